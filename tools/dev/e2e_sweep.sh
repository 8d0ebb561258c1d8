#!/bin/bash
# usage (GPU box): tools/dev/e2e_sweep.sh <files>  -- smilextract_hip --set mfcc12_0_d_a on <files> x 10 s WAVs in /dev/shm, stage timing
# for a few thread counts / chunk sizes (development aid for the file-to-file route)
N=${1:-8000}
D=/dev/shm/e2e_sweep; rm -rf $D; mkdir -p $D/in $D/out
python - $N $D <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from opensmile_amd import synth
from oracle import lldo
n, d = int(sys.argv[1]), sys.argv[2]
u = [synth.utterance(2 + i, 160000) for i in range(32)]
with open(d + "/list.txt", "w") as f:
    for i in range(n):
        p = f"{d}/in/u{i:05d}.wav"; lldo.write_wav(p, u[i % 32]); f.write(p + "\n")
PY
export LD_LIBRARY_PATH=$PWD/opensmile_amd:$LD_LIBRARY_PATH SMILEHIP_TIMING=1
X="opensmile_amd/smilextract_hip --set mfcc12_0_d_a -filelist $D/list.txt -outdir $D/out -O 1"
$X 2>/dev/null
for t in 8 16 32; do for c in 128 256 256 512; do
  t0=$(date +%s%N); o=$(SMILEHIP_IO_THREADS=$t $X --chunk-files $c 2>&1 | sed 's/smilextract_hip timing: //' | tr '\n' ' '); t1=$(date +%s%N)
  echo "threads $t chunk $c: wall $(( (t1 - t0) / 1000000 )) ms; $o"
done; done
rm -rf $D
