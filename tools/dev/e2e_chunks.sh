set -u
R=$GRAFT_REPO_ROOT
cd $R
python - <<'PY'
import os, sys, shutil
sys.path.insert(0, os.getcwd())
from oracle import lldo
from opensmile_amd import synth
d = "/dev/shm/e2ec"; shutil.rmtree(d, ignore_errors=True); os.makedirs(d + "/in"); os.makedirs(d + "/out")
u = [synth.utterance(2 + i, 160000) for i in range(32)]
for i in range(32): lldo.write_wav(f"{d}/in/u{i:05d}.wav", u[i])
for i in range(32, 8000): shutil.copyfile(f"{d}/in/u{i % 32:05d}.wav", f"{d}/in/u{i:05d}.wav")
open(d + "/list.txt", "w").write("\n".join(f"{d}/in/u{i:05d}.wav" for i in range(8000)) + "\n")
PY
export LD_LIBRARY_PATH=$R/opensmile_amd:${LD_LIBRARY_PATH:-}
for cf in ${E2E_CHUNKS:-256 512 1024 2048}; do
  for serial in 0 1; do
    best=9
    for rep in 1 2 3; do
      s=$(date +%s%N)
      if [ $serial = 1 ]; then SMILEHIP_E2E_SERIAL=1 SMILEHIP_TIMING=1 ./opensmile_amd/smilextract_hip --set mfcc12_0_d_a -filelist /dev/shm/e2ec/list.txt -outdir /dev/shm/e2ec/out -O 1 --chunk-files $cf 2> /tmp/t.err
      else SMILEHIP_TIMING=1 ./opensmile_amd/smilextract_hip --set mfcc12_0_d_a -filelist /dev/shm/e2ec/list.txt -outdir /dev/shm/e2ec/out -O 1 --chunk-files $cf 2> /tmp/t.err; fi
      e=$(date +%s%N)
      ms=$(( (e - s) / 1000000 ))
      echo "chunk $cf serial $serial wall_ms $ms $(grep -o "since the first ingest.*" /tmp/t.err | cut -c1-560)"
    done
  done
done
rm -rf /dev/shm/e2ec
