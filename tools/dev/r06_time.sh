set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_time; mkdir -p $O
cd $R
python - ${1:-600} <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from oracle import lldo
from opensmile_amd import synth
lldo.write_wav("/tmp/in.wav", synth.utterance(5, int(float(sys.argv[1]) * 16000)))
PY
export LD_LIBRARY_PATH=$R/opensmile_amd:$R/oracle/_ref:${LD_LIBRARY_PATH:-}
cd $R/opensmile_amd/plugin
for c in "mfcc/MFCC12_0_D_A.conf -O" "is09-13/IS09_emotion.conf -lldhtkoutput"; do
  set -- $c
  rm -f $O/trace.txt
  s=$(date +%s%N)
  SMILEHIP_PLUGIN_FUSE=0 SMILEHIP_PLUGIN_TIMING=1 SMILEHIP_PLUGIN_TRACE=$O/trace.txt $R/oracle/_ref/SMILExtract -C $R/oracle/_ref/config/$1 -I /tmp/in.wav $2 /tmp/o.htk -l 0 2> /dev/null
  e=$(date +%s%N)
  echo "$1 ms=$(( (e - s) / 1000000 )) $(grep '^block\|^time' $O/trace.txt | tr '\n' ' ')"
done
