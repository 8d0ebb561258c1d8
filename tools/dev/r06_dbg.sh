# usage (GPU box): bash tools/dev/r06_dbg.sh [seconds]   -- the plugin's block-per-tick mode on one file per big set: wall time, block counters
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_dbg; mkdir -p $O
SEC=${1:-60}
cd $R
python - $SEC <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from oracle import lldo
from opensmile_amd import synth
lldo.write_wav("/tmp/in.wav", synth.utterance(5, int(float(sys.argv[1]) * 16000)))
PY
export LD_LIBRARY_PATH=$R/opensmile_amd:$R/oracle/_ref:${LD_LIBRARY_PATH:-}
cd $R/opensmile_amd/plugin
for c in "mfcc/MFCC12_0_D_A.conf -O" "is09-13/IS09_emotion.conf -lldhtkoutput" "compare16/ComParE_2016.conf -lldhtkoutput" "egemaps/v02/eGeMAPSv02.conf -lldhtkoutput"; do
  set -- $c
  conf=$1; opt=$2
  rm -f $O/trace.txt
  s=$(date +%s%N)
  SMILEHIP_PLUGIN_COMPONENTS=none $R/oracle/_ref/SMILExtract -C $R/oracle/_ref/config/$conf -I /tmp/in.wav $opt /tmp/o_cpu.htk -l 0 2> /dev/null
  e=$(date +%s%N)
  SMILEHIP_PLUGIN_FUSE=0 SMILEHIP_PLUGIN_DEBUG=${DBG:-0} SMILEHIP_PLUGIN_TRACE=$O/trace.txt $R/oracle/_ref/SMILExtract -C $R/oracle/_ref/config/$conf -I /tmp/in.wav $opt /tmp/o.htk -l 1 2> $O/dbg_$(basename $conf).log
  rc=$?
  f=$(date +%s%N)
  echo "$conf rc=$rc cpu_ms=$(( (e - s) / 1000000 )) block_ms=$(( (f - e) / 1000000 )) same=$(cmp -s /tmp/o.htk /tmp/o_cpu.htk && echo yes || echo NO) $(grep '^block\|^cFramer' $O/trace.txt | tr '\n' ' ')"
done
