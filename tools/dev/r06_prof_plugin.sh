# usage (GPU box): bash tools/dev/r06_prof_plugin.sh <conf relative to config/> <output option> [seconds]   -- rocprofv3 kernel table of the plugin inside SMILExtract
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_prof; mkdir -p $O
conf=$1; opt=$2; SEC=${3:-60}
cd $R
python - $SEC <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from oracle import lldo
from opensmile_amd import synth
lldo.write_wav("/tmp/in.wav", synth.utterance(5, int(float(sys.argv[1]) * 16000)))
PY
export LD_LIBRARY_PATH=$R/opensmile_amd:$R/oracle/_ref:${LD_LIBRARY_PATH:-}
export TMPDIR=/tmp
cd $R/opensmile_amd/plugin
tag=$(basename $conf .conf)
SMILEHIP_PLUGIN_TRACE=$O/trace_$tag.txt rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$tag -- $R/oracle/_ref/SMILExtract -C $R/oracle/_ref/config/$conf -I /tmp/in.wav $opt /tmp/o.htk -l 1 > $O/run_$tag.log 2>&1
cp $(find $O/p_$tag -name '*kernel_stats.csv' | head -1) $O/${tag}_kernel_stats.csv; rm -rf $O/p_$tag
cut -d, -f1-4,7 $O/${tag}_kernel_stats.csv | cut -c1-150 | head -25
grep -v "\.cpu\| 0$" $O/trace_$tag.txt | tr '\n' ' '
