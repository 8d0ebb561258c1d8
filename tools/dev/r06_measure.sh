#!/bin/bash
# usage (GPU box, via gpurun): tools/dev/r06_measure.sh <tag> <section> [<section> ...]   -> gpurun_out/<tag>/
# One script for the round's evidence (replaces the r04_run*.sh one-offs). Sections:
#   tests      the whole GPU suite (+ smoke)              tests:<pytest args>   a subset, e.g. tests:tests/test_gpu_is09.py
#   bench      the default bench line (configs 2-5, PCIe figures, CPU baselines) + the driver's short form
#   benchc:N   bench.py --config N --no-cpu-baseline (N = 2..5)
#   stats:N    rocprofv3 --kernel-trace --stats of bench.py --config N, default and SMILEHIP_SERIAL=1 (each kernel alone)
#   pmc:N      SQ / LDS / GRBM counter passes + FETCH_SIZE / WRITE_SIZE (each pass its own run) of config N at a reduced batch
#              -> pmc_c<N>.txt (tools/pmc_summary.py), r06_pmc_c<N>.json (tools/pmc_counters_json.py)
#   traffic    tools/pmc_traffic_configs.sh (HBM bytes per kernel of configs 3-5, per frame)
#   e2e        tools/bench_e2e.py (file-to-file route)        plugin   tools/bench_plugin.py (60 s and 600 s files)
#   smileapi   tools/bench_smileapi.py (10 min pushed in 1 s pieces through cExternalAudioSource)
#   sweep      tools/plugin_config_sweep.py (the 40 runnable shipped files through the plugin)
#   pmc64:N    only the FP64 / INT64 instruction-class pass of config N (-> pmc64_c<N>.txt)
#   pmcx:N     counter sets of the caller's choice, PMCX_SETS="A B C;D E" (one pass per ';' part) -> pmcx_c<N>.txt
#   sh:<cmd>   any command (quoted), output to sh_<n>.log
set -u
R=$GRAFT_REPO_ROOT
T=$1; shift
O=$R/gpurun_out/$T
mkdir -p $O
declare -A UTTS=( [2]=1000 [3]=2000 [4]=1500 [5]=6000 )
[ "${PMC_FULL:-0}" = 1 ] && UTTS=( [2]=1000 [3]=10000 [4]=12500 [5]=125000 )     # the bench's own batch sizes
declare -A STEPS=( [2]=3 [3]=2 [4]=1 [5]=1 )
n=0
for sec in "$@"; do
  cd $R
  case $sec in
    tests)
      ( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
      grep -n "^FAILED\|^ERROR\|passed\|failed\|pytest rc" $O/pytest_gpu.log | cut -c1-300 | head -12
      timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log ;;
    tests:*)
      ( time timeout 1200 python -m pytest ${sec#tests:} -m gpu -q -x ) > $O/pytest_sub_$n.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sub_$n.log
      grep -n "^FAILED\|^ERROR\|^E  \|passed\|failed\|pytest rc" $O/pytest_sub_$n.log | cut -c1-400 | head -20 ;;
    bench)
      ( time timeout 1500 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
      cut -c1-300 $O/bench_default.json
      timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-h2d > $O/bench_c2_short.json 2>> $O/bench_default.err
      cut -c1-300 $O/bench_c2_short.json ;;
    benchc:*)
      c=${sec#benchc:}
      if [ $c = 2 ]; then X="--no-configs --no-h2d"; else X="--config $c"; fi
      timeout 900 python bench.py $X --no-cpu-baseline > $O/bench_c$c.json 2> $O/bench_c$c.err; tail -2 $O/bench_c$c.err
      python - $O/bench_c$c.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "max_abs_err", "cells_bit_identical", "cells_checked", "frame_scaled_err")}, d["roofline"].get("kernel_ms"), d["roofline"].get("frac"))
except Exception as e:
    print("no line:", e)
PY
      ;;
    stats:*)
      c=${sec#stats:}
      if [ $c = 2 ]; then X="--no-configs --no-h2d"; else X="--config $c"; fi
      cd /tmp && export TMPDIR=/tmp
      for mode in default serial; do
        if [ $mode = serial ]; then export SMILEHIP_SERIAL=1; else unset SMILEHIP_SERIAL; fi
        [ $c = 2 ] && [ $mode = serial ] && continue
        rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c$c -- python $R/bench.py $X --no-cpu-baseline > $O/stats_c${c}_$mode.json 2> $O/stats_c$c.log
        cp $(find $O/stats_c$c -name '*kernel_stats.csv' | head -1) $O/c${c}_${mode}_kernel_stats.csv; rm -rf $O/stats_c$c
        cut -d, -f1-4,7 $O/c${c}_${mode}_kernel_stats.csv | cut -c1-160 | head -12
      done
      unset SMILEHIP_SERIAL ;;
    pmc:*)
      c=${sec#pmc:}
      if [ $c = 2 ]; then X="--no-configs --no-h2d"; else X="--config $c"; fi
      BENCH="python $R/bench.py $X --utts ${UTTS[$c]} --steps ${STEPS[$c]} --warmup 1 --no-cpu-baseline"
      cd /tmp && export TMPDIR=/tmp
      i=0
      for set in \
       "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
       "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" \
       "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" \
       "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_IFETCH" \
       "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64" \
       "FETCH_SIZE" "WRITE_SIZE" ; do
        i=$((i+1))
        timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_c$c/p$i -- $BENCH > $O/pmc_c${c}_p$i.log 2>&1
      done
      python $R/tools/pmc_summary.py $O/pmc_c$c $O/pmc_c$c.txt > /dev/null 2>&1
      rm -rf $O/pmc_c$c
      python $R/tools/pmc_counters_json.py $O/pmc_c$c.txt > $O/r06_pmc_c$c.json 2>/dev/null
      python - $O/r06_pmc_c$c.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:12]:
    print("%-46s %8.2f ms  valu %.2f lds %.2f wait %.2f hbm %.3f" % (k[:46], v["ms"], v["valu_busy_frac"], v["lds_pipe_frac"], v["wait_inst_frac"] or 0, v["hbm_frac"] or 0))
PY
      ;;
    pmcx:*)                                              # pmcx:N  counter sets of the caller's choice: PMCX_SETS="A B C;D E" (one pass per ';' part) -> pmcx_c<N>.txt
      c=${sec#pmcx:}
      if [ $c = 2 ]; then X="--no-configs --no-h2d"; else X="--config $c"; fi
      cd /tmp && export TMPDIR=/tmp
      i=0
      IFS=';' read -ra SETS <<< "${PMCX_SETS:-SQ_WAVES SQ_WAVE_CYCLES}"
      for set in "${SETS[@]}"; do
        i=$((i+1))
        timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcx_c$c/p$i -- python $R/bench.py $X --utts ${UTTS[$c]} --steps ${STEPS[$c]} --warmup 1 --no-cpu-baseline > $O/pmcx_c${c}_p$i.log 2>&1
      done
      python $R/tools/pmc_summary.py $O/pmcx_c$c $O/pmcx_c$c.txt > /dev/null 2>&1
      rm -rf $O/pmcx_c$c
      grep -c "^==" $O/pmcx_c$c.txt ;;
    pmc64:*)
      c=${sec#pmc64:}
      if [ $c = 2 ]; then X="--no-configs --no-h2d"; else X="--config $c"; fi
      cd /tmp && export TMPDIR=/tmp
      timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU --output-format csv -d $O/pmc64_c$c/p1 -- python $R/bench.py $X --utts ${UTTS[$c]} --steps ${STEPS[$c]} --warmup 1 --no-cpu-baseline > $O/pmc64_c${c}.log 2>&1
      python $R/tools/pmc_summary.py $O/pmc64_c$c $O/pmc64_c$c.txt > /dev/null 2>&1
      rm -rf $O/pmc64_c$c
      grep -c "^==" $O/pmc64_c$c.txt ;;
    traffic) bash tools/pmc_traffic_configs.sh $T/traffic > $O/traffic.log 2>&1; cp $O/traffic/pmc_traffic_c*.json $O/ 2>/dev/null; tail -3 $O/traffic.log | cut -c1-300 ;;
    e2e) timeout 900 python tools/bench_e2e.py > $O/e2e.jsonl 2> $O/e2e.err; cut -c1-400 $O/e2e.jsonl; tail -3 $O/e2e.err ;;
    plugin) ( timeout 900 python tools/bench_plugin.py --seconds 60 --skip-per-component; echo '{"note": "ten-minute files (one hour for MFCC12_0_D_A is 6 x this): CPU binary, the block-per-tick path, the default (fused batch fed from the wave level), the fused source component"}'; timeout 1200 python tools/bench_plugin.py --seconds 600 --no-per-component ) > $O/plugin_throughput.jsonl 2> $O/plugin.err; cut -c1-260 $O/plugin_throughput.jsonl | grep -v fused_source; tail -3 $O/plugin.err ;;
    smileapi) timeout 1200 python tools/bench_smileapi.py --seconds 600 > $O/smileapi_throughput.jsonl 2> $O/smileapi.err; cut -c1-300 $O/smileapi_throughput.jsonl; tail -3 $O/smileapi.err ;;
    sweep) timeout 1200 python tools/plugin_config_sweep.py > $O/plugin_config_sweep.jsonl 2> $O/sweep.err; grep -c identical $O/plugin_config_sweep.jsonl ;;
    sh:*) n=$((n+1)); ( eval "${sec#sh:}" ) > $O/sh_$n.log 2>&1; tail -15 $O/sh_$n.log | cut -c1-300 ;;
    *) echo "unknown section $sec" ;;
  esac
  n=$((n+1))
done
ls $O | head -60
