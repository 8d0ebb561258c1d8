cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 300 python bench.py > gpurun_out/r03b/bench_c2.json 2> gpurun_out/r03b/c2.err
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03b/bench_c2_short.json 2>> gpurun_out/r03b/c2.err
for c in 3 4 5; do timeout 500 python bench.py --config $c > gpurun_out/r03b/bench_c$c.json 2> gpurun_out/r03b/c$c.err; done
timeout 300 python tools/bench_sets.py --sets mfcc,plp,is09,compare,f0,compare_full --utts 1000 --steps 10 2>/dev/null | grep set > gpurun_out/r03b/chains.jsonl
timeout 300 python tools/bench_sets.py --sets egemaps --utts 10000 --seconds 3 --steps 5 --func 2>/dev/null | grep set >> gpurun_out/r03b/chains.jsonl
for f in gpurun_out/r03b/bench_c*.json; do echo $f; cut -c1-330 $f; done
cat gpurun_out/r03b/chains.jsonl | cut -c1-200
