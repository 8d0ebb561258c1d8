R=$GRAFT_REPO_ROOT; cd $R
for c in 4 5; do for x in 0 1; do
  SMILEHIP_JITTER_XD=$x timeout 900 python bench.py --config $c --no-cpu-baseline > /tmp/o.json 2>/tmp/o.err || tail -2 /tmp/o.err
  python - $c $x <<'PY'
import json,sys
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
k=d['roofline']['kernels_ms_per_step']
print('config',sys.argv[1],'xd',sys.argv[2],'ms_per_step',round(d['ms_per_step'],2),'bits',d.get('accuracy',{}).get('cells_bit_identical'),'of',d.get('accuracy',{}).get('cells_checked'),'jitter',[v for a,v in k.items() if 'jitter_runs' in a])
PY
done; done
