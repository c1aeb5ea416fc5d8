cd $GRAFT_REPO_ROOT
NIR_PROFILE_SHAPES=1 python bench.py --model cars --batch 16 --cands 10 --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['ms_per_step_one_batch_in_flight'])
for k,v in sorted(d['roofline']['kernels_us_per_step'].items(), key=lambda kv:-kv[1]): print('%8.2f  %s'%(v,k))"
