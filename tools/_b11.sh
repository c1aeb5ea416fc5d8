cd $GRAFT_REPO_ROOT
for v in 2000 100000 1000000; do
python bench.py --config C3_cars --vocab $v --sub none --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us_per_step']; print('C3 vocab $v', d['value'], d['ms_per_step'], [ (n,v) for n,v in k.items() if 'lstm16' in n])"
python bench.py --config C3_cars --vocab $v --uniform --sub none --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us_per_step']; print('C3 uniform vocab $v', d['value'], d['ms_per_step'], [ (n,v) for n,v in k.items() if 'lstm16' in n])"
done
