cd $GRAFT_REPO_ROOT
for s in 2 3 4 6 8; do
GPU_MAX_HW_QUEUES=8 python bench.py --config C3_cars --streams $s --sub none --no-cpu-baseline --steps 300 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C3 streams $s', d['value'], d['ms_per_step'])"
done
for s in 4 6 8; do
GPU_MAX_HW_QUEUES=8 python bench.py --config C2_match_tensor --streams $s --sub none --no-cpu-baseline --steps 600 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C2 streams $s', d['value'], d['ms_per_step'])"
done
