cd $GRAFT_REPO_ROOT
for cfg in "--batch 16 --cands 10" "--batch 16 --cands 10 --dtype bf16" "--batch 64 --cands 50 --steps 30" "--batch 64 --cands 50 --steps 30 --dtype bf16"; do
  python bench.py --model cars $cfg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['config']['ms_per_step_one_batch_in_flight'], d['roofline']['kernels_us_per_step'])"
done
