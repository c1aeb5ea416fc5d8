cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fold.py tests/test_gpu_cars_session.py tests/test_gpu_parity.py -m gpu -x -q -k "fold or cars or duet" 2>&1 | tail -3
python bench.py --sub C4_duet,C5_cars_bf16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('HEAD', d['value'], d['ms_per_step'], d['config']['ms_per_step_one_batch_in_flight'], d['roofline']['kernel'], d['roofline'].get('frac')); print(d['roofline']['kernels_us_per_step'])
for k,v in d['config']['sub'].items(): print(k, v.get('pairs_per_s'), v.get('ms_per_step'), (v.get('roofline') or {}).get('kernels_us_per_step'))"
