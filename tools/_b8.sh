cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -x -q -k "match_tensor or ranker or fuzz_esm or vocabulary or graphed or batches_in_flight" 2>&1 | tail -3
python bench.py --config C2_match_tensor --sub NS_match_tensor_50 --no-cpu-baseline --steps 400 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C2', d['value'], d['ms_per_step'], d['config']['ms_per_step_one_batch_in_flight'], d['roofline']['kernel'], d['roofline'].get('frac')); print(d['roofline']['kernels_us_per_step'])
for k,v in d['config']['sub'].items(): print(k, v.get('pairs_per_s'), v.get('ms_per_step'), v.get('ms_per_step_one_batch_in_flight'), (v.get('roofline') or {}).get('kernels_us_per_step'))"
