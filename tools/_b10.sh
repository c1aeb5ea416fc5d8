cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "duet or c4_full" 2>&1 | tail -3
python bench.py --config C4_duet --sub none --no-cpu-baseline --steps 12 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('DUET', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline'].get('achieved'), d['roofline'].get('frac')); print(d['roofline']['kernels_us_per_step'])"
