#!/bin/bash
# fused DUET: parity tests, then C4 DUET bench fused vs unfused
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "duet or c4_full" 2>&1 | tail -3
python bench.py --config C4_duet --sub none --steps 30 --warmup 5 2>&1 | tail -1 | python -c "
import sys, json
l = json.loads(sys.stdin.readline())
print('FUSED', l['value'], l['ms_per_step'], json.dumps(l['roofline'].get('kernels_us_per_step')))"
