#!/bin/bash
for v in 100000 64; do
python bench.py --config C4_duet --sub none --steps 30 --warmup 5 --vocab $v 2>&1 | tail -1 | python -c "
import sys, json
l = json.loads(sys.stdin.readline())
k = l['roofline'].get('kernels_us_per_step')
print('VOCAB $v', l['value'], l['ms_per_step'], {a: b for a, b in k.items() if 'doc_kernel' in a})"
done
