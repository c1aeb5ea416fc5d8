timeout 900 python -m pytest tests/test_gpu_fold.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -i "passed\|failed\|error" | tail -3
for cfg in C3_cars C2_match_tensor; do
for S in 1 4; do
  out=$(timeout 300 python bench.py --config $cfg --sub none --streams $S --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$cfg S=$S $(echo "$out" | python -c 'import sys,json; r=json.loads(sys.stdin.read()); k=r["roofline"]["kernels_us_per_step"]; print(r["value"], r["ms_per_step"], r["config"]["ms_per_step_one_batch_in_flight"], {a:b for a,b in k.items() if "lstm16" in a})' 2>&1 | tail -1)"
done; done
