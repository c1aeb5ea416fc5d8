timeout 1200 python -m pytest tests/test_gpu_cars_session.py -x -q -m gpu 2>&1 | grep -i "passed\|failed\|error\|assert" | tail -3
timeout 600 python -m pytest tests -x -q -m gpu -k "decode or golden or cars" 2>&1 | grep -i "passed\|failed\|error" | tail -3
for cfg in C5_cars_bf16 C3_cars; do
for S in 1 4; do
  st=400; case $cfg in C5_cars_bf16) st=40;; esac
  out=$(timeout 300 python bench.py --config $cfg --sub none --streams $S --steps $st --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$cfg S=$S $(echo "$out" | python -c 'import sys,json; r=json.loads(sys.stdin.read()); k=r["roofline"]["kernels_us_per_step"]; print(r["value"], r["ms_per_step"], {a:b for a,b in k.items() if "lstm_step" in a or "gemm" in a})' 2>&1 | tail -1)"
done; done
