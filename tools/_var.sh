timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linear" 2>&1 | grep -i "passed\|failed\|error\|assert" | tail -8
for v in 0 1; do
  out=$(NIR_TUNE=lstm_var=$v timeout 300 python bench.py --config C3_cars --sub none --streams 4 --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "C3 v=$v $(echo "$out" | python -c 'import sys,json; r=json.loads(sys.stdin.read()); k=r["roofline"]["kernels_us_per_step"]; print(r["value"], r["ms_per_step"], {a:b for a,b in k.items() if "gemm" in a})' 2>&1 | tail -1)"
done
