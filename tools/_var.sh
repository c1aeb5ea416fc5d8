timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linear" 2>&1 | grep -i "passed\|failed\|error\|assert" | tail -8
for S in 1 4; do
  out=$(timeout 300 python bench.py --config C3_cars --sub none --streams $S --steps 400 --warmup 40 --cpu-seconds 3 2>/dev/null | tail -1)
  echo "C3 S=$S $(echo "$out" | python -c 'import sys,json; r=json.loads(sys.stdin.read()); k=r["roofline"]["kernels_us_per_step"]; print(r["value"], r["ms_per_step"], r["cpu_baseline"]["max_abs_diff_vs_gpu_softmax"], {a:b for a,b in k.items() if "gemm" in a})' 2>&1 | tail -1)"
done
