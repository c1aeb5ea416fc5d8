export BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for it in 1 2 3 4 5 6 7 8; do
for W in 1 8; do
  BENCH_EMULATE_WORLD=$W timeout 300 python bench.py --config C3_cars --sub none --streams $((4 + 4*(it%2))) --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/e_${W}_$it.out 2> gpurun_out/e_${W}_$it.err
  echo "W=$W it=$it rc=$? $(cut -c1-100 gpurun_out/e_${W}_$it.out)"
done; done
grep -l "Error\|error" gpurun_out/e_*_*.err
