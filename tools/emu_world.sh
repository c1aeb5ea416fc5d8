export BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for cfg in C3_cars C2_match_tensor; do
for W in 1 2 4 8; do
for S in 2 4 8; do
  out=$(BENCH_EMULATE_WORLD=$W timeout 300 python bench.py --config $cfg --sub none --streams $S --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$cfg W=$W S=$S $(echo "$out" | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"], r["config"].get("ms_per_step_one_batch_in_flight"), r["config"].get("host_enqueue_ms_per_step"))' 2>&1 | tail -1)"
done; done; done
