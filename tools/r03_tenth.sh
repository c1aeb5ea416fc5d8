#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_bench_flow.py -q -m gpu > gpurun_out/t10.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/t10.log
for CFG in C3_cars C5_cars_bf16; do
for W in 1 2 4 8; do
  steps=160; [ $CFG = C5_cars_bf16 ] && steps=48
  BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=$W timeout 600 python bench.py --config $CFG --sub none --no-cpu-baseline --steps $steps > gpurun_out/emu_${CFG}_w$W.log 2>gpurun_out/emu_${CFG}_w$W.err; echo "emu $CFG W=$W rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/emu_${CFG}_w$W.log").read().strip().splitlines()[-1])
    print("$CFG W=$W", d["value"], d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"], d["config"]["parallelism"][:60])
except Exception as e: print("parse fail", e)
PY
done
done
