#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_bench_flow.py "tests/test_gpu_train.py::test_drmm_trainable_embeddings_vs_reference" -q -m gpu > gpurun_out/t3.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/t3.log
timeout 1200 python bench.py > gpurun_out/bench_default.log 2>gpurun_out/bench_default.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/bench_default.log; tail -5 gpurun_out/bench_default.err
for W in 1 2 4 8; do
  BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=$W timeout 600 python bench.py --sub none --no-cpu-baseline --steps 64 > gpurun_out/emu_c3_w$W.log 2>gpurun_out/emu_c3_w$W.err; echo "emu C3 W=$W rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/emu_c3_w$W.log").read().strip().splitlines()[-1])
    print("C3 W=$W", d["value"], d["ms_per_step"], d["config"]["parallelism"][:200])
except Exception as e: print("parse fail", e)
PY
  tail -2 gpurun_out/emu_c3_w$W.err
done
for W in 1 8; do
  BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=$W timeout 600 python bench.py --config C5_cars_bf16 --sub none --no-cpu-baseline --steps 32 > gpurun_out/emu_c5_w$W.log 2>gpurun_out/emu_c5_w$W.err; echo "emu C5 W=$W rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/emu_c5_w$W.log").read().strip().splitlines()[-1])
    print("C5 W=$W", d["value"], d["ms_per_step"], d["config"]["parallelism"][:200])
except Exception as e: print("parse fail", e)
PY
  tail -2 gpurun_out/emu_c5_w$W.err
done
