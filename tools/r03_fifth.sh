#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_hardening.py tests/test_gpu_bench_flow.py -q -m gpu -s > gpurun_out/t5.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|DRMM_OVERLAP_REPORT|^FAILED|Error" gpurun_out/t5.log | head -30
for W in 1 2 4 8; do
  BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=$W timeout 600 python bench.py --sub none --no-cpu-baseline --steps 160 > gpurun_out/emu_c3_w$W.log 2>gpurun_out/emu_c3_w$W.err; echo "emu C3 W=$W rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/emu_c3_w$W.log").read().strip().splitlines()[-1])
    print("C3 W=$W", d["value"], d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"], d["config"]["batches_in_flight"])
except Exception as e: print("parse fail", e)
PY
  tail -2 gpurun_out/emu_c3_w$W.err
done
for W in 1 2 4 8; do
  BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=$W timeout 600 python bench.py --config C5_cars_bf16 --sub none --no-cpu-baseline --steps 48 > gpurun_out/emu_c5_w$W.log 2>gpurun_out/emu_c5_w$W.err; echo "emu C5 W=$W rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/emu_c5_w$W.log").read().strip().splitlines()[-1])
    print("C5 W=$W", d["value"], d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"], d["config"]["batches_in_flight"])
except Exception as e: print("parse fail", e)
PY
done
for L in 4 16; do
BENCH_SHARD_LANES=$L GPU_MAX_HW_QUEUES=16 BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=8 timeout 600 python bench.py --sub none --no-cpu-baseline --steps 160 --nbatches 16 > gpurun_out/emu_c3_w8_l$L.log 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/emu_c3_w8_l$L.log").read().strip().splitlines()[-1]); print("C3 W=8 lanes $L", d["value"], d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"])
PY
done
