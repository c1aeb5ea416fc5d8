#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for L in 6 8 12; do
  BENCH_SHARD_LANES=$L GPU_MAX_HW_QUEUES=16 BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=8 timeout 600 python bench.py --sub none --no-cpu-baseline --steps 240 --nbatches 24 > gpurun_out/emu_lanes_$L.log 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/emu_lanes_$L.log").read().strip().splitlines()[-1]); print("C3 W=8 pair lanes $L q16", d["value"], d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"])
PY
  BENCH_SHARD_LANES=$L BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=8 timeout 600 python bench.py --sub none --no-cpu-baseline --steps 240 --nbatches 24 > gpurun_out/emu_lanes_$L.log 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/emu_lanes_$L.log").read().strip().splitlines()[-1]); print("C3 W=8 pair lanes $L q8", d["value"], d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"])
PY
done
