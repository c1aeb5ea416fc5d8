#!/bin/bash
# Per-rank step time of an N = W run on ONE GPU: BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=W gives this process rank 0's 1/W
# share of the sharded CARS step over a real 1-rank RCCL group (W-shard gather buffers; scores meaningless; cross-GPU collective latency NOT
# included).  usage (GPU box, repo root):  bash tools/emu_world.sh [config ...]      knobs: BENCH_SHARD_AXIS=auto|candidate|pair,
# BENCH_GATHER_EVERY=<steps merged per graph and all-gather, pair axis>, BENCH_SHARD_LANES=<lanes>
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
mkdir -p gpurun_out
for CFG in ${@:-C3_cars C5_cars_bf16}; do
  steps=256; [ "$CFG" = C5_cars_bf16 ] && steps=64
  for W in 1 2 4 8; do
    BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=$W timeout 600 python bench.py --config $CFG --sub none --no-cpu-baseline --steps $steps \
      > gpurun_out/emu_${CFG}_w$W.log 2> gpurun_out/emu_${CFG}_w$W.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/emu_${CFG}_w$W.log").read().strip().splitlines()[-1])
    print("$CFG W=$W  %.1f pairs/s  %.5f ms/step  host %.5f ms/step  %s" % (d["value"], d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"], d["config"]["parallelism"][:90]))
except Exception as e:
    print("$CFG W=$W failed:", e)
PY
  done
done
