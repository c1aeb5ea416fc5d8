#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "4 4 16" "4 8 32" "8 4 32" "8 8 64" "2 4 16"; do
  set -- $cfg
  BENCH_GATHER_EVERY=$1 BENCH_SHARD_LANES=$2 BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=8 timeout 600 python bench.py --sub none --no-cpu-baseline --steps 256 --nbatches $3 > gpurun_out/emu_kg.log 2>gpurun_out/emu_kg.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/emu_kg.log").read().strip().splitlines()[-1]); print("C3 W=8 KG=$1 lanes=$2", d["value"], d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"])
except Exception as e:
    print("fail $cfg", e); print(open("gpurun_out/emu_kg.err").read()[-800:])
PY
done
BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=1 timeout 600 python bench.py --sub none --no-cpu-baseline --steps 256 > gpurun_out/emu_kg.log 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/emu_kg.log").read().strip().splitlines()[-1]); print("C3 W=1 default", d["value"], d["ms_per_step"])
PY
for W in 8; do
BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=$W timeout 600 python bench.py --config C5_cars_bf16 --sub none --no-cpu-baseline --steps 64 > gpurun_out/emu_kg.log 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/emu_kg.log").read().strip().splitlines()[-1]); print("C5 W=$W default", d["value"], d["ms_per_step"])
PY
done
