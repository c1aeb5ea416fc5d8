#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cars_session.py tests/test_gpu_envelope.py -q -m gpu -k "tail_over or cars" > gpurun_out/t11.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/t11.log
for cfg in "4 4 16" "4 8 32" "8 4 32" "2 4 16" "1 4 16"; do
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
for W in 1 2 4; do
BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=$W timeout 600 python bench.py --sub none --no-cpu-baseline --steps 256 > gpurun_out/emu_kg.log 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/emu_kg.log").read().strip().splitlines()[-1]); print("C3 W=$W default", d["value"], d["ms_per_step"])
PY
done
for W in 1 2 4 8; do
BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=$W timeout 600 python bench.py --config C5_cars_bf16 --sub none --no-cpu-baseline --steps 64 > gpurun_out/emu_kg.log 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/emu_kg.log").read().strip().splitlines()[-1]); print("C5 W=$W default", d["value"], d["ms_per_step"])
PY
done
