#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_bench_flow.py -q -m gpu > gpurun_out/t4.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/t4.log
timeout 900 python bench.py --sub C5_cars_bf16 > gpurun_out/bench_h2d.log 2>gpurun_out/bench_h2d.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_h2d.log").read().strip().splitlines()[-1])
c=d["config"]
print("headline", d["value"], "h2d", c["pairs_per_s_with_host_ids_h2d"], c["h2d_inclusive_over_resident"], c["h2d_stream"])
print("C5 stream", json.dumps(c["sub"].get("C5_stream"))[:1500])
PY
tail -5 gpurun_out/bench_h2d.err
timeout 900 python bench.py --config C5_stream > gpurun_out/bench_c5stream.log 2>gpurun_out/bench_c5stream.err; echo "c5stream rc=$?"
tail -c 1800 gpurun_out/bench_c5stream.log; tail -3 gpurun_out/bench_c5stream.err
for W in 1 8; do
 for K in 4 8; do
  BENCH_KSTEP=$K BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=$W timeout 600 python bench.py --sub none --no-cpu-baseline --steps 64 > gpurun_out/emu_c3_w${W}_k$K.log 2>gpurun_out/emu_c3_w${W}_k$K.err; echo "emu C3 W=$W K=$K rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/emu_c3_w${W}_k$K.log").read().strip().splitlines()[-1])
    print("C3 W=$W K=$K", d["value"], d["ms_per_step"])
except Exception as e: print("parse fail", e)
PY
 done
done
for W in 1 8; do
  BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=$W timeout 600 python bench.py --config C5_cars_bf16 --sub none --no-cpu-baseline --steps 32 > gpurun_out/emu_c5_w$W.log 2>gpurun_out/emu_c5_w$W.err; echo "emu C5 W=$W rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/emu_c5_w$W.log").read().strip().splitlines()[-1])
    print("C5 W=$W", d["value"], d["ms_per_step"])
except Exception as e: print("parse fail", e)
PY
done
# staged (eager collectives) for comparison
BENCH_NO_COLL_CAPTURE=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=8 timeout 600 python bench.py --sub none --no-cpu-baseline --steps 64 > gpurun_out/emu_c3_w8_staged.log 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/emu_c3_w8_staged.log").read().strip().splitlines()[-1]); print("C3 W=8 staged", d["value"], d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"])
PY
