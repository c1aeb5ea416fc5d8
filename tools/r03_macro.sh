#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_bench_flow.py -q -m gpu > gpurun_out/t12.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t12.log
for MB in 1 2 4 8; do
BENCH_MACRO_BATCH=$MB timeout 900 python bench.py --sub none --no-cpu-baseline > gpurun_out/bench_mb$MB.log 2>gpurun_out/bench_mb$MB.err; echo "bench MB=$MB rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_mb$MB.log").read().strip().splitlines()[-1]); c=d["config"]
print("MB=$MB headline", d["value"], d["ms_per_step"], "1-in-flight", c["ms_per_step_one_batch_in_flight"], "h2d", c["pairs_per_s_with_host_ids_h2d"], c["h2d_inclusive_over_resident"], "diff", c["overlapped_vs_serial_max_abs_diff"], c["batches_in_flight"])
PY
tail -2 gpurun_out/bench_mb$MB.err
done
for SM in 1 2 4; do
BENCH_STREAM_MACRO=$SM timeout 900 python bench.py --config C5_stream > gpurun_out/bench_c5s.log 2>gpurun_out/bench_c5s.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_c5s.log").read().strip().splitlines()[-1]); c=d["config"]
print("C5 stream macro $SM", d["value"], c.get("seconds"), c.get("batches"), c.get("left_over_batches_not_timed"), c.get("error"))
PY
done
