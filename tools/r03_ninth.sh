#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fold.py -q -m gpu > gpurun_out/t9.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/t9.log
for S in 1 2; do
NIR_TUNE=lstm_s=$S BENCH_NO_H2D=1 timeout 900 python bench.py --sub none --no-cpu-baseline > gpurun_out/bench_c3_s$S.log 2>gpurun_out/bench_c3_s$S.err; echo "bench lstm_s=$S rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_c3_s$S.log").read().strip().splitlines()[-1])
print("lstm_s=$S headline", d["value"], d["ms_per_step"], d["config"]["ms_per_step_one_batch_in_flight"])
print(json.dumps(d["roofline"]["kernels_us_per_step"]))
PY
done
