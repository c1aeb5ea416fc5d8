#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_cars_session.py tests/test_gpu_parity.py tests/test_gpu_envelope.py tests/test_gpu_fold.py -q -m gpu -k "cars or CARS or decode or session" > gpurun_out/t8.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/t8.log
timeout 600 python tools/decode_profile.py > gpurun_out/decode_profile.log 2>&1; echo "decode profile rc=$?"; cat gpurun_out/decode_profile.log | head -14
BENCH_NO_H2D=1 timeout 900 python bench.py --sub C5_cars_bf16 --no-cpu-baseline > gpurun_out/bench_c3.log 2>gpurun_out/bench_c3.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_c3.log").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["config"]["ms_per_step_one_batch_in_flight"])
print(json.dumps(d["roofline"]["kernels_us_per_step"]))
for e in d["sub"]: print(e["name"], e["pairs_per_s"], e["ms_per_step"], e["kernel"], e["frac"], e.get("error"))
print(json.dumps(d["config"]["sub"]["C3_cars_with_decode"]))
PY
