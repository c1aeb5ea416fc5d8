#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_cars_session.py tests/test_gpu_hardening.py -q -m gpu > gpurun_out/t7.log 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/t7.log
timeout 600 python tools/decode_profile.py > gpurun_out/decode_profile.log 2>&1; echo "decode profile rc=$?"; cat gpurun_out/decode_profile.log | head -12
timeout 900 python - <<'PY'
import json, os, sys, time
sys.argv = ["bench.py", "--sub", "none"]
sys.path.insert(0, os.getcwd())
import bench, torch
args = bench.parse(); env = bench.Env()
head = dict(bench.CONFIGS["C3_cars"])
print("decode", json.dumps(bench.decode_record(head, args, env)))
torch.cuda.empty_cache()
print("train cars", json.dumps(bench.train_record("CARS", dict(head), args, env)))
print("train mt", json.dumps(bench.train_record("MATCH_TENSOR", dict(bench.CONFIGS["C2_match_tensor"]), args, env)))
PY
for L in 4 8; do
BENCH_SHARD_LANES=$L BENCH_NO_H2D=1 BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=8 timeout 600 python bench.py --config C5_cars_bf16 --sub none --no-cpu-baseline --steps 48 --nbatches 16 > gpurun_out/emu_c5_w8_l$L.log 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/emu_c5_w8_l$L.log").read().strip().splitlines()[-1]); print("C5 W=8 lanes $L", d["value"], d["ms_per_step"], d["config"]["batches_in_flight"])
PY
done
