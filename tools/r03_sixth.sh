#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/decode_profile.py > gpurun_out/decode_profile.log 2>&1; echo "decode profile rc=$?"; cat gpurun_out/decode_profile.log | tail -25
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/gputests.log 2>&1; echo "gpu rc=$?"
tail -6 gpurun_out/gputests.log
timeout 1500 python bench.py > gpurun_out/bench_default.log 2>gpurun_out/bench_default.err; echo "bench rc=$?"
tail -3 gpurun_out/bench_default.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_default.log").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["config"]["pairs_per_s_with_host_ids_h2d"])
for e in d["sub"]: print(e["name"], e["pairs_per_s"], e["ms_per_step"], e["kernel"], e["frac"], e.get("error"))
for k in ("C3_cars_with_decode","train_C3_cars_update","train_C2_match_tensor_update"): print(k, json.dumps(d["config"]["sub"].get(k))[:900])
PY
