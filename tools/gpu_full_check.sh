#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/gputests.log 2>&1; echo "gpu rc=$?"
tail -6 gpurun_out/gputests.log
timeout 1500 python bench.py > gpurun_out/bench_default.log 2>gpurun_out/bench_default.err; echo "bench rc=$?"
tail -3 gpurun_out/bench_default.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_default.log").read().strip().splitlines()[-1])
c=d["config"]
print("headline", d["value"], d["ms_per_step"], c["ms_per_step_one_batch_in_flight"], c["pairs_per_s_with_host_ids_h2d"], c["h2d_inclusive_over_resident"], c["macro_batch"])
print({k:v for k,v in d["roofline"].items() if k!="kernels_us_per_step" and not k.startswith("sub.")})
for e in d["sub"]: print(e["name"], e["pairs_per_s"], e["ms_per_step"], e["kernel"], e["frac"], e.get("error"))
PY
python __graft_entry__.py --smoke 2>&1 | tail -6
