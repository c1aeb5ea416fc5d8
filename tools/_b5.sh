cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r02_default.json 2> gpurun_out/bench_r02_default.err; echo rc=$?
tail -3 gpurun_out/bench_r02_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_r02_default.json").read().strip().splitlines()[-1])
print("HEAD", d["value"], d["ms_per_step"], d["config"]["ms_per_step_one_batch_in_flight"], d["roofline"]["kernel"], d["roofline"].get("frac"), d["cpu_baseline"])
for k,v in d["config"]["sub"].items():
    if "error" in v: print(k, "ERROR", v["error"]); continue
    r=v.get("roofline") or {}
    print(k, v.get("pairs_per_s"), v.get("ms_per_step"), v.get("ms_per_step_one_batch_in_flight"), r.get("kernel"), r.get("avg_us"), r.get("bound"), r.get("achieved"), r.get("frac"))
PY
