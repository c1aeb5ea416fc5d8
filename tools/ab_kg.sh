# alternate macro-batch policies of the driver's 20-step region on ONE box: bash tools/ab_kg.sh
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for kg in 8 7 10 5; do
    BENCH_MACRO_BATCH=$kg timeout 200 python bench.py --steps 20 --warmup 5 --sub none --no-cpu-baseline > /tmp/b.log 2>&1
    python - <<PY
import json
l=[x for x in open("/tmp/b.log") if x.startswith("{")][-1]
d=json.loads(l); print("kg=$kg rep=$rep", d["value"], d["ms_per_step"], d["config"]["macro_batch"], d["config"]["lanes"], d["power"]["package_w"])
PY
  done
done
