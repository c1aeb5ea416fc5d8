#!/bin/bash
# round-2 "before" numbers: GPU test suite + one bench line per config (serial, eager, per-kernel stats from the library's own events)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02_base
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for cfg in "cars --batch 16 --cands 10" "cars --batch 64 --cands 50 --steps 20" "duet --batch 64 --cands 50 --dlen 290 --qlen 8 --steps 20" "drmm --batch 64 --cands 50 --dlen 290 --uniform --vocab 1000000 --steps 100" "esm --batch 64 --cands 50 --dlen 290 --uniform --vocab 1000000 --steps 100" "match_tensor --batch 32 --cands 50"; do
  name=$(echo $cfg | tr ' ' '_' | tr -d '-')
  timeout 600 python bench.py --model $cfg --no-cpu-baseline 2>$OUT/$name.err | tail -1 > $OUT/$name.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json"))
    print("$name", d["value"], d["ms_per_step"], d["config"]["ms_per_step_one_batch_in_flight"], d["roofline"]["kernels_us_per_step"])
except Exception as e:
    print("$name FAILED", e)
PY
done
