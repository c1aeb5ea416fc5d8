#!/bin/bash
# rocprofv3 per-kernel durations of one bench configuration, serial eager launches (every launch attributed):
#   gpurun -- 'bash tools/prof_stats.sh <tag> <bench.py args...>'   -> gpurun_out/prof_<tag>/ + condensed CSV printed
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $REPO/bench.py "$@" --streams 1 --no-graph --steps 30 --warmup 5 --no-cpu-baseline > $OUT/run.log 2>&1
f=$(find $OUT -name "s_kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total_us_all_launches", round(tot / 1e3, 1))
for r in rows[:25]:
    print("%-90s calls=%6s avg_us=%9.2f pct=%5.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
