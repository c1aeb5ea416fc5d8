#!/bin/bash
# L2 / vector-L1 counters of one kernel: bash tools/pmc_l2.sh <filter> <cmd...>
# (the TA_* counter set is left out: that pass did not finish within 13 minutes on the pool)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_l2
rm -rf $OUT; mkdir -p $OUT
FILT=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p$i -o p -- "$@" > $OUT/p$i.log 2>&1
done
python - $OUT "$FILT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if sys.argv[2] not in k: continue
    print(k)
    for c, v in sorted(d.items()): print("   %-34s %16.0f per launch" % (c, v / cnt[(k, c)]))
PY
grep -h -i "error\|invalid\|not found" $OUT/p*.log | head -5
find $OUT -name "*.csv" -delete
