"""Condense gpurun_out/prof/ (tools/capture_profiles.sh) into profiles/: kernel stats CSVs, PMC summary, traffic.json.
usage: python tools/derive_traffic.py [round_tag]      (default r01)"""
import csv
import glob
import io
import json
import os
import shutil
import sys
from collections import defaultdict
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_summary  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    if not f:
        raise SystemExit("missing " + pattern)
    return f[0]


shutil.copy(one("stats_serial/**/*kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(one("stats_default/**/*kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats_4_in_flight.csv"))
shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, tag + "_bench_line.json"))
buf = io.StringIO()
with redirect_stdout(buf):
    pmc_summary.main([os.path.dirname(one("pmc_%s/**/*counter_collection.csv" % s)) for s in ("fetch", "write", "sq", "mfma")])
open(os.path.join(dst, tag + "_pmc_summary.txt"), "w").write(buf.getvalue())


def mean_counter(pass_name, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(one("pmc_%s/**/*counter_collection.csv" % pass_name))):
        if r["Counter_Name"] == counter:
            a = acc[pmc_summary.short(r["Kernel_Name"]).replace(", ", ",")]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


fetch, write = mean_counter("fetch", "FETCH_SIZE"), mean_counter("write", "WRITE_SIZE")
kern = {}
for k in fetch:
    if k.startswith("lstm_mfma_kernel") or k.startswith("mt_"):
        kern[k] = {"FETCH_SIZE_KB": round(fetch[k], 1), "WRITE_SIZE_KB": round(write.get(k, 0.0), 1),
                   "bytes_per_launch": int((2 * fetch[k] + write.get(k, 0.0)) * 1024),
                   "note": "(2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes, %s capture" % tag}
json.dump({"workload": "bench.py default (match_tensor 32x10, q4, d64), one batch in flight, eager launches", "kernels": kern},
          open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(kern, indent=1))
