"""Condense rocprofv3 counter_collection CSVs: mean counter value per dispatch, per kernel.
usage: python tools/pmc_summary.py dir_or_csv [...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("nir::", "")
    if name.startswith("void "):
        name = name[5:]
    return name.split("(")[0][:64]


def main(paths):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    meta = {}
    for p in paths:
        files = glob.glob(os.path.join(p, "*counter_collection.csv")) if os.path.isdir(p) else [p]
        for f in files:
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                a = acc[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
                meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
    for k in sorted(acc, key=lambda k: -sum(v[0] for v in acc[k].values())):
        g, wg, lds, vg, sg = meta[k]
        print("%s   [grid=%s wg=%s lds=%s vgpr=%s sgpr=%s]" % (k, g, wg, lds, vg, sg))
        for c in sorted(acc[k]):
            tot, n = acc[k][c]
            print("    %-30s %18.1f   (mean of %d dispatches)" % (c, tot / n, n))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
