"""Condense gpurun_out/r02prof (tools/capture_r02.sh) into per-config summaries: <cfg>/kernel_stats.csv (rocprofv3 --stats),
<cfg>/pmc.json (mean counter per launch per kernel) and traffic.json ((2*FETCH_SIZE + WRITE_SIZE) KiB per launch -- the x2 is the
gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md section HBM).  usage: python tools/derive_r02.py <dir>; run
`python tools/derive_r02.py --install <dir>` in the authoring container to copy the summaries into profiles/."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def short(k):
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*$", "", k)
    return k.replace("nir::", "").replace(", ", ",")


def counters(d):
    """mean counter value per launch, per kernel; a kernel launched with several grids in one step (e.g. the query and the
    document recurrence) is reported for its LARGEST grid only -- that is the launch bench.py's roofline prices."""
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(lambda: [0.0, 0])))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            grid = int(r.get("Grid_Size", 0) or 0)
            a = acc[short(r["Kernel_Name"])][grid][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    out = {}
    for k, grids in acc.items():
        g = max(grids)
        out[k] = {c: v[0] / v[1] for c, v in grids[g].items()}
        out[k]["grid_size"] = g
    return out


def condense(root):
    traffic = {}
    for cdir in sorted(glob.glob(os.path.join(root, "*", ""))):
        cfg = os.path.basename(os.path.dirname(cdir))
        st = glob.glob(os.path.join(cdir, "stats", "**", "*kernel_stats.csv"), recursive=True)
        if st:
            shutil.copy(st[0], os.path.join(cdir, "kernel_stats.csv"))
        pmc = {}
        for p in ("fetch", "write", "sq"):
            for k, d in counters(os.path.join(cdir, p)).items():
                pmc.setdefault(k, {}).update({c: round(v, 1) for c, v in d.items()})
        if pmc:
            json.dump(pmc, open(os.path.join(cdir, "pmc.json"), "w"), indent=1, sort_keys=True)
        t = {}
        for k, d in pmc.items():
            if "FETCH_SIZE" in d:
                t[k] = {"FETCH_SIZE_KB": d["FETCH_SIZE"], "WRITE_SIZE_KB": d.get("WRITE_SIZE", 0.0),
                        "bytes_per_launch": int((2 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)) * 1024)}
        traffic[cfg] = t
    json.dump({"note": "(2*FETCH_SIZE + WRITE_SIZE) KiB per launch, mean over the launches of the serial eager bench.py run of each config; "
                       "separate --pmc passes (tools/capture_r02.sh); kernels launched with several shapes in one step are averaged over all of them",
               "configs": traffic}, open(os.path.join(root, "traffic.json"), "w"), indent=1, sort_keys=True)


def install(root):
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    for cdir in sorted(glob.glob(os.path.join(root, "*", ""))):
        cfg = os.path.basename(os.path.dirname(cdir))
        for f, name in (("kernel_stats.csv", "r02_%s_kernel_stats.csv"), ("pmc.json", "r02_%s_pmc.json")):
            if os.path.exists(os.path.join(cdir, f)):
                shutil.copy(os.path.join(cdir, f), os.path.join(dst, name % cfg))
    shutil.copy(os.path.join(root, "traffic.json"), os.path.join(dst, "traffic.json"))


if __name__ == "__main__":
    if sys.argv[1] == "--install":
        install(sys.argv[2])
    else:
        condense(sys.argv[1])
