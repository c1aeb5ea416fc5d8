"""Condense gpurun_out/<round>prof (tools/capture_profiles.sh) into per-config summaries: <cfg>/kernel_stats.csv (rocprofv3 --stats),
<cfg>/pmc.json (mean counter per launch per kernel) and pmc_summary.json: per config and kernel
  bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB  (the x2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md section HBM),
  mfma_busy        = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)   (matrix-pipe busy fraction, chip-wide, while
                     the kernel runs; GRBM_GUI_ACTIVE is summed over the 8 XCDs).
usage: python tools/derive_profiles.py <dir>; `python tools/derive_profiles.py --install <dir> [prefix]` in the authoring container copies the
summaries into profiles/ (prefix default r03); `--check <dir> <bench_detail.json>` verifies that every record's dominant kernel is in its capture."""
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
    k = k.replace("nir::", "").replace(", ", ",")
    return re.sub(r"(,false)+>$", ">", k)       # defaulted trailing template arguments: the library's profile label omits them (<4,4,8,false,false> = <4,4,8>)


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
        for p in ("fetch", "write", "sq", "stall"):
            for k, d in counters(os.path.join(cdir, p)).items():
                pmc.setdefault(k, {}).update({c: round(v, 1) for c, v in d.items()})
        if pmc:
            json.dump(pmc, open(os.path.join(cdir, "pmc.json"), "w"), indent=1, sort_keys=True)
        t = {}
        for k, d in pmc.items():
            if "FETCH_SIZE" in d:
                t[k] = {"FETCH_SIZE_KB": d["FETCH_SIZE"], "WRITE_SIZE_KB": d.get("WRITE_SIZE", 0.0),
                        "bytes_per_launch": int((2 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)) * 1024)}
                if d.get("GRBM_GUI_ACTIVE"):
                    t[k]["mfma_busy"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * d["GRBM_GUI_ACTIVE"] / 8.0), 5)
                    t[k]["gui_active_cycles_per_xcd"] = round(d["GRBM_GUI_ACTIVE"] / 8.0, 1)
                    t[k]["SQ_INSTS_MFMA"] = d.get("SQ_INSTS_MFMA")
        traffic[cfg] = t
    json.dump({"note": "bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8); mean over the "
                       "launches of the serial eager bench.py run of each config; separate --pmc passes (tools/capture_profiles.sh); a kernel launched with "
                       "several grids in one step is reported for its largest grid",
               "configs": traffic}, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1, sort_keys=True)


def install(root, prefix="r03"):
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    for cdir in sorted(glob.glob(os.path.join(root, "*", ""))):
        cfg = os.path.basename(os.path.dirname(cdir))
        for f, name in (("kernel_stats.csv", "%s_%s_kernel_stats.csv"), ("pmc.json", "%s_%s_pmc.json")):
            if os.path.exists(os.path.join(cdir, f)):
                shutil.copy(os.path.join(cdir, f), os.path.join(dst, name % (prefix, cfg)))
    new = json.load(open(os.path.join(root, "pmc_summary.json")))
    old_path = os.path.join(dst, "pmc_summary.json")
    if os.path.exists(old_path):          # a partial re-capture replaces only the configs it holds
        old = json.load(open(old_path))
        old["configs"].update(new["configs"])
        new["configs"] = old["configs"]
    json.dump(new, open(old_path, "w"), indent=1, sort_keys=True)


def check(root, detail_path):
    """every record of the bench run (headline + sub-records with a roofline) must find its dominant kernel -- template arguments included -- in
    the kernel_stats.csv captured for that config: a profile of another instantiation proves nothing about the timed one.  Exit code 1 otherwise."""
    detail = json.load(open(detail_path))
    recs = {detail["headline"]["name"]: detail["headline"]}
    recs.update(detail.get("sub", {}))
    bad = []
    for cfg, rec in sorted(recs.items()):
        st = os.path.join(root, cfg, "kernel_stats.csv")
        kern = ((rec.get("roofline") or {}).get("kernel") or "").split("[")[0]
        if not os.path.exists(st) or not kern:
            continue
        names = {short(r["Name"]) for r in csv.DictReader(open(st))}
        fam = kern.replace("[gather]", "").replace("[fused]", "")
        ok = fam in names or ("<" not in fam and any(n.split("<")[0] == fam for n in names))
        print("%-22s %-40s %s" % (cfg, kern, "ok" if ok else "MISSING from the capture"))
        if not ok:
            bad.append(cfg)
    return 1 if bad else 0


if __name__ == "__main__":
    if sys.argv[1] == "--install":
        install(sys.argv[2], *(sys.argv[3:4]))
    elif sys.argv[1] == "--check":
        sys.exit(check(sys.argv[2], sys.argv[3]))
    else:
        condense(sys.argv[1])
