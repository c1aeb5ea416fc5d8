"""One steady-state iteration of a serial loop from a rocprofv3 kernel trace: every kernel between two consecutive launches of a marker kernel
(default: flag_publish_kernel, the last kernel of a predict()), with start offset, duration, gap to the previous kernel's end, grid.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $REPO/tools/dropin_loop.py --modes default --batch 16 --decode 0
    python tools/iter_timeline.py $OUT [--marker flag_publish] [--which -3]
"""
import argparse
import csv
import glob
import os


def short(name):
    name = name.replace("void ", "").replace("nir::", "")
    return name.split("(")[0][:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--marker", default="flag_publish")
    ap.add_argument("--which", type=int, default=-3, help="which marker-to-marker interval (negative: from the end)")
    a = ap.parse_args()
    rows = []
    for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"),
                         int(r.get("Grid_Size_X", 0) or 0) // max(1, int(r.get("Workgroup_Size_X", 1) or 1)), int(r.get("Workgroup_Size_X", 1) or 1)))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    i0, i1 = marks[a.which - 1], marks[a.which]
    seg = rows[i0 + 1:i1 + 1]
    t0 = rows[i0][1]
    print("interval: %.1f us from the previous marker's end to this marker's end; %d kernels, sum of durations %.1f us" % (
        (seg[-1][1] - t0) / 1e3, len(seg), sum(e - s for s, e, *_ in seg) / 1e3))
    prev_end = t0
    for s, e, n, q, g, w in seg:
        print("%9.1f  dur %8.1f  gap %7.1f  q%-3s wg %6d x %4d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, g, w, n))
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main()
