"""Occupancy timeline of the TIMED bench run (hipGraph replays on 4 lanes) from a rocprofv3 kernel trace:

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $REPO/bench.py --sub none --no-cpu-baseline --steps 200
    python tools/timeline.py $OUT [--tail-ms 15]

Over the last --tail-ms of the trace (steady state: replays only): wall time, time with NO kernel resident, time with exactly one / two / more,
per kernel: launches, mean duration, how much of its time it ran beside another kernel, and per queue the share of the wall it had a kernel
running.  Answers "do the lanes overlap, and what is the chip doing when they do not".
"""
import argparse
import csv
import glob
import os
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("nir::", "")
    return name.split("(")[0][:44]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--tail-ms", type=float, default=15.0)
    ap.add_argument("--skip-ms", type=float, default=1.0, help="drop this much before the last kernel (teardown)")
    ap.add_argument("--scan", type=float, default=0.0, help="print launches / queues per bin of this many ms and exit")
    ap.add_argument("--at-ms", type=float, default=None, help="window start, ms after the first kernel of the trace")
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"),
                         int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1),
                         int(r.get("Workgroup_Size_X", 1) or 1) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)))
    rows.sort()
    base = rows[0][0]
    if a.scan:
        bins = defaultdict(lambda: [0, set(), 0])
        for s, e, n, q, g, w in rows:
            b = (s - base) // int(a.scan * 1e6)
            bins[b][0] += 1; bins[b][1].add(q); bins[b][2] += e - s
        for b in sorted(bins):
            print("%8.1f ms: %5d launches, %d queues, kernel time / bin %.2f" % (b * a.scan, bins[b][0], len(bins[b][1]), bins[b][2] / (a.scan * 1e6)))
        return
    if a.at_ms is not None:
        t0 = base + int(a.at_ms * 1e6)
        t1 = t0 + int(a.tail_ms * 1e6)
    else:
        # default: the last stretch in which launches arrive on several queues (the graphed, multi-lane timed region)
        multi = [e for s, e, n, q, g, w in rows if n.startswith("lstm16") and g // max(w, 1) >= 200]
        t1 = max(multi) - int(a.skip_ms * 1e6)
        t0 = t1 - int(a.tail_ms * 1e6)
    win = [(max(s, t0), min(e, t1), n, q, g, w) for s, e, n, q, g, w in rows if e > t0 and s < t1]
    ev = []
    for s, e, n, q, g, w in win:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth_t = defaultdict(int)
    d, prev = 0, t0
    for t, k in ev:
        depth_t[d] += t - prev
        prev = t
        d += k
    depth_t[d] += t1 - prev
    wall = t1 - t0
    print("window %.1f ms, %d kernel launches, %d queues" % (wall / 1e6, len(win), len({q for *_, q, _, _ in win})))
    for k in sorted(depth_t):
        print("  %d kernel(s) resident: %5.1f %%" % (k, 100.0 * depth_t[k] / wall))
    # per kernel: time, overlap share (time during which some other kernel was also running)
    iv = sorted((s, e) for s, e, *_ in win)
    per = defaultdict(lambda: [0, 0, 0, 0])
    import bisect
    starts = [s for s, e in iv]
    for s, e, n, q, g, w in win:
        ov = 0
        # overlap with others: sum of pairwise overlaps, capped at own duration
        i = bisect.bisect_left(starts, s - 2_000_000)
        for s2, e2 in iv[i:]:
            if s2 >= e:
                break
            if (s2, e2) == (s, e):
                continue
            ov += max(0, min(e, e2) - max(s, s2))
        p = per[n]
        p[0] += 1; p[1] += e - s; p[2] += min(ov, e - s); p[3] = max(p[3], g // max(w, 1))
    tot = sum(p[1] for p in per.values())
    print("sum of kernel time / wall = %.2f" % (tot / wall))
    print("%-46s %6s %9s %8s %8s %7s" % ("kernel", "n", "mean us", "% wall", "beside%", "wgs"))
    for n, p in sorted(per.items(), key=lambda kv: -kv[1][1])[:24]:
        print("%-46s %6d %9.1f %8.1f %8.0f %7d" % (n, p[0], p[1] / p[0] / 1e3, 100.0 * p[1] / wall, 100.0 * p[2] / max(p[1], 1), p[3]))
    perq = defaultdict(int)
    for s, e, n, q, g, w in win:
        perq[q] += e - s
    print("per queue busy share:", {q: round(v / wall, 2) for q, v in sorted(perq.items())})
    # gaps inside one queue between consecutive kernels (the dependent-launch gap the lanes see)
    byq = defaultdict(list)
    for s, e, n, q, g, w in win:
        byq[q].append((s, e))
    gaps = []
    for q, l in byq.items():
        l.sort()
        for (s0, e0), (s1, e1) in zip(l, l[1:]):
            gaps.append(max(0, s1 - e0))
    gaps.sort()
    if gaps:
        print("in-queue gap between consecutive kernels: median %.1f us, p90 %.1f us, mean %.1f us, sum/wall/queue %.2f" % (
            gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, sum(gaps) / len(gaps) / 1e3, sum(gaps) / wall / max(len(byq), 1)))


if __name__ == "__main__":
    main()
