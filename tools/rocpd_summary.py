"""Summarise a rocprofv3 rocpd (.db) capture: per-kernel call count / avg / total duration, and, when the capture
has PMC samples, per-kernel mean counter values per dispatch.   usage: python tools/rocpd_summary.py file.db [...]"""
import sqlite3
import sys


def short(name):
    name = name.replace("nir::", "")
    if name.startswith("void "):
        name = name[5:]
    return name.split("(")[0][:70]


def main(paths):
    for path in paths:
        c = sqlite3.connect(path)
        print("== %s" % path)
        rows = c.execute("select name, count(*), avg(end-start), sum(end-start), min(end-start), max(end-start) "
                         "from kernels group by name order by 4 desc").fetchall()
        tot = sum(r[3] for r in rows) or 1
        print("%-72s %7s %10s %10s %10s %6s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
        for n, cnt, avg, sm, mn, mx in rows[:25]:
            print("%-72s %7d %10.2f %10.2f %10.2f %6.1f" % (short(n), cnt, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * sm / tot))
        try:
            pm = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                           "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        except sqlite3.Error:
            pm = []
        if pm:
            print("-- mean counter value per dispatch")
            cur = None
            for kn, cn, v, cnt in pm:
                if kn != cur:
                    cur = kn
                    print("  %s" % short(kn))
                print("      %-28s %16.1f   (n=%d)" % (cn, v, cnt))
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
