"""One training step out of a rocprofv3 --kernel-trace database of tools/train_profile.py (rocpd sqlite): kernels per step, per-name totals and,
with --order, the launch sequence of one step (short names, microseconds).  python tools/train_step_trace.py gpurun_out/trainprof/tp_results.db"""
import re
import sqlite3
import sys


def short(n):
    n = n.replace(".kd", "")
    if n.startswith("_ZN3nir"):
        m = re.match(r"_ZN3nir(\d+)", n)
        return "NIR:" + n[m.end():m.end() + int(m.group(1))]
    if n.startswith("void nir::") or n.startswith("nir::"):
        return "NIR:" + re.sub(r"^(void )?nir::", "", n).split("(")[0].split("<")[0]
    for key in ("MulFunctor", "CUDAFunctor_add", "FillFunctor", "sum_functor", "CatArrayBatchedCopy", "direct_copy", "masked_fill", "FusedAdam",
                "SoftMaxBackward", "SoftMaxForward", "softmax_warp_backward", "softmax_warp_forward", "gather", "MaxOps", "NormTwo", "LpNorm", "CompareEq",
                "CompareFunctor", "bitwise_not", "arange", "exp_kernel", "neg_kernel", "DivFunctor", "MeanOps", "radixSort", "AbsFunctor", "sign",
                "triu_tril", "reciprocal", "clamp", "BitwiseOr", "MaxNan", "remainder", "multi_tensor_apply", "fillBuffer", "copyBuffer", "index"):
        if key in n:
            return "T:" + key
    return "T:" + n[:40]


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = db.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
    idx = [i for i, r in enumerate(rows) if "FusedAdam" in r[0]]
    per = 2 if len(idx) > 2 and idx[1] - idx[0] < 5 else 1                     # fused Adam = 1-2 launches per step
    a, b = idx[-2 * per - 1] + 1, idx[-per - 1] + 1
    step = rows[a:b]
    tot = {}
    for n, s, e in step:
        k = short(n)
        c, u = tot.get(k, (0, 0.0))
        tot[k] = (c + 1, u + (e - s) / 1000.0)
    print("%d kernels, kernel-sum %.0f us, span %.0f us" % (len(step), sum(v[1] for v in tot.values()), (step[-1][2] - step[0][1]) / 1000.0))
    nir = sum(v[0] for k, v in tot.items() if k.startswith("NIR:"))
    print("library kernels %d (%.0f us), tensor glue %d (%.0f us)" % (nir, sum(v[1] for k, v in tot.items() if k.startswith("NIR:")), len(step) - nir,
                                                                      sum(v[1] for k, v in tot.items() if not k.startswith("NIR:"))))
    for k, (c, u) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("%-36s %4d %9.1f" % (k, c, u))
    if "--order" in sys.argv:                     # start offset, duration, gap to the latest end so far (idle device time in front of the kernel)
        t0, prev = step[0][1], step[0][1]
        for n, s, e in step:
            print("  %9.1f  %-34s %7.1f  gap %6.1f" % ((s - t0) / 1000.0, short(n), (e - s) / 1000.0, (s - prev) / 1000.0))
            prev = max(prev, e)


if __name__ == "__main__":
    main()
