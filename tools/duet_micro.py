"""Micro-harness for the fused DUET document kernel (csrc/duet_fused.hip) at the C4 shape: ms per forward over many launches and, with a
timing build (NIR_VARIANT=dft NIR_VARIANT_FLAGS=-DDF_TIMING python -m context_attentive_ir_amd.build; --lib dft), the phase clocks of
workgroup 3000: prologue, GEMM 1, pooling epilogue, GEMM 2, fc2 epilogue (shader-clock cycles).

    python tools/duet_micro.py [--B 64 --N 50 --DL 290 --iters 20 --lib dft --no-planes]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--N", type=int, default=50)
    ap.add_argument("--QL", type=int, default=6)
    ap.add_argument("--DL", type=int, default=290)
    ap.add_argument("--V", type=int, default=100000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--lib", default="")
    ap.add_argument("--no-planes", action="store_true")
    a = ap.parse_args()
    from context_attentive_ir_amd import lib
    if a.lib:
        lib.LIB_PATH = os.path.join(ROOT, "context_attentive_ir_amd", "libneuroir_hip_%s.so" % a.lib)
    from helpers import build_model
    m = build_model("DUET", vocab=a.V, device="cuda", max_query_len=a.QL, max_doc_len=a.DL)
    m.table_planes = not a.no_planes
    r = np.random.default_rng(5)
    z = lambda *s: torch.from_numpy(np.minimum(r.zipf(1.2, size=s), a.V - 1).astype("int64")).cuda()
    q, d = z(a.B, a.QL), z(a.B, a.N, a.DL)
    ql = torch.full((a.B,), a.QL, dtype=torch.int64, device="cuda")
    dl = torch.full((a.B, a.N), a.DL, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            s = m(q, ql, d, dl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            s = m(q, ql, d, dl)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print("forward %.3f ms  -> %.3f M pairs/s   (planes=%s, checksum %.6f)" % (ms, a.B * a.N / ms / 1e3, not a.no_planes, float(s.sum())))
    L = lib.load()
    if hasattr(L, "nir_debug_duet_timing"):
        buf = (C.c_longlong * 16)()
        L.nir_debug_duet_timing.argtypes = [C.c_void_p]
        L.nir_debug_duet_timing(buf)
        t = list(buf)[:6]
        names = ["prologue", "GEMM 1", "pool epilogue", "GEMM 2", "fc2 epilogue"]
        print("phase cycles (workgroup 3000, shader clock):", {n: t[i + 1] - t[i] for i, n in enumerate(names)}, "total", t[5] - t[0])


if __name__ == "__main__":
    main()
