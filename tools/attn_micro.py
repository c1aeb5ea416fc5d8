"""Micro-harness for the fused attention pooling of the CARS document encoder (csrc/cars_attn.hip) at a chosen shape: per-kernel time of
encode_document from the library profiler and, with a timing build
    NIR_VARIANT=aptime NIR_VARIANT_FLAGS=-DAP_TIMING python -m context_attentive_ir_amd.build      (then --lib aptime)
the phase stamps of workgroup 7, tile 5 of attn_pool_pipe_kernel: MMA waves (k-loop, tanh/row-dot epilogue, barrier wait) and IO waves
(softmax + weighted sum of tile k-2, staging of tile k, load issue of tile k+1, barrier wait), shader-clock cycles.

    python tools/attn_micro.py [--M 4480 --T 64 --iters 20 --lib aptime --fp32-rows 0|1 --dtype f32|bf16]
"""
import os
os.environ.setdefault("NIR_DEBUG_TUNABLES", "1")
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=4480)
    ap.add_argument("--T", type=int, default=64)
    ap.add_argument("--V", type=int, default=20000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--lib", default="")
    ap.add_argument("--fp32-rows", type=int, default=0)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--io-prio", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from context_attentive_ir_amd import lib
    if a.lib:
        lib.LIB_PATH = os.path.join(ROOT, "context_attentive_ir_amd", "libneuroir_hip_%s.so" % a.lib)
    from helpers import build_model
    L = lib.load()
    m = build_model("CARS", vocab=a.V, device="cuda")
    m.compute_dtype = a.dtype
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(4, a.V, (1, 1, a.M, a.T), generator=g).cuda()
    lens = torch.full((1, 1, a.M), a.T, dtype=torch.int64).cuda()
    def timed(rows, prio):
        L.nir_debug_set_tunable(b"attn_fp32_rows", rows)
        L.nir_debug_set_tunable(b"attn_io_prio", prio)
        for _ in range(3):
            m.encode_document(ids, lens)
        torch.cuda.synchronize()
        L.nir_profile_enable(1)
        for _ in range(a.iters):
            m.encode_document(ids, lens)
        torch.cuda.synchronize()
        L.nir_profile_enable(0)
        buf = C.create_string_buffer(1 << 16)
        L.nir_profile_report(buf, len(buf))
        out = {}
        for line in buf.value.decode().strip().splitlines():
            k, cnt, ms = line.rsplit(",", 2)
            out[k.split("[")[0]] = float(ms) / int(cnt) * 1e3
        return out

    # variants interleaved in ONE process (box-to-box and run-to-run differences are ~5 %, larger than the effects looked for)
    variants = [(0, 0), (1, 0), (0, 1), (1, 1)] if a.dtype == "f32" else [(0, 0), (0, 1)]
    acc = {v: [] for v in variants}
    for rep in range(a.reps):
        for v in variants:
            acc[v].append(timed(*v))
    for v in variants:
        att = sorted(next(x for k, x in r.items() if k.startswith("attn_pool")) for r in acc[v])
        rec = sorted(next(x for k, x in r.items() if k.startswith("lstm16")) for r in acc[v])
        print("fp32_rows %d io_prio %d : attention median %.1f us (min %.1f max %.1f)   recurrence median %.1f us" % (
            v[0], v[1], att[len(att) // 2], att[0], att[-1], rec[len(rec) // 2]))
    L.nir_debug_set_tunable(b"attn_fp32_rows", a.fp32_rows)
    L.nir_debug_set_tunable(b"attn_io_prio", a.io_prio)
    m.encode_document(ids, lens)
    torch.cuda.synchronize()
    if a.lib:
        raw = C.CDLL(lib.LIB_PATH)
        if hasattr(raw, "nir_debug_attn_timing"):
            out = (C.c_longlong * 16)()
            raw.nir_debug_attn_timing(out)
            t = list(out)
            print("stamps", t[:8])
            print("MMA waves : k-loop %d  epilogue %d  barrier wait %d   (iteration %d cycles)" % (t[1] - t[0], t[2] - t[1], t[3] - t[2], t[3] - t[0]))
            print("IO waves  : softmax of tile k-2 %d, its weighted sum + store %d" % (t[8] - t[4], t[5] - t[8]))
            print("IO waves  : finish tile k-2 %d  stage tile k %d  issue loads k+1 %d   (busy %d cycles, starts %+d after the MMA waves)" % (
                t[5] - t[4], t[6] - t[5], t[7] - t[6], t[7] - t[4], t[4] - t[0]))


if __name__ == "__main__":
    main()
