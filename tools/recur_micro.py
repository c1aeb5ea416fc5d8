"""Micro-harness for the folded BiLSTM recurrence (nir_bilstm_folded_fwd) at a chosen shape: launch time over many launches and, with a
trace build (NIR_VARIANT=trace NIR_VARIANT_FLAGS=-DNIR_PT_TRACE python -m context_attentive_ir_amd.build; --lib trace), the per-wave phase
clocks of workgroup (0, 0): matrix phase (LDS reads + MFMAs until the last result is back), gate phase (gate math, LDS writes, store),
barrier wait -- shader-clock cycles per step.

    python tools/recur_micro.py [--M 4480 --T 64 --H 128 --iters 50 --lib trace --dtype f32]
"""
import os
os.environ.setdefault("NIR_DEBUG_TUNABLES", "1")
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=4480)
    ap.add_argument("--T", type=int, default=64)
    ap.add_argument("--H", type=int, default=128)
    ap.add_argument("--V", type=int, default=20000)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--lib", default="")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--lens", default="full", help="full | ragged (uniform in [1, T])")
    ap.add_argument("--ids", default="uniform", help="uniform | zipf | hot (64 distinct rows: the row gathers hit in L2)")
    ap.add_argument("--tune", default="", help="name=value,... through nir_debug_set_tunable (lstm_w16=3: two sequence groups per workgroup)")
    a = ap.parse_args()
    name = "libneuroir_hip%s.so" % ("_" + a.lib if a.lib else "")
    L = C.CDLL(os.path.join(ROOT, "context_attentive_ir_amd", name))
    vp, i64 = C.c_void_p, C.c_int64
    L.nir_bilstm_folded_fwd.restype = C.c_int
    L.nir_bilstm_folded_fwd.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, i64, i64, C.c_int, C.c_int, C.c_int, vp]
    L.nir_debug_set_buffer.argtypes = [vp]
    for kv in filter(None, a.tune.split(",")):
        k, v = kv.split("=")
        L.nir_debug_set_tunable(k.encode(), int(v))
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(3)
    H, T, M, V = a.H, a.T, a.M, a.V
    bf = a.dtype == "bf16"
    table = (torch.randn(V, 8 * H, generator=g) * 0.5).to(dev)
    if bf:
        table = table.to(torch.bfloat16)
    if a.ids == "hot":
        ids = torch.randint(1, 65, (M, T), generator=g).to(dev)
    elif a.ids == "zipf":
        import numpy as np
        r = np.random.default_rng(5).zipf(1.3, size=M * T)
        ids = torch.from_numpy(np.minimum(r, V - 1).astype("int64")).view(M, T).to(dev)
    else:
        ids = torch.randint(1, V, (M, T), generator=g).to(dev)
    lens = None
    if a.lens == "ragged":
        lens = torch.randint(1, T + 1, (M,), generator=g).to(dev)
    whh = ((torch.rand(2, 4 * H, H, generator=g) * 2 - 1) / H ** 0.5).to(dev)
    out = torch.empty(M, T, 2 * H, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    trace = torch.zeros(16 * 8, dtype=torch.int64, device=dev)
    L.nir_debug_set_buffer(vp(trace.data_ptr()))
    st = vp(torch.cuda.current_stream().cuda_stream)

    def run():
        rc = L.nir_bilstm_folded_fwd(vp(table.data_ptr()), 1 if bf else 0, vp(ids.data_ptr()), vp(lens.data_ptr()) if lens is not None else None, vp(whh.data_ptr()), vp(out.data_ptr()),
                                     vp(err.data_ptr()), M, V, T, H, 2, st)
        assert rc == 0, rc
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    wgs = (M + 15) // 16 * 2
    print("ids=%s " % a.ids + "M=%d T=%d H=%d %s: %.1f us / launch, %d workgroups (%.2f rounds of 256), %.3f us per step and round, checksum %.6f err %d"
          % (M, T, H, a.dtype, us, wgs, wgs / 256.0, us / T / max(1, -(-wgs // 256)), float(out.double().abs().sum()), int(err.item())))
    tr = trace.cpu().view(16, 8)
    if int(tr[:, 5].max()) > 0:
        print("wave  hw_id(simd,slot)  matrix   gates  barrier   total  lds-in  tile0   [s_memtime ticks / step]")
        for w in range(16):
            n = float(tr[w, 5]) or 1.0
            hw = int(tr[w, 4])
            print("%4d  simd %d slot %2d   %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f" % (w, (hw >> 4) & 3, hw & 15, tr[w, 0] / n, tr[w, 1] / n, tr[w, 2] / n, tr[w, 3] / n,
                                                                             tr[w, 6] / n, tr[w, 7] / n))


if __name__ == "__main__":
    sys.exit(main())
