"""dW = dY^T X (+ db) of the training step: the LDS-staged workgroup-tile kernel against the register-blocked one (tunable wgrad_no_lds)
and torch, per shape: max |diff| relative to the fp64 product and microseconds per launch.

    python tools/wgrad_micro.py [--shapes 71680x1024x300,71680x512x128]
"""
import argparse
import json
import os
import sys

os.environ.setdefault("NIR_DEBUG_TUNABLES", "1")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_attentive_ir_amd import lib  # noqa: E402


def run(L, dy, x, db, set_=True, iters=20):
    M, N = dy.shape
    K = x.shape[1]
    dw = torch.empty(N, K, device="cuda")
    call = lambda: lib.check(L.nir_linear_wgrad_bias_set_f32(lib.ptr(dy), N, lib.ptr(x), K, None, None, 0, lib.ptr(dw), K, lib.ptr(db), M, N, K,
                                                             lib.stream()), "wgrad")
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    return dw, e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="71680x1024x300,71680x512x128,8960x1024x300,20000x256x512,4480x768x256")
    ap.add_argument("--tiles", type=int, default=0, help="tunable wgrad_lds_tiles: LDS kernel for M >= 256 once the dW has this many 128 x 128 tiles")
    ap.add_argument("--min-rows", type=int, default=0)
    a = ap.parse_args()
    L = lib.load()
    for sh in a.shapes.split(","):
        M, N, K = (int(v) for v in sh.split("x"))
        g = torch.Generator(device="cuda").manual_seed(1)
        dy = torch.randn(M, N, device="cuda", generator=g) * 1e-3
        x = torch.randn(M, K, device="cuda", generator=g)
        ref = (dy.double().t() @ x.double())
        refb = dy.double().sum(0)
        rec = {"shape": sh, "gflop": 2e-9 * M * N * K}
        L.nir_debug_set_tunable(b"wgrad_lds_tiles", a.tiles)
        L.nir_debug_set_tunable(b"wgrad_min_rows", a.min_rows)
        for name, flag in (("lds", 0), ("regs", 1)):
            L.nir_debug_set_tunable(b"wgrad_no_lds", flag)
            db = torch.empty(N, device="cuda")
            dw, us = run(L, dy, x, db)
            rec[name + "_us"] = round(us, 1)
            rec[name + "_tflops"] = round(rec["gflop"] / us * 1e-3, 1)
            rec[name + "_err"] = float((dw.double() - ref).abs().max() / ref.abs().max())
            rec[name + "_db_err"] = float((db.double() - refb).abs().max() / refb.abs().max())
        L.nir_debug_set_tunable(b"wgrad_no_lds", 0)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dy.t() @ x
        t0.record()
        for _ in range(10):
            dy.t() @ x
        t1.record()
        torch.cuda.synchronize()
        rec["torch_us"] = round(t0.elapsed_time(t1) / 10 * 1e3, 1)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
