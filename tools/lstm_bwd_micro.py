"""The BPTT kernel of the train-mode encoders (nir_lstm_train_bwd) alone, at one shape: microseconds per launch.  With --lib <variant> a
NIR_VARIANT build is timed instead (NIR_VARIANT=nomfma NIR_VARIANT_FLAGS=-DNIR_BW_NOMFMA python -m context_attentive_ir_amd.build; likewise
NIR_BW_NOSTORE / NIR_BW_NOLOAD): the timing ablations of the step (their results are wrong by construction).

    python tools/lstm_bwd_micro.py [--M 1120 --T 64 --H 128 --nd 2] [--lib nomfma]
"""
import argparse
import json
import os
import sys

os.environ.setdefault("NIR_DEBUG_TUNABLES", "1")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_attentive_ir_amd import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=1120)
    ap.add_argument("--T", type=int, default=64)
    ap.add_argument("--H", type=int, default=128)
    ap.add_argument("--nd", type=int, default=2)
    ap.add_argument("--lib", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--w8", type=int, default=0, help="tunable lstm_bwd_w8: 2 = the four-wave x two-tile form")
    a = ap.parse_args()
    if a.lib:
        lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "context_attentive_ir_amd", "libneuroir_hip_%s.so" % a.lib)
    L = lib.load()
    L.nir_debug_set_tunable(b"lstm_bwd_w8", a.w8)
    M, T, H, nd = a.M, a.T, a.H, a.nd
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.rand(*s, device="cuda", generator=g)
    act, cst, dout = r(M, T, nd, 4 * H), r(M, T, nd, H) - 0.5, (r(M, T, nd * H) - 0.5) * 1e-3
    whh = (r(nd, 4 * H, H) - 0.5) * 0.2
    lens = torch.full((M,), T, device="cuda", dtype=torch.int64)
    dg = torch.empty(M, T, nd * 4 * H, device="cuda")
    call = lambda: lib.check(L.nir_lstm_train_bwd(lib.ptr(dout), None, None, None, lib.ptr(act), lib.ptr(cst), None, lib.ptr(lens), lib.ptr(whh), lib.ptr(dg),
                                                  None, None, M, T, H, nd, lib.stream()), "nir_lstm_train_bwd")
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    print(json.dumps({"lib": a.lib or "product", "w8": a.w8, "M": M, "T": T, "H": H, "nd": nd, "us": round(us, 1), "us_per_step": round(us / T, 2),
                      "checksum": float(dg.abs().sum())}))


if __name__ == "__main__":
    main()
