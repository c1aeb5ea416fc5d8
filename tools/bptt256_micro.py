"""nir_lstm256_bptt alone (the T step launches of the 256-unit encoders' BPTT) at one shape: microseconds per BPTT and per step, eagerly and as one
hipGraph.  --lib <variant>: a NIR_VARIANT build (NIR_VARIANT=bNOMMA NIR_VARIANT_FLAGS=-DNIR_B256_NOMMA python -m context_attentive_ir_amd.build;
likewise NIR_B256_NOPART / NIR_B256_NODGX / NIR_B256_NOA): the timing ablations of the step (their results are wrong by construction).

    python tools/bptt256_micro.py [--M 1120 --T 64 --nd 2] [--lib bNOMMA]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_attentive_ir_amd import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=1120)
    ap.add_argument("--T", type=int, default=64)
    ap.add_argument("--nd", type=int, default=2)
    ap.add_argument("--lib", default="")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    if a.lib:
        lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "context_attentive_ir_amd", "libneuroir_hip_%s.so" % a.lib)
    L = lib.load()
    M, T, nd, H = a.M, a.T, a.nd, 256
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.rand(*s, device="cuda", generator=g)           # noqa: E731
    act, cst, dout = r(M, T, nd, 4 * H), r(M, T, nd, H) - 0.5, (r(M, T, nd * H) - 0.5) * 1e-3
    whh = (r(nd, 4 * H, H) - 0.5) * 0.1
    lens = torch.full((M,), T, device="cuda", dtype=torch.int64)
    dg = torch.empty(M, T, nd * 4 * H, device="cuda")
    ws = torch.empty(L.nir_lstm256_bptt_workspace_bytes(M, nd), dtype=torch.uint8, device="cuda")
    call = lambda: lib.check(L.nir_lstm256_bptt(lib.ptr(dout), lib.ptr(act), lib.ptr(cst), lib.ptr(lens), lib.ptr(whh), lib.ptr(dg), M, T, nd,     # noqa: E731
                                                lib.ptr(ws), ws.numel(), lib.stream()), "nir_lstm256_bptt")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    eager = e0.elapsed_time(e1) / a.iters * 1e3
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        call()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            call()
        torch.cuda.synchronize()
        e0.record(st)
        for _ in range(a.iters):
            gr.replay()
        e1.record(st)
    torch.cuda.synchronize()
    graphed = e0.elapsed_time(e1) / a.iters * 1e3
    print("%-8s M=%d T=%d nd=%d: eager %.0f us (%.1f per step), graphed %.0f us (%.1f per step)" % (a.lib or "product", M, T, nd, eager, eager / T, graphed, graphed / T))


if __name__ == "__main__":
    main()
