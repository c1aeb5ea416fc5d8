"""Timing of the four-workgroup resident-weight recurrence (csrc/lstm_cluster.hip): one launch alone and four in flight.
    python tools/cluster_micro.py [--M 1120 --T 64 --iters 50 --streams 4]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_attentive_ir_amd import lib  # noqa: E402
from context_attentive_ir_amd.encoders.rnn_encoder import lstm_cat_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=1120)
    ap.add_argument("--T", type=int, default=64)
    ap.add_argument("--V", type=int, default=100000)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--mode", type=int, default=1)
    ap.add_argument("--hint", type=int, default=0, help="batches-in-flight hint on every stream (> 1: the 32-sequence-group kernel)")
    ap.add_argument("--lib", default="", help="variant library suffix (NIR_VARIANT build)")
    ap.add_argument("--trace", action="store_true", help="with a -DNIR_CL_TRACE variant: phase-segment clocks of wave 0 of workgroup 0")
    a = ap.parse_args()
    if a.lib:
        lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "context_attentive_ir_amd", "libneuroir_hip_%s.so" % a.lib)
    L = lib.load()
    g = torch.Generator().manual_seed(0)
    lstm = torch.nn.LSTM(300, 256, bidirectional=True, batch_first=True)
    wih, whh, bih, bhh = [t.detach().cuda().contiguous() for t in lstm_cat_weights(lstm)]
    table = (torch.randn(a.V, 300, generator=g) * 0.5).cuda()
    rows = lib.fold_lstm_table(table, wih, bih, bhh, 256, 2, "f32")
    frag = torch.empty(L.nir_lstm256_whh_frag_bytes(2), dtype=torch.uint8, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    lib.check(L.nir_lstm256_pack_whh_frag(lib.ptr(whh), 2, lib.ptr(frag), lib.ptr(err), lib.stream()), "pack")
    M, T = a.M, a.T
    ns = max(1, a.streams)
    ids = [torch.randint(1, a.V, (M, T), generator=g).cuda() for _ in range(ns)]
    lens = [torch.full((M,), T, dtype=torch.int64).cuda() for _ in range(ns)]
    wss = [torch.empty(L.nir_lstm256_workspace_bytes(M, 2), dtype=torch.uint8, device="cuda") for _ in range(ns)]
    outs = [torch.empty((M, 512) if a.mode else (M, T, 512), device="cuda") for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    if a.hint:
        lib.set_batches_in_flight(a.hint, streams + [torch.cuda.current_stream()])

    def launch(k):
        lib.check(L.nir_lstm256_rows_fwd(lib.ptr(rows), lib.ptr(ids[k]), lib.ptr(lens[k]), lib.ptr(frag), lib.ptr(outs[k]), a.mode, lib.ptr(err),
                                         M, a.V, T, 2, lib.ptr(wss[k]), wss[k].numel(), lib.stream()), "fwd")
    for _ in range(3):
        launch(0)
    torch.cuda.synchronize()
    if a.trace:
        dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
        L.nir_debug_set_buffer(lib.ptr(dbg))
        launch(0)
        torch.cuda.synchronize()
        v = dbg.cpu().tolist()
        n = max(1, v[8])
        names = ["LDS fill + row take-over", "barrier", "publish (+ requests of rank 1)", "LDS reads + MFMAs", "gate math, h writes", "requests of rank 0", "wait for pieces", "polls x1000"]
        print("per phase (s_memtime ticks / 10 = shader cycles): " + "; ".join("%s %.0f" % (names[i], v[i] / n * (1000 if i == 7 else 1)) for i in range(8)) + "; phases %d" % n)
        L.nir_debug_set_buffer(None)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        launch(0)
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / a.iters * 1e6
    for k in range(ns):
        with torch.cuda.stream(streams[k]):
            launch(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        for k in range(ns):
            with torch.cuda.stream(streams[k]):
                launch(k)
    torch.cuda.synchronize()
    many = (time.perf_counter() - t0) / (a.iters * ns) * 1e6
    flops = 2.0 * M * T * 2 * 1024 * 256
    print("M=%d T=%d mode=%d: alone %.1f us per launch (%.1f TFLOP/s useful, %.2f us per step); %d in flight %.1f us per launch (%.1f TFLOP/s); err=%d" % (
        M, T, a.mode, alone, flops / alone / 1e6, alone / T, ns, many, flops / many / 1e6, int(err.item())))


if __name__ == "__main__":
    main()
