"""Micro-harness for the CARS session tail (nir_cars_rank_session_rows: click attention, session LSTMs, cross attention, ranknet) at a chosen
macro-batch: per-kernel time from the library profiler (serial, HIP events) and the time of the whole tail as ONE hipGraph replay.

    python tools/tail_micro.py [--B 128 --S 7 --N 10 --iters 30]
"""
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
    ap.add_argument("--B", type=int, default=128)
    ap.add_argument("--S", type=int, default=7)
    ap.add_argument("--N", type=int, default=10)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    from context_attentive_ir_amd import lib
    from helpers import build_model
    L = lib.load()
    m = build_model("CARS", vocab=2000, device="cuda")
    g = torch.Generator().manual_seed(3)
    pq = (torch.rand(a.B, a.S, 256, generator=g) * 2 - 1).cuda()
    docs = (torch.rand(a.B, a.S, a.N, 256, generator=g) * 2 - 1).cuda()
    lab = torch.zeros(a.B, a.S, a.N)
    lab[:, :, 0] = 1
    lab = lab.cuda()
    for _ in range(3):
        m._rank_session(pq, docs, lab)
    torch.cuda.synchronize()
    L.nir_profile_enable(1)
    for _ in range(a.iters):
        m._rank_session(pq, docs, lab)
    torch.cuda.synchronize()
    L.nir_profile_enable(0)
    buf = C.create_string_buffer(1 << 16)
    L.nir_profile_report(buf, len(buf))
    tot = 0.0
    for line in buf.value.decode().strip().splitlines():
        k, cnt, ms = line.rsplit(",", 2)
        per_call = float(ms) / a.iters * 1e3
        tot += per_call
        print("%-52s %3.0f launches/tail  %8.2f us/tail  (%6.2f us each)" % (k, int(cnt) / a.iters, per_call, float(ms) / int(cnt) * 1e3))
    print("sum of kernels %.1f us per tail (serial, HIP events around every launch)" % tot)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        m._rank_session(pq, docs, lab)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            m._rank_session(pq, docs, lab)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
    print("whole tail as one hipGraph replay: %.1f us" % (e0.elapsed_time(e1) / a.iters * 1e3))


if __name__ == "__main__":
    main()
