"""Micro-benchmark of the LSTM recurrence.  usage: python tools/bench_lstm.py M T H [I]  (I given -> fused variant)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from context_attentive_ir_amd import lib
M, T, H = (int(x) for x in sys.argv[1:4])
I = int(sys.argv[4]) if len(sys.argv) > 4 else 0
L = lib.load()
dev = "cuda"
whh = torch.randn(2, 4 * H, H, device=dev) / H ** 0.5
lens = torch.full((M,), T, dtype=torch.long, device=dev)
out = torch.empty(M, T, 2 * H, device=dev)
if I:
    x = torch.randn(M, T, I, device=dev); wih = torch.randn(8 * H, I, device=dev) / I ** 0.5
    b1 = torch.randn(8 * H, device=dev) * .1; b2 = torch.randn(8 * H, device=dev) * .1
    def run():
        lib.check(L.nir_bilstm_fused_fwd(lib.ptr(x), I, lib.ptr(wih), lib.ptr(b1), lib.ptr(b2), lib.ptr(lens), lib.ptr(whh), None, None, lib.ptr(out), None, None, M, T, H, 2, lib.stream()), "lstm")
else:
    gates = torch.randn(M, T, 8 * H, device=dev) * 0.5
    def run():
        lib.check(L.nir_bilstm_fwd(lib.ptr(gates), lib.ptr(lens), lib.ptr(whh), None, None, lib.ptr(out), None, None, M, T, H, 2, lib.stream()), "lstm")
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
fl = M * 2 * T * 2 * 4 * H * (H + I)
print("S=%s M=%d T=%d H=%d I=%d: %.1f us/launch, %.2f us/step, %.2f TFLOP/s" % (os.environ.get("NIR_LSTM_S", "auto"), M, T, H, I, us, us / T, fl / us / 1e6))
