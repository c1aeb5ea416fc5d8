"""Micro-benchmark of nir_linear_planes_f32 (pre-split fp16 term planes).  usage: python tools/bench_gemm_planes.py M N K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from context_attentive_ir_amd import lib
M, N, K = (int(x) for x in sys.argv[1:4])
L = lib.load(); dev = "cuda"
a = torch.rand(M, K, device=dev) - 0.5; w = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev); c = torch.empty(M, N, device=dev)
KP = (K + 7) // 8 * 8
a1, a2 = lib.split_f16x2(a, KP); w1, w2 = lib.split_f16x2(w, KP)
def run(): lib.check(L.nir_linear_planes_f32(lib.ptr(a1), lib.ptr(a2), KP, None, 0, 0, 0, 0, lib.ptr(w1), lib.ptr(w2), KP, lib.ptr(b), lib.ptr(c), N, M, N, KP, 0, lib.stream()), "p")
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50; e0.record()
for _ in range(n): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print("planes M=%d N=%d K=%d: %.1f us, %.2f TFLOP/s fp32-equivalent" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
