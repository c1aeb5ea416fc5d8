"""Micro-benchmark of nir_linear_f32.  usage: python tools/bench_gemm.py M N K [gather] [bounded]  (bounded: operands in (-1, 1), the fp16 two-term form)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from context_attentive_ir_amd import lib
M, N, K = (int(x) for x in sys.argv[1:4])
gather = "gather" in sys.argv[4:]
ACT = 0x100 if "bounded" in sys.argv[4:] else 0
if os.environ.get("NIR_TOOL_LIB"):
    lib.LIB_PATH = os.environ["NIR_TOOL_LIB"]          # A/B against a variant build (NIR_VARIANT=... python -m context_attentive_ir_amd.build)
L = lib.load(); dev = "cuda"
w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev); c = torch.empty(M, N, device=dev)
if gather:
    V = 100000; table = torch.randn(V, K, device=dev); ids = torch.randint(0, V, (M,), device=dev)
    def run(): lib.check(L.nir_linear_f32(None, 0, lib.ptr(ids), lib.ptr(table), K, 1, 1, lib.ptr(w), K, lib.ptr(b), None, lib.ptr(c), N, M, N, K, ACT, lib.stream()), "g")
else:
    a = torch.rand(M, K, device=dev) * 2 - 1
    def run(): lib.check(L.nir_linear_f32(lib.ptr(a), K, None, None, 0, 0, 0, lib.ptr(w), K, lib.ptr(b), None, lib.ptr(c), N, M, N, K, ACT, lib.stream()), "g")
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 100; e0.record()
for _ in range(n): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
ref = (table[ids] if gather else a).double() @ w.double().t() + b.double()
print("max |err| vs float64 %.2e" % float((c.double() - ref).abs().max()))
print("M=%d N=%d K=%d gather=%s: %.1f us, %.2f TFLOP/s, A-read %.0f GB/s" % (M, N, K, gather, us, 2.0 * M * N * K / us / 1e6, M * K * 4 / us / 1e3))
