"""nir_linear_f32 at one shape, 50 launches captured into ONE hipGraph (no host launch overhead, no events between kernels): us per launch.
    python tools/gemm_graph_micro.py M N K [tunable=value ...]"""
import os
os.environ.setdefault("NIR_DEBUG_TUNABLES", "1")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from context_attentive_ir_amd import lib
M, N, K = (int(x) for x in sys.argv[1:4])
L = lib.load()
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    lib.check(L.nir_debug_set_tunable(k.encode(), int(v)), "tunable")
dev = "cuda"
a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); c = torch.empty(M, N, device=dev)
def run(): lib.check(L.nir_linear_f32(lib.ptr(a), K, None, None, 0, 0, 0, lib.ptr(w), K, lib.ptr(b), None, lib.ptr(c), N, M, N, K, 0, lib.stream()), "g")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(50): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 500 * 1e3
print("M=%d N=%d K=%d %s: %.2f us per launch, %.1f TFLOP/s" % (M, N, K, " ".join(sys.argv[4:]), us, 2.0 * M * N * K / us / 1e6))
