"""Do two document-BiLSTM launches on different streams co-run?   usage: python tools/lstm_concurrency.py [nstreams]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from context_attentive_ir_amd import lib
L = lib.load(); dev = "cuda"
M, T, H, I = 320, 64, 70, 40
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2
wih = torch.randn(8 * H, I, device=dev) / I ** 0.5; whh = torch.randn(2, 4 * H, H, device=dev) / H ** 0.5
b1 = torch.randn(8 * H, device=dev) * .1; b2 = torch.randn(8 * H, device=dev) * .1
lens = torch.full((M,), T, dtype=torch.long, device=dev)
xs = [torch.randn(M, T, I, device=dev) for _ in range(ns)]; outs = [torch.empty(M, T, 2 * H, device=dev) for _ in range(ns)]
streams = [torch.cuda.Stream() for _ in range(ns)]
def run(j):
    with torch.cuda.stream(streams[j]):
        lib.check(L.nir_bilstm_fused_fwd(lib.ptr(xs[j]), I, lib.ptr(wih), lib.ptr(b1), lib.ptr(b2), lib.ptr(lens), lib.ptr(whh), None, None,
                                         lib.ptr(outs[j]), None, None, M, T, H, 2, lib.stream()), "lstm")
for j in range(ns): run(j)
torch.cuda.synchronize()
for label, order in (("one stream ", [0] * (20 * ns)), ("%d streams  " % ns, list(range(ns)) * 20)):
    torch.cuda.synchronize(); t = time.perf_counter()
    for j in order: run(j)
    torch.cuda.synchronize()
    print("%s: %.1f us per launch" % (label, (time.perf_counter() - t) / len(order) * 1e6))
