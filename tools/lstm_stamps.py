import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from context_attentive_ir_amd import lib
L = lib.load(); dev = "cuda"
M, T, H, I = 320, 64, 70, 40
x = torch.randn(M, T, I, device=dev); wih = torch.randn(8 * H, I, device=dev) / I ** 0.5
whh = torch.randn(2, 4 * H, H, device=dev) / H ** 0.5
b1 = torch.randn(8 * H, device=dev) * .1; b2 = torch.randn(8 * H, device=dev) * .1
lens = torch.full((M,), T, dtype=torch.long, device=dev); out = torch.empty(M, T, 2 * H, device=dev)
dbg = torch.zeros(64, dtype=torch.int64, device=dev)
def run(): lib.check(L.nir_bilstm_fused_fwd(lib.ptr(x), I, lib.ptr(wih), lib.ptr(b1), lib.ptr(b2), lib.ptr(lens), lib.ptr(whh), None, None, lib.ptr(out), None, None, M, T, H, 2, lib.stream()), "lstm")
for _ in range(3): run()
L.nir_debug_set_buffer(lib.ptr(dbg)); run(); torch.cuda.synchronize(); L.nir_debug_set_buffer(None)
d = dbg.cpu().view(8, 8)
for w in range(4):
    t = d[w, :5].tolist()
    print("wave %d: mfma+partial-store %d | wait barrier1 %d | cell %d | wait barrier2 %d | step total %d cycles" % (w, t[1]-t[0], t[2]-t[1], t[3]-t[2], t[4]-t[3], t[4]-t[0]))
