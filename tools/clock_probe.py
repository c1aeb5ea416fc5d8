import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from context_attentive_ir_amd import lib
L = lib.load()
out = torch.zeros(2, dtype=torch.int64, device="cuda"); sink = torch.zeros(1, device="cuda")
def probe(tag, iters, blocks, reps=1):
    for _ in range(reps):
        L.nir_debug_clock_probe(lib.ptr(out), iters, blocks, lib.ptr(sink), lib.stream())
    torch.cuda.synchronize()
    c, w = out.tolist()
    print("%-40s iters=%d blocks=%d: %.0f MHz (%.1f us)" % (tag, iters, blocks, c / w * 100.0, w / 100.0))
probe("cold, 1 block", 20000, 1)
probe("cold, 1 block again", 20000, 1)
probe("256 blocks", 20000, 256)
probe("2048 blocks", 20000, 2048)
probe("2048 blocks long", 400000, 2048)
probe("after long: 1 block", 20000, 1)
probe("320 blocks x50 back-to-back short", 20000, 320, 50)
time.sleep(0.5)
probe("after 0.5 s idle: 320 blocks", 20000, 320)
