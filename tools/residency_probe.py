import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from context_attentive_ir_amd import lib
L = lib.load()
out = torch.zeros(2 + 3 * 2048, dtype=torch.int64, device="cuda"); sink = torch.zeros(1, device="cuda")
for blocks in (256, 320, 512, 768, 1024, 2048):
    out.zero_()
    L.nir_debug_clock_probe(lib.ptr(out), 20000, blocks, lib.ptr(sink), lib.stream()); torch.cuda.synchronize()
    d = out.cpu().numpy()[2:2 + 3 * blocks].reshape(-1, 3)
    st, en, hw = d[:, 0], d[:, 1], d[:, 2]
    t0 = st.min()
    late = (st - t0) > 0.5 * np.median(en - st)
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = (np.arange(blocks) % 8) * 1000 + se * 100 + sh * 20 + cu
    print("blocks=%4d: total %.1f us, block median %.1f us, started late: %d, distinct CU slots %d" % (
        blocks, (en.max() - t0) / 100.0, np.median(en - st) / 100.0, late.sum(), len(set(key.tolist()))))
