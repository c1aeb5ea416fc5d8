import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from context_attentive_ir_amd import lib, synth
from context_attentive_ir_amd.config import default_args
from context_attentive_ir_amd.detinit import fill_module_
from context_attentive_ir_amd.rankers import MatchTensor
L = lib.load(); V = 100000
m = fill_module_(MatchTensor(default_args("MATCH_TENSOR", src_vocab_size=V))).eval().cuda()
ex = {k: v.cuda() for k, v in synth.ranker_batch(32, 10, 4, 64, V).items()}
dbg = torch.zeros(64 + 4 * 400, dtype=torch.int64, device="cuda")
for _ in range(3): m(ex["que_rep"], ex["que_len"], ex["doc_rep"], ex["doc_len"])
L.nir_debug_set_buffer(lib.ptr(dbg))
m(ex["que_rep"], ex["que_len"], ex["doc_rep"], ex["doc_len"]); torch.cuda.synchronize()
L.nir_debug_set_buffer(None)
d = dbg.cpu().numpy()[64:].reshape(-1, 4)[:320]
st, en, hw = d[:, 0], d[:, 1], d[:, 2]
t0 = st.min()
print("blocks %d; start spread %.1f us; last end %.1f us; block duration us: min %.1f median %.1f max %.1f" % (
    len(st), (st.max() - t0) / 100.0, (en.max() - t0) / 100.0, (en - st).min() / 100.0, np.median(en - st) / 100.0, (en - st).max() / 100.0))
# HW_ID bits (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]...
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
xcc = np.arange(320) % 8
key = xcc * 1000 + se * 100 + sh * 20 + cu
import collections
cnt = collections.Counter(key.tolist())
print("distinct (xcd,se,sh,cu) slots used: %d ; blocks per slot histogram: %s" % (len(cnt), sorted(collections.Counter(cnt.values()).items())))
order = np.argsort(st)
print("first 12 starts (us):", np.round((st[order][:12] - t0) / 100.0, 2), " last 12 starts:", np.round((st[order][-12:] - t0) / 100.0, 2))
