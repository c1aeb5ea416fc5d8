import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from context_attentive_ir_amd import lib, synth
from context_attentive_ir_amd.config import default_args
from context_attentive_ir_amd.detinit import fill_module_
from context_attentive_ir_amd.rankers import MatchTensor
L = lib.load()
V = 100000
m = fill_module_(MatchTensor(default_args("MATCH_TENSOR", src_vocab_size=V))).eval().cuda()
ex = {k: v.cuda() for k, v in synth.ranker_batch(32, 10, 4, 64, V).items()}
dbg = torch.zeros(64 + 4 * 400, dtype=torch.int64, device="cuda")
for _ in range(3): m(ex["que_rep"], ex["que_len"], ex["doc_rep"], ex["doc_len"])
L.nir_debug_set_buffer(lib.ptr(dbg))
m(ex["que_rep"], ex["que_len"], ex["doc_rep"], ex["doc_len"]); torch.cuda.synchronize()
L.nir_debug_set_buffer(None)
d = dbg.cpu()[:64].view(8, 8)
names = ["start", "prologue done", "phase1 done", "after sync", "phase2 done", "end"]
for w in range(4):
    t = d[w, :6].tolist()
    print("wave %d: " % w + ", ".join("%s +%d" % (names[i], t[i] - t[0]) for i in range(1, 6)), "(cycles @100MHz wall? see note)")
