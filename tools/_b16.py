import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from helpers import build_model
from context_attentive_ir_amd import lib, synth
m = build_model("DUET", vocab=100000, device="cuda", max_query_len=4, max_doc_len=290)
ex = synth.ranker_batch(64, 50, 4, 290, 100000, seed=1, full_length=True)
a = [ex[k].cuda() for k in ("que_rep", "que_len", "doc_rep", "doc_len")]
for dbg in (0, 1, 2, 4, 6):
    with lib.tunable("debug", dbg, 0):
        for _ in range(3): m(*a)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): m(*a)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print("dbg=%d  %.3f ms/step" % (dbg, dt * 1e3))
