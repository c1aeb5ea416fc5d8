import ctypes, torch, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from helpers import build_model
from context_attentive_ir_amd import lib, synth
m = build_model("DUET", vocab=100000, device="cuda", max_query_len=4, max_doc_len=290)
ex = synth.ranker_batch(64, 50, 4, 290, 100000, seed=1, full_length=True)
a = [ex[k].cuda() for k in ("que_rep", "que_len", "doc_rep", "doc_len")]
for _ in range(3): m(*a)
torch.cuda.synchronize()
L = ctypes.CDLL(lib.load()._name)
buf = (ctypes.c_longlong * 16)()
L.nir_debug_duet_timing(buf)
t = list(buf)[:6]
print("phases (cycles): prologue %d gemm1 %d epi1 %d gemm2 %d epi2 %d total %d" % (t[1]-t[0], t[2]-t[1], t[3]-t[2], t[4]-t[3], t[5]-t[4], t[5]-t[0]))
