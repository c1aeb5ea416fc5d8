"""Parity spot checks at shapes outside the regular test grid (long documents, many candidates).  usage: python tools/extra_shapes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import build_model, cpu_state_dict
from oracle import neuroir_cpu as O
from context_attentive_ir_amd import synth

def batch(B, N, QL, DL, V, seed):
    rng = np.random.default_rng(seed)
    ql = rng.integers(1, QL + 1, size=B); dl = rng.integers(1, DL + 1, size=(B, N)); ql[0] = QL; dl[0, 0] = DL
    q = rng.integers(4, V, size=(B, QL)); d = rng.integers(4, V, size=(B, N, DL))
    q[np.arange(QL)[None] >= ql[:, None]] = 0; d[np.arange(DL)[None, None] >= dl[..., None]] = 0
    return [torch.from_numpy(x.astype(np.int64)) for x in (q, ql, d, dl)]

for name, fn, shape in (("MATCH_TENSOR", O.match_tensor_scores, (3, 5, 6, 290)), ("MATCH_TENSOR", O.match_tensor_scores, (2, 50, 4, 64)),
                        ("MATCH_TENSOR", O.match_tensor_scores, (70, 20, 4, 64)), ("ESM", O.esm_scores, (5, 50, 7, 290))):
    m = build_model(name, vocab=500, device="cuda")
    q, ql, d, dl = batch(*shape, 500, 3)
    ref = fn(cpu_state_dict(m), q, ql, d, dl)
    got = m(q.cuda(), ql.cuda(), d.cuda(), dl.cuda()).cpu()
    print("%-13s B,N,QL,DL=%s  max|diff| = %.2e" % (name, shape, float((got - ref).abs().max())))
    assert float((got - ref).abs().max()) < 1e-4
m = build_model("CARS", vocab=500, device="cuda")
b = synth.session_batch(2, 3, 50, 4, 64, 500, 5)
sd = cpu_state_dict(m)
ref = O.cars_scores(sd, b["source_words"], b["source_lens"], b["document_words"], b["document_lens"], b["document_labels"])
pooled, _, _ = m.encode(b["source_words"].cuda(), b["source_lens"].cuda())
s, _, _ = m.rank_document(pooled, b["document_words"].cuda(), b["document_lens"].cuda(), b["document_labels"].cuda())
got = s.cpu()
print("CARS          B,S,N=2,3,50      max|diff| = %.2e" % float((got - ref.view_as(got)).abs().max()))
assert float((got - ref.view_as(got)).abs().max()) < 1e-4
print("ok")
