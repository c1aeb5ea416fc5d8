"""GPU (-m gpu): the HIP RNNEncoder outside the hyparam-pinned 1-layer LSTM -- GRU, stacked layers, bridge, use_last = False, initial
states (/root/reference/neuroir/encoders/rnn_encoder.py:14-185) -- against the reference's own outputs (golden/rnn_encoder.npz) and the
CPU oracle on larger seeded shapes.  Tolerance 1e-5 on fp32 states (the recurrences chain up to 3 x 40 steps)."""
import numpy as np
import pytest
import torch

from conftest import T
from oracle import neuroir_cpu as O
from test_oracle_golden import RNN_CFGS, rnn_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(sd, x, cfg, dropout=0.0):
    from context_attentive_ir_amd.encoders.rnn_encoder import RNNEncoder
    G = 4 if cfg["rnn_type"] == "LSTM" else 3
    ndir = 2 if cfg["bidirectional"] else 1
    hid = sd["rnns.0.weight_ih_l0"].shape[0] // G * ndir
    enc = RNNEncoder(cfg["rnn_type"], x.shape[2], cfg["bidirectional"], cfg["nlayers"], hid, dropout, use_bridge=cfg["use_bridge"],
                     use_last=cfg["use_last"])
    assert set(enc.state_dict()) == set(sd), "state-dict keys differ from the reference's"
    enc.load_state_dict(sd)
    return enc.to(DEV).eval()


def unsort(fin_sorted, lens):
    """reference final states are in length-sorted batch order; the HIP encoder keeps the original order"""
    if lens is None:
        return fin_sorted
    order = torch.sort(lens, 0, True)[1]
    out = torch.empty_like(fin_sorted)
    out[:, order] = fin_sorted
    return out


@pytest.mark.parametrize("name", sorted(RNN_CFGS))
def test_rnn_encoder_matches_reference_fixture(name):
    g, sd, x, lens, init = rnn_fixture(name)
    enc = build(sd, x, RNN_CFGS[name])
    fin, mem = enc(x.to(DEV), lens.to(DEV) if lens is not None else None, tuple(t.to(DEV) for t in init) if init is not None else None)
    assert float((mem.cpu() - T(g[name + ".bank"])).abs().max()) < 1e-5
    h = fin[0] if isinstance(fin, tuple) else fin
    assert float((h.cpu() - unsort(T(g[name + ".h"]), lens)).abs().max()) < 1e-5
    if isinstance(fin, tuple):
        assert float((fin[1].cpu() - unsort(T(g[name + ".c"]), lens)).abs().max()) < 1e-5


@pytest.mark.parametrize("rnn_type,bi,nl,hid,inp,M,Tn,bridge,last", [("GRU", True, 3, 80, 50, 37, 40, False, True), ("GRU", False, 1, 256, 300, 20, 12, True, True),
                                                                      ("LSTM", True, 2, 256, 64, 33, 20, False, False), ("LSTM", True, 3, 140, 30, 18, 25, True, True),
                                                                      ("GRU", True, 2, 24, 8, 9, 9, True, False)])   # (bridge over several layers mixes sort-neighbours: distinct lengths)
def test_rnn_encoder_matches_oracle(rnn_type, bi, nl, hid, inp, M, Tn, bridge, last):
    from context_attentive_ir_amd.detinit import fill_module_
    from context_attentive_ir_amd.encoders.rnn_encoder import RNNEncoder
    enc = fill_module_(RNNEncoder(rnn_type, inp, bi, nl, hid, 0.0, use_bridge=bridge, use_last=last), 77).eval()
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    rng = np.random.default_rng(5)
    x = T(rng.normal(size=(M, Tn, inp)).astype(np.float32))
    lens = T(rng.permutation(np.arange(M) % Tn + 1).astype(np.int64))
    if int(lens.max()) < Tn:
        lens[0] = Tn
    cfg = dict(rnn_type=rnn_type, bidirectional=bi, nlayers=nl, use_bridge=bridge, use_last=last)
    fin_o, mem_o = O.rnn_encoder_general(sd, "", x, lens, **cfg)
    enc.to(DEV)
    fin, mem = enc(x.to(DEV), lens.to(DEV))
    assert float((mem.cpu() - mem_o).abs().max()) < 1e-5
    order = torch.sort(lens, 0, True)[1]                      # the oracle's (= reference's) sort; ties permute equal-length rows only

    def same(a, b_sorted):
        b = torch.empty_like(b_sorted)
        b[:, order] = b_sorted
        return float((a.cpu() - b).abs().max())
    if isinstance(fin, tuple):
        assert same(fin[0], fin_o[0]) < 1e-5 and same(fin[1], fin_o[1]) < 1e-5
    else:
        assert same(fin, fin_o) < 1e-5


def test_rnn_encoder_rejects_what_it_does_not_cover():
    from context_attentive_ir_amd.encoders.rnn_encoder import RNNEncoder
    with pytest.raises(NotImplementedError):
        RNNEncoder("RNN", 8, True, 1, 16)
    enc = RNNEncoder("GRU", 8, True, 2, 16, dropout=0.3).to(DEV).train()
    with pytest.raises(NotImplementedError):
        enc(torch.zeros(2, 3, 8, device=DEV), None)


@pytest.mark.parametrize("tag,rnn_type,nlayers", [("gru2", "GRU", 2), ("lstm2", "LSTM", 2), ("gru1", "GRU", 1)])
def test_match_tensor_with_general_encoders_golden(tag, rnn_type, nlayers):
    """MATCH_TENSOR built with GRU / stacked encoders (mtensor.py:36-49 passes rnn_type / nlayers through): the HIP ranker against the
    reference's own scores and encoder states (golden/match_tensor_general.npz), tolerance 1e-4 on scores like the other rankers."""
    from conftest import load_golden
    from helpers import build_model
    g = load_golden("match_tensor_general")
    m = build_model("MATCH_TENSOR", device=DEV, rnn_type=rnn_type, nlayers=nlayers)
    s, (hq, hd, pq, pd) = m(T(g[tag + ".que_rep"]).to(DEV), T(g[tag + ".que_len"]).to(DEV), T(g[tag + ".doc_rep"]).to(DEV),
                            T(g[tag + ".doc_len"]).to(DEV), return_parts=True)
    assert float((hq.cpu() - T(g[tag + ".enc_q"])).abs().max()) < 2e-5 and float((hd.cpu() - T(g[tag + ".enc_d"])).abs().max()) < 2e-5
    assert float((s.cpu() - T(g[tag + ".scores"])).abs().max()) < 1e-4


def test_match_tensor_gru_oracle_long_documents():
    from helpers import build_model, cpu_state_dict
    rng = np.random.default_rng(9)
    m = build_model("MATCH_TENSOR", vocab=300, device=DEV, rnn_type="GRU", nlayers=2)
    B, N, QL, DL = 3, 5, 7, 130
    qlen = rng.integers(1, QL + 1, size=B); dlen = rng.integers(1, DL + 1, size=(B, N)); qlen[0] = QL; dlen[0, 0] = DL
    q = rng.integers(4, 300, size=(B, QL)); d = rng.integers(4, 300, size=(B, N, DL))
    q[np.arange(QL)[None] >= qlen[:, None]] = 0
    d[np.arange(DL)[None, None] >= dlen[..., None]] = 0
    q, ql, d, dl = (T(x.astype(np.int64)) for x in (q, qlen, d, dlen))
    ref = O.match_tensor_general_scores(cpu_state_dict(m), q, ql, d, dl, "GRU", 2)[0]
    got = m(q.to(DEV), ql.to(DEV), d.to(DEV), dl.to(DEV))
    assert float((got.cpu() - ref).abs().max()) < 1e-4
