"""CPU: the oracle (oracle/neuroir_cpu.py) against the golden vectors produced by the REAL reference
(tests/golden/generate.py).  This is what pins the oracle; tolerance 1e-6 (same ATen ops, same machine class)."""
import numpy as np
import pytest
import torch

from conftest import T, load_golden
from helpers import build_model, cpu_state_dict
from oracle import neuroir_cpu as O

TOL = 1e-6


def _close(a, b, tol=TOL):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=0, atol=tol)


def test_esm():
    g = load_golden("esm")
    sd = cpu_state_dict(build_model("ESM"))
    s = O.esm_scores(sd, T(g["que_rep"]), T(g["que_len"]), T(g["doc_rep"]), T(g["doc_len"]))
    _close(s, g["scores"])
    _close(O.predict_softmax(s), g["softmax"])
    assert s[1, 2].item() == 0.0  # all-PAD document -> zero vector -> cosine 0


def test_match_tensor():
    g = load_golden("match_tensor")
    sd = cpu_state_dict(build_model("MATCH_TENSOR"))
    args = [T(g[k]) for k in ("que_rep", "que_len", "doc_rep", "doc_len")]
    hq, hd, pq, pd = O.match_tensor_parts(sd, *args)
    _close(hq, g["enc_q"]); _close(hd, g["enc_d"]); _close(pq, g["proj_q"]); _close(pd, g["proj_d"])
    _close(O.match_tensor_scores(sd, *args), g["scores"], 2e-6)


@pytest.mark.parametrize("tag", ["safe", "overlap"])
def test_drmm(tag):
    g = load_golden("drmm_" + tag)
    sd = cpu_state_dict(build_model("DRMM"))
    q, d = T(g["que_rep"]), T(g["doc_rep"])
    gate, cos, hist = O.drmm_parts(sd, q, d)
    _close(gate, g["gate"]); _close(cos, g["cos"])
    np.testing.assert_array_equal(hist.numpy(), g["hist"])       # integer counts: bit-exact
    _close(O.drmm_scores(sd, q, T(g["que_len"]), d, T(g["doc_len"])), g["scores"], 1e-5)


def test_duet():
    g = load_golden("duet")
    QL, DL = g["que_rep"].shape[1], g["doc_rep"].shape[2]
    sd = cpu_state_dict(build_model("DUET", max_query_len=QL, max_doc_len=DL))
    q, d = T(g["que_rep"]), T(g["doc_rep"])
    _close(O.duet_local(sd, q, d), g["local"])
    _close(O.duet_distributed(sd, q, d), g["dist"])
    _close(O.duet_scores(sd, q, None, d, None), g["scores"])


@pytest.mark.parametrize("tag", ["oneclick", "multiclick"])
def test_cars(tag):
    g = load_golden("cars_" + tag)
    sd = cpu_state_dict(build_model("CARS"))
    q, ql, d, dl, lab = (T(g[k]) for k in ("source_words", "source_lens", "document_words", "document_lens", "document_labels"))
    pooled, enc = O.cars_encode(sd, q, ql)
    _close(pooled, g["pooled_q"]); _close(enc, g["enc_q"])
    docs = O.cars_encode_document(sd, d, dl)
    _close(docs, g["pooled_docs"])
    _close(O.cars_encode_clicks(sd, docs, lab), g["encoded_clicks"])
    s = O.cars_rank_document(sd, pooled, d, dl, lab)
    _close(s, g["click_scores"], 2e-6)
    _close(O.bce_with_logits(s, lab), g["ranking_loss"])
    _close(O.predict_softmax(s), g["softmax"])


def test_m_match_tensor():
    g = load_golden("m_match_tensor")
    sd = cpu_state_dict(build_model("M_MATCH_TENSOR", tgt_vocab_size=int(g["tgt_vocab_size"])))
    src, sl, d, dl = (T(g[k]) for k in ("source_words", "source_lens", "document_words", "document_lens"))
    _close(O.m_match_tensor_encode(sd, src, sl), g["projected_queries"])
    s = O.m_match_tensor_scores(sd, src, sl, d, dl)
    _close(s, g["scores"], 2e-6)
    _close(O.predict_softmax(s), g["softmax"])
    B, S = src.shape[:2]                                        # suggestion side: session states and the greedy decode
    bank, h, c = O.session_decoder_states(O._strip_encoder_nesting(sd), "session_query_encoder", T(g["projected_queries"]).max(1)[0].view(B, S, -1))
    _close(bank, g["session_bank"], 2e-6); _close(h, g["dec_h"], 2e-6); _close(c, g["dec_c"], 2e-6)
    p = O.plain_greedy_decode(sd, sd["embedder.word_embeddings.make_embedding.emb_luts.0.weight"], h, c, int(g["max_len"]), T(g["tgt2src"]))
    assert torch.equal(p.view(B, S - 1, -1), T(g["predictions"]))


def test_mnsrf():
    g = load_golden("mnsrf")
    sd = cpu_state_dict(build_model("MNSRF", tgt_vocab_size=int(g["tgt_vocab_size"])))
    src, sl, d, dl = (T(g[k]) for k in ("source_words", "source_lens", "document_words", "document_lens"))
    mem, sess = O.mnsrf_encode(sd, src, sl)
    _close(mem, g["memory_bank"]); _close(sess, g["session_bank"])
    s = O.mnsrf_scores(sd, src, sl, d, dl)
    _close(s, g["scores"], 5e-6)
    _close(O.predict_softmax(s), g["softmax"], 2e-6)
    B, S = src.shape[:2]
    bank, h, c = O.session_decoder_states(O._strip_encoder_nesting(sd), "session_query_encoder", mem)
    _close(bank, g["session_bank"], 2e-6); _close(h, g["dec_h"], 2e-6); _close(c, g["dec_c"], 2e-6)
    p = O.plain_greedy_decode(sd, sd["embedder.word_embeddings.make_embedding.emb_luts.0.weight"], h, c, int(g["max_len"]), T(g["tgt2src"]))
    assert torch.equal(p.view(B, S - 1, -1), T(g["predictions"]))


def test_losses_and_metrics():
    g = load_golden("losses_metrics")
    s, y = T(g["scores"]), T(g["labels"])
    _close(O.bce_with_logits(s, y), g["bce"]); _close(O.softmax_nll(s, y), g["softmax_nll"])
    _close(O.predict_softmax(s), g["softmax"])
    pred = np.argsort(-g["softmax"], kind="stable")
    np.testing.assert_array_equal(pred, g["predictions"])
    assert O.mean_average_precision(pred, g["labels"]) == pytest.approx(float(g["MAP"]), abs=1e-12)
    assert O.mean_reciprocal_rank(pred, g["labels"]) == pytest.approx(float(g["MRR"]), abs=1e-12)
    assert O.precision_at_k(pred, g["labels"], 1) == pytest.approx(float(g["P1"]), abs=1e-12)
    assert O.precision_at_k(pred, g["labels"], 3) == pytest.approx(float(g["P3"]), abs=1e-12)


CARS_CFG = {"full": {}, "qoff": dict(query_session_off=True), "doff": dict(doc_session_off=True)}


@pytest.mark.parametrize("tag", ["full", "qoff", "doff"])
def test_cars_decode_and_switches(tag):
    """Oracle restatement of the session switches, the decoder-initialisation states, the inner-attention pools and the
    greedy decoder against the real reference (tests/golden/generate.py:gen_cars_decode)."""
    g = load_golden("cars_decode")
    V = int(g["meta_vocab"])
    kw = CARS_CFG[tag]
    sd = cpu_state_dict(build_model("CARS", tgt_vocab_size=V, **kw))
    q, ql, d, dl, lab = (T(g[k]) for k in ("source_words", "source_lens", "document_words", "document_lens", "document_labels"))
    B, S, _ = q.shape
    pooled, enc = O.cars_encode(sd, q, ql)
    q_on, d_on = not kw.get("query_session_off", False), not kw.get("doc_session_off", False)
    scores, states, attns = O.cars_rank_document_full(sd, pooled, d, dl, lab, q_on=q_on, d_on=d_on)
    _close(scores, g[tag + "_click_scores"], 2e-6)
    _close(states[0], g[tag + "_dec_h"], 2e-6); _close(states[1], g[tag + "_dec_c"], 2e-6)
    if q_on:
        _close(attns[0], g[tag + "_inner_q"], 2e-6)
    if d_on:
        _close(attns[1], g[tag + "_inner_d"], 2e-6)
    pred = O.cars_decode(sd, states, int(g["max_len"]), B, S - 1, enc, ql, attns, tgt2src=T(g["tgt2src"]))
    assert (pred.numpy() == g[tag + "_predictions"]).all()


def test_cars_both_sessions_off():
    g = load_golden("cars_decode")
    sd = cpu_state_dict(build_model("CARS", tgt_vocab_size=int(g["meta_vocab"]), query_session_off=True, doc_session_off=True,
                                    turn_recommender_off=True))
    q, ql, d, dl, lab = (T(g[k]) for k in ("source_words", "source_lens", "document_words", "document_lens", "document_labels"))
    pooled, _ = O.cars_encode(sd, q, ql)
    scores, states, _ = O.cars_rank_document_full(sd, pooled, d, dl, lab, q_on=False, d_on=False, recommender=False)
    _close(scores, g["bothoff_click_scores"], 2e-6)
    assert states is None


def test_match_tensor_train_gradients():
    """The oracle's differentiable train-mode MatchTensor + BCE against the gradients of the reference's own backward
    (tests/golden/generate.py:gen_train; dropout 0)."""
    g = load_golden("match_tensor_train")
    sd = {k: v.clone().requires_grad_(not k.startswith("word_embeddings")) for k, v in cpu_state_dict(build_model("MATCH_TENSOR")).items()}
    q, ql, d, dl, lab = (T(g["b0_" + k]) for k in ("que_rep", "que_len", "doc_rep", "doc_len", "label"))
    s = O.match_tensor_train_scores(sd, q, ql, d, dl)
    _close(s, g["scores0"], 2e-6)
    loss = O.bce_with_logits(s, lab)
    _close(loss, g["loss0"], 1e-6)
    loss.backward()
    for k, v in sd.items():
        if torch.is_tensor(v) and v.requires_grad:
            _close(v.grad, g["grad_" + k], 2e-6)


RNN_CFGS = dict(gru2_bi_bridge=dict(rnn_type="GRU", bidirectional=True, nlayers=2, use_bridge=True, use_last=False),
                lstm2_uni=dict(rnn_type="LSTM", bidirectional=False, nlayers=2, use_bridge=False, use_last=True),
                lstm2_bi_cat=dict(rnn_type="LSTM", bidirectional=True, nlayers=2, use_bridge=True, use_last=False),
                lstm1_uni_init=dict(rnn_type="LSTM", bidirectional=False, nlayers=1, use_bridge=False, use_last=True))


def rnn_fixture(name):
    g = load_golden("rnn_encoder")
    pre = name + "."
    sd = {k[len(pre) + 3:]: T(v) for k, v in g.items() if k.startswith(pre + "sd.")}
    lens = T(g[pre + "lens"]) if g[pre + "lens"].size else None
    init = (T(g[pre + "init_h"]), T(g[pre + "init_c"])) if pre + "init_h" in g else None
    return g, sd, T(g[pre + "x"]), lens, init


@pytest.mark.parametrize("name", sorted(RNN_CFGS))
def test_rnn_encoder_general(name):
    """oracle.rnn_encoder_general against the reference's RNNEncoder outside the 1-layer LSTM (GRU, stacked layers, bridge, use_last=False,
    initial states): rnn_encoder.npz."""
    g, sd, x, lens, init = rnn_fixture(name)
    fin, mem = O.rnn_encoder_general(sd, "", x, lens, init=init, **RNN_CFGS[name])
    assert float((mem - T(g[name + ".bank"])).abs().max()) < 1e-6
    h = fin[0] if isinstance(fin, tuple) else fin
    assert float((h - T(g[name + ".h"])).abs().max()) < 1e-6
    if isinstance(fin, tuple):
        assert float((fin[1] - T(g[name + ".c"])).abs().max()) < 1e-6


@pytest.mark.parametrize("tag,rnn_type,nlayers", [("gru2", "GRU", 2), ("lstm2", "LSTM", 2), ("gru1", "GRU", 1)])
def test_match_tensor_general(tag, rnn_type, nlayers):
    """oracle.match_tensor_general_scores against the reference MatchTensor built with GRU / stacked encoders (match_tensor_general.npz)."""
    g = load_golden("match_tensor_general")
    m = build_model("MATCH_TENSOR", rnn_type=rnn_type, nlayers=nlayers)
    s, hq, hd = O.match_tensor_general_scores(cpu_state_dict(m), T(g[tag + ".que_rep"]), T(g[tag + ".que_len"]), T(g[tag + ".doc_rep"]),
                                              T(g[tag + ".doc_len"]), rnn_type, nlayers)
    assert float((hq - T(g[tag + ".enc_q"])).abs().max()) < 1e-6 and float((hd - T(g[tag + ".enc_d"])).abs().max()) < 1e-6
    assert float((s - T(g[tag + ".scores"])).abs().max()) < 1e-5
