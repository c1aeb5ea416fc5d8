#!/usr/bin/env python
"""Generate golden parity fixtures by running the REAL reference (/root/reference) on CPU.

Runs only in the authoring container (the reference never travels to the GPU box).
Weights come from context_attentive_ir_amd.detinit (counter-based, keyed by state-dict
name) and are load_state_dict-ed into the reference modules, so the fixtures carry only
inputs + expected outputs; every consumer regenerates identical weights from the key names.

Compat shims are installed before importing neuroir; none of them
changes arithmetic.

    python tests/golden/generate.py          # rewrites tests/golden/*.npz
"""
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

# ---- shims -------------------------------------------------------------------------
_pt = types.ModuleType("prettytable")


class _PT(object):
    def __init__(self, *a, **k):
        self.field_names, self.align = [], {}

    def add_row(self, *a, **k):
        pass


_pt.PrettyTable = _PT
sys.modules["prettytable"] = _pt
if not hasattr(np, "float_"):
    np.float_ = np.float64
_orig_mf = torch.Tensor.masked_fill_


def _mf(self, mask, value):
    if mask.dtype == torch.uint8:
        mask = mask.bool()
    return _orig_mf(self, mask, value)


torch.Tensor.masked_fill_ = _mf

from neuroir.rankers.esm import ESM  # noqa: E402
from neuroir.rankers.mtensor import MatchTensor  # noqa: E402
from neuroir.rankers import drmm as ref_drmm  # noqa: E402
from neuroir.rankers.duet import DUET  # noqa: E402
from neuroir.multitask.cars import CARS  # noqa: E402
from neuroir.models.ranker import Ranker  # noqa: E402
from neuroir.eval import ltorank  # noqa: E402
from neuroir import hyparam  # noqa: E402


class _NumpyProxy(object):
    """numpy >= 1.24 refuses the ragged (hist, edges) tuple from apply_along_axis
    (/root/reference/neuroir/rankers/drmm.py:71-75); return an object array of pairs instead."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def apply_along_axis(fn, axis, arr):
        assert axis == 2
        out = np.empty(arr.shape[:2] + (2,), dtype=object)
        for a in range(arr.shape[0]):
            for b in range(arr.shape[1]):
                h, e = fn(arr[a, b])
                out[a, b, 0], out[a, b, 1] = h, e
        return out


ref_drmm.numpy = _NumpyProxy()

from context_attentive_ir_amd.detinit import det_state_dict  # noqa: E402

SEED = 1013
V = 200
E = 300


def base_args(model, **kw):
    a = dict(emsize=E, src_vocab_size=V, dropout_emb=0.2, dropout=0.2, dropout_rnn=0.2,
             max_doc_len=200, max_query_len=10, num_candidates=10, use_word=True,
             fix_embeddings=True, model_type=model)
    a.update(hyparam.get_model_specific_params(model, "arch"))
    a.update(kw)
    return Namespace(**a)


def load_det(model):
    sd = model.state_dict()
    model.load_state_dict(det_state_dict({k: v.shape for k, v in sd.items()}, SEED))
    model.eval()
    return model


def rand_ids(rng, shape, lens, lo=4, hi=V):
    ids = rng.integers(lo, hi, size=shape, dtype=np.int64)
    pos = np.arange(shape[-1])
    ids[pos >= np.asarray(lens)[..., None]] = 0
    return ids


def save(name, **arrs):
    meta = dict(torch_version=torch.__version__, numpy_version=np.__version__, seed=SEED, vocab=V, emsize=E)
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()},
                        **{"meta_" + k: np.asarray(str(v)) for k, v in meta.items()})
    print("wrote", name)


def T(x):
    return torch.from_numpy(np.asarray(x))


@torch.no_grad()
def gen_esm():
    rng = np.random.default_rng(1)
    B, N, QL, DL = 3, 5, 6, 16
    qlen = np.array([6, 4, 3]); dlen = rng.integers(1, DL + 1, size=(B, N)); dlen[0, 0] = DL; dlen[1, 2] = 0
    q = rand_ids(rng, (B, QL), qlen); d = rand_ids(rng, (B, N, DL), dlen)
    dlen[1, 2] = 1  # all-PAD document (zero mean vector -> cosine 0); length field is unused by ESM
    m = load_det(ESM(base_args("ESM")))
    s = m(T(q), T(qlen), T(d), T(dlen))
    save("esm", que_rep=q, que_len=qlen, doc_rep=d, doc_len=dlen, scores=s, softmax=torch.softmax(s, -1))


@torch.no_grad()
def gen_match_tensor():
    rng = np.random.default_rng(2)
    B, N, QL, DL = 3, 4, 6, 20
    qlen = np.array([6, 3, 1]); dlen = rng.integers(1, DL + 1, size=(B, N)); dlen[0, 0] = DL; dlen[2, 1] = 1
    q = rand_ids(rng, (B, QL), qlen, hi=40); d = rand_ids(rng, (B, N, DL), dlen, hi=40)  # small range -> exact matches
    m = load_det(MatchTensor(base_args("MATCH_TENSOR")))
    tq, tql, td, tdl = T(q), T(qlen), T(d), T(dlen)
    s = m(tq, tql, td, tdl)
    # key intermediates (same ops as mtensor.py:76-100)
    eq = m.linear_projection(m.word_embeddings(tq.unsqueeze(2)))
    ed = m.linear_projection(m.word_embeddings(td.view(B * N, DL).unsqueeze(2)))
    _, hq = m.query_encoder(eq, tql)
    _, hd = m.document_encoder(ed, tdl.reshape(-1))
    save("match_tensor", que_rep=q, que_len=qlen, doc_rep=d, doc_len=dlen, scores=s,
         softmax=torch.softmax(s, -1), enc_q=hq, enc_d=hd, proj_q=m.query_projection(hq),
         proj_d=m.document_projection(hd))


@torch.no_grad()
def gen_match_tensor_general():
    """MATCH_TENSOR with the encoder configurations hyparam does not pin but the constructor admits (rnn_type GRU, stacked layers):
    the reference ranker end to end."""
    out = {}
    for tag, kw in (("gru2", dict(rnn_type="GRU", nlayers=2)), ("lstm2", dict(rnn_type="LSTM", nlayers=2)), ("gru1", dict(rnn_type="GRU", nlayers=1))):
        rng = np.random.default_rng(12)
        B, N, QL, DL = 3, 4, 6, 20
        qlen = np.array([6, 3, 1]); dlen = rng.integers(1, DL + 1, size=(B, N)); dlen[0, 0] = DL; dlen[2, 1] = 1
        q = rand_ids(rng, (B, QL), qlen, hi=40); d = rand_ids(rng, (B, N, DL), dlen, hi=40)
        m = load_det(MatchTensor(base_args("MATCH_TENSOR", **kw)))
        tq, tql, td, tdl = T(q), T(qlen), T(d), T(dlen)
        s = m(tq, tql, td, tdl)
        eq = m.linear_projection(m.word_embeddings(tq.unsqueeze(2)))
        ed = m.linear_projection(m.word_embeddings(td.view(B * N, DL).unsqueeze(2)))
        _, hq = m.query_encoder(eq, tql)
        _, hd = m.document_encoder(ed, tdl.reshape(-1))
        out.update({tag + ".que_rep": q, tag + ".que_len": qlen, tag + ".doc_rep": d, tag + ".doc_len": dlen, tag + ".scores": s,
                    tag + ".enc_q": hq, tag + ".enc_d": hd})
    save("match_tensor_general", **out)


@torch.no_grad()
def gen_drmm():
    rng = np.random.default_rng(3)
    B, N, QL, DL = 3, 4, 5, 24
    m = load_det(ref_drmm.DRMM(base_args("DRMM")))
    for tag, (qlo, qhi, dlo, dhi) in dict(safe=(4, 60, 60, V), overlap=(4, 50, 4, 50)).items():
        qlen = np.array([5, 3, 2]); dlen = rng.integers(1, DL + 1, size=(B, N)); dlen[0, 0] = DL
        q = rand_ids(rng, (B, QL), qlen, qlo, qhi); d = rand_ids(rng, (B, N, DL), dlen, dlo, dhi)
        tq, td = T(q), T(d)
        s = m(tq, T(qlen), td, T(dlen))
        # cosine + histogram intermediates, op-for-op as drmm.py:45-75
        eq = m.word_embeddings(tq.unsqueeze(2)); ed = m.word_embeddings(td.view(B * N, DL).unsqueeze(2))
        eqx = torch.stack([eq] * N, 1).view(B * N, QL, -1)
        cos = torch.nn.functional.cosine_similarity(torch.stack([eqx] * DL, 2), torch.stack([ed] * QL, 1), 3)
        hist = np.stack([[np.histogram(r, bins=m.bins)[0] for r in pair] for pair in cos.numpy()])
        edge = np.min(np.abs(cos.numpy()[..., None] - np.array([-1, -.5, 0, .5, 1.0])), -1)
        edge[cos.numpy() == 0] = 1.0  # exact zeros (PAD rows) are stable
        save("drmm_" + tag, que_rep=q, que_len=qlen, doc_rep=d, doc_len=dlen, scores=s, cos=cos, hist=hist,
             min_edge_dist=np.asarray(edge.min()), gate=m.gating_network(eq))


@torch.no_grad()
def gen_duet():
    rng = np.random.default_rng(4)
    B, N, QL, DL = 2, 3, 5, 24
    qlen = np.array([5, 3]); dlen = rng.integers(4, DL + 1, size=(B, N)); dlen[0, 0] = DL
    q = rand_ids(rng, (B, QL), qlen, hi=40); d = rand_ids(rng, (B, N, DL), dlen, hi=40)
    m = load_det(DUET(base_args("DUET", max_doc_len=DL, max_query_len=QL)))
    tq, td = T(q), T(d)
    s = m(tq, T(qlen), td, T(dlen))
    loc = m.local_model(tq, td)
    save("duet", que_rep=q, que_len=qlen, doc_rep=d, doc_len=dlen, scores=s, local=loc, dist=s - loc)


@torch.no_grad()
def gen_cars():
    rng = np.random.default_rng(5)
    B, S, N, QL, DL = 2, 3, 4, 5, 12
    qlen = rng.integers(1, QL + 1, size=(B, S)); qlen[0, 0] = QL
    dlen = rng.integers(1, DL + 1, size=(B, S, N)); dlen[0, 0, 0] = DL
    q = rand_ids(rng, (B, S, QL), qlen); d = rand_ids(rng, (B, S, N, DL), dlen)
    for tag in ("oneclick", "multiclick"):
        lab = np.zeros((B, S, N), np.float32)
        for b in range(B):
            for s_ in range(S):
                k = 1 if tag == "oneclick" else int(rng.integers(1, 4))
                lab[b, s_, rng.choice(N, k, replace=False)] = 1.0
        m = load_det(CARS(base_args("CARS", tgt_vocab_size=V)))
        tq, tql, td, tdl, tl = T(q), T(qlen), T(d), T(dlen), T(lab)
        pooled, enc_q, _ = m.encode(tq, tql)
        pooled_docs = m.encode_document(td, tdl)
        clicks = m.encode_clicks(pooled_docs, tl)
        scores, _, _ = m.rank_document(pooled, td, tdl, tl)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, tl)
        save("cars_" + tag, source_words=q, source_lens=qlen, document_words=d, document_lens=dlen,
             document_labels=lab, pooled_q=pooled, enc_q=enc_q, pooled_docs=pooled_docs,
             encoded_clicks=clicks, click_scores=scores, softmax=torch.softmax(scores, -1), ranking_loss=loss)


def gen_train(model="MATCH_TENSOR", fixture="match_tensor_train", seed=23, final_keys=("output.weight", "conv.weight", "linear_projection.weight",
                                                                                      "document_encoder.rnns.0.weight_hh_l0"), disjoint=False, fix_embeddings=True):
    """Training step of the real reference (models/ranker.py:192-230): dropout 0, Adam lr 1e-3, grad clipping 10,
    5 updates alternating over two batches -> loss trajectory; gradients of the first backward (before clipping)."""
    rng = np.random.default_rng(seed)
    B, N, QL, DL = 4, 3, 5, 11
    batches = []
    for _ in range(2):
        qlen = rng.integers(1, QL + 1, size=B); dlen = rng.integers(1, DL + 1, size=(B, N)); qlen[0] = QL; dlen[0, 0] = DL
        if disjoint:   # no shared tokens: no cosine lands on the cos == 1 histogram edge
            q = rand_ids(rng, (B, QL), qlen, 4, V // 2); d = rand_ids(rng, (B, N, DL), dlen, V // 2, V)
        else:
            q = rand_ids(rng, (B, QL), qlen); d = rand_ids(rng, (B, N, DL), dlen)
        lab = np.zeros((B, N), np.int64)
        lab[np.arange(B), rng.integers(0, N, size=B)] = 1
        batches.append(dict(que_rep=q, que_len=qlen, doc_rep=d, doc_len=dlen, label=lab))
    args = base_args(model, dropout_emb=0.0, dropout=0.0, dropout_rnn=0.0, optimizer="adam", learning_rate=0.001, weight_decay=0,
                     momentum=0, grad_clipping=10.0, fix_embeddings=fix_embeddings, max_query_len=QL, max_doc_len=DL)
    vocab = list(range(V))
    r = Ranker(args, vocab)
    load_det(r.network)
    r.init_optimizer()
    b0 = batches[0]
    r.network.train()
    s = r.network(T(b0["que_rep"]), T(b0["que_len"]), T(b0["doc_rep"]), T(b0["doc_len"]))
    loss0 = r.criterion(s, T(b0["label"]).float())
    r.optimizer.zero_grad()
    loss0.backward()
    out = {}
    for bi, b in enumerate(batches):
        for k, v in b.items():
            out["b%d_%s" % (bi, k)] = v
    for name, p in r.network.named_parameters():
        if p.grad is not None and p.numel() <= 20000:
            out["grad_" + name] = p.grad.detach().clone()
        elif p.grad is not None:      # large tensors: every 37th element + the norm (keeps the fixture small)
            out["gradsub37_" + name] = p.grad.detach().flatten()[::37].clone()
            out["gradnorm_" + name] = p.grad.detach().norm()
    out["scores0"], out["loss0"] = s.detach(), loss0.detach()
    r.optimizer.zero_grad()
    losses = []
    for step in range(5):
        b = batches[step % 2]
        losses.append(float(r.update({k: T(v) for k, v in b.items()})))
    out["losses"] = np.asarray(losses, np.float64)
    final = r.network.state_dict()
    for k in final_keys:
        out["final_" + k] = final[k].detach().clone() if final[k].numel() <= 20000 else final[k].detach().flatten()[::37].clone()
    save(fixture, **out)


def gen_duet_train():
    gen_train("DUET", "duet_train", 31, ("local_model.conv1d.weight", "local_model.fc3.weight", "distributed_model.conv_d1.weight",
                                         "distributed_model.fc2.weight", "distributed_model.fc4.bias"))


def gen_drmm_train():
    gen_train("DRMM", "drmm_train", 37, ("gating_network.weight.weight", "ffnn.0.weight", "ffnn.1.bias", "output.weight"), disjoint=True)


def gen_drmm_train_free():
    """config.py:94 default fix_embeddings=False: the embedding table trains through the gating network (the histograms are constants,
    rankers/drmm.py:70-75)."""
    gen_train("DRMM", "drmm_train_free", 41, ("gating_network.weight.weight", "ffnn.0.weight", "output.weight", "word_embeddings.make_embedding.emb_luts.0.weight"),
              disjoint=True, fix_embeddings=False)


def gen_cars_train():
    """Training step of the real reference Multitask (models/multitask.py:161-223) for CARS, all dropouts 0: the three losses and
    selected gradients of the first backward, then the total-loss trajectory of 4 Adam updates alternating over two batches."""
    from neuroir.models.multitask import Multitask
    rng = np.random.default_rng(29)
    B, S, N, QL, DL, TL = 2, 3, 4, 5, 9, 6
    batches = []
    for _ in range(2):
        qlen = rng.integers(1, QL + 1, size=(B, S)); qlen[0, 0] = QL
        dlen = rng.integers(1, DL + 1, size=(B, S, N)); dlen[0, 0, 0] = DL
        tlen = rng.integers(2, TL + 1, size=(B, S - 1)); tlen[0, 0] = TL
        q = rand_ids(rng, (B, S, QL), qlen); d = rand_ids(rng, (B, S, N, DL), dlen)
        tw = rand_ids(rng, (B, S - 1, TL), tlen); ts = rand_ids(rng, (B, S - 1, TL), tlen)
        lab = np.zeros((B, S, N), np.float32)
        for b in range(B):
            for s_ in range(S):
                lab[b, s_, rng.choice(N, int(rng.integers(1, 3)), replace=False)] = 1.0
        batches.append(dict(source_words=q, source_lens=qlen, document_words=d, document_lens=dlen, document_labels=lab,
                            target_words=tw, target_seq=ts, target_lens=tlen))
    args = base_args("CARS", tgt_vocab_size=V, dropout_emb=0.0, dropout=0.0, dropout_rnn=0.0, optimizer="adam", learning_rate=0.001,
                     weight_decay=0, momentum=0, grad_clipping=10.0, fix_embeddings=True)
    vocab = list(range(V))
    m = Multitask(args, vocab, vocab)
    load_det(m.network)
    m.init_optimizer()
    m.network.train()
    b0 = {k: T(v) for k, v in batches[0].items()}
    loss = m.network(source_rep=b0["source_words"], source_len=b0["source_lens"], target_rep=b0["target_words"], target_len=b0["target_lens"],
                     target_seq=b0["target_seq"], document_rep=b0["document_words"], document_len=b0["document_lens"],
                     document_label=b0["document_labels"])
    total = (1 - args.alpha) * loss["ranking_loss"] + args.alpha * loss["suggestion_loss"] + loss["regularization"]
    m.optimizer.zero_grad()
    total.backward()
    out = {}
    for bi, b in enumerate(batches):
        for k, v in b.items():
            out["b%d_%s" % (bi, k)] = v
    out["ranking_loss"], out["suggestion_loss"], out["regularization"], out["total_loss"] = (
        loss["ranking_loss"].detach(), loss["suggestion_loss"].detach(), loss["regularization"].detach(), total.detach())
    keep = ("q_attn.3.weight", "d_attn.0.bias", "click_attn.3.weight", "session_query_attn.weight", "session_doc_attn.bias", "ranknet._linear_layers.2.weight",
            "ranknet._linear_layers.1.bias", "q_projection.linear.bias", "transform_hid.linear.bias", "transform_cell.linear.bias", "dec_attn.weight",
            "decoder.decoder.attn.linear_in.weight", "decoder.decoder.rnn.bias_hh_l0", "token_prob_predictor1.weight", "session_query_inner_attn.3.weight",
            "session_doc_inner_attn.0.bias", "query_encoder.encoder.rnns.0.bias_ih_l0", "document_encoder.encoder.rnns.0.bias_hh_l0_reverse",
            "session_query_encoder.encoder.rnns.0.bias_ih_l0", "session_doc_encoder.encoder.rnns.0.bias_hh_l0", "shared_session_projector.linear.weight")
    norms = {}
    for name, p in m.network.named_parameters():
        if p.grad is not None:
            norms[name] = float(p.grad.norm())
            if name in keep and p.numel() <= 20000:          # small fixtures: big matrices are pinned through their norms
                out["grad_" + name] = p.grad.detach().clone()
    out["grad_norm_names"] = np.asarray(sorted(norms))
    out["grad_norms"] = np.asarray([norms[k] for k in sorted(norms)], np.float64)
    m.optimizer.zero_grad()
    losses = []
    for step in range(4):
        b = {k: T(v) for k, v in batches[step % 2].items()}
        losses.append(float(m.update(b)["total_loss"]))
    out["losses"] = np.asarray(losses, np.float64)
    save("cars_train", **out)


def gen_session_train(model, fixture, keep, seed):
    """Training step of the real reference Multitask (models/multitask.py:161-223) for M_MATCH_TENSOR / MNSRF, all dropouts 0: both losses
    and selected gradients of the first backward (all gradient norms), then the total-loss trajectory of 4 Adam updates over two batches."""
    from neuroir.models.multitask import Multitask
    rng = np.random.default_rng(seed)
    B, S, N, QL, DL, TL = 2, 3, 3, 5, 8, 6
    batches = []
    for _ in range(2):
        qlen = rng.integers(1, QL + 1, size=(B, S)); qlen[0, 0] = QL
        dlen = rng.integers(1, DL + 1, size=(B, S, N)); dlen[0, 0, 0] = DL
        tlen = rng.integers(2, TL + 1, size=(B, S - 1)); tlen[0, 0] = TL
        q = rand_ids(rng, (B, S, QL), qlen); d = rand_ids(rng, (B, S, N, DL), dlen)
        tw = rand_ids(rng, (B, S - 1, TL), tlen, hi=50); ts = rand_ids(rng, (B, S - 1, TL), tlen, hi=50)
        lab = np.zeros((B, S, N), np.float32)
        for b in range(B):
            for s_ in range(S):
                lab[b, s_, rng.choice(N, int(rng.integers(1, 3)), replace=False)] = 1.0
        batches.append(dict(source_words=q, source_lens=qlen, document_words=d, document_lens=dlen, document_labels=lab,
                            target_words=tw, target_seq=ts, target_lens=tlen))
    args = base_args(model, tgt_vocab_size=50, dropout_emb=0.0, dropout=0.0, dropout_rnn=0.0, optimizer="adam", learning_rate=0.001,
                     weight_decay=0, momentum=0, grad_clipping=10.0, fix_embeddings=True)
    m = Multitask(args, list(range(V)), list(range(50)))
    from context_attentive_ir_amd.detinit import fill_module_
    fill_module_(m.network, SEED)
    m.init_optimizer()
    m.network.train()
    b0 = {k: T(v) for k, v in batches[0].items()}
    loss = m.network(source_rep=b0["source_words"], source_len=b0["source_lens"], target_rep=b0["target_words"], target_len=b0["target_lens"],
                     target_seq=b0["target_seq"], document_rep=b0["document_words"], document_len=b0["document_lens"],
                     document_label=b0["document_labels"])
    total = (1 - args.alpha) * loss["ranking_loss"] + args.alpha * loss["suggestion_loss"]
    m.optimizer.zero_grad()
    total.backward()
    out = {}
    for bi, b in enumerate(batches):
        for k, v in b.items():
            out["b%d_%s" % (bi, k)] = v
    out["ranking_loss"], out["suggestion_loss"], out["total_loss"] = loss["ranking_loss"].detach(), loss["suggestion_loss"].detach(), total.detach()
    norms = {}
    for name, p in m.network.named_parameters():
        if p.grad is not None:
            norms[name] = float(p.grad.norm())
            if name in keep and p.numel() <= 20000:
                out["grad_" + name] = p.grad.detach().clone()
    out["grad_norm_names"] = np.asarray(sorted(norms))
    out["grad_norms"] = np.asarray([norms[k] for k in sorted(norms)], np.float64)
    m.optimizer.zero_grad()
    losses = []
    for step in range(4):
        b = {k: T(v) for k, v in batches[step % 2].items()}
        losses.append(float(m.update(b)["total_loss"]))
    out["losses"] = np.asarray(losses, np.float64)
    out["alpha"] = float(args.alpha)
    save(fixture, **out)


def gen_mmt_train():
    gen_session_train("M_MATCH_TENSOR", "m_match_tensor_train", ("output.weight", "conv.weight", "query_projection.bias", "generator.bias",
                                                                  "decoder.decoder.rnn.bias_hh_l0", "session_query_encoder.encoder.rnns.0.bias_ih_l0",
                                                                  "query_encoder.encoder.rnns.0.bias_ih_l0", "exact_match_channel.alpha"), 61)


def gen_mnsrf_train():
    gen_session_train("MNSRF", "mnsrf_train", ("projection.linear.bias", "generator.bias", "decoder.decoder.rnn.bias_hh_l0",
                                               "session_query_encoder.encoder.rnns.0.bias_ih_l0", "query_encoder.encoder.rnns.0.bias_ih_l0",
                                               "document_encoder.encoder.rnns.0.bias_hh_l0_reverse"), 67)


@torch.no_grad()
def gen_cars_decode():
    """CARS suggestion side + session switches from the real reference: decoder-initialisation states, inner-attention
    pools, greedy decode predictions (cars.py:382-456, 706-791); click scores with query_session_off / doc_session_off /
    both off (cars.py:185-188, 329-410, 485-533)."""
    rng = np.random.default_rng(17)
    B, S, N, QL, DL, MAXLEN = 3, 4, 5, 6, 9, 5
    qlen = rng.integers(1, QL + 1, size=(B, S)); qlen[0, 0] = QL
    dlen = rng.integers(1, DL + 1, size=(B, S, N)); dlen[0, 0, 0] = DL
    q = rand_ids(rng, (B, S, QL), qlen); d = rand_ids(rng, (B, S, N, DL), dlen)
    lab = np.zeros((B, S, N), np.float32)
    for b in range(B):
        for s_ in range(S):
            lab[b, s_, rng.choice(N, int(rng.integers(1, 3)), replace=False)] = 1.0
    tq, tql, td, tdl, tl = T(q), T(qlen), T(d), T(dlen), T(lab)
    tgt2src = rng.permutation(V).astype(np.int64)           # src_dict[tgt_dict[i]]: tgt_dict = identity, src_dict = a permutation
    tgt_dict, src_dict = list(range(V)), [int(x) for x in tgt2src]
    out = dict(source_words=q, source_lens=qlen, document_words=d, document_lens=dlen, document_labels=lab, tgt2src=tgt2src,
               max_len=MAXLEN)
    for tag, kw in (("full", {}), ("qoff", dict(query_session_off=True)), ("doff", dict(doc_session_off=True))):
        m = load_det(CARS(base_args("CARS", tgt_vocab_size=V, **kw)))
        pooled, enc_q, _ = m.encode(tq, tql)
        scores, states, attns = m.rank_document(pooled, td, tdl, tl)
        dec = m.decode(states=states, max_len=MAXLEN, src_dict=src_dict, tgt_dict=tgt_dict, batch_size=B, session_len=S - 1,
                       use_cuda=False, encoded_source=enc_q, source_len=tql, session_attns=attns)
        out[tag + "_click_scores"] = scores
        out[tag + "_dec_h"], out[tag + "_dec_c"] = states
        if attns[0] is not None:
            out[tag + "_inner_q"] = attns[0]
        if attns[1] is not None:
            out[tag + "_inner_d"] = attns[1]
        out[tag + "_predictions"] = dec["predictions"]
    m = load_det(CARS(base_args("CARS", tgt_vocab_size=V, query_session_off=True, doc_session_off=True, turn_recommender_off=True)))
    pooled, _, _ = m.encode(tq, tql)
    out["bothoff_click_scores"] = m.rank_document(pooled, td, tdl, tl)[0]
    save("cars_decode", **out)


@torch.no_grad()
def gen_losses_metrics():
    rng = np.random.default_rng(6)
    B, N = 6, 10
    s = T(rng.normal(size=(B, N)).astype(np.float32))
    lab = np.zeros((B, N), np.int64)
    for b in range(B):
        lab[b, rng.choice(N, int(rng.integers(1, 4)), replace=False)] = 1
    y = T(lab).float()
    bce = torch.nn.BCEWithLogitsLoss()(s, y)                      # models/ranker.py:55-69
    nll = Ranker.compute_loss(s, y)                               # models/ranker.py:79-89
    sm = torch.softmax(s, -1)                                     # models/ranker.py:258
    pred = np.argsort(-sm.numpy(), kind="stable")                 # main/ranker.py:257 (stable for determinism)
    save("losses_metrics", scores=s, labels=lab, bce=bce, softmax_nll=nll, softmax=sm, predictions=pred,
         MAP=np.asarray(ltorank.MAP(pred, lab)), MRR=np.asarray(ltorank.MRR(pred, lab)),
         P1=np.asarray(ltorank.precision_at_k(pred, lab, 1)), P3=np.asarray(ltorank.precision_at_k(pred, lab, 3)),
         R3=np.asarray(ltorank.recall_at_k(pred, lab, 3)), NDCG3=np.asarray(ltorank.NDCG_at_k(pred, lab, 3)))


@torch.no_grad()
def gen_m_match_tensor():
    """Session-aware MatchTensor (multitask/mmtensor.py): encode -> rank_document, eval mode."""
    from neuroir.multitask.mmtensor import M_MATCH_TENSOR
    args = base_args("M_MATCH_TENSOR", tgt_vocab_size=50)
    net = M_MATCH_TENSOR(args).eval()
    from context_attentive_ir_amd.detinit import fill_module_
    fill_module_(net, SEED)
    rng = np.random.default_rng(SEED + 21)
    B, S, N, QL, DL = 2, 3, 4, 5, 17
    slen = rng.integers(1, QL + 1, size=(B, S)); slen[0, 0] = QL
    dlen = rng.integers(1, DL + 1, size=(B, S, N)); dlen[0, 0, 0] = DL
    src = rand_ids(rng, (B, S, QL), slen); docs = rand_ids(rng, (B, S, N, DL), dlen)
    docs[0, 0, 1, :3] = src[0, 0, :3]                              # some exact matches
    pq, session_bank, states = net.encode(T(src), T(slen))
    scores = net.rank_document(T(src), pq, session_bank, T(docs), T(dlen))
    # suggestion side (mmtensor.py:94-125, 281-325): decoder-initialisation states and the greedy decode
    tgt2src = rng.permutation(V)[:50].astype(np.int64)
    tgt_dict, src_dict = list(range(50)), [int(x) for x in tgt2src]
    dec = net.decode(states=states, max_len=6, src_dict=src_dict, tgt_dict=tgt_dict, batch_size=B, session_len=S - 1, use_cuda=False)
    save("m_match_tensor", source_words=src, source_lens=slen, document_words=docs, document_lens=dlen,
         projected_queries=pq, scores=scores, softmax=torch.softmax(scores, -1), tgt_vocab_size=50, session_bank=session_bank,
         dec_h=states[0], dec_c=states[1], tgt2src=tgt2src, max_len=6, predictions=dec["predictions"])


@torch.no_grad()
def gen_mnsrf():
    """MNSRF (multitask/mnsrf.py): encode -> rank_document, eval mode."""
    from neuroir.multitask.mnsrf import MNSRF
    from context_attentive_ir_amd.detinit import fill_module_
    args = base_args("MNSRF", tgt_vocab_size=50)
    net = fill_module_(MNSRF(args).eval(), SEED)
    rng = np.random.default_rng(SEED + 33)
    B, S, N, QL, DL = 2, 3, 3, 4, 9
    slen = rng.integers(1, QL + 1, size=(B, S)); slen[0, 0] = QL
    dlen = rng.integers(1, DL + 1, size=(B, S, N)); dlen[0, 0, 0] = DL
    src = rand_ids(rng, (B, S, QL), slen); docs = rand_ids(rng, (B, S, N, DL), dlen)
    mem, sess, states = net.encode(T(src), T(slen))
    scores = net.rank_document(T(src), mem, sess, T(docs), T(dlen))
    tgt2src = rng.permutation(V)[:50].astype(np.int64)               # suggestion side (mnsrf.py:88-112, 251-296)
    tgt_dict, src_dict = list(range(50)), [int(x) for x in tgt2src]
    dec = net.decode(states=states, max_len=6, src_dict=src_dict, tgt_dict=tgt_dict, batch_size=B, session_len=S - 1, use_cuda=False)
    save("mnsrf", source_words=src, source_lens=slen, document_words=docs, document_lens=dlen, memory_bank=mem,
         session_bank=sess, scores=scores, softmax=torch.softmax(scores, -1), tgt_vocab_size=50, dec_h=states[0], dec_c=states[1],
         tgt2src=tgt2src, max_len=6, predictions=dec["predictions"])


@torch.no_grad()
def gen_rnn_encoder():
    """The reference RNNEncoder itself (encoders/rnn_encoder.py) outside the hyparam-pinned 1-layer LSTM: GRU, stacked layers, bridge,
    use_last = False, initial states.  Final states are saved as the reference returns them (length-SORTED batch order) together with
    the sort indices; lengths are distinct so that the (unstable) sort is unambiguous."""
    from neuroir.encoders.rnn_encoder import RNNEncoder
    from context_attentive_ir_amd.detinit import fill_module_
    rng = np.random.default_rng(SEED + 55)
    out = {}
    cfgs = dict(gru2_bi_bridge=dict(rnn_type="GRU", input_size=10, bidirectional=True, num_layers=2, hidden_size=24, use_bridge=True, use_last=False),
                lstm2_uni=dict(rnn_type="LSTM", input_size=10, bidirectional=False, num_layers=2, hidden_size=16, use_bridge=False, use_last=True),
                lstm2_bi_cat=dict(rnn_type="LSTM", input_size=12, bidirectional=True, num_layers=2, hidden_size=20, use_bridge=True, use_last=False),
                # (a GRU with initial states is not a working configuration of the reference: `if init_states:` on a tensor raises, :77)
                lstm1_uni_init=dict(rnn_type="LSTM", input_size=8, bidirectional=False, num_layers=1, hidden_size=12, use_bridge=False, use_last=True))
    for name, kw in cfgs.items():
        enc = fill_module_(RNNEncoder(dropout=0.0, **kw).eval(), SEED + len(name))
        M, Tn = 4, 7
        x = T(rng.normal(size=(M, Tn, kw["input_size"])).astype(np.float32))
        lens = T(np.array([5, 7, 1, 3], dtype=np.int64))
        init = None
        if name == "lstm1_uni_init":
            init = (T(rng.normal(size=(1, M, 12)).astype(np.float32) * 0.5), T(rng.normal(size=(1, M, 12)).astype(np.float32) * 0.5))
            lens = None                                           # (with lengths the reference would feed the unsorted states to sorted rows)
        final, bank = enc(x, lens, init)
        out[name + ".x"] = x
        out[name + ".lens"] = lens if lens is not None else np.zeros(0, np.int64)
        out[name + ".bank"] = bank
        if init is not None:
            out[name + ".init_h"], out[name + ".init_c"] = init
        if isinstance(final, tuple):
            out[name + ".h"], out[name + ".c"] = final
        else:
            out[name + ".h"] = final
        if lens is not None:
            out[name + ".sort_idx"] = torch.sort(lens, 0, True)[1]
        for k, v in enc.state_dict().items():
            out[name + ".sd." + k] = v
    save("rnn_encoder", **out)


def gen_batchify():
    """Input contract: the reference's own collate functions on ragged synthetic examples."""
    from neuroir.inputters.ranker.vector import batchify as ranker_batchify
    from neuroir.inputters.multitask.vector import batchify as session_batchify
    rng = np.random.default_rng(SEED + 77)
    B, N = 5, 4
    qlens = rng.integers(1, 7, size=B); dlens = rng.integers(1, 23, size=(B, N))
    qflat = rng.integers(4, V, size=int(qlens.sum())); dflat = rng.integers(4, V, size=int(dlens.sum()))
    labels = rng.integers(0, 2, size=(B, N))
    batch, qo, do = [], 0, 0
    for b in range(B):
        q = qflat[qo:qo + qlens[b]]; qo += qlens[b]
        docs = []
        for n in range(N):
            docs.append(torch.LongTensor(dflat[do:do + dlens[b, n]])); do += dlens[b, n]
        batch.append({"id": b, "query_words": torch.LongTensor(q), "doc_words": docs, "label": torch.LongTensor(labels[b]),
                      "num_candidates": N, "max_doc_len": int(dlens[b].max()), "max_query_len": int(qlens[b])})
    out = ranker_batchify(batch)
    arrs = dict(r_qlens=qlens, r_dlens=dlens, r_qflat=qflat, r_dflat=dflat, r_labels=labels,
                **{"r_out_" + k: out[k] for k in ("doc_rep", "doc_len", "que_rep", "que_len", "label")})
    # sessions: S fixed per batch, per-session maxima differ
    Bs, S, Ns = 3, 4, 3
    sess = []
    for b in range(Bs):
        ql, dl, tl = int(rng.integers(2, 6)), int(rng.integers(3, 12)), int(rng.integers(2, 6))
        sl = rng.integers(1, ql + 1, size=S); sl[rng.integers(0, S)] = ql
        dls = rng.integers(1, dl + 1, size=(S, Ns)); dls[0, 0] = dl
        tls = rng.integers(1, tl + 1, size=S - 1); tls[0] = tl
        sw = rng.integers(4, V, size=(S, ql)); sw[np.arange(ql)[None] >= sl[:, None]] = 0
        dw = rng.integers(4, V, size=(S, Ns, dl)); dw[np.arange(dl)[None, None] >= dls[..., None]] = 0
        tw = rng.integers(4, V, size=(S - 1, tl)); tw[np.arange(tl)[None] >= tls[:, None]] = 0
        ts = rng.integers(4, V, size=(S - 1, tl)); ts[np.arange(tl)[None] >= tls[:, None]] = 0
        lab = rng.integers(0, 2, size=(S, Ns))
        sess.append({"id": b, "source_tokens": None, "target_tokens": None, "session_len": S, "num_candidates": Ns,
                     "source_words": torch.LongTensor(sw), "source_lens": torch.LongTensor(sl),
                     "target_words": torch.LongTensor(tw), "target_lens": torch.LongTensor(tls), "target_seq": torch.LongTensor(ts),
                     "document_words": torch.LongTensor(dw), "document_lens": torch.LongTensor(dls),
                     "document_labels": torch.LongTensor(lab), "max_source_len": ql, "max_target_len": tl,
                     "max_document_len": dl})
    sout = session_batchify(sess)
    keys = ("source_words", "source_lens", "target_words", "target_lens", "target_seq", "document_words", "document_lens",
            "document_labels")
    for b, ex in enumerate(sess):
        for k in keys:
            arrs["s_in%d_%s" % (b, k)] = ex[k]
    arrs.update({"s_out_" + k: sout[k] for k in keys})
    arrs["s_out_document_labels_is_float"] = np.asarray(sout["document_labels"].dtype == torch.float32)
    save("batchify", **arrs)


def gen_samplers():
    """Batch composition of the reference's length-bucketing samplers under a fixed numpy seed."""
    from neuroir.inputters.ranker.data import SortedBatchSampler as RankerSampler
    from neuroir.inputters.multitask.data import SortedBatchSampler as SessionSampler
    rng = np.random.default_rng(SEED + 5)
    lengths = np.stack([rng.integers(5, 40, size=57), rng.integers(1, 8, size=57)], 1)
    out = {"r_lengths": lengths}
    for shuffle in (False, True):
        np.random.seed(SEED)
        out["r_flat_shuffle%d" % shuffle] = np.asarray(list(RankerSampler([tuple(l) for l in lengths], 8, shuffle=shuffle)))
    slens = rng.integers(2, 6, size=61)
    out["s_lengths"] = slens
    for shuffle in (False, True):
        np.random.seed(SEED)
        out["s_flat_shuffle%d" % shuffle] = np.asarray(list(SessionSampler(list(slens), 4, shuffle=shuffle)))
    save("samplers", **out)


if __name__ == "__main__":
    torch.manual_seed(SEED)
    torch.set_num_threads(4)
    only = set(sys.argv[1:])          # e.g. `generate.py cars_decode` regenerates one fixture family
    gens = dict(esm=gen_esm, match_tensor=gen_match_tensor, drmm=gen_drmm, duet=gen_duet, cars=gen_cars, cars_decode=gen_cars_decode, train=gen_train, duet_train=gen_duet_train, drmm_train=gen_drmm_train, drmm_train_free=gen_drmm_train_free, cars_train=gen_cars_train,
                losses_metrics=gen_losses_metrics, batchify=gen_batchify, samplers=gen_samplers, m_match_tensor=gen_m_match_tensor,
                mnsrf=gen_mnsrf, rnn_encoder=gen_rnn_encoder, match_tensor_general=gen_match_tensor_general, mmt_train=gen_mmt_train, mnsrf_train=gen_mnsrf_train)
    for name, fn in gens.items():
        if not only or name in only:
            fn()
