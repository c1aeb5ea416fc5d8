"""GPU (-m gpu): the training step (autograd.py over csrc/train.hip) -- operator gradients against torch-CPU autograd, the
MatchTensor backward against the reference's own gradients, Ranker.update against the reference's 5-step loss trajectory
(tests/golden/match_tensor_train.npz), and a dropout run replayed through the oracle with the product's own masks.
Tolerance: 1e-4 relative to the largest entry of each gradient (north_star's 1e-4, applied to gradients)."""
import numpy as np
import pytest
import torch

from conftest import T, load_golden
from helpers import build_model, cpu_state_dict
from oracle import neuroir_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b, tol=1e-4):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    scale = max(float(np.abs(b).max()), 1e-5)     # (a mathematically-zero gradient is rounding noise ~1e-10 on both sides)
    err = float(np.abs(a - b).max()) / scale
    assert err <= tol, "relative error %.3g > %.3g (scale %.3g)" % (err, tol, scale)


@pytest.mark.parametrize("M,N,K,act", [(37, 40, 300, None), (130, 50, 30, "tanh"), (4100, 18, 1071, "relu"), (5, 1, 20, None),
                                       (20480, 560, 40, None), (13000, 300, 260, "tanh")])
def test_linear_backward(M, N, K, act):
    from context_attentive_ir_amd import autograd as A
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    y = torch.nn.functional.linear(xr, wr, br)
    y = torch.tanh(y) if act == "tanh" else torch.relu(y) if act == "relu" else y
    y.backward(dy)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    yd = A.linear(xd, wd, bd, act)
    yd.backward(dy.to(DEV))
    _rel(yd, y, 2e-5); _rel(xd.grad, xr.grad); _rel(wd.grad, wr.grad); _rel(bd.grad, br.grad)


@pytest.mark.parametrize("H,I,M,T_,bi", [(15, 40, 7, 6, True), (70, 40, 33, 20, True), (128, 300, 19, 12, True), (64, 256, 16, 7, False),
                                         (128, 64, 3, 64, True), (1, 4, 2, 3, True)])
def test_bilstm_backward(H, I, M, T_, bi):
    """Train-mode recurrence + BPTT against torch autograd through the oracle's pack/sort/nn.LSTM restatement."""
    from context_attentive_ir_amd import autograd as A
    from context_attentive_ir_amd.detinit import fill_module_
    nd = 2 if bi else 1
    lstm = fill_module_(torch.nn.LSTM(I, H, 1, bidirectional=bi, batch_first=True), seed=9)
    g = torch.Generator().manual_seed(H + M)
    x = torch.randn(M, T_, I, generator=g); lens = torch.randint(1, T_ + 1, (M,), generator=g); lens[0] = T_
    dout = torch.randn(M, T_, nd * H, generator=g)
    sd = {"e.rnns.0." + k: v.detach().clone().requires_grad_(True) for k, v in lstm.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    _, ref = O.rnn_encode(sd, "e", xr, lens, bidirectional=bi)
    ref.backward(dout)
    lstm = lstm.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    out = A.bilstm(xd, lens.to(DEV), lstm)
    out.backward(dout.to(DEV))
    _rel(out, ref, 2e-5); _rel(xd.grad, xr.grad)
    for k, p in lstm.named_parameters():
        _rel(p.grad, sd["e.rnns.0." + k].grad)


def test_embed_backward_skips_pad_and_accumulates():
    from context_attentive_ir_amd import autograd as A
    V, E = 30, 8
    table = torch.randn(V, E)
    ids = torch.tensor([[1, 2, 0, 2], [5, 0, 2, 1]])
    dout = torch.randn(2, 4, E)
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    emb.weight.data.copy_(table)
    emb(ids).backward(dout)
    td = table.to(DEV).requires_grad_(True)
    A.embed(ids.to(DEV), td).backward(dout.to(DEV))
    _rel(td.grad, emb.weight.grad, 1e-6)


def test_match_tensor_gradients_vs_reference():
    g = load_golden("match_tensor_train")
    from context_attentive_ir_amd import autograd as A
    m = build_model("MATCH_TENSOR", device=DEV, dropout_emb=0.0).train()
    m.word_embeddings.table.requires_grad_(False)
    q, ql, d, dl, lab = (T(g["b0_" + k], DEV) for k in ("que_rep", "que_len", "doc_rep", "doc_len", "label"))
    s = m(q, ql, d, dl)
    _rel(s, g["scores0"], 2e-5)
    loss = A.bce_with_logits(s, lab.float())
    _rel(loss, g["loss0"], 1e-5)
    loss.backward()
    for name, p in m.named_parameters():
        if p.requires_grad:
            _rel(p.grad, g["grad_" + name])


def test_ranker_update_matches_reference_loss_trajectory():
    g = load_golden("match_tensor_train")
    from context_attentive_ir_amd.config import default_args
    from context_attentive_ir_amd.detinit import fill_module_
    from context_attentive_ir_amd.wrappers import Ranker
    args = default_args("MATCH_TENSOR", src_vocab_size=int(g["meta_vocab"]), dropout_emb=0.0, dropout=0.0, dropout_rnn=0.0, optimizer="adam",
                        learning_rate=0.001, weight_decay=0, momentum=0, grad_clipping=10.0, fix_embeddings=True)
    r = Ranker(args)
    fill_module_(r.network, 1013)
    r.cuda()
    r.init_optimizer()
    losses = []
    for step in range(5):
        b = {k: T(g["b%d_%s" % (step % 2, k)]) for k in ("que_rep", "que_len", "doc_rep", "doc_len", "label")}
        losses.append(float(r.update(b)))
    np.testing.assert_allclose(np.asarray(losses), g["losses"], rtol=1e-4, atol=0)
    sd = r.network.state_dict()
    for k in ("output.weight", "conv.weight", "linear_projection.weight", "document_encoder.rnns.0.weight_hh_l0"):
        _rel(sd[k], g["final_" + k], 2e-4)
    assert r.updates == 5


def test_match_tensor_dropout_replayed_through_oracle():
    """Train mode with dropout 0.2: the product's own keep masks are replayed through the differentiable oracle."""
    from context_attentive_ir_amd import autograd as A
    from context_attentive_ir_amd import synth
    V = 400
    m = build_model("MATCH_TENSOR", vocab=V, device=DEV, dropout_emb=0.2).train()
    m.word_embeddings.table.requires_grad_(False)
    ex = synth.ranker_batch(3, 4, 5, 17, V, seed=3, full_length=False)
    A.DROPOUT.manual_seed(77)
    A.DROPOUT.record, A.DROPOUT.masks = True, []
    try:
        s = m(ex["que_rep"].to(DEV), ex["que_len"].to(DEV), ex["doc_rep"].to(DEV), ex["doc_len"].to(DEV))
    finally:
        A.DROPOUT.record = False
    masks = [k.cpu() for k in A.DROPOUT.masks]
    assert len(masks) == 2 and 0.7 < float(masks[1].float().mean()) < 0.9
    lab = ex["label"].float()
    A.bce_with_logits(s, lab.to(DEV)).backward()
    sd = {k: v.clone().requires_grad_(not k.startswith("word_embeddings")) for k, v in cpu_state_dict(m).items()}
    ref = O.match_tensor_train_scores(sd, ex["que_rep"], ex["que_len"], ex["doc_rep"], ex["doc_len"], masks=masks, p_drop=0.2)
    _rel(s, ref, 5e-5)
    O.bce_with_logits(ref, lab).backward()
    for name, p in m.named_parameters():
        if p.requires_grad:
            _rel(p.grad, sd[name].grad)
    # a second forward draws different masks
    s2 = m(ex["que_rep"].to(DEV), ex["que_len"].to(DEV), ex["doc_rep"].to(DEV), ex["doc_len"].to(DEV))
    assert float((s2 - s).abs().max()) > 1e-4


def _cars_train_batch(g, i, dev):
    keys = ("source_words", "source_lens", "document_words", "document_lens", "document_labels", "target_words", "target_seq", "target_lens")
    return {k: T(g["b%d_%s" % (i, k)], dev) for k in keys}


def test_cars_losses_and_gradients_vs_reference():
    """CARS train-mode forward (ranking + suggestion + regularisation) and its backward against the real reference
    (tests/golden/generate.py:gen_cars_train, all dropouts 0)."""
    g = load_golden("cars_train")
    m = build_model("CARS", vocab=int(g["meta_vocab"]), tgt_vocab_size=int(g["meta_vocab"]), device=DEV, dropout_emb=0.0, dropout=0.0,
                    dropout_rnn=0.0).train()
    m.embedder.word_embeddings.table.requires_grad_(False)
    b = _cars_train_batch(g, 0, DEV)
    loss = m(source_rep=b["source_words"], source_len=b["source_lens"], target_rep=b["target_words"], target_len=b["target_lens"],
             target_seq=b["target_seq"], document_rep=b["document_words"], document_len=b["document_lens"], document_label=b["document_labels"])
    _rel(loss["ranking_loss"], g["ranking_loss"], 2e-5); _rel(loss["suggestion_loss"], g["suggestion_loss"], 2e-5)
    _rel(loss["regularization"], g["regularization"], 1e-5)
    total = 0.9 * loss["ranking_loss"] + 0.1 * loss["suggestion_loss"] + loss["regularization"]
    _rel(total, g["total_loss"], 2e-5)
    total.backward()
    grads = dict(m.named_parameters())
    for k in g:
        if k.startswith("grad_") and k not in ("grad_norms", "grad_norm_names"):
            _rel(grads[k[5:]].grad, g[k])
    for name, ref in zip(g["grad_norm_names"], g["grad_norms"]):
        got = float(grads[str(name)].grad.norm())
        assert abs(got - ref) <= 1e-4 * max(ref, 1e-3), (name, got, ref)


def test_multitask_update_matches_reference_loss_trajectory():
    g = load_golden("cars_train")
    from context_attentive_ir_amd.config import default_args
    from context_attentive_ir_amd.detinit import fill_module_
    from context_attentive_ir_amd.wrappers import Multitask
    V = int(g["meta_vocab"])
    args = default_args("CARS", src_vocab_size=V, tgt_vocab_size=V, dropout_emb=0.0, dropout=0.0, dropout_rnn=0.0, optimizer="adam",
                        learning_rate=0.001, weight_decay=0, momentum=0, grad_clipping=10.0, fix_embeddings=True)
    mt = Multitask(args)
    fill_module_(mt.network, 1013)
    mt.cuda()
    mt.init_optimizer()
    losses = [float(mt.update(_cars_train_batch(g, step % 2, "cpu"))["total_loss"]) for step in range(4)]
    np.testing.assert_allclose(np.asarray(losses), g["losses"], rtol=1e-4, atol=0)
    # eval-mode inference still works after training (weights repacked, folded tables rebuilt)
    mt.network.eval()
    ex = _cars_train_batch(g, 0, "cpu")
    out = mt.predict(ex, suggest=False)["click_scores"]
    assert torch.isfinite(out).all()


def test_cars_train_with_dropout_runs_and_is_stochastic():
    from context_attentive_ir_amd import autograd as A
    g = load_golden("cars_train")
    m = build_model("CARS", vocab=int(g["meta_vocab"]), tgt_vocab_size=int(g["meta_vocab"]), device=DEV).train()
    b = _cars_train_batch(g, 0, DEV)
    kw = dict(source_rep=b["source_words"], source_len=b["source_lens"], target_rep=b["target_words"], target_len=b["target_lens"],
              target_seq=b["target_seq"], document_rep=b["document_words"], document_len=b["document_lens"], document_label=b["document_labels"])
    A.DROPOUT.manual_seed(5)
    l1 = m(**kw)
    l2 = m(**kw)
    A.DROPOUT.manual_seed(5)
    l3 = m(**kw)
    assert torch.isfinite(l1["ranking_loss"]) and torch.isfinite(l1["suggestion_loss"])
    assert float((l1["ranking_loss"] - l2["ranking_loss"]).abs()) > 1e-6          # different masks
    assert float((l1["ranking_loss"] - l3["ranking_loss"]).abs()) < 1e-6          # same seed stream -> same masks
    (0.9 * l1["ranking_loss"] + 0.1 * l1["suggestion_loss"]).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
