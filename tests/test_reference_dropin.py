"""Drop-in check against the REAL reference wrappers (authoring container only: /root/reference is not on the GPU box).

The reference's own `neuroir.models.ranker.Ranker` / `neuroir.models.multitask.Multitask` are constructed with this package's
network classes monkey-patched in (the two import lines a maintainer changes) and everything the
drivers main/ranker.py / main/multitask.py do with a model object is exercised up to the kernel launch: construction from
`config.get_model_args`, state-dict key identity with the reference's own networks, load_embeddings, init_optimizer
(fix_embeddings freezes `network.word_embeddings` / `network.embedder.word_embeddings`), save -> load, checkpoint ->
load_checkpoint.  predict/update on CPU tensors must fail loudly (no CPU fallback)."""
import os
import sys
import types
from argparse import Namespace

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the authoring container")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, REF)
    pt = types.ModuleType("prettytable")

    class _PT(object):
        def __init__(self, *a, **k):
            self.field_names, self.align = [], {}

        def add_row(self, *a, **k):
            pass

    pt.PrettyTable = _PT
    sys.modules.setdefault("prettytable", pt)
    if not hasattr(np, "float_"):
        np.float_ = np.float64
    orig_load = torch.load                        # SURVEY Appendix D shim 5: the reference pickles Namespace / Vocabulary objects

    def _load(*a, **k):
        k.setdefault("weights_only", False)
        return orig_load(*a, **k)

    torch.load = _load
    import neuroir.models.multitask as rm
    import neuroir.models.ranker as rr
    from neuroir import hyparam
    from neuroir.inputters.vocabulary import Vocabulary
    yield types.SimpleNamespace(rr=rr, rm=rm, hyparam=hyparam, Vocabulary=Vocabulary)
    torch.load = orig_load
    sys.path.remove(REF)


def _vocab(ref, n):
    v = ref.Vocabulary()
    for i in range(n):
        v.add("tok%d" % i)
    return v


def _args(ref, model, **kw):
    a = dict(emsize=300, dropout_emb=0.2, dropout=0.2, dropout_rnn=0.2, max_doc_len=20, max_query_len=6, num_candidates=4,
             use_word=True, fix_embeddings=True, model_type=model, optimizer="adam", learning_rate=0.001, weight_decay=0,
             momentum=0, grad_clipping=10)
    a.update(ref.hyparam.get_model_specific_params(model, "arch"))
    a.update(kw)
    return Namespace(**a)


@pytest.mark.parametrize("model,attr", [("ESM", "ESM"), ("MATCH_TENSOR", "MatchTensor"), ("DRMM", "DRMM"), ("DUET", "DUET")])
def test_reference_ranker_with_swapped_network(ref, model, attr, monkeypatch, tmp_path):
    from context_attentive_ir_amd import rankers
    vocab = _vocab(ref, 40)
    own = ref.rr.Ranker(_args(ref, model), vocab)                       # the reference's own network: the key set to match
    ref_keys = {k: tuple(v.shape) for k, v in own.network.state_dict().items()}
    monkeypatch.setattr(ref.rr, attr, getattr(rankers, attr))
    m = ref.rr.Ranker(_args(ref, model), vocab)
    assert type(m.network).__module__.startswith("context_attentive_ir_amd")
    assert {k: tuple(v.shape) for k, v in m.network.state_dict().items()} == ref_keys
    assert m.count_parameters() == own.count_parameters()
    # load_embeddings -> init_word_vectors on the swapped network's word_embeddings
    emb = tmp_path / "emb.txt"
    emb.write_text("\n".join("tok%d %s" % (i, " ".join(["%.3f" % (0.01 * i)] * 300)) for i in range(5)))
    m.load_embeddings(["tok%d" % i for i in range(5)], str(emb))
    row = m.network.word_embeddings.table[vocab["tok3"]]
    assert torch.allclose(row, torch.full((300,), 0.03), atol=1e-6)
    if model != "ESM":
        m.init_optimizer()
        assert all(not p.requires_grad for p in m.network.word_embeddings.parameters())
        m.checkpoint(str(tmp_path / "c.mdl"), 3)
        m2, epoch = ref.rr.Ranker.load_checkpoint(str(tmp_path / "c.mdl"), use_gpu=False)
        assert epoch == 3 and type(m2.network) is type(m.network)
    m.save(str(tmp_path / "m.mdl"))
    m3 = ref.rr.Ranker.load(str(tmp_path / "m.mdl"))
    for k, v in m.network.state_dict().items():
        assert torch.equal(v, m3.network.state_dict()[k])
    own.network.load_state_dict(m.network.state_dict())                 # and the reference's own class loads our weights strictly
    ex = dict(que_rep=torch.zeros(2, 6, dtype=torch.long), que_len=torch.full((2,), 6), doc_rep=torch.zeros(2, 4, 20, dtype=torch.long),
              doc_len=torch.full((2, 4), 20), label=torch.zeros(2, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.predict(ex)


@pytest.mark.parametrize("kw", [{}, dict(query_session_off=True), dict(doc_session_off=True)])
def test_reference_multitask_with_swapped_cars(ref, kw, monkeypatch, tmp_path):
    from context_attentive_ir_amd.multitask import CARS
    src, tgt = _vocab(ref, 40), _vocab(ref, 30)
    own = ref.rm.Multitask(_args(ref, "CARS", **kw), src, tgt)
    ref_keys = {k: tuple(v.shape) for k, v in own.network.state_dict().items()}
    monkeypatch.setattr(ref.rm, "CARS", CARS)
    m = ref.rm.Multitask(_args(ref, "CARS", **kw), src, tgt)
    assert type(m.network) is CARS
    assert {k: tuple(v.shape) for k, v in m.network.state_dict().items()} == ref_keys
    m.init_optimizer()
    assert all(not p.requires_grad for p in m.network.embedder.word_embeddings.parameters())
    m.save(str(tmp_path / "m.mdl"))
    m2 = ref.rm.Multitask.load(str(tmp_path / "m.mdl"))
    assert type(m2.network) is CARS
    own.network.load_state_dict(m.network.state_dict())
    # every attribute / method Multitask.predict touches on the network exists with the reference's signature
    for name in ("encode", "rank_document", "decode", "forward", "embedder"):
        assert hasattr(m.network, name)
    ex = dict(source_words=torch.zeros(2, 3, 6, dtype=torch.long), source_lens=torch.full((2, 3), 6),
              document_words=torch.zeros(2, 3, 4, 20, dtype=torch.long), document_lens=torch.full((2, 3, 4), 20),
              document_labels=torch.zeros(2, 3, 4), session_len=3, ids=["a", "b"], batch_size=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.predict(ex)


def test_reference_cars_itself_fails_with_unequal_encoder_sizes(ref, monkeypatch):
    """`multitask/cars.py` of this package refuses nhid_query != nhid_document.  That mirrors the reference: its session loop multiplies
    session_doc_attn(states) [B, k, nhid_document] with the pooled QUERY [B, nhid_query, 1] (cars.py:357-360), so with the document-session
    encoder and the ranker on (every hyparam configuration) its own forward raises for unequal sizes -- reproduced here against the real class."""
    from neuroir.multitask.cars import CARS as RefCARS
    from context_attentive_ir_amd import synth
    orig_mf = torch.Tensor.masked_fill_                       # SURVEY Appendix D shim: the reference masks with uint8 tensors
    monkeypatch.setattr(torch.Tensor, "masked_fill_", lambda self, mask, value: orig_mf(self, mask.bool() if mask.dtype == torch.uint8 else mask, value))
    from context_attentive_ir_amd.multitask import CARS
    kw = dict(src_vocab_size=60, tgt_vocab_size=60, nhid_query=16, nhid_document=24, nhid_session_query=8, nhid_session_document=8, nhid_decoder=8,
              emsize=12, max_query_len=4)
    a = _args(ref, "CARS", **kw)
    net = RefCARS(a)
    net.eval()
    ex = synth.session_batch(2, 3, 3, 4, 6, 60, seed=1)
    tgt = torch.ones(2, 2, 5, dtype=torch.long)                    # [B, S-1, TL]: the next query of every step but the last
    call = lambda n: n(ex["source_words"], ex["source_lens"], tgt, torch.full((2, 2), 5), tgt, ex["document_words"], ex["document_lens"],      # noqa: E731
                       ex["document_labels"])
    with pytest.raises(RuntimeError, match="batch2"):                     # the bmm of cars.py:361 inside the reference's encode_session
        call(net)
    ok = RefCARS(_args(ref, "CARS", **dict(kw, nhid_document=16)))       # the same inputs with equal sizes go through: the failure is the size mismatch
    ok.eval()
    assert torch.isfinite(call(ok)["ranking_loss"])
    with pytest.raises(NotImplementedError, match="nhid_query == nhid_document"):
        CARS(a)


def test_predict_text_outputs_equal_the_reference_tail(ref):
    """wrappers.Multitask._suggestion_text (the host tail of predict for a batch in the reference's collate layout) against the reference's own
    code: `tens2sen` over the decoder's ids per step, `ex_ids`, `targets`, `src_sequences` exactly as models/multitask.py:294-316 builds them --
    what main/multitask.py:validate_official reads from `outputs`."""
    from neuroir.utils.misc import tens2sen
    from context_attentive_ir_amd.config import default_args
    from context_attentive_ir_amd.wrappers import Multitask
    tgt = _vocab(ref, 30)
    w = Multitask(default_args("CARS", src_vocab_size=40, tgt_vocab_size=len(tgt)), tgt_dict=tgt)
    B, S, ML = 3, 4, 6
    g = torch.Generator().manual_seed(5)
    pred = torch.randint(0, len(tgt) + 3, (B, S - 1, ML), generator=g)           # incl. BOS / EOS / PAD and ids past the dictionary
    pred[0, 0, :] = 2                                                              # only BOS: an empty sentence
    pred[1, 1, 2] = 3                                                              # EOS in the middle
    src_tok = [[["<s>"] + ["q%d_%d_%d" % (b, s, j) for j in range(2 + s)] + ["</s>"] for s in range(S)] for b in range(B)]
    ex = {"ids": ["sess%d_" % b for b in range(B)], "batch_size": B, "session_len": S, "source_tokens": src_tok,
          "target_tokens": [[src_tok[b][s] for s in range(1, S)] for b in range(B)]}
    got = w._suggestion_text(ex, pred)
    # the reference's tail, verbatim semantics
    want_pred, want_tgt, want_src = [], [], []
    for sidx in range(S - 1):
        want_pred.extend(tens2sen(pred[:, sidx, :], tgt, None))
        for bidx in range(B):
            tokens = ex["target_tokens"][bidx][sidx]
            want_tgt.append([" ".join(tokens[1:-1])])
            want_src.append(" ".join(" ".join(q[1:-1]) for q in ex["source_tokens"][bidx][0:sidx + 1]))
    assert got["predictions"] == want_pred and got["targets"] == want_tgt and got["src_sequences"] == want_src
    assert got["ex_ids"] == [_id + str(i) for i in range(S) for _id in ex["ids"]]
    assert got["prediction_ids"] is pred and len(got["predictions"]) == B * (S - 1)
