"""GPU (-m gpu): the session stream (inputters.session_stream + graph_runner.StreamingSessionPredictor): int32 wire block -> device widening ->
captured CARS ranking step per session length, against the CPU oracle on the same sessions."""
import numpy as np
import pytest
import torch

from helpers import cpu_state_dict
from oracle import neuroir_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_widen_ids_kernel():
    from context_attentive_ir_amd import lib
    g = torch.Generator().manual_seed(3)
    for n in (0, 1, 3, 4, 1023, 4096, 100003):
        src = torch.randint(-5, 2 ** 31 - 1, (n + 4,), generator=g, dtype=torch.int64).to(torch.int32).to(DEV)
        dst = torch.full((n + 4,), -77, dtype=torch.int64, device=DEV)
        lib.check(lib.load().nir_widen_ids_i32(lib.ptr(src), lib.ptr(dst), n, lib.stream()), "widen")
        assert torch.equal(dst[:n].cpu(), src[:n].cpu().long())
        assert (dst[n:] == -77).all()                       # nothing written past n


def _model(V, dtype="f32"):
    from context_attentive_ir_amd.config import default_args
    from context_attentive_ir_amd.detinit import fill_module_
    from context_attentive_ir_amd.wrappers import Multitask
    mt = Multitask(default_args("CARS", src_vocab_size=V, tgt_vocab_size=300))
    fill_module_(mt.network, 1013)
    mt.network.compute_dtype = dtype
    mt.cuda()
    mt.network.eval()
    return mt


def _oracle_probs(sd, ex):
    return O.predict_softmax(O.cars_scores(sd, ex["source_words"], ex["source_lens"], ex["document_words"], ex["document_lens"], ex["document_labels"]))


@pytest.mark.parametrize("producers", [1, 2])
def test_stream_map_equals_oracle_on_200_sessions(producers):
    """A 208-session slice of the synthetic stream (Poisson session lengths, ragged token lengths, reference-sampler batches of 8 equal-length
    sessions, 10 candidates): every batch's click probabilities within 1e-4 of the oracle and MAP (eval/ltorank.py:4-26 over the argsort,
    main/multitask.py:285-287) IDENTICAL."""
    from context_attentive_ir_amd.eval.ltorank import MAP
    from context_attentive_ir_amd.graph_runner import StreamingSessionPredictor
    from context_attentive_ir_amd.inputters import SyntheticSessionCorpus
    V, B, N = 3000, 8, 10
    corpus = SyntheticSessionCorpus(n_sessions=208, n_cands=N, qlen=5, dlen=24, vocab=V, seed=5, pool=8, full_length=False, s_max=9)
    batches = corpus.batches(B, seed=3)
    assert 15 <= len(batches) <= 26 and all(len({int(corpus.lengths[i]) for i in b}) == 1 for b in batches)
    mt = _model(V)
    sd = cpu_state_dict(mt.network)
    sp = StreamingSessionPredictor(mt, N, 5, 24, B, max_session_len=9, lanes=2, slots=2)
    got = {}
    stats = sp.run(corpus, batches, on_result=lambda k, idx, probs: got.__setitem__(k, probs.clone()), producers=producers)
    assert stats["batches"] == len(batches) and sorted(got) == list(range(len(batches)))
    maps_gpu, maps_ref = [], []
    for k, idx in enumerate(batches):
        ex = corpus.batch_tensors(idx)
        ref = _oracle_probs(sd, ex)
        np.testing.assert_allclose(got[k].numpy(), ref.numpy(), rtol=0, atol=1e-4)
        lab = ex["document_labels"].reshape(-1, N).numpy().astype(int)
        maps_gpu.append(MAP(np.argsort(-got[k].reshape(-1, N).numpy(), axis=1, kind="stable"), lab))
        maps_ref.append(MAP(np.argsort(-ref.reshape(-1, N).numpy(), axis=1, kind="stable"), lab))
    assert maps_gpu == maps_ref
    mt.network.check_ids()


def test_stream_bf16_and_cycling():
    """bf16 folded tables through the same pipeline (config 5): probabilities within the bf16 bound of the oracle; min_seconds cycles the
    batch list and keeps every lane busy."""
    from context_attentive_ir_amd.graph_runner import StreamingSessionPredictor
    from context_attentive_ir_amd.inputters import SyntheticSessionCorpus
    V, B, N = 3000, 4, 7
    corpus = SyntheticSessionCorpus(n_sessions=64, n_cands=N, qlen=4, dlen=16, vocab=V, seed=9, pool=4, s_max=6)
    batches = corpus.batches(B, seed=1)
    mt = _model(V, "bf16")
    sd = cpu_state_dict(mt.network)
    sp = StreamingSessionPredictor(mt, N, 4, 16, B, max_session_len=6, lanes=2, slots=2)
    got = {}
    sp.run(corpus, batches, on_result=lambda k, idx, probs: got.__setitem__(k, probs.clone()))
    for k, idx in enumerate(batches):
        ref = _oracle_probs(sd, corpus.batch_tensors(idx))
        assert float((got[k] - ref).abs().max()) <= 2e-2
    r = sp.run(corpus, batches, min_seconds=0.5)
    assert r["batches"] > len(batches) and r["pairs_per_s"] > 0 and r["seconds"] >= 0.5


def test_stream_macro_batches_equal_single_batches():
    """macro = 3: three sampler batches of one session length per wire block / graph replay (Multitask.predict_groups) give the probabilities
    of the same batches scored one by one -- every batch keeps its own click count -- and the oracle's."""
    from context_attentive_ir_amd.graph_runner import StreamingSessionPredictor
    from context_attentive_ir_amd.inputters import SyntheticSessionCorpus
    V, B, N, MK = 3000, 4, 6, 3
    corpus = SyntheticSessionCorpus(n_sessions=160, n_cands=N, qlen=4, dlen=16, vocab=V, seed=13, pool=6, full_length=False, s_max=5)
    for i in range(0, 160, 7):                      # multi-click sessions: batches differ in their batch-wide click count
        p = corpus.pool[int(corpus.lengths[i])]
        p["document_labels"][corpus.slot[i], 0, :3] = 1.0
    batches = corpus.batches(B, seed=2)
    mt = _model(V)
    sd = cpu_state_dict(mt.network)
    sp = StreamingSessionPredictor(mt, N, 4, 16, MK * B, max_session_len=5, lanes=2, slots=2, macro=MK)
    merged, rest = sp.merge_batches(corpus, batches, MK)
    assert merged and len(merged) * MK + len(rest) == len(batches) and all(len(m) == MK * B for m in merged)
    got = {}
    sp.run(corpus, merged, on_result=lambda k, idx, probs: got.__setitem__(k, probs.clone()))
    for k, idx in enumerate(merged):
        for j in range(MK):
            ex = corpus.batch_tensors(idx[j * B:(j + 1) * B])
            np.testing.assert_allclose(got[k][j * B:(j + 1) * B].numpy(), _oracle_probs(sd, ex).numpy(), rtol=0, atol=1e-4)


def test_pair_mode_block_with_shipped_click_count_equals_whole_batch():
    """sharding.StreamShardPlan mode 'pair' on ONE process: the predictor of simulated rank r scores batch_size / world whole sessions of every
    batch with the batch's click count shipped in the wire block (`click_max`, int32 on the device); the blocks of all simulated ranks put
    together equal the whole batch scored at once (and the oracle), including batches whose count lives on another rank."""
    from context_attentive_ir_amd import sharding
    from context_attentive_ir_amd.graph_runner import StreamingSessionPredictor
    from context_attentive_ir_amd.inputters import SyntheticSessionCorpus
    V, B, N, world = 3000, 8, 7, 4
    corpus = SyntheticSessionCorpus(n_sessions=96, n_cands=N, qlen=4, dlen=16, vocab=V, seed=17, pool=6, full_length=False, s_max=5, multi_click=True)
    for body in corpus.pool.values():
        body["document_labels"][0, 0, :] = 1.0
        body["_clicks"] = (body["document_labels"] != 0).sum(-1).max(-1).astype(np.int32)
    bs = corpus.batches(B, seed=4)
    mt = _model(V)
    sd = cpu_state_dict(mt.network)
    blocks, differs = {}, 0
    for r in range(world):
        plan = sharding.StreamShardPlan(world, r, "pair", batch_size=B)
        sp = StreamingSessionPredictor(mt, N, 4, 16, B, max_session_len=5, lanes=2, slots=2, plan=plan, gather="none")
        assert sp.B == B // world and sp.groups == 1
        sp.run(corpus, bs, on_result=lambda k, idx, p, r=r: blocks.__setitem__((k, r), p.clone()))
        differs += sum(int(corpus.click_max(plan.mine(j, bs)[1]) < corpus.click_max(bs[j])) for j in range(len(bs)))
    assert differs > 0                                   # some block's own count is below its batch's: the shipped count matters
    for k, idx in enumerate(bs):
        got = torch.cat([blocks[(k, r)] for r in range(world)])
        ref = _oracle_probs(sd, corpus.batch_tensors(idx))
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=1e-4)
