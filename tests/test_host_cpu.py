"""CPU (-m "not gpu"): host logic -- C-ABI symbols, API surface / state-dict keys, loud failure without a GPU,
metrics, synthetic data, candidate sharding over a 2-rank gloo group."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from helpers import build_model


def test_library_exports_every_declared_symbol():
    from context_attentive_ir_amd import lib
    hdr = open(os.path.join(ROOT, "include", "neuroir_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nir_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(lib.SIGNATURES), (declared ^ set(lib.SIGNATURES))
    L = lib.load()                              # dlopen + resolve every symbol (no compute call)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.nir_version() >= 100


def test_argument_errors_are_reported_without_a_gpu():
    from context_attentive_ir_amd import lib
    L = lib.load()
    rc = L.nir_esm_score(None, None, 1, 1, 1, 1, None, 1, 300, None, None)
    assert rc < 0 and b"null" in L.nir_last_error_string()
    assert L.nir_bilstm_supported(128) == 1 and L.nir_bilstm_supported(129) == 0


def test_no_cpu_fallback():
    m = build_model("ESM")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 4, dtype=torch.long), torch.ones(2, dtype=torch.long),
          torch.zeros(2, 3, 8, dtype=torch.long), torch.ones(2, 3, dtype=torch.long))


REF_KEYS = {  # SURVEY.md Appendix C (probed from the reference)
    "ESM": ["word_embeddings.make_embedding.emb_luts.0.weight"],
    "DRMM": ["word_embeddings.make_embedding.emb_luts.0.weight", "gating_network.weight.weight", "gating_network.weight.bias",
             "ffnn.0.weight", "ffnn.0.bias", "ffnn.1.weight", "ffnn.1.bias", "output.weight", "output.bias"],
}


def test_state_dict_layouts():
    for kind, keys in REF_KEYS.items():
        assert list(build_model(kind).state_dict().keys()) == keys
    mt = build_model("MATCH_TENSOR").state_dict()
    assert mt["query_encoder.rnns.0.weight_hh_l0_reverse"].shape == (60, 15)
    assert mt["document_encoder.rnns.0.weight_ih_l0"].shape == (280, 40)
    assert mt["conv3.weight"].shape == (6, 51, 3, 7) and mt["exact_match_channel.alpha"].shape == (1,)
    assert sum(v.numel() for k, v in mt.items() if "emb_luts" not in k) == 104390
    du = build_model("DUET", max_query_len=4, max_doc_len=290).state_dict()
    assert du["local_model.conv1d.weight"].shape == (300, 290, 1) and du["distributed_model.fc2.weight"].shape == (1, 284)
    assert sum(v.numel() for k, v in du.items() if "emb_luts" not in k) == 989992
    ca = build_model("CARS").state_dict()
    for k, shp in {"query_encoder.encoder.rnns.0.weight_hh_l0": (512, 128), "session_doc_encoder.encoder.rnns.0.weight_ih_l0": (2048, 256),
                   "ranknet._linear_layers.0.weight": (512, 1024), "shared_session_projector.linear.weight": (256, 1024),
                   "decoder.decoder.attn.linear_out.weight": (512, 1024), "q_attn.3.weight": (1, 256)}.items():
        assert tuple(ca[k].shape) == shp, k


def test_config_surface():
    import argparse
    from context_attentive_ir_amd import config
    p = argparse.ArgumentParser(); config.add_model_args(p)
    a = p.parse_args(["--max_doc_len", "64", "--fix_embeddings", "true"])
    a.model_type = "match_tensor"
    a = config.update_model_args(a)
    m = config.get_model_args(a)
    assert m.nhid_doc == 140 and m.max_doc_len == 64 and m.fix_embeddings is True and not hasattr(m, "early_stop")
    old = argparse.Namespace(dropout=0.2, nhid_doc=140); new = argparse.Namespace(dropout=0.5, nhid_doc=10)
    o = config.override_model_args(old, new)
    assert o.dropout == 0.5 and o.nhid_doc == 140


def test_metrics_match_golden():
    from context_attentive_ir_amd.eval import MAP, MRR, precision_at_k, rank_candidates
    g = load_golden("losses_metrics")
    pred = rank_candidates(g["softmax"])
    np.testing.assert_array_equal(pred, g["predictions"])
    assert MAP(pred, g["labels"]) == pytest.approx(float(g["MAP"]), abs=1e-12)
    assert MRR(pred, g["labels"]) == pytest.approx(float(g["MRR"]), abs=1e-12)
    assert precision_at_k(pred, g["labels"], 3) == pytest.approx(float(g["P3"]), abs=1e-12)
    with pytest.raises(ZeroDivisionError):
        MAP(pred[:1], np.zeros_like(g["labels"][:1]))


def test_synth_shapes_and_padding():
    from context_attentive_ir_amd import synth
    b = synth.ranker_batch(4, 5, 6, 32, 1000, seed=1, full_length=False)
    assert b["doc_rep"].shape == (4, 5, 32) and b["doc_rep"].dtype == torch.int64
    pos = torch.arange(32)
    assert ((b["doc_rep"] == 0) == (pos >= b["doc_len"].unsqueeze(-1))).all()
    assert (b["label"].sum(1) == 1).all() and int(b["doc_rep"].max()) < 1000 and int(b["doc_rep"][b["doc_rep"] > 0].min()) >= 4
    s = synth.session_batch(2, 3, 4, 5, 12, 500, multi_click=True)
    assert s["document_words"].shape == (2, 3, 4, 12) and s["document_labels"].dtype == torch.float32
    assert (s["document_labels"].sum(-1) >= 1).all()


def test_shard_bounds_cover_all_candidates():
    from context_attentive_ir_amd.sharding import shard_bounds
    for N, G in [(50, 8), (10, 8), (10, 2), (3, 4), (64, 1)]:
        got = []
        for r in range(G):
            lo, hi, per = shard_bounds(N, G, r)
            got += list(range(lo, hi))
            assert hi - lo <= per == -(-N // G)
        assert got == list(range(N))


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from context_attentive_ir_amd import sharding
from context_attentive_ir_amd.detinit import fill_module_
from context_attentive_ir_amd.rankers import ESM
from context_attentive_ir_amd.config import default_args
from context_attentive_ir_amd import synth
from oracle import neuroir_cpu as O
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[4]
dist.init_process_group("gloo", rank=rank, world_size=world)
sd = {k: v for k, v in fill_module_(ESM(default_args("ESM", src_vocab_size=300))).state_dict().items()}
for N in (5, 10, 1):
    ex = synth.ranker_batch(3, N, 4, 16, 300, seed=N, full_length=False)
    q, ql = ex["que_rep"], ex["que_len"]
    fn = lambda d, l: O.esm_scores(sd, q, ql, d, l)          # the CPU oracle stands in for the HIP scorer
    full = fn(ex["doc_rep"], ex["doc_len"])
    got = sharding.sharded_scores(fn, ex["doc_rep"], ex["doc_len"])
    assert got.shape == full.shape and torch.equal(got, full), (rank, N, (got - full).abs().max())
# CARS: shard the document encoder, all-gather pooled vectors, replicated session part (oracle stands in for HIP)
from context_attentive_ir_amd.multitask import CARS
sdc = {k: v for k, v in fill_module_(CARS(default_args("CARS", src_vocab_size=300))).state_dict().items()}
for N in (5, 3):
    ex = synth.session_batch(2, 3, N, 4, 10, 300, seed=N, full_length=False, multi_click=True)
    enc = lambda d, l: O.cars_encode_document(sdc, d, l)
    full = enc(ex["document_words"], ex["document_lens"])
    got = sharding.sharded_pooled_docs(enc, ex["document_words"], ex["document_lens"])
    assert got.shape == full.shape and torch.allclose(got, full, atol=1e-6), (rank, N)
    pooled, _ = O.cars_encode(sdc, ex["source_words"], ex["source_lens"])
    s_full = O.cars_encode_session(sdc, pooled, full, O.cars_encode_clicks(sdc, full, ex["document_labels"]))
    s_got = O.cars_encode_session(sdc, pooled, got, O.cars_encode_clicks(sdc, got, ex["document_labels"]))
    assert torch.allclose(s_got, s_full, atol=1e-6)
    # sharded ranker MLP: clicks / sessions from the gathered documents, scores of this rank's own pooled slice, score slices gathered
    got2, own = sharding.sharded_pooled_docs(enc, ex["document_words"], ex["document_lens"], return_local=True)
    assert torch.allclose(got2, full, atol=1e-6) and own.shape[2] == (N + world - 1) // world
    s_loc = O.cars_encode_session(sdc, pooled, own, O.cars_encode_clicks(sdc, got2, ex["document_labels"]))
    s_all = sharding.gather_session_scores(s_loc, N)
    assert s_all.shape == s_full.shape and torch.allclose(s_all, s_full, atol=1e-6), (rank, N)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_candidate_sharding_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % r) in o, o


_SESSION_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from context_attentive_ir_amd import sharding, synth
from context_attentive_ir_amd.detinit import fill_module_
from context_attentive_ir_amd.config import default_args
from context_attentive_ir_amd.multitask import CARS
from oracle import neuroir_cpu as O
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[4]
dist.init_process_group("gloo", rank=rank, world_size=world)
sd = {k: v for k, v in fill_module_(CARS(default_args("CARS", src_vocab_size=300))).state_dict().items()}
enc_q = lambda q, l: O.cars_encode(sd, q, l)[0]                      # the CPU oracle stands in for the HIP entry points
enc_d = lambda d, l: O.cars_encode_document(sd, d, l)
def tail(pq, docs, lab, lab_all):
    return O.predict_softmax(O.cars_encode_session(sd, pq, docs, O.cars_encode_clicks(sd, docs, lab, labels_all=lab_all)))
for B, S, N in ((4, 3, 5), (3, 2, 9), (1, 3, 2), (5, 2, 4)):        # B, N divisible / ragged / smaller than the world
    ex = synth.session_batch(B, S, N, 4, 10, 300, seed=10 * B + N, full_length=False, multi_click=True)
    if B > 1:                                                         # one session with many clicks: the batch-wide m lives on ONE rank
        ex["document_labels"][B - 1, 0, :] = 1.0
    full = O.predict_softmax(O.cars_scores(sd, ex["source_words"], ex["source_lens"], ex["document_words"], ex["document_lens"],
                                            ex["document_labels"]))
    plan = sharding.SessionShardPlan(B, S, N, world, rank)
    got = sharding.session_sharded_click_probs(plan, enc_q, enc_d, tail, ex)
    assert got.shape == full.shape and torch.allclose(got, full, atol=1e-6), (rank, B, N, float((got - full).abs().max()))
    got2 = sharding.session_sharded_click_probs(plan, enc_q, enc_d, tail, ex, via_gather=True)       # the capturable form of the exchange
    assert torch.equal(got2, got), (rank, B, N)
    auto = sharding.SessionShardPlan(B, S, N, world, rank, axis="auto")
    assert auto.aligned == (B % world == 0) and auto.exchange_bytes(256) == (0 if auto.aligned else plan.exchange_bytes(256))
    got3 = sharding.session_sharded_click_probs(auto, enc_q, enc_d, tail, ex)       # pair axis (whole sessions per rank) when B % world == 0
    assert got3.shape == full.shape and torch.allclose(got3, full, atol=1e-6), (rank, B, N, auto.axis)
    # a block evaluated WITHOUT the batch-wide labels differs whenever its own max click count is smaller: the quirk is really exercised
    cnt = lambda l: int((l.reshape(-1, N) != 0).sum(1).max())
    if B > 1 and world > 1 and rank == 0 and cnt(plan.own(ex["document_labels"])) < min(cnt(ex["document_labels"]), N - 1):
        loc = tail(enc_q(plan.own(ex["source_words"]), plan.own(ex["source_lens"])),
                   O.cars_encode_document(sd, plan.own(ex["document_words"]), plan.own(ex["document_lens"])),
                   plan.own(ex["document_labels"]), None)
        assert not torch.allclose(loc, full[:plan.bper], atol=1e-6)
    assert plan.exchange_bytes(256) == (world - 1) * plan.bper * S * plan.per * 256 * 4
# the software-pipelined form: ONE all_to_all_single per step carries the pooled slices of step k and the probabilities of step k-1
B, S, N = 4, 3, 6
exs = [synth.session_batch(B, S, N, 4, 10, 300, seed=77 + i, full_length=False, multi_click=True) for i in range(3)]
plan = sharding.SessionShardPlan(B, S, N, world, rank)
outs = sharding.pipelined_session_sharded_probs(plan, enc_q, enc_d, tail, exs)
assert len(outs) == 3
for ex, got in zip(exs, outs):
    full = O.predict_softmax(O.cars_scores(sd, ex["source_words"], ex["source_lens"], ex["document_words"], ex["document_lens"],
                                            ex["document_labels"]))
    assert got.shape == full.shape and torch.allclose(got, full, atol=1e-6), (rank, float((got - full).abs().max()))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", [2, 4])
def test_session_sharded_tail_gloo(tmp_path, world):
    """candidate-sharded encode -> all-to-all -> session-sharded tail -> all-gather (sharding.SessionShardPlan) reproduces the unsharded
    click probabilities on every rank, including the batch-wide click-mask quirk (m comes from the replicated labels)."""
    script = tmp_path / "worker.py"
    script.write_text(_SESSION_WORKER)
    port = str(31500 + (os.getpid() * 7 + world) % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % r) in o, o


def _grad_sync_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from context_attentive_ir_amd.config import default_args
        from context_attentive_ir_amd.detinit import fill_module_
        from context_attentive_ir_amd.wrappers import Multitask, Ranker
        ok = True
        for w in (Ranker(default_args("DRMM", src_vocab_size=50)), Multitask(default_args("CARS", src_vocab_size=50, tgt_vocab_size=40))):
            fill_module_(w.network, 1013)
            params = [p for p in w.network.parameters() if p.requires_grad]
            for i, p in enumerate(params):
                p.grad = None if (rank == 1 and i == 0) else torch.full_like(p, float(rank + 1) * (i + 1))     # rank 1 never used param 0
            ok &= w.sync_gradients() is False                      # not parallelised: untouched
            ok &= float(params[1].grad.flatten()[0]) == float(rank + 1) * 2
            w.parallelize()
            ok &= w.sync_gradients() is True
            for i, p in enumerate(params):                         # mean over ranks of (rank+1)*(i+1); param 0: (1 + 0) / 2
                want = 0.5 if i == 0 else 1.5 * (i + 1)
                ok &= bool(torch.all(p.grad == want))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_update_gradient_sync_two_ranks():
    """Ranker.update / Multitask.update under parallelize(): gradients are averaged over the ranks before clipping
    (WrapperBase.sync_gradients), the role nn.DataParallel's backward plays in the reference (models/ranker.py:341-346)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + os.getpid() % 100
    procs = [ctx.Process(target=_grad_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def _async_gather_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from context_attentive_ir_amd import sharding
    try:
        B, N = 3, 7
        full = [torch.arange(B * N, dtype=torch.float32).view(B, N) + 100 * k for k in range(3)]
        per = (N + world - 1) // world
        handles = []
        for k in range(3):                       # three batches in flight before the first wait
            lo, hi, _ = sharding.shard_bounds(N, world, rank)
            loc = torch.zeros(B, per)
            loc[:, :hi - lo] = full[k][:, lo:hi]
            handles.append(sharding.ScoreGather(loc, N))
        ok = all(torch.equal(h.wait(), full[k]) for k, h in enumerate(handles))
        try:                                     # the softmax forms are device-only: no CPU arithmetic in the product
            handles[0].softmax()
            ok = False
        except RuntimeError:
            pass
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_async_score_gather_two_ranks():
    """ScoreGather (async all-gather consumed a batch later) returns the same [B,N] as the blocking gather."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    procs = [ctx.Process(target=_async_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_batchify_matches_reference_collate():
    """Row a0: ranker_batchify / session_batchify reproduce the reference's collate outputs (fixture generated by
    tests/golden/generate.py from neuroir.inputters.*.vector.batchify) bit for bit, dtypes included."""
    from context_attentive_ir_amd.inputters import ranker_batchify, session_batchify
    g = load_golden("batchify")
    qlens, dlens = g["r_qlens"], g["r_dlens"]
    B, N = dlens.shape
    batch, qo, do = [], 0, 0
    for b in range(B):
        q = g["r_qflat"][qo:qo + qlens[b]]; qo += qlens[b]
        docs = []
        for n in range(N):
            docs.append(torch.from_numpy(g["r_dflat"][do:do + dlens[b, n]].astype(np.int64))); do += dlens[b, n]
        batch.append({"id": b, "query_words": torch.from_numpy(q.astype(np.int64)), "doc_words": docs,
                      "label": torch.from_numpy(g["r_labels"][b].astype(np.int64)), "num_candidates": N,
                      "max_doc_len": int(dlens[b].max()), "max_query_len": int(qlens[b])})
    out = ranker_batchify(batch)
    for k in ("doc_rep", "doc_len", "que_rep", "que_len", "label"):
        assert out[k].dtype == torch.int64 and np.array_equal(out[k].numpy(), g["r_out_" + k]), k
    assert out["batch_size"] == B and out["ids"] == list(range(B))
    # all fields are views of one contiguous buffer (single H2D copy)
    base = out["_buffer"].data_ptr()
    assert all(base <= out[k].data_ptr() < base + out["_buffer"].numel() for k in ("doc_rep", "doc_len", "que_rep", "que_len", "label"))

    keys = ("source_words", "source_lens", "target_words", "target_lens", "target_seq", "document_words", "document_lens",
            "document_labels")
    sess, b = [], 0
    while "s_in%d_source_words" % b in g:
        ex = {k: torch.from_numpy(g["s_in%d_%s" % (b, k)]) for k in keys}
        ex.update(id=b, session_len=ex["source_words"].shape[0], num_candidates=ex["document_lens"].shape[1],
                  max_source_len=ex["source_words"].shape[1], max_target_len=ex["target_words"].shape[1],
                  max_document_len=ex["document_words"].shape[2])
        sess.append(ex); b += 1
    sout = session_batchify(sess)
    for k in keys:
        assert np.array_equal(sout[k].numpy(), g["s_out_" + k]), k
        assert sout[k].dtype == (torch.float32 if k == "document_labels" else torch.int64), k
    assert bool(g["s_out_document_labels_is_float"])
    sess[1]["session_len"] = 99
    with pytest.raises(AssertionError):
        session_batchify(sess)


def test_validate_official_matches_reference_loop():
    """eval.validate_official == the reference's per-batch loop (main/ranker.py:236-297): mean over batches of the
    batch metrics, driven here by a stand-in predictor (CPU tensors) so the host logic is covered without a GPU."""
    from context_attentive_ir_amd.eval import validate_official, MAP, MRR, precision_at_k

    class Stub(object):
        def predict(self, ex):
            return ex["_scores"]

    rng = np.random.default_rng(5)
    batches = []
    for _ in range(7):
        B, N = int(rng.integers(1, 6)), 6
        lab = np.zeros((B, N), dtype=np.int64)
        lab[np.arange(B), rng.integers(0, N, size=B)] = 1
        batches.append({"_scores": torch.from_numpy(rng.standard_normal((B, N)).astype(np.float32)), "label": torch.from_numpy(lab)})
    got = validate_official(batches, Stub(), depth=2)
    ref = {"map": [], "mrr": [], "prec@1": [], "prec@3": [], "prec@5": []}
    for ex in batches:
        pred = np.argsort(-ex["_scores"].numpy(), kind="stable")
        lab = ex["label"].numpy()
        ref["map"].append(MAP(pred, lab)); ref["mrr"].append(MRR(pred, lab))
        for k in (1, 3, 5):
            ref["prec@%d" % k].append(precision_at_k(pred, lab, k))
    for k, v in ref.items():
        assert abs(got[k] - float(np.mean(v))) < 1e-12, k
    assert got["examples"] == sum(b["label"].shape[0] for b in batches)
    # CARS layout: dict output, [B,S,N] rows flattened to B*S
    class Stub2(object):
        def predict(self, ex):
            return {"click_scores": ex["_scores"], "predictions": None}
    ex = {"_scores": torch.randn(2, 3, 5), "document_labels": torch.eye(5)[torch.randint(0, 5, (2, 3))]}
    r = validate_official([ex], Stub2())
    assert r["examples"] == 6 and 0 < r["map"] <= 1


def test_samplers_match_reference_batches():
    """inputters.samplers reproduce the reference samplers' index sequence under the same numpy seed (fixture from
    neuroir.inputters.{ranker,multitask}.data.SortedBatchSampler)."""
    from context_attentive_ir_amd.inputters import length_sorted_batches, session_length_batches, flat_indices
    g = load_golden("samplers")
    seed = int(str(g["meta_seed"]))
    for shuffle in (False, True):
        np.random.seed(seed)
        got = flat_indices(length_sorted_batches(g["r_lengths"], 8, shuffle=shuffle))
        assert got == g["r_flat_shuffle%d" % shuffle].tolist()
        np.random.seed(seed)
        batches = session_length_batches(g["s_lengths"], 4, shuffle=shuffle)
        assert flat_indices(batches) == g["s_flat_shuffle%d" % shuffle].tolist()
        assert all(len(set(g["s_lengths"][b].tolist())) == 1 and len(b) == 4 for b in batches)


def test_prefetching_stream_order_and_shutdown():
    from context_attentive_ir_amd.inputters import PrefetchingBatchStream, ranker_batchify, flat_examples, length_sorted_batches
    rng = np.random.default_rng(3)
    n, N = 23, 3
    q = [rng.integers(4, 99, size=rng.integers(1, 6)) for _ in range(n)]
    d = [[rng.integers(4, 99, size=rng.integers(1, 12)) for _ in range(N)] for _ in range(n)]
    ex = flat_examples(q, d, rng.integers(0, 2, size=(n, N)))
    batches = length_sorted_batches([(max(len(x) for x in dd), len(qq)) for qq, dd in zip(q, d)], 4, shuffle=True,
                                    rng=np.random.RandomState(0))
    stream = PrefetchingBatchStream(ex, batches, ranker_batchify, depth=2, pin=False)
    got = list(stream)
    assert len(got) == len(batches) == 6
    for b, idx in zip(got, batches):
        ref = ranker_batchify([ex[int(i)] for i in idx])
        assert b["ids"] == [int(i) for i in idx] and torch.equal(b["doc_rep"], ref["doc_rep"]) and torch.equal(b["que_rep"], ref["que_rep"])
    it = iter(stream)            # abandoning the iterator early must not leave the producer blocked
    next(it)
    it.close()
    import threading
    assert not any(t.name == "nir-batch-prefetch" and t.is_alive() for t in threading.enumerate())


def test_session_stream_corpus_and_wire_layout():
    """SURVEY.md 8(d) stream: S ~ clip(Poisson(4.84)+2, 2, 16); batches = the reference sampler's composition (equal-length, full batches);
    the int32 wire block round-trips every field and keeps 16-byte aligned field starts."""
    from context_attentive_ir_amd.inputters import SyntheticSessionCorpus, WireLayout
    c = SyntheticSessionCorpus(n_sessions=6000, n_cands=5, qlen=4, dlen=12, vocab=500, seed=2, pool=6, full_length=False)
    assert c.lengths.min() >= 2 and c.lengths.max() <= 16 and abs(float(c.lengths.mean()) - 6.84) < 0.15
    bs = c.batches(16, seed=4)
    assert all(len(b) == 16 and len({int(c.lengths[i]) for i in b}) == 1 for b in bs)
    dropped = len(c) - 16 * len(bs)
    assert 0 <= dropped < 16 * len(np.unique(c.lengths))               # only the remainder of each length cluster is dropped
    assert bs != c.batches(16, shuffle=False)                           # shuffled batch order, same content
    assert sorted(map(sorted, bs)) == sorted(map(sorted, c.batches(16, shuffle=False)))
    lay = WireLayout(16, 7, 5, 4, 12)
    assert all(o % 4 == 0 for o in lay.offset.values()) and lay.nbytes % 16 == 0
    buf = np.full(c.layout(16, 16).nbytes, 0xAB, np.uint8)
    lay = c.collate_into(bs[0], buf)
    ref = c.batch_tensors(bs[0])
    host, dev_like = lay.views(buf), lay.views(torch.from_numpy(buf))
    for k, v in ref.items():
        assert host[k].shape == tuple(v.shape) and (host[k] == v.numpy()).all() and (dev_like[k].numpy() == v.numpy()).all(), k
    assert ref["document_words"].dtype == torch.int64 and host["document_words"].dtype == np.int32      # half the bytes on the wire
    wide = lay.wide_views(torch.from_numpy(buf[:4 * lay.n_int].view(np.int32).astype(np.int64)))
    assert all(torch.equal(wide[k], ref[k]) for k in wide)
    pos = np.arange(12)
    assert ((ref["document_words"].numpy() == 0) == (pos >= ref["document_lens"].numpy()[..., None])).all()


def test_stream_shard_plan_covers_every_batch_once():
    """sharding.StreamShardPlan: 'batch' mode deals whole sampler batches round-robin (short last round = fillers that are dropped),
    'pair' mode cuts every (macro-)batch into world blocks of whole sessions; unpack() restores batch order from the rank-major gather."""
    from context_attentive_ir_amd import sharding
    from context_attentive_ir_amd.inputters import SyntheticSessionCorpus
    c = SyntheticSessionCorpus(n_sessions=500, n_cands=3, qlen=3, dlen=6, vocab=60, seed=5, pool=8, full_length=False, multi_click=True)
    bs = c.batches(8, seed=1)
    lengths_of = lambda idx: int(c.lengths[idx[0]])     # noqa: E731
    for world in (1, 2, 3, 4):
        seen = []
        plans = [sharding.StreamShardPlan(world, r, "batch") for r in range(world)]
        for j in range(plans[0].rounds(len(bs))):
            real = dict(plans[0].members(j, len(bs)))
            for r, p in enumerate(plans):
                k, own, whole = p.mine(j, bs)
                assert own == whole == list(bs[k])
                if r in real:
                    assert real[r] == k
                    seen.append(k)
                else:
                    assert k == j * world                     # filler of a short last round
        assert seen == list(range(len(bs)))
    with pytest.raises(ValueError):
        sharding.StreamShardPlan(3, 0, "pair", batch_size=8)
    macro = [bs[0] + bs[0]]                                   # two batches of one length back to back
    for world in (2, 4):
        plans = [sharding.StreamShardPlan(world, r, "pair", batch_size=8) for r in range(world)]
        owns = [p.mine(0, macro)[1] for p in plans]
        assert sorted(x for o in owns for x in o) == sorted(macro[0]) and all(len(o) == 2 * 8 // world for o in owns)
        S, N = lengths_of(macro[0]), 3
        full = torch.arange(16 * S * N, dtype=torch.float32).view(16, S, N)
        pos = {x: i for i, x in enumerate(macro[0][:8])}
        blocks = []
        for o in owns:                                       # what each rank would send: the rows of ITS sessions, group-major
            rows = [g * 8 + pos[x] for g in range(2) for x in o[g * (8 // world):(g + 1) * (8 // world)]]
            blk = torch.zeros(plans[0].block_elems(16, S + 2, N))
            blk[:len(rows) * S * N] = full[rows].reshape(-1)
            blocks.append(blk)
        (k, idx, got), = plans[0].unpack(0, torch.stack(blocks), macro, lengths_of, N)
        assert k == 0 and idx == macro[0] and torch.equal(got, full)
    assert (c.click_max(macro[0], 8) == np.array([c.click_max(bs[0])] * 2)).all() and c.click_max(bs[0]) >= 1
    lay = c.collate_into(bs[0][:4], np.zeros(c.layout(16, 8, 1).nbytes, np.uint8), whole=bs[0], batch_size=8)
    assert lay.groups == 1 and lay.offset["click_max"] % 4 == 0 and lay.B == 4


_STREAM_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from context_attentive_ir_amd import sharding
from context_attentive_ir_amd.detinit import fill_module_
from context_attentive_ir_amd.config import default_args
from context_attentive_ir_amd.multitask import CARS
from context_attentive_ir_amd.inputters import SyntheticSessionCorpus
from context_attentive_ir_amd.eval.ltorank import MAP, rank_candidates
from oracle import neuroir_cpu as O
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[4]
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.set_num_threads(2)
sd = {k: v for k, v in fill_module_(CARS(default_args("CARS", src_vocab_size=300))).state_dict().items()}
corpus = SyntheticSessionCorpus(n_sessions=200, n_cands=9, qlen=4, dlen=10, vocab=300, seed=11, pool=12, full_length=False, s_max=5, multi_click=True)
for S_, body in corpus.pool.items():            # one session per length clicks EVERY candidate of its first query: the batch-wide count of a
    body["document_labels"][0, 0, :] = 1.0      # batch that holds it lives on ONE rank in 'pair' mode
    body["_clicks"] = (body["document_labels"] != 0).sum(-1).max(-1).astype(np.int32)
bs = corpus.batches(8, seed=3)
N = corpus.N
def score(ex, m):                                   # the CPU oracle stands in for the graph-replayed HIP predict
    pq = O.cars_encode(sd, ex["source_words"], ex["source_lens"])[0]
    docs = O.cars_encode_document(sd, ex["document_words"], ex["document_lens"])
    lab_all = None
    if m is not None:                               # the batch's click count as shipped by the collator -> a label row with m clicks
        lab_all = torch.zeros(1, 1, N); lab_all[0, 0, :int(m)] = 1.0
    return O.predict_softmax(O.cars_encode_session(sd, pq, docs, O.cars_encode_clicks(sd, docs, ex["document_labels"], labels_all=lab_all)))
single = {}
for k, idx in enumerate(bs):                        # the single-rank stream: every batch whole
    single[k] = score(corpus.batch_tensors(idx), None)
def stream_map(res):
    p = torch.cat([res[k].reshape(-1, N) for k in sorted(res)]); t = torch.cat([corpus.batch_tensors(bs[k])["document_labels"].reshape(-1, N) for k in sorted(res)])
    return MAP(rank_candidates(p.numpy()), t.numpy())
ref_map = stream_map(single)
modes = ["batch"] + (["pair"] if 8 % world == 0 else [])
for mode in modes:
    plan = sharding.StreamShardPlan(world, rank, mode, batch_size=8)
    got = {}
    def on_result(k, idx, probs):
        assert idx == list(bs[k]) and k not in got
        got[k] = probs
    n = sharding.sharded_stream_probs(plan, score, corpus, bs, on_result=on_result)
    assert n == len(bs) and sorted(got) == list(range(len(bs))), (mode, n, len(bs))
    for k in got:
        assert got[k].shape == single[k].shape and torch.allclose(got[k], single[k], atol=1e-6), (mode, k, float((got[k] - single[k]).abs().max()))
    assert stream_map(got) == ref_map, (mode, stream_map(got), ref_map)
# the quirk is exercised: some batch has a rank block whose own max click count is below the batch's
own_lt = 0
if world > 1 and 8 % world == 0:
    plan = sharding.StreamShardPlan(world, rank, "pair", batch_size=8)
    for j in range(len(bs)):
        _, own, whole = plan.mine(j, bs)
        own_lt += int(corpus.click_max(own) < corpus.click_max(whole))
t = torch.tensor([float(own_lt)]); dist.all_reduce(t)
assert world == 1 or 8 % world or t.item() > 0
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", len(bs), "batches, MAP", round(float(ref_map), 4))
"""


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_session_stream_gloo(tmp_path, world):
    """BASELINE configs[4] as a multi-rank stream (sharding.StreamShardPlan + sharded_stream_probs): both modes deliver EVERY batch's click
    probabilities on EVERY rank, equal to the single-rank stream, with identical MAP on a 200-session slice -- including the batch-wide
    click count in 'pair' mode, where a rank sees only a block of the batch's sessions."""
    script = tmp_path / "worker.py"
    script.write_text(_STREAM_WORKER)
    port = str(33500 + (os.getpid() * 7 + world) % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % r) in o, o


def test_drmm_self_cosine_bins_reproduce_the_oracle_at_exact_matches():
    """rankers.drmm.histogram_bin == numpy.histogram(bins=[-1,-.5,0,.5,1,1]) element by element, and self_cosine_bins(table)[v] is the bin
    the oracle's materialised [B*N,QL,DL,E] cosine (reference drmm.py:59-75) gives EVERY q_id == d_id hit -- the per-row class table the
    HIP kernel looks up (host logic only; the GPU side is tests/test_gpu_parity.py::test_drmm_golden_overlap_exact)."""
    from context_attentive_ir_amd.rankers.drmm import histogram_bin, self_cosine_bins
    from helpers import cpu_state_dict
    from oracle import neuroir_cpu as O
    one = np.float32(1)
    c = np.array([-1.5, -1, -0.7, -0.5, -0.2, 0, 0.3, 0.5, 0.9, np.nextafter(one, np.float32(0)), 1.0, np.nextafter(one, np.float32(2)), np.nan], np.float32)
    want = [int(np.argmax(h)) if h.sum() else -1 for h in (np.histogram([x], bins=O.DRMM_BINS)[0] for x in c)]
    assert histogram_bin(c).tolist() == want and histogram_bin(torch.from_numpy(c)).tolist() == want
    V = 500
    m = build_model("DRMM", vocab=V)
    sd = cpu_state_dict(m)
    bins = self_cosine_bins(m.word_embeddings.table)
    assert bins.dtype == torch.int8 and bins.shape == (V,) and int(bins[0]) == 2            # PAD row: cos = 0 -> [0,.5)
    assert {int(b) for b in bins[1:].unique()} == {-1, 3, 4}                                  # >1 dropped, <1, ==1: all three occur
    rng = np.random.default_rng(7)
    B, N, QL, DL = 3, 4, 5, 30
    q = torch.from_numpy(rng.integers(1, 60, size=(B, QL))); d = torch.from_numpy(rng.integers(0, 60, size=(B, N, DL)))
    _, cos, _ = O.drmm_parts(sd, q, d)
    hit = (q.view(B, 1, QL, 1) == d.view(B, N, 1, DL)).reshape(B * N, QL, DL)
    assert int(hit.sum()) > 20
    got = torch.from_numpy(histogram_bin(cos.numpy()))
    exp = bins[q].view(B, 1, QL, 1).expand(B, N, QL, DL).reshape(B * N, QL, DL)
    assert torch.equal(got[hit], exp[hit])
    g = load_golden("drmm_overlap")                        # and on the real reference's fixture: rebuild its histogram from the table
    m2 = build_model("DRMM")
    t = m2.word_embeddings.table.detach()
    b2 = self_cosine_bins(t)
    qg, dg = torch.from_numpy(g["que_rep"]), torch.from_numpy(g["doc_rep"])
    Bg, Ng, DLg = dg.shape
    QLg = qg.shape[1]
    tn = t / t.norm(dim=1, keepdim=True).clamp_min(1e-8)
    cs = torch.einsum("bqe,bnde->bnqd", tn[qg], tn[dg]).reshape(Bg * Ng, QLg, DLg)      # any rounding: only used away from the edges
    hb = torch.from_numpy(histogram_bin(cs.numpy())).long()
    hit = (qg.view(Bg, 1, QLg, 1) == dg.view(Bg, Ng, 1, DLg)).reshape(Bg * Ng, QLg, DLg)
    hb[hit] = b2[qg].long().view(Bg, 1, QLg, 1).expand(Bg, Ng, QLg, DLg).reshape(Bg * Ng, QLg, DLg)[hit]
    hist = torch.stack([(hb == k).sum(-1) for k in range(5)], -1).numpy()
    np.testing.assert_array_equal(hist, g["hist"].astype(np.int64))


def test_debug_tunables_are_frozen_in_a_product_process():
    """nir_debug_set_tunable only works when the library is loaded with NIR_DEBUG_TUNABLES (tests / profilers); a product process gets
    NIR_ERR_BAD_ARG and the switches stay what the environment said at load time (no caller can steer another's entry points)."""
    lib_path = os.path.join(ROOT, "context_attentive_ir_amd", "libneuroir_hip.so")
    code = ("import ctypes,sys; L=ctypes.CDLL(%r); L.nir_last_error_string.restype=ctypes.c_char_p; "
            "rc=L.nir_debug_set_tunable(b'no_fork',1); print(rc, L.nir_last_error_string().decode()[:60])" % lib_path)
    env = {k: v for k, v in os.environ.items() if k != "NIR_DEBUG_TUNABLES"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rc, msg = out.stdout.strip().split(" ", 1)
    assert int(rc) != 0 and "frozen" in msg
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, NIR_DEBUG_TUNABLES="1"), capture_output=True, text=True, timeout=120)
    assert out.stdout.strip().split(" ", 1)[0] == "0"


def test_native_rank_metrics_equal_the_numpy_definitions():
    """eval.ltorank MAP / MRR / precision_at_k take the library's host loop (nir_host_rank_metric: one C pass per metric, round 6) for plain
    [rows, n] int64 / float32|int64|float64 arrays and the numpy form otherwise: both equal the reference's definitions
    (neuroir/eval/ltorank.py:4-47, 104-123), MAP raises on a row without a relevant candidate."""
    from context_attentive_ir_amd.eval import ltorank as L
    rng = np.random.default_rng(0)
    for dt in (np.float32, np.int64, np.float64):
        for rows, n in ((112, 10), (7, 50), (1, 1), (896, 10)):
            s = rng.random((rows, n)).astype(np.float32)
            lab = np.zeros((rows, n), dt)
            for r in range(rows):
                lab[r, rng.choice(n, int(rng.integers(1, min(n, 4) + 1)), replace=False)] = 1
            p = np.argsort(-s)
            hit = np.take_along_axis(lab, p, 1) == 1
            ap = (((np.cumsum(hit, 1) / np.arange(1, n + 1)) * hit).sum(1) / hit.sum(1)).mean()
            mrr = np.where(hit.any(1), 1.0 / (hit.argmax(1) + 1), 0.0).mean()
            assert L._native(0, p, lab) is not None                                       # the C pass is the one taken
            assert abs(L.MAP(p, lab) - ap) < 1e-12 and abs(L.MRR(p, lab) - mrr) < 1e-12
            for k in (1, 3, 5):
                if n >= k:
                    assert abs(L.precision_at_k(p, lab, k) - hit[:, :k].sum(1).mean() / k) < 1e-12
            pf = np.asfortranarray(p)                                                     # not C-contiguous: the numpy form
            assert pf.flags.c_contiguous or L._native(0, pf, lab) is None
            assert abs(L.MAP(pf, lab) - ap) < 1e-12 and abs(L.MRR(pf, lab) - mrr) < 1e-12
    lab[3] = 0
    with pytest.raises(ZeroDivisionError):
        L.MAP(p, lab)
    with pytest.raises(ZeroDivisionError):
        L.MAP(np.asfortranarray(p), lab)
    assert abs(L.MRR(p, lab) - L.MRR(np.asfortranarray(p), lab)) < 1e-12                  # a row without a relevant candidate counts 0


def test_lstm256_bptt_workspace_size_is_a_host_function():
    """nir_lstm256_bptt_workspace_bytes (csrc/lstm256_bptt.hip): eight dh partials + dc per (direction, sequence, unit), ping-pong, + alignment slack;
    0 for a direction count the entry point rejects -- callable without a GPU."""
    from context_attentive_ir_amd import lib
    L = lib.load()
    assert L.nir_lstm256_bptt_workspace_bytes(1120, 2) == 2 * 9 * 2 * 1120 * 256 * 4 + 256
    assert L.nir_lstm256_bptt_workspace_bytes(7, 1) == 2 * 9 * 1 * 7 * 256 * 4 + 256
    assert L.nir_lstm256_bptt_workspace_bytes(7, 3) == 0
