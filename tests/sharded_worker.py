"""Worker of tests/test_gpu_bench_flow.py::test_two_rank_sharding_reproduces_single_rank_scores (launched by torch.distributed.run,
2 ranks on one GPU, gloo): candidate-sharded scoring must reproduce the unsharded score matrix on every rank."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from context_attentive_ir_amd import synth  # noqa: E402
from context_attentive_ir_amd.config import default_args  # noqa: E402
from context_attentive_ir_amd.detinit import fill_module_  # noqa: E402
from context_attentive_ir_amd.wrappers import Multitask, Ranker  # noqa: E402


def main():
    torch.cuda.set_device(0)
    backend = os.environ.get("SHARD_BACKEND", "gloo")          # "nccl" with ONE rank: the RCCL device-gather path of the stream (communication stream,
    if backend == "nccl":                                       # per-slot gather buffers), which gloo never takes
        dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    V = 1500
    # rankers: N = 7 candidates over 2 ranks (4 + 3, padded shard)
    ex = synth.ranker_batch(5, 7, 4, 33, V, seed=3, full_length=False)
    for kind in (("MATCH_TENSOR", "ESM") if backend == "gloo" else ()):
        r = Ranker(default_args(kind, src_vocab_size=V))
        fill_module_(r.network, 1013)
        r.cuda()
        full = r.predict(ex).cpu()
        r.parallelize()
        shard = r.predict(ex).cpu()
        assert torch.allclose(full, shard, atol=1e-6), (kind, float((full - shard).abs().max()))
    # CARS: candidate-sharded document encoding + all-gather of pooled documents, session part replicated
    if backend == "gloo":
        sex = synth.session_batch(3, 4, 5, 4, 21, V, seed=5, full_length=False, multi_click=True)
        mt = Multitask(default_args("CARS", src_vocab_size=V, tgt_vocab_size=300))
        fill_module_(mt.network, 1013)
        mt.cuda()
        full = mt.predict(sex, suggest=False)["click_scores"].cpu()
        mt.parallelize()
        shard = mt.predict(sex, suggest=False)["click_scores"].cpu()
        assert torch.allclose(full, shard, atol=1e-6), float((full - shard).abs().max())
    # the session stream over the 2 ranks (graph_runner.StreamingSessionPredictor under a sharding.StreamShardPlan; gloo: the all-gather of
    # the probabilities goes through the host): every rank receives every batch, equal to the single-rank stream, identical MAP
    import numpy as np
    from context_attentive_ir_amd import sharding
    from context_attentive_ir_amd.eval.ltorank import MAP, rank_candidates
    from context_attentive_ir_amd.graph_runner import StreamingSessionPredictor
    from context_attentive_ir_amd.inputters import SyntheticSessionCorpus
    world, rank = dist.get_world_size(), dist.get_rank()
    V, B, N = 1500, 8, 6
    corpus = SyntheticSessionCorpus(n_sessions=200, n_cands=N, qlen=4, dlen=16, vocab=V, seed=21, pool=10, full_length=False, s_max=6, multi_click=True)
    for body in corpus.pool.values():                   # one all-click query per length: 'pair' blocks disagree about the batch-wide count
        body["document_labels"][0, 0, :] = 1.0
        body["_clicks"] = (body["document_labels"] != 0).sum(-1).max(-1).astype(np.int32)
    bs = corpus.batches(B, seed=2)
    mt2 = Multitask(default_args("CARS", src_vocab_size=V, tgt_vocab_size=300))
    fill_module_(mt2.network, 1013)
    mt2.cuda()
    single = {}
    StreamingSessionPredictor(mt2, N, 4, 16, B, max_session_len=6, lanes=2, slots=2).run(
        corpus, bs, on_result=lambda k, idx, p: single.__setitem__(k, p.clone()))
    labs = torch.cat([corpus.batch_tensors(b)["document_labels"].reshape(-1, N) for b in bs]).numpy()
    smap = MAP(rank_candidates(torch.cat([single[k].reshape(-1, N) for k in range(len(bs))]).numpy()), labs)
    for mode, macro in (("batch", 1), ("pair", 1), ("pair", 2)):
        plan = sharding.StreamShardPlan(world, rank, mode, batch_size=B)
        sp = StreamingSessionPredictor(mt2, N, 4, 16, macro * B, max_session_len=6, lanes=2, slots=2, macro=macro, plan=plan,
                                       gather="auto" if backend == "gloo" else "device")     # (a 1-rank world would default to no gather)
        assert sp.gather == ("host" if backend == "gloo" else "device") and sp.B == (macro * B // world if mode == "pair" else B)
        stream, _ = sp.merge_batches(corpus, bs, macro) if macro > 1 else (bs, [])
        got = {}
        st = sp.run(corpus, stream, on_result=lambda k, idx, p: got.__setitem__(k, (list(idx), p.clone())))
        assert sorted(got) == list(range(len(stream))) and st["batches"] == plan.rounds(len(stream)), (mode, len(got), len(stream))
        for k, (idx, p) in got.items():
            assert idx == list(stream[k])
            for j in range(macro):
                ref = single[bs.index(idx[j * B:(j + 1) * B])]
                assert float((p[j * B:(j + 1) * B] - ref).abs().max()) < 1e-6, (mode, macro, k, j)
        if macro == 1:
            gmap = MAP(rank_candidates(torch.cat([got[k][1].reshape(-1, N) for k in range(len(bs))]).numpy()), labs)
            assert gmap == smap, (mode, gmap, smap)
        tot = torch.tensor([float(st["pairs"])], device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tot)
        assert int(tot.item()) == sum(len(b) * int(corpus.lengths[b[0]]) * N for b in stream), mode     # every pair scored exactly once
    dist.barrier()
    if dist.get_rank() == 0:
        print("SHARDED_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
