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
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    V = 1500
    # rankers: N = 7 candidates over 2 ranks (4 + 3, padded shard)
    ex = synth.ranker_batch(5, 7, 4, 33, V, seed=3, full_length=False)
    for kind in ("MATCH_TENSOR", "ESM"):
        r = Ranker(default_args(kind, src_vocab_size=V))
        fill_module_(r.network, 1013)
        r.cuda()
        full = r.predict(ex).cpu()
        r.parallelize()
        shard = r.predict(ex).cpu()
        assert torch.allclose(full, shard, atol=1e-6), (kind, float((full - shard).abs().max()))
    # CARS: candidate-sharded document encoding + all-gather of pooled documents, session part replicated
    sex = synth.session_batch(3, 4, 5, 4, 21, V, seed=5, full_length=False, multi_click=True)
    mt = Multitask(default_args("CARS", src_vocab_size=V, tgt_vocab_size=300))
    fill_module_(mt.network, 1013)
    mt.cuda()
    full = mt.predict(sex, suggest=False)["click_scores"].cpu()
    mt.parallelize()
    shard = mt.predict(sex, suggest=False)["click_scores"].cpu()
    assert torch.allclose(full, shard, atol=1e-6), float((full - shard).abs().max())
    dist.barrier()
    if dist.get_rank() == 0:
        print("SHARDED_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
