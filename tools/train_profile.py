"""The training step of one model (Ranker.update / Multitask.update, eager) on its bench batch shape, in a loop -- for
rocprofv3 --kernel-trace --stats (cd /tmp; TMPDIR=/tmp): which kernels the step is made of, HIP operators and tensor glue alike.

    python tools/train_profile.py [--kind MATCH_TENSOR|DUET|DRMM|CARS] [--steps 10]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="MATCH_TENSOR")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--graphed", action="store_true", help="the step as one hipGraph (wrappers.GraphedUpdate)")
    a = ap.parse_args()
    cfg = {"MATCH_TENSOR": "C2_match_tensor", "DUET": "C4_duet", "DRMM": "C4_drmm", "CARS": bench.HEADLINE, "MNSRF": "X3_mnsrf",
           "M_MATCH_TENSOR": "X3_m_match_tensor"}[a.kind]
    c = dict(bench.CONFIGS[cfg])
    if a.kind == "DUET":
        c.update(batch=8, cands=10)              # the C4 inference batch (3 200 x 290) does not fit a training step's activations
    dev = torch.device("cuda:0")
    extra = dict(optimizer="adam", learning_rate=0.001, weight_decay=0, momentum=0, grad_clipping=10.0, fix_embeddings=True)
    from helpers import default_args, fill_module_
    from context_attentive_ir_amd.wrappers import Multitask, Ranker
    multi = a.kind in ("CARS", "MNSRF", "M_MATCH_TENSOR")
    if multi:
        w = Multitask(default_args(a.kind, src_vocab_size=c["vocab"], tgt_vocab_size=30000, **extra))
    else:
        w = Ranker(default_args(a.kind, src_vocab_size=c["vocab"], max_query_len=c["qlen"], max_doc_len=c["dlen"], **extra))
    fill_module_(w.network, 1013)
    w.cuda()
    w.init_optimizer()
    w.id_check_interval = 0
    batches = bench.make_batches(c, 4, 0, dev)
    if multi:                                   # teacher-forcing targets as in bench.train_record
        for b in batches:
            src = b["source_words"][:, 1:]
            B_, S1, QL = src.shape
            tw = torch.zeros(B_, S1, QL + 2, dtype=torch.int64, device=dev)
            tw[..., 0] = 2
            tw[..., 1:QL + 1] = src
            tw[..., QL + 1] = 3
            b["target_words"], b["target_seq"] = tw, tw % 30000
            b["target_lens"] = torch.full((B_, S1), QL + 2, dtype=torch.int64, device=dev)
    step = w.update
    if a.graphed:
        from context_attentive_ir_amd.wrappers import GraphedUpdate
        step = GraphedUpdate(w)
    for i in range(3):
        step(batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(batches[i % 4])
    torch.cuda.synchronize()
    print("%s.update: %.3f ms per step (%s)" % (a.kind, (time.perf_counter() - t0) / a.steps * 1e3, "graphed" if a.graphed else "eager"))


if __name__ == "__main__":
    main()
