"""aten-level view of one eager training step (torch.profiler, CUDA activities): every op that launched device work, with its input shapes and
the Python line that called it -- where the tensor glue of the step comes from.  python tools/train_ops.py [CARS|MATCH_TENSOR]"""
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "CARS"
    cfg = {"MATCH_TENSOR": "C2_match_tensor", "CARS": bench.HEADLINE}[kind]
    c = dict(bench.CONFIGS[cfg])
    dev = torch.device("cuda:0")
    extra = dict(optimizer="adam", learning_rate=0.001, weight_decay=0, momentum=0, grad_clipping=10.0, fix_embeddings=True)
    from helpers import default_args, fill_module_
    from context_attentive_ir_amd.wrappers import Multitask, Ranker
    if kind == "CARS":
        w = Multitask(default_args("CARS", src_vocab_size=c["vocab"], tgt_vocab_size=30000, **extra))
    else:
        w = Ranker(default_args(kind, src_vocab_size=c["vocab"], max_query_len=c["qlen"], max_doc_len=c["dlen"], **extra))
    fill_module_(w.network, 1013)
    w.cuda(); w.init_optimizer(); w.id_check_interval = 0
    batches = bench.make_batches(c, 2, 0, dev)
    if kind == "CARS":
        for b in batches:
            src = b["source_words"][:, 1:]
            B_, S1, QL = src.shape
            tw = torch.zeros(B_, S1, QL + 2, dtype=torch.int64, device=dev)
            tw[..., 0] = 2; tw[..., 1:QL + 1] = src; tw[..., QL + 1] = 3
            b["target_words"], b["target_seq"] = tw, tw % 30000
            b["target_lens"] = torch.full((B_, S1), QL + 2, dtype=torch.int64, device=dev)
    for i in range(3):
        w.update(batches[i % 2])
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False, with_stack=True) as prof:
        w.update(batches[0])
        torch.cuda.synchronize()
    rows = defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
        if dt <= 0 or not ev.name.startswith("aten::"):
            continue
        site = ""
        for fr in (ev.stack or []):
            if "context_attentive_ir_amd" in fr and "autograd.py" not in fr.split(":")[0][-12:]:
                site = fr.split("context_attentive_ir_amd/")[-1][:70]
                break
        if not site:
            for fr in (ev.stack or []):
                if "context_attentive_ir_amd" in fr:
                    site = fr.split("context_attentive_ir_amd/")[-1][:70]
                    break
        key = (ev.name, str(ev.input_shapes)[:60], site or ("<autograd engine>" if not ev.stack else ev.stack[0][-60:]))
        rows[key][0] += 1
        rows[key][1] += dt
    tot = sum(v[1] for v in rows.values())
    print("%d aten ops with device time, %.0f us" % (sum(v[0] for v in rows.values()), tot))
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1])[:70]:
        print("%4d %8.1f  %-28s %-60s %s" % (v[0], v[1], k[0], k[1], k[2]))


if __name__ == "__main__":
    main()
