"""The reference's validation loop, timed: one `model.predict(ex)` per batch, `scores.cpu().numpy()`, argsort, MAP / MRR / P@1,3,5 -- nothing
else in flight (main/multitask.py:280-290, main/ranker.py:254-262) -- on pinned host batches as the reference's DataLoader(pin_memory=True)
hands them over.  Modes: r5 = round 5's defaults (blocking id check per call, eager launches); deferred_eager = the pinned error word, eager;
default = round 6's defaults (pinned error word + the shape-keyed hipGraph cache inside predict()).

    python tools/dropin_loop.py [--iters 200] [--model CARS|MATCH_TENSOR] [--batch 16,128] [--decode 0,1]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from context_attentive_ir_amd import synth  # noqa: E402
from context_attentive_ir_amd.config import default_args  # noqa: E402
from context_attentive_ir_amd.detinit import fill_module_  # noqa: E402
from context_attentive_ir_amd.eval.validate import reference_loop  # noqa: E402
from context_attentive_ir_amd.wrappers import Multitask, Ranker  # noqa: E402


def breakdown(model, batches, iters, suggest):
    """where a loop iteration goes on the host: time inside predict() (enqueue), inside `.cpu()` (the device finishing + D2H), the rest (numpy,
    metrics) -- perf_counter stamps around the same statements as eval.validate.reference_loop"""
    import numpy as np
    from context_attentive_ir_amd.eval.ltorank import MAP, MRR, precision_at_k
    tp = tc = tm = 0.0
    with torch.no_grad():
        for i in range(iters):
            ex = batches[i % len(batches)]
            rows = ex["source_words"].shape[0] * ex["source_words"].shape[1]
            t0 = time.perf_counter()
            outputs = model.predict(ex, suggest=suggest)
            t1 = time.perf_counter()
            scores = outputs["click_scores"].view(rows, -1).contiguous()
            host = scores.cpu()
            t2 = time.perf_counter()
            labels = ex["document_labels"].view(rows, -1).contiguous().numpy()
            predictions = np.argsort(-host.numpy())
            MAP(predictions, labels), MRR(predictions, labels), precision_at_k(predictions, labels, 1), precision_at_k(predictions, labels, 3), precision_at_k(predictions, labels, 5)
            t3 = time.perf_counter()
            tp, tc, tm = tp + t1 - t0, tc + t2 - t1, tm + t3 - t2
    return {"predict_us": round(tp / iters * 1e6, 1), "cpu_wait_us": round(tc / iters * 1e6, 1), "metrics_us": round(tm / iters * 1e6, 1)}


def set_mode(model, mode):
    model.id_check_interval = 1
    model.id_check = "blocking" if mode == "r5" else "deferred"
    model.args.predict_graphs = mode == "default"
    model.clear_predict_graphs()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--model", default="CARS")
    ap.add_argument("--batch", default="16,128")
    ap.add_argument("--decode", default="0,1")
    ap.add_argument("--cands", type=int, default=10)
    ap.add_argument("--modes", default="r5,deferred_eager,default")
    ap.add_argument("--vocab", type=int, default=100000)
    ap.add_argument("--breakdown", action="store_true")
    a = ap.parse_args()
    V = a.vocab
    is_sess = a.model in ("CARS", "MNSRF", "M_MATCH_TENSOR")
    if is_sess:
        model = Multitask(default_args(a.model, src_vocab_size=V, tgt_vocab_size=30000))
    else:
        model = Ranker(default_args(a.model, src_vocab_size=V, max_query_len=4, max_doc_len=64))
    fill_module_(model.network, 1013)
    model.cuda()
    for B in [int(x) for x in a.batch.split(",")]:
        if is_sess:
            batches = [{k: v.pin_memory() for k, v in synth.session_batch(B, 7, a.cands, 4, 64, V, seed=50 + i).items()} for i in range(8)]
            pairs = B * 7 * a.cands
        else:
            batches = [{k: v.pin_memory() for k, v in synth.ranker_batch(B, a.cands, 4, 64, V, seed=50 + i).items()} for i in range(8)]
            pairs = B * a.cands
        for dec in ([int(x) for x in a.decode.split(",")] if is_sess else [0]):
            ref = None
            for mode in a.modes.split(","):
                set_mode(model, mode)
                run = lambda n: reference_loop(batches, model, n, suggest=bool(dec))      # noqa: E731
                run(16)
                torch.cuda.synchronize()
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    maps = run(a.iters)
                    dt = (time.perf_counter() - t0) / a.iters
                    best = dt if best is None else min(best, dt)
                if ref is None:
                    ref = maps
                extra = breakdown(model, batches, a.iters, bool(dec)) if (a.breakdown and is_sess) else None
                print(json.dumps({"breakdown": extra, "model": a.model, "batch": B, "cands": a.cands, "decode": dec, "mode": mode, "ms_per_call": round(best * 1e3, 4),
                                  "pairs_per_s": round(pairs / best, 1), "map_equal_first_mode": maps == ref,
                                  "graphs": None if model._graphs is None else [model._graphs.captures, model._graphs.replays]}), flush=True)
    model.check_ids()


if __name__ == "__main__":
    main()
