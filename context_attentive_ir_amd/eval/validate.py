"""validate_official -- the evaluation loop that drives the hot path, with the reference's result keys and
averaging (/root/reference/main/ranker.py:236-297 for the rankers, main/multitask.py:262-332 for CARS' ranking
metrics): per batch, rank the candidates by descending predicted score and compute MAP / MRR / P@1,3,5; the result
is the plain mean of the per-batch values (AverageMeter.update(val) with n = 1).

Where the reference synchronises every batch (`scores.cpu().numpy()` right after `predict`), this loop keeps
`depth` batches in flight: batch k's scores are copied D2H into a pinned buffer behind an event while batches k+1..
are already being scored, and the (host-side) metrics of batch k are computed when its event has fired.
CARS: rows are the B*S (session, query) pairs, exactly as main/multitask.py:282-287 flattens them; the
suggestion metrics (BLEU/ROUGE/EM/F1) need `decode`, which is outside the hot path.
"""
import collections

import numpy as np
import torch

from .ltorank import MAP, MRR, precision_at_k, rank_candidates

_KS = (1, 3, 5)


def _metrics(scores, labels):
    pred = rank_candidates(scores)
    out = {"map": MAP(pred, labels), "mrr": MRR(pred, labels)}
    for k in _KS:
        if pred.shape[1] >= k:
            out["prec@%d" % k] = precision_at_k(pred, labels, k)
    return out


def validate_official(data_loader, model, depth=4):
    """data_loader: iterable of batch dicts (inputters.*_batchify layout); model: wrappers.Ranker / Multitask or a
    graph_runner.GraphedPredictor-like object with .predict(ex).  Returns {'map','mrr','prec@1','prec@3','prec@5',
    'examples'} (a prec@k entry is absent when the candidate lists are shorter than k, where the reference asserts)."""
    sums, nb, examples = collections.defaultdict(float), 0, 0
    inflight = collections.deque()

    def retire():
        nonlocal nb, examples
        host, ev, labels = inflight.popleft()
        if ev is not None:
            ev.synchronize()
        for k, v in _metrics(host.numpy(), labels).items():
            sums[k] += v
        nb += 1
        examples += labels.shape[0]

    with torch.no_grad():
        for ex in data_loader:
            out = model.predict(ex)
            if isinstance(out, dict):                       # Multitask.predict
                scores = out["click_scores"]
                labels = ex["document_labels"]
            else:
                scores, labels = out, ex["label"]
            scores = scores.reshape(-1, scores.shape[-1])
            labels = labels.reshape(-1, labels.shape[-1]).cpu().numpy()
            if scores.is_cuda:
                host = torch.empty(scores.shape, dtype=scores.dtype).pin_memory()
                host.copy_(scores, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            else:
                host, ev = scores, None
            inflight.append((host, ev, labels))
            while len(inflight) > depth:
                retire()
        while inflight:
            retire()
    result = {k: v / nb for k, v in sums.items()} if nb else {}
    result["examples"] = examples
    return result


def reference_loop(batches, model, iters=None, suggest=True):
    """The reference's validation loop AS WRITTEN (main/multitask.py:280-290 for a Multitask wrapper, main/ranker.py:254-262 for a Ranker): one
    `model.predict(ex)` per batch, `scores.cpu().numpy()`, argsort, MAP / MRR / P@1,3,5 -- every batch is synchronised on before the next
    one is issued, nothing else is in flight.  What `tools/dropin_loop.py` and bench.py's `dropin_loop_*` records time.  batches: list of
    batch dicts (cycled for `iters` calls); -> list of per-batch MAP values."""
    maps = []
    n = len(batches) if iters is None else int(iters)
    session = hasattr(model, "tgt_dict")
    with torch.no_grad():
        for i in range(n):
            ex = batches[i % len(batches)]
            if session:
                rows = ex["source_words"].shape[0] * ex["source_words"].shape[1]
                outputs = model.predict(ex) if suggest else model.predict(ex, suggest=False)
                scores = outputs["click_scores"].view(rows, -1).contiguous()
                labels = ex["document_labels"].view(rows, -1).contiguous().numpy()
            else:
                scores = model.predict(ex)
                labels = ex["label"].numpy()
            predictions = np.argsort(-scores.cpu().numpy())
            maps.append(MAP(predictions, labels))
            MRR(predictions, labels)
            precision_at_k(predictions, labels, 1)
            if predictions.shape[1] >= 3:
                precision_at_k(predictions, labels, 3)
            if predictions.shape[1] >= 5:
                precision_at_k(predictions, labels, 5)
            if session and suggest and outputs["predictions"] is not None:
                outputs["predictions"].cpu()                 # (the reference turns the token ids into strings on the host here)
    return maps
