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
