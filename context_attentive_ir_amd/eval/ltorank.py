"""Ranking metrics with the reference's definitions (/root/reference/neuroir/eval/ltorank.py:4-47,104-123),
vectorised with numpy: AP is taken over ALL ranked candidates (so MAP@10 <=> N=10) and a row needs at least
one relevant candidate (the reference divides by the number of relevant documents)."""
import numpy as np


def rank_candidates(scores):
    """Descending ranking of the candidates of each query (main/ranker.py:257).  Stable, so ties resolve by
    candidate index on every platform (the reference's default argsort is not stable, Appendix E6)."""
    return np.argsort(-np.asarray(scores), axis=-1, kind="stable")


def _hits(predictions, target):
    predictions, target = np.asarray(predictions), np.asarray(target)
    assert predictions.shape == target.shape and predictions.ndim == 2
    return np.take_along_axis(target, predictions, axis=1) == 1


def MAP(predictions, target):
    hit = _hits(predictions, target)
    nrel = hit.sum(1)
    if (nrel == 0).any():
        raise ZeroDivisionError("MAP needs at least one relevant candidate per query")
    prec = np.cumsum(hit, 1) / np.arange(1, hit.shape[1] + 1)
    return float(((prec * hit).sum(1) / nrel).mean())


def MRR(predictions, target):
    hit = _hits(predictions, target)
    first = np.where(hit.any(1), hit.argmax(1), -1)
    return float(np.where(first >= 0, 1.0 / (first + 1), 0.0).mean())


def precision_at_k(predictions, target, k):
    hit = _hits(predictions, target)
    assert hit.shape[1] >= k, "Precision@K cannot be computed, invalid value of K."
    return float(hit[:, :k].sum(1).mean() / k)
