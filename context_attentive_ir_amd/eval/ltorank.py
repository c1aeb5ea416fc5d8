"""Ranking metrics with the reference's definitions (/root/reference/neuroir/eval/ltorank.py:4-47,104-123),
vectorised with numpy: AP is taken over ALL ranked candidates (so MAP@10 <=> N=10) and a row needs at least
one relevant candidate (the reference divides by the number of relevant documents)."""
import numpy as np


def rank_candidates(scores):
    """Descending ranking of the candidates of each query (main/ranker.py:257).  Stable, so ties resolve by
    candidate index on every platform (the reference's default argsort is not stable, Appendix E6)."""
    return np.argsort(-np.asarray(scores), axis=-1, kind="stable")


_LAST = [None, None, None]


def _hits(predictions, target):
    """hit[r, j] = the j-th ranked candidate of row r is relevant.  The reference's loops call MAP, MRR and precision_at_k (x3) on the SAME
    (predictions, target) pair per batch (main/ranker.py:258-262): the gather is memoised on the identity of the last pair (the arrays are
    kept referenced, so an id cannot be recycled; callers do not mutate them between the five calls)."""
    if _LAST[0] is predictions and _LAST[1] is target:
        return _LAST[2]
    p, t = np.asarray(predictions), np.asarray(target)
    assert p.shape == t.shape and p.ndim == 2
    hit = np.take_along_axis(t, p, axis=1) == 1
    _LAST[0], _LAST[1], _LAST[2] = predictions, target, hit
    return hit


_RANKS = {}
_DT = {np.dtype(np.float32): 0, np.dtype(np.int64): 1, np.dtype(np.float64): 2}
_FN = [None]


def _native(what, predictions, target, k=0):
    """the metric from the library's host loop (nir_host_rank_metric: one C pass over the batch instead of ~10 numpy calls of ~3 us each), or
    None when the arrays are not plain [rows, n] int64 / float32|int64|float64 host arrays (then the numpy form below runs)."""
    if not (type(predictions) is np.ndarray and type(target) is np.ndarray and predictions.dtype == np.int64 and predictions.ndim == 2
            and predictions.shape == target.shape and predictions.flags.c_contiguous and target.flags.c_contiguous and predictions.size):
        return None
    code = _DT.get(target.dtype)
    if code is None:
        return None
    fn = _FN[0]
    if fn is None:
        from .. import lib
        fn = _FN[0] = lib.load().nir_host_rank_metric
    rows, n = predictions.shape
    v = fn(what, predictions.__array_interface__["data"][0], target.__array_interface__["data"][0], code, rows, n, k)
    if v == -2.0:
        return None
    return v


def MAP(predictions, target):
    v = _native(0, predictions, target)
    if v is not None:
        if v == -1.0:
            raise ZeroDivisionError("MAP needs at least one relevant candidate per query")
        return v
    hit = _hits(predictions, target)
    nrel = hit.sum(1)
    if not nrel.all():
        raise ZeroDivisionError("MAP needs at least one relevant candidate per query")
    n = hit.shape[1]
    inv = _RANKS.get(n)
    if inv is None:
        inv = _RANKS[n] = 1.0 / np.arange(1, n + 1)
    prec = np.cumsum(hit, 1) * inv
    return float(((prec * hit).sum(1) / nrel).mean())


def MRR(predictions, target):
    v = _native(1, predictions, target)
    if v is not None:
        return v
    hit = _hits(predictions, target)
    first = hit.argmax(1)
    rr = 1.0 / (first + 1)
    rr[~hit[np.arange(hit.shape[0]), first]] = 0.0          # a row without a relevant candidate
    return float(rr.mean())


def precision_at_k(predictions, target, k):
    assert np.shape(predictions)[1] >= k, "Precision@K cannot be computed, invalid value of K."
    v = _native(2, predictions, target, int(k))
    if v is not None:
        return v
    hit = _hits(predictions, target)
    assert hit.shape[1] >= k, "Precision@K cannot be computed, invalid value of K."
    return float(hit[:, :k].sum() / (k * hit.shape[0]))
