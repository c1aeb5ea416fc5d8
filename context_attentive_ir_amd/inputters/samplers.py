"""Length-bucketing batch samplers with the reference's batch composition
(/root/reference/neuroir/inputters/ranker/data.py:37-56 and neuroir/inputters/multitask/data.py:42-72), so that every
batch has (nearly) one shape -- which is also what lets graph_runner.GraphedPredictor replay one captured hipGraph per
shape bucket.

* rankers: examples sorted by (longest document desc, query length desc, random tie-break), cut into consecutive
  batches, the batches (not their contents) shuffled;
* sessions (CARS): examples clustered by session length, each cluster cut into full batches (the remainder of a
  cluster is dropped, as in the reference), the batches shuffled.

`rng` is anything with numpy's legacy `random_sample` / `shuffle` (the `numpy.random` module itself by default): with the
same seed these functions draw the same numbers in the same order as the reference's samplers, so the batches are
identical -- tests/golden/samplers.npz pins that.  Vectorised (one lexsort) instead of a python tuple per example.
"""
import numpy as np


def length_sorted_batches(lengths, batch_size, shuffle=True, rng=np.random):
    """lengths: [(max_doc_len, query_len)] per example -> list of index arrays, one per batch."""
    lengths = np.asarray(lengths, dtype=np.int64).reshape(-1, 2)
    tie = rng.random_sample(len(lengths))
    order = np.lexsort((tie, -lengths[:, 1], -lengths[:, 0]))          # primary key last
    batches = [order[i:i + batch_size] for i in range(0, len(order), batch_size)]
    if shuffle:
        rng.shuffle(batches)
    return batches


def session_length_batches(session_lengths, batch_size, shuffle=True, rng=np.random):
    """session_lengths: number of queries per session -> list of index lists; every batch holds sessions of ONE length."""
    session_lengths = np.asarray(session_lengths, dtype=np.int64)
    batches = []
    _, first = np.unique(session_lengths, return_index=True)
    for key in session_lengths[np.sort(first)]:                         # clusters in order of first appearance (dict order)
        idx = np.flatnonzero(session_lengths == key)
        full = len(idx) // batch_size * batch_size
        batches.extend(idx[i:i + batch_size].tolist() for i in range(0, full, batch_size))
    if shuffle:
        rng.shuffle(batches)
    return batches


def flat_indices(batches):
    """The reference's samplers hand torch's DataLoader one flat index sequence (batch after batch)."""
    return [int(i) for b in batches for i in b]
