"""Synthetic MSMARCO-shaped sessions: ids 0-3 reserved (PAD/UNK/BOS/EOS,
neuroir/inputters/constants.py:1-4), token ids Zipf(s~1.07) over [4, V), zero padding after the true length,
exactly one clicked candidate per query unless `multi_click`.  Layouts are the batchify() contracts of
neuroir/inputters/ranker/vector.py:39-90 and neuroir/inputters/multitask/vector.py:82-149."""
import numpy as np
import torch


def _zipf_ids(rng, shape, V, s=1.07):
    # inverse-CDF sampling of a truncated Zipf over ranks 1..V-4 (vectorised, deterministic given rng)
    n = V - 4
    u = rng.random(size=shape)
    if abs(s - 1.0) < 1e-9:
        r = np.exp(u * np.log(n))
    else:
        a = 1.0 - s
        r = (u * (n ** a - 1.0) + 1.0) ** (1.0 / a)
    return (np.clip(r.astype(np.int64), 1, n) - 1) + 4


def _pad(ids, lens):
    pos = np.arange(ids.shape[-1])
    ids[pos >= lens[..., None]] = 0
    return ids


def _lengths(rng, shape, L, full, mean, std, lo):
    if full:
        return np.full(shape, L, np.int64)
    return np.clip(np.rint(rng.normal(mean, std, size=shape)), min(lo, L), L).astype(np.int64)


def ranker_batch(B, N, QL, DL, V, seed=1013, full_length=True, uniform=False, multi_click=False):
    """-> dict(que_rep [B,QL], que_len [B], doc_rep [B,N,DL], doc_len [B,N], label [B,N]) of int64 CPU tensors."""
    rng = np.random.default_rng(seed)
    draw = (lambda sh: rng.integers(4, V, size=sh, dtype=np.int64)) if uniform else (lambda sh: _zipf_ids(rng, sh, V))
    qlen = _lengths(rng, (B,), QL, full_length, 3.84, 1.5, 1)
    dlen = _lengths(rng, (B, N), DL, full_length, 63.4, 25.0, 8)
    q, d = _pad(draw((B, QL)), qlen), _pad(draw((B, N, DL)), dlen)
    lab = np.zeros((B, N), np.int64)
    for b in range(B):
        k = int(rng.integers(1, 7)) if multi_click else 1
        lab[b, rng.choice(N, min(k, N), replace=False)] = 1
    return {k: torch.from_numpy(v) for k, v in
            dict(que_rep=q, que_len=qlen, doc_rep=d, doc_len=dlen, label=lab).items()}


def session_batch(B, S, N, QL, DL, V, seed=1013, full_length=True, multi_click=False):
    """CARS layout: source_words [B,S,QL], source_lens [B,S], document_words [B,S,N,DL], document_lens [B,S,N],
    document_labels [B,S,N] float32."""
    rng = np.random.default_rng(seed)
    qlen = _lengths(rng, (B, S), QL, full_length, 3.84, 1.5, 1)
    dlen = _lengths(rng, (B, S, N), DL, full_length, 63.4, 25.0, 8)
    q, d = _pad(_zipf_ids(rng, (B, S, QL), V), qlen), _pad(_zipf_ids(rng, (B, S, N, DL), V), dlen)
    lab = np.zeros((B, S, N), np.float32)
    for b in range(B):
        for s in range(S):
            k = int(rng.integers(1, 7)) if multi_click else 1
            lab[b, s, rng.choice(N, min(k, N), replace=False)] = 1.0
    return {k: torch.from_numpy(v) for k, v in
            dict(source_words=q, source_lens=qlen, document_words=d, document_lens=dlen, document_labels=lab).items()}
