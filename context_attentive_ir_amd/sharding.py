"""Candidate-axis sharding over the GPUs of one node.

Every (query, candidate) pair of the rankers is independent given the query, so rank g scores the slice
doc_rep[:, n_g:n_{g+1}] of the N candidates (queries, weights and the embedding table are replicated) and the
only exchange is ONE all-gather of fp32 scores [B, ceil(N/G)] -> [B, N] -- KB-sized, latency-bound, so a single
all_gather_into_tensor (RCCL over the xGMI mesh when the group's backend is 'nccl'; 'gloo' in the CPU tests).
N not divisible by G: shards are padded to ceil(N/G) by repeating the last candidate, the padded scores are
dropped after the gather.  Softmax over N, losses and MAP run after the gather, replicated.
"""
import torch
import torch.distributed as dist


def shard_bounds(N, world, rank):
    """Half-open candidate range [lo, hi) of `rank` and the padded per-rank width."""
    per = (N + world - 1) // world
    lo = min(rank * per, N)
    return lo, min(lo + per, N), per


def shard_candidates(doc_rep, doc_len, world, rank):
    """Slice [B,N,DL]/[B,N] along candidates and pad to the common width (repeat last real candidate)."""
    N = doc_rep.shape[1]
    lo, hi, per = shard_bounds(N, world, rank)
    if hi > lo:
        d, l = doc_rep[:, lo:hi], doc_len[:, lo:hi]
    else:  # rank beyond the candidates (N < world): score a dummy copy of candidate 0, dropped later
        d, l = doc_rep[:, :1], doc_len[:, :1]
    pad = per - d.shape[1]
    if pad > 0:
        d = torch.cat([d, d[:, -1:].expand(-1, pad, -1)], 1)
        l = torch.cat([l, l[:, -1:].expand(-1, pad)], 1)
    return d.contiguous(), l.contiguous()


def gather_scores(local, N, group=None):
    """all-gather [B,per] from every rank -> [B,N] on every rank (padding removed)."""
    world = dist.get_world_size(group)
    B, per = local.shape
    out = torch.empty(world * B, per, device=local.device, dtype=local.dtype)   # rank-major concatenation
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out.view(world, B, per).permute(1, 0, 2).reshape(B, world * per)[:, :N].contiguous()


def gathered_softmax(local, N, group=None):
    """Blocking all-gather of the score shards [B,per] + softmax over the N candidates -> [B,N] on every rank.
    The softmax reads the rank-major gather buffer directly (nir_softmax_gathered: no permute/slice copies).  Device
    tensors only -- there is no CPU arithmetic in this package (the gloo tests exercise gather_scores / ScoreGather.wait,
    which only move data)."""
    if not local.is_cuda:
        raise RuntimeError("gathered_softmax needs score shards on a ROCm device (no CPU fallback)")
    from . import lib
    world = dist.get_world_size(group)
    B, per = local.shape
    buf = torch.empty(world * B, per, device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(buf, local.contiguous(), group=group)
    probs = torch.empty(B, N, device=local.device, dtype=local.dtype)
    lib.check(lib.load().nir_softmax_gathered(lib.ptr(buf), lib.ptr(probs), None, world, B, per, N, lib.stream()),
              "nir_softmax_gathered")
    return probs


class ScoreGather(object):
    """Asynchronous form of gather_scores: the all-gather is issued on the backend's communication stream when
    the object is built and `wait()` / `softmax()` (typically called one batch later) consume it.  Between the two
    the caller's stream is free, so batch k's gather overlaps batch k+1's scoring (SURVEY.md section 8e:
    "overlap batch k's gather with batch k+1's document encoding").  `out` may be a caller-owned, reused
    [world*B, per] buffer (no allocation per batch)."""

    def __init__(self, local, N, group=None, out=None):
        world = dist.get_world_size(group)
        self.B, self.per = local.shape
        self.N, self.world = N, world
        self.local = local.contiguous()                  # kept alive until wait()
        self.out = out if out is not None else torch.empty(world * self.B, self.per, device=local.device, dtype=local.dtype)
        self.work = dist.all_gather_into_tensor(self.out, self.local, group=group, async_op=True)

    def wait(self):
        """-> raw scores [B,N] (query-major, padding removed)."""
        self.work.wait()                                 # device backends: the current stream waits, the host does not
        s = self.out.view(self.world, self.B, self.per).permute(1, 0, 2).reshape(self.B, self.world * self.per)
        return s[:, :self.N].contiguous()

    def softmax(self, probs=None):
        """-> softmax over the N gathered candidates, [B,N]: one kernel straight off the rank-major gather buffer
        (device tensors only; no CPU fallback)."""
        if not self.out.is_cuda:
            raise RuntimeError("ScoreGather.softmax needs device tensors (no CPU fallback); use wait() to get the raw scores")
        from . import lib
        self.work.wait()
        if probs is None:
            probs = torch.empty(self.B, self.N, device=self.out.device, dtype=self.out.dtype)
        lib.check(lib.load().nir_softmax_gathered(lib.ptr(self.out), lib.ptr(probs), None, self.world, self.B, self.per, self.N,
                                                  lib.stream()), "nir_softmax_gathered")
        return probs


def sharded_scores(score_fn, doc_rep, doc_len, group=None):
    """score_fn(doc_shard [B,per,DL], len_shard [B,per]) -> [B,per]; returns the full [B,N] on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return score_fn(doc_rep, doc_len)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    d, l = shard_candidates(doc_rep, doc_len, world, rank)
    return gather_scores(score_fn(d, l), doc_rep.shape[1], group)


# ----------------------------------------------------------------------------------------------------------
# CARS: document encoding (~96 % of the FLOPs) shards by candidate exactly like the rankers,
# but encode_clicks needs ALL N pooled document vectors of a query, so the exchange is one all-gather of the pooled
# vectors [B,S,ceil(N/G),D] (C3: 229 KB per rank); click attention, session LSTMs and the ranknet then run
# replicated on every rank (cheap) and produce the full [B,S,N] scores with no second collective.
# ----------------------------------------------------------------------------------------------------------
def shard_session_candidates(document_words, document_lens, world, rank):
    """[B,S,N,DL] / [B,S,N] -> this rank's candidate slice, padded to the common width."""
    B, S, N, DL = document_words.shape
    d, l = shard_candidates(document_words.reshape(B * S, N, DL), document_lens.reshape(B * S, N), world, rank)
    per = d.shape[1]
    return d.view(B, S, per, DL), l.view(B, S, per)


def gather_pooled_docs(local, N, group=None):
    """all-gather pooled document vectors [B,S,per,D] from every rank -> [B,S,N,D] (padding removed)."""
    world = dist.get_world_size(group)
    B, S, per, D = local.shape
    out = torch.empty(world * B * S, per * D, device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(out, local.reshape(B * S, per * D).contiguous(), group=group)
    return out.view(world, B, S, per, D).permute(1, 2, 0, 3, 4).reshape(B, S, world * per, D)[:, :, :N].contiguous()


def sharded_pooled_docs(encode_fn, document_words, document_lens, group=None, return_local=False):
    """encode_fn(doc_shard [B,S,per,DL], len_shard [B,S,per]) -> pooled [B,S,per,D]; returns [B,S,N,D] everywhere
    (return_local: also this rank's own pooled shard [B,S,per,D], None without a process group -- the ranker then scores the
    shard only and the score slices are gathered with gather_session_scores)."""
    if not (dist.is_available() and dist.is_initialized()):
        full = encode_fn(document_words, document_lens)
        return (full, None) if return_local else full
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    d, l = shard_session_candidates(document_words, document_lens, world, rank)
    local = encode_fn(d, l)
    full = gather_pooled_docs(local, document_words.shape[2], group)
    return (full, local) if return_local else full


def gather_session_scores(local, N, group=None):
    """all-gather click-score slices [B,S,per] from every rank -> [B,S,N] on every rank (padding removed)."""
    B, S, per = local.shape
    return gather_scores(local.reshape(B * S, per), N, group).view(B, S, N)
