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


# ----------------------------------------------------------------------------------------------------------
# CARS, round 3: candidate-sharded encode -> ALL-TO-ALL -> SESSION-sharded tail -> all-gather of the click probabilities.
#
# With the all-gather above every rank still ran the query encoder, click pooling, both session LSTMs and the attention for ALL B
# sessions: a replicated tail that capped 8-GPU strong scaling at 1.9x (C3) / 4.0x (C5) before any link latency.  Sessions are
# independent of each other (cars.py:306-458 iterates the session axis with batch-parallel ops only; the one batch-wide quantity, the
# click mask's max click count m of cars.py:285-289, is a function of the replicated labels), so after encoding its ceil(N/G) candidates for
# all B sessions rank g needs the N pooled vectors of ITS ceil(B/G) sessions only:
#   stage A   queries of my sessions -> pooled_q [bper,S,D];   my candidate slice of all sessions -> pooled [G*bper,S,per,D]
#   exchange  all_to_all_single: chunk r of my buffer (the sessions of rank r) goes to rank r          ((G-1)/G of bper*G*S*per*D*4 B out)
#   stage B   [G,bper,S,per,D] -> [bper,S,N,D]; clicks + session LSTMs + ranknet + softmax for my sessions -> probs [bper,S,N]
#   gather    all_gather_into_tensor of the probabilities (KBs) -> [B,S,N] on every rank
# Nothing heavier than the KB-sized gather is replicated.  B or N not divisible by G: sessions / candidates are padded by repeating the
# last one and dropped after the exchange / gather.
# ----------------------------------------------------------------------------------------------------------
class SessionShardPlan(object):
    """axis:
      "candidate"  every rank encodes its slice of the N candidates of EVERY session (the split of SURVEY.md 8e), then the exchange above;
      "pair"       the flattened (session, candidate) axis is cut into G contiguous chunks ("shard the flattened B*N pair
                   axis in contiguous chunks -- equally valid").  With B % G == 0 a chunk is exactly B/G whole sessions x all N candidates:
                   every rank encodes 1/G of the candidate documents WITHOUT padding the candidate axis (N = 50 over 8 ranks pads to 56) and
                   already holds everything its sessions' tail needs -- the exchange disappears, only the all-gather of the scores remains.
                   Requires B % G == 0;
      "auto"       "pair" when B % G == 0, else "candidate"."""

    def __init__(self, B, S, N, world, rank, axis="candidate"):
        self.B, self.S, self.N, self.world, self.rank = int(B), int(S), int(N), int(world), int(rank)
        if axis == "auto":
            axis = "pair" if self.B % self.world == 0 else "candidate"
        if axis not in ("candidate", "pair"):
            raise ValueError("axis must be 'candidate', 'pair' or 'auto'")
        if axis == "pair" and self.B % self.world:
            raise ValueError("pair-axis sharding in whole sessions needs B %% world == 0 (B=%d, world=%d)" % (self.B, self.world))
        self.axis = axis
        self.aligned = axis == "pair"
        self.per = self.N if self.aligned else (self.N + self.world - 1) // self.world   # candidates per rank and session (padded)
        self.bper = (self.B + self.world - 1) // self.world         # sessions per rank (padded)
        self._ids = {}

    def session_ids(self, device):
        """LongTensor [G*bper]: session b of slot i (slots past B repeat the last session); rank r owns slots [r*bper, (r+1)*bper)."""
        k = str(device)
        if k not in self._ids:
            self._ids[k] = torch.clamp(torch.arange(self.world * self.bper, device=device), max=self.B - 1)
        return self._ids[k]

    def own(self, t):
        """rows of a [B, ...] tensor that belong to this rank's sessions -> [bper, ...]."""
        ids = self.session_ids(t.device)[self.rank * self.bper:(self.rank + 1) * self.bper]
        return t.index_select(0, ids).contiguous()

    def doc_shard(self, document_words, document_lens):
        """[B,S,N,DL] / [B,S,N] -> this rank's candidate slice of every (padded) session: [G*bper,S,per,DL] / [G*bper,S,per]
        (pair axis: all N candidates of this rank's own sessions, [bper,S,N,DL] / [bper,S,N])."""
        if self.aligned:
            return self.own(document_words), self.own(document_lens)
        d, l = shard_session_candidates(document_words, document_lens, self.world, self.rank)
        ids = self.session_ids(d.device)
        return d.index_select(0, ids).contiguous(), l.index_select(0, ids).contiguous()

    def exchange(self, pooled_shard, group=None, out=None):
        """all-to-all of the pooled candidate slices [G*bper,S,per,D]: afterwards out[r] = rank r's candidates of MY sessions,
        out [G,bper,S,per,D].  world 1: a view, no collective."""
        G, bper, S, per = self.world, self.bper, self.S, self.per
        D = pooled_shard.shape[-1]
        if self.aligned:                                     # pair axis: the rank already holds all N candidates of its sessions
            assert tuple(pooled_shard.shape) == (bper, S, per, D), tuple(pooled_shard.shape)
            return pooled_shard.view(1, bper, S, per, D)
        assert tuple(pooled_shard.shape) == (G * bper, S, per, D), tuple(pooled_shard.shape)
        if G == 1 and not (dist.is_available() and dist.is_initialized()):
            return pooled_shard.view(1, bper, S, per, D)
        if out is None:
            out = torch.empty(G, bper, S, per, D, device=pooled_shard.device, dtype=pooled_shard.dtype)
        dist.all_to_all_single(out.view(G * bper, S, per, D), pooled_shard.contiguous(), group=group)
        return out

    def exchange_via_gather(self, pooled_shard, group=None, out=None):
        """The same exchange through all_gather_into_tensor: every rank receives EVERY slice and keeps the chunk of its own sessions --
        G x the bytes of the all-to-all (C5 at 8 ranks: 25.7 MB in per rank and step instead of 3.2 MB), but a collective RCCL can replay
        from a captured hipGraph (RCCL 2.26.6: a captured all_to_all_single hangs, all_gather_into_tensor captures and replays;
        tools/rccl_capture_probe.py).  out (optional): [G, G*bper, S, per, D] gather buffer.  -> view [G,bper,S,per,D]."""
        G, bper, S, per = self.world, self.bper, self.S, self.per
        D = pooled_shard.shape[-1]
        if self.aligned:
            return self.exchange(pooled_shard, group)
        assert tuple(pooled_shard.shape) == (G * bper, S, per, D), tuple(pooled_shard.shape)
        if G == 1 and not (dist.is_available() and dist.is_initialized()):
            return pooled_shard.view(1, bper, S, per, D)
        if out is None:
            out = torch.empty(G, G * bper, S, per, D, device=pooled_shard.device, dtype=pooled_shard.dtype)
        dist.all_gather_into_tensor(out.view(G * G * bper, S, per, D), pooled_shard.contiguous(), group=group)
        return out[:, self.rank * bper:(self.rank + 1) * bper]

    def assemble(self, recv):
        """[G,bper,S,per,D] (rank-major candidate slices) -> pooled documents of my sessions [bper,S,N,D] (padding dropped)."""
        G, bper, S, per, D = recv.shape
        return recv.permute(1, 2, 0, 3, 4).reshape(bper, S, G * per, D)[:, :, :self.N].contiguous()

    def gather(self, local, group=None, out=None):
        """all-gather the per-rank blocks [bper,S,N] -> [B,S,N] on every rank (padded sessions dropped)."""
        G, bper = self.world, self.bper
        assert local.shape[0] == bper
        if G == 1 and not (dist.is_available() and dist.is_initialized()):
            return local[:self.B]
        if out is None:
            out = torch.empty((G * bper,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out[:self.B]

    def exchange_bytes(self, D, itemsize=4):
        """bytes this rank sends over xGMI per step for the pooled vectors (the chunk it keeps is excluded; pair axis: none)."""
        return 0 if self.aligned else (self.world - 1) * self.bper * self.S * self.per * D * itemsize


def session_sharded_click_probs(plan, encode_q, encode_docs, session_tail, ex, group=None, via_gather=False):
    """The whole sharded step in terms of three callables (the HIP entry points on a GPU; the CPU oracle in the gloo tests):
         encode_q(source_words [b,S,QL], source_lens [b,S])                  -> pooled queries [b,S,D]
         encode_docs(document_words [b,S,n,DL], document_lens [b,S,n])       -> pooled documents [b,S,n,D]
         session_tail(pooled_q [b,S,D], docs [b,S,N,D], labels [b,S,N], labels_all [B,S,N]) -> click probabilities [b,S,N]
       -> [B,S,N] on every rank."""
    pq = encode_q(plan.own(ex["source_words"]), plan.own(ex["source_lens"]))
    d, l = plan.doc_shard(ex["document_words"], ex["document_lens"])
    docs = plan.assemble((plan.exchange_via_gather if via_gather else plan.exchange)(encode_docs(d, l), group))
    labels = ex["document_labels"]
    probs = session_tail(pq, docs, plan.own(labels), labels)
    return plan.gather(probs, group)


class SessionShardPipeline(object):
    """ONE collective per step for the session-sharded CARS step, software-pipelined over the steps of a lane (HIP stream).

    The two exchanges of SessionShardPlan -- pooled candidate slices out (all-to-all), click probabilities back (all-gather) -- ride in
    the SAME all_to_all_single: chunk r of the send buffer = [ my candidate slice of rank r's sessions, step k | my click probabilities,
    step k-1 ] (the probabilities are KBs and simply replicated into every chunk), so after the exchange a rank holds the N pooled vectors
    of its sessions for step k AND everybody's probabilities of step k-1.  Per step and lane the host then issues one compute segment
    ( tail of step k-1 ; encode of step k  -- one hipGraph replay on a GPU ) and one collective, instead of two graphs and two collectives;
    collectives of all lanes run on the process group's communication stream in issue order, identical on every rank.
    Results lag one step: `flush()` after the last step delivers its probabilities."""

    def __init__(self, plan, D, device, dtype=torch.float32):
        if plan.aligned:
            raise ValueError("pair-axis plans have no exchange to pipeline: encode -> tail -> all-gather of the probabilities per step")
        self.plan, self.D = plan, int(D)
        G, bper, S, per, N = plan.world, plan.bper, plan.S, plan.per, plan.N
        self.n_pool, self.n_prob = bper * S * per * self.D, bper * S * N
        self.send = torch.zeros(G, self.n_pool + self.n_prob, device=device, dtype=dtype)
        self.recv = torch.zeros_like(self.send)

    # -- compute-side views (written / read inside the captured segment) ----------------------------------------------------------------
    def put_pooled(self, pooled_shard):
        """pooled_shard [G*bper,S,per,D] (chunk r = the sessions of rank r) -> the pooled section of the send buffer."""
        self.send[:, :self.n_pool].copy_(pooled_shard.reshape(self.plan.world, self.n_pool))

    def put_probs(self, probs_own):
        """probs_own [bper,S,N] -> replicated into the probability section of every chunk."""
        self.send[:, self.n_pool:].copy_(probs_own.reshape(1, self.n_prob).expand(self.plan.world, self.n_prob))

    def got_pooled(self):
        """-> [G,bper,S,per,D]: rank r's candidate slice of MY sessions (strided view of the receive buffer)."""
        p = self.plan
        return self.recv[:, :self.n_pool].view(p.world, p.bper, p.S, p.per, self.D)

    def got_probs(self):
        """-> [B,S,N]: the click probabilities of the PREVIOUS step, every session (rank-major = session order)."""
        p = self.plan
        return self.recv[:, self.n_pool:].reshape(p.world * p.bper, p.S, p.N)[:p.B]

    # -- the one collective ------------------------------------------------------------------------------------------------------------
    def exchange(self, group=None):
        if self.plan.world == 1 and not (dist.is_available() and dist.is_initialized()):
            self.recv.copy_(self.send)
        else:
            dist.all_to_all_single(self.recv, self.send, group=group)


def pipelined_session_sharded_probs(plan, encode_q, encode_docs, session_tail, batches, group=None, D=None):
    """Reference driver of SessionShardPipeline over a list of batches (one lane): -> list of [B,S,N] probabilities, one per batch.
    Same callables as session_sharded_click_probs (HIP entry points on a GPU; the CPU oracle in the gloo tests)."""
    pipe, prev, out = None, None, []
    for ex in list(batches) + [None]:
        cur = None
        if ex is not None:
            pq = encode_q(plan.own(ex["source_words"]), plan.own(ex["source_lens"]))
            d, l = plan.doc_shard(ex["document_words"], ex["document_lens"])
            pooled = encode_docs(d, l)
            if pipe is None:
                pipe = SessionShardPipeline(plan, pooled.shape[-1] if D is None else D, pooled.device, pooled.dtype)
            cur = (pq, ex["document_labels"])
        if prev is not None:                                 # tail of the previous step: its pooled documents arrived with the last exchange
            docs = plan.assemble(pipe.got_pooled())
            pipe.put_probs(session_tail(prev[0], docs, plan.own(prev[1]), prev[1]))
        if ex is not None:
            pipe.put_pooled(pooled)
        pipe.exchange(group)
        if prev is not None:
            out.append(pipe.got_probs().clone())
        prev = cur
    return out


# ----------------------------------------------------------------------------------------------------------
# The session STREAM over G ranks (BASELINE.json configs[4]; the reference feeds nn.DataParallel from its length-bucketing sampler,
# neuroir/inputters/multitask/data.py:42-72 + models/multitask.py:402-407).
#
# The unit of the stream is a sampler batch (B sessions of ONE length S, N candidates each).  Two ways to spread it, both without any
# exchange of encoder outputs -- the only collective is the all-gather of the click probabilities, issued asynchronously and consumed one
# round later, so it overlaps the next batch's H2D / widen / encode:
#   "batch"  round j = sampler batches jG .. jG+G-1, rank r scores batch jG+r WHOLE (its own click count m).  Throughput mode: per-rank
#            work is a full batch, a round's batches may have different session lengths (gather blocks are padded to the longest).
#   "pair"   every sampler batch is cut along the flattened (session, candidate) pair axis into G blocks of B/G whole sessions (the
#            aligned split of SessionShardPlan); rank r scores block r of EVERY batch with the click count m of the WHOLE batch
#            (cars.py:285-289 -- computed on the host by the collator, an integer).  Latency mode: a batch finishes in 1/G of the time.
# A stream whose batch count is not a multiple of G ("batch" mode) ends with a short round: the idle ranks re-score the round's first
# batch and their block is dropped.
# ----------------------------------------------------------------------------------------------------------
class StreamShardPlan(object):
    def __init__(self, world, rank, mode="batch", batch_size=None):
        self.world, self.rank, self.mode = int(world), int(rank), mode
        if mode not in ("batch", "pair"):
            raise ValueError("mode must be 'batch' or 'pair'")
        if mode == "pair":
            if batch_size is None or int(batch_size) % self.world:
                raise ValueError("pair-mode stream sharding cuts a batch into whole sessions per rank: batch_size %% world must be 0")
            self.bper = int(batch_size) // self.world
        self.batch_size = None if batch_size is None else int(batch_size)

    def rounds(self, nbatches):
        return nbatches if self.mode == "pair" else (nbatches + self.world - 1) // self.world

    def members(self, j, nbatches):
        """the sampler batches of round j as [(rank, batch_no)] (real ones only)."""
        if self.mode == "pair":
            return [(r, j) for r in range(self.world)]
        return [(r, j * self.world + r) for r in range(self.world) if j * self.world + r < nbatches]

    def mine(self, j, batches, rank=None):
        """-> (batch_no, this rank's session indices, the indices of the whole batch) for round j."""
        r = self.rank if rank is None else rank
        if self.mode == "pair":
            idx = list(batches[j])
            groups = len(idx) // self.batch_size               # macro-batches: `groups` sampler batches back to back
            own = [x for g in range(groups) for x in idx[g * self.batch_size + r * self.bper:g * self.batch_size + (r + 1) * self.bper]]
            return j, own, idx
        k = j * self.world + r
        if k >= len(batches):
            k = j * self.world                                # short last round: filler, dropped after the gather
        return k, list(batches[k]), list(batches[k])

    def block_elems(self, B, S, N):
        """elements of one rank's gather block for a round whose longest session has S queries."""
        return (self.bper * (B // self.batch_size) if self.mode == "pair" else B) * S * N

    def unpack(self, j, recv, batches, lengths_of, N):
        """recv [G, block] (rank-major) -> [(batch_no, idx, probs [B,S,N])] of round j (padding / fillers dropped)."""
        out = []
        if self.mode == "pair":
            idx = list(batches[j])
            S = lengths_of(idx)
            groups = len(idx) // self.batch_size
            n = groups * self.bper * S * N
            blocks = recv[:, :n].reshape(self.world, groups, self.bper, S, N)
            out.append((j, idx, blocks.permute(1, 0, 2, 3, 4).reshape(len(idx), S, N)))
            return out
        for r, k in self.members(j, len(batches)):
            idx = list(batches[k])
            S = lengths_of(idx)
            out.append((k, idx, recv[r, :len(idx) * S * N].reshape(len(idx), S, N)))
        return out


def sharded_stream_probs(plan, score_fn, corpus, batches, group=None, on_result=None):
    """Reference driver of the sharded stream in terms of ONE callable (the graph-replayed HIP predict on a GPU --
    graph_runner.StreamingSessionPredictor does the same with device buffers; the CPU oracle in the gloo tests):
         score_fn(ex: the batch tensors of this rank's sessions, click_max: int or None) -> click probabilities [b,S,N]
    The all-gather of round j is issued asynchronously and consumed after round j+1 has been scored (double-buffered).
    on_result(batch_no, idx, probs [B,S,N]) is called on EVERY rank for every sampler batch, in batch order.  -> number of batches delivered."""
    lengths_of = lambda idx: int(corpus.lengths[idx[0]])     # noqa: E731
    N = corpus.N
    max_S = max(lengths_of(b) for b in batches)
    max_B = max(len(b) for b in batches)
    block = plan.block_elems(max_B, max_S, N)
    recv = [torch.zeros(plan.world, block), torch.zeros(plan.world, block)]
    send = [torch.zeros(block), torch.zeros(block)]
    have_pg = dist.is_available() and dist.is_initialized()
    pending, delivered = None, 0

    def deliver(p):
        j, work = p
        if work is not None:
            work.wait()
        n = 0
        for k, idx, probs in plan.unpack(j, recv[j % 2], batches, lengths_of, N):
            if on_result is not None:
                on_result(k, idx, probs.clone())
            n += 1
        return n

    for j in range(plan.rounds(len(batches))):
        k, own, whole = plan.mine(j, batches)
        ex = corpus.batch_tensors(own)
        probs = score_fn(ex, corpus.click_max(whole, plan.batch_size) if plan.mode == "pair" else None)
        s = send[j % 2]
        s.zero_()
        s[:probs.numel()] = probs.reshape(-1).float().cpu()
        if have_pg:
            work = dist.all_gather_into_tensor(recv[j % 2].view(-1), s, group=group, async_op=True)
        else:
            assert plan.world == 1
            recv[j % 2][0].copy_(s)
            work = None
        if pending is not None:
            delivered += deliver(pending)
        pending = (j, work)
    if pending is not None:
        delivered += deliver(pending)
    return delivered
