"""hipGraph replay of Ranker.predict / Multitask.predict for fixed batch shapes.

At MSMARCO-shaped batch sizes a forward is a dozen short kernels (~200 us), so host launch overhead dominates an
eager call.  GraphedPredictor captures one predict() into a hipGraph over STATIC input buffers; each call copies the
ids (pinned host or device tensors) into those buffers on the capture stream and replays the graph.  Shapes are
fixed at capture time (one predictor per (B, N, QL, DL) bucket -- the reference's length-bucketing samplers,
neuroir/inputters/ranker/data.py:37-56, already group batches by shape).
"""
import collections
import ctypes
import os
import time

import torch

from . import lib


class GraphedPredictor(object):
    def __init__(self, wrapper, example, warmup=3, queue_ahead=True):
        """wrapper: wrappers.Ranker or wrappers.Multitask (already .cuda()); example: a batch dict of that shape.
        queue_ahead: the host may enqueue the next batch while the previous one is still executing (it then only
        waits for the previous H2D copy to have left the staging buffer).  Best for a single predictor (1.44 M vs 1.31 M
        pairs/s at C2); with several predictors round-robin on their own streams pass False -- each then waits for its
        own previous replay, which keeps the hardware queues shallow (2.18 M vs 1.80 M at 4 predictors)."""
        self.queue_ahead = queue_ahead
        if not wrapper.use_cuda:
            raise RuntimeError("GraphedPredictor needs a wrapper on a ROCm device (call .cuda() first)")
        self.wrapper = wrapper
        self.replay_done = None           # event: the previous replay has finished
        self.copy_done = None             # event: the H2D copy out of the staging buffer has been executed
        self.clone_done = None            # event on the caller's stream: its copy of the previous output has been made
        self.fast_path_calls = 0          # batches that arrived packed (inputters.*_batchify / pack()) and took one memmove
        dev = next(wrapper.network.parameters()).device
        self.stream = torch.cuda.Stream(device=dev)
        # one device byte buffer + one pinned staging buffer hold every input tensor (16-byte aligned slots):
        # a step costs ONE H2D copy; the captured graph reads typed views of the device buffer
        self.slots, off = {}, 0
        for k, v in example.items():
            if torch.is_tensor(v) and not k.startswith("_"):
                nbytes = v.numel() * v.element_size()
                self.slots[k] = (off, nbytes, v.dtype, tuple(v.shape))
                off = (off + nbytes + 15) // 16 * 16
        self.dev_buf = torch.empty(max(off, 16), dtype=torch.uint8, device=dev)
        self.host_buf = torch.empty(max(off, 16), dtype=torch.uint8).pin_memory()
        self.static = {k: self.dev_buf[o:o + n].view(dt).view(shape) for k, (o, n, dt, shape) in self.slots.items()}
        self._host_views = {k: self.host_buf[o:o + n].view(dt).view(shape) for k, (o, n, dt, shape) in self.slots.items()}
        for k, v in example.items():
            if k in self.static:
                self.static[k].copy_(v)
        self.stream.wait_stream(torch.cuda.current_stream(dev))    # the initial copies above ran on the caller's stream
        # scratch requested during warm-up / capture belongs to this predictor (lib.workspace_owner): the addresses baked
        # into the graph cannot be superseded by another predictor or an eager caller that shares the stream handle
        with torch.cuda.stream(self.stream), lib.workspace_owner(self):
            for _ in range(warmup):
                self._call()
            self.stream.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) may query events
                self.out = self._call()

    def pack(self, ex, pin=True):
        """Copy a batch into ONE (pinned) host buffer laid out like this predictor's static inputs; the returned dict
        (typed views + '_buffer') goes through predict()'s packed path: one memmove + one H2D copy."""
        buf = torch.empty(self.host_buf.numel(), dtype=torch.uint8)
        if pin:
            buf = buf.pin_memory()
        out = {"_buffer": buf}
        for k, (o, n, dt, shape) in self.slots.items():
            out[k] = buf[o:o + n].view(dt).view(shape)
            out[k].copy_(ex[k])
        return out

    def _staging_free(self):
        """Host-side wait until the previous H2D out of the staging buffer has run.  Waiting for that copy -- not for the
        whole previous replay -- lets the host queue the next batch behind the one executing (bounded: one ahead)."""
        ev = self.copy_done if self.queue_ahead else self.replay_done
        if ev is not None:
            while not ev.query():        # spin: a blocking synchronize can put the host thread to sleep, and the wake-up
                time.sleep(0)            # (tens of us) is of the order of a whole C2 batch; sleep(0) only yields the GIL
                                         # (e.g. to inputters.PrefetchingBatchStream's collating thread)

    def _h2d(self):
        self.dev_buf.copy_(self.host_buf, non_blocking=True)     # stream order keeps it behind the previous replay
        if self.copy_done is None:
            self.copy_done = torch.cuda.Event()
        self.copy_done.record(self.stream)

    def _call(self):
        if hasattr(self.wrapper, "tgt_dict"):      # Multitask: the ranking path (the greedy decoder is not part of the captured step)
            return self.wrapper.predict(self.static, suggest=False)["click_scores"]
        return self.wrapper.predict(self.static)

    def predict(self, ex, clone=True):
        """ex must have the captured shapes; returns the softmax scores (a fresh tensor unless clone=False)."""
        packed = ex.get("_buffer")
        if packed is not None:   # a batch from inputters.*_batchify / pack() in this predictor's field order: already one buffer
            base = packed.data_ptr()
            if ex.get("_layout_ok") is not self:       # validated once per batch object, not once per call
                if packed.is_cuda or packed.numel() != self.host_buf.numel() or any(
                        ex[k].data_ptr() - base != o or tuple(ex[k].shape) != shape or ex[k].dtype != dt
                        for k, (o, n, dt, shape) in self.slots.items()):
                    packed = None
                else:
                    ex["_layout_ok"] = self
        on_host = packed is not None or all(not ex[k].is_cuda for k in self.slots)
        caller = torch.cuda.current_stream()
        # cross-stream ordering (only where there is a hazard, so that several predictors fed from one host thread still
        # overlap): device inputs were produced on the caller's stream; and the clone of the PREVIOUS call's output, made on
        # the caller's stream below, must have read self.out before this replay overwrites it
        if not on_host:
            self.stream.wait_stream(caller)
        if self.clone_done is not None:
            self.stream.wait_event(self.clone_done)
            self.clone_done = None
        with torch.cuda.stream(self.stream):
            if packed is not None:
                # one memmove into this predictor's own pinned staging buffer, one H2D.  (Copying straight out of the
                # caller's buffers measured slower: H2D from ever-changing pinned regions 2.10 M pairs/s vs 2.35 M from
                # the fixed staging buffer, and letting the host run ahead without the sync 1.08 M.)
                self._staging_free()
                ctypes.memmove(self.host_buf.data_ptr(), base, self.host_buf.numel())
                self._h2d()
                self.fast_path_calls += 1
            elif on_host:
                self._staging_free()
                for k, hv in self._host_views.items():
                    src = ex[k]
                    if src.shape != hv.shape:
                        raise RuntimeError("GraphedPredictor captured %s with shape %s, got %s" % (k, tuple(hv.shape), tuple(src.shape)))
                    if src.dtype != hv.dtype:
                        raise RuntimeError("GraphedPredictor captured %s as %s, got %s" % (k, hv.dtype, src.dtype))
                    # plain memmove: torch's copy_ goes parallel above 32K elements, and waking the OpenMP team
                    # costs 10-20 ms per call on a many-core host (measured, tools/graph_probe.py)
                    src = src.contiguous()
                    ctypes.memmove(hv.data_ptr(), src.data_ptr(), hv.numel() * hv.element_size())
                self._h2d()
            else:
                for k, buf in self.static.items():
                    src = ex[k]
                    if src.shape != buf.shape:
                        raise RuntimeError("GraphedPredictor captured %s with shape %s, got %s" % (k, tuple(buf.shape), tuple(src.shape)))
                    buf.copy_(src, non_blocking=True)
            self.graph.replay()
            if not self.queue_ahead:
                if self.replay_done is None:
                    self.replay_done = torch.cuda.Event()
                self.replay_done.record(self.stream)
        caller.wait_stream(self.stream)
        # the copy is allocated AND written on the caller's stream (after the wait): the caching allocator then tracks it
        # on the stream that uses it, and no block is handed back while another stream still reads it
        if not clone:
            return self.out
        out = self.out.clone()
        self.clone_done = torch.cuda.Event()
        self.clone_done.record(caller)
        return out


class CheckedTensor(torch.Tensor):
    """The scores a wrapper's predict() returns: a plain tensor whose `.cpu()` / `.tolist()` -- the caller's own synchronisation in the
    reference's drivers (`scores.cpu().numpy()`, main/ranker.py:255) -- also reads the pinned error word (lib.Flags.poll: no device round
    trip), so an out-of-vocabulary id raises IndexError in the same loop iteration as the reference's nn.Embedding.  Operators see a plain
    tensor (`__torch_function__` disabled: no dispatch overhead, results are plain tensors)."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    def cpu(self, *args, **kwargs):
        out = torch.Tensor.cpu(self, *args, **kwargs)
        f = getattr(self, "_nir_flags", None)
        if f is not None:
            f.poll()
        return out

    def tolist(self):
        out = torch.Tensor.tolist(self)
        f = getattr(self, "_nir_flags", None)
        if f is not None:
            f.poll()
        return out


def checked(t, flags):
    """t as a CheckedTensor bound to `flags` (lib.Flags); None / non-tensors pass through"""
    if not torch.is_tensor(t) or flags is None or not t.is_cuda:
        return t
    c = t.as_subclass(CheckedTensor)
    c._nir_flags = flags
    return c


class PredictGraphCache(object):
    """Shape-keyed hipGraph cache INSIDE Ranker.predict / Multitask.predict (round 6): the reference's drivers call `model.predict(ex)` once
    per batch and synchronise on the scores (main/ranker.py:254-257, main/multitask.py:280-287); an eager call is host-bound (a CARS batch
    is ~45 launches for 0.3 ms of kernels).  The first `min_calls - 1` calls of a key run eagerly; the next one captures the eager body
    over static input buffers (scratch owned by the cache: lib.workspace_owner, shared by its graphs) and every later call is
        [small host fields -> one pinned staging block -> ONE H2D | large pinned fields: direct H2D | device fields: D2D] -> replay -> clone
    on the CALLER's current stream, OPTIMISTICALLY (see call()) (round 6b: the bracketed step is the graph's FIRST KERNEL -- nir_gather_fields reads a per-call table of
    source addresses from pinned memory and copies pinned host tensors over PCIe itself, staged fields and device tensors alike: a replay
    issues no hipMemcpyAsync at all).  Key = field shapes / dtypes + the call's flavour + a weights token (sum of parameter versions, first
    data pointer, the network's path switches): training, load_state_dict or a switch change re-captures; `clear()` on .cuda() / .cpu().
    At most `max_entries` graphs (LRU)."""
    BIG = 64 << 10          # host fields of at least this many bytes that are already pinned skip the staging copy
    # pinned host fields larger than this go through hipMemcpyAsync (SDMA) in front of the replay instead of the gather kernel's own PCIe reads
    # (None: never) -- A/B switch for tools/dropin_loop.py
    SDMA_FROM = int(os.environ["NIR_GATHER_SDMA_FROM"]) if os.environ.get("NIR_GATHER_SDMA_FROM") else None

    def __init__(self, wrapper, max_entries=32, min_calls=2):
        self.w = wrapper
        self.max_entries, self.min_calls = int(max_entries), int(min_calls)
        self.entries = collections.OrderedDict()
        self.seen = {}
        self.params = None
        self.captures = self.replays = 0
        # ONE capture stream and ONE scratch owner for all entries: the library's workspaces are keyed by (owner, device, stream), so every graph of
        # this cache uses the same grow-only scratch (a C3 call at --test_batch_size 128 needs ~1 GB: 32 shapes must not mean 32 GB).  Safe because
        # the graphs of one wrapper are replayed one after the other on the caller's stream; a wrapper shared by threads that call predict() on
        # different streams at the same time is not supported (the reference's predict is not re-entrant either).
        self.side = None
        self.owner = type("PredictScratch", (), {})()

    def clear(self):
        self.entries.clear()
        self.seen.clear()
        self.params = None
        self.owner = type("PredictScratch", (), {})()

    def token(self):
        if self.params is None:
            self.params = list(self.w.network.parameters())
        ps = self.params
        v = 0
        for p in ps:
            v += p._version
        # every plain-valued attribute of the network is a potential path switch (fold_embeddings, compute_dtype, fuse_* ...): part of the token
        sw = tuple(x for x in self.w.network.__dict__.values() if x is None or isinstance(x, (bool, int, float, str)))
        return (v, ps[0].data_ptr() if ps else 0, lib.GRAPH_EPOCH[0], sw)

    EAGER = object()          # call(): "run this call eagerly"

    def call(self, ex, fields, flavour, body, finish=None):
        """One predict() through the cache -> its outputs, or PredictGraphCache.EAGER when the caller must run the eager body itself (first
        sightings of a shape, an un-capturable call).  body(static_ex) = the eager predict over a dict of device tensors; finish(static_out)
        (optional) turns the graph's static outputs into the fresh tensors handed to the caller.

        OPTIMISTIC replay: an entry is found by the SHAPES of the batch alone and replayed at once; the weights token (a walk over the
        parameters' versions, ~15 us of host time in front of the device's chain) is computed while the device already runs the graph.  A
        mismatch -- the weights changed since the capture: rare, once per training epoch -- discards that replay's outputs (the stale graph read
        retired but still allocated packs) and re-captures over the current weights before anything is returned."""
        skey = (tuple((tuple(ex[k].shape), ex[k].dtype) for k in fields), flavour)
        ent = self.entries.get(skey)
        if ent is False:
            return self.EAGER
        if ent is not None:
            # memory safety of the optimistic order: a superseded pack is freed only when some PackCache rebuilds (lib.PACK_EPOCH); while the
            # epoch is the capture's, everything the graph references is alive and the replay may go first.  Otherwise: verify, then replay.
            fresh = ent.epoch == lib.PACK_EPOCH[0]
            if fresh:
                self._replay(ent, ex)
            if self.token() == ent.token:
                if not fresh:
                    ent.epoch = lib.PACK_EPOCH[0]                     # (another model repacked; this graph's packs are the current ones)
                    self._replay(ent, ex)
                self.entries.move_to_end(skey)
                return self._outputs(ent, finish)
            del self.entries[skey]                                    # stale: fall through to a fresh capture of this (recurring) shape
            self.seen[skey] = self.min_calls
        n = self.seen.get(skey, 0) + 1
        if n < self.min_calls:
            if len(self.seen) > 4096:
                self.seen.clear()
            self.seen[skey] = n
            return self.EAGER
        self.seen.pop(skey, None)
        while len(self.entries) >= self.max_entries:
            self.entries.popitem(last=False)
        try:
            ent = self._capture(ex, fields, body)
            ent.token, ent.epoch = self.token(), lib.PACK_EPOCH[0]
        except RuntimeError as e:                                     # an un-capturable call stays eager (recorded, not retried)
            import logging
            logging.getLogger(__name__).warning("predict graph capture failed, staying eager for this shape: %s", e)
            self.entries[skey] = False
            return self.EAGER
        self.entries[skey] = ent
        self._replay(ent, ex)
        return self._outputs(ent, finish)

    def _capture(self, ex, fields, body):
        dev = next(self.w.network.parameters()).device
        ent = type("PredictGraph", (), {})()
        order = sorted(fields, key=lambda k: ex[k].numel() * ex[k].element_size())
        slots, off = [], 0
        for k in order:
            v = ex[k]
            nbytes = v.numel() * v.element_size()
            slots.append((k, off, nbytes, v.dtype, tuple(v.shape)))
            off = (off + nbytes + 15) // 16 * 16
        ent.slots = slots
        ent.dev_buf = torch.empty(max(off, 16), dtype=torch.uint8, device=dev)
        ent.host_buf = torch.empty(max(off, 16), dtype=torch.uint8).pin_memory()
        ent.static = {k: ent.dev_buf[o:o + n].view(dt).view(shape) for k, o, n, dt, shape in slots}
        ent.replayed, ent.hold = None, None
        # the per-call source table {address, destination offset, bytes} x fields, in pinned memory the gather kernel reads
        ent.table = torch.zeros(3 * len(slots), dtype=torch.int64).pin_memory()
        ent.table_np = ent.table.numpy()
        L = lib.load()
        dp = ctypes.c_void_p()
        lib.check(L.nir_host_device_pointer(ctypes.c_void_p(ent.host_buf.data_ptr()), ctypes.byref(dp)), "nir_host_device_pointer")
        ent.stage_dev = int(dp.value)                                 # device-visible address of the staging block
        ent.total = sum((n + 15) // 16 * 16 for _, _, n, _, _ in slots)
        self._fill(ent, ex)
        cur = torch.cuda.current_stream(dev)

        def step():
            lib.check(L.nir_gather_fields(ctypes.c_void_p(ent.table.data_ptr()), len(slots), lib.ptr(ent.dev_buf), ent.total, lib.stream()),
                      "nir_gather_fields")
            return body(ent.static)
        if self.side is None or self.side.device != dev:
            self.side = torch.cuda.Stream(device=dev)
        side = self.side
        side.wait_stream(cur)
        with torch.cuda.stream(side), lib.workspace_owner(self.owner):
            step()                                                    # warm-up under the entry's own scratch (sizes it, builds packs)
            side.synchronize()
            ent.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ent.graph, stream=side, capture_error_mode="thread_local"):
                ent.out = step()
        cur.wait_stream(side)
        self.captures += 1
        return ent

    def _fill(self, ent, ex):
        """the source table of this call: a device tensor or a large pinned host tensor is read where it lies, everything else is memmoved
        into the entry's pinned staging block first.  The sources are kept referenced until the replay that reads them has finished."""
        if ent.replayed is not None:
            ent.replayed.synchronize()                                # the previous replay has read its table / staging block / sources
        L = lib.load()
        base, tab, hold = ent.host_buf.data_ptr(), ent.table_np, []
        dp = ctypes.c_void_p()
        for i, (k, o, n, dt, shape) in enumerate(ent.slots):
            src = ex[k]
            if not src.is_contiguous():
                src = src.contiguous()
            addr = 0
            if self.SDMA_FROM is not None and n >= self.SDMA_FROM and not src.is_cuda and src.is_pinned():
                ent.static[k].copy_(src, non_blocking=True)
                hold.append(src)
                tab[3 * i], tab[3 * i + 1], tab[3 * i + 2] = ent.stage_dev + o, o, 0
                continue
            if src.is_cuda:
                addr = src.data_ptr()
            elif n >= self.BIG and src.is_pinned() and L.nir_host_device_pointer(ctypes.c_void_p(src.data_ptr()), ctypes.byref(dp)) == 0:
                addr = int(dp.value)
            if addr and addr % 16 == 0:
                hold.append(src)
            else:
                ctypes.memmove(base + o, src.data_ptr(), n)
                addr = ent.stage_dev + o
            tab[3 * i], tab[3 * i + 1], tab[3 * i + 2] = addr, o, n
        ent.hold = hold

    def _replay(self, ent, ex):
        """source table of this call -> replay (gather kernel + the captured predict) on the caller's current stream"""
        self._fill(ent, ex)
        ent.graph.replay()
        if ent.replayed is None:
            ent.replayed = torch.cuda.Event()
        ent.replayed.record()
        self.replays += 1

    def _outputs(self, ent, finish):
        """fresh copies of the graph's static outputs (same structure as the eager body's); finish(static_out) (optional) makes them itself --
        e.g. the final softmax run eagerly from the static raw scores into a new tensor: one launch instead of an in-graph softmax plus a copy"""
        out = ent.out
        if finish is not None:
            return finish(out)
        if isinstance(out, dict):
            return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()}
        return out.clone()


class StreamingSessionPredictor(object):
    """Sustained, H2D-INCLUSIVE CARS ranking over a session stream (BASELINE.json configs[4]).

    Per lane (HIP stream): `slots` pinned host staging buffers, one device wire buffer, one device int64 buffer, and one captured
    hipGraph per session length S seen so far:  [ nir_widen_ids_i32 (int32 wire block -> int64 ids/lengths) | Multitask.predict
    ranking path ]  reading fixed device addresses.  A batch costs the host one in-place collate (producer thread, numpy `take` into
    the pinned slot), one H2D of the int32 block (half the bytes of the reference's LongTensor batch), one graph replay and one small
    D2H of the click probabilities.  Batches go round-robin over the lanes; a host slot is recycled as soon as ITS H2D has executed
    (not the whole replay), so with >= 2 slots per lane the producer fills slot k+1 while slot k is in flight.
    Shapes: batches of exactly `batch_size` sessions of one length (the reference sampler's composition,
    neuroir/inputters/multitask/data.py:42-72); a new length is captured on first use (outside any timed region: `prepare`)."""

    def __init__(self, wrapper, n_cands, qlen, dlen, batch_size, max_session_len=16, lanes=2, slots=2, macro=1, plan=None, group=None,
                 gather="auto"):
        """macro > 1: a wire block / graph replay carries `macro` batches of one session length as ONE macro-batch (Multitask.predict_groups:
        one pass over the session weights for all of them, every batch keeps its own click count); `batch_size` is then macro x the
        sampler's batch size and run() expects index lists of that length (see merge_batches).

        plan (sharding.StreamShardPlan) + group: the stream over the ranks of a process group (BASELINE.json configs[4]).  Every rank builds
        the same batch list; per round this rank collates / scores its share (mode "batch": one whole batch of the round; mode "pair":
        batch_size / world whole sessions of every batch, the batch's click count riding in the wire block) and the click probabilities of
        all ranks are all-gathered: with the RCCL backend on the lane's own COMMUNICATION stream -- [ wait for the replay | all-gather | D2H of
        the gathered block ] -- while the lane's compute stream already runs the next batch's H2D / widen / encode; with gloo (flow tests)
        through the host at hand-over time.  Every rank delivers every batch's probabilities (on_result).
        gather: "device" (RCCL) / "host" (gloo) / "none" (each rank keeps its own share: weak form) / "auto" (by backend)."""
        from .inputters.session_stream import WireLayout
        if not wrapper.use_cuda:
            raise RuntimeError("StreamingSessionPredictor needs a wrapper on a ROCm device (call .cuda() first)")
        self.macro = max(1, int(macro))
        if int(batch_size) % self.macro:
            raise RuntimeError("batch_size must be macro x the sampler's batch size")
        self.wrapper, self.N, self.QL, self.DL = wrapper, int(n_cands), int(qlen), int(dlen)
        self.plan, self.group = plan, group
        self.B_all = int(batch_size)                       # sessions of a (macro-)batch of the stream
        pair = plan is not None and plan.mode == "pair"
        if pair and plan.batch_size * self.macro != self.B_all:
            raise RuntimeError("pair-mode plan was built for batches of %d sessions, the stream has %d x %d" % (plan.batch_size, self.macro, self.B_all // self.macro))
        self.B = self.B_all // plan.world if pair else self.B_all      # sessions THIS rank scores per replay
        self.groups = self.macro if pair else 0            # click counts shipped in the wire block
        self.gather = gather
        if plan is None or plan.world == 1 and group is None:
            self.gather = "none" if gather == "auto" else gather
        elif gather == "auto":
            import torch.distributed as dist
            self.gather = "device" if dist.get_backend(group) == "nccl" else "host"
        self.emulated = False                              # plan.world > ranks of the group: this process times ONE rank's share (tuning aid)
        if self.gather != "none":
            import torch.distributed as dist
            self.emulated = dist.get_world_size(group) != plan.world
        self.WireLayout = WireLayout
        self.dev = next(wrapper.network.parameters()).device
        big = WireLayout(self.B, max_session_len, self.N, self.QL, self.DL, self.groups)
        self.max_S = int(max_session_len)
        self.lanes = []
        G = plan.world if plan is not None else 1
        blk = self.B * max_session_len * self.N             # one rank's (padded) probability block
        self.block = blk
        for _ in range(max(1, lanes)):
            ln = {"stream": torch.cuda.Stream(device=self.dev),
                  "wire": torch.empty(big.nbytes, dtype=torch.uint8, device=self.dev),
                  "wide": torch.empty(big.n_int, dtype=torch.int64, device=self.dev),
                  "host": [torch.empty(big.nbytes, dtype=torch.uint8).pin_memory() for _ in range(max(1, slots))],
                  "res": [torch.empty((G if self.gather != "none" else 1) * blk, dtype=torch.float32).pin_memory() for _ in range(max(1, slots))],
                  "copied": [None] * max(1, slots), "done": [None] * max(1, slots), "graphs": {}, "owners": {}}
            if self.gather == "device":
                # per slot: the rank's block as the replay left it + everybody's blocks; the communication stream owns them until `done`
                ln["comm"] = torch.cuda.Stream(device=self.dev)
                ln["gsend"] = [torch.zeros(blk, device=self.dev) for _ in range(max(1, slots))]
                ln["grecv"] = [torch.zeros(G * blk, device=self.dev) for _ in range(max(1, slots))]
                ln["replayed"] = [None] * max(1, slots)
            ln["host_np"] = [h.numpy() for h in ln["host"]]
            self.lanes.append(ln)
        lib.set_batches_in_flight(len(self.lanes), [ln["stream"] for ln in self.lanes])

    # ---- capture -----------------------------------------------------------------------------------------------------
    def _step(self, ln, lay):
        wide = lay.wide_views(ln["wide"])
        lib.check(lib.load().nir_widen_ids_i32(lib.ptr(ln["wire"]), lib.ptr(ln["wide"]), lay.n_int, lib.stream()), "nir_widen_ids_i32")
        wv = lay.views(ln["wire"])
        ex = dict(wide, document_labels=wv["document_labels"])
        if lay.groups:                                     # a slice of `groups` batches: their click counts came with the block (int32, as shipped)
            return self.wrapper.predict_groups(ex, lay.groups, click_max=wv["click_max"])
        if self.macro > 1:
            return self.wrapper.predict_groups(ex, self.macro)
        return self.wrapper.predict(ex, suggest=False)["click_scores"]

    def prepare(self, lengths, example=None):
        """capture the graphs of these session lengths on every lane (warm-up included).  `example` = (corpus, idx): a real batch used as
        the capture input (else zeros: PAD ids, length-0 sequences -- valid inputs)."""
        for S in sorted({int(s) for s in lengths}):
            if S > self.max_S:
                raise RuntimeError("session length %d > max_session_len %d" % (S, self.max_S))
            lay = self.WireLayout(self.B, S, self.N, self.QL, self.DL, self.groups)
            for ln in self.lanes:
                if S in ln["graphs"]:
                    continue
                owner = ln["owners"].setdefault(S, type("ws", (), {})())
                with torch.cuda.stream(ln["stream"]), lib.workspace_owner(owner):
                    ln["wire"].zero_()
                    if example is not None and int(example[0].lengths[example[1][0]]) == S:
                        self._collate(example[0], example[1], ln["host_np"][0])
                        ln["wire"][:lay.nbytes].copy_(ln["host"][0][:lay.nbytes])
                    for _ in range(2):
                        self._step(ln, lay)
                    ln["stream"].synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=ln["stream"], capture_error_mode="thread_local"):
                        out = self._step(ln, lay)
                ln["graphs"][S] = (g, out, lay)

    # ---- one batch ---------------------------------------------------------------------------------------------------
    def slot_free(self, lane, slot):
        ev = self.lanes[lane]["copied"][slot]
        return ev is None or ev.query()

    def submit(self, lane, slot, S):
        """the wire block of a batch of length-S sessions sits in host slot (lane, slot): H2D, replay, D2H of the probabilities."""
        ln = self.lanes[lane]
        g, out, lay = ln["graphs"][S]
        with torch.cuda.stream(ln["stream"]):
            ln["wire"][:lay.nbytes].copy_(ln["host"][slot][:lay.nbytes], non_blocking=True)
            if ln["copied"][slot] is None:
                ln["copied"][slot], ln["done"][slot] = torch.cuda.Event(), torch.cuda.Event()
            ln["copied"][slot].record(ln["stream"])
            g.replay()
            if self.gather != "device":
                ln["res"][slot][:lay.pairs].copy_(out.reshape(-1), non_blocking=True)
                ln["done"][slot].record(ln["stream"])
            else:
                # the slot's gather buffers are free again once the communication stream has finished the slot's previous round
                # (`done`, recorded there); the compute stream only parks its block and moves on to the next batch
                if ln["replayed"][slot] is None:
                    ln["replayed"][slot] = torch.cuda.Event()
                else:
                    ln["stream"].wait_event(ln["done"][slot])
                ln["gsend"][slot][:lay.pairs].copy_(out.reshape(-1), non_blocking=True)
                ln["replayed"][slot].record(ln["stream"])
        if self.gather == "device":
            import torch.distributed as dist
            G = self.plan.world
            with torch.cuda.stream(ln["comm"]):
                ln["comm"].wait_event(ln["replayed"][slot])
                # issue order of the collectives = submission order of the batches, identical on every rank
                work = dist.all_gather_into_tensor(ln["grecv"][slot][:self.block] if self.emulated else ln["grecv"][slot], ln["gsend"][slot],
                                                   group=self.group, async_op=True)
                work.wait()                                 # (the communication stream waits, not the host)
                ln["res"][slot][:G * self.block].copy_(ln["grecv"][slot], non_blocking=True)
                ln["done"][slot].record(ln["comm"])
        return lay

    def result(self, lane, slot, lay):
        """click probabilities [B,S,N] of the batch submitted from (lane, slot) -- host tensor (valid until the slot is re-submitted).
        (With a gather: this rank's own block; run() hands the gathered rounds over through on_result.)"""
        self.lanes[lane]["done"][slot].synchronize()
        o = self.plan.rank * self.block if self.gather == "device" else 0
        return self.lanes[lane]["res"][slot][o:o + lay.pairs].view(lay.B, lay.S, lay.N)

    def gathered(self, lane, slot):
        """[G, block] view of the host result slot: rank-major probability blocks of the round submitted from (lane, slot).  gloo: the
        all-gather happens HERE, on the host, in hand-over order (identical on every rank)."""
        ln = self.lanes[lane]
        ln["done"][slot].synchronize()
        G = self.plan.world
        if self.gather == "host":
            import torch.distributed as dist
            if "hsend" not in ln:
                ln["hsend"], ln["hrecv"] = torch.zeros(self.block), torch.zeros(G * self.block)
            ln["hsend"].copy_(ln["res"][slot][:self.block])
            dist.all_gather_into_tensor(ln["hrecv"], ln["hsend"], group=self.group)
            return ln["hrecv"].view(G, self.block)
        return ln["res"][slot][:G * self.block].view(G, self.block)

    def _collate(self, corpus, idx, host_np):
        """this rank's share of the (macro-)batch `idx` into a pinned slot."""
        if self.plan is not None and self.plan.mode == "pair":
            r, bs, bper = self.plan.rank, self.plan.batch_size, self.plan.bper
            own = [x for g in range(len(idx) // bs) for x in idx[g * bs + r * bper:g * bs + (r + 1) * bper]]
            return corpus.collate_into(own, host_np, whole=idx, batch_size=bs)
        return corpus.collate_into(idx, host_np)

    @staticmethod
    def merge_batches(corpus, batches, macro):
        """the sampler's batches (equal-length sessions each) -> macro-batches: `macro` batches of ONE session length concatenated (left-over
        batches of a length are dropped from the macro list and returned separately).  Batch composition is untouched; only the order in
        which batches are scored changes."""
        by_len, out, rest = {}, [], []
        for b in batches:
            by_len.setdefault(int(corpus.lengths[b[0]]), []).append(list(b))
        for S_, bl in by_len.items():
            full = len(bl) // macro * macro
            for i in range(0, full, macro):
                out.append([x for b in bl[i:i + macro] for x in b])
            rest.extend(bl[full:])
        return out, rest

    # ---- the whole pipeline ----------------------------------------------------------------------------------------------
    def run(self, corpus, batches, on_result=None, min_seconds=None, max_batches=None, producers=1, cycle=False):
        """Stream `batches` (index lists into `corpus`; cycled when min_seconds -- or max_batches with cycle=True -- asks for more) through the lanes.
        on_result(batch_no, idx, probs[B,S,N] host tensor): called in submission order once a batch's results are on the host.
        Under a process group every rank must run the SAME number of rounds (each round is a collective): pass max_batches (rounds), not
        min_seconds.  -> dict(batches (= rounds submitted by this rank), pairs (scored by this rank), seconds, pairs_per_s, h2d_bytes)."""
        if min_seconds and self.gather != "none":
            raise RuntimeError("a gathered stream runs a fixed number of rounds on every rank: use max_batches (+ cycle=True), not min_seconds")
        import queue
        import threading
        L, R = len(self.lanes), len(self.lanes[0]["host"])
        nslots = L * R
        plan = self.plan
        lengths_of = lambda idx: int(corpus.lengths[idx[0]])     # noqa: E731
        self.prepare({lengths_of(b) for b in batches})
        # the stream in ROUNDS: without a plan round k = batch k; "pair": round k = this rank's sessions of batch k; "batch": round k = batch kG + rank
        nrounds = len(batches) if plan is None else plan.rounds(len(batches))

        def share(k):
            """-> (index list handed to _collate, is it a real batch of this rank?)"""
            j = k % nrounds
            if plan is None or plan.mode == "pair":
                return batches[j], True
            b = j * plan.world + plan.rank
            return (batches[b], True) if b < len(batches) else (batches[j * plan.world], False)

        total = None if min_seconds else (nrounds if max_batches is None else (max_batches if cycle else min(nrounds, max_batches)))
        free_q, ready = [queue.Queue() for _ in range(producers)], queue.Queue()
        stop = threading.Event()
        err = []

        def produce(pi):
            k = pi                                          # producer pi fills rounds pi, pi + P, ... into slots k % nslots
            try:
                while not stop.is_set() and (total is None or k < total):
                    try:
                        free_q[pi].get(timeout=0.05)            # token: slot (k % nslots) may be overwritten
                    except queue.Empty:
                        continue
                    idx, real = share(k)
                    s = k % nslots
                    lay = self._collate(corpus, idx, self.lanes[s % L]["host_np"][s // L])
                    ready.put((k, s, lay.S, real))
                    k += producers
            except BaseException as e:                      # surface collation errors in the consumer
                err.append(e)
                ready.put(None)

        def hand_over(k, ln_i, sl_i, lay):
            """results of round k to the consumer: every batch of the round (gathered) or this rank's own batch"""
            if self.gather == "none":
                idx, real = share(k)
                if real:
                    on_result(k, idx, self.result(ln_i, sl_i, lay))
                return
            for bno, idx, probs in plan.unpack(k % nrounds, self.gathered(ln_i, sl_i), batches, lengths_of, self.N):
                on_result(bno, idx, probs)

        # slot s is owned by producer (batch k % producers) -- with nslots % producers == 0 a slot always belongs to the same producer
        if nslots % producers:
            raise RuntimeError("lanes * slots must be a multiple of the producer count")
        for s in range(nslots):
            free_q[s % producers].put(s)
        threads = [threading.Thread(target=produce, args=(pi,), name="nir-session-collate-%d" % pi, daemon=True) for pi in range(producers)]
        for t in threads:
            t.start()
        pending, inflight, waiting_copy = {}, [], []
        nxt = done = 0
        pairs = nbytes = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            while True:
                # recycle host slots whose H2D has executed (with a result consumer: once the slot's results were handed over, below)
                while on_result is None and waiting_copy and self.slot_free(*waiting_copy[0][:2]):
                    ln_i, sl_i, s = waiting_copy.pop(0)
                    free_q[s % producers].put(s)
                # hand finished results over in submission order
                while inflight and (on_result is None or self.lanes[inflight[0][1]]["done"][inflight[0][2]].query()):
                    k, ln_i, sl_i, lay = inflight.pop(0)
                    if on_result is not None:
                        hand_over(k, ln_i, sl_i, lay)
                        free_q[(sl_i * L + ln_i) % producers].put(sl_i * L + ln_i)
                    done += 1
                if total is not None and nxt >= total:
                    if not inflight:
                        break
                    time.sleep(0)
                    continue
                if min_seconds and time.perf_counter() - t0 >= min_seconds:
                    total = nxt
                    continue
                if nxt not in pending:
                    try:
                        item = ready.get(timeout=0.0005)
                    except queue.Empty:
                        continue
                    if item is None:
                        raise err[0]
                    pending[item[0]] = item
                    if nxt not in pending:
                        continue
                k, s, S, real = pending.pop(nxt)
                ln_i, sl_i = s % L, s // L
                lay = self.submit(ln_i, sl_i, S)
                if on_result is None:
                    waiting_copy.append((ln_i, sl_i, s))
                inflight.append((k, ln_i, sl_i, lay))
                pairs += lay.pairs if real else 0           # (a filler of a short last round is scored and dropped)
                nbytes += lay.nbytes
                nxt += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            stop.set()
            for t in threads:
                t.join(timeout=5.0)
        return {"batches": nxt, "pairs": pairs, "seconds": dt, "pairs_per_s": pairs / dt, "h2d_bytes": nbytes,
                "h2d_GBps": nbytes / dt / 1e9, "lanes": L, "slots_per_lane": R, "producers": producers}
