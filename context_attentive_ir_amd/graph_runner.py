"""hipGraph replay of Ranker.predict / Multitask.predict for fixed batch shapes.

At MSMARCO-shaped batch sizes a forward is a dozen short kernels (~200 us), so host launch overhead dominates an
eager call.  GraphedPredictor captures one predict() into a hipGraph over STATIC input buffers; each call copies the
ids (pinned host or device tensors) into those buffers on the capture stream and replays the graph.  Shapes are
fixed at capture time (one predictor per (B, N, QL, DL) bucket -- the reference's length-bucketing samplers,
neuroir/inputters/ranker/data.py:37-56, already group batches by shape).
"""
import ctypes
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
