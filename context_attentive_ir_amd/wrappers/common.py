"""Host-side surface shared by the Ranker / Multitask wrappers: everything main/ranker.py and main/multitask.py call on the
model object besides predict/update (/root/reference/neuroir/models/ranker.py:95-190, 262-346; models/multitask.py:62-160,
319-407).  Pure bookkeeping -- no arithmetic on activations happens here."""
import copy
import logging

import torch
import torch.optim as optim

logger = logging.getLogger(__name__)


class WrapperBase(object):
    optimizer = None
    # Token ids are validated on the device without a host round trip (invalid ids are read as PAD and a bit of the device's error word is
    # set: lib.Flags).  The reference's nn.Embedding raises IndexError at the offending call.  Here, by default (`id_check = "deferred"`),
    # every predict() / update() ends with nir_flag_publish -- a non-zero word is written into pinned host memory by the device -- and the
    # host word is read WITHOUT a device round trip (a) by `.cpu()` / `.tolist()` of the returned scores and (b) at the entry of the next
    # predict() / update(): in the reference's drivers, which synchronise on every batch's scores (main/ranker.py:255, main/multitask.py:284),
    # the IndexError surfaces in the same iteration (Ranker) or at the next call (Multitask, whose driver reshapes the scores first) -- at most
    # one call late, at no cost.  `id_check = "blocking"` reads the device word back after every `id_check_interval`-th call (the error at the
    # offending call, one synchronisation per call); `id_check_interval = 0` switches the per-call work off (the caller calls check_ids(), the
    # blocking read, whenever it wants).  With a deferred check an invalid id has been read as PAD and -- in update() -- the optimizer step of
    # that batch applied before the error surfaces.
    id_check = "deferred"
    id_check_interval = 1
    _id_calls = 0
    # hipGraph replay inside predict(): graph_runner.PredictGraphCache (args.predict_graphs = False switches it off).  A shape is captured at its
    # `predict_graph_min_calls`-th sighting: a capture costs 3-5 ms (tools/_scratch measurement: CARS 3.1-3.3 ms, with decode 4.6-5.6 ms, MatchTensor
    # 1.9-3.2 ms) and a replay saves 0.3-0.45 ms per call against the eager launches, so a shape pays for its graph after ~8-12 calls -- capturing at the
    # 8th sighting is the ski-rental choice (never more than ~2x the cost of the better of "always eager" / "capture at once") for data whose batches
    # are padded to per-batch maxima and repeat a shape only a few times; steady shapes lose seven eager calls once.
    predict_graph_min_calls = 8
    predict_graph_max = 32
    _graphs = None
    _board = None

    def _flags(self):
        if self._board is None:
            from .. import lib
            self._board = lib.flags(next(self.network.parameters()).device)
        return self._board

    def check_ids(self):
        """Synchronising: raise IndexError / RuntimeError if a forward since the last check saw an invalid id, out-of-range weights or a
        recurrence cluster that timed out."""
        from .. import autograd as A
        if self.use_cuda:
            self._flags().check()
        if hasattr(self.network, "check_ids"):
            self.network.check_ids()
        A.check_ids()

    def _poll_ids(self):
        """entry of predict() / update(): the pinned host word of the previous calls (no device round trip)"""
        b = self._board
        if b is None:
            return
        if self.id_check_interval > 0 and self.id_check == "deferred":
            b.poll()
        elif b.host_np[0] & 4:                              # a cluster time-out is never left to the caller (id_check_interval = 0 included)
            b.poll()

    def _softmax_rows(self, s, out=None):
        """softmax over the last axis of the raw scores `s` (contiguous) -> probabilities; with the deferred id check the same launch
        publishes the device's error word (nir_softmax_rows_publish) -> (probs, published)."""
        from .. import lib
        if out is None:
            out = torch.empty_like(s)
        rows, n = s.numel() // s.shape[-1], s.shape[-1]
        L = lib.load()
        if rows > 0 and self.use_cuda and self.id_check == "deferred" and (self.id_check_interval > 0 or getattr(self.network, "uses_cluster", False)):
            f = self._flags()
            if f.mapped:
                rc = L.nir_softmax_rows_publish(lib.ptr(s), lib.ptr(out), rows, n, lib.ptr(f.dev), lib.C.c_void_p(f.host.data_ptr()), lib.stream())
                if rc == 0:
                    return out, True
                f.mapped = False
        lib.check(L.nir_softmax_rows(lib.ptr(s), lib.ptr(out), rows, n, lib.stream()), "nir_softmax_rows")
        return out, False

    def _maybe_check_ids(self, published=False):
        if not self.use_cuda:
            return
        if self.id_check_interval <= 0:
            if getattr(self.network, "uses_cluster", False) and self.id_check == "deferred" and not published:
                self._flags().publish()                     # (the cluster recurrence's time-out bit is published whatever the id check does)
            return
        if self.id_check == "deferred":
            if published or self._flags().publish():        # capturable: part of a captured predict
                return
        if torch.cuda.is_current_stream_capturing():
            return
        self._id_calls += 1
        if self._id_calls >= self.id_check_interval:
            self._id_calls = 0
            self.check_ids()

    def _checked(self, t):
        """the returned scores: `.cpu()` also reads the pinned error word (graph_runner.CheckedTensor)"""
        if self.id_check_interval > 0 and self.id_check == "deferred" and self._board is not None and self._board.mapped:
            from ..graph_runner import checked
            return checked(t, self._board)
        return t

    def _graphed(self, ex, fields, flavour, body, finish=None):
        """this predict() through the hipGraph cache -> its outputs, or None: run the eager body."""
        if not (self.use_cuda and getattr(self.args, "predict_graphs", True)) or getattr(self, "parallel", False):
            return None
        if torch.cuda.is_current_stream_capturing():        # an outer capture (graph_runner.GraphedPredictor, bench.py) records the eager body
            return None
        from .. import lib
        if lib.load().nir_profile_enable(-1):               # per-kernel event timing is on: events cannot be recorded into a capture
            return None
        if self._graphs is None:
            from ..graph_runner import PredictGraphCache
            self._graphs = PredictGraphCache(self, self.predict_graph_max, self.predict_graph_min_calls)
        for k in fields:
            if not torch.is_tensor(ex.get(k)):
                return None
        if self.network.training:
            self.network.eval()                             # (the key holds the network's plain attributes, `training` among them)
        out = self._graphs.call(ex, fields, flavour, body, finish)
        return None if out is self._graphs.EAGER else out

    def clear_predict_graphs(self):
        if self._graphs is not None:
            self._graphs.clear()

    # the module whose `.word_vec_size / .init_word_vectors / .parameters()` the drivers use
    def _word_embeddings(self):
        net = self.network
        return net.word_embeddings if hasattr(net, "word_embeddings") else net.embedder.word_embeddings

    def count_parameters(self):
        return sum(p.numel() for p in self.network.parameters() if p.requires_grad)

    def layer_wise_parameters(self):
        rows = [(n, list(p.shape), p.numel()) for n, p in self.network.named_parameters() if p.requires_grad]
        w = max([len(r[0]) for r in rows] + [10])
        return "\n".join("%-*s %20s %12d" % (w, n, s, c) for n, s, c in rows)

    def load_embeddings(self, words, embedding_file):
        """Pre-trained vectors for the dictionary tokens in `words` (contract of models/ranker.py:105-150): a text file with one token and
        emsize floats per line, optionally led by a "count dim" line; a token that occurs several times after dictionary normalisation gets
        the mean of its vectors; tokens without a vector keep their initialisation."""
        layer = self._word_embeddings()
        width = layer.word_vec_size + 1
        canon = getattr(self.src_dict, "normalize", None) or (lambda tok: tok)
        wanted = {w for w in words if w in self.src_dict}
        logger.info("Loading pre-trained embeddings for %d words from %s" % (len(wanted), embedding_file))
        total, seen = {}, {}
        with open(embedding_file) as fh:
            for lineno, line in enumerate(fh):
                fields = line.rstrip().split(" ")
                if lineno == 0 and len(fields) == 2:       # header of the word2vec text format
                    continue
                if len(fields) != width:
                    raise AssertionError("%s:%d: expected a token and %d values, got %d fields" % (embedding_file, lineno + 1, width - 1, len(fields)))
                tok = canon(fields[0])
                if tok not in wanted:
                    continue
                vec = torch.tensor([float(x) for x in fields[1:]])
                if tok in total:
                    total[tok] += vec
                    seen[tok] += 1
                else:
                    total[tok], seen[tok] = vec, 1
        vectors = {tok: v / seen[tok] for tok, v in total.items()}
        layer.init_word_vectors(self.src_dict, vectors, self.args.fix_embeddings)
        logger.info("Loaded %d embeddings (%.2f%%)" % (len(vectors), 100.0 * len(vectors) / max(len(wanted), 1)))

    def init_optimizer(self, state_dict=None, use_gpu=True):
        """models/ranker.py:152-190: freeze the embedding table when fix_embeddings, build sgd/adam/adamax/adadelta over the
        free parameters, optionally restore its state."""
        if self.args.fix_embeddings:
            for p in self._word_embeddings().parameters():
                p.requires_grad = False
        parameters = [p for p in self.network.parameters() if p.requires_grad]
        a = self.args
        # dropout masks of the HIP train-mode forwards come from a counter-based stream (autograd.DROPOUT): tie it to the run's seed
        # (args.random_seed, main/ranker.py:49; else torch's) and to the rank, so runs / ranks / resumed runs do not replay one sequence
        from .. import autograd as A
        rank = 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank = dist.get_rank()
        except Exception:  # pragma: no cover
            pass
        seed = getattr(a, "random_seed", None)
        A.DROPOUT.manual_seed((int(seed) if seed is not None else int(torch.initial_seed())) * 1000003 + rank)
        # capturable: the step counters live on the device, so a whole update can be captured into a hipGraph (GraphedUpdate); same arithmetic
        cap = dict(capturable=True) if (use_gpu and torch.cuda.is_available()) else {}
        if a.optimizer == "sgd":
            self.optimizer = optim.SGD(parameters, a.learning_rate, momentum=a.momentum, weight_decay=a.weight_decay)
        elif a.optimizer == "adam":
            # fused: the whole Adam update of all parameters is a handful of multi-tensor launches.  The default (foreach) form of a capturable Adam
            # walks its 0-dim device step counters one tensor at a time -- ~7 tiny kernels per parameter (bias corrections), 500 of the 690
            # tensor-glue launches of a CARS step (torch.profiler: 144 aten::div_ + 137 mul + ...); the arithmetic is the same Adam.
            self.optimizer = optim.Adam(parameters, a.learning_rate, weight_decay=a.weight_decay, **cap, **(dict(fused=True) if cap else {}))
        elif a.optimizer == "adamax":
            self.optimizer = optim.Adamax(parameters, a.learning_rate, weight_decay=a.weight_decay, **cap)
        elif a.optimizer == "adadelta":
            self.optimizer = optim.Adadelta(parameters, a.learning_rate, weight_decay=a.weight_decay, **cap)
        else:
            raise RuntimeError("Unsupported optimizer: %s" % a.optimizer)
        if state_dict is not None:
            ds = state_dict.pop("_nir_dropout_state", None) if isinstance(state_dict, dict) else None
            if ds is not None:                      # resume: continue the mask stream where the checkpoint left it
                A.DROPOUT.seed, A.DROPOUT.counter = int(ds[0]), int(ds[1])
            # torch's Optimizer.load_state_dict takes the SAVED groups' hyper-parameters and keeps only `params` of the live ones: a checkpoint
            # (written without the run-time flavour flags, like the reference's, models/ranker.py:283-292) would turn the freshly built capturable /
            # fused optimizer into a plain one -- GraphedUpdate could not capture it and eager steps would fall back to one host sync per
            # parameter.  The run-time flags of the fresh groups are re-applied after the load.
            runtime = [{k: g[k] for k in ("capturable", "fused", "foreach", "differentiable", "maximize") if k in g} for g in self.optimizer.param_groups]
            self.optimizer.load_state_dict(state_dict)
            for g, flags in zip(self.optimizer.param_groups, runtime):
                g.update(flags)
                if torch.is_tensor(g.get("lr")):
                    g["lr"] = float(g["lr"])
            if use_gpu:
                dev = parameters[0].device if parameters else None
                for state in self.optimizer.state.values():
                    for k, v in state.items():
                        if isinstance(v, torch.Tensor):
                            # a capturable optimizer keeps its step counter as a float32 tensor on the parameters' device
                            state[k] = v.to(device=dev, dtype=torch.float32) if (k == "step" and cap) else v.to(dev)
                        elif k == "step" and cap:
                            state[k] = torch.tensor(float(v), dtype=torch.float32, device=dev)

    def sync_gradients(self):
        """Multi-rank training step (replaces what nn.DataParallel's backward did for the reference, models/ranker.py:341-346,
        models/multitask.py:402-407: replicas see different slices of the batch and their gradients are reduced).  One process per GPU:
        after backward, before clip_grad_norm_, the gradients of all ranks of `self.group` are AVERAGED with one flat all-reduce
        (RCCL when the group's backend is 'nccl'), so every rank applies the same update and the replicas -- which must start from the
        same weights -- never diverge; the candidate-sharded predict then gathers scores of ONE model.  No-op unless parallelize() was
        called and a process group with more than one rank is initialised."""
        if not getattr(self, "parallel", False):
            return False
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False
        group = getattr(self, "group", None)
        world = dist.get_world_size(group)
        if world == 1:
            return False
        params = [p for p in self.network.parameters() if p.requires_grad]
        for p in params:                      # a parameter unused on this rank still takes part in the reduction
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        dist.all_reduce(flat, group=group)
        flat.div_(world)
        off = 0
        for p in params:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
        return True

    def _params(self, extra=None):
        state = copy.copy(self.network.state_dict())
        state.pop("fixed_embedding", None)
        p = {"state_dict": {k: v.cpu() for k, v in state.items()}, "src_dict": self.src_dict, "args": self.args}
        if hasattr(self, "tgt_dict"):
            p["tgt_dict"] = self.tgt_dict
        p.update(extra or {})
        return p

    def save(self, filename):
        try:
            torch.save(self._params(), filename)
        except BaseException:
            logger.warning("WARN: Saving failed... continuing anyway.")

    def checkpoint(self, filename, epoch):
        if self.optimizer is None:
            raise RuntimeError("No optimizer set.")
        try:
            from .. import autograd as A
            opt = dict(self.optimizer.state_dict())
            # the reference's checkpoint holds a float learning rate and no fused / capturable flags (models/ranker.py:283-292): GraphedUpdate's
            # device-tensor rate and this build's optimizer flavour are run-time choices, re-made by init_optimizer on the resuming side
            groups = []
            for g in opt.get("param_groups", []):
                g = dict(g)
                if torch.is_tensor(g.get("lr")):
                    g["lr"] = float(g["lr"])
                for k in ("fused", "capturable", "foreach"):
                    if k in g:
                        g[k] = None if k != "capturable" else False
                groups.append(g)
            opt["param_groups"] = groups
            opt["_nir_dropout_state"] = (A.DROPOUT.seed, A.DROPOUT.counter)
            torch.save(self._params({"epoch": epoch, "optimizer": opt}), filename)
        except BaseException:
            logger.warning("WARN: Saving failed... continuing anyway.")

    def cuda(self):
        self.use_cuda = True
        self.network = self.network.cuda()
        self._board = None
        self.clear_predict_graphs()
        return self

    def cpu(self):
        self.use_cuda = False
        self.network = self.network.cpu()
        self._board = None
        self.clear_predict_graphs()
        return self


class GraphedUpdate(object):
    """The whole training step of a wrapper -- train-mode forward, losses, backward, gradient clipping, optimizer step
    (models/ranker.py:192-230, models/multitask.py:161-223) -- as ONE hipGraph per batch shape.

    The eager update() is host-bound: a CARS step enqueues ~900 launches from Python / autograd for 9.6 ms of kernels (27 ms per step).
    Captured, the step costs one graph launch.  What capture needs and gets:
      * static inputs: the batch is copied into per-shape device buffers;
      * fresh dropout masks on every replay: the mask seed lives in device memory (nir_dropout_dev_f32), the graph's first node advances it;
      * a capturable optimizer (init_optimizer builds Adam / Adamax / Adadelta with capturable=True on a GPU);
      * no host synchronisation inside the step: id validation stays deferred (`wrapper.network.check_ids()` whenever the caller wants it).
    The first batch of every shape is an ordinary eager update (it is that step, and it creates the optimizer state); the graph is captured
    right after it and replayed from the second batch of the shape on.  Multi-rank gradient averaging is not captured: with a process group
    the call falls back to update().  Returns the loss dict of device tensors (valid until the next call)."""

    def __init__(self, wrapper):
        self.w = wrapper
        self.graphs = {}
        self.seed = None
        self.stream = None

    def _hyper(self):
        """Host-side hyper-parameters that the capture would freeze: part of the graph key, so a change re-captures instead of being ignored.
        The learning rate of a capturable optimizer is excluded -- it lives in a device tensor (below) that the captured kernels read."""
        a = self.w.args
        groups = tuple((None if torch.is_tensor(g["lr"]) else float(g["lr"]), float(g.get("weight_decay", 0.0)), float(g.get("momentum", 0.0)))
                       for g in self.w.optimizer.param_groups)
        return groups + (float(getattr(a, "grad_clipping", 0.0)),)

    def _device_lr(self, dev):
        """The reference's training loops decay the rate in place every epoch (`optimizer.param_groups[0]['lr'] *= args.lr_decay`,
        main/ranker.py:204, main/multitask.py:228).  A Python float would be baked into the captured kernels' arguments; a capturable
        optimizer (Adam / Adamax / Adadelta here) takes the rate as a device tensor instead, `*=` then updates it in place and every replay
        reads the current value.  (Assigning a new float, or SGD, which has no capturable form: the key changes and the step is re-captured.)"""
        for g in self.w.optimizer.param_groups:
            if g.get("capturable") and not torch.is_tensor(g["lr"]):
                g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=dev)

    def _key(self, ex):
        return tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(ex.items()) if torch.is_tensor(v))

    def __call__(self, ex):
        from .. import autograd as A
        w = self.w
        if w.optimizer is None:
            raise RuntimeError("No optimizer set.")
        if getattr(w, "group", None) is not None or getattr(w, "parallel", False):
            return w.update(ex)
        dev = next(w.network.parameters()).device
        self._device_lr(dev)
        key = (self._key(ex), self._hyper())
        ent = self.graphs.get(key)
        if ent is None:
            # a host-side hyper-parameter changed (SGD's per-epoch `lr *= lr_decay`, a new float rate): the graphs captured under the old
            # values can never be replayed again -- drop them (graph, static inputs, private pool) instead of growing by one set of shapes
            # per epoch
            for k in [k for k in self.graphs if k[1] != key[1]]:
                del self.graphs[k]
        caller = torch.cuda.current_stream(dev)
        if ent is None:
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=dev)
            # The eager first step of a shape runs on the stream the capture will use: autograd's AccumulateGrad nodes remember the stream they
            # were created on, and nodes of an earlier step that are still alive would meet the captured backward on a different stream
            # (torch warns, and would synchronise the two).  Stale graphs of earlier eager steps are collected first.
            import gc
            self.stream.wait_stream(caller)
            with torch.cuda.stream(self.stream):
                out = w.update(ex)                               # this batch's step, eager
                # only detached values survive into the capture: with the eager step's autograd graph still referenced (through its loss), ending
                # the capture crashed inside hipStreamEndCapture (ROCm 7.0 / torch 2.10; reproduced in isolation, fine once the reference is dropped)
                out = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()} if isinstance(out, dict) else out.detach()
                static = {k: (v.to(dev).clone() if torch.is_tensor(v) else v) for k, v in ex.items()}
                if self.seed is None:
                    self.seed = torch.full((1,), (A.DROPOUT.seed * 2654435761 + 12345) & 0x7FFFFFFFFFFF, dtype=torch.int64, device=dev)
            gc.collect()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            w.optimizer.zero_grad(set_to_none=True)
            A.DROPOUT.device_seed, A.DROPOUT.site = self.seed, 0
            try:
                with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                    self.seed.add_(1)
                    loss = w._update_body(static)
                    # keep the VALUES only: a loss that still carried its grad_fn would keep the captured step's autograd graph -- and its
                    # AccumulateGrad nodes, bound to the capture stream -- alive for as long as this object lives, and a later eager update() on
                    # another stream would meet them (stream-mismatch warning + synchronisation)
                    loss = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in loss.items()} if isinstance(loss, dict) else loss.detach()
            finally:
                A.DROPOUT.device_seed = None
            self.graphs[key] = (g, static, loss)
            caller.wait_stream(self.stream)
            return out
        g, static, loss = ent
        for k, v in ex.items():
            if torch.is_tensor(v):
                static[k].copy_(v, non_blocking=True)
        g.replay()
        # a replay changes the parameters WITHOUT bumping their version counters (the in-place optimizer kernels were dispatched once, at capture): every
        # version-keyed cache -- the weight packs / folded tables of the predict path (lib.PackCache), the predict() graph cache -- would keep serving
        # the weights of the capture.  Round 6 (found with tools/_scratch: predict() after six graphed steps differed from a fresh model with the same
        # state dict by 0.58): bump the counters explicitly, no kernel involved.
        torch.autograd.graph.increment_version(self._trained())
        w.updates += 1
        return loss

    def _trained(self):
        ps = getattr(self, "_trained_params", None)
        if ps is None or self._trained_opt is not self.w.optimizer:
            ps = self._trained_params = [p for g in self.w.optimizer.param_groups for p in g["params"]]
            self._trained_opt = self.w.optimizer
        return ps
