"""Ranker -- high-level wrapper with the call shapes of neuroir.models.ranker.Ranker
(/root/reference/neuroir/models/ranker.py:25-346) for the hot-path models: build the network from
`args.model_type`, `.cuda()`, `predict(ex)` = softmax(network(...), -1), `loss(ex)` and save/load of
`{state_dict, args}`; `update(ex)` is the training step (models/ranker.py:192-230) on the HIP autograd operators.

Multi-GPU: instead of the reference's nn.DataParallel batch split (models/ranker.py:341-346) the wrapper can
shard the CANDIDATE axis over the ranks of a torch.distributed group and all-gather the scores (sharding.py).
"""
import torch

from .. import lib, sharding
from ..rankers import DRMM, DUET, ESM, MatchTensor
from .common import WrapperBase

NETWORKS = {"ESM": ESM, "DUET": DUET, "DRMM": DRMM, "MATCH_TENSOR": MatchTensor}
BCE_MODELS = {"DUET", "DRMM", "MATCH_TENSOR"}


class Ranker(WrapperBase):
    def __init__(self, args, src_dict=None, state_dict=None):
        self.args = args
        self.src_dict = src_dict
        if src_dict is not None:
            self.args.src_vocab_size = len(src_dict)
        self.updates, self.use_cuda, self.parallel = 0, False, False
        kind = args.model_type.upper()
        if kind not in NETWORKS:
            raise RuntimeError("Unsupported model: %s (hot-path models: %s)" % (args.model_type, sorted(NETWORKS)))
        self.kind = kind
        self.network = NETWORKS[kind](args)
        if state_dict:
            self.network.load_state_dict(state_dict)
        self.group = None

    # -- device / parallel (cuda()/cpu() in WrapperBase) ------------------------------------------
    def parallelize(self, group=None):
        """Candidate-axis sharding over `group` (default WORLD) -- replaces nn.DataParallel."""
        self.parallel = True
        self.group = group

    # -- prediction --------------------------------------------------------------------------------
    def _inputs(self, ex):
        keys = ("que_rep", "que_len", "doc_rep", "doc_len")
        return [ex[k].cuda(non_blocking=True) if self.use_cuda else ex[k] for k in keys]

    @torch.no_grad()
    def scores(self, ex):
        """raw network scores [B,N] (models/ranker.py:257)."""
        self.network.eval()
        q, ql, d, dl = self._inputs(ex)
        if self.parallel:
            return sharding.sharded_scores(lambda dd, ll: self.network(q, ql, dd, ll), d, dl, group=self.group)
        return self.network(q, ql, d, dl)

    _FIELDS = ("que_rep", "que_len", "doc_rep", "doc_len")

    def _predict_body(self, ex):
        s = self.scores(ex).contiguous()
        out, published = self._softmax_rows(s)
        self._maybe_check_ids(published)
        return out

    @torch.no_grad()
    def predict(self, ex):
        """softmax over the candidates (models/ranker.py:236-260).  From the second call of a batch shape on, the call replays a captured
        hipGraph (WrapperBase._graph_entry); an out-of-vocabulary id raises IndexError from the scores' `.cpu()` or from the next call
        (WrapperBase.id_check)."""
        self._poll_ids()
        if self.parallel and sharding.dist.is_available() and sharding.dist.is_initialized():
            self.network.eval()
            q, ql, d, dl = self._inputs(ex)
            world, rank = sharding.dist.get_world_size(self.group), sharding.dist.get_rank(self.group)
            dd, ll = sharding.shard_candidates(d, dl, world, rank)
            return sharding.gathered_softmax(self.network(q, ql, dd, ll), d.shape[1], self.group)
        # the captured part ends at the raw scores; the softmax (+ the publication of the error word) runs eagerly into a fresh tensor
        out = self._graphed(ex, self._FIELDS, None, lambda e: self.scores(e).contiguous(), self._finish_scores)
        return self._checked(self._predict_body(ex) if out is None else out)

    def _finish_scores(self, s):
        out, published = self._softmax_rows(s)
        self._maybe_check_ids(published)
        return out

    @torch.no_grad()
    def predict_many(self, exs, out=None):
        """softmax scores of several equal-shape batches as ONE macro-batch -> [k,B,N] (every (query, candidate) pair of these rankers is
        independent of the rest of its batch, so this is plain concatenation along the query axis): one launch sequence over k x the pairs
        instead of k -- a C2 batch (320 documents) fills 40 of 256 CUs with recurrence workgroups.  out (optional): [k*B,N] result buffer."""
        self.network.eval()
        cols = [self._inputs(e) for e in exs]
        q, ql, d, dl = (torch.cat([c[i] for c in cols]) if len(cols) > 1 else cols[0][i] for i in range(4))
        s = self.network(q, ql, d, dl).contiguous()
        if out is None:
            out = torch.empty_like(s)
        lib.check(lib.load().nir_softmax_rows(lib.ptr(s), lib.ptr(out), s.shape[0], s.shape[1], lib.stream()), "nir_softmax_rows")
        return out.view(len(exs), -1, s.shape[1])

    @torch.no_grad()
    def loss(self, ex):
        """forward + criterion of the model (BCEWithLogits, models/ranker.py:55-69); ESM has none."""
        if self.kind not in BCE_MODELS:
            raise RuntimeError("%s has no training criterion (main/ranker.py:414)" % self.kind)
        s = self.scores(ex).contiguous()
        y = ex["label"].to(s.device).float().contiguous()
        out = torch.empty(1, device=s.device)
        lib.check(lib.load().nir_rank_loss_bce(lib.ptr(s), lib.ptr(y), s.shape[0], s.shape[1], lib.ptr(out),
                                               lib.stream()), "nir_rank_loss_bce")
        return out[0]

    def update(self, ex):
        """models/ranker.py:192-230: train-mode forward -> criterion -> backward -> clip_grad_norm(grad_clipping) -> step.
        Forward/backward of the network and the criterion run on the HIP operators (autograd.py); the optimiser update is
        torch.optim, as in the reference."""
        if self.optimizer is None:
            raise RuntimeError("No optimizer set.")
        if self.kind not in BCE_MODELS:
            raise RuntimeError("%s has no training criterion (main/ranker.py:414)" % self.kind)
        if not hasattr(self.network, "_forward_train"):
            raise NotImplementedError("%s has no train-mode forward" % self.kind)
        self._poll_ids()
        self.optimizer.zero_grad()
        loss = self._update_body(ex)
        self.updates += 1
        self._maybe_check_ids()
        return loss

    def _update_body(self, ex):
        """forward -> criterion -> backward -> (gradient averaging) -> clipping -> optimizer step; no host synchronisation, so that
        common.GraphedUpdate can capture it into one hipGraph (gradients must be cleared by the caller)."""
        from .. import autograd as A
        self.network.train()
        A.STEP.begin()                         # in-place accumulation of the parameter gradients of A.linear, transposes once per step
        try:
            q, ql, d, dl = self._inputs(ex)
            labels = ex["label"].float()
            labels = labels.cuda(non_blocking=True) if self.use_cuda else labels
            scores = self.network(q, ql, d, dl)
            loss = A.bce_with_logits(scores, labels)
            loss.backward()
        except BaseException:
            A.STEP.abort()
            raise
        A.STEP.end()
        self.sync_gradients()                 # multi-rank: average the gradients of all ranks (WrapperBase.sync_gradients)
        torch.nn.utils.clip_grad_norm_(self.network.parameters(), self.args.grad_clipping)
        self.optimizer.step()
        return loss

    # -- persistence (save / checkpoint in WrapperBase) ---------------------------------------------
    @staticmethod
    def load(filename, new_args=None):
        saved = torch.load(filename, map_location="cpu", weights_only=False)
        args = saved["args"]
        if new_args is not None:
            from ..config import override_model_args
            args = override_model_args(args, new_args)
        return Ranker(args, saved.get("src_dict"), saved["state_dict"])

    @staticmethod
    def load_checkpoint(filename, use_gpu=True):
        """models/ranker.py:315-327 -> (model with its optimizer restored, epoch)."""
        saved = torch.load(filename, map_location="cpu", weights_only=False)
        model = Ranker(saved["args"], saved.get("src_dict"), saved["state_dict"])
        if use_gpu:
            model.cuda()
        model.init_optimizer(saved["optimizer"], use_gpu)
        return model, saved["epoch"]
