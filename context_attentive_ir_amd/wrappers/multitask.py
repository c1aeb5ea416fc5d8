"""Multitask -- wrapper with the call shapes of neuroir.models.multitask.Multitask
(/root/reference/neuroir/models/multitask.py:24-407) for CARS, M_MATCH_TENSOR and MNSRF, ranking side only:
predict(ex) -> {'click_scores': softmax over candidates [B,S,N]} (multitask.py:262-279)."""
import torch

from .. import lib
from ..multitask import CARS, M_MATCH_TENSOR, MNSRF

NETWORKS = {"CARS": CARS, "M_MATCH_TENSOR": M_MATCH_TENSOR, "MNSRF": MNSRF}


class Multitask(object):
    def __init__(self, args, src_dict=None, tgt_dict=None, state_dict=None):
        self.args = args
        self.src_dict, self.tgt_dict = src_dict, tgt_dict
        if src_dict is not None:
            self.args.src_vocab_size = len(src_dict)
        if tgt_dict is not None:
            self.args.tgt_vocab_size = len(tgt_dict)
        self.type = args.model_type.upper()
        if self.type not in NETWORKS:
            raise RuntimeError("Unsupported model: %s (hot-path multitask models: %s)" % (args.model_type, sorted(NETWORKS)))
        self.network = NETWORKS[self.type](args)
        if state_dict:
            self.network.load_state_dict(state_dict)
        self.updates, self.use_cuda, self.parallel = 0, False, False
        self.group = None

    def parallelize(self, group=None):
        """Candidate-axis sharding of the document encoder over `group` (replaces nn.DataParallel,
        models/multitask.py:402-407)."""
        self.parallel = True
        self.group = group

    def cuda(self):
        self.use_cuda = True
        self.network = self.network.cuda()
        return self

    def _dev(self, t):
        return t.cuda(non_blocking=True) if self.use_cuda else t

    @torch.no_grad()
    def scores(self, ex):
        self.network.eval()
        if self.type != "CARS":     # models/multitask.py:271-278: encode -> rank_document(source, memory, session, docs, lens)
            src = self._dev(ex["source_words"])
            memory_bank, session_bank, _ = self.network.encode(src, self._dev(ex["source_lens"]))
            return self.network.rank_document(src, memory_bank, session_bank, self._dev(ex["document_words"]),
                                              self._dev(ex["document_lens"]))
        pooled, _, _ = self.network.encode(self._dev(ex["source_words"]), self._dev(ex["source_lens"]))
        s, _, _ = self.network.rank_document(pooled, self._dev(ex["document_words"]), self._dev(ex["document_lens"]),
                                             self._dev(ex["document_labels"]), group=self.group, shard=self.parallel)
        return s

    @torch.no_grad()
    def predict(self, ex):
        s = self.scores(ex).contiguous()
        out = torch.empty_like(s)
        lib.check(lib.load().nir_softmax_rows(lib.ptr(s), lib.ptr(out), s.shape[0] * s.shape[1], s.shape[2],
                                              lib.stream()), "nir_softmax_rows")
        return {"click_scores": out, "predictions": None}

    def update(self, ex):
        raise NotImplementedError("training step is the next scope row, SURVEY.md section 8f rank 1")

    def save(self, filename):
        state = {k: v.cpu() for k, v in self.network.state_dict().items()}
        torch.save({"state_dict": state, "src_dict": self.src_dict, "tgt_dict": self.tgt_dict, "args": self.args}, filename)

    @staticmethod
    def load(filename):
        saved = torch.load(filename, map_location="cpu", weights_only=False)
        return Multitask(saved["args"], saved.get("src_dict"), saved.get("tgt_dict"), saved["state_dict"])
