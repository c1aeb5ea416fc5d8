"""DRMM -- deep relevance matching model (drop-in for neuroir.rankers.drmm.DRMM,
/root/reference/neuroir/rankers/drmm.py:10-98).

The reference materialises two [B*N,QL,DL,E] tensors, takes their cosine, copies it to the HOST for a
per-row numpy.histogram and copies the counts back.  nir_drmm_score streams each document row once,
keeps the QL normalised query rows in LDS, bins cosines in registers and finishes gate softmax / FFN /
output in the same kernel -- no host round trip.  Bin edges follow numpy.histogram(bins=[-1,-.5,0,.5,1,1]).
Exact token matches: the reference's cosine of a row with itself is <1 / ==1 / >1 by its reduction order and lands
in [.5,1) / {1} / nowhere; that bin is a pure function of the embedding row, computed here once per table version exactly as the
reference's host path computes it (`self_cosine_bins`) and looked up by the kernel for every q_id == d_id hit -- the integer histograms are
then the reference's.  Cosines of DIFFERENT rows within fp32 rounding of a bin edge (O(1e-6) of all pairs) may still land in a neighbouring bin.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import autograd as A
from .. import lib
from ..constants import PAD
from ..modules import Embeddings


def histogram_bin(c):
    """numpy.histogram(x, bins=[-1,-.5,0,.5,1,1]) as a per-element bin index: [-1,-.5) [-.5,0) [0,.5) [.5,1) {1} -> 0..4, anything outside
    [-1,1] (or NaN) -> -1 = dropped.  c: float32 numpy array or torch tensor; returns int8 of the same kind."""
    if torch.is_tensor(c):
        b = (c >= -0.5).to(torch.int8) + (c >= 0).to(torch.int8) + (c >= 0.5).to(torch.int8) + (c >= 1.0).to(torch.int8)
        return torch.where((c >= -1.0) & (c <= 1.0), b, torch.full_like(b, -1))
    c = np.asarray(c, np.float32)
    b = (c >= -0.5).astype(np.int8) + (c >= 0).astype(np.int8) + (c >= 0.5).astype(np.int8) + (c >= 1.0).astype(np.int8)
    return np.where((c >= -1.0) & (c <= 1.0), b, np.int8(-1)).astype(np.int8)


def self_cosine_bins(table, host=True):
    """[V] int8 on table.device: the histogram bin of cosine_similarity(row, row) for every embedding row (drmm.py:66-75 at a q_id == d_id hit).
    host=True: computed by the HOST's ATen kernel on a CPU copy of the table -- the reference's CPU path bit for bit (one D2H copy of the
    table per table version; synchronising, so not inside a graph capture).  host=False: computed on the device (what the reference gives
    with --cuda; no synchronisation -- used while a trainable table changes every step)."""
    t = table.detach().float()
    if host:
        t = t.cpu()
        return torch.from_numpy(histogram_bin(F.cosine_similarity(t, t, 1).numpy())).to(table.device)
    return histogram_bin(F.cosine_similarity(t, t, 1)).contiguous()


class GatingNetwork(nn.Module):
    """Term gating network parameters (drmm.py:87-98): Linear(emsize -> 1), softmax over query slots."""

    def __init__(self, emsize):
        super().__init__()
        self.weight = nn.Linear(emsize, 1)


class DRMM(nn.Module, lib.IdCheck):
    def __init__(self, args):
        super().__init__()
        self.word_embeddings = Embeddings(args.emsize, args.src_vocab_size, PAD)
        self.emb_drop = nn.Dropout(p=args.dropout_emb)
        self.nbins = args.nbins
        if self.nbins != 5:
            raise NotImplementedError("DRMM bins are fixed to [-1,-.5,0,.5,1,1] in the reference (drmm.py:22-23)")
        self.bins = [-1.0, -0.5, 0, 0.5, 1.0, 1.0]
        self.gating_network = GatingNetwork(args.emsize)
        self.ffnn = nn.Sequential(nn.Linear(self.nbins, 1), nn.Linear(1, 1))
        self.output = nn.Linear(1, 1)
        self._pack, self._pack_nobins, self._bins = lib.PackCache(), lib.PackCache(), lib.PackCache(retain=1)
        # Exact token matches (q_id == d_id; the cosine is 1 +- a few ulp):
        #   "reference" (default): the bin the reference's HOST path gives that row (self_cosine_bins; <1 -> [.5,1), ==1 -> {1}, >1 -> dropped) --
        #               the integer histograms equal the reference's;
        #   "numpy":    the kernel's own fp32 cosine binned literally (round 1-4 behaviour: right distribution, different rows);
        #   "snap":     opt-in, intentional deviation: |cos - 1| <= 4 ulp counts as 1, every exact match lands in {1} on every device.
        self.exact_match_policy = getattr(args, "drmm_exact_match_policy", "reference")

    def _self_bins(self):
        table = self.word_embeddings.table
        host = not (self.training and table.requires_grad)      # a table that trains changes every step: device-side rounding, no D2H
        return self._bins.get([table, host], lambda: self_cosine_bins(table, host))

    def _weights(self, bins=True):
        """bins=False: ids are not vocabulary ids (the dropout path addresses a per-batch row table by position) -> no self-bin lookup."""
        if self.exact_match_policy not in ("reference", "numpy", "snap"):
            raise ValueError("exact_match_policy must be 'reference', 'numpy' or 'snap'")
        sb = self._self_bins() if (bins and self.exact_match_policy == "reference") else None

        def build():
            pk = lib.Packed(lib.DrmmWeights, dict(
                gate_w=self.gating_network.weight.weight, gate_b=self.gating_network.weight.bias,
                ffnn0_w=self.ffnn[0].weight, ffnn0_b=self.ffnn[0].bias, ffnn1_w=self.ffnn[1].weight,
                ffnn1_b=self.ffnn[1].bias, out_w=self.output.weight, out_b=self.output.bias), dict(snap_one=int(self.exact_match_policy == "snap")))
            if sb is not None:
                pk.keep["self_bin"] = sb
                pk.struct.self_bin = sb.data_ptr()
            return pk
        params = [p for n, p in self.named_parameters() if not n.startswith("word_embeddings")] + [self.exact_match_policy, sb]
        return (self._pack if bins else self._pack_nobins).get(params, build)

    def _hist(self, q, d, table, bins=True):
        """Matching histograms [B*N, QL, 5] from the scoring kernel (constants w.r.t. the parameters, as in the reference where
        they pass through numpy, drmm.py:70-75)."""
        B, QL = q.shape
        N, DL = d.shape[1], d.shape[2]
        hist = torch.empty(B * N, QL, 5, device=q.device, dtype=torch.float32)
        scratch = torch.empty(B, N, device=q.device, dtype=torch.float32)
        lib.check(lib.load().nir_drmm_score(lib.ptr(q), lib.ptr(d), B, N, QL, DL, lib.ptr(table), table.shape[0], table.shape[1],
                                            self._weights(bins).ref(), lib.ptr(scratch), lib.ptr(hist), lib.stream()), "nir_drmm_score")
        return hist

    def _forward_train(self, q, d):
        """Train-mode forward (drmm.py:29-84) on the autograd operators of autograd.py: the trainable part is the gating network and the
        three tiny Linear layers; the histograms come from the same HIP kernel as in eval.  With embedding dropout the kernel reads
        the dropped rows: they are handed over as a [B*QL + M*DL, E] row table addressed by position."""
        B, QL = q.shape
        N, DL = d.shape[1], d.shape[2]
        M = B * N
        table = self.word_embeddings.table
        p = self.emb_drop.p
        eq = A.dropout(A.embed(q, table), p, True)                                              # [B,QL,E]
        gate = torch.softmax(A.linear(eq, self.gating_network.weight.weight, self.gating_network.weight.bias).squeeze(2), 1)
        if p > 0:
            ed = A.dropout(A.embed(d.reshape(M, DL), table), p, True)
            rows = torch.cat((eq.detach().reshape(B * QL, -1), ed.detach().reshape(M * DL, -1)), 0).contiguous()
            pos = torch.arange(B * QL + M * DL, device=q.device, dtype=torch.int64)
            hist = self._hist(pos[:B * QL].view(B, QL).contiguous(), pos[B * QL:].view(B, N, DL).contiguous(), rows, bins=False)
        else:
            hist = self._hist(q, d, table.detach())
        z = A.linear(A.linear(hist, self.ffnn[0].weight, self.ffnn[0].bias), self.ffnn[1].weight, self.ffnn[1].bias).squeeze(2)
        pooled = (z.view(B, N, QL) * gate.unsqueeze(1)).sum(2, keepdim=True)                    # [B,N,1]
        return A.linear(pooled, self.output.weight, self.output.bias).view(B, N)

    def forward(self, batch_queries, query_len, batch_docs, doc_len, return_hist=False):
        assert batch_queries.shape[0] == batch_docs.shape[0]
        table = self.word_embeddings.table
        lib.require_device(batch_queries, batch_docs, table)
        q, d = self._clean_ids(batch_queries, batch_docs, self.word_embeddings.table.shape[0])
        if self.training and not return_hist:
            # fix_embeddings=False (config.py:94, the reference's default): the table trains through the gating network only -- the
            # histograms are constants of the graph (the reference detaches them through numpy, drmm.py:70-75; _hist reads detached rows)
            return self._forward_train(q, d)
        B, QL = q.shape
        N, DL = d.shape[1], d.shape[2]
        w = self._weights()
        scores = torch.empty(B, N, device=q.device, dtype=torch.float32)
        if B == 0 and not return_hist:
            return scores
        hist = torch.empty(B * N, QL, 5, device=q.device, dtype=torch.float32) if return_hist else None
        lib.check(lib.load().nir_drmm_score(lib.ptr(q), lib.ptr(d), B, N, QL, DL, lib.ptr(table), table.shape[0],
                                            table.shape[1], w.ref(), lib.ptr(scores), lib.ptr(hist), lib.stream()),
                  "nir_drmm_score")
        return (scores, hist) if return_hist else scores
