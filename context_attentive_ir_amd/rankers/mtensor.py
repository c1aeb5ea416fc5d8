"""MatchTensor (drop-in for neuroir.rankers.mtensor.MatchTensor, /root/reference/neuroir/rankers/mtensor.py:24-158).

Same constructor args, attribute names and state-dict keys; forward enqueues the whole pipeline with one
C-ABI call (nir_matchtensor_score): gather+projection GEMM, gate GEMMs, two BiLSTM recurrences, channel
projections, per-query weight folding and the fused interaction/conv/max-pool head.  The [B*N,51,QL,DL]
match tensor of the reference is never built (csrc/mtensor.hip).
"""
import torch
import torch.nn as nn

from .. import lib
from ..constants import PAD
from ..encoders import RNNEncoder
from ..encoders.rnn_encoder import lstm_cat_weights
from ..modules import Embeddings


class ExactMatchChannel(nn.Module):
    """alpha * [q_id == d_id] channel (mtensor.py:134-158); alpha ~ U(0,1)."""

    def __init__(self):
        super().__init__()
        self.alpha = nn.Parameter(torch.rand(1))


class MatchTensor(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.word_embeddings = Embeddings(args.emsize, args.src_vocab_size, PAD)
        self.emb_drop = nn.Dropout(p=args.dropout_emb)
        self.linear_projection = nn.Linear(args.emsize, args.featsize)
        self.query_encoder = RNNEncoder(args.rnn_type, args.featsize, args.bidirection, args.nlayers,
                                        args.nhid_query, args.dropout_rnn)
        self.document_encoder = RNNEncoder(args.rnn_type, args.featsize, args.bidirection, args.nlayers,
                                           args.nhid_doc, args.dropout_rnn)
        self.query_projection = nn.Linear(args.nhid_query, args.nchannels)
        self.document_projection = nn.Linear(args.nhid_doc, args.nchannels)
        self.exact_match_channel = ExactMatchChannel()
        self.conv1 = nn.Conv2d(args.nchannels + 1, args.nfilters, (3, 3), padding=1)
        self.conv2 = nn.Conv2d(args.nchannels + 1, args.nfilters, (3, 5), padding=(1, 2))
        self.conv3 = nn.Conv2d(args.nchannels + 1, args.nfilters, (3, 7), padding=(1, 3))
        self.relu = nn.ReLU()
        self.conv = nn.Conv2d(args.nfilters * 3, args.match_filter_size, (1, 1))
        self.output = nn.Linear(args.match_filter_size, 1)
        if not args.bidirection:
            raise NotImplementedError("HIP MatchTensor expects bidirection=True (hyparam.py:88-105)")
        self._dims = dict(F=args.featsize, Hq=args.nhid_query // 2, Hd=args.nhid_doc // 2, C=args.nchannels,
                          NF=args.nfilters, MF=args.match_filter_size)
        self._pack = lib.PackCache()

    def _weights(self):
        def build():
            q = lstm_cat_weights(self.query_encoder.rnns[0])
            d = lstm_cat_weights(self.document_encoder.rnns[0])
            t = dict(proj_w=self.linear_projection.weight, proj_b=self.linear_projection.bias,
                     q_wih=q[0], q_whh=q[1], q_bih=q[2], q_bhh=q[3], d_wih=d[0], d_whh=d[1], d_bih=d[2], d_bhh=d[3],
                     qproj_w=self.query_projection.weight, qproj_b=self.query_projection.bias,
                     dproj_w=self.document_projection.weight, dproj_b=self.document_projection.bias,
                     alpha=self.exact_match_channel.alpha,
                     conv1_w=self.conv1.weight, conv1_b=self.conv1.bias, conv2_w=self.conv2.weight,
                     conv2_b=self.conv2.bias, conv3_w=self.conv3.weight, conv3_b=self.conv3.bias,
                     conv_w=self.conv.weight, conv_b=self.conv.bias, out_w=self.output.weight, out_b=self.output.bias)
            return lib.Packed(lib.MatchTensorWeights, t, self._dims)
        params = [p for n, p in self.named_parameters() if not n.startswith("word_embeddings")]
        return self._pack.get(params, build)

    def forward(self, batch_queries, query_len, batch_docs, doc_len, return_parts=False):
        assert batch_queries.shape[0] == batch_docs.shape[0]
        if self.training and (self.emb_drop.p > 0):
            raise NotImplementedError("HIP MatchTensor implements the eval-mode forward (SURVEY.md Appendix E7)")
        table = self.word_embeddings.table
        lib.require_device(batch_queries, batch_docs, query_len, doc_len, table)
        L = lib.load()
        q, d = lib.ids64(batch_queries), lib.ids64(batch_docs)
        ql, dl = lib.ids64(query_len), lib.ids64(doc_len.reshape(-1))
        B, QL = q.shape
        N, DL = d.shape[1], d.shape[2]
        w = self._weights()
        dev = q.device
        nbytes = L.nir_matchtensor_workspace_bytes(B, N, QL, DL, w.ref())
        ws = lib.workspace(nbytes, dev)
        scores = torch.empty(B, N, device=dev, dtype=torch.float32)
        if B == 0 and not return_parts:
            return scores
        parts = [None] * 4
        if return_parts:
            dm = self._dims
            parts = [torch.empty(B, QL, 2 * dm["Hq"], device=dev), torch.empty(B * N, DL, 2 * dm["Hd"], device=dev),
                     torch.empty(B, QL, dm["C"], device=dev), torch.empty(B * N, DL, dm["C"], device=dev)]
        lib.check(L.nir_matchtensor_score(lib.ptr(q), lib.ptr(ql), lib.ptr(d), lib.ptr(dl), B, N, QL, DL,
                                          lib.ptr(table), table.shape[0], table.shape[1], w.ref(),
                                          lib.ptr(ws), ws.numel(), lib.ptr(scores), *[lib.ptr(p) for p in parts],
                                          lib.stream()), "nir_matchtensor_score")
        return (scores, parts) if return_parts else scores
