"""MatchTensor (drop-in for neuroir.rankers.mtensor.MatchTensor, /root/reference/neuroir/rankers/mtensor.py:24-158).

Same constructor args, attribute names and state-dict keys; forward enqueues the whole pipeline with one
C-ABI call (nir_matchtensor_score): gather+projection GEMM, gate GEMMs, two BiLSTM recurrences, channel
projections, per-query weight folding and the fused interaction/conv/max-pool head.  The [B*N,51,QL,DL]
match tensor of the reference is never built (csrc/mtensor.hip).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import autograd as A
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


def interaction_bounded(m):
    """Host-side bound check, once per weight version: the encoder outputs lie in (-1, 1), so |Pq| and |Pd| are at most the largest
    L1 row norm of the projection plus |bias|, and the folded operand U is at most 3 max|conv_w| max|Pq|.  Below 2^15 the interaction
    GEMM of the head may use the fp16 two-term split (csrc/mtensor.hip, mt_head_kernel<true>)."""
    with torch.no_grad():
        def reach(lin):
            return float((lin.weight.detach().abs().sum(1) + lin.bias.detach().abs()).max())
        pq, pd = reach(m.query_projection), reach(m.document_projection)
        cw = max(float(c.weight.detach().abs().max()) for c in (m.conv1, m.conv2, m.conv3))
        return max(pd, pq, 3.0 * cw * pq) < 32768.0


def attach_projection_fragments(pk):
    """document_projection as the B operand of the head kernel's fused channel projection (csrc/mtensor.hip): [C, 2Hd] zero-padded to
    [64][k rounded up to 32], split into two fp16 terms and re-ordered into MFMA fragments [k/32][4 column tiles][2 terms][64 lanes][8]."""
    wd = pk.keep["dproj_w"]
    C_, K = wd.shape
    if not (pk.struct.bounded and wd.is_cuda and C_ <= 64 and K % 4 == 0 and K >= 8):
        return pk
    kp = (K + 31) // 32 * 32
    pad = torch.zeros(64, kp, device=wd.device, dtype=torch.float32)
    pad[:C_, :K] = wd
    planes = torch.stack(lib.split_f16x2(pad, kp))                                   # [2 terms, 64, kp] int16
    pk.keep["dproj_frag"] = planes.view(2, 4, 16, kp // 32, 4, 8).permute(3, 1, 0, 4, 2, 5).contiguous()
    pk.struct.dproj_frag = pk.keep["dproj_frag"].data_ptr()
    return pk


def train_head(mod, q, d, pq, pd):
    """Differentiable interaction head (mtensor.py:100-131) shared by MatchTensor and M_MATCH_TENSOR: q [B,QL] / d [B,N,DL] ids, projected
    queries pq [B,QL,C] and documents pd [B*N,DL,C] -> scores [B,N].  `mod` holds exact_match_channel, conv1..3, conv, output."""
    B, QL = q.shape
    N, DL = d.shape[1], d.shape[2]
    M = B * N
    C = pq.shape[-1]
    pqe = pq.unsqueeze(1).expand(B, N, QL, C).reshape(M, QL, C)
    prod = pqe.unsqueeze(2) * pd.unsqueeze(1)                                              # [M,QL,DL,C]
    em = (q.unsqueeze(1).expand(B, N, QL).reshape(M, QL).unsqueeze(2) == d.reshape(M, DL).unsqueeze(1)).float()
    em = em * mod.exact_match_channel.alpha
    T = torch.cat((prod, em.unsqueeze(3)), 3).permute(0, 3, 1, 2).contiguous()            # [M,C+1,QL,DL]
    feats = []
    if A.mt_conv3_supported(T, mod.conv1, mod.conv2, mod.conv3):
        # direct convolutions, forward and backward (csrc/mt_conv_train.hip): no [M QL DL, C kh kw] patch rows, ReLU and the concatenation fused
        feats = [A.mt_conv3(T, mod.conv1, mod.conv2, mod.conv3)]
    for conv in (() if feats else (mod.conv1, mod.conv2, mod.conv3)):
        kh, kw = conv.kernel_size
        if 2 * conv.padding[0] == kh - 1 and 2 * conv.padding[1] == kw - 1 and DL * (kw | 1) * 4 <= 65536:      # (the kernel pads its LDS rows to an odd stride)
            rows = A.im2col_rows(T, (kh, kw), conv.padding)                                # [M*QL*DL,(C+1)*kh*kw], one launch (HIP)
        else:
            cols = F.unfold(T, (kh, kw), padding=conv.padding)                             # [M,(C+1)*kh*kw,QL*DL]
            rows = cols.transpose(1, 2).reshape(M * QL * DL, -1)
        feats.append(A.linear(rows, conv.weight.reshape(conv.out_channels, -1), conv.bias, act="relu"))
    g = A.linear(feats[0] if len(feats) == 1 else torch.cat(feats, 1), mod.conv.weight.reshape(mod.conv.out_channels, -1), mod.conv.bias)
    g = g.view(M, QL * DL, -1).max(1)[0]
    return A.linear(g, mod.output.weight, mod.output.bias).view(B, N)


class MatchTensor(nn.Module, lib.IdCheck):
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
        # eval mode: fold embedding -> Linear(E->F) -> LSTM input projection into one table per encoder (csrc/lstm_fold.hip)
        self.fold_embeddings = getattr(args, "fold_embeddings", True)
        self._fold = lib.PackCache(retain=1)

    def _folded_tables(self, w):
        table = self.word_embeddings.table

        def build():
            L = lib.load()
            V, E = table.shape
            F_ = self._dims["F"]
            x = torch.empty(V, F_, device=table.device, dtype=torch.float32)           # projected table x[v] = W_p table[v] + b_p
            t = table.detach().float().contiguous()
            lib.check(L.nir_linear_f32(lib.ptr(t), E, None, None, 0, 0, 0, lib.ptr(w.keep["proj_w"]), E, lib.ptr(w.keep["proj_b"]), None,
                                       lib.ptr(x), F_, V, F_, E, 0, lib.stream()), "nir_linear_f32")
            fq = lib.fold_lstm_table(x, w.keep["q_wih"], w.keep["q_bih"], w.keep["q_bhh"], self._dims["Hq"], 2, "f32")
            fd = lib.fold_lstm_table(x, w.keep["d_wih"], w.keep["d_bih"], w.keep["d_bhh"], self._dims["Hd"], 2, "f32")
            return fq, fd
        params = [table, self.linear_projection.weight, self.linear_projection.bias] + list(self.query_encoder.rnns[0].parameters()) \
            + list(self.document_encoder.rnns[0].parameters())
        return self._fold.get(params, build)

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
            pk = attach_projection_fragments(lib.Packed(lib.MatchTensorWeights, t, dict(self._dims, bounded=int(interaction_bounded(self)))))
            # W_hh of the folded recurrences goes onto the fp16 matrix cores as a two-term split: needs |w| < 2^15 (else: unfolded fp32 path)
            pk.rec_ok = max(float(q[1].detach().abs().max()), float(d[1].detach().abs().max())) < 32768.0
            return pk
        params = [p for n, p in self.named_parameters() if not n.startswith("word_embeddings")]
        return self._pack.get(params, build)

    def _forward_train(self, q, ql, d, dl):
        """Train-mode forward (mtensor.py:62-131 with dropout active), differentiable: lookups, projections, both BiLSTMs, the
        three convolutions (im2col rows x filter matrix), the 1x1 convolution and the output layer run on the HIP operators of
        autograd.py, im2col included (patch rows in one launch); the broadcast product, the exact-match comparison and the global max are
        tensor glue."""
        B, QL = q.shape
        N, DL = d.shape[1], d.shape[2]
        M = B * N
        table = self.word_embeddings.table
        p = self.emb_drop.p
        eq = A.dropout(A.embed(q, table), p, True)
        ed = A.dropout(A.embed(d.reshape(M, DL), table), p, True)
        xq = A.linear(eq, self.linear_projection.weight, self.linear_projection.bias)
        xd = A.linear(ed, self.linear_projection.weight, self.linear_projection.bias)
        hq = A.bilstm(xq, ql, self.query_encoder.rnns[0])
        hd = A.bilstm(xd, dl.reshape(-1), self.document_encoder.rnns[0])
        pq = A.linear(hq, self.query_projection.weight, self.query_projection.bias)            # [B,QL,C]
        pd = A.linear(hd, self.document_projection.weight, self.document_projection.bias)      # [M,DL,C]
        return train_head(self, q, d, pq, pd)

    def _generic_encoders(self):
        """encoder configurations outside the fused / folded kernels (rnn_encoder.py:28-60 admits GRU and stacked layers; hyparam pins the
        1-layer LSTM): the encoders then run as RNNEncoder modules and the head takes their states (nir_matchtensor_score_encoded)"""
        e = self.query_encoder
        return e.cell != 0 or e.nlayers != 1

    def _forward_generic(self, batch_queries, query_len, batch_docs, doc_len, return_parts):
        table = self.word_embeddings.table
        lib.require_device(batch_queries, batch_docs, query_len, doc_len, table)
        L = lib.load()
        q, d = self._clean_ids(batch_queries, batch_docs, table.shape[0])
        ql, dl = lib.ids64(query_len), lib.ids64(doc_len.reshape(-1))
        B, QL = q.shape
        N, DL = d.shape[1], d.shape[2]
        dev, dm = q.device, self._dims
        w = self._weights()
        scores = torch.empty(B, N, device=dev, dtype=torch.float32)
        if B == 0:
            return (scores, [None] * 4) if return_parts else scores
        t = table.detach().float().contiguous()
        E, F_ = t.shape[1], dm["F"]

        def project(ids):                                        # embedding gather fused into the Linear(E -> F) GEMM
            x = torch.empty(ids.numel(), F_, device=dev, dtype=torch.float32)
            lib.check(L.nir_linear_f32(None, 0, lib.ptr(ids), lib.ptr(t), E, 1, 1, lib.ptr(w.keep["proj_w"]), E, lib.ptr(w.keep["proj_b"]), None,
                                       lib.ptr(x), F_, ids.numel(), F_, E, 0, lib.stream()), "nir_linear_f32")
            return x
        hq = self.query_encoder(project(q).view(B, QL, F_), ql)[1].contiguous()
        hd = self.document_encoder(project(d).view(B * N, DL, F_), dl)[1].contiguous()
        ws = lib.workspace(L.nir_matchtensor_workspace_bytes(B, N, QL, DL, w.ref()), dev)
        pq = torch.empty(B, QL, dm["C"], device=dev) if return_parts else None
        pd = torch.empty(B * N, DL, dm["C"], device=dev) if return_parts else None
        lib.check(L.nir_matchtensor_score_encoded(lib.ptr(q), lib.ptr(d), lib.ptr(hq), lib.ptr(hd), B, N, QL, DL, w.ref(), lib.ptr(ws), ws.numel(),
                                                  lib.ptr(scores), lib.ptr(pq), lib.ptr(pd), lib.stream()), "nir_matchtensor_score_encoded")
        return (scores, [hq, hd, pq, pd]) if return_parts else scores

    def forward(self, batch_queries, query_len, batch_docs, doc_len, return_parts=False):
        assert batch_queries.shape[0] == batch_docs.shape[0]
        if self._generic_encoders():
            if self.training:
                raise NotImplementedError("HIP MatchTensor trains the hyparam configuration (1-layer LSTM encoders, autograd.py); "
                                          "GRU / stacked encoders are eval-only")
            return self._forward_generic(batch_queries, query_len, batch_docs, doc_len, return_parts)
        if self.training:
            lib.require_device(batch_queries, batch_docs, query_len, doc_len, self.word_embeddings.table)
            q, d = self._clean_ids(batch_queries, batch_docs, self.word_embeddings.table.shape[0])
            return self._forward_train(q, lib.ids64(query_len), d, lib.ids64(doc_len))
        table = self.word_embeddings.table
        lib.require_device(batch_queries, batch_docs, query_len, doc_len, table)
        L = lib.load()
        fold = self.fold_embeddings and self._dims["Hq"] >= 4 and self._dims["Hd"] >= 4 and self._weights().rec_ok
        if fold:      # the folded recurrences validate ids in-kernel
            q, d = lib.ids64(batch_queries), lib.ids64(batch_docs)
        else:
            q, d = self._clean_ids(batch_queries, batch_docs, table.shape[0])
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
        if fold:
            fq, fd = self._folded_tables(w)
            lib.check(L.nir_matchtensor_score_folded(lib.ptr(q), lib.ptr(ql), lib.ptr(d), lib.ptr(dl), B, N, QL, DL, lib.ptr(fq), lib.ptr(fd),
                                                     lib.DTYPE_F32, table.shape[0], w.ref(), lib.ptr(ws), ws.numel(), lib.ptr(scores),
                                                     *[lib.ptr(p) for p in parts], lib.ptr(self._flag_word(dev)), lib.stream()),
                      "nir_matchtensor_score_folded")
        else:
            lib.check(L.nir_matchtensor_score(lib.ptr(q), lib.ptr(ql), lib.ptr(d), lib.ptr(dl), B, N, QL, DL,
                                              lib.ptr(table), table.shape[0], table.shape[1], w.ref(),
                                              lib.ptr(ws), ws.numel(), lib.ptr(scores), *[lib.ptr(p) for p in parts],
                                              lib.stream()), "nir_matchtensor_score")
        return (scores, parts) if return_parts else scores
