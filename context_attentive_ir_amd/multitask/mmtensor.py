"""M_MATCH_TENSOR -- ranking side of the session-aware MatchTensor (drop-in for
neuroir.multitask.mmtensor.M_MATCH_TENSOR, /root/reference/neuroir/multitask/mmtensor.py:10-189).

Its document ranking is MatchTensor applied to the B*S (session, query) rows of a batch: `encode` gives the projected
queries (mmtensor.py:70-88) and `rank_document` the scores [B,S,N] (mmtensor.py:127-189), both through the same HIP
pipeline as rankers.MatchTensor (one nir_matchtensor_score call; the [B*S*N,51,QL,DL] match tensor is never built).
The session-level encoder, the suggestion decoder and the generator only feed query suggestion (`decode`), which is
outside the hot path: their parameters are kept under the reference's state-dict keys so checkpoints load with
strict=True, `decode` raises, and `encode` returns None for session_bank / states.
"""
import torch
import torch.nn as nn

from .. import autograd as A
from .. import lib
from . import suggest
from ..encoders.rnn_encoder import lstm_cat_weights
from ..rankers.mtensor import ExactMatchChannel, train_head
from .layers import Embedder, Encoder


class _PlainDecoderParams(nn.Module):
    """`decoder.decoder.rnn.*` (RNNDecoder without attention); parameters only."""

    def __init__(self, emsize, nhid):
        super().__init__()
        self.rnn = nn.LSTM(emsize, nhid, 1, batch_first=True)


class M_MATCH_TENSOR(nn.Module, lib.IdCheck):
    def __init__(self, args):
        super().__init__()
        if args.rnn_type != "LSTM" or not args.bidirection or args.nlayers != 1:
            raise NotImplementedError("HIP M_MATCH_TENSOR expects the reference configuration: 1-layer bidirectional LSTM encoders (GRU / stacked layers fail in "
                                      "the reference's own session loop: rnn_encoder.py:77-91 on the states it hands back as init_states)")
        self.embedder = Embedder(args.emsize, args.src_vocab_size, args.dropout_emb)
        self.linear_projection = nn.Linear(args.emsize, args.featsize)
        self.query_encoder = Encoder(args.rnn_type, args.featsize, args.bidirection, args.nlayers, args.nhid_query,
                                     args.dropout_rnn)
        self.document_encoder = Encoder(args.rnn_type, args.featsize, args.bidirection, args.nlayers,
                                        args.nhid_document, args.dropout_rnn)
        self.query_projection = nn.Linear(args.nhid_query, args.nchannels)
        self.document_projection = nn.Linear(args.nhid_document, args.nchannels)
        self.exact_match_channel = ExactMatchChannel()
        self.conv1 = nn.Conv2d(args.nchannels + 1, args.nfilters, (3, 3), padding=1)
        self.conv2 = nn.Conv2d(args.nchannels + 1, args.nfilters, (3, 5), padding=(1, 2))
        self.conv3 = nn.Conv2d(args.nchannels + 1, args.nfilters, (3, 7), padding=(1, 3))
        self.relu = nn.ReLU()
        self.conv = nn.Conv2d(args.nfilters * 3, args.match_filter_size, (1, 1))
        self.output = nn.Linear(args.match_filter_size, 1)
        # suggestion side: parameters only (state-dict compatibility)
        self.nhid_session = args.nhid_session
        self.session_query_encoder = Encoder(args.rnn_type, args.nchannels, False, args.nlayers, args.nhid_session,
                                             args.dropout_rnn)
        self.decoder = nn.Module()
        self.decoder.decoder = _PlainDecoderParams(args.emsize, args.nhid_session)
        self.dropout = nn.Dropout(args.dropout)
        self.generator = nn.Linear(args.nhid_session, args.tgt_vocab_size)
        self.regularize_coeff = args.regularize_coeff
        self.dec_dropout_p = float(args.dropout_rnn)        # RNNDecoder.dropout (decoders/decoder.py:87), train mode only
        self._dims = dict(F=args.featsize, Hq=args.nhid_query // 2, Hd=args.nhid_document // 2, C=args.nchannels,
                          NF=args.nfilters, MF=args.match_filter_size)
        self._pack = lib.PackCache()
        # eval mode: embedding -> Linear(E->F) -> LSTM input projection folded into one table per encoder, as rankers.MatchTensor does
        self.fold_embeddings = getattr(args, "fold_embeddings", True)
        self._fold = lib.PackCache(retain=1)

    def _folded_tables(self, w):
        table = self.embedder.word_embeddings.table

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
        params = [table, self.linear_projection.weight, self.linear_projection.bias] + list(self.query_encoder.encoder.rnns[0].parameters()) \
            + list(self.document_encoder.encoder.rnns[0].parameters())
        return self._fold.get(params, build)

    def _weights(self):
        def build():
            q = lstm_cat_weights(self.query_encoder.encoder.rnns[0])
            d = lstm_cat_weights(self.document_encoder.encoder.rnns[0])
            t = dict(proj_w=self.linear_projection.weight, proj_b=self.linear_projection.bias,
                     q_wih=q[0], q_whh=q[1], q_bih=q[2], q_bhh=q[3], d_wih=d[0], d_whh=d[1], d_bih=d[2], d_bhh=d[3],
                     qproj_w=self.query_projection.weight, qproj_b=self.query_projection.bias,
                     dproj_w=self.document_projection.weight, dproj_b=self.document_projection.bias,
                     alpha=self.exact_match_channel.alpha,
                     conv1_w=self.conv1.weight, conv1_b=self.conv1.bias, conv2_w=self.conv2.weight,
                     conv2_b=self.conv2.bias, conv3_w=self.conv3.weight, conv3_b=self.conv3.bias,
                     conv_w=self.conv.weight, conv_b=self.conv.bias, out_w=self.output.weight, out_b=self.output.bias)
            from ..rankers.mtensor import attach_projection_fragments, interaction_bounded
            pk = attach_projection_fragments(lib.Packed(lib.MatchTensorWeights, t, dict(self._dims, bounded=int(interaction_bounded(self)))))
            # W_hh of the folded recurrences goes onto the fp16 matrix cores as a two-term split: needs |w| < 2^15 (else: unfolded fp32 path)
            pk.rec_ok = max(float(q[1].detach().abs().max()), float(d[1].detach().abs().max())) < 32768.0
            return pk
        skip = ("embedder.", "session_query_encoder.", "decoder.", "generator.")
        params = [p for n, p in self.named_parameters() if not n.startswith(skip)]
        return self._pack.get(params, build)

    def _check_eval(self):
        if self.training and (self.dropout.p > 0 or self.embedder.dropout.p > 0):
            raise NotImplementedError("HIP M_MATCH_TENSOR implements the eval-mode forward (SURVEY.md Appendix E7)")

    def encode(self, source_rep, source_len):
        """source_rep [B,S,QL] ids, source_len [B,S] -> (projected_queries [B*S,QL,C], session_bank [B,S,nhid_session],
        states = (h, c) [1,(S-1)*B,nhid_session])  (mmtensor.py:70-125)."""
        self._check_eval()
        table = self.embedder.word_embeddings.table
        lib.require_device(source_rep, source_len, table)
        L, st = lib.load(), lib.stream()
        B, S, QL = source_rep.shape
        ids = lib.ids64(source_rep.reshape(B * S, QL))
        M, E, F_, C = B * S * QL, table.shape[1], self._dims["F"], self._dims["C"]
        x = torch.empty(B * S, QL, F_, device=ids.device, dtype=torch.float32)
        lib.check(L.nir_linear_f32(None, 0, lib.ptr(ids), lib.ptr(table), E, 1, 1, lib.ptr(self.linear_projection.weight), E,
                                   lib.ptr(self.linear_projection.bias), None, lib.ptr(x), F_, M, F_, E, 0, st), "nir_linear_f32")
        _, enc = self.query_encoder(x, source_len.reshape(-1))
        pq = torch.empty(B * S, QL, C, device=ids.device, dtype=torch.float32)
        Hq2 = enc.shape[2]
        lib.check(L.nir_linear_f32(lib.ptr(enc), Hq2, None, None, 0, 0, 0, lib.ptr(self.query_projection.weight), Hq2,
                                   lib.ptr(self.query_projection.bias), None, lib.ptr(pq), C, M, C, Hq2, 0, st), "nir_linear_f32")
        self._src_len = source_len                     # rank_document's signature carries no query lengths
        # suggestion side (mmtensor.py:88-124): max over the query positions -> unidirectional session LSTM -> session bank and the
        # decoder's initial states (the state after every query but the last, step-major along the batch axis)
        mem = torch.empty(B * S, C, device=ids.device, dtype=torch.float32)
        lib.check(L.nir_maxpool_time_f32(lib.ptr(pq), B * S, QL, C, lib.ptr(mem), st), "nir_maxpool_time_f32")
        session_bank, states = suggest.session_states(mem.view(B, S, C), self.session_query_encoder.encoder.rnns[0])
        return pq, session_bank, states

    def rank_document(self, source_rep, projected_queries, session_bank, document_rep, document_len, source_len=None):
        """-> scores [B,S,N]  (mmtensor.py:127-189).  The query side is re-derived from the ids inside the fused call
        (bitwise the same values as `projected_queries`); `source_len` defaults to the lengths given to encode()."""
        self._check_eval()
        table = self.embedder.word_embeddings.table
        src_len = source_len if source_len is not None else getattr(self, "_src_len", None)
        if src_len is None:
            raise RuntimeError("rank_document needs the query lengths: call encode() first or pass source_len")
        if src_len.numel() != source_rep.shape[0] * source_rep.shape[1]:
            raise RuntimeError("rank_document: %d query lengths for %d x %d queries (lengths cached by encode() belong to another "
                               "batch? pass source_len)" % (src_len.numel(), source_rep.shape[0], source_rep.shape[1]))
        lib.require_device(source_rep, document_rep, document_len, src_len, table)
        L = lib.load()
        B, S, N, DL = document_rep.shape
        QL = source_rep.shape[2]
        w = self._weights()
        fold = self.fold_embeddings and self._dims["Hq"] >= 4 and self._dims["Hd"] >= 4 and w.rec_ok
        if fold:      # the folded recurrences validate ids in-kernel
            q, d = lib.ids64(source_rep.reshape(B * S, QL)), lib.ids64(document_rep.reshape(B * S, N, DL))
        else:
            q, d = self._clean_ids(source_rep.reshape(B * S, QL), document_rep.reshape(B * S, N, DL), table.shape[0])
        ql, dl = lib.ids64(src_len.reshape(-1)), lib.ids64(document_len.reshape(-1))
        ws = lib.workspace(L.nir_matchtensor_workspace_bytes(B * S, N, QL, DL, w.ref()), q.device)
        scores = torch.empty(B * S, N, device=q.device, dtype=torch.float32)
        if B * S > 0 and fold:
            fq, fd = self._folded_tables(w)
            lib.check(L.nir_matchtensor_score_folded(lib.ptr(q), lib.ptr(ql), lib.ptr(d), lib.ptr(dl), B * S, N, QL, DL, lib.ptr(fq), lib.ptr(fd),
                                                     lib.DTYPE_F32, table.shape[0], w.ref(), lib.ptr(ws), ws.numel(), lib.ptr(scores),
                                                     None, None, None, None, lib.ptr(self._flag_word(q.device)), lib.stream()),
                      "nir_matchtensor_score_folded")
        elif B * S > 0:
            lib.check(L.nir_matchtensor_score(lib.ptr(q), lib.ptr(ql), lib.ptr(d), lib.ptr(dl), B * S, N, QL, DL,
                                              lib.ptr(table), table.shape[0], table.shape[1], w.ref(), lib.ptr(ws), ws.numel(),
                                              lib.ptr(scores), None, None, None, None, lib.stream()), "nir_matchtensor_score")
        return scores.view(B, S, N)

    def forward(self, source_rep, source_len, target_rep, target_len, target_seq, document_rep, document_len, document_label):
        """mmtensor.py:191-259 (train mode) -> {'ranking_loss', 'suggestion_loss'}, differentiable through the HIP operators of autograd.py:
        the projected queries are computed ONCE and feed both the interaction head and, max-pooled, the session LSTM whose states start
        the teacher-forced decoder -- as in the reference, where encode()'s tensor is handed to rank_document()."""
        table = self.embedder.word_embeddings.table
        lib.require_device(source_rep, document_rep, table)
        B, S, QL = source_rep.shape
        N, DL = document_rep.shape[2], document_rep.shape[3]
        q, d = self._clean_ids(source_rep.reshape(B * S, QL), document_rep.reshape(B * S, N, DL), table.shape[0])
        pe, p = self.embedder.dropout.p, self.dropout.p
        lp = self.linear_projection
        xq = A.linear(A.dropout(A.embed(q, table), pe, True), lp.weight, lp.bias)
        hq = A.dropout(suggest.bilstm_train(xq, lib.ids64(source_len.reshape(-1)), self.query_encoder.encoder.rnns[0]), p, True)
        pq = A.linear(hq, self.query_projection.weight, self.query_projection.bias)               # [B*S,QL,C]
        mem = pq.max(1)[0].view(B, S, -1)                                                         # all positions take part (:88-90)
        h_steps, c_steps = A.lstm_seq(mem, self.session_query_encoder.encoder.rnns[0])            # session states, every step
        # ranking (mmtensor.py:127-189; the dropout on the encoded documents is `self.dropout` as well, :153)
        xd = A.linear(A.dropout(A.embed(d.reshape(B * S * N, DL), table), pe, True), lp.weight, lp.bias)
        hd = A.dropout(suggest.bilstm_train(xd, lib.ids64(document_len.reshape(-1)), self.document_encoder.encoder.rnns[0]), p, True)
        pd = A.linear(hd, self.document_projection.weight, self.document_projection.bias)
        scores = train_head(self, q, d, pq, pd).view(B, S, N)
        return {"ranking_loss": A.bce_with_logits(scores, document_label.float()),
                "suggestion_loss": suggest.suggestion_loss(self, h_steps, c_steps, target_rep, target_seq)}

    def decode(self, states, max_len, src_dict, tgt_dict, batch_size, session_len, use_cuda=True, tgt2src=None, **kwargs):
        """mmtensor.py:281-325 (greedy, decoder without attention) -> {'predictions': LongTensor [batch_size, session_len, max_len]}."""
        self._check_eval()
        return suggest.greedy_decode(self, states, max_len, src_dict, tgt_dict, batch_size, session_len, self.embedder.word_embeddings.table,
                                     self.decoder.decoder.rnn, self.generator, tgt2src)
