"""MNSRF -- ranking side of the multi-task neural session relevance framework (drop-in for
neuroir.multitask.mnsrf.MNSRF, /root/reference/neuroir/multitask/mnsrf.py:10-162).

encode():        BiLSTM over every query of a session + max pooling over time -> memory_bank [B,S,nhid_query];
                 a unidirectional session LSTM over the S pooled queries   -> session_bank [B,S,nhid_session]
rank_document(): BiLSTM + max pooling over every candidate document; score = tanh(W [q_t ; t > 0 ? s_t : 0] + b) . d
One C-ABI call each (nir_mnsrf_encode / nir_mnsrf_score, csrc/mnsrf.hip).  The hidden sizes (256 per direction, 1024
session units) are beyond the register-resident recurrences, so the LSTMs run in the streaming form (one MFMA GEMM +
one cell kernel per time step).  The suggestion decoder / generator are parameter containers only (`decode` raises).
"""
import torch
import torch.nn as nn
from collections import OrderedDict

from .. import autograd as A
from .. import lib
from . import suggest
from ..encoders.rnn_encoder import lstm_cat_weights
from .layers import Embedder, Encoder
from .mmtensor import _PlainDecoderParams


class MNSRF(nn.Module, lib.IdCheck):
    def __init__(self, args):
        super().__init__()
        if args.rnn_type != "LSTM" or not args.bidirection or args.nlayers != 1:
            raise NotImplementedError("HIP MNSRF expects the reference configuration: 1-layer bidirectional LSTM encoders (GRU / stacked layers fail in "
                                      "the reference's own session loop: rnn_encoder.py:77-91 on the states it hands back as init_states)")
        self.embedder = Embedder(args.emsize, args.src_vocab_size, args.dropout_emb)
        self.query_encoder = Encoder(args.rnn_type, args.emsize, args.bidirection, args.nlayers, args.nhid_query, args.dropout_rnn)
        self.document_encoder = Encoder(args.rnn_type, args.emsize, args.bidirection, args.nlayers, args.nhid_document,
                                        args.dropout_rnn)
        self.nhid_session = args.nhid_session
        self.session_query_encoder = Encoder(args.rnn_type, args.nhid_query, False, args.nlayers, args.nhid_session,
                                             args.dropout_rnn)
        self.decoder = nn.Module()
        self.decoder.decoder = _PlainDecoderParams(args.emsize, args.nhid_session)
        self.projection = nn.Sequential(OrderedDict([("linear", nn.Linear(args.nhid_query + args.nhid_session, args.nhid_document)),
                                                     ("tanh", nn.Tanh())]))
        self.dropout = nn.Dropout(args.dropout)
        self.generator = nn.Linear(args.nhid_session, args.tgt_vocab_size)
        self.regularize_coeff = args.regularize_coeff
        self.dec_dropout_p = float(args.dropout_rnn)        # RNNDecoder.dropout (decoders/decoder.py:87), train mode only
        self._dims = dict(Hq=args.nhid_query // 2, Hd=args.nhid_document // 2, HS=args.nhid_session)
        self._pack = lib.PackCache()

    def _weights(self):
        def build():
            q = lstm_cat_weights(self.query_encoder.encoder.rnns[0])
            d = lstm_cat_weights(self.document_encoder.encoder.rnns[0])
            s = lstm_cat_weights(self.session_query_encoder.encoder.rnns[0])
            t = dict(q_wih=q[0], q_whh=q[1], q_bih=q[2], q_bhh=q[3], d_wih=d[0], d_whh=d[1], d_bih=d[2], d_bhh=d[3],
                     s_wih=s[0], s_whh=s[1], s_bih=s[2], s_bhh=s[3],
                     proj_w=self.projection.linear.weight, proj_b=self.projection.linear.bias)
            return lib.Packed(lib.MnsrfWeights, t, self._dims)
        skip = ("embedder.", "decoder.", "generator.")
        return self._pack.get([p for n, p in self.named_parameters() if not n.startswith(skip)], build)

    def _check_eval(self):
        if self.training and (self.dropout.p > 0 or self.embedder.dropout.p > 0):
            raise NotImplementedError("HIP MNSRF implements the eval-mode forward (SURVEY.md Appendix E7)")

    def encode(self, source_rep, source_len):
        """source_rep [B,S,QL], source_len [B,S] -> (memory_bank [B,S,nhid_query], session_bank [B,S,nhid_session],
        states = (h, c) [1,(S-1)*B,nhid_session])  (mnsrf.py:62-114)."""
        self._check_eval()
        table = self.embedder.word_embeddings.table
        lib.require_device(source_rep, source_len, table)
        L = lib.load()
        B, S, QL = source_rep.shape
        src, sl = lib.ids64(source_rep.reshape(B * S, QL)), lib.ids64(source_len.reshape(-1))
        w = self._weights()
        ws = lib.workspace(L.nir_mnsrf_workspace_bytes(B, S, 0, QL, 1, w.ref()), src.device)
        mem = torch.empty(B, S, 2 * self._dims["Hq"], device=src.device, dtype=torch.float32)
        sess = torch.empty(B, S, self._dims["HS"], device=src.device, dtype=torch.float32)
        if B > 0:
            lib.check(L.nir_mnsrf_encode(lib.ptr(src), lib.ptr(sl), B, S, QL, lib.ptr(table), table.shape[0], table.shape[1],
                                         w.ref(), lib.ptr(ws), ws.numel(), lib.ptr(mem), lib.ptr(sess), lib.stream()),
                      "nir_mnsrf_encode")
        self._src_len = source_len
        # the decoder's initial states (mnsrf.py:96-112): the session LSTM's state after every query but the last, step-major along the
        # batch axis -- the fused encode keeps only h, so the (tiny) session recurrence is run once more with its cell states written out
        states = suggest.session_states(mem, self.session_query_encoder.encoder.rnns[0])[1] if B > 0 and S > 1 else None
        return mem, sess, states

    def rank_document(self, source_rep, memory_bank, session_bank, document_rep, document_len, source_len=None):
        """-> scores [B,S,N]  (mnsrf.py:116-162).  The query side is re-derived from the ids inside the fused call (the
        same values as memory_bank / session_bank); `source_len` defaults to the lengths given to encode()."""
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
        src, d = self._clean_ids(source_rep.reshape(B * S, QL), document_rep.reshape(B * S * N, DL), table.shape[0])
        sl, dl = lib.ids64(src_len.reshape(-1)), lib.ids64(document_len.reshape(-1))
        w = self._weights()
        ws = lib.workspace(L.nir_mnsrf_workspace_bytes(B, S, N, QL, DL, w.ref()), src.device)
        scores = torch.empty(B, S, N, device=src.device, dtype=torch.float32)
        if B > 0:
            lib.check(L.nir_mnsrf_score(lib.ptr(src), lib.ptr(sl), lib.ptr(d), lib.ptr(dl), B, S, N, QL, DL, lib.ptr(table),
                                        table.shape[0], table.shape[1], w.ref(), lib.ptr(ws), ws.numel(), lib.ptr(scores),
                                        lib.stream()), "nir_mnsrf_score")
        return scores

    def forward(self, source_rep, source_len, target_rep, target_len, target_seq, document_rep, document_len, document_label):
        """mnsrf.py:164-232 (train mode) -> {'ranking_loss', 'suggestion_loss'}, differentiable through the HIP operators of autograd.py
        (the 256-per-direction encoders run as two unidirectional lstm_seq passes, suggest.bilstm_train)."""
        table = self.embedder.word_embeddings.table
        lib.require_device(source_rep, document_rep, table)
        B, S, QL = source_rep.shape
        N, DL = document_rep.shape[2], document_rep.shape[3]
        q, d = self._clean_ids(source_rep.reshape(B * S, QL), document_rep.reshape(B * S * N, DL), table.shape[0])
        pe, p = self.embedder.dropout.p, self.dropout.p
        eq = A.dropout(A.embed(q, table), pe, True)
        mem = A.dropout(suggest.bilstm_train(eq, lib.ids64(source_len.reshape(-1)), self.query_encoder.encoder.rnns[0]), p, True)
        mem = mem.max(1)[0].view(B, S, -1)                                                        # mnsrf.py:79-83
        h_steps, c_steps = A.lstm_seq(mem, self.session_query_encoder.encoder.rnns[0])
        ed = A.dropout(A.embed(d, table), pe, True)
        docs = suggest.bilstm_train(ed, lib.ids64(document_len.reshape(-1)), self.document_encoder.encoder.rnns[0]).max(1)[0].view(B, S, N, -1)
        sess_in = torch.cat((torch.zeros_like(h_steps[:, :1]), h_steps[:, 1:]), 1)                # zeros at the first query, s_t afterwards (:139-146)
        comb = A.linear(torch.cat((mem, sess_in), 2), self.projection.linear.weight, self.projection.linear.bias, act="tanh")
        scores = (comb.unsqueeze(2) * docs).sum(3)
        return {"ranking_loss": A.bce_with_logits(scores, document_label.float()),
                "suggestion_loss": suggest.suggestion_loss(self, h_steps, c_steps, target_rep, target_seq)}

    def decode(self, states, max_len, src_dict, tgt_dict, batch_size, session_len, use_cuda=True, tgt2src=None, **kwargs):
        """mnsrf.py:251-296 (greedy, decoder without attention) -> {'predictions': LongTensor [batch_size, session_len, max_len]}."""
        self._check_eval()
        return suggest.greedy_decode(self, states, max_len, src_dict, tgt_dict, batch_size, session_len, self.embedder.word_embeddings.table,
                                     self.decoder.decoder.rnn, self.generator, tgt2src)
