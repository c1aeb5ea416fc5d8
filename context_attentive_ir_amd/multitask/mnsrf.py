"""MNSRF -- ranking side of the multi-task neural session relevance framework (drop-in for
neuroir.multitask.mnsrf.MNSRF, /root/reference/neuroir/multitask/mnsrf.py:10-162).

encode():        BiLSTM over every query of a session + max pooling over time -> memory_bank [B,S,nhid_query];
                 a unidirectional session LSTM over the S pooled queries   -> session_bank [B,S,nhid_session]
rank_document(): BiLSTM + max pooling over every candidate document; score = tanh(W [q_t ; t > 0 ? s_t : 0] + b) . d
One C-ABI call each (nir_mnsrf_encode_states / nir_mnsrf_rank, csrc/mnsrf.hip).  At the reference's sizes (256 units per direction, 1024
session units) the encoders run on the resident-weight cluster recurrence (csrc/lstm_cluster.hip: W_hh spread over four CUs, gate rows
gathered from a table folded once per weight version, max over time fused) and the session LSTM on the fp16-term step kernel -- built
once per parameter version by `_weights()`; other sizes, a training model or `fold_embeddings=False` take the streaming form (one GEMM +
one cell launch per time step).  The suggestion decoder / generator are parameter containers (greedy decode in multitask/suggest.py).
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
        # retain = 1: the pack owns the two folded gate tables (1.6 GB at V = 100 000); alternating train / eval must not pin eight dead sets
        self._pack = lib.PackCache(retain=1)
        # the folded gate tables (V x 2048 floats per encoder) + pre-split recurrent weights of the resident-weight path: on in eval mode while
        # both tables stay under `fold_budget_bytes` (100 000 words: 1.6 GB)
        self.fold_embeddings = getattr(args, "fold_embeddings", True)
        self.fold_budget_bytes = 64 << 30
        # the four-CU cluster recurrence (csrc/lstm_cluster.hip) needs its four members co-resident; a cluster that cannot make progress raises bit 2
        # of the device's error word and leaves (bounded poll).  Fail safe: the first such event switches this model to the streaming recurrence
        # (nir_birnn_steps_fwd behind nir_mnsrf_*) for good -- lib.Flags runs _cluster_failed before it raises the failed call's RuntimeError.
        self.resident_recurrence = True
        self.uses_cluster = True                            # the wrappers publish / poll the error word even with id_check_interval = 0
        self._cluster_registered = None

    def _cluster_failed(self):
        if self.resident_recurrence:
            import logging
            logging.getLogger(__name__).warning("MNSRF: a recurrence cluster timed out; switching to the streaming recurrence (slower, placement independent)")
        self.resident_recurrence = False
        A.CLUSTER_TRAIN_FWD = False

    def _use_fold(self, table):
        per = table.shape[0] * 8 * 256 * 4
        return (self.fold_embeddings and not self.training and self._dims["Hq"] == 256 and self._dims["Hd"] == 256
                and 2 * per <= self.fold_budget_bytes)

    def _weights(self):
        table = self.embedder.word_embeddings.table
        fold = self._use_fold(table)
        # the resident-weight recurrence itself needs only the reference's sizes and an embedding width the per-batch form is sized for
        resident = (self.resident_recurrence and not self.training and self._dims["Hq"] == 256 and self._dims["Hd"] == 256
                    and (fold or table.shape[1] <= 300))
        if table.is_cuda and self._cluster_registered != table.device:
            lib.flags(table.device).on_cluster_timeout(self._cluster_failed)
            self._cluster_registered = table.device

        def build():
            q = lstm_cat_weights(self.query_encoder.encoder.rnns[0])
            d = lstm_cat_weights(self.document_encoder.encoder.rnns[0])
            s = lstm_cat_weights(self.session_query_encoder.encoder.rnns[0])
            t = dict(q_wih=q[0], q_whh=q[1], q_bih=q[2], q_bhh=q[3], d_wih=d[0], d_whh=d[1], d_bih=d[2], d_bhh=d[3],
                     s_wih=s[0], s_whh=s[1], s_bih=s[2], s_bhh=s[3],
                     proj_w=self.projection.linear.weight, proj_b=self.projection.linear.bias)
            pk = lib.Packed(lib.MnsrfWeights, t, self._dims)
            dev = pk.keep["q_whh"].device
            L = lib.load()
            HS = self._dims["HS"]
            if dev.type == "cuda":
                pk.struct.err = self._flag_word(dev).data_ptr()       # run time: the cluster recurrence's time-out bit (the device's one error word)

            def packed(fn, *args):
                """a W_hh fragment, or None when a weight lies outside the fp16 range of the split (one blocking flag read per weight version, like
                CARS): the pointer then stays NULL and the entry points take the exact fp32 recurrence"""
                flag = torch.zeros(1, dtype=torch.int32, device=dev)
                frag = fn(flag)
                return frag if int(flag.item()) == 0 else None

            if HS % 32 == 0 and dev.type == "cuda":            # session LSTM: W_hh as fp16 term pairs in MFMA-fragment order
                def s_frag(flag):
                    frag = torch.empty(max(1, L.nir_lstm_step_whh_frag_bytes(HS)), dtype=torch.uint8, device=dev)
                    lib.check(L.nir_lstm_step_pack_whh_frag(lib.ptr(pk.keep["s_whh"]), HS, lib.ptr(frag), lib.ptr(flag), lib.stream()), "nir_lstm_step_pack_whh_frag")
                    return frag
                frag = packed(s_frag)
                if frag is not None:
                    pk.keep["s_whh_frag"] = frag
                    pk.struct.s_whh_frag = frag.data_ptr()
            if resident and dev.type == "cuda":
                for k in ("q", "d"):
                    def c_frag(flag, k=k):
                        frag = torch.empty(L.nir_lstm256_whh_frag_bytes(2), dtype=torch.uint8, device=dev)
                        lib.check(L.nir_lstm256_pack_whh_frag(lib.ptr(pk.keep[k + "_whh"]), 2, lib.ptr(frag), lib.ptr(flag), lib.stream()), "nir_lstm256_pack_whh_frag")
                        return frag
                    frag = packed(c_frag)
                    if frag is None:
                        continue
                    pk.keep[k + "_whh_frag"] = frag
                    setattr(pk.struct, k + "_whh_frag", frag.data_ptr())
                    if fold:         # else: per-batch gate rows (one gather-GEMM per call in the folded order), same recurrence
                        ft = lib.fold_lstm_table(table.detach(), pk.keep[k + "_wih"], pk.keep[k + "_bih"], pk.keep[k + "_bhh"], 256, 2, "f32")
                        pk.keep[k + "_fold"] = ft
                        setattr(pk.struct, k + "_fold", ft.data_ptr())
            return pk
        skip = ("decoder.", "generator.") + (() if fold else ("embedder.",))
        params = [p for n, p in self.named_parameters() if not n.startswith(skip)] + [fold, resident]
        return self._pack.get(params, build)

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
        src, _ = self._clean_ids(source_rep.reshape(B * S, QL), None, table.shape[0])
        sl = lib.ids64(source_len.reshape(-1))
        w = self._weights()
        ws = lib.workspace(L.nir_mnsrf_workspace_bytes(B, S, 0, QL, 1, w.ref()), src.device)
        HS = self._dims["HS"]
        mem = torch.empty(B, S, 2 * self._dims["Hq"], device=src.device, dtype=torch.float32)
        sess = torch.empty(B, S, HS, device=src.device, dtype=torch.float32)
        # the decoder's initial states (mnsrf.py:96-112): the session LSTM's (h, c) after every query but the last, step-major along the batch
        # axis -- written by the same call (the session recurrence runs once)
        states = None
        if B > 0 and S > 1:
            states = (torch.empty(1, (S - 1) * B, HS, device=src.device, dtype=torch.float32),
                      torch.empty(1, (S - 1) * B, HS, device=src.device, dtype=torch.float32))
        if B > 0:
            lib.check(L.nir_mnsrf_encode_states(lib.ptr(src), lib.ptr(sl), B, S, QL, lib.ptr(table), table.shape[0], table.shape[1],
                                                w.ref(), lib.ptr(ws), ws.numel(), lib.ptr(mem), lib.ptr(sess),
                                                lib.ptr(states[0]) if states else None, lib.ptr(states[1]) if states else None, lib.stream()),
                      "nir_mnsrf_encode_states")
        self._src_len = source_len
        return mem, sess, states

    def rank_document(self, source_rep, memory_bank, session_bank, document_rep, document_len, source_len=None):
        """-> scores [B,S,N]  (mnsrf.py:116-162) from the query side encode() returned.  memory_bank / session_bank None: the query side
        is re-derived from the ids inside one fused call (`source_len` then defaults to the lengths given to encode())."""
        self._check_eval()
        table = self.embedder.word_embeddings.table
        lib.require_device(source_rep, document_rep, document_len, table)
        L = lib.load()
        B, S, N, DL = document_rep.shape
        QL = source_rep.shape[2]
        w = self._weights()
        scores = torch.empty(B, S, N, device=document_rep.device, dtype=torch.float32)
        if memory_bank is not None and session_bank is not None:
            d, _ = self._clean_ids(document_rep.reshape(B * S * N, DL), None, table.shape[0])
            dl = lib.ids64(document_len.reshape(-1))
            mem, sess = memory_bank.float().contiguous(), session_bank.float().contiguous()
            if tuple(mem.shape[:2]) != (B, S) or tuple(sess.shape[:2]) != (B, S):
                raise RuntimeError("rank_document: memory_bank %s / session_bank %s do not belong to %d x %d queries" % (
                    tuple(mem.shape), tuple(sess.shape), B, S))
            ws = lib.workspace(L.nir_mnsrf_workspace_bytes(B, S, N, 1, DL, w.ref()), d.device)
            if B > 0:
                lib.check(L.nir_mnsrf_rank(lib.ptr(mem), lib.ptr(sess), lib.ptr(d), lib.ptr(dl), B, S, N, DL, lib.ptr(table), table.shape[0],
                                           table.shape[1], w.ref(), lib.ptr(ws), ws.numel(), lib.ptr(scores), lib.stream()), "nir_mnsrf_rank")
            return scores
        src_len = source_len if source_len is not None else getattr(self, "_src_len", None)
        if src_len is None:
            raise RuntimeError("rank_document needs the query lengths: call encode() first or pass source_len")
        if src_len.numel() != source_rep.shape[0] * source_rep.shape[1]:
            raise RuntimeError("rank_document: %d query lengths for %d x %d queries (lengths cached by encode() belong to another "
                               "batch? pass source_len)" % (src_len.numel(), source_rep.shape[0], source_rep.shape[1]))
        lib.require_device(src_len)
        src, d = self._clean_ids(source_rep.reshape(B * S, QL), document_rep.reshape(B * S * N, DL), table.shape[0])
        sl, dl = lib.ids64(src_len.reshape(-1)), lib.ids64(document_len.reshape(-1))
        ws = lib.workspace(L.nir_mnsrf_workspace_bytes(B, S, N, QL, DL, w.ref()), src.device)
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
