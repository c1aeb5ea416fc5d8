"""CARS -- context attentive ranking and suggestion, RANKING path (drop-in for neuroir.multitask.cars.CARS,
/root/reference/neuroir/multitask/cars.py:13-540,671-691).

Scope: encode, encode_document, encode_clicks, encode_session/rank, rank_document and the
ranking loss run on hand-written HIP kernels.  The query-suggestion decoder (cars.py:605-657,706-791) is out
of scope: its parameters are kept (same state-dict keys, so reference checkpoints load strictly) but
`decode`/the suggestion loss raise NotImplementedError.

HIP mapping
  encode / encode_document : nir_cars_encode  = gather fused into the gate GEMM (fp32 MFMA) -> BiLSTM recurrence
                             -> attention MLP GEMM+tanh -> masked softmax + weighted sum (one wave per sequence)
  encode_clicks + session  : nir_cars_rank_session = click attention with the reference's batch-dependent mask
                             quirk (Appendix E2), then the sequential session loop (cross attention over previous
                             states incl. the zero state, ranknet maxout, two LSTM steps) with no host sync.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import lib
from ..encoders.rnn_encoder import lstm_cat_weights
from ..modules import Maxout
from .layers import Embedder, Encoder


def _attn_mlp(n, p):
    return nn.Sequential(nn.Linear(n, n), nn.Tanh(), nn.Dropout(p=p), nn.Linear(n, 1))


def _projector(i, o, p, bias):
    return nn.Sequential(OrderedDict([("dropout", nn.Dropout(p=p)), ("linear", nn.Linear(i, o, bias=bias))]))


class _SuggestionDecoderParams(nn.Module):
    """Parameter container for `decoder.decoder.*` (RNNDecoder: LSTM + general attention), kept only so that
    reference checkpoints load; evaluated by nobody on the ranking path."""

    def __init__(self, emsize, nhid):
        super().__init__()
        self.rnn = nn.LSTM(emsize, nhid, 1, batch_first=True)
        self.attn = nn.Module()
        self.attn.linear_in = nn.Linear(nhid, nhid, bias=False)
        self.attn.linear_out = nn.Linear(2 * nhid, nhid, bias=False)


class CARS(nn.Module):
    def __init__(self, args):
        super().__init__()
        if args.rnn_type != "LSTM" or not args.bidirection or args.nlayers != 1 or args.pool_type != "attn":
            raise NotImplementedError("HIP CARS supports the reference configuration (hyparam.py:197-225): "
                                      "LSTM, bidirectional, 1 layer, pool_type='attn'")
        if args.query_session_off or args.doc_session_off or args.turn_ranker_off:
            raise NotImplementedError("HIP CARS needs both session encoders and the ranker on")
        p = args.dropout
        self.embedder = Embedder(args.emsize, args.src_vocab_size, args.dropout_emb)
        self.query_encoder = Encoder(args.rnn_type, args.emsize, True, 1, args.nhid_query, args.dropout_rnn)
        self.document_encoder = Encoder(args.rnn_type, args.emsize, True, 1, args.nhid_document, args.dropout_rnn)
        self.q_attn = _attn_mlp(args.nhid_query, p)
        self.d_attn = _attn_mlp(args.nhid_document, p)
        self.nhid_session_query = args.nhid_session_query
        self.session_query_encoder = Encoder(args.rnn_type, args.nhid_query, False, 1, args.nhid_session_query, args.dropout_rnn)
        self.session_query_attn = nn.Linear(args.nhid_session_query, args.nhid_query)
        self.session_query_inner_attn = _attn_mlp(args.nhid_session_query, p)
        self.click_attn = _attn_mlp(args.nhid_document, p)
        self.nhid_session_document = args.nhid_session_document
        self.session_doc_encoder = Encoder(args.rnn_type, args.nhid_document, False, 1, args.nhid_session_document, args.dropout_rnn)
        self.session_doc_attn = nn.Linear(args.nhid_session_document, args.nhid_document)
        self.session_doc_inner_attn = _attn_mlp(args.nhid_session_document, p)
        sess = args.nhid_session_query + args.nhid_session_document
        self.shared_session_projector = _projector(sess, args.nhid_document, p, False)
        self.q_projection = _projector(args.nhid_query, args.nhid_document, p, True)
        self.private_session_projector1 = _projector(sess, args.nhid_document, p, False)
        self.ranknet = Maxout(args.nhid_document * 4, 3, [256, 128, 1], [2, 2, 2])
        self.no_recommender = args.turn_recommender_off
        if not self.no_recommender:  # suggestion-only parameters: containers for checkpoint compatibility
            self.private_session_projector2 = _projector(sess, args.nhid_document, p, False)
            self.transform_hid = _projector(sess, args.nhid_decoder, p, True)
            self.transform_cell = _projector(sess, args.nhid_decoder, p, True)
            self.decoder = nn.Module()
            self.decoder.decoder = _SuggestionDecoderParams(args.emsize, args.nhid_decoder)
            self.dec_attn = nn.Linear(args.nhid_query, args.nhid_decoder, bias=False)
            self.token_prob_predictor1 = nn.Linear(args.nhid_decoder, args.nhid_document, bias=False)
            self.token_prob_predictor2 = nn.Linear(args.nhid_document, args.tgt_vocab_size, bias=False)
        self.dropout = nn.Dropout(args.dropout)
        self.regularize_coeff = args.regularize_coeff
        self.no_ranker = False
        self.no_query_session_encoding = self.no_document_session_encoding = False
        self.pool_type = args.pool_type
        self.lambda1, self.lambda2 = args.lambda1, args.lambda2
        if args.nhid_query != args.nhid_document or args.nhid_session_query != args.nhid_session_document:
            raise NotImplementedError("HIP CARS expects nhid_query == nhid_document and equal session sizes")
        self._dims = dict(D=args.nhid_document, HS=args.nhid_session_query)
        self._pq, self._pd, self._ps = lib.PackCache(), lib.PackCache(), lib.PackCache()
        # Inference-time folding of the embedding table into the LSTM input projection (csrc/lstm_fold.hip): on in eval
        # mode while the two folded tables (V x 8H each) stay under `fold_budget_bytes`; `compute_dtype` "bf16" selects the
        # bf16 folded table + bf16 MFMA recurrence (BASELINE config 5), "f32" is the parity path.
        self.fold_embeddings = getattr(args, "fold_embeddings", True)
        self.fold_budget_bytes = 64 << 30
        self.compute_dtype = getattr(args, "compute_dtype", "f32")
        self._fq, self._fd = lib.PackCache(), lib.PackCache()
        self._err_flag = None

    # ---- weight packing -------------------------------------------------------------------------
    def _enc_weights(self, which):
        enc = (self.query_encoder if which == "q" else self.document_encoder).encoder
        attn = self.q_attn if which == "q" else self.d_attn
        cache = self._pq if which == "q" else self._pd

        def build():
            wih, whh, bih, bhh = lstm_cat_weights(enc.rnns[0])
            return lib.Packed(lib.CarsEncoderWeights,
                              dict(wih=wih, whh=whh, bih=bih, bhh=bhh, attn0_w=attn[0].weight, attn0_b=attn[0].bias,
                                   attn3_w=attn[3].weight, attn3_b=attn[3].bias), dict(H=enc.hidden))
        return cache.get(list(enc.parameters()) + list(attn.parameters()), build)

    def _session_weights(self):
        def build():
            sq, sd = self.session_query_encoder.encoder.rnns[0], self.session_doc_encoder.encoder.rnns[0]
            mo = self.ranknet._linear_layers
            t = dict(click0_w=self.click_attn[0].weight, click0_b=self.click_attn[0].bias,
                     click3_w=self.click_attn[3].weight, click3_b=self.click_attn[3].bias,
                     sq_attn_w=self.session_query_attn.weight, sq_attn_b=self.session_query_attn.bias,
                     sd_attn_w=self.session_doc_attn.weight, sd_attn_b=self.session_doc_attn.bias,
                     sq_wih=sq.weight_ih_l0, sq_whh=sq.weight_hh_l0, sq_bih=sq.bias_ih_l0, sq_bhh=sq.bias_hh_l0,
                     sd_wih=sd.weight_ih_l0, sd_whh=sd.weight_hh_l0, sd_bih=sd.bias_ih_l0, sd_bhh=sd.bias_hh_l0,
                     qproj_w=self.q_projection.linear.weight, qproj_b=self.q_projection.linear.bias,
                     shared_w=self.shared_session_projector.linear.weight,
                     priv1_w=self.private_session_projector1.linear.weight,
                     mo0_w=mo[0].weight, mo0_b=mo[0].bias, mo1_w=mo[1].weight, mo1_b=mo[1].bias,
                     mo2_w=mo[2].weight, mo2_b=mo[2].bias)
            return lib.Packed(lib.CarsSessionWeights, t, self._dims)
        mods = [self.click_attn, self.session_query_attn, self.session_doc_attn, self.session_query_encoder,
                self.session_doc_encoder, self.q_projection, self.shared_session_projector,
                self.private_session_projector1, self.ranknet]
        return self._ps.get([p for m in mods for p in m.parameters()], build)

    def _check_eval(self):
        if self.training:
            raise NotImplementedError("HIP CARS implements the eval-mode forward (dropout is RNG-dependent, "
                                      "SURVEY.md Appendix E7)")

    def _folded_table(self, which, w):
        """[V, 8H] folded gate table of one encoder (fp32 or bf16), rebuilt when the table or the LSTM weights change."""
        table = self.embedder.word_embeddings.table
        enc = (self.query_encoder if which == "q" else self.document_encoder).encoder
        cache = self._fq if which == "q" else self._fd
        dt = self.compute_dtype
        return cache.get([table] + list(enc.parameters()) + [dt],
                         lambda: lib.fold_lstm_table(table, w.keep["wih"], w.keep["bih"], w.keep["bhh"], w.struct.H, 2, dt))

    def _use_fold(self, table, H):
        if not self.fold_embeddings or self.training:
            return False
        per = table.shape[0] * 8 * H * (2 if self.compute_dtype == "bf16" else 4)
        return 2 * per <= self.fold_budget_bytes and H >= 8 and (2 * H) % 64 == 0

    def check_ids(self):
        """Raise IndexError if any folded-path call since the last check saw a token id outside the vocabulary (the
        reference's nn.Embedding raises at the lookup; the kernels clamp to row 0 and set a device flag instead of faulting).
        Reads one int from the device, i.e. synchronises -- call it outside latency-critical loops."""
        if self._err_flag is not None and int(self._err_flag.item()) != 0:
            self._err_flag.zero_()
            raise IndexError("index out of range in self (token id outside [0, src_vocab_size))")

    def _encode_seqs(self, which, ids, lens, want_encoded):
        table = self.embedder.word_embeddings.table
        lib.require_device(ids, lens, table)
        L = lib.load()
        ids, lens = lib.ids64(ids), lib.ids64(lens)
        M, T = ids.shape
        w = self._enc_weights(which)
        H2 = 2 * w.struct.H
        dev = ids.device
        if self._use_fold(table, w.struct.H):
            folded = self._folded_table(which, w)
            if self._err_flag is None or self._err_flag.device != dev:
                self._err_flag = torch.zeros(1, dtype=torch.int32, device=dev)
            ws = lib.workspace(L.nir_cars_encode_folded_workspace_bytes(M, T, w.ref()), dev)
            pooled = torch.empty(M, H2, device=dev, dtype=torch.float32)
            encoded = torch.empty(M, T, H2, device=dev, dtype=torch.float32) if want_encoded else None
            lib.check(L.nir_cars_encode_folded(lib.ptr(ids), lib.ptr(lens), M, T, lib.ptr(folded), lib.DTYPES[self.compute_dtype],
                                               table.shape[0], w.ref(), lib.ptr(ws), ws.numel(), lib.ptr(pooled),
                                               lib.ptr(encoded), lib.ptr(self._err_flag), lib.stream()), "nir_cars_encode_folded")
            return pooled, encoded
        ws = lib.workspace(L.nir_cars_encode_workspace_bytes(M, T, table.shape[1], w.ref()), dev)
        pooled = torch.empty(M, H2, device=dev, dtype=torch.float32)
        encoded = torch.empty(M, T, H2, device=dev, dtype=torch.float32) if want_encoded else None
        lib.check(L.nir_cars_encode(lib.ptr(ids), lib.ptr(lens), M, T, lib.ptr(table), table.shape[0], table.shape[1],
                                    w.ref(), lib.ptr(ws), ws.numel(), lib.ptr(pooled), lib.ptr(encoded), lib.stream()),
                  "nir_cars_encode")
        return pooled, encoded

    # ---- reference API ----------------------------------------------------------------------------
    def encode(self, queries, query_length):
        """cars.py:193-225 -> (pooled [B,S,2H], encoded [B*S,QL,2H], hidden=None)."""
        self._check_eval()
        B, S, QL = queries.shape
        pooled, enc = self._encode_seqs("q", queries.reshape(B * S, QL), query_length.reshape(-1), True)
        return pooled.view(B, S, -1), enc, None

    def encode_document(self, docs, docs_length):
        """cars.py:227-260 -> pooled docs [B,S,N,2H]."""
        self._check_eval()
        B, S, N, DL = docs.shape
        pooled, _ = self._encode_seqs("d", docs.reshape(B * S * N, DL), docs_length.reshape(-1), False)
        return pooled.view(B, S, N, -1)

    def _rank_session(self, pooled_q, pooled_docs, labels, want_clicks=False):
        lib.require_device(pooled_q, pooled_docs, labels)
        L = lib.load()
        B, S, N, D = pooled_docs.shape
        w = self._session_weights()
        dev = pooled_docs.device
        ws = lib.workspace(L.nir_cars_session_workspace_bytes(B, S, N, w.ref()), dev)
        pq, pdv, lab = pooled_q.float().contiguous(), pooled_docs.float().contiguous(), labels.float().contiguous()
        scores = torch.empty(B, S, N, device=dev, dtype=torch.float32)
        clicks = torch.empty(B, S, D, device=dev, dtype=torch.float32) if want_clicks else None
        lib.check(L.nir_cars_rank_session(lib.ptr(pq), lib.ptr(pdv), lib.ptr(lab), B, S, N, w.ref(), lib.ptr(ws),
                                          ws.numel(), lib.ptr(scores), lib.ptr(clicks), lib.stream()),
                  "nir_cars_rank_session")
        return scores, clicks

    def encode_clicks(self, docs, doc_labels):
        """cars.py:262-304 -> [B,S,2H] (computed by the same kernel family as rank_document)."""
        self._check_eval()
        B, S, N, D = docs.shape
        dummy_q = torch.zeros(B, S, D, device=docs.device)
        return self._rank_session(dummy_q, docs, doc_labels, want_clicks=True)[1]

    def rank_document(self, pooled_rep, document_rep, document_len, document_label, group=None, shard=False):
        """cars.py:522-540 -> (click_scores [B,S,N], hidden_states=None, session_attns=(None, None)).
        The decoder-initialisation states are suggestion-only and not produced.
        shard=True: candidate-sharded document encoding over the torch.distributed `group` + one all-gather of the
        pooled document vectors (sharding.sharded_pooled_docs); the session part runs replicated."""
        self._check_eval()
        if shard:
            from .. import sharding
            encoded_docs = sharding.sharded_pooled_docs(self.encode_document, document_rep, document_len, group)
        else:
            encoded_docs = self.encode_document(document_rep, document_len)
        scores, _ = self._rank_session(pooled_rep, encoded_docs, document_label)
        return scores, None, (None, None)

    def forward(self, source_rep, source_len, target_rep, target_len, target_seq, document_rep, document_len,
                document_label):
        """cars.py:542-669, ranking branch: {'ranking_loss': BCE-with-logits over [B,S,N], 'suggestion_loss': None}."""
        pooled, _, _ = self.encode(source_rep, source_len)
        scores, _, _ = self.rank_document(pooled, document_rep, document_len, document_label)
        lab = document_label.float().contiguous()
        loss = torch.empty(1, device=scores.device, dtype=torch.float32)
        rows = scores.shape[0] * scores.shape[1]
        lib.check(lib.load().nir_rank_loss_bce(lib.ptr(scores), lib.ptr(lab), rows, scores.shape[2], lib.ptr(loss),
                                               lib.stream()), "nir_rank_loss_bce")
        return {"ranking_loss": loss[0], "suggestion_loss": None, "click_scores": scores}

    def decode(self, **kwargs):
        raise NotImplementedError("CARS.decode (query suggestion, cars.py:706-791) is outside the accelerated "
                                  "ranking hot path (SURVEY.md section 8f, rank 4)")
