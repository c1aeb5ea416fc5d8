"""CARS -- context attentive ranking and suggestion (drop-in for neuroir.multitask.cars.CARS,
/root/reference/neuroir/multitask/cars.py:13-791).

Everything CARS computes at inference time runs on hand-written HIP kernels through the C-ABI:

  encode / encode_document : eval mode: nir_cars_encode_folded -- embedding table folded into the LSTM input projection
                             once per weight version (csrc/lstm_fold.hip), recurrence gathers gate rows by token id, attention
                             MLP with tanh + Linear(D,1) in the GEMM epilogue, masked softmax + weighted sum.
                             (fold_embeddings=False: nir_cars_encode, gather fused into the per-batch gate GEMM.)
  encode_clicks + session  : nir_cars_rank_session (csrc/cars_session.hip) -- click attention with the reference's
                             batch-dependent mask quirk (Appendix E2), both session LSTM chains, cross attention over the
                             previous states incl. the zero state, maxout ranknet; honours query_session_off /
                             doc_session_off / turn_ranker_off (cars.py:185-188, 329-410, 485-533); on request also the
                             decoder-initialisation states and inner-attention pools (cars.py:382-456).
  decode                   : nir_cars_decode_greedy (csrc/cars_decode.hip) -- greedy suggestion, LSTM step + Luong general
                             attention + the 256 -> V_tgt projection, no host round trip per step.
  forward                  : the ranking loss (BCE-with-logits, cars.py:603); the suggestion loss / training mode is the
                             training-step row.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import autograd as A
from .. import lib
from ..constants import BOS, PAD
from ..encoders.rnn_encoder import lstm_cat_weights
from ..modules import Maxout
from .layers import Embedder, Encoder


def _attn_mlp(n, p):
    return nn.Sequential(nn.Linear(n, n), nn.Tanh(), nn.Dropout(p=p), nn.Linear(n, 1))


def _projector(i, o, p, bias):
    return nn.Sequential(OrderedDict([("dropout", nn.Dropout(p=p)), ("linear", nn.Linear(i, o, bias=bias))]))


class _RNNDecoderParams(nn.Module):
    """Parameters of `decoder.decoder.*` (RNNDecoder: nn.LSTM + GlobalAttention('general'), decoders/decoder.py:69-117) under
    the reference's state-dict keys; evaluated by nir_cars_decode_greedy, never by torch."""

    def __init__(self, emsize, nhid):
        super().__init__()
        self.rnn = nn.LSTM(emsize, nhid, 1, batch_first=True)
        self.attn = nn.Module()
        self.attn.linear_in = nn.Linear(nhid, nhid, bias=False)
        self.attn.linear_out = nn.Linear(2 * nhid, nhid, bias=False)


def _wsum(x, w):
    """sum_t w[b,t] x[b,t,:] (the reference's torch.bmm(x^T, w), cars.py:671-691) as a broadcast product + reduction: tensor glue of the train-mode
    forward, no library GEMM on the path."""
    return (x * w.unsqueeze(2)).sum(1)


class CARS(nn.Module, lib.IdCheck):
    def __init__(self, args):
        super().__init__()
        if args.rnn_type != "LSTM" or not args.bidirection or args.nlayers != 1 or args.pool_type != "attn":
            # rnn_type='GRU' and nlayers > 1 are not working configurations of the reference's session models either: the session loop hands
            # the previous state back as init_states (cars.py:378-402), and rnn_encoder.py:77 evaluates `if init_states:` on a tensor (GRU:
            # RuntimeError) / :79-91 splits the states by `nlayers` (2 layers: IndexError) -- verified against /root/reference.  The general
            # RNNEncoder (GRU, stacked layers, bridge) serves MATCH_TENSOR, which has no initial states.
            raise NotImplementedError("HIP CARS supports the reference configuration (hyparam.py:197-225): LSTM, bidirectional, 1 layer, "
                                      "pool_type='attn' (GRU / stacked layers fail in the reference's own session loop too)")
        if getattr(args, "attn_type", "general") != "general":
            raise NotImplementedError("HIP CARS decoder supports attn_type='general' (hyparam.py:206)")
        p = args.dropout
        self.no_ranker = bool(args.turn_ranker_off)
        self.no_recommender = bool(args.turn_recommender_off)
        self.no_query_session_encoding = bool(args.query_session_off)
        self.no_document_session_encoding = bool(args.doc_session_off)
        q_on, d_on = not self.no_query_session_encoding, not self.no_document_session_encoding
        need_docs = not (self.no_ranker and self.no_document_session_encoding)
        if (q_on and d_on and args.nhid_session_query != args.nhid_session_document) or args.nhid_query != args.nhid_document:
            raise NotImplementedError("HIP CARS expects nhid_query == nhid_document and equal session sizes")
        self.embedder = Embedder(args.emsize, args.src_vocab_size, args.dropout_emb)
        self.query_encoder = Encoder(args.rnn_type, args.emsize, True, 1, args.nhid_query, args.dropout_rnn)
        self.q_attn = _attn_mlp(args.nhid_query, p)
        if need_docs:                                       # cars.py:29-58
            self.document_encoder = Encoder(args.rnn_type, args.emsize, True, 1, args.nhid_document, args.dropout_rnn)
            self.d_attn = _attn_mlp(args.nhid_document, p)
        sess = 0
        if q_on:                                            # cars.py:60-79
            self.nhid_session_query = args.nhid_session_query
            self.session_query_encoder = Encoder(args.rnn_type, args.nhid_query, False, 1, args.nhid_session_query, args.dropout_rnn)
            self.session_query_attn = nn.Linear(args.nhid_session_query, args.nhid_query)
            self.session_query_inner_attn = _attn_mlp(args.nhid_session_query, p)
            sess += args.nhid_session_query
        if d_on:                                            # cars.py:82-107
            self.click_attn = _attn_mlp(args.nhid_document, p)
            self.nhid_session_document = args.nhid_session_document
            self.session_doc_encoder = Encoder(args.rnn_type, args.nhid_document, False, 1, args.nhid_session_document, args.dropout_rnn)
            self.session_doc_attn = nn.Linear(args.nhid_session_document, args.nhid_document)
            self.session_doc_inner_attn = _attn_mlp(args.nhid_session_document, p)
            sess += args.nhid_session_document
        self.session_rep_size = sess
        if sess > 0:
            self.shared_session_projector = _projector(sess, args.nhid_document, p, False)
        if not self.no_ranker:                              # cars.py:116-132
            self.q_projection = _projector(args.nhid_query, args.nhid_document, p, True)
            if sess > 0:
                self.private_session_projector1 = _projector(sess, args.nhid_document, p, False)
            self.ranknet = Maxout(args.nhid_document * 4, 3, [256, 128, 1], [2, 2, 2])
        if not self.no_recommender:                         # cars.py:134-178
            if sess == 0:
                raise ValueError("Both session-level RNNs cannot be off!")
            self.private_session_projector2 = _projector(sess, args.nhid_document, p, False)
            self.transform_hid = _projector(sess, args.nhid_decoder, p, True)
            self.transform_cell = _projector(sess, args.nhid_decoder, p, True)
            self.decoder = nn.Module()
            self.decoder.decoder = _RNNDecoderParams(args.emsize, args.nhid_decoder)
            self.dec_attn = nn.Linear(args.nhid_query, args.nhid_decoder, bias=False)
            pred_in = args.nhid_document if need_docs else args.emsize
            self.token_prob_predictor1 = nn.Linear(args.nhid_decoder, pred_in, bias=False)
            self.token_prob_predictor2 = nn.Linear(pred_in, args.tgt_vocab_size, bias=False)
        self.dropout = nn.Dropout(args.dropout)
        self.dec_dropout_p = float(args.dropout_rnn)        # RNNDecoder.dropout (decoders/decoder.py:87), train mode only
        self.regularize_coeff = args.regularize_coeff
        self.pool_type = args.pool_type
        self.lambda1, self.lambda2 = args.lambda1, args.lambda2
        hs = args.nhid_session_query if q_on else args.nhid_session_document
        self._dims = dict(D=args.nhid_document, HS=hs, HDEC=args.nhid_decoder, q_on=int(q_on), d_on=int(d_on),
                          rank_on=int(not self.no_ranker))
        self._pq, self._pd, self._ps, self._pdec = lib.PackCache(), lib.PackCache(), lib.PackCache(), lib.PackCache()
        # Inference-time folding of the embedding table into the LSTM input projection (csrc/lstm_fold.hip): on in eval
        # mode while the two folded tables (V x 8H each) stay under `fold_budget_bytes`; `compute_dtype` "f32_split2" = the opt-in precision tier
        # of round 5 (fp32 tables, h as ONE fp16 term in the recurrent product and the attention GEMM: NIR_DTYPE_F32_SPLIT2); "bf16" selects the
        # bf16 folded table + bf16 MFMA recurrence (BASELINE config 5), "f32" is the parity path.
        self.fold_embeddings = getattr(args, "fold_embeddings", True)
        self.fuse_attention_pooling = True   # attention MLP + masked softmax + weighted sum as one kernel (csrc/cars_attn.hip)
        self.fuse_decoder_argmax = True      # decode: 256 -> V_tgt projection + arg-max as one kernel, no [Bd, V_tgt] logits
        self.fold_decoder_step = True        # decode: per-token gate rows of the decoder LSTM folded into a [V, 4HD] table + fp16-term recurrent product
        self.fold_decoder_query = True       # decode: attn.linear_in folded into a second memory bank (one GEMM per decode instead of one per step)
        self.fold_budget_bytes = 64 << 30
        self.compute_dtype = getattr(args, "compute_dtype", "f32")
        self._fq, self._fd = lib.PackCache(retain=1), lib.PackCache(retain=1)

    # ---- weight packing -------------------------------------------------------------------------
    def _enc_weights(self, which):
        enc = (self.query_encoder if which == "q" else self.document_encoder).encoder
        attn = self.q_attn if which == "q" else self.d_attn
        cache = self._pq if which == "q" else self._pd

        def build():
            wih, whh, bih, bhh = lstm_cat_weights(enc.rnns[0])
            bounded = float(attn[0].weight.detach().abs().max()) < 32768.0
            # bit 1: embedding table and W_ih inside the fp16 split's range -> the per-batch gather-GEMM (fold_embeddings off / table over the
            # fold budget) runs on the fp16 two-term form
            table = self.embedder.word_embeddings.table
            gb = float(table.detach().abs().max()) < 32768.0 and float(wih.detach().abs().max()) < 32768.0
            # the folded recurrences run W_hh on the fp16 matrix cores (two-term split / single term): outside that range the encoder
            # takes the per-batch path with the exact fp32 recurrence (one check per weight version; bit 2 of `bounded` tells nir_cars_encode)
            rec_ok = float(whh.detach().abs().max()) < 32768.0
            pk = lib.Packed(lib.CarsEncoderWeights,
                            dict(wih=wih, whh=whh, bih=bih, bhh=bhh, attn0_w=attn[0].weight, attn0_b=attn[0].bias,
                                 attn3_w=attn[3].weight, attn3_b=attn[3].bias),
                            dict(H=enc.hidden, bounded=int(bounded) | (int(gb) << 1) | (int(rec_ok) << 2)))
            pk.rec_ok = rec_ok
            L = lib.load()
            nb = L.nir_lstm_whh_frag_bytes(enc.hidden, 2)
            if nb and pk.rec_ok and whh.is_cuda:     # W_hh pre-split in the lane order of the folded fp32 recurrence (once per weight version)
                pk.keep["whh_frag"] = torch.empty(nb, dtype=torch.uint8, device=whh.device)
                lib.check(L.nir_lstm_pack_whh_frag(lib.ptr(pk.keep["whh"]), enc.hidden, 2, lib.ptr(pk.keep["whh_frag"]), None, lib.stream()),
                          "nir_lstm_pack_whh_frag")
                pk.struct.whh_frag = pk.keep["whh_frag"].data_ptr()
            if bounded and 2 * enc.hidden == 256 and self.fuse_attention_pooling and attn[0].weight.is_cuda:
                # operand of the fused attention-pooling kernel (csrc/cars_attn.hip): attn0_w as two fp16 term planes in MFMA-fragment
                # order [K/32][16 column tiles][2 terms][64 lanes][8], built once per weight version
                planes = torch.stack(lib.split_f16x2(pk.keep["attn0_w"], 256))                 # [2, 256, 256] int16
                pk.keep["attn_frag"] = planes.view(2, 16, 16, 8, 4, 8).permute(3, 1, 0, 4, 2, 5).contiguous()
                pk.struct.attn_frag = pk.keep["attn_frag"].data_ptr()
            return pk
        return cache.get(list(enc.parameters()) + list(attn.parameters()) + [self.embedder.word_embeddings.table], build)

    def _session_modules(self):
        names = ["click_attn", "session_query_attn", "session_doc_attn", "session_query_encoder", "session_doc_encoder",
                 "q_projection", "shared_session_projector", "private_session_projector1", "ranknet",
                 "session_query_inner_attn", "session_doc_inner_attn", "transform_hid", "transform_cell"]
        return [getattr(self, n) for n in names if hasattr(self, n)]

    def _session_weights(self):
        def build():
            t = {}
            if hasattr(self, "click_attn"):
                sd = self.session_doc_encoder.encoder.rnns[0]
                t.update(click0_w=self.click_attn[0].weight, click0_b=self.click_attn[0].bias,
                         click3_w=self.click_attn[3].weight, click3_b=self.click_attn[3].bias,
                         sd_attn_w=self.session_doc_attn.weight, sd_attn_b=self.session_doc_attn.bias,
                         sd_wih=sd.weight_ih_l0, sd_whh=sd.weight_hh_l0, sd_bih=sd.bias_ih_l0, sd_bhh=sd.bias_hh_l0,
                         sd_inner0_w=self.session_doc_inner_attn[0].weight, sd_inner0_b=self.session_doc_inner_attn[0].bias,
                         sd_inner3_w=self.session_doc_inner_attn[3].weight, sd_inner3_b=self.session_doc_inner_attn[3].bias)
            if hasattr(self, "session_query_encoder"):
                sq = self.session_query_encoder.encoder.rnns[0]
                t.update(sq_attn_w=self.session_query_attn.weight, sq_attn_b=self.session_query_attn.bias,
                         sq_wih=sq.weight_ih_l0, sq_whh=sq.weight_hh_l0, sq_bih=sq.bias_ih_l0, sq_bhh=sq.bias_hh_l0,
                         sq_inner0_w=self.session_query_inner_attn[0].weight, sq_inner0_b=self.session_query_inner_attn[0].bias,
                         sq_inner3_w=self.session_query_inner_attn[3].weight, sq_inner3_b=self.session_query_inner_attn[3].bias)
            if not self.no_ranker:
                mo = self.ranknet._linear_layers
                t.update(qproj_w=self.q_projection.linear.weight, qproj_b=self.q_projection.linear.bias,
                         mo0_w=mo[0].weight, mo0_b=mo[0].bias, mo1_w=mo[1].weight, mo1_b=mo[1].bias,
                         mo2_w=mo[2].weight, mo2_b=mo[2].bias)
                if self.session_rep_size:
                    t.update(shared_w=self.shared_session_projector.linear.weight,
                             priv1_w=self.private_session_projector1.linear.weight)
            if not self.no_recommender:
                t.update(th_w=self.transform_hid.linear.weight, th_b=self.transform_hid.linear.bias,
                         tc_w=self.transform_cell.linear.weight, tc_b=self.transform_cell.linear.bias)
            pk = lib.Packed(lib.CarsSessionWeights, t, self._dims)
            # W_hh of the session LSTMs as fp16 term planes in the step kernel's lane order, once per weight version; a weight outside the
            # split's range leaves the fragment pointer NULL (fp32-MFMA steps)
            L = lib.load()
            HS_ = self._dims["HS"]
            nb = L.nir_lstm_step_whh_frag_bytes(HS_)
            for key in ("sq", "sd"):
                if nb and (key + "_whh") in pk.keep:
                    dev = pk.keep[key + "_whh"].device
                    frag, flag = torch.empty(nb, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
                    lib.check(L.nir_lstm_step_pack_whh_frag(lib.ptr(pk.keep[key + "_whh"]), HS_, lib.ptr(frag), lib.ptr(flag), lib.stream()),
                              "nir_lstm_step_pack_whh_frag")
                    if int(flag.item()) == 0:
                        pk.keep[key + "_whh_frag"] = frag
                        setattr(pk.struct, key + "_whh_frag", frag.data_ptr())
            b3 = int(all(float(pk.keep[k].abs().max()) < 32768.0 for k in ("sq_wih", "sd_wih") if k in pk.keep))
            pk.struct.rank_bounded = b3 << 3        # bit 3: the hoisted input projections of the session LSTMs may use the fp16 two-term split
            if not self.no_ranker:      # [W_q | W_shared + W_priv1] and the query-side attention projection, once per version
                L = lib.load()
                na, nb = lib.C.c_size_t(0), lib.C.c_size_t(0)
                L.nir_cars_session_pack_floats(pk.ref(), lib.C.byref(na), lib.C.byref(nb))
                dev = self.q_projection.linear.weight.device
                pk.keep["wrank"] = torch.empty(max(na.value, 1), device=dev)
                pk.keep["attn_ut"] = torch.empty(max(nb.value, 1), device=dev)
                lib.check(L.nir_cars_session_pack(pk.ref(), lib.ptr(pk.keep["wrank"]), lib.ptr(pk.keep["attn_ut"]), lib.stream()),
                          "nir_cars_session_pack")
                pk.struct.wrank = pk.keep["wrank"].data_ptr()
                pk.struct.attn_ut = pk.keep["attn_ut"].data_ptr()
                # rank features [q', d, |q' - d|, q' d]: d and the inputs of q' are pooled encoder / session states in (-1, 1), so
                # |q'| <= max_row(sum |W_rank| + |b|); with that and the layer's weights below 2^15 the first maxout GEMM takes the fp16
                # two-term split (one check per weight version)
                D_ = self._dims["D"]
                kr = na.value // D_ if D_ else 0
                qb = float((pk.keep["wrank"][:D_ * kr].view(D_, kr).abs().sum(1) + pk.keep["qproj_b"].abs()).max()) if kr else 0.0
                b0 = int(qb + 1.0 < 32768.0 and float(pk.keep["mo0_w"].abs().max()) < 32768.0)
                y0 = (qb + 1.0) * float(pk.keep["mo0_w"].abs().sum(1).max()) + float(pk.keep["mo0_b"].abs().max())     # bound of layer 0's outputs
                b1 = int(b0 and y0 < 32768.0 and float(pk.keep["mo1_w"].abs().max()) < 32768.0)
                b2 = int("click0_w" in pk.keep and float(pk.keep["click0_w"].abs().max()) < 32768.0)
                pk.struct.rank_bounded = b0 | (b1 << 1) | (b2 << 2) | (b3 << 3)
            return pk
        return self._ps.get([p for m in self._session_modules() for p in m.parameters()], build)

    def _decoder_weights(self):
        def build():
            rnn, att = self.decoder.decoder.rnn, self.decoder.decoder.attn
            t = dict(rnn_wih=rnn.weight_ih_l0, rnn_whh=rnn.weight_hh_l0, rnn_bih=rnn.bias_ih_l0, rnn_bhh=rnn.bias_hh_l0,
                     attn_in_w=att.linear_in.weight, attn_out_w=att.linear_out.weight, dec_attn_w=self.dec_attn.weight,
                     pred1_w=self.token_prob_predictor1.weight, pred2_w=self.token_prob_predictor2.weight)
            ints = dict(HD=rnn.hidden_size, DQ=self.dec_attn.weight.shape[1], P=self.token_prob_predictor1.weight.shape[0],
                        KS=self.session_rep_size, VT=self.token_prob_predictor2.weight.shape[0])
            pk = lib.Packed(lib.CarsDecoderWeights, t, ints)
            a = self.shared_session_projector.linear.weight.detach().float().contiguous()
            b = self.private_session_projector2.linear.weight.detach().float().contiguous()
            pk.keep["sess_w"] = torch.empty_like(a)
            lib.check(lib.load().nir_add_f32(lib.ptr(a), lib.ptr(b), lib.ptr(pk.keep["sess_w"]), a.numel(), lib.stream()), "nir_add_f32")
            pk.struct.sess_w = pk.keep["sess_w"].data_ptr()
            # token_prob_predictor2 as pre-split fp16 A-fragments for the fused projection + arg-max kernel (csrc/cars_decode.hip)
            w2 = pk.keep["pred2_w"]
            VT, P = w2.shape
            if self.fuse_decoder_argmax and P == 256 and w2.is_cuda and float(w2.abs().max()) < 32768.0:
                vp = (VT + 15) // 16 * 16
                pad = torch.zeros(vp, P, device=w2.device, dtype=torch.float32)
                pad[:VT] = w2
                planes = torch.stack(lib.split_f16x2(pad, P))                         # [2 terms, vp, P] int16
                pk.keep["pred2_frag"] = planes.view(2, vp // 16, 16, P // 32, 4, 8).permute(1, 3, 0, 4, 2, 5).contiguous()
                pk.struct.pred2_frag = pk.keep["pred2_frag"].data_ptr()
            if self.fold_decoder_query and w2.is_cuda:
                HDa, DQ = self.dec_attn.weight.shape
                a_t = pk.keep["attn_in_w"].t().contiguous()                            # [HD(k), HD(o)]
                d_t = pk.keep["dec_attn_w"].t().contiguous()                           # [DQ, HD(o)]
                wq = torch.empty(HDa, DQ, device=w2.device, dtype=torch.float32)       # wq[k, d] = sum_o W_in[o, k] W_dec_attn[o, d]
                lib.check(lib.load().nir_linear_f32(lib.ptr(a_t), HDa, None, None, 0, 0, 0, lib.ptr(d_t), HDa, None, None, lib.ptr(wq), DQ, HDa, DQ, HDa,
                                                    0, lib.stream()), "nir_linear_f32")
                pk.keep["attn_q_w"] = wq
                pk.struct.attn_q_w = wq.data_ptr()
            # the decoder LSTM's input is the previous token's embedding alone: its gate half is a per-token row, folded once per weight version;
            # W_hh as fp16 term fragments (a weight outside the split's range keeps the fp32-MFMA step)
            table = self.embedder.word_embeddings.table
            HD = int(rnn.hidden_size)
            nb = lib.load().nir_lstm_step_whh_frag_bytes(HD)
            if (self.fold_decoder_step and self.fold_embeddings and nb and table.is_cuda and table.shape[1] == rnn.input_size
                    and table.shape[0] * 4 * HD * 4 <= self.fold_budget_bytes):
                frag, flag = torch.empty(nb, dtype=torch.uint8, device=table.device), torch.zeros(1, dtype=torch.int32, device=table.device)
                lib.check(lib.load().nir_lstm_step_pack_whh_frag(lib.ptr(pk.keep["rnn_whh"]), HD, lib.ptr(frag), lib.ptr(flag), lib.stream()),
                          "nir_lstm_step_pack_whh_frag")
                if int(flag.item()) == 0:
                    pk.keep["rnn_whh_frag"] = frag
                    pk.keep["rnn_gate_fold"] = lib.fold_lstm_table(table, pk.keep["rnn_wih"], pk.keep["rnn_bih"], pk.keep["rnn_bhh"], HD, 1, "f32")
                    pk.struct.rnn_whh_frag = frag.data_ptr()
                    pk.struct.rnn_gate_fold = pk.keep["rnn_gate_fold"].data_ptr()
            return pk
        mods = [self.decoder, self.dec_attn, self.token_prob_predictor1, self.token_prob_predictor2,
                self.shared_session_projector, self.private_session_projector2]
        return self._pdec.get([p for m in mods for p in m.parameters()] + [self.embedder.word_embeddings.table, self.fold_decoder_step,
                                                                         self.fuse_decoder_argmax, self.fold_decoder_query], build)

    def _check_eval(self):
        if self.training:
            raise RuntimeError("CARS.encode / encode_document / rank_document / decode are the inference entry points: call them in "
                               "eval mode (the train-mode path is forward(), which runs the differentiable operators with dropout)")

    def _folded_table(self, which, w):
        """[V, 8H] folded gate table of one encoder (fp32 or bf16), rebuilt when the table or the LSTM weights change."""
        table = self.embedder.word_embeddings.table
        enc = (self.query_encoder if which == "q" else self.document_encoder).encoder
        cache = self._fq if which == "q" else self._fd
        dt = "bf16" if self.compute_dtype == "bf16" else "f32"          # ("f32_split2" reads the fp32 table)
        return cache.get([table] + list(enc.parameters()) + [dt],
                         lambda: lib.fold_lstm_table(table, w.keep["wih"], w.keep["bih"], w.keep["bhh"], w.struct.H, 2, dt))

    def _use_fold(self, table, H):
        if not self.fold_embeddings or self.training:
            return False
        per = table.shape[0] * 8 * H * (2 if self.compute_dtype == "bf16" else 4)
        return 2 * per <= self.fold_budget_bytes and H >= 8 and (2 * H) % 64 == 0

    def _encode_seqs(self, which, ids, lens, want_encoded):
        table = self.embedder.word_embeddings.table
        lib.require_device(ids, lens, table)
        L = lib.load()
        ids, lens = lib.ids64(ids), lib.ids64(lens)
        M, T = ids.shape
        w = self._enc_weights(which)
        H2 = 2 * w.struct.H
        dev = ids.device
        pooled = torch.empty(M, H2, device=dev, dtype=torch.float32)
        encoded = torch.empty(M, T, H2, device=dev, dtype=torch.float32) if want_encoded else None
        if self._use_fold(table, w.struct.H) and w.rec_ok:
            folded = self._folded_table(which, w)
            ws = lib.workspace(L.nir_cars_encode_folded_workspace_bytes(M, T, w.ref()), dev)
            lib.check(L.nir_cars_encode_folded(lib.ptr(ids), lib.ptr(lens), M, T, lib.ptr(folded), lib.DTYPES[self.compute_dtype],
                                               table.shape[0], w.ref(), lib.ptr(ws), ws.numel(), lib.ptr(pooled),
                                               lib.ptr(encoded), lib.ptr(self._flag_word(dev)), lib.stream()), "nir_cars_encode_folded")
            return pooled, encoded
        ids, _ = self._clean_ids(ids, None, table.shape[0])       # the per-batch gather-GEMM path has no in-kernel id check
        ws = lib.workspace(L.nir_cars_encode_workspace_bytes(M, T, table.shape[1], w.ref()), dev)
        lib.check(L.nir_cars_encode(lib.ptr(ids), lib.ptr(lens), M, T, lib.ptr(table), table.shape[0], table.shape[1],
                                    w.ref(), lib.ptr(ws), ws.numel(), lib.ptr(pooled), lib.ptr(encoded), lib.stream()),
                  "nir_cars_encode")
        return pooled, encoded

    # ---- reference API ----------------------------------------------------------------------------
    def encode(self, queries, query_length):
        """cars.py:193-225 -> (pooled [B,S,2H], encoded [B*S,QL,2H], hidden=None).
        (`hidden`, the final BiLSTM state in length-sorted order, has no consumer in the reference.)"""
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

    def session_query_side(self, pooled_q):
        """The two GEMMs of the session tail that read the pooled queries only (nir_cars_session_query_side: the session attention's keys U and the
        query chain's hoisted input projection gq), on the CURRENT stream -> (U, gq) for rank_document(.., query_side=...).  A caller that encodes
        the queries on a side stream next to the document encoder issues them there: off the one-batch critical path (cars.py:346-378)."""
        lib.require_device(pooled_q)
        L = lib.load()
        B, S, D = pooled_q.shape
        w = self._session_weights()
        nch = int(bool(w.struct.q_on)) + int(bool(w.struct.d_on))
        if nch == 0 or B == 0:
            return None
        HS, dev = self._dims["HS"], pooled_q.device
        pq = pooled_q.float().contiguous()
        U = torch.empty(B * S, nch * HS + nch, device=dev, dtype=torch.float32) if w.struct.rank_on else None
        gq = torch.empty(B * S, 4 * HS, device=dev, dtype=torch.float32) if w.struct.q_on else None
        lib.check(L.nir_cars_session_query_side(lib.ptr(pq), B, S, w.ref(), lib.ptr(U), lib.ptr(gq), lib.stream()), "nir_cars_session_query_side")
        return U, gq

    def _rank_session(self, pooled_q, pooled_docs, labels, want_clicks=False, want_states=False, rank_docs=None, labels_all=None,
                      labels_groups=None, click_max=None, query_side=None):
        """rank_docs [B,S,NR,D] (optional): the ranker scores only this slice of the candidates -> scores [B,S,NR]; clicks and sessions
        still see all of pooled_docs (candidate-sharded callers).
        labels_all [B_all,S,N] (optional): the inputs are a block of the sessions of a larger batch (session-sharded tail,
        sharding.SessionShardPlan); the click mask's batch-wide max click count (cars.py:285-289) is taken over labels_all.
        labels_groups [G,B_all,S,N] (optional, instead of labels_all): the B sessions are G equal blocks taken from G DIFFERENT batches (merged
        so that the session weights are streamed once for all of them); block g takes its count from labels_groups[g].
        click_max int32 [G] on the device (optional, instead of both): the counts themselves, one per block of B/G sessions -- a rank that holds
        a slice of a batch and received the batch's count with it (sharding.StreamShardPlan, mode "pair")."""
        lib.require_device(pooled_q, pooled_docs, labels)
        L = lib.load()
        B, S, D = pooled_q.shape
        N = pooled_docs.shape[2] if pooled_docs is not None else 1
        w = self._session_weights()
        dev = pooled_q.device
        ws = lib.workspace(L.nir_cars_session_workspace_bytes(B, S, N, w.ref()), dev)
        pq = pooled_q.float().contiguous()
        pdv = pooled_docs.float().contiguous() if pooled_docs is not None else None
        lab = labels.float().contiguous() if labels is not None else None
        lab_all = labels_all.float().contiguous() if labels_all is not None else None
        HS, HDEC = self._dims["HS"], self._dims["HDEC"]
        rd = rank_docs.float().contiguous() if rank_docs is not None else None
        NR = rd.shape[2] if rd is not None else N
        scores = torch.empty(B, S, NR, device=dev, dtype=torch.float32) if not self.no_ranker else None
        clicks = torch.empty(B, S, D, device=dev, dtype=torch.float32) if want_clicks else None
        extra, outs = None, {}
        if want_states:
            extra = lib.CarsSessionOutputs()
            if not self.no_query_session_encoding:
                outs["inner_q"] = torch.empty(B, S, HS, device=dev)
            if not self.no_document_session_encoding:
                outs["inner_d"] = torch.empty(B, S, HS, device=dev)
            outs["dec_h"] = torch.empty(1, (S - 1) * B, HDEC, device=dev)
            outs["dec_c"] = torch.empty(1, (S - 1) * B, HDEC, device=dev)
            for k, v in outs.items():
                setattr(extra, k, v.data_ptr())
        mg, spg = None, 0
        if labels_groups is not None:
            if lab_all is not None or B % labels_groups.shape[0]:
                raise RuntimeError("labels_groups excludes labels_all and needs B divisible by the number of groups")
            lg = labels_groups.float().contiguous()
            ng = lg.shape[0]
            mg, spg = torch.empty(ng, dtype=torch.int32, device=dev), B // ng
            lib.check(L.nir_cars_click_max(lib.ptr(lg), ng, lg.numel() // (ng * N), N, lib.ptr(mg), lib.stream()), "nir_cars_click_max")
        if click_max is not None:
            if lab_all is not None or mg is not None or click_max.dtype != torch.int32 or not click_max.is_cuda or B % click_max.numel():
                raise RuntimeError("click_max: int32 device tensor [G] with B % G == 0, instead of labels_all / labels_groups")
            mg, spg = click_max, B // click_max.numel()
        pre_u, pre_gq = query_side if query_side is not None else (None, None)
        lib.check(L.nir_cars_rank_session_pre(lib.ptr(pq), lib.ptr(pdv), lib.ptr(lab), B, S, N, w.ref(), lib.ptr(ws), ws.numel(),
                                              lib.ptr(scores), lib.ptr(clicks), lib.C.byref(extra) if extra is not None else None,
                                              lib.ptr(rd), NR if rd is not None else 0, lib.ptr(lab_all),
                                              lab_all.numel() // N if lab_all is not None else 0, lib.ptr(mg), spg, lib.ptr(pre_u), lib.ptr(pre_gq),
                                              lib.stream()),
                  "nir_cars_rank_session")
        return scores, clicks, outs

    def encode_clicks(self, docs, doc_labels):
        """cars.py:262-304 -> [B,S,2H] (computed by the same kernel family as rank_document)."""
        self._check_eval()
        if self.no_document_session_encoding:
            raise RuntimeError("encode_clicks needs the document session encoder (doc_session_off=False)")
        B, S, N, D = docs.shape
        dummy_q = torch.zeros(B, S, D, device=docs.device)
        return self._rank_session(dummy_q, docs, doc_labels, want_clicks=True)[1]

    def rank_document(self, pooled_rep, document_rep, document_len, document_label, group=None, shard=False, want_states=None,
                      labels_groups=None, after_documents=None, query_side=None, encoded_docs=None):
        """cars.py:522-540 -> (click_scores [B,S,N] (or [] when the ranker is off), hidden_states, session_attns).
        hidden_states = (transform_hid(h), transform_cell(c)) [1,(S-1)*B,nhid_decoder] and session_attns = (inner_q, inner_d)
        [B,S,HS] are the decoder inputs (cars.py:382-456); they are produced when `want_states` (default: whenever the
        recommender is on, like the reference), otherwise returned as None / (None, None).
        shard=True: candidate-sharded document encoding over the torch.distributed `group`, all-to-all of the pooled vectors, clicks /
        sessions / ranknet for this rank's block of sessions, all-gather of the scores (sharding.SessionShardPlan); when the decoder
        states are wanted: all-gather of the pooled vectors and a replicated session part (the states cover every session).
        labels_groups [G,B0,S,N] (optional): the B = G*B0 sessions are G whole batches merged into one macro-batch (Multitask.predict_many);
        batch g keeps the click count of its own labels.
        after_documents (optional callable): run between the document encoder and the session tail (a caller that produced `pooled_rep` on
        a side stream joins it here).  query_side (optional): session_query_side(pooled_rep) computed by that caller (unsharded path only).
        encoded_docs (optional, unsharded path): encode_document(document_rep, document_len) the caller already issued (it wanted the document
        encoder's first launch in front of its side branch)."""
        self._check_eval()
        if want_states is None:
            want_states = not self.no_recommender
        own = None
        if shard:
            encoded_docs = None
        if shard and after_documents is not None:
            after_documents()
            after_documents = None
        from .. import sharding
        if shard and sharding.dist.is_available() and sharding.dist.is_initialized() and not want_states and not self.no_ranker:
            # session-sharded tail (sharding.SessionShardPlan): candidate slice of every session -> all-to-all -> clicks / sessions / ranknet
            # for this rank's sessions only -> all-gather of the raw scores
            B, S, N = document_rep.shape[:3]
            plan = sharding.SessionShardPlan(B, S, N, sharding.dist.get_world_size(group), sharding.dist.get_rank(group), axis="auto")
            d, l = plan.doc_shard(document_rep, document_len)
            docs = plan.assemble(plan.exchange(self.encode_document(d, l), group))
            s_own = self._rank_session(plan.own(pooled_rep), docs, plan.own(document_label), labels_all=document_label)[0]
            return plan.gather(s_own, group).contiguous(), None, (None, None)
        if not (self.no_ranker and self.no_document_session_encoding):
            if shard:
                from .. import sharding
                encoded_docs, own = sharding.sharded_pooled_docs(self.encode_document, document_rep, document_len, group, return_local=True)
            elif encoded_docs is None:
                encoded_docs = self.encode_document(document_rep, document_len)
        if after_documents is not None:
            after_documents()
        # candidate-sharded: the ranker MLP scores this rank's slice only (clicks / sessions need every pooled candidate and stay
        # replicated); the score slices are gathered afterwards
        scores, _, outs = self._rank_session(pooled_rep, encoded_docs, document_label, want_states=want_states,
                                             rank_docs=own if not self.no_ranker else None, labels_groups=labels_groups, query_side=query_side)
        if own is not None and scores is not None:
            from .. import sharding
            scores = sharding.gather_session_scores(scores, document_rep.shape[2], group)
        states = (outs["dec_h"], outs["dec_c"]) if want_states else None
        attns = (outs.get("inner_q"), outs.get("inner_d")) if want_states else (None, None)
        return (scores if scores is not None else []), states, attns

    # ---- train-mode forward (cars.py:542-669 with dropout active) ----------------------------------------------------------
    # Differentiable: every lookup, Linear, BiLSTM, LSTM cell and the ranking loss run on the HIP operators of autograd.py; softmax /
    # masking / batched weighted sums (bmm) / concatenations / the label sort / max-pooling of the maxout / log-softmax of the
    # suggestion loss are tensor glue.
    @staticmethod
    def _mlp_logits(mlp, x, p):
        a = A.dropout(A.linear(x, mlp[0].weight, mlp[0].bias, act="tanh"), p, True)
        return A.linear(a, mlp[3].weight, mlp[3].bias).squeeze(-1)

    def _pool_train(self, mlp, enc, lens, p):
        T = enc.shape[1]
        mask = torch.arange(T, device=enc.device).unsqueeze(0) < lens.unsqueeze(1)
        return A.softmax_pool(self._mlp_logits(mlp, enc, p), mask, enc)

    def _encode_train(self, which, ids, lens):
        table = self.embedder.word_embeddings.table
        enc_mod = (self.query_encoder if which == "q" else self.document_encoder).encoder
        x = A.dropout(A.embed(ids, table), self.embedder.dropout.p, True)
        enc = A.dropout(A.bilstm(x, lens, enc_mod.rnns[0]), self.dropout.p, True)
        return self._pool_train(self.q_attn if which == "q" else self.d_attn, enc, lens, self.dropout.p), enc

    def _proj(self, seq, x):
        return A.linear(A.dropout(x, seq.dropout.p, True), seq.linear.weight, seq.linear.bias)

    def _forward_train(self, source_rep, source_len, target_rep, target_len, target_seq, document_rep, document_len, document_label):
        B, S, QL = source_rep.shape
        N, DL = document_rep.shape[2], document_rep.shape[3]
        p = self.dropout.p
        q_on, d_on = not self.no_query_session_encoding, not self.no_document_session_encoding
        src_lens = lib.ids64(source_len).reshape(-1)
        pooled_q, enc_q = self._encode_train("q", lib.ids64(source_rep).reshape(B * S, QL), src_lens)
        pooled_q = pooled_q.view(B, S, -1)
        docs = clicks = None
        if not (self.no_ranker and self.no_document_session_encoding):
            docs, _ = self._encode_train("d", lib.ids64(document_rep).reshape(B * S * N, DL), lib.ids64(document_len).reshape(-1))
            docs = docs.view(B, S, N, -1)
            if d_on:                                                   # encode_clicks (cars.py:262-304)
                lab = document_label.reshape(B * S, N)
                order = torch.sort(lab, dim=1, descending=True, stable=True)[1]
                sd = torch.gather(docs.reshape(B * S, N, -1), 1, order.unsqueeze(2).expand(-1, -1, docs.shape[-1]))
                count = (lab != 0).sum(1)
                pos = torch.arange(N, device=lab.device).unsqueeze(0)
                keep = (pos < count.unsqueeze(1)) | (pos >= count.max())          # the batch-dependent mask quirk (Appendix E2)
                clicks = A.softmax_pool(self._mlp_logits(self.click_attn, sd, p), keep, sd).view(B, S, -1)
        # ---- encode_session (cars.py:306-458), re-ordered: the two session LSTMs read pooled queries / click-pooled documents only, never a
        # ranking output, so their S steps run first (the only loop left); everything the reference computes per step from the states collected
        # so far -- the query-conditioned session attentions, projections, pair features, maxout ranker, inner attentions -- is then ONE batched
        # call over all steps with a causal mask (step t sees states 0..t), instead of S copies of every node in the autograd graph.
        dev = pooled_q.device
        sq_rnn = self.session_query_encoder.encoder.rnns[0] if q_on else None
        sd_rnn = self.session_doc_encoder.encoder.rnns[0] if d_on else None
        # the two session LSTMs: the input side of all S steps is one linear each, the S steps run inside the sequence buffers (A.lstm_gx)
        QH = QC = DH = DC = None                                        # [B,S,H] raw states (decoder initial states) and cell states
        if q_on:
            QH, QC = A.lstm_gx(A.linear(pooled_q, sq_rnn.weight_ih_l0, sq_rnn.bias_ih_l0), sq_rnn)
        if d_on:
            DH, DC = A.lstm_gx(A.linear(clicks, sd_rnn.weight_ih_l0, sd_rnn.bias_ih_l0), sd_rnn)
        QHd = A.dropout(QH, p, True) if q_on else None                  # the dropped copies both attentions read (one mask per state)
        DHd = A.dropout(DH, p, True) if d_on else None
        causal = torch.ones(S, S, dtype=torch.bool, device=dev).tril().unsqueeze(0)          # [1, step t, state j <= t]

        def attend(Hd_, lin):
            """step t attends over [0, h_0 .. h_{t-1}] with its pooled query (cars.py:520-560)"""
            st = torch.cat((torch.zeros(B, 1, Hd_.shape[2], device=dev), Hd_[:, :S - 1]), 1)          # [B,S,H]
            sc = (A.linear(st, lin.weight, lin.bias).unsqueeze(1) * pooled_q.unsqueeze(2)).sum(3)     # [B, t, j]
            return A.softmax_pool(sc, causal[0], st).view(B, S, -1)                                   # [B,S,H]: step t's rows share st[b]

        def inner(Hd_, mlp):
            """step t pools h_0 .. h_t with the inner attention (cars.py:562-600); its dropout draws a fresh mask at every step"""
            a = A.linear(Hd_, mlp[0].weight, mlp[0].bias, act="tanh")                                  # [B, j, H]
            if p > 0:
                a = A.dropout(a.unsqueeze(1).expand(B, S, S, a.shape[2]).contiguous(), p, True)
                lg = A.linear(a, mlp[3].weight, mlp[3].bias).squeeze(-1)                               # [B, t, j]
            else:
                lg = A.linear(a, mlp[3].weight, mlp[3].bias).squeeze(-1).unsqueeze(1).expand(B, S, S)
            return A.softmax_pool(lg, causal[0], Hd_).view(B, S, -1)                                   # [B,S,H]

        scores_all = None
        if not self.no_ranker:
            parts = ([attend(QHd, self.session_query_attn)] if q_on else []) + ([attend(DHd, self.session_doc_attn)] if d_on else [])
            D = docs.shape[-1]
            qp = self._proj(self.q_projection, pooled_q.reshape(B * S, -1))
            qx = qp.unsqueeze(1).expand(B * S, N, D).reshape(B * S * N, D)
            if parts:
                # the reference expands the session representation to [B,N,.] BEFORE the two projectors (cars.py:484-508): their
                # dropout draws one mask per candidate, not one per query
                sess = torch.cat(parts, 2)
                sx = sess.unsqueeze(2).expand(B, S, N, sess.shape[2]).reshape(B * S * N, sess.shape[2])
                qx = qx + self._proj(self.shared_session_projector, sx) + self._proj(self.private_session_projector1, sx)
            dx = docs.reshape(B * S * N, D)
            x = torch.cat((qx, dx, (qx - dx).abs(), qx * dx), 1)
            for layer, o, pool in zip(self.ranknet._linear_layers, self.ranknet._output_dims, self.ranknet._pool_sizes):
                x = A.linear(x, layer.weight, layer.bias).view(B * S * N, o, pool).max(-1)[0]
            scores_all = x.view(B, S, N)
        inner_q = inner(QHd, self.session_query_inner_attn) if q_on else None
        inner_d = inner(DHd, self.session_doc_inner_attn) if d_on else None
        out = {"ranking_loss": None, "suggestion_loss": None}
        if not self.no_ranker:
            click_scores = scores_all
            out["ranking_loss"] = A.bce_with_logits(click_scores, document_label.float())
            out["click_scores"] = click_scores
        if not self.no_recommender:                                    # teacher-forced decoder (cars.py:605-657)
            Bd = B * (S - 1)
            hid = torch.cat([h for h in (QH, DH) if h is not None], 2)            # [B,S,.]
            cell = torch.cat([c for c in (QC, DC) if c is not None], 2)
            dec_h = self._proj(self.transform_hid, hid[:, :-1].transpose(0, 1).reshape(Bd, -1))     # (step, session) row order, like the reference
            dec_c = self._proj(self.transform_cell, cell[:, :-1].transpose(0, 1).reshape(Bd, -1))
            cs = torch.cat([a for a in (inner_q, inner_d) if a is not None], 2)[:, :-1].reshape(Bd, -1)
            tgt = lib.ids64(target_rep).reshape(Bd, -1)
            tseq = lib.ids64(target_seq).reshape(Bd, -1)
            TL = tgt.shape[1]
            temb = A.dropout(A.embed(tgt, self.embedder.word_embeddings.table), self.embedder.dropout.p, True)
            mem = enc_q.view(B, S, QL, -1)[:, :-1].reshape(Bd, QL, -1)
            mem = A.linear(mem, self.dec_attn.weight)
            mlen = lib.ids64(source_len)[:, :-1].reshape(-1)
            rnn, att = self.decoder.decoder.rnn, self.decoder.decoder.attn
            h_all, _ = A.lstm_seq(temb, rnn, dec_h, dec_c)                            # [Bd,TL,HD]
            align = (A.linear(h_all, att.linear_in.weight).unsqueeze(2) * mem.unsqueeze(1)).sum(3)      # [Bd,TL,QL] (tiny: tensor glue, no library GEMM)
            mask = torch.arange(QL, device=dev).unsqueeze(0) < mlen.unsqueeze(1)                       # [Bd,QL]: the TL rows of a session share it
            ctx = A.softmax_pool(align, mask, mem, mask_div=align.shape[1]).view(Bd, align.shape[1], -1)
            dec_out = A.linear(torch.cat((ctx, h_all), 2), att.linear_out.weight, act="tanh")
            dec_out = A.dropout(dec_out, self.dec_dropout_p, True)[:, :-1]
            po = A.linear(dec_out, self.token_prob_predictor1.weight)
            sess = self._proj(self.shared_session_projector, cs) + self._proj(self.private_session_projector2, cs)
            po = A.dropout(po + sess.unsqueeze(1), p, True)
            out["suggestion_loss"] = A.suggestion_loss(A.linear(po, self.token_prob_predictor2.weight), tseq[:, 1:], PAD, self.regularize_coeff)
            _ = TL
        if not self.no_ranker and not self.no_recommender:
            if q_on and d_on:
                w1 = self.shared_session_projector.linear.weight.norm(2)
                w2 = self.private_session_projector1.linear.weight.norm(2) + self.private_session_projector2.linear.weight.norm(2)
                out["regularization"] = self.lambda1 * w1 + self.lambda2 * w2
            else:
                out["regularization"] = None
        return out

    def forward(self, source_rep, source_len, target_rep, target_len, target_seq, document_rep, document_len,
                document_label):
        """cars.py:542-669.  Eval mode: the ranking loss on the inference kernels.  Train mode: differentiable ranking +
        suggestion (+ regularisation) losses (_forward_train)."""
        if self.training:
            lib.require_device(source_rep, document_rep, self.embedder.word_embeddings.table)
            return self._forward_train(source_rep, source_len, target_rep, target_len, target_seq, document_rep, document_len, document_label)
        pooled, _, _ = self.encode(source_rep, source_len)
        scores, _, _ = self.rank_document(pooled, document_rep, document_len, document_label, want_states=False)
        out = {"ranking_loss": None, "suggestion_loss": None, "click_scores": scores if not self.no_ranker else None}
        if not self.no_ranker:
            lab = document_label.float().contiguous()
            loss = torch.empty(1, device=scores.device, dtype=torch.float32)
            rows = scores.shape[0] * scores.shape[1]
            lib.check(lib.load().nir_rank_loss_bce(lib.ptr(scores), lib.ptr(lab), rows, scores.shape[2], lib.ptr(loss),
                                                   lib.stream()), "nir_rank_loss_bce")
            out["ranking_loss"] = loss[0]
        return out

    def decode(self, states, max_len, src_dict, tgt_dict, batch_size, session_len, use_cuda, encoded_source, source_len,
               session_attns, tgt2src=None):
        """cars.py:706-791 (greedy): -> {'predictions': LongTensor [batch_size, session_len, max_len]} in target-vocabulary ids.
        `session_len` is the number of decoded queries per session (the caller passes S-1, models/multitask.py:286).
        The reference maps each predicted token back to a source-vocabulary id on the host (tgt_dict[idx] -> word ->
        src_dict[word]); here that mapping is ONE device lookup table `tgt2src` [V_tgt] (built from the two dictionaries on
        first use and cached; identity when no dictionaries are given)."""
        self._check_eval()
        if self.no_recommender:
            raise RuntimeError("decode needs the recommender (turn_recommender_off=False)")
        assert all(s is not None for s in states)
        L = lib.load()
        dec_h, dec_c = (s.reshape(-1, s.shape[-1]).float().contiguous() for s in states)
        B, SD = int(batch_size), int(session_len)
        Bd = B * SD
        if dec_h.shape[0] != Bd:
            raise RuntimeError("decode: %d initial states for %d x %d decode rows" % (dec_h.shape[0], B, SD))
        dev = dec_h.device
        enc = encoded_source.float().contiguous()
        rows_src, QL = enc.shape[0], enc.shape[1]
        lens = lib.ids64(source_len.reshape(-1))
        if rows_src != B * (SD + 1) or lens.numel() != rows_src:
            raise RuntimeError("decode: encoded_source must hold batch_size*(session_len+1) query rows")
        i = torch.arange(Bd, device=dev)
        rowmap = ((i // SD) * (SD + 1) + i % SD).contiguous()                       # the reference's [:, :-1] selection
        cat = [a for a in session_attns if a is not None]
        session_cat = torch.cat(cat, 2).reshape(rows_src, -1).float().contiguous() if cat else None
        if tgt2src is None:
            tgt2src = self._tgt2src(src_dict, tgt_dict, dev)
        w = self._decoder_weights()
        table = self.embedder.word_embeddings.table
        ws = lib.workspace(L.nir_cars_decode_workspace_bytes(rows_src, Bd, QL, w.ref()), dev)
        preds = torch.empty(Bd, int(max_len), dtype=torch.int64, device=dev)
        lib.check(L.nir_cars_decode_greedy(lib.ptr(dec_h), lib.ptr(dec_c), lib.ptr(enc), lib.ptr(lens), rows_src, QL, lib.ptr(rowmap), Bd,
                                           lib.ptr(session_cat), lib.ptr(table), table.shape[0], table.shape[1], lib.ptr(tgt2src), BOS,
                                           int(max_len), w.ref(), lib.ptr(ws), ws.numel(), lib.ptr(preds), lib.stream()),
                  "nir_cars_decode_greedy")
        return {"predictions": preds.view(B, SD, int(max_len))}

    def _tgt2src(self, src_dict, tgt_dict, dev):
        if src_dict is None or tgt_dict is None:
            return None
        key = (id(src_dict), id(tgt_dict), len(src_dict), len(tgt_dict), str(dev))
        if getattr(self, "_lut_key", None) != key:
            n = self.token_prob_predictor2.weight.shape[0]
            lut = [int(src_dict[tgt_dict[i]]) if i < len(tgt_dict) else 0 for i in range(n)]
            self._lut, self._lut_key = torch.tensor(lut, dtype=torch.int64, device=dev), key
        return self._lut
