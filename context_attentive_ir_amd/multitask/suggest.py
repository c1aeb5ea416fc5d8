"""Suggestion side shared by M_MATCH_TENSOR and MNSRF (the reference's two session models whose decoder has no attention):

* session_states: the unidirectional session LSTM over the S pooled queries of every session -> the session bank AND the decoder's
  initial states, i.e. the LSTM state after every query but the last, concatenated along the batch axis in STEP-major order
  (/root/reference/neuroir/multitask/mmtensor.py:94-124, mnsrf.py:88-112);
* greedy_decode: embedding -> Decoder(attn_type='none') LSTM step -> generator -> arg-max -> target id mapped to its source id and
  fed back (mmtensor.py:281-325, mnsrf.py:251-296) -- nir_decode_greedy_plain, no host synchronisation.
"""
import torch

from .. import autograd as A
from .. import lib
from ..constants import BOS, PAD


def session_states(x, lstm):
    """x [B,S,I] pooled queries, lstm = the session encoder's single-layer unidirectional nn.LSTM (parameter container)
    -> (session_bank [B,S,HS], (h [1,(S-1)*B,HS], c [1,(S-1)*B,HS]))."""
    L, st = lib.load(), lib.stream()
    B, S, I = x.shape
    HS = lstm.hidden_size
    dev = x.device
    wih, whh = lstm.weight_ih_l0.detach().float().contiguous(), lstm.weight_hh_l0.detach().float().contiguous().view(1, 4 * HS, HS)
    bih, bhh = lstm.bias_ih_l0.detach().float().contiguous(), lstm.bias_hh_l0.detach().float().contiguous()
    xf = x.float().contiguous()
    gates = torch.empty(B * S, 4 * HS, device=dev, dtype=torch.float32)
    lib.check(L.nir_linear_f32(lib.ptr(xf), I, None, None, 0, 0, 0, lib.ptr(wih), I, lib.ptr(bih), lib.ptr(bhh), lib.ptr(gates), 4 * HS,
                               B * S, 4 * HS, I, 0, st), "nir_linear_f32")
    bank = torch.empty(B, S, HS, device=dev, dtype=torch.float32)
    cst = torch.empty(B, S, HS, device=dev, dtype=torch.float32)
    ws = lib.workspace(L.nir_bilstm_steps_workspace_bytes(B, HS), dev)
    lib.check(L.nir_birnn_steps_fwd(0, lib.ptr(gates), None, lib.ptr(whh), None, None, None, lib.ptr(bank), lib.ptr(cst), None, None, B, S, HS, 1,
                                    lib.ptr(ws), ws.numel(), st), "nir_birnn_steps_fwd")
    # states after queries 0 .. S-2, step-major along the batch axis (torch.cat(states[:-1], dim=1) of the reference)
    h = bank[:, :S - 1].transpose(0, 1).reshape(1, (S - 1) * B, HS).contiguous()
    c = cst[:, :S - 1].transpose(0, 1).reshape(1, (S - 1) * B, HS).contiguous()
    return bank, (h, c)


def tgt2src_lut(owner, src_dict, tgt_dict, n, dev):
    """[V_tgt] device lookup table target id -> source id (the reference maps every predicted token through tgt_dict[idx] -> word ->
    src_dict[word] on the host); cached on `owner`; None (identity) without dictionaries."""
    if src_dict is None or tgt_dict is None:
        return None
    key = (id(src_dict), id(tgt_dict), len(src_dict), len(tgt_dict), str(dev))
    if getattr(owner, "_lut_key", None) != key:
        lut = [int(src_dict[tgt_dict[i]]) if i < len(tgt_dict) else 0 for i in range(n)]
        owner._lut, owner._lut_key = torch.tensor(lut, dtype=torch.int64, device=dev), key
    return owner._lut


def greedy_decode(owner, states, max_len, src_dict, tgt_dict, batch_size, session_len, table, dec_rnn, generator, tgt2src=None):
    """-> {'predictions': LongTensor [batch_size, session_len, max_len]} in target-vocabulary ids; `session_len` = decoded queries per
    session (the caller passes S-1, models/multitask.py:286)."""
    L = lib.load()
    dec_h, dec_c = (s.reshape(-1, s.shape[-1]).float().contiguous() for s in states)
    Bd, H = dec_h.shape
    if Bd != int(batch_size) * int(session_len):
        raise RuntimeError("decode: %d initial states for %d x %d decode rows" % (Bd, batch_size, session_len))
    dev = dec_h.device
    gw, gb = generator.weight.detach().float().contiguous(), generator.bias.detach().float().contiguous()
    VT = gw.shape[0]
    if tgt2src is None:
        tgt2src = tgt2src_lut(owner, src_dict, tgt_dict, VT, dev)
    t = table.detach().float().contiguous()
    p = [getattr(dec_rnn, n).detach().float().contiguous() for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    # the decoder LSTM's input is the previous token's embedding alone: gate rows folded per token + W_hh as fp16 term fragments, once per weight
    # version (a weight outside the split's range, H % 32 != 0 or `fold_decoder_step = False` on the model keep the fp32 step)
    def build():
        nb = L.nir_lstm_step_whh_frag_bytes(H)
        if not (getattr(owner, "fold_decoder_step", True) and nb and t.is_cuda and t.shape[0] * 4 * H * 4 <= getattr(owner, "fold_budget_bytes", 64 << 30)):
            return None
        frag, flag = torch.empty(nb, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        lib.check(L.nir_lstm_step_pack_whh_frag(lib.ptr(p[1]), H, lib.ptr(frag), lib.ptr(flag), lib.stream()), "nir_lstm_step_pack_whh_frag")
        if int(flag.item()) != 0:
            return None
        return lib.fold_lstm_table(t, p[0], p[2], p[3], H, 1, "f32"), frag
    cache = getattr(owner, "_pdec_plain", None)
    if cache is None:
        cache = owner._pdec_plain = lib.PackCache(retain=1)
    fold = cache.get([table] + [getattr(dec_rnn, n) for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
                     + [getattr(owner, "fold_decoder_step", True)], build)
    ws = lib.workspace(L.nir_decode_greedy_plain_workspace_bytes(Bd, H, VT), dev)
    preds = torch.empty(Bd, int(max_len), dtype=torch.int64, device=dev)
    lib.check(L.nir_decode_greedy_plain_folded(lib.ptr(dec_h), lib.ptr(dec_c), Bd, H, lib.ptr(t), t.shape[0], t.shape[1], lib.ptr(p[0]), lib.ptr(p[1]),
                                               lib.ptr(p[2]), lib.ptr(p[3]), lib.ptr(gw), lib.ptr(gb), VT, lib.ptr(tgt2src), BOS, int(max_len),
                                               lib.ptr(fold[0]) if fold else None, lib.ptr(fold[1]) if fold else None, lib.ptr(ws),
                                               ws.numel(), lib.ptr(preds), lib.stream()), "nir_decode_greedy_plain_folded")
    return {"predictions": preds.view(int(batch_size), int(session_len), int(max_len))}


def bilstm_train(x, lens, lstm):
    """Train-mode BiLSTM memory bank [M,T,2H] for any hidden size (autograd.bilstm holds the wide-H form since round 4)."""
    return A.bilstm(x, lens, lstm)


def suggestion_loss(model, h_steps, c_steps, target_rep, target_seq):
    """Teacher-forced decoding loss shared by M_MATCH_TENSOR and MNSRF (mmtensor.py:224-254, mnsrf.py:198-228): the decoder without
    attention starts from the session state after queries 0 .. S-2 (rows step-major: torch.cat(states[:-1], 1), paired with the
    batch-major target rows exactly as the reference pairs them), generator, masked NLL summed over time and averaged over rows, plus the
    entropy regulariser.  h_steps / c_steps [B,S,HS] differentiable session states."""
    B, S, HS = h_steps.shape
    Bd = B * (S - 1)
    dec_h = h_steps[:, :S - 1].transpose(0, 1).reshape(Bd, HS)
    dec_c = c_steps[:, :S - 1].transpose(0, 1).reshape(Bd, HS)
    tgt = lib.ids64(target_rep.reshape(Bd, -1))
    seq = lib.ids64(target_seq.reshape(Bd, -1))
    emb = A.dropout(A.embed(tgt, model.embedder.word_embeddings.table), model.embedder.dropout.p, True)
    h_all, _ = A.lstm_seq(emb, model.decoder.decoder.rnn, dec_h, dec_c)                  # [Bd,TL,HS]
    h_all = A.dropout(h_all, model.dec_dropout_p, True)                                   # RNNDecoder's own dropout (dropout_rnn)
    # (the last step's logits feed nothing: sliced off in front of the generator)
    logits = A.linear(h_all[:, :-1], model.generator.weight, model.generator.bias)
    return A.suggestion_loss(logits, seq[:, 1:], PAD, model.regularize_coeff)
