"""ctypes binding of libneuroir_hip.so (the C-ABI declared in include/neuroir_hip.h).

There is NO CPU fallback: if the shared library is missing or a tensor is not on a ROCm device the
call raises RuntimeError (mirrors how the reference surfaces torch errors).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libneuroir_hip.so")

c_fp = C.c_void_p      # const float*   (device)
c_ip = C.c_void_p      # const int64_t* (device)
c_st = C.c_void_p      # hipStream_t


def _struct(name, float_fields, int_fields=()):
    fields = [(f, c_fp) for f in float_fields] + [(f, C.c_int) for f in int_fields]
    return type(name, (C.Structure,), {"_fields_": fields})


DrmmWeights = _struct("nir_drmm_weights",
                      ["gate_w", "gate_b", "ffnn0_w", "ffnn0_b", "ffnn1_w", "ffnn1_b", "out_w", "out_b"], ["snap_one"])
DrmmWeights = type("nir_drmm_weights", (C.Structure,), {"_fields_": list(DrmmWeights._fields_) + [("self_bin", C.c_void_p)]})
MatchTensorWeights = _struct(
    "nir_matchtensor_weights",
    ["proj_w", "proj_b", "q_wih", "q_whh", "q_bih", "q_bhh", "d_wih", "d_whh", "d_bih", "d_bhh",
     "qproj_w", "qproj_b", "dproj_w", "dproj_b", "alpha", "conv1_w", "conv1_b", "conv2_w", "conv2_b",
     "conv3_w", "conv3_b", "conv_w", "conv_b", "out_w", "out_b"],
    ["F", "Hq", "Hd", "C", "NF", "MF", "bounded"])
MatchTensorWeights = type("nir_matchtensor_weights", (C.Structure,), {"_fields_": list(MatchTensorWeights._fields_) + [("dproj_frag", C.c_void_p)]})
DuetWeights = _struct(
    "nir_duet_weights",
    ["l_conv_w", "l_conv_b", "l_fc1_w", "l_fc1_b", "l_fc2_w", "l_fc2_b", "l_fc3_w", "l_fc3_b",
     "convq_w", "convq_b", "convd1_w", "convd1_b", "convd2_w", "convd2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
     "fc3_w", "fc3_b", "fc4_w", "fc4_b"],
    ["NF", "pool", "bounded"])
DuetWeights = type("nir_duet_weights", (C.Structure,), {"_fields_": list(DuetWeights._fields_) + [
    (f, C.c_void_p) for f in ("table_h1", "table_h2", "convd1_h1", "convd1_h2", "convd2_h1", "convd2_h2")] + [("EP", C.c_int)] + [
    ("fw1", C.c_void_p), ("fw2", C.c_void_p), ("K1P", C.c_int), ("ftable", C.c_void_p), ("fw1c", C.c_void_p), ("EPT", C.c_int)]})
CarsEncoderWeights = _struct(
    "nir_cars_encoder_weights",
    ["wih", "whh", "bih", "bhh", "attn0_w", "attn0_b", "attn3_w", "attn3_b"], ["H", "bounded"])
CarsEncoderWeights = type("nir_cars_encoder_weights", (C.Structure,), {"_fields_": list(CarsEncoderWeights._fields_) + [("attn_frag", C.c_void_p),
                                                                                                   ("whh_frag", C.c_void_p)]})
CarsSessionWeights = _struct(
    "nir_cars_session_weights",
    ["click0_w", "click0_b", "click3_w", "click3_b", "sq_attn_w", "sq_attn_b", "sd_attn_w", "sd_attn_b",
     "sq_wih", "sq_whh", "sq_bih", "sq_bhh", "sd_wih", "sd_whh", "sd_bih", "sd_bhh", "qproj_w", "qproj_b",
     "shared_w", "priv1_w", "mo0_w", "mo0_b", "mo1_w", "mo1_b", "mo2_w", "mo2_b", "wrank", "attn_ut",
     "sq_inner0_w", "sq_inner0_b", "sq_inner3_w", "sq_inner3_b", "sd_inner0_w", "sd_inner0_b", "sd_inner3_w", "sd_inner3_b",
     "th_w", "th_b", "tc_w", "tc_b"],
    ["D", "HS", "HDEC", "q_on", "d_on", "rank_on", "rank_bounded"])
CarsSessionWeights = type("nir_cars_session_weights", (C.Structure,), {"_fields_": list(CarsSessionWeights._fields_) + [("sq_whh_frag", C.c_void_p),
                                                                                                   ("sd_whh_frag", C.c_void_p)]})
class CarsDecoderWeights(C.Structure):
    _fields_ = [(f, c_fp) for f in ("rnn_wih", "rnn_whh", "rnn_bih", "rnn_bhh", "attn_in_w", "attn_out_w", "dec_attn_w", "pred1_w",
                                    "pred2_w", "sess_w")] + [(f, C.c_int) for f in ("HD", "DQ", "P", "KS")] + [("VT", C.c_int64), ("pred2_frag", C.c_void_p), ("rnn_gate_fold", C.c_void_p),
                                                                                                           ("rnn_whh_frag", C.c_void_p), ("attn_q_w", C.c_void_p)]


CarsSessionOutputs = _struct("nir_cars_session_outputs", ["inner_q", "inner_d", "dec_h", "dec_c"])

MnsrfWeights = _struct(
    "nir_mnsrf_weights",
    ["q_wih", "q_whh", "q_bih", "q_bhh", "d_wih", "d_whh", "d_bih", "d_bhh", "s_wih", "s_whh", "s_bih", "s_bhh",
     "proj_w", "proj_b"], ["Hq", "Hd", "HS"])
MnsrfWeights = type("nir_mnsrf_weights", (C.Structure,), {"_fields_": list(MnsrfWeights._fields_) + [
    ("q_fold", C.c_void_p), ("d_fold", C.c_void_p), ("q_whh_frag", C.c_void_p), ("d_whh_frag", C.c_void_p), ("s_whh_frag", C.c_void_p), ("err", C.c_void_p)]})

_i, _l, _z = C.c_int, C.c_int64, C.c_size_t
# name -> (restype, argtypes); must list EVERY symbol include/neuroir_hip.h declares (tests check this)
SIGNATURES = {
    "nir_version": (_i, []),
    "nir_last_error_string": (C.c_char_p, []),
    "nir_debug_clock_probe": (_i, [C.c_void_p, _i, _i, C.c_void_p, c_st]),
    "nir_debug_set_buffer": (_i, [C.c_void_p]),
    "nir_set_stream_batches_in_flight": (_i, [c_st, _i]),
    "nir_debug_set_tunable": (_i, [C.c_char_p, _i]),
    "nir_profile_enable": (_i, [_i]),
    "nir_profile_report": (_i, [C.c_char_p, _z]),
    "nir_split_f16x2": (_i, [c_fp, _l, _i, _l, _i, C.c_void_p, C.c_void_p, c_st]),
    "nir_linear_planes_f32": (_i, [C.c_void_p, C.c_void_p, _l, c_ip, _l, _l, _i, _i, C.c_void_p, C.c_void_p, _l, c_fp, c_fp, _l, _l, _i, _i, _i, c_st]),
    "nir_sanitize_ids": (_i, [c_ip, _l, c_ip, _l, _l, c_ip, c_ip, C.c_void_p, c_st]),
    "nir_flag_publish": (_i, [C.c_void_p, C.c_void_p, c_st]),
    "nir_gather_fields": (_i, [C.c_void_p, _i, C.c_void_p, _l, c_st]),
    "nir_host_device_pointer": (_i, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "nir_softmax_rows_publish": (_i, [c_fp, c_fp, _l, _i, C.c_void_p, C.c_void_p, c_st]),
    "nir_host_rank_metric": (C.c_double, [_i, C.c_void_p, C.c_void_p, _i, _l, _i, _i]),
    "nir_widen_ids_i32": (_i, [C.c_void_p, C.c_void_p, _l, c_st]),
    "nir_linear_f32": (_i, [c_fp, _l, c_ip, c_fp, _i, _l, _l, c_fp, _l, c_fp, c_fp, c_fp, _l, _l, _i, _i, _i, c_st]),
    "nir_rowdot_f32": (_i, [c_fp, _l, c_fp, c_fp, c_fp, _l, _i, _i, c_st]),
    "nir_bilstm_fwd": (_i, [c_fp, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _l, _i, _i, _i, c_st]),
    "nir_bilstm_supported": (_i, [_i]),
    "nir_bilstm_fused_fwd": (_i, [c_fp, _i, c_fp, c_fp, c_fp, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _l, _i, _i, _i, c_st]),
    "nir_bilstm_steps_workspace_bytes": (_z, [_l, _i]),
    "nir_bilstm_steps_fwd": (_i, [c_fp, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _l, _i, _i, _i, C.c_void_p, _z, c_st]),
    "nir_birnn_steps_fwd": (_i, [_i, c_fp, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _l, _i, _i, _i, C.c_void_p, _z, c_st]),
    "nir_maxpool_time_f32": (_i, [c_fp, _l, _i, _i, c_fp, c_st]),
    "nir_lstm256_whh_frag_bytes": (_z, [_i]),
    "nir_lstm256_pack_whh_frag": (_i, [c_fp, _i, C.c_void_p, C.c_void_p, c_st]),
    "nir_lstm256_workspace_bytes": (_z, [_l, _i]),
    "nir_lstm256_rows_fwd": (_i, [c_fp, c_ip, c_ip, C.c_void_p, c_fp, _i, C.c_void_p, _l, _l, _i, _i, C.c_void_p, _z, c_st]),
    "nir_lstm256_train_fwd": (_i, [c_fp, c_ip, C.c_void_p, c_fp, c_fp, c_fp, C.c_void_p, _l, _i, _i, C.c_void_p, C.c_size_t, c_st]),
    "nir_lstm256_bptt_workspace_bytes": (_z, [_l, _i]),
    "nir_lstm256_bptt": (_i, [c_fp, c_fp, c_fp, c_ip, c_fp, c_fp, _l, _i, _i, C.c_void_p, C.c_size_t, c_st]),
    "nir_decode_greedy_plain_workspace_bytes": (_z, [_l, _i, _l]),
    "nir_decode_greedy_plain": (_i, [c_fp, c_fp, _l, _i, c_fp, _l, _i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _l, c_ip, _l, _i, C.c_void_p, _z, c_ip,
                                     c_st]),
    "nir_decode_greedy_plain_folded": (_i, [c_fp, c_fp, _l, _i, c_fp, _l, _i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _l, c_ip, _l, _i, c_fp, C.c_void_p,
                                            C.c_void_p, _z, c_ip, c_st]),
    "nir_mnsrf_workspace_bytes": (_z, [_l, _i, _i, _i, _i, C.POINTER(MnsrfWeights)]),
    "nir_mnsrf_encode": (_i, [c_ip, c_ip, _l, _i, _i, c_fp, _l, _i, C.POINTER(MnsrfWeights), C.c_void_p, _z, c_fp, c_fp, c_st]),
    "nir_mnsrf_encode_states": (_i, [c_ip, c_ip, _l, _i, _i, c_fp, _l, _i, C.POINTER(MnsrfWeights), C.c_void_p, _z, c_fp, c_fp, c_fp, c_fp, c_st]),
    "nir_mnsrf_rank": (_i, [c_fp, c_fp, c_ip, c_ip, _l, _i, _i, _i, c_fp, _l, _i, C.POINTER(MnsrfWeights), C.c_void_p, _z, c_fp, c_st]),
    "nir_mnsrf_score": (_i, [c_ip, c_ip, c_ip, c_ip, _l, _i, _i, _i, _i, c_fp, _l, _i, C.POINTER(MnsrfWeights), C.c_void_p, _z,
                            c_fp, c_st]),
    "nir_softmax_rows": (_i, [c_fp, c_fp, _l, _i, c_st]),
    "nir_softmax_gathered": (_i, [c_fp, c_fp, c_fp, _i, _l, _i, _i, c_st]),
    "nir_rank_loss_bce": (_i, [c_fp, c_fp, _l, _i, c_fp, c_st]),
    "nir_rank_loss_softmax_nll": (_i, [c_fp, c_fp, _l, _i, c_fp, c_st]),
    "nir_esm_score": (_i, [c_ip, c_ip, _i, _i, _i, _i, c_fp, _l, _i, c_fp, c_st]),
    "nir_drmm_score": (_i, [c_ip, c_ip, _i, _i, _i, _i, c_fp, _l, _i, C.POINTER(DrmmWeights), c_fp, c_fp, c_st]),
    "nir_matchtensor_workspace_bytes": (_z, [_i, _i, _i, _i, C.POINTER(MatchTensorWeights)]),
    "nir_matchtensor_score": (_i, [c_ip, c_ip, c_ip, c_ip, _i, _i, _i, _i, c_fp, _l, _i,
                                   C.POINTER(MatchTensorWeights), C.c_void_p, _z, c_fp, c_fp, c_fp, c_fp, c_fp, c_st]),
    "nir_matchtensor_score_encoded": (_i, [c_ip, c_ip, c_fp, c_fp, _i, _i, _i, _i, C.POINTER(MatchTensorWeights), C.c_void_p, _z, c_fp, c_fp, c_fp,
                                           c_st]),
    "nir_matchtensor_score_folded": (_i, [c_ip, c_ip, c_ip, c_ip, _i, _i, _i, _i, C.c_void_p, C.c_void_p, _i, _l,
                                          C.POINTER(MatchTensorWeights), C.c_void_p, _z, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_void_p, c_st]),
    "nir_duet_workspace_bytes": (_z, [_i, _i, _i, _i, _i, C.POINTER(DuetWeights)]),
    "nir_duet_score": (_i, [c_ip, c_ip, _i, _i, _i, _i, c_fp, _l, _i, C.POINTER(DuetWeights), C.c_void_p, _z,
                            c_fp, c_fp, c_fp, c_st]),
    "nir_cars_encode_workspace_bytes": (_z, [_l, _i, _i, C.POINTER(CarsEncoderWeights)]),
    "nir_cars_encode": (_i, [c_ip, c_ip, _l, _i, c_fp, _l, _i, C.POINTER(CarsEncoderWeights), C.c_void_p, _z,
                             c_fp, c_fp, c_st]),
    "nir_lstm_whh_frag_bytes": (_z, [_i, _i]),
    "nir_lstm_pack_whh_frag": (_i, [c_fp, _i, _i, C.c_void_p, C.c_void_p, c_st]),
    "nir_lstm_fold_table_bytes": (_z, [_l, _i, _i, _i]),
    "nir_lstm_fold_table_workspace_bytes": (_z, [_l, _i, _i, _i, _i]),
    "nir_lstm_fold_table": (_i, [c_fp, _l, _i, c_fp, c_fp, c_fp, _i, _i, C.c_void_p, _i, C.c_void_p, _z, c_st]),
    "nir_bilstm_folded_fwd": (_i, [C.c_void_p, _i, c_ip, c_ip, c_fp, c_fp, C.c_void_p, _l, _l, _i, _i, _i, c_st]),
    "nir_cars_encode_folded_workspace_bytes": (_z, [_l, _i, C.POINTER(CarsEncoderWeights)]),
    "nir_cars_encode_folded": (_i, [c_ip, c_ip, _l, _i, C.c_void_p, _i, _l, C.POINTER(CarsEncoderWeights), C.c_void_p, _z,
                                    c_fp, c_fp, C.c_void_p, c_st]),
    "nir_cars_session_workspace_bytes": (_z, [_i, _i, _i, C.POINTER(CarsSessionWeights)]),
    "nir_linear_wgrad_f32": (_i, [c_fp, _l, c_fp, _l, c_ip, c_fp, _i, c_fp, _l, _l, _i, _i, c_st]),
    "nir_colsum_f32": (_i, [c_fp, _l, _l, _i, c_fp, c_st]),
    "nir_linear_wgrad_set_f32": (_i, [c_fp, _l, c_fp, _l, c_ip, c_fp, _i, c_fp, _l, _l, _i, _i, c_st]),
    "nir_linear_wgrad_bias_f32": (_i, [c_fp, _l, c_fp, _l, c_ip, c_fp, _i, c_fp, _l, c_fp, _l, _i, _i, c_st]),
    "nir_linear_wgrad_group_f32": (_i, [_i] + [C.c_void_p] * 10 + [c_st]),
    "nir_linear_wgrad_bias_set_f32": (_i, [c_fp, _l, c_fp, _l, c_ip, c_fp, _i, c_fp, _l, c_fp, _l, _i, _i, c_st]),
    "nir_colsum_set_f32": (_i, [c_fp, _l, _l, _i, c_fp, c_st]),
    "nir_seq_rows": (_i, [c_ip, _l, _i, _i, C.c_void_p, C.c_void_p, c_st]),
    "nir_linear_wgrad_rows_set_f32": (_i, [c_fp, _l, _l, c_fp, _l, _l, C.c_void_p, C.c_void_p, _l, _i, _i, c_fp, _l, c_fp, _i, _i, c_st]),
    "nir_transpose_f32": (_i, [c_fp, _i, _i, c_fp, c_st]),
    "nir_transpose_group_f32": (_i, [_i, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_st]),
    "nir_lstm_train_fwd": (_i, [c_fp, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _l, _i, _i, _i, c_st]),
    "nir_lstm_perm_weights": (_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _i, _i, _i, c_fp, c_fp, c_st]),
    "nir_lstm_train_fwd_split": (_i, [c_fp, c_ip, c_ip, c_fp, c_fp, c_fp, c_fp, C.c_void_p, _l, _i, _i, _i, c_st]),
    "nir_lstm_train_bwd": (_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_ip, c_fp, c_fp, c_fp, c_fp, _l, _i, _i, _i, c_st]),
    "nir_lstm_cell_fwd": (_i, [c_fp, c_fp, c_fp, c_fp, c_fp, _l, _i, c_st]),
    "nir_lstm_cell_bwd": (_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _l, _i, c_st]),
    "nir_lstm_cell_seq_fwd": (_i, [c_fp, _l, c_fp, c_fp, c_fp, _l, c_fp, _l, c_fp, _l, c_fp, _l, _l, _i, c_st]),
    "nir_lstm_cell_seq_bwd": (_i, [c_fp, _l, c_fp, c_fp, _l, c_fp, c_fp, _l, c_fp, _l, c_fp, _l, c_fp, _l, c_fp, _l, _i, c_st]),
    "nir_lstm_cell_seq_bwd_masked": (_i, [c_fp, _l, c_fp, c_fp, c_fp, _l, c_fp, _l, c_fp, _l, c_fp, _l, c_fp, c_ip, _i, _i, _l, _i, c_st]),
    "nir_dropout_f32": (_i, [c_fp, c_fp, C.c_void_p, _l, C.c_float, C.c_uint64, c_st]),
    "nir_dropout_dev_f32": (_i, [c_fp, c_fp, C.c_void_p, _l, C.c_float, C.c_void_p, C.c_uint64, c_st]),
    "nir_mask_scale_f32": (_i, [c_fp, C.c_void_p, C.c_float, c_fp, _l, c_st]),
    "nir_act_bwd_f32": (_i, [c_fp, c_fp, c_fp, _l, _i, c_st]),
    "nir_im2col_rows_f32": (_i, [c_fp, _l, _i, _i, _i, _i, _i, _i, _i, c_fp, c_st]),
    "nir_mt_conv3_supported": (_i, [_i, _i, _i, _i]),
    "nir_mt_conv3_fwd": (_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _l, _i, _i, _i, _i, c_fp, c_st]),
    "nir_mt_conv3_wt_floats": (C.c_size_t, [_i, _i]),
    "nir_mt_conv3_partial_floats": (C.c_size_t, [_l, _i, _i]),
    "nir_mt_conv3_bwd": (_i, [c_fp, c_fp, c_fp, c_fp, c_fp, _l, _i, _i, _i, _i, c_fp, c_fp, c_fp, c_st]),
    "nir_col2im_rows_f32": (_i, [c_fp, _l, _i, _i, _i, _i, _i, _i, _i, c_fp, c_st]),
    "nir_rank_loss_bce_bwd": (_i, [c_fp, c_fp, c_fp, c_fp, _l, c_st]),
    "nir_softmax_nll_ent_fwd": (_i, [c_fp, _l, c_ip, _l, _l, _i, c_fp, c_fp, c_fp, C.c_void_p, c_st]),
    "nir_softmax_nll_ent_bwd": (_i, [c_fp, _l, c_ip, _l, c_fp, c_fp, c_fp, c_fp, _l, _i, c_fp, c_st]),
    "nir_softmax_pool_fwd": (_i, [c_fp, C.c_void_p, _l, _l, c_fp, _l, _i, _i, _i, c_fp, c_fp, c_st]),
    "nir_softmax_pool_bwd": (_i, [c_fp, c_fp, c_fp, _l, _i, _i, _i, c_fp, c_fp, c_st]),
    "nir_embed_f32": (_i, [c_ip, c_fp, _l, _i, _l, c_fp, C.c_void_p, c_st]),
    "nir_embed_bwd_f32": (_i, [c_ip, c_fp, _l, _i, _l, c_fp, _l, c_st]),
    "nir_add_f32": (_i, [c_fp, c_fp, c_fp, _l, c_st]),
    "nir_cars_decode_workspace_bytes": (_z, [_l, _l, _i, C.POINTER(CarsDecoderWeights)]),
    "nir_cars_decode_greedy": (_i, [c_fp, c_fp, c_fp, c_ip, _l, _i, c_ip, _l, c_fp, c_fp, _l, _i, c_ip, _l, _i,
                                    C.POINTER(CarsDecoderWeights), C.c_void_p, _z, c_ip, c_st]),
    "nir_cars_session_pack_floats": (_z, [C.POINTER(CarsSessionWeights), C.POINTER(_z), C.POINTER(_z)]),
    "nir_cars_session_pack": (_i, [C.POINTER(CarsSessionWeights), c_fp, c_fp, c_st]),
    "nir_cars_rank_session": (_i, [c_fp, c_fp, c_fp, _i, _i, _i, C.POINTER(CarsSessionWeights), C.c_void_p, _z,
                                   c_fp, c_fp, C.POINTER(CarsSessionOutputs), c_st]),
    "nir_cars_rank_session_shard": (_i, [c_fp, c_fp, c_fp, _i, _i, _i, C.POINTER(CarsSessionWeights), C.c_void_p, _z,
                                         c_fp, c_fp, C.POINTER(CarsSessionOutputs), c_fp, _i, c_st]),
    "nir_cars_rank_session_rows": (_i, [c_fp, c_fp, c_fp, _i, _i, _i, C.POINTER(CarsSessionWeights), C.c_void_p, _z,
                                        c_fp, c_fp, C.POINTER(CarsSessionOutputs), c_fp, _i, c_fp, _l, C.c_void_p, _i, c_st]),
    "nir_cars_session_query_side": (_i, [c_fp, _i, _i, C.POINTER(CarsSessionWeights), c_fp, c_fp, c_st]),
    "nir_cars_rank_session_pre": (_i, [c_fp, c_fp, c_fp, _i, _i, _i, C.POINTER(CarsSessionWeights), C.c_void_p, _z,
                                       c_fp, c_fp, C.POINTER(CarsSessionOutputs), c_fp, _i, c_fp, _l, C.c_void_p, _i, c_fp, c_fp, c_st]),
    "nir_cars_click_max": (_i, [c_fp, _i, _i, _i, C.c_void_p, c_st]),
    "nir_lstm_step_whh_frag_bytes": (_z, [_i]),
    "nir_lstm_step_pack_whh_frag": (_i, [c_fp, _i, C.c_void_p, C.c_void_p, c_st]),
}

DTYPE_F32, DTYPE_BF16, DTYPE_F32_SPLIT2 = 0, 1, 2
DTYPES = {"f32": DTYPE_F32, "fp32": DTYPE_F32, "bf16": DTYPE_BF16, "f32_split2": DTYPE_F32_SPLIT2}

_lib = None


def load():
    """dlopen the in-tree library (never builds implicitly; run `python -m context_attentive_ir_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libneuroir_hip.so not found at %s -- build it with `python -m context_attentive_ir_amd.build` "
                "(hipcc, gfx950). The HIP path has no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().nir_last_error_string().decode("utf-8", "replace")
        kind = "argument error" if rc < 0 else "hipError_t"
        raise RuntimeError("%s failed (%s %d): %s" % (what, kind, rc, msg))


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("context_attentive_ir_amd runs on a ROCm device only (got a %s tensor); "
                               "there is no CPU fallback -- move the model and inputs to cuda" % t.device)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def set_batches_in_flight(n, streams=None):
    """the library's scheduling hint (nir_set_stream_batches_in_flight) for `streams` (torch streams; default: the current one): n independent
    batches are kept in flight next to the calls enqueued there; n < 1 removes the entry (= 1)."""
    L = load()
    for s in ([torch.cuda.current_stream()] if streams is None else streams):
        L.nir_set_stream_batches_in_flight(C.c_void_p(s.cuda_stream), int(n))
        if int(n) < 1:
            _BIF.pop(s.cuda_stream, None)
        else:
            _BIF[s.cuda_stream] = int(n)


_BIF = {}


def batches_in_flight():
    """the hint of the current stream (1 without one): the host-side mirror of the library's per-stream table"""
    return _BIF.get(torch.cuda.current_stream().cuda_stream, 1) if _BIF else 1


def ids64(t):
    """int64 contiguous view of an id / length tensor (the reference feeds torch.LongTensor)."""
    if t.dtype != torch.int64:
        t = t.long()
    return t.contiguous()


class Packed(object):
    """A ctypes weight struct + the tensors that back its pointers (kept alive together)."""

    def __init__(self, struct_cls, tensors, ints=None):
        self.keep = {k: v.detach().to(torch.float32).contiguous() for k, v in tensors.items()}
        for k, v in self.keep.items():
            if v.data_ptr() % 16:
                self.keep[k] = v.clone()
        self.struct = struct_cls()
        for k, v in self.keep.items():
            setattr(self.struct, k, v.data_ptr())
        for k, v in (ints or {}).items():
            setattr(self.struct, k, int(v))

    def ref(self):
        return C.byref(self.struct)


# bumped whenever any PackCache rebuilds (= the only moment a superseded pack can be FREED): a hipGraph captured at epoch e references live packs
# for as long as the epoch is still e, whatever happened to the parameters meanwhile (graph_runner.PredictGraphCache's optimistic replay)
PACK_EPOCH = [0]


class PackCache(object):
    """Re-pack weights only when a parameter was modified (tensor._version) or moved.

    Keys may mix tensors and plain hashables (e.g. a dtype name).  `tensor._version` is bumped by every in-place op
    (optimizer steps, load_state_dict, `with torch.no_grad(): p.copy_()`), but NOT by writes through `p.data`; code that
    writes that way must call invalidate().  Superseded packs are kept alive (the last RETAIN of them): a hipGraph
    captured earlier still holds their device pointers, so freeing them on a repack would let the graph read freed memory.
    """
    RETAIN = 8

    def __init__(self, retain=None):
        """retain: superseded packs kept alive (default RETAIN); the folded [V, 8H] tables pass 1 -- alternating train and eval would
        otherwise pin eight dead multi-hundred-MB tables; a graph captured over an older table must be re-captured after training anyway."""
        self.key, self.val, self.retired = None, None, []
        if retain is not None:
            self.RETAIN = int(retain)

    def invalidate(self):
        self.key = None

    def get(self, params, builder):
        key = tuple((p.data_ptr(), p._version, str(p.device)) if torch.is_tensor(p) else p for p in params)
        if key != self.key:
            if self.val is not None:
                self.retired = (self.retired + [self.val])[-self.RETAIN:] if self.RETAIN > 0 else []
            self.val, self.key = builder(), key
            PACK_EPOCH[0] += 1
        return self.val


_WS = {}
_WS_RETIRED = []
_WS_OWNER = [None]


class workspace_owner(object):
    """Context manager: scratch requested inside the block belongs to `owner` (any object; buffers are stored on it as
    `_nir_ws`) instead of the shared per-(device, stream) pool.  graph_runner.GraphedPredictor captures under it, so the
    pointers baked into its hipGraph stay valid for the predictor's whole life no matter what other callers do."""

    def __init__(self, owner):
        self.owner = owner

    def __enter__(self):
        self.prev = _WS_OWNER[0]
        _WS_OWNER[0] = self.owner
        if not hasattr(self.owner, "_nir_ws"):
            self.owner._nir_ws = {}
        return self

    def __exit__(self, *exc):
        _WS_OWNER[0] = self.prev
        return False


def workspace(nbytes, device):
    """Scratch buffer per (device, stream) -- or per workspace_owner; the C library never allocates device memory.
    Buffers only grow (geometrically); a superseded buffer is retired, never freed, because a captured hipGraph may still
    hold its address."""
    owner = _WS_OWNER[0]
    pool = _WS if owner is None else owner._nir_ws
    k = (str(device), torch.cuda.current_stream().cuda_stream)
    buf = pool.get(k)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:      # (an owner's superseded buffers live as long as the owner -- its graphs -- does; the shared pool's for the process)
            (_WS_RETIRED if owner is None else owner.__dict__.setdefault("_nir_ws_retired", [])).append(buf)
        grow = 0 if buf is None else 2 * buf.numel()
        buf = torch.empty(max(int(nbytes), grow, 1 << 20), dtype=torch.uint8, device=device)
        pool[k] = buf
    return buf


def fold_lstm_table(table, wih, bih, bhh, H, ndir, dtype):
    """nir_lstm_fold_table: [V, ndir*4H] folded gate table (fp32, or bf16 stored as int16) -- weight packing, run once
    per parameter version (PackCache)."""
    L = load()
    V, E = table.shape
    dt = DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
    out = torch.empty(V, ndir * 4 * H, device=table.device, dtype=torch.float32 if dt == DTYPE_F32 else torch.int16)
    ws = torch.empty(max(1, L.nir_lstm_fold_table_workspace_bytes(V, E, H, ndir, dt)), dtype=torch.uint8, device=table.device)
    t = table.detach().float().contiguous()
    check(L.nir_lstm_fold_table(ptr(t), V, E, ptr(wih), ptr(bih), ptr(bhh), H, ndir, ptr(out), dt, ptr(ws), ws.numel(), stream()),
          "nir_lstm_fold_table")
    return out


# bumped by anything that changes which kernels an entry point launches without touching a weight (debug tunables): part of the key of
# graph_runner.PredictGraphCache, so a graph captured before the change is not replayed after it
GRAPH_EPOCH = [0]


class tunable(object):
    """Context manager: `with lib.tunable("lstm_mfma16", 1): ...` (nir_debug_set_tunable; restores `restore` on exit)."""

    def __init__(self, name, value, restore):
        self.name, self.value, self.restore = name.encode(), int(value), int(restore)

    def __enter__(self):
        check(load().nir_debug_set_tunable(self.name, self.value), "nir_debug_set_tunable")
        GRAPH_EPOCH[0] += 1
        return self

    def __exit__(self, *exc):
        load().nir_debug_set_tunable(self.name, self.restore)
        GRAPH_EPOCH[0] += 1
        return False


class Flags(object):
    """The error word of one device + its pinned, device-mapped host mirror (round 6).

    Every kernel that can detect an error ORs into `dev` (bit 0: token id outside the vocabulary; bit 1: recurrent weights outside the fp16
    range of a split recurrence; bit 2: a recurrence cluster timed out) and never synchronises.  `publish()` enqueues nir_flag_publish at the
    end of a predict() / update(): a non-zero word lands in `host` (pinned memory the device writes directly).  `poll()` reads the host word --
    no device round trip -- and raises what it finds: called at the entry of the next wrapper call and from the `.cpu()` of the returned
    scores, i.e. right after the caller's own synchronisation in the reference's drivers (main/ranker.py:255, main/multitask.py:284).
    `check()` is the blocking form (reads the device word)."""

    def __init__(self, device):
        self.device = device
        self.dev = torch.zeros(1, dtype=torch.int32, device=device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.host_np = self.host.numpy()
        self.mapped = True
        self.cluster_handlers = []                          # weak references to callables run when bit 2 (cluster time-out) is found

    def on_cluster_timeout(self, method):
        """register a bound method (kept weakly) that makes its owner safe after a recurrence cluster timed out (MNSRF: switch to the streaming
        recurrence) -- run BEFORE the RuntimeError of the failed call is raised, so the caller's retry already takes the safe path"""
        import weakref
        self.cluster_handlers.append(weakref.WeakMethod(method))

    def publish(self):
        """capturable; -> False when the host word cannot be written by the device (then only check() sees errors)"""
        if self.mapped:
            rc = load().nir_flag_publish(ptr(self.dev), C.c_void_p(self.host.data_ptr()), stream())
            if rc != 0:
                self.mapped = False
        return self.mapped

    def _raise(self, v):
        if v & 4:
            live = []
            for ref in self.cluster_handlers:
                fn = ref()
                if fn is not None:
                    fn()
                    live.append(ref)
            self.cluster_handlers = live
            raise RuntimeError("a recurrence cluster (csrc/lstm_cluster.hip) waited for a partner workgroup that never became resident; the results of "
                               "that call are invalid -- the models on this device have been switched to the streaming recurrence: re-issue the batch")
        if v & 2:
            raise RuntimeError("recurrent weights outside the fp16 range of the split-fp16 MFMA recurrence (|w_hh| >= 2^15); "
                               "results of that call are invalid -- the exact fp32 recurrence handles such weights")
        raise IndexError("index out of range in self (token id outside [0, src_vocab_size))")

    def take(self):
        """blocking read-and-clear of the device word -> its value"""
        v = int(self.dev.item())
        if v != 0:
            self.dev.zero_()
        self.host_np[0] = 0
        return v

    def poll(self):
        if self.host_np[0] != 0:
            torch.cuda.synchronize(self.device)         # error path only: later publishes of the same sticky word have landed
            self._raise(self.take())

    def check(self):
        v = self.take()
        if v != 0:
            self._raise(v)


_FLAGS = {}


def flags(device):
    """the Flags of `device` (a torch.device with an index, e.g. tensor.device)"""
    key = (device.type, device.index)
    f = _FLAGS.get(key)
    if f is None:
        f = _FLAGS[key] = Flags(device)
    return f


class IdCheck(object):
    """Mixin of the network mirrors: `q, d = self._clean_ids(q, d, V)` validates token ids on the device (nir_sanitize_ids, one
    launch, no sync) and `check_ids()` raises the IndexError the reference's nn.Embedding would have raised at the lookup.
    `validate_ids = False` skips the launch for callers that guarantee 0 <= id < V themselves.  The flag every kernel of a device
    writes is ONE word (lib.flags(device).dev): `_flag_word(device)` hands out its tensor."""
    validate_ids = True
    _flag_dev = None

    def _flag_word(self, device):
        f = flags(device)
        self._flag_dev = device
        return f.dev

    def _clean_ids(self, a, b, V):
        a = ids64(a)
        b = ids64(b) if b is not None else None
        if not self.validate_ids:
            return a, b
        flag = self._flag_word(a.device)
        oa = torch.empty_like(a)
        ob = torch.empty_like(b) if b is not None else None
        check(load().nir_sanitize_ids(ptr(a), a.numel(), ptr(b), b.numel() if b is not None else 0, int(V), ptr(oa), ptr(ob), ptr(flag),
                                      stream()), "nir_sanitize_ids")
        return oa, ob

    def check_ids(self):
        """Synchronising: raise IndexError / RuntimeError if any forward on this network's device since the last check set an error bit."""
        if self._flag_dev is not None:
            flags(self._flag_dev).check()


def split_f16x2(x, cols_pad=None):
    """nir_split_f16x2: fp32 [rows, cols] -> two fp16 term planes [rows, cols_pad] (stored as int16 tensors)."""
    x = x.detach().float().contiguous()
    rows, cols = x.shape
    cp = (cols + 7) // 8 * 8 if cols_pad is None else int(cols_pad)
    p1 = torch.empty(rows, cp, dtype=torch.int16, device=x.device)
    p2 = torch.empty_like(p1)
    check(load().nir_split_f16x2(ptr(x), rows, cols, cols, cp, ptr(p1), ptr(p2), stream()), "nir_split_f16x2")
    return p1, p2
