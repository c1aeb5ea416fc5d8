"""Flag / hyper-parameter surface of the hot-path models.

Mirror of the reference's config API for the five hot-path models:
  * add_model_args / get_model_args / update_model_args / override_model_args keep the call
    shapes of /root/reference/neuroir/config.py:33,98,115,123;
  * the per-model fixed hyper-parameters keep the values of /root/reference/neuroir/hyparam.py
    (ESM :3-8, DUET :34-46, DRMM :78-86, MATCH_TENSOR :88-105, CARS :197-225).
Everything is table-driven here; models outside the hot path are not listed (they keep
running on the reference's own stock-PyTorch classes).
"""
import argparse
import logging

logger = logging.getLogger(__name__)

_LSTM = dict(rnn_type="LSTM", bidirection=True, nlayers=1, dropout_rnn=0.2)

MODEL_ARCHITECTURE = {
    "ESM": dict(arch={}, data={}),
    "DUET": dict(arch=dict(nfilters=300, local_filter_size=1, dist_filter_size=3, pool_size=5),
                 data=dict(src_vocab_size=None, force_pad=True, fix_embeddings=True)),
    "DRMM": dict(arch=dict(nbins=5), data=dict(src_vocab_size=None, fix_embeddings=True)),
    "MATCH_TENSOR": dict(arch=dict(_LSTM, featsize=40, nhid_query=30, nhid_doc=140, nchannels=50,
                                   nfilters=6, match_filter_size=20),
                         data=dict(src_vocab_size=None, fix_embeddings=True)),
    "MNSRF": dict(arch=dict(_LSTM, nhid_query=512, nhid_document=512, nhid_session=1024, regularize_coeff=0.1, alpha=0.5),
                  data=dict(tgt_vocab_size=30000, fix_embeddings=True)),
    "M_MATCH_TENSOR": dict(arch=dict(_LSTM, featsize=40, nhid_query=30, nhid_document=140, nhid_session=300, nchannels=50,
                                     nfilters=6, match_filter_size=20, regularize_coeff=0.1, alpha=0.5),
                           data=dict(max_doc_len=100, max_query_len=10, tgt_vocab_size=30000, fix_embeddings=True)),
    "CARS": dict(arch=dict(_LSTM, nhid_query=256, nhid_document=256, nhid_click=512,
                           nhid_session_query=512, nhid_session_document=512, nhid_decoder=512,
                           query_session_off=False, doc_session_off=False, attn_type="general",
                           mlp_nhid=150, pool_type="attn", regularize_coeff=0.1, alpha=0.1,
                           lambda1=0.01, lambda2=0.0001, turn_ranker_off=False,
                           turn_recommender_off=False),
                 data=dict(tgt_vocab_size=30000, fix_embeddings=True)),
}

MODEL_OPTIONS = {"model_type", "emsize", "use_word", "use_char_ngram", "copy_attn", "resue_copy_attn", "force_copy"}
MODEL_OPTIMIZER = {"fix_embeddings", "optimizer", "learning_rate", "momentum", "weight_decay", "rnn_padding",
                   "dropout_rnn", "dropout", "dropout_emb", "cuda", "grad_clipping", "lr_decay"}
DATA_OPTIONS = {"max_doc_len", "max_query_len", "num_candidates", "force_pad"}

# (flag, type, default, group) -- same flags/defaults as the reference CLI (config.py:38-96)
_FLAGS = [
    ("max_doc_len", int, 200, "data"), ("max_query_len", int, 10, "data"), ("num_candidates", int, 10, "data"),
    ("use_word", "bool", True, "model"), ("use_char_ngram", int, 0, "model"), ("emsize", int, 300, "model"),
    ("rnn_type", str, "LSTM", "model"), ("bidirection", "bool", True, "model"), ("nlayers", int, 1, "model"),
    ("attn_type", str, "general", "model"), ("copy_attn", "bool", False, "model"),
    ("force_copy", "bool", False, "model"), ("reuse_copy_attn", "bool", False, "model"),
    ("dropout_emb", float, 0.2, "optim"), ("dropout_rnn", float, 0.2, "optim"), ("dropout", float, 0.2, "optim"),
    ("optimizer", str, "adam", "optim"), ("learning_rate", float, 0.001, "optim"), ("lr_decay", float, 0.95, "optim"),
    ("grad_clipping", float, 10, "optim"), ("early_stop", int, 5, "optim"), ("weight_decay", float, 0, "optim"),
    ("momentum", float, 0, "optim"), ("fix_embeddings", "bool", False, "optim"),
]


def str2bool(v):
    return v.lower() in ("yes", "true", "t", "1", "y")


def get_model_specific_params(model_name, field):
    return MODEL_ARCHITECTURE[model_name.upper()][field]


def add_model_args(parser):
    parser.register("type", "bool", str2bool)
    groups = {}
    for flag, typ, default, grp in _FLAGS:
        g = groups.setdefault(grp, parser.add_argument_group(grp))
        g.add_argument("--" + flag, type=typ, default=default)


def get_model_args(args):
    """Keep only model/optimizer/data args and overlay the model's fixed 'arch' dict."""
    keep = MODEL_OPTIONS | MODEL_OPTIMIZER | DATA_OPTIONS
    vals = {k: v for k, v in vars(args).items() if k in keep}
    vals.update(get_model_specific_params(args.model_type, "arch"))
    return argparse.Namespace(**vals)


def update_model_args(args):
    vals = dict(vars(args))
    vals.update(get_model_specific_params(args.model_type, "data"))
    return argparse.Namespace(**vals)


def override_model_args(old_args, new_args):
    """Saved architecture args win; optimizer args come from the new run."""
    old, new = vars(old_args), vars(new_args)
    for k in list(old):
        if k in new and old[k] != new[k] and k in MODEL_OPTIMIZER:
            logger.info("Overriding saved %s: %s --> %s", k, old[k], new[k])
            old[k] = new[k]
    return argparse.Namespace(**old)


def default_args(model_type, **overrides):
    """Convenience: a fully populated Namespace for `model_type` (CLI defaults + data + arch dicts)."""
    vals = {flag: default for flag, _, default, _ in _FLAGS}
    vals["model_type"] = model_type
    vals.update(get_model_specific_params(model_type, "data"))
    vals.update(get_model_specific_params(model_type, "arch"))
    vals.update(overrides)
    return argparse.Namespace(**vals)
