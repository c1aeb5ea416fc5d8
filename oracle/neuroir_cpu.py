"""CPU oracle for the neuroir encode-and-rank hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-CPU (fp32) restatement of the reference algorithms on the hot path
(SURVEY.md section 8a), written as pure functions over a state dict `sd` whose keys are the
reference's own state-dict keys.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product package
(context_attentive_ir_amd) never does, and it has no CPU fallback.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which tests/golden/generate.py produced by running the real reference
(/root/reference, torch 2.10 CPU) on the same inputs and the same deterministic weights.

The dataflow deliberately keeps the reference's tensor materialisations (broadcast copies,
the host-side numpy histogram) so that timing this module is a fair "port" CPU baseline.
Each function cites the reference lines it restates.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

EMB = "make_embedding.emb_luts.0.weight"


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def embed(sd, prefix, ids):
    """neuroir/modules/embeddings.py:243-252 -- row gather; PAD row is zero in the table itself."""
    return F.embedding(ids, sd[prefix + "." + EMB])


def rnn_encode(sd, prefix, x, lengths, bidirectional=True, init=None):
    """neuroir/encoders/rnn_encoder.py:62-141 with nlayers=1, rnn_type='LSTM', use_last=True.

    sort by length (desc) -> pack -> nn.LSTM -> unpack -> unsort -> zero-pad back to x.size(1).
    Returns (final_state, memory_bank).  final_state stays in sorted order like the reference.
    """
    w_ih = sd[prefix + ".rnns.0.weight_ih_l0"]
    hid, inp = w_ih.shape[0] // 4, w_ih.shape[1]
    cache = sd.setdefault("__lstm_modules__", {})       # cache lives and dies with this state dict
    key = (prefix, bidirectional)
    lstm = cache.get(key)
    if lstm is None:
        lstm = torch.nn.LSTM(inp, hid, 1, batch_first=True, bidirectional=bidirectional).to(w_ih.dtype)   # (float64 state dicts: the
        names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
        if bidirectional:
            names += [n + "_reverse" for n in names]
        lstm.load_state_dict({n: sd[prefix + ".rnns.0." + n] for n in names})
        lstm.eval()
        cache[key] = lstm
    if w_ih.requires_grad:          # differentiable use (training-step tests): run the same nn.LSTM on the leaf tensors of `sd`
        names = [n for n, _ in lstm.named_parameters()]
        mod, params = lstm, {n: sd[prefix + ".rnns.0." + n] for n in names}
        lstm = lambda *a: torch.func.functional_call(mod, params, a)    # noqa: E731
    if lengths is None:
        out, fin = lstm(x, init) if init is not None else lstm(x)
        return fin, out
    slen, order = torch.sort(lengths, 0, True)
    packed = pack_padded_sequence(x[order], slen.tolist(), batch_first=True)
    out, fin = lstm(packed)
    out = pad_packed_sequence(out, batch_first=True)[0]
    out = out[torch.sort(order, 0)[1]]
    if out.size(1) < x.size(1):
        out = torch.cat([out, out.new_zeros(out.size(0), x.size(1) - out.size(1), out.size(2))], 1)
    return fin, out


def rnn_encoder_general(sd, prefix, x, lengths, rnn_type="LSTM", bidirectional=True, nlayers=1, use_last=True, use_bridge=False, init=None):
    """neuroir/encoders/rnn_encoder.py:62-185 in full: one single-layer nn.LSTM / nn.GRU per layer (`rnns.{i}`), sort -> pack -> rnn ->
    unpack per layer (:93-117), `use_last` selection or concatenation over layers (:119-132), bridge = Linear + ReLU on every state viewed
    as [-1, total_hidden_dim] (:159-185).  `sd` keys are `prefix + rnns.{i}.*` / `prefix + bridge.{j}.*` (prefix '' or ending in '.').
    Final states stay in length-sorted order like the reference; dropout between layers is the eval-mode identity."""
    ndir = 2 if bidirectional else 1
    mods = []
    for i in range(nlayers):
        w_ih = sd["%srnns.%d.weight_ih_l0" % (prefix, i)]
        G = 4 if rnn_type == "LSTM" else 3
        m = getattr(torch.nn, rnn_type)(w_ih.shape[1], w_ih.shape[0] // G, 1, batch_first=True, bidirectional=bidirectional).to(w_ih.dtype)
        m.load_state_dict({k[len("%srnns.%d." % (prefix, i)):]: v for k, v in sd.items() if isinstance(k, str) and k.startswith("%srnns.%d." % (prefix, i))})
        mods.append(m.eval())
    order = inv = slen = None
    cur = x
    if lengths is not None:
        slen, order = torch.sort(lengths, 0, True)
        inv = torch.sort(order, 0)[1]
        cur = pack_padded_sequence(x[order], slen.tolist(), batch_first=True)
    ist = []
    if init is not None:                                          # :76-91 (tuple states only: `if init_states:` on a tensor raises)
        hs, cs = init
        hs, cs = hs.split(nlayers, 0), cs.split(nlayers, 0)
        ist = [(hs[i], cs[i]) for i in range(nlayers)]
    bank, h_n, c_n = [], [], []
    for i in range(nlayers):
        if i != 0 and lengths is not None:
            cur = pack_padded_sequence(cur, slen.tolist(), batch_first=True)
        cur, st = mods[i](cur, ist[i]) if ist else mods[i](cur)
        if isinstance(st, tuple):
            h_n.append(st[0]); c_n.append(st[1])
        else:
            h_n.append(st)
        if lengths is not None:
            cur = pad_packed_sequence(cur, batch_first=True)[0]
        if not use_last or i == nlayers - 1:
            bank.append(cur[inv] if lengths is not None else cur)
    if use_last:
        mem = bank[-1]
        fin = (h_n[-1], c_n[-1]) if c_n else h_n[-1]
    else:
        mem = torch.cat(bank, 2)
        fin = (torch.cat(h_n, 0), torch.cat(c_n, 0)) if c_n else torch.cat(h_n, 0)
    if use_bridge:
        hid = mods[0].hidden_size
        tot = hid * (1 if use_last else nlayers)

        def bottle(j, states):
            return F.relu(F.linear(states.reshape(-1, tot), sd["%sbridge.%d.weight" % (prefix, j)], sd["%sbridge.%d.bias" % (prefix, j)])).view(states.shape)
        fin = tuple(bottle(j, t) for j, t in enumerate(fin)) if isinstance(fin, tuple) else bottle(0, fin)
    if mem.size(1) < x.size(1):
        mem = torch.cat([mem, mem.new_zeros(mem.size(0), x.size(1) - mem.size(1), mem.size(2))], 1)
    return fin, mem


# ------------------------------------------------------------------------------------------
# ESM  (neuroir/rankers/esm.py:19-45)
# ------------------------------------------------------------------------------------------
@torch.no_grad()
def esm_scores(sd, q, q_len, d, d_len):
    B, N, DL = d.shape
    qm = embed(sd, "word_embeddings", q).mean(1)                       # divides by padded QL
    dm = embed(sd, "word_embeddings", d.view(B * N, DL)).mean(1).view(B, N, -1)
    return F.cosine_similarity(qm.unsqueeze(1).expand_as(dm), dm, dim=2)


# ------------------------------------------------------------------------------------------
# MatchTensor  (neuroir/rankers/mtensor.py:62-131, 144-158)
# ------------------------------------------------------------------------------------------
@torch.no_grad()
def match_tensor_parts(sd, q, q_len, d, d_len):
    B, QL = q.shape
    N, DL = d.shape[1], d.shape[2]
    xq = _lin(sd, "linear_projection", embed(sd, "word_embeddings", q))
    xd = _lin(sd, "linear_projection", embed(sd, "word_embeddings", d.view(B * N, DL)))
    _, hq = rnn_encode(sd, "query_encoder", xq, q_len)
    _, hd = rnn_encode(sd, "document_encoder", xd, d_len.reshape(-1))
    pq = _lin(sd, "query_projection", hq)                              # [B,QL,C]
    pd = _lin(sd, "document_projection", hd)                           # [B*N,DL,C]; == bias at padded t
    return hq, hd, pq, pd


@torch.no_grad()
def match_tensor_general_scores(sd, q, q_len, d, d_len, rnn_type="LSTM", nlayers=1):
    """neuroir/rankers/mtensor.py:62-131 with any encoder configuration of its constructor (:36-49: rnn_type, nlayers; use_last = True)."""
    B, N, DL = d.shape
    eq = _lin(sd, "linear_projection", embed(sd, "word_embeddings", q))
    ed = _lin(sd, "linear_projection", embed(sd, "word_embeddings", d.reshape(B * N, DL)))
    _, hq = rnn_encoder_general(sd, "query_encoder.", eq, q_len, rnn_type, True, nlayers)
    _, hd = rnn_encoder_general(sd, "document_encoder.", ed, d_len.reshape(-1), rnn_type, True, nlayers)
    return _match_tensor_head(sd, q, d, _lin(sd, "query_projection", hq), _lin(sd, "document_projection", hd)), hq, hd


@torch.no_grad()
def match_tensor_scores(sd, q, q_len, d, d_len):
    hq, hd, pq, pd = match_tensor_parts(sd, q, q_len, d, d_len)
    return _match_tensor_head(sd, q, d, pq, pd)


def _match_tensor_head(sd, q, d, pq, pd):
    """mtensor.py:100-131: broadcast product, exact-match channel, three convolutions, 1x1 convolution, global max, output layer."""
    B, QL = q.shape
    N, DL = d.shape[1], d.shape[2]
    C = pq.shape[-1]
    # the reference materialises both broadcast operands (torch.stack) before multiplying
    pq_x = pq.unsqueeze(1).expand(B, N, QL, C).reshape(B * N, QL, 1, C).expand(-1, -1, DL, -1).contiguous()
    pd_x = pd.unsqueeze(1).expand(-1, QL, -1, -1).contiguous()
    prod = pq_x * pd_x
    qi = q.unsqueeze(1).expand(B, N, QL).reshape(B * N, QL, 1)
    exact = (qi == d.view(B * N, 1, DL)).float() * sd["exact_match_channel.alpha"]  # PAD==PAD counts
    t = torch.cat((prod, exact.unsqueeze(3)), 3).permute(0, 3, 1, 2)   # [B*N, C+1, QL, DL]
    feats = [F.conv2d(t, sd["conv%d.weight" % k], sd["conv%d.bias" % k], padding=(1, k))
             for k in (1, 2, 3)]
    g = F.conv2d(F.relu(torch.cat(feats, 1)), sd["conv.weight"], sd["conv.bias"])   # [B*N,20,QL,DL]
    pooled = g.flatten(2).max(2)[0]                                    # max over all (padded) positions
    return _lin(sd, "output", pooled).view(B, N)


def match_tensor_train_scores(sd, q, q_len, d, d_len, masks=None, p_drop=0.0):
    """Train-mode MatchTensor forward (mtensor.py:62-131 with emb_drop active), DIFFERENTIABLE (sd holds leaf tensors with
    requires_grad).  `masks` = (keep_q [B,QL,E], keep_d [B*N,DL,E]) replays externally drawn inverted-dropout masks, so that a
    product forward and this restatement see identical noise; None = no dropout."""
    B, QL = q.shape
    N, DL = d.shape[1], d.shape[2]
    eq = embed(sd, "word_embeddings", q)
    ed = embed(sd, "word_embeddings", d.view(B * N, DL))
    if masks is not None:
        eq = eq * masks[0].float() / (1.0 - p_drop)
        ed = ed * masks[1].float() / (1.0 - p_drop)
    xq, xd = _lin(sd, "linear_projection", eq), _lin(sd, "linear_projection", ed)
    _, hq = rnn_encode(sd, "query_encoder", xq, q_len)
    _, hd = rnn_encode(sd, "document_encoder", xd, d_len.reshape(-1))
    pq, pd = _lin(sd, "query_projection", hq), _lin(sd, "document_projection", hd)
    C = pq.shape[-1]
    prod = pq.unsqueeze(1).expand(B, N, QL, C).reshape(B * N, QL, 1, C) * pd.unsqueeze(1)
    qi = q.unsqueeze(1).expand(B, N, QL).reshape(B * N, QL, 1)
    exact = (qi == d.view(B * N, 1, DL)).float() * sd["exact_match_channel.alpha"]
    t = torch.cat((prod, exact.unsqueeze(3)), 3).permute(0, 3, 1, 2)
    feats = [F.conv2d(t, sd["conv%d.weight" % k], sd["conv%d.bias" % k], padding=(1, k)) for k in (1, 2, 3)]
    g = F.conv2d(F.relu(torch.cat(feats, 1)), sd["conv.weight"], sd["conv.bias"])
    return _lin(sd, "output", g.flatten(2).max(2)[0]).view(B, N)


# ------------------------------------------------------------------------------------------
# MNSRF, ranking side  (neuroir/multitask/mnsrf.py:62-114 encode, :116-162 rank_document)
# ------------------------------------------------------------------------------------------
def _strip_encoder_nesting(sd):
    out = {}
    for k, v in sd.items():
        if k.startswith("embedder."):
            out[k[len("embedder."):]] = v
        elif ".encoder.rnns." in k:
            out[k.replace(".encoder.", ".", 1)] = v
        else:
            out[k] = v
    return out


@torch.no_grad()
def mnsrf_encode(sd, source_rep, source_len):
    """-> (memory_bank [B,S,nhid_query], session_bank [B,S,nhid_session])"""
    m = _strip_encoder_nesting(sd)
    B, S, QL = source_rep.shape
    _, enc = rnn_encode(m, "query_encoder", embed(m, "word_embeddings", source_rep.reshape(B * S, QL)), source_len.reshape(-1))
    memory_bank = enc.max(1)[0].view(B, S, -1)                    # max over all (padded) positions, mnsrf.py:235-237
    # the reference steps the session LSTM one query at a time carrying (h, c): the same as one pass over the S queries
    _, session_bank = rnn_encode(m, "session_query_encoder", memory_bank, None, bidirectional=False)
    return memory_bank, session_bank


@torch.no_grad()
def mnsrf_scores(sd, source_rep, source_len, document_rep, document_len):
    """-> scores [B,S,N]"""
    m = _strip_encoder_nesting(sd)
    memory_bank, session_bank = mnsrf_encode(sd, source_rep, source_len)
    B, S, N, DL = document_rep.shape
    _, enc = rnn_encode(m, "document_encoder", embed(m, "word_embeddings", document_rep.reshape(B * S * N, DL)), document_len.reshape(-1))
    docs = enc.max(1)[0].view(B, S, N, -1)
    sess_in = torch.cat([torch.zeros_like(session_bank[:, :1]), session_bank[:, 1:]], 1)   # zeros at t = 0, s_t afterwards
    comb = torch.tanh(_lin(m, "projection.linear", torch.cat([memory_bank, sess_in], 2)))   # [B,S,nhid_document]
    return (comb.unsqueeze(2) * docs).sum(3)


# ------------------------------------------------------------------------------------------
# Suggestion side of MNSRF / M_MATCH_TENSOR: session-LSTM states that initialise the decoder, greedy decode without attention
# ------------------------------------------------------------------------------------------
@torch.no_grad()
def session_decoder_states(sd, prefix, pooled):
    """mmtensor.py:94-124 / mnsrf.py:88-112: the unidirectional session LSTM is stepped one query at a time carrying (h, c); the
    states after every query but the last are concatenated along the batch axis (step-major).  pooled [B,S,I]; sd keys
    `prefix.rnns.0.*` -> (session_bank [B,S,HS], h [1,(S-1)*B,HS], c [1,(S-1)*B,HS])."""
    B, S, _ = pooled.shape
    hidden, hs, cs, bank = None, [], [], []
    for qidx in range(S):
        hidden, rep = rnn_encode(sd, prefix, pooled[:, qidx:qidx + 1], None, bidirectional=False, init=hidden)
        bank.append(rep.squeeze(1)); hs.append(hidden[0]); cs.append(hidden[1])
    return torch.stack(bank, 1), torch.cat(hs[:-1], 1), torch.cat(cs[:-1], 1)


@torch.no_grad()
def plain_greedy_decode(sd, table, h, c, max_len, tgt2src=None, bos=2, dec="decoder.decoder.rnn", gen="generator"):
    """mmtensor.py:281-325 / mnsrf.py:251-296: embed(tgt) -> one decoder-LSTM step from the running state -> generator -> soft-max ->
    arg-max (first index on ties, torch.max) -> the prediction; the next input is its source-vocabulary id.  h, c [1,Bd,H] -> [Bd,max_len]."""
    W_ih, W_hh, b_ih, b_hh = (sd["%s.%s_l0" % (dec, n)] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
    h, c = h[0], c[0]
    tgt = torch.full((h.shape[0],), bos, dtype=torch.int64)
    preds = []
    for _ in range(max_len):
        g = F.linear(F.embedding(tgt, table), W_ih, b_ih) + F.linear(h, W_hh, b_hh)
        i, f, gg, o = g.chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        p = torch.softmax(_lin(sd, gen, h), 1).max(1)[1]
        preds.append(p)
        tgt = tgt2src[p] if tgt2src is not None else p
    return torch.stack(preds, 1)


# ------------------------------------------------------------------------------------------
# M_MATCH_TENSOR, ranking side  (neuroir/multitask/mmtensor.py:70-88 encode, :127-189 rank_document):
# MatchTensor over the B*S (session, query) rows; only the module nesting of the state dict differs.
# ------------------------------------------------------------------------------------------
def _mmt_as_match_tensor(sd):
    out = {}
    for k, v in sd.items():
        if k.startswith("embedder."):
            out[k[len("embedder."):]] = v
        elif k.startswith(("query_encoder.encoder.", "document_encoder.encoder.")):
            out[k.replace(".encoder.", ".", 1)] = v
        elif not k.startswith(("session_query_encoder.", "decoder.", "generator.")):
            out[k] = v
    return out


@torch.no_grad()
def m_match_tensor_encode(sd, source_rep, source_len):
    """-> projected_queries [B*S, QL, C]"""
    m = _mmt_as_match_tensor(sd)
    B, S, QL = source_rep.shape
    xq = _lin(m, "linear_projection", embed(m, "word_embeddings", source_rep.reshape(B * S, QL)))
    _, hq = rnn_encode(m, "query_encoder", xq, source_len.reshape(-1))
    return _lin(m, "query_projection", hq)


@torch.no_grad()
def m_match_tensor_scores(sd, source_rep, source_len, document_rep, document_len):
    """-> scores [B, S, N]"""
    B, S, N, DL = document_rep.shape
    s = match_tensor_scores(_mmt_as_match_tensor(sd), source_rep.reshape(B * S, -1), source_len.reshape(-1),
                            document_rep.reshape(B * S, N, DL), document_len.reshape(B * S, N))
    return s.view(B, S, N)


# ------------------------------------------------------------------------------------------
# DRMM  (neuroir/rankers/drmm.py:29-84, 95-98)
# ------------------------------------------------------------------------------------------
DRMM_BINS = [-1.0, -0.5, 0, 0.5, 1.0, 1.0]


@torch.no_grad()
def drmm_parts(sd, q, d):
    B, QL = q.shape
    N, DL = d.shape[1], d.shape[2]
    eq = embed(sd, "word_embeddings", q)
    gate = F.softmax(_lin(sd, "gating_network.weight", eq).squeeze(2), 1)           # over all QL slots
    ed = embed(sd, "word_embeddings", d.view(B * N, DL))
    eq_x = eq.unsqueeze(1).expand(B, N, QL, -1).reshape(B * N, QL, 1, -1).expand(-1, -1, DL, -1).contiguous()
    ed_x = ed.unsqueeze(1).expand(-1, QL, -1, -1).contiguous()
    cos = F.cosine_similarity(eq_x, ed_x, 3)                                        # [B*N,QL,DL]
    c = cos.numpy()
    hist = np.empty(c.shape[:2] + (5,), np.float32)
    for a in range(c.shape[0]):                                                     # host histogram, drmm.py:71-75
        for b in range(c.shape[1]):
            hist[a, b] = np.histogram(c[a, b], bins=DRMM_BINS)[0]
    return gate, cos, torch.from_numpy(hist)


def drmm_scores_from_hist(sd, gate, hist, B, N):
    z = _lin(sd, "ffnn.1", _lin(sd, "ffnn.0", hist)).squeeze(2).view(B, N, -1)
    s = (z * gate.unsqueeze(1)).sum(2, keepdim=True)
    return _lin(sd, "output", s).view(B, N)


@torch.no_grad()
def drmm_scores(sd, q, q_len, d, d_len):
    gate, _, hist = drmm_parts(sd, q, d)
    return drmm_scores_from_hist(sd, gate, hist, d.shape[0], d.shape[1])


# ------------------------------------------------------------------------------------------
# DUET  (neuroir/rankers/duet.py:28-59, 77-121, 148-208)
# ------------------------------------------------------------------------------------------
@torch.no_grad()
def duet_local(sd, q, d):
    B, QL = q.shape
    N, DL = d.shape[1], d.shape[2]
    m = (d.view(B, N, DL, 1) == q.view(B, 1, 1, QL)).float().view(B * N, DL, QL)    # PAD==PAD -> 1
    u = torch.tanh(F.conv1d(m, sd["local_model.conv1d.weight"], sd["local_model.conv1d.bias"]))
    u = torch.tanh(_lin(sd, "local_model.fc1", u)).squeeze(2)
    v = torch.tanh(_lin(sd, "local_model.fc2", u))
    return torch.tanh(_lin(sd, "local_model.fc3", v)).view(B, N)


@torch.no_grad()
def duet_distributed(sd, q, d, pool_size=5):
    B, QL = q.shape
    N, DL = d.shape[1], d.shape[2]
    p = "distributed_model."
    eq = embed(sd, "word_embeddings", q)
    ed = embed(sd, "word_embeddings", d.view(B * N, DL))
    cq = torch.tanh(F.conv1d(eq.transpose(1, 2), sd[p + "conv_q.weight"], sd[p + "conv_q.bias"]))
    cd = torch.tanh(F.conv1d(ed.transpose(1, 2), sd[p + "conv_d1.weight"], sd[p + "conv_d1.bias"]))
    qv = torch.tanh(_lin(sd, p + "fc1", cq.max(2)[0]))                              # [B,F]
    dd = torch.tanh(F.conv1d(F.max_pool1d(cd, pool_size, 1), sd[p + "conv_d2.weight"], sd[p + "conv_d2.bias"]))
    had = (qv.view(B, 1, -1, 1).expand(B, N, -1, dd.size(2)).reshape(B * N, -1, dd.size(2)).contiguous() * dd)
    m1 = torch.tanh(_lin(sd, p + "fc2", had)).squeeze(2)                            # Linear(DL-6 -> 1) over t
    m2 = torch.tanh(_lin(sd, p + "fc3", m1))
    return torch.tanh(_lin(sd, p + "fc4", m2)).view(B, N)


@torch.no_grad()
def duet_scores(sd, q, q_len, d, d_len):
    return duet_local(sd, q, d) + duet_distributed(sd, q, d)


# ------------------------------------------------------------------------------------------
# CARS ranking path  (neuroir/multitask/cars.py:193-540, 671-691; modules/maxout.py:70-84)
# ------------------------------------------------------------------------------------------
def _seq_mask(lengths, max_len):
    return torch.arange(max_len).unsqueeze(0) < lengths.unsqueeze(1)               # utils/misc.py:65-74


def _attn_pool(sd, p, h, mask):
    """cars.py:671-691 -- softmax(mask(w2.tanh(W1 h + b1) + b2)) weighted sum over time."""
    a = _lin(sd, p + ".3", torch.tanh(_lin(sd, p + ".0", h))).squeeze(2)
    a = a.masked_fill(~mask, float("-inf"))
    return torch.bmm(h.transpose(1, 2), F.softmax(a, 1).unsqueeze(2)).squeeze(2)


@torch.no_grad()
def cars_encode(sd, q, q_len):
    """cars.py:193-225 -> (pooled [B,S,256], encoded [B*S,QL,256])."""
    B, S, QL = q.shape
    lens = q_len.reshape(-1)
    _, h = rnn_encode(sd, "query_encoder.encoder", embed(sd, "embedder.word_embeddings", q.view(B * S, QL)), lens)
    return _attn_pool(sd, "q_attn", h, _seq_mask(lens, QL)).view(B, S, -1), h


@torch.no_grad()
def cars_encode_document(sd, d, d_len):
    """cars.py:227-260 -> pooled docs [B,S,N,256]."""
    B, S, N, DL = d.shape
    lens = d_len.reshape(-1)
    _, h = rnn_encode(sd, "document_encoder.encoder",
                      embed(sd, "embedder.word_embeddings", d.view(B * S * N, DL)), lens)
    return _attn_pool(sd, "d_attn", h, _seq_mask(lens, DL)).view(B, S, N, -1)


@torch.no_grad()
def cars_encode_clicks(sd, docs, labels, labels_all=None):
    """cars.py:262-304 incl. the batch-dependent mask quirk.
    labels_all (tests of the session-sharded multi-GPU tail only): docs / labels are a block of the sessions of a larger batch whose
    full label matrix is labels_all -- the reference takes m = max click count over the WHOLE batch (:285-289), so a block must too."""
    B, S, N, H = docs.shape
    order = labels.sort(dim=2, descending=True, stable=True)[1]
    sdocs = torch.gather(docs, 2, order.unsqueeze(3).expand(-1, -1, -1, H)).view(B * S, N, H)
    count = (labels.view(B * S, N) != 0).sum(1)
    m = int(count.max()) if labels_all is None else int((labels_all.reshape(-1, N) != 0).sum(1).max())
    keep = torch.ones(B * S, N, dtype=torch.bool)
    keep[:, :m] = torch.arange(m).unsqueeze(0) < count.unsqueeze(1)
    a = _lin(sd, "click_attn.3", torch.tanh(_lin(sd, "click_attn.0", sdocs))).squeeze(2)
    a = F.softmax(a.masked_fill(~keep, float("-inf")), 1)
    return torch.bmm(sdocs.transpose(1, 2), a.unsqueeze(2)).squeeze(2).view(B, S, H)


def _maxout(sd, p, x, dims=(256, 128, 1), pool=2):
    for i, o in enumerate(dims):                                                    # maxout.py:70-84
        x = _lin(sd, "%s._linear_layers.%d" % (p, i), x).view(*x.shape[:-1], o, pool).max(-1)[0]
    return x


def _cars_rank(sd, qv, sq, sdv, docs):
    """cars.py:460-520."""
    B, N, H = docs.shape
    sess = torch.cat((sq, sdv), 1)
    qp = _lin(sd, "q_projection.linear", qv) + _lin(sd, "shared_session_projector.linear", sess) \
        + _lin(sd, "private_session_projector1.linear", sess)
    qx = qp.unsqueeze(1).expand(B, N, H).reshape(B * N, H)
    dx = docs.reshape(B * N, H)
    feats = torch.cat((qx, dx, (qx - dx).abs(), qx * dx), 1)
    return _maxout(sd, "ranknet", feats).view(B, N)


def _lstm_step(sd, p, x, state):
    """single-step unidirectional LSTM (cars.py:378-380, 400-402), PyTorch gate order i,f,g,o."""
    h, c = state
    g = F.linear(x, sd[p + ".weight_ih_l0"], sd[p + ".bias_ih_l0"]) + F.linear(h, sd[p + ".weight_hh_l0"], sd[p + ".bias_hh_l0"])
    i, f, gg, o = g.chunk(4, 1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c), c


def _cars_rank_sw(sd, qv, sess_parts, docs):
    """cars.py:460-520 with the session switches: session_rep = cat of the encoders that are on (:485-500)."""
    B, N, H = docs.shape
    qp = _lin(sd, "q_projection.linear", qv)
    if sess_parts:
        sess = torch.cat(sess_parts, 1)
        qp = qp + _lin(sd, "shared_session_projector.linear", sess) + _lin(sd, "private_session_projector1.linear", sess)
    qx = qp.unsqueeze(1).expand(B, N, H).reshape(B * N, H)
    dx = docs.reshape(B * N, H)
    feats = torch.cat((qx, dx, (qx - dx).abs(), qx * dx), 1)
    return _maxout(sd, "ranknet", feats).view(B, N)


def _inner_pool(sd, p, states):
    """cars.py:385-388 / 407-410: self-attention pool over the states produced so far (without the zero state)."""
    st = torch.stack(states, 1)
    w = F.softmax(_lin(sd, p + ".3", torch.tanh(_lin(sd, p + ".0", st))).squeeze(2), 1)
    return torch.bmm(st.transpose(1, 2), w.unsqueeze(2)).squeeze(2)


@torch.no_grad()
def cars_session_full(sd, pooled_q, pooled_docs, clicks, q_on=True, d_on=True, rank_on=True, recommender=True):
    """cars.py:306-458 -> (click scores [B,S,N] or None, (dec_h, dec_c) or None, (inner_q, inner_d)).

    dec_h / dec_c = transform_hid / transform_cell of cat(h_q, h_d) for steps 0..S-2, concatenated along the BATCH axis in
    step-major order ([1,(S-1)*B,nhid_decoder], cars.py:431-445); inner_* [B,S,HS]."""
    B, S, _ = pooled_q.shape
    qs, ds = [], []
    qstate = dstate = None
    if q_on:
        HS = sd["session_query_attn.weight"].shape[1]
        qs.append(pooled_q.new_zeros(B, HS))
        qstate = (pooled_q.new_zeros(B, HS), pooled_q.new_zeros(B, HS))
    if d_on:
        HSd = sd["session_doc_attn.weight"].shape[1]
        ds.append(pooled_q.new_zeros(B, HSd))
        dstate = (pooled_q.new_zeros(B, HSd), pooled_q.new_zeros(B, HSd))
    scores, hid, cell, inner_q, inner_d = [], [], [], [], []
    for t in range(S):
        qv = pooled_q[:, t]

        def attend(states, p):
            st = torch.stack(states, 1)                                             # incl. the initial zero state
            w = F.softmax(torch.bmm(_lin(sd, p, st), qv.unsqueeze(2)).squeeze(2), 1)
            return torch.bmm(st.transpose(1, 2), w.unsqueeze(2)).squeeze(2)

        if rank_on:
            parts = []
            if q_on:
                parts.append(attend(qs, "session_query_attn"))
            if d_on:
                parts.append(attend(ds, "session_doc_attn"))                        # keyed by the QUERY vector (:361)
            scores.append(_cars_rank_sw(sd, qv, parts, pooled_docs[:, t]))
        h_parts, c_parts = [], []
        if q_on:
            qstate = _lstm_step(sd, "session_query_encoder.encoder.rnns.0", qv, qstate)
            qs.append(qstate[0]); h_parts.append(qstate[0]); c_parts.append(qstate[1])
            inner_q.append(_inner_pool(sd, "session_query_inner_attn", qs[1:]))
        if d_on:
            dstate = _lstm_step(sd, "session_doc_encoder.encoder.rnns.0", clicks[:, t], dstate)
            ds.append(dstate[0]); h_parts.append(dstate[0]); c_parts.append(dstate[1])
            inner_d.append(_inner_pool(sd, "session_doc_inner_attn", ds[1:]))
        if h_parts:
            hid.append(torch.cat(h_parts, 1)); cell.append(torch.cat(c_parts, 1))
    states = None
    if recommender and hid and S > 1:
        states = (_lin(sd, "transform_hid.linear", torch.cat(hid[:-1], 0)).unsqueeze(0),
                  _lin(sd, "transform_cell.linear", torch.cat(cell[:-1], 0)).unsqueeze(0))
    attns = (torch.stack(inner_q, 1) if inner_q else None, torch.stack(inner_d, 1) if inner_d else None)
    return (torch.stack(scores, 1) if scores else None), states, attns


@torch.no_grad()
def cars_encode_session(sd, pooled_q, pooled_docs, clicks):
    """cars.py:306-458, ranking outputs only -> click scores [B,S,N]."""
    return cars_session_full(sd, pooled_q, pooled_docs, clicks, recommender=False)[0]


@torch.no_grad()
def cars_rank_document(sd, pooled_q, d, d_len, labels):
    """cars.py:522-540 -> click_scores [B,S,N]."""
    docs = cars_encode_document(sd, d, d_len)
    return cars_encode_session(sd, pooled_q, docs, cars_encode_clicks(sd, docs, labels))


@torch.no_grad()
def cars_rank_document_full(sd, pooled_q, d, d_len, labels, q_on=True, d_on=True, rank_on=True, recommender=True):
    """cars.py:522-540 with the switches -> (click_scores, states, session_attns)."""
    docs = clicks = None
    if rank_on or d_on:
        docs = cars_encode_document(sd, d, d_len)
        if d_on:
            clicks = cars_encode_clicks(sd, docs, labels)
    return cars_session_full(sd, pooled_q, docs, clicks, q_on, d_on, rank_on, recommender)


@torch.no_grad()
def cars_scores(sd, q, q_len, d, d_len, labels):
    pooled, _ = cars_encode(sd, q, q_len)
    return cars_rank_document(sd, pooled, d, d_len, labels)


@torch.no_grad()
def cars_decode(sd, states, max_len, B, SD, encoded_source, source_len, session_attns, tgt2src=None, bos=2):
    """cars.py:706-791 (greedy) with RNNDecoder (decoders/decoder.py:94-168, rnn_decoder.py:19-88) and Luong 'general'
    attention (modules/global_attention.py:98-196) -> predictions [B, SD, max_len] (target-vocabulary ids).
    tgt2src: LongTensor [V_tgt] = src_dict[tgt_dict[i]] (the reference maps through the two dicts on the host, :783-787)."""
    h, c = states[0][0], states[1][0]
    QL, DQ = encoded_source.shape[1], encoded_source.shape[2]
    mem = encoded_source.view(B, SD + 1, QL, DQ)[:, :-1].reshape(B * SD, QL, DQ)
    mem = F.linear(mem, sd["dec_attn.weight"])
    mlen = source_len.view(B, SD + 1)[:, :-1].reshape(-1)
    mask = _seq_mask(mlen, QL)
    cat = [a for a in session_attns if a is not None]
    sess = None
    if cat:
        cs = torch.cat(cat, 2)[:, :-1].reshape(B * SD, -1)
        sess = _lin(sd, "shared_session_projector.linear", cs) + _lin(sd, "private_session_projector2.linear", cs)
    tgt = torch.full((B * SD,), bos, dtype=torch.long)
    table = sd["embedder.word_embeddings.make_embedding.emb_luts.0.weight"]
    preds = []
    for _ in range(max_len):
        h, c = _lstm_step(sd, "decoder.decoder.rnn", table[tgt], (h, c))
        qv = F.linear(h, sd["decoder.decoder.attn.linear_in.weight"])
        align = torch.bmm(mem, qv.unsqueeze(2)).squeeze(2).masked_fill(~mask, float("-inf"))
        ctx = torch.bmm(F.softmax(align, 1).unsqueeze(1), mem).squeeze(1)
        out = torch.tanh(F.linear(torch.cat((ctx, h), 1), sd["decoder.decoder.attn.linear_out.weight"]))
        out = F.linear(out, sd["token_prob_predictor1.weight"])
        if sess is not None:
            out = out + sess
        prob = F.softmax(F.linear(out, sd["token_prob_predictor2.weight"]), 1)
        w = prob.max(1)[1]
        preds.append(w)
        tgt = tgt2src[w] if tgt2src is not None else w
    return torch.stack(preds, 1).view(B, SD, max_len)


# ------------------------------------------------------------------------------------------
# losses / predict softmax / metrics
# ------------------------------------------------------------------------------------------
def bce_with_logits(scores, labels):
    """models/ranker.py:55-69, cars.py:603 -- mean over all B*N (B*S*N) entries."""
    return F.binary_cross_entropy_with_logits(scores, labels.float())


def softmax_nll(scores, labels):
    """models/ranker.py:79-89."""
    return -(F.log_softmax(scores, -1) * labels.float()).sum(1).mean()


def predict_softmax(scores):
    """models/ranker.py:258, models/multitask.py:279."""
    return F.softmax(scores, -1)


def mean_average_precision(pred, target):
    """eval/ltorank.py:4-26 (AP over all ranked candidates; needs >=1 relevant per row)."""
    tot = 0.0
    for row_p, row_t in zip(np.asarray(pred), np.asarray(target)):
        hits, ap = 0, 0.0
        for rank, idx in enumerate(row_p):
            if row_t[idx] == 1:
                hits += 1
                ap += hits / (rank + 1)
        tot += ap / hits
    return tot / len(pred)


def mean_reciprocal_rank(pred, target):
    """eval/ltorank.py:104-123."""
    tot = 0.0
    for row_p, row_t in zip(np.asarray(pred), np.asarray(target)):
        for rank, idx in enumerate(row_p):
            if row_t[idx] == 1:
                tot += 1.0 / (rank + 1)
                break
    return tot / len(pred)


def precision_at_k(pred, target, k):
    """eval/ltorank.py:29-47."""
    pred, target = np.asarray(pred), np.asarray(target)
    return float(np.mean([np.count_nonzero(t[p[:k]]) / k for p, t in zip(pred, target)]))


MODEL_FNS = {"ESM": esm_scores, "MATCH_TENSOR": match_tensor_scores, "DRMM": drmm_scores, "DUET": duet_scores}
