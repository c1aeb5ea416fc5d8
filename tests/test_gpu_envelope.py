"""GPU (-m gpu): the shape envelope of the drop-in classes -- the shapes the reference's own launch scripts and data produce
(scripts/ranker.sh:18-19 pads to max_query_len 20 x max_doc_len 200; README.md:81: queries up to 40 tokens; config.py:42:
--num_candidates is free) must run, and match the CPU oracle, not raise a capacity error."""
import numpy as np
import pytest
import torch

from helpers import build_model, cpu_state_dict
from oracle import neuroir_cpu as O
from test_gpu_parity import _close, _synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("QL,DL,B,N", [(20, 200, 2, 3), (22, 200, 1, 2), (42, 290, 1, 2), (40, 24, 1, 2), (64, 40, 1, 2), (17, 64, 2, 2),
                                       (33, 100, 1, 3), (16, 200, 2, 2), (4, 400, 1, 2)])
@pytest.mark.parametrize("exact", [0, 1])
def test_match_tensor_long_shapes(QL, DL, B, N, exact):
    """mt_head_kernel processes queries in passes of 16 positions and falls back to 32-wide document chunks when the document planes are
    long: LDS no longer grows with QL.  Both interaction forms (fp16 two-term MFMA, exact fp32 MFMA) against the oracle."""
    from context_attentive_ir_amd import lib
    V = 300
    m = build_model("MATCH_TENSOR", vocab=V, device=DEV)
    rng = np.random.default_rng(QL * 1000 + DL)
    q, ql, d, dl = _synth(rng, B, N, QL, DL, V)
    # plant exact matches at the pass boundaries (query positions 15/16/17, 31/32): the exact-match channel reads q[i-1..i+1]
    for i in (0, 15, 16, 17, 31, 32, QL - 1):
        if i < int(ql[0]):
            d[0, 0, (7 * i + 3) % DL] = q[0, i]
    with lib.tunable("exact_f32", exact, 0):
        s = m(q.to(DEV), ql.to(DEV), d.to(DEV), dl.to(DEV))
    _close(s, O.match_tensor_scores(cpu_state_dict(m), q, ql, d, dl))


def test_match_tensor_doc_too_long_is_an_argument_error():
    """Documents beyond what one workgroup's LDS can hold are refused before anything is enqueued; the library stays usable."""
    V = 300
    m = build_model("MATCH_TENSOR", vocab=V, device=DEV)
    rng = np.random.default_rng(5)
    q, ql, d, dl = _synth(rng, 1, 1, 4, 512, V)
    with pytest.raises(RuntimeError, match="LDS"):
        m(q.to(DEV), ql.to(DEV), d.to(DEV), dl.to(DEV))
    q, ql, d, dl = _synth(rng, 1, 2, 6, 24, V)
    _close(m(q.to(DEV), ql.to(DEV), d.to(DEV), dl.to(DEV)), O.match_tensor_scores(cpu_state_dict(m), q, ql, d, dl))


@pytest.mark.parametrize("B,S,N,kw", [(3, 4, 100, {}), (2, 3, 65, {}), (1, 70, 5, {}), (2, 66, 70, dict(query_session_off=True)),
                                      (2, 5, 130, dict(doc_session_off=True))])
def test_cars_many_candidates_long_sessions(B, S, N, kw):
    """--num_candidates is free in the reference (config.py:42) and the session axis has no bound (cars.py:346): N > 64 takes the
    LDS-staged click pooling, S > 63 the dynamically sized attention logits."""
    from context_attentive_ir_amd import synth
    V = 2000
    m = build_model("CARS", vocab=V, tgt_vocab_size=300, device=DEV, **kw)
    ex = synth.session_batch(B, S, N, 4, 12, V, seed=B * 100 + S + N, full_length=False, multi_click=True)
    sd = cpu_state_dict(m)
    q_on, d_on = not kw.get("query_session_off", False), not kw.get("doc_session_off", False)
    pooled_ref, _ = O.cars_encode(sd, ex["source_words"], ex["source_lens"])
    s_ref, st_ref, at_ref = O.cars_rank_document_full(sd, pooled_ref, ex["document_words"], ex["document_lens"], ex["document_labels"],
                                                      q_on=q_on, d_on=d_on)
    dex = {k: v.to(DEV) for k, v in ex.items()}
    pooled, _, _ = m.encode(dex["source_words"], dex["source_lens"])
    s, st, at = m.rank_document(pooled, dex["document_words"], dex["document_lens"], dex["document_labels"])
    _close(s, s_ref)
    _close(st[0], st_ref[0], 5e-5)
    for a, b in zip(at, at_ref):
        assert (a is None) == (b is None)
        if a is not None:
            _close(a, b, 5e-5)
    if d_on:
        clicks = m.encode_clicks(m.encode_document(dex["document_words"], dex["document_lens"]), dex["document_labels"])
        _close(clicks, O.cars_encode_clicks(sd, O.cars_encode_document(sd, ex["document_words"], ex["document_lens"]), ex["document_labels"]), 2e-5)


@pytest.mark.parametrize("B,N,QL,DL", [(2, 3, 40, 64), (1, 2, 26, 200), (1, 2, 64, 30), (1, 2, 120, 16)])
def test_drmm_long_queries(B, N, QL, DL):
    """Queries longer than 25 tokens (README.md:81: up to 40): LDS-atomic histogram instantiation; exact counts on edge-safe ids."""
    rng = np.random.default_rng(QL)
    V = 5000
    m = build_model("DRMM", vocab=V, device=DEV)
    q, ql, d, dl = _synth(rng, B, N, QL, DL, V)
    q[q > 0] = q[q > 0] % 1000 + 4; d[d > 0] = d[d > 0] % 3000 + 1500      # disjoint vocab halves: edge-safe
    sd = cpu_state_dict(m)
    gate, cos, hist_ref = O.drmm_parts(sd, q, d)
    s, hist = m(q.to(DEV), ql.to(DEV), d.to(DEV), dl.to(DEV), return_hist=True)
    c = cos.numpy(); edge = np.abs(c[..., None] - np.array([-1, -.5, 0, .5, 1.0])).min(-1); edge[c == 0] = 1
    safe = (edge > 2e-6).all(axis=(1, 2))
    np.testing.assert_array_equal(hist.cpu().numpy()[safe], hist_ref.numpy()[safe])
    ref = O.drmm_scores_from_hist(sd, gate, hist_ref, B, N).reshape(-1)
    _close(s.reshape(-1)[torch.from_numpy(safe)], ref[torch.from_numpy(safe)], 2e-4)
