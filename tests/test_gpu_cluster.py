"""GPU (-m gpu): the four-workgroup resident-weight recurrence for 256 units per direction (csrc/lstm_cluster.hip) through the C-ABI, against
torch's CPU nn.LSTM over packed sequences (what the reference's RNNEncoder runs, encoders/rnn_encoder.py:62-141) -- memory bank with zeros
beyond each length (mode 0) and MNSRF's max over time (mode 1, mnsrf.py:79-83).  fp32 parity: 3e-5 absolute on |h| < 1."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _reference_bank(lstm, x, lens):
    T = x.shape[1]
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False)
    out, _ = lstm(packed)
    out, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=T)
    return out


def _run(lstm, table, ids, lens, mode, use_ids=True, hint=0):
    from context_attentive_ir_amd import lib
    from context_attentive_ir_amd.encoders.rnn_encoder import lstm_cat_weights
    L = lib.load()
    nd = 2 if lstm.bidirectional else 1
    M, T = ids.shape
    wih, whh, bih, bhh = [t.detach().to(DEV).contiguous() for t in lstm_cat_weights(lstm)]
    ids_d = ids.to(DEV)
    if use_ids:
        rows = lib.fold_lstm_table(table.to(DEV), wih, bih, bhh, 256, nd, "f32")
        idp, R = lib.ptr(ids_d), table.shape[0]
    else:       # per-batch gates in the folded order: row = m*T + t
        rows = lib.fold_lstm_table(table[ids.reshape(-1)].to(DEV).contiguous(), wih, bih, bhh, 256, nd, "f32")
        idp, R = None, M * T
    frag = torch.empty(L.nir_lstm256_whh_frag_bytes(nd), dtype=torch.uint8, device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    lib.check(L.nir_lstm256_pack_whh_frag(lib.ptr(whh), nd, lib.ptr(frag), lib.ptr(err), lib.stream()), "pack")
    ws = torch.empty(L.nir_lstm256_workspace_bytes(M, nd), dtype=torch.uint8, device=DEV)
    out = torch.full((M, T, nd * 256) if mode == 0 else (M, nd * 256), 7.0, device=DEV)
    lens_d = lens.to(DEV)
    if hint:
        lib.set_batches_in_flight(hint)              # the per-stream scheduling hint must not change the numbers
    try:
        lib.check(L.nir_lstm256_rows_fwd(lib.ptr(rows), idp, lib.ptr(lens_d), lib.ptr(frag), lib.ptr(out), mode, lib.ptr(err), M, R, T, nd,
                                         lib.ptr(ws), ws.numel(), lib.stream()), "nir_lstm256_rows_fwd")
    finally:
        if hint:
            lib.set_batches_in_flight(0)
    torch.cuda.synchronize()
    assert int(err.item()) == 0, int(err.item())
    del ids_d
    return out.cpu()


@pytest.mark.parametrize("M,T,bi,full", [(37, 7, True, False), (16, 4, True, True), (5, 3, False, False), (112, 4, True, False),
                                         (1120, 64, True, False), (1120, 64, True, True), (49, 33, True, False), (130, 12, False, False),
                                         (300, 20, True, False), (257, 9, False, False)])
def test_lstm256_cluster_matches_packed_lstm(M, T, bi, full):
    g = torch.Generator().manual_seed(M * 100 + T)
    V, E = 500, 300
    lstm = torch.nn.LSTM(E, 256, bidirectional=bi, batch_first=True)
    with torch.no_grad():
        for p in lstm.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.08)
    table = torch.randn(V, E, generator=g) * 0.5
    ids = torch.randint(1, V, (M, T), generator=g)
    lens = torch.full((M,), T, dtype=torch.int64) if full else torch.randint(1, T + 1, (M,), generator=g)
    lens[0] = T
    if not full and M > 3:
        lens[1] = 1
    with torch.no_grad():
        ref = _reference_bank(lstm, table[ids], lens)
    bank = _run(lstm, table, ids, lens, 0)
    np.testing.assert_allclose(bank.numpy(), ref.numpy(), rtol=0, atol=3e-5)
    assert float(bank[1, int(lens[1]):].abs().max() if int(lens[1]) < T else 0.0) == 0.0       # exact zeros beyond the length
    pooled = _run(lstm, table, ids, lens, 1)
    np.testing.assert_allclose(pooled.numpy(), ref.max(1)[0].numpy(), rtol=0, atol=3e-5)
    if M >= 256:                                       # with the several-batches-in-flight hint set: the same numbers
        bank32 = _run(lstm, table, ids, lens, 0, hint=4)
        np.testing.assert_allclose(bank32.numpy(), ref.numpy(), rtol=0, atol=3e-5)
        pooled32 = _run(lstm, table, ids, lens, 1, hint=4)
        np.testing.assert_allclose(pooled32.numpy(), ref.max(1)[0].numpy(), rtol=0, atol=3e-5)
    if M <= 112:                                                                                  # per-batch gate rows (ids == NULL)
        bank2 = _run(lstm, table, ids, lens, 0, use_ids=False)
        np.testing.assert_allclose(bank2.numpy(), bank.numpy(), rtol=0, atol=1e-5)      # (the two gate GEMMs differ in shape -> kernel -> rounding)


def test_lstm256_cluster_concurrent_launches_on_four_streams():
    """Four launches in flight on four streams (clusters of different launches compete for CUs): same results as alone, no timeout flag."""
    g = torch.Generator().manual_seed(3)
    V, E, M, T = 500, 300, 640, 32
    lstm = torch.nn.LSTM(E, 256, bidirectional=True, batch_first=True)
    table = torch.randn(V, E, generator=g) * 0.5
    ids = [torch.randint(1, V, (M, T), generator=g) for _ in range(4)]
    lens = [torch.randint(1, T + 1, (M,), generator=g) for _ in range(4)]
    alone = [_run(lstm, table, i, l, 1) for i, l in zip(ids, lens)]
    from context_attentive_ir_amd import lib
    from context_attentive_ir_amd.encoders.rnn_encoder import lstm_cat_weights
    L = lib.load()
    wih, whh, bih, bhh = [t.detach().to(DEV).contiguous() for t in lstm_cat_weights(lstm)]
    rows = lib.fold_lstm_table(table.to(DEV), wih, bih, bhh, 256, 2, "f32")
    frag = torch.empty(L.nir_lstm256_whh_frag_bytes(2), dtype=torch.uint8, device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    lib.check(L.nir_lstm256_pack_whh_frag(lib.ptr(whh), 2, lib.ptr(frag), lib.ptr(err), lib.stream()), "pack")
    streams = [torch.cuda.Stream() for _ in range(4)]
    wss = [torch.empty(L.nir_lstm256_workspace_bytes(M, 2), dtype=torch.uint8, device=DEV) for _ in range(4)]
    outs = [torch.empty(M, 512, device=DEV) for _ in range(4)]
    di, dl = [i.to(DEV) for i in ids], [l.to(DEV) for l in lens]
    torch.cuda.synchronize()
    for rep in range(5):
        for k in range(4):
            with torch.cuda.stream(streams[k]):
                lib.check(L.nir_lstm256_rows_fwd(lib.ptr(rows), lib.ptr(di[k]), lib.ptr(dl[k]), lib.ptr(frag), lib.ptr(outs[k]), 1, lib.ptr(err),
                                                 M, V, T, 2, lib.ptr(wss[k]), wss[k].numel(), lib.stream()), "fwd")
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    for k in range(4):
        assert torch.equal(outs[k].cpu(), alone[k])


def test_cluster_timeout_fails_safe_to_the_streaming_recurrence():
    """a cluster that cannot make progress (forced here: the debug tunable cl_poll_limit = -1 makes the first unsuccessful poll
    give up) raises bit 2 of the device's error word.  The wrappers publish / poll that word even with id_check_interval = 0: the failed call's
    results raise RuntimeError at the caller's synchronisation, the model switches to the streaming recurrence (nir_birnn_steps_fwd behind
    nir_mnsrf_*) BEFORE the error is raised, and the re-issued batch equals the oracle."""
    from context_attentive_ir_amd import lib, synth
    from context_attentive_ir_amd.config import default_args
    from context_attentive_ir_amd.detinit import fill_module_
    from context_attentive_ir_amd.wrappers import Multitask
    from oracle import neuroir_cpu as O
    V = 300
    w = Multitask(default_args("MNSRF", src_vocab_size=V, tgt_vocab_size=40))
    fill_module_(w.network, 29)
    sd = {k: v.detach().cpu().float() for k, v in w.network.state_dict().items()}
    w.cuda()
    w.predict_graph_min_calls = 2
    w.id_check_interval = 0                                     # the caller opted out of the id check: the cluster bit is still not silent
    ex = synth.session_batch(6, 5, 8, 5, 21, V, seed=3)
    ref = torch.softmax(O.mnsrf_scores(sd, ex["source_words"], ex["source_lens"], ex["document_words"], ex["document_lens"]), -1)
    good = w.predict(ex, suggest=False)["click_scores"].cpu()
    assert float((good - ref.view_as(good)).abs().max()) < 1e-4 and w.network.resident_recurrence
    assert w.network._weights().struct.d_whh_frag
    with lib.tunable("cl_poll_limit", -1, 0):
        out = w.predict(ex, suggest=False)                      # enqueued, nothing raised yet
        torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="streaming recurrence"):
        w.predict(ex, suggest=False)                            # entry of the next call: the pinned word holds bit 2
    del out
    assert not w.network.resident_recurrence and not w.network._weights().struct.d_whh_frag
    for _ in range(3):                                          # eager, capture, replay -- all on the streaming recurrence
        again = w.predict(ex, suggest=False)["click_scores"].cpu()
        assert float((again - ref.view_as(again)).abs().max()) < 1e-4
    w.check_ids()
    from context_attentive_ir_amd import autograd as A
    A.CLUSTER_TRAIN_FWD = True                                  # (module-level switch: restore for the tests that follow)
