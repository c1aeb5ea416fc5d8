"""RNNEncoder on the HIP path.

Mirror of neuroir.encoders.RNNEncoder (/root/reference/neuroir/encoders/rnn_encoder.py:14-141) for the
configuration the hot path uses: rnn_type='LSTM', nlayers=1, batch_first.  Parameters live in an nn.LSTM
(`rnns.0.*`) purely as a container so that state-dict keys match; the forward never calls it:
   gates = x W_ih^T + b_ih + b_hh   -> nir_linear_f32  (fp32 MFMA GEMM, both directions at once)
   recurrence                       -> nir_bilstm_fwd  (masking instead of sort + pack + unpack)
Returns (final_state, memory_bank) like the reference; final_state is (h_n, c_n) in ORIGINAL batch order
(the reference leaves it in length-sorted order, an artefact nobody on the hot path consumes, Appendix E4).
"""
import torch
import torch.nn as nn

from .. import lib


def lstm_cat_weights(lstm):
    """[fwd; rev] concatenation in the layout nir_bilstm_fwd / the gate GEMM expect."""
    sfx = ["", "_reverse"] if lstm.bidirectional else [""]
    wih = torch.cat([getattr(lstm, "weight_ih_l0" + s) for s in sfx], 0)
    whh = torch.stack([getattr(lstm, "weight_hh_l0" + s) for s in sfx], 0)
    bih = torch.cat([getattr(lstm, "bias_ih_l0" + s) for s in sfx], 0)
    bhh = torch.cat([getattr(lstm, "bias_hh_l0" + s) for s in sfx], 0)
    return wih, whh, bih, bhh


class RNNEncoder(nn.Module):
    def __init__(self, rnn_type, input_size, bidirectional, num_layers, hidden_size, dropout=0.0,
                 use_bridge=False, use_last=True):
        super().__init__()
        if rnn_type != "LSTM" or num_layers != 1 or use_bridge:
            raise NotImplementedError("HIP RNNEncoder supports rnn_type='LSTM', nlayers=1, no bridge "
                                      "(the hot-path configuration, hyparam.py:88-105,197-225)")
        ndir = 2 if bidirectional else 1
        assert hidden_size % ndir == 0
        self.nlayers, self.ndir, self.hidden = 1, ndir, hidden_size // ndir
        self.rnns = nn.ModuleList([nn.LSTM(input_size, self.hidden, 1, bidirectional=bidirectional, batch_first=True)])
        self.dropout = nn.Dropout(dropout)
        self._pack = lib.PackCache()

    def packed(self):
        lstm = self.rnns[0]
        return self._pack.get(list(lstm.parameters()), lambda: [t.detach().float().contiguous() for t in lstm_cat_weights(lstm)])

    def forward(self, emb, lengths=None, init_states=None):
        lib.require_device(emb, lengths)
        L = lib.load()
        M, T, I = emb.shape
        H, ND = self.hidden, self.ndir
        wih, whh, bih, bhh = self.packed()
        x = emb.float().contiguous()
        st = lib.stream()
        out = torch.empty(M, T, ND * H, device=emb.device, dtype=torch.float32)
        hn = torch.empty(ND, M, H, device=emb.device, dtype=torch.float32)
        cn = torch.empty_like(hn)
        h0 = c0 = None
        if init_states is not None:
            h0, c0 = (s.float().contiguous() for s in init_states)
        lens = lib.ids64(lengths) if lengths is not None else None
        if I <= 64 and H <= 128:   # narrow inputs: W_ih lives in registers inside the recurrence, no gate tensor
            lib.check(L.nir_bilstm_fused_fwd(lib.ptr(x), I, lib.ptr(wih), lib.ptr(bih), lib.ptr(bhh), lib.ptr(lens),
                                             lib.ptr(whh), lib.ptr(h0), lib.ptr(c0), lib.ptr(out), lib.ptr(hn),
                                             lib.ptr(cn), M, T, H, ND, st), "nir_bilstm_fused_fwd")
            return (hn, cn), out
        gates = torch.empty(M * T, ND * 4 * H, device=emb.device, dtype=torch.float32)
        lib.check(L.nir_linear_f32(lib.ptr(x), I, None, None, 0, 0, 0, lib.ptr(wih), I, lib.ptr(bih), lib.ptr(bhh),
                                   lib.ptr(gates), ND * 4 * H, M * T, ND * 4 * H, I, 0, st), "nir_linear_f32")
        if H > 128:   # beyond the register-resident recurrences: streaming form, one GEMM + one cell kernel per step
            ws = lib.workspace(L.nir_bilstm_steps_workspace_bytes(M, H), emb.device)
            lib.check(L.nir_bilstm_steps_fwd(lib.ptr(gates), lib.ptr(lens), lib.ptr(whh), lib.ptr(h0), lib.ptr(c0), lib.ptr(out),
                                             lib.ptr(hn), lib.ptr(cn), M, T, H, ND, lib.ptr(ws), ws.numel(), st),
                      "nir_bilstm_steps_fwd")
            return (hn, cn), out
        lib.check(L.nir_bilstm_fwd(lib.ptr(gates), lib.ptr(lens), lib.ptr(whh), lib.ptr(h0), lib.ptr(c0), lib.ptr(out),
                                   lib.ptr(hn), lib.ptr(cn), M, T, H, ND, st), "nir_bilstm_fwd")
        return (hn, cn), out
