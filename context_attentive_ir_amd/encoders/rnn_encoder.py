"""RNNEncoder on the HIP path.

Mirror of neuroir.encoders.RNNEncoder (/root/reference/neuroir/encoders/rnn_encoder.py:14-185): rnn_type 'LSTM' or 'GRU', any number of
stacked layers (one single-layer module per layer, `rnns.{i}.*`, as the reference builds them), `use_last`, optional bridge.  Parameters
live in nn.LSTM / nn.GRU / nn.Linear modules purely as containers so that state-dict keys match; the forward never calls them:
   gates = x W_ih^T + b_ih (+ b_hh for the LSTM) -> nir_linear_f32  (MFMA GEMM, both directions at once)
   recurrence                                    -> LSTM: nir_bilstm_fused_fwd / nir_bilstm_fwd / nir_bilstm_steps_fwd by size;
                                                    GRU: nir_birnn_steps_fwd (streaming form: one GEMM + one cell kernel per step)
   bridge                                        -> nir_linear_f32 with the ReLU epilogue
(masking instead of sort + pack + unpack).  The hot path of hyparam.py:88-105,197-225 is the 1-layer LSTM; the folded-table kernels of the
model classes cover that configuration, this class covers the rest of the constructor's envelope.
Returns (final_state, memory_bank) like the reference; final states are in ORIGINAL batch order (the reference leaves them in
length-sorted order, an artefact nobody on the hot path consumes, Appendix E4).  Dropout between layers is the eval-mode identity; in
train mode with dropout > 0 and more than one layer the forward raises (training runs through autograd.py's 1-layer LSTM operators).
"""
import torch
import torch.nn as nn

from .. import lib

CELLS = {"LSTM": 0, "GRU": 1}


def rnn_cat_weights(rnn):
    """[fwd; rev] concatenation in the layout the gate GEMM / the recurrences expect: w_ih [ND*G*H, I], w_hh [ND, G*H, H], b_ih [ND*G*H],
    b_hh [ND*G*H] (G = 4 gates for the LSTM, 3 for the GRU)."""
    sfx = ["", "_reverse"] if rnn.bidirectional else [""]
    wih = torch.cat([getattr(rnn, "weight_ih_l0" + s) for s in sfx], 0)
    whh = torch.stack([getattr(rnn, "weight_hh_l0" + s) for s in sfx], 0)
    bih = torch.cat([getattr(rnn, "bias_ih_l0" + s) for s in sfx], 0)
    bhh = torch.cat([getattr(rnn, "bias_hh_l0" + s) for s in sfx], 0)
    return wih, whh, bih, bhh


lstm_cat_weights = rnn_cat_weights      # (older name, used by the model classes)


class RNNEncoder(nn.Module):
    def __init__(self, rnn_type, input_size, bidirectional, num_layers, hidden_size, dropout=0.0,
                 use_bridge=False, use_last=True):
        super().__init__()
        if rnn_type not in CELLS:
            raise NotImplementedError("HIP RNNEncoder supports rnn_type 'LSTM' and 'GRU' (the reference's getattr(nn, rnn_type) also admits "
                                      "'RNN'; no configuration of the reference uses it)")
        ndir = 2 if bidirectional else 1
        assert hidden_size % ndir == 0
        self.rnn_type, self.cell = rnn_type, CELLS[rnn_type]
        self.nlayers, self.ndir, self.hidden, self.use_last = int(num_layers), ndir, hidden_size // ndir, use_last
        self.rnns = nn.ModuleList()
        for i in range(self.nlayers):
            isz = input_size if i == 0 else self.hidden * ndir
            self.rnns.append(getattr(nn, rnn_type)(isz, self.hidden, 1, bidirectional=bidirectional, batch_first=True))
        self.dropout = nn.Dropout(dropout)
        self.use_bridge = use_bridge
        if use_bridge:                                            # rnn_encoder.py:143-157
            nl = 1 if use_last else self.nlayers
            self.total_hidden_dim = self.hidden * nl
            self.bridge = nn.ModuleList([nn.Linear(self.total_hidden_dim, self.total_hidden_dim, bias=True)
                                         for _ in range(2 if rnn_type == "LSTM" else 1)])
        self._pack = lib.PackCache()

    def packed(self, layer=0):
        def build():
            return [[t.detach().float().contiguous() for t in rnn_cat_weights(r)] for r in self.rnns]
        return self._pack.get([p for r in self.rnns for p in r.parameters()], build)[layer]

    def _lstm_layer(self, x, lens, h0, c0, w):
        L = lib.load()
        M, T, I = x.shape
        H, ND = self.hidden, self.ndir
        wih, whh, bih, bhh = w
        st = lib.stream()
        out = torch.empty(M, T, ND * H, device=x.device, dtype=torch.float32)
        hn = torch.empty(ND, M, H, device=x.device, dtype=torch.float32)
        cn = torch.empty_like(hn)
        if I <= 64 and H <= 128:   # narrow inputs: W_ih lives in registers inside the recurrence, no gate tensor
            lib.check(L.nir_bilstm_fused_fwd(lib.ptr(x), I, lib.ptr(wih), lib.ptr(bih), lib.ptr(bhh), lib.ptr(lens),
                                             lib.ptr(whh), lib.ptr(h0), lib.ptr(c0), lib.ptr(out), lib.ptr(hn),
                                             lib.ptr(cn), M, T, H, ND, st), "nir_bilstm_fused_fwd")
            return (hn, cn), out
        gates = torch.empty(M * T, ND * 4 * H, device=x.device, dtype=torch.float32)
        lib.check(L.nir_linear_f32(lib.ptr(x), I, None, None, 0, 0, 0, lib.ptr(wih), I, lib.ptr(bih), lib.ptr(bhh),
                                   lib.ptr(gates), ND * 4 * H, M * T, ND * 4 * H, I, 0, st), "nir_linear_f32")
        if H > 128:   # beyond the register-resident recurrences: streaming form, one GEMM + one cell kernel per step
            ws = lib.workspace(L.nir_bilstm_steps_workspace_bytes(M, H), x.device)
            lib.check(L.nir_bilstm_steps_fwd(lib.ptr(gates), lib.ptr(lens), lib.ptr(whh), lib.ptr(h0), lib.ptr(c0), lib.ptr(out),
                                             lib.ptr(hn), lib.ptr(cn), M, T, H, ND, lib.ptr(ws), ws.numel(), st),
                      "nir_bilstm_steps_fwd")
            return (hn, cn), out
        lib.check(L.nir_bilstm_fwd(lib.ptr(gates), lib.ptr(lens), lib.ptr(whh), lib.ptr(h0), lib.ptr(c0), lib.ptr(out),
                                   lib.ptr(hn), lib.ptr(cn), M, T, H, ND, st), "nir_bilstm_fwd")
        return (hn, cn), out

    def _gru_layer(self, x, lens, h0, w):
        L = lib.load()
        M, T, I = x.shape
        H, ND = self.hidden, self.ndir
        wih, whh, bih, bhh = w
        st = lib.stream()
        out = torch.empty(M, T, ND * H, device=x.device, dtype=torch.float32)
        hn = torch.empty(ND, M, H, device=x.device, dtype=torch.float32)
        gates = torch.empty(M * T, ND * 3 * H, device=x.device, dtype=torch.float32)
        lib.check(L.nir_linear_f32(lib.ptr(x), I, None, None, 0, 0, 0, lib.ptr(wih), I, lib.ptr(bih), None,
                                   lib.ptr(gates), ND * 3 * H, M * T, ND * 3 * H, I, 0, st), "nir_linear_f32")
        ws = lib.workspace(L.nir_bilstm_steps_workspace_bytes(M, H), x.device)
        lib.check(L.nir_birnn_steps_fwd(1, lib.ptr(gates), lib.ptr(lens), lib.ptr(whh), lib.ptr(bhh), lib.ptr(h0), None, lib.ptr(out),
                                        None, lib.ptr(hn), None, M, T, H, ND, lib.ptr(ws), ws.numel(), st), "nir_birnn_steps_fwd")
        return hn, out

    def _bridge(self, hidden):
        """rnn_encoder.py:159-185: Linear + ReLU on every state, rows = states viewed as [-1, total_hidden_dim]."""
        L = lib.load()

        def one(linear, states):
            x = states.contiguous().view(-1, self.total_hidden_dim)
            y = torch.empty_like(x)
            D = self.total_hidden_dim
            wt, bs = linear.weight.detach().float().contiguous(), linear.bias.detach().float().contiguous()    # (referenced until enqueued)
            lib.check(L.nir_linear_f32(lib.ptr(x), D, None, None, 0, 0, 0, lib.ptr(wt), D, lib.ptr(bs), None, lib.ptr(y), D, x.shape[0], D, D,
                                       2, lib.stream()), "nir_linear_f32")   # NIR_ACT_RELU
            return y.view(states.shape)
        if isinstance(hidden, tuple):
            return tuple(one(layer, hidden[ix]) for ix, layer in enumerate(self.bridge))
        return one(self.bridge[0], hidden)

    def forward(self, emb, lengths=None, init_states=None):
        lib.require_device(emb, lengths)
        if self.training and self.nlayers > 1 and self.dropout.p > 0:
            raise NotImplementedError("HIP RNNEncoder: dropout between stacked layers is implemented as the eval-mode identity; "
                                      "train with dropout_rnn = 0 or one layer (autograd.py covers the 1-layer LSTM)")
        ND = self.ndir
        x = emb.float().contiguous()
        lens = lib.ids64(lengths) if lengths is not None else None
        h_all = c_all = None
        if init_states is not None:                               # [(layers*directions), M, H] (pair for the LSTM): layer i takes its ND rows
            if isinstance(init_states, tuple):
                h_all, c_all = (s.float().contiguous() for s in init_states)
            else:
                h_all = init_states.float().contiguous()
        bank, h_fin, c_fin = [], [], []
        for i in range(self.nlayers):
            h0 = h_all[i * ND:(i + 1) * ND].contiguous() if h_all is not None else None
            c0 = c_all[i * ND:(i + 1) * ND].contiguous() if c_all is not None else None
            if self.cell == 0:
                (hn, cn), x = self._lstm_layer(x, lens, h0, c0, self.packed(i))
                c_fin.append(cn)
            else:
                hn, x = self._gru_layer(x, lens, h0, self.packed(i))
            h_fin.append(hn)
            if not self.use_last or i == self.nlayers - 1:
                bank.append(x)
        if self.use_last:
            memory_bank = bank[-1]
            final = (h_fin[-1], c_fin[-1]) if c_fin else h_fin[-1]
        else:
            memory_bank = torch.cat(bank, 2) if len(bank) > 1 else bank[0]
            final = (torch.cat(h_fin, 0), torch.cat(c_fin, 0)) if c_fin else torch.cat(h_fin, 0)
        if self.use_bridge:
            # rnn_encoder.py:166-172 views the states as [-1, total_hidden_dim]: with use_last = False and several layers one bridge row is the
            # concatenation of `nlayers` NEIGHBOURING BATCH ROWS of the length-sorted batch (a quirk of the view, kept): the bridge then
            # has to see the reference's sorted order.  (Equal lengths: torch.sort is not stable, the reference's pairing is then its own
            # sort's choice -- there is nothing to reproduce.)
            mixes_rows = (not self.use_last) and self.nlayers > 1 and lens is not None
            if mixes_rows:
                order = torch.sort(lens, 0, True)[1]
                srt = tuple(t[:, order] for t in final) if isinstance(final, tuple) else final[:, order]
                br = self._bridge(srt)

                def back(t):
                    o = torch.empty_like(t)
                    o[:, order] = t
                    return o
                final = tuple(back(t) for t in br) if isinstance(br, tuple) else back(br)
            else:
                final = self._bridge(final)
        return final, memory_bank
