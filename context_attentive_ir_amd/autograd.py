"""Differentiable operators of the training step -- torch.autograd.Function wrappers whose
forward AND backward run on the hand-written HIP kernels of csrc/train.hip / csrc/gemm.hip:

    linear      y = act(x W^T + b)          fwd nir_linear_f32; bwd: act' (nir_act_bwd_f32), dX = dY W (nir_linear_f32 on the
                                            transposed weight), dW = dY^T X (nir_linear_wgrad_f32), db (nir_colsum_f32)
    embed       rows of the embedding table (nir_embed_f32; bwd scatter-add nir_embed_bwd_f32, skipped for a frozen table)
    dropout     counter-based inverted dropout (nir_dropout_f32; the keep mask is kept for the backward and for parity replays)
    bilstm      RNNEncoder in train mode: gate GEMM + nir_lstm_train_fwd (saves gate activations / cell states);
                bwd: BPTT nir_lstm_train_bwd -> dgates, then dW_ih / dW_hh / db / dx as GEMMs
    bce_with_logits   nir_rank_loss_bce / nir_rank_loss_bce_bwd

What stays in torch in the train-mode model forwards (models call these functions and glue them with tensor reshapes,
concatenations, element-wise products and max-reductions) is stated in DESIGN.md section 8; every matrix product, recurrence,
lookup and loss above is HIP.  Optimiser state updates (Adam/SGD) and clip_grad_norm use torch.optim like the reference.
"""
import torch
from torch.autograd import Function

from . import lib

ACT = {None: 0, "none": 0, "tanh": 1, "relu": 2}


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _linear_raw(x2, w, b, act):
    """x2 [M,K] contiguous, w [N,K] -> [M,N] through nir_linear_f32."""
    L = lib.load()
    M, K = x2.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x2.device, dtype=torch.float32)
    if M:
        lib.check(L.nir_linear_f32(lib.ptr(x2), K, None, None, 0, 0, 0, lib.ptr(w), K, lib.ptr(b), None, lib.ptr(y), N, M, N, K, act,
                                   lib.stream()), "nir_linear_f32")
    return y


class StepScope(object):
    """Per-training-step state of the operators (opened by the wrappers' _update_body around forward + backward):

    * gradients of PARAMETERS that enter `linear` are not formed use by use: every use parks its (dY, X) pair, and when the scope closes each
      parameter gets ONE weight-gradient launch over the row-concatenation of its pairs -- dW = sum_u dY_u^T X_u = [dY_1; dY_2; ..]^T [X_1; X_2; ..]
      -- with its bias's column sum from the same launch -- accumulated into a persistent buffer that becomes `p.grad` (the buffers of a step are
      views of one arena, zero-filled by ONE launch: no memset node per sliced weight-gradient launch).  Left to
      autograd, a weight used at every step of the session / decoder loops produced one small launch and one gradient tensor per use plus an
      `add` kernel per pair (a CARS step: ~140 weight-gradient launches, ~110 column sums, ~350 adds);
    * the transposed weight of the data-gradient GEMM (dX = dY W) is formed once per weight and step, not once per use.
    Nothing here synchronises with the host, so a captured step (wrappers.GraphedUpdate) replays it as is."""

    def __init__(self):
        self.active = False
        self.bufs = {}            # id(param) -> (param, persistent gradient buffer)
        self.pend_w = {}          # id(weight) -> (param, [(dY [M,N], X [M,K]), ..]) of this step
        self.pend_b = {}          # id(bias)   -> (param, [dY, ..])
        self.wt = {}              # (data_ptr, version, shape) -> (source tensor, transposed weight), this step
        self.pair_b = {}          # id(weight) -> bias parameter parked with the same dY tensors
        self.arena_key, self.arenas = None, []
        self.tprev, self.tcur = [], []      # weights whose transposes the previous / this step asked for (grouped into one launch)

    def begin(self):
        self.active, self.pend_w, self.pend_b, self.wt, self.pair_b = True, {}, {}, {}, {}
        self.tcur = []
        # a caller that cleared gradients IN PLACE (zero_grad(set_to_none=False)) left last step's buffer installed as p.grad: autograd would
        # accumulate into it and end() would overwrite / double it -- the buffer is this scope's, so it is detached from the parameter here
        for p, buf in self.bufs.values():
            if p.grad is buf:
                p.grad = None

    def abort(self):
        self.active, self.pend_w, self.pend_b, self.wt, self.pair_b = False, {}, {}, {}, {}

    def grad_buffer(self, p):
        ent = self.bufs.get(id(p))
        if ent is None or ent[0] is not p or ent[1].shape != p.shape or ent[1].device != p.device:
            raise KeyError("StepScope: no gradient buffer laid out for this parameter")
        return ent[1]

    def _layout(self, params):
        """the step's gradient buffers as views of ONE arena (zero-filled by one launch per step; the weight gradients then accumulate: no
        memset node per sliced launch -- a CARS step had 67 of them); re-laid only when the set of parameters changes"""
        key = tuple((id(p), p.numel(), str(p.device)) for p in params)
        if key != self.arena_key:
            by_dev = {}
            for p in params:
                by_dev.setdefault(str(p.device), []).append(p)
            self.bufs, self.arenas = {}, []
            for ps in by_dev.values():
                offs, n = [], 0
                for p in ps:
                    offs.append(n)
                    n += (p.numel() + 63) // 64 * 64
                arena = torch.empty(n, device=ps[0].device, dtype=torch.float32)
                self.arenas.append(arena)
                for p, o in zip(ps, offs):
                    self.bufs[id(p)] = (p, arena[o:o + p.numel()].view(p.shape))
            self.arena_key = key
        for a in self.arenas:
            a.zero_()

    def park_w(self, p, d, x2, bias=None):
        self.pend_w.setdefault(id(p), (p, []))[1].append((d, x2))
        if bias is not None:
            self.pair_b[id(p)] = bias        # the bias that shares this weight's dY: its column sum rides in the weight-gradient launch

    def park_b(self, p, d):
        self.pend_b.setdefault(id(p), (p, []))[1].append(d)

    def end(self):
        """one weight-gradient launch / one column sum per parameter, then p.grad = buffer (plus whatever autograd itself accumulated for the
        parameter elsewhere, e.g. a norm regulariser)"""
        L = lib.load()
        done, fused_b, group = [], set(), []
        seen, params = set(), []
        for p in [e[0] for e in self.pend_w.values()] + [e[0] for e in self.pend_b.values()]:
            if id(p) not in seen:
                seen.add(id(p)); params.append(p)
        self._layout(params)
        for p, pairs in self.pend_w.values():
            pairs = [(d, x) for d, x in pairs if d.shape[0]]
            buf = self.grad_buffer(p)
            N, K = p.shape
            if pairs:
                d = pairs[0][0] if len(pairs) == 1 else torch.cat([a for a, _ in pairs], 0)
                x = pairs[0][1] if len(pairs) == 1 else torch.cat([b for _, b in pairs], 0)
                # the bias of the same nn.Linear, parked with exactly these dY tensors: its column sum comes out of the same launch
                bp = self.pair_b.get(id(p))
                bent = self.pend_b.get(id(bp)) if bp is not None else None
                bbuf = None
                if bent is not None and len(bent[1]) == len(self.pend_w[id(p)][1]) and all(a is b_[0] for a, b_ in zip(bent[1], self.pend_w[id(p)][1])):
                    bbuf = self.grad_buffer(bp)
                    done.append((bp, bbuf))
                    fused_b.add(id(bp))
                group.append((d, x, buf, bbuf, d.shape[0], N, K))
            done.append((p, buf))
        if group:
            # ALL weight gradients of the step through one call: the small ones share launches (nir_linear_wgrad_group_f32)
            n = len(group)
            PA, LA, IA = lib.C.c_void_p * n, lib.C.c_int64 * n, lib.C.c_int * n
            args = (PA(*[g[0].data_ptr() for g in group]), LA(*[g[5] for g in group]), PA(*[g[1].data_ptr() for g in group]), LA(*[g[6] for g in group]),
                    PA(*[g[2].data_ptr() for g in group]), LA(*[g[6] for g in group]), PA(*[(g[3].data_ptr() if g[3] is not None else None) for g in group]),
                    LA(*[g[4] for g in group]), IA(*[g[5] for g in group]), IA(*[g[6] for g in group]))
            lib.check(L.nir_linear_wgrad_group_f32(n, *args, lib.stream()), "nir_linear_wgrad_group_f32")
        for p, ds in self.pend_b.values():
            if id(p) in fused_b:
                continue
            ds = [d for d in ds if d.shape[0]]
            buf = self.grad_buffer(p)
            N = p.shape[0]
            if ds:
                d = ds[0] if len(ds) == 1 else torch.cat(ds, 0)
                lib.check(L.nir_colsum_f32(lib.ptr(d), N, d.shape[0], N, lib.ptr(buf), lib.stream()), "nir_colsum_f32")
            done.append((p, buf))
        for p, buf in done:
            if p.grad is None:
                p.grad = buf
            else:
                p.grad.add_(buf)
        self.tprev = [q for q in self.tcur if _is_param(q) and q.dim() == 2][:256]
        self.abort()


STEP = StepScope()
SPLIT_TRAIN_FWD = True     # _BiLSTM.forward at 64 < H <= 128: the split-fp16 matrix-core recurrence (False: the fp32 MFMA one, any H <= 128)
CLUSTER_TRAIN_FWD = True   # bilstm at 256 units per direction: forward on the cluster recurrence (False: two unidirectional lstm_seq passes)
TWO_STREAM_BPTT = True     # _BiLSTM256.backward (FUSED_BPTT256 = False): the reverse direction's step chain on a side stream
FUSED_BPTT256 = True       # _BiLSTM256.backward: one launch per step for both directions, cell gradients + the recurrent product (nir_lstm256_bptt);
                           # False: rounds 3-5's masked cell kernel + GEMM per step and direction
PACKED_WGRAD = False       # _BiLSTM.backward: reduce the weight gradients over a list of the valid (t < length) positions only


def _is_param(t):
    return isinstance(t, torch.nn.Parameter) and t.requires_grad and t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda


def _transpose(w):
    key = (w.data_ptr(), w._version, tuple(w.shape))
    if STEP.active and key in STEP.wt:
        return STEP.wt[key][1]
    L = lib.load()
    if STEP.active and _is_param(w):
        STEP.tcur.append(w)
        if any(q is w for q in STEP.tprev):
            # a weight the previous step transposed too: transpose ALL of that step's weights now, in one launch (nir_transpose_group_f32)
            ws = [q for q in STEP.tprev if q.is_cuda and q.device == w.device and (q.data_ptr(), q._version, tuple(q.shape)) not in STEP.wt]
            outs = [torch.empty(q.shape[1], q.shape[0], device=q.device, dtype=torch.float32) for q in ws]
            n = len(ws)
            PA, IA = lib.C.c_void_p * n, lib.C.c_int * n
            lib.check(L.nir_transpose_group_f32(n, PA(*[q.data_ptr() for q in ws]), IA(*[q.shape[0] for q in ws]), IA(*[q.shape[1] for q in ws]),
                                                PA(*[o.data_ptr() for o in outs]), lib.stream()), "nir_transpose_group_f32")
            for q, o in zip(ws, outs):
                STEP.wt[(q.data_ptr(), q._version, tuple(q.shape))] = (q, o)
            return STEP.wt[key][1]
    R, Cc = w.shape
    out = torch.empty(Cc, R, device=w.device, dtype=torch.float32)
    lib.check(L.nir_transpose_f32(lib.ptr(w), R, Cc, lib.ptr(out), lib.stream()), "nir_transpose_f32")
    if STEP.active:
        STEP.wt[key] = (w, out)      # the source is held for the step: its address cannot be handed to another same-shape temporary while the entry lives
    return out


def _wgrad(dy2, lddy, x2, ldx, M, N, K):
    """dW [N,K] = sum_m dy[m,:N]^T x[m,:K]; dy2 / x2 may be column windows of wider row-major buffers (lddy / ldx)."""
    L = lib.load()
    dw = torch.empty(N, K, device=x2.device, dtype=torch.float32)          # "=" form: no zero fill (a step has ~140 of these)
    lib.check(L.nir_linear_wgrad_set_f32(lib.ptr(dy2) if M else lib.ptr(dw), lddy, lib.ptr(x2) if M else lib.ptr(dw), ldx, None, None, 0, lib.ptr(dw), K, M, N,
                                         K, lib.stream()), "nir_linear_wgrad_set_f32")
    return dw


def _wgrad_bias(dy2, lddy, x2, ldx, M, N, K):
    """(dW, db) of one nn.Linear from ONE pass over dY (nir_linear_wgrad_bias_set_f32)."""
    L = lib.load()
    dw = torch.empty(N, K, device=x2.device, dtype=torch.float32)
    db = torch.empty(N, device=x2.device, dtype=torch.float32)
    lib.check(L.nir_linear_wgrad_bias_set_f32(lib.ptr(dy2) if M else lib.ptr(dw), lddy, lib.ptr(x2) if M else lib.ptr(dw), ldx, None, None, 0, lib.ptr(dw), K,
                                              lib.ptr(db), M, N, K, lib.stream()), "nir_linear_wgrad_bias_set_f32")
    return dw, db


def _colsum(dy2, ld, M, N):
    L = lib.load()
    out = torch.empty(N, device=dy2.device, dtype=torch.float32)
    lib.check(L.nir_colsum_set_f32(lib.ptr(dy2) if M else lib.ptr(out), ld, M, N, lib.ptr(out), lib.stream()), "nir_colsum_set_f32")
    return out


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, b, act):
        lib.require_device(x, w)
        shp = x.shape
        x2 = _f32c(x).reshape(-1, shp[-1])
        wc = _f32c(w)
        y = _linear_raw(x2, wc, _f32c(b) if b is not None else None, act)
        ctx.act, ctx.has_b, ctx.shp = act, b is not None, shp
        # parameters whose gradients this step accumulates in place (StepScope)
        ctx.wp = w if (STEP.active and _is_param(w)) else None
        ctx.bp = b if (STEP.active and b is not None and _is_param(b)) else None
        ctx.save_for_backward(x2, wc, y if act else None)
        return y.view(*shp[:-1], wc.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        L = lib.load()
        M, K = x2.shape
        N = w.shape[0]
        d = _f32c(dy).reshape(M, N)
        if ctx.act:
            dpre = torch.empty_like(d)
            lib.check(L.nir_act_bwd_f32(lib.ptr(d), lib.ptr(y), lib.ptr(dpre), d.numel(), ctx.act, lib.stream()), "nir_act_bwd_f32")
            d = dpre
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _linear_raw(d, _transpose(w), None, 0).view(ctx.shp)
        want_b = ctx.has_b and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if ctx.wp is not None and STEP.active:
                # (no tensor returned: formed with the parameter's other uses when the step scope closes)
                STEP.park_w(ctx.wp, d, x2, bias=ctx.bp if want_b else None)
            elif want_b and not (ctx.bp is not None and STEP.active):
                dw, db = _wgrad_bias(d, N, x2, K, M, N, K)
                want_b = False
            else:
                dw = _wgrad(d, N, x2, K, M, N, K)
        if want_b:
            if ctx.bp is not None and STEP.active:
                STEP.park_b(ctx.bp, d)
            else:
                db = _colsum(d, N, M, N)
        return dx, dw, db, None


def linear(x, weight, bias=None, act=None):
    """act(x W^T + b) over the last dim of x; act in {None, 'tanh', 'relu'}."""
    return _Linear.apply(x, weight, bias, ACT[act])


_ID_FLAGS = {}


class _Im2colRows(Function):
    """[M,C,H,W] -> patch rows [M*H*W, C*kh*kw] of a stride-1 'same' convolution (the A operand of the filter GEMM) and back."""

    @staticmethod
    def forward(ctx, x, kh, kw, ph, pw):
        xc = _f32c(x)
        M, C, H, W = xc.shape
        out = torch.empty(M * H * W, C * kh * kw, dtype=torch.float32, device=xc.device)
        lib.check(lib.load().nir_im2col_rows_f32(lib.ptr(xc), M, C, H, W, kh, kw, ph, pw, lib.ptr(out), lib.stream()), "nir_im2col_rows_f32")
        ctx.geom = (M, C, H, W, kh, kw, ph, pw)
        return out

    @staticmethod
    def backward(ctx, drows):
        M, C, H, W, kh, kw, ph, pw = ctx.geom
        d = _f32c(drows)
        dx = torch.empty(M, C, H, W, dtype=torch.float32, device=d.device)
        lib.check(lib.load().nir_col2im_rows_f32(lib.ptr(d), M, C, H, W, kh, kw, ph, pw, lib.ptr(dx), lib.stream()), "nir_col2im_rows_f32")
        return dx, None, None, None, None


def im2col_rows(x, kernel_size, padding):
    """Patch rows of Conv2d(kernel_size, stride 1, padding) over x [M,C,H,W]: [M*H*W, C*kh*kw] with k = (c, dy, dx) like conv.weight.reshape(out, -1)."""
    (kh, kw), (ph, pw) = kernel_size, padding
    lib.require_device(x)
    return _Im2colRows.apply(x, int(kh), int(kw), int(ph), int(pw))


class _MTConv3(Function):
    """relu(conv3x3 | conv3x5 | conv3x7)(T) as feature rows [M*H*W, 3*NF] -- the direct-convolution kernels of csrc/mt_conv_train.hip."""

    @staticmethod
    def forward(ctx, T, w1, b1, w2, b2, w3, b3):
        lib.require_device(T, w1)
        L = lib.load()
        Tc = _f32c(T)
        M, C1, H, W = Tc.shape
        NF = w1.shape[0]
        ws = [_f32c(t) for t in (w1, b1, w2, b2, w3, b3)]
        out = torch.empty(M * H * W, 3 * NF, device=Tc.device, dtype=torch.float32)
        lib.check(L.nir_mt_conv3_fwd(lib.ptr(Tc), *[lib.ptr(t) for t in ws], M, C1, H, W, NF, lib.ptr(out), lib.stream()), "nir_mt_conv3_fwd")
        ctx.save_for_backward(Tc, ws[0], ws[2], ws[4], out)
        return out

    @staticmethod
    def backward(ctx, dout):
        Tc, w1, w2, w3, out = ctx.saved_tensors
        L = lib.load()
        M, C1, H, W = Tc.shape
        NF = w1.shape[0]
        d = _f32c(dout)
        dpre = torch.empty_like(d)
        lib.check(L.nir_act_bwd_f32(lib.ptr(d), lib.ptr(out), lib.ptr(dpre), d.numel(), 2, lib.stream()), "nir_act_bwd_f32")
        dev = d.device
        need_w = any(ctx.needs_input_grad[1:])
        dT = torch.empty_like(Tc) if ctx.needs_input_grad[0] else None
        wt = torch.empty(L.nir_mt_conv3_wt_floats(C1, NF), device=dev) if dT is not None else None
        NW = NF * C1 * 45
        part = torch.empty(L.nir_mt_conv3_partial_floats(M, C1, NF) // NW, NW, device=dev) if need_w else None
        lib.check(L.nir_mt_conv3_bwd(lib.ptr(dpre), lib.ptr(Tc), lib.ptr(w1), lib.ptr(w2), lib.ptr(w3), M, C1, H, W, NF, lib.ptr(dT), lib.ptr(wt), lib.ptr(part),
                                     lib.stream()), "nir_mt_conv3_bwd")
        grads = [None] * 6
        if need_w:
            dw = _colsum(part, NW, part.shape[0], NW)
            db = _colsum(dpre, 3 * NF, dpre.shape[0], 3 * NF)
            o = 0
            for g, w in enumerate((w1, w2, w3)):
                n = w.numel()
                grads[2 * g] = dw[o:o + n].view_as(w)
                grads[2 * g + 1] = db[g * NF:(g + 1) * NF]
                o += n
        return (dT,) + tuple(grads)


def mt_conv3_supported(T, conv1, conv2, conv3):
    """the three MatchTensor convolutions in their reference shapes ((3,3)/(3,5)/(3,7), 'same' padding, NF = 6 filters over 51 channels)?"""
    ks = [(3, 3), (3, 5), (3, 7)]
    ok = all(tuple(c.kernel_size) == k and tuple(c.padding) == (1, k[1] // 2) and tuple(c.stride) == (1, 1) and c.bias is not None
             for c, k in zip((conv1, conv2, conv3), ks))
    ok = ok and conv1.out_channels == conv2.out_channels == conv3.out_channels and T.is_cuda
    return bool(ok and lib.load().nir_mt_conv3_supported(conv1.out_channels, T.shape[1], T.shape[2], T.shape[3]))


def mt_conv3(T, conv1, conv2, conv3):
    return _MTConv3.apply(T, conv1.weight, conv1.bias, conv2.weight, conv2.bias, conv3.weight, conv3.bias)


def id_flag(device):
    """device int32 flag the train-mode lookups set for an id outside [0, V) (the reference's nn.Embedding raises IndexError)"""
    _ID_FLAGS[(device.type, device.index)] = device
    return lib.flags(device).dev


def check_ids():
    """Synchronising: raise IndexError / RuntimeError if a train-mode operator since the last check set an error bit (the one word of the
    device, lib.Flags: an id outside the vocabulary; weights outside the fp16 range of the split recurrence -- SPLIT_TRAIN_FWD = False
    selects the fp32 one --; a recurrence cluster that timed out -- CLUSTER_TRAIN_FWD = False selects the step-by-step recurrence)."""
    for device in list(_ID_FLAGS.values()):
        lib.flags(device).check()


class _SuggestLossRows(Function):
    """logits [R,V], target [R] -> (nll [R], ent [R]) of nir_softmax_nll_ent_fwd; the backward writes dlogits in one pass."""

    @staticmethod
    def forward(ctx, logits, target, pad):
        lib.require_device(logits)
        L = lib.load()
        z = _f32c(logits)
        R, V = z.shape
        t = lib.ids64(target).reshape(R).contiguous()
        nll = torch.empty(R, device=z.device)
        ent = torch.empty(R, device=z.device)
        lse = torch.empty(R, device=z.device)
        lib.check(L.nir_softmax_nll_ent_fwd(lib.ptr(z), V, lib.ptr(t), int(pad), R, V, lib.ptr(nll), lib.ptr(ent), lib.ptr(lse), lib.ptr(id_flag(z.device)),
                                            lib.stream()), "nir_softmax_nll_ent_fwd")
        ctx.save_for_backward(z, t, lse, ent)
        ctx.pad = int(pad)
        return nll, ent

    @staticmethod
    def backward(ctx, gnll, gent):
        z, t, lse, ent = ctx.saved_tensors
        R, V = z.shape
        gn = _f32c(gnll) if gnll is not None else torch.zeros(R, device=z.device)
        ge = _f32c(gent) if gent is not None else None
        dz = torch.empty_like(z)
        CH = 65535                      # rows per launch (grid.y of the kernel): larger batches run in row chunks instead of failing mid-backward
        for r0 in range(0, R, CH):
            r1 = min(R, r0 + CH)
            lib.check(lib.load().nir_softmax_nll_ent_bwd(lib.ptr(z[r0:r1]), V, lib.ptr(t[r0:r1]), ctx.pad, lib.ptr(lse[r0:r1]), lib.ptr(ent[r0:r1]), lib.ptr(gn[r0:r1]),
                                                         lib.ptr(ge[r0:r1]) if ge is not None else None, r1 - r0, V, lib.ptr(dz[r0:r1]), lib.stream()),
                      "nir_softmax_nll_ent_bwd")
        return dz, None, None


def suggestion_loss(logits, target, pad, regularize_coeff=0.0):
    """The multitask models' suggestion loss (multitask.py:203-216) from decoder logits [Bd, L, V] and targets [Bd, L]:
    mean over Bd of sum_L( -log_softmax(logits)[target] * (target != pad) ) (+ regularize_coeff * sum_L sum_V p log p)."""
    Bd, TL, V = logits.shape
    nll, ent = _SuggestLossRows.apply(logits.reshape(Bd * TL, V), target.reshape(-1), pad)
    loss = nll.view(Bd, TL).sum(1).mean()
    if regularize_coeff > 0:
        loss = loss + (ent.view(Bd, TL).sum(1) * regularize_coeff).mean()
    return loss


class _SoftmaxPool(Function):
    """logits [R,T], mask (bool [MR,T] or None; row of r = (r // mdiv) % MR), values [R/G,T,D] -> out [R,D] = softmax(masked logits) @ values."""

    @staticmethod
    def forward(ctx, logits, mask, mdiv, values, G):
        lib.require_device(logits, values)
        L = lib.load()
        z, v = _f32c(logits), _f32c(values)
        R, T = z.shape
        D = v.shape[2]
        mk = mask.to(torch.uint8).contiguous() if mask is not None else None
        w = torch.empty(R, T, device=z.device)
        out = torch.empty(R, D, device=z.device)
        lib.check(L.nir_softmax_pool_fwd(lib.ptr(z), lib.ptr(mk), int(mdiv), mk.shape[0] if mk is not None else 1, lib.ptr(v), R, int(G), T, D, lib.ptr(w),
                                         lib.ptr(out), lib.stream()), "nir_softmax_pool_fwd")
        ctx.save_for_backward(w, v)
        ctx.G = int(G)
        return out

    @staticmethod
    def backward(ctx, dout):
        w, v = ctx.saved_tensors
        R, T = w.shape
        D = v.shape[2]
        d = _f32c(dout)
        dz = torch.empty_like(w)
        dv = torch.empty_like(v) if ctx.needs_input_grad[3] else None
        lib.check(lib.load().nir_softmax_pool_bwd(lib.ptr(w), lib.ptr(d), lib.ptr(v), R, ctx.G, T, D, lib.ptr(dz), lib.ptr(dv), lib.stream()),
                  "nir_softmax_pool_bwd")
        return dz, None, None, dv, None


def softmax_pool(logits, mask, values, mask_div=1):
    """softmax(logits.masked_fill(~mask, -inf), -1) @ values in one launch each way.  logits [..., T]; values [V0, T, D] with V0 dividing the number
    of logit rows R (G = R / V0 consecutive rows share a value block); mask bool [MR, T] or None, row of r = (r // mask_div) % MR.
    -> [R, D]"""
    T = logits.shape[-1]
    z = logits.reshape(-1, T)
    R = z.shape[0]
    if R == 0 or values.shape[0] == 0:                       # empty batch: nothing to launch (differentiable zeros of the right shape)
        return z.sum(1, keepdim=True) * values.sum((0, 1)).unsqueeze(0)
    G = R // values.shape[0]
    if R % max(1, values.shape[0]):
        raise RuntimeError("softmax_pool: %d logit rows are not a multiple of the %d value blocks" % (R, values.shape[0]))
    if G * T * 8 > 64 * 1024 or T > 8192:
        # outside the kernel's LDS budget (G T <= 8192: target_len x query_len of the CARS decoder attention, S x S of the causal session
        # attention): the same expression as tensor glue, differentiable through torch -- any shape the reference accepts still runs
        zz = z
        if mask is not None:
            rows = (torch.arange(R, device=z.device) // int(mask_div)) % mask.shape[0]
            zz = z.masked_fill(~mask.bool()[rows], float("-inf"))
        w = torch.softmax(zz, -1)
        return torch.bmm(w.view(values.shape[0], G, T), values.float()).reshape(R, values.shape[2])
    return _SoftmaxPool.apply(z, mask, mask_div, values, G)


class _Embed(Function):
    @staticmethod
    def forward(ctx, ids, table, pad_idx):
        lib.require_device(ids, table)
        L = lib.load()
        flat = lib.ids64(ids).reshape(-1)
        V, E = table.shape
        out = torch.empty(flat.numel(), E, device=table.device, dtype=torch.float32)
        tc = _f32c(table)
        lib.check(L.nir_embed_f32(lib.ptr(flat), lib.ptr(tc), V, E, flat.numel(), lib.ptr(out), lib.ptr(id_flag(table.device)), lib.stream()),
                  "nir_embed_f32")
        ctx.save_for_backward(flat)
        ctx.dims, ctx.pad = (V, E), pad_idx
        return out.view(*ids.shape, E)

    @staticmethod
    def backward(ctx, dout):
        if not ctx.needs_input_grad[1]:
            return None, None, None
        (flat,) = ctx.saved_tensors
        V, E = ctx.dims
        dt = torch.zeros(V, E, device=dout.device, dtype=torch.float32)
        d = _f32c(dout).reshape(-1, E)
        lib.check(lib.load().nir_embed_bwd_f32(lib.ptr(flat), lib.ptr(d), V, E, flat.numel(), lib.ptr(dt), ctx.pad, lib.stream()),
                  "nir_embed_bwd_f32")
        return None, dt, None


def embed(ids, table, pad_idx=0):
    return _Embed.apply(ids, table, pad_idx)


class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        L = lib.load()
        xc = _f32c(x)
        y = torch.empty_like(xc)
        keep = torch.empty(xc.numel(), dtype=torch.uint8, device=xc.device)
        if DROPOUT.device_seed is not None:       # captured training step: the seed lives on the device, `seed` is the call site's salt
            lib.check(L.nir_dropout_dev_f32(lib.ptr(xc), lib.ptr(y), lib.ptr(keep), xc.numel(), float(p), lib.ptr(DROPOUT.device_seed), int(seed),
                                            lib.stream()), "nir_dropout_dev_f32")
        else:
            lib.check(L.nir_dropout_f32(lib.ptr(xc), lib.ptr(y), lib.ptr(keep), xc.numel(), float(p), int(seed), lib.stream()), "nir_dropout_f32")
        ctx.p = float(p)
        ctx.save_for_backward(keep)
        ctx.mark_non_differentiable(keep)
        return y, keep

    @staticmethod
    def backward(ctx, dy, _dkeep):
        (keep,) = ctx.saved_tensors
        d = _f32c(dy)
        dx = torch.empty_like(d)
        lib.check(lib.load().nir_mask_scale_f32(lib.ptr(d), lib.ptr(keep), 1.0 / (1.0 - ctx.p), lib.ptr(dx), d.numel(), lib.stream()),
                  "nir_mask_scale_f32")
        return dx, None, None


class DropoutState(object):
    """Seed stream of the training step: every dropout site draws the next 64-bit seed (reproducible under `manual_seed`);
    `masks` records the keep masks of the last forward so a parity test can replay them through the oracle."""

    def __init__(self, seed=1013):
        self.seed, self.counter, self.masks, self.record = int(seed), 0, [], False
        self.device_seed = None         # int64 [1] device tensor while a training step is being captured / replayed (wrappers.GraphedUpdate)
        self.site = 0

    def manual_seed(self, seed):
        self.seed, self.counter = int(seed), 0

    def next(self):
        if self.device_seed is not None:   # the per-step seed comes from the device; sites are numbered within the step
            self.site += 1
            return self.site
        self.counter += 1
        return (self.seed * 0x9E3779B97F4A7C15 + self.counter * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


DROPOUT = DropoutState()


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    y, keep = _Dropout.apply(x, p, DROPOUT.next())
    if DROPOUT.record:
        DROPOUT.masks.append(keep.view(x.shape))
    return y


_ROW_IDS = {}
_ROW_IDS_RETIRED = []


def _row_ids(n, dev):
    """0 .. n-1 (int64) on the device: the "token ids" of a per-batch gate tensor read by the folded-table recurrence.  ONE buffer per device
    that only grows (a prefix of a longer 0 .. N-1 is the same list); a superseded buffer is retired, never freed -- a hipGraph captured by
    GraphedUpdate holds its address (the old per-size cache dropped its entries above 16 sizes and replays then read recycled memory)."""
    key = str(dev)
    t = _ROW_IDS.get(key)
    if t is None or t.numel() < n:
        if t is not None:
            _ROW_IDS_RETIRED.append(t)
        t = _ROW_IDS[key] = torch.arange(max(int(n), 2 * (t.numel() if t is not None else 0), 1 << 16), device=dev, dtype=torch.int64)
    return t[:n]


class _BiLSTM(Function):
    """x [M,T,I], lens [M] (or None), optional initial state h0/c0 [ND,M,H] and the nn.LSTM parameters (per direction: w_ih
    [4H,I], w_hh [4H,H], b_ih, b_hh) -> (memory bank [M,T,ND*H], zero at t >= length; cell states [M,T,ND,H])."""

    @staticmethod
    def forward(ctx, x, lens, nd, h0, c0, *params):
        lib.require_device(x)
        L = lib.load()
        M, T, I = x.shape
        wih = torch.cat([params[4 * d] for d in range(nd)], 0).float().contiguous()
        whh = torch.stack([params[4 * d + 1] for d in range(nd)], 0).float().contiguous()
        H = whh.shape[2]
        if H > 128:
            raise NotImplementedError("train-mode recurrence (nir_lstm_train_fwd / _bwd) supports H <= 128 per direction (got %d); "
                                      "wider single-direction LSTMs go through lstm_seq" % H)
        if h0 is not None and nd != 1:
            raise NotImplementedError("initial states are supported for one direction only (the backward's h_{t-1} of the reverse "
                                      "direction at t = length-1 would have to be h0[1])")
        x2 = _f32c(x).reshape(M * T, I)
        dev = x.device
        out = torch.empty(M, T, nd * H, device=dev)
        act = torch.empty(M, T, nd, 4 * H, device=dev)
        cst = torch.empty(M, T, nd, H, device=dev)
        if lens is not None:
            cst.zero_()          # positions past a sequence's length are never written by the kernel
        lens64 = lib.ids64(lens) if lens is not None else None
        h0c = _f32c(h0) if h0 is not None else None
        c0c = _f32c(c0) if c0 is not None else None
        if SPLIT_TRAIN_FWD and 64 < H <= 128 and h0 is None and M * T >= 16 and T <= 512:
            # the fp32-accurate split-fp16 recurrence of the inference path with train-mode stores (3 fp16 MFMAs per k-block for 32 fp32 ones):
            # the input GEMM writes the gates in the folded order [row][dir][unit][gate] (weights permuted once per step), rows are their own ids
            ps = [_f32c(t) for t in params]
            wperm = torch.empty(nd * 4 * H, I, device=dev)
            bperm = torch.empty(nd * 4 * H, device=dev)
            rev = ps[4:7] if nd == 2 else [None, None, None]
            lib.check(L.nir_lstm_perm_weights(lib.ptr(ps[0]), lib.ptr(ps[2]), lib.ptr(ps[3]), lib.ptr(rev[0]), lib.ptr(rev[2]) if nd == 2 else None,
                                              lib.ptr(ps[7]) if nd == 2 else None, H, nd, I, lib.ptr(wperm), lib.ptr(bperm), lib.stream()),
                      "nir_lstm_perm_weights")
            gates = _linear_raw(x2, wperm, bperm, 0)
            lib.check(L.nir_lstm_train_fwd_split(lib.ptr(gates), lib.ptr(_row_ids(M * T, dev)), lib.ptr(lens64), lib.ptr(whh), lib.ptr(out), lib.ptr(act),
                                                 lib.ptr(cst), lib.ptr(id_flag(dev)), M, T, H, nd, lib.stream()), "nir_lstm_train_fwd_split")
        else:
            bias = torch.cat([params[4 * d + 2] + params[4 * d + 3] for d in range(nd)], 0).float().contiguous()
            gates = _linear_raw(x2, wih, bias, 0)
            lib.check(L.nir_lstm_train_fwd(lib.ptr(gates), lib.ptr(lens64), lib.ptr(whh), lib.ptr(h0c), lib.ptr(c0c), lib.ptr(out), lib.ptr(act),
                                           lib.ptr(cst), None, None, M, T, H, nd, lib.stream()), "nir_lstm_train_fwd")
        ctx.nd, ctx.dims = nd, (M, T, I, H)
        e = torch.empty(0)
        ctx.save_for_backward(x2, lens64 if lens64 is not None else e, wih, whh, out, act, cst, h0c if h0c is not None else e,
                              c0c if c0c is not None else e)
        ctx.has_lens, ctx.has_init = lens64 is not None, h0c is not None
        return out, cst

    @staticmethod
    def backward(ctx, dout, dcst):
        x2, lens64, wih, whh, out, act, cst, h0, c0 = ctx.saved_tensors
        L = lib.load()
        nd = ctx.nd
        M, T, I, H = ctx.dims
        G = nd * 4 * H
        d = _f32c(dout) if dout is not None else torch.zeros(M, T, nd * H, device=x2.device)
        dc = _f32c(dcst) if dcst is not None else None
        dgates = torch.empty(M, T, G, device=d.device)
        dh0 = torch.empty(nd, M, H, device=d.device) if ctx.has_init else None
        dc0 = torch.empty(nd, M, H, device=d.device) if ctx.has_init else None
        lib.check(L.nir_lstm_train_bwd(lib.ptr(d), None, None, lib.ptr(dc), lib.ptr(act), lib.ptr(cst), lib.ptr(c0) if ctx.has_init else None,
                                       lib.ptr(lens64) if ctx.has_lens else None, lib.ptr(whh), lib.ptr(dgates), lib.ptr(dh0), lib.ptr(dc0),
                                       M, T, H, nd, lib.stream()), "nir_lstm_train_bwd")
        dg2 = dgates.view(M * T, G)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _linear_raw(dg2, _transpose(wih), None, 0).view(M, T, I)
        if not ctx.has_init:
            # dW_hh straight from the saved states: h of the PREVIOUS recurrence step is the row before (forward direction) / after (reverse
            # direction; `out` is zero at t >= length = the reverse direction's zero initial state) the gate row, and the first step of a
            # sequence has none (period T, skip 0 / T-1) -- no shifted [M,T,H] copy of the states per direction.  PACKED_WGRAD: all three
            # reductions over a device-built list of the valid positions only (pays for batches of short sequences, costs ~18 % on full ones)
            dev = d.device
            rows = cnt = [None, None]
            if PACKED_WGRAD and ctx.has_lens and M * T < 2 ** 31:
                offs = torch.empty(2, M + 1, device=dev, dtype=torch.int32)
                rows = torch.empty(2, M * T, device=dev, dtype=torch.int32)
                lib.check(L.nir_seq_rows(lib.ptr(lens64), M, T, 0, lib.ptr(offs[0]), lib.ptr(rows[0]), lib.stream()), "nir_seq_rows")
                cnt = [offs[0, M:], offs[0, M:]]
                rows = [rows[0], rows[0]]
                dwih = torch.empty(G, I, device=dev)
                db = torch.empty(G, device=dev)
                lib.check(L.nir_linear_wgrad_rows_set_f32(lib.ptr(dg2), G, 0, lib.ptr(x2), I, 0, lib.ptr(rows[0]), lib.ptr(cnt[0]), M * T, 0, 0,
                                                          lib.ptr(dwih), I, lib.ptr(db), G, I, lib.stream()), "nir_linear_wgrad_rows_set_f32")
            else:
                dwih, db = _wgrad_bias(dg2, G, x2, I, M * T, G, I)
            grads = []
            o2 = out.view(M * T, nd * H)
            for dd in range(nd):
                dwhh = torch.empty(4 * H, H, device=dev)
                lib.check(L.nir_linear_wgrad_rows_set_f32(lib.C.c_void_p(dg2.data_ptr() + dd * 4 * H * 4), G, 0, lib.C.c_void_p(o2.data_ptr() + dd * H * 4),
                                                          nd * H, -1 if dd == 0 else 1, lib.ptr(rows[dd]), lib.ptr(cnt[dd]), M * T, T,
                                                          0 if dd == 0 else T - 1, lib.ptr(dwhh), H, None, 4 * H, H, lib.stream()),
                          "nir_linear_wgrad_rows_set_f32")
                sl = slice(dd * 4 * H, (dd + 1) * 4 * H)
                grads += [dwih[sl], dwhh, db[sl], db[sl].clone()]
            return (dx, None, None, dh0, dc0) + tuple(grads)
        dwih, db = _wgrad_bias(dg2, G, x2, I, M * T, G, I)
        grads = []
        for dd in range(nd):
            # h of the PREVIOUS recurrence step: forward direction t-1 (h0 at t = 0), reverse direction t+1 (out is zero at
            # t >= length, which is exactly the zero initial state of the reverse direction at t = length-1)
            hd = out[:, :, dd * H:(dd + 1) * H]
            z = h0[dd].unsqueeze(1) if (ctx.has_init and dd == 0) else torch.zeros(M, 1, H, device=d.device)
            hprev = (torch.cat([z, hd[:, :-1]], 1) if dd == 0 else torch.cat([hd[:, 1:], torch.zeros(M, 1, H, device=d.device)], 1))
            hprev = hprev.contiguous().view(M * T, H)
            dwhh = torch.zeros(4 * H, H, device=d.device)
            lib.check(L.nir_linear_wgrad_f32(lib.C.c_void_p(dg2.data_ptr() + dd * 4 * H * 4), G, lib.ptr(hprev), H, None, None, 0,
                                             lib.ptr(dwhh), H, M * T, 4 * H, H, lib.stream()), "nir_linear_wgrad_f32")
            s = slice(dd * 4 * H, (dd + 1) * 4 * H)
            grads += [dwih[s], dwhh, db[s], db[s].clone()]
        return (dx, None, None, dh0, dc0) + tuple(grads)



_SIDE = {}


def _side_stream(dev):
    s = _SIDE.get(str(dev))
    if s is None:
        s = _SIDE[str(dev)] = torch.cuda.Stream(device=dev)
    return s



class _BiLSTM256(Function):
    """Train-mode encoder with 256 units per direction: x [M,T,I], lens [M] -> memory bank [M,T,ND*256] (zero past each length).  Forward = ONE launch
    of the four-workgroup cluster recurrence of the predict path with activation / cell-state stores (nir_lstm256_train_fwd; gates in the folded
    order from one GEMM); backward = BPTT with ONE launch per step for both directions (nir_lstm256_bptt: the step's gate gradients + the recurrent
    product as eight K-slice partials on the fp32 matrix cores; FUSED_BPTT256 = False: the masked cell kernel + one GEMM per step and direction of
    rounds 3-5), weight gradients from the saved states shifted by a row."""

    @staticmethod
    def forward(ctx, x, lens, nd, *params):
        lib.require_device(x)
        L = lib.load()
        M, T, I = x.shape
        H = 256
        dev = x.device
        ps = [_f32c(t) for t in params]
        x2 = _f32c(x).reshape(M * T, I)
        wperm = torch.empty(nd * 4 * H, I, device=dev)
        bperm = torch.empty(nd * 4 * H, device=dev)
        lib.check(L.nir_lstm_perm_weights(lib.ptr(ps[0]), lib.ptr(ps[2]), lib.ptr(ps[3]), lib.ptr(ps[4]) if nd == 2 else None,
                                          lib.ptr(ps[6]) if nd == 2 else None, lib.ptr(ps[7]) if nd == 2 else None, H, nd, I, lib.ptr(wperm), lib.ptr(bperm),
                                          lib.stream()), "nir_lstm_perm_weights")
        gates = _linear_raw(x2, wperm, bperm, 0)
        whh = torch.stack([ps[4 * d + 1] for d in range(nd)], 0).contiguous()
        frag = torch.empty(L.nir_lstm256_whh_frag_bytes(nd), dtype=torch.uint8, device=dev)
        flag = id_flag(dev)
        lib.check(L.nir_lstm256_pack_whh_frag(lib.ptr(whh), nd, lib.ptr(frag), lib.ptr(flag), lib.stream()), "nir_lstm256_pack_whh_frag")
        # exchange scratch of the cluster recurrence: the library pool (per device and stream -- or owned by the capturing GraphedUpdate /
        # predictor --, grow-only, superseded buffers retired, never freed: a captured hipGraph holds the address)
        ws = lib.workspace(L.nir_lstm256_workspace_bytes(M, nd), dev)
        lens64 = lib.ids64(lens)
        out = torch.empty(M, T, nd * H, device=dev)
        act = torch.empty(M, T, nd, 4 * H, device=dev)
        cst = torch.empty(M, T, nd, H, device=dev)
        lib.check(L.nir_lstm256_train_fwd(lib.ptr(gates), lib.ptr(lens64), lib.ptr(frag), lib.ptr(out), lib.ptr(act), lib.ptr(cst), lib.ptr(flag), M, T, nd,
                                          lib.ptr(ws), ws.numel(), lib.stream()), "nir_lstm256_train_fwd")
        wih = torch.cat([ps[4 * d] for d in range(nd)], 0) if ctx.needs_input_grad[0] else torch.empty(0)
        ctx.save_for_backward(x2, lens64, wih, whh, out, act, cst)
        ctx.nd, ctx.dims = nd, (M, T, I)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, lens64, wih, whh, out, act, cst = ctx.saved_tensors
        L = lib.load()
        nd = ctx.nd
        M, T, I = ctx.dims
        H, G = 256, nd * 1024
        dev = x2.device
        st = lib.stream()
        d = _f32c(dout)
        dgx = torch.empty(M, T, G, device=dev)
        if FUSED_BPTT256:
            ws = torch.empty(L.nir_lstm256_bptt_workspace_bytes(M, nd), dtype=torch.uint8, device=dev)
            lib.check(L.nir_lstm256_bptt(lib.ptr(d), lib.ptr(act), lib.ptr(cst), lib.ptr(lens64), lib.ptr(whh), lib.ptr(dgx), M, T, nd, lib.ptr(ws),
                                         ws.numel(), st), "nir_lstm256_bptt")
            return _BiLSTM256._param_grads(ctx, dgx)
        keep = []
        # the two directions are independent chains of 2 T small launches each (a [M,256] x [256,1024] GEMM is latency-bound): the reverse
        # direction runs on a side stream (forked / joined through events: inside a captured step two branches of the graph).  Everything both
        # streams touch (dgx, the saved tensors) was allocated before the fork; the side stream's temporaries live and die there.
        main = torch.cuda.current_stream()
        side = _side_stream(dev) if (nd == 2 and TWO_STREAM_BPTT) else None
        wts = [_transpose(whh[dd]) for dd in range(nd)]          # [H, 4H]: dh_prev = dg W_hh
        if side is not None:
            side.wait_stream(main)
        for dd in range(nd):
            with torch.cuda.stream(side if (side is not None and dd == 1) else main):
                st = lib.stream()
                wt = wts[dd]
                dh_rec = dc_rec = None
                for t in (range(T - 1, -1, -1) if dd == 0 else range(T)):
                    tprev = t - 1 if dd == 0 else t + 1               # position of the previous recurrence step
                    cp = _off(cst, ((tprev * nd + dd) * H) * 4) if 0 <= tprev < T else None
                    dcn = torch.empty(M, H, device=dev)
                    lib.check(L.nir_lstm_cell_seq_bwd_masked(_off(d, (t * nd * H + dd * H) * 4), T * nd * H, lib.ptr(dh_rec), lib.ptr(dc_rec),
                                                             _off(act, ((t * nd + dd) * 4 * H) * 4), T * G, _off(cst, ((t * nd + dd) * H) * 4), T * nd * H,
                                                             cp, T * nd * H, _off(dgx, (t * G + dd * 4 * H) * 4), T * G, lib.ptr(dcn), lib.ptr(lens64), t,
                                                             tprev, M, H, st), "nir_lstm_cell_seq_bwd_masked")
                    keep.append((dh_rec, dc_rec))
                    dc_rec = dcn
                    if t != (0 if dd == 0 else T - 1):
                        dh_rec = torch.empty(M, H, device=dev)
                        lib.check(L.nir_linear_f32(_off(dgx, (t * G + dd * 4 * H) * 4), T * G, None, None, 0, 0, 0, lib.ptr(wt), 4 * H, None, None,
                                                   lib.ptr(dh_rec), H, M, H, 4 * H, 0, st), "nir_linear_f32")
        if side is not None:
            main.wait_stream(side)
        return _BiLSTM256._param_grads(ctx, dgx)

    @staticmethod
    def _param_grads(ctx, dgx):
        x2, lens64, wih, whh, out, act, cst = ctx.saved_tensors
        L = lib.load()
        nd = ctx.nd
        M, T, I = ctx.dims
        H, G = 256, nd * 1024
        dev = x2.device
        st = lib.stream()
        dg2 = dgx.view(M * T, G)
        dx = _linear_raw(dg2, _transpose(wih), None, 0).view(M, T, I) if ctx.needs_input_grad[0] else None
        dwih, db = _wgrad_bias(dg2, G, x2, I, M * T, G, I)
        o2 = out.view(M * T, nd * H)
        grads = []
        for dd in range(nd):
            dwhh = torch.empty(4 * H, H, device=dev)
            lib.check(L.nir_linear_wgrad_rows_set_f32(_off(dg2, dd * 4 * H * 4), G, 0, _off(o2, dd * H * 4), nd * H, -1 if dd == 0 else 1, None, None, M * T, T,
                                                      0 if dd == 0 else T - 1, lib.ptr(dwhh), H, None, 4 * H, H, st), "nir_linear_wgrad_rows_set_f32")
            sl = slice(dd * 4 * H, (dd + 1) * 4 * H)
            grads += [dwih[sl], dwhh, db[sl], db[sl].clone()]
        return (dx, None, None) + tuple(grads)


def _lstm_params(lstm):
    sfx = ["", "_reverse"] if lstm.bidirectional else [""]
    params = []
    for s in sfx:
        params += [getattr(lstm, "weight_ih_l0" + s), getattr(lstm, "weight_hh_l0" + s), getattr(lstm, "bias_ih_l0" + s),
                   getattr(lstm, "bias_hh_l0" + s)]
    return len(sfx), params


def bilstm(x, lens, lstm):
    """RNNEncoder body in train mode for an nn.LSTM(1 layer, batch_first) parameter container -> memory bank [M,T,ND*H], any hidden size:
    the register-resident training recurrence (_BiLSTM) up to H = 128 per direction; beyond it one unidirectional pass of lstm_seq per
    direction -- the reverse direction over each sequence's valid part read backwards (packed-sequence semantics: zero past the length;
    states past a sequence's end never reach a kept output).  Every train-mode caller (MatchTensor, CARS, MNSRF, M_MATCH_TENSOR) goes
    through here, so a wide encoder trains wherever it evaluates."""
    nd, params = _lstm_params(lstm)
    H = lstm.hidden_size
    if H <= 128:
        return _BiLSTM.apply(x, lens, nd, None, None, *params)[0]
    M, T, _ = x.shape
    if H == 256 and CLUSTER_TRAIN_FWD and T <= 1024 and M * T > 0 and M * T < 2 ** 31:
        ln_ = lens if lens is not None else torch.full((M,), T, device=x.device, dtype=torch.int64)
        return _BiLSTM256.apply(x, ln_, nd, *params)
    dev = x.device
    ln = lens.to(dev).view(M, 1) if lens is not None else torch.full((M, 1), T, device=dev, dtype=torch.int64)
    pos = torch.arange(T, device=dev).view(1, T)
    valid = (pos < ln).unsqueeze(2).float()

    class _Dir(object):                                                          # one direction's parameters under the names lstm_seq reads
        def __init__(self, sfx):
            for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
                setattr(self, n, getattr(lstm, n + sfx))
    fwd = lstm_seq(x, _Dir(""))[0] * valid
    if nd == 1:
        return fwd
    ridx = (ln - 1 - pos).clamp(min=0)                                             # position read at reverse step t
    xr = torch.gather(x, 1, ridx.unsqueeze(2).expand(M, T, x.shape[2])) * valid
    rev = lstm_seq(xr, _Dir("_reverse"))[0] * valid
    rev = torch.gather(rev, 1, ridx.unsqueeze(2).expand(M, T, H)) * valid            # back to time order
    return torch.cat((fwd, rev), 2)


class _LSTMCell(Function):
    @staticmethod
    def forward(ctx, gates, c_prev):
        L = lib.load()
        g = _f32c(gates)
        B, H4 = g.shape
        H = H4 // 4
        cp = _f32c(c_prev) if c_prev is not None else None
        act = torch.empty_like(g)
        c = torch.empty(B, H, device=g.device)
        h = torch.empty(B, H, device=g.device)
        lib.check(L.nir_lstm_cell_fwd(lib.ptr(g), lib.ptr(cp), lib.ptr(act), lib.ptr(c), lib.ptr(h), B, H, lib.stream()), "nir_lstm_cell_fwd")
        ctx.save_for_backward(act, c, cp if cp is not None else torch.empty(0))
        ctx.has_cp = cp is not None
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        act, c, cp = ctx.saved_tensors
        B, H = c.shape
        dg = torch.empty_like(act)
        dcp = torch.empty_like(c)
        # the contiguous copies stay referenced until the launch is enqueued: as temporaries inside the call they were freed one by one, and
        # with BOTH gradients arriving as strided slices (torch.stack's backward) the second copy reused the first one's block
        dhc = _f32c(dh) if dh is not None else None
        dcc = _f32c(dc) if dc is not None else None
        lib.check(lib.load().nir_lstm_cell_bwd(lib.ptr(dhc), lib.ptr(dcc), lib.ptr(act), lib.ptr(c), lib.ptr(cp) if ctx.has_cp else None,
                                               lib.ptr(dg), lib.ptr(dcp), B, H, lib.stream()), "nir_lstm_cell_bwd")
        return dg, (dcp if ctx.has_cp else None)


def _off(t, nbytes):
    return lib.C.c_void_p(t.data_ptr() + int(nbytes))


class _LSTMSeq(Function):
    """gx [M,T,4H] (input projection of every step, b_ih included), W_hh [4H,H], b_hh, optional h0 / c0 [M,H] -> (h [M,T,H], c [M,T,H]).
    Per step ONE recurrent GEMM and ONE cell kernel that reads / writes the sequence buffers in place (round 5: the op-by-op form cost a step an
    add, a stack slice and, in the backward, add + fill + copy + add around the same two kernels); BPTT here, dW_hh / db_hh from ONE launch over
    the saved states shifted by a row (nir_linear_wgrad_rows_set_f32)."""

    @staticmethod
    def forward(ctx, gx, whh, bhh, h0, c0):
        lib.require_device(gx, whh)
        L = lib.load()
        g = _f32c(gx)
        M, T, G = g.shape
        H = G // 4
        w, b = _f32c(whh), _f32c(bhh)
        h0c = _f32c(h0) if h0 is not None else None
        c0c = _f32c(c0) if c0 is not None else None
        dev = g.device
        hs = torch.empty(M, T, H, device=dev)
        cs = torch.empty(M, T, H, device=dev)
        act = torch.empty(M, T, G, device=dev)
        gh = torch.empty(M, G, device=dev)
        st = lib.stream()
        for t in range(T if M else 0):                       # (an empty batch launches nothing)
            hp, ldh = (h0c, H) if t == 0 else (_off(hs, (t - 1) * H * 4), T * H)
            have_h = M > 0 and (t > 0 or h0c is not None)
            if have_h:
                lib.check(L.nir_linear_f32(lib.ptr(hp) if t == 0 else hp, ldh, None, None, 0, 0, 0, lib.ptr(w), H, lib.ptr(b), None, lib.ptr(gh), G, M, G, H, 0, st),
                          "nir_linear_f32")
            cp, ldcp = (lib.ptr(c0c), H) if t == 0 else (_off(cs, (t - 1) * H * 4), T * H)
            lib.check(L.nir_lstm_cell_seq_fwd(_off(g, t * G * 4), T * G, lib.ptr(gh) if have_h else None, None if have_h else lib.ptr(b), cp, ldcp,
                                              _off(act, t * G * 4), T * G, _off(cs, t * H * 4), T * H, _off(hs, t * H * 4), T * H, M, H, st),
                      "nir_lstm_cell_seq_fwd")
        e = torch.empty(0)
        ctx.save_for_backward(w, hs, cs, act, h0c if h0c is not None else e, c0c if c0c is not None else e)
        ctx.has_h0, ctx.has_c0 = h0c is not None, c0c is not None
        return hs, cs

    @staticmethod
    def backward(ctx, dhs, dcs):
        w, hs, cs, act, h0, c0 = ctx.saved_tensors
        L = lib.load()
        M, T, H = hs.shape
        G = 4 * H
        dev = hs.device
        st = lib.stream()
        d1 = _f32c(dhs) if dhs is not None else None
        d2 = _f32c(dcs) if dcs is not None else None
        dgx = torch.empty(M, T, G, device=dev)
        wt = _transpose(w)                                     # [H, 4H]: dh_{t-1} = dg_t W_hh
        need_h0 = ctx.has_h0 and ctx.needs_input_grad[3]
        dh_rec = dc_rec = None
        keep = []                                              # (buffers stay referenced until their launches are enqueued)
        for t in range(T - 1 if M else -1, -1, -1):
            cp, ldcp = (_off(cs, (t - 1) * H * 4), T * H) if t > 0 else ((lib.ptr(c0), H) if ctx.has_c0 else (None, 0))
            dcn = torch.empty(M, H, device=dev)
            lib.check(L.nir_lstm_cell_seq_bwd(_off(d1, t * H * 4) if d1 is not None else None, T * H, lib.ptr(dh_rec),
                                              _off(d2, t * H * 4) if d2 is not None else None, T * H, lib.ptr(dc_rec), _off(act, t * G * 4), T * G,
                                              _off(cs, t * H * 4), T * H, cp, ldcp, _off(dgx, t * G * 4), T * G, lib.ptr(dcn), M, H, st),
                      "nir_lstm_cell_seq_bwd")
            keep.append((dh_rec, dc_rec))
            dc_rec = dcn
            if (t > 0 or need_h0) and M > 0:
                dh_rec = torch.empty(M, H, device=dev)
                lib.check(L.nir_linear_f32(_off(dgx, t * G * 4), T * G, None, None, 0, 0, 0, lib.ptr(wt), G, None, None, lib.ptr(dh_rec), H, M, H, G, 0, st),
                          "nir_linear_f32")
            else:
                dh_rec = None
        dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw = torch.empty(G, H, device=dev)
            db = torch.empty(G, device=dev)
            if M * T:
                # dW_hh = sum_(m, t >= 1) dg[m,t]^T h[m,t-1]: the states one row back, the first step of every sequence skipped; db = all rows of dg
                lib.check(L.nir_linear_wgrad_rows_set_f32(lib.ptr(dgx), G, 0, lib.ptr(hs), H, -1, None, None, M * T, T, 0, lib.ptr(dw), H, lib.ptr(db), G, H, st),
                          "nir_linear_wgrad_rows_set_f32")
                if ctx.has_h0:                                 # + dg[:,0]^T h0
                    lib.check(L.nir_linear_wgrad_f32(lib.ptr(dgx), T * G, lib.ptr(h0), H, None, None, 0, lib.ptr(dw), H, M, G, H, st), "nir_linear_wgrad_f32")
            else:
                dw.zero_(); db.zero_()
        if M == 0:
            dh_rec = torch.zeros(0, H, device=dev) if need_h0 else None
            dc_rec = torch.zeros(0, H, device=dev)
        return dgx, dw, db, (dh_rec if need_h0 else None), (dc_rec if ctx.has_c0 and ctx.needs_input_grad[4] else None)


def lstm_gx(gx, lstm, h0=None, c0=None):
    """one direction of an nn.LSTM parameter container over precomputed input gates gx [M,T,4H] (x W_ih^T + b_ih) -> (h, c) of every step"""
    return _LSTMSeq.apply(gx, lstm.weight_hh_l0, lstm.bias_hh_l0, h0, c0)


def lstm_seq(x, lstm, h0=None, c0=None):
    """Unidirectional LSTM over full-length sequences x [M,T,I] with an optional initial state ([M,H] each) -> (h of every step
    [M,T,H], c of every step [M,T,H]), any hidden size: the input projection of all steps is one GEMM, each step is the recurrent GEMM and
    the cell kernel working inside the sequence buffers (_LSTMSeq; T is a session or a query: <= ~20)."""
    return lstm_gx(linear(x, lstm.weight_ih_l0, lstm.bias_ih_l0), lstm, h0, c0)


class _BCE(Function):
    @staticmethod
    def forward(ctx, scores, labels):
        L = lib.load()
        s, y = _f32c(scores), _f32c(labels)
        loss = torch.empty(1, device=s.device)
        n = s.shape[-1]
        lib.check(L.nir_rank_loss_bce(lib.ptr(s), lib.ptr(y), s.numel() // n, n, lib.ptr(loss), lib.stream()), "nir_rank_loss_bce")
        ctx.save_for_backward(s, y)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        s, y = ctx.saved_tensors
        ds = torch.empty_like(s)
        gg = _f32c(g).reshape(1)
        lib.check(lib.load().nir_rank_loss_bce_bwd(lib.ptr(s), lib.ptr(y), lib.ptr(gg), lib.ptr(ds), s.numel(), lib.stream()),
                  "nir_rank_loss_bce_bwd")
        return ds, None


def bce_with_logits(scores, labels):
    """mean BCE-with-logits over all entries (models/ranker.py:55-69, multitask/cars.py:603)."""
    return _BCE.apply(scores, labels)
