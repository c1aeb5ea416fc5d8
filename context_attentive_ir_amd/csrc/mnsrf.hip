// MNSRF ranking side (neuroir/multitask/mnsrf.py:62-162): 256-per-direction BiLSTM encoders with max pooling over time, a
// 1024-unit session LSTM over the queries of a session, tanh projection of [query ; session state] and a dot product
// with every candidate document.
//
// Its hidden sizes are beyond the register-resident recurrences of lstm.hip / lstm_mfma.hip (W_hh of one direction is
// 1 MB at H = 256, 16 MB at H = 1024), so the recurrence here is the streaming form: per time step ONE GEMM
// h_{t-1} W_hh^T on the matrix cores (gemm.hip; W_hh is re-read from L2/HBM every step, which is what bounds it) and
// ONE cell kernel that adds the pre-computed input part of the gates (the embedding gather-GEMM), applies the
// packed-sequence semantics (per-sequence length mask, reverse direction starting at len-1) and writes h_t.
// Any H; used by nir_bilstm_steps_fwd (RNNEncoder for H > 128) and by the MNSRF entry points below.
#include "common.hpp"
#include <algorithm>

namespace nir {

int launch_linear(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                  int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                  int64_t ldc, int64_t M, int N, int K, int act, hipStream_t st);
// csrc/lstm_cluster.hip: the resident-weight recurrence for 256 units per direction (four-workgroup clusters)
size_t lstm256_xbuf_bytes(int64_t M, int ND);
int launch_fold_permute(const float* w_ih, const float* b_ih, const float* b_hh, int H, int ndir, int E, float* wperm, float* bperm, int64_t* iota, int64_t niota,
                        hipStream_t st);
int launch_lstm256_cluster(const float* rows, const int64_t* ids, const int64_t* lens, const void* wfrag, float* out, int mode, int* err,
                           int64_t M, int64_t R, int T, int ND, void* xbuf, size_t xbuf_bytes, hipStream_t st);

// hw [M,4H] = h_{t-1} W_hh^T (null at the first step of a zero initial state); gates = hw + gin[m][t_m][dir]
__global__ __launch_bounds__(256) void lstm_step_cell_kernel(const float* __restrict__ hw, const float* __restrict__ gin,
                                                             const int64_t* __restrict__ lens, float* __restrict__ hstate,
                                                             float* __restrict__ cstate, float* __restrict__ out, float* __restrict__ cst,
                                                             int64_t M, int T, int H, int ND, int dir, int step) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * H) return;
    const int64_t m = i / H;
    const int j = (int)(i - m * H);
    int len = lens ? (int)lens[m] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    if (step >= len) return;                                   // finished sequence: state carried, output stays zero
    const int t = dir == 0 ? step : len - 1 - step;
    const float* g = gin + ((m * T + t) * ND + dir) * 4 * (int64_t)H;
    float gi = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
    if (hw) {
        const float* r = hw + m * 4 * (int64_t)H;
        gi += r[j]; gf += r[H + j]; gg += r[2 * H + j]; go += r[3 * H + j];
    }
    const float c = fast_sigmoid(gf) * cstate[i] + fast_sigmoid(gi) * fast_tanh(gg);
    const float h = fast_sigmoid(go) * fast_tanh(c);
    cstate[i] = c;
    hstate[i] = h;
    out[(m * T + t) * ND * (int64_t)H + (int64_t)dir * H + j] = h;
    if (cst) cst[(m * T + t) * ND * (int64_t)H + (int64_t)dir * H + j] = c;      // cell state of every step (decoder initialisation)
}

// GRU step (torch.nn.GRU gate order r, z, n): gin = x W_ih^T + b_ih; hw = h_{t-1} W_hh^T + b_hh (null at the first step of a zero state:
// then hw = b_hh);  r = s(gin_r + hw_r), z = s(gin_z + hw_z), n = tanh(gin_n + r hw_n), h = (1 - z) n + z h_{t-1}.
__global__ __launch_bounds__(256) void gru_step_cell_kernel(const float* __restrict__ hw, const float* __restrict__ bhh, const float* __restrict__ gin,
                                                            const int64_t* __restrict__ lens, float* __restrict__ hstate, float* __restrict__ out,
                                                            int64_t M, int T, int H, int ND, int dir, int step) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * H) return;
    const int64_t m = i / H;
    const int j = (int)(i - m * H);
    int len = lens ? (int)lens[m] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    if (step >= len) return;
    const int t = dir == 0 ? step : len - 1 - step;
    const float* g = gin + ((m * T + t) * ND + dir) * 3 * (int64_t)H;
    const float* r = hw ? hw + m * 3 * (int64_t)H : bhh;
    const float gr = fast_sigmoid(g[j] + r[j]);
    const float gz = fast_sigmoid(g[H + j] + r[H + j]);
    const float gn = fast_tanh(g[2 * H + j] + gr * r[2 * H + j]);
    const float h = (1.0f - gz) * gn + gz * hstate[i];
    hstate[i] = h;
    out[(m * T + t) * ND * (int64_t)H + (int64_t)dir * H + j] = h;
}

__global__ void fill_f32_kernel(float* p, float v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void copy_f32_kernel(const float* s, float* d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = s[i];
}
static dim3 g1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

size_t lstm_steps_ws_floats(int64_t M, int H) { return 2 * ((size_t)M * H * 2 + (size_t)M * 4 * H); }   // one set per direction

// gates_in [M,T,ND*4H] (both biases included), w_hh [ND,4H,H]; out [M,T,ND*H] zero beyond each length; hn/cn [ND,M,H] or null.
// The two directions are independent chains of T x (GEMM, cell): with one batch in flight the reverse direction runs on
// the side stream (own state / scratch set), with several batches in flight ForkJoin is a no-op and they run back to back.
int launch_birnn_steps(int cell, const float* gin, const int64_t* lens, const float* whh, const float* bhh, const float* h0, const float* c0,
                       float* out, float* hn, float* cn, int64_t M, int T, int H, int ND, float* ws, hipStream_t st, float* cst = nullptr);
int launch_bilstm_steps(const float* gin, const int64_t* lens, const float* whh, const float* h0, const float* c0, float* out,
                        float* hn, float* cn, int64_t M, int T, int H, int ND, float* ws, hipStream_t st) {
    return launch_birnn_steps(0, gin, lens, whh, nullptr, h0, c0, out, hn, cn, M, T, H, ND, ws, st);
}
// cell 0: LSTM (gates_in carries both biases, G = 4); cell 1: GRU (gates_in carries b_ih, b_hh [ND,3H] joins the recurrent product, G = 3)
int launch_birnn_steps(int cell, const float* gin, const int64_t* lens, const float* whh, const float* bhh, const float* h0, const float* c0,
                       float* out, float* hn, float* cn, int64_t M, int T, int H, int ND, float* ws, hipStream_t st, float* cst) {
    const int G = cell == 1 ? 3 : 4;
    hipLaunchKernelGGL(fill_f32_kernel, g1(M * T * ND * H), dim3(256), 0, st, out, 0.f, M * T * ND * (int64_t)H);
    if (cst) hipLaunchKernelGGL(fill_f32_kernel, g1(M * T * ND * H), dim3(256), 0, st, cst, 0.f, M * T * ND * (int64_t)H);
    ForkJoin fj(st);
    if (ND == 2) fj.fork();
    for (int dir = 0; dir < ND; ++dir) {
        hipStream_t ds = dir == 1 ? fj.side : st;
        float* hs = ws + (size_t)dir * ((size_t)M * H * 2 + (size_t)M * 4 * H);
        float* cs = hs + M * H;
        float* hw = cs + M * H;
        const float* w = whh + (int64_t)dir * G * H * H;
        const float* bh = bhh ? bhh + (int64_t)dir * G * H : nullptr;
        if (h0) hipLaunchKernelGGL(copy_f32_kernel, g1(M * H), dim3(256), 0, ds, h0 + (int64_t)dir * M * H, hs, M * (int64_t)H);
        else hipLaunchKernelGGL(fill_f32_kernel, g1(M * H), dim3(256), 0, ds, hs, 0.f, M * (int64_t)H);
        if (c0) hipLaunchKernelGGL(copy_f32_kernel, g1(M * H), dim3(256), 0, ds, c0 + (int64_t)dir * M * H, cs, M * (int64_t)H);
        else hipLaunchKernelGGL(fill_f32_kernel, g1(M * H), dim3(256), 0, ds, cs, 0.f, M * (int64_t)H);
        for (int step = 0; step < T; ++step) {
            const bool skip_gemm = step == 0 && !h0;          // h_{-1} = 0
            if (!skip_gemm) NIR_PROPAGATE(launch_linear(hs, H, nullptr, nullptr, 0, 0, 0, w, H, bh, nullptr, hw, G * H, M, G * H, H, NIR_ACT_NONE, ds));
            ProfScope ps(cell == 1 ? "gru_step_cell_kernel" : "lstm_step_cell_kernel", ds);
            if (cell == 1)
                hipLaunchKernelGGL(gru_step_cell_kernel, g1(M * H), dim3(256), 0, ds, skip_gemm ? nullptr : hw, bh, gin, lens, hs, out, M, T, H, ND,
                                   dir, step);
            else
                hipLaunchKernelGGL(lstm_step_cell_kernel, g1(M * H), dim3(256), 0, ds, skip_gemm ? nullptr : hw, gin, lens, hs, cs, out, cst, M, T,
                                   H, ND, dir, step);
        }
        if (hn) hipLaunchKernelGGL(copy_f32_kernel, g1(M * H), dim3(256), 0, ds, hs, hn + (int64_t)dir * M * H, M * (int64_t)H);
        if (cn && cell == 0) hipLaunchKernelGGL(copy_f32_kernel, g1(M * H), dim3(256), 0, ds, cs, cn + (int64_t)dir * M * H, M * (int64_t)H);
    }
    if (ND == 2) fj.join();
    NIR_CHECK_LAUNCH("nir_bilstm_steps_fwd");
    return 0;
}

// max over ALL T positions of [M,T,D] (mnsrf.py:235-237: padded positions hold zeros and take part)
__global__ __launch_bounds__(256) void maxpool_time_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t M, int T, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * D) return;
    const int64_t m = i / D;
    const int d = (int)(i - m * D);
    const float* p = x + m * T * (int64_t)D + d;
    float v = p[0];
    for (int t = 1; t < T; ++t) v = fmaxf(v, p[(int64_t)t * D]);
    y[i] = v;
}

// comb[b,t,:] = [memory_bank[b,t,:] ; t == 0 ? 0 : session_bank[b,t,:]]   (mnsrf.py:143-148)
__global__ void mnsrf_concat_kernel(const float* mem, const float* sess, float* comb, int64_t B, int S, int Dq, int HS) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int W = Dq + HS;
    if (i >= B * S * W) return;
    const int64_t r = i / W;
    const int c = (int)(i - r * W);
    const int t = (int)(r % S);
    comb[i] = c < Dq ? mem[r * Dq + c] : (t == 0 ? 0.f : sess[r * HS + (c - Dq)]);
}

// scores[r,n] = sum_f proj[r,f] * docs[r,n,f]   (mnsrf.py:152-155); one wave per (r,n)
__global__ __launch_bounds__(256) void mnsrf_dot_kernel(const float* __restrict__ proj, const float* __restrict__ docs,
                                                        float* __restrict__ scores, int64_t R, int N, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= R * N) return;
    const float* a = proj + (pair / N) * D;
    const float* b = docs + pair * D;
    float s = 0.f;
    for (int f = lane; f < D; f += 64) s += a[f] * b[f];
    s = wave_sum(s);
    if (lane == 0) scores[pair] = s;
}

// decoder-init states from batch-major banks: dst[(t * B + b), :] = src[b, t, :] for t < S - 1   (mnsrf.py:96-112, torch.cat(states[:-1], 1))
__global__ void mnsrf_dec_states_kernel(const float* __restrict__ hbank, const float* __restrict__ cbank, float* __restrict__ dec_h,
                                        float* __restrict__ dec_c, int64_t B, int S, int HS) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)(S - 1) * B * HS) return;
    const int64_t r = i / HS;
    const int c = (int)(i - r * HS);
    const int64_t t = r / B, b = r - t * B;
    const int64_t si = (b * S + t) * HS + c;
    if (dec_h) dec_h[i] = hbank[si];
    if (dec_c && cbank) dec_c[i] = cbank[si];
}

// Round 5: the resident-weight path.  With the folded gate tables and the pre-split recurrent weights in the struct (all optional; the host
// mirror builds them once per weight version) the encoders run as ONE launch each -- lstm_cluster_kernel, max over time fused, no memory
// bank -- and the session LSTM as one lstm_step16_kernel launch per query (W_hh as fp16 term pairs, input side hoisted into one GEMM):
// 14 launches per batch instead of 300.
// (q_/d_whh_frag without q_/d_fold -- a table that trains, or one over the fold budget: the same recurrence over PER-BATCH gate rows, written in
// the folded order by one gather-GEMM with W_ih permuted on the fly; rows addressed by position, ids == NULL)
static bool mnsrf_fast_q(const nir_mnsrf_weights* w) { return w->q_whh_frag && w->Hq == 256; }
static bool mnsrf_fast_d(const nir_mnsrf_weights* w) { return w->d_whh_frag && w->Hd == 256; }
static bool mnsrf_fast_s(const nir_mnsrf_weights* w) { return w->s_whh_frag && w->HS % 32 == 0; }

struct MnsrfPlan { float *gin, *enc, *lstm_ws, *mem, *sgin, *sess, *comb, *proj, *docs, *hs, *cs, *h16, *wperm, *bperm; void* xbuf; size_t xbuf_bytes; size_t bytes; };

static MnsrfPlan mnsrf_plan(void* ws, size_t cap, int64_t B, int S, int N, int QL, int DL, const nir_mnsrf_weights* w, int E = 300) {
    Workspace a(ws, cap);
    MnsrfPlan p;
    const int64_t Mq = B * S, Md = B * S * N;
    const int Hq = w->Hq, Hd = w->Hd, HS = w->HS;
    const bool fq = mnsrf_fast_q(w), fd = mnsrf_fast_d(w) || Md == 0;
    // gate pre-activations of the encoder being run: the streaming encoders and the per-batch form of the cluster recurrence (no folded table);
    // the memory bank only for the encoders that still stream
    const int64_t grows = std::max<int64_t>((fq && w->q_fold) ? 0 : Mq * QL, ((fd && w->d_fold) || Md == 0) ? 0 : Md * DL);
    const int64_t rows = std::max<int64_t>(fq ? 0 : Mq * QL, fd ? 0 : Md * DL);
    const int Hmax = std::max(Hq, Hd);
    p.gin = a.take<float>((size_t)grows * 8 * Hmax);
    p.enc = a.take<float>((size_t)rows * 2 * Hmax);
    p.wperm = a.take<float>((fq && !w->q_fold) || (mnsrf_fast_d(w) && !w->d_fold) ? (size_t)8 * Hmax * E : 0);
    p.bperm = a.take<float>((size_t)8 * Hmax);
    p.lstm_ws = a.take<float>(std::max(lstm_steps_ws_floats(std::max<int64_t>(fq ? 0 : Mq, fd ? 0 : Md), Hmax), mnsrf_fast_s(w) ? (size_t)0 : lstm_steps_ws_floats(B, HS)));
    p.mem = a.take<float>((size_t)Mq * 2 * Hq);
    p.sgin = a.take<float>((size_t)Mq * 4 * HS);
    p.sess = a.take<float>((size_t)Mq * HS);
    p.comb = a.take<float>((size_t)Mq * (2 * Hq + HS));
    p.proj = a.take<float>((size_t)Mq * 2 * Hd);
    p.docs = a.take<float>((size_t)Md * 2 * Hd);
    // session LSTM, step-major: slot t + 1 = the state after query t (slot 0 = the zero state, never touched); fp32 h, c and h as fp16 term pairs
    p.hs = a.take<float>((size_t)(S + 1) * B * HS);
    p.cs = a.take<float>((size_t)(S + 1) * B * HS);
    p.h16 = a.take<float>((size_t)(S + 1) * B * HS);
    p.xbuf_bytes = std::max(fq ? lstm256_xbuf_bytes(Mq, 2) : (size_t)0, mnsrf_fast_d(w) && Md ? lstm256_xbuf_bytes(Md, 2) : (size_t)0);
    p.xbuf = a.take<char>(p.xbuf_bytes);
    p.bytes = align_up(a.off, 256);
    return p;
}

// out[b, t, :] = steps[t + 1][b][:]   (session bank, batch-major, from the step-major states)
__global__ void mnsrf_bank_kernel(const float* __restrict__ steps, float* __restrict__ out, int64_t B, int S, int HS) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * S * HS) return;
    const int64_t r = i / HS;
    const int c = (int)(i - r * HS);
    const int64_t b = r / S;
    const int t = (int)(r - b * S);
    out[i] = steps[((int64_t)(t + 1) * B + b) * HS + c];
}

// queries: ids [B*S,QL] -> memory_bank [B,S,2Hq] (BiLSTM + max over time), session_bank [B,S,HS] (session LSTM); optional decoder-init
// states dec_h / dec_c [(S-1)*B, HS] = the session LSTM's state after queries 0 .. S-2, step-major along the batch axis (mnsrf.py:96-112)
static int mnsrf_encode(const int64_t* src, const int64_t* src_len, int64_t B, int S, int QL, const float* table, int64_t V, int E,
                        const nir_mnsrf_weights* w, const MnsrfPlan& p, float* mem, float* sess, float* dec_h, float* dec_c, hipStream_t st) {
    const int64_t Mq = B * S;
    const int Hq = w->Hq, HS = w->HS;
    if (mnsrf_fast_q(w) && w->q_fold) {
        NIR_PROPAGATE(launch_lstm256_cluster(w->q_fold, src, src_len, w->q_whh_frag, mem, 1, w->err, Mq, V, QL, 2, p.xbuf, p.xbuf_bytes, st));
    } else if (mnsrf_fast_q(w)) {
        NIR_PROPAGATE(launch_fold_permute(w->q_wih, w->q_bih, w->q_bhh, Hq, 2, E, p.wperm, p.bperm, nullptr, 0, st));
        NIR_PROPAGATE(launch_linear(nullptr, 0, src, table, E, 1, 1, p.wperm, E, p.bperm, nullptr, p.gin, 8 * Hq, Mq * QL, 8 * Hq, E, NIR_ACT_NONE, st));
        NIR_PROPAGATE(launch_lstm256_cluster(p.gin, nullptr, src_len, w->q_whh_frag, mem, 1, w->err, Mq, Mq * QL, QL, 2, p.xbuf, p.xbuf_bytes, st));
    } else {
        NIR_PROPAGATE(launch_linear(nullptr, 0, src, table, E, 1, 1, w->q_wih, E, w->q_bih, w->q_bhh, p.gin, 8 * Hq, Mq * QL, 8 * Hq, E, NIR_ACT_NONE, st));
        NIR_PROPAGATE(launch_bilstm_steps(p.gin, src_len, w->q_whh, nullptr, nullptr, p.enc, nullptr, nullptr, Mq, QL, Hq, 2, p.lstm_ws, st));
        hipLaunchKernelGGL(maxpool_time_kernel, g1(Mq * 2 * Hq), dim3(256), 0, st, p.enc, mem, Mq, QL, 2 * Hq);
    }
    // session LSTM: B sequences of S steps over the pooled queries (hidden state carried from query to query); input side hoisted into one GEMM
    NIR_PROPAGATE(launch_linear(mem, 2 * Hq, nullptr, nullptr, 0, 0, 0, w->s_wih, 2 * Hq, w->s_bih, w->s_bhh, p.sgin, 4 * HS, Mq, 4 * HS, 2 * Hq, NIR_ACT_NONE, st));
    const int64_t slot = B * (int64_t)HS;
    if (mnsrf_fast_s(w)) {
        LstmStepArgs a;
        a.x[0] = a.x[1] = nullptr; a.xid[0] = a.xid[1] = nullptr; a.xstride[0] = a.xstride[1] = 0;
        a.wih[0] = a.wih[1] = w->s_wih; a.whh[0] = a.whh[1] = w->s_whh; a.bih[0] = a.bih[1] = w->s_bih; a.bhh[0] = a.bhh[1] = w->s_bhh;
        a.chain0 = 0; a.B = (int)B; a.I = 2 * Hq; a.H = HS;
        a.gxstride = (int64_t)S * 4 * HS;
        a.whh_frag[0] = w->s_whh_frag;
        for (int t = 0; t < S; ++t) {
            a.gx[0] = p.sgin + (int64_t)t * 4 * HS;
            a.hprev[0] = t ? p.hs + t * slot : nullptr; a.cprev[0] = t ? p.cs + t * slot : nullptr;
            a.hnext[0] = p.hs + (t + 1) * slot; a.cnext[0] = p.cs + (t + 1) * slot;
            a.h16prev[0] = t ? reinterpret_cast<const _Float16*>(p.h16 + t * slot) : nullptr;
            a.h16next[0] = reinterpret_cast<_Float16*>(p.h16 + (t + 1) * slot);
            NIR_PROPAGATE(launch_lstm_step(a, 1, st));
        }
        hipLaunchKernelGGL(mnsrf_bank_kernel, g1(Mq * HS), dim3(256), 0, st, p.hs, sess, B, S, HS);
        if (S > 1 && dec_h) hipLaunchKernelGGL(copy_f32_kernel, g1((S - 1) * slot), dim3(256), 0, st, p.hs + slot, dec_h, (S - 1) * slot);
        if (S > 1 && dec_c) hipLaunchKernelGGL(copy_f32_kernel, g1((S - 1) * slot), dim3(256), 0, st, p.cs + slot, dec_c, (S - 1) * slot);
    } else {
        // streaming form; the cell states of every step are only written when the decoder states are wanted ([B,S,HS] in p.cs)
        NIR_PROPAGATE(launch_birnn_steps(0, p.sgin, nullptr, w->s_whh, nullptr, nullptr, nullptr, sess, nullptr, nullptr, B, S, HS, 1, p.lstm_ws, st,
                                         dec_c ? p.cs : nullptr));
        if (S > 1 && (dec_h || dec_c))
            hipLaunchKernelGGL(mnsrf_dec_states_kernel, g1((S - 1) * slot), dim3(256), 0, st, sess, dec_c ? p.cs : nullptr, dec_h, dec_c, B, S, HS);
    }
    NIR_CHECK_LAUNCH("nir_mnsrf_encode");
    return 0;
}

// scores from the query side (memory_bank [B,S,2Hq], session_bank [B,S,HS]) and the candidate documents
static int mnsrf_rank(const float* mem, const float* sess, const int64_t* doc_ids, const int64_t* doc_lens, int64_t B, int S, int N, int DL,
                      const float* table, int64_t V, int E, const nir_mnsrf_weights* w, const MnsrfPlan& p, float* scores, hipStream_t st) {
    const int64_t Mq = B * S, Md = B * S * N;
    const int Hq = w->Hq, Hd = w->Hd, HS = w->HS;
    // documents: BiLSTM -> max over time
    if (mnsrf_fast_d(w) && w->d_fold) {
        NIR_PROPAGATE(launch_lstm256_cluster(w->d_fold, doc_ids, doc_lens, w->d_whh_frag, p.docs, 1, w->err, Md, V, DL, 2, p.xbuf, p.xbuf_bytes, st));
    } else if (mnsrf_fast_d(w)) {
        NIR_PROPAGATE(launch_fold_permute(w->d_wih, w->d_bih, w->d_bhh, Hd, 2, E, p.wperm, p.bperm, nullptr, 0, st));
        NIR_PROPAGATE(launch_linear(nullptr, 0, doc_ids, table, E, 1, 1, p.wperm, E, p.bperm, nullptr, p.gin, 8 * Hd, Md * DL, 8 * Hd, E, NIR_ACT_NONE, st));
        NIR_PROPAGATE(launch_lstm256_cluster(p.gin, nullptr, doc_lens, w->d_whh_frag, p.docs, 1, w->err, Md, Md * DL, DL, 2, p.xbuf, p.xbuf_bytes, st));
    } else {
        NIR_PROPAGATE(launch_linear(nullptr, 0, doc_ids, table, E, 1, 1, w->d_wih, E, w->d_bih, w->d_bhh, p.gin, 8 * Hd, Md * DL, 8 * Hd, E, NIR_ACT_NONE, st));
        NIR_PROPAGATE(launch_bilstm_steps(p.gin, doc_lens, w->d_whh, nullptr, nullptr, p.enc, nullptr, nullptr, Md, DL, Hd, 2, p.lstm_ws, st));
        hipLaunchKernelGGL(maxpool_time_kernel, g1(Md * 2 * Hd), dim3(256), 0, st, p.enc, p.docs, Md, DL, 2 * Hd);
    }
    // tanh projection of [query ; session state] and the dot product with every candidate
    const int KC = 2 * Hq + HS;
    hipLaunchKernelGGL(mnsrf_concat_kernel, g1(Mq * KC), dim3(256), 0, st, mem, sess, p.comb, B, S, 2 * Hq, HS);
    NIR_PROPAGATE(launch_linear(p.comb, KC, nullptr, nullptr, 0, 0, 0, w->proj_w, KC, w->proj_b, nullptr, p.proj, 2 * Hd, Mq, 2 * Hd, KC, NIR_ACT_TANH, st));
    hipLaunchKernelGGL(mnsrf_dot_kernel, dim3((unsigned)((Md + 3) / 4)), dim3(256), 0, st, p.proj, p.docs, scores, Mq, N, 2 * Hd);
    NIR_CHECK_LAUNCH("nir_mnsrf_rank");
    return 0;
}

}  // namespace nir

extern "C" size_t nir_bilstm_steps_workspace_bytes(int64_t M, int H) {
    return nir::lstm_steps_ws_floats(M, H) * sizeof(float) + 256;
}

extern "C" int nir_bilstm_steps_fwd(const float* gates_in, const int64_t* lengths, const float* w_hh, const float* h0, const float* c0,
                                    float* out, float* hn, float* cn, int64_t M, int T, int H, int ndir, void* workspace,
                                    size_t workspace_bytes, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(gates_in && w_hh && out && workspace, "bilstm_steps: null pointer");
    NIR_REQUIRE(M >= 0 && T > 0 && H > 0 && (ndir == 1 || ndir == 2), "bilstm_steps: bad dims");
    NIR_REQUIRE(workspace_bytes >= nir_bilstm_steps_workspace_bytes(M, H), "bilstm_steps: workspace too small");
    if (M == 0) return 0;
    return launch_bilstm_steps(gates_in, lengths, w_hh, h0, c0, out, hn, cn, M, T, H, ndir, (float*)workspace, (hipStream_t)stream);
}

extern "C" int nir_maxpool_time_f32(const float* x, int64_t M, int T, int D, float* y, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(x && y && M >= 0 && T > 0 && D > 0, "maxpool_time: bad arguments");
    if (M == 0) return 0;
    hipLaunchKernelGGL(maxpool_time_kernel, g1(M * D), dim3(256), 0, (hipStream_t)stream, x, y, M, T, D);
    NIR_CHECK_LAUNCH("nir_maxpool_time_f32");
    return 0;
}

extern "C" int nir_birnn_steps_fwd(int cell, const float* gates_in, const int64_t* lengths, const float* w_hh, const float* b_hh, const float* h0,
                                   const float* c0, float* out, float* c_steps, float* hn, float* cn, int64_t M, int T, int H, int ndir,
                                   void* workspace, size_t workspace_bytes, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(cell == NIR_CELL_LSTM || cell == NIR_CELL_GRU, "birnn_steps: cell must be NIR_CELL_LSTM or NIR_CELL_GRU");
    NIR_REQUIRE(gates_in && w_hh && out && workspace, "birnn_steps: null pointer");
    NIR_REQUIRE(cell == NIR_CELL_LSTM || b_hh, "birnn_steps: the GRU needs b_hh (it sits inside the reset-gate product)");
    NIR_REQUIRE(M >= 0 && T > 0 && H > 0 && (ndir == 1 || ndir == 2), "birnn_steps: bad dims");
    NIR_REQUIRE(workspace_bytes >= nir_bilstm_steps_workspace_bytes(M, H), "birnn_steps: workspace too small");
    if (M == 0) return 0;
    return launch_birnn_steps(cell, gates_in, lengths, w_hh, cell == NIR_CELL_GRU ? b_hh : nullptr, h0, c0, out, hn, cn, M, T, H, ndir,
                              (float*)workspace, (hipStream_t)stream, cell == NIR_CELL_LSTM ? c_steps : nullptr);
}

extern "C" size_t nir_mnsrf_workspace_bytes(int64_t B, int S, int N, int QL, int DL, const nir_mnsrf_weights* w) {
    if (!w) return 0;
    return nir::mnsrf_plan(nullptr, 0, B, S, N, QL, DL, w).bytes + 256;      // (sized for E = 300, the reference's emsize; larger E: see mnsrf_plan_e)
}

extern "C" int nir_mnsrf_encode_states(const int64_t* source_ids, const int64_t* source_lens, int64_t B, int S, int QL, const float* table,
                                       int64_t V, int E, const nir_mnsrf_weights* w, void* workspace, size_t workspace_bytes,
                                       float* memory_bank, float* session_bank, float* dec_h, float* dec_c, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(source_ids && source_lens && table && w && workspace && memory_bank && session_bank, "mnsrf_encode: null pointer");
    NIR_REQUIRE(B >= 0 && S > 0 && QL > 0 && E > 0 && V > 0, "mnsrf_encode: bad dims");
    NIR_REQUIRE(workspace_bytes >= nir_mnsrf_workspace_bytes(B, S, 0, QL, 1, w), "mnsrf_encode: workspace too small");
    if (B == 0) return 0;
    const MnsrfPlan p = mnsrf_plan(workspace, workspace_bytes, B, S, 0, QL, 1, w);
    NIR_REQUIRE(E <= 300 || (w->q_fold != nullptr) || !w->q_whh_frag, "mnsrf: the per-batch cluster form is sized for emsize <= 300");
    return mnsrf_encode(source_ids, source_lens, B, S, QL, table, V, E, w, p, memory_bank, session_bank, dec_h, dec_c, (hipStream_t)stream);
}

extern "C" int nir_mnsrf_encode(const int64_t* source_ids, const int64_t* source_lens, int64_t B, int S, int QL, const float* table,
                                int64_t V, int E, const nir_mnsrf_weights* w, void* workspace, size_t workspace_bytes,
                                float* memory_bank, float* session_bank, nir_stream_t stream) {
    return nir_mnsrf_encode_states(source_ids, source_lens, B, S, QL, table, V, E, w, workspace, workspace_bytes, memory_bank, session_bank,
                                   nullptr, nullptr, stream);
}

extern "C" int nir_mnsrf_rank(const float* memory_bank, const float* session_bank, const int64_t* doc_ids, const int64_t* doc_lens, int64_t B,
                              int S, int N, int DL, const float* table, int64_t V, int E, const nir_mnsrf_weights* w, void* workspace,
                              size_t workspace_bytes, float* scores, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(memory_bank && session_bank && doc_ids && doc_lens && table && w && workspace && scores, "mnsrf_rank: null pointer");
    NIR_REQUIRE(B >= 0 && S > 0 && N > 0 && DL > 0 && E > 0 && V > 0, "mnsrf_rank: bad dims");
    NIR_REQUIRE(workspace_bytes >= nir_mnsrf_workspace_bytes(B, S, N, 1, DL, w), "mnsrf_rank: workspace too small");
    if (B == 0) return 0;
    const MnsrfPlan p = mnsrf_plan(workspace, workspace_bytes, B, S, N, 1, DL, w);
    NIR_REQUIRE(E <= 300 || (w->d_fold != nullptr) || !w->d_whh_frag, "mnsrf: the per-batch cluster form is sized for emsize <= 300");
    return mnsrf_rank(memory_bank, session_bank, doc_ids, doc_lens, B, S, N, DL, table, V, E, w, p, scores, (hipStream_t)stream);
}

extern "C" int nir_mnsrf_score(const int64_t* source_ids, const int64_t* source_lens, const int64_t* doc_ids, const int64_t* doc_lens,
                               int64_t B, int S, int N, int QL, int DL, const float* table, int64_t V, int E,
                               const nir_mnsrf_weights* w, void* workspace, size_t workspace_bytes, float* scores,
                               nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(source_ids && source_lens && doc_ids && doc_lens && table && w && workspace && scores, "mnsrf_score: null pointer");
    NIR_REQUIRE(B >= 0 && S > 0 && N > 0 && QL > 0 && DL > 0 && E > 0 && V > 0, "mnsrf_score: bad dims");
    NIR_REQUIRE(workspace_bytes >= nir_mnsrf_workspace_bytes(B, S, N, QL, DL, w), "mnsrf_score: workspace too small");
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const MnsrfPlan p = mnsrf_plan(workspace, workspace_bytes, B, S, N, QL, DL, w);
    NIR_REQUIRE(E <= 300 || (w->d_fold && w->q_fold) || !(w->d_whh_frag || w->q_whh_frag), "mnsrf: the per-batch cluster form is sized for emsize <= 300");
    NIR_PROPAGATE(mnsrf_encode(source_ids, source_lens, B, S, QL, table, V, E, w, p, p.mem, p.sess, nullptr, nullptr, st));
    return mnsrf_rank(p.mem, p.sess, doc_ids, doc_lens, B, S, N, DL, table, V, E, w, p, scores, st);
}
