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

struct MnsrfPlan { float *gin, *enc, *lstm_ws, *mem, *sgin, *sess, *comb, *proj, *docs; size_t bytes; };

static MnsrfPlan mnsrf_plan(void* ws, size_t cap, int64_t B, int S, int N, int QL, int DL, const nir_mnsrf_weights* w) {
    Workspace a(ws, cap);
    MnsrfPlan p;
    const int64_t Mq = B * S, Md = B * S * N;
    const int Hq = w->Hq, Hd = w->Hd, HS = w->HS;
    const int64_t rows = std::max<int64_t>(Mq * QL, Md * DL);
    const int Hmax = std::max(Hq, Hd);
    p.gin = a.take<float>((size_t)rows * 8 * Hmax);                     // gate pre-activations of the encoder being run
    p.enc = a.take<float>((size_t)rows * 2 * Hmax);                     // its memory bank
    p.lstm_ws = a.take<float>(std::max(lstm_steps_ws_floats(std::max(Mq, Md), Hmax), lstm_steps_ws_floats(B, HS)));
    p.mem = a.take<float>((size_t)Mq * 2 * Hq);
    p.sgin = a.take<float>((size_t)Mq * 4 * HS);
    p.sess = a.take<float>((size_t)Mq * HS);
    p.comb = a.take<float>((size_t)Mq * (2 * Hq + HS));
    p.proj = a.take<float>((size_t)Mq * 2 * Hd);
    p.docs = a.take<float>((size_t)Md * 2 * Hd);
    p.bytes = align_up(a.off, 256);
    return p;
}

// queries: ids [B*S,QL] -> memory_bank [B,S,2Hq] (BiLSTM + max over time), session_bank [B,S,HS] (session LSTM)
static int mnsrf_encode(const int64_t* src, const int64_t* src_len, int64_t B, int S, int QL, const float* table, int E,
                        const nir_mnsrf_weights* w, const MnsrfPlan& p, float* mem, float* sess, hipStream_t st) {
    const int64_t Mq = B * S;
    const int Hq = w->Hq, HS = w->HS;
    NIR_PROPAGATE(launch_linear(nullptr, 0, src, table, E, 1, 1, w->q_wih, E, w->q_bih, w->q_bhh, p.gin, 8 * Hq, Mq * QL, 8 * Hq, E, NIR_ACT_NONE, st));
    NIR_PROPAGATE(launch_bilstm_steps(p.gin, src_len, w->q_whh, nullptr, nullptr, p.enc, nullptr, nullptr, Mq, QL, Hq, 2, p.lstm_ws, st));
    hipLaunchKernelGGL(maxpool_time_kernel, g1(Mq * 2 * Hq), dim3(256), 0, st, p.enc, mem, Mq, QL, 2 * Hq);
    // session LSTM: B sequences of S steps over the pooled queries (hidden state carried from query to query)
    NIR_PROPAGATE(launch_linear(mem, 2 * Hq, nullptr, nullptr, 0, 0, 0, w->s_wih, 2 * Hq, w->s_bih, w->s_bhh, p.sgin, 4 * HS, Mq, 4 * HS, 2 * Hq, NIR_ACT_NONE, st));
    NIR_PROPAGATE(launch_bilstm_steps(p.sgin, nullptr, w->s_whh, nullptr, nullptr, sess, nullptr, nullptr, B, S, HS, 1, p.lstm_ws, st));
    NIR_CHECK_LAUNCH("nir_mnsrf_encode");
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
    return nir::mnsrf_plan(nullptr, 0, B, S, N, QL, DL, w).bytes + 256;
}

extern "C" int nir_mnsrf_encode(const int64_t* source_ids, const int64_t* source_lens, int64_t B, int S, int QL, const float* table,
                                int64_t V, int E, const nir_mnsrf_weights* w, void* workspace, size_t workspace_bytes,
                                float* memory_bank, float* session_bank, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(source_ids && source_lens && table && w && workspace && memory_bank && session_bank, "mnsrf_encode: null pointer");
    NIR_REQUIRE(B >= 0 && S > 0 && QL > 0 && E > 0 && V > 0, "mnsrf_encode: bad dims");
    NIR_REQUIRE(workspace_bytes >= nir_mnsrf_workspace_bytes(B, S, 0, QL, 1, w), "mnsrf_encode: workspace too small");
    if (B == 0) return 0;
    const MnsrfPlan p = mnsrf_plan(workspace, workspace_bytes, B, S, 0, QL, 1, w);
    return mnsrf_encode(source_ids, source_lens, B, S, QL, table, E, w, p, memory_bank, session_bank, (hipStream_t)stream);
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
    const int64_t Mq = B * S, Md = B * S * N;
    const int Hq = w->Hq, Hd = w->Hd, HS = w->HS;
    NIR_PROPAGATE(mnsrf_encode(source_ids, source_lens, B, S, QL, table, E, w, p, p.mem, p.sess, st));
    // documents: gather-GEMM -> BiLSTM -> max over time
    NIR_PROPAGATE(launch_linear(nullptr, 0, doc_ids, table, E, 1, 1, w->d_wih, E, w->d_bih, w->d_bhh, p.gin, 8 * Hd, Md * DL, 8 * Hd, E, NIR_ACT_NONE, st));
    NIR_PROPAGATE(launch_bilstm_steps(p.gin, doc_lens, w->d_whh, nullptr, nullptr, p.enc, nullptr, nullptr, Md, DL, Hd, 2, p.lstm_ws, st));
    hipLaunchKernelGGL(maxpool_time_kernel, g1(Md * 2 * Hd), dim3(256), 0, st, p.enc, p.docs, Md, DL, 2 * Hd);
    // tanh projection of [query ; session state] and the dot product with every candidate
    const int KC = 2 * Hq + HS;
    hipLaunchKernelGGL(mnsrf_concat_kernel, g1(Mq * KC), dim3(256), 0, st, p.mem, p.sess, p.comb, B, S, 2 * Hq, HS);
    NIR_PROPAGATE(launch_linear(p.comb, KC, nullptr, nullptr, 0, 0, 0, w->proj_w, KC, w->proj_b, nullptr, p.proj, 2 * Hd, Mq, 2 * Hd, KC, NIR_ACT_TANH, st));
    hipLaunchKernelGGL(mnsrf_dot_kernel, dim3((unsigned)((Md + 3) / 4)), dim3(256), 0, st, p.proj, p.docs, scores, Mq, N, 2 * Hd);
    NIR_CHECK_LAUNCH("nir_mnsrf_score");
    return 0;
}
