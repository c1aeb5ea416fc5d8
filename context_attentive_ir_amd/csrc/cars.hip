// CARS ranking path (neuroir/multitask/cars.py).
//
// nir_cars_encode       : CARS.encode / encode_document (:193-260) = embedding gather fused into the LSTM gate
//                         GEMM (K=E, N=8H, fp32 MFMA) -> BiLSTM recurrence -> attention MLP GEMM + tanh -> logit row
//                         dot -> masked softmax + weighted sum (apply_pooling, :671-691), one wave per sequence.
// nir_cars_rank_session : encode_clicks (:262-304, incl. the batch-dependent mask quirk) + the sequential session
//                         loop of encode_session (:306-458) restricted to what ranking needs: cross attention over
//                         the previous session states (incl. the initial zero state), rank() with the maxout
//                         ranknet (:460-520), and the two single-step session LSTMs.  Every step is enqueued from
//                         here -- no host sync anywhere (the reference syncs at .cpu().numpy() and lengths.tolist()).
#include "common.hpp"

namespace nir {

int launch_linear(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                  int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                  int64_t ldc, int64_t M, int N, int K, int act, hipStream_t st);
int launch_linear_ex(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                     int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                     int64_t ldc, int64_t M, int N, int K, int act, const float* add, int64_t ldadd, hipStream_t st);
constexpr int ACT_MAXOUT2 = 16;
constexpr int ACT_TANH_ROWDOT16 = 17;
int launch_bilstm_folded(const void* pt, int pt_dtype, const int64_t* ids, const int64_t* lens, const float* whh, float* out,
                         int* err, int64_t M, int64_t V, int T, int H, int ND, hipStream_t st);
int launch_rowdot(const float* x, int64_t ldx, const float* w, const float* b, float* out, int64_t M, int K, int act,
                  hipStream_t st);
int launch_bilstm(const float* gin, const int64_t* lens, const float* whh, const float* h0, const float* c0,
                  float* out, float* hn, float* cn, int64_t M, int T, int H, int ND, hipStream_t st);

// ------------------------------------------------------------------------------------------------------
// attention pooling: one wave per sequence; logits [M,T], h [M,T,D] (D % 4 == 0, D <= 1024)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_pool_kernel(const float* __restrict__ h, const float* __restrict__ logits,
                                                        const int64_t* __restrict__ lens, int64_t M, int T, int D,
                                                        float* __restrict__ pooled) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    int len = lens ? (int)lens[m] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    const float* lg = logits + m * T;
    float mx = -INFINITY;
    for (int t = lane; t < len; t += 64) mx = fmaxf(mx, lg[t]);
    mx = wave_max(mx);
    float den = 0.f;
    for (int t = lane; t < len; t += 64) den += expf(lg[t] - mx);
    den = wave_sum(den);
    const int nch = D >> 2;
    for (int c = lane; c < nch; c += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < len; ++t) {
            float p = expf(lg[t] - mx) / den;
            float4 v = *reinterpret_cast<const float4*>(h + (m * T + t) * D + 4 * c);
            acc.x = fmaf(p, v.x, acc.x); acc.y = fmaf(p, v.y, acc.y); acc.z = fmaf(p, v.z, acc.z); acc.w = fmaf(p, v.w, acc.w);
        }
        *reinterpret_cast<float4*>(pooled + m * D + 4 * c) = acc;
    }
}

// attention pooling over logit partials (the fused GEMM epilogue ACT_TANH_ROWDOT16 leaves NP = D/16 partial sums per row):
// logit[t] = sum_j lpart[t][j] + b3;  p = softmax over t < len;  pooled = sum_t p_t h_t.  One wave per sequence; the
// probabilities are computed once (lane = time step) and broadcast through LDS, every lane then owns 4 channels per pass.
__global__ __launch_bounds__(256) void attn_pool2_kernel(const float* __restrict__ h, const float* __restrict__ lpart, int NP,
                                                         const float* __restrict__ b3, const int64_t* __restrict__ lens,
                                                         int64_t M, int T, int D, float* __restrict__ pooled) {
    extern __shared__ float pr[];                    // [4][T]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t m = (int64_t)blockIdx.x * 4 + wave;
    if (m >= M) return;
    float* pw = pr + wave * T;
    int len = lens ? (int)lens[m] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    const float bias = b3 ? b3[0] : 0.f;
    float mx = -INFINITY;
    for (int t = lane; t < len; t += 64) {
        const float* lp = lpart + (m * T + t) * NP;
        float s = bias;
        for (int j = 0; j < NP; j += 4) {
            const float4 v = *reinterpret_cast<const float4*>(lp + j);
            s += (v.x + v.y) + (v.z + v.w);
        }
        pw[t] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float den = 0.f;
    for (int t = lane; t < len; t += 64) {
        const float e = expf(pw[t] - mx);
        pw[t] = e;
        den += e;
    }
    den = wave_sum(den);
    const float inv = 1.0f / den;                    // len == 0 -> 0/0 = NaN rows, like softmax over an all -inf row
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS writes are visible to its own lanes
    const int nch = D >> 2;
    for (int c = lane; c < nch; c += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* hp = h + m * T * D + 4 * c;
        for (int t = 0; t < len; ++t) {
            const float p = pw[t] * inv;
            const float4 v = *reinterpret_cast<const float4*>(hp + (int64_t)t * D);
            acc.x = fmaf(p, v.x, acc.x); acc.y = fmaf(p, v.y, acc.y); acc.z = fmaf(p, v.z, acc.z); acc.w = fmaf(p, v.w, acc.w);
        }
        if (len == 0) acc = make_float4(NAN, NAN, NAN, NAN);
        *reinterpret_cast<float4*>(pooled + m * D + 4 * c) = acc;
    }
}

struct EncPlan {
    float *gates, *enc, *a1, *logit;
    size_t bytes;
};
static EncPlan enc_plan(void* ws, size_t cap, int64_t M, int T, int H) {
    Workspace a(ws, cap);
    EncPlan p;
    p.gates = a.take<float>((size_t)M * T * 8 * H);
    p.enc = a.take<float>((size_t)M * T * 2 * H);
    p.a1 = a.take<float>((size_t)M * T * 2 * H);
    p.logit = a.take<float>((size_t)M * T);
    p.bytes = align_up(a.off, 256);
    return p;
}

// ------------------------------------------------------------------------------------------------------
// session kernels
// ------------------------------------------------------------------------------------------------------
// m = max over rows of count_nonzero(labels[row, :])   (cars.py:285-289, batch-wide)
__global__ __launch_bounds__(256) void click_maxcount_kernel(const float* labels, int rows, int N, int* mout) {
    __shared__ int part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int best = 0;
    for (int r = wave; r < rows; r += 4) {
        int c = 0;
        for (int k = lane; k < N; k += 64) c += labels[(int64_t)r * N + k] != 0.f;
        c = (int)wave_sum((float)c);
        best = max(best, c);
    }
    if (lane == 0) part[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) mout[0] = max(max(part[0], part[1]), max(part[2], part[3]));
}

// one wave per (b,s) row, N <= 64: stable descending rank by label, attend over {rank < count} U {rank >= m}
__global__ __launch_bounds__(256) void click_pool_kernel(const float* __restrict__ docs, const float* __restrict__ e,
                                                         const float* __restrict__ labels, const int* __restrict__ mptr,
                                                         int rows, int N, int D, float* __restrict__ clicks) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int m = mptr[0];
    const float lab = lane < N ? labels[(int64_t)r * N + lane] : -INFINITY;
    int rank = 0;
    for (int k = 0; k < N; ++k) {
        float lk = __shfl(lab, k, 64);
        rank += (lk > lab) || (lk == lab && k < lane);
    }
    const int count = (int)wave_sum((lane < N && lab != 0.f) ? 1.f : 0.f);
    const bool keep = lane < N && (rank < count || rank >= m);
    const float lg = keep ? e[(int64_t)r * N + lane] : -INFINITY;
    const float mx = wave_max(lg);
    const float ex = keep ? expf(lg - mx) : 0.f;
    const float p = ex / wave_sum(ex);   // all masked -> NaN, exactly like softmax of all -inf in the reference
    const int nch = D >> 2;
    for (int c = lane; c < nch; c += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < N; ++k) {
            float pk = __shfl(p, k, 64);
            float4 v = *reinterpret_cast<const float4*>(docs + ((int64_t)r * N + k) * D + 4 * c);
            acc.x = fmaf(pk, v.x, acc.x); acc.y = fmaf(pk, v.y, acc.y); acc.z = fmaf(pk, v.z, acc.z); acc.w = fmaf(pk, v.w, acc.w);
        }
        *reinterpret_cast<float4*>(clicks + (int64_t)r * D + 4 * c) = acc;
    }
}

// Batched cross attention over the session states (cars.py:348-366), one workgroup per (session b, step t):
//   logit_k = inter[k][b] . q[b,t]  for k = 0..t   (inter_k = W states_k + bias; state 0 is the zero vector)
//   out     = sum_k softmax(logit)_k * states[k][b]
// for both the query-session and the document-session states (both keyed by the QUERY vector, :350,361), written
// next to q[b,t] into xcat[(b,t)] = [q ; sq ; sd], the input row of the rank() projection.
__global__ __launch_bounds__(256) void session_attend_kernel(const float* __restrict__ interQ, const float* __restrict__ Qs,
                                                             const float* __restrict__ interD, const float* __restrict__ Ds,
                                                             const float* __restrict__ q, int B, int S, int D, int HS,
                                                             float* __restrict__ xcat) {
    __shared__ float lg[2][64];
    const int bt = blockIdx.x, b = bt / S, t = bt % S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nstates = t + 1;
    const float* qv = q + (int64_t)bt * D;
    float* orow = xcat + (int64_t)bt * (D + 2 * HS);
    for (int e = wave; e < 2 * nstates; e += 4) {
        const int which = e / nstates, k = e % nstates;
        const float* in = (which ? interD : interQ) + ((int64_t)k * B + b) * D;
        float sacc = 0.f;
        for (int f = lane; f < D; f += 64) sacc += in[f] * qv[f];
        sacc = wave_sum(sacc);
        if (lane == 0) lg[which][k] = sacc;
    }
    for (int f = threadIdx.x; f < D; f += 256) orow[f] = qv[f];
    __syncthreads();
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const float* st = which ? Ds : Qs;
        float mx = -INFINITY;
        for (int k = 0; k < nstates; ++k) mx = fmaxf(mx, lg[which][k]);
        float den = 0.f;
        for (int k = 0; k < nstates; ++k) den += expf(lg[which][k] - mx);
        for (int f = threadIdx.x; f < HS; f += 256) {
            float acc = 0.f;
            for (int k = 0; k < nstates; ++k) acc = fmaf(expf(lg[which][k] - mx) / den, st[((int64_t)k * B + b) * HS + f], acc);
            orow[D + which * HS + f] = acc;
        }
    }
}

// wsum[o,k] = w1[o,k] + w2[o,k]   (W_shared + W_priv1)  and  wcat[o,:] = [wq[o,:] | wsum[o,:]]
__global__ void concat_weights_kernel(const float* w1, int K1, const float* w2, const float* w3, int K2, int O, float* wcat) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K = K1 + K2;
    if (i < (int64_t)O * K) {
        int o = (int)(i / K), k = (int)(i % K);
        float v;
        if (k < K1) v = w1[(int64_t)o * K1 + k];
        else {
            v = w2[(int64_t)o * K2 + (k - K1)];
            if (w3) v += w3[(int64_t)o * K2 + (k - K1)];
        }
        wcat[i] = v;
    }
}

// feats[(b,t,n)] = [q', d, |q'-d|, q'*d]   (cars.py:514-518); q' row = (b,t), d row = (b,t,n)
__global__ void rank_feats_kernel(const float* qp, const float* docs, int N, int D, int64_t rows, float* feats) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * D) {
        int64_t r = i / D;
        int f = (int)(i % D);
        float q = qp[(r / N) * D + f], d = docs[r * D + f];
        float* o = feats + r * 4 * D;
        o[f] = q; o[D + f] = d; o[2 * D + f] = fabsf(q - d); o[3 * D + f] = q * d;
    }
}

// LSTM cell on gate pre-activations [B,4HS] (i,f,g,o) laid out with row stride gstride: updates c in place, writes h.
__global__ void lstm_cell_kernel(const float* gates, int64_t gstride, float* c, float* hout, int B, int HS) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)B * HS) {
        int b = (int)(i / HS), j = (int)(i % HS);
        const float* g = gates + (int64_t)b * gstride;
        float cc = fast_sigmoid(g[HS + j]) * c[i] + fast_sigmoid(g[j]) * fast_tanh(g[2 * HS + j]);
        c[i] = cc;
        hout[i] = fast_sigmoid(g[3 * HS + j]) * fast_tanh(cc);
    }
}

__global__ void fill_kernel(float* p, float v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct SessPlan {
    float *a1, *e, *clicks, *Qs, *Ds, *interQ, *interD, *gxq, *gxd, *gates_q, *gates_d, *cq, *cd, *xcat, *wrank, *qp, *feats, *y0, *y1;
    int* m;
    size_t bytes;
};
static SessPlan sess_plan(void* ws, size_t cap, int B, int S, int N, int D, int HS) {
    Workspace a(ws, cap);
    SessPlan p;
    const size_t R = (size_t)B * S * N, BS = (size_t)B * S;
    p.a1 = a.take<float>(R * D);
    p.e = a.take<float>(R);
    p.clicks = a.take<float>(BS * D);
    p.Qs = a.take<float>((size_t)S * B * HS);            // states 0..S-1 (state 0 = zeros)
    p.Ds = a.take<float>((size_t)S * B * HS);
    p.interQ = a.take<float>((size_t)S * B * D);
    p.interD = a.take<float>((size_t)S * B * D);
    p.gxq = a.take<float>(BS * 4 * HS);                  // x W_ih^T + b_ih + b_hh for every (b,t)
    p.gxd = a.take<float>(BS * 4 * HS);
    p.gates_q = a.take<float>((size_t)B * 4 * HS);
    p.gates_d = a.take<float>((size_t)B * 4 * HS);
    p.cq = a.take<float>((size_t)B * HS);
    p.cd = a.take<float>((size_t)B * HS);
    p.xcat = a.take<float>(BS * (D + 2 * HS));
    p.wrank = a.take<float>((size_t)D * (D + 2 * HS));
    p.qp = a.take<float>(BS * D);
    p.feats = a.take<float>(R * 4 * D);
    p.y0 = a.take<float>(R * 256);
    p.y1 = a.take<float>(R * 128);
    p.m = a.take<int>(4);
    p.bytes = align_up(a.off, 256);
    return p;
}

static inline dim3 g1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace nir

extern "C" size_t nir_cars_encode_workspace_bytes(int64_t M, int T, int E, const nir_cars_encoder_weights* w) {
    if (!w || M < 0 || T <= 0) return 0;
    return nir::enc_plan(nullptr, 0, M, T, w->H).bytes;
}

extern "C" int nir_cars_encode(const int64_t* ids, const int64_t* lens, int64_t M, int T, const float* table, int64_t V,
                               int E, const nir_cars_encoder_weights* w, void* workspace, size_t workspace_bytes,
                               float* pooled, float* encoded, nir_stream_t stream) {
    using namespace nir;
    hipStream_t st = (hipStream_t)stream;
    NIR_REQUIRE(ids && lens && table && w && pooled, "cars_encode: null pointer");
    NIR_REQUIRE(M >= 0 && T > 0 && V > 0 && E > 0, "cars_encode: bad dims");
    NIR_REQUIRE(nir_bilstm_supported(w->H) && (2 * w->H) % 4 == 0, "cars_encode: hidden size %d unsupported", w->H);
    if (M == 0) return 0;
    const int H = w->H, D = 2 * H;
    EncPlan p = enc_plan(workspace, workspace_bytes, M, T, H);
    if (!workspace || p.bytes > workspace_bytes) {
        set_error("cars_encode: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
        return NIR_ERR_WORKSPACE;
    }
    float* enc = encoded ? encoded : p.enc;
    NIR_PROPAGATE(launch_linear(nullptr, 0, ids, table, E, 1, 1, w->wih, E, w->bih, w->bhh, p.gates, 8 * H, M * T, 8 * H, E, NIR_ACT_NONE, st));
    NIR_PROPAGATE(launch_bilstm(p.gates, lens, w->whh, nullptr, nullptr, enc, nullptr, nullptr, M, T, H, 2, st));
    NIR_PROPAGATE(launch_linear(enc, D, nullptr, nullptr, 0, 0, 0, w->attn0_w, D, w->attn0_b, nullptr, p.a1, D, M * T, D, D, NIR_ACT_TANH, st));
    NIR_PROPAGATE(launch_rowdot(p.a1, D, w->attn3_w, w->attn3_b, p.logit, M * T, D, NIR_ACT_NONE, st));
    {
        ProfScope ps("attn_pool_kernel", st);
        hipLaunchKernelGGL(attn_pool_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, enc, p.logit, lens, M, T, D, pooled);
    }
    NIR_CHECK_LAUNCH("attn_pool_kernel");
    return 0;
}

namespace nir {
struct EncFoldPlan { float *enc, *lpart; size_t bytes; };
static EncFoldPlan enc_fold_plan(void* ws, size_t cap, int64_t M, int T, int H) {
    Workspace a(ws, cap);
    EncFoldPlan p;
    p.enc = a.take<float>((size_t)M * T * 2 * H);
    p.lpart = a.take<float>((size_t)M * T * (2 * H / 16));
    p.bytes = align_up(a.off, 256);
    return p;
}
}  // namespace nir

extern "C" size_t nir_cars_encode_folded_workspace_bytes(int64_t M, int T, const nir_cars_encoder_weights* w) {
    if (!w || M < 0 || T <= 0) return 0;
    return nir::enc_fold_plan(nullptr, 0, M, T, w->H).bytes;
}

// CARS.encode / encode_document over a folded table (nir_lstm_fold_table): 3 launches -- recurrence gathering its gate
// pre-activations by token id, attention MLP GEMM with the tanh + Linear(D,1) epilogue, masked softmax + weighted sum.
extern "C" int nir_cars_encode_folded(const int64_t* ids, const int64_t* lens, int64_t M, int T, const void* folded, int dtype,
                                      int64_t V, const nir_cars_encoder_weights* w, void* workspace, size_t workspace_bytes,
                                      float* pooled, float* encoded, int* err_flag, nir_stream_t stream) {
    using namespace nir;
    hipStream_t st = (hipStream_t)stream;
    NIR_REQUIRE(ids && lens && folded && w && pooled, "cars_encode_folded: null pointer");
    NIR_REQUIRE(M >= 0 && T > 0 && V > 0, "cars_encode_folded: bad dims");
    NIR_REQUIRE(w->H >= 8 && w->H <= 128 && (2 * w->H) % 64 == 0, "cars_encode_folded: hidden size %d unsupported", w->H);
    if (M == 0) return 0;
    const int H = w->H, D = 2 * H, NP = D / 16;
    EncFoldPlan p = enc_fold_plan(workspace, workspace_bytes, M, T, H);
    if (!workspace || p.bytes > workspace_bytes) {
        set_error("cars_encode_folded: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
        return NIR_ERR_WORKSPACE;
    }
    float* enc = encoded ? encoded : p.enc;
    NIR_PROPAGATE(launch_bilstm_folded(folded, dtype, ids, lens, w->whh, enc, err_flag, M, V, T, H, 2, st));
    NIR_PROPAGATE(launch_linear_ex(enc, D, nullptr, nullptr, 0, 0, 0, w->attn0_w, D, w->attn0_b, nullptr, p.lpart, NP, M * T, D, D,
                                   ACT_TANH_ROWDOT16, w->attn3_w, 0, st));
    {
        ProfScope ps("attn_pool2_kernel", st);
        hipLaunchKernelGGL(attn_pool2_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), (size_t)4 * T * 4, st, enc, p.lpart, NP,
                           w->attn3_b, lens, M, T, D, pooled);
    }
    NIR_CHECK_LAUNCH("attn_pool2_kernel");
    return 0;
}

extern "C" size_t nir_cars_session_workspace_bytes(int B, int S, int N, const nir_cars_session_weights* w) {
    if (!w || B < 0 || S <= 0 || N <= 0) return 0;
    return nir::sess_plan(nullptr, 0, B, S, N, w->D, w->HS).bytes;
}

extern "C" int nir_cars_rank_session(const float* pooled_q, const float* pooled_docs, const float* labels, int B, int S,
                                     int N, const nir_cars_session_weights* w, void* workspace, size_t workspace_bytes,
                                     float* click_scores, float* clicks_out, nir_stream_t stream) {
    using namespace nir;
    hipStream_t st = (hipStream_t)stream;
    NIR_REQUIRE(pooled_q && pooled_docs && labels && w && click_scores, "cars_rank_session: null pointer");
    NIR_REQUIRE(B >= 0 && S > 0 && N > 0, "cars_rank_session: bad dims");
    NIR_REQUIRE(N <= 64, "cars_rank_session: %d candidates > 64 unsupported", N);
    NIR_REQUIRE(S <= 64, "cars_rank_session: session length %d > 64 unsupported", S);
    NIR_REQUIRE(w->D % 4 == 0 && w->HS % 4 == 0, "cars_rank_session: D/HS must be multiples of 4");
    if (B == 0) return 0;
    const int D = w->D, HS = w->HS;
    SessPlan p = sess_plan(workspace, workspace_bytes, B, S, N, D, HS);
    if (!workspace || p.bytes > workspace_bytes) {
        set_error("cars_rank_session: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
        return NIR_ERR_WORKSPACE;
    }
    const int64_t BS = (int64_t)B * S, R = BS * N;
    float* clicks = clicks_out ? clicks_out : p.clicks;
    ProfScope ps_all("cars_rank_session[all kernels]", st);
    // The session states depend only on the queries and the clicks, never on the rank outputs, so the two session
    // LSTM chains run first (x projections batched over all steps; the chains run concurrently on two streams) and
    // attention + ranknet are then evaluated ONCE for all (session, step) pairs instead of once per step.
    ForkJoin fj(st);
    fj.fork();
    {   // ---- query-session chain (side stream): Qs[t+1] = LSTM(q_t), t = 0..S-2   (cars.py:378-380)
        hipStream_t qs = fj.side;
        NIR_PROPAGATE(launch_linear(pooled_q, D, nullptr, nullptr, 0, 0, 0, w->sq_wih, D, w->sq_bih, w->sq_bhh, p.gxq, 4 * HS, BS, 4 * HS, D, NIR_ACT_NONE, qs));
        hipLaunchKernelGGL(fill_kernel, g1((int64_t)B * HS), dim3(256), 0, qs, p.Qs, 0.f, (int64_t)B * HS);   // state 0 = zeros
        hipLaunchKernelGGL(fill_kernel, g1((int64_t)B * HS), dim3(256), 0, qs, p.cq, 0.f, (int64_t)B * HS);
        for (int t = 0; t + 1 < S; ++t) {
            const float* gates = p.gxq + (int64_t)t * 4 * HS;      // row b at stride S*4HS
            int64_t gstride = (int64_t)S * 4 * HS;
            if (t > 0) {   // h_0 = 0: the first step needs no recurrent GEMM
                NIR_PROPAGATE(launch_linear_ex(p.Qs + (int64_t)t * B * HS, HS, nullptr, nullptr, 0, 0, 0, w->sq_whh, HS, nullptr, nullptr, p.gates_q, 4 * HS, B, 4 * HS, HS, NIR_ACT_NONE, gates, gstride, qs));
                gates = p.gates_q;
                gstride = 4 * HS;
            }
            hipLaunchKernelGGL(lstm_cell_kernel, g1((int64_t)B * HS), dim3(256), 0, qs, gates, gstride, p.cq, p.Qs + (int64_t)(t + 1) * B * HS, B, HS);
        }
        NIR_CHECK_LAUNCH("session query LSTM");
        NIR_PROPAGATE(launch_linear(p.Qs, HS, nullptr, nullptr, 0, 0, 0, w->sq_attn_w, HS, w->sq_attn_b, nullptr, p.interQ, D, (int64_t)S * B, D, HS, NIR_ACT_NONE, qs));
    }
    // ---- document-session chain (main stream): encode_clicks (cars.py:262-304), then Ds[t+1] = LSTM(clicks_t)
    NIR_PROPAGATE(launch_linear(pooled_docs, D, nullptr, nullptr, 0, 0, 0, w->click0_w, D, w->click0_b, nullptr, p.a1, D, R, D, D, NIR_ACT_TANH, st));
    NIR_PROPAGATE(launch_rowdot(p.a1, D, w->click3_w, w->click3_b, p.e, R, D, NIR_ACT_NONE, st));
    hipLaunchKernelGGL(click_maxcount_kernel, dim3(1), dim3(256), 0, st, labels, (int)BS, N, p.m);
    hipLaunchKernelGGL(click_pool_kernel, dim3((unsigned)((BS + 3) / 4)), dim3(256), 0, st, pooled_docs, p.e, labels, p.m, (int)BS, N, D, clicks);
    NIR_CHECK_LAUNCH("click_pool_kernel");
    NIR_PROPAGATE(launch_linear(clicks, D, nullptr, nullptr, 0, 0, 0, w->sd_wih, D, w->sd_bih, w->sd_bhh, p.gxd, 4 * HS, BS, 4 * HS, D, NIR_ACT_NONE, st));
    hipLaunchKernelGGL(fill_kernel, g1((int64_t)B * HS), dim3(256), 0, st, p.Ds, 0.f, (int64_t)B * HS);
    hipLaunchKernelGGL(fill_kernel, g1((int64_t)B * HS), dim3(256), 0, st, p.cd, 0.f, (int64_t)B * HS);
    for (int t = 0; t + 1 < S; ++t) {
        const float* gates = p.gxd + (int64_t)t * 4 * HS;
        int64_t gstride = (int64_t)S * 4 * HS;
        if (t > 0) {
            NIR_PROPAGATE(launch_linear_ex(p.Ds + (int64_t)t * B * HS, HS, nullptr, nullptr, 0, 0, 0, w->sd_whh, HS, nullptr, nullptr, p.gates_d, 4 * HS, B, 4 * HS, HS, NIR_ACT_NONE, gates, gstride, st));
            gates = p.gates_d;
            gstride = 4 * HS;
        }
        hipLaunchKernelGGL(lstm_cell_kernel, g1((int64_t)B * HS), dim3(256), 0, st, gates, gstride, p.cd, p.Ds + (int64_t)(t + 1) * B * HS, B, HS);
    }
    NIR_CHECK_LAUNCH("session doc LSTM");
    NIR_PROPAGATE(launch_linear(p.Ds, HS, nullptr, nullptr, 0, 0, 0, w->sd_attn_w, HS, w->sd_attn_b, nullptr, p.interD, D, (int64_t)S * B, D, HS, NIR_ACT_NONE, st));
    const int KR = D + 2 * HS;
    hipLaunchKernelGGL(concat_weights_kernel, g1((int64_t)D * KR), dim3(256), 0, st, w->qproj_w, D, w->shared_w, w->priv1_w, 2 * HS, D, p.wrank);
    fj.join();
    // ---- batched over all (b,t): cross attention (incl. the zero state), rank projection, ranknet (cars.py:348-366,460-520)
    hipLaunchKernelGGL(session_attend_kernel, dim3((unsigned)BS), dim3(256), 0, st, p.interQ, p.Qs, p.interD, p.Ds, pooled_q, B, S, D, HS, p.xcat);
    NIR_CHECK_LAUNCH("session_attend_kernel");
    NIR_PROPAGATE(launch_linear(p.xcat, KR, nullptr, nullptr, 0, 0, 0, p.wrank, KR, w->qproj_b, nullptr, p.qp, D, BS, D, KR, NIR_ACT_NONE, st));
    hipLaunchKernelGGL(rank_feats_kernel, g1(R * D), dim3(256), 0, st, p.qp, pooled_docs, N, D, R, p.feats);
    NIR_CHECK_LAUNCH("rank_feats_kernel");
    // maxout 1024 -> 256 -> 128 -> 1 (pool 2): the pairwise max is fused into the GEMM epilogues
    NIR_PROPAGATE(launch_linear_ex(p.feats, 4 * D, nullptr, nullptr, 0, 0, 0, w->mo0_w, 4 * D, w->mo0_b, nullptr, p.y0, 256, R, 512, 4 * D, ACT_MAXOUT2, nullptr, 0, st));
    NIR_PROPAGATE(launch_linear_ex(p.y0, 256, nullptr, nullptr, 0, 0, 0, w->mo1_w, 256, w->mo1_b, nullptr, p.y1, 128, R, 256, 256, ACT_MAXOUT2, nullptr, 0, st));
    NIR_PROPAGATE(launch_linear_ex(p.y1, 128, nullptr, nullptr, 0, 0, 0, w->mo2_w, 128, w->mo2_b, nullptr, click_scores, 1, R, 2, 128, ACT_MAXOUT2, nullptr, 0, st));
    return 0;
}
