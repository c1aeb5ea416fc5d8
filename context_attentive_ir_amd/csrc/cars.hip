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
constexpr int ACT_BOUNDED = 0x100;      // operands bounded by 2^15: the split-precision GEMM may use its fp16 two-term form
int launch_bilstm_folded(const void* pt, int pt_dtype, const int64_t* ids, const int64_t* lens, const float* whh, float* out,
                         int* err, int64_t M, int64_t V, int T, int H, int ND, hipStream_t st, int out_f16 = 0, const void* whh_frag = nullptr);
int launch_rowdot(const float* x, int64_t ldx, const float* w, const float* b, float* out, int64_t M, int K, int act,
                  hipStream_t st);
int launch_bilstm(const float* gin, const int64_t* lens, const float* whh, const float* h0, const float* c0,
                  float* out, float* hn, float* cn, int64_t M, int T, int H, int ND, hipStream_t st);
// cars_attn.hip
bool attn_pool_fused_usable(int D, int T);
int launch_attn_pool_fused(const float* h, const void* wfrag, const float* b0, const float* w3, const float* b3, const int64_t* lens, int64_t M,
                           int T, float* pooled, int one_term, hipStream_t st, int in_f16 = 0);
bool attn_pool_pipe_selected(int64_t M, int T);
bool bilstm_folded_split_out_ok(int pt_dtype, int H, int T);
int launch_fold_permute(const float* w_ih, const float* b_ih, const float* b_hh, int H, int ndir, int E, float* wperm, float* bperm, int64_t* iota, int64_t niota,
                        hipStream_t st);

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
    float *gates, *enc, *a1, *logit, *wperm, *bperm;
    int64_t* iota;
    size_t bytes;
};
static EncPlan enc_plan(void* ws, size_t cap, int64_t M, int T, int H, int E) {
    Workspace a(ws, cap);
    EncPlan p;
    p.gates = a.take<float>((size_t)M * T * 8 * H);
    p.enc = a.take<float>((size_t)M * T * 2 * H);
    p.a1 = a.take<float>((size_t)M * T * 2 * H);
    p.logit = a.take<float>((size_t)M * T);
    p.wperm = a.take<float>((size_t)8 * H * E);
    p.bperm = a.take<float>((size_t)8 * H);
    p.iota = a.take<int64_t>((size_t)M * T);
    p.bytes = align_up(a.off, 256);
    return p;
}

}  // namespace nir

extern "C" size_t nir_cars_encode_workspace_bytes(int64_t M, int T, int E, const nir_cars_encoder_weights* w) {
    if (!w || M < 0 || T <= 0) return 0;
    return nir::enc_plan(nullptr, 0, M, T, w->H, E).bytes;
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
    EncPlan p = enc_plan(workspace, workspace_bytes, M, T, H, E);
    if (!workspace || p.bytes > workspace_bytes) {
        set_error("cars_encode: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
        return NIR_ERR_WORKSPACE;
    }
    float* enc = encoded ? encoded : p.enc;
    // Round 4: the per-batch form shares the recurrence and the attention pooling of the folded form.  The gather-GEMM writes the gate
    // pre-activations of THIS batch in the folded table's row layout ([dir][unit][gate]: W_ih permuted on the fly, 8H x E elements), and that
    // [M*T, 8H] tensor is handed to the folded recurrence as a "table" of M*T rows with the row number as token id -- same kernels
    // (fp16-split MFMA recurrence with the pre-split W_hh, term pairs to the fused attention pipeline) instead of the round-1 recurrence
    // (347 us per 1 120 documents against 75) and the three-launch attention.  What remains of the per-batch cost is the gather-GEMM itself.
    // (bit 2 of `bounded`: |W_hh| < 2^15, host-checked -- outside it the fp16 split overflows: the exact fp32 recurrence below takes over)
    if (H >= 32 && (2 * H) % 64 == 0 && H <= 128 && M * T < ((int64_t)1 << 31) && (w->bounded & 4) && !tun(g_tun.exact_f32) && !tun(g_tun.nofold_old)) {
        NIR_PROPAGATE(launch_fold_permute(w->wih, w->bih, w->bhh, H, 2, E, p.wperm, p.bperm, p.iota, M * T, st));
        // (bit 1 of `bounded`: |table|, |W_ih| < 2^15 host-checked -> the gather-GEMM takes the fp16 two-term split, 3 MFMAs per product instead of 6)
        NIR_PROPAGATE(launch_linear_ex(nullptr, 0, ids, table, E, 1, 1, p.wperm, E, p.bperm, nullptr, p.gates, 8 * H, M * T, 8 * H, E,
                                       NIR_ACT_NONE | ((w->bounded & 2) ? ACT_BOUNDED : 0), nullptr, 0, st));
        const bool fused_attn = w->attn_frag && (w->bounded & 1) && attn_pool_fused_usable(D, T) && !tun(g_tun.attn_unfused);
        const bool inside = fused_attn && !encoded && attn_pool_pipe_selected(M, T);
        const int enc16 = inside && bilstm_folded_split_out_ok(NIR_DTYPE_F32, H, T) && !tun(g_tun.attn_fp32_rows) ? 2 : 0;
        NIR_PROPAGATE(launch_bilstm_folded(p.gates, NIR_DTYPE_F32, p.iota, lens, w->whh, enc, nullptr, M, M * T, T, H, 2, st, enc16, w->whh_frag));
        if (fused_attn)
            return launch_attn_pool_fused(enc, w->attn_frag, w->attn0_b, w->attn3_w, w->attn3_b, lens, M, T, pooled, 0, st, enc16);
        const int NP = D / 16;
        NIR_PROPAGATE(launch_linear_ex(enc, D, nullptr, nullptr, 0, 0, 0, w->attn0_w, D, w->attn0_b, nullptr, p.a1, NP, M * T, D, D,
                                       ACT_TANH_ROWDOT16 | ((w->bounded & 1) ? ACT_BOUNDED : 0), w->attn3_w, 0, st));
        {
            ProfScope ps("attn_pool2_kernel", st);
            hipLaunchKernelGGL(attn_pool2_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), (size_t)4 * T * 4, st, enc, p.a1, NP, w->attn3_b, lens, M, T, D, pooled);
        }
        NIR_CHECK_LAUNCH("attn_pool2_kernel");
        return 0;
    }
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
    // NIR_DTYPE_F32_SPLIT2 (round 5, opt-in precision tier): the fp32 folded table with h as ONE fp16 term in the recurrent product and the
    // attention GEMM (2 MFMAs per block instead of 3, fp16 rows between the two kernels) -- where both kernels of the large-launch pipeline
    // apply (H = 128, T in {4..64}, enough tiles); anywhere else the call is the plain fp32-accurate one
    const bool split2 = dtype == NIR_DTYPE_F32_SPLIT2;
    if (split2) dtype = NIR_DTYPE_F32;
    const int H = w->H, D = 2 * H, NP = D / 16;
    EncFoldPlan p = enc_fold_plan(workspace, workspace_bytes, M, T, H);
    if (!workspace || p.bytes > workspace_bytes) {
        set_error("cars_encode_folded: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
        return NIR_ERR_WORKSPACE;
    }
    float* enc = encoded ? encoded : p.enc;
    const bool fused_attn = w->attn_frag && (w->bounded & 1) && attn_pool_fused_usable(D, T) && !tun(g_tun.attn_unfused) && !tun(g_tun.exact_f32);
    // bf16 encoder whose per-token states stay inside this call and go to the attention pipeline: they travel as fp16 (the pipeline
    // takes single fp16 terms from a bf16 encoder anyway) -- half the bytes written by the recurrence and read by the pooling
    // fp32-accurate encoder in the same situation: the recurrence already forms the two fp16 terms of every h_t for its own next step and
    // hands exactly those to the pipeline (mode 2: same 4 bytes per element, [4 x term 1 | 4 x term 2] per group of 4 units) -- the
    // pipeline's IO waves then copy 16 bytes per lane into their LDS planes instead of re-splitting fp32 rows (VALU-bound before)
    const bool inside = fused_attn && !encoded && attn_pool_pipe_selected(M, T);
    const int enc16 = !inside ? 0 : (dtype == NIR_DTYPE_BF16 ? (H > 64 ? 1 : 0) :
                                     (bilstm_folded_split_out_ok(dtype, H, T) && !tun(g_tun.attn_fp32_rows) ? (split2 ? 3 : 2) : 0));
    NIR_PROPAGATE(launch_bilstm_folded(folded, dtype, ids, lens, w->whh, enc, err_flag, M, V, T, H, 2, st, enc16, dtype == NIR_DTYPE_F32 ? w->whh_frag : nullptr));
    // enc = o * tanh(c) lies in (-1,1); the attention weights are bounded (checked by the host when it packs them)
    if (fused_attn)
        return launch_attn_pool_fused(enc, w->attn_frag, w->attn0_b, w->attn3_w, w->attn3_b, lens, M, T, pooled, dtype == NIR_DTYPE_BF16, st, enc16);
    NIR_PROPAGATE(launch_linear_ex(enc, D, nullptr, nullptr, 0, 0, 0, w->attn0_w, D, w->attn0_b, nullptr, p.lpart, NP, M * T, D, D,
                                   ACT_TANH_ROWDOT16 | ((w->bounded & 1) ? ACT_BOUNDED : 0), w->attn3_w, 0, st));
    {
        ProfScope ps("attn_pool2_kernel", st);
        hipLaunchKernelGGL(attn_pool2_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), (size_t)4 * T * 4, st, enc, p.lpart, NP,
                           w->attn3_b, lens, M, T, D, pooled);
    }
    NIR_CHECK_LAUNCH("attn_pool2_kernel");
    return 0;
}

