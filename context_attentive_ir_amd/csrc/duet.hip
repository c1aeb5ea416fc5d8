// DUET (neuroir/rankers/duet.py): local model (:77-121) + distributed model (:148-208).
//
// local : M[j,i] = [d_j == q_i]  (PAD==PAD counts);  U[f,i] = tanh(b_f + sum_j W[f,j] M[j,i]) -- evaluated
//         SPARSELY: only matching (j,i) add the contiguous weight row Wt[j][:] (the reference runs a dense
//         Conv1d over a [B*N, DL, QL] 0/1 tensor);  then fc1 over i, fc2 (GEMM), fc3 (row dot), tanh each.
// dist  : conv_q / conv_d1 = Conv1d(E->NF, k=3) as fp32-MFMA GEMMs with K = 3E whose A rows are gathered from
//         the embedding table on the fly (no [B*N, DL, E] block is materialised); max-pool(5, stride 1) over t;
//         conv_d2 (1x1) as GEMM; Hadamard with the query vector, Linear over positions, fc3, fc4.
// The document branch (conv_d1 -> pool -> conv_d2 -> Hadamard . fc2) runs as ONE kernel per document tile when the pack carries
// the fragment-ordered weight planes (duet_fused.hip); the GEMM-per-layer chain below it is the general fallback.
#include "common.hpp"

namespace nir {

int launch_linear(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                  int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                  int64_t ldc, int64_t M, int N, int K, int act, hipStream_t st);
int launch_linear_planes(const void* a1, const void* a2, int64_t lda, const int64_t* ids, int64_t rows_per_seq, int64_t seq_stride, int EP,
                         int taps, const void* w1, const void* w2, int64_t ldw, const float* bias, float* c, int64_t ldc, int64_t M, int N, int K,
                         int act, const float* add, int64_t ldadd, hipStream_t st);
int launch_rowdot(const float* x, int64_t ldx, const float* w, const float* b, float* out, int64_t M, int K, int act,
                  hipStream_t st);
// duet_fused.hip
bool duet_doc_usable(int NF, int P, int E, int DL, int K1P);
size_t duet_doc_partial_floats(int64_t M, int DL, int P, bool planes, int EPT);
bool duet_planes_fit(int EPT);
int launch_duet_doc(const int64_t* d_ids, const float* table, int E, int DL, int64_t M, int N, const void* wf1, int K1P, const void* wf2,
                    const void* ftab, const void* wf1c, int EPT, const float* b1, const float* b2, const float* fc2w, const float* fc2b,
                    const float* qv, int NF, int P, float* partial, float* m1, hipStream_t st);

static bool duet_table_planes(const nir_duet_weights* w, int E) {
    return w->ftable && w->fw1c && w->EPT >= E && duet_planes_fit(w->EPT);
}

static bool duet_fused(const nir_duet_weights* w, int E, int DL) {
    return w->bounded && w->fw1 && w->fw2 && !tun(g_tun.duet_unfused) && !tun(g_tun.exact_f32) && duet_doc_usable(w->NF, w->pool, E, DL, w->K1P);
}

// one workgroup per pair: u[pair][f] = tanh(fc1_b + sum_i fc1_w[i] * tanh(conv_b[f] + sum_{j: d_j==q_i} Wt[j][f]))
// The QL x DL id comparisons do not depend on f: they are done ONCE per pair (a wave per query position, ballot +
// mbcnt compaction keeps the matching j in ascending order, so the sums below are taken in the reference's order)
// and every filter thread then walks only the (few) matches.  The first version repeated all QL*DL comparisons in
// every one of the NF filter threads (C4: 480 us per launch).
__global__ __launch_bounds__(256) void duet_local_kernel(const int64_t* __restrict__ q_ids, const int64_t* __restrict__ d_ids,
                                                         const float* __restrict__ wt /*[DL][NF]*/, const float* __restrict__ cb,
                                                         const float* __restrict__ fc1w, const float* __restrict__ fc1b, int N,
                                                         int QL, int DL, int NF, float* __restrict__ u) {
    extern __shared__ int64_t dsh[];              // [DL] doc ids | [QL] query ids | int cnt[QL] | int list[QL][DL]
    int64_t* qsh = dsh + DL;
    int* cnt = (int*)(qsh + QL);
    int* list = cnt + QL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pair = blockIdx.x;
    const int b = (int)(pair / N);
    for (int j = threadIdx.x; j < DL; j += 256) dsh[j] = d_ids[pair * DL + j];
    for (int i = threadIdx.x; i < QL; i += 256) qsh[i] = q_ids[(int64_t)b * QL + i];
    __syncthreads();
    for (int i = wave; i < QL; i += 4) {          // ordered compaction of {j : d_j == q_i}
        const int64_t qid = qsh[i];
        int n = 0;
        for (int j0 = 0; j0 < DL; j0 += 64) {
            const int j = j0 + lane;
            const bool hit = j < DL && dsh[j] == qid;
            const unsigned long long m = __ballot(hit);
            if (hit) list[i * DL + n + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = j;
            n += __popcll(m);
        }
        if (lane == 0) cnt[i] = n;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < NF; f += 256) {
        float acc = fc1b[0];
        for (int i = 0; i < QL; ++i) {
            float s = cb[f];
            const int n = cnt[i];
            for (int m = 0; m < n; ++m) s += wt[(int64_t)list[i * DL + m] * NF + f];   // coalesced row read
            acc = fmaf(fc1w[i], fast_tanh(s), acc);
        }
        u[pair * NF + f] = fast_tanh(acc);
    }
}

// out[r][f] = max_{t < T} x[r][t][f]      (global max-pool of the query conv, duet.py:178)
__global__ __launch_bounds__(256) void colmax_kernel(const float* x, float* out, int T, int NF) {
    const int64_t r = blockIdx.x;
    for (int f = threadIdx.x; f < NF; f += 256) {
        float m = -INFINITY;
        for (int t = 0; t < T; ++t) m = fmaxf(m, x[(r * T + t) * NF + f]);
        out[r * NF + f] = m;
    }
}

// pooled[m][t][f] = max_{dt < P} x[m][t+dt][f], t < Tin-P+1   (max_pool1d(P, stride 1), duet.py:180); float4 over f.
// A thread produces a run of TR consecutive positions of one 16-byte column: TR + P - 1 loads for TR outputs (the
// one-output-per-thread form issued P loads per output and ran at 4.1 TB/s of HBM traffic).
constexpr int MP_TR = 8, MP_PMAX = 8;
__global__ __launch_bounds__(256) void maxpool_t_kernel(const float* __restrict__ x, float* __restrict__ out, int Tin, int P, int NF4,
                                                        int64_t M) {
    const int Tout = Tin - P + 1;
    const int nrun = (Tout + MP_TR - 1) / MP_TR;
    const int64_t total = M * nrun * NF4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int f4 = (int)(e % NF4);
        const int64_t r = e / NF4;
        const int run = (int)(r % nrun);
        const int64_t m = r / nrun;
        const int t0 = run * MP_TR;
        const float4* src = reinterpret_cast<const float4*>(x) + (m * Tin + t0) * NF4 + f4;
        float4 v[MP_TR + MP_PMAX - 1];
#pragma unroll
        for (int k = 0; k < MP_TR + MP_PMAX - 1; ++k)
            v[k] = (k < MP_TR + P - 1 && t0 + k < Tin) ? src[(int64_t)k * NF4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        float4* dst = reinterpret_cast<float4*>(out) + (m * Tout + t0) * NF4 + f4;
#pragma unroll
        for (int k = 0; k < MP_TR; ++k) {
            if (t0 + k < Tout) {
                float4 a = v[k];
#pragma unroll
                for (int dt = 1; dt < MP_PMAX; ++dt)
                    if (dt < P) {
                        a.x = fmaxf(a.x, v[k + dt].x); a.y = fmaxf(a.y, v[k + dt].y);
                        a.z = fmaxf(a.z, v[k + dt].z); a.w = fmaxf(a.w, v[k + dt].w);
                    }
                dst[(int64_t)k * NF4] = a;
            }
        }
    }
}

// The same pooling writing its output as the two fp16 term planes the pre-split GEMM consumes (rows padded to EP = 4*NF4P
// columns, padding zeroed): conv_d2's A operand needs no split pass and no fp32 round trip.
__global__ __launch_bounds__(256) void maxpool_t_planes_kernel(const float* __restrict__ x, _Float16* __restrict__ o1, _Float16* __restrict__ o2,
                                                               int Tin, int P, int NF4, int NF4P, int64_t M) {
    const int Tout = Tin - P + 1;
    const int nrun = (Tout + MP_TR - 1) / MP_TR;
    const int64_t total = M * nrun * NF4P;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int f4 = (int)(e % NF4P);
        const int64_t r = e / NF4P;
        const int run = (int)(r % nrun);
        const int64_t m = r / nrun;
        const int t0 = run * MP_TR;
        const bool real = f4 < NF4;
        const float4* src = reinterpret_cast<const float4*>(x) + (m * Tin + t0) * NF4 + (real ? f4 : 0);
        float4 v[MP_TR + MP_PMAX - 1];
#pragma unroll
        for (int k = 0; k < MP_TR + MP_PMAX - 1; ++k)
            v[k] = (k < MP_TR + P - 1 && t0 + k < Tin) ? src[(int64_t)k * NF4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int k = 0; k < MP_TR; ++k) {
            if (t0 + k < Tout) {
                float4 a = v[k];
#pragma unroll
                for (int dt = 1; dt < MP_PMAX; ++dt)
                    if (dt < P) {
                        a.x = fmaxf(a.x, v[k + dt].x); a.y = fmaxf(a.y, v[k + dt].y);
                        a.z = fmaxf(a.z, v[k + dt].z); a.w = fmaxf(a.w, v[k + dt].w);
                    }
                if (!real) a = make_float4(0.f, 0.f, 0.f, 0.f);
                typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
                const h2_t p01 = __builtin_amdgcn_cvt_pkrtz(a.x, a.y), p23 = __builtin_amdgcn_cvt_pkrtz(a.z, a.w);
                const h2_t q01 = __builtin_amdgcn_cvt_pkrtz((a.x - (float)p01[0]) * 2048.f, (a.y - (float)p01[1]) * 2048.f);
                const h2_t q23 = __builtin_amdgcn_cvt_pkrtz((a.z - (float)p23[0]) * 2048.f, (a.w - (float)p23[1]) * 2048.f);
                const int64_t off = ((m * Tout + t0 + k) * NF4P + f4) * 4;
                *reinterpret_cast<uint2*>(o1 + off) = make_uint2(__builtin_bit_cast(unsigned, p01), __builtin_bit_cast(unsigned, p23));
                *reinterpret_cast<uint2*>(o2 + off) = make_uint2(__builtin_bit_cast(unsigned, q01), __builtin_bit_cast(unsigned, q23));
            }
        }
    }
}

// m1[pair][f] = tanh(fc2_b + sum_t fc2_w[t] * qv[b][f] * dd[pair][t][f])     (Hadamard + Linear over positions)
// One workgroup per pair streams its [T, NF] block once: 16-byte lanes over f, the positions split over
// 256 / (NF/4) thread groups with 4 loads in flight each, partial sums folded through LDS.  (The first version gave a
// thread one f and walked all T positions with a single dependent 4-byte load in flight: 2.4 TB/s.)
__global__ __launch_bounds__(256) void duet_hadamard_kernel(const float* __restrict__ dd, const float* __restrict__ qv,
                                                            const float* __restrict__ fc2w, const float* __restrict__ fc2b, int N,
                                                            int T, int NF, float* __restrict__ m1) {
    extern __shared__ __attribute__((aligned(16))) float4 hsm[];       // [groups][NF/4]
    const int64_t pair = blockIdx.x;
    const int b = (int)(pair / N);
    const int NF4 = NF >> 2, groups = 256 / NF4;
    const int f4 = threadIdx.x % NF4, tg = threadIdx.x / NF4;
    if (tg < groups) {
        const float4 q = reinterpret_cast<const float4*>(qv + (int64_t)b * NF)[f4];
        const float4* src = reinterpret_cast<const float4*>(dd + pair * T * (int64_t)NF) + f4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int t = tg;
        for (; t + 3 * groups < T; t += 4 * groups) {
            float4 v[4];
            float wv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = src[(int64_t)(t + k * groups) * NF4]; wv[k] = fc2w[t + k * groups]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc.x = fmaf(wv[k], q.x * v[k].x, acc.x); acc.y = fmaf(wv[k], q.y * v[k].y, acc.y);
                acc.z = fmaf(wv[k], q.z * v[k].z, acc.z); acc.w = fmaf(wv[k], q.w * v[k].w, acc.w);
            }
        }
        for (; t < T; t += groups) {
            const float4 v = src[(int64_t)t * NF4];
            const float wv = fc2w[t];
            acc.x = fmaf(wv, q.x * v.x, acc.x); acc.y = fmaf(wv, q.y * v.y, acc.y);
            acc.z = fmaf(wv, q.z * v.z, acc.z); acc.w = fmaf(wv, q.w * v.w, acc.w);
        }
        hsm[tg * NF4 + f4] = acc;
    }
    __syncthreads();
    if (threadIdx.x < NF4) {
        float4 a = hsm[threadIdx.x];
        for (int g2 = 1; g2 < groups; ++g2) {
            const float4 o = hsm[g2 * NF4 + threadIdx.x];
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        const float bias = fc2b[0];
        reinterpret_cast<float4*>(m1 + pair * NF)[threadIdx.x] =
            make_float4(fast_tanh(a.x + bias), fast_tanh(a.y + bias), fast_tanh(a.z + bias), fast_tanh(a.w + bias));
    }
}

__global__ void add2_kernel(const float* a, const float* b, float* out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

struct DuetPlan {
    float *u, *v, *sloc, *cq, *qmax, *qv, *cd, *pooled, *dd, *m1, *m2, *sdist;
    size_t bytes;
};

static DuetPlan duet_plan(void* ws, size_t cap, int B, int N, int QL, int DL, int NF, int P, bool fused, bool tplanes, int EPT) {
    Workspace a(ws, cap);
    const size_t M = (size_t)B * N;
    const int Tc = DL - 2, Tp = Tc - P + 1;
    DuetPlan p;
    p.u = a.take<float>(M * NF);
    p.v = a.take<float>(M * NF);
    p.sloc = a.take<float>(M);
    p.cq = a.take<float>((size_t)B * (QL - 2) * NF);
    p.qmax = a.take<float>((size_t)B * NF);
    p.qv = a.take<float>((size_t)B * NF);
    if (fused) {                                      // the fused kernel keeps conv_d1 / pooled / conv_d2 on chip: only per-tile fc2 partials
        p.cd = a.take<float>(duet_doc_partial_floats((int64_t)M, DL, P, tplanes, EPT));
        p.pooled = p.dd = nullptr;
    } else {
        p.cd = a.take<float>(M * Tc * NF);
        p.pooled = a.take<float>(M * Tp * (NF + 8));  // fp32 [M*Tp, NF] or two fp16 planes [M*Tp, EP <= NF + 8]
        p.dd = a.take<float>(M * Tp * NF);
    }
    p.m1 = a.take<float>(M * NF);
    p.m2 = a.take<float>(M * NF);
    p.sdist = a.take<float>(M);
    p.bytes = align_up(a.off, 256);
    return p;
}

}  // namespace nir

extern "C" size_t nir_duet_workspace_bytes(int B, int N, int QL, int DL, int E, const nir_duet_weights* w) {
    if (!w || B < 0 || N <= 0 || QL < 3 || DL < w->pool + 2) return 0;
    return nir::duet_plan(nullptr, 0, B, N, QL, DL, w->NF, w->pool, nir::duet_fused(w, E, DL), nir::duet_table_planes(w, E), w->EPT).bytes;
}

extern "C" int nir_duet_score(const int64_t* q_ids, const int64_t* d_ids, int B, int N, int QL, int DL,
                              const float* table, int64_t V, int E, const nir_duet_weights* w, void* workspace,
                              size_t workspace_bytes, float* scores, float* local_out, float* dist_out,
                              nir_stream_t stream) {
    using namespace nir;
    hipStream_t st = (hipStream_t)stream;
    NIR_REQUIRE(q_ids && d_ids && table && w && scores, "duet: null pointer");
    NIR_REQUIRE(B >= 0 && N > 0 && V > 0 && E > 0, "duet: bad dims");
    NIR_REQUIRE(QL >= 3, "duet: query length %d < dist_filter_size 3 (Conv1d would be empty)", QL);
    NIR_REQUIRE(DL >= w->pool + 2, "duet: doc length %d too short for conv(3) + max_pool(%d)", DL, w->pool);
    NIR_REQUIRE(w->pool >= 1 && w->pool <= 8, "duet: pool_size %d unsupported (1..8)", w->pool);
    NIR_REQUIRE(w->NF % 4 == 0 && w->NF <= 1024, "duet: nfilters %d must be a multiple of 4 and <= 1024", w->NF);
    NIR_REQUIRE((size_t)(DL + QL) * 8 + (size_t)(QL + (size_t)QL * DL) * 4 <= 64 * 1024,
                "duet: QL=%d x DL=%d exact-match lists exceed 64 KB of LDS", QL, DL);
    if (B == 0) return 0;
    const int NF = w->NF, P = w->pool, Tc = DL - 2, Tp = Tc - P + 1;
    const bool fused = duet_fused(w, E, DL);
    const bool tplanes = duet_table_planes(w, E);
    DuetPlan p = duet_plan(workspace, workspace_bytes, B, N, QL, DL, NF, P, fused, tplanes, w->EPT);
    if (!workspace || p.bytes > workspace_bytes) {
        set_error("duet: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
        return NIR_ERR_WORKSPACE;
    }
    const int64_t M = (int64_t)B * N;
    const int bnd = w->bounded ? 0x100 : 0;          // ACT_BOUNDED (gemm.hip): operands < 2^15 -> fp16 two-term split allowed
    float* sloc = local_out ? local_out : p.sloc;
    float* sdist = dist_out ? dist_out : p.sdist;
    // ---- local model (duet.py:77-121)
    {
        ProfScope ps("duet_local_kernel", st);
        hipLaunchKernelGGL(duet_local_kernel, dim3((unsigned)M), dim3(256), (size_t)(DL + QL) * 8 + (size_t)(QL + QL * DL) * 4, st, q_ids, d_ids,
                           w->l_conv_w, w->l_conv_b, w->l_fc1_w, w->l_fc1_b, N, QL, DL, NF, p.u);
    }
    NIR_CHECK_LAUNCH("duet_local_kernel");
    NIR_PROPAGATE(launch_linear(p.u, NF, nullptr, nullptr, 0, 0, 0, w->l_fc2_w, NF, w->l_fc2_b, nullptr, p.v, NF, M, NF, NF, NIR_ACT_TANH, st));
    NIR_PROPAGATE(launch_rowdot(p.v, NF, w->l_fc3_w, w->l_fc3_b, sloc, M, NF, NIR_ACT_TANH, st));
    // ---- distributed model, query side (duet.py:172,178,183)
    NIR_PROPAGATE(launch_linear(nullptr, 0, q_ids, table, E, QL - 2, QL, w->convq_w, 3 * E, w->convq_b, nullptr, p.cq, NF, (int64_t)B * (QL - 2), NF, 3 * E, NIR_ACT_TANH, st));
    {
        ProfScope ps("colmax_kernel", st);
        hipLaunchKernelGGL(colmax_kernel, dim3(B), dim3(256), 0, st, p.cq, p.qmax, QL - 2, NF);
    }
    NIR_CHECK_LAUNCH("colmax_kernel");
    NIR_PROPAGATE(launch_linear(p.qmax, NF, nullptr, nullptr, 0, 0, 0, w->fc1_w, NF, w->fc1_b, nullptr, p.qv, NF, B, NF, NF, NIR_ACT_TANH, st));
    // ---- distributed model, document side (duet.py:174,180,185)
    const bool planes = w->bounded && w->EP > 0 && w->table_h1 && w->table_h2 && w->convd1_h1 && w->convd1_h2 && w->convd2_h1 && w->convd2_h2 &&
                        w->EP % 8 == 0 && w->EP >= NF && w->EP <= NF + 8 && w->EP >= E;
    if (fused) {
        NIR_PROPAGATE(launch_duet_doc(d_ids, table, E, DL, M, N, w->fw1, w->K1P, w->fw2, tplanes ? w->ftable : nullptr, w->fw1c, w->EPT, w->convd1_b,
                                      w->convd2_b, w->fc2_w, w->fc2_b, p.qv, NF, P, p.cd, p.m1, st));
    } else if (planes) {
        // pre-split fp16 term planes end to end: table planes gathered by id (3 taps) -> conv_d1 + tanh (fp32) -> pooling writes
        // planes -> conv_d2 + tanh; no operand is split inside a GEMM
        const int EP = w->EP;
        NIR_PROPAGATE(launch_linear_planes(w->table_h1, w->table_h2, EP, d_ids, Tc, DL, EP, 3, w->convd1_h1, w->convd1_h2, 3 * EP, w->convd1_b, p.cd, NF,
                                           M * Tc, NF, 3 * EP, NIR_ACT_TANH, nullptr, 0, st));
        _Float16* pp1 = reinterpret_cast<_Float16*>(p.pooled);
        _Float16* pp2 = pp1 + (size_t)M * Tp * EP;
        {
            const int64_t total = (int64_t)M * ((Tp + MP_TR - 1) / MP_TR) * (EP / 4);
            ProfScope ps("maxpool_t_planes_kernel", st);
            hipLaunchKernelGGL(maxpool_t_planes_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20)), dim3(256), 0, st, p.cd, pp1,
                               pp2, Tc, P, NF / 4, EP / 4, (int64_t)M);
        }
        NIR_CHECK_LAUNCH("maxpool_t_planes_kernel");
        NIR_PROPAGATE(launch_linear_planes(pp1, pp2, EP, nullptr, 0, 0, 0, 0, w->convd2_h1, w->convd2_h2, EP, w->convd2_b, p.dd, NF, M * Tp, NF, EP,
                                           NIR_ACT_TANH, nullptr, 0, st));
    } else {
        NIR_PROPAGATE(launch_linear(nullptr, 0, d_ids, table, E, Tc, DL, w->convd1_w, 3 * E, w->convd1_b, nullptr, p.cd, NF, M * Tc, NF, 3 * E, NIR_ACT_TANH | bnd, st));
        {
            const int64_t total = (int64_t)M * ((Tp + MP_TR - 1) / MP_TR) * (NF / 4);
            ProfScope ps("maxpool_t_kernel", st);
            hipLaunchKernelGGL(maxpool_t_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20)), dim3(256), 0, st, p.cd,
                               p.pooled, Tc, P, NF / 4, (int64_t)M);
        }
        NIR_CHECK_LAUNCH("maxpool_t_kernel");
        NIR_PROPAGATE(launch_linear(p.pooled, NF, nullptr, nullptr, 0, 0, 0, w->convd2_w, NF, w->convd2_b, nullptr, p.dd, NF, M * Tp, NF, NF, NIR_ACT_TANH | bnd, st));
    }
    // ---- Hadamard + fc2 over positions, fc3, fc4 (duet.py:187-207)
    if (!fused) {
        ProfScope ps("duet_hadamard_kernel", st);
        hipLaunchKernelGGL(duet_hadamard_kernel, dim3((unsigned)M), dim3(256), (size_t)(256 / (NF / 4)) * (NF / 4) * 16, st, p.dd, p.qv, w->fc2_w,
                           w->fc2_b, N, Tp, NF, p.m1);
        NIR_CHECK_LAUNCH("duet_hadamard_kernel");
    }
    NIR_PROPAGATE(launch_linear(p.m1, NF, nullptr, nullptr, 0, 0, 0, w->fc3_w, NF, w->fc3_b, nullptr, p.m2, NF, M, NF, NF, NIR_ACT_TANH, st));
    NIR_PROPAGATE(launch_rowdot(p.m2, NF, w->fc4_w, w->fc4_b, sdist, M, NF, NIR_ACT_TANH, st));
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, sloc, sdist, scores, M);
    NIR_CHECK_LAUNCH("add2_kernel");
    return 0;
}
