// DUET distributed model, document branch, fused per document tile (neuroir/rankers/duet.py:174-201):
//     emb gather -> conv_d1 (k = 3) + tanh -> max_pool1d(P, stride 1) -> conv_d2 (1x1) + tanh -> Hadamard with the query vector
//     -> Linear over positions (fc2)
// The unfused chain writes conv_d1's [M, DL-2, NF] output, the pooled tensor and conv_d2's output to HBM and reads each back
// (C4: 9.3 GB of counter traffic for 1.1 GB of embedding rows).  Here one workgroup owns 64 consecutive conv positions of one
// document and keeps everything on chip:
//   GEMM 1  D1[64, NFP] = A[64, 3E] W1^T      A = three shifted views of the gathered embedding rows (fp32 from HBM/L2, split into
//                                             two fp16 terms on the way into LDS); W1 = pre-split fp16 term planes stored by the
//                                             host in MFMA-fragment order, streamed L2 -> VGPR (every wave owns 80 filter columns,
//                                             so a W fragment has exactly one consumer and never needs LDS)
//   pool    rows of an accumulator tile live in the wave that owns the column: the 5-row window is one ds_bpermute per value
//   GEMM 2  D2[64, NFP] = P[64, NFP] W2^T     P = pooled tile as fp16 term planes in LDS (fragment order), W2 streamed like W1
//   fc2     partial[doc][tile][f] = sum_rows fc2_w[t] * D2[row][f]   (rows outside the tile's share / the document get weight 0)
// A second tiny kernel folds the tiles: m1[pair][f] = tanh(fc2_b + qv[b][f] * sum_tile partial).
// Arithmetic: two-term fp16 split (x = h1 + 2^-11 h2', three v_mfma_f32_16x16x32_f16 per k-block, two accumulator sets) as in
// gemm3_kernel<., true> -- needs |table|, |weights| < 2^15 (host-checked `bounded`); activations are tanh outputs.
// One workgroup (4 waves, one per SIMD, 160 accumulator registers each) per CU; 104 KB of LDS.
#include <mutex>
#include "common.hpp"

namespace nir {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DF_ROWS = 64;                 // conv positions per workgroup
constexpr int DF_NFP = 320;                 // filter columns (4 waves x 5 tiles x 16)
constexpr int DF_CT = 5;                    // column tiles per wave
constexpr int DF_RT = DF_ROWS / 16;         // row tiles
constexpr int DF_KG = DF_ROWS * 8 + 32;     // halves per k-group block [row][8] (+64 B so the 4 k-groups start in different banks)
constexpr int DF_S2 = DF_NFP / 32;          // k-steps of GEMM 2
constexpr int DF_A_HALVES = 2 * 2 * 4 * DF_KG;          // A stage: [2 buffers][2 terms][4 k-groups][DF_KG]
constexpr int DF_P_HALVES = 2 * DF_S2 * 4 * DF_KG;      // P planes: [2 terms][DF_S2][4 k-groups][DF_KG]
constexpr size_t DF_LDS = (size_t)(DF_A_HALVES + DF_P_HALVES) * 2;

struct DuetDocArgs {
    const int64_t* d_ids;       // [M, DL]
    const float* table;         // [V, E]
    const _Float16* wf1;        // [S1][20 col tiles][2 terms][64 lanes][8]
    const _Float16* wf2;        // [DF_S2][20][2][64][8]
    const float *b1, *b2;       // [NF]
    const float* fc2w;          // [PL]
    float* partial;             // [M][ntile][DF_NFP]
    int E, DL, S1, NF, PL, P, ntile, TPv;
};

__device__ __forceinline__ void df_split_store(unsigned short* base, int row, int kg, int e0, const float4& v) {
    const fp16x2_t a01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), a23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    const float r0 = (v.x - (float)a01[0]) * 2048.0f, r1 = (v.y - (float)a01[1]) * 2048.0f;
    const float r2 = (v.z - (float)a23[0]) * 2048.0f, r3 = (v.w - (float)a23[1]) * 2048.0f;
    const fp16x2_t b01 = __builtin_amdgcn_cvt_pkrtz(r0, r1), b23 = __builtin_amdgcn_cvt_pkrtz(r2, r3);
    unsigned short* d = base + kg * DF_KG + row * 8 + e0;
    *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, a23));
    *reinterpret_cast<uint2*>(d + 4 * DF_KG) = make_uint2(__builtin_bit_cast(unsigned, b01), __builtin_bit_cast(unsigned, b23));
}

__device__ __forceinline__ float df_bperm(float v, int byte_idx) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_idx, __builtin_bit_cast(int, v)));
}

// In-place accumulate in AGPRs.  Written as inline assembly: with the builtin, hipcc assigns the result of each accumulator chain to a
// different register tuple than its loop-carried input and rotates 28 of the 40 tuples through VGPRs on every k-step (112
// v_accvgpr_* moves per 60 MFMAs).  The operands come straight from ds_read / global_load (s_waitcnt is still compiler-inserted);
// the accumulators are first read by VALU code after DF_MMA_DRAIN.
#define DF_MMA(ACC, A, W) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(W))
#define DF_MMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")

// 12 MFMAs of one column tile (4 row tiles x 3 term products): the cross terms go to acx, the leading term to acc
__device__ __forceinline__ void df_mma_col(f32x4 (&acc)[DF_RT], f32x4 (&acx)[DF_RT], const f16x8 (&af)[DF_RT][2], const f16x8& w0,
                                           const f16x8& w1) {
#pragma unroll
    for (int i = 0; i < DF_RT; ++i) DF_MMA(acx[i], af[i][1], w0);
#pragma unroll
    for (int i = 0; i < DF_RT; ++i) DF_MMA(acx[i], af[i][0], w1);
#pragma unroll
    for (int i = 0; i < DF_RT; ++i) DF_MMA(acc[i], af[i][0], w0);
}

__global__ __launch_bounds__(256, 1) void duet_doc_kernel(DuetDocArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short dsm[];
    unsigned short* As = dsm;
    unsigned short* Pp = dsm + DF_A_HALVES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int64_t doc = blockIdx.x / p.ntile;
    const int tile = (int)(blockIdx.x % p.ntile);
    const int t0 = tile * p.TPv;
    const int E = p.E, S1 = p.S1, K1 = 3 * p.E;

    // ---- A operand: row `arow` of the tile = conv position t0 + arow = tokens t, t+1, t+2 (positions past the document are clamped;
    // their rows only feed pooled rows that get weight 0).  Tap s of the row starts at table + id_s * E; the offsets are pre-biased by
    // the tap's k offset so that element k of the concatenated row is table[off_s + k].
    const int arow = tid >> 2, aq = tid & 3;
    int64_t off0, off1, off2;
    {
        int tok = t0 + arow;
        tok = tok < p.DL - 3 ? tok : p.DL - 3;
        const int64_t* idp = p.d_ids + doc * p.DL + tok;
        off0 = idp[0] * (int64_t)E;
        off1 = idp[1] * (int64_t)E - E;
        off2 = idp[2] * (int64_t)E - 2 * E;
    }
    const float* const table = p.table;
    float4 ra0, ra1;
#define DF_LOAD_A(S)                                                                                      \
    {                                                                                                     \
        int k_ = 32 * (S) + 4 * aq;                                                                       \
        int ka_ = k_ < K1 ? k_ : K1 - 4, kb_ = k_ + 16 < K1 ? k_ + 16 : K1 - 4;                           \
        const int64_t oa_ = ka_ < E ? off0 : (ka_ < 2 * E ? off1 : off2);                                 \
        const int64_t ob_ = kb_ < E ? off0 : (kb_ < 2 * E ? off1 : off2);                                 \
        ra0 = *reinterpret_cast<const float4*>(table + oa_ + ka_);                                        \
        ra1 = *reinterpret_cast<const float4*>(table + ob_ + kb_);                                        \
    }
#define DF_STORE_A(BUF)                                                                                   \
    {                                                                                                     \
        df_split_store(As + (BUF) * (2 * 4 * DF_KG), arow, (aq >> 1), 4 * (aq & 1), ra0);                 \
        df_split_store(As + (BUF) * (2 * 4 * DF_KG), arow, (aq >> 1) + 2, 4 * (aq & 1), ra1);             \
    }
    // ---- W operands: fragment-ordered planes, 1 KB contiguous per wave-level load; one register set, a column tile's pair is
    // re-loaded for the next k-step as soon as its 12 MFMAs are issued (the other four tiles' MFMAs cover the L2 latency)
    const _Float16* wp1 = p.wf1 + ((int64_t)(DF_CT * wave) * 2 * 64 + lane) * 8;
    const _Float16* wp2 = p.wf2 + ((int64_t)(DF_CT * wave) * 2 * 64 + lane) * 8;
    constexpr int WSTEP = 20 * 2 * 64 * 8;
    f16x8 w[DF_CT][2];
    f32x4 acc[DF_CT][DF_RT], acx[DF_CT][DF_RT];
#pragma unroll
    for (int j = 0; j < DF_CT; ++j)
#pragma unroll
        for (int i = 0; i < DF_RT; ++i) {
            acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            acx[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const int foff = g * DF_KG + c16 * 8;       // fragment address of the A-side operand: [term][k-group = lane >> 4][row][8]

    // ================= GEMM 1: conv_d1 =================
    DF_LOAD_A(0)
#pragma unroll
    for (int j = 0; j < DF_CT; ++j) {
        w[j][0] = *reinterpret_cast<const f16x8*>(wp1 + (j * 2) * 512);
        w[j][1] = *reinterpret_cast<const f16x8*>(wp1 + (j * 2 + 1) * 512);
    }
    DF_STORE_A(0)
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < S1; ++s) {
        const int sn = s + 1 < S1 ? s + 1 : S1 - 1;           // the last step re-loads its own operands (branch-free)
        DF_LOAD_A(sn)
        const unsigned short* ab = As + (s & 1) * (2 * 4 * DF_KG) + foff;
        f16x8 af[DF_RT][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < DF_RT; ++i) af[i][t] = *reinterpret_cast<const f16x8*>(ab + t * 4 * DF_KG + i * 128);
        const _Float16* wn = wp1 + (int64_t)sn * WSTEP;
#pragma unroll
        for (int j = 0; j < DF_CT; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            df_mma_col(acc[j], acx[j], af, w[j][0], w[j][1]);
            __builtin_amdgcn_sched_barrier(0);
            w[j][0] = *reinterpret_cast<const f16x8*>(wn + (j * 2) * 512);
            w[j][1] = *reinterpret_cast<const f16x8*>(wn + (j * 2 + 1) * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
        DF_STORE_A((s + 1) & 1)
        __syncthreads();
    }

    DF_MMA_DRAIN();
    // ================= tanh, max-pool over rows, split into the P planes =================
    // C layout of the 16x16 tile: column = lane & 15, row = 4 * (lane >> 4) + r.  Pooled row pr needs rows pr .. pr+P-1: the rest of
    // this lane's quad and the quad of the next 16-lane group (the next row tile's group 0 for group 3), fetched by ds_bpermute.
    {
        const int nb_idx = ((lane + 16) & 63) * 4;
        const int P = p.P;
#pragma unroll
        for (int j = 0; j < DF_CT; ++j) {
            const int col = 80 * wave + 16 * j + c16;
            const float bias = col < p.NF ? p.b1[col] : 0.f;
            float v[DF_RT + 1][4], x[DF_RT + 1][4];
#pragma unroll
            for (int i = 0; i < DF_RT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[i][r] = fast_tanh(fmaf(acx[j][i][r], 1.0f / 2048.0f, acc[j][i][r]) + bias);
                    x[i][r] = df_bperm(v[i][r], nb_idx);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) x[DF_RT][r] = x[DF_RT - 1][r];     // rows past the tile: pooled rows >= 60 are never used
            const int sk = col >> 5, kg = (col >> 3) & 3, e = col & 7;
            unsigned short* dst = Pp + (sk * 4 + kg) * DF_KG + e;
#pragma unroll
            for (int i = 0; i < DF_RT; ++i) {
                float cat[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    cat[r] = v[i][r];
                    cat[4 + r] = g < 3 ? x[i][r] : x[i + 1][r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float m = cat[r];
#pragma unroll
                    for (int q = 1; q < 5; ++q) m = q < P ? fmaxf(m, cat[r + q]) : m;
                    const fp16x2_t h1 = __builtin_amdgcn_cvt_pkrtz(m, 0.f);
                    const fp16x2_t h2 = __builtin_amdgcn_cvt_pkrtz((m - (float)h1[0]) * 2048.0f, 0.f);
                    const int row = 16 * i + 4 * g + r;
                    dst[row * 8] = __builtin_bit_cast(unsigned, h1) & 0xFFFFu;
                    dst[DF_S2 * 4 * DF_KG + row * 8] = __builtin_bit_cast(unsigned, h2) & 0xFFFFu;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DF_CT; ++j)
#pragma unroll
        for (int i = 0; i < DF_RT; ++i) {
            acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            acx[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int j = 0; j < DF_CT; ++j) {
        w[j][0] = *reinterpret_cast<const f16x8*>(wp2 + (j * 2) * 512);
        w[j][1] = *reinterpret_cast<const f16x8*>(wp2 + (j * 2 + 1) * 512);
    }
    __syncthreads();

    // ================= GEMM 2: conv_d2 (A = P planes, static in LDS: no barriers) =================
#pragma unroll 1
    for (int s2 = 0; s2 < DF_S2; ++s2) {
        const int sn = s2 + 1 < DF_S2 ? s2 + 1 : DF_S2 - 1;
        f16x8 af[DF_RT][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < DF_RT; ++i)
                af[i][t] = *reinterpret_cast<const f16x8*>(Pp + (t * DF_S2 + s2) * 4 * DF_KG + foff + i * 128);
        const _Float16* wn = wp2 + (int64_t)sn * WSTEP;
#pragma unroll
        for (int j = 0; j < DF_CT; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            df_mma_col(acc[j], acx[j], af, w[j][0], w[j][1]);
            __builtin_amdgcn_sched_barrier(0);
            w[j][0] = *reinterpret_cast<const f16x8*>(wn + (j * 2) * 512);
            w[j][1] = *reinterpret_cast<const f16x8*>(wn + (j * 2 + 1) * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    DF_MMA_DRAIN();
    // ================= tanh, fc2 over the tile's rows =================
    {
        float wrow[DF_RT][4];
#pragma unroll
        for (int i = 0; i < DF_RT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pr = 16 * i + 4 * g + r;
                const bool ok = pr < p.TPv && t0 + pr < p.PL;
                wrow[i][r] = p.fc2w[ok ? t0 + pr : 0] * (ok ? 1.0f : 0.0f);
            }
        float* out = p.partial + ((int64_t)blockIdx.x) * DF_NFP;
#pragma unroll
        for (int j = 0; j < DF_CT; ++j) {
            const int col = 80 * wave + 16 * j + c16;
            const float bias = col < p.NF ? p.b2[col] : 0.f;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < DF_RT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sum = fmaf(wrow[i][r], fast_tanh(fmaf(acx[j][i][r], 1.0f / 2048.0f, acc[j][i][r]) + bias), sum);
            sum += df_bperm(sum, ((lane + 16) & 63) * 4);
            sum += df_bperm(sum, ((lane + 32) & 63) * 4);
            if (g == 0) out[col] = sum;
        }
    }
}

// m1[pair][f] = tanh(fc2_b + qv[b][f] * sum_tile partial[pair][tile][f])
__global__ void duet_doc_finish_kernel(const float* __restrict__ partial, const float* __restrict__ qv, const float* __restrict__ fc2b, int N,
                                       int NF, int ntile, int64_t total, float* __restrict__ m1) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t pair = i / NF;
    const int f = (int)(i % NF);
    float s = 0.f;
    for (int t = 0; t < ntile; ++t) s += partial[(pair * ntile + t) * DF_NFP + f];
    m1[i] = fast_tanh(fc2b[0] + qv[(pair / N) * NF + f] * s);
}

// Tiling of a document: 64 conv positions per workgroup give 64 - (P-1) pooled rows; the pooled rows are shared out evenly.
void duet_doc_tiling(int DL, int P, int* ntile, int* tpv) {
    const int PL = DL - 2 - P + 1, cap = DF_ROWS - (P - 1);
    *ntile = (PL + cap - 1) / cap;
    *tpv = (PL + *ntile - 1) / *ntile;
}

bool duet_doc_usable(int NF, int P, int E, int DL, int K1P) {
    return NF <= DF_NFP && NF % 4 == 0 && P >= 1 && P <= 5 && E % 4 == 0 && E >= 4 && DL >= P + 2 && K1P % 32 == 0 && K1P >= 3 * E;
}

size_t duet_doc_partial_floats(int64_t M, int DL, int P) {
    int nt, tpv;
    duet_doc_tiling(DL, P, &nt, &tpv);
    return (size_t)M * nt * DF_NFP;
}

int launch_duet_doc(const int64_t* d_ids, const float* table, int E, int DL, int64_t M, int N, const void* wf1, int K1P, const void* wf2,
                    const float* b1, const float* b2, const float* fc2w, const float* fc2b, const float* qv, int NF, int P, float* partial,
                    float* m1, hipStream_t st) {
    NIR_REQUIRE(duet_doc_usable(NF, P, E, DL, K1P), "duet_doc: unsupported shape NF=%d pool=%d E=%d DL=%d K1P=%d", NF, P, E, DL, K1P);
    if (M == 0) return 0;
    DuetDocArgs a;
    a.d_ids = d_ids; a.table = table; a.wf1 = (const _Float16*)wf1; a.wf2 = (const _Float16*)wf2; a.b1 = b1; a.b2 = b2; a.fc2w = fc2w;
    a.partial = partial; a.E = E; a.DL = DL; a.S1 = K1P / 32; a.NF = NF; a.PL = DL - 2 - P + 1; a.P = P;
    duet_doc_tiling(DL, P, &a.ntile, &a.TPv);
    static std::once_flag once;
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)duet_doc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DF_LDS); });
    {
        ProfScope ps(prof_shape_name("duet_doc_kernel", M * a.ntile * DF_ROWS, NF, 3 * E), st);
        hipLaunchKernelGGL(duet_doc_kernel, dim3((unsigned)(M * a.ntile)), dim3(256), DF_LDS, st, a);
    }
    NIR_CHECK_LAUNCH("duet_doc_kernel");
    {
        const int64_t total = M * NF;
        ProfScope ps("duet_doc_finish_kernel", st);
        hipLaunchKernelGGL(duet_doc_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, partial, qv, fc2b, N, NF, a.ntile, total,
                           m1);
    }
    NIR_CHECK_LAUNCH("duet_doc_finish_kernel");
    return 0;
}

}  // namespace nir
