// CARS.apply_pooling fused (neuroir/multitask/cars.py:671-691 with the {q,d}_attn MLP of :200-223 / :238-258):
//     logits = Linear(D,1)(tanh(Linear(D,D)(h)))  ->  masked softmax over the sequence  ->  pooled = sum_t p_t h_t
// for D = 2H = 256.  The layer chain ran this as a split-precision GEMM with a tanh + row-dot epilogue (gemm3h_kernel, 79 us at C3,
// logit partials to HBM) plus a pooling kernel that read the encoder output a second time (23 us).  Here one workgroup owns 64
// consecutive rows of the flattened (sequence, step) axis = 64 / T whole sequences and touches the encoder output once:
//   rows -> split into two fp16 terms, LDS planes in MFMA-fragment order (read again from L2 as fp32 for the weighted sum)
//   GEMM D1[64, 256] = h W0^T with W0 as pre-split fragment-ordered planes streamed L2 -> VGPR (every wave owns 64 columns); the
//        k-loop is written MFMA by MFMA with the loads pinned between them, as in duet_fused.hip (same operand keep-alive fences
//        against the VALU-after-MFMA WAR hazard of inline-assembly MFMAs)
//   tanh, times w3, summed over the wave's columns (in-lane + DPP over the 16 lanes of a row group), the four waves' partials meet in
//        LDS; softmax over each sequence's valid steps; the weighted sum of the fp32 rows is reduced over the waves through LDS.
// Requires D == 256, T in {4, 8, 16, 32, 64}, |W0| < 2^15 (encoder outputs are o * tanh(c), inside (-1, 1)).
#include <algorithm>
#include <mutex>
#include <type_traits>
#include "common.hpp"

namespace nir {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AP_D = 256;                   // 2H
constexpr int AP_ROWS = 64, AP_RT = 4, AP_CT = 4, AP_S = AP_D / 32;
constexpr int AP_KG = AP_ROWS * 8 + 32;     // halves per k-group block [row][8] (+64 B)
constexpr int AP_PLANE_HALVES = 2 * AP_S * 4 * AP_KG;            // [2 terms][8 k-steps][4 k-groups][KG]
constexpr int AP_RED_FLOATS = 4 * 16 * AP_D;                     // weighted-sum partials [4 waves][<= 16 sequences][256] (overlays the planes)
constexpr size_t AP_LDS = (size_t)(AP_PLANE_HALVES * 2 > AP_RED_FLOATS * 4 ? AP_PLANE_HALVES * 2 : AP_RED_FLOATS * 4) + (4 * 64 + 64) * 4;

#define AP_MMA(ACC, A, W) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(W))
#define AP_MMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")

__device__ __forceinline__ void ap_mma_n(int n, f32x4 (&acc)[AP_CT][AP_RT], f32x4 (&acx)[AP_CT][AP_RT], const f16x8 (&af)[AP_RT][2],
                                         const f16x8 (&w)[AP_CT][2]) {
    const int j = n / (3 * AP_RT), ph = (n / AP_RT) % 3, i = n % AP_RT;
    if (ph == 0) AP_MMA(acx[j][i], af[i][1], w[j][0]);
    else if (ph == 1) AP_MMA(acx[j][i], af[i][0], w[j][1]);
    else AP_MMA(acc[j][i], af[i][0], w[j][0]);
}

// max / sum over aligned groups of T lanes (T a power of two, 4..64): DPP inside 16-lane rows, ds_bpermute only across rows
__device__ __forceinline__ float ap_group_max(float v, int T) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    if (T >= 8) v = fmaxf(v, dpp_mov<0x141>(v));
    if (T >= 16) v = fmaxf(v, dpp_mov<0x140>(v));
    if (T >= 32) v = fmaxf(v, __shfl_xor(v, 16));
    if (T >= 64) v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}
__device__ __forceinline__ float ap_group_sum(float v, int T) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    if (T >= 8) v += dpp_mov<0x141>(v);
    if (T >= 16) v += dpp_mov<0x140>(v);
    if (T >= 32) v += __shfl_xor(v, 16);
    if (T >= 64) v += __shfl_xor(v, 32);
    return v;
}

struct AttnPoolArgs {
    const float* h;             // [M*T, 256] encoder output
    const _Float16* wf;         // W0 fragments [8 k-steps][16 col tiles][2 terms][64 lanes][8]
    const float *b0, *w3, *b3;  // [256], [256], [1]
    const int64_t* lens;        // [M] or null
    float* pooled;              // [M, 256]
    int64_t M;
    int T, logT;
    int io_prio;                // pipelined kernel: issue priority of the IO waves (0 = like the MMA waves)
};

__global__ __launch_bounds__(256, 1) void attn_pool_fused_kernel(AttnPoolArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short asm_[];
    unsigned short* Pp = asm_;                                                   // term planes, later the weighted-sum partials
    float* rowpart = reinterpret_cast<float*>(asm_ + (AP_LDS - (4 * 64 + 64) * 4) / 2);   // [4 waves][64 rows]
    float* prob = rowpart + 4 * 64;                                              // [64 rows]
    constexpr int KG = AP_KG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int T = p.T;
    const int64_t row0 = (int64_t)blockIdx.x * AP_ROWS, nrows = p.M * T;

    // ---- the tile's rows: thread (wave, lane) handles columns 4 lane .. 4 lane + 3 of rows wave + 4 j (coalesced 1 KB per wave and row):
    // split into the two fp16 terms (x = h1 + 2^-11 h2') and stored in fragment order [term][k-step][k-group][row][8].  Rows past the
    // end belong to sequences that are never written: clamped, not predicated (a predicated load is a branch per load).
    const _Float16* wp = p.wf + ((int64_t)(AP_CT * wave) * 2 * 64 + lane) * 8;
    constexpr int WSTEP = 16 * 2 * 64 * 8;
    f16x8 w[AP_CT][2], wb[AP_CT][2], afa[AP_RT][2], afb[AP_RT][2];
    {
        float4 hv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int64_t r = row0 + wave + 4 * j;
            r = r < nrows ? r : nrows - 1;
            hv[j] = *reinterpret_cast<const float4*>(p.h + r * AP_D + 4 * lane);
        }
#pragma unroll
        for (int j = 0; j < AP_CT; ++j) {
            w[j][0] = *reinterpret_cast<const f16x8*>(wp + (j * 2) * 512);
            w[j][1] = *reinterpret_cast<const f16x8*>(wp + (j * 2 + 1) * 512);
        }
        const int k = 4 * lane, sk = k >> 5, kg = (k >> 3) & 3, e0 = k & 7;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 v = hv[j];
            const fp16x2_t a01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), a23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
            const fp16x2_t b01 = __builtin_amdgcn_cvt_pkrtz((v.x - (float)a01[0]) * 2048.0f, (v.y - (float)a01[1]) * 2048.0f);
            const fp16x2_t b23 = __builtin_amdgcn_cvt_pkrtz((v.z - (float)a23[0]) * 2048.0f, (v.w - (float)a23[1]) * 2048.0f);
            unsigned short* d = Pp + (sk * 4 + kg) * KG + (wave + 4 * j) * 8 + e0;
            *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, a23));
            *reinterpret_cast<uint2*>(d + AP_S * 4 * KG) = make_uint2(__builtin_bit_cast(unsigned, b01), __builtin_bit_cast(unsigned, b23));
        }
    }
    f32x4 acc[AP_CT][AP_RT], acx[AP_CT][AP_RT];
#pragma unroll
    for (int j = 0; j < AP_CT; ++j)
#pragma unroll
        for (int i = 0; i < AP_RT; ++i) {
            acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            acx[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    __syncthreads();
    const int foff = g * KG + c16 * 8;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < AP_RT; ++i) afa[i][t] = *reinterpret_cast<const f16x8*>(Pp + t * AP_S * 4 * KG + foff + i * 128);

    // ================= D1 = h W0^T (A planes static in LDS: no barriers) =================
    // Two fragment register sets, every k-step written MFMA by MFMA with the next step's loads pinned between them, operand
    // keep-alive fences against the VALU-after-MFMA WAR hazard of the inline-assembly MFMAs: see duet_fused.hip.  (A variant with one
    // set and two workgroups per CU -- 128 + 128 registers -- was slower: 77 us against 62 us at C3.)
#define AP_KEEP_HEAD(WN, AFN)                                                             \
    _Pragma("unroll") for (int j_ = AP_CT - 2; j_ < AP_CT; ++j_) asm volatile("" ::"v"(WN[j_][0]), "v"(WN[j_][1])); \
    _Pragma("unroll") for (int i_ = 0; i_ < AP_RT; ++i_) asm volatile("" ::"v"(AFN[i_][0]), "v"(AFN[i_][1]));
#define AP_KEEP(WC, AFC)                                                                  \
    _Pragma("unroll") for (int j_ = 0; j_ < AP_CT; ++j_) asm volatile("" ::"v"(WC[j_][0]), "v"(WC[j_][1]));   \
    _Pragma("unroll") for (int i_ = 0; i_ < AP_RT; ++i_) asm volatile("" ::"v"(AFC[i_][0]), "v"(AFC[i_][1]));
#define AP_STEP(S, AFC, AFN, WC, WN)                                                      \
    {                                                                                     \
        const int sn_ = (S) + 1 < AP_S ? (S) + 1 : AP_S - 1;                              \
        const _Float16* wn_ = wp + (int64_t)sn_ * WSTEP;                                  \
        const unsigned short* pn_ = Pp + sn_ * 4 * KG + foff;                             \
        _Pragma("clang loop unroll(full)") for (int n_ = 0; n_ < 3 * AP_RT * AP_CT; ++n_) { \
            ap_mma_n(n_, acc, acx, AFC, WC);                                              \
            if (n_ == 8) { AP_KEEP_HEAD(WN, AFN) }                                        \
            if (n_ % 6 == 2 && n_ / 6 < 2 * AP_CT)                                        \
                WN[(n_ / 6) >> 1][(n_ / 6) & 1] = *reinterpret_cast<const f16x8*>(wn_ + (n_ / 6) * 512); \
            if (n_ >= 9 && (n_ - 9) % 5 == 0 && (n_ - 9) / 5 < 2 * AP_RT)                 \
                AFN[((n_ - 9) / 5) % AP_RT][((n_ - 9) / 5) / AP_RT] =                     \
                    *reinterpret_cast<const f16x8*>(pn_ + (((n_ - 9) / 5) / AP_RT) * AP_S * 4 * KG + (((n_ - 9) / 5) % AP_RT) * 128); \
            __builtin_amdgcn_sched_barrier(0);                                            \
        }                                                                                 \
        AP_KEEP(WC, AFC)                                                                  \
    }
#pragma unroll 1
    for (int s = 0; s < AP_S; s += 2) {
        AP_STEP(s, afa, afb, w, wb)
        AP_STEP(s + 1, afb, afa, wb, w)
    }
    AP_MMA_DRAIN();
    // the fp32 rows again for the weighted sum (L2 hits; holding them across the k-loop cost 64 VGPRs and pushed fragments into AGPR
    // spills): issued here, consumed after the softmax
    float4 hv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        int64_t r = row0 + wave + 4 * j;
        r = r < nrows ? r : nrows - 1;
        hv[j] = *reinterpret_cast<const float4*>(p.h + r * AP_D + 4 * lane);
    }

    // ================= logits: tanh, times w3, summed over the columns =================
    {
        constexpr float C2 = 2.8853900817779268f;          // 2 log2(e): tanh(x) = 1 - 2 / (1 + 2^(C2 x))
        float rs[AP_RT][4];
#pragma unroll
        for (int i = 0; i < AP_RT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) rs[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < AP_CT; ++j) {
            const int col = 64 * wave + 16 * j + c16;
            const float bz = p.b0[col] * C2, w3c = p.w3[col];
#pragma unroll
            for (int i = 0; i < AP_RT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = fmaf(fmaf(acx[j][i][r], 1.0f / 2048.0f, acc[j][i][r]), C2, bz);
                    const float th = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z)), 1.0f);
                    rs[i][r] = fmaf(w3c, th, rs[i][r]);
                }
        }
#pragma unroll
        for (int i = 0; i < AP_RT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {                  // sum over the 16 lanes (columns) of the row group
                float v = rs[i][r];
                v += dpp_mov<0xB1>(v);
                v += dpp_mov<0x4E>(v);
                v += dpp_mov<0x141>(v);
                v += dpp_mov<0x140>(v);
                if (c16 == 0) rowpart[wave * 64 + 16 * i + 4 * g + r] = v;
            }
    }
    __syncthreads();
    // ================= masked softmax over each sequence (wave 0: lane = row of the tile) =================
    if (wave == 0) {
        const int64_t r = row0 + lane;
        const int64_t seq = r >> p.logT;                   // T is a power of two: no 64-bit division per lane
        const int t = (int)(r - seq * T);
        int len = T;
        if (p.lens && seq < p.M) len = (int)p.lens[seq];
        len = len < 0 ? 0 : (len > T ? T : len);
        const bool ok = seq < p.M && t < len;
        const float lg = (rowpart[lane] + rowpart[64 + lane]) + (rowpart[128 + lane] + rowpart[192 + lane]) + p.b3[0];
        const float mx = ap_group_max(ok ? lg : -INFINITY, T);
        const float e = ok ? __expf(lg - mx) : 0.f;
        const float den = ap_group_sum(e, T);
        prob[lane] = e / den;                              // len == 0: 0/0 = NaN, like softmax over an all -inf row
    }
    __syncthreads();
    // ================= pooled = sum_t p_t h_t =================
    {
        float4* red = reinterpret_cast<float4*>(Pp);       // [4 waves][RS sequences][64 float4]  (the planes are dead)
        const int RS = AP_ROWS / T, per = T / 4;           // rows of one sequence held by this thread (rows wave + 4 j): T / 4
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 16; ++j) {                     // row wave + 4 j belongs to sequence j / per of the tile
            const float pr = prob[wave + 4 * j];
            a.x = fmaf(pr, hv[j].x, a.x); a.y = fmaf(pr, hv[j].y, a.y); a.z = fmaf(pr, hv[j].z, a.z); a.w = fmaf(pr, hv[j].w, a.w);
            if (((j + 1) & (per - 1)) == 0) {                      // last row of its sequence held by this thread
                red[(wave * RS + (j >> (p.logT - 2))) * 64 + lane] = a;
                a = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
        for (int idx = tid; idx < RS * 64; idx += 256) {
            const int q = idx >> 6, c = idx & 63;
            const int64_t seq = (row0 >> p.logT) + q;
            if (seq < p.M) {
                const float4 a0 = red[(0 * RS + q) * 64 + c], a1 = red[(1 * RS + q) * 64 + c], a2 = red[(2 * RS + q) * 64 + c],
                             a3 = red[(3 * RS + q) * 64 + c];
                float4 o;
                o.x = (a0.x + a1.x) + (a2.x + a3.x); o.y = (a0.y + a1.y) + (a2.y + a3.y);
                o.z = (a0.z + a1.z) + (a2.z + a3.z); o.w = (a0.w + a1.w) + (a2.w + a3.w);
                *reinterpret_cast<float4*>(p.pooled + seq * AP_D + 4 * c) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same computation as a persistent, role-specialised pipeline (many tiles per CU: document encoders).  The single-role kernel
// above spends ~3 us of an 11 us tile in MFMAs; prologue (row loads, split) and epilogue (softmax, weighted sum) leave the matrix
// pipe idle and one workgroup per CU cannot overlap them.  Here a workgroup has 8 waves: waves 0-3 (one per SIMD) only run the GEMM
// and the tanh / row-dot epilogue of tile k, waves 4-7 (one per SIMD) meanwhile finish tile k-1 (softmax, weighted sum, store) and
// stage tile k+1 (load, split, store planes) -- two plane buffers, one barrier per tile:
//     iteration it:   MMA waves: G(it-1)  reads planes[(it-1)&1], writes rowpart[(it-1)&1]
//                     IO waves : S(it-2)  reads rowpart[it&1];   L(it) writes planes[it&1]
// Register budget 256 per lane (2 waves per SIMD): MMA waves 128 accumulator AGPRs + W fragments (two sets, the next k-step's loads
// spread behind the MFMAs) + ONE A-fragment set whose row tile i is re-read for the next k-step 7+ MFMA slots after its last use;
// MFMA order is row tile outermost.  IO waves work in halves of 8 rows (32 registers per half).  The IO waves never synchronise among
// themselves: each computes the 64-row softmax redundantly (lane = row) and owns 64 of the 256 output columns.
// ---------------------------------------------------------------------------------------------------------------------
#ifdef AP_TIMING
__device__ long long ap_dbg[16];
#define AP_T(I) if (blockIdx.x == 7 && lane == 0 && it == 5) ap_dbg[I] = __builtin_readcyclecounter();
#else
#define AP_T(I)
#endif
// Plane layout of the pipeline (round 6: LDS bank conflicts, 7.4 extra cycles per LDS instruction in round 5's PMC captures = a quarter of the
// launch's CU-cycles): [2 terms][8 k-steps S][4 k-groups g][64 row positions][8 halves], blocks of exactly 1 KB, row r of block (S, g) at position
//     r ^ x(S, g),   x = 4 (S & 1) + 2 (g >> 1).
// * MMA waves (ds_read_b128: lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... , bank = (a / 4) mod 64): a group mixes rows {0-3,12-15} of
//   k-group 2j with rows {4-11} of k-group 2j + 1; with the block stride a multiple of 256 B and ONE x for both k-groups of a pair the 16 lanes cover 16
//   distinct 16-byte slots (the old 64-byte pad between blocks shifted k-group 2j + 1 by four slots: 2-way in every group, +4 cycles per read).
// * IO waves (ds_write_b64 / ds_read_b64: 16 / 32 contiguous lanes = ONE row, 8 (S & 1, g) blocks x 2 halves): the same row sits at the same offset of
//   every block, so only a per-block displacement separates the lanes: x spreads the four (S & 1, g >> 1) classes over 4 x 32 B of the 128-byte bank
//   window -- 2-way (k-groups 2j / 2j + 1 must stay congruent for the MMA reads) instead of 4-way on every staging write (+12 cycles each).
constexpr int AP2_KG = AP_ROWS * 8;                              // halves per (S, g) block: 1 KB, no pad
constexpr int AP2_PLANE_HALVES = 2 * AP_S * 4 * AP2_KG;         // [2 terms][8 k-steps][4 k-groups][KG]
constexpr size_t AP2_LDS = (size_t)2 * AP2_PLANE_HALVES * 2 + (2 * 4 * 64 + 4 * 64 + 2 * AP_D) * 4;

// ONE: W0 and the rows enter as single fp16 terms (bf16 encoders).  ROW1 (round 5, the "split2" tier: <false, 1>): the ROWS are single fp16
// terms (the recurrence's one-term hand-over) but W0 keeps its two terms: two MFMAs per fragment pair (row . w1, row . w2').
template <bool ONE, bool ROW1>
__device__ __forceinline__ void ap2_mma_n(int n, f32x4 (&acc)[AP_CT][AP_RT], f32x4 (&acx)[AP_CT][AP_RT], const f16x8 (&af)[AP_RT][2],
                                          const f16x8 (&w)[AP_CT][2]) {
    const int i = n / (3 * AP_CT), ph = (n / AP_CT) % 3, j = n % AP_CT;      // row tile outermost
    if (ph == 2) AP_MMA(acc[j][i], af[i][0], w[j][0]);
    else if (ONE) return;                                                      // leading fp16 term only (bf16 encoders)
    else if (ph == 0) { if (!ROW1) AP_MMA(acx[j][i], af[i][1], w[j][0]); }
    else AP_MMA(acx[j][i], af[i][0], w[j][1]);
}

// ONE: the encoder runs in bf16 (bf16 folded table + bf16 recurrence): h and W0 enter the attention MLP as single fp16 terms (11
// mantissa bits, still 3 more than the bf16 operands upstream) -- one MFMA per fragment pair instead of three, one term plane.
// IN = 1 (implies ONE): the encoder output arrives as fp16 rows (bf16-table recurrence with fp16 output): the IO waves copy it into the
// term plane as it is -- half the HBM read, no conversion.
// IN = 2 (two terms): the fp32-accurate recurrence hands over the two fp16 terms it formed for its own next step (lstm16_pt_h2_kernel,
// out_f16 = 2): per row and group of 4 columns 16 bytes = [4 x leading term | 4 x residual x 2^11].  The IO waves load the same 16 bytes
// per lane as for an fp32 row and store the halves into the two planes -- no split arithmetic (it was 16 x ~14 VALU per lane and tile and
// made the IO waves, not the matrix pipe, the bound of this kernel).
#ifndef AP_FMAMIX
#define AP_FMAMIX 1
#endif
// a += s * (the four fp16 values of u), fp32 accumulate: the fp16 operands are read by v_fma_mix_f32 (op_sel picks the half, op_sel_hi marks
// the source as fp16); exact in fp32 like convert-then-fma
__device__ __forceinline__ void ap_fma_mix(float4& a, const uint2 u, const float s) {
    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a.x) : "v"(u.x), "v"(s));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a.y) : "v"(u.x), "v"(s));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a.z) : "v"(u.y), "v"(s));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a.w) : "v"(u.y), "v"(s));
}

template <bool ONE, int IN>
__global__ __launch_bounds__(512, 1) void attn_pool_pipe_kernel(AttnPoolArgs p, int64_t ntiles) {
    constexpr bool IN16 = IN == 1;
    constexpr bool ROW1 = ONE || IN16;                // the rows have no residual plane
    extern __shared__ __attribute__((aligned(16))) unsigned short asm2_[];
    constexpr int KG = AP2_KG, PH = AP2_PLANE_HALVES;
    float* rowpart = reinterpret_cast<float*>(asm2_ + 2 * PH);      // [2 buffers][4 waves][64 rows]
    float* probw = rowpart + 2 * 4 * 64;                                          // [4 IO waves][64 rows]
    float* bw = probw + 4 * 64;                                                   // [256] b0 * 2 log2(e), [256] w3: read by every tile's epilogue
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w4 = wave & 3;
    // (from global memory the eight values a lane needs were re-loaded at the start of every tile's epilogue: one exposed L2 round trip per tile)
    if (tid < AP_D) bw[tid] = p.b0[tid] * 2.8853900817779268f;
    else bw[tid] = p.w3[tid - AP_D];
    __syncthreads();
    const bool mma_role = wave < 4;
    const int g = lane >> 4, c16 = lane & 15;
    const int T = p.T;
    const int64_t nrows = p.M * T;
    const int64_t G = gridDim.x;
    const int64_t nk = (ntiles - blockIdx.x + G - 1) / G;                         // tiles of this workgroup: blockIdx.x + k G

    if (mma_role) {
        // ================================================= MMA waves =================================================
        // W fragments are addressed as (wave-uniform base + compile-time offset) + one 32-bit lane offset: with a per-lane 64-bit
        // pointer the unrolled last k-steps kept one precomputed pointer per fragment live across the tile loop, 25 registers went to
        // scratch and every reload carried an s_waitcnt vmcnt(0) into the MFMA stream (12 per tile)
        // The MMA wave is the tile's critical chain (k-loop, then ITS epilogue): it issues ahead of the IO wave that shares its SIMD.  Measured with
        // the conflict-free plane layout (tools/attn_micro.py, M = 8 960, same process): layout alone 1.07 x the round-5 kernel (the IO waves, no longer
        // throttled by 4-way staging conflicts, take issue slots at the wrong time), priority alone 1.05 x, both 0.957 x.
        __builtin_amdgcn_s_setprio(3);
        const _Float16* wbase = p.wf + (int64_t)(AP_CT * __builtin_amdgcn_readfirstlane(w4)) * 2 * 64 * 8;
        uint32_t wlane = (uint32_t)lane * 16u;                       // bytes; made opaque once per k-step (below) so that the addresses
        auto ldw = [&](int off_halves) {                             // are recomputed (2 VALU) instead of hoisted out of the tile loop
            return *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(wbase + off_halves) + (uint64_t)wlane);
        };
        constexpr int WSTEP = 16 * 2 * 64 * 8;
        // row position of this lane's A rows inside block (S, g): c16 ^ x(S & 1, g >> 1) -- one offset for even, one for odd k-steps
        const int foff_e = g * KG + (c16 ^ (2 * (g >> 1))) * 8, foff_o = g * KG + (c16 ^ (4 + 2 * (g >> 1))) * 8;
        f16x8 wa[AP_CT][2], wb[AP_CT][2], af[AP_RT][2];
        f32x4 acc[AP_CT][AP_RT], acx[AP_CT][AP_RT];
#pragma unroll
        for (int j = 0; j < AP_CT; ++j) {
            wa[j][0] = ldw((j * 2) * 512);
            wa[j][1] = ONE ? wa[j][0] : ldw((j * 2 + 1) * 512);
        }
        for (int64_t it = 0; it < nk + 2; ++it) {
            if (w4 == 0) { AP_T(0) }
            if (it >= 1 && it <= nk) {
                const unsigned short* Pp = asm2_ + ((it - 1) & 1) * PH;
                float* rp = rowpart + ((it - 1) & 1) * 4 * 64;
#pragma unroll
                for (int j = 0; j < AP_CT; ++j)
#pragma unroll
                    for (int i = 0; i < AP_RT; ++i) {
                        acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        acx[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                for (int i = 0; i < AP_RT - 1; ++i) {      // row tile 3 of a k-step is read at MFMAs 7, 8 of that step
                    af[i][0] = *reinterpret_cast<const f16x8*>(Pp + foff_e + i * 128);
                    af[i][1] = *reinterpret_cast<const f16x8*>(Pp + AP_S * 4 * KG + foff_e + i * 128);
                }
                // k-step S with W set WC; WN receives the fragments of step S+1 (after step 7: step 0 again, for the next tile).  A
                // fragments of row tile i for step S+1 are re-read into af[i] at MFMA 12 i + 19 / + 20 (>= 7 slots after their last use;
                // row tile 3: MFMAs 7 / 8 of the step itself); the last step reads none -- the next tile's planes are not complete yet.
#define AP2_STEP(S, WC, WN, LAST, ODD)                                                    \
                {                                                                         \
                    const int sn_ = ((S) + 1) & (AP_S - 1);                               \
                    asm volatile("" : "+v"(wlane));                                       \
                    const unsigned short* pc_ = Pp + (S) * 4 * KG + ((ODD) ? foff_o : foff_e);  \
                    const unsigned short* pn_ = Pp + sn_ * 4 * KG + ((ODD) ? foff_e : foff_o);  \
                    _Pragma("clang loop unroll(full)") for (int n_ = 0; n_ < 3 * AP_RT * AP_CT; ++n_) { \
                        ap2_mma_n<ONE, ROW1>(n_, acc, acx, af, WC);                             \
                        if (n_ % 6 == 2 && n_ / 6 < 2 * AP_CT && !(ONE && ((n_ / 6) & 1))) { /* ONE: the residual-term fragments are never read */ \
                            asm volatile("" ::"v"(WN[(n_ / 6) >> 1][(n_ / 6) & 1]));      \
                            WN[(n_ / 6) >> 1][(n_ / 6) & 1] = ldw(sn_ * WSTEP + (n_ / 6) * 512);  \
                        }                                                                 \
                        if (n_ == 7 || n_ == 8) {                                         \
                            asm volatile("" ::"v"(af[3][n_ - 7]));                        \
                            af[3][n_ - 7] = *reinterpret_cast<const f16x8*>(pc_ + (n_ - 7) * AP_S * 4 * KG + 3 * 128); \
                        }                                                                 \
                        if (!(LAST) && n_ >= 19 && (n_ - 19) % 12 < 2 && (n_ - 19) / 12 < AP_RT - 1) { \
                            asm volatile("" ::"v"(af[(n_ - 19) / 12][(n_ - 19) % 12]));   \
                            af[(n_ - 19) / 12][(n_ - 19) % 12] =                          \
                                *reinterpret_cast<const f16x8*>(pn_ + ((n_ - 19) % 12) * AP_S * 4 * KG + ((n_ - 19) / 12) * 128); \
                        }                                                                 \
                        __builtin_amdgcn_sched_barrier(0);                                \
                    }                                                                     \
                    _Pragma("unroll") for (int j_ = 0; j_ < AP_CT; ++j_) asm volatile("" ::"v"(WC[j_][0]), "v"(WC[j_][1])); \
                    _Pragma("unroll") for (int i_ = 0; i_ < AP_RT; ++i_) asm volatile("" ::"v"(af[i_][0]), "v"(af[i_][1])); \
                }
#pragma unroll 1
                for (int s = 0; s < AP_S - 2; s += 2) {
                    AP2_STEP(s, wa, wb, 0, 0)
                    AP2_STEP(s + 1, wb, wa, 0, 1)
                }
                AP2_STEP(AP_S - 2, wa, wb, 0, 0)
                AP2_STEP(AP_S - 1, wb, wa, 1, 1)
                if (w4 == 0) { AP_T(1) }
                AP_MMA_DRAIN();
                {   // logits: tanh, times w3, summed over this wave's 64 columns
                    // sum_c w3_c tanh(z_c) = -2 ( sum_c w3_c / (1 + 2^(C2 z_c)) - sum_c w3_c / 2 ): the lane accumulates w3_c / (1 + 2^..) starting from
                    // -1/2 of its four w3 (one FMA per value less than forming tanh first); the factor -2 is applied once per row after the reduction
                    constexpr float C2 = 2.8853900817779268f;
                    float rs[AP_RT][4], bzv[AP_CT], w3v[AP_CT];
#pragma unroll
                    for (int j = 0; j < AP_CT; ++j) {
                        const int col = 64 * w4 + 16 * j + c16;
                        bzv[j] = bw[col];
                        w3v[j] = bw[AP_D + col];
                    }
                    const float r0 = -0.5f * ((w3v[0] + w3v[1]) + (w3v[2] + w3v[3]));
#pragma unroll
                    for (int i = 0; i < AP_RT; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) rs[i][r] = r0;
#pragma unroll
                    for (int j = 0; j < AP_CT; ++j) {
#pragma unroll
                        for (int i = 0; i < AP_RT; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float z = fmaf(fmaf(acx[j][i][r], 1.0f / 2048.0f, acc[j][i][r]), C2, bzv[j]);
                                rs[i][r] = fmaf(w3v[j], __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z)), rs[i][r]);
                            }
                    }
#pragma unroll
                    for (int i = 0; i < AP_RT; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = rs[i][r];
                            v += dpp_mov<0xB1>(v);
                            v += dpp_mov<0x4E>(v);
                            v += dpp_mov<0x141>(v);
                            v += dpp_mov<0x140>(v);
                            if (c16 == 0) rp[w4 * 64 + 16 * i + 4 * g + r] = -2.0f * v;
                        }
                }
            }
            if (w4 == 0) { AP_T(2) }
            __syncthreads();
            if (w4 == 0) { AP_T(3) }
        }
    } else {
        // ================================================== IO waves ==================================================
        // IO wave w4 owns the COLUMN block [64 w4, 64 w4 + 64) of every tile: lane = (row subgroup lane >> 4, 4 columns 4 (lane & 15) ..),
        // rows 4 q + subgroup.  It stages that block of tile it into planes[it&1] and later reads only that block back for the weighted
        // sum, so the IO waves never touch each other's data and need no barrier among themselves.  Iteration it:
        //   (1) finish tile it-2: softmax from rowpart[it&1] (every IO wave computes all 64 rows, lane = row), weighted sum from that
        //       tile's own term planes, still in planes[it&1] (h = h1 + 2^-11 h2' to 2^-22: no second read of the encoder output)
        //   (2) split the rows of tile it -- loaded during the previous iteration -- into planes[it&1]
        //   (3) issue the loads of tile it+1: in flight across the barrier (64 registers held): no memory latency exposed here
        // Optional static issue priority above the MMA wave of the same SIMD (tunable attn_io_prio; measured in tools/attn_micro.py: the two
        // waves of a SIMD share its issue bandwidth, so work moved between them does not net -- the priority only decides WHO waits)
        if (p.io_prio) __builtin_amdgcn_s_setprio(2);
        float* pw = probw + w4 * 64;
        const int sub = lane >> 4, per = T / 4;
        const int col = 64 * w4 + 4 * c16;
        const int poff = ((col >> 5) * 4 + ((col >> 3) & 3)) * KG + (col & 7);
        // row r of this lane's blocks sits at position r ^ x, x = 4 (S & 1) + 2 (g >> 1) = 4 xq + 2 xs: r = 4 q + sub -> 4 (q ^ xq) + (sub ^ 2 xs):
        // one base for even and one for odd q, the rest of q stays a compile-time offset
        const int xq = c16 >> 3, sx = sub ^ (2 * ((c16 >> 2) & 1));
        const int roff_e = (4 * xq + sx) * 8, roff_o = (4 * (1 ^ xq) + sx) * 8;
        float4 ld[IN16 ? 1 : 16];
        uint2 ld16[IN16 ? 16 : 1];
        // The length of row `lane`'s sequence travels with the tile's rows: requested IN FRONT of them (vmcnt retires in order), first read where
        // the rows are consumed anyway, then handed down a two-deep register ring to the iteration that finishes the tile.  Loaded inside the
        // finish phase instead, its wait covered the 16 row loads in flight for tile k+1 as well: every tile paid a full memory round trip there
        // (11.5 K of a 15 K-cycle iteration, tools/attn_micro.py).
        const float b3v = p.b3[0];
        int len_ld = T, len_1 = T, len_2 = T;
        auto load_rows = [&](int64_t k) {
            const int64_t row0 = (blockIdx.x + k * G) * AP_ROWS;
            {
                const int64_t seq = (row0 + lane) >> p.logT;
                len_ld = T;
                if (p.lens) len_ld = (int)p.lens[seq < p.M ? seq : p.M - 1];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int64_t r = row0 + 4 * q + sub;
                r = r < nrows ? r : nrows - 1;             // rows past the end belong to sequences that are never written
                if (IN16) ld16[q] = *reinterpret_cast<const uint2*>(reinterpret_cast<const _Float16*>(p.h) + r * AP_D + col);
                else ld[q] = *reinterpret_cast<const float4*>(p.h + r * AP_D + col);
            }
        };
        if (nk > 0) load_rows(0);
        for (int64_t it = 0; it < nk + 2; ++it) {
            unsigned short* Pb = asm2_ + (it & 1) * PH + poff;
            unsigned short* Pbe = Pb + roff_e;
            unsigned short* Pbo = Pb + roff_o;
            if (w4 == 0) { AP_T(4) }
            if (it >= 2) {
                const int64_t rowS = (blockIdx.x + (it - 2) * G) * AP_ROWS;
                const float* rp = rowpart + (it & 1) * 4 * 64;
                {
                    const int64_t r = rowS + lane;
                    const int64_t seq = r >> p.logT;                   // T is a power of two: no 64-bit division per lane
                    const int t = (int)(r - seq * T);
                    int len = len_2;
                    len = len < 0 ? 0 : (len > T ? T : len);
                    const bool ok = seq < p.M && t < len;
                    const float lg = (rp[lane] + rp[64 + lane]) + (rp[128 + lane] + rp[192 + lane]) + b3v;
                    const float mx = ap_group_max(ok ? lg : -INFINITY, T);
                    const float e = ok ? __expf(lg - mx) : 0.f;
                    const float den = ap_group_sum(e, T);
                    pw[lane] = e / den;                    // len == 0: 0/0 = NaN, like softmax over an all -inf row
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if (w4 == 0) { AP_T(8) }
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                // in two halves of 8 rows: ALL LDS reads of a half are issued before its first use (40 registers).  Written row by row the
                // compiler waited for every row's three reads before converting it -- 16 serialised LDS round trips per tile, each several
                // hundred cycles while the MMA waves stream their fragments: 8-10 K cycles for ~300 VALU instructions (tools/attn_micro.py)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    uint2 u1[8], u2[ROW1 ? 1 : 8];
                    float prr[8];
#pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8) {
                        const int row = 4 * (8 * hb + q8) + sub;
                        const unsigned short* src = ((q8 & 1) ? Pbo : Pbe) + ((8 * hb + q8) >> 1) * 64;
                        u1[q8] = *reinterpret_cast<const uint2*>(src);
                        if (!ROW1) u2[q8] = *reinterpret_cast<const uint2*>(src + AP_S * 4 * KG);
                        prr[q8] = pw[row];
                    }
#pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8) {
                        const int q = 8 * hb + q8;
                        const fp16x2_t a01 = __builtin_bit_cast(fp16x2_t, u1[q8].x), a23 = __builtin_bit_cast(fp16x2_t, u1[q8].y);
                        const float pr = prr[q8];
#if AP_FMAMIX
                        // v_fma_mix_f32 reads the fp16 halves in place: 1 (one term) or 2 (two terms) instructions per element instead of a
                        // convert per term plus the FMAs -- the IO wave shares its SIMD's issue slots with an MMA wave
                        (void)a01; (void)a23;
                        ap_fma_mix(a, u1[q8], pr);
                        if (!ROW1) ap_fma_mix(a, u2[q8], pr * (1.0f / 2048.0f));
#else
                        if (ROW1) {
                            a.x = fmaf(pr, (float)a01[0], a.x);
                            a.y = fmaf(pr, (float)a01[1], a.y);
                            a.z = fmaf(pr, (float)a23[0], a.z);
                            a.w = fmaf(pr, (float)a23[1], a.w);
                        } else {
                            const fp16x2_t b01 = __builtin_bit_cast(fp16x2_t, u2[q8].x), b23 = __builtin_bit_cast(fp16x2_t, u2[q8].y);
                            a.x = fmaf(pr, fmaf((float)b01[0], 1.0f / 2048.0f, (float)a01[0]), a.x);
                            a.y = fmaf(pr, fmaf((float)b01[1], 1.0f / 2048.0f, (float)a01[1]), a.y);
                            a.z = fmaf(pr, fmaf((float)b23[0], 1.0f / 2048.0f, (float)a23[0]), a.z);
                            a.w = fmaf(pr, fmaf((float)b23[1], 1.0f / 2048.0f, (float)a23[1]), a.w);
                        }
#endif
                        if (((q + 1) & (per - 1)) == 0) {              // sequence complete: fold the four row subgroups, subgroup 0 stores
                            a.x += __shfl_xor(a.x, 16); a.y += __shfl_xor(a.y, 16); a.z += __shfl_xor(a.z, 16); a.w += __shfl_xor(a.w, 16);
                            a.x += __shfl_xor(a.x, 32); a.y += __shfl_xor(a.y, 32); a.z += __shfl_xor(a.z, 32); a.w += __shfl_xor(a.w, 32);
                            const int64_t seq = (rowS >> p.logT) + (q >> (p.logT - 2));
                            if (sub == 0 && seq < p.M) *reinterpret_cast<float4*>(p.pooled + seq * AP_D + col) = a;
                            a = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                }
            }
            if (w4 == 0) { AP_T(5) }
            if (it < nk) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    unsigned short* d = ((q & 1) ? Pbo : Pbe) + (q >> 1) * 64;
                    if (IN16) {
                        *reinterpret_cast<uint2*>(d) = ld16[q];
                        continue;
                    }
                    const float4 v = ld[q];
                    if (IN == 2) {                             // the two terms as the recurrence formed them
                        *reinterpret_cast<uint2*>(d) = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
                        *reinterpret_cast<uint2*>(d + AP_S * 4 * KG) = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
                        continue;
                    }
                    const fp16x2_t a01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), a23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
                    *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, a23));
                    if (!ONE) {
                        const fp16x2_t b01 = __builtin_amdgcn_cvt_pkrtz((v.x - (float)a01[0]) * 2048.0f, (v.y - (float)a01[1]) * 2048.0f);
                        const fp16x2_t b23 = __builtin_amdgcn_cvt_pkrtz((v.z - (float)a23[0]) * 2048.0f, (v.w - (float)a23[1]) * 2048.0f);
                        *reinterpret_cast<uint2*>(d + AP_S * 4 * KG) = make_uint2(__builtin_bit_cast(unsigned, b01), __builtin_bit_cast(unsigned, b23));
                    }
                }
                if (w4 == 0) { AP_T(6) }
                len_2 = len_1;                               // tile it-1 is finished next
                len_1 = len_ld;                              // tile it (its rows were consumed just above: the value has arrived)
                if (it + 1 < nk) load_rows(it + 1);
            } else {
                len_2 = len_1;
            }
            if (w4 == 0) { AP_T(7) }
            __syncthreads();
        }
    }
}

#ifdef AP_TIMING
}  // namespace nir
extern "C" int nir_debug_attn_timing(long long* out16) { return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(nir::ap_dbg), sizeof(long long) * 16); }
namespace nir {
#endif

bool attn_pool_fused_usable(int D, int T) { return D == AP_D && (T == 4 || T == 8 || T == 16 || T == 32 || T == 64); }

static int ap_cu_count() {
    static const int ncu = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }();
    return ncu;
}
// true when launch_attn_pool_fused(M, T) runs the role-specialised pipeline (several tiles per CU, or forced by the tunable)
bool attn_pool_pipe_selected(int64_t M, int T) {
    const int64_t tiles = (M * T + AP_ROWS - 1) / AP_ROWS;
    const int pipe = tun(g_tun.attn_unfused_pipe);     // 0: by size, 1: never, 2: always (tests)
    return (tiles >= 2 * (int64_t)ap_cu_count() && pipe == 0) || pipe == 2;
}

int launch_attn_pool_fused(const float* h, const void* wfrag, const float* b0, const float* w3, const float* b3, const int64_t* lens, int64_t M,
                           int T, float* pooled, int one_term, hipStream_t st, int in_f16) {
    NIR_REQUIRE(h && wfrag && b0 && w3 && b3 && pooled && attn_pool_fused_usable(AP_D, T), "attn_pool_fused: bad args (T=%d)", T);
    if (M == 0) return 0;
    AttnPoolArgs a;
    a.h = h; a.wf = (const _Float16*)wfrag; a.b0 = b0; a.w3 = w3; a.b3 = b3; a.lens = lens; a.pooled = pooled; a.M = M; a.T = T; a.logT = __builtin_ctz((unsigned)T);
    a.io_prio = tun(g_tun.attn_io_prio);
    static std::once_flag once;
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)attn_pool_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AP_LDS); });
    const int64_t tiles = (M * T + AP_ROWS - 1) / AP_ROWS;
    // (profile label = the kernel that runs: the role-specialised pipeline by template arguments, else the single-role kernel)
    const bool pipe_sel = attn_pool_pipe_selected(M, T);
    const char* pname = !pipe_sel ? "attn_pool_fused_kernel" : in_f16 == 1 ? "attn_pool_pipe_kernel<true,1>" : in_f16 == 3 ? "attn_pool_pipe_kernel<false,1>" :
                        in_f16 == 2 ? "attn_pool_pipe_kernel<false,2>" :
                        one_term ? "attn_pool_pipe_kernel<true,0>" : "attn_pool_pipe_kernel<false,0>";
    ProfScope ps(prof_shape_name(pname, M * T, AP_D, AP_D), st);
    static std::once_flag once2;
    std::call_once(once2, [] {
        (void)hipFuncSetAttribute((const void*)attn_pool_pipe_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AP2_LDS);
        (void)hipFuncSetAttribute((const void*)attn_pool_pipe_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AP2_LDS);
        (void)hipFuncSetAttribute((const void*)attn_pool_pipe_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AP2_LDS);
        (void)hipFuncSetAttribute((const void*)attn_pool_pipe_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AP2_LDS);
        (void)hipFuncSetAttribute((const void*)attn_pool_pipe_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AP2_LDS);
    });
    const int ncu = ap_cu_count();
    if (pipe_sel) {                                                        // several tiles per CU: the role-specialised pipeline
        const dim3 grid((unsigned)std::min<int64_t>(tiles, ncu));
        NIR_REQUIRE(in_f16 != 2 || !one_term, "attn_pool_fused: term-pair rows come from the fp32-accurate encoder (two terms)");
        if (in_f16 == 1) hipLaunchKernelGGL((attn_pool_pipe_kernel<true, 1>), grid, dim3(512), AP2_LDS, st, a, tiles);
        else if (in_f16 == 3) hipLaunchKernelGGL((attn_pool_pipe_kernel<false, 1>), grid, dim3(512), AP2_LDS, st, a, tiles);   // fp16 rows, two-term W0
        else if (in_f16 == 2) hipLaunchKernelGGL((attn_pool_pipe_kernel<false, 2>), grid, dim3(512), AP2_LDS, st, a, tiles);
        else if (one_term) hipLaunchKernelGGL((attn_pool_pipe_kernel<true, 0>), grid, dim3(512), AP2_LDS, st, a, tiles);
        else hipLaunchKernelGGL((attn_pool_pipe_kernel<false, 0>), grid, dim3(512), AP2_LDS, st, a, tiles);
    } else {
        NIR_REQUIRE(!in_f16, "attn_pool_fused: fp16 input is only taken by the pipelined kernel (attn_pool_pipe_selected)");
        hipLaunchKernelGGL(attn_pool_fused_kernel, dim3((unsigned)tiles), dim3(256), AP_LDS, st, a);
    }
    NIR_CHECK_LAUNCH("attn_pool_fused_kernel");
    return 0;
}

}  // namespace nir
