// DUET distributed model, document branch, fused per tile of conv positions (neuroir/rankers/duet.py:174-201):
//     emb gather -> conv_d1 (k = 3) + tanh -> max_pool1d(P, stride 1) -> conv_d2 (1x1) + tanh -> Hadamard with the query vector
//     -> Linear over positions (fc2)
// The GEMM-per-layer chain writes conv_d1's [M, DL-2, NF] output, the pooled tensor and conv_d2's output to HBM and reads each back
// (C4: 9.3 GB of counter traffic for 1.1 GB of embedding rows).  Here one workgroup owns 16*RT consecutive conv positions and keeps
// everything on chip:
//   GEMM 1  D1[rows, NFP] = A[rows, 3E] W1^T  A = three shifted views of the gathered embedding rows (fp32 from HBM/L2, split into
//                                             two fp16 terms on the way into LDS); W1 = pre-split fp16 term planes stored by the
//                                             host in MFMA-fragment order, streamed L2 -> VGPR (every wave owns 80 filter columns,
//                                             so a W fragment has exactly one consumer and never needs LDS)
//   pool    rows of an accumulator tile live in the wave that owns the column: the P-row window is one ds_bpermute per value
//   GEMM 2  D2[rows, NFP] = P[rows, NFP] W2^T P = pooled tile as fp16 term planes in LDS (fragment order), W2 streamed like W1
//   fc2     partial[tile][slot][f] = sum_rows fc2_w[t] * D2[row][f]   (rows whose pooling window leaves the document get weight 0)
// A second tiny kernel folds the tiles of a document: m1[pair][f] = tanh(fc2_b + qv[b][f] * sum partial).
// Arithmetic: two-term fp16 split (x = h1 + 2^-11 h2', three v_mfma_f32_16x16x32_f16 per k-block, two accumulator sets) as in
// gemm3_kernel<., true> -- needs |table|, |weights| < 2^15 (host-checked `bounded`); activations are tanh outputs.
// The GEMM phases are bound by the W stream out of L2 (every CU streams all of W1 / W2 per tile: 40 KB per k-step; measured with 64-row
// tiles: TCC busy 95 %, 13 TB/s of L2 reads), so the tile is as tall as registers and LDS allow, one workgroup (4 waves, one per SIMD)
// per CU.  Two forms of GEMM 1:
//   * plane mode (the pack carries the embedding table as fp16 term planes, nir_duet_weights.ftable): the tile's token rows are ONE
//     contiguous range of the flattened id array; they are brought into LDS once (not once per tap), by LDS-direct loads that need no
//     registers and no split arithmetic, two column chunks ahead of their use, and handed over through an LDS flag instead of a
//     barrier.  k-steps run row tile by row tile (A fragments double-buffered per row tile), which leaves room for 96-row tiles
//     (documents of >= 98 - pool positions; 64 rows otherwise): 160 accumulator AGPRs + 80 accumulator VGPRs, 133 KB of LDS.
//   * fp32 table (no planes): 64-row tiles, the rows of a k-step are gathered per tap, split on the way into LDS, one barrier per step.
// Tiles are cut from the FLATTENED (document, position) axis when documents are long enough (a tile then touches at most two
// documents): 92 of 96 rows carry useful pooled rows.
#include <algorithm>
#include <mutex>
#include "common.hpp"

namespace nir {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int DF_NFP = 320;                 // filter columns (4 waves x 5 tiles x 16)
constexpr int DF_CT = 5;                    // column tiles per wave
constexpr int DF_S2 = DF_NFP / 32;          // k-steps of GEMM 2
constexpr int DF_RT = 4;                    // row tiles of the launched instantiation
constexpr int DF_ROWS = 16 * DF_RT;         // conv positions per workgroup

struct DuetDocArgs {
    const int64_t* d_ids;       // [M, DL]
    const float* table;         // [V, E]
    const _Float16* wf1;        // [S1][20 col tiles][2 terms][64 lanes][8]
    const _Float16* wf2;        // [DF_S2][20][2][64][8]
    const float *b1, *b2;       // [NF]
    const float* fc2w;          // [PL]
    float* partial;             // [tiles][2 slots][DF_NFP]
    int64_t M;
    int E, DL, S1, NF, Tc, PL, P;
    int flat;                   // 1: tiles cut from the flattened (doc, position) axis with stride TS; 0: ntile tiles per document
    int TS, ntile, TPv;
    // plane mode (PL): the embedding table as pre-split fp16 term planes [V][2 terms][EPT] (EPT = E rounded up to a multiple of 32, zero
    // padded) and conv_d1 in chunk-major k order: k-step 3c + u = elements 32c .. 32c+31 of tap u
    const _Float16* ftab;
    const _Float16* wf1c;       // [3 C][20][2][64][8]
    int EPT, C, ids_off;        // C = EPT / 32 column chunks; byte offset of the tile's token ids in LDS
};

__device__ __forceinline__ void df_split_store(unsigned short* base, int kgs, int row, int kg, int e0, const float4& v) {
    const fp16x2_t a01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), a23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    const float r0 = (v.x - (float)a01[0]) * 2048.0f, r1 = (v.y - (float)a01[1]) * 2048.0f;
    const float r2 = (v.z - (float)a23[0]) * 2048.0f, r3 = (v.w - (float)a23[1]) * 2048.0f;
    const fp16x2_t b01 = __builtin_amdgcn_cvt_pkrtz(r0, r1), b23 = __builtin_amdgcn_cvt_pkrtz(r2, r3);
    unsigned short* d = base + kg * kgs + row * 8 + e0;
    *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, a23));
    *reinterpret_cast<uint2*>(d + 4 * kgs) = make_uint2(__builtin_bit_cast(unsigned, b01), __builtin_bit_cast(unsigned, b23));
}

// tanh(x) for z = 2 log2(e) x already formed:  1 - 2 / (1 + 2^z)   (v_exp_f32, v_rcp_f32; saturates correctly at +-inf)
constexpr float DF_2LOG2E = 2.8853900817779268f;
__device__ __forceinline__ float df_tanh_z(float z) {
#ifdef DF_X_NOTANH
    return z;
#endif
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z)), 1.0f);
}

__device__ __forceinline__ float df_bperm(float v, int byte_idx) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_idx, __builtin_bit_cast(int, v)));
}

// In-place accumulate in AGPRs.  Written as inline assembly: with the builtin, hipcc assigns the result of each accumulator chain to a
// different register tuple than its loop-carried input and rotates most tuples through VGPRs on every k-step (112 v_accvgpr_* moves per
// 60 MFMAs).  The operands come straight from ds_read / global_load (s_waitcnt is still compiler-inserted); the accumulators are first
// read by VALU code after DF_MMA_DRAIN.
#define DF_MMA(ACC, A, W) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(W))
#define DF_MMA_V(ACC, A, W) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(W))
// first product of an accumulator chain: C = 0 (no zero fill of 240 registers between the two GEMMs)
#define DF_MMA0(ACC, A, W) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(ACC) : "v"(A), "v"(W))
#define DF_MMA0_V(ACC, A, W) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(ACC) : "v"(A), "v"(W))
#ifdef DF_TIMING
__device__ long long df_dbg[16];
#define DF_T(I) if (blockIdx.x == 3000 && threadIdx.x == 0) df_dbg[I] = __builtin_readcyclecounter();
#else
#define DF_T(I)
#endif
#define DF_MMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")
// Accumulator element -> VGPR at the point of use.  Left to the compiler, every accumulator is copied out of its AGPR right behind the
// last MFMA (240 v_accvgpr_read in a row for the 96-row tile: the epilogue then spills, and so do the GEMM loops around it).
__device__ __forceinline__ float df_acc(const f32x4& a, int r, bool from_agpr) {      // from_agpr folds to a constant after unrolling
    if (!from_agpr) return a[r];              // VGPR-resident accumulators; the 64-row kernel without table planes keeps the allocation it was validated with
    float x;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a[r]));
    return x;
}

// MFMA number n of a k-step (n is a compile-time constant after unrolling): column tile n / (3 RT), term pair (n / RT) % 3, row tile n % RT
template <int RT>
__device__ __forceinline__ void df_mma_n(int n, f32x4 (&acc)[DF_CT][RT], f32x4 (&acx)[DF_CT][RT], const f16x8 (&af)[RT][2],
                                         const f16x8 (&w)[DF_CT][2]) {
    const int j = n / (3 * RT), ph = (n / RT) % 3, i = n % RT;
    if (ph == 0) DF_MMA(acx[j][i], af[i][1], w[j][0]);
    else if (ph == 1) DF_MMA(acx[j][i], af[i][0], w[j][1]);
    else DF_MMA(acc[j][i], af[i][0], w[j][0]);
}

// one half (two elements) of a staged float4: split into the two fp16 terms and store 4 bytes into each term plane
__device__ __forceinline__ void df_split_store_half(unsigned short* base, int kgs, int row, int kg, int e0, float x, float y) {
    const fp16x2_t a = __builtin_amdgcn_cvt_pkrtz(x, y);
    const fp16x2_t b = __builtin_amdgcn_cvt_pkrtz((x - (float)a[0]) * 2048.0f, (y - (float)a[1]) * 2048.0f);
    unsigned short* d = base + kg * kgs + row * 8 + e0;
    *reinterpret_cast<unsigned*>(d) = __builtin_bit_cast(unsigned, a);
    *reinterpret_cast<unsigned*>(d + 4 * kgs) = __builtin_bit_cast(unsigned, b);
}

template <int RT>
struct DfLayout {
    static constexpr int ROWS = 16 * RT;
    static constexpr int KG = ROWS * 8 + 32;            // halves per k-group block [row][8] (+64 B: the 4 k-groups start in different banks)
    static constexpr int A_HALVES = 2 * 2 * 4 * KG;     // A stage: [2 buffers][2 terms][4 k-groups][KG]
    static constexpr int P_HALVES = 2 * DF_S2 * 4 * KG; // P planes: [2 terms][DF_S2][4 k-groups][KG]
    static constexpr size_t LDS = (size_t)(A_HALVES + P_HALVES) * 2;
    static constexpr int LPT = ROWS * 8 / 256;          // A-stage float4 loads per thread and k-step
};

// Plane mode: the token rows a tile needs are ONE contiguous range of the flattened id array -- conv row (d, t), tap u reads token
// (doc0 + d) DL + t + u = F0 + pr + 2 d + u -- so the tile's ROWS + 6 token rows are brought in once (not once per tap), as fp16 term
// planes that need no arithmetic, by LDS-direct loads that need no registers: global_load_lds_dwordx4 writes a wave's 64 x 16 bytes to
// consecutive LDS addresses, so the tile is laid out [column chunk c][term][k-group][token row][8 halves] with a chunk padded to NI
// wave-loads.  Chunk c + 1 is requested during the first two k-steps of chunk c and taken over (s_waitcnt vmcnt + barrier) at the top
// of the third: the k-loop has one barrier per THREE k-steps and no VALU work beyond address arithmetic.
template <int RT>
struct DfLayoutP {
    static constexpr int ROWS = 16 * RT;
    static constexpr int RP = ROWS + 6;                 // token rows: ROWS + 2 taps + 2 per document boundary (at most two)
    static constexpr int NI = (8 * RP + 63) / 64;       // wave-level loads per chunk (2 terms x 4 k-groups x RP pieces of 16 bytes)
    static constexpr int CHB = NI * 1024;               // bytes per chunk
    static constexpr int NSLOT = (NI + 3) / 4;          // loads per wave and chunk
};

// POOL: pooling window known at compile time (5 = the reference's pool_size), or 0 = read it from the arguments (1..5)
template <int RT, int POOL, bool PL>
__global__ __launch_bounds__(256, 1) void duet_doc_kernel(DuetDocArgs p) {
    using L = DfLayout<RT>;
    constexpr int KG = L::KG, LPT = L::LPT;
    extern __shared__ __attribute__((aligned(16))) unsigned short dsm[];
    unsigned short* As = dsm;
    unsigned short* Pp = PL ? dsm : dsm + L::A_HALVES;          // plane mode: the P planes take the token tile's place after GEMM 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int E = p.E, S1 = p.S1, K1 = 3 * p.E, Tc = p.Tc;
    // first conv row of the tile: document doc0, position t00 (one 64-bit division per workgroup; rows use 32-bit offsets from it)
    int64_t doc0;
    int t00;
    if (p.flat) {
        const int64_t R0 = (int64_t)blockIdx.x * p.TS;
        doc0 = R0 / Tc;
        t00 = (int)(R0 - doc0 * Tc);
    } else {
        doc0 = blockIdx.x / p.ntile;
        t00 = (int)(blockIdx.x % p.ntile) * p.TPv;
    }
    // row pr of the tile -> (document offset d in 0..2, position t); flattened tiles run on into the following documents
    auto row_pos = [&](int pr, int& d, int& t) {
        t = t00 + pr;
        d = 0;
        if (p.flat) {
            if (t >= Tc) { t -= Tc; d = 1; }
            if (t >= Tc) { t -= Tc; d = 2; }
            if (doc0 + d >= p.M) { d = (int)(p.M - 1 - doc0); t = Tc - 1; }     // past the last document: clamp (weight 0 later)
        } else {
            t = t < Tc - 1 ? t : Tc - 1;                                            // per-document tiles stay inside their document
        }
    };

    // ---- W operands: fragment-ordered planes, 1 KB contiguous per wave-level load; one register set, a column tile's pair is
    // re-loaded for the next k-step as soon as its MFMAs are issued (the other four tiles' MFMAs cover the L2 latency)
    const _Float16* wp1 = (PL ? p.wf1c : p.wf1) + ((int64_t)(DF_CT * wave) * 2 * 64 + lane) * 8;
    const _Float16* wp2 = p.wf2 + ((int64_t)(DF_CT * wave) * 2 * 64 + lane) * 8;
    constexpr int WSTEP = 20 * 2 * 64 * 8;
    f16x8 w[DF_CT][2];
    f32x4 acc[DF_CT][RT], acx[DF_CT][RT];
#define DF_ZERO_ACC()                                                                     \
    _Pragma("unroll") for (int j_ = 0; j_ < DF_CT; ++j_)                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < RT; ++i_) {                               \
            acc[j_][i_] = f32x4{0.f, 0.f, 0.f, 0.f};                                      \
            acx[j_][i_] = f32x4{0.f, 0.f, 0.f, 0.f};                                      \
        }
    if constexpr (!PL) { DF_ZERO_ACC() }
    const int foff = g * KG + c16 * 8;          // fragment address of the A-side operand: [term][k-group = lane >> 4][row][8]

    // ================= GEMM 1: conv_d1 =================
    // One wave per SIMD: whatever is not an MFMA has to be issued in the 12 idle issue cycles behind each MFMA, so every k-step is
    // written as its 15 RT MFMAs with the other work of the step pinned between them (sched_barrier after every MFMA):
    //   W fragments of step s+1  -> the other register set, one load every 6 MFMAs (a full step of slack: with a single set re-loaded
    //                               tile by tile, hipcc's loop-header s_waitcnt merge waited for loads issued ~200 cycles earlier)
    //   A rows of step s+1       -> loaded during step s-1 (register set RAC), split + stored to LDS in the first third of step s
    //   A rows of step s+2       -> global loads into the other set (RAN)
    //   barrier                  -> after MFMA 8 LPT + 6; A fragments of step s+1 are read from LDS in the second half (set AFN)
    // WAR hazard of the inline-assembly MFMAs: a VALU write to a VGPR that an MFMA issued fewer than ~8 slots earlier reads as A/B
    // operand corrupts that operand (LLVM pads this for its own MFMAs -- SMFMA16x16ReadVgprVALUWarWaitStates -- but cannot see inside
    // the asm, and happily re-uses a fragment register for split arithmetic right after its last MFMA: deterministic 1e-3 errors).
    // DF_KEEP / DF_KEEP_HEAD are empty asm uses that keep the fragment registers of a step live until its end, and those read by the
    // last MFMAs of a step until MFMA 8 of the next one, so nothing can be allocated on top of them in between.
    DF_T(0)
    f16x8 wb[DF_CT][2], afa[RT][2], afb[RT][2];
#define DF_LOAD_W(W, PTR)                                                                 \
    _Pragma("unroll") for (int j_ = 0; j_ < DF_CT; ++j_) {                                \
        W[j_][0] = *reinterpret_cast<const f16x8*>((PTR) + (j_ * 2) * 512);               \
        W[j_][1] = *reinterpret_cast<const f16x8*>((PTR) + (j_ * 2 + 1) * 512);           \
    }
#define DF_KEEP_HEAD(WN, AFN)                                                             \
    _Pragma("unroll") for (int j_ = DF_CT - 2; j_ < DF_CT; ++j_) asm volatile("" ::"v"(WN[j_][0]), "v"(WN[j_][1])); \
    _Pragma("unroll") for (int i_ = 0; i_ < RT; ++i_) asm volatile("" ::"v"(AFN[i_][0]), "v"(AFN[i_][1]));
#define DF_KEEP(WC, AFC)                                                                  \
    _Pragma("unroll") for (int j_ = 0; j_ < DF_CT; ++j_) asm volatile("" ::"v"(WC[j_][0]), "v"(WC[j_][1]));   \
    _Pragma("unroll") for (int i_ = 0; i_ < RT; ++i_) asm volatile("" ::"v"(AFC[i_][0]), "v"(AFC[i_][1]));
    // Plane mode runs its k-steps ROW TILE by row tile (15 MFMAs: 3 term products x 5 column tiles), not column tile by column tile: a row
    // tile's A fragments are then live for 15 MFMAs instead of the whole step and are double-buffered per row tile (16 registers instead
    // of 32 RT), which is what lets the 96-row tile (240 accumulators) fit.  Per step: the W fragments of the next step into the other
    // set, in the order the next step needs them and in its first third (all ten are read by the next step's first ten MFMAs), the next
    // row tile's fragments at MFMAs 5 and 7 of a row tile.  Every redefinition is preceded by an empty asm use of the old value, so the register is never free between its last MFMA
    // read and its reload (the VALU-after-MFMA WAR hazard described below).
    // With 240 accumulator AGPRs out of 256 the register allocator starts rotating accumulator tuples through VGPRs around the asm MFMAs
    // (and reads them there before the MFMA has written them back): the last row tile of the 96-row kernel accumulates in VGPRs instead.
    constexpr int VACC = RT > 4 ? RT - 4 : 0;
#ifndef DF_WSTRIDE
#define DF_WSTRIDE 2
#endif
#ifndef DF_AFM
#define DF_AFM 1
#endif
#ifndef DF_G1FIRST        // 1: the first k-step of GEMM 1 starts the accumulators with C = 0 (GEMM 2 does, its first step is peeled); here it would be a
#define DF_G1FIRST 0      // uniform branch per MFMA of the step body shared by all chunks -- measured: GEMM 1 50 K -> 61 K cycles per tile (every branch
#endif                    // ends a scheduling region).  0: 240 zero writes per tile, rematerialised by hipcc right in front of the first MFMA (~1 K cycles)
#ifndef DF_ROT            // 1: every workgroup starts its k walk at its own chunk (measured: no effect, see below); 0: canonical k order
#define DF_ROT 0
#endif
#ifndef DF_X_NOTILE       // timing ablations (tools/duet_micro.py): results are wrong with either set
#define DF_X_NOTILE 0
#endif
#ifndef DF_X_NOBAR
#define DF_X_NOBAR 0
#endif
    f16x8 af[2][2];
#define DF_STEPR(WC, WN, WNP, CURB, NXTB, TOFF, ROW, FIRST, EXTRA)                        \
    {                                                                                     \
        constexpr int WS_ = DF_WSTRIDE;          /* every fragment is needed in the first ten MFMAs of the next step: request early */ \
        _Pragma("clang loop unroll(full)") for (int n_ = 0; n_ < 15 * RT; ++n_) {         \
            const int i_ = n_ / 15, m_ = n_ % 15, j_ = m_ % 5, b_ = i_ & 1;               \
            if (i_ >= RT - VACC) {                                                        \
                if (m_ < 5) { if (FIRST) DF_MMA0_V(acx[j_][i_], af[b_][1], WC[j_][0]); else DF_MMA_V(acx[j_][i_], af[b_][1], WC[j_][0]); } \
                else if (m_ < 10) DF_MMA_V(acx[j_][i_], af[b_][0], WC[j_][1]);            \
                else { if (FIRST) DF_MMA0_V(acc[j_][i_], af[b_][0], WC[j_][0]); else DF_MMA_V(acc[j_][i_], af[b_][0], WC[j_][0]); } \
            } else {                                                                      \
                if (m_ < 5) { if (FIRST) DF_MMA0(acx[j_][i_], af[b_][1], WC[j_][0]); else DF_MMA(acx[j_][i_], af[b_][1], WC[j_][0]); } \
                else if (m_ < 10) DF_MMA(acx[j_][i_], af[b_][0], WC[j_][1]);              \
                else { if (FIRST) DF_MMA0(acc[j_][i_], af[b_][0], WC[j_][0]); else DF_MMA(acc[j_][i_], af[b_][0], WC[j_][0]); } \
            }                                                                             \
            EXTRA                                                                         \
            if (n_ >= 3 && (n_ - 3) % WS_ == 0 && (n_ - 3) / WS_ < 2 * DF_CT) {           \
                const int k_ = (n_ - 3) / WS_, t_ = k_ < DF_CT ? 0 : 1, jj_ = k_ % DF_CT; \
                asm volatile("" ::"v"(WN[jj_][t_]));                                      \
                WN[jj_][t_] = *reinterpret_cast<const f16x8*>((WNP) + (jj_ * 2 + t_) * 512); \
            }                                                                             \
            if (m_ == DF_AFM - 1 || m_ == DF_AFM) asm volatile("" ::"v"(af[b_ ^ 1][m_ == DF_AFM - 1 ? 1 : 0]));  \
            if (m_ == DF_AFM || m_ == DF_AFM + 1) {                                       \
                const int tt_ = m_ == DF_AFM ? 1 : 0;                                     \
                const unsigned char* fb_ = i_ + 1 < RT ? (CURB) : (NXTB);                 \
                const int in_ = i_ + 1 < RT ? i_ + 1 : 0;                                 \
                af[b_ ^ 1][tt_] = *reinterpret_cast<const f16x8*>(fb_ + tt_ * (TOFF) + ROW(in_)); \
            }                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                            \
        }                                                                                 \
    }
    static_assert(!PL || RT % 2 == 0, "row-tile double buffering of the A fragments needs an even RT");
    if constexpr (!PL) {
    // ---- A operand: conv row (doc, t) reads tokens t, t+1, t+2 of its document (rows past the end are clamped; they only feed
    // pooled rows that get weight 0).  Tap s of a row starts at table + id_s * E; the offsets are pre-biased by the tap's k offset so
    // that element k of the concatenated row is table[off_s + k].
    const int aq = tid & 7;
    int64_t off[LPT][3];
#pragma unroll
    for (int h = 0; h < LPT; ++h) {
        int d, t;
        row_pos((tid >> 3) + 32 * h, d, t);
        const int64_t* idp = p.d_ids + (doc0 + d) * p.DL + t;
        off[h][0] = idp[0] * (int64_t)E;
        off[h][1] = idp[1] * (int64_t)E - E;
        off[h][2] = idp[2] * (int64_t)E - 2 * E;
    }
    const float* const table = p.table;
    float4 ra[LPT], rb[LPT];
#define DF_LOAD_A(RA, S)                                                                                  \
    {                                                                                                     \
        int k_ = 32 * (S) + 4 * aq;                                                                       \
        k_ = k_ < K1 ? k_ : K1 - 4;                              /* k >= 3E: any finite value (W1 is zero there) */ \
        _Pragma("unroll") for (int h_ = 0; h_ < LPT; ++h_) {                                              \
            const int64_t o_ = k_ < E ? off[h_][0] : (k_ < 2 * E ? off[h_][1] : off[h_][2]);              \
            RA[h_] = *reinterpret_cast<const float4*>(table + o_ + k_);                                   \
        }                                                                                                 \
    }
#define DF_STORE_A(RA, BUF)                                                                               \
    {                                                                                                     \
        _Pragma("unroll") for (int h_ = 0; h_ < LPT; ++h_)                                                \
            df_split_store(As + (BUF) * (2 * 4 * KG), KG, (tid >> 3) + 32 * h_, aq >> 1, 4 * (aq & 1), RA[h_]); \
    }
#define DF_STEP1(S, AFC, AFN, WC, WN, RAC, RAN)                                           \
    {                                                                                     \
        const int s1_ = (S) + 1 < S1 ? (S) + 1 : S1 - 1;      /* past the end: re-load the last step's operands (branch-free) */ \
        const int s2_ = (S) + 2 < S1 ? (S) + 2 : S1 - 1;                                  \
        const _Float16* wn_ = wp1 + (int64_t)s1_ * WSTEP;                                 \
        unsigned short* an_ = As + (((S) + 1) & 1) * (2 * 4 * KG);                        \
        _Pragma("clang loop unroll(full)") for (int n_ = 0; n_ < 15 * RT; ++n_) {                          \
            df_mma_n<RT>(n_, acc, acx, AFC, WC);                                          \
            if (n_ == 8) { DF_KEEP_HEAD(WN, AFN) }                                        \
            if (n_ % 6 == 2 && n_ / 6 < 2 * DF_CT)                                        \
                WN[(n_ / 6) >> 1][(n_ / 6) & 1] = *reinterpret_cast<const f16x8*>(wn_ + (n_ / 6) * 512); \
            if (n_ % 4 == 1 && n_ / 4 < 2 * LPT) {                                        \
                const int h_ = (n_ / 4) >> 1;                                             \
                if ((n_ / 4) & 1) df_split_store_half(an_, KG, (tid >> 3) + 32 * h_, aq >> 1, 4 * (aq & 1) + 2, RAC[h_].z, RAC[h_].w); \
                else df_split_store_half(an_, KG, (tid >> 3) + 32 * h_, aq >> 1, 4 * (aq & 1), RAC[h_].x, RAC[h_].y); \
            }                                                                             \
            if (n_ == 8 * LPT + 2) DF_LOAD_A(RAN, s2_)                                    \
            if (n_ == 8 * LPT + 6) lds_barrier();   /* LDS-only: __syncthreads() carries s_waitcnt vmcnt(0) -- the A rows and W fragments requested a few MFMAs earlier */ \
            if (n_ >= 8 * LPT + 8 && (n_ - 8 * LPT - 8) % 3 == 0 && (n_ - 8 * LPT - 8) / 3 < 2 * RT) { \
                const int q_ = (n_ - 8 * LPT - 8) / 3;                                    \
                AFN[q_ % RT][q_ / RT] = *reinterpret_cast<const f16x8*>(an_ + (q_ / RT) * 4 * KG + foff + (q_ % RT) * 128); \
            }                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                            \
        }                                                                                 \
        DF_KEEP(WC, AFC)                                                                  \
    }
    DF_LOAD_A(ra, 0)
    DF_LOAD_W(w, wp1)
    DF_STORE_A(ra, 0)
    DF_LOAD_A(ra, (1 < S1 ? 1 : 0))
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < RT; ++i) afa[i][t] = *reinterpret_cast<const f16x8*>(As + t * 4 * KG + foff + i * 128);
    DF_T(1)
    {
        int s = 0;
#pragma unroll 1
        for (; s + 1 < S1; s += 2) {
            DF_STEP1(s, afa, afb, w, wb, ra, rb)
            DF_STEP1(s + 1, afb, afa, wb, w, rb, ra)
        }
        if (s < S1) DF_STEP1(s, afa, afb, w, wb, ra, rb)
    }


    } else {
        // ================= GEMM 1, plane mode =================
        using LP = DfLayoutP<RT>;
        constexpr int RP = LP::RP, NI = LP::NI, CHB = LP::CHB, NSLOT = LP::NSLOT;
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        unsigned char* const Ab = reinterpret_cast<unsigned char*>(dsm);
        int* const ids_s = reinterpret_cast<int*>(Ab + p.ids_off);
        const int C = p.C;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        // The 256 workgroups stream the same W fragments (the GEMM phases run at ~1 650 cycles per 96-row k-step against 1 440 of MFMA
        // issue: 40 KB of fragments per step and CU, ~13 TB/s of L2 reads chip-wide).  DF_ROT = 1 lets every workgroup walk the column
        // chunks (and GEMM 2 its k-steps) from its own starting point -- logical chunk ci is column chunk ci + rot (mod C) -- in case the
        // limit were workgroups asking the same L2 channels for the same lines at the same time: no difference (2.00 vs 2.00 ms), off.
        const int rot1 = DF_ROT ? (int)(blockIdx.x % (unsigned)C) : 0;
#define DF_PHYS1(CI_) ((CI_) + rot1 < C ? (CI_) + rot1 : (CI_) + rot1 - C)
        // Everything the prologue reads from memory that does not depend on the token ids is requested first, with branch-free (clamped)
        // addresses -- a predicated load is its own exec-masked block with s_waitcnt vmcnt(0) behind it: the tile's token ids, the fc2
        // row weights (used by the last epilogue: weight = fc2_w[t] for an own pooled row whose window stays inside its document, else
        // 0; slot 0 = the tile's first document, slot 1 = the next one) and the W fragments of step 0 share one round trip.
        typedef __attribute__((address_space(3))) volatile int* lds_vint_t;
        const lds_vint_t flag_s = (lds_vint_t)(lds_ptr_t)(Ab + p.ids_off + 1280);                  // [chunk]: waves whose requests have landed
        {
            static_assert(RP <= 256, "one token id per thread");
            const int64_t F0 = doc0 * p.DL + t00, Fm = p.M * p.DL - 1;       // token row i of the tile = flattened token F0 + i (clamped)
            const int64_t f = F0 + (tid < RP ? tid : RP - 1);
            // the low dword of the 64-bit id: with the 8-byte load the dead upper half was re-used at once and hipcc waited for the load
            // (a round trip) before it issued the fc2 / W requests below
            const int idv = reinterpret_cast<const int*>(p.d_ids)[2 * (f < Fm ? f : Fm)];
            const int tr = tid < 16 * RT ? tid : 16 * RT - 1;
            int t = t00 + tr;
            const bool s1 = p.flat && t >= Tc;
            t = s1 ? t - Tc : t;
            const bool ok = tr < (p.flat ? p.TS : p.TPv) && t < p.PL && doc0 + (s1 ? 1 : 0) < p.M;
            const float wv_ = p.fc2w[ok ? t : 0] * (ok ? 1.0f : 0.0f);
            DF_LOAD_W(w, wp1 + (int64_t)(3 * DF_PHYS1(0)) * WSTEP)
            if (tid < 32) flag_s[tid] = 0;
            if (tid < RP) ids_s[tid] = idv;
            if (tid < 16 * RT) {
                float* const ws_ = reinterpret_cast<float*>(Ab + p.ids_off + 512);
                ws_[2 * tid] = s1 ? 0.f : wv_;          // interleaved {slot 0, slot 1}: one packed FMA per value in the epilogue
                ws_[2 * tid + 1] = s1 ? wv_ : 0.f;
            }
        }
        __syncthreads();
        // load slot k of this wave = pieces 64 (4k + wave) .. +63 of a chunk: piece -> (term, k-group, token row); pad pieces repeat the last one
        const _Float16* gb[NSLOT];
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            int q = 64 * (4 * k + wave) + lane;
            q = q < 8 * RP ? q : 8 * RP - 1;
            const int tt = q / (4 * RP), r2 = q - tt * 4 * RP, kgl = r2 / RP, tau = r2 - kgl * RP;
            gb[k] = p.ftab + ((int64_t)ids_s[tau] * 2 + tt) * p.EPT + kgl * 8;
        }
        // The request is inline assembly on purpose: for the builtin, hipcc puts s_waitcnt vmcnt(0) in front of every later ds_read (an LDS
        // read with no alias scope is assumed to depend on every pending LDS-DMA write), which also drains the W prefetch of the step.
        const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)dsm;
#define DF_TILE_LOAD(K, CC)                                                                                     \
        if (4 * (K) + wv < NI) {                                                                                \
            const _Float16* g_ = gb[K] + DF_PHYS1(CC) * 32;                                                     \
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds0 + (CC) * CHB + (4 * (K) + wv) * 1024), "v"(g_) \
                         : "memory", "m0");                                                                     \
        }
        // fragment address of row tile i (lane part): [k-group g][token row pr + 2 d][8 halves]; d = document boundaries before the row
        int arow[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int pr = 16 * i + c16, tt0 = t00 + pr;
            const int d = p.flat ? (tt0 >= Tc ? 1 : 0) + (tt0 >= 2 * Tc ? 1 : 0) : 0;
            arow[i] = (g * RP + pr + 2 * d) * 16;
        }
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) DF_TILE_LOAD(k, 0)
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            if (1 < C) { DF_TILE_LOAD(k, 1) }
        }
        if (!DF_G1FIRST) { DF_ZERO_ACC() }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunks 0 and 1 are in LDS
        int fl_ = 0;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t) af[0][t] = *reinterpret_cast<const f16x8*>(Ab + t * 4 * RP * 16 + arow[0]);
        DF_T(1)
        // k-step (chunk CI, tap U): MFMAs on the current sets; W fragments of the next step; the next step's A fragments from (CI, U+1) or
        // (CI+1, 0).  Requests complete in order, so a W fragment waits for every token-row request issued before it: chunk CI+2 is
        // requested in steps U = 0, 1 of chunk CI BEHIND the step's W loads (the W fragments of the next step are never stuck behind a
        // gather that may go to HBM; those of the step after have a step and a half of slack).  Hand-over without a barrier (ten barriers
        // per tile cost 4.5 K of 52 K cycles: the waves drift apart on their W streams and a barrier turns the mean delay into the maximum):
        // at the top of step U = 1 of chunk CI+1 a wave waits for ITS requests of chunk CI+2 (s_waitcnt vmcnt: 20 W loads + 2 requests
        // are younger) and counts itself in an LDS flag; a step and a half later, before the first fragment read of chunk CI+2, every wave
        // checks that the flag says four (it does unless a wave is more than a step behind; then it polls).
#define DF_AROW(I) arow[I]
#define DF_STEP1P(CB, CI, U, WC, WN)                                                      \
        {                                                                                 \
            const bool more_ = (CI) + 1 < C;                                              \
            const int sn_ = (U) < 2 ? 3 * DF_PHYS1(CI) + (U) + 1 : (more_ ? 3 * DF_PHYS1((CI) + 1) : 3 * DF_PHYS1(CI) + 2); \
            const _Float16* wn_ = wp1 + (int64_t)sn_ * WSTEP;                             \
            const unsigned char* ac_ = (CB) + (U) * 16;                                   \
            const unsigned char* an_ = (U) < 2 ? (CB) + ((U) + 1) * 16 : (more_ ? (CB) + CHB : (CB) + 32); \
            DF_STEPR(WC, WN, wn_, ac_, an_, 4 * RP * 16, DF_AROW, DF_G1FIRST && (U) == 0 && (CI) == 0,  \
                     if (!DF_X_NOTILE && (U) < 2 && (n_ == 24 || n_ == 26) && 2 * (U) + (n_ - 24) / 2 < NSLOT) { \
                         if ((CI) + 2 < C) { DF_TILE_LOAD(2 * (U) + (n_ - 24) / 2, (CI) + 2) }   \
                     }                                                                    \
                     if (!DF_X_NOBAR && (U) == 1 && n_ == 0 && (CI) >= 1) {               \
                         if ((CI) + 1 < C) {          /* publish: this wave's part of chunk CI+1 has landed */ \
                             if ((CI) + 2 < C) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");  \
                             else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");       \
                             if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"((unsigned)(size_t)(flag_s + (CI) + 1)), "v"(1) : "memory"); \
                         }                                                                \
                     }                                                                    \
                     if (!DF_X_NOBAR && (U) == 2 && n_ == 15 * (RT - 1) - 14 && (CI) >= 1) fl_ = flag_s[(CI) + 1 < C ? (CI) + 1 : (CI)]; \
                     if (!DF_X_NOBAR && (U) == 2 && n_ == 15 * (RT - 1) - 4 && (CI) >= 1 && (CI) + 1 < C) { \
                         while (__builtin_amdgcn_readfirstlane(fl_) < 4) fl_ = flag_s[(CI) + 1];  \
                         asm volatile("" ::: "memory");                                   \
                     }) \
        }
        {
            const unsigned char* cb = Ab;
            int c = 0;
#pragma unroll 1
            for (; c + 1 < C; c += 2) {       // C is even (EPT % 64 == 0): a tail of three steps behind the loop made hipcc permute all accumulators at the exit
                DF_STEP1P(cb, c, 0, w, wb)
                DF_STEP1P(cb, c, 1, wb, w)
                DF_STEP1P(cb, c, 2, w, wb)
                cb += CHB;
                DF_STEP1P(cb, c + 1, 0, wb, w)
                DF_STEP1P(cb, c + 1, 1, w, wb)
                DF_STEP1P(cb, c + 1, 2, wb, w)
                cb += CHB;
            }
        }
    }
    DF_T(2)
    DF_MMA_DRAIN();
    if constexpr (PL) lds_barrier();       // the P planes overwrite the token tile: every wave is past its last fragment read
    // ================= tanh, max-pool over rows, split into the P planes =================
    // C layout of the 16x16 tile: column = lane & 15, row = 4 * (lane >> 4) + r.  Pooled row pr needs rows pr .. pr+P-1: the rest of
    // this lane's quad and the quad of the next 16-lane group (the next row tile's group 0 for group 3), fetched by ds_bpermute.
    {
        const int nb_idx = ((lane + 16) & 63) * 4;
        const int P = POOL ? POOL : p.P;
#pragma unroll
        for (int j = 0; j < DF_CT; ++j) {
            __builtin_amdgcn_sched_barrier(0);      // one column tile at a time: interleaving the tiles spilled (RT = 6)
            const int col = 80 * wave + 16 * j + c16;
            const float biasz = (col < p.NF ? p.b1[col] : 0.f) * DF_2LOG2E;
            float v[RT][4], x[RT + 1][4];
            if constexpr (PL) {                                            // packed fp32 math: the same operations, two values per issue slot
                const f32x2 k1 = {1.0f / 2048.0f, 1.0f / 2048.0f}, k2 = {DF_2LOG2E, DF_2LOG2E}, b2 = {biasz, biasz}, one = {1.f, 1.f}, m2 = {-2.f, -2.f};
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    const bool ag = i < RT - (RT > 4 ? RT - 4 : 0);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 ax = {df_acc(acx[j][i], 2 * h, ag), df_acc(acx[j][i], 2 * h + 1, ag)};
                        const f32x2 ac = {df_acc(acc[j][i], 2 * h, ag), df_acc(acc[j][i], 2 * h + 1, ag)};
                        const f32x2 z = __builtin_elementwise_fma(__builtin_elementwise_fma(ax, k1, ac), k2, b2);
                        const f32x2 den = f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])} + one;
                        const f32x2 t2 = __builtin_elementwise_fma(m2, f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])}, one);
                        v[i][2 * h] = t2[0];
                        v[i][2 * h + 1] = t2[1];
                        x[i][2 * h] = df_bperm(t2[0], nb_idx);
                        x[i][2 * h + 1] = df_bperm(t2[1], nb_idx);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[i][r] = df_tanh_z(fmaf(fmaf(acx[j][i][r], 1.0f / 2048.0f, acc[j][i][r]), DF_2LOG2E, biasz));
                        x[i][r] = df_bperm(v[i][r], nb_idx);
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) x[RT][r] = x[RT - 1][r];           // rows past the tile: the last P-1 pooled rows are never used
            const int sk = col >> 5, kg = (col >> 3) & 3, e = col & 6, odd = col & 1;
            unsigned short* dst = Pp + (sk * 4 + kg) * KG + e;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                float c[8], m[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    c[r] = v[i][r];
                    c[4 + r] = g < 3 ? x[i][r] : x[i + 1][r];
                }
                if (POOL == 5) {                                           // shared partial maxima: 7 v_max / v_max3 for the four windows
                    const float t = fmaxf(c[3], c[4]), u = fmaxf(c[1], c[2]), q = fmaxf(c[5], c[6]);
                    m[0] = fmaxf(fmaxf(c[0], u), t);
                    m[1] = fmaxf(fmaxf(u, t), c[5]);
                    m[2] = fmaxf(fmaxf(c[2], t), q);
                    m[3] = fmaxf(fmaxf(t, q), c[7]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        m[r] = c[r];
#pragma unroll
                        for (int q = 1; q < 5; ++q) m[r] = q < P ? fmaxf(m[r], c[r + q]) : m[r];
                    }
                }
                // Two neighbouring columns share a dword of a P plane: the even lane stores rows r = 0, 2 of the pair, the odd lane rows
                // 1, 3 (values swapped through DPP quad_perm [1,0,3,2]) -- 4-byte stores instead of twice as many 2-byte ones, which
                // collided on LDS banks (SQ_LDS_BANK_CONFLICT was 10 % of the kernel's cycles).
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const float mine = odd ? m[2 * hh + 1] : m[2 * hh], give = odd ? m[2 * hh] : m[2 * hh + 1];
                    const float got = dpp_mov<0xB1>(give);
                    const float lo = odd ? got : mine, hi = odd ? mine : got;
                    const fp16x2_t h1 = __builtin_amdgcn_cvt_pkrtz(lo, hi);
                    const f32x2 rs = (f32x2{lo, hi} - f32x2{(float)h1[0], (float)h1[1]}) * f32x2{2048.0f, 2048.0f};      // v_pk_add / v_pk_mul
                    const fp16x2_t h2 = __builtin_amdgcn_cvt_pkrtz(rs[0], rs[1]);
                    const int row = 16 * i + 4 * g + 2 * hh + odd;
                    *reinterpret_cast<unsigned*>(dst + row * 8) = __builtin_bit_cast(unsigned, h1);
                    *reinterpret_cast<unsigned*>(dst + DF_S2 * 4 * KG + row * 8) = __builtin_bit_cast(unsigned, h2);
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!PL) {
#pragma unroll
        for (int j = 0; j < DF_CT; ++j)
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                acx[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int rot2 = PL && DF_ROT ? (int)(blockIdx.x % (unsigned)DF_S2) : 0;
    DF_LOAD_W(w, wp2 + (int64_t)rot2 * WSTEP)
    __syncthreads();
    if constexpr (PL) {
#pragma unroll
        for (int t = 0; t < 2; ++t) af[0][t] = *reinterpret_cast<const f16x8*>(Pp + rot2 * 4 * KG + t * DF_S2 * 4 * KG + foff);
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < RT; ++i) afa[i][t] = *reinterpret_cast<const f16x8*>(Pp + t * DF_S2 * 4 * KG + foff + i * 128);
    }
    DF_T(3)

    // ================= GEMM 2: conv_d2 (A = P planes, static in LDS: no barriers) =================
#define DF_STEP2(S, AFC, AFN, WC, WN)                                                     \
    {                                                                                     \
        const int sn_ = (S) + 1 < DF_S2 ? (S) + 1 : DF_S2 - 1;                            \
        const _Float16* wn_ = wp2 + (int64_t)sn_ * WSTEP;                                 \
        const unsigned short* pn_ = Pp + sn_ * 4 * KG + foff;                             \
        _Pragma("clang loop unroll(full)") for (int n_ = 0; n_ < 15 * RT; ++n_) {                          \
            df_mma_n<RT>(n_, acc, acx, AFC, WC);                                          \
            if (n_ == 8) { DF_KEEP_HEAD(WN, AFN) }                                        \
            if (n_ % 6 == 2 && n_ / 6 < 2 * DF_CT)                                        \
                WN[(n_ / 6) >> 1][(n_ / 6) & 1] = *reinterpret_cast<const f16x8*>(wn_ + (n_ / 6) * 512); \
            if (n_ >= 11 && n_ % 6 == 5 && (n_ - 11) / 6 < 2 * RT)     /* after the head fence, which still names AFN */ \
                AFN[((n_ - 11) / 6) % RT][((n_ - 11) / 6) / RT] =                         \
                    *reinterpret_cast<const f16x8*>(pn_ + (((n_ - 11) / 6) / RT) * DF_S2 * 4 * KG + (((n_ - 11) / 6) % RT) * 128); \
            __builtin_amdgcn_sched_barrier(0);                                            \
        }                                                                                 \
        DF_KEEP(WC, AFC)                                                                  \
    }
#define DF_PROW(I) ((I) * 256)
#define DF_PHYS2(S_) ((S_) + rot2 < DF_S2 ? (S_) + rot2 : (S_) + rot2 - DF_S2)
#define DF_STEP2R(S, WC, WN, FIRST)                                                       \
    {                                                                                     \
        const int sc_ = DF_PHYS2(S), sn_ = DF_PHYS2((S) + 1 < DF_S2 ? (S) + 1 : DF_S2 - 1); \
        const _Float16* wn_ = wp2 + (int64_t)sn_ * WSTEP;                                 \
        const unsigned char* pc_ = reinterpret_cast<const unsigned char*>(Pp + sc_ * 4 * KG + foff);  \
        const unsigned char* pn_ = reinterpret_cast<const unsigned char*>(Pp + sn_ * 4 * KG + foff);  \
        DF_STEPR(WC, WN, wn_, pc_, pn_, DF_S2 * 4 * KG * 2, DF_PROW, FIRST, )             \
    }
    if constexpr (PL) {
        DF_STEP2R(0, w, wb, true)            // starts every accumulator chain (C = 0)
        DF_STEP2R(1, wb, w, false)
#pragma unroll 1
        for (int s2 = 2; s2 < DF_S2; s2 += 2) {
            DF_STEP2R(s2, w, wb, false)
            DF_STEP2R(s2 + 1, wb, w, false)
        }
    } else {
#pragma unroll 1
        for (int s2 = 0; s2 < DF_S2; s2 += 2) {
            DF_STEP2(s2, afa, afb, w, wb)
            DF_STEP2(s2 + 1, afb, afa, wb, w)
        }
    }

    DF_T(4)
    DF_MMA_DRAIN();
    // ================= tanh, fc2 over the tile's rows =================
    // Row weight = fc2_w[t] when the row is one of the tile's own pooled rows and its window stays inside the document, else 0.  A
    // flattened tile touches at most two documents: slot 0 = the document of the tile's first row, slot 1 = the next one.
    {
        const int own = p.flat ? p.TS : p.TPv;
        float wrow[PL ? 1 : RT][4], w1row[PL ? 1 : RT][4];
        const float* const wrow_s = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(dsm) + p.ids_off + 512);   // plane mode
#pragma unroll
        for (int i = 0; i < (PL ? 0 : RT); ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pr = 16 * i + 4 * g + r;
                int t = t00 + pr;
                const bool s1 = p.flat && t >= Tc;               // an owned row is in the tile's first document or the one after it
                t = s1 ? t - Tc : t;
                const bool ok = pr < own && t < p.PL && doc0 + (s1 ? 1 : 0) < p.M;
                const float wv = p.fc2w[ok ? t : 0] * (ok ? 1.0f : 0.0f);
                wrow[i][r] = s1 ? 0.f : wv;
                w1row[i][r] = s1 ? wv : 0.f;
            }
        float* out = p.partial + (int64_t)blockIdx.x * 2 * DF_NFP;
#pragma unroll
        for (int j = 0; j < DF_CT; ++j) {
            __builtin_amdgcn_sched_barrier(0);      // one column tile at a time: interleaving the tiles spilled (RT = 6)
            const int col = 80 * wave + 16 * j + c16;
            const float biasz = (col < p.NF ? p.b2[col] : 0.f) * DF_2LOG2E;
            float sum = 0.f, sum1 = 0.f;
            if constexpr (PL) {
                // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32: two values per issue slot; one wave per SIMD, the epilogue is issue-bound)
                f32x2 s2 = {0.f, 0.f};
                const f32x2 k1 = {1.0f / 2048.0f, 1.0f / 2048.0f}, k2 = {DF_2LOG2E, DF_2LOG2E}, b2 = {biasz, biasz};
                const f32x2 one = {1.f, 1.f}, m2 = {-2.f, -2.f};
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    const bool ag = i < RT - (RT > 4 ? RT - 4 : 0);
                    const f32x4 wlo = *reinterpret_cast<const f32x4*>(wrow_s + 2 * (16 * i + 4 * g));        // rows 4g, 4g+1: {w, w1, w, w1}
                    const f32x4 whi = *reinterpret_cast<const f32x4*>(wrow_s + 2 * (16 * i + 4 * g) + 4);    // rows 4g+2, 4g+3
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 ax = {df_acc(acx[j][i], 2 * h, ag), df_acc(acx[j][i], 2 * h + 1, ag)};
                        const f32x2 ac = {df_acc(acc[j][i], 2 * h, ag), df_acc(acc[j][i], 2 * h + 1, ag)};
                        const f32x2 z = __builtin_elementwise_fma(__builtin_elementwise_fma(ax, k1, ac), k2, b2);
                        const f32x2 den = f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])} + one;
                        const f32x2 d2 = __builtin_elementwise_fma(m2, f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])}, one);
                        const f32x4 wq = h ? whi : wlo;
                        s2 = __builtin_elementwise_fma(f32x2{wq[0], wq[1]}, f32x2{d2[0], d2[0]}, s2);
                        s2 = __builtin_elementwise_fma(f32x2{wq[2], wq[3]}, f32x2{d2[1], d2[1]}, s2);
                    }
                }
                sum = s2[0];
                sum1 = s2[1];
            } else {
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float d2 = df_tanh_z(fmaf(fmaf(acx[j][i][r], 1.0f / 2048.0f, acc[j][i][r]), DF_2LOG2E, biasz));
                        sum = fmaf(wrow[i][r], d2, sum);
                        sum1 = fmaf(w1row[i][r], d2, sum1);
                    }
            }
            sum += df_bperm(sum, ((lane + 16) & 63) * 4);
            sum += df_bperm(sum, ((lane + 32) & 63) * 4);
            sum1 += df_bperm(sum1, ((lane + 16) & 63) * 4);
            sum1 += df_bperm(sum1, ((lane + 32) & 63) * 4);
            if (g == 0) {
                out[col] = sum;
                out[DF_NFP + col] = sum1;
            }
        }
    }
    DF_T(5)
}

// m1[pair][f] = tanh(fc2_b + qv[b][f] * sum over the tiles (and slots) that hold pooled rows of the pair's document)
__global__ void duet_doc_finish_kernel(const float* __restrict__ partial, const float* __restrict__ qv, const float* __restrict__ fc2b, int N,
                                       int NF, int flat, int TS, int Tc, int PL, int ntile, int64_t total, float* __restrict__ m1) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t pair = i / NF;
    const int f = (int)(i % NF);
    float s = 0.f;
    if (flat) {
        const int64_t ta = pair * Tc / TS, tb = (pair * Tc + PL - 1) / TS;
        for (int64_t t = ta; t <= tb; ++t) {
            const int slot = (int)(pair - t * TS / Tc);          // 0 or 1
            s += partial[(t * 2 + slot) * DF_NFP + f];
        }
    } else {
        for (int t = 0; t < ntile; ++t) s += partial[((pair * ntile + t) * 2) * DF_NFP + f];
    }
    m1[i] = fast_tanh(fc2b[0] + qv[(pair / N) * NF + f] * s);
}

// Tiling: a tile of `rows` conv positions yields rows - (P-1) pooled rows.  Long documents (>= that many positions): tiles are cut
// from the flattened axis with that stride.  Short documents: ntile tiles per document, the pooled rows shared out evenly.
static void duet_doc_tiling(int64_t M, int DL, int P, int rows, DuetDocArgs* a, int64_t* tiles) {
    const int Tc = DL - 2, PL = Tc - P + 1, cap = rows - (P - 1);
    a->Tc = Tc; a->PL = PL; a->TS = cap;
    a->flat = Tc >= cap ? 1 : 0;
    a->ntile = (PL + cap - 1) / cap;
    a->TPv = (PL + a->ntile - 1) / a->ntile;
    *tiles = a->flat ? (M * Tc + cap - 1) / cap : M * a->ntile;
}

bool duet_doc_usable(int NF, int P, int E, int DL, int K1P) {
    return NF <= DF_NFP && NF % 4 == 0 && P >= 1 && P <= 5 && E % 4 == 0 && E >= 4 && DL >= P + 2 && K1P % 32 == 0 && K1P >= 3 * E;
}

// Rows per tile: with the table planes the tile is 96 rows when documents are long enough for flattened 96-row tiles (W fragments are
// streamed once per tile: 1.5x fewer L2 bytes and MFMA-free epilogue cycles per row), else 64.
// LDS of a plane-mode tile: the token tile (C column chunks) or the P planes that later take its place, ids + fc2 weights + flags
template <int RT>
static size_t duet_pl_lds(int C) {
    return std::max((size_t)C * DfLayoutP<RT>::CHB, (size_t)DfLayout<RT>::P_HALVES * 2) + 1536;
}
// plane mode needs the whole token tile in LDS (EPT / 32 chunks) and one flag per chunk: wide embeddings fall back to the fp32-table form
bool duet_planes_fit(int EPT) {
    return EPT > 0 && EPT % 64 == 0 && EPT / 32 <= 32 && duet_pl_lds<DF_RT>(EPT / 32) <= 160 * 1024;
}
int duet_doc_rows(bool planes, int DL, int P, int EPT) {
    return planes && !tun(g_tun.duet_rows64) && DL - 2 >= 96 - (P - 1) && duet_pl_lds<6>(EPT / 32) <= 160 * 1024 ? 96 : DF_ROWS;
}

size_t duet_doc_partial_floats(int64_t M, int DL, int P, bool planes, int EPT) {
    DuetDocArgs a;
    int64_t tiles;
    duet_doc_tiling(M, DL, P, duet_doc_rows(planes, DL, P, EPT), &a, &tiles);
    return (size_t)tiles * 2 * DF_NFP;
}

template <int RT, bool PLN>
static void duet_doc_launch_t(const DuetDocArgs& a, int64_t tiles, size_t lds, int P, hipStream_t st) {
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)duet_doc_kernel<RT, 5, PLN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)duet_doc_kernel<RT, 0, PLN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (P == 5) hipLaunchKernelGGL((duet_doc_kernel<RT, 5, PLN>), dim3((unsigned)tiles), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((duet_doc_kernel<RT, 0, PLN>), dim3((unsigned)tiles), dim3(256), lds, st, a);
}

int launch_duet_doc(const int64_t* d_ids, const float* table, int E, int DL, int64_t M, int N, const void* wf1, int K1P, const void* wf2,
                    const void* ftab, const void* wf1c, int EPT, const float* b1, const float* b2, const float* fc2w, const float* fc2b,
                    const float* qv, int NF, int P, float* partial, float* m1, hipStream_t st) {
    NIR_REQUIRE(duet_doc_usable(NF, P, E, DL, K1P), "duet_doc: unsupported shape NF=%d pool=%d E=%d DL=%d K1P=%d", NF, P, E, DL, K1P);
    const bool planes = ftab && wf1c && EPT >= E && duet_planes_fit(EPT);
    if (M == 0) return 0;
    DuetDocArgs a;
    a.d_ids = d_ids; a.table = table; a.wf1 = (const _Float16*)wf1; a.wf2 = (const _Float16*)wf2; a.b1 = b1; a.b2 = b2; a.fc2w = fc2w;
    a.partial = partial; a.M = M; a.E = E; a.DL = DL; a.S1 = K1P / 32; a.NF = NF; a.P = P;
    a.ftab = (const _Float16*)ftab; a.wf1c = (const _Float16*)wf1c; a.EPT = EPT; a.C = EPT / 32; a.ids_off = 0;
    const int rows = duet_doc_rows(planes, DL, P, EPT);
    int64_t tiles;
    duet_doc_tiling(M, DL, P, rows, &a, &tiles);
    {
        ProfScope ps(prof_shape_name("duet_doc_kernel", tiles * rows, NF, 3 * E), st);
        if (!planes) {
            duet_doc_launch_t<DF_RT, false>(a, tiles, DfLayout<DF_RT>::LDS, P, st);
        } else if (rows == 96) {
            const size_t lds = duet_pl_lds<6>(a.C);
            a.ids_off = (int)(lds - 1536);
            duet_doc_launch_t<6, true>(a, tiles, lds, P, st);
        } else {
            const size_t lds = duet_pl_lds<DF_RT>(a.C);
            a.ids_off = (int)(lds - 1536);
            duet_doc_launch_t<DF_RT, true>(a, tiles, lds, P, st);
        }
    }
    NIR_CHECK_LAUNCH("duet_doc_kernel");
    {
        const int64_t total = M * NF;
        ProfScope ps("duet_doc_finish_kernel", st);
        hipLaunchKernelGGL(duet_doc_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, partial, qv, fc2b, N, NF, a.flat, a.TS,
                           a.Tc, a.PL, a.ntile, total, m1);
    }
    NIR_CHECK_LAUNCH("duet_doc_finish_kernel");
    return 0;
}

}  // namespace nir

#ifdef DF_TIMING
extern "C" int nir_debug_duet_timing(long long* out16) {
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(nir::df_dbg), sizeof(long long) * 16);
}
#endif
