// nir_linear_f32 / nir_rowdot_f32: nn.Linear and Conv1d-as-GEMM with the embedding gather fused into the
// A-operand load.  fp32 MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products/accumulation at the fp32 vector
// rate, one VGPR per operand per lane (cdna guide section 3).
//
// Tiling: 256 threads = 4 waves, block tile 64(M) x 64(N) x 64(K); each wave owns a 32x32 accumulator
// (16 VGPRs).  Operand tiles are staged in LDS k-contiguous with a +4-float row pad (row stride 68 floats):
// a lane's ds_read_b128 then covers 4 consecutive k of its row and the 16-lane read groups hit 16 distinct
// 16-byte slots (conflict-free).  The 4 floats feed 4 successive MFMAs; the k-order inside an 8-wide group
// is permuted identically for A and W (lane>>5 selects which half), which leaves the sum unchanged.
#include "common.hpp"
#include <algorithm>
#include <mutex>
#include <stdlib.h>

namespace nir {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* a;
    int64_t lda;
    const int64_t* ids;
    const float* table;
    int E;
    int64_t rows_per_seq, seq_stride;
    const float* w;
    int64_t ldw;
    const float* bias;
    const float* bias2;
    float* c;
    int64_t ldc;
    int64_t M;
    int N, K, act;
    const float* add;    // optional addend [M, ldadd] added before the activation (e.g. precomputed x W_ih^T)
    int64_t ldadd;
};

constexpr int ACT_MAXOUT2 = 16;   // internal: out[m, n/2] = max(v[m,n], v[m,n+1])  (maxout pool 2, maxout.py:77-81)
// internal: the attention MLP's second layer fused into the first one's epilogue (cars.py:671-691, Linear -> Tanh -> Linear(D,1)):
// out[m, n/16] = sum over the 16 columns n..n+15 of tanh(v[m,n]) * add[n]   (add = the [N] weight row of the second layer;
// ldadd unused); the consumer sums the N/16 partials of a row, so the [M,N] hidden activation never reaches HBM.
constexpr int ACT_TANH_ROWDOT16 = 17;
// flag OR-ed into `act` by internal callers: every element of both operands is bounded by 2^15 in magnitude (tanh / sigmoid
// outputs, embedding tables and weights checked when they are packed) -> the large-GEMM path may use the two-term fp16 split
constexpr int ACT_BOUNDED = 0x100;

// epilogue shared by both kernels: bias / addend / activation / optional pairwise maxout over adjacent columns
__device__ __forceinline__ void gemm_store(const GemmArgs& p, int64_t m, int n, float v, float bsum) {
    v += bsum;
    if (p.act == ACT_TANH_ROWDOT16) {
        v = n < p.N ? fast_tanh(v) * p.add[n] : 0.f;
        v += dpp_mov<0xB1>(v);     // 16-lane row reduction (the 16 lanes of a DPP row hold 16 consecutive columns)
        v += dpp_mov<0x4E>(v);
        v += dpp_mov<0x141>(v);
        v += dpp_mov<0x140>(v);
        if (m < p.M && n < p.N && !(n & 15)) p.c[m * p.ldc + (n >> 4)] = v;
        return;
    }
    if (p.add && m < p.M && n < p.N) v += p.add[m * p.ldadd + n];
    if (p.act == NIR_ACT_TANH) v = fast_tanh(v);
    else if (p.act == NIR_ACT_RELU) v = fmaxf(v, 0.f);
    if (p.act == ACT_MAXOUT2) {
        const float other = dpp_mov<0xB1>(v);          // partner column = adjacent lane (quad_perm [1,0,3,2])
        v = fmaxf(v, other);
        if (m < p.M && n < p.N && !(n & 1)) p.c[m * p.ldc + (n >> 1)] = v;
    } else if (m < p.M && n < p.N) {
        p.c[m * p.ldc + n] = v;
    }
}

// BK = 32 (36 KB of LDS, 4 blocks = 16 waves per CU) beats BK = 64 (68 KB, 2 blocks per CU) on this path's shapes, whose K
// is short (300..900) so a block lives only 5-30 k-tiles and prologue/epilogue must hide behind other blocks: measured
// 71680x1024x300 gather 68 -> 86 TFLOP/s, 921600x300x900 86 -> 96, 4096^3 107 -> 104.
#ifndef NIR_GEMM_BK
#define NIR_GEMM_BK 32
#endif
constexpr int BM = 64, BN = 64, BK = NIR_GEMM_BK, LDS_LD = BK + 4;   // odd multiple of 4 floats: conflict-free ds_read_b128
constexpr int TPR = BK / 4, RPP = 256 / TPR;                // threads per tile row, rows per pass
constexpr int LPT = BM * BK / 4 / 256;                        // float4 loads per thread per operand tile (4)

// Software pipeline: the global loads of tile t+1 (gathered embedding rows for the A operand) are issued before the
// MFMAs of tile t and land in the other LDS buffer afterwards -- one barrier per tile.
template <bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][LDS_LD]
    float* Ws = smem + 2 * BM * LDS_LD;        // [2][BN][LDS_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware block order (blocks are dealt round-robin to the 8 XCDs, each with its own L2): the N-blocks that
    // share one A row-block get ids 8 apart, i.e. the same XCD back to back, so the (gathered) A rows are fetched
    // from HBM once and re-read from that L2.
    const int nb = (p.N + BN - 1) / BN;
    const int64_t mb = (p.M + BM - 1) / BM;
    const int64_t bid = blockIdx.x;
    const int64_t tq = bid >> 3;
    const int64_t mblk = (tq / nb) * 8 + (bid & 7);
    if (mblk >= mb) return;
    const int64_t m0 = mblk * BM;
    const int n0 = (int)(tq % nb) * BN;
    const int lr = tid / TPR, lk = (tid % TPR) * 4;   // TPR threads cover one BK-float row; rows lr + RPP*i

    const float* arow[LPT];
    int64_t aidx[LPT];
    bool aval[LPT];
    const float* wrow[LPT];
    bool wval[LPT];
    // conv-style gather with 2..3 taps (K = taps*E): the tap rows are resolved once per block as well, so a k-tile load
    // is ONE dependent access (table row) instead of two (id, then row) plus an integer division
    const bool taps3 = p.ids && p.K > p.E && p.K <= 3 * p.E;
    const float* arow1[LPT];
    const float* arow2[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        int64_t m = m0 + lr + RPP * i;
        aval[i] = m < p.M;
        arow[i] = arow1[i] = arow2[i] = nullptr;
        aidx[i] = 0;
        if (aval[i]) {
            if (p.ids) {
                aidx[i] = (m / p.rows_per_seq) * p.seq_stride + (m % p.rows_per_seq);
                // single-segment gather (K <= E): resolve the row pointer once, not once per k-tile
                if (p.K <= p.E) arow[i] = p.table + p.ids[aidx[i]] * (int64_t)p.E;
                if (taps3) {   // pointers pre-biased by the tap's k offset: element k of tap s is arow_s[k]
                    arow[i] = p.table + p.ids[aidx[i]] * (int64_t)p.E;
                    arow1[i] = p.table + p.ids[aidx[i] + 1] * (int64_t)p.E - p.E;
                    arow2[i] = p.K > 2 * p.E ? p.table + p.ids[aidx[i] + 2] * (int64_t)p.E - 2 * p.E : arow1[i];
                }
            } else {
                arow[i] = p.a + m * p.lda;
            }
        }
        int n = n0 + lr + RPP * i;
        wval[i] = n < p.N;
        wrow[i] = p.w + (int64_t)(wval[i] ? n : 0) * p.ldw;
    }

    float4 ra[LPT], rw[LPT];
    auto load_tile = [&](int k0) {
        const int k = k0 + lk;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            rw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (VEC) {
                if (k < p.K) {
                    if (aval[i]) {
                        const float* src;
                        if (taps3) {
                            src = (k < p.E ? arow[i] : (k < 2 * p.E ? arow1[i] : arow2[i])) + k;
                        } else if (p.ids && p.K > p.E) {
                            int seg = k / p.E;
                            src = p.table + p.ids[aidx[i] + seg] * (int64_t)p.E + (k - seg * p.E);
                        } else {
                            src = arow[i] + k;
                        }
                        ra[i] = *reinterpret_cast<const float4*>(src);
                    }
                    if (wval[i]) rw[i] = *reinterpret_cast<const float4*>(wrow[i] + k);
                }
            } else {
                float ta[4] = {0.f, 0.f, 0.f, 0.f}, tw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int kk = k + e;
                    if (kk < p.K) {
                        if (aval[i]) {
                            if (p.ids && p.K > p.E) {
                                int seg = kk / p.E;
                                ta[e] = p.table[p.ids[aidx[i] + seg] * (int64_t)p.E + (kk - seg * p.E)];
                            } else {
                                ta[e] = arow[i][kk];
                            }
                        }
                        if (wval[i]) tw[e] = wrow[i][kk];
                    }
                }
                ra[i] = make_float4(ta[0], ta[1], ta[2], ta[3]);
                rw[i] = make_float4(tw[0], tw[1], tw[2], tw[3]);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            *reinterpret_cast<float4*>(&As[(buf * BM + lr + RPP * i) * LDS_LD + lk]) = ra[i];
            *reinterpret_cast<float4*>(&Ws[(buf * BN + lr + RPP * i) * LDS_LD + lk]) = rw[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);   // in flight during the MFMAs below
        const float* ap = &As[(buf * BM + wm * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * 4];
        const float* bp = &Ws[(buf * BN + wn * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * 4];
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            float4 a4 = *reinterpret_cast<const float4*>(ap + q * 8);
            float4 b4 = *reinterpret_cast<const float4*>(bp + q * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);        // the other buffer was last read in iteration kt-1
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int n = n0 + wn * 32 + (lane & 31);
    float bsum = 0.f;
    if (n < p.N) {
        if (p.bias) bsum += p.bias[n];
        if (p.bias2) bsum += p.bias2[n];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        gemm_store(p, m, n, acc[r], bsum);
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Small-M path (session LSTM steps, ranknet, attention MLPs on a handful of rows): the 64x64 tiling would leave
// most CUs idle and serialise K.  Here one workgroup owns ONE 16x16 output tile, its 4 waves split K, operands are
// read straight from L2 as MFMA fragments (v_mfma_f32_16x16x4_f32; lane group g = lane>>4 takes k = 16q+4g..+3 for
// both operands, one float4 per 4 MFMAs) and the 4 partial tiles are summed through LDS.
template <bool VEC>
__global__ __launch_bounds__(256) void gemm16_kernel(GemmArgs p) {
    __shared__ float red[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int64_t m = (int64_t)blockIdx.x * 16 + i;
    const int n = blockIdx.y * 16 + i;
    const bool mval = m < p.M, nval = n < p.N;
    const float* arow = nullptr;
    int64_t aidx = 0;
    if (mval) {
        if (p.ids) {
            aidx = (m / p.rows_per_seq) * p.seq_stride + (m % p.rows_per_seq);
            if (p.K <= p.E) arow = p.table + p.ids[aidx] * (int64_t)p.E;
        } else {
            arow = p.a + m * p.lda;
        }
    }
    const float* wrow = p.w + (int64_t)(nval ? n : 0) * p.ldw;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nq = (p.K + 15) / 16;
    if (VEC && !p.ids && nq >= 32) {               // K >= 512 (shorter K: one chunk per wave, the look-ahead only over-fetches: 7 -> 10 us at K = 256)
        // dense operands: k-groups in chunks of 4, two chunks in ping-pong -- the loads of the next chunk are issued before the MFMAs
        // of the current one.  (The `#pragma unroll 4` below is not honoured on this runtime-strided loop: every k-group paid its own
        // L2 round trip, 20 in a row at K = 1280.)  Unconditional loads from clamped addresses, masked afterwards.
        constexpr int CH = 4;
        const float* abase = p.a + (mval ? m : 0) * p.lda;
        auto loadc = [&](int q0, float4 (&a4)[CH], float4 (&b4)[CH]) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int k = 16 * (q0 + 4 * c) + 4 * g;
                const int kc = k < p.K ? k : 0;
                a4[c] = *reinterpret_cast<const float4*>(abase + kc);
                b4[c] = *reinterpret_cast<const float4*>(wrow + kc);
            }
        };
        auto mmac = [&](int q0, float4 (&a4)[CH], float4 (&b4)[CH]) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int k = 16 * (q0 + 4 * c) + 4 * g;
                const float ma = (mval && k < p.K) ? 1.f : 0.f, mb = (nval && k < p.K) ? 1.f : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c].x * ma, b4[c].x * mb, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c].y * ma, b4[c].y * mb, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c].z * ma, b4[c].z * mb, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c].w * ma, b4[c].w * mb, acc, 0, 0, 0);
            }
        };
        float4 a0[CH], b0[CH], a1[CH], b1[CH];
        loadc(wave, a0, b0);
        for (int q0 = wave; q0 < nq; q0 += 8 * CH) {
            loadc(q0 + 4 * CH, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mmac(q0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            loadc(q0 + 8 * CH, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mmac(q0 + 4 * CH, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else
#pragma unroll 4
    for (int q = wave; q < nq; q += 4) {
        const int k = 16 * q + 4 * g;
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (VEC) {
            if (k < p.K) {
                if (mval) {
                    const float* src;
                    if (p.ids && p.K > p.E) {
                        int seg = k / p.E;
                        src = p.table + p.ids[aidx + seg] * (int64_t)p.E + (k - seg * p.E);
                    } else {
                        src = arow + k;
                    }
                    a4 = *reinterpret_cast<const float4*>(src);
                }
                if (nval) b4 = *reinterpret_cast<const float4*>(wrow + k);
            }
        } else {
            float ta[4] = {0.f, 0.f, 0.f, 0.f}, tw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int kk = k + e;
                if (kk < p.K) {
                    if (mval) {
                        if (p.ids && p.K > p.E) {
                            int seg = kk / p.E;
                            ta[e] = p.table[p.ids[aidx + seg] * (int64_t)p.E + (kk - seg * p.E)];
                        } else {
                            ta[e] = arow[kk];
                        }
                    }
                    if (nval) tw[e] = wrow[kk];
                }
            }
            a4 = make_float4(ta[0], ta[1], ta[2], ta[3]);
            b4 = make_float4(tw[0], tw[1], tw[2], tw[3]);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
        // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + r
        const int nn = blockIdx.y * 16 + (lane & 15);
        float bsum = 0.f;
        if (nn < p.N) {
            if (p.bias) bsum += p.bias[nn];
            if (p.bias2) bsum += p.bias2[nn];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t mm = (int64_t)blockIdx.x * 16 + (lane >> 4) * 4 + r;
            const float v = red[0][r * 64 + lane] + red[1][r * 64 + lane] + red[2][r * 64 + lane] + red[3][r * 64 + lane];
            gemm_store(p, mm, nn, v, bsum);
        }
    }
}

// Mid-size path (hundreds to a few thousand rows: CARS' maxout layers over B*S*N candidate rows).  Same exact-fp32 arithmetic and K split
// as gemm16_kernel, but a workgroup owns a 32x32 output block: a wave's two A and two W fragments of a k-group feed 16 MFMAs instead of
// 4 (half the L2 bytes per flop, twice the MFMA work behind every load).  1120x512x1024: 43 -> 27 us.
template <int RA>                                  // RA row tiles x 2 column tiles of 16x16 per workgroup (16 RA rows x 32 columns)
__global__ __launch_bounds__(256) void gemm32_kernel(GemmArgs p) {
    __shared__ float red[4][2 * RA][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int64_t mbase = (int64_t)blockIdx.x * (16 * RA);
    const int nbase = blockIdx.y * 32;
    const float* ar[RA];
    const float* wr[2];
    float am[RA], wm[2];
#pragma unroll
    for (int h = 0; h < RA; ++h) {
        const int64_t m = mbase + 16 * h + i;
        am[h] = m < p.M ? 1.f : 0.f;
        ar[h] = p.a + (m < p.M ? m : p.M - 1) * p.lda + 4 * g;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = nbase + 16 * h + i;
        wm[h] = n < p.N ? 1.f : 0.f;
        wr[h] = p.w + (int64_t)(n < p.N ? n : p.N - 1) * p.ldw + 4 * g;
    }
    f32x4 acc[RA][2];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nq = p.K / 16;                       // K % 16 == 0 (launcher)
    // Two operand sets in ping-pong: the loads of k-group q+4 are issued before the MFMAs of k-group q (the strided k-loop is not
    // unrolled by the compiler, and without the sched_barriers every load sinks next to its first use).  With one workgroup per CU
    // (the launcher picks RA for a single round) nothing else hides the L2 round trip of a k-group.  A k-group past the end re-reads
    // group 0 and is masked to zero.
    auto loadk = [&](int q, float4 (&a4)[RA], float4 (&b4)[2], float& km) {
        km = q < nq ? 1.f : 0.f;
        const int qc = q < nq ? q : 0;
#pragma unroll
        for (int h = 0; h < RA; ++h) a4[h] = *reinterpret_cast<const float4*>(ar[h] + 16 * qc);
#pragma unroll
        for (int h = 0; h < 2; ++h) b4[h] = *reinterpret_cast<const float4*>(wr[h] + 16 * qc);
    };
    auto mma = [&](float4 (&a4)[RA], float4 (&b4)[2], float km) {
#pragma unroll
        for (int h = 0; h < RA; ++h) {             // rows / columns past the edge contribute zeros (clamped address, 0/1 mask)
            const float m = am[h] * km;
            a4[h].x *= m; a4[h].y *= m; a4[h].z *= m; a4[h].w *= m;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            b4[h].x *= wm[h]; b4[h].y *= wm[h]; b4[h].z *= wm[h]; b4[h].w *= wm[h];
        }
#pragma unroll
        for (int a = 0; a < RA; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[a].x, b4[b].x, acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[a].y, b4[b].y, acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[a].z, b4[b].z, acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[a].w, b4[b].w, acc[a][b], 0, 0, 0);
            }
    };
    float4 a0[RA], b0[2], a1[RA], b1[2];
    float k0, k1;
    loadk(wave, a0, b0, k0);
    for (int q = wave; q < nq; q += 8) {
        loadk(q + 4, a1, b1, k1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0, k0);
        __builtin_amdgcn_sched_barrier(0);
        loadk(q + 8, a0, b0, k0);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1, k1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][2 * a + b][r * 64 + lane] = acc[a][b][r];
    __syncthreads();
    // wave w finishes the 16x16 tiles t = w, w + 4, ..: t = (a, b); C/D layout col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
    for (int t = wave; t < 2 * RA; t += 4) {
        const int a = t >> 1, b = t & 1;
        const int nn = nbase + 16 * b + (lane & 15);
        float bsum = 0.f;
        if (nn < p.N) {
            if (p.bias) bsum += p.bias[nn];
            if (p.bias2) bsum += p.bias2[nn];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t mm = mbase + 16 * a + (lane >> 4) * 4 + r;
            const float v = (red[0][t][r * 64 + lane] + red[1][t][r * 64 + lane]) + (red[2][t][r * 64 + lane] + red[3][t][r * 64 + lane]);
            gemm_store(p, mm, nn, v, bsum);
        }
    }
}

// Skinny-N path (N <= 64 over many rows: MatchTensor's 300 -> 40 embedding projection and 140 -> 50 channel projection on
// M = B*N*DL token rows).  The 64x64 tiling wastes 37 % of its N tile there and, with K = 140..300, spends most of a
// block's life in prologue/epilogue.  Here W (zero padded to 16*NT x Kp) is resident in LDS for the whole workgroup, a wave
// owns a 16-row tile, reads its A rows straight from global/L2 as MFMA fragments (lane (i = lane & 15, g = lane >> 4)
// takes k = 16q + 4g .. +3 of row i: one float4 feeds 4 v_mfma_f32_16x16x4_f32 per 16-column tile) with the loads of the
// next 4 k-groups in flight behind the MFMAs of the current 4, and loops over tiles persistently.
constexpr int SK_WAVES = 8, SK_CH = 4;

template <int NT>
__global__ __launch_bounds__(64 * SK_WAVES) void gemm_skinny_kernel(GemmArgs p, int G) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Kp = G * 16, LD = Kp + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int nch = (G + SK_CH - 1) / SK_CH;
    const int64_t ntiles = (p.M + 15) / 16;
    const int64_t tstep = (int64_t)gridDim.x * SK_WAVES;
    int64_t tile = (int64_t)blockIdx.x * SK_WAVES + wave;

    auto row_ptr = [&](int64_t t) -> const float* {
        const int64_t m = t * 16 + i;
        if (t >= ntiles || m >= p.M) return nullptr;
        if (p.ids) return p.table + p.ids[(m / p.rows_per_seq) * p.seq_stride + (m % p.rows_per_seq)] * (int64_t)p.E;
        return p.a + m * p.lda;
    };
    auto load_chunk = [&](const float* arow, int c, float4 (&dst)[SK_CH]) {
#pragma unroll
        for (int j = 0; j < SK_CH; ++j) {
            const int k = (c * SK_CH + j) * 16 + 4 * g;
            dst[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (arow && k < p.K) dst[j] = *reinterpret_cast<const float4*>(arow + k);
        }
    };
    // the first tile's id -> row -> data chain starts before W is staged, so the two latencies overlap
    const float* arow = row_ptr(tile);
    float4 a0[SK_CH], a1[SK_CH];
    load_chunk(arow, 0, a0);
    float bsum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = 16 * t + i;
        bsum[t] = 0.f;
        if (n < p.N) {
            if (p.bias) bsum[t] += p.bias[n];
            if (p.bias2) bsum[t] += p.bias2[n];
        }
    }
    {   // W -> LDS, 8 loads in flight per thread (a load->store loop would expose one memory round trip per iteration)
        const int kq = Kp / 4, total = 16 * NT * kq;
        for (int e0 = tid; e0 < total; e0 += 8 * 64 * SK_WAVES) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 64 * SK_WAVES;
                const int n = e / kq, k = (e - n * kq) * 4;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < total && n < p.N && k < p.K) v[u] = *reinterpret_cast<const float4*>(p.w + (int64_t)n * p.ldw + k);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 64 * SK_WAVES;
                const int n = e / kq, k = (e - n * kq) * 4;
                if (e < total) *reinterpret_cast<float4*>(&smem[n * LD + k]) = v[u];
            }
        }
    }
    __syncthreads();
    const float* wl = smem + i * LD + 4 * g;
    for (; tile < ntiles; tile += tstep) {
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto mma_chunk = [&](int c, const float4 (&a)[SK_CH]) {
#pragma unroll
            for (int j = 0; j < SK_CH; ++j) {
                const int q = c * SK_CH + j;
                if (q < G) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float4 b4 = *reinterpret_cast<const float4*>(wl + 16 * t * LD + q * 16);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b4.x, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b4.y, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b4.z, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b4.w, acc[t], 0, 0, 0);
                    }
                }
            }
        };
        const float* next_row = nullptr;
        for (int c = 0; c < nch; c += 2) {
            if (c + 1 < nch) load_chunk(arow, c + 1, a1);
            else next_row = row_ptr(tile + tstep);           // last chunk pair: start the next tile's id -> row chain
            mma_chunk(c, a0);
            if (c + 2 < nch) load_chunk(arow, c + 2, a0);
            else if (c + 1 < nch) next_row = row_ptr(tile + tstep);
            if (c + 1 < nch) mma_chunk(c + 1, a1);
        }
        load_chunk(next_row, 0, a0);                          // in flight behind this tile's epilogue
        arow = next_row;
        // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) gemm_store(p, tile * 16 + g * 4 + r, 16 * t + i, acc[t][r], bsum[t]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-precision GEMM: fp32 operands, fp32 result, products on the bf16 matrix cores.
//   Every fp32 operand element is split into three bf16 terms (common.hpp: split3) while its tile is staged into LDS; the
//   six leading cross terms go through v_mfma_f32_32x32x16_bf16 with fp32 accumulation (smallest terms first).  bf16 x bf16
//   products are exact in fp32, so the result carries fp32-class error (measured: max |err| 6e-7 vs 1.2e-6 for the plain
//   fp32 chain at K = 900) while the matrix pipe does 6 bf16 instructions where the f32 MFMA path needs 16 instruction
//   slots: peak = 2.5 PFLOP/s / 6 = 417 TFLOP/s fp32-equivalent, 2.65 x the 157 TFLOP/s fp32 MFMA roof.
// Tiling: 128(M) x 128(N) per workgroup, 2 x 2 waves of 64 x 64 (4 accumulator tiles = 64 VGPRs), BK = 16 fp32 = one bf16
// MFMA k-block.  LDS per operand, stage and term: [k-half][128 rows][8 bf16] (a 32-lane half-wave reads 512 contiguous bytes:
// conflict-free ds_read_b128) -> 2 operands x 2 stages x 3 terms x 4 KB = 48 KB, 3 workgroups per CU.  Global loads of tile
// t+1 are in flight during the 24 MFMAs of tile t; the split (VALU) runs in the shadow of the other waves' MFMAs.
// Operand addressing (dense / embedding gather / conv taps), epilogues and the XCD-aware block order are gemm_kernel's.
// ---------------------------------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int G3_BM = 128, G3_BN = 128, G3_BK = 16;
constexpr int G3_HALF = 128 * 8 + 32;                 // one k-half: 128 rows x 8 bf16, + 64 B so the two halves sit 16 banks apart
constexpr int G3_PLANE = 2 * G3_HALF;                 // bf16 elements per (operand, stage, term): [k-half][row][8]

// Split four k-consecutive fp32 values into the three bf16 term planes and store them (8 bytes per plane).  The split
// TRUNCATES (term = top 16 bits of the running residual; the residual stays exact and keeps its sign): three terms still cover
// 24 mantissa bits, and v_perm_b32 packs two terms per instruction -- 5.5 VALU ops per element instead of ~25 for
// round-to-nearest splits, which would make this kernel VALU-bound (measured: 97 TFLOP/s, no better than the f32 MFMA path).
__device__ __forceinline__ void g3_split_store(unsigned short* base, int row, int kq, const float4& v) {
    const unsigned x0 = __float_as_uint(v.x), x1 = __float_as_uint(v.y), x2 = __float_as_uint(v.z), x3 = __float_as_uint(v.w);
    const unsigned a01 = __builtin_amdgcn_perm(x1, x0, 0x07060302u), a23 = __builtin_amdgcn_perm(x3, x2, 0x07060302u);
    const float r0 = v.x - __uint_as_float(x0 & 0xFFFF0000u), r1 = v.y - __uint_as_float(x1 & 0xFFFF0000u);
    const float r2 = v.z - __uint_as_float(x2 & 0xFFFF0000u), r3 = v.w - __uint_as_float(x3 & 0xFFFF0000u);
    const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1), y2 = __float_as_uint(r2), y3 = __float_as_uint(r3);
    const unsigned b01 = __builtin_amdgcn_perm(y1, y0, 0x07060302u), b23 = __builtin_amdgcn_perm(y3, y2, 0x07060302u);
    const float s0 = r0 - __uint_as_float(y0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(y1 & 0xFFFF0000u);
    const float s2 = r2 - __uint_as_float(y2 & 0xFFFF0000u), s3 = r3 - __uint_as_float(y3 & 0xFFFF0000u);
    const unsigned c01 = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    const unsigned c23 = __builtin_amdgcn_perm(__float_as_uint(s3), __float_as_uint(s2), 0x07060302u);
    unsigned short* d = base + (kq >> 1) * G3_HALF + row * 8 + (kq & 1) * 4;
    *reinterpret_cast<uint2*>(d) = make_uint2(a01, a23);
    *reinterpret_cast<uint2*>(d + G3_PLANE) = make_uint2(b01, b23);
    *reinterpret_cast<uint2*>(d + 2 * G3_PLANE) = make_uint2(c01, c23);
}

// Two-term fp16 split (operands bounded by 2^15, e.g. tanh/sigmoid outputs, embeddings, weights): x = h1 + 2^-11 h2' with
// h1 = fp16_rtz(x) (v_cvt_pkrtz_f16_f32 converts and packs two values per instruction; the residual is exact) and
// h2' = fp16(2^11 (x - h1)).  Leading products go to one accumulator set, the two cross terms (scaled by 2^11) to a second one:
// 3 fp16 MFMAs per k-block instead of 6 bf16 ones, ~4 VALU ops per element instead of 5.5, 2 LDS planes instead of 3.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void g3_split_store_h2(unsigned short* base, int row, int kq, const float4& v) {
    const fp16x2_t a01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), a23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    const float r0 = (v.x - (float)a01[0]) * 2048.0f, r1 = (v.y - (float)a01[1]) * 2048.0f;
    const float r2 = (v.z - (float)a23[0]) * 2048.0f, r3 = (v.w - (float)a23[1]) * 2048.0f;
    const fp16x2_t b01 = __builtin_amdgcn_cvt_pkrtz(r0, r1), b23 = __builtin_amdgcn_cvt_pkrtz(r2, r3);
    unsigned short* d = base + (kq >> 1) * G3_HALF + row * 8 + (kq & 1) * 4;
    *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, a23));
    *reinterpret_cast<uint2*>(d + G3_PLANE) = make_uint2(__builtin_bit_cast(unsigned, b01), __builtin_bit_cast(unsigned, b23));
}

// MODE 0: dense A; 1: embedding gather, one row per A row (K <= E); 2: conv taps (E < K <= 3E, one table row per tap).
// The mode is a template parameter and every load is unconditional (row pointers of out-of-range rows are clamped to a valid
// row -- their products land in accumulator rows / columns the epilogue never stores), so the k-loop has no branches: a
// predicated load would become its own basic block with a vmcnt(0) in front of it.  Only the K tail tile masks its operands.
// KS: k-tiles of 16 per pipeline stage (one barrier and one prefetch per KS tiles)
template <int MODE, bool H2, int KS = 1>
__global__ __launch_bounds__(256, 2) void gemm3_kernel(GemmArgs p) {
    constexpr int NTERM = H2 ? 2 : 3;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem3[];
    unsigned short* As = smem3;                        // [2 stages][KS tiles][NTERM terms][G3_PLANE]
    unsigned short* Ws = smem3 + 2 * KS * NTERM * G3_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nb = (p.N + G3_BN - 1) / G3_BN;
    const int64_t mb = (p.M + G3_BM - 1) / G3_BM;
    const int64_t bid = blockIdx.x;
    const int64_t tq = bid >> 3;
    const int64_t mblk = (tq / nb) * 8 + (bid & 7);
    if (mblk >= mb) return;
    const int64_t m0 = mblk * G3_BM;
    const int n0 = (int)(tq % nb) * G3_BN;
    const int lr = tid >> 2, kq = tid & 3;             // thread loads rows lr and lr + 64, k = 4*kq .. +3 of the k-tile

    const float* arow[2];
    const float* arow1[2];
    const float* arow2[2];
    const float* wrow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int64_t m = m0 + lr + 64 * i;
        m = m < p.M ? m : p.M - 1;
        arow1[i] = arow2[i] = nullptr;
        if (MODE == 0) {
            arow[i] = p.a + m * p.lda;
        } else {
            const int64_t ai = (m / p.rows_per_seq) * p.seq_stride + (m % p.rows_per_seq);
            arow[i] = p.table + p.ids[ai] * (int64_t)p.E;
            if (MODE == 2) {   // pointers pre-biased by the tap's k offset: element k of tap s is arow_s[k]
                arow1[i] = p.table + p.ids[ai + 1] * (int64_t)p.E - p.E;
                arow2[i] = p.K > 2 * p.E ? p.table + p.ids[ai + 2] * (int64_t)p.E - 2 * p.E : arow1[i];
            }
        }
        int n = n0 + lr + 64 * i;
        n = n < p.N ? n : p.N - 1;
        wrow[i] = p.w + (int64_t)n * p.ldw;
    }
    // Two register sets: the tile loaded in iteration kt is staged into LDS at the END of iteration kt + 1 (prefetch distance two k-stages).
    // Round 6: with one set (tile kt + 1 requested above the MFMAs of tile kt and staged right behind them) an iteration took ~4 000 cycles
    // against 768 of MFMA issue per wave -- the HBM / L2 round trip of the operand loads, not the matrix pipe, set the pace of every gemm3 launch
    struct TileRegs { float4 a[KS][2], w[KS][2]; };
    TileRegs R0, R1;
    auto load_tile = [&](TileRegs& R, int k00, bool tail) {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            int k = k00 + j * G3_BK + 4 * kq;
            const bool kv = k < p.K;
            k = kv ? k : p.K - 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* src = arow[i];
                if (MODE == 2) src = k < p.E ? arow[i] : (k < 2 * p.E ? arow1[i] : arow2[i]);
                R.a[j][i] = *reinterpret_cast<const float4*>(src + k);
                R.w[j][i] = *reinterpret_cast<const float4*>(wrow[i] + k);
                if (tail && !kv) {
                    R.a[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    R.w[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    };
    auto store_tile = [&](const TileRegs& R, int buf) {
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned short* ad = As + (buf * KS + j) * NTERM * G3_PLANE;
                unsigned short* wd = Ws + (buf * KS + j) * NTERM * G3_PLANE;
                if (H2) {
                    g3_split_store_h2(ad, lr + 64 * i, kq, R.a[j][i]);
                    g3_split_store_h2(wd, lr + 64 * i, kq, R.w[j][i]);
                } else {
                    g3_split_store(ad, lr + 64 * i, kq, R.a[j][i]);
                    g3_split_store(wd, lr + 64 * i, kq, R.w[j][i]);
                }
            }
    };

    f32x16 acc[2][2], acx[H2 ? 2 : 1][H2 ? 2 : 1];     // acx: the 2^11-scaled cross terms of the fp16 split
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[a][b][r] = 0.0f;
                if (H2) acx[a][b][r] = 0.0f;
            }

    const int nk = (p.K + KS * G3_BK - 1) / (KS * G3_BK);          // pipeline stages of KS k-tiles
    const bool ktail = (p.K % (KS * G3_BK)) != 0;
    const int KSTEP = KS * G3_BK;
    load_tile(R0, 0, nk == 1 && ktail);
    store_tile(R0, 0);
    if (nk > 1) load_tile(R1, KSTEP, nk == 2 && ktail);        // tile 1 is in flight while tile 0 is multiplied
    __syncthreads();
    // fragment address inside a plane: [k-half = lane >> 5][row][8]
    const int foff_a = (lane >> 5) * G3_HALF + (wm * 64 + (lane & 31)) * 8;
    const int foff_w = (lane >> 5) * G3_HALF + (wn * 64 + (lane & 31)) * 8;
    auto mma_tile = [&](int buf) {
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const unsigned short* ab = As + (buf * KS + j) * NTERM * G3_PLANE + foff_a;
        const unsigned short* wb = Ws + (buf * KS + j) * NTERM * G3_PLANE + foff_w;
        if constexpr (H2) {
            f16x8 af[2][2], wf[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i][t] = *reinterpret_cast<const f16x8*>(ab + t * G3_PLANE + i * 32 * 8);
                    wf[i][t] = *reinterpret_cast<const f16x8*>(wb + t * G3_PLANE + i * 32 * 8);
                }
#define G3_H2(ACC, TA, TW)                                                                                       \
            ACC[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][TA], wf[0][TW], ACC[0][0], 0, 0, 0);        \
            ACC[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][TA], wf[1][TW], ACC[0][1], 0, 0, 0);        \
            ACC[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][TA], wf[0][TW], ACC[1][0], 0, 0, 0);        \
            ACC[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][TA], wf[1][TW], ACC[1][1], 0, 0, 0);
            G3_H2(acx, 1, 0) G3_H2(acx, 0, 1) G3_H2(acc, 0, 0)
#undef G3_H2
        } else {
            bf16x8 af[2][3], wf[2][3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i][t] = *reinterpret_cast<const bf16x8*>(ab + t * G3_PLANE + i * 32 * 8);
                    wf[i][t] = *reinterpret_cast<const bf16x8*>(wb + t * G3_PLANE + i * 32 * 8);
                }
            }
            // six cross terms, smallest first; the four accumulator tiles are interleaved inside each term so that consecutive
            // MFMAs never depend on each other
#define G3_TERM(TA, TW)                                                                                          \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][TA], wf[0][TW], acc[0][0], 0, 0, 0);       \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][TA], wf[1][TW], acc[0][1], 0, 0, 0);       \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][TA], wf[0][TW], acc[1][0], 0, 0, 0);       \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][TA], wf[1][TW], acc[1][1], 0, 0, 0);
            G3_TERM(2, 0) G3_TERM(1, 1) G3_TERM(0, 2) G3_TERM(1, 0) G3_TERM(0, 1) G3_TERM(0, 0)
#undef G3_TERM
        }
      }
    };
    // iteration kt: request tile kt + 2 into the set tile kt left free, multiply tile kt from LDS, stage tile kt + 1 (requested one iteration ago).
    // Steady state (tile kt + 2 is a full one): no masking code in the loop
    // NIR_G3_NOLOAD / NIR_G3_NOMMA / NIR_G3_NOSTORE: timing ablations of the steady-state loop (instrumented variant builds only: results are wrong)
#ifdef NIR_G3_NOLOAD
#define G3_LOAD(...)
#else
#define G3_LOAD(...) load_tile(__VA_ARGS__)
#endif
#ifdef NIR_G3_NOMMA
#define G3_MMA(b)
#else
#define G3_MMA(b) mma_tile(b)
#endif
#ifdef NIR_G3_NOSTORE
#define G3_STORE(...)
#else
#define G3_STORE(...) store_tile(__VA_ARGS__)
#endif
    int kt = 0;
    for (; kt + 4 < nk; kt += 2) {                     // tiles kt + 2 and kt + 3 are full ones: a masked (tail) tile is waited for right behind its request
        G3_LOAD(R0, (kt + 2) * KSTEP, false);
        __builtin_amdgcn_sched_barrier(0);             // keep the loads ABOVE the MFMAs (the scheduler sinks them to their first use)
        G3_MMA(0);
        __builtin_amdgcn_sched_barrier(0);
        G3_STORE(R1, 1);
        lds_barrier();                                 // LDS only: __syncthreads() also waits for the tile requested two stages ahead (vmcnt(0))
        G3_LOAD(R1, (kt + 3) * KSTEP, false);
        __builtin_amdgcn_sched_barrier(0);
        G3_MMA(1);
        __builtin_amdgcn_sched_barrier(0);
        G3_STORE(R0, 0);
        lds_barrier();                                 // LDS only: __syncthreads() also waits for the tile requested two stages ahead (vmcnt(0))
    }
#undef G3_LOAD
#undef G3_MMA
#undef G3_STORE
    // the last one to four tiles (kt is even: tile kt sits in LDS buffer 0, tile kt + 1 -- if any -- in R1)
    for (; kt < nk; ++kt) {
        const bool odd = (kt & 1) != 0;
        if (kt + 2 < nk) {
            if (odd) load_tile(R1, (kt + 2) * KSTEP, (kt + 2 == nk - 1) && ktail);
            else load_tile(R0, (kt + 2) * KSTEP, (kt + 2 == nk - 1) && ktail);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_tile(kt & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            if (odd) store_tile(R0, 0);
            else store_tile(R1, 1);
            lds_barrier();                                 // LDS only: __syncthreads() also waits for the tile requested two stages ahead (vmcnt(0))
        }
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wn * 64 + b * 32 + (lane & 31);
        float bsum = 0.f;
        if (n < p.N) {
            if (p.bias) bsum += p.bias[n];
            if (p.bias2) bsum += p.bias2[n];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[a][b][r];
                if (H2) v = fmaf(acx[a][b][r], 1.0f / 2048.0f, v);
                gemm_store(p, m, n, v, bsum);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same two-term fp16 GEMM with PRE-SPLIT operands: both operands arrive as fp16 term planes (x1 = fp16_rtz(x),
// x2 = fp16(2^11 (x - x1)), nir_split_f16x2 or written directly by the producing kernel), so staging a tile is four 16-byte
// copies per thread and the k-loop holds no VALU work besides addressing -- the in-kernel split made gemm3 VALU-bound.
//   A planes: dense [M, lda] (MODE 0) or rows of plane TABLES [V, EP] gathered by token id, one row per conv tap (MODE 2, K =
//   taps*EP; EP is the row length padded to a multiple of 8 so that no 16-byte chunk straddles a tap).  W planes: [N, K].
// Tile 128 x 128, BK = 32 (two fp16 k-blocks: 24 MFMAs per wave between barriers), LDS 64 KB -> 2 workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------------
struct GemmPlaneArgs {
    const _Float16 *a1, *a2;      // dense A planes (MODE 0) or plane tables (MODE 2)
    int64_t lda;                  // row stride of the A planes / tables, in elements (multiple of 8)
    const int64_t* ids;
    int64_t rows_per_seq, seq_stride;
    int EP, taps;
    const _Float16 *w1, *w2;      // [N, K] planes, row stride ldw
    int64_t ldw;
    GemmArgs ep;                  // epilogue description (bias, act, add, c, ldc, M, N) -- operand fields unused
    int K;                        // padded K (multiple of 8)
};
constexpr int GP_BK = 32;
constexpr int GP_CHUNK = 128 * 8 + 32;            // one (k-block, k-half) chunk: 128 rows x 8 halves (+64 B bank offset)
constexpr int GP_PLANE = 4 * GP_CHUNK;            // per (operand, stage, term): [k-block 2][k-half 2][row][8]

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_h2p_kernel(GemmPlaneArgs q) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smemp[];
    unsigned short* As = smemp;                        // [2 stages][2 terms][GP_PLANE]
    unsigned short* Ws = smemp + 2 * 2 * GP_PLANE;
    const GemmArgs& p = q.ep;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nb = (p.N + G3_BN - 1) / G3_BN;
    const int64_t mb = (p.M + G3_BM - 1) / G3_BM;
    const int64_t bid = blockIdx.x;
    const int64_t tq = bid >> 3;
    const int64_t mblk = (tq / nb) * 8 + (bid & 7);
    if (mblk >= mb) return;
    const int64_t m0 = mblk * G3_BM;
    const int n0 = (int)(tq % nb) * G3_BN;
    const int lr = tid >> 1, hs = tid & 1;             // thread stages row lr, k-block hs (16 k = two 16-byte chunks) of every plane

    int64_t m = m0 + lr;
    m = m < p.M ? m : p.M - 1;
    int64_t arow[3] = {0, 0, 0};                       // element offsets of the row (per tap) inside the A planes
    if (MODE == 0) {
        arow[0] = m * q.lda;
    } else {
        const int64_t ai = (m / q.rows_per_seq) * q.seq_stride + (m % q.rows_per_seq);
#pragma unroll
        for (int t = 0; t < 3; ++t) arow[t] = (t < q.taps ? q.ids[ai + t] : q.ids[ai]) * q.lda - (int64_t)t * q.EP;   // pre-biased by the tap's k offset
    }
    int n = n0 + lr;
    n = n < p.N ? n : p.N - 1;
    const int64_t wrow = (int64_t)n * q.ldw;

    uint4 ra[2][2], rw[2][2];                          // [term][k-half chunk]
    // `tail` is a literal at every call site: the steady-state loop carries no masking code (a masked overwrite of a register
    // that a load is still filling forces an s_waitcnt vmcnt in front of the MFMA block, exposing the whole load latency)
    auto load_tile = [&](int k0, bool tail) {
        int k = k0 + 16 * hs;
        bool kv = true, kv2 = true;
        if (tail) {
            kv = k < q.K;
            k = kv ? k : 0;                            // K is a multiple of 8 but maybe not of 32: tail chunks are zeroed
            kv2 = kv && k + 8 < q.K;
        }
        int64_t ao = arow[0];
        if (MODE == 2) ao = k < q.EP ? arow[0] : (k < 2 * q.EP ? arow[1] : arow[2]);
        const int k2 = kv2 ? k + 8 : k;
        int64_t ao2 = ao;
        if (MODE == 2) ao2 = k2 < q.EP ? arow[0] : (k2 < 2 * q.EP ? arow[1] : arow[2]);
        ra[0][0] = *reinterpret_cast<const uint4*>(q.a1 + ao + k);
        ra[1][0] = *reinterpret_cast<const uint4*>(q.a2 + ao + k);
        ra[0][1] = *reinterpret_cast<const uint4*>(q.a1 + ao2 + k2);
        ra[1][1] = *reinterpret_cast<const uint4*>(q.a2 + ao2 + k2);
        rw[0][0] = *reinterpret_cast<const uint4*>(q.w1 + wrow + k);
        rw[1][0] = *reinterpret_cast<const uint4*>(q.w2 + wrow + k);
        rw[0][1] = *reinterpret_cast<const uint4*>(q.w1 + wrow + k2);
        rw[1][1] = *reinterpret_cast<const uint4*>(q.w2 + wrow + k2);
        if (tail) {
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            if (!kv) { ra[0][0] = z; ra[1][0] = z; rw[0][0] = z; rw[1][0] = z; }
            if (!kv2) { ra[0][1] = z; ra[1][1] = z; rw[0][1] = z; rw[1][1] = z; }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int off = (buf * 2 + t) * GP_PLANE + (hs * 2 + c) * GP_CHUNK + lr * 8;
                *reinterpret_cast<uint4*>(As + off) = ra[t][c];
                *reinterpret_cast<uint4*>(Ws + off) = rw[t][c];
            }
    };
    f32x16 acc[2][2], acx[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.0f; acx[a][b][r] = 0.0f; }

    const int nk = (q.K + GP_BK - 1) / GP_BK;
    const bool ktail = (q.K % GP_BK) != 0;
    if (nk == 1 && ktail) load_tile(0, true);
    else load_tile(0, false);
    store_tile(0);
    __syncthreads();
    const int foff_a = (lane >> 5) * GP_CHUNK + (wm * 64 + (lane & 31)) * 8;
    const int foff_w = (lane >> 5) * GP_CHUNK + (wn * 64 + (lane & 31)) * 8;
    auto mma_tile = [&](int buf) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const unsigned short* ab = As + buf * 2 * GP_PLANE + kb * 2 * GP_CHUNK + foff_a;
            const unsigned short* wb = Ws + buf * 2 * GP_PLANE + kb * 2 * GP_CHUNK + foff_w;
            f16x8 af[2][2], wf[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i][t] = *reinterpret_cast<const f16x8*>(ab + t * GP_PLANE + i * 32 * 8);
                    wf[i][t] = *reinterpret_cast<const f16x8*>(wb + t * GP_PLANE + i * 32 * 8);
                }
#define GP_T(ACC, TA, TW)                                                                                        \
            ACC[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][TA], wf[0][TW], ACC[0][0], 0, 0, 0);        \
            ACC[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][TA], wf[1][TW], ACC[0][1], 0, 0, 0);        \
            ACC[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][TA], wf[0][TW], ACC[1][0], 0, 0, 0);        \
            ACC[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][TA], wf[1][TW], ACC[1][1], 0, 0, 0);
            GP_T(acx, 1, 0) GP_T(acx, 0, 1) GP_T(acc, 0, 0)
#undef GP_T
        }
    };
    for (int kt = 0; kt + 2 < nk; ++kt) {              // steady state: the next tile is a full one
        load_tile((kt + 1) * GP_BK, false);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile(kt & 1);
        __builtin_amdgcn_sched_barrier(0);
        store_tile((kt & 1) ^ 1);
        __syncthreads();
    }
    if (nk >= 2) {                                     // second-to-last tile prefetches the (possibly partial) last one
        const int kt = nk - 2;
        if (ktail) load_tile((kt + 1) * GP_BK, true);
        else load_tile((kt + 1) * GP_BK, false);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile(kt & 1);
        __builtin_amdgcn_sched_barrier(0);
        store_tile((kt & 1) ^ 1);
        __syncthreads();
    }
    mma_tile((nk - 1) & 1);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int nn = n0 + wn * 64 + b * 32 + (lane & 31);
        float bsum = 0.f;
        if (nn < p.N) {
            if (p.bias) bsum += p.bias[nn];
            if (p.bias2) bsum += p.bias2[nn];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t mm = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                gemm_store(p, mm, nn, fmaf(acx[a][b][r], 1.0f / 2048.0f, acc[a][b][r]), bsum);
            }
    }
}

// x [rows, cols] fp32 (row stride ld) -> two fp16 term planes [rows, cols_pad] (zero padded): p1 = fp16_rtz(x), p2 = fp16(2^11 (x - p1))
__global__ void split_f16x2_kernel(const float* __restrict__ x, int64_t rows, int cols, int64_t ld, int cols_pad, _Float16* __restrict__ p1,
                                   _Float16* __restrict__ p2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols_pad) return;
    const int64_t r = i / cols_pad;
    const int c = (int)(i % cols_pad);
    const float v = c < cols ? x[r * ld + c] : 0.f;
    const fp16x2_t a = __builtin_amdgcn_cvt_pkrtz(v, 0.f);
    p1[i] = (_Float16)a[0];
    p2[i] = (_Float16)((v - (float)a[0]) * 2048.0f);
}

int launch_split_f16x2(const float* x, int64_t rows, int cols, int64_t ld, int cols_pad, void* p1, void* p2, hipStream_t st) {
    NIR_REQUIRE(x && p1 && p2 && rows >= 0 && cols > 0 && cols_pad >= cols && cols_pad % 8 == 0, "split_f16x2: bad args (cols_pad %% 8 == 0)");
    if (rows == 0) return 0;
    const int64_t n = rows * cols_pad;
    hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, rows, cols, ld, cols_pad, (_Float16*)p1, (_Float16*)p2);
    NIR_CHECK_LAUNCH("split_f16x2_kernel");
    return 0;
}

// C = act(A W^T + bias) from pre-split planes.  Dense: a1/a2 [M, lda]; gathered (ids != NULL): a1/a2 are plane tables [V, lda] and row m
// is the concatenation of `taps` table rows (EP elements each).  K = padded reduction length (taps * EP or the padded dense K).
int launch_linear_planes(const void* a1, const void* a2, int64_t lda, const int64_t* ids, int64_t rows_per_seq, int64_t seq_stride, int EP,
                         int taps, const void* w1, const void* w2, int64_t ldw, const float* bias, float* c, int64_t ldc, int64_t M, int N, int K,
                         int act, const float* add, int64_t ldadd, hipStream_t st) {
    NIR_REQUIRE(a1 && a2 && w1 && w2 && c && M >= 0 && N > 0 && K > 0, "linear_planes: bad args");
    NIR_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0, "linear_planes: K and the row strides must be multiples of 8");
    NIR_REQUIRE(!ids || (taps >= 1 && taps <= 3 && EP % 8 == 0 && K == taps * EP), "linear_planes: gathered A needs K == taps*EP, EP %% 8 == 0");
    if (M == 0) return 0;
    GemmPlaneArgs q;
    q.a1 = (const _Float16*)a1; q.a2 = (const _Float16*)a2; q.lda = lda; q.ids = ids; q.rows_per_seq = rows_per_seq; q.seq_stride = seq_stride;
    q.EP = EP; q.taps = taps; q.w1 = (const _Float16*)w1; q.w2 = (const _Float16*)w2; q.ldw = ldw; q.K = K;
    q.ep = GemmArgs{nullptr, 0, nullptr, nullptr, 0, 0, 0, nullptr, 0, bias, nullptr, c, ldc, M, N, K, act, add, ldadd};
    const int64_t mb3 = (M + G3_BM - 1) / G3_BM;
    const int nb3 = (N + G3_BN - 1) / G3_BN;
    constexpr size_t lds = (size_t)2 * 2 * 2 * GP_PLANE * 2;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)gemm_h2p_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gemm_h2p_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    ProfScope ps(prof_shape_name(ids ? "gemm_h2p_kernel[gather]" : "gemm_h2p_kernel", M, N, K), st);
    dim3 grid((unsigned)(8 * nb3 * ((mb3 + 7) / 8)));
    if (ids) hipLaunchKernelGGL(gemm_h2p_kernel<2>, grid, dim3(256), lds, st, q);
    else hipLaunchKernelGGL(gemm_h2p_kernel<0>, grid, dim3(256), lds, st, q);
    NIR_CHECK_LAUNCH("gemm_h2p_kernel");
    return 0;
}

template <int NT>
static void launch_skinny(const GemmArgs& p, int G, size_t lds, hipStream_t st) {
    static std::once_flag once;       // one-time opt-in to > 64 KB of dynamic LDS, race-free
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); });
    const int64_t ntiles = (p.M + 15) / 16;
    const unsigned grid = (unsigned)std::min<int64_t>((ntiles + SK_WAVES - 1) / SK_WAVES, 512);
    hipLaunchKernelGGL(gemm_skinny_kernel<NT>, dim3(grid), dim3(64 * SK_WAVES), lds, st, p, G);
}

// One wave per output row.
__global__ __launch_bounds__(256) void rowdot_kernel(const float* x, int64_t ldx, const float* w, const float* b,
                                                     float* out, int64_t M, int K, int act) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* xr = x + m * ldx;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += xr[k] * w[k];
    s = wave_sum(s);
    if (lane == 0) {
        float v = s + (b ? b[0] : 0.f);
        if (act == NIR_ACT_TANH) v = fast_tanh(v);
        else if (act == NIR_ACT_RELU) v = fmaxf(v, 0.f);
        out[m] = v;
    }
}

int launch_linear_ex(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                     int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                     int64_t ldc, int64_t M, int N, int K, int act, const float* add, int64_t ldadd, hipStream_t st) {
    const bool bounded = (act & ACT_BOUNDED) != 0;
    act &= ~ACT_BOUNDED;
    NIR_REQUIRE(M >= 0 && N > 0 && K > 0, "linear: bad dims M=%lld N=%d K=%d", (long long)M, N, K);
    NIR_REQUIRE(w && c, "linear: null weight/output");
    NIR_REQUIRE(ids ? (table != nullptr && E > 0 && rows_per_seq > 0) : (a != nullptr), "linear: null A operand");
    if (M == 0) return 0;
    NIR_REQUIRE(act != ACT_MAXOUT2 || (N % 2 == 0), "linear: maxout epilogue needs an even N");
    NIR_REQUIRE(act != ACT_TANH_ROWDOT16 || (N % 16 == 0 && add != nullptr), "linear: tanh-rowdot epilogue needs N %% 16 == 0 and the weight row");
    GemmArgs p{a, lda, ids, table, E, rows_per_seq, seq_stride, w, ldw, bias, bias2, c, ldc, M, N, K, act, add, ldadd};
    bool vec = (K % 4 == 0) && (ldw % 4 == 0) && (((uintptr_t)w & 15) == 0);
    if (ids) vec = vec && (E % 4 == 0) && (((uintptr_t)table & 15) == 0);
    else vec = vec && (lda % 4 == 0) && (((uintptr_t)a & 15) == 0);
    const int64_t mb = (M + BM - 1) / BM;
    const int nb = (N + BN - 1) / BN;
    const int skG = (K + 15) / 16, skNT = (N + 15) / 16;
    const size_t sk_lds = (size_t)16 * skNT * (skG * 16 + 4) * 4;
    const bool exact_f32 = tun(g_tun.exact_f32) != 0;                    // force the f32-MFMA kernels everywhere
    const int64_t mb3 = (M + G3_BM - 1) / G3_BM;
    const int nb3 = (N + G3_BN - 1) / G3_BN;
    const int mode3 = !ids ? 0 : (K <= E ? 1 : (K <= 3 * E ? 2 : -1));
    // (>= 96 workgroups of 128 x 128: below that the 64 x 64-tile fp32 kernel wins -- 4480 x 256 x 256, 70 of them, measured 6.9 us per call on
    // the fp16 two-term form against 5.2 us)
    if (vec && !exact_f32 && mode3 >= 0 && N >= 96 && K >= 32 && mb3 * nb3 >= 96) {
        // large GEMMs: fp32 accuracy from three-term bf16 splits on the bf16 matrix cores (2.65 x the f32 MFMA roof)
        ProfScope ps(prof_shape_name(bounded ? (ids ? "gemm3h_kernel[gather]" : "gemm3h_kernel") : (ids ? "gemm3_kernel[gather]" : "gemm3_kernel"), M, N, K), st);
        dim3 grid((unsigned)(8 * nb3 * ((mb3 + 7) / 8)));
        // fp16 two-term form: two k-tiles per pipeline stage (one barrier and one prefetch per 32 of K; 66 KB of LDS: still two workgroups per CU):
        // 8 960 x 512 x 1 024 65.5 -> 57.4 us, the gather-GEMM 573 440 x 1 024 x 300 2.135 -> 2.096 ms (tunable gemm3_ks = 1: one tile per stage)
        if (bounded && tun(g_tun.gemm3_ks) != 1 && K >= 64) {
            constexpr size_t lds2 = (size_t)2 * 2 * 2 * 2 * G3_PLANE * 2;
            static std::once_flag once;
            std::call_once(once, [] {
                (void)hipFuncSetAttribute((const void*)gemm3_kernel<0, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                (void)hipFuncSetAttribute((const void*)gemm3_kernel<1, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            });
            if (mode3 == 0) { hipLaunchKernelGGL((gemm3_kernel<0, true, 2>), grid, dim3(256), lds2, st, p); NIR_CHECK_LAUNCH("gemm3h ks2"); return 0; }
            if (mode3 == 1) { hipLaunchKernelGGL((gemm3_kernel<1, true, 2>), grid, dim3(256), lds2, st, p); NIR_CHECK_LAUNCH("gemm3h ks2"); return 0; }
        }
        if (bounded) {
            constexpr size_t lds = (size_t)2 * 2 * 2 * G3_PLANE * 2;
            if (mode3 == 0) hipLaunchKernelGGL((gemm3_kernel<0, true>), grid, dim3(256), lds, st, p);
            else if (mode3 == 1) hipLaunchKernelGGL((gemm3_kernel<1, true>), grid, dim3(256), lds, st, p);
            else hipLaunchKernelGGL((gemm3_kernel<2, true>), grid, dim3(256), lds, st, p);
        } else {
            constexpr size_t lds = (size_t)2 * 2 * 3 * G3_PLANE * 2;
            if (mode3 == 0) hipLaunchKernelGGL((gemm3_kernel<0, false>), grid, dim3(256), lds, st, p);
            else if (mode3 == 1) hipLaunchKernelGGL((gemm3_kernel<1, false>), grid, dim3(256), lds, st, p);
            else hipLaunchKernelGGL((gemm3_kernel<2, false>), grid, dim3(256), lds, st, p);
        }
        NIR_CHECK_LAUNCH("nir_linear_f32[bf16x3]");
        return 0;
    }
    if (N <= 64 && vec && M >= 4096 && (!ids || K <= E) && sk_lds <= 128 * 1024 && act != ACT_MAXOUT2 && act != ACT_TANH_ROWDOT16 && !tun(g_tun.no_skinny)) {
        ProfScope ps(prof_shape_name(ids ? "gemm_skinny_kernel[gather]" : "gemm_skinny_kernel", M, N, K), st);
        if (skNT == 1) launch_skinny<1>(p, skG, sk_lds, st);
        else if (skNT == 2) launch_skinny<2>(p, skG, sk_lds, st);
        else if (skNT == 3) launch_skinny<3>(p, skG, sk_lds, st);
        else launch_skinny<4>(p, skG, sk_lds, st);
    } else if (mb * nb < 160 && !tun(g_tun.no_gemm16) && !ids && vec && K % 16 == 0 && ((M + 31) / 32) * ((N + 31) / 32) >= 200) {
        // mid-size: 32x32 output blocks still give >= 200 workgroups
        ProfScope ps(prof_shape_name("gemm32_kernel", M, N, K), st);
        // rows per workgroup (16 RA) chosen for the fewest rounds of workgroups over the CUs times the work per workgroup: 1120 x 512 is
        // 288 workgroups of 64 rows (two rounds on 256 CUs, 28 us) but 224 of 80 rows (one round)
        static const int ncu = [] {
            int dev = 0;
            hipDeviceProp_t prop;
            return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                       ? prop.multiProcessorCount : 256;
        }();
        const int64_t ncol = (N + 31) / 32;
        int best = 2;
        int64_t best_cost = INT64_MAX;
        for (int ra = 2; ra <= (K >= 512 ? 6 : 2); ++ra) {     // short K: prologue / reduction dominate, the smallest block wins (1120 x 256 x 256)
            const int64_t wgs = ((M + 16 * ra - 1) / (16 * ra)) * ncol;
            const int64_t cost = ((wgs + ncu - 1) / ncu) * ra;
            if (cost < best_cost) { best_cost = cost; best = ra; }
        }
        const dim3 grid((unsigned)((M + 16 * best - 1) / (16 * best)), (unsigned)ncol);
        switch (best) {
            case 2: hipLaunchKernelGGL(gemm32_kernel<2>, grid, dim3(256), 0, st, p); break;
            case 3: hipLaunchKernelGGL(gemm32_kernel<3>, grid, dim3(256), 0, st, p); break;
            case 4: hipLaunchKernelGGL(gemm32_kernel<4>, grid, dim3(256), 0, st, p); break;
            case 5: hipLaunchKernelGGL(gemm32_kernel<5>, grid, dim3(256), 0, st, p); break;
            default: hipLaunchKernelGGL(gemm32_kernel<6>, grid, dim3(256), 0, st, p); break;
        }
    } else if (mb * nb < 160 && !tun(g_tun.no_gemm16)) {
        // too few 64x64 tiles to fill 256 CUs: one 16x16 tile per workgroup, K split over the waves
        ProfScope ps(prof_shape_name(ids ? "gemm16_kernel[gather]" : "gemm16_kernel", M, N, K), st);
        dim3 grid((unsigned)((M + 15) / 16), (unsigned)((N + 15) / 16));
        if (vec) hipLaunchKernelGGL(gemm16_kernel<true>, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(gemm16_kernel<false>, grid, dim3(256), 0, st, p);
    } else {
        ProfScope ps(prof_shape_name(ids ? "gemm_kernel[gather]" : "gemm_kernel", M, N, K), st);
        constexpr size_t lds = (size_t)2 * (BM + BN) * LDS_LD * 4;   // 36864 B at BK = 32 (the opt-in only matters for BK = 64)
        static std::once_flag once;
        std::call_once(once, [] {
            (void)hipFuncSetAttribute((const void*)gemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        });
        dim3 grid((unsigned)(8 * nb * ((mb + 7) / 8)));
        if (vec) hipLaunchKernelGGL(gemm_kernel<true>, grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL(gemm_kernel<false>, grid, dim3(256), lds, st, p);
    }
    NIR_CHECK_LAUNCH("nir_linear_f32");
    return 0;
}

int launch_linear(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                  int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                  int64_t ldc, int64_t M, int N, int K, int act, hipStream_t st) {
    return launch_linear_ex(a, lda, ids, table, E, rows_per_seq, seq_stride, w, ldw, bias, bias2, c, ldc, M, N, K, act,
                            nullptr, 0, st);
}

int launch_rowdot(const float* x, int64_t ldx, const float* w, const float* b, float* out, int64_t M, int K, int act,
                  hipStream_t st) {
    NIR_REQUIRE(x && w && out && K > 0, "rowdot: bad args");
    if (M == 0) return 0;
    {
        ProfScope ps("rowdot_kernel", st);
        hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, ldx, w, b, out, M, K, act);
    }
    NIR_CHECK_LAUNCH("nir_rowdot_f32");
    return 0;
}

}  // namespace nir

extern "C" int nir_linear_f32(const float* a, int64_t lda, const int64_t* ids, const float* table, int E,
                              int64_t rows_per_seq, int64_t seq_stride, const float* w, int64_t ldw,
                              const float* bias, const float* bias2, float* c, int64_t ldc, int64_t M, int N, int K,
                              int act, nir_stream_t stream) {
    return nir::launch_linear(a, lda, ids, table, E, rows_per_seq, seq_stride, w, ldw, bias, bias2, c, ldc, M, N, K,
                              act, (hipStream_t)stream);
}

extern "C" int nir_rowdot_f32(const float* x, int64_t ldx, const float* w, const float* b, float* out, int64_t M,
                              int K, int act, nir_stream_t stream) {
    return nir::launch_rowdot(x, ldx, w, b, out, M, K, act, (hipStream_t)stream);
}

namespace nir { int launch_split_f16x2(const float* x, int64_t rows, int cols, int64_t ld, int cols_pad, void* p1, void* p2, hipStream_t st); }
extern "C" int nir_split_f16x2(const float* x, int64_t rows, int cols, int64_t ld, int cols_pad, void* p1, void* p2, nir_stream_t stream) {
    return nir::launch_split_f16x2(x, rows, cols, ld, cols_pad, p1, p2, (hipStream_t)stream);
}

extern "C" int nir_linear_planes_f32(const void* a1, const void* a2, int64_t lda, const int64_t* ids, int64_t rows_per_seq, int64_t seq_stride,
                                     int EP, int taps, const void* w1, const void* w2, int64_t ldw, const float* bias, float* c, int64_t ldc,
                                     int64_t M, int N, int K, int act, nir_stream_t stream) {
    return nir::launch_linear_planes(a1, a2, lda, ids, rows_per_seq, seq_stride, EP, taps, w1, w2, ldw, bias, c, ldc, M, N, K, act, nullptr, 0,
                                     (hipStream_t)stream);
}
