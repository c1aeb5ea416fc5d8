// Training-step kernels (neuroir/models/ranker.py:192-230, models/multitask.py:161-223):
// the backward halves of the FLOP-carrying operators plus the train-mode forwards that have to save activations.
//
//   nir_linear_wgrad_f32   dW[n,k] += sum_m dY[m,n] X[m,k]   (X dense or gathered embedding rows)  -- fp32 MFMA, "TN" GEMM
//   nir_colsum_f32         db[n]   += sum_m dY[m,n]
//   nir_transpose_f32      W^T for the data-gradient GEMM dX = dY W (nir_linear_f32 computes A B^T)
//   nir_lstm_train_fwd     LSTM recurrence that also stores the gate activations and cell states of every step
//   nir_lstm_train_bwd     BPTT: pre-activation gate gradients of every step (dW_ih, dW_hh, db, dx follow as GEMMs)
//   nir_embed_f32 / nir_embed_bwd_f32   embedding lookup (+ dropout mask) / scatter-add of row gradients
//   nir_dropout_f32        counter-based Bernoulli mask (splitmix64 of seed ^ index): the mask is an OUTPUT, so a parity
//                          test can replay exactly the same mask through the oracle
//   nir_act_bwd_f32, nir_bce_bwd_f32, nir_softmax_nll_bwd_f32   element-wise backward pieces
#include "common.hpp"
#include <mutex>
#include <algorithm>

namespace nir {
int launch_bilstm_mfma16(const float* gin, const int64_t* lens, const float* whh, const float* h0, const float* c0, float* out, float* hn,
                         float* cn, int64_t M, int T, int H, int ND, hipStream_t st, float* act, float* cst);

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------------
// dW[n,k] += sum_m dY[m,n] * X[m,k].  One wave owns a 32x32 tile of dW and one slice of M; v_mfma_f32_32x32x2_f32 takes
// A[i = n][kk = m] = dY[m0 + (lane >> 5)][n0 + (lane & 31)] and B[kk = m][j = k] = X[m0 + (lane >> 5)][k0 + (lane & 31)] straight
// from global memory (both reads are 128-byte coalesced rows), 8 row pairs in flight; slices are combined with atomicAdd.
// ---------------------------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* dy; int64_t lddy;
    const float* x; int64_t ldx;
    const int64_t* ids; const float* table; int E;     // gathered X: row m = table[ids[m]]
    float* dw; int64_t lddw;
    int64_t M; int N, K;
    int64_t mslice;
    int store;                                         // 1: one slice covers M -- every element is written once, plain stores (dw needs no zero fill)
    float* db;                                         // optional: bias gradient db[n] = sum_m dy[m,n] -- the waves of k-tile 0 hold dy's column values anyway (round 5)
    // row list (round 5): reduction row r reads dY row rows[r] + dyd and X row rows[r] + xd; the number of rows is read on the device (*mcount;
    // M is its upper bound, `slices` the slice count the grid was sized for) -- padded (t >= length) positions of a sequence batch are not visited
    const int32_t* rows; const int32_t* mcount; int64_t dyd, xd; int slices;
    unsigned magic;                                    // ceil(2^32 / period): row % period through one mulhi (rows < 2^31)
    int period, skip;                                  // period > 0: X counts as zero on reduction rows r with r % period == skip (the first / last step of
                                                       // a [M,T] sequence batch has no previous state: the row before / after belongs to its neighbour)
};
__device__ __forceinline__ bool wgrad_skip(const WgradArgs& p, int64_t row) {
    if (p.period <= 0) return false;
    if (p.period == 1) return true;
    const unsigned r = (unsigned)row, q = __umulhi(r, p.magic);          // q in {floor(r / period), + 1}
    int rem = (int)(r - q * (unsigned)p.period);
    if (rem < 0) rem += p.period;
    return rem == p.skip;
}
__device__ __forceinline__ void wgrad_range(const WgradArgs& p, int64_t slice, int64_t& ms, int64_t& me) {
    if (p.mcount) {
        const int64_t Mt = min((int64_t)*p.mcount, p.M);
        const int64_t msl = ((Mt + p.slices - 1) / p.slices + 31) / 32 * 32;
        ms = slice * msl; me = min(Mt, ms + msl);
    } else {
        ms = slice * p.mslice; me = min(p.M, ms + p.mslice);
    }
}

// NB x KB register blocking: a wave owns a (32 NB) x (32 KB) tile of dW -- NB + KB operand values per lane and m for NB KB MFMAs (the 1 x 1
// form loads two values per MFMA and was load-bound: 0.26 of the fp32 pipe on the [71680] x [1024, 300] gradient of the CARS input projection)
// RB row pairs per iteration (2 RB reduction rows): the small gradients of a step (M of a few hundred rows, one slice or a few) are bound by the
// load -> MFMA round trip of each iteration, not by bandwidth -- 1 x 1 blocking takes 32 rows per trip (32 loads in flight per lane)
// SH (1 x 1 only): the four waves of a workgroup take four SLICES of the same tile and add their tiles up in LDS before the atomics -- a tiny dW
// under a long reduction (the [81920] x [20, 18] gradient of MatchTensor's 1x1 convolution: 1 280 slices) otherwise queues a thousand atomics on
// each of its 360 addresses (78 us for 12 MB of operands).
template <int NB, int KB, int RB, bool SH>
__device__ __forceinline__ void wgrad_body(const WgradArgs& p, const int64_t bid) {      // bid = workgroup index inside this gradient's grid
    static_assert(!SH || (NB == 1 && KB == 1), "shared-tile mode: one 32 x 32 tile per wave");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt = (p.N + 32 * NB - 1) / (32 * NB), kt = (p.K + 32 * KB - 1) / (32 * KB);
    const int64_t wid = bid * 4 + wave;
    const int64_t tiles = (int64_t)nt * kt;
    const int64_t slice = SH ? (bid / tiles) * 4 + wave : wid / tiles;
    const int tile = SH ? (int)(bid % tiles) : (int)(wid % tiles);
    const int n0 = (tile / kt) * 32 * NB, k0 = (tile % kt) * 32 * KB;
    int64_t ms, me;
    wgrad_range(p, slice, ms, me);
    if (ms >= me) {
        if (p.store && p.mcount && slice == 0) me = ms = 0;           // (an empty row list in "=" mode still writes its zeros; slice 0 only:
                                                                     // the spare waves of the last workgroup have slice >= slices)
        else if (SH) me = ms = 0;                                    // (stays for the workgroup's reduction with a zero tile)
        else return;
    }
    const int half = lane >> 5;
    int n[NB], k[KB];
    bool nv[NB], kv[KB];
#pragma unroll
    for (int i = 0; i < NB; ++i) { n[i] = n0 + 32 * i + (lane & 31); nv[i] = n[i] < p.N; n[i] = nv[i] ? n[i] : 0; }
#pragma unroll
    for (int j = 0; j < KB; ++j) { k[j] = k0 + 32 * j + (lane & 31); kv[j] = k[j] < p.K; k[j] = kv[j] ? k[j] : 0; }
    f32x16 acc[NB][KB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool do_db = p.db != nullptr && k0 == 0;      // wave-uniform
    float bs[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) bs[i] = 0.f;
    for (int64_t m = ms; m < me; m += 2 * RB) {
        float a[NB][RB], b[KB][RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int64_t mm = m + 2 * u + half;
            const bool mv = mm < me;
            int64_t mc = mv ? mm : ms;                                           // clamped row, masked value: no predicated loads
            if (p.rows) mc = mv ? (int64_t)p.rows[mc] : 0;
            const bool xs = wgrad_skip(p, mc);
            const float* dr = p.dy + (mc + p.dyd) * p.lddy;
            const float* xr = p.ids ? p.table + p.ids[mc] * (int64_t)p.E : p.x + (mc + (xs ? 0 : p.xd)) * p.ldx;
#pragma unroll
            for (int i = 0; i < NB; ++i) a[i][u] = (mv && nv[i]) ? dr[n[i]] : 0.f;
#pragma unroll
            for (int j = 0; j < KB; ++j) b[j][u] = (mv && kv[j] && !xs) ? xr[k[j]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < RB; ++u)
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < KB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][u], b[j][u], acc[i][j], 0, 0, 0);
        if (do_db) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int u = 0; u < RB; u += 4) bs[i] += (a[i][u] + a[i][u + 1]) + (a[i][u + 2] + a[i][u + 3]);
        }
    }
    if constexpr (SH) {
        if (!p.store) {                                  // (store mode = one slice: its wave writes alone, below)
            __shared__ float red[4][17][64];
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[0][0][r];
            red[wave][16][lane] = bs[0];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {                // wave w adds up accumulator registers 4 w .. 4 w + 3 of the four tiles
                const int r = 4 * wave + q;
                const float v = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
                const int nn = n0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (nn < p.N && kv[0]) atomicAdd(p.dw + (int64_t)nn * p.lddw + k[0], v);
            }
            if (wave == 0 && p.db != nullptr && k0 == 0) {
                float t = (red[0][16][lane] + red[1][16][lane]) + (red[2][16][lane] + red[3][16][lane]);
                t += __shfl_xor(t, 32);
                if (half == 0 && nv[0]) atomicAdd(p.db + n[0], t);
            }
            return;
        }
        if (me == ms && !(p.mcount && slice == 0)) return;
    }
    if (do_db) {                                        // lanes l and l + 32 hold the even / odd rows of column n
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const float t = bs[i] + __shfl_xor(bs[i], 32);
            if (half == 0 && nv[i]) {
                if (p.store) p.db[n[i]] = t;
                else atomicAdd(p.db + n[i], t);
            }
        }
    }
    // C/D layout: col = lane & 31 (k), row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5) (n)
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (nn < p.N && kv[j]) {
                    if (p.store) p.dw[(int64_t)nn * p.lddw + k[j]] = acc[i][j][r];
                    else atomicAdd(p.dw + (int64_t)nn * p.lddw + k[j], acc[i][j][r]);
                }
            }
}

template <int NB, int KB, int RB = 8, bool SH = false>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs p) {
    wgrad_body<NB, KB, RB, SH>(p, (int64_t)blockIdx.x);
}

// GROUPED small gradients (round 5): a training step ends with ~35 weight gradients of a few hundred rows, each its own launch of a few dozen
// workgroups walking a dependent load -> MFMA chain (15-40 us apiece at a few percent of the chip, 0.5 ms per CARS step back to back).  One
// launch takes up to WG_GROUP of them: the descriptors travel as kernel arguments (no table in memory: captured by a hipGraph as they are),
// a workgroup finds its gradient from the prefix of workgroup counts and runs the same 1 x 1 body -- all chains in flight together.
constexpr int WG_GROUP = 20;
struct WgradGroup {
    int n;
    int bstart[WG_GROUP + 1];                              // first workgroup of every gradient; bstart[n] = grid size
    int sh[WG_GROUP];                                      // shared-tile mode of the gradient
    WgradArgs a[WG_GROUP];
};
__global__ __launch_bounds__(256) void wgrad_group_kernel(WgradGroup g) {
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.bstart[i + 1]) ++i;             // (uniform)
    const int64_t bid = (int64_t)blockIdx.x - g.bstart[i];
    if (g.sh[i]) wgrad_body<1, 1, 16, true>(g.a[i], bid);
    else wgrad_body<1, 1, 16, false>(g.a[i], bid);
}

// LDS-staged form for the big gradients (round 5).  The register-blocked kernel above has every wave fetch its own 16 x (64 + 64) operand
// block from L2 for 32 MFMAs: at the fp32 MFMA rate that is 16 B / cycle / CU of L2 reads -- the [71680] x [1024, 300] gradient of the CARS
// input projection ran at 0.37 of the pipe on it.  Here a workgroup of WN x WK waves owns a (64 WN) x (64 WK) tile of dW: each 16-row stage of
// dY[.., 64 WN] and X[.., 64 WK] is fetched ONCE (float4, through registers, double-buffered in LDS: the loads of stage s + 1 are in flight
// while stage s feeds the MFMAs) and every wave reads its 64-column halves from LDS -- (WN + WK) / (2 WN WK) of the L2 reads (0.35 at 2 x 5).
// Row stride of the planes = 32 mod 64 words: lanes l and l + 32 (rows 2u, 2u + 1) fall in different banks.
// Wave tile = NBW x KBW blocks of 32 x 32 (2 x 2 by default).  The 128 x 320 tile of the K = 300 gradients is also built from 4 x 2 waves of
// 1 x 5 blocks: EIGHT waves, two per SIMD -- ten waves of 2 x 2 sit 3,3,2,2 on the four SIMDs and the kernel is MFMA-bound on the loaded pair.
template <int WN, int WK, int NBW = 2, int KBW = 2>
__global__ __launch_bounds__(64 * WN * WK) void wgrad_lds_kernel(WgradArgs p) {
    constexpr int BN = 32 * NBW * WN, BK = 32 * KBW * WK, SA = BN + 32, SB = BK + 32, NT = 64 * WN * WK;
    constexpr int ITEMS = 4 * (BN + BK), LA = (ITEMS + NT - 1) / NT;     // float4 items of one 16-row stage
    extern __shared__ float wg_lds[];
    float* sA = wg_lds;
    float* sB = wg_lds + 2 * 16 * SA;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wn = wave / WK, wk = wave % WK;
    const int nt = (p.N + BN - 1) / BN, kt = (p.K + BK - 1) / BK;
    const int tiles = nt * kt;
    const int64_t slice = blockIdx.x / tiles;
    const int tile = (int)(blockIdx.x % tiles);
    const int n0 = (tile / kt) * BN, k0 = (tile % kt) * BK;
    int64_t ms, me;
    wgrad_range(p, slice, ms, me);
    if (ms >= me) {                                                // block-uniform
        if (p.store && p.mcount && slice == 0) me = ms = 0;
        else return;
    }
    const int half = lane >> 5;
    float4 pre[LA];
    auto gload = [&](int64_t m0) {
#pragma unroll
        for (int q = 0; q < LA; ++q) {
            const int idx = (int)threadIdx.x + q * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < 4 * BN) {
                const int row = idx / (BN / 4), c = (idx % (BN / 4)) * 4;
                const int64_t mm = m0 + row;
                if (mm < me && n0 + c < p.N) {
                    const int64_t rr = p.rows ? (int64_t)p.rows[mm] : mm;
                    v = *(const float4*)(p.dy + (rr + p.dyd) * p.lddy + n0 + c);
                }
            } else if (idx < ITEMS) {
                const int j = idx - 4 * BN;
                const int row = j / (BK / 4), c = (j % (BK / 4)) * 4;
                const int64_t mm = m0 + row;
                if (mm < me && k0 + c < p.K) {
                    const int64_t rr = p.rows ? (int64_t)p.rows[mm] : mm;
                    if (!wgrad_skip(p, rr)) {
                        const float* xr = p.ids ? p.table + p.ids[rr] * (int64_t)p.E : p.x + (rr + p.xd) * p.ldx;
                        v = *(const float4*)(xr + k0 + c);
                    }
                }
            }
            pre[q] = v;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < LA; ++q) {
            const int idx = (int)threadIdx.x + q * NT;
            if (idx < 4 * BN) {
                const int row = idx / (BN / 4), c = (idx % (BN / 4)) * 4;
                *(float4*)(sA + (buf * 16 + row) * SA + c) = pre[q];
            } else if (idx < ITEMS) {
                const int j = idx - 4 * BN;
                const int row = j / (BK / 4), c = (j % (BK / 4)) * 4;
                *(float4*)(sB + (buf * 16 + row) * SB + c) = pre[q];
            }
        }
    };
    f32x16 acc[NBW][KBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int j = 0; j < KBW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool do_db = p.db != nullptr && k0 == 0 && wk == 0;      // wave-uniform
    float bs[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) bs[i] = 0.f;
    gload(ms);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int64_t m = ms; m < me; m += 16, buf ^= 1) {
        const bool more = m + 16 < me;                             // block-uniform
        if (more) gload(m + 16);
        const float* ap = sA + (buf * 16 + half) * SA + wn * 32 * NBW + (lane & 31);
        const float* bp = sB + (buf * 16 + half) * SB + wk * 32 * KBW + (lane & 31);
        float a[NBW][8], b[KBW][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < NBW; ++i) a[i][u] = ap[2 * u * SA + 32 * i];
#pragma unroll
            for (int j = 0; j < KBW; ++j) b[j][u] = bp[2 * u * SB + 32 * j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NBW; ++i)
#pragma unroll
                for (int j = 0; j < KBW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][u], b[j][u], acc[i][j], 0, 0, 0);
        if (do_db) {
#pragma unroll
            for (int i = 0; i < NBW; ++i) bs[i] += ((a[i][0] + a[i][1]) + (a[i][2] + a[i][3])) + ((a[i][4] + a[i][5]) + (a[i][6] + a[i][7]));
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
    }
    if (do_db) {
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const float t = bs[i] + __shfl_xor(bs[i], 32);
            const int nn = n0 + wn * 32 * NBW + 32 * i + (lane & 31);
            if (half == 0 && nn < p.N) {
                if (p.store) p.db[nn] = t;
                else atomicAdd(p.db + nn, t);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int j = 0; j < KBW; ++j) {
            const int kk = k0 + wk * 32 * KBW + 32 * j + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + wn * 32 * NBW + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (nn < p.N && kk < p.K) {
                    if (p.store) p.dw[(int64_t)nn * p.lddw + kk] = acc[i][j][r];
                    else atomicAdd(p.dw + (int64_t)nn * p.lddw + kk, acc[i][j][r]);
                }
            }
        }
}

template <int WN, int WK, int NBW = 2, int KBW = 2>
static void wgrad_lds_launch(const WgradArgs& a, int64_t blocks, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * 16 * (32 * NBW * WN + 32 + 32 * KBW * WK + 32) * sizeof(float);
    if (lds > 64 * 1024) {
        static std::once_flag once;
        std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)wgrad_lds_kernel<WN, WK, NBW, KBW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    }
    hipLaunchKernelGGL((wgrad_lds_kernel<WN, WK, NBW, KBW>), dim3((unsigned)blocks), dim3(64 * WN * WK), lds, st, a);
}

// Row list of a padded sequence batch: rows = { m T + t : t0 <= t < len[m] } in (m, t) order, offs[m] = its start, offs[M] = the count.
// The weight gradients of the recurrent encoders reduce over these rows only (the reference packs its sequences: layers.py:52-66).
__global__ __launch_bounds__(1024) void seq_rows_scan_kernel(const int64_t* __restrict__ lens, int64_t M, int T, int t0, int32_t* __restrict__ offs) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < M; base += 1024) {
        const int64_t m = base + threadIdx.x;
        int v = 0;
        if (m < M) {
            const int64_t l = lens[m];
            v = (int)max((int64_t)0, min((int64_t)T, l) - t0);
        }
        int x = v;                                                 // inclusive wave scan
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int pre = carry;
        for (int w = 0; w < wave; ++w) pre += wsum[w];
        if (m < M) offs[m] = pre + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = pre + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) offs[M] = carry;
}
__global__ __launch_bounds__(256) void seq_rows_fill_kernel(const int64_t* __restrict__ lens, const int32_t* __restrict__ offs, int64_t M, int T, int t0,
                                                           int32_t* __restrict__ rows) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * T) return;
    const int64_t m = i / T;
    const int t = (int)(i - m * T);
    if (t >= t0 && t < lens[m]) rows[offs[m] + (t - t0)] = (int32_t)i;
}

// out[n] += sum_m x[m*ld + n]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ld, int64_t M, int N, float* __restrict__ out,
                                                     int64_t mslice, int store) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int64_t ms = (int64_t)blockIdx.y * mslice, me = min(M, ms + mslice);
    if (n >= N) return;
    float s = 0.f;
    for (int64_t m = ms; m < me; ++m) s += x[m * ld + n];
    if (store) out[n] = s;
    else atomicAdd(out + n, s);
}

__global__ void transpose_kernel(const float* __restrict__ in, int R, int Cc, float* __restrict__ out) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        t[i][threadIdx.x] = (r < R && c < Cc) ? in[(int64_t)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (c < Cc && r < R) out[(int64_t)c * R + r] = t[threadIdx.x][i];
    }
}

// up to TG_GROUP transposes in one launch (descriptors as kernel arguments, like wgrad_group_kernel): the data-gradient GEMM of every
// nn.Linear of a step needs its weight transposed, ~30 launches of a few tiles each
constexpr int TG_GROUP = 48;
struct TransposeGroup {
    int n;
    int bstart[TG_GROUP + 1];
    const float* in[TG_GROUP];
    float* out[TG_GROUP];
    int R[TG_GROUP], Cc[TG_GROUP];
};
__global__ void transpose_group_kernel(TransposeGroup g) {
    __shared__ float t[32][33];
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.bstart[i + 1]) ++i;
    const int b = (int)blockIdx.x - g.bstart[i];
    const int R = g.R[i], Cc = g.Cc[i], tx = (Cc + 31) / 32;
    const float* __restrict__ in = g.in[i];
    float* __restrict__ out = g.out[i];
    const int c0 = (b % tx) * 32, r0 = (b / tx) * 32;
    for (int q = threadIdx.y; q < 32; q += 8) {
        const int r = r0 + q, c = c0 + threadIdx.x;
        t[q][threadIdx.x] = (r < R && c < Cc) ? in[(int64_t)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int q = threadIdx.y; q < 32; q += 8) {
        const int c = c0 + q, r = r0 + threadIdx.x;
        if (c < Cc && r < R) out[(int64_t)c * R + r] = t[threadIdx.x][q];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Train-mode LSTM recurrence.  One workgroup = SQ sequences of one direction, one thread per gate row j (4H <= 512 threads):
// its W_hh row lives in registers (H <= 128), h_{t-1} of the SQ sequences in LDS.  Saves act[m,t,dir,4H] (i,f,g,o after
// their non-linearities) and cst[m,t,dir,H] (c_t) for the backward pass; out is zero at t >= length (pack/unpack semantics).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TSQ = 4;
struct LstmTrainArgs {
    const float* gin;      // [M,T,ND*4H]
    const int64_t* lens;
    const float* whh;      // [ND,4H,H]
    float* out;            // [M,T,ND*H]
    float* act;            // [M,T,ND,4H]
    float* cst;            // [M,T,ND,H]
    float* hn;             // [ND,M,H] or NULL
    float* cn;
    const float* h0;       // [ND,M,H] or NULL
    const float* c0;
    int64_t M;
    int T, H, ND;
};

template <int HP>
__global__ __launch_bounds__(512) void lstm_train_fwd_kernel(LstmTrainArgs p) {
    extern __shared__ float sm[];
    float* hs = sm;                       // [TSQ][H]
    float* gs = sm + TSQ * p.H;           // [TSQ][4H]
    const int j = threadIdx.x, H = p.H, H4 = 4 * H, T = p.T;
    const int dir = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * TSQ;
    const bool jv = j < H4;
    float w[HP];
#pragma unroll
    for (int k = 0; k < HP; ++k) w[k] = (jv && k < H) ? p.whh[((int64_t)dir * H4 + j) * H + k] : 0.f;
    int len[TSQ];
    int tmax = 0;
#pragma unroll
    for (int s = 0; s < TSQ; ++s) {
        int l = 0;
        if (m0 + s < p.M) {
            l = p.lens ? (int)p.lens[m0 + s] : T;
            l = l < 0 ? 0 : (l > T ? T : l);
        }
        len[s] = l;
        tmax = max(tmax, l);
    }
    float c[TSQ];
#pragma unroll
    for (int s = 0; s < TSQ; ++s) {
        c[s] = 0.f;
        if (j < H) {
            const bool v = m0 + s < p.M;
            const int64_t si = ((int64_t)dir * p.M + m0 + s) * H + j;
            hs[s * H + j] = (v && p.h0) ? p.h0[si] : 0.f;
            if (v && p.c0) c[s] = p.c0[si];
        }
    }
    __syncthreads();
    const int gate = jv ? j / H : 0;
    for (int step = 0; step < tmax; ++step) {
#pragma unroll
        for (int s = 0; s < TSQ; ++s) {
            if (step < len[s] && jv) {
                const int t = dir == 0 ? step : len[s] - 1 - step;
                float a = p.gin[((m0 + s) * T + t) * (int64_t)(p.ND * H4) + dir * H4 + j];
                const float* hv = hs + s * H;
#pragma unroll
                for (int k = 0; k < HP; ++k)
                    if (k < H) a = fmaf(w[k], hv[k], a);
                a = gate == 2 ? tanhf(a) : 1.0f / (1.0f + expf(-a));
                gs[s * H4 + j] = a;
                p.act[(((m0 + s) * T + t) * p.ND + dir) * (int64_t)H4 + j] = a;
            }
        }
        __syncthreads();
        if (j < H) {
#pragma unroll
            for (int s = 0; s < TSQ; ++s) {
                if (step < len[s]) {
                    const int t = dir == 0 ? step : len[s] - 1 - step;
                    const float* g = gs + s * H4;
                    c[s] = g[H + j] * c[s] + g[j] * g[2 * H + j];
                    const float h = g[3 * H + j] * tanhf(c[s]);
                    hs[s * H + j] = h;
                    p.out[((m0 + s) * T + t) * (int64_t)(p.ND * H) + dir * H + j] = h;
                    p.cst[(((m0 + s) * T + t) * p.ND + dir) * (int64_t)H + j] = c[s];
                }
            }
        }
        __syncthreads();
    }
    if (j < H) {
#pragma unroll
        for (int s = 0; s < TSQ; ++s) {
            if (m0 + s < p.M) {
                for (int t = len[s]; t < T; ++t) p.out[((m0 + s) * T + t) * (int64_t)(p.ND * H) + dir * H + j] = 0.f;
                const int64_t si = ((int64_t)dir * p.M + m0 + s) * H + j;
                if (p.hn) p.hn[si] = hs[s * H + j];
                if (p.cn) p.cn[si] = c[s];
            }
        }
    }
}

// BPTT.  dout [M,T,ND*H] (gradient of the memory bank), dhn/dcn [ND,M,H] (gradient of the final state, may be NULL) ->
// dgates [M,T,ND*4H] = gradient w.r.t. the gate PRE-activations (zero at t >= length), dh0/dc0 [ND,M,H] (may be NULL).
// One thread per gate row j.  dh_{t-1} = dgates_t W_hh: thread (q = j / H, k = j % H) accumulates the rows [q*H, (q+1)*H) of
// column k (coalesced reads of W_hh rows from L2), the four partial sums meet in LDS.
struct LstmBwdArgs {
    const float* dout; const float* dhn; const float* dcn;
    const float* dcst;     // [M,T,ND,H] gradient flowing into the stored cell states (decoder-initialisation path), or NULL
    const float* act; const float* cst; const float* c0;
    const int64_t* lens; const float* whh;
    float* dgates; float* dh0; float* dc0;
    int64_t M; int T, H, ND;
};

__global__ __launch_bounds__(512) void lstm_train_bwd_kernel(LstmBwdArgs p) {
    extern __shared__ float sm[];
    const int H = p.H, H4 = 4 * H, T = p.T;
    float* dg = sm;                       // [TSQ][4H] gate-preactivation grads of the current step
    float* dhr = dg + TSQ * H4;           // [TSQ][H]  recurrent dh
    float* part = dhr + TSQ * H;          // [4][TSQ][H]
    const int j = threadIdx.x;
    const int dir = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * TSQ;
    const bool jv = j < H4;
    const int q = jv ? j / H : 0, k = jv ? j % H : 0;
    int len[TSQ];
    int tmax = 0;
#pragma unroll
    for (int s = 0; s < TSQ; ++s) {
        int l = 0;
        if (m0 + s < p.M) {
            l = p.lens ? (int)p.lens[m0 + s] : T;
            l = l < 0 ? 0 : (l > T ? T : l);
        }
        len[s] = l;
        tmax = max(tmax, l);
    }
    float dc[TSQ];
#pragma unroll
    for (int s = 0; s < TSQ; ++s) {
        dc[s] = 0.f;
        if (j < H) {
            const bool v = m0 + s < p.M;
            const int64_t si = ((int64_t)dir * p.M + m0 + s) * H + j;
            dhr[s * H + j] = (v && p.dhn) ? p.dhn[si] : 0.f;
            if (v && p.dcn) dc[s] = p.dcn[si];
        }
    }
    __syncthreads();
    for (int step = tmax - 1; step >= 0; --step) {
        if (j < H) {
#pragma unroll
            for (int s = 0; s < TSQ; ++s) {
                float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f;
                if (step < len[s]) {
                    const int t = dir == 0 ? step : len[s] - 1 - step;
                    const int64_t row = (m0 + s) * T + t;
                    const float* a = p.act + (row * p.ND + dir) * (int64_t)H4;
                    const float i_ = a[j], f_ = a[H + j], g_ = a[2 * H + j], o_ = a[3 * H + j];
                    const float ct = p.cst[(row * p.ND + dir) * (int64_t)H + j];
                    float cprev = 0.f;
                    if (step > 0) {
                        const int tp = dir == 0 ? step - 1 : len[s] - step;
                        cprev = p.cst[(((m0 + s) * T + tp) * p.ND + dir) * (int64_t)H + j];
                    } else if (p.c0) {
                        cprev = p.c0[((int64_t)dir * p.M + m0 + s) * H + j];
                    }
                    const float th = tanhf(ct);
                    const float dh = p.dout[row * (int64_t)(p.ND * H) + dir * H + j] + dhr[s * H + j];
                    go = dh * th * o_ * (1.f - o_);
                    float dct = dc[s] + dh * o_ * (1.f - th * th);
                    if (p.dcst) dct += p.dcst[(row * p.ND + dir) * (int64_t)H + j];
                    gi = dct * g_ * i_ * (1.f - i_);
                    gf = dct * cprev * f_ * (1.f - f_);
                    gg = dct * i_ * (1.f - g_ * g_);
                    dc[s] = dct * f_;
                    float* o = p.dgates + row * (int64_t)(p.ND * H4) + dir * H4;
                    o[j] = gi; o[H + j] = gf; o[2 * H + j] = gg; o[3 * H + j] = go;
                }
                dg[s * H4 + j] = gi; dg[s * H4 + H + j] = gf; dg[s * H4 + 2 * H + j] = gg; dg[s * H4 + 3 * H + j] = go;
            }
        }
        __syncthreads();
        if (jv) {   // dh_{t-1}[s][k] = sum_jj dg[s][jj] * whh[jj][k]; this thread sums jj in [q*H, (q+1)*H)
            float a[TSQ];
#pragma unroll
            for (int s = 0; s < TSQ; ++s) a[s] = 0.f;
            const float* wp = p.whh + ((int64_t)dir * H4 + q * H) * H + k;
            for (int jj = 0; jj < H; ++jj) {
                const float wv = wp[(int64_t)jj * H];
#pragma unroll
                for (int s = 0; s < TSQ; ++s) a[s] = fmaf(dg[s * H4 + q * H + jj], wv, a[s]);
            }
#pragma unroll
            for (int s = 0; s < TSQ; ++s) part[(q * TSQ + s) * H + k] = a[s];
        }
        __syncthreads();
        if (j < H) {
#pragma unroll
            for (int s = 0; s < TSQ; ++s) {
                if (step < len[s])    // sequences that have not started yet (step >= len) keep the final-state gradient
                    dhr[s * H + j] = (part[(0 * TSQ + s) * H + j] + part[(1 * TSQ + s) * H + j]) + (part[(2 * TSQ + s) * H + j] + part[(3 * TSQ + s) * H + j]);
            }
        }
        __syncthreads();
    }
    if (j < H) {
#pragma unroll
        for (int s = 0; s < TSQ; ++s) {
            if (m0 + s < p.M) {
                const int64_t si = ((int64_t)dir * p.M + m0 + s) * H + j;
                if (p.dh0) p.dh0[si] = dhr[s * H + j];
                if (p.dc0) p.dc0[si] = dc[s];
                for (int t = len[s]; t < T; ++t) {
                    float* o = p.dgates + ((m0 + s) * T + t) * (int64_t)(p.ND * H4) + dir * H4;
                    o[j] = 0.f; o[H + j] = 0.f; o[2 * H + j] = 0.f; o[3 * H + j] = 0.f;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// BPTT on the matrix cores with W_hh RESIDENT (the kernel above re-reads W_hh from L2 in every step and multiplies on the VALU:
// 1.75 ms at the C3 document shape, 6.8 % of the fp32-MFMA roof).  16 sequences x one direction per workgroup, 4 waves (one per SIMD,
// up to 512 registers each):
//     dh_{t-1}[unit, seq] = sum_jj W_hh[jj, unit] * dg_t[seq, jj]      as   D[16 units x 16 seqs] += A[16 units x 4] B[4 x 16 seqs]
// on v_mfma_f32_16x16x4_f32 (exact fp32: gradients span too many binades for the fp16 split).  Wave w owns the unit tiles 2w, 2w+1
// and keeps their columns of W_hh as A fragments in registers for all T steps (the reduction index runs over (gate, unit) = 4 x HP
// values, HP = H rounded up to 4: NKS = HP MFMA k-steps per tile and step, fragment lane (unit = lane & 15, q = lane >> 4) holds
// W_hh[g*H + 4c + q][unit] for k-step (g, c)).  The C/D layout hands a lane (seq = lane & 15, units 4*(lane >> 4) + r of each tile):
// the lane that receives dh_{t-1} of a cell is the lane that computes that cell's gate gradients in the next step -- dh never leaves
// registers.  Per step: gate gradients of the lane's 8 cells (loads prefetched one step ahead) -> dgates to HBM and, as the B operand, to
// LDS [seq][q][k-step] (fp32, 16 B reads feed four k-steps) -> barrier -> NKS MFMAs per tile -> next step.  Two LDS buffers, one barrier
// per step.
// ---------------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NTW = unit tiles per wave: 2 -> 4 waves (one per SIMD, W_hh fragments fill half the register file); 1 -> 8 waves, two per SIMD with 256
// registers each (round 5: with 256 of 512 registers in fragments the prefetched cell inputs of the 4-wave form aliased the registers the
// gate-gradient stores read from, and the compiler waited for the stores before issuing the prefetch -- 2.8 us of every 8.9 us step)
template <int NKS, int NTW = 2>
__global__ __launch_bounds__(512 / NTW, 1) void lstm_train_bwd_mfma_kernel(LstmBwdArgs p) {
    constexpr int SEQ = 16, NTH = 512 / NTW, NWV = 8 / NTW;
    constexpr int HP = NKS;                                // padded hidden size (k-steps per gate x 4 ... = 4 * HP / 4)
    constexpr int RS = NKS + 4;                            // LDS row stride in floats: with NKS (a multiple of 64 banks at H = 128) every lane of a
                                                           // ds_read_b128 hit the same four banks -- 7 us of the 17 us step
    extern __shared__ __attribute__((aligned(16))) float smb[];
    float* dgs = smb;                                      // [2][SEQ][4][RS]
    int* lens_s = reinterpret_cast<int*>(smb + 2 * SEQ * 4 * RS);
    const int H = p.H, H4 = 4 * H, T = p.T, ND = p.ND;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sq = lane & 15, pq = lane >> 4;
    const int dir = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * SEQ;
    const bool sv = m0 + sq < p.M;
    if (tid < SEQ) {
        int l = 0;
        if (m0 + tid < p.M) {
            l = p.lens ? (int)p.lens[m0 + tid] : T;
            l = l < 0 ? 0 : (l > T ? T : l);
        }
        lens_s[tid] = l;
    }
    for (int e = tid; e < 2 * SEQ * 4 * RS; e += NTH) dgs[e] = 0.f;
    __syncthreads();
    int tmax = 0;
#pragma unroll
    for (int s2 = 0; s2 < SEQ; ++s2) tmax = max(tmax, lens_s[s2]);
    const int len = lens_s[sq];
    // A fragments: tile tau = NTW*wave + i, output unit = 16*tau + (lane & 15); k-step ks = g*(HP/4) + c reads W_hh[g*H + 4c + pq][unit]
    float afr[NTW][NKS];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int uo = 16 * (NTW * wave + i) + sq;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int g = ks / (HP / 4), c = ks % (HP / 4), ui = 4 * c + pq;
            // unconditional load from a clamped index, masked by a multiply: a select lets the compiler predicate the load, and every
            // predicated load became its own exec-masked block with an s_waitcnt vmcnt(0) behind it (256 serialised round trips)
            afr[i][ks] = p.whh[((int64_t)dir * H4 + (int64_t)g * H + (ui < H ? ui : H - 1)) * H + (uo < H ? uo : H - 1)] * ((uo < H && ui < H) ? 1.f : 0.f);
        }
    }
    // the lane's cells: sequence sq, units ub[i] + r (r = 0..3) of its two tiles
    int ub[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) ub[i] = 16 * (NTW * wave + i) + 4 * pq;
    float dh[NTW][4], dc[NTW][4];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = ub[i] + r;
            const int64_t si = ((int64_t)dir * p.M + m0 + sq) * H + u;
            dh[i][r] = (sv && u < H && p.dhn) ? p.dhn[si] : 0.f;
            dc[i][r] = (sv && u < H && p.dcn) ? p.dcn[si] : 0.f;
        }
    // the cell inputs of a step are requested a full step ahead (under the previous step's MFMAs), unconditionally from clamped addresses:
    // per-cell conditional loads serialised eight memory round trips per step (18.8 us per step measured)
    struct CellIn { float a[NTW][4][4]; float ct[NTW][4], cp[NTW][4], dy[NTW][4], dcs[NTW][4]; };
    const bool vec = (H & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.act) | reinterpret_cast<uintptr_t>(p.cst) | reinterpret_cast<uintptr_t>(p.dout) |
                                       reinterpret_cast<uintptr_t>(p.dgates) | reinterpret_cast<uintptr_t>(p.dcst) | reinterpret_cast<uintptr_t>(p.c0)) & 15) == 0;
    // H % 4 == 2 (MatchTensor's 70): 8-byte pieces instead -- two per row and tile
    const bool vec2 = !vec && (H & 1) == 0 && ((reinterpret_cast<uintptr_t>(p.act) | reinterpret_cast<uintptr_t>(p.cst) | reinterpret_cast<uintptr_t>(p.dout) |
                                                 reinterpret_cast<uintptr_t>(p.dgates) | reinterpret_cast<uintptr_t>(p.dcst) | reinterpret_cast<uintptr_t>(p.c0)) & 7) == 0;
    auto load_cells = [&](int step, CellIn& ci) {
        const bool on_ = step >= 0 && step < len;
        const int st_ = on_ ? step : 0;
        const int t_ = dir == 0 ? st_ : (len > 0 ? len - 1 - st_ : 0);
        const int64_t row_ = (sv ? m0 + sq : m0) * T + t_;
        const float* a = p.act + (row_ * ND + dir) * (int64_t)H4;
        const float* cs = p.cst + (row_ * ND + dir) * (int64_t)H;
        const int tp = dir == 0 ? st_ - 1 : len - st_;                  // position of the previous step's cell state (step > 0)
        const float* cpv = (st_ > 0) ? p.cst + (((sv ? m0 + sq : m0) * T + (tp < 0 ? 0 : (tp >= T ? T - 1 : tp))) * ND + dir) * (int64_t)H
                                     : (p.c0 ? p.c0 + ((int64_t)dir * p.M + (sv ? m0 + sq : m0)) * H : nullptr);
        const float* dyp = p.dout + row_ * (int64_t)(ND * H) + dir * H;
        const float* dcp = p.dcst ? p.dcst + (row_ * ND + dir) * (int64_t)H : nullptr;
        if (vec) {                                       // H % 4 == 0: the lane's four units are one 16-byte piece of every row
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int u = ub[i] < H ? ub[i] : 0;
                auto ld4 = [&](const float* q, float (&dst)[4]) {
                    const float4 v = *reinterpret_cast<const float4*>(q + u);
                    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                };
#pragma unroll
                for (int g = 0; g < 4; ++g) ld4(a + g * H, ci.a[i][g]);
                ld4(cs, ci.ct[i]);
                ld4(dyp, ci.dy[i]);
                if (cpv) ld4(cpv, ci.cp[i]);
                else ci.cp[i][0] = ci.cp[i][1] = ci.cp[i][2] = ci.cp[i][3] = 0.f;
                if (dcp) ld4(dcp, ci.dcs[i]);
                else ci.dcs[i][0] = ci.dcs[i][1] = ci.dcs[i][2] = ci.dcs[i][3] = 0.f;
            }
            return;
        }
        if (vec2) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int u0 = ub[i] < H ? ub[i] : 0, u1 = ub[i] + 2 < H ? ub[i] + 2 : u0;      // units past H: any valid address (masked below)
                auto ld22 = [&](const float* q, float (&dst)[4]) {
                    const float2 v0 = *reinterpret_cast<const float2*>(q + u0), v1 = *reinterpret_cast<const float2*>(q + u1);
                    dst[0] = v0.x; dst[1] = v0.y; dst[2] = v1.x; dst[3] = v1.y;
                };
#pragma unroll
                for (int g = 0; g < 4; ++g) ld22(a + g * H, ci.a[i][g]);
                ld22(cs, ci.ct[i]);
                ld22(dyp, ci.dy[i]);
                if (cpv) ld22(cpv, ci.cp[i]);
                else ci.cp[i][0] = ci.cp[i][1] = ci.cp[i][2] = ci.cp[i][3] = 0.f;
                if (dcp) ld22(dcp, ci.dcs[i]);
                else ci.dcs[i][0] = ci.dcs[i][1] = ci.dcs[i][2] = ci.dcs[i][3] = 0.f;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = ub[i] + r < H ? ub[i] + r : H - 1;
#pragma unroll
                for (int g = 0; g < 4; ++g) ci.a[i][g][r] = a[g * H + u];
                ci.ct[i][r] = cs[u];
                ci.cp[i][r] = cpv ? cpv[u] : 0.f;
                ci.dy[i][r] = dyp[u];
                ci.dcs[i][r] = dcp ? dcp[u] : 0.f;
            }
    };
    CellIn cin;
    load_cells(tmax - 1, cin);
    for (int step = tmax - 1; step >= 0; --step) {
        float* dgw = dgs + (step & 1) * SEQ * 4 * RS;
        const bool on = step < len;                                    // this sequence takes part in the step
        const int t = dir == 0 ? step : len - 1 - step;
        const int64_t row = (m0 + sq) * T + (on ? t : 0);
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            float gv[4][4];                                  // [gate][r] of this tile
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = ub[i] + r;
                const bool cv = on && sv && u < H;
                const float i_ = cin.a[i][0][r], f_ = cin.a[i][1][r], g_ = cin.a[i][2][r], o_ = cin.a[i][3][r];
                const float th = tanhf(cin.ct[i][r]);
                const float dht = cin.dy[i][r] + dh[i][r];
                const float dct = dc[i][r] + dht * o_ * (1.f - th * th) + cin.dcs[i][r];
                const float gi = cv ? dct * g_ * i_ * (1.f - i_) : 0.f;
                const float gf = cv ? dct * cin.cp[i][r] * f_ * (1.f - f_) : 0.f;
                const float gg = cv ? dct * i_ * (1.f - g_ * g_) : 0.f;
                const float go = cv ? dht * th * o_ * (1.f - o_) : 0.f;
                gv[0][r] = gi; gv[1][r] = gf; gv[2][r] = gg; gv[3][r] = go;
                if (cv) {
                    dc[i][r] = dct * f_;
// NIR_BW_NOSTORE / NIR_BW_NOLOAD / NIR_BW_NOMFMA: timing ablations of the step (instrumented variant builds only, results are wrong with any set)
#ifndef NIR_BW_NOSTORE
                    if (!vec && !vec2) {
                        float* o = p.dgates + row * (int64_t)(ND * H4) + dir * H4;
                        o[u] = gi; o[H + u] = gf; o[2 * H + u] = gg; o[3 * H + u] = go;
                    }
#endif
                }
                // B operand: k-step (g, c = u / 4), q = r  ->  dgw[sq][r][g * HP/4 + c]   (zero for padded units / idle sequences)
                if (u < HP) {
                    float* d = dgw + (sq * 4 + r) * RS + (u >> 2);
                    d[0] = gi; d[HP / 4] = gf; d[2 * (HP / 4)] = gg; d[3 * (HP / 4)] = go;
                }
            }
#ifndef NIR_BW_NOSTORE
            if (vec && on && sv && ub[i] < H) {
                float* o = p.dgates + row * (int64_t)(ND * H4) + dir * H4 + ub[i];
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(o + g * H) = make_float4(gv[g][0], gv[g][1], gv[g][2], gv[g][3]);
            }
            if (vec2 && on && sv && ub[i] < H) {
                float* o = p.dgates + row * (int64_t)(ND * H4) + dir * H4 + ub[i];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *reinterpret_cast<float2*>(o + g * H) = make_float2(gv[g][0], gv[g][1]);
                    if (ub[i] + 2 < H) *reinterpret_cast<float2*>(o + g * H + 2) = make_float2(gv[g][2], gv[g][3]);
                }
            }
#endif
        }
#ifndef NIR_BW_NOLOAD
        load_cells(step - 1, cin);                                     // lands under this step's MFMAs
#endif
        lds_barrier();                                   // LDS only: the prefetched cell inputs and the dgates stores stay in flight under the MFMAs
        f32x4 acc[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* bp = dgw + (sq * 4 + pq) * RS;
#ifndef NIR_BW_NOMFMA
#pragma unroll
        for (int k4 = 0; k4 < NKS / 4; ++k4) {
            const float4 b = *reinterpret_cast<const float4*>(bp + 4 * k4);
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[i][4 * k4 + 0], b.x, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[i][4 * k4 + 1], b.y, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[i][4 * k4 + 2], b.z, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[i][4 * k4 + 3], b.w, acc[i], 0, 0, 0);
            }
        }
#else
        acc[0][0] = bp[0] + afr[0][0] + afr[1][NKS - 1];
#endif
        // sequences that have not started yet (step >= len) keep the final-state gradient
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (on) dh[i][r] = acc[i][r];
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = ub[i] + r;
            if (sv && u < H) {
                const int64_t si = ((int64_t)dir * p.M + m0 + sq) * H + u;
                if (p.dh0) p.dh0[si] = dh[i][r];
                if (p.dc0) p.dc0[si] = dc[i][r];
            }
        }
    // zero the gate gradients of the padded steps: one wave per (sequence, step) row, coalesced
    __syncthreads();
    for (int s_ = 0; s_ < SEQ && m0 + s_ < p.M; ++s_)
        for (int t2 = lens_s[s_] + wave; t2 < T; t2 += NWV) {
            float* o = p.dgates + ((m0 + s_) * T + t2) * (int64_t)(ND * H4) + dir * H4;
            for (int col = lane; col < H4; col += 64) o[col] = 0.f;
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// element-wise pieces
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// keep[i] = uniform(seed, i) >= p ; y = x * keep / (1 - p)
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ keep, int64_t n, float pdrop,
                               uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float u = (float)(splitmix(seed ^ (uint64_t)i * 0xD1342543DE82EF95ull) >> 40) * (1.0f / 16777216.0f);
    const bool k = u >= pdrop;
    keep[i] = k ? 1 : 0;
    y[i] = k ? x[i] / (1.0f - pdrop) : 0.f;
}
// the same with the seed read from device memory (a captured training step replays with a fresh seed: the graph's first node advances it)
__global__ void dropout_dev_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ keep, int64_t n, float pdrop,
                                   const uint64_t* __restrict__ seed_dev, uint64_t salt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t seed = splitmix(seed_dev[0] * 0x9E3779B97F4A7C15ull + salt * 0xD1B54A32D192ED03ull);
    const float u = (float)(splitmix(seed ^ (uint64_t)i * 0xD1342543DE82EF95ull) >> 40) * (1.0f / 16777216.0f);
    const bool k = u >= pdrop;
    keep[i] = k ? 1 : 0;
    y[i] = k ? x[i] / (1.0f - pdrop) : 0.f;
}
__global__ void mask_scale_kernel(const float* __restrict__ x, const unsigned char* __restrict__ keep, float scale, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = keep[i] ? x[i] * scale : 0.f;
}
// dx = dy * f'(y) for y = f(x): act 1 tanh (1 - y^2), 2 relu (y > 0), 3 sigmoid y (1 - y)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = y[i];
    const float d = act == 1 ? 1.f - v * v : act == 2 ? (v > 0.f ? 1.f : 0.f) : act == 3 ? v * (1.f - v) : 1.f;
    dx[i] = dy[i] * d;
}
// d mean-BCE-with-logits / d score = (sigmoid(s) - y) * gscale / n
__global__ void bce_bwd_kernel(const float* __restrict__ s, const float* __restrict__ y, const float* __restrict__ gout, float* __restrict__ ds, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ds[i] = (1.0f / (1.0f + expf(-s[i])) - y[i]) * gout[0] / (float)n;
}
// Suggestion loss rows (neuroir/models/multitask.py:203-216, seq2seq loss): for every decoder row r with logits z [V] and target t,
//   nll[r] = -(log_softmax z)[t] (0 when t == pad),   ent[r] = sum_v p_v log p_v (the entropy regulariser's row term),   lse[r] = logsumexp z.
// One workgroup per row, two passes over the row (the second one out of L2): the [rows, V] log-softmax, its exp and their product are never
// written (the reference's five full-size temporaries: 16 elementwise / reduction launches per step at V = 30 000).
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}
__global__ __launch_bounds__(256) void softmax_nll_ent_fwd_kernel(const float* __restrict__ z, int64_t ld, const int64_t* __restrict__ target, int64_t pad,
                                                                  int V, float* __restrict__ nll, float* __restrict__ ent, float* __restrict__ lse,
                                                                  int* err) {
    __shared__ float red[4];
    const float* zr = z + (int64_t)blockIdx.x * ld;
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) m = fmaxf(m, zr[v]);
    m = block_reduce(m, red, true);
    float s = 0.f, q = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float d = zr[v] - m, e = __expf(d);
        s += e;
        q = fmaf(e, d, q);
    }
    s = block_reduce(s, red, false);
    q = block_reduce(q, red, false);
    if (threadIdx.x == 0) {
        const float ls = __logf(s);
        const int64_t t = target[blockIdx.x];
        const bool ok = t >= 0 && t < V;
        if (!ok && err) atomicOr(err, 1);
        lse[blockIdx.x] = m + ls;
        ent[blockIdx.x] = q / s - ls;
        nll[blockIdx.x] = (ok && t != pad) ? (m + ls) - zr[t] : 0.f;
    }
}
// dz[r, v] = p_v (ga + gb (log p_v - ent[r])) - ga [v == t],   ga = gnll[r] (0 for a pad target), gb = gent[r]
__global__ __launch_bounds__(256) void softmax_nll_ent_bwd_kernel(const float* __restrict__ z, int64_t ld, const int64_t* __restrict__ target, int64_t pad,
                                                                  const float* __restrict__ lse, const float* __restrict__ ent,
                                                                  const float* __restrict__ gnll, const float* __restrict__ gent, int V,
                                                                  float* __restrict__ dz) {
    const int64_t r = blockIdx.y;
    const int64_t t = target[r];
    const float ga = (t != pad && t >= 0 && t < V) ? gnll[r] : 0.f, gb = gent ? gent[r] : 0.f;
    const float l = lse[r], e0 = ent[r];
    const float* zr = z + r * ld;
    float* dr = dz + r * (int64_t)V;
    const int v = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (v + 3 < V && (ld & 3) == 0 && (V & 3) == 0) {
        const float4 x = *reinterpret_cast<const float4*>(zr + v);
        float4 o;
        float lp = x.x - l; o.x = __expf(lp) * fmaf(gb, lp - e0, ga);
        lp = x.y - l; o.y = __expf(lp) * fmaf(gb, lp - e0, ga);
        lp = x.z - l; o.z = __expf(lp) * fmaf(gb, lp - e0, ga);
        lp = x.w - l; o.w = __expf(lp) * fmaf(gb, lp - e0, ga);
        if (t >= v && t < v + 4) (&o.x)[t - v] -= ga;
        *reinterpret_cast<float4*>(dr + v) = o;
    } else {
        for (int j = v; j < min(V, v + 4); ++j) {
            const float lp = zr[j] - l;
            dr[j] = __expf(lp) * fmaf(gb, lp - e0, ga) - (j == t ? ga : 0.f);
        }
    }
}
// Masked softmax + weighted sum of the attention sites of the training forwards (cars.py:262-304, 520-600 and the decoder's global attention):
//   w[r, :] = softmax(logits[r, :] where mask, -inf elsewhere),   out[r, :] = sum_t w[r, t] V[r / G, t, :]
// (G consecutive rows share one value block: the causal session attentions and the decoder steps).  mask row of r = (r / mdiv) % mmod.
// The op-by-op form is ~6 launches forward (not, copy, masked_fill, softmax, mul, sum -- the product a [R, T, D] temporary) and ~8 backward.
__global__ __launch_bounds__(256) void softmax_pool_fwd_kernel(const float* __restrict__ logits, const unsigned char* __restrict__ mask, int64_t mdiv,
                                                               int64_t mmod, const float* __restrict__ V, int G, int T, int D, float* __restrict__ w_out,
                                                               float* __restrict__ out) {
    extern __shared__ float ws[];                       // [T]
    __shared__ float red[4];
    const int64_t r = blockIdx.x, g = r / G;
    const unsigned char* mr = mask ? mask + ((r / mdiv) % mmod) * T : nullptr;
    const float* lr = logits + r * T;
    float m = -INFINITY;
    for (int t = threadIdx.x; t < T; t += 256) {
        const float x = (!mr || mr[t]) ? lr[t] : -INFINITY;
        ws[t] = x;
        m = fmaxf(m, x);
    }
    m = block_reduce(m, red, true);
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) {
        const float e = expf(ws[t] - m);
        ws[t] = e;
        s += e;
    }
    s = block_reduce(s, red, false);
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
        const float w = ws[t] / s;
        ws[t] = w;
        w_out[r * T + t] = w;
    }
    __syncthreads();
    const float* vb = V + g * (int64_t)T * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc = fmaf(ws[t], vb[(int64_t)t * D + d], acc);
        out[r * D + d] = acc;
    }
}
// one workgroup per value block g (its G rows):  dw[r,t] = dout[r,:] . V[g,t,:];  dlogits[r,t] = w[r,t] (dw[r,t] - sum_t' w[r,t'] dw[r,t']);
// dV[g,t,:] = sum_r w[r,t] dout[r,:]
__global__ __launch_bounds__(256) void softmax_pool_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dout, const float* __restrict__ V,
                                                               int G, int T, int D, float* __restrict__ dlogits, float* __restrict__ dV) {
    extern __shared__ float sm[];                       // wl [G][T], dw [G][T]
    float* wl = sm;
    float* dw = sm + G * T;
    const int64_t g = blockIdx.x, r0 = g * G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < G * T; e += 256) wl[e] = w[r0 * T + e];
    const float* vb = V + g * (int64_t)T * D;
    // (row, t) pairs: lanes over d, FOUR pairs per wave and trip -- their loads are in flight together (the causal session attentions are 16
    // workgroups of 49 pairs: one pair per trip was 12 dependent L2 round trips, 30 us for a few kilobytes)
    for (int pq0 = 4 * wave; pq0 < G * T; pq0 += 16) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const float* dr[4];
        const float* vr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pq = min(pq0 + u, G * T - 1), rl = pq / T, t = pq - rl * T;
            dr[u] = dout + (r0 + rl) * D;
            vr[u] = vb + (int64_t)t * D;
        }
        for (int d = lane; d < D; d += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = fmaf(dr[u][d], vr[u][d], a[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float sum = wave_sum(a[u]);
            if (lane == 0 && pq0 + u < G * T) dw[pq0 + u] = sum;
        }
    }
    __syncthreads();
    for (int rl = threadIdx.x; rl < G; rl += 256) {
        float dot = 0.f;
        for (int t = 0; t < T; ++t) dot = fmaf(wl[rl * T + t], dw[rl * T + t], dot);
        for (int t = 0; t < T; ++t) {
            const float wv = wl[rl * T + t];
            // (a masked position has w = 0 exactly: its gradient is 0 whatever dw is; 0 * inf cannot occur, dw is finite)
            dlogits[(r0 + rl) * T + t] = wv * (dw[rl * T + t] - dot);
        }
    }
    if (dV) {
        float* dvb = dV + g * (int64_t)T * D;
        for (int d = threadIdx.x; d < D; d += 256)
            for (int t = 0; t < T; ++t) {
                float a = 0.f;
                for (int rl = 0; rl < G; ++rl) a = fmaf(wl[rl * T + t], dout[(r0 + rl) * D + d], a);
                dvb[(int64_t)t * D + d] = a;
            }
    }
}

// embedding lookup: out[m,:] = table[ids[m],:]  (optionally with an inverted-dropout keep mask that is also returned)
__global__ void embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table, int64_t V, int E, int64_t M, float* __restrict__ out,
                             int* err) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * (E / 4)) return;
    const int64_t m = i / (E / 4);
    const int c = (int)(i % (E / 4)) * 4;
    int64_t id = ids[m];
    if (id < 0 || id >= V) { if (err) atomicOr(err, 1); id = 0; }
    *reinterpret_cast<float4*>(out + m * E + c) = *reinterpret_cast<const float4*>(table + id * E + c);
}
__global__ void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout, int64_t V, int E, int64_t M, float* __restrict__ dtable,
                                 int64_t pad_idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * E) return;
    const int64_t m = i / E;
    const int c = (int)(i % E);
    const int64_t id = ids[m];
    if (id < 0 || id >= V || id == pad_idx) return;          // nn.Embedding(padding_idx): the PAD row receives no gradient
    atomicAdd(dtable + id * E + c, dout[i]);
}

// Single LSTM cell step on summed gate pre-activations g [B,4H] (i,f,g,o): any hidden size (session LSTMs H = 512, decoder).
// act [B,4H] keeps the gate activations for the backward.
__global__ void lstm_cell_fwd_kernel(const float* __restrict__ g, const float* __restrict__ cprev, float* __restrict__ act, float* __restrict__ c,
                                     float* __restrict__ h, int64_t B, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int64_t b = i / H;
    const int j = (int)(i % H);
    const float* gr = g + b * 4 * H;
    const float gi = 1.0f / (1.0f + expf(-gr[j])), gf = 1.0f / (1.0f + expf(-gr[H + j]));
    const float gg = tanhf(gr[2 * H + j]), go = 1.0f / (1.0f + expf(-gr[3 * H + j]));
    float* ar = act + b * 4 * H;
    ar[j] = gi; ar[H + j] = gf; ar[2 * H + j] = gg; ar[3 * H + j] = go;
    const float cn = gf * (cprev ? cprev[i] : 0.f) + gi * gg;
    c[i] = cn;
    h[i] = go * tanhf(cn);
}
// dh, dc (either may be NULL) -> dg [B,4H] (pre-activation), dcprev [B,H]
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dc, const float* __restrict__ act,
                                     const float* __restrict__ c, const float* __restrict__ cprev, float* __restrict__ dg,
                                     float* __restrict__ dcprev, int64_t B, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int64_t b = i / H;
    const int j = (int)(i % H);
    const float* ar = act + b * 4 * H;
    const float gi = ar[j], gf = ar[H + j], gg = ar[2 * H + j], go = ar[3 * H + j];
    const float th = tanhf(c[i]);
    const float dhh = dh ? dh[i] : 0.f;
    const float dct = (dc ? dc[i] : 0.f) + dhh * go * (1.f - th * th);
    float* dr = dg + b * 4 * H;
    dr[j] = dct * gg * gi * (1.f - gi);
    dr[H + j] = dct * (cprev ? cprev[i] : 0.f) * gf * (1.f - gf);
    dr[2 * H + j] = dct * gi * (1.f - gg * gg);
    dr[3 * H + j] = dhh * th * go * (1.f - go);
    dcprev[i] = dct * gf;
}

// The same cell inside a [B,T,.] sequence buffer (autograd._LSTMSeq): the step's gates are gx (row stride ldgx: the input projection of all steps) +
// gh ([B,4H] contiguous: the recurrent GEMM of this step, bias included) or + bias (first step without an initial state); act / c / h are written
// into the step's column of the sequence buffers (row strides).  No per-step add, stack or copy kernels around it.
__global__ void lstm_cell_seq_fwd_kernel(const float* __restrict__ gx, int64_t ldgx, const float* __restrict__ gh, const float* __restrict__ bias,
                                         const float* __restrict__ cprev, int64_t ldcp, float* __restrict__ act, int64_t ldact, float* __restrict__ c,
                                         int64_t ldc, float* __restrict__ h, int64_t ldh, int64_t B, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int64_t b = i / H;
    const int j = (int)(i % H);
    const float* xr = gx + b * ldgx;
    float p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = xr[q * H + j] + (gh ? gh[b * 4 * H + q * H + j] : (bias ? bias[q * H + j] : 0.f));
    const float gi = 1.0f / (1.0f + expf(-p[0])), gf = 1.0f / (1.0f + expf(-p[1]));
    const float gg = tanhf(p[2]), go = 1.0f / (1.0f + expf(-p[3]));
    float* ar = act + b * ldact;
    ar[j] = gi; ar[H + j] = gf; ar[2 * H + j] = gg; ar[3 * H + j] = go;
    const float cn = gf * (cprev ? cprev[b * ldcp + j] : 0.f) + gi * gg;
    c[b * ldc + j] = cn;
    h[b * ldh + j] = go * tanhf(cn);
}
// dh = dh1 (strided, the consumers of this step's h) + dh2 (contiguous, from the next step's recurrent GEMM), dc likewise; any may be NULL
// (lens != NULL: the step is position t of padded sequences -- a row with t >= lens[b] takes no part: zero gate gradients, zero dc_prev, its
//  incoming gradients ignored; c_prev exists where 0 <= tprev < lens[b], tprev = the position of the previous recurrence step)
__global__ void lstm_cell_seq_bwd_kernel(const float* __restrict__ dh1, int64_t ld1, const float* __restrict__ dh2, const float* __restrict__ dc1,
                                         int64_t ldc1, const float* __restrict__ dc2, const float* __restrict__ act, int64_t ldact,
                                         const float* __restrict__ c, int64_t ldc, const float* __restrict__ cprev, int64_t ldcp, float* __restrict__ dg,
                                         int64_t lddg, float* __restrict__ dcprev, int64_t B, int H, const int64_t* __restrict__ lens, int t, int tprev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int64_t b = i / H;
    const int j = (int)(i % H);
    if (lens) {
        const int64_t l = lens[b];
        if (t >= l) {
            float* dr = dg + b * lddg;
            dr[j] = 0.f; dr[H + j] = 0.f; dr[2 * H + j] = 0.f; dr[3 * H + j] = 0.f;
            dcprev[i] = 0.f;
            return;
        }
        if (tprev < 0 || tprev >= l) cprev = nullptr;
    }
    const float* ar = act + b * ldact;
    const float gi = ar[j], gf = ar[H + j], gg = ar[2 * H + j], go = ar[3 * H + j];
    const float th = tanhf(c[b * ldc + j]);
    const float dhh = (dh1 ? dh1[b * ld1 + j] : 0.f) + (dh2 ? dh2[i] : 0.f);
    const float dct = (dc1 ? dc1[b * ldc1 + j] : 0.f) + (dc2 ? dc2[i] : 0.f) + dhh * go * (1.f - th * th);
    float* dr = dg + b * lddg;
    dr[j] = dct * gg * gi * (1.f - gi);
    dr[H + j] = dct * (cprev ? cprev[b * ldcp + j] : 0.f) * gf * (1.f - gf);
    dr[2 * H + j] = dct * gi * (1.f - gg * gg);
    dr[3 * H + j] = dhh * th * go * (1.f - go);
    dcprev[i] = dct * gf;
}

static inline dim3 g1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

// ---- im2col as ROWS (training forwards of the 2-D convolutions, rankers/mtensor.py:108-121): out[(m, y, x)][(c, dy, dx)] =
// in[m, c, y + dy - ph, x + dx - pw] (0 outside), stride 1, "same" geometry (2 ph = kh - 1, 2 pw = kw - 1) -- the A operand of the
// filter GEMM, in ONE launch.  (F.unfold runs one im2col kernel per sample -- 960 launches per MatchTensor step at C2 -- and yields
// [M, K, L], which the GEMM needs transposed: one more copy of the largest tensor of the step.)  k is the fastest output index: coalesced
// stores; the loads walk a [C, H, W] block that stays in L1 / L2.
__global__ __launch_bounds__(256) void im2col_rows_kernel(const float* __restrict__ in, int C, int H, int W, int kh, int kw, int ph, int pw,
                                                          float* __restrict__ out) {
    // one workgroup per (m, y): a thread owns patch columns k = tid, tid + 256, .. (its (c, dy, dx) and source row are found once per k,
    // 32-bit arithmetic) and walks x: the W stores of a wave instruction are 64 consecutive k of one output row
    const int64_t m = blockIdx.x / H;
    const int y = (int)(blockIdx.x - m * H);
    const int K = C * kh * kw;
    float* const orow = out + ((m * H + y) * (int64_t)W) * K;
    for (int k = threadIdx.x; k < K; k += 256) {
        const int c = k / (kh * kw), r = k - c * kh * kw, dy = r / kw, dx = r - dy * kw;
        const int yy = y + dy - ph, x0 = dx - pw;
        const bool rowok = yy >= 0 && yy < H;
        const float* const src = in + ((m * C + c) * H + (rowok ? yy : 0)) * (int64_t)W;
        for (int x = 0; x < W; ++x) {
            const int xx = x + x0;
            const bool ok = rowok && xx >= 0 && xx < W;
            const float v = src[ok ? xx : 0];                                  // clamped address, no predicated load
            orow[(int64_t)x * K + k] = ok ? v : 0.f;
        }
    }
}

// Its backward (col2im from rows): din[m, c, y, x] = sum over (dy, dx) of drows[(m, y - dy + ph, x - dx + pw)][(c, dy, dx)].  One workgroup
// per (m, y), channels in groups of CG: for every tap row dy it stages the [W][CG * kw] slice of the source row y - dy + ph in LDS
// (coalesced runs of kw floats) and every thread sums its (c, x) outputs over dx in a fixed order -- deterministic (no atomics), no
// transposed copy of drows, one coalesced store per output.
constexpr int C2I_MAXOUT = 8;                                 // outputs per thread and channel group: CG * W <= 2048
__global__ __launch_bounds__(256) void col2im_rows_kernel(const float* __restrict__ drows, int C, int H, int W, int kh, int kw, int ph, int pw, int CG,
                                                          float* __restrict__ din) {
    extern __shared__ float c2i_slice[];                     // [W][CG * kw | 1]
    const int64_t m = blockIdx.x / H;
    const int y = (int)(blockIdx.x - m * H);
    const int K = C * kh * kw;
    for (int c0 = 0; c0 < C; c0 += CG) {
        const int cg = C - c0 < CG ? C - c0 : CG, per = cg * kw, nout = cg * W;
        const int perp = per | 1;                            // odd row stride: the sums below read a column of the slice (x varies across the lanes)
        float acc[C2I_MAXOUT];
#pragma unroll
        for (int j = 0; j < C2I_MAXOUT; ++j) acc[j] = 0.f;
        for (int dy = 0; dy < kh; ++dy) {
            const int ys = y - dy + ph;                      // source row whose tap dy lands on y (uniform)
            if (ys < 0 || ys >= H) continue;
            const float* src = drows + ((m * H + ys) * (int64_t)W) * K + (int64_t)c0 * kh * kw + dy * kw;
            __syncthreads();
            for (int i = threadIdx.x; i < W * per; i += 256) {
                const int xs = i / per, q = i - xs * per, cl = q / kw, dx = q - cl * kw;
                c2i_slice[xs * perp + q] = src[(int64_t)xs * K + cl * kh * kw + dx];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < C2I_MAXOUT; ++j) {
                const int o = threadIdx.x + 256 * j;
                if (o < nout) {
                    const int cl = o / W, x = o - cl * W;
                    float a = acc[j];
                    for (int dx = 0; dx < kw; ++dx) {
                        const int xs = x - dx + pw;
                        if (xs >= 0 && xs < W) a += c2i_slice[xs * perp + cl * kw + dx];
                    }
                    acc[j] = a;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < C2I_MAXOUT; ++j) {
            const int o = threadIdx.x + 256 * j;
            if (o < nout) {
                const int cl = o / W, x = o - cl * W;
                din[((m * C + c0 + cl) * H + y) * W + x] = acc[j];
            }
        }
    }
}

}  // namespace nir

extern "C" int nir_lstm_cell_fwd(const float* gates, const float* c_prev, float* act, float* c, float* h, int64_t B, int H, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(gates && act && c && h && B >= 0 && H > 0, "lstm_cell_fwd: bad args");
    if (B == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, g1(B * H), dim3(256), 0, (hipStream_t)stream, gates, c_prev, act, c, h, B, H);
    NIR_CHECK_LAUNCH("lstm_cell_fwd_kernel");
    return 0;
}
extern "C" int nir_lstm_cell_bwd(const float* dh, const float* dc, const float* act, const float* c, const float* c_prev, float* dgates,
                                 float* dc_prev, int64_t B, int H, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(act && c && dgates && dc_prev && B >= 0 && H > 0, "lstm_cell_bwd: bad args");
    if (B == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, g1(B * H), dim3(256), 0, (hipStream_t)stream, dh, dc, act, c, c_prev, dgates, dc_prev, B, H);
    NIR_CHECK_LAUNCH("lstm_cell_bwd_kernel");
    return 0;
}

struct WgradRows { const int32_t* rows = nullptr; const int32_t* mcount = nullptr; int64_t dyd = 0, xd = 0; int period = 0, skip = 0; };
// collects the small (1 x 1 path) gradients of nir_linear_wgrad_group_f32 and launches them WG_GROUP at a time
struct WgradCollector {
    nir::WgradGroup g;
    hipStream_t st;
    explicit WgradCollector(hipStream_t s) : st(s) { g.n = 0; g.bstart[0] = 0; }
    int flush() {
        using namespace nir;
        if (g.n == 0) return 0;
        ProfScope ps(prof_shape_name("wgrad_group_kernel", g.n, g.bstart[g.n], 0), st);
        hipLaunchKernelGGL(wgrad_group_kernel, dim3((unsigned)g.bstart[g.n]), dim3(256), 0, st, g);
        NIR_CHECK_LAUNCH("wgrad_group_kernel");
        g.n = 0;
        return 0;
    }
    int add(const nir::WgradArgs& a, int64_t blocks, bool sh) {
        if (g.n == nir::WG_GROUP || (int64_t)g.bstart[g.n] + blocks > (1 << 30)) NIR_PROPAGATE(flush());
        g.a[g.n] = a; g.sh[g.n] = sh ? 1 : 0;
        g.bstart[g.n + 1] = g.bstart[g.n] + (int)blocks;
        ++g.n;
        return 0;
    }
};
static int wgrad_impl(const float* dy, int64_t lddy, const float* x, int64_t ldx, const int64_t* ids, const float* table, int E, float* dw,
                      int64_t lddw, int64_t M, int N, int K, bool set, hipStream_t st, float* db = nullptr, WgradRows rl = WgradRows(),
                      WgradCollector* grp = nullptr) {
    using namespace nir;
    NIR_REQUIRE(dy && dw && (ids ? (table != nullptr && E >= K) : (x != nullptr)), "linear_wgrad: null pointer");
    NIR_REQUIRE(M >= 0 && N > 0 && K > 0, "linear_wgrad: bad dims");
    if (M == 0) {
        if (set && db) NIR_PROPAGATE((int)hipMemsetAsync(db, 0, (size_t)N * 4, st));
        if (set) return (int)hipMemset2DAsync(dw, (size_t)lddw * 4, 0, (size_t)K * 4, (size_t)N, st);
        return 0;
    }
    // LDS-staged workgroup tiles for the big gradients (float4 rows: widths, strides and bases in units of 16 bytes)
    const int64_t xstride = ids ? (int64_t)E : ldx;
    const bool lds_ok = (M >= 32768 || (M >= 256 && (int64_t)((N + 127) / 128) * ((K + 127) / 128) >= (tun(g_tun.wgrad_lds_tiles) > 0 ? tun(g_tun.wgrad_lds_tiles) : 192))) && N >= 128 && K >= 128 && N % 4 == 0 && K % 4 == 0 && lddy % 4 == 0 && xstride % 4 == 0 &&
                        ((uintptr_t)dy & 15) == 0 && ((uintptr_t)(ids ? table : x) & 15) == 0 && !tun(g_tun.wgrad_no_lds);
    if (lds_ok) {
        int wk = 2;                                                    // K tile of 64 wk columns: least padding, then the widest
        int64_t best = -1;
        for (int c = 2; c <= 5; ++c) {
            const int64_t padded = (int64_t)((K + 64 * c - 1) / (64 * c)) * 64 * c;
            if (best < 0 || padded < best || (padded == best && c > wk)) { best = padded; wk = c; }
        }
        const int64_t tiles = (int64_t)((N + 127) / 128) * ((K + 64 * wk - 1) / (64 * wk));
        int64_t slices = std::max<int64_t>(1, std::min<int64_t>((M + 255) / 256, (512 + tiles - 1) / tiles));
        const int64_t mslice = ((M + slices - 1) / slices + 15) / 16 * 16;
        slices = (M + mslice - 1) / mslice;
        const int store = set && slices == 1;
        if (set && !store) NIR_PROPAGATE((int)hipMemset2DAsync(dw, (size_t)lddw * 4, 0, (size_t)K * 4, (size_t)N, st));
        if (set && !store && db) NIR_PROPAGATE((int)hipMemsetAsync(db, 0, (size_t)N * 4, st));
        WgradArgs a{dy, lddy, x, ldx, ids, table, E, dw, lddw, M, N, K, mslice, store, db, rl.rows, rl.mcount, rl.dyd, rl.xd, (int)slices, rl.period > 1 ? (unsigned)((((uint64_t)1 << 32) + rl.period - 1) / rl.period) : 0u, rl.period, rl.skip};
        ProfScope ps(prof_shape_name("wgrad_lds_kernel", M, N, K), st);
        switch (wk) {
            case 2: wgrad_lds_launch<2, 2>(a, tiles * slices, st); break;
            case 3: wgrad_lds_launch<2, 3>(a, tiles * slices, st); break;
            case 4: wgrad_lds_launch<2, 4>(a, tiles * slices, st); break;
            default:                                                    // 128 x 320: eight waves of 1 x 5 blocks (tunable wgrad_lds_tiles = -5: ten of 2 x 2)
                if (tun(g_tun.wgrad_lds_tiles) == -5) wgrad_lds_launch<2, 5>(a, tiles * slices, st);
                else wgrad_lds_launch<4, 2, 1, 5>(a, tiles * slices, st);
                break;
        }
        NIR_CHECK_LAUNCH("wgrad_lds_kernel");
        return 0;
    }
    // 2 x 2 blocking once there is enough work for it to pay (big M) and the tile is not mostly padding
    const bool big = M >= 4096 && N >= 64 && K >= 64;      // (small M with a big N x K: measured slower than 1 x 1 with 64-row slices)
    const int nb = big ? 2 : 1, kb = big ? 2 : 1;
    const int64_t tiles = (int64_t)((N + 32 * nb - 1) / (32 * nb)) * ((K + 32 * kb - 1) / (32 * kb));
    // slices of >= 256 rows (big M) / >= 64 rows (1 x 1: a slice is then two to eight load -> MFMA round trips), as many as fill the chip
    const int64_t minrows = (big && M >= 4096) ? 256 : std::max(32, tun(g_tun.wgrad_min_rows) > 0 ? tun(g_tun.wgrad_min_rows) : 64);
    int64_t slices = std::max<int64_t>(1, std::min<int64_t>((M + minrows - 1) / minrows, (4096 + tiles - 1) / tiles));
    const int64_t mslice = ((M + slices - 1) / slices + 31) / 32 * 32;
    slices = (M + mslice - 1) / mslice;
    const int store = set && slices == 1;
    if (set && !store) NIR_PROPAGATE((int)hipMemset2DAsync(dw, (size_t)lddw * 4, 0, (size_t)K * 4, (size_t)N, st));
    if (set && !store && db) NIR_PROPAGATE((int)hipMemsetAsync(db, 0, (size_t)N * 4, st));
    WgradArgs a{dy, lddy, x, ldx, ids, table, E, dw, lddw, M, N, K, mslice, store, db, rl.rows, rl.mcount, rl.dyd, rl.xd, (int)slices, rl.period > 1 ? (unsigned)((((uint64_t)1 << 32) + rl.period - 1) / rl.period) : 0u, rl.period, rl.skip};
    if (grp && !big && !set) {                       // (accumulating small gradient: joins the group launch)
        const bool sh = slices >= 8 && tiles <= 64;
        return grp->add(a, sh ? tiles * ((slices + 3) / 4) : (tiles * slices + 3) / 4, sh);
    }
    ProfScope ps(prof_shape_name("wgrad_kernel", M, N, K), st);
    if (big) hipLaunchKernelGGL((wgrad_kernel<2, 2>), dim3((unsigned)((tiles * slices + 3) / 4)), dim3(256), 0, st, a);
    else if (slices >= 8 && tiles <= 64)             // small dW, long reduction: slices of one tile share a workgroup (LDS add before the atomics)
        hipLaunchKernelGGL((wgrad_kernel<1, 1, 16, true>), dim3((unsigned)(tiles * ((slices + 3) / 4))), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad_kernel<1, 1, 16>), dim3((unsigned)((tiles * slices + 3) / 4)), dim3(256), 0, st, a);
    NIR_CHECK_LAUNCH("wgrad_kernel");
    return 0;
}
extern "C" int nir_linear_wgrad_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const int64_t* ids, const float* table, int E,
                                    float* dw, int64_t lddw, int64_t M, int N, int K, nir_stream_t stream) {
    return wgrad_impl(dy, lddy, x, ldx, ids, table, E, dw, lddw, M, N, K, false, (hipStream_t)stream);
}
extern "C" int nir_linear_wgrad_set_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const int64_t* ids, const float* table, int E,
                                        float* dw, int64_t lddw, int64_t M, int N, int K, nir_stream_t stream) {
    return wgrad_impl(dy, lddy, x, ldx, ids, table, E, dw, lddw, M, N, K, true, (hipStream_t)stream);
}

extern "C" int nir_linear_wgrad_bias_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const int64_t* ids, const float* table, int E,
                                         float* dw, int64_t lddw, float* db, int64_t M, int N, int K, nir_stream_t stream) {
    NIR_REQUIRE(db != nullptr, "linear_wgrad_bias: null bias gradient");
    return wgrad_impl(dy, lddy, x, ldx, ids, table, E, dw, lddw, M, N, K, false, (hipStream_t)stream, db);
}
extern "C" int nir_linear_wgrad_group_f32(int n, const float* const* dy, const int64_t* lddy, const float* const* x, const int64_t* ldx, float* const* dw,
                                          const int64_t* lddw, float* const* db, const int64_t* M, const int* N, const int* K, nir_stream_t stream) {
    NIR_REQUIRE(n >= 0 && (n == 0 || (dy && lddy && x && ldx && dw && lddw && db && M && N && K)), "linear_wgrad_group: null array");
    WgradCollector col((hipStream_t)stream);
    for (int i = 0; i < n; ++i)
        NIR_PROPAGATE(wgrad_impl(dy[i], lddy[i], x[i], ldx[i], nullptr, nullptr, 0, dw[i], lddw[i], M[i], N[i], K[i], false, (hipStream_t)stream, db[i], WgradRows(), &col));
    return col.flush();
}
extern "C" int nir_linear_wgrad_bias_set_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const int64_t* ids, const float* table, int E,
                                             float* dw, int64_t lddw, float* db, int64_t M, int N, int K, nir_stream_t stream) {
    NIR_REQUIRE(db != nullptr, "linear_wgrad_bias: null bias gradient");
    return wgrad_impl(dy, lddy, x, ldx, ids, table, E, dw, lddw, M, N, K, true, (hipStream_t)stream, db);
}

extern "C" int nir_seq_rows(const int64_t* lengths, int64_t M, int T, int t_begin, int32_t* offs, int32_t* rows, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(lengths && offs && rows, "seq_rows: null pointer");
    NIR_REQUIRE(M >= 0 && T > 0 && t_begin >= 0 && M * (int64_t)T < ((int64_t)1 << 31), "seq_rows: bad dims (M T < 2^31)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(seq_rows_scan_kernel, dim3(1), dim3(1024), 0, st, lengths, M, T, t_begin, offs);
    NIR_CHECK_LAUNCH("seq_rows_scan_kernel");
    if (M) {
        hipLaunchKernelGGL(seq_rows_fill_kernel, dim3((unsigned)((M * T + 255) / 256)), dim3(256), 0, st, lengths, offs, M, T, t_begin, rows);
        NIR_CHECK_LAUNCH("seq_rows_fill_kernel");
    }
    return 0;
}
extern "C" int nir_linear_wgrad_rows_set_f32(const float* dy, int64_t lddy, int64_t dy_row_delta, const float* x, int64_t ldx, int64_t x_row_delta,
                                             const int32_t* rows, const int32_t* count, int64_t max_rows, int period, int skip, float* dw, int64_t lddw,
                                             float* db, int N, int K, nir_stream_t stream) {
    NIR_REQUIRE((rows != nullptr) == (count != nullptr), "linear_wgrad_rows: rows and count go together");
    NIR_REQUIRE(period >= 0 && (period == 0 || (skip >= 0 && skip < period && max_rows < ((int64_t)1 << 31))), "linear_wgrad_rows: bad period / skip");
    WgradRows rl;
    rl.rows = rows; rl.mcount = count; rl.dyd = dy_row_delta; rl.xd = x_row_delta; rl.period = period; rl.skip = skip;
    return wgrad_impl(dy, lddy, x, ldx, nullptr, nullptr, 0, dw, lddw, max_rows, N, K, true, (hipStream_t)stream, db, rl);
}

static int colsum_impl(const float* x, int64_t ld, int64_t M, int N, float* out, bool set, hipStream_t st) {
    using namespace nir;
    NIR_REQUIRE(x && out && M >= 0 && N > 0, "colsum: bad args");
    if (M == 0) return set ? (int)hipMemsetAsync(out, 0, (size_t)N * 4, st) : 0;
    const int64_t mslice = std::max<int64_t>(64, (M + 255) / 256);
    const int64_t slices = (M + mslice - 1) / mslice;
    const int store = set && slices == 1;
    if (set && !store) NIR_PROPAGATE((int)hipMemsetAsync(out, 0, (size_t)N * 4, st));
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)slices), dim3(256), 0, st, x, ld, M, N, out, mslice, store);
    NIR_CHECK_LAUNCH("colsum_kernel");
    return 0;
}
extern "C" int nir_colsum_f32(const float* x, int64_t ld, int64_t M, int N, float* out, nir_stream_t stream) {
    return colsum_impl(x, ld, M, N, out, false, (hipStream_t)stream);
}
extern "C" int nir_colsum_set_f32(const float* x, int64_t ld, int64_t M, int N, float* out, nir_stream_t stream) {
    return colsum_impl(x, ld, M, N, out, true, (hipStream_t)stream);
}

extern "C" int nir_transpose_f32(const float* in, int R, int Cc, float* out, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(in && out && R > 0 && Cc > 0, "transpose: bad args");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((Cc + 31) / 32), (unsigned)((R + 31) / 32)), dim3(32, 8), 0, (hipStream_t)stream, in, R, Cc, out);
    NIR_CHECK_LAUNCH("transpose_kernel");
    return 0;
}

extern "C" int nir_transpose_group_f32(int n, const float* const* in, const int* R, const int* Cc, float* const* out, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(n >= 0 && (n == 0 || (in && R && Cc && out)), "transpose_group: null array");
    TransposeGroup g;
    g.n = 0; g.bstart[0] = 0;
    auto flush = [&]() {
        if (g.n == 0) return 0;
        hipLaunchKernelGGL(transpose_group_kernel, dim3((unsigned)g.bstart[g.n]), dim3(32, 8), 0, (hipStream_t)stream, g);
        NIR_CHECK_LAUNCH("transpose_group_kernel");
        g.n = 0;
        return 0;
    };
    for (int i = 0; i < n; ++i) {
        NIR_REQUIRE(in[i] && out[i] && R[i] > 0 && Cc[i] > 0, "transpose_group: bad item %d", i);
        const int64_t tiles = (int64_t)((Cc[i] + 31) / 32) * ((R[i] + 31) / 32);
        if (g.n == TG_GROUP || (int64_t)g.bstart[g.n] + tiles > (1 << 30)) NIR_PROPAGATE(flush());
        g.in[g.n] = in[i]; g.out[g.n] = out[i]; g.R[g.n] = R[i]; g.Cc[g.n] = Cc[i];
        g.bstart[g.n + 1] = g.bstart[g.n] + (int)tiles;
        ++g.n;
    }
    return flush();
}

extern "C" int nir_lstm_train_fwd(const float* gates_in, const int64_t* lengths, const float* w_hh, const float* h0, const float* c0, float* out,
                                  float* act, float* cst, float* hn, float* cn, int64_t M, int T, int H, int ndir, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(gates_in && w_hh && out && act && cst, "lstm_train_fwd: null pointer");
    NIR_REQUIRE(M >= 0 && T > 0 && (ndir == 1 || ndir == 2) && H >= 1 && H <= 128, "lstm_train_fwd: bad dims (H <= 128)");
    if (M == 0) return 0;
    if (H >= 33 && !tun(g_tun.lstm_valu)) {
        // the inference recurrence on the f32 matrix cores (csrc/lstm_mfma.hip: 16 sequences per workgroup, W_hh resident in registers, one
        // barrier per step) with the activation / cell-state stores of train mode: 4.9 ms -> 0.4 ms at the C3 document shape against the
        // one-thread-per-gate-row kernel below (kept for H <= 32 and as the `lstm_valu` cross-check)
        const int rc = launch_bilstm_mfma16(gates_in, lengths, w_hh, h0, c0, out, hn, cn, M, T, H, ndir, (hipStream_t)stream, act, cst);
        if (rc != NIR_ERR_UNSUPPORTED) return rc;
    }
    LstmTrainArgs a{gates_in, lengths, w_hh, out, act, cst, hn, cn, h0, c0, M, T, H, ndir};
    const dim3 grid((unsigned)((M + TSQ - 1) / TSQ), (unsigned)ndir);
    const int threads = (4 * H + 63) / 64 * 64;
    const size_t lds = (size_t)TSQ * 5 * H * 4;
    ProfScope ps(prof_shape_name("lstm_train_fwd_kernel", M, T, H), (hipStream_t)stream);
    if (H <= 32) hipLaunchKernelGGL(lstm_train_fwd_kernel<32>, grid, dim3(threads), lds, (hipStream_t)stream, a);
    else if (H <= 64) hipLaunchKernelGGL(lstm_train_fwd_kernel<64>, grid, dim3(threads), lds, (hipStream_t)stream, a);
    else if (H <= 96) hipLaunchKernelGGL(lstm_train_fwd_kernel<96>, grid, dim3(threads), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(lstm_train_fwd_kernel<128>, grid, dim3(threads), lds, (hipStream_t)stream, a);
    NIR_CHECK_LAUNCH("lstm_train_fwd_kernel");
    return 0;
}

extern "C" int nir_lstm_train_bwd(const float* dout, const float* dhn, const float* dcn, const float* dcst, const float* act, const float* cst, const float* c0,
                                  const int64_t* lengths, const float* w_hh, float* dgates, float* dh0, float* dc0, int64_t M, int T, int H,
                                  int ndir, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(dout && act && cst && w_hh && dgates, "lstm_train_bwd: null pointer");
    NIR_REQUIRE(M >= 0 && T > 0 && (ndir == 1 || ndir == 2) && H >= 1 && H <= 128, "lstm_train_bwd: bad dims (H <= 128)");
    if (M == 0) return 0;
    LstmBwdArgs a{dout, dhn, dcn, dcst, act, cst, c0, lengths, w_hh, dgates, dh0, dc0, M, T, H, ndir};
    // W_hh resident on the matrix cores for the hidden sizes that fill 4 waves x 2 unit tiles (H <= 128); tunable lstm_valu keeps the VALU form
    const int hp = (H + 3) / 4 * 4;
    // (H % 4 != 0 has no 16-byte cell IO: with scalar loads MatchTensor's H = 70 measured 666 against 427 us at 320 sequences, 1430 against
    // 1790 us at 2560; even H uses 8-byte pieces, odd H keeps the scalar form and this kernel only from 1024 sequences on)
    if (!tun(g_tun.lstm_valu) && H >= 16 && (hp == 32 || hp == 64 || hp == 72 || hp == 96 || hp == 128) && (H % 2 == 0 || M >= 1024)) {
        const size_t ldm = (size_t)2 * 16 * 4 * (hp + 4) * 4 + 16 * 4;
        ProfScope ps(prof_shape_name("lstm_train_bwd_mfma_kernel", M, T, H), (hipStream_t)stream);
        const dim3 grid((unsigned)((M + 15) / 16), (unsigned)ndir);
        hipStream_t st = (hipStream_t)stream;
        // more than four unit tiles: eight waves x one tile (tunable lstm_bwd_w8 = 2: the four-wave form; bit-identical results) -- C3 documents
        // 566 -> 443 us, MatchTensor's H = 70 418 -> 372 us
        const bool w8 = tun(g_tun.lstm_bwd_w8) != 2;
        if (hp == 128 && w8) hipLaunchKernelGGL((lstm_train_bwd_mfma_kernel<128, 1>), grid, dim3(512), ldm, st, a);
        else if (hp == 96 && w8) hipLaunchKernelGGL((lstm_train_bwd_mfma_kernel<96, 1>), grid, dim3(512), ldm, st, a);
        else if (hp == 72 && w8) hipLaunchKernelGGL((lstm_train_bwd_mfma_kernel<72, 1>), grid, dim3(512), ldm, st, a);
        else if (hp == 32) hipLaunchKernelGGL(lstm_train_bwd_mfma_kernel<32>, grid, dim3(256), ldm, st, a);
        else if (hp == 64) hipLaunchKernelGGL(lstm_train_bwd_mfma_kernel<64>, grid, dim3(256), ldm, st, a);
        else if (hp == 72) hipLaunchKernelGGL(lstm_train_bwd_mfma_kernel<72>, grid, dim3(256), ldm, st, a);
        else if (hp == 96) hipLaunchKernelGGL(lstm_train_bwd_mfma_kernel<96>, grid, dim3(256), ldm, st, a);
        else hipLaunchKernelGGL(lstm_train_bwd_mfma_kernel<128>, grid, dim3(256), ldm, st, a);
        NIR_CHECK_LAUNCH("lstm_train_bwd_mfma_kernel");
        return 0;
    }
    const int threads = (4 * H + 63) / 64 * 64;
    const size_t lds = (size_t)TSQ * (4 * H + H + 4 * H) * 4;
    ProfScope ps(prof_shape_name("lstm_train_bwd_kernel", M, T, H), (hipStream_t)stream);
    hipLaunchKernelGGL(lstm_train_bwd_kernel, dim3((unsigned)((M + TSQ - 1) / TSQ), (unsigned)ndir), dim3(threads), lds, (hipStream_t)stream, a);
    NIR_CHECK_LAUNCH("lstm_train_bwd_kernel");
    return 0;
}

extern "C" int nir_dropout_f32(const float* x, float* y, unsigned char* keep, int64_t n, float p, uint64_t seed, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(x && y && keep && n >= 0 && p >= 0.f && p < 1.f, "dropout: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, x, y, keep, n, p, seed);
    NIR_CHECK_LAUNCH("dropout_kernel");
    return 0;
}
extern "C" int nir_dropout_dev_f32(const float* x, float* y, unsigned char* keep, int64_t n, float p, const uint64_t* seed_dev, uint64_t salt,
                                   nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(x && y && keep && seed_dev && n >= 0 && p >= 0.f && p < 1.f, "dropout_dev: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_dev_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, x, y, keep, n, p, seed_dev, salt);
    NIR_CHECK_LAUNCH("dropout_dev_kernel");
    return 0;
}
extern "C" int nir_mask_scale_f32(const float* x, const unsigned char* keep, float scale, float* y, int64_t n, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(x && y && keep && n >= 0, "mask_scale: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(mask_scale_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, x, keep, scale, y, n);
    NIR_CHECK_LAUNCH("mask_scale_kernel");
    return 0;
}
extern "C" int nir_act_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, int act, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(dy && y && dx && n >= 0, "act_bwd: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_bwd_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n, act);
    NIR_CHECK_LAUNCH("act_bwd_kernel");
    return 0;
}
extern "C" int nir_rank_loss_bce_bwd(const float* scores, const float* labels, const float* grad_out, float* dscores, int64_t n, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(scores && labels && grad_out && dscores && n >= 0, "bce_bwd: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(bce_bwd_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, scores, labels, grad_out, dscores, n);
    NIR_CHECK_LAUNCH("bce_bwd_kernel");
    return 0;
}
extern "C" int nir_softmax_nll_ent_fwd(const float* logits, int64_t ld, const int64_t* target, int64_t pad, int64_t R, int V, float* nll, float* ent,
                                       float* lse, int* err_flag, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(logits && target && nll && ent && lse && R >= 0 && V > 0 && ld >= V, "softmax_nll_ent_fwd: bad args");
    if (R == 0) return 0;
    hipLaunchKernelGGL(softmax_nll_ent_fwd_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, ld, target, pad, V, nll, ent, lse, err_flag);
    NIR_CHECK_LAUNCH("softmax_nll_ent_fwd_kernel");
    return 0;
}
extern "C" int nir_softmax_nll_ent_bwd(const float* logits, int64_t ld, const int64_t* target, int64_t pad, const float* lse, const float* ent,
                                       const float* grad_nll, const float* grad_ent, int64_t R, int V, float* dlogits, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(logits && target && lse && ent && grad_nll && dlogits && R >= 0 && R < 65536 && V > 0 && ld >= V, "softmax_nll_ent_bwd: bad args (rows < 65536)");
    if (R == 0) return 0;
    hipLaunchKernelGGL(softmax_nll_ent_bwd_kernel, dim3((unsigned)((V + 1023) / 1024), (unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, ld, target, pad,
                       lse, ent, grad_nll, grad_ent, V, dlogits);
    NIR_CHECK_LAUNCH("softmax_nll_ent_bwd_kernel");
    return 0;
}
extern "C" int nir_lstm_cell_seq_fwd(const float* gx, int64_t ldgx, const float* gh, const float* bias, const float* c_prev, int64_t ldcp, float* act,
                                     int64_t ldact, float* c, int64_t ldc, float* h, int64_t ldh, int64_t B, int H, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(gx && act && c && h && B >= 0 && H > 0, "lstm_cell_seq_fwd: bad args");
    if (B == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_seq_fwd_kernel, g1(B * H), dim3(256), 0, (hipStream_t)stream, gx, ldgx, gh, bias, c_prev, ldcp, act, ldact, c, ldc, h, ldh, B, H);
    NIR_CHECK_LAUNCH("lstm_cell_seq_fwd_kernel");
    return 0;
}
extern "C" int nir_lstm_cell_seq_bwd(const float* dh_step, int64_t ld_dh, const float* dh_rec, const float* dc_step, int64_t ld_dc, const float* dc_rec,
                                     const float* act, int64_t ldact, const float* c, int64_t ldc, const float* c_prev, int64_t ldcp, float* dgates,
                                     int64_t lddg, float* dc_prev, int64_t B, int H, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(act && c && dgates && dc_prev && B >= 0 && H > 0, "lstm_cell_seq_bwd: bad args");
    if (B == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_seq_bwd_kernel, g1(B * H), dim3(256), 0, (hipStream_t)stream, dh_step, ld_dh, dh_rec, dc_step, ld_dc, dc_rec, act, ldact, c,
                       ldc, c_prev, ldcp, dgates, lddg, dc_prev, B, H, nullptr, 0, 0);
    NIR_CHECK_LAUNCH("lstm_cell_seq_bwd_kernel");
    return 0;
}
extern "C" int nir_lstm_cell_seq_bwd_masked(const float* dh_step, int64_t ld_dh, const float* dh_rec, const float* dc_rec, const float* act, int64_t ldact,
                                            const float* c, int64_t ldc, const float* c_prev, int64_t ldcp, float* dgates, int64_t lddg, float* dc_prev,
                                            const int64_t* lengths, int t, int t_prev, int64_t B, int H, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(act && c && dgates && dc_prev && lengths && B >= 0 && H > 0, "lstm_cell_seq_bwd_masked: bad args");
    if (B == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_seq_bwd_kernel, g1(B * H), dim3(256), 0, (hipStream_t)stream, dh_step, ld_dh, dh_rec, (const float*)nullptr, (int64_t)0, dc_rec,
                       act, ldact, c, ldc, c_prev, ldcp, dgates, lddg, dc_prev, B, H, lengths, t, t_prev);
    NIR_CHECK_LAUNCH("lstm_cell_seq_bwd_kernel");
    return 0;
}
extern "C" int nir_softmax_pool_fwd(const float* logits, const unsigned char* mask, int64_t mask_div, int64_t mask_mod, const float* values, int64_t R,
                                    int G, int T, int D, float* weights, float* out, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(logits && values && weights && out, "softmax_pool_fwd: null pointer");
    NIR_REQUIRE(R >= 0 && G >= 1 && R % G == 0 && T >= 1 && T <= 8192 && D >= 1 && (!mask || (mask_div >= 1 && mask_mod >= 1)), "softmax_pool_fwd: bad dims");
    if (R == 0) return 0;
    hipLaunchKernelGGL(softmax_pool_fwd_kernel, dim3((unsigned)R), dim3(256), (size_t)T * 4, (hipStream_t)stream, logits, mask, mask_div, mask_mod, values, G, T, D,
                       weights, out);
    NIR_CHECK_LAUNCH("softmax_pool_fwd_kernel");
    return 0;
}
extern "C" int nir_softmax_pool_bwd(const float* weights, const float* dout, const float* values, int64_t R, int G, int T, int D, float* dlogits,
                                    float* dvalues, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(weights && dout && values && dlogits, "softmax_pool_bwd: null pointer");
    NIR_REQUIRE(R >= 0 && G >= 1 && R % G == 0 && T >= 1 && D >= 1 && (int64_t)G * T * 8 <= 64 * 1024, "softmax_pool_bwd: bad dims (G T <= 8192)");
    if (R == 0) return 0;
    hipLaunchKernelGGL(softmax_pool_bwd_kernel, dim3((unsigned)(R / G)), dim3(256), (size_t)G * T * 8, (hipStream_t)stream, weights, dout, values, G, T, D, dlogits,
                       dvalues);
    NIR_CHECK_LAUNCH("softmax_pool_bwd_kernel");
    return 0;
}
extern "C" int nir_embed_f32(const int64_t* ids, const float* table, int64_t V, int E, int64_t M, float* out, int* err_flag, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(ids && table && out && V > 0 && E > 0 && E % 4 == 0 && M >= 0, "embed: bad args (E %% 4 == 0)");
    if (M == 0) return 0;
    hipLaunchKernelGGL(embed_kernel, g1(M * (E / 4)), dim3(256), 0, (hipStream_t)stream, ids, table, V, E, M, out, err_flag);
    NIR_CHECK_LAUNCH("embed_kernel");
    return 0;
}
extern "C" int nir_embed_bwd_f32(const int64_t* ids, const float* dout, int64_t V, int E, int64_t M, float* dtable, int64_t pad_idx, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(ids && dout && dtable && V > 0 && E > 0 && M >= 0, "embed_bwd: bad args");
    if (M == 0) return 0;
    hipLaunchKernelGGL(embed_bwd_kernel, g1(M * E), dim3(256), 0, (hipStream_t)stream, ids, dout, V, E, M, dtable, pad_idx);
    NIR_CHECK_LAUNCH("embed_bwd_kernel");
    return 0;
}
extern "C" int nir_im2col_rows_f32(const float* in, int64_t M, int C, int H, int W, int kh, int kw, int ph, int pw, float* out, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(in && out && M >= 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0, "im2col_rows: bad args");
    NIR_REQUIRE(2 * ph == kh - 1 && 2 * pw == kw - 1, "im2col_rows: only the 'same' geometry (2 pad = kernel - 1, stride 1); got kernel %dx%d pad %dx%d", kh, kw, ph, pw);
    if (M == 0) return 0;
    NIR_REQUIRE(M * H < (int64_t)1 << 31, "im2col_rows: too many rows for one launch");
    ProfScope ps("im2col_rows_kernel", (hipStream_t)stream);
    hipLaunchKernelGGL(im2col_rows_kernel, dim3((unsigned)(M * H)), dim3(256), 0, (hipStream_t)stream, in, C, H, W, kh, kw, ph, pw, out);
    NIR_CHECK_LAUNCH("im2col_rows_kernel");
    return 0;
}
extern "C" int nir_col2im_rows_f32(const float* drows, int64_t M, int C, int H, int W, int kh, int kw, int ph, int pw, float* din, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(drows && din && M >= 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0, "col2im_rows: bad args");
    NIR_REQUIRE(2 * ph == kh - 1 && 2 * pw == kw - 1, "col2im_rows: only the 'same' geometry (2 pad = kernel - 1, stride 1)");
    NIR_REQUIRE(W <= 2048 && kw <= 256, "col2im_rows: W = %d x kw = %d exceeds the LDS slice", W, kw);
    if (M == 0) return 0;
    NIR_REQUIRE(M * H < (int64_t)1 << 31, "col2im_rows: too many rows for one launch");
    int CG = std::min(C, std::min(2048 / W, (16384 / W - 1) / kw));      // outputs per thread <= 8, slice <= 64 KB
    CG = CG < 1 ? 1 : CG;
    // the slice rows are padded to an odd stride: validate the allocation that is actually requested (an even kw at the limit -- W = 2048,
    // kw = 8 -- asks for 72 KB although W * kw * 4 is exactly 64 KB)
    NIR_REQUIRE((size_t)W * (size_t)((CG * kw) | 1) * 4 <= 64 * 1024, "col2im_rows: W = %d x kw = %d exceeds the 64 KB LDS slice", W, kw);
    ProfScope ps("col2im_rows_kernel", (hipStream_t)stream);
    hipLaunchKernelGGL(col2im_rows_kernel, dim3((unsigned)(M * H)), dim3(256), (size_t)W * (CG * kw | 1) * 4, (hipStream_t)stream, drows, C, H, W, kh, kw, ph, pw, CG,
                       din);
    NIR_CHECK_LAUNCH("col2im_rows_kernel");
    return 0;
}
