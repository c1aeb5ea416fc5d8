// MatchTensor (neuroir/rankers/mtensor.py:62-131,144-158): embed -> Linear(300->40) -> BiLSTM(q: 2x15, d: 2x70)
// -> Linear(->50) x2 -> match tensor T[c,i,j] = Pq[i,c]*Pd[j,c] (+ exact-match channel) -> 3 convs + ReLU
// -> 1x1 conv -> global max -> Linear(20->1).
//
// What the reference materialises and this path does not: three [B*N,QL,DL,50] broadcast copies and the
// [B*N,51,QL,DL] match tensor (`cat` is its top CPU cost).  Here the match tensor never exists:
//   conv_k[f,i,j] = b + sum_{dj,c} ( sum_di W_k[f,c,di,dj] * Pq[i+di-1,c] ) * Pd[j+dj-pw,c]  + exact-match taps
// The inner parenthesis `U` depends on the QUERY only: it is folded once per query (mt_fold_kernel) and shared
// by all N candidates, cutting the per-pair conv work 3x (the di sum) -- U is read through the scalar cache as
// wave-uniform SGPR operands, the document projections sit transposed in LDS (one lane per doc position), and
// ReLU / 1x1 conv / global max-pool / output Linear are fused in registers + wave shuffles.
#include "common.hpp"
#include <stdlib.h>
#include <algorithm>

namespace nir {

int launch_linear(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                  int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                  int64_t ldc, int64_t M, int N, int K, int act, hipStream_t st);
int launch_bilstm_fused(const float* x, int I, const float* wih, const float* bih, const float* bhh, const int64_t* lens,
                        const float* whh, const float* h0, const float* c0, float* out, float* hn, float* cn, int64_t M,
                        int T, int H, int ND, hipStream_t st);

int launch_bilstm_folded(const void* pt, int pt_dtype, const int64_t* ids, const int64_t* lens, const float* whh, float* out,
                         int* err, int64_t M, int64_t V, int T, int H, int ND, hipStream_t st, int out_f16 = 0, const void* whh_frag = nullptr);

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CP = 56;     // channels padded 50 -> 56 so an 8-wide K group never straddles a conv tap (dj)
constexpr int NFC = 6;     // filters per conv (hyparam.py:101)
constexpr int MFC = 20;    // match_filter_size
constexpr int JTMAX = 64;  // document positions per chunk: 64 (2 MFMA column tiles) or, when LDS is short (long documents), 32
constexpr int MTGMAX = 3;  // row tiles (32 rows) of the folded operand per pass: 96 rows = 16 whole query positions x 6 filters

struct MtHeadW {
    const float* conv_w[3];
    const float* conv_b[3];
    const float *alpha, *cw, *cb, *ow, *ob;
    int C;
    unsigned long long* dbg;
    // fused channel projection of the documents (H2 head only; dpf == NULL: Pd is read from memory): encoder output hd [pairs*DL, HD2],
    // document_projection as fp16 term planes in MFMA B-fragment order [ceil(HD2/32)][4 column tiles][2 terms][64 lanes][8], bias [C]
    const float* hd;
    const void* dpf;
    const float* dpb;
    int HD2;
};

__host__ __device__ inline int mt_kp(int k) { return (3 + 2 * k) * CP; }          // padded K of conv k: 168, 280, 392
__host__ __device__ inline int mt_koff(int k) { return k == 0 ? 0 : (k == 1 ? 3 * CP : 8 * CP); }
constexpr int KTOT = 15 * CP;                                                    // 840

// A operand of the interaction GEMM, folded once per query and shared by its N candidates:
//   U[b][mt][k][r][kk],  r = i*6 + f (row inside the 32-row tile mt),  kk = dj*CP + c
//   U = sum_di W_k[f][c][di][dj] * Pq[b][i+di-1][c]      (zero for c >= C, rows >= 6*QL)
// grid (B, 3 convs, FOLD_Z element slices), block 256
constexpr int FOLD_Z = 4;
// fp16 two-term form (H2) of the same operand, for the v_mfma_f32_16x16x32_f16 interaction GEMM: every conv tap is padded to 64 channels
// (two 32-wide k-steps) and stored as MFMA A-fragments,
//   Uh[b][tap 0..14][half 0..1][row tile 0..2MT-1][term 0..1][lane 64][8],   lane = 16 * (c % 32 / 8) + row % 16,
// x = h1 + 2^-11 h2' with h1 = fp16_rtz(x), h2' = fp16(2^11 (x - h1))  (requires |U| < 2^15: host-checked `bounded`).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CH = 64;     // channels per tap in the H2 form
__host__ __device__ inline int mt_tapoff(int k) { return k == 0 ? 0 : (k == 1 ? 3 : 8); }
__host__ __device__ inline size_t mt_uh_halves(int MT) { return (size_t)15 * 2 * (2 * MT) * 2 * 64 * 8; }

template <bool H2>
__global__ __launch_bounds__(256) void mt_fold_kernel(const float* __restrict__ pq, MtHeadW w, int QL, int MT, float* __restrict__ U) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];   // Pq[QL][C] | W_k[NF][C+1][3][kw]
    const int b = blockIdx.x, k = blockIdx.y, C = w.C;
    const int kw = 3 + 2 * k, Kp = mt_kp(k);
    const int nw = NFC * (C + 1) * 3 * kw;
    float* pqs = fsm;
    float* wks = fsm + ((QL * C + 3) & ~3);
    {   // batched staging: all global loads of a batch are in flight before the first LDS write
        const float* pqb = pq + (int64_t)b * QL * C;
        const float* wk = w.conv_w[k];
        for (int base = 0; base < nw; base += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * 256 + threadIdx.x;
                v[u] = e < nw ? wk[e] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * 256 + threadIdx.x;
                if (e < nw) wks[e] = v[u];
            }
        }
        for (int e = threadIdx.x; e < QL * C; e += 256) pqs[e] = pqb[e];
    }
    __syncthreads();
    const int rows = MT * 32;
    if (H2) {
        _Float16* uh = reinterpret_cast<_Float16*>(U) + (int64_t)b * mt_uh_halves(MT);
        for (int e = threadIdx.x + 256 * blockIdx.z; e < rows * kw * (CH / 2); e += 256 * FOLD_Z) {   // two channels per thread
            const int r = e / (kw * (CH / 2)), rem = e - r * (kw * (CH / 2));
            const int dj = rem / (CH / 2), c = 2 * (rem - dj * (CH / 2));
            const int i = r / NFC, f = r - i * NFC;
            float a0 = 0.f, a1 = 0.f;
            if (i < QL) {
#pragma unroll
                for (int di = 0; di < 3; ++di) {
                    const int ii = i + di - 1;
                    if (ii >= 0 && ii < QL) {
                        if (c < C) a0 = fmaf(wks[((f * (C + 1) + c) * 3 + di) * kw + dj], pqs[ii * C + c], a0);
                        if (c + 1 < C) a1 = fmaf(wks[((f * (C + 1) + c + 1) * 3 + di) * kw + dj], pqs[ii * C + c + 1], a1);
                    }
                }
            }
            const fp16x2_t h1 = __builtin_amdgcn_cvt_pkrtz(a0, a1);
            const fp16x2_t h2 = __builtin_amdgcn_cvt_pkrtz((a0 - (float)h1[0]) * 2048.0f, (a1 - (float)h1[1]) * 2048.0f);
            const int tk = mt_tapoff(k) + dj, half = c >> 5, kg = (c >> 3) & 3, e8 = c & 7, rt = r >> 4, row16 = r & 15;
            _Float16* d = uh + ((((int64_t)(tk * 2 + half) * (2 * MT) + rt) * 2) * 64 + kg * 16 + row16) * 8 + e8;
            *reinterpret_cast<unsigned*>(d) = __builtin_bit_cast(unsigned, h1);
            *reinterpret_cast<unsigned*>(d + 64 * 8) = __builtin_bit_cast(unsigned, h2);
        }
        return;
    }
    float* ub = U + (int64_t)b * rows * KTOT;   // per query: [mt][k-slab][32][Kp] laid out as consecutive slabs
    for (int e = threadIdx.x + 256 * blockIdx.z; e < rows * Kp; e += 256 * FOLD_Z) {
        const int r = e / Kp, kk = e - r * Kp;
        const int dj = kk / CP, c = kk - dj * CP;
        const int i = r / NFC, f = r - i * NFC;
        float acc = 0.f;
        if (c < C && i < QL) {
#pragma unroll
            for (int di = 0; di < 3; ++di) {
                const int ii = i + di - 1;
                if (ii >= 0 && ii < QL) acc = fmaf(wks[((f * (C + 1) + c) * 3 + di) * kw + dj], pqs[ii * C + c], acc);
            }
        }
        const int mt = r >> 5, rr = r & 31;
        ub[(int64_t)mt * 32 * KTOT + (int64_t)mt_koff(k) * 32 + (int64_t)rr * Kp + kk] = acc;
    }
}

// One workgroup (4 waves) per (query, candidate) pair.  Per 64-position chunk of the document:
//   phase 1 (MFMA): for each conv k and 32-column tile, D[32 rows x 32 cols] += U[rows][kk] * Pd[j+dj-pw][c]
//            with B read straight out of the transposed, zero-haloed LDS image PdT[c][j] (an im2col view: the K
//            index kk = dj*CP + c only shifts the column), A streamed from L2 as one float4 per 4 MFMAs;
//   phase 2 (VALU): one lane per (query position, doc position): + bias + exact-match taps, ReLU, 1x1 conv,
//            running max-pool;  finally max over lanes/waves and the output Linear.
// Queries longer than 16 tokens are processed in passes of 16 query positions (MTGMAX row tiles) per chunk, so the conv-output tile Y
// -- and with it the LDS footprint -- does not grow with the query length (mtensor.py:100-131 takes any max_query_len; scripts/ranker.sh
// pads to 20 x 200).
// dynamic LDS: PdT[CP][DLP] | Y[3][min(MT,3)*32][JT+1] | small weights | dids[DL]
// H2 = true: phase 1 on v_mfma_f32_16x16x32_f16 with both operands in the two-term fp16 form -- 720 MFMAs of 16 cycles per pair instead
// of 840 of 64 (fp32 32x32x2): U arrives as A-fragments from mt_fold_kernel<true>, the document projections are split into two fp16
// planes Pdh[term][position + 3][64 channels (+8 pad)] when they are staged (position-major: a B fragment = 16 bytes of one row).
constexpr int CPH = CH + 8;     // halves per position row of a Pdh plane (144 B pitch: 16-byte reads of 16 consecutive rows hit 16 bank groups)
template <bool H2>
__global__ __launch_bounds__(256, 3) void mt_head_kernel(const float* __restrict__ pd, const float* __restrict__ U,
                                                      const int64_t* __restrict__ q_ids, const int64_t* __restrict__ d_ids,
                                                      MtHeadW w, int B, int N, int QL, int DL, int MT, int jts,
                                                      float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = w.C;
    const int JT = 1 << jts, NCT = JT >> 5;        // chunk width (64 or 32 positions) and its 32-column tiles
    const int nchunk = (DL + JT - 1) >> jts;
    const int DLP = nchunk * JT + 9;               // halo: 3 left, >= 3 right; odd pitch -> the transposed prologue
                                                   // writes pdt[c*DLP + j] (consecutive c per lane) are conflict-free
    const int MTG = MT < MTGMAX ? MT : MTGMAX;     // row tiles per pass
    const int ngroup = (MT + MTG - 1) / MTG;
    const int rows = MTG * 32;                     // rows of Y
    const int YLD = JT + 1;
    float* pdt = smem;                             // [CP][DLP]   (H2: Pdh[2 terms][DLH][CPH] halves in the same region)
    const int DLH = nchunk * JT + 6;               // positions + 3 halo rows on each side
    _Float16* pdh = reinterpret_cast<_Float16*>(smem);
    const int pd_floats = H2 ? (2 * DLH * CPH + 1) / 2 : CP * DLP;
    float* Y = pdt + ((pd_floats + 3) & ~3);       // [3][rows][YLD]
    float* wsm = Y + 3 * rows * YLD;               // cw[20*18] | cb[20] | bias[18] | wm[3 convs: 6*3*kw] (270)
    float* cw_s = wsm;
    float* cb_s = cw_s + MFC * 3 * NFC;
    float* bias_s = cb_s + MFC;
    float* wm_s = bias_s + 3 * NFC;
    float* ow_s = wm_s + 270;                      // output Linear weights [MFC] + bias
    int64_t* dids = (int64_t*)(wsm + ((MFC * 3 * NFC + MFC + 3 * NFC + 270 + MFC + 2 + 1) & ~1));   // [DL]
    int64_t* qsh = dids + DL;                                                             // [QL]
    __shared__ float wmax[4][MFC];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware order: the N candidates of one query reuse the same folded operand U (107 KB); give them block ids
    // 8 apart so they run on ONE XCD back to back and U is served by that L2 instead of the fabric
    // (round-1 PMC: FETCH_SIZE 15.7 MB per launch = U re-fetched for every pair).
    const int bid = blockIdx.x;
    const int tq = bid >> 3;
    const int b = (tq / N) * 8 + (bid & 7);
    if (b >= B) return;
    const int64_t pair = (int64_t)b * N + (tq % N);

#define MT_STAMP(slot) do { if (w.dbg && blockIdx.x == 0 && lane == 0) w.dbg[wave * 8 + (slot)] = clock64(); } while (0)
    MT_STAMP(0);
    if (w.dbg && tid == 0) { w.dbg[64 + 4 * blockIdx.x] = wall_clock64(); w.dbg[64 + 4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg(63492 /*HW_REG_HW_ID, 32 bits*/); }
    // ---- prologue: every global load is issued before anything waits on it (batched), then the LDS images are built
    const float* pdm = pd + pair * DL * C;
    const int npd = DL * C;
    constexpr int PB = 8;                                  // Pd elements (H2: pairs of elements) per thread per batch
    float pv[PB], pv2[PB];
    const bool fuse_proj = H2 && w.dpf != nullptr;
#pragma unroll
    for (int u = 0; u < PB; ++u) {
        const int e = u * 256 + tid;
        if (fuse_proj) {
            pv[u] = pv2[u] = 0.f;
        } else if (H2) {                                   // C even: a pair never straddles two positions; clamped, masked below
            const int e2 = 2 * e < npd ? 2 * e : 0;
            pv[u] = pdm[e2];
            pv2[u] = pdm[e2 + 1];
        } else {
            pv[u] = e < npd ? pdm[e] : 0.f;
        }
    }
    float cwv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = u * 256 + tid;
        cwv[u] = e < MFC * 3 * NFC ? w.cw[e] : 0.f;
    }
    const float cbv = tid < MFC ? w.cb[tid] : 0.f;
    const float owv = tid < MFC ? w.ow[tid] : (tid == MFC ? w.ob[0] : 0.f);
    const float biasv = tid < 3 * NFC ? w.conv_b[tid / NFC][tid % NFC] : 0.f;
    float wmv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {   // exact-match channel weights W_k[f][C][di][dj] -> wm_s[k-slab][(di*kw + dj)*6 + f]
        const int e = u * 256 + tid;
        wmv[u] = 0.f;
        if (e < 270) {
            const int k = e < 54 ? 0 : (e < 144 ? 1 : 2);
            const int kw = 3 + 2 * k, r = e - (k == 0 ? 0 : (k == 1 ? 54 : 144));
            const int f = r % NFC, t = r / NFC, dj = t % kw, di = t / kw;
            wmv[u] = w.conv_w[k][((f * (C + 1) + C) * 3 + di) * kw + dj];
        }
    }
    const float alpha = w.alpha[0];
    const int64_t did0 = tid < DL ? d_ids[pair * DL + tid] : 0;
    const int64_t qid0 = tid < QL ? q_ids[(int64_t)b * QL + tid] : 0;

    if (H2) {
        for (int e = tid; e < (2 * DLH * CPH + 1) / 2; e += 256) pdt[e] = 0.f;       // zero halo rows and padded channels of both planes
        __syncthreads();
        auto put = [&](int e, float x0, float x1) {            // elements 2e, 2e+1 -> the two term planes
            const int j = (2 * e) / C, c = 2 * e - j * C;
            const fp16x2_t h1 = __builtin_amdgcn_cvt_pkrtz(x0, x1);
            const fp16x2_t h2 = __builtin_amdgcn_cvt_pkrtz((x0 - (float)h1[0]) * 2048.0f, (x1 - (float)h1[1]) * 2048.0f);
            _Float16* d = pdh + (j + 3) * CPH + c;
            *reinterpret_cast<unsigned*>(d) = __builtin_bit_cast(unsigned, h1);
            *reinterpret_cast<unsigned*>(d + DLH * CPH) = __builtin_bit_cast(unsigned, h2);
        };
        if (fuse_proj) {
            // Pd = hd Wd^T + b computed here (replaces the [M*DL,140] x [140,50] projection launch and the Pd round trip): wave = 16
            // channels, four 16-position row tiles per 64-position block; A fragments straight from the fp32 encoder output (split
            // in registers), B fragments pre-split by the host.  Padded document positions give the bias, as in the reference.
            const int g4 = lane >> 4, c16 = lane & 15, c = 16 * wave + c16, HD2 = w.HD2, SP = (HD2 + 31) / 32;
            const _Float16* wfp = reinterpret_cast<const _Float16*>(w.dpf) + ((int64_t)wave * 2 * 64 + lane) * 8;
            const float* hdm = w.hd + pair * DL * (int64_t)HD2;
            const float bias = c < C ? w.dpb[c] : 0.f;
            const int nblk = (DL + 63) >> 6;
            for (int jt = 0; jt < nblk; ++jt) {
                f32x4 pa[4], px[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { pa[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; px[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                for (int sp = 0; sp < SP; ++sp) {
                    const f16x8 wf0 = *reinterpret_cast<const f16x8*>(wfp + (int64_t)sp * (4 * 2 * 64 * 8));
                    const f16x8 wf1 = *reinterpret_cast<const f16x8*>(wfp + (int64_t)sp * (4 * 2 * 64 * 8) + 512);
                    // the lane's 8 k values start at k0; 2Hd is a multiple of 4, not of 8: each float4 is loaded from a clamped address and
                    // multiplied by a 0/1 mask (no predicated loads)
                    const int k0 = 32 * sp + 8 * g4;
                    const int ka = k0 + 4 <= HD2 ? k0 : HD2 - 4, kb = k0 + 8 <= HD2 ? k0 + 4 : HD2 - 4;
                    const float ma = k0 + 4 <= HD2 ? 1.f : 0.f, mb = k0 + 8 <= HD2 ? 1.f : 0.f;
                    float4 x0[4], x1[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int j = jt * 64 + 16 * i + c16;
                        j = j < DL ? j : DL - 1;
                        const float* hp = hdm + (int64_t)j * HD2;
                        x0[i] = *reinterpret_cast<const float4*>(hp + ka);
                        x1[i] = *reinterpret_cast<const float4*>(hp + kb);
                        x0[i].x *= ma; x0[i].y *= ma; x0[i].z *= ma; x0[i].w *= ma;
                        x1[i].x *= mb; x1[i].y *= mb; x1[i].z *= mb; x1[i].w *= mb;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const fp16x2_t a0 = __builtin_amdgcn_cvt_pkrtz(x0[i].x, x0[i].y), a1 = __builtin_amdgcn_cvt_pkrtz(x0[i].z, x0[i].w);
                        const fp16x2_t a2 = __builtin_amdgcn_cvt_pkrtz(x1[i].x, x1[i].y), a3 = __builtin_amdgcn_cvt_pkrtz(x1[i].z, x1[i].w);
                        const fp16x2_t b0 = __builtin_amdgcn_cvt_pkrtz((x0[i].x - (float)a0[0]) * 2048.0f, (x0[i].y - (float)a0[1]) * 2048.0f);
                        const fp16x2_t b1 = __builtin_amdgcn_cvt_pkrtz((x0[i].z - (float)a1[0]) * 2048.0f, (x0[i].w - (float)a1[1]) * 2048.0f);
                        const fp16x2_t b2 = __builtin_amdgcn_cvt_pkrtz((x1[i].x - (float)a2[0]) * 2048.0f, (x1[i].y - (float)a2[1]) * 2048.0f);
                        const fp16x2_t b3 = __builtin_amdgcn_cvt_pkrtz((x1[i].z - (float)a3[0]) * 2048.0f, (x1[i].w - (float)a3[1]) * 2048.0f);
                        const f16x8 h1 = __builtin_bit_cast(f16x8, make_uint4(__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1),
                                                                                    __builtin_bit_cast(unsigned, a2), __builtin_bit_cast(unsigned, a3)));
                        const f16x8 h2 = __builtin_bit_cast(f16x8, make_uint4(__builtin_bit_cast(unsigned, b0), __builtin_bit_cast(unsigned, b1),
                                                                                    __builtin_bit_cast(unsigned, b2), __builtin_bit_cast(unsigned, b3)));
                        px[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h2, wf0, px[i], 0, 0, 0);
                        px[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, wf1, px[i], 0, 0, 0);
                        pa[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, wf0, pa[i], 0, 0, 0);
                    }
                }
                if (c < C) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int j = jt * 64 + 16 * i + 4 * g4 + r;
                            if (j < DL) {
                                const float v = fmaf(px[i][r], 1.0f / 2048.0f, pa[i][r]) + bias;
                                const _Float16 a = (_Float16)__builtin_amdgcn_cvt_pkrtz(v, 0.f)[0];
                                _Float16* d = pdh + (j + 3) * CPH + c;
                                d[0] = a;
                                d[DLH * CPH] = (_Float16)((v - (float)a) * 2048.0f);
                            }
                        }
                }
            }
        } else {
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int e = u * 256 + tid;
            if (2 * e < npd) put(e, pv[u], pv2[u]);
        }
        for (int e = PB * 256 + tid; 2 * e < npd; e += 256) put(e, pdm[2 * e], pdm[2 * e + 1]);   // long documents
        }
    } else {
    for (int e = tid; e < CP * DLP; e += 256) pdt[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PB; ++u) {
        const int e = u * 256 + tid;
        if (e < npd) {
            const int j = e / C, c = e - j * C;
            pdt[c * DLP + j + 3] = pv[u];
        }
    }
    for (int e = PB * 256 + tid; e < npd; e += 256) {      // long documents: remaining elements
        const int j = e / C, c = e - j * C;
        pdt[c * DLP + j + 3] = pdm[e];
    }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = u * 256 + tid;
        if (e < MFC * 3 * NFC) cw_s[e] = cwv[u];
        if (e < 270) wm_s[e] = alpha * wmv[u];
    }
    if (tid < MFC) cb_s[tid] = cbv;
    if (tid <= MFC) ow_s[tid] = owv;
    if (tid < 3 * NFC) bias_s[tid] = biasv;
    if (tid < DL) dids[tid] = did0;
    for (int j = 256 + tid; j < DL; j += 256) dids[j] = d_ids[pair * DL + j];
    if (tid < QL) qsh[tid] = qid0;
    __syncthreads();
    MT_STAMP(1);

    float zmax[MFC];
#pragma unroll
    for (int g = 0; g < MFC; ++g) zmax[g] = -INFINITY;

    const float* ub = U + (int64_t)b * (MT * 32) * KTOT;      // per query: all MT row tiles of the folded operand
    const int g2 = lane >> 5, col = lane & 31;
    for (int ch = 0; ch < nchunk; ++ch)
    for (int grp = 0; grp < ngroup; ++grp) {
        const int j0 = ch * JT;
        const int mt0 = grp * MTG, mtn = MT - mt0 < MTG ? MT - mt0 : MTG;     // this pass: row tiles [mt0, mt0 + mtn)
        const int i0 = ngroup > 1 ? 16 * grp : 0;                            // = query positions [i0, i0 + nqi)
        const int nqi = QL - i0 < (ngroup > 1 ? 16 : QL) ? QL - i0 : (ngroup > 1 ? 16 : QL);
        // ---- phase 1: tasks (conv k, row tile mt, column tile nt), heaviest first, each handed to the least-loaded wave
        // (weights 7:5:3 taps).  Round-robin gave waves 0/1 a 7+3 pair and waves 2/3 a single 5 at MT = 1 (measured
        // 56 K vs 30 K cycles before the barrier); this gives 7, 7, 5+3, 5+3.
        const int ntask = 3 * mtn * NCT;
        int wload[4] = {0, 0, 0, 0};
        for (int task = 0; task < ntask; ++task) {
            const int k = 2 - task / (mtn * NCT);
            int wsel = 0;
#pragma unroll
            for (int x = 1; x < 4; ++x) wsel = wload[x] < wload[wsel] ? x : wsel;
#pragma unroll
            for (int x = 0; x < 4; ++x) wload[x] += x == wsel ? 3 + 2 * k : 0;
            if (wsel != wave) continue;
            const int rem = task % (mtn * NCT);
            const int mtl = NCT == 2 ? rem >> 1 : rem, nt = NCT == 2 ? rem & 1 : 0, mt = mt0 + mtl;
            if (H2) {
                // 32 rows x 32 positions of conv k: 2 x 2 tiles of 16 x 16, per tap two 32-channel k-steps, three MFMAs per product
                // block (cross terms into acx, scaled by 2^-11 at the end); the A fragments of the next k-step are in flight (L2)
                // while the 12 MFMAs of the current one run
                const int kw2 = 3 + 2 * k, pw2 = k + 1, g4 = lane >> 4, c16 = lane & 15;
                const _Float16* ua = reinterpret_cast<const _Float16*>(U) + (int64_t)b * mt_uh_halves(MT) +
                                     ((((int64_t)mt_tapoff(k) * 2) * (2 * MT) + 2 * mt) * 2 * 64 + lane) * 8;
                const int64_t astep = (int64_t)(2 * MT) * 2 * 64 * 8;                 // one (tap, half) k-step
                const _Float16* bb = pdh + (j0 + nt * 32 + c16 - pw2 + 3) * CPH + 8 * g4;
                f32x4 acc[2][2], acx[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acx[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                // Two A-fragment sets in ping-pong (the step count 2 kw2 is even): a step first pins the hand-over of its own set (loaded
                // during the previous step), then issues the loads of the other set for the next step, then runs its 12 MFMAs.  With one
                // set copied at the top of each step the compiler moved the copies to where the old values died -- two MFMAs into the
                // step, behind an s_waitcnt for the loads issued a few instructions earlier: one exposed L2 round trip per k-step.
                f16x8 fa[2][2], fb[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t) fa[i][t] = *reinterpret_cast<const f16x8*>(ua + (i * 2 + t) * 512);
                const int nstep = 2 * kw2;
#define MT_H2_STEP(ST, CUR, NXT)                                                                                              \
                {                                                                                                             \
                    f16x8 bf[2][2];                                                                                           \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
                        _Pragma("unroll") for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(CUR[i][t]));                     \
                    const int sn = (ST) + 1 < nstep ? (ST) + 1 : (ST);                                                        \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
                        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                         \
                            NXT[i][t] = *reinterpret_cast<const f16x8*>(ua + sn * astep + (i * 2 + t) * 512);                 \
                    const _Float16* bp = bb + ((ST) >> 1) * CPH + 32 * ((ST) & 1);                                            \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                         \
                            bf[j][t] = *reinterpret_cast<const f16x8*>(bp + t * DLH * CPH + j * 16 * CPH);                    \
                    __builtin_amdgcn_sched_barrier(0);                                                                        \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
                            acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(CUR[i][1], bf[j][0], acx[i][j], 0, 0, 0);      \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
                            acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(CUR[i][0], bf[j][1], acx[i][j], 0, 0, 0);      \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(CUR[i][0], bf[j][0], acc[i][j], 0, 0, 0);      \
                    __builtin_amdgcn_sched_barrier(0);                                                                        \
                }
                for (int st = 0; st < nstep; st += 2) {
                    MT_H2_STEP(st, fa, fb)
                    MT_H2_STEP(st + 1, fb, fa)
                }
#undef MT_H2_STEP
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float* yk = Y + ((int64_t)k * rows + mtl * 32 + 16 * i + 4 * g4) * YLD + nt * 32 + 16 * j + c16;
#pragma unroll
                        for (int r = 0; r < 4; ++r) yk[r * YLD] = fmaf(acx[i][j][r], 1.0f / 2048.0f, acc[i][j][r]);
                    }
                continue;
            }
            const int kw = 3 + 2 * k, pw = k + 1, Kp = kw * CP;
            const float* arow = ub + (int64_t)mt * 32 * KTOT + (int64_t)mt_koff(k) * 32 + (int64_t)col * Kp + 4 * g2;
            const float* bcol = pdt + (4 * g2) * DLP + (j0 + nt * 32 + col) - pw + 3;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // A fragments (U, served by L2) are prefetched one tap ahead: 7 float4 in flight while the 28 MFMAs of
            // the current tap run (round-1 PMC: 66 % of wave cycles were s_waitcnt on these loads).
            constexpr int NG = CP / 8;
            float4 an[NG];
#pragma unroll
            for (int u = 0; u < NG; ++u) an[u] = *reinterpret_cast<const float4*>(arow + u * 8);
            for (int dj = 0; dj < kw; ++dj) {
                float4 ac[NG];
#pragma unroll
                for (int u = 0; u < NG; ++u) ac[u] = an[u];
                if (dj + 1 < kw) {
#pragma unroll
                    for (int u = 0; u < NG; ++u) an[u] = *reinterpret_cast<const float4*>(arow + (dj + 1) * CP + u * 8);
                }
                // B operands (LDS) one channel group ahead of their MFMAs: left to itself the compiler issued each
                // ds_read right before its MFMA behind an lgkmcnt(0), i.e. one exposed LDS round trip per 64-cycle MFMA
                const float* bp0 = bcol + dj;
                float bn0 = bp0[0], bn1 = bp0[DLP], bn2 = bp0[2 * DLP], bn3 = bp0[3 * DLP];
#pragma unroll
                for (int u = 0; u < NG; ++u) {
                    const float b0 = bn0, b1 = bn1, b2 = bn2, b3 = bn3;
                    if (u + 1 < NG) {
                        const float* bp = bcol + ((u + 1) * 8) * DLP + dj;
                        bn0 = bp[0]; bn1 = bp[DLP]; bn2 = bp[2 * DLP]; bn3 = bp[3 * DLP];
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs (the scheduler sinks them)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u].x, b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u].y, b1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u].z, b2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u].w, b3, acc, 0, 0, 0);
                }
            }
            float* yk = Y + ((int64_t)k * rows + mtl * 32) * YLD + nt * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) yk[((r & 3) + 8 * (r >> 2) + 4 * g2) * YLD] = acc[r];
        }
        MT_STAMP(2);
        __syncthreads();
        MT_STAMP(3);
        // ---- phase 2: positions (i, j) of this chunk
        typedef const float __attribute__((address_space(4))) * const_fp;      // constant address space: uniform -> s_load
        const const_fp cw_c = (const_fp)(uintptr_t)w.cw, cb_c = (const_fp)(uintptr_t)w.cb;
        for (int pos = tid; pos < nqi * JT; pos += 256) {
            const int il = pos >> jts, jl = pos & (JT - 1), i = i0 + il, j = j0 + jl;
            float v[3 * NFC];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int f = 0; f < NFC; ++f) v[k * NFC + f] = Y[((int64_t)k * rows + il * NFC + f) * YLD + jl] + bias_s[k * NFC + f];
            // exact-match channel: alpha * [q_id == d_id], PAD==PAD counts (mtensor.py:144-158).  The 7 document ids
            // around j are read once into registers (a rolled loop of 21 dependent LDS reads cost ~2 K cycles per
            // position); hits are rare, so the weight adds stay in a rolled, rarely taken branch
            int64_t dwin[7];
#pragma unroll
            for (int dd = 0; dd < 7; ++dd) {
                const int jj = j + dd - 3;
                dwin[dd] = (jj >= 0 && jj < DL) ? dids[jj] : (int64_t)-1 - (int64_t)dd;   // sentinel < 0: never equals an id
            }
#pragma unroll
            for (int di = 0; di < 3; ++di) {
                const int ii = i + di - 1;
                const int64_t qid = (ii >= 0 && ii < QL) ? qsh[ii] : (int64_t)-100;
                unsigned hit = 0;
#pragma unroll
                for (int dd = 0; dd < 7; ++dd) hit |= (dwin[dd] == qid ? 1u : 0u) << dd;
                if (hit) {
#pragma unroll 1
                    for (int dd = 0; dd < 7; ++dd) {
                        if (!((hit >> dd) & 1u)) continue;
#pragma unroll 1
                        for (int k = 0; k < 3; ++k) {
                            const int kw = 3 + 2 * k, pw = k + 1, dj = dd - 3 + pw;
                            if (dj < 0 || dj >= kw) continue;
                            const float* wm = wm_s + (k == 0 ? 0 : (k == 1 ? 54 : 144)) + (di * kw + dj) * NFC;
#pragma unroll
                            for (int f = 0; f < NFC; ++f) v[k * NFC + f] += wm[f];
                        }
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < 3 * NFC; ++f) v[f] = fmaxf(v[f], 0.f);
            if (j < DL) {
#pragma unroll
                for (int g = 0; g < MFC; ++g) {
                    // 1x1 conv weights straight from global memory: the index is wave-uniform, so they arrive through
                    // the scalar cache as SGPR operands of the FMAs (as LDS reads every one of the 380 values cost an
                    // exposed ds_read round trip per position)
                    float z = cb_c[g];
#pragma unroll
                    for (int f = 0; f < 3 * NFC; ++f) z = fmaf(cw_c[g * 3 * NFC + f], v[f], z);
                    zmax[g] = fmaxf(zmax[g], z);
                }
            }
        }
        MT_STAMP(4);
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < MFC; ++g) {
        float v = wave_max(zmax[g]);
        if (lane == 0) wmax[wave][g] = v;
    }
    __syncthreads();
    if (wave == 0) {   // global max over the 4 waves, then the output Linear as one wave reduction
        float t = 0.f;
        if (lane < MFC) t = ow_s[lane] * fmaxf(fmaxf(wmax[0][lane], wmax[1][lane]), fmaxf(wmax[2][lane], wmax[3][lane]));
        t = wave_sum(t);
        if (lane == 0) scores[pair] = t + ow_s[MFC];
    }
    MT_STAMP(5);
    if (w.dbg && tid == 0) w.dbg[64 + 4 * blockIdx.x + 1] = wall_clock64();
}

static size_t mt_head_lds(int QL, int DL, int MT, bool h2, int jts) {
    const int JT = 1 << jts, MTG = MT < MTGMAX ? MT : MTGMAX;
    const int nchunk = (DL + JT - 1) / JT, DLP = nchunk * JT + 9, DLH = nchunk * JT + 6;
    const size_t pdf = ((h2 ? (size_t)(2 * DLH * CPH + 1) / 2 : (size_t)CP * DLP) + 3) & ~(size_t)3;
    size_t fl = pdf + (size_t)3 * MTG * 32 * (JT + 1) + ((MFC * 3 * NFC + MFC + 3 * NFC + 270 + MFC + 2 + 1) & ~1);
    return fl * 4 + (size_t)(DL + QL) * 8;
}

struct MtPlan {
    float *xq, *xd, *hq, *hd, *pq, *pd, *U;
    size_t bytes;
};

static MtPlan mt_plan(void* ws, size_t cap, int B, int N, int QL, int DL, const nir_matchtensor_weights* w) {
    Workspace a(ws, cap);
    const size_t Mq = (size_t)B * QL, Md = (size_t)B * N * DL;
    MtPlan p;
    p.xq = a.take<float>(Mq * w->F);
    p.xd = a.take<float>(Md * w->F);
    p.hq = a.take<float>(Mq * 2 * w->Hq);
    p.hd = a.take<float>(Md * 2 * w->Hd);
    p.pq = a.take<float>(Mq * w->C);
    p.pd = a.take<float>(Md * w->C);
    // fp32 form: [B][MT*32 rows][KTOT]; fp16 two-term fragment form: mt_uh_halves(MT) halves per query (larger: 64-channel taps)
    p.U = a.take<float>((size_t)B * std::max((size_t)((6 * QL + 31) / 32) * 32 * KTOT, (mt_uh_halves((6 * QL + 31) / 32) + 1) / 2));
    p.bytes = align_up(a.off, 256);
    return p;
}

}  // namespace nir

extern "C" size_t nir_matchtensor_workspace_bytes(int B, int N, int QL, int DL, const nir_matchtensor_weights* w) {
    if (!w || B < 0 || N <= 0 || QL <= 0 || DL <= 0) return 0;
    return nir::mt_plan(nullptr, 0, B, N, QL, DL, w).bytes;
}

namespace nir {
static int matchtensor_impl(const int64_t* q_ids, const int64_t* q_len, const int64_t* d_ids, const int64_t* d_len, int B, int N, int QL,
                            int DL, const float* table, int64_t V, int E, const void* fold_q, const void* fold_d, int fold_dtype,
                            const nir_matchtensor_weights* w, void* workspace, size_t workspace_bytes, float* scores, float* enc_q,
                            float* enc_d, float* proj_q, float* proj_d, int* err_flag, hipStream_t st, const float* given_q = nullptr,
                            const float* given_d = nullptr) {
    const bool folded = fold_q != nullptr && fold_d != nullptr;
    const bool given = given_q != nullptr && given_d != nullptr;      // encoder states supplied by the caller (any RNNEncoder configuration)
    NIR_REQUIRE(q_ids && d_ids && (given || (q_len && d_len)) && (table || folded || given) && w && scores, "match_tensor: null pointer");
    NIR_REQUIRE(B >= 0 && N > 0 && QL > 0 && DL > 0 && V > 0 && E > 0, "match_tensor: bad dims");
    NIR_REQUIRE(w->NF == NFC && w->MF == MFC, "match_tensor: nfilters=%d match_filter_size=%d unsupported (6, 20)", w->NF, w->MF);
    NIR_REQUIRE(w->C >= 1 && w->C <= CP - 6, "match_tensor: nchannels=%d unsupported (<= 50)", w->C);
    NIR_REQUIRE(given || (nir_bilstm_supported(w->Hq) && nir_bilstm_supported(w->Hd)), "match_tensor: hidden size unsupported");
    NIR_REQUIRE(given || (w->F >= 1 && w->F <= 64), "match_tensor: featsize %d unsupported (1..64)", w->F);
    NIR_REQUIRE(QL <= 256, "match_tensor: query length %d > 256 unsupported", QL);
    if (B == 0) return 0;
    MtPlan p = mt_plan(workspace, workspace_bytes, B, N, QL, DL, w);
    if (!workspace || p.bytes > workspace_bytes) {
        set_error("match_tensor: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
        return NIR_ERR_WORKSPACE;
    }
    const int64_t Mq = (int64_t)B * QL, Md = (int64_t)B * N * DL;
    float* hq = given ? const_cast<float*>(given_q) : (enc_q ? enc_q : p.hq);
    float* hd = given ? const_cast<float*>(given_d) : (enc_d ? enc_d : p.hd);
    float* pq = proj_q ? proj_q : p.pq;
    float* pd = proj_d ? proj_d : p.pd;
    MtHeadW hw;
    hw.conv_w[0] = w->conv1_w; hw.conv_w[1] = w->conv2_w; hw.conv_w[2] = w->conv3_w;
    hw.conv_b[0] = w->conv1_b; hw.conv_b[1] = w->conv2_b; hw.conv_b[2] = w->conv3_b;
    hw.alpha = w->alpha; hw.cw = w->conv_w; hw.cb = w->conv_b; hw.ow = w->out_w; hw.ob = w->out_b;
    hw.C = w->C;
    hw.dbg = g_debug_buf;
    hw.hd = nullptr; hw.dpf = nullptr; hw.dpb = nullptr; hw.HD2 = 2 * w->Hd;
    const int MT = (NFC * QL + 31) / 32;
    // every shape / LDS feasibility check comes before the first launch (nothing is enqueued for a call that cannot finish)
    // two-term fp16 interaction GEMM when the host vouches for |U|, |Pd| < 2^15 (bounds derived from the projection / conv weights)
    const bool h2 = w->bounded && w->C % 2 == 0 && w->C <= CH && !tun(g_tun.exact_f32);
    // chunks of 64 document positions; of 32 when the document planes of a long document leave too little LDS for a 64-wide Y tile
    const int jts = mt_head_lds(QL, DL, MT, h2, 6) <= 160 * 1024 - 512 ? 6 : 5;
    const size_t lds = mt_head_lds(QL, DL, MT, h2, jts);
    const size_t flds = (size_t)(((QL * w->C + 3) & ~3) + NFC * (w->C + 1) * 3 * 7) * 4;
    NIR_REQUIRE(lds <= 160 * 1024 - 512, "match_tensor: doc length %d needs %zu bytes of LDS (> 160 KiB) at any query length", DL, lds);
    NIR_REQUIRE(flds <= 160 * 1024 - 512, "match_tensor: QL=%d needs %zu bytes of LDS for the query fold (> 160 KiB)", QL, flds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(h2 ? (const void*)mt_head_kernel<true> : (const void*)mt_head_kernel<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("match_tensor: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
            return (int)e;
        }
    }
    if (flds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(h2 ? (const void*)mt_fold_kernel<true> : (const void*)mt_fold_kernel<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds);
        if (e != hipSuccess) {
            set_error("match_tensor: cannot reserve %zu bytes of LDS for the query fold: %s", flds, hipGetErrorString(e));
            return (int)e;
        }
    }
    // The query chain (tiny, latency-bound) runs on a side stream concurrently with the document chain.
    ForkJoin fj(st);
    fj.fork();
    // Round 6: with the fork active (one batch in flight) the DOCUMENT chain is enqueued first -- a replayed hipGraph dispatches its nodes in creation
    // order, and with the query side's three launches created in front the document recurrence (the critical path) started ~10 us later: the
    // reference's loop 0.191 -> 0.180 ms per call at C2.  Without the fork (several batches in flight: one stream) the order of rounds 2-5 stays.
    auto doc_side = [&]() -> int {
    // ---- document side (mtensor.py:80-110): gather fused into the projection GEMM; the LSTM input projection
    // (I = featsize = 40) is fused into the recurrence, so the [tokens, 8H] gate tensor is never written;
    // padded positions of the channel projection give the bias (E3)
    if (given) {
    } else if (folded) {
        NIR_PROPAGATE(launch_bilstm_folded(fold_d, fold_dtype, d_ids, d_len, w->d_whh, hd, err_flag, (int64_t)B * N, V, DL, w->Hd, 2, st));
    } else {
        NIR_PROPAGATE(launch_linear(nullptr, 0, d_ids, table, E, 1, 1, w->proj_w, E, w->proj_b, nullptr, p.xd, w->F, Md, w->F, E, NIR_ACT_NONE, st));
        NIR_PROPAGATE(launch_bilstm_fused(p.xd, w->F, w->d_wih, w->d_bih, w->d_bhh, d_len, w->d_whh, nullptr, nullptr, hd, nullptr, nullptr, (int64_t)B * N, DL, w->Hd, 2, st));
    }
    // the channel projection of the documents runs inside the head kernel when the host supplied its fragment planes (fp16 two-term
    // head only; 2Hd a multiple of 4 for the 16-byte row loads); a requested proj_d output keeps the separate GEMM
    const bool fuse_proj = h2 && w->dproj_frag && (2 * w->Hd) % 4 == 0 && 2 * w->Hd >= 8 && w->C <= 64 && !tun(g_tun.no_skinny);
    if (fuse_proj) { hw.hd = hd; hw.dpf = w->dproj_frag; hw.dpb = w->dproj_b; }
    if (!fuse_proj || proj_d)
        NIR_PROPAGATE(launch_linear(hd, 2 * w->Hd, nullptr, nullptr, 0, 0, 0, w->dproj_w, 2 * w->Hd, w->dproj_b, nullptr, pd, w->C, Md, w->C, 2 * w->Hd, NIR_ACT_NONE, st));
        return 0;
    };
    auto query_side = [&]() -> int {
    {   // ---- query side: gather+projection GEMM -> BiLSTM -> channel projection -> per-query weight folding
        hipStream_t qs = fj.side;
        if (given) {    // states come from the caller
        } else if (folded) {   // embedding, Linear(E->F) and the LSTM input projection folded into one table lookup (nir_lstm_fold_table)
            NIR_PROPAGATE(launch_bilstm_folded(fold_q, fold_dtype, q_ids, q_len, w->q_whh, hq, err_flag, B, V, QL, w->Hq, 2, qs));
        } else {
            NIR_PROPAGATE(launch_linear(nullptr, 0, q_ids, table, E, 1, 1, w->proj_w, E, w->proj_b, nullptr, p.xq, w->F, Mq, w->F, E, NIR_ACT_NONE, qs));
            NIR_PROPAGATE(launch_bilstm_fused(p.xq, w->F, w->q_wih, w->q_bih, w->q_bhh, q_len, w->q_whh, nullptr, nullptr, hq, nullptr, nullptr, B, QL, w->Hq, 2, qs));
        }
        NIR_PROPAGATE(launch_linear(hq, 2 * w->Hq, nullptr, nullptr, 0, 0, 0, w->qproj_w, 2 * w->Hq, w->qproj_b, nullptr, pq, w->C, Mq, w->C, 2 * w->Hq, NIR_ACT_NONE, qs));
        {
            ProfScope ps("mt_fold_kernel", qs);
            if (h2) hipLaunchKernelGGL(mt_fold_kernel<true>, dim3(B, 3, FOLD_Z), dim3(256), flds, qs, pq, hw, QL, MT, p.U);
            else hipLaunchKernelGGL(mt_fold_kernel<false>, dim3(B, 3, FOLD_Z), dim3(256), flds, qs, pq, hw, QL, MT, p.U);
        }
        NIR_CHECK_LAUNCH("mt_fold_kernel");
    }
        return 0;
    };
    if (fj.ok) {
        NIR_PROPAGATE(doc_side());
        NIR_PROPAGATE(query_side());
    } else {
        NIR_PROPAGATE(query_side());
        NIR_PROPAGATE(doc_side());
    }
    fj.join();
    if (tun(g_tun.debug)) {
        int nb = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, h2 ? (const void*)mt_head_kernel<true> : (const void*)mt_head_kernel<false>, 256, lds);
        hipFuncAttributes fa;
        hipFuncGetAttributes(&fa, h2 ? (const void*)mt_head_kernel<true> : (const void*)mt_head_kernel<false>);
        fprintf(stderr, "[nir] mt_head_kernel: lds dyn=%zu static=%zu regs=%d maxThreads=%d -> max active blocks/CU=%d\n", lds,
                (size_t)fa.sharedSizeBytes, fa.numRegs, fa.maxThreadsPerBlock, nb);
    }
    {
        ProfScope ps("mt_head_kernel", st);
        if (h2) hipLaunchKernelGGL(mt_head_kernel<true>, dim3((unsigned)(8 * N * ((B + 7) / 8))), dim3(256), lds, st, pd, p.U, q_ids, d_ids, hw,
                                   B, N, QL, DL, MT, jts, scores);
        else hipLaunchKernelGGL(mt_head_kernel<false>, dim3((unsigned)(8 * N * ((B + 7) / 8))), dim3(256), lds, st, pd, p.U, q_ids, d_ids,
                                hw, B, N, QL, DL, MT, jts, scores);
    }
    NIR_CHECK_LAUNCH("mt_head_kernel");
    return 0;
}
}  // namespace nir

extern "C" int nir_matchtensor_score(const int64_t* q_ids, const int64_t* q_len, const int64_t* d_ids,
                                     const int64_t* d_len, int B, int N, int QL, int DL, const float* table, int64_t V,
                                     int E, const nir_matchtensor_weights* w, void* workspace, size_t workspace_bytes,
                                     float* scores, float* enc_q, float* enc_d, float* proj_q, float* proj_d,
                                     nir_stream_t stream) {
    return nir::matchtensor_impl(q_ids, q_len, d_ids, d_len, B, N, QL, DL, table, V, E, nullptr, nullptr, 0, w, workspace, workspace_bytes, scores,
                                 enc_q, enc_d, proj_q, proj_d, nullptr, (hipStream_t)stream);
}

extern "C" int nir_matchtensor_score_folded(const int64_t* q_ids, const int64_t* q_len, const int64_t* d_ids, const int64_t* d_len, int B,
                                            int N, int QL, int DL, const void* folded_q, const void* folded_d, int dtype, int64_t V,
                                            const nir_matchtensor_weights* w, void* workspace, size_t workspace_bytes, float* scores,
                                            float* enc_q, float* enc_d, float* proj_q, float* proj_d, int* err_flag, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(folded_q && folded_d, "match_tensor_folded: null folded table");
    return matchtensor_impl(q_ids, q_len, d_ids, d_len, B, N, QL, DL, nullptr, V, 1, folded_q, folded_d, dtype, w, workspace, workspace_bytes,
                            scores, enc_q, enc_d, proj_q, proj_d, err_flag, (hipStream_t)stream);
}


extern "C" int nir_matchtensor_score_encoded(const int64_t* q_ids, const int64_t* d_ids, const float* enc_q, const float* enc_d, int B, int N,
                                             int QL, int DL, const nir_matchtensor_weights* w, void* workspace, size_t workspace_bytes,
                                             float* scores, float* proj_q, float* proj_d, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(enc_q && enc_d, "match_tensor_encoded: null encoder states");
    return matchtensor_impl(q_ids, nullptr, d_ids, nullptr, B, N, QL, DL, nullptr, 1, 1, nullptr, nullptr, 0, w, workspace, workspace_bytes, scores,
                            nullptr, nullptr, proj_q, proj_d, nullptr, (hipStream_t)stream, enc_q, enc_d);
}
