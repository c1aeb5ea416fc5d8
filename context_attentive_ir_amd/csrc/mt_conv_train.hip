// Training step of the MatchTensor interaction head (neuroir/rankers/mtensor.py:108-121): the three parallel Conv2d(C1 -> NF, (3,3) / (3,5) /
// (3,7), 'same' padding) + ReLU over the [M, C1, H, W] match tensor as DIRECT convolutions -- forward, data gradient, weight gradient.
//
// The im2col + GEMM form of round 3 materialises [M H W, C1 kh kw] patch rows (351 MB at the C2 shape for the 3x7 filter) to multiply them
// by a [., NF = 6] filter matrix: 6 output columns on a 32-wide MFMA tile, and col2im reads the same volume back (3 x 188 us + 0.26 ms
// im2col + 6 skinny GEMMs per step).  The arithmetic is 2.3 GFLOP per pass and the operands are 17 MB: these kernels keep the filter taps in
// SGPRs (wave-uniform scalar loads), the activations in registers / LDS, and never form the patch matrix.
//
// Layouts: T [M, C1, H, W] fp32; filters in their nn.Conv2d layout w_g [NF, C1, 3, kw_g], kw_g = 3, 5, 7 (g = 0, 1, 2); feature rows
// out [M H W, 3 NF] (column g NF + o), the rows the 1x1 convolution reads (a concatenation of the three ReLU outputs).
#include "common.hpp"
#include <algorithm>
#include <mutex>

namespace nir {

struct MtConvArgs {
    const float* T;           // [M, C1, H, W]
    const float* w[3];        // [NF, C1, 3, kw_g]
    const float* b[3];        // [NF]
    float* out;               // fwd: [M H W, 3 NF] (ReLU applied)
    const float* dpre;        // bwd: gradient of the pre-activation [M H W, 3 NF]
    float* dT;                // bwd-data: [M, C1, H, W]
    const float* wt;          // bwd-data: the filters as wt[tap (g, dy, dx)][o][C1P] (mt_conv3_pack_wt_kernel), C1P = C1 rounded up to 4
    float* part;              // bwd-weight: [M, NW] per-sample partial sums, NW = NF C1 3 (3 + 5 + 7), the three filters back to back
    int64_t M;
    int C1, H, W;
};

// ---------------------------------------------------------------------------------------------------------------------
// forward: one thread per position (m, y, x), 3 NF accumulators; per input channel the 3 x 7 window is read once (the 3x3 and 3x5
// filters see its middle columns) and meets the channel's 45 NF taps from SGPRs
// ---------------------------------------------------------------------------------------------------------------------
// blockIdx.y = output group j: filters o = j NG .. of every size (one position per thread with all 3 NF outputs is 1 280 waves at the C2 shape --
// barely one per SIMD, every scalar / vector load wait exposed: 240 us; three groups 130 us, six groups (NG = 1) 114 us; staging the tile in LDS
// instead of reading the windows through L1: 182 us -- the waits are the per-channel scalar tap loads, not the window)
template <int NF, int NG>
__global__ __launch_bounds__(256) void mt_conv3_fwd_kernel(MtConvArgs p) {
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int o0 = blockIdx.y * NG;
    const int HW = p.H * p.W;
    if (pos >= p.M * HW) return;
    const int64_t m = pos / HW;
    const int r = (int)(pos - m * HW), y = r / p.W, x = r - y * p.W;
    float acc[3][NG];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int o = 0; o < NG; ++o) acc[g][o] = p.b[g][o0 + o];
    const float* tb = p.T + m * (int64_t)p.C1 * HW;
    bool ok[3][7];
    int off[3][7];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 7; ++dx) {
            const int yy = y + dy - 1, xx = x + dx - 3;
            ok[dy][dx] = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            off[dy][dx] = ok[dy][dx] ? yy * p.W + xx : r;
        }
    for (int c = 0; c < p.C1; ++c) {
        const float* tc = tb + (int64_t)c * HW;
        float v[3][7];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) {
                const float t = tc[off[dy][dx]];                                  // (clamped address, masked value: no predicated loads)
                v[dy][dx] = ok[dy][dx] ? t : 0.f;
            }
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int kw = 2 * g + 3;
            const float* wg = p.w[g] + ((int64_t)o0 * p.C1 + c) * 3 * kw;       // + o C1 3 kw
#pragma unroll
            for (int o = 0; o < NG; ++o)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2 * g + 3; ++dx)
                        acc[g][o] = fmaf(v[dy][dx + 2 - g], wg[((int64_t)o * p.C1 * 3 + dy) * kw + dx], acc[g][o]);
        }
    }
    float* orow = p.out + pos * (3 * NF);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int o = 0; o < NG; ++o) orow[g * NF + o0 + o] = fmaxf(acc[g][o], 0.f);
}

// ---------------------------------------------------------------------------------------------------------------------
// data gradient: dT[m, c, y, x] = sum_g sum_o sum_(dy, dx) dpre[(m, y - dy + 1, x - dx + pw_g), g NF + o] w_g[o, c, dy, dx].
// One workgroup per sample m stages its dpre rows [H W, 3 NF] in LDS; one thread per position keeps ALL C1 channel sums in registers
// (C1 a compile-time constant) and walks the 21 taps: 3 NF gradient values from LDS per tap, C1 x (NF .. 3 NF) FMAs against SGPR taps.
// ---------------------------------------------------------------------------------------------------------------------
// filters -> wt[tap][o][C1P]: tap = (g, dy, dx) with the three filters back to back (9 + 15 + 21 = 45 taps); the C1 taps of one (tap, o) are then
// one contiguous scalar-load run in the data-gradient kernel
__global__ void mt_conv3_pack_wt_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3, int NF, int C1, int C1P,
                                        float* __restrict__ wt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 45 * NF * C1P) return;
    const int c = i % C1P, o = (i / C1P) % NF, tap = i / (C1P * NF);
    const int g = tap < 9 ? 0 : tap < 24 ? 1 : 2, kw = 2 * g + 3, tl = tap - (g == 0 ? 0 : g == 1 ? 9 : 24), dy = tl / kw, dx = tl % kw;
    const float* w = g == 0 ? w1 : g == 1 ? w2 : w3;
    wt[i] = c < C1 ? w[(((int64_t)o * C1 + c) * 3 + dy) * kw + dx] : 0.f;
}

// (blockIdx.y = channel group of CG = C1 / 3 channels: three times the waves, a third of the accumulators and FMAs each, no reduction)
template <int NF, int C1, int CG>
__global__ __launch_bounds__(256) void mt_conv3_bwd_data_kernel(MtConvArgs p) {
    extern __shared__ float dps[];                      // [H W][3 NF + 1]
    static_assert(C1 % CG == 0, "channel groups");
    constexpr int LD = 3 * NF + 1, C1P = (C1 + 3) / 4 * 4;
    const int c0 = blockIdx.y * CG;
    const int HW = p.H * p.W;
    const int64_t m = blockIdx.x;
    const float* dp = p.dpre + m * (int64_t)HW * 3 * NF;
    for (int e = threadIdx.x; e < HW * 3 * NF; e += 256) dps[(e / (3 * NF)) * LD + e % (3 * NF)] = dp[e];
    __syncthreads();
    for (int r = threadIdx.x; r < HW; r += 256) {
        const int y = r / p.W, x = r - y * p.W;
        float acc[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[c] = 0.f;
        // (tap loops stay rolled: fully unrolled the body is 13 770 FMAs of straight-line code, more than the instruction cache)
#pragma unroll 1
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll 1
            for (int dx = 0; dx < 7; ++dx) {
                // the output position whose window holds (y, x) at tap (dy, dx - 3): (y - dy + 1, x - (dx - 3))
                const int yy = y - dy + 1, xx = x - dx + 3;
                const bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                const float* src = dps + (ok ? yy * p.W + xx : r) * LD;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const int kw = 2 * g + 3, dxg = dx - 2 + g;               // tap column inside filter g
                    if (dxg < 0 || dxg >= kw) continue;                        // wave-uniform: the narrower filters skip the outer columns
                    const int tap = (g == 0 ? 0 : g == 1 ? 9 : 24) + dy * kw + dxg;
#pragma unroll 1
                    for (int o = 0; o < NF; ++o) {
                        const float dd = src[g * NF + o];
                        const float d = ok ? dd : 0.f;
                        const float* wo = p.wt + ((int64_t)tap * NF + o) * C1P + c0;      // uniform address: scalar loads
#pragma unroll
                        for (int c = 0; c < CG; ++c) acc[c] = fmaf(d, wo[c], acc[c]);
                    }
                }
            }
        float* o_ = p.dT + (m * (int64_t)C1 + c0) * HW + r;
#pragma unroll
        for (int c = 0; c < CG; ++c) o_[(int64_t)c * HW] = acc[c];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient, per-sample partial sums (reduced over m by the column-sum kernel: no atomics, fixed summation order inside a sample):
//   part[m][w_g[o, c, dy, dx]] = sum_(y, x) dpre[(m, y, x), g NF + o] T[m, c, y + dy - 1, x + dx - pw_g]
// One workgroup per sample, wave = filter row dy, lane = channel c (C1 <= 64).  The sample's T tile sits in LDS [c][y][x]; a lane walks its
// channel's row y + dy - 1 and meets every (g, o, dx) whose window holds that element: the gradient value of that pairing is the same for
// all lanes of the wave -- a scalar load.  45 NF accumulators per lane.  blockIdx.y = one of MT_XS column ranges of the output positions, each with
// its own partial row (more waves in flight: the scalar loads of a position are otherwise exposed in front of its 90 FMAs).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MT_XS = 2;
#ifndef MT_CG
#define MT_CG 17          // channels per workgroup of the data gradient (3 groups; 17 groups of 3 measured: 186 against 140 us)
#endif
template <int NF>
__global__ __launch_bounds__(192) void mt_conv3_bwd_weight_kernel(MtConvArgs p) {
    extern __shared__ float ts[];                       // [C1][CS], CS = H (W + 1) made odd: lanes are channels, an even stride is a bank conflict
    const int HW = p.H * p.W, WP = p.W + 1, CS = (p.H * WP) | 1;
    const int64_t m = blockIdx.x;
    const float* tb = p.T + m * (int64_t)p.C1 * HW;
    for (int e = threadIdx.x; e < p.C1 * HW; e += 192) {
        const int c = e / HW, r = e - c * HW;
        ts[c * CS + (r / p.W) * WP + r % p.W] = tb[e];
    }
    __syncthreads();
    const int c = threadIdx.x & 63, dy = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool cv = c < p.C1;
    const int cc = cv ? c : 0;
    float acc[3][NF][7];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int o = 0; o < NF; ++o)
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) acc[g][o][dx] = 0.f;
    const float* dp = p.dpre + m * (int64_t)HW * 3 * NF;
    const int xchunk = (p.W + MT_XS - 1) / MT_XS, xb = blockIdx.y * xchunk, xe = min(p.W, xb + xchunk);
    for (int y = 0; y < p.H; ++y) {                     // output row y reads input row y + dy - 1
        const int yy = y + dy - 1;
        if (yy < 0 || yy >= p.H) continue;              // wave-uniform
        const float* trow = ts + cc * CS + yy * WP;
        for (int x = xb; x < xe; ++x) {                 // output position (y, x): uniform across the wave
            const float* d = dp + ((int64_t)y * p.W + x) * 3 * NF;
            float dv[3 * NF];
#pragma unroll
            for (int j = 0; j < 3 * NF; ++j) dv[j] = d[j];                     // scalar loads (uniform address)
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) {
                const int xx = x + dx - 3;
                if (xx < 0 || xx >= p.W) continue;      // wave-uniform
                const float t = trow[xx];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const int dxg = dx - 2 + g;
                    if (dxg < 0 || dxg >= 2 * g + 3) continue;
#pragma unroll
                    for (int o = 0; o < NF; ++o) acc[g][o][dxg] = fmaf(t, dv[g * NF + o], acc[g][o][dxg]);
                }
            }
        }
    }
    if (!cv) return;
    const int64_t NW = (int64_t)NF * p.C1 * 3 * 15;
    float* pr = p.part + (m * MT_XS + blockIdx.y) * NW;
    int64_t base = 0;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int kw = 2 * g + 3;
#pragma unroll
        for (int o = 0; o < NF; ++o)
#pragma unroll
            for (int dx = 0; dx < 2 * g + 3; ++dx) pr[base + (((int64_t)o * p.C1 + c) * 3 + dy) * kw + dx] = acc[g][o][dx];
        base += (int64_t)NF * p.C1 * 3 * kw;
    }
}

static bool mt_conv3_shape_ok(int NF, int C1, int H, int W) {
    return NF == 6 && C1 >= 1 && C1 <= 64 && H >= 1 && W >= 1 && (int64_t)H * W * (3 * NF + 1) * 4 <= 64 * 1024 && (int64_t)C1 * ((H * (W + 1)) | 1) * 4 <= 150 * 1024;
}

}  // namespace nir

using namespace nir;

// 1 when nir_mt_conv3_* serves the shape (NF = 6 filters per size; the data gradient additionally needs C1 = 51, the reference's default
// nchannels + 1) -- callers fall back to im2col rows + GEMM otherwise
extern "C" int nir_mt_conv3_supported(int NF, int C1, int H, int W) {
    return (mt_conv3_shape_ok(NF, C1, H, W) && C1 == 51) ? 1 : 0;
}

extern "C" int nir_mt_conv3_fwd(const float* T, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                                int64_t M, int C1, int H, int W, int NF, float* out, nir_stream_t stream) {
    NIR_REQUIRE(T && w1 && b1 && w2 && b2 && w3 && b3 && out, "mt_conv3_fwd: null pointer");
    NIR_REQUIRE(M >= 0 && nir_mt_conv3_supported(NF, C1, H, W), "mt_conv3_fwd: unsupported shape (NF = 6, C1 = 51)");
    if (M == 0) return 0;
    MtConvArgs a{};
    a.T = T; a.w[0] = w1; a.w[1] = w2; a.w[2] = w3; a.b[0] = b1; a.b[1] = b2; a.b[2] = b3; a.out = out; a.M = M; a.C1 = C1; a.H = H; a.W = W;
    ProfScope ps(prof_shape_name("mt_conv3_fwd_kernel", M * H * W, 3 * NF, C1 * 45), (hipStream_t)stream);
    hipLaunchKernelGGL((mt_conv3_fwd_kernel<6, 1>), dim3((unsigned)((M * H * W + 255) / 256), 6), dim3(256), 0, (hipStream_t)stream, a);
    NIR_CHECK_LAUNCH("mt_conv3_fwd_kernel");
    return 0;
}

extern "C" size_t nir_mt_conv3_partial_floats(int64_t M, int C1, int NF) { return (size_t)M * MT_XS * NF * C1 * 45; }

extern "C" size_t nir_mt_conv3_wt_floats(int C1, int NF) { return (size_t)45 * NF * ((C1 + 3) / 4 * 4); }

extern "C" int nir_mt_conv3_bwd(const float* dpre, const float* T, const float* w1, const float* w2, const float* w3, int64_t M, int C1, int H, int W, int NF,
                                float* dT, float* wt_workspace, float* partial, nir_stream_t stream) {
    NIR_REQUIRE(dpre && T && w1 && w2 && w3, "mt_conv3_bwd: null pointer");
    NIR_REQUIRE(M >= 0 && nir_mt_conv3_supported(NF, C1, H, W), "mt_conv3_bwd: unsupported shape (NF = 6, C1 = 51)");
    if (M == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    MtConvArgs a{};
    a.T = T; a.w[0] = w1; a.w[1] = w2; a.w[2] = w3; a.dpre = dpre; a.dT = dT; a.part = partial; a.M = M; a.C1 = C1; a.H = H; a.W = W;
    if (dT) {
        NIR_REQUIRE(wt_workspace != nullptr, "mt_conv3_bwd: the data gradient needs the nir_mt_conv3_wt_floats() workspace");
        const int C1P = (C1 + 3) / 4 * 4;
        hipLaunchKernelGGL(mt_conv3_pack_wt_kernel, dim3((unsigned)((45 * NF * C1P + 255) / 256)), dim3(256), 0, st, w1, w2, w3, NF, C1, C1P, wt_workspace);
        NIR_CHECK_LAUNCH("mt_conv3_pack_wt_kernel");
        a.wt = wt_workspace;
        ProfScope ps(prof_shape_name("mt_conv3_bwd_data_kernel", M * H * W, C1, 3 * NF * 15), st);
        hipLaunchKernelGGL((mt_conv3_bwd_data_kernel<6, 51, MT_CG>), dim3((unsigned)M, 51 / MT_CG), dim3(256), (size_t)H * W * (3 * NF + 1) * 4, st, a);
        NIR_CHECK_LAUNCH("mt_conv3_bwd_data_kernel");
    }
    if (partial) {
        const size_t lds = (size_t)C1 * ((H * (W + 1)) | 1) * 4;
        if (lds > 64 * 1024) {
            static std::once_flag once;
            std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)mt_conv3_bwd_weight_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); });
        }
        ProfScope ps(prof_shape_name("mt_conv3_bwd_weight_kernel", M * H * W, 3 * NF, C1 * 45), st);
        hipLaunchKernelGGL(mt_conv3_bwd_weight_kernel<6>, dim3((unsigned)M, MT_XS), dim3(192), lds, st, a);
        NIR_CHECK_LAUNCH("mt_conv3_bwd_weight_kernel");
    }
    return 0;
}
