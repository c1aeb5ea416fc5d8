// BPTT of a 256-unit-per-direction LSTM encoder (MNSRF's query / document encoders in train mode: neuroir/multitask/mnsrf.py:62-114 under
// models/multitask.py:161-223; the forward is nir_lstm256_train_fwd, csrc/lstm_cluster.hip).
//
// Rounds 3-5 ran one masked cell kernel + one [M,1024] x [1024,256] GEMM per step and direction (2 x 2 T launches on two streams; the GEMM's 72
// workgroups made it latency-sized: 25 us for 0.6 GFLOP).  Here ONE launch per step does both directions, the cell gradients and the product:
//
//   grid (row blocks of 16 RT sequences, KS = 8 unit slices, directions).  Member km of a row block owns units [32 km, 32 km + 32):
//     1. gate gradients of ITS cells (rows x 32 units) from the saved activations / cell states, the step's dout and the dh partials / dc of the
//        previous launch  ->  dgates (HBM, the operand of the weight gradients and of dx) and, as the B operand, LDS [row][k & 3][k >> 2];
//     2. partial dh_{prev}[row, all 256 units] = dg[row, its 128 gate rows] W_hh[its 128 gate rows, :] on v_mfma_f32_16x16x4_f32 (exact fp32 products:
//        gate gradients span too many binades for the fp16 term split of the forward), the 128 KB slice of W_hh read once per workgroup and launch
//        (requested before phase 1, lands under it), every A fragment used for all RT sequence tiles;
//     3. the partial goes to dhp[km][dir][row][256]; the owner of a cell adds the eight partials in a fixed order in the next launch
//        (deterministic: no atomics).
//   K is split, not N: a member never recomputes another member's cells, and the B operand of 80 rows is 46 KB of LDS.
//   RT = 5 at M = 1120 x 2 directions: 14 x 8 x 2 = 224 workgroups = one round on 256 CUs, 320 MFMAs per wave and step.
//   Measured (MI355X, M = 1120, T = 64, both directions): 25 us per step launch against 2 x (8 + 25) us of the cell kernel + GEMM pair; the step moves
//   ~66 MB (activations 9, gate gradients 9, partials 18 + 18, states / dout / dc 12) in 128-byte runs at ~2.6 TB/s: traffic-bound, the 9.3 us of
//   fp32 MFMA work per workgroup hide under it (RT = 1..4, i.e. two smaller workgroups per CU, measured slower: 1.87 / 2.0 / 1.9 / 2.27 ms per 64 steps
//   against 1.59; four K slices of 64 units x RT = 3 -- half the partials -- 1.585 ms: the template keeps the parameter, the host uses 8).
//   Also measured and not kept: touching the NEXT step's activations / states / dout during this launch (so that they wait in the Infinity Cache):
//   26.5 us per step against 24.8 -- every extra request lengthens the launch's one operand burst.
#include "common.hpp"
#include <algorithm>

namespace nir {

typedef float f32x4_b __attribute__((ext_vector_type(4)));

struct Bptt256Args {
    const float* dout;       // [M,T,ND*256]   gradient of the memory bank
    const float* act;        // [M,T,ND,1024]  gate activations i,f,g,o
    const float* cst;        // [M,T,ND,256]   cell states
    const int64_t* lens;     // [M] or NULL
    const float* whh;        // [ND,1024,256]
    const float* dhp_in;     // [KS][ND][M][256] partial dh from the previous launch (NULL: first step)
    const float* dc_in;      // [ND][M][256] (NULL: first step)
    float* dhp_out;          // NULL: last step (no product)
    float* dc_out;           // [ND][M][256]
    float* dgx;              // [M,T,ND*1024]
    int64_t M;
    int T, ND, s;
    int dir0;                // first direction of this launch (grid.z = directions of the launch)
};

constexpr int B256_KS = 8;          // (host: workspace sizing = the largest split)

template <int RT, bool FIRST, int KS>
__global__ __launch_bounds__(512, 1) void lstm256_bptt_step_kernel(Bptt256Args p) {
    constexpr int H = 256, H4 = 1024, KU = H / KS, RS = KU + 4, ROWS = 16 * RT, NSP = KU / 32, RPP = 512 / KU;   // sub-passes per tile, rows per sub-pass
    __shared__ __attribute__((aligned(16))) float bs[ROWS * 4 * RS];       // [row][k & 3][k >> 2], k = gate * 32 + local unit
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sq = lane & 15, pq = lane >> 4;
    const int km = blockIdx.y, dir = blockIdx.z + p.dir0;
    const int64_t m0 = (int64_t)blockIdx.x * ROWS, M = p.M;
    const int T = p.T, ND = p.ND;
    const int t = dir == 0 ? T - 1 - p.s : p.s, tprev = dir == 0 ? t - 1 : t + 1;
    const bool prod = p.dhp_out != nullptr;
    // A fragments of the wave's two unit tiles (units 32 wave .. 32 wave + 31): lane (unit = sq, k = 4 ks + pq) -- requested first, used in phase 2
    float afr[2][KU];
    if (prod) {
        const float* w = p.whh + (int64_t)dir * H4 * H + 32 * wave + sq;
#pragma unroll
        for (int ks = 0; ks < KU; ++ks) {
            const int kl = 4 * ks + pq, row = (kl / KU) * H + km * KU + (kl % KU);
#pragma unroll
#ifdef NIR_B256_NOA
            for (int i = 0; i < 2; ++i) afr[i][ks] = (float)(row + i);
#else
            for (int i = 0; i < 2; ++i) afr[i][ks] = w[(int64_t)row * H + 16 * i];
#endif
        }
    }
    // phase 1: the member's cells.  Thread -> (row tid >> 5 of a 16-row pass, local unit tid & 31): 128-byte runs of every operand row.  All operands
    // of all RT passes are requested up front, unconditionally from clamped addresses (a per-pass conditional load chain is 4 dependent memory
    // round trips per pass); the masks are selects on the loaded values
    const int ul = tid % KU, u = km * KU + ul, rsub = tid / KU;
    const int tpc = tprev < 0 ? 0 : (tprev >= T ? T - 1 : tprev);
    constexpr int NP = RT * NSP;                               // passes of RPP rows: pass = tile * NSP + sub-pass, row = 16 tile + RPP sub + rsub
    float ra[NP][4], rc[NP], rcp[NP], rdo[NP], rq[NP][KS], rdc[NP];
    int64_t rlen[NP];
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        const int64_t m = m0 + (pass / NSP) * 16 + (pass % NSP) * RPP + rsub, mc = m < M ? m : M - 1;
        rlen[pass] = p.lens ? p.lens[mc] : (int64_t)T;
    }
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        const int64_t m = m0 + (pass / NSP) * 16 + (pass % NSP) * RPP + rsub, mc = m < M ? m : M - 1;
        const int64_t pos = mc * T + t;
        const float* a = p.act + (pos * ND + dir) * (int64_t)H4 + u;
#pragma unroll
        for (int g = 0; g < 4; ++g) ra[pass][g] = a[g * H];
        rc[pass] = p.cst[(pos * ND + dir) * (int64_t)H + u];
        rcp[pass] = p.cst[((mc * T + tpc) * ND + dir) * (int64_t)H + u];
        rdo[pass] = p.dout[pos * (int64_t)(ND * H) + dir * H + u];
        if (!FIRST) {
// NIR_B256_NOPART / NIR_B256_NOMMA / NIR_B256_NODGX / NIR_B256_NOA: timing ablations (instrumented variant builds only; results are wrong)
#ifdef NIR_B256_NOPART
#pragma unroll
            for (int k = 0; k < KS; ++k) rq[pass][k] = 0.f;
#else
#pragma unroll
            for (int k = 0; k < KS; ++k) rq[pass][k] = p.dhp_in[(((int64_t)k * ND + dir) * M + mc) * H + u];
#endif
            rdc[pass] = p.dc_in[((int64_t)dir * M + mc) * H + u];
        }
    }
    // per sequence tile: cells -> LDS -> barrier (LDS only: the later tiles' operands stay in flight) -> the tile's MFMAs -> partial out.
    // Tile p's matrix work runs while the operands of tiles p+1.. are still arriving (the serial form -- all cells, one barrier, all MFMAs -- spent
    // ~10 us in the memory phase and ~9 us in the MFMA phase of a 26.6 us step)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int sp = 0; sp < NSP; ++sp) {
            const int pass = rt * NSP + sp;
            const int rl = rt * 16 + sp * RPP + rsub;
            const int64_t m = m0 + rl;
            float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, dcn = 0.f;
            if (m < M) {
                const int64_t l = rlen[pass];
                const int len = l < 0 ? 0 : (l > T ? T : (int)l);
                if (t < len) {
                    const float i_ = ra[pass][0], f_ = ra[pass][1], g_ = ra[pass][2], o_ = ra[pass][3];
                    const float th = tanhf(rc[pass]);
                    const float cp = (tprev >= 0 && tprev < len) ? rcp[pass] : 0.f;
                    float dh = 0.f, dc = 0.f;
                    if (!FIRST) {
                        const float* q = rq[pass];
                        if (KS == 8) dh = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4 % KS] + q[5 % KS]) + (q[6 % KS] + q[7 % KS]));
                        else dh = (q[0] + q[1]) + (q[2] + q[3]);
                        dc = rdc[pass];
                    }
                    const float dhh = rdo[pass] + dh;
                    const float dct = dc + dhh * o_ * (1.f - th * th);
                    gi = dct * g_ * i_ * (1.f - i_);
                    gf = dct * cp * f_ * (1.f - f_);
                    gg = dct * i_ * (1.f - g_ * g_);
                    go = dhh * th * o_ * (1.f - o_);
                    dcn = dct * f_;
                }
#ifndef NIR_B256_NODGX
                float* o = p.dgx + (m * T + t) * (int64_t)(ND * H4) + dir * H4;
                o[u] = gi; o[H + u] = gf; o[2 * H + u] = gg; o[3 * H + u] = go;
#endif
                p.dc_out[((int64_t)dir * M + m) * H + u] = dcn;
            }
            if (prod) {
                float* d = bs + (rl * 4 + (ul & 3)) * RS + (ul >> 2);          // k = g * KU + ul -> (k & 3, k >> 2) = (ul & 3, g KU/4 + (ul >> 2))
                d[0] = gi; d[KU / 4] = gf; d[2 * (KU / 4)] = gg; d[3 * (KU / 4)] = go;
            }
        }
        if (!prod) continue;
        lds_barrier();
        // partial dh_prev[unit, seq] += W_hh[k, unit] dg[seq, k] over the member's 4 KU gate rows, sequence tile rt
        f32x4_b acc0 = (f32x4_b){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        const float* bp = bs + ((16 * rt + sq) * 4 + pq) * RS;
#ifndef NIR_B256_NOMMA
#pragma unroll
        for (int k4 = 0; k4 < KU / 4; ++k4) {
            const float4 b = *reinterpret_cast<const float4*>(bp + 4 * k4);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[0][4 * k4 + 0], b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[1][4 * k4 + 0], b.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[0][4 * k4 + 1], b.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[1][4 * k4 + 1], b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[0][4 * k4 + 2], b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[1][4 * k4 + 2], b.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[0][4 * k4 + 3], b.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[1][4 * k4 + 3], b.w, acc1, 0, 0, 0);
        }
#else
        acc0[0] = bp[0] + afr[0][0] + afr[1][KU - 1];
#endif
        // D lane (seq = sq, units 4 pq + r of the tile) -> dhp[km][dir][m][unit]
        const int64_t mo = m0 + 16 * rt + sq;
#ifdef NIR_B256_NOPART
        if (mo < M && acc0[0] == 12345.678f) {
#else
        if (mo < M) {
#endif
            float* o = p.dhp_out + (((int64_t)km * ND + dir) * M + mo) * H + 32 * wave + 4 * pq;
            *reinterpret_cast<float4*>(o) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
            *reinterpret_cast<float4*>(o + 16) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
        }
    }
}

static size_t bptt256_ws_floats(int64_t M, int ND) { return (size_t)2 * (B256_KS + 1) * ND * M * 256; }

}  // namespace nir

extern "C" size_t nir_lstm256_bptt_workspace_bytes(int64_t M, int ndir) {
    return (M >= 0 && (ndir == 1 || ndir == 2)) ? nir::bptt256_ws_floats(M, ndir) * sizeof(float) + 256 : 0;
}

extern "C" int nir_lstm256_bptt(const float* dout, const float* act, const float* cst, const int64_t* lengths, const float* w_hh, float* dgates,
                                int64_t M, int T, int ndir, void* workspace, size_t workspace_bytes, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(dout && act && cst && w_hh && dgates && M >= 0 && T >= 0 && (ndir == 1 || ndir == 2), "lstm256_bptt: bad arguments");
    if (M == 0 || T == 0) return 0;
    NIR_REQUIRE(workspace && workspace_bytes >= nir_lstm256_bptt_workspace_bytes(M, ndir), "lstm256_bptt: workspace too small (%zu < %zu)",
                workspace_bytes, nir_lstm256_bptt_workspace_bytes(M, ndir));
    NIR_REQUIRE(M * (int64_t)T < (1ll << 40), "lstm256_bptt: M * T too large");
    hipStream_t st = (hipStream_t)stream;
    float* ws = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const size_t nhp = (size_t)B256_KS * ndir * M * 256, ndc = (size_t)ndir * M * 256;
    float* dhp[2] = {ws, ws + nhp};
    float* dcb[2] = {ws + 2 * nhp, ws + 2 * nhp + ndc};
    const int64_t tiles = (M + 15) / 16;
    const int KS = 8;            // (KS = 4 x RT = 3 -- half the partial traffic, 192 workgroups, 11 us of MFMA work each -- measured equal: 1.585 vs 1.590 ms)
    int RT = (int)((tiles * KS * ndir + 255) / 256);
    RT = RT < 1 ? 1 : (RT > 5 ? 5 : RT);
    // (Measured and not kept: the two directions as two chains of half-size launches on two streams -- 112 + 112 workgroups that could drift apart, one
    // direction's operand burst under the other's MFMA phase: 1 602 / 1 630 us per BPTT (eager / graphed) against 1 590 / 1 593 for one launch per step.)
    const dim3 grid((unsigned)((tiles + RT - 1) / RT), (unsigned)KS, (unsigned)ndir);
    ProfScope ps(prof_shape_name("lstm256_bptt_steps", (long long)M, T, 256), st);      // ONE label for the T launches of lstm256_bptt_step_kernel
    {
      hipStream_t sl = st;
      for (int s = 0; s < T; ++s) {
        Bptt256Args a;
        a.dout = dout; a.act = act; a.cst = cst; a.lens = lengths; a.whh = w_hh; a.dgx = dgates;
        a.dhp_in = s ? dhp[(s + 1) & 1] : nullptr;
        a.dc_in = s ? dcb[(s + 1) & 1] : nullptr;
        a.dhp_out = s + 1 < T ? dhp[s & 1] : nullptr;
        a.dc_out = dcb[s & 1];
        a.M = M; a.T = T; a.ND = ndir; a.s = s; a.dir0 = 0;
#define NIR_B256_LAUNCH(rt)                                                                                        \
    do {                                                                                                           \
        if (s == 0) hipLaunchKernelGGL((lstm256_bptt_step_kernel<rt, true, 8>), grid, dim3(512), 0, sl, a);        \
        else hipLaunchKernelGGL((lstm256_bptt_step_kernel<rt, false, 8>), grid, dim3(512), 0, sl, a);              \
    } while (0)
        switch (RT) {
            case 1: NIR_B256_LAUNCH(1); break;
            case 2: NIR_B256_LAUNCH(2); break;
            case 3: NIR_B256_LAUNCH(3); break;
            case 4: NIR_B256_LAUNCH(4); break;
            default: NIR_B256_LAUNCH(5); break;
        }
#undef NIR_B256_LAUNCH
      }
    }
    NIR_CHECK_LAUNCH("lstm256_bptt_step_kernel");
    return 0;
}
