// nir_bilstm_fwd / nir_bilstm_fused_fwd: the recurrence of RNNEncoder (neuroir/encoders/rnn_encoder.py:62-141).
//
// One workgroup owns S sequences of ONE direction for the whole recurrence (time loop = workgroup barriers only).
// Mapping: a QUAD of 4 lanes owns hidden unit j; lane q of the quad holds the W_hh rows of all four gates (i,f,g,o)
// of unit j restricted to its quarter of K (4 x KP/4 = KP floats in VGPRs for all T steps), reads only that quarter
// of h_{t-1} from LDS (ds_read_b128, 4 distinct conflict-free addresses per wave), accumulates with packed FMAs and
// the four partial gate sums are combined across the quad with two DPP quad_perm adds -- no gate exchange through
// LDS.  Every lane then has i,f,g,o of its unit, applies the cell update with c_t in a register, lane q==0 writes
// h_t to the ping-pong LDS buffer and to HBM.  One LDS-only barrier per step.
// (First version: one gate column per thread + broadcast reads of all of h + gate exchange via LDS: 4x the LDS
// reads, two barriers, 2.1-3.4 us/step at 2.4 GHz -- latency-bound with ~2 LDS round trips in flight; profiles/.)
//
// Two input modes:
//   IP == 0 : `gates_in` = x W_ih^T + b (one big fp32-MFMA GEMM, nir_linear_f32) is streamed from HBM, prefetched
//             two steps ahead into registers; lane q adds gate q's pre-activation (CARS: I = 300);
//   IP  > 0 : "fused" -- the lane also keeps its quarter of the four W_ih rows (I <= 64) in VGPRs and the x tile of
//             its S sequences is staged once in LDS: no global loads in the loop and the [tokens, 8H] gate tensor
//             never exists (MatchTensor: I = 40; saves a GEMM launch and a 2 x 46 MB HBM round trip).
//
// Global accesses inside the loop are branch-free raw-buffer ops (out-of-range lanes are dropped / read 0 by the
// hardware bounds check) so the compiler can count outstanding VMEM ops exactly instead of falling back to
// `s_waitcnt vmcnt(0)` (a full store round trip) every step.
//
// Variable length = masking: a sequence stops updating at step >= len; the reverse direction walks t = len-1-step
// (packed-sequence semantics) -- no sort, no pack, no host sync (the reference does lengths.tolist(), :73).
#include "common.hpp"
#include <stdlib.h>
#include <string>

namespace nir {

typedef float v2f __attribute__((ext_vector_type(2)));

// matrix-core variants (lstm_mfma.hip); return NIR_ERR_UNSUPPORTED when the shape has no instantiation
int launch_bilstm_mfma16(const float* gin, const int64_t* lens, const float* whh, const float* h0, const float* c0,
                         float* out, float* hn, float* cn, int64_t M, int T, int H, int ND, hipStream_t st, float* act = nullptr, float* cst = nullptr);
int launch_bilstm_mfma(const float* gin, const int64_t* lens, const float* whh, const float* h0, const float* c0,
                       float* out, float* hn, float* cn, int64_t M, int T, int H, int ND, hipStream_t st);

struct LstmArgs {
    const float* gin;       // [M,T,ND*4H]            (IP == 0)
    const float* x;         // [M,T,I]                (IP  > 0)
    const float* wih;       // [ND*4H, I]             (IP  > 0)
    const float* bih;       // [ND*4H]                (IP  > 0)
    const float* bhh;       // [ND*4H]                (IP  > 0)
    const int64_t* lens;    // [M] or null
    const float* whh;       // [ND,4H,H]
    const float* h0;        // [ND,M,H] or null
    const float* c0;
    float* out;             // [M,T,ND*H]
    float* hn;              // [ND,M,H] or null
    float* cn;
    int64_t M;
    int T, H, ND, I;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// sum over the 4 lanes of a quad; every lane gets the total (DPP quad_perm [1,0,3,2] then [2,3,0,1])
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    return v;
}

// Occupancy note (measured, tools/bench_lstm.py): a second workgroup only becomes resident next to the first when the
// kernel stays <= 168 VGPRs (3 waves/SIMD).  Forcing that budget here makes the compiler spill inside the time loop
// (fused S=3: 109 -> 512 us), so the kernel keeps its natural ~200 VGPRs and one workgroup per CU.
template <int KP, int S, int IP>
__global__ __launch_bounds__(4 * KP) void lstm_rec_kernel(LstmArgs p) {
    constexpr int NT = 4 * KP;
    constexpr int KQ = KP / 4;            // k's per lane (multiple of 4)
    constexpr int QS = KQ + 4;            // padded quarter stride in LDS (keeps the 4 quad addresses on distinct banks)
    constexpr int HB = 4 * QS;            // floats per sequence in one h buffer
    constexpr int IQ = IP / 4;            // x elements per lane (fused mode)
    constexpr uint32_t OOB = 0x7FFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hbuf = smem;                   // [2][S][HB]  ping-pong
    float* xbuf = smem + 2 * S * HB;      // [S][T][IP]  (fused mode only)

    const int tid = threadIdx.x;
    const int j = tid >> 2, q = tid & 3;
    const int dir = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * S;
    const int H = p.H, T = p.T;
    const int nvalid = (int)min((int64_t)S, p.M - m0);
    const int G = p.ND * 4 * H;   // gates_in row width
    const int OW = p.ND * H;      // out row width
    const bool uvalid = j < H;

    int len[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int l = 0;
        if (s < nvalid) {
            l = p.lens ? (int)p.lens[m0 + s] : T;
            l = l < 0 ? 0 : (l > T ? T : l);
        }
        len[s] = l;
    }
    int tmax = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) tmax = max(tmax, len[s]);

    // my quarter of the four gate rows of unit j -> registers (packed pairs for v_pk_fma_f32)
    v2f w[4][KQ / 2];
    v2f wi[4][IP > 0 ? IQ / 2 : 1];
    float bias = 0.f;   // lane q carries the bias of gate q (added once per quad)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t row = (int64_t)dir * 4 * H + (int64_t)g * H + (uvalid ? j : 0);
        const float* wr = p.whh + row * H;
#pragma unroll
        for (int k = 0; k < KQ; k += 2) {
            const int k0 = q * KQ + k;
            w[g][k / 2].x = (uvalid && k0 < H) ? wr[k0] : 0.f;
            w[g][k / 2].y = (uvalid && k0 + 1 < H) ? wr[k0 + 1] : 0.f;
        }
        if (IP > 0) {
            const float* wir = p.wih + row * p.I;
#pragma unroll
            for (int k = 0; k < IQ; k += 2) {
                const int k0 = q * IQ + k;
                wi[g][k / 2].x = (uvalid && k0 < p.I) ? wir[k0] : 0.f;
                wi[g][k / 2].y = (uvalid && k0 + 1 < p.I) ? wir[k0 + 1] : 0.f;
            }
            if (uvalid && g == q) bias = p.bih[row] + p.bhh[row];
        }
    }
    if (IP > 0) {  // stage the x tile of my sequences: xbuf[s][t][k], zero padded to IP
        const int per = T * IP;
        for (int e = tid; e < S * per; e += NT) {
            int s = e / per, r = e - s * per, t = r / IP, k = r - t * IP;
            float v = 0.f;
            if (s < nvalid && k < p.I) v = p.x[((m0 + s) * T + t) * p.I + k];
            xbuf[e] = v;
        }
    }
    // initial state (h index k lives at hbuf[(k / KQ) * QS + k % KQ])
    float creg[(S + 3) / 4];              // c_t of the sequences this lane finalises (s = 4r + q)
    for (int e = tid; e < 2 * S * HB; e += NT) hbuf[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < (S + 3) / 4; ++r) {
        creg[r] = 0.f;
        const int s = 4 * r + q;
        if (uvalid && s < nvalid) {
            const int64_t si = ((int64_t)dir * p.M + m0 + s) * H + j;
            if (p.c0) creg[r] = p.c0[si];
            if (p.h0) hbuf[s * HB + (j / KQ) * QS + (j % KQ)] = p.h0[si];
        }
    }
    __syncthreads();

    // per-workgroup buffer descriptors (wave-uniform: built from blockIdx only)
    const __amdgpu_buffer_rsrc_t out_rs = make_rsrc(p.out + m0 * T * OW, (uint32_t)nvalid * T * OW * 4u);
    const __amdgpu_buffer_rsrc_t gin_rs =
        make_rsrc(IP == 0 ? (const void*)(p.gin + m0 * T * G) : (const void*)p.out, IP == 0 ? (uint32_t)nvalid * T * G * 4u : 0u);
    const uint32_t gcol = (uint32_t)(dir * 4 * H + q * H + j) * 4u;   // lane q fetches gate q of unit j

    auto load_gin = [&](int step, float (&dst)[S]) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int l = len[s];
            const int t = dir == 0 ? step : l - 1 - step;
            const uint32_t off = (uvalid && step < l) ? (uint32_t)((s * T + t) * G) * 4u + gcol : OOB;
            dst[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gin_rs, off, 0, 0));  // OOB -> 0
        }
    };

    // Sequences are distributed over the quad for the cell update: lane q finalises sequences s with (s & 3) == q
    // (every lane has all gate totals after the quad reduction), so the transcendental work is not done 4x.
    constexpr int R = (S + 3) / 4;
    auto do_step = [&](int step, float (&cur)[S]) {
        const float* hrd = hbuf + (step & 1) * S * HB;
        float* hwr = hbuf + ((step + 1) & 1) * S * HB;
        float gin_now[S];
#pragma unroll
        for (int s = 0; s < S; ++s) gin_now[s] = IP > 0 ? bias : cur[s];
        if (IP == 0 && step + 2 < tmax) load_gin(step + 2, cur);   // two steps ahead
        // ---- stage A: partial gate sums of every sequence (no LDS writes in between -> reads pipeline freely)
        float tot[S][4];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            v2f acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[g].x = (g == q) ? gin_now[s] : 0.f;
                acc[g].y = 0.f;
            }
            float4 hv[KQ / 4];
#pragma unroll
            for (int k = 0; k < KQ / 4; ++k) hv[k] = *reinterpret_cast<const float4*>(hrd + s * HB + q * QS + 4 * k);
            if (IP > 0) {
                const int l = len[s];
                int t = dir == 0 ? step : l - 1 - step;
                t = t < 0 ? 0 : (t >= T ? T - 1 : t);
                const float* xr = xbuf + (s * T + t) * IP + q * IQ;
                float4 xv[IQ / 4 > 0 ? IQ / 4 : 1];
#pragma unroll
                for (int k = 0; k < IQ / 4; ++k) xv[k] = *reinterpret_cast<const float4*>(xr + 4 * k);
#pragma unroll
                for (int k = 0; k < IQ / 4; ++k) {
                    const v2f lo = {xv[k].x, xv[k].y}, hi = {xv[k].z, xv[k].w};
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[g] = __builtin_elementwise_fma(wi[g][2 * k], lo, acc[g]);
                        acc[g] = __builtin_elementwise_fma(wi[g][2 * k + 1], hi, acc[g]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < KQ / 4; ++k) {
                const v2f lo = {hv[k].x, hv[k].y}, hi = {hv[k].z, hv[k].w};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[g] = __builtin_elementwise_fma(w[g][2 * k], lo, acc[g]);
                    acc[g] = __builtin_elementwise_fma(w[g][2 * k + 1], hi, acc[g]);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) tot[s][g] = quad_sum(acc[g].x + acc[g].y);
        }
        // ---- stage B: cell update; lane q owns sequences 4r + q
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float gi = tot[4 * r][0], gf = tot[4 * r][1], gg = tot[4 * r][2], go = tot[4 * r][3];
            int l = len[4 * r], sidx = 4 * r;
#pragma unroll
            for (int u = 1; u < 4; ++u) {
                if (4 * r + u < S) {
                    const bool mine = q == u;
                    gi = mine ? tot[4 * r + u][0] : gi;
                    gf = mine ? tot[4 * r + u][1] : gf;
                    gg = mine ? tot[4 * r + u][2] : gg;
                    go = mine ? tot[4 * r + u][3] : go;
                    l = mine ? len[4 * r + u] : l;
                    sidx = mine ? 4 * r + u : sidx;
                }
            }
            const bool owner = (4 * r + q < S);            // lanes without a sequence of their own shadow seq 4r
            const float c = fast_sigmoid(gf) * creg[r] + fast_sigmoid(gi) * fast_tanh(gg);
            const float h = fast_sigmoid(go) * fast_tanh(c);
            const bool act = uvalid && owner && step < l;
            if (act) creg[r] = c;
            const int hidx = sidx * HB + (j / KQ) * QS + (j % KQ);
            // h_t -> the other LDS buffer (a finished sequence carries its frozen state over)
            if (uvalid && owner) hwr[hidx] = act ? h : hrd[hidx];
            const int t = dir == 0 ? step : l - 1 - step;
            const uint32_t off = act ? (uint32_t)((sidx * T + t) * OW + dir * H + j) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h), out_rs, off, 0, 0);   // OOB lanes dropped
        }
        lds_barrier();
    };

    float preA[S], preB[S];
    if (IP == 0) {
        if (tmax > 0) load_gin(0, preA);
        if (tmax > 1) load_gin(1, preB);
    }
    int step = 0;
    for (; step + 2 <= tmax; step += 2) {
        do_step(step, preA);
        do_step(step + 1, preB);
    }
    if (step < tmax) {
        do_step(step, preA);
        ++step;
    }

    // zero the padded tail (pad_packed_sequence) and emit final states (lane q owns sequences 4r + q)
    if (uvalid) {
        const float* hfin = hbuf + (step & 1) * S * HB;
#pragma unroll
        for (int r = 0; r < (S + 3) / 4; ++r) {
            const int s = 4 * r + q;
            if (s < nvalid) {
                int l = 0;
#pragma unroll
                for (int u = 0; u < S; ++u) l = (u == s) ? len[u] : l;
                const int64_t m = m0 + s;
                for (int t = l; t < T; ++t) p.out[(m * T + t) * OW + (int64_t)dir * H + j] = 0.f;
                const int64_t si = ((int64_t)dir * p.M + m) * H + j;
                if (p.hn) p.hn[si] = hfin[s * HB + (j / KQ) * QS + (j % KQ)];
                if (p.cn) p.cn[si] = creg[r];
            }
        }
    }
}

template <int KP, int S, int IP>
static int launch_one(const LstmArgs& p, hipStream_t st) {
    static const std::string pname =
        std::string(IP > 0 ? "lstm_rec_kernel[fused]<" : "lstm_rec_kernel<") + std::to_string(KP) + ">";
    const size_t lds = (size_t)(2 * S * 4 * (KP / 4 + 4) + (IP > 0 ? S * p.T * IP : 0)) * 4;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)lstm_rec_kernel<KP, S, IP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("bilstm: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
            return (int)e;
        }
    }
    if (tun(g_tun.debug)) {
        int nb = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)lstm_rec_kernel<KP, S, IP>, 4 * KP, lds);
        fprintf(stderr, "[nir] %s S=%d: grid=%lld x %d, block=%d, lds=%zu B, max active blocks/CU=%d\n", pname.c_str(), S,
                (long long)((p.M + S - 1) / S), p.ND, 4 * KP, lds, nb);
    }
    ProfScope ps(prof_shape_name(pname.c_str(), (long long)p.M, p.T, p.H), st);
    hipLaunchKernelGGL((lstm_rec_kernel<KP, S, IP>), dim3((unsigned)((p.M + S - 1) / S), (unsigned)p.ND), dim3(4 * KP), lds, st, p);
    NIR_CHECK_LAUNCH("nir_bilstm_fwd");
    return 0;
}

template <int KP, int IP>
static int launch_s(const LstmArgs& p, int S, hipStream_t st) {
    switch (S) {
        case 1: return launch_one<KP, 1, IP>(p, st);
        case 2: return launch_one<KP, 2, IP>(p, st);
        case 3: return launch_one<KP, 3, IP>(p, st);
        case 4: return launch_one<KP, 4, IP>(p, st);
        default:
            if (IP > 0) return launch_one<KP, 4, IP>(p, st);   // fused variant is built for S <= 4
            return launch_one<KP, 8, 0>(p, st);
    }
}

template <int IP>
static int launch_kp(const LstmArgs& p, int S, hipStream_t st) {
    const int KP = (p.H + 15) / 16 * 16;
    switch (KP) {
        case 16: return launch_s<16, IP>(p, S, st);
        case 32: return launch_s<32, IP>(p, S, st);
        case 48: return launch_s<48, IP>(p, S, st);
        case 64: return launch_s<64, IP>(p, S, st);
        case 80: return launch_s<80, IP>(p, S, st);
        case 96: return launch_s<96, IP>(p, S, st);
        case 112: return launch_s<112, IP>(p, S, st);
        default: return launch_s<128, IP>(p, S, st);
    }
}

static int pick_s(int64_t seqdirs, bool fused) {
    // Sequences per workgroup.  Measured on MI355X (tools/bench_lstm.py, profiles/): one recurrence step of a
    // workgroup costs ~(0.85 + 0.29*S) us, S == 1 fits two workgroups per CU (VGPR-limited), S > 1 one; a launch
    // runs ceil(workgroups / resident slots) rounds.  Pick the S that minimises rounds x step cost.
    const int cand[5] = {1, 2, 3, 4, 8};
    int best = 1;
    double best_cost = 1e30;
    for (int i = 0; i < 5; ++i) {
        const int S = cand[i];
        if (fused && S > 4) continue;
        const int64_t wgs = (seqdirs + S - 1) / S;
        const int64_t slots = 256 * (S == 1 ? 2 : 1);
        const double rounds = (double)((wgs + slots - 1) / slots);
        const double cost = rounds * (0.85 + 0.29 * S);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = S; }
    }
    {   // tuning override
        const int v = tun(g_tun.lstm_s);
        if (v == 1 || v == 2 || v == 3 || v == 4 || v == 8) best = v;
    }
    if (fused && best > 4) best = 4;
    return best;
}

int launch_bilstm(const float* gin, const int64_t* lens, const float* whh, const float* h0, const float* c0,
                  float* out, float* hn, float* cn, int64_t M, int T, int H, int ND, hipStream_t st) {
    NIR_REQUIRE(gin && whh && out, "bilstm: null pointer");
    NIR_REQUIRE(M >= 0 && T > 0 && (ND == 1 || ND == 2), "bilstm: bad dims M=%lld T=%d ndir=%d", (long long)M, T, ND);
    NIR_REQUIRE(H >= 1 && H <= 128, "bilstm: hidden size %d per direction unsupported (1..128)", H);
    NIR_REQUIRE((int64_t)8 * T * ND * 4 * H * 4 < 0x7FFFFFF0LL, "bilstm: T*H too large for 32-bit tile offsets");
    if (M == 0) return 0;
    // Measured (tools/bench_lstm.py).  From ~1000 sequences up the 16-sequence 16x16x4-MFMA layout wins (H = 128:
    // M = 1120 353 vs 439 us, M = 22400 3.86 vs 7.06 ms).  Below that: the quad/VALU kernel when its workgroup fills the
    // 4 SIMDs evenly (KP/16 waves a multiple of 4: H = 128 439 us vs 520 us for the 4x4x1 layout at M = 1120), the
    // 4x4x1-MFMA recurrence for the unbalanced sizes (H = 70: 5 waves).
    if (!tun(g_tun.lstm_valu)) {   // many sequences, wide H: 16 sequences per workgroup on 16x16x4 MFMAs
        int rc = launch_bilstm_mfma16(gin, lens, whh, h0, c0, out, hn, cn, M, T, H, ND, st);
        if (rc != NIR_ERR_UNSUPPORTED) return rc;
    }
    const bool valu_balanced = (((H + 15) / 16) % 4) == 0;
    if (!valu_balanced && !tun(g_tun.lstm_valu)) {
        int rc = launch_bilstm_mfma(gin, lens, whh, h0, c0, out, hn, cn, M, T, H, ND, st);
        if (rc != NIR_ERR_UNSUPPORTED) return rc;
    }
    LstmArgs p{gin, nullptr, nullptr, nullptr, nullptr, lens, whh, h0, c0, out, hn, cn, M, T, H, ND, 0};
    return launch_kp<0>(p, pick_s(M * ND, false), st);
}

int launch_bilstm_fused_mfma(const float* x, int I, const float* wih, const float* bih, const float* bhh, const int64_t* lens,
                             const float* whh, const float* h0, const float* c0, float* out, float* hn, float* cn, int64_t M,
                             int T, int H, int ND, hipStream_t st);

int launch_bilstm_fused(const float* x, int I, const float* wih, const float* bih, const float* bhh, const int64_t* lens,
                        const float* whh, const float* h0, const float* c0, float* out, float* hn, float* cn, int64_t M,
                        int T, int H, int ND, hipStream_t st) {
    NIR_REQUIRE(x && wih && bih && bhh && whh && out, "bilstm_fused: null pointer");
    NIR_REQUIRE(M >= 0 && T > 0 && (ND == 1 || ND == 2), "bilstm_fused: bad dims");
    NIR_REQUIRE(H >= 1 && H <= 128 && I >= 1 && I <= 64, "bilstm_fused: H=%d (1..128) / I=%d (1..64) unsupported", H, I);
    if (M == 0) return 0;
    if (!tun(g_tun.lstm_valu)) {   // matrix-core recurrence (lstm_mfma.hip) when the shape has an instantiation
        int rc = launch_bilstm_fused_mfma(x, I, wih, bih, bhh, lens, whh, h0, c0, out, hn, cn, M, T, H, ND, st);
        if (rc != NIR_ERR_UNSUPPORTED) return rc;
    }
    LstmArgs p{nullptr, x, wih, bih, bhh, lens, whh, h0, c0, out, hn, cn, M, T, H, ND, I};
    int S = pick_s(M * ND, true);
    const int IP = I <= 48 ? 48 : 64;
    while (S > 1 && (size_t)S * T * IP * 4 > 96 * 1024) S >>= 1;   // keep the x tile comfortably inside LDS
    NIR_REQUIRE((size_t)S * T * IP * 4 <= 140 * 1024, "bilstm_fused: sequence length %d too long for the LDS x tile", T);
    return IP == 48 ? launch_kp<48>(p, S, st) : launch_kp<64>(p, S, st);
}

}  // namespace nir

extern "C" int nir_bilstm_supported(int H) { return H >= 1 && H <= 128; }

extern "C" int nir_bilstm_fwd(const float* gates_in, const int64_t* lengths, const float* w_hh, const float* h0,
                              const float* c0, float* out, float* hn, float* cn, int64_t M, int T, int H, int ndir,
                              nir_stream_t stream) {
    return nir::launch_bilstm(gates_in, lengths, w_hh, h0, c0, out, hn, cn, M, T, H, ndir, (hipStream_t)stream);
}

extern "C" int nir_bilstm_fused_fwd(const float* x, int I, const float* w_ih, const float* b_ih, const float* b_hh,
                                    const int64_t* lengths, const float* w_hh, const float* h0, const float* c0,
                                    float* out, float* hn, float* cn, int64_t M, int T, int H, int ndir,
                                    nir_stream_t stream) {
    return nir::launch_bilstm_fused(x, I, w_ih, b_ih, b_hh, lengths, w_hh, h0, c0, out, hn, cn, M, T, H, ndir,
                                    (hipStream_t)stream);
}
