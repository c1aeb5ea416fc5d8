// nir_bilstm_fwd: the recurrent half of RNNEncoder (neuroir/encoders/rnn_encoder.py:62-141, nn.LSTM).
//
// The input half (x W_ih^T + b_ih + b_hh for every token, both directions) is one big MFMA GEMM
// (nir_linear_f32) whose result `gates_in` this kernel consumes step by step.
//
// Design (CDNA4): one workgroup owns S sequences of ONE direction for the whole recurrence, so the time
// loop needs only workgroup barriers.  Thread c of the 4*KP threads owns gate column c = gate*KP + unit and
// keeps its W_hh row (<=128 floats) in VGPRs for all T steps; h_{t-1} of the S sequences lives in LDS and is
// read as wave-uniform (broadcast) ds_read_b128.  Gate pre-activations are exchanged through LDS so that one
// thread per (sequence, unit) applies the cell update with c_t kept in a register.  S is chosen by the host so
// that small batches still spread over all 256 CUs (latency-bound regime) while large batches reuse each
// register-resident W_hh row S times per step.  Variable length = masking: a sequence simply stops updating
// at step >= len; the reverse direction walks t = len-1-step, exactly the packed-sequence semantics, with no
// sort, no pack and no host sync (the reference does lengths.tolist(), rnn_encoder.py:73).
#include "common.hpp"
#include <string>

namespace nir {

struct LstmArgs {
    const float* gin;       // [M,T,ND*4H]
    const int64_t* lens;    // [M] or null
    const float* whh;       // [ND,4H,H]
    const float* h0;        // [ND,M,H] or null
    const float* c0;
    float* out;             // [M,T,ND*H]
    float* hn;              // [ND,M,H] or null
    float* cn;
    int64_t M;
    int T, H, ND;
};

template <int KP, int S>
__global__ __launch_bounds__(4 * KP) void lstm_rec_kernel(LstmArgs p) {
    constexpr int NT = 4 * KP;
    constexpr int R = (S * KP + NT - 1) / NT;  // cell-update rounds per thread
    __shared__ __attribute__((aligned(16))) float hbuf[S * KP];
    __shared__ __attribute__((aligned(16))) float gbuf[S * 4 * KP];
    __shared__ int slen[S];

    const int tid = threadIdx.x;
    const int g = tid / KP, j = tid % KP;
    const int dir = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * S;
    const int H = p.H, T = p.T;
    const int64_t G = (int64_t)p.ND * 4 * H;   // gates_in row width
    const int64_t OW = (int64_t)p.ND * H;      // out row width

    if (tid < S) {
        int64_t m = m0 + tid;
        int l = 0;
        if (m < p.M) {
            l = p.lens ? (int)p.lens[m] : T;
            l = l < 0 ? 0 : (l > T ? T : l);
        }
        slen[tid] = l;
    }
    // recurrent weights of my gate column -> registers
    float w[KP];
    {
        const bool valid = j < H;
        const float* wr = p.whh + ((int64_t)dir * 4 * H + (int64_t)g * H + (valid ? j : 0)) * H;
#pragma unroll
        for (int k = 0; k < KP; ++k) w[k] = (valid && k < H) ? wr[k] : 0.f;
    }
    // initial state
    float creg[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int pidx = tid + r * NT;
        creg[r] = 0.f;
        if (pidx < S * KP) {
            int s = pidx / KP, jj = pidx % KP;
            int64_t m = m0 + s;
            float hv = 0.f;
            if (jj < H && m < p.M) {
                if (p.h0) hv = p.h0[((int64_t)dir * p.M + m) * H + jj];
                if (p.c0) creg[r] = p.c0[((int64_t)dir * p.M + m) * H + jj];
            }
            hbuf[pidx] = hv;
        }
    }
    __syncthreads();
    int tmax = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) tmax = max(tmax, slen[s]);

    const int64_t gcol = (int64_t)dir * 4 * H + (int64_t)g * H + j;
    auto load_gin = [&](int step, float (&dst)[S]) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            int l = slen[s];
            dst[s] = 0.f;
            if (j < H && step < l) {
                int t = dir == 0 ? step : l - 1 - step;
                dst[s] = p.gin[((m0 + s) * T + t) * G + gcol];
            }
        }
    };
    float pre[S];
    if (tmax > 0) load_gin(0, pre);

    for (int step = 0; step < tmax; ++step) {
        float acc[S];
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = pre[s];
        if (step + 1 < tmax) load_gin(step + 1, pre);  // in flight during the mat-vec below
#pragma unroll
        for (int k = 0; k < KP; k += 4) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float4 hv = *reinterpret_cast<const float4*>(&hbuf[s * KP + k]);  // wave-uniform broadcast
                acc[s] = fmaf(w[k], hv.x, acc[s]);
                acc[s] = fmaf(w[k + 1], hv.y, acc[s]);
                acc[s] = fmaf(w[k + 2], hv.z, acc[s]);
                acc[s] = fmaf(w[k + 3], hv.w, acc[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) gbuf[(s * 4 + g) * KP + j] = acc[s];
        lds_barrier();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int pidx = tid + r * NT;
            if (pidx < S * KP) {
                int s = pidx / KP, jj = pidx % KP;
                int l = slen[s];
                if (jj < H && step < l) {
                    float gi = gbuf[(s * 4 + 0) * KP + jj], gf = gbuf[(s * 4 + 1) * KP + jj];
                    float gg = gbuf[(s * 4 + 2) * KP + jj], go = gbuf[(s * 4 + 3) * KP + jj];
                    float c = fast_sigmoid(gf) * creg[r] + fast_sigmoid(gi) * fast_tanh(gg);
                    float h = fast_sigmoid(go) * fast_tanh(c);
                    creg[r] = c;
                    hbuf[pidx] = h;
                    int t = dir == 0 ? step : l - 1 - step;
                    p.out[((m0 + s) * T + t) * OW + (int64_t)dir * H + jj] = h;
                }
            }
        }
        lds_barrier();
    }
    // zero the padded tail (pad_packed_sequence) and emit final states
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int pidx = tid + r * NT;
        if (pidx < S * KP) {
            int s = pidx / KP, jj = pidx % KP;
            int64_t m = m0 + s;
            if (jj < H && m < p.M) {
                for (int t = slen[s]; t < T; ++t) p.out[(m * T + t) * OW + (int64_t)dir * H + jj] = 0.f;
                if (p.hn) p.hn[((int64_t)dir * p.M + m) * H + jj] = hbuf[pidx];
                if (p.cn) p.cn[((int64_t)dir * p.M + m) * H + jj] = creg[r];
            }
        }
    }
}

template <int KP>
static int launch_kp(const LstmArgs& p, int S, hipStream_t st) {
    static const std::string pname = "lstm_rec_kernel<" + std::to_string(KP) + ">";
    ProfScope ps(pname.c_str(), st);
    dim3 block(4 * KP);
    auto grid = [&](int s) { return dim3((unsigned)((p.M + s - 1) / s), (unsigned)p.ND); };
    switch (S) {
        case 1: hipLaunchKernelGGL((lstm_rec_kernel<KP, 1>), grid(1), block, 0, st, p); break;
        case 2: hipLaunchKernelGGL((lstm_rec_kernel<KP, 2>), grid(2), block, 0, st, p); break;
        case 4: hipLaunchKernelGGL((lstm_rec_kernel<KP, 4>), grid(4), block, 0, st, p); break;
        default: hipLaunchKernelGGL((lstm_rec_kernel<KP, 8>), grid(8), block, 0, st, p); break;
    }
    NIR_CHECK_LAUNCH("nir_bilstm_fwd");
    return 0;
}

int launch_bilstm(const float* gin, const int64_t* lens, const float* whh, const float* h0, const float* c0,
                  float* out, float* hn, float* cn, int64_t M, int T, int H, int ND, hipStream_t st) {
    NIR_REQUIRE(gin && whh && out, "bilstm: null pointer");
    NIR_REQUIRE(M >= 0 && T > 0 && (ND == 1 || ND == 2), "bilstm: bad dims M=%lld T=%d ndir=%d", (long long)M, T, ND);
    NIR_REQUIRE(H >= 1 && H <= 128, "bilstm: hidden size %d per direction unsupported (1..128)", H);
    if (M == 0) return 0;
    LstmArgs p{gin, lens, whh, h0, c0, out, hn, cn, M, T, H, ND};
    // sequences per workgroup: spread small batches over the 256 CUs, amortise W_hh for large ones
    const int64_t seqdirs = M * ND;
    int S = 8;
    if (seqdirs <= 512) S = 1;
    else if (seqdirs <= 1024) S = 2;
    else if (seqdirs <= 4096) S = 4;
    const int KP = (H + 15) / 16 * 16;
    switch (KP) {
        case 16: return launch_kp<16>(p, S, st);
        case 32: return launch_kp<32>(p, S, st);
        case 48: return launch_kp<48>(p, S, st);
        case 64: return launch_kp<64>(p, S, st);
        case 80: return launch_kp<80>(p, S, st);
        case 96: return launch_kp<96>(p, S, st);
        case 112: return launch_kp<112>(p, S, st);
        default: return launch_kp<128>(p, S, st);
    }
}

}  // namespace nir

extern "C" int nir_bilstm_supported(int H) { return H >= 1 && H <= 128; }

extern "C" int nir_bilstm_fwd(const float* gates_in, const int64_t* lengths, const float* w_hh, const float* h0,
                              const float* c0, float* out, float* hn, float* cn, int64_t M, int T, int H, int ndir,
                              nir_stream_t stream) {
    return nir::launch_bilstm(gates_in, lengths, w_hh, h0, c0, out, hn, cn, M, T, H, ndir, (hipStream_t)stream);
}
