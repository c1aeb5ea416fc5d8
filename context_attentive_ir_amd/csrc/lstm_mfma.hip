// BiLSTM recurrence on the matrix cores for narrow inputs (fused input projection): MatchTensor's encoders.
//
// v_mfma_f32_4x4x1_16b_f32 computes 16 independent 4x4 outer products per instruction (probed layout,
// tools/mfma4_layout.hip: block b = lane/4, A[i] from lane 4b+i, B[j] from lane 4b+j, D[i][j] in lane 4b+j reg i).
// With A = z[seq = lane&3][k] (the same 4 values replicated over the 16 blocks) and B = W[col = lane][k] one
// instruction advances 64 gate columns x 4 sequences by one k:  M = 4, so a workgroup needs only FOUR sequences to
// use the matrix pipe (the 16x16 / 32x32 shapes would need 16/32 and starve a small batch of workgroups).
//
// One workgroup = 4 sequences x 1 direction, 4 waves.  z = [h_{t-1} ; x_t] (K = H + I, zero padded) lives in a
// ping-pong LDS buffer; the 4 waves split K evenly (perfect SIMD balance for any H -- the VALU kernel's H = 70 gave a
// 5-wave workgroup with 2 waves on one SIMD) and every wave covers all gate columns with its W slice in VGPRs
// (NG*KQ floats).  Per step: 7 ds_read_b128 + NG*KQ MFMAs per wave, partial sums -> LDS (one b128 per column group),
// barrier, (unit, seq) threads add the 4 partials + bias, apply the cell update (c_t in a register), write h_t and
// x_{t+1} (prefetched from HBM one step ahead) into the other z buffer and h_t to HBM through a branch-free
// raw-buffer store, barrier.  Masking / reverse-direction semantics as in lstm.hip.
#include "common.hpp"
#include <string>

namespace nir {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LstmMfmaArgs {
    const float* x;         // [M,T,I]
    const float* wih;       // [ND*4H, I]
    const float* bih;       // [ND*4H]
    const float* bhh;       // [ND*4H]
    const int64_t* lens;    // [M] or null
    const float* whh;       // [ND,4H,H]
    const float* h0;        // [ND,M,H] or null
    const float* c0;
    float* out;             // [M,T,ND*H]
    float* hn;              // [ND,M,H] or null
    float* cn;
    int64_t M;
    int T, H, ND, I;
    unsigned long long* dbg;
};

// S = sequences a workgroup owns (3 or 4; the MFMA always carries 4 rows, row 3 stays zero for S = 3).  Cell tasks per
// workgroup = S*H: with S = 4 and H = 70 that is 280 > 256 threads, i.e. two tasks on the critical path of every thread
// that has any; S = 3 gives 210 tasks = one per thread (cell phase ~halved) and 214 instead of 160 workgroups at
// M = 320 -- still one round on 256 CUs.  TPT = cell tasks per thread: 1 whenever S*H <= 256 (thread = (unit tid / S,
// sequence tid % S)), 2 only for S = 4 with 4H > 256.  The launcher picks S from a rounds x step-cost model.
template <int NG, int KQ, int S, int TPT>
__global__ __launch_bounds__(256) void lstm_mfma_kernel(LstmMfmaArgs p) {
    constexpr int NW = 4, KZ = NW * KQ, NC = 64 * NG, SEQ = 4;
    constexpr uint32_t OOB = 0x7FFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* z = smem;                               // [2][SEQ][KZ]
    float* part = z + 2 * SEQ * KZ;                // [NW][NC][SEQ]
    float* bias_s = part + NW * NC * SEQ;          // [NC]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * S;
    const int H = p.H, I = p.I, T = p.T, H4 = 4 * H;
    const int nvalid = (int)min((int64_t)S, p.M - m0);
    const int OW = p.ND * H;
    const int myseq = tid & 3;                     // the sequence this thread serves in every role

    int len4[SEQ];
#pragma unroll
    for (int s = 0; s < SEQ; ++s) {
        int l = 0;
        if (s < nvalid) {
            l = p.lens ? (int)p.lens[m0 + s] : T;
            l = l < 0 ? 0 : (l > T ? T : l);
        }
        len4[s] = l;
    }
    const int tmax = max(max(len4[0], len4[1]), max(len4[2], len4[3]));
    const int mylen = myseq == 0 ? len4[0] : myseq == 1 ? len4[1] : myseq == 2 ? len4[2] : len4[3];

    // my K slice of every gate column I cover -> registers
    float wreg[NG][KQ];
#pragma unroll
    for (int cg = 0; cg < NG; ++cg) {
        const int col = 64 * cg + lane;
        const bool cv = col < H4;
        const int64_t row = (int64_t)dir * H4 + (cv ? col : 0);
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const int k = wave * KQ + kk;
            float v = 0.f;
            if (cv) {
                if (k < H) v = p.whh[row * H + k];
                else if (k < H + I) v = p.wih[row * I + (k - H)];
            }
            wreg[cg][kk] = v;
        }
    }
    for (int c = tid; c < NC; c += 256) bias_s[c] = c < H4 ? p.bih[(int64_t)dir * H4 + c] + p.bhh[(int64_t)dir * H4 + c] : 0.f;
    for (int e = tid; e < 2 * SEQ * KZ; e += 256) z[e] = 0.f;
    __syncthreads();

    // x role: thread (i = tid >> 2, seq = tid & 3) moves x[seq][t][i] into z[.][seq][H + i]
    const bool xrole = (tid >> 2) < I;
    const int xi = tid >> 2;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + m0 * T * I), 0,
                                                                           (int)((uint32_t)nvalid * T * I * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(p.out + m0 * T * OW, 0,
                                                                             (int)((uint32_t)nvalid * T * OW * 4u), 0x00020000);
    auto load_x = [&](int step) -> float {
        const int t = dir == 0 ? step : mylen - 1 - step;
        const uint32_t off = (xrole && step < mylen) ? (uint32_t)((myseq * T + t) * I + xi) * 4u : OOB;
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_rs, off, 0, 0));   // OOB -> 0
    };
    // cell role.  TPT == 2: thread (unit cj = tid >> 1, sequence pair cs = 2*(tid & 1)) finalises sequences cs and cs+1;
    //             TPT == 1: thread (unit cj = tid / S, sequence cs = tid % S) finalises one sequence.
    static_assert(TPT == 1 || S == 4, "two tasks per thread only in the 4-sequence layout");
    const int cj = TPT == 2 ? tid >> 1 : tid / S;
    const int cs = TPT == 2 ? 2 * (tid & 1) : tid - S * cj;
    const bool cuv = cj < H;
    const int clen0 = cs == 0 ? len4[0] : cs == 1 ? len4[1] : cs == 2 ? len4[2] : len4[3];
    const int clen1 = TPT == 2 ? (cs == 0 ? len4[1] : len4[3]) : 0;
    float creg0 = 0.f, creg1 = 0.f;
    if (cuv) {   // initial state: h0 / c0
        if (cs < nvalid) {
            const int64_t si = ((int64_t)dir * p.M + m0 + cs) * H + cj;
            if (p.c0) creg0 = p.c0[si];
            if (p.h0) z[cs * KZ + cj] = p.h0[si];
        }
        if (TPT == 2 && cs + 1 < nvalid) {
            const int64_t si = ((int64_t)dir * p.M + m0 + cs + 1) * H + cj;
            if (p.c0) creg1 = p.c0[si];
            if (p.h0) z[(cs + 1) * KZ + cj] = p.h0[si];
        }
    }
    if (xrole) z[myseq * KZ + H + xi] = load_x(0);
    float xnext = load_x(1);
    __syncthreads();

    for (int step = 0; step < tmax; ++step) {
        const float* zc = z + (step & 1) * SEQ * KZ;
        float* zn = z + ((step + 1) & 1) * SEQ * KZ;
#define LM_STAMP(slot) do { if (p.dbg && step == 10 && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) p.dbg[wave * 8 + (slot)] = clock64(); } while (0)
        LM_STAMP(0);
        const float xcur = xnext;                 // x_{step+1}, loaded one step ago
        xnext = load_x(step + 2);                 // in flight during this step
        // ---- gate pre-activations: 4 sequences x all columns x my K slice on the matrix pipe
        f32x4 acc[NG];
#pragma unroll
        for (int cg = 0; cg < NG; ++cg) acc[cg] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* zr = zc + myseq * KZ + wave * KQ;
#pragma unroll
        for (int q4 = 0; q4 < KQ / 4; ++q4) {
            const float4 zv = *reinterpret_cast<const float4*>(zr + 4 * q4);
#pragma unroll
            for (int cg = 0; cg < NG; ++cg) acc[cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(zv.x, wreg[cg][4 * q4 + 0], acc[cg], 0, 0, 0);
#pragma unroll
            for (int cg = 0; cg < NG; ++cg) acc[cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(zv.y, wreg[cg][4 * q4 + 1], acc[cg], 0, 0, 0);
#pragma unroll
            for (int cg = 0; cg < NG; ++cg) acc[cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(zv.z, wreg[cg][4 * q4 + 2], acc[cg], 0, 0, 0);
#pragma unroll
            for (int cg = 0; cg < NG; ++cg) acc[cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(zv.w, wreg[cg][4 * q4 + 3], acc[cg], 0, 0, 0);
        }
#pragma unroll
        for (int cg = 0; cg < NG; ++cg)
            *reinterpret_cast<f32x4*>(part + ((wave * NC) + 64 * cg + lane) * SEQ) = acc[cg];   // [seq0..3] of my column
        LM_STAMP(1);
        lds_barrier();
        LM_STAMP(2);
        // ---- cell update: 4 wave partials + bias per gate, c_t stays in a register
        if constexpr (TPT == 2) {
            const int jj = cuv ? cj : 0;
            float ga[4], gb[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = g * H + jj;
                const float bsv = bias_s[col];
                float2 t0 = *reinterpret_cast<const float2*>(part + (0 * NC + col) * SEQ + cs);
                float2 t1 = *reinterpret_cast<const float2*>(part + (1 * NC + col) * SEQ + cs);
                float2 t2 = *reinterpret_cast<const float2*>(part + (2 * NC + col) * SEQ + cs);
                float2 t3 = *reinterpret_cast<const float2*>(part + (3 * NC + col) * SEQ + cs);
                ga[g] = bsv + t0.x + t1.x + t2.x + t3.x;
                gb[g] = bsv + t0.y + t1.y + t2.y + t3.y;
            }
            const float c0 = fast_sigmoid(ga[1]) * creg0 + fast_sigmoid(ga[0]) * fast_tanh(ga[2]);
            const float c1 = fast_sigmoid(gb[1]) * creg1 + fast_sigmoid(gb[0]) * fast_tanh(gb[2]);
            const float h0v = fast_sigmoid(ga[3]) * fast_tanh(c0);
            const float h1v = fast_sigmoid(gb[3]) * fast_tanh(c1);
            const bool act0 = cuv && step < clen0, act1 = cuv && step < clen1;
            if (act0) creg0 = c0;
            if (act1) creg1 = c1;
            if (cuv) {   // a finished sequence carries its state over
                zn[cs * KZ + cj] = act0 ? h0v : zc[cs * KZ + cj];
                zn[(cs + 1) * KZ + cj] = act1 ? h1v : zc[(cs + 1) * KZ + cj];
            }
            const int t0i = dir == 0 ? step : clen0 - 1 - step, t1i = dir == 0 ? step : clen1 - 1 - step;
            const uint32_t off0 = act0 ? (uint32_t)((cs * T + t0i) * OW + dir * H + cj) * 4u : OOB;
            const uint32_t off1 = act1 ? (uint32_t)(((cs + 1) * T + t1i) * OW + dir * H + cj) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h0v), out_rs, off0, 0, 0);   // OOB lanes dropped
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h1v), out_rs, off1, 0, 0);
        } else {
            const int jj = cuv ? cj : 0;
            float ga[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = g * H + jj;
                ga[g] = bias_s[col] + part[(0 * NC + col) * SEQ + cs] + part[(1 * NC + col) * SEQ + cs] +
                        part[(2 * NC + col) * SEQ + cs] + part[(3 * NC + col) * SEQ + cs];
            }
            const float c0 = fast_sigmoid(ga[1]) * creg0 + fast_sigmoid(ga[0]) * fast_tanh(ga[2]);
            const float h0v = fast_sigmoid(ga[3]) * fast_tanh(c0);
            const bool act0 = cuv && step < clen0;
            if (act0) creg0 = c0;
            if (cuv) zn[cs * KZ + cj] = act0 ? h0v : zc[cs * KZ + cj];
            const int t0i = dir == 0 ? step : clen0 - 1 - step;
            const uint32_t off0 = act0 ? (uint32_t)((cs * T + t0i) * OW + dir * H + cj) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h0v), out_rs, off0, 0, 0);   // OOB lanes dropped
        }
        if (xrole) zn[myseq * KZ + H + xi] = xcur;
        LM_STAMP(3);
        lds_barrier();
        LM_STAMP(4);
    }

    // zero the padded tail (pad_packed_sequence) and emit final states
    const float* zf = z + (tmax & 1) * SEQ * KZ;
    if (cuv) {
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int sq = cs + u;
            if (sq < nvalid) {
                const int l = u == 0 ? clen0 : clen1;
                const int64_t m = m0 + sq;
                for (int t = l; t < T; ++t) p.out[(m * T + t) * OW + (int64_t)dir * H + cj] = 0.f;
                const int64_t si = ((int64_t)dir * p.M + m) * H + cj;
                if (p.hn) p.hn[si] = zf[sq * KZ + cj];
                if (p.cn) p.cn[si] = u == 0 ? creg0 : creg1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Unfused variant (wide inputs, e.g. CARS: H = 128, I = 300): z = h_{t-1} only; the gate pre-activations
// x W_ih^T + b come from the big gather-GEMM (`gates_in`) and are added in the cell update, prefetched one step ahead.
// Waves = NWC column slices x 4 K-quarters; thread (unit = tid >> 2, seq = tid & 3) owns one cell.
// ------------------------------------------------------------------------------------------------------------------
struct LstmMfmaGinArgs {
    const float* gin;       // [M,T,ND*4H]
    const int64_t* lens;
    const float* whh;       // [ND,4H,H]
    const float* h0;
    const float* c0;
    float* out;             // [M,T,ND*H]
    float* hn;
    float* cn;
    int64_t M;
    int T, H, ND;
    float* act;             // train-mode forward only (lstm_mfma16_gin_kernel<.., true>): [M,T,ND,4H] gate activations i,f,g,o and
    float* cst;             //   [M,T,ND,H] cell states of every valid step, for the BPTT kernel (csrc/train.hip)
};

template <int NG, int KQ, int NWC>
__global__ __launch_bounds__(256 * NWC) void lstm_mfma_gin_kernel(LstmMfmaGinArgs p) {
    constexpr int NWK = 4, KZ = NWK * KQ, NC = 64 * NG * NWC, SEQ = 4, NT = 256 * NWC;
    constexpr uint32_t OOB = 0x7FFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* z = smem;                               // [2][SEQ][KZ]
    float* part = z + 2 * SEQ * KZ;                // [NWK][NC][SEQ]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = wave & 3, ch = wave >> 2;       // my K quarter / column slice
    const int dir = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * SEQ;
    const int H = p.H, T = p.T, H4 = 4 * H;
    const int nvalid = (int)min((int64_t)SEQ, p.M - m0);
    const int OW = p.ND * H, G = p.ND * H4;
    const int myseq = tid & 3, cj = tid >> 2;      // cell role: (unit cj, sequence myseq); needs 4H <= NT
    const bool cuv = cj < H;

    int len4[SEQ];
#pragma unroll
    for (int s = 0; s < SEQ; ++s) {
        int l = 0;
        if (s < nvalid) {
            l = p.lens ? (int)p.lens[m0 + s] : T;
            l = l < 0 ? 0 : (l > T ? T : l);
        }
        len4[s] = l;
    }
    const int tmax = max(max(len4[0], len4[1]), max(len4[2], len4[3]));
    const int mylen = myseq == 0 ? len4[0] : myseq == 1 ? len4[1] : myseq == 2 ? len4[2] : len4[3];

    float wreg[NG][KQ];
#pragma unroll
    for (int cg = 0; cg < NG; ++cg) {
        const int col = 64 * (ch * NG + cg) + lane;
        const bool cv = col < H4;
        const float* wr = p.whh + ((int64_t)dir * H4 + (cv ? col : 0)) * H;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const int k = kq * KQ + kk;
            wreg[cg][kk] = (cv && k < H) ? wr[k] : 0.f;
        }
    }
    for (int e = tid; e < 2 * SEQ * KZ; e += NT) z[e] = 0.f;
    __syncthreads();
    float creg = 0.f;
    if (cuv && myseq < nvalid) {
        const int64_t si = ((int64_t)dir * p.M + m0 + myseq) * H + cj;
        if (p.c0) creg = p.c0[si];
        if (p.h0) z[myseq * KZ + cj] = p.h0[si];
    }
    const __amdgpu_buffer_rsrc_t gin_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gin + m0 * T * G), 0,
                                                                             (int)((uint32_t)nvalid * T * G * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(p.out + m0 * T * OW, 0,
                                                                             (int)((uint32_t)nvalid * T * OW * 4u), 0x00020000);
    auto load_gin = [&](int step, float (&dst)[4]) {
        const int t = dir == 0 ? step : mylen - 1 - step;
        const bool ok = cuv && step < mylen;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t off = ok ? (uint32_t)((myseq * T + t) * G + dir * H4 + g * H + cj) * 4u : OOB;
            dst[g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gin_rs, off, 0, 0));   // OOB -> 0
        }
    };
    float gcur[4], gnext[4];
    load_gin(0, gcur);
    __syncthreads();

    for (int step = 0; step < tmax; ++step) {
        const float* zc = z + (step & 1) * SEQ * KZ;
        float* zn = z + ((step + 1) & 1) * SEQ * KZ;
        load_gin(step + 1, gnext);                 // lands during this step's MFMA phase
        f32x4 acc[NG];
#pragma unroll
        for (int cg = 0; cg < NG; ++cg) acc[cg] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* zr = zc + myseq * KZ + kq * KQ;
#pragma unroll
        for (int q4 = 0; q4 < KQ / 4; ++q4) {
            const float4 zv = *reinterpret_cast<const float4*>(zr + 4 * q4);
#pragma unroll
            for (int cg = 0; cg < NG; ++cg) acc[cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(zv.x, wreg[cg][4 * q4 + 0], acc[cg], 0, 0, 0);
#pragma unroll
            for (int cg = 0; cg < NG; ++cg) acc[cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(zv.y, wreg[cg][4 * q4 + 1], acc[cg], 0, 0, 0);
#pragma unroll
            for (int cg = 0; cg < NG; ++cg) acc[cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(zv.z, wreg[cg][4 * q4 + 2], acc[cg], 0, 0, 0);
#pragma unroll
            for (int cg = 0; cg < NG; ++cg) acc[cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(zv.w, wreg[cg][4 * q4 + 3], acc[cg], 0, 0, 0);
        }
#pragma unroll
        for (int cg = 0; cg < NG; ++cg)
            *reinterpret_cast<f32x4*>(part + ((kq * NC) + 64 * (ch * NG + cg) + lane) * SEQ) = acc[cg];
        lds_barrier();
        {   // ---- cell update: one (unit, sequence) per thread
            const int jj = cuv ? cj : 0;
            float g4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = g * H + jj;
                g4[g] = gcur[g] + part[(0 * NC + col) * SEQ + myseq] + part[(1 * NC + col) * SEQ + myseq] +
                        part[(2 * NC + col) * SEQ + myseq] + part[(3 * NC + col) * SEQ + myseq];
            }
            const float c = fast_sigmoid(g4[1]) * creg + fast_sigmoid(g4[0]) * fast_tanh(g4[2]);
            const float h = fast_sigmoid(g4[3]) * fast_tanh(c);
            const bool act = cuv && step < mylen;
            if (act) creg = c;
            if (cuv) zn[myseq * KZ + cj] = act ? h : zc[myseq * KZ + cj];
            const int t = dir == 0 ? step : mylen - 1 - step;
            const uint32_t off = act ? (uint32_t)((myseq * T + t) * OW + dir * H + cj) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h), out_rs, off, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) gcur[g] = gnext[g];
        }
        lds_barrier();
    }
    if (cuv && myseq < nvalid) {
        const float* zf = z + (tmax & 1) * SEQ * KZ;
        const int64_t m = m0 + myseq;
        for (int t = mylen; t < T; ++t) p.out[(m * T + t) * OW + (int64_t)dir * H + cj] = 0.f;
        const int64_t si = ((int64_t)dir * p.M + m) * H + cj;
        if (p.hn) p.hn[si] = zf[myseq * KZ + cj];
        if (p.cn) p.cn[si] = creg;
    }
}

template <int NG, int KQ, int NWC>
static int launch_mfma_gin(const LstmMfmaGinArgs& p, hipStream_t st) {
    static const std::string pname = "lstm_mfma_gin_kernel<" + std::to_string(NG) + "," + std::to_string(KQ) + "," + std::to_string(NWC) + ">";
    constexpr size_t lds = (size_t)(2 * 4 * 4 * KQ + 4 * 64 * NG * NWC * 4) * 4;
    ProfScope ps(prof_shape_name(pname.c_str(), (long long)p.M, p.T, p.H), st);
    hipLaunchKernelGGL((lstm_mfma_gin_kernel<NG, KQ, NWC>), dim3((unsigned)((p.M + 3) / 4), (unsigned)p.ND), dim3(256 * NWC), lds, st, p);
    NIR_CHECK_LAUNCH("nir_bilstm_fwd[mfma]");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Wide hidden sizes with many sequences (CARS: H = 128, thousands of documents): 16 sequences per workgroup on
// v_mfma_f32_16x16x4_f32, 16 waves.
//   * D = W_tile (16 gate rows x K = H) * h^T (K x 16 sequences).  All 16 MFMA columns carry a sequence (the 4x4x1
//     layout above fills 4), the 8-pass MFMA holds the matrix pipe for 32 cycles but the issue port for 4, and with 4
//     waves per SIMD one wave's cell update runs in the shadow of the others' MFMAs.
//   * gate rows are interleaved as row = 4*unit + gate, so the C/D layout (lane (col = lane & 15, rows 4*(lane >> 4)
//     + r)) hands ONE lane the four gates i,f,g,o of (unit = 4*tile + (lane >> 4), sequence = lane & 15) in its four
//     accumulator registers: no partial sums, no exchange, c_t / h_t stay in registers.
//   * wave w owns tiles w and w + 16 (4 hidden units each, two interleaved accumulate chains), K is not split, so a
//     step needs ONE barrier (h ping-pong in LDS); its slice of W_hh (2 x 4*ceil(H/16) floats per lane) stays in VGPRs
//     for all T steps.  The input part of the gates (gates_in, biases included) is prefetched one step ahead.
// ------------------------------------------------------------------------------------------------------------------
template <int G, int NT, bool TR = false>
__global__ __launch_bounds__(1024) void lstm_mfma16_gin_kernel(LstmMfmaGinArgs p) {
    constexpr int SEQ = 16, KP = 16 * G, ZLD = KP + 4, NW = 16;
    constexpr uint32_t OOB = 0x7FFFFFF0u;
    __shared__ __attribute__((aligned(16))) float z[2][SEQ][ZLD];
    __shared__ int lens_s[SEQ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sq = lane & 15, kq = lane >> 4;       // operand view: (row/col sq, k quarter kq); result view: (sequence sq, local unit kq)
    const int dir = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * SEQ;
    const int H = p.H, T = p.T, H4 = 4 * H;
    const int nvalid = (int)min((int64_t)SEQ, p.M - m0);
    const int OW = p.ND * H, GW = p.ND * H4;
    const int ntiles = (H + 3) / 4;

    if (tid < SEQ) {
        int l = 0;
        if (tid < nvalid) {
            l = p.lens ? (int)p.lens[m0 + tid] : T;
            l = l < 0 ? 0 : (l > T ? T : l);
        }
        lens_s[tid] = l;
    }
    for (int e = tid; e < 2 * SEQ * ZLD; e += 1024) (&z[0][0][0])[e] = 0.f;
    __syncthreads();
    int tmax = 0;
#pragma unroll
    for (int s2 = 0; s2 < SEQ; ++s2) tmax = max(tmax, lens_s[s2]);
    const int mylen = lens_s[sq];

    float wreg[NT][4 * G];
    float creg[NT], hreg[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tile = wave + NW * t;
        const int unit_a = 4 * tile + (sq >> 2), gate_a = sq & 3;
        const bool av = unit_a < H;
        const float* wr = p.whh + ((int64_t)dir * H4 + (int64_t)gate_a * H + (av ? unit_a : 0)) * H;
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * q + 4 * kq + j;
                wreg[t][4 * q + j] = (av && k < H) ? wr[k] : 0.f;
            }
        const int unit_d = 4 * tile + kq;
        creg[t] = 0.f;
        hreg[t] = 0.f;
        if (unit_d < H && sq < nvalid) {
            const int64_t si = ((int64_t)dir * p.M + m0 + sq) * H + unit_d;
            if (p.c0) creg[t] = p.c0[si];
            if (p.h0) { hreg[t] = p.h0[si]; z[0][sq][unit_d] = hreg[t]; }
        }
    }
    const __amdgpu_buffer_rsrc_t gin_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gin + m0 * T * GW), 0,
                                                                             (int)((uint32_t)nvalid * T * GW * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(p.out + m0 * T * OW, 0,
                                                                             (int)((uint32_t)nvalid * T * OW * 4u), 0x00020000);
    // train mode: the gate activations and cell states of every valid step go to memory for the backward (same OOB-offset = dropped trick)
    const __amdgpu_buffer_rsrc_t act_rs = __builtin_amdgcn_make_buffer_rsrc(TR ? p.act + m0 * T * GW : p.out, 0,
                                                                             TR ? (int)((uint32_t)nvalid * T * GW * 4u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t cst_rs = __builtin_amdgcn_make_buffer_rsrc(TR ? p.cst + m0 * T * OW : p.out, 0,
                                                                             TR ? (int)((uint32_t)nvalid * T * OW * 4u) : 0, 0x00020000);
    auto load_gin = [&](int step, float (&dst)[NT][4]) {
        const int t_ = dir == 0 ? step : mylen - 1 - step;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int unit_d = 4 * (wave + NW * t) + kq;
            const bool ok = unit_d < H && step < mylen;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t off = ok ? (uint32_t)((sq * T + t_) * GW + dir * H4 + r * H + unit_d) * 4u : OOB;
                dst[t][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gin_rs, off, 0, 0));   // OOB -> 0
            }
        }
    };
    float gcur[NT][4], gnext[NT][4];
    load_gin(0, gcur);
    __syncthreads();

    const bool has1 = wave < ntiles;
    const bool has2 = NT > 1 && wave + NW < ntiles;    // wave-uniform
    for (int step = 0; step < tmax; ++step) {
        const float* zc = &z[step & 1][0][0];
        float* zn = &z[(step + 1) & 1][0][0];
        load_gin(step + 1, gnext);                   // lands during this step's MFMAs
        const bool live = step < mylen;
        const int tt = dir == 0 ? step : mylen - 1 - step;
        auto cell_tile = [&](int t, const f32x4& acc) {
            const int unit_d = 4 * (wave + NW * t) + kq;
            const bool dv = unit_d < H;
            const float gi = fast_sigmoid(acc[0] + gcur[t][0]);
            const float gf = fast_sigmoid(acc[1] + gcur[t][1]);
            const float gg = fast_tanh(acc[2] + gcur[t][2]);
            const float go = fast_sigmoid(acc[3] + gcur[t][3]);
            const float cn = gf * creg[t] + gi * gg;
            const float hn = go * fast_tanh(cn);
            const bool act = dv && live;
            creg[t] = act ? cn : creg[t];            // a finished sequence carries its state over
            hreg[t] = act ? hn : hreg[t];
            if (dv) zn[sq * ZLD + unit_d] = hreg[t];
            const uint32_t off = act ? (uint32_t)((sq * T + tt) * OW + dir * H + unit_d) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hn), out_rs, off, 0, 0);   // OOB lanes dropped
            if (TR) {      // act [M,T,ND,4H] (gate-major inside a direction), cst [M,T,ND,H]
                const uint32_t ao = act ? (uint32_t)(((sq * T + tt) * p.ND + dir) * H4 + unit_d) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gi), act_rs, ao, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gf), act_rs, act ? ao + (uint32_t)H * 4u : OOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gg), act_rs, act ? ao + (uint32_t)H * 8u : OOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(go), act_rs, act ? ao + (uint32_t)H * 12u : OOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(cn), cst_rs, act ? (uint32_t)(((sq * T + tt) * p.ND + dir) * H + unit_d) * 4u : OOB, 0, 0);
            }
        };
        if (has1) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* zr = zc + sq * ZLD + 4 * kq;
            if (has2) {
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const float4 zf = *reinterpret_cast<const float4*>(zr + 16 * q);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 0], zf.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[NT - 1][4 * q + 0], zf.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 1], zf.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[NT - 1][4 * q + 1], zf.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 2], zf.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[NT - 1][4 * q + 2], zf.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 3], zf.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[NT - 1][4 * q + 3], zf.w, acc1, 0, 0, 0);
                }
                cell_tile(0, acc0);
                cell_tile(NT - 1, acc1);
            } else {
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const float4 zf = *reinterpret_cast<const float4*>(zr + 16 * q);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 0], zf.x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 1], zf.y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 2], zf.z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 3], zf.w, acc0, 0, 0, 0);
                }
                cell_tile(0, acc0);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) gcur[t][r] = gnext[t][r];
        lds_barrier();
    }

    // zero the padded tail (pad_packed_sequence) and emit final states
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int unit_d = 4 * (wave + NW * t) + kq;
        if (unit_d < H && sq < nvalid) {
            const int64_t m = m0 + sq;
            for (int t2 = mylen; t2 < T; ++t2) p.out[(m * T + t2) * OW + (int64_t)dir * H + unit_d] = 0.f;
            const int64_t si = ((int64_t)dir * p.M + m) * H + unit_d;
            if (p.hn) p.hn[si] = hreg[t];
            if (p.cn) p.cn[si] = creg[t];
        }
    }
}

template <int G, int NT>
static int launch_mfma16_gin(const LstmMfmaGinArgs& p, hipStream_t st) {
    static const std::string pname = "lstm_mfma16_gin_kernel<" + std::to_string(G) + "," + std::to_string(NT) + ">";
    static const std::string tname = "lstm_train_fwd_mfma16_kernel<" + std::to_string(G) + "," + std::to_string(NT) + ">";
    const bool tr = p.act != nullptr;
    ProfScope ps(prof_shape_name(tr ? tname.c_str() : pname.c_str(), (long long)p.M, p.T, p.H), st);
    if (tr) hipLaunchKernelGGL((lstm_mfma16_gin_kernel<G, NT, true>), dim3((unsigned)((p.M + 15) / 16), (unsigned)p.ND), dim3(1024), 0, st, p);
    else hipLaunchKernelGGL((lstm_mfma16_gin_kernel<G, NT, false>), dim3((unsigned)((p.M + 15) / 16), (unsigned)p.ND), dim3(1024), 0, st, p);
    NIR_CHECK_LAUNCH("nir_bilstm_fwd[mfma16]");
    return 0;
}

// 16-sequence layout: pays once there are enough sequences to give most CUs a workgroup (otherwise the quad/VALU
// kernel spreads a small batch over more CUs).  NIR_LSTM_MFMA16=0/1 forces it off/on.
int launch_bilstm_mfma16(const float* gin, const int64_t* lens, const float* whh, const float* h0, const float* c0,
                         float* out, float* hn, float* cn, int64_t M, int T, int H, int ND, hipStream_t st, float* act, float* cst) {
    const int f = tun(g_tun.lstm_mfma16);       // -1 auto, 0 off, 1 forced
    // (train-mode forward, act != NULL: always -- the alternative there is the scalar kernel of csrc/train.hip)
    if (!act && (f >= 0 ? f == 0 : ((M + 15) / 16) * ND < 128)) return NIR_ERR_UNSUPPORTED;
    if (H < 33 || H > 128 || (int64_t)16 * T * ND * 4 * H * 4 >= 0x7FFFFFF0LL) return NIR_ERR_UNSUPPORTED;
    LstmMfmaGinArgs p{gin, lens, whh, h0, c0, out, hn, cn, M, T, H, ND, act, cst};
    const int G = (H + 15) / 16;
    if (H <= 64) {
        if (G == 3) return launch_mfma16_gin<3, 1>(p, st);
        return launch_mfma16_gin<4, 1>(p, st);
    }
    switch (G) {
        case 5: return launch_mfma16_gin<5, 2>(p, st);
        case 6: return launch_mfma16_gin<6, 2>(p, st);
        case 7: return launch_mfma16_gin<7, 2>(p, st);
        default: return launch_mfma16_gin<8, 2>(p, st);
    }
}

int launch_bilstm_mfma(const float* gin, const int64_t* lens, const float* whh, const float* h0, const float* c0,
                       float* out, float* hn, float* cn, int64_t M, int T, int H, int ND, hipStream_t st) {
    if ((int64_t)4 * T * ND * 4 * H * 4 >= 0x7FFFFFF0LL || H < 17) return NIR_ERR_UNSUPPORTED;
    LstmMfmaGinArgs p{gin, lens, whh, h0, c0, out, hn, cn, M, T, H, ND, nullptr, nullptr};
    if (H <= 64) return launch_mfma_gin<4, 16, 1>(p, st);
    if (H <= 96) return launch_mfma_gin<3, 24, 2>(p, st);
    if (H <= 128) return launch_mfma_gin<4, 32, 2>(p, st);
    return NIR_ERR_UNSUPPORTED;
}

template <int NG, int KQ, int S, int TPT>
static int launch_mfma_s(const LstmMfmaArgs& p, hipStream_t st) {
    static const std::string pname = "lstm_mfma_kernel<" + std::to_string(NG) + "," + std::to_string(KQ) + "," + std::to_string(S) + "," + std::to_string(TPT) + ">";
    constexpr size_t lds = (size_t)(2 * 4 * 4 * KQ + 4 * 64 * NG * 4 + 64 * NG) * 4;
    ProfScope ps(prof_shape_name(pname.c_str(), (long long)p.M, p.T, p.H), st);
    hipLaunchKernelGGL((lstm_mfma_kernel<NG, KQ, S, TPT>), dim3((unsigned)((p.M + S - 1) / S), (unsigned)p.ND), dim3(256), lds, st, p);
    NIR_CHECK_LAUNCH("nir_bilstm_fused_fwd[mfma]");
    return 0;
}

// Layout choice.  4H <= 256: four sequences, one cell task per thread.  Otherwise (3H <= 256 holds for every
// instantiated NG <= 5) three sequences with one task per thread against four with two: rounds over the 256 CUs x
// per-step cost (measured at H = 70: the S = 3 step costs ~0.85 of the S = 4 / two-task step).
template <int NG, int KQ>
static int launch_mfma(const LstmMfmaArgs& p, hipStream_t st) {
    if constexpr (NG <= 4) {
        if (4 * p.H <= 256) return launch_mfma_s<NG, KQ, 4, 1>(p, st);
    }
    bool three = false;
    const int force = tun(g_tun.lstm_mfma_s);
    if (force) {
        three = force == 3;
    } else if (batches_in_flight(st) <= 1) {
        // (with several batches in flight the 4-sequence layout wins: all 4 MFMA rows carry work and 160 workgroups of
        //  one batch leave CUs for the next -- measured 3.69 M vs 3.36 M pairs/s at 4 batches in flight)
        const int64_t wg4 = ((p.M + 3) / 4) * p.ND, wg3 = ((p.M + 2) / 3) * p.ND;
        three = (double)((wg3 + 255) / 256) * 0.85 < (double)((wg4 + 255) / 256);
    }
    return three ? launch_mfma_s<NG, KQ, 3, 1>(p, st) : launch_mfma_s<NG, KQ, 4, 2>(p, st);
}

// ------------------------------------------------------------------------------------------------------------------
// The same 16-sequence / 16-wave layout with the input projection fused (z = [h ; x], K = H + I): for narrow inputs
// (MatchTensor) once there are enough sequences to give most CUs a 16-sequence workgroup.  At the C2 batch (640
// sequences = 40 workgroups) the 4x4x1 layout above is faster (102 vs 176 us); at 6400 sequences this one needs one
// round of ~180 us where the 4x4x1 layout needs 6.
// ------------------------------------------------------------------------------------------------------------------
template <int G, int NT>
__global__ __launch_bounds__(1024) void lstm_mfma16_kernel(LstmMfmaArgs p) {
    constexpr int SEQ = 16, KP = 16 * G, ZLD = KP + 4, NW = 16;
    constexpr uint32_t OOB = 0x7FFFFFF0u;
    __shared__ __attribute__((aligned(16))) float z[2][SEQ][ZLD];
    __shared__ int lens_s[SEQ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sq = lane & 15, kq = lane >> 4;
    const int dir = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * SEQ;
    const int H = p.H, I = p.I, T = p.T, H4 = 4 * H;
    const int nvalid = (int)min((int64_t)SEQ, p.M - m0);
    const int OW = p.ND * H;
    const int ntiles = (H + 3) / 4;

    if (tid < SEQ) {
        int l = 0;
        if (tid < nvalid) {
            l = p.lens ? (int)p.lens[m0 + tid] : T;
            l = l < 0 ? 0 : (l > T ? T : l);
        }
        lens_s[tid] = l;
    }
    for (int e = tid; e < 2 * SEQ * ZLD; e += 1024) (&z[0][0][0])[e] = 0.f;
    __syncthreads();
    int tmax = 0;
#pragma unroll
    for (int s2 = 0; s2 < SEQ; ++s2) tmax = max(tmax, lens_s[s2]);
    const int mylen = lens_s[sq];

    float wreg[NT][4 * G];
    float bias[NT][4];
    float creg[NT], hreg[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tile = wave + NW * t;
        const int unit_a = 4 * tile + (sq >> 2), gate_a = sq & 3;
        const bool av = unit_a < H;
        const int64_t wrow = (int64_t)dir * H4 + (int64_t)gate_a * H + (av ? unit_a : 0);
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * q + 4 * kq + j;
                float v = 0.f;
                if (av) {
                    if (k < H) v = p.whh[wrow * H + k];
                    else if (k < H + I) v = p.wih[wrow * I + (k - H)];
                }
                wreg[t][4 * q + j] = v;
            }
        const int unit_d = 4 * tile + kq;
        const bool dv = unit_d < H;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t bi = (int64_t)dir * H4 + (int64_t)r * H + (dv ? unit_d : 0);
            bias[t][r] = dv ? p.bih[bi] + p.bhh[bi] : 0.f;
        }
        creg[t] = 0.f;
        hreg[t] = 0.f;
        if (dv && sq < nvalid) {
            const int64_t si = ((int64_t)dir * p.M + m0 + sq) * H + unit_d;
            if (p.c0) creg[t] = p.c0[si];
            if (p.h0) { hreg[t] = p.h0[si]; z[0][sq][unit_d] = hreg[t]; }
        }
    }

    // x role: thread e = tid -> (sequence xs = e / I, input xi = e % I) moves x[xs][t][xi] into z[.][xs][H + xi]  (16*I <= 1024)
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + m0 * T * I), 0,
                                                                           (int)((uint32_t)nvalid * T * I * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(p.out + m0 * T * OW, 0,
                                                                             (int)((uint32_t)nvalid * T * OW * 4u), 0x00020000);
    const int xs = tid / I, xi = tid - xs * I;
    const bool xrole = xs < SEQ;
    const int xl = xrole ? lens_s[xs] : 0;
    auto load_x = [&](int step) -> float {
        const int t = dir == 0 ? step : xl - 1 - step;
        const uint32_t off = step < xl ? (uint32_t)((xs * T + t) * I + xi) * 4u : OOB;
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_rs, off, 0, 0));   // OOB -> 0
    };
    if (xrole) z[0][xs][H + xi] = load_x(0);
    float xnext = load_x(1);
    __syncthreads();

    const bool has1 = wave < ntiles;
    const bool has2 = NT > 1 && wave + NW < ntiles;    // wave-uniform
    for (int step = 0; step < tmax; ++step) {
        const float* zc = &z[step & 1][0][0];
        float* zn = &z[(step + 1) & 1][0][0];
        const float xcur = xnext;
        xnext = load_x(step + 2);                    // in flight during this step
        const bool live = step < mylen;
        const int tt = dir == 0 ? step : mylen - 1 - step;
        auto cell_tile = [&](int t, const f32x4& acc) {
            const int unit_d = 4 * (wave + NW * t) + kq;
            const bool dv = unit_d < H;
            const float gi = fast_sigmoid(acc[0] + bias[t][0]);
            const float gf = fast_sigmoid(acc[1] + bias[t][1]);
            const float gg = fast_tanh(acc[2] + bias[t][2]);
            const float go = fast_sigmoid(acc[3] + bias[t][3]);
            const float cn = gf * creg[t] + gi * gg;
            const float hn = go * fast_tanh(cn);
            const bool act = dv && live;
            creg[t] = act ? cn : creg[t];            // a finished sequence carries its state over
            hreg[t] = act ? hn : hreg[t];
            if (dv) zn[sq * ZLD + unit_d] = hreg[t];
            const uint32_t off = act ? (uint32_t)((sq * T + tt) * OW + dir * H + unit_d) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hn), out_rs, off, 0, 0);   // OOB lanes dropped
        };
        if (has1) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* zr = zc + sq * ZLD + 4 * kq;
            if (has2) {
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const float4 zf = *reinterpret_cast<const float4*>(zr + 16 * q);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 0], zf.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[NT - 1][4 * q + 0], zf.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 1], zf.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[NT - 1][4 * q + 1], zf.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 2], zf.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[NT - 1][4 * q + 2], zf.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 3], zf.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[NT - 1][4 * q + 3], zf.w, acc1, 0, 0, 0);
                }
                cell_tile(0, acc0);
                cell_tile(NT - 1, acc1);
            } else {
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const float4 zf = *reinterpret_cast<const float4*>(zr + 16 * q);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 0], zf.x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 1], zf.y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 2], zf.z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][4 * q + 3], zf.w, acc0, 0, 0, 0);
                }
                cell_tile(0, acc0);
            }
        }
        if (xrole) zn[xs * ZLD + H + xi] = xcur;
        lds_barrier();
    }

    // zero the padded tail (pad_packed_sequence) and emit final states
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int unit_d = 4 * (wave + NW * t) + kq;
        if (unit_d < H && sq < nvalid) {
            const int64_t m = m0 + sq;
            for (int t2 = mylen; t2 < T; ++t2) p.out[(m * T + t2) * OW + (int64_t)dir * H + unit_d] = 0.f;
            const int64_t si = ((int64_t)dir * p.M + m) * H + unit_d;
            if (p.hn) p.hn[si] = hreg[t];
            if (p.cn) p.cn[si] = creg[t];
        }
    }
}

template <int G, int NT>
static int launch_mfma16(const LstmMfmaArgs& p, hipStream_t st) {
    static const std::string pname = "lstm_mfma16_kernel<" + std::to_string(G) + "," + std::to_string(NT) + ">";
    ProfScope ps(prof_shape_name(pname.c_str(), (long long)p.M, p.T, p.H), st);
    hipLaunchKernelGGL((lstm_mfma16_kernel<G, NT>), dim3((unsigned)((p.M + 15) / 16), (unsigned)p.ND), dim3(1024), 0, st, p);
    NIR_CHECK_LAUNCH("nir_bilstm_fused_fwd[mfma16]");
    return 0;
}

// NIR_ERR_UNSUPPORTED when the shape has no instantiation (H <= 128, I <= 64, H + I <= 160) or there are too few sequences
static int launch_bilstm_fused_mfma16(const LstmMfmaArgs& p, hipStream_t st) {
    const int f = tun(g_tun.lstm_mfma16);
    if (f >= 0 ? f == 0 : ((p.M + 15) / 16) * p.ND < 160) return NIR_ERR_UNSUPPORTED;   // H = 70: 1024 seqs 178 vs 169 us, 3200 358 vs 509
    if ((int64_t)16 * p.T * max(p.I, p.ND * p.H) * 4 >= 0x7FFFFFF0LL || 16 * p.I > 1024) return NIR_ERR_UNSUPPORTED;
    const int G = (p.H + p.I + 15) / 16, NT = (p.H + 3) / 4 > 16 ? 2 : 1;
#define NIR_M16_CASE(g) if (G == g) return NT == 2 ? launch_mfma16<g, 2>(p, st) : launch_mfma16<g, 1>(p, st);
    NIR_M16_CASE(1) NIR_M16_CASE(2) NIR_M16_CASE(3) NIR_M16_CASE(4) NIR_M16_CASE(5)
    NIR_M16_CASE(6) NIR_M16_CASE(7) NIR_M16_CASE(8) NIR_M16_CASE(9) NIR_M16_CASE(10)
#undef NIR_M16_CASE
    return NIR_ERR_UNSUPPORTED;
}

// returns NIR_ERR_UNSUPPORTED when the shape has no instantiation (caller falls back to the VALU kernel)
int launch_bilstm_fused_mfma(const float* x, int I, const float* wih, const float* bih, const float* bhh, const int64_t* lens,
                             const float* whh, const float* h0, const float* c0, float* out, float* hn, float* cn, int64_t M,
                             int T, int H, int ND, hipStream_t st) {
    if (I > 64 || (int64_t)4 * T * max(I, ND * H) * 4 >= 0x7FFFFFF0LL) return NIR_ERR_UNSUPPORTED;
    LstmMfmaArgs p{x, wih, bih, bhh, lens, whh, h0, c0, out, hn, cn, M, T, H, ND, I, g_debug_buf};
    {   // enough sequences for 16-sequence workgroups on most CUs: the 16x16x4 layout
        const int rc = launch_bilstm_fused_mfma16(p, st);
        if (rc != NIR_ERR_UNSUPPORTED) return rc;
    }
    const int NG = (4 * H + 63) / 64;
    const int KQ = ((H + I + 15) / 16) * 4;
#define NIR_MFMA_CASE(ng, kq) if (NG == ng && KQ == kq) return launch_mfma<ng, kq>(p, st);
    NIR_MFMA_CASE(1, 8) NIR_MFMA_CASE(1, 12) NIR_MFMA_CASE(1, 16) NIR_MFMA_CASE(1, 20)
    NIR_MFMA_CASE(2, 12) NIR_MFMA_CASE(2, 16) NIR_MFMA_CASE(2, 20) NIR_MFMA_CASE(2, 24)
    NIR_MFMA_CASE(3, 16) NIR_MFMA_CASE(3, 20) NIR_MFMA_CASE(3, 24) NIR_MFMA_CASE(3, 28)
    NIR_MFMA_CASE(4, 20) NIR_MFMA_CASE(4, 24) NIR_MFMA_CASE(4, 28) NIR_MFMA_CASE(4, 32)
    NIR_MFMA_CASE(5, 24) NIR_MFMA_CASE(5, 28) NIR_MFMA_CASE(5, 32) NIR_MFMA_CASE(5, 36)
#undef NIR_MFMA_CASE
    return NIR_ERR_UNSUPPORTED;
}

}  // namespace nir
