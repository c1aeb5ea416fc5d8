// CARS session part (neuroir/multitask/cars.py:262-520): encode_clicks, encode_session, rank -- and, on request, the
// session outputs the suggestion decoder consumes (decoder-initialisation states and inner-attention pools, :382-456).
//
// The session states depend only on the queries and the clicks, never on the rank outputs, so the two session LSTM chains
// run first and cross attention + ranknet are evaluated ONCE for all (session, step) pairs.  Launch plan (C3: 15 launches,
// the reference issues ~25 torch ops x S steps and two host syncs):
//   1 click-attention MLP (GEMM, tanh + Linear(D,1) fused into the epilogue)          [doc chain on]
//   1 click_pool2 (batch-wide max click count, stable label sort, quirk mask, softmax, weighted sum)
//   1 U = pooled_q [W_sq^T | W_sd^T | b_sq | b_sd]: the cross-attention projection moved to the query side,
//       logit_k = (W s_k + b) . q  ==  s_k . (W^T q) + b . q   -- independent of the LSTM chains
//   S-1 session_lstm_step: BOTH chains in one launch; a workgroup owns 4 hidden units (16 gate rows, gate-interleaved so
//       one lane ends up with i,f,g,o of its (unit, session)), x W_ih^T + h W_hh^T on v_mfma_f32_16x16x4_f32 with K split
//       over the 4 waves, cell update in the same kernel -- no gate tensor, no separate cell kernel, no fill kernels
//   1 session_attend2 (softmax over the t+1 previous states incl. the zero state, weighted sums, [q; sq; sd] rows)
//   1 rank projection GEMM (W_q | W_shared + W_priv1 packed once per weight version), 1 feature kernel, 3 maxout GEMMs
#include "common.hpp"
#include <algorithm>

namespace nir {

int launch_linear(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                  int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                  int64_t ldc, int64_t M, int N, int K, int act, hipStream_t st);
int launch_linear_ex(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                     int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                     int64_t ldc, int64_t M, int N, int K, int act, const float* add, int64_t ldadd, hipStream_t st);
constexpr int ACT_MAXOUT2 = 16;
constexpr int ACT_BOUNDED = 0x100;      // (gemm.hip) operands bounded by 2^15: the split-precision GEMM may use its fp16 two-term form
constexpr int ACT_TANH_ROWDOT16 = 17;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave per (b,s) row, N <= 64: batch-wide m = max_rows count_nonzero(labels) (cars.py:285-289), stable descending
// rank by label, attend over {rank < count} U {rank >= m} (Appendix E2), logits e_k = sum of the NP epilogue partials + b3
__global__ __launch_bounds__(256) void click_pool2_kernel(const float* __restrict__ docs, const float* __restrict__ epart, int NP,
                                                          const float* __restrict__ b3, const float* __restrict__ labels,
                                                          const float* __restrict__ labels_all, int rows_all,
                                                          const int* __restrict__ mg, int rpg,
                                                          int rows, int N, int D, float* __restrict__ clicks) {
    // labels_all [rows_all, N]: the label matrix the batch-wide click count m is taken over -- the rows of this call, or (session-sharded
    // callers) those of the whole global batch, of which `labels` [rows, N] is this rank's block of sessions.
    // mg != NULL: the rows of this call come from several batches (rpg consecutive rows per batch) whose m was computed beforehand
    // (nir_cars_click_max): row r uses mg[r / rpg] and the scan below is skipped.
    __shared__ int part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (!mg) {   // every workgroup recomputes m from the (tiny) label matrix: no extra launch, no cross-workgroup dependency
        // (N <= 64: one label per lane, the row's count is the population count of a ballot -- a DPP wave_sum per row made this
        // prologue 112 x ~150 cycles per workgroup at the C5 shape)
        int best = 0;
        for (int r0 = wave; r0 < rows_all; r0 += 4 * 16) { // 16 independent row loads in flight per wave (each is an L2 round trip)
            float lv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {               // unconditional loads from clamped addresses (a predicated load becomes its own
                const int r = r0 + 4 * j;                // exec-masked block with a full wait behind it), masked afterwards
                lv[j] = labels_all[(int64_t)(r < rows_all ? r : 0) * N + (lane < N ? lane : 0)];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) best = max(best, (int)__popcll(__ballot(lv[j] != 0.f && lane < N && r0 + 4 * j < rows_all)));
        }
        if (lane == 0) part[wave] = best;
    }
    __syncthreads();
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const int m = mg ? mg[r / rpg] : max(max(part[0], part[1]), max(part[2], part[3]));
    const float lab = lane < N ? labels[(int64_t)r * N + lane] : -INFINITY;
    int rank = 0;
#pragma unroll 8
    for (int k = 0; k < N; ++k) {
        const float lk = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lab), k));   // k is wave-uniform: no LDS permute
        rank += (lk > lab) || (lk == lab && k < lane);
    }
    const int count = (int)wave_sum((lane < N && lab != 0.f) ? 1.f : 0.f);
    const bool keep = lane < N && (rank < count || rank >= m);
    float lg = -INFINITY;
    if (keep) {
        const float* lp = epart + ((int64_t)r * N + lane) * NP;
        float s = b3[0];
        for (int j = 0; j < NP; j += 4) {
            const float4 v = *reinterpret_cast<const float4*>(lp + j);
            s += (v.x + v.y) + (v.z + v.w);
        }
        lg = s;
    }
    const float mx = wave_max(lg);
    const float ex = keep ? expf(lg - mx) : 0.f;
    const float p = ex / wave_sum(ex);   // all masked -> NaN, exactly like softmax of all -inf in the reference
    const int nch = D >> 2;
    for (int c = lane; c < nch; c += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int k = 0; k < N; ++k) {                     // 8 independent 1 KB row reads in flight
            const float pk = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p), k));
            const float4 v = *reinterpret_cast<const float4*>(docs + ((int64_t)r * N + k) * D + 4 * c);
            acc.x = fmaf(pk, v.x, acc.x); acc.y = fmaf(pk, v.y, acc.y); acc.z = fmaf(pk, v.z, acc.z); acc.w = fmaf(pk, v.w, acc.w);
        }
        *reinterpret_cast<float4*>(clicks + (int64_t)r * D + 4 * c) = acc;
    }
}

// m[g] = max over the rows of group g of count_nonzero(labels[g, row, :])  (cars.py:285-289): one workgroup per group
__global__ __launch_bounds__(256) void click_max_kernel(const float* __restrict__ labels, int rows, int N, int* __restrict__ out) {
    __shared__ int part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* lg = labels + (int64_t)blockIdx.x * rows * N;
    int best = 0;
    if (N <= 64) {
        // a thread per row, all of a row's labels requested before the first is counted (one wave per row with a 64-lane reduction took 14.7 us
        // for 112 rows of 10 labels: 28 dependent round trips per wave)
        for (int r0 = 0; r0 < rows; r0 += 256) {
            const int r = r0 + (int)threadIdx.x;
            const float* row = lg + (int64_t)(r < rows ? r : rows - 1) * N;
            int c = 0;
#pragma unroll 8
            for (int k = 0; k < N; ++k) c += row[k] != 0.f ? 1 : 0;
            best = max(best, r < rows ? c : 0);
        }
        best = (int)wave_max((float)best);
    } else {
        for (int r = wave; r < rows; r += 4) {
            float c = 0.f;
            for (int k = lane; k < N; k += 64) c += lg[(int64_t)r * N + k] != 0.f ? 1.f : 0.f;
            best = max(best, (int)wave_sum(c));
        }
    }
    if (lane == 0) part[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = max(max(part[0], part[1]), max(part[2], part[3]));
}

// The same for N > 64 candidates (config.py:42: --num_candidates is free): a lane owns candidates lane, lane + 64, ..; the row's labels and
// softmax weights are staged in LDS (dynamic: 4 waves x 2 x N floats).  Not a hot shape: plain loops.
__global__ __launch_bounds__(256) void click_pool_big_kernel(const float* __restrict__ docs, const float* __restrict__ epart, int NP,
                                                             const float* __restrict__ b3, const float* __restrict__ labels,
                                                             const float* __restrict__ labels_all, int rows_all,
                                                             const int* __restrict__ mg, int rpg, int rows, int N, int D,
                                                             float* __restrict__ clicks) {
    extern __shared__ float csm[];
    __shared__ int part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (!mg) {
        int best = 0;
        for (int r = wave; r < rows_all; r += 4) {
            float c = 0.f;
            for (int k = lane; k < N; k += 64) c += labels_all[(int64_t)r * N + k] != 0.f ? 1.f : 0.f;
            best = max(best, (int)wave_sum(c));
        }
        if (lane == 0) part[wave] = best;
    }
    __syncthreads();
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;                                 // (no workgroup barrier below this line)
    const int m = mg ? mg[r / rpg] : max(max(part[0], part[1]), max(part[2], part[3]));
    float* lab = csm + (size_t)wave * 2 * N;
    float* pw = lab + N;
    float cnt = 0.f;
    for (int k = lane; k < N; k += 64) {
        const float v = labels[(int64_t)r * N + k];
        lab[k] = v;
        cnt += v != 0.f ? 1.f : 0.f;
    }
    const int count = (int)wave_sum(cnt);
    __builtin_amdgcn_wave_barrier();                       // one wave: its DS operations execute in issue order
    float mx = -INFINITY;
    for (int c0 = 0; c0 < N; c0 += 64) {
        const int c = c0 + lane;
        const float lc = c < N ? lab[c] : -INFINITY;
        int rank = 0;
        for (int k = 0; k < N; ++k) {
            const float lk = lab[k];                       // wave-uniform address: LDS broadcast
            rank += (lk > lc) || (lk == lc && k < c);
        }
        const bool keep = c < N && (rank < count || rank >= m);
        float lg = -INFINITY;
        if (keep) {
            const float* lp = epart + ((int64_t)r * N + c) * NP;
            float sacc = b3[0];
            for (int j = 0; j < NP; j += 4) {
                const float4 v = *reinterpret_cast<const float4*>(lp + j);
                sacc += (v.x + v.y) + (v.z + v.w);
            }
            lg = sacc;
        }
        if (c < N) pw[c] = lg;
        mx = fmaxf(mx, lg);
    }
    mx = wave_max(mx);
    float den = 0.f;
    for (int c = lane; c < N; c += 64) {
        const float lg = pw[c];
        const float ex = lg == -INFINITY ? 0.f : expf(lg - mx);
        pw[c] = ex;
        den += ex;
    }
    den = wave_sum(den);
    __builtin_amdgcn_wave_barrier();
    const int nch = D >> 2;
    for (int c = lane; c < nch; c += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int k = 0; k < N; ++k) {
            const float pk = pw[k] / den;                  // all masked -> 0 / 0 = NaN, exactly like softmax of all -inf in the reference
            const float4 v = *reinterpret_cast<const float4*>(docs + ((int64_t)r * N + k) * D + 4 * c);
            acc.x = fmaf(pk, v.x, acc.x); acc.y = fmaf(pk, v.y, acc.y); acc.z = fmaf(pk, v.z, acc.z); acc.w = fmaf(pk, v.w, acc.w);
        }
        *reinterpret_cast<float4*>(clicks + (int64_t)r * D + 4 * c) = acc;
    }
}

// One LSTM cell step for up to two independent chains in one launch (session LSTMs: cars.py:378-380, 400-402 via
// rnn_encoder.py:76-102; decoder LSTM: decoders/decoder.py:94-95).  A workgroup owns 4 hidden units = 16 gate rows,
// interleaved as row = 4*unit + gate so the 16x16 MFMA's C/D layout hands ONE lane the four gates i,f,g,o of its
// (unit, batch row): gates = x W_ih^T + h W_hh^T on v_mfma_f32_16x16x4_f32 with K split over the 4 waves, operands read
// straight from L2 as MFMA fragments, partial tiles summed through LDS, cell update in the same kernel.
// (LstmStepArgs: common.hpp)

// NB = batch tiles of 16 rows that share one pass over the weights (B <= 16 NB): a weight fragment is loaded once per k-group and
// feeds NB MFMA chains.  With the one-tile-at-a-time loop a 64-row batch (the C5 shape) walked the 6 MB of gate weights four times
// per step and chain: 27 us per step against 8 us at B = 16.
// UG = unit groups (of 4 units = one 16-row MFMA tile) per workgroup: the [rows, I + H] operand is re-read once per workgroup COLUMN -- at the
// greedy decoders' 768 rows and H = 512, 128 columns x 2.5 MB = 320 MB of L2 reads per step (78 us, 5 TB/s); two groups per workgroup halve it.
template <int NB, int UG = 1>
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmStepArgs p) {
    __shared__ float red[4][UG * NB][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int ch = blockIdx.y + p.chain0;
    const int u0 = blockIdx.x * 4 * UG;
    const int I = p.I, H = p.H;
    // A operand rows: row i = 4*unit_local + gate  ->  weight row gate*H + u0 + 4 ug + unit_local
    const float* wi[UG];
    const float* wh[UG];
#pragma unroll
    for (int ug = 0; ug < UG; ++ug) {
        const int ua = u0 + 4 * ug + (i >> 2);
        const int arow = (i & 3) * H + (ua < H ? ua : H - 1);
        wi[ug] = p.wih[ch] + (int64_t)arow * I;
        wh[ug] = p.whh[ch] + (int64_t)arow * H;
    }
    const float* hprev = p.hprev[ch];
    const float* cprev = p.cprev[ch];
    const float* gxp = p.gx[ch];
    const int nq1 = gxp ? 0 : ((I + 15) >> 4), nq2 = hprev ? ((H + 15) >> 4) : 0;
    // result view: lane = (batch column i, unit_local g), registers r = gates i,f,g,o
    float bias[UG][4];
#pragma unroll
    for (int ug = 0; ug < UG; ++ug) {
        const int ud = u0 + 4 * ug + g;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[ug][r] = (ud < H && !gxp) ? p.bih[ch][r * H + ud] + p.bhh[ch][r * H + ud] : 0.f;
    }
    // (batch slabs of 16 NB rows across workgroups -- blockIdx.z -- like lstm_step16_kernel: the greedy decoders step 768 rows at a macro-batch of 8)
    for (int b0 = blockIdx.z * 16 * NB; b0 < p.B; b0 += gridDim.z * 16 * NB) {
        const float* xr[NB];
        const float* hr[NB];
        float bm[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int b = b0 + 16 * t + i;
            const bool bv = b < p.B;
            bm[t] = bv ? 1.f : 0.f;                       // rows past the batch: clamped address, zeroed operand
            const int bc = bv ? b : p.B - 1;
            const int64_t xr_i = p.xid[ch] ? p.xid[ch][bc] : (int64_t)bc;
            xr[t] = p.x[ch] + xr_i * p.xstride[ch];
            hr[t] = hprev ? hprev + (int64_t)bc * H : nullptr;
        }
        f32x4 acc[UG][NB];
#pragma unroll
        for (int ug = 0; ug < UG; ++ug)
#pragma unroll
            for (int t = 0; t < NB; ++t) acc[ug][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // ONE list of k-groups over both products (x W_ih^T: groups 0 .. nq1-1, h W_hh^T: the rest), walked in chunks of CH per wave with
        // all of a chunk's operand loads issued before its first MFMA: at I + H = 768 and B <= 16 a wave's 12 groups are a single chunk,
        // i.e. ONE L2 / HBM round trip per step.  (Two separate walks -- 4 + 8 groups per wave in chunks of 4 -- were three dependent
        // round trips: 9.4 us per session step against ~6 MB of weights; `#pragma unroll` on the runtime-strided loop was not honoured
        // either.)  Unconditional loads from a clamped k, masked afterwards.
        constexpr int CH = NB == 1 ? 12 : (NB == 2 ? 8 : (UG == 1 ? 6 : 4));
        const int nqt = nq1 + nq2;
        for (int q0 = wave; q0 < nqt; q0 += 4 * CH) {
            float4 a4[CH][UG], b4[CH][NB];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int q = q0 + 4 * c;
                const bool second = q >= nq1;                 // wave-uniform
                const int K = second ? H : I;
                const int k = 16 * (second ? q - nq1 : q) + 4 * g;
                const float km = (q < nqt && k < K) ? 1.f : 0.f;
                const int kc = (q < nqt && k < K) ? k : 0;
#pragma unroll
                for (int ug = 0; ug < UG; ++ug) {
                    a4[c][ug] = *reinterpret_cast<const float4*>((second ? wh[ug] : wi[ug]) + kc);
                    a4[c][ug].x *= km; a4[c][ug].y *= km; a4[c][ug].z *= km; a4[c][ug].w *= km;
                }
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    b4[c][t] = *reinterpret_cast<const float4*>(((second && hr[t]) ? hr[t] : xr[t]) + kc);
                    b4[c][t].x *= bm[t]; b4[c][t].y *= bm[t]; b4[c][t].z *= bm[t]; b4[c][t].w *= bm[t];
                }
            }
            __builtin_amdgcn_sched_barrier(0);         // without it the scheduler sinks every load next to its first use again
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int ug = 0; ug < UG; ++ug)
#pragma unroll
                    for (int t = 0; t < NB; ++t) {
                        acc[ug][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c][ug].x, b4[c][t].x, acc[ug][t], 0, 0, 0);
                        acc[ug][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c][ug].y, b4[c][t].y, acc[ug][t], 0, 0, 0);
                        acc[ug][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c][ug].z, b4[c][t].z, acc[ug][t], 0, 0, 0);
                        acc[ug][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c][ug].w, b4[c][t].w, acc[ug][t], 0, 0, 0);
                    }
        }
#pragma unroll
        for (int ug = 0; ug < UG; ++ug)
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][ug * NB + t][r * 64 + lane] = acc[ug][t][r];
        __syncthreads();
        for (int s_ = wave; s_ < UG * NB; s_ += 4) {         // wave w finishes (unit group, batch tile) slots w, w + 4, ..
            const int ug = s_ / NB, t = s_ - ug * NB;
            const int b = b0 + 16 * t + i;
            const int ud = u0 + 4 * ug + g;
            const bool uv = ud < H;
            float g4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float gxv = 0.f;
                if (gxp) gxv = gxp[(int64_t)(b < p.B ? b : p.B - 1) * p.gxstride + (uv ? ud : 0) + r * H];
                float bsel = bias[0][r];
#pragma unroll
                for (int q = 1; q < UG; ++q) bsel = ug == q ? bias[q][r] : bsel;
                g4[r] = (bsel + gxv) + ((red[0][s_][r * 64 + lane] + red[1][s_][r * 64 + lane]) + (red[2][s_][r * 64 + lane] + red[3][s_][r * 64 + lane]));
            }
            if (b < p.B && uv) {
                const int64_t si = (int64_t)b * H + ud;
                const float c0 = cprev ? cprev[si] : 0.f;
                const float cn = fast_sigmoid(g4[1]) * c0 + fast_sigmoid(g4[0]) * fast_tanh(g4[2]);
                p.cnext[ch][si] = cn;
                p.hnext[ch][si] = fast_sigmoid(g4[3]) * fast_tanh(cn);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same step with the recurrent product on the fp16 matrix cores (two-term split, fp32-class: w = w1 + 2^-11 w2', h = h1 + 2^-11 h2',
// w.h = w1.h1 + 2^-11 (w1.h2' + w2'.h1), see lstm_fold.hip).  At the session shapes the fp32-MFMA form above is bound by its own matrix
// work: B = 128 rows x 4 HS x HS per chain and step is 1 024 v_mfma_f32_16x16x4_f32 per workgroup (32 cycles each) in two passes over
// the weights -- 23.6 us per step, six steps per tail.  Here a k-block of 32 is three 16-cycle MFMAs, W_hh arrives pre-split in lane
// order (no conversion), h arrives as the term pairs the previous step wrote, and the input side comes from the hoisted GEMM (gx).
// ---------------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));

// UG unit groups (4 units each) per workgroup: every workgroup reads the WHOLE previous state (B x H term pairs) as its B operand, so with 4
// units per workgroup (H / 4 workgroups per chain) the state was fetched from L2 128 times per chain and step -- 32 MB per pass at B = 64
// against 4 MB of weights; a k-block's state fragments now feed UG x 3 MFMAs per batch tile.
template <int NB, int UG, int CK>
__global__ __launch_bounds__(256) void lstm_step16_kernel(LstmStepArgs p) {
    __shared__ float red[4][UG][NB][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int ch = blockIdx.y + p.chain0;
    const int H = p.H, KB = H >> 5, H8 = H >> 3;
    const int NUG = H >> 2;                                   // unit groups of the chain
    const _Float16* hp16 = p.h16prev[ch];
    const float* cprev = p.cprev[ch];
    const float* gxp = p.gx[ch];
    // result view: lane = (batch column i, unit_local g), registers r = gates i,f,g,o
    const f16x8s* wf = reinterpret_cast<const f16x8s*>(p.whh_frag[ch]) + lane;
    size_t wbase[UG];
    int udv[UG];
#pragma unroll
    for (int u = 0; u < UG; ++u) {
        const int ug = blockIdx.x * UG + u;
        wbase[u] = (size_t)(ug < NUG ? ug : NUG - 1) * KB * 2 * 64;       // groups past the end (H / 4 not a multiple of UG): clamped, never stored
        udv[u] = 4 * ug + g;
    }
    // epilogue work items: (unit group u, batch tile t) pairs, item e = u * NB + t handled by wave e % 4
    constexpr int NE = UG * NB, EPW = (NE + 3) / 4;
    // batch tiles beyond the first 16 NB rows: one workgroup per slab (blockIdx.z) -- the slabs of a step are independent, and a second pass
    // inside the workgroup was a second serialised round trip to L2
    for (int b0 = blockIdx.z * 16 * NB; b0 < p.B; b0 += gridDim.z * 16 * NB) {
        float gxv[EPW][4];
#pragma unroll
        for (int k = 0; k < EPW; ++k) {
            const int e = wave + 4 * k, u = e / NB, t = e % NB;       // (u, t) are compile-time per k only up to the wave offset: plain integer math
            const int bq = b0 + 16 * t + i;
            const int ud = 4 * (blockIdx.x * UG + u) + g;
            const int64_t brow = bq < p.B ? bq : p.B - 1;
            const int udc = ud < H ? ud : H - 1;
            int64_t grow = p.gxid[ch] ? p.gxid[ch][brow] : brow;
            if (p.gxkey && ch == 0) {                                       // (wave-uniform) the previous decode step's arg-max key -> source token id
                const ulonglong2* kb_ = reinterpret_cast<const ulonglong2*>(p.gxkey + brow * ARGMAX_KEY_BUCKETS);
                unsigned long long km_ = 0ull;
#pragma unroll
                for (int q_ = 0; q_ < ARGMAX_KEY_BUCKETS / 2; ++q_) {
                    const ulonglong2 kk_ = kb_[q_];
                    km_ = kk_.x > km_ ? kk_.x : km_;
                    km_ = kk_.y > km_ ? kk_.y : km_;
                }
                const int64_t w_ = argmax_key_index(km_);
                const int64_t nx_ = p.gxmap ? p.gxmap[w_] : w_;
                grow = (nx_ >= 0 && nx_ < p.gxV) ? nx_ : 1;
            }
            const float* gr = gxp + grow * p.gxstride;
            if (p.gx_unit_major) {
                const float4 v = *reinterpret_cast<const float4*>(gr + 4 * udc);
                gxv[k][0] = v.x; gxv[k][1] = v.y; gxv[k][2] = v.z; gxv[k][3] = v.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) gxv[k][r] = gr[udc + r * H];
            }
            if (e >= NE) gxv[k][0] = gxv[k][1] = gxv[k][2] = gxv[k][3] = 0.f;
        }
        f32x4 acc[UG][NB], acx[UG][NB];
#pragma unroll
        for (int u = 0; u < UG; ++u)
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                acc[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acx[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        if (hp16) {
            const _Float16* hr[NB];
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                const int b = b0 + 16 * t + i;
                hr[t] = hp16 + ((int64_t)(b < p.B ? b : p.B - 1) * H8 + g) * 16;      // rows past the batch: clamped (their columns are never stored)
            }
            for (int q0 = wave; q0 < KB; q0 += 4 * CK) {
                f16x8s w1[CK][UG], w2[CK][UG], h1[CK][NB], h2[CK][NB];
#pragma unroll
                for (int c = 0; c < CK; ++c) {
                    const int kb = q0 + 4 * c;
                    const int kc = kb < KB ? kb : 0;             // past the end: re-read block 0, skipped below (wave-uniform)
#pragma unroll
                    for (int u = 0; u < UG; ++u) {
                        w1[c][u] = wf[wbase[u] + (size_t)(kc * 2) * 64];
                        w2[c][u] = wf[wbase[u] + (size_t)(kc * 2 + 1) * 64];
                        if (kb >= KB) {                          // past the end: zero weights, the MFMAs below run unconditionally (a wave-uniform
                            w1[c][u] = f16x8s{};                 // `continue` around them cost phi copies of every accumulator: 693 v_accvgpr moves,
                            w2[c][u] = f16x8s{};                 // 512 registers and 32 bytes of scratch per lane in the <4, 4, 2> instantiation;
                        }                                        // 391 registers and none now.  Slabs of 96 rows (<6, 4>: 256 workgroups at 768
                    }                                            // rows, one round over the chip) still spill 140-190 bytes: not built)
#pragma unroll
                    for (int t = 0; t < NB; ++t) {
                        h1[c][t] = *reinterpret_cast<const f16x8s*>(hr[t] + (int64_t)kc * 64);
                        h2[c][t] = *reinterpret_cast<const f16x8s*>(hr[t] + (int64_t)kc * 64 + 8);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < CK; ++c) {
#pragma unroll
                    for (int u = 0; u < UG; ++u)
#pragma unroll
                        for (int t = 0; t < NB; ++t) {
                            acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[c][u], h1[c][t], acc[u][t], 0, 0, 0);
                            acx[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[c][u], h2[c][t], acx[u][t], 0, 0, 0);
                            acx[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[c][u], h1[c][t], acx[u][t], 0, 0, 0);
                        }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UG; ++u)
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][u][t][r * 64 + lane] = fmaf(acx[u][t][r], 1.0f / 2048.0f, acc[u][t][r]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EPW; ++k) {
            const int e = wave + 4 * k, u = e / NB, t = e % NB;
            if (e >= NE) break;
            const int b = b0 + 16 * t + i;
            const int ud = 4 * (blockIdx.x * UG + u) + g;
            float g4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                g4[r] = gxv[k][r] + ((red[0][u][t][r * 64 + lane] + red[1][u][t][r * 64 + lane]) + (red[2][u][t][r * 64 + lane] + red[3][u][t][r * 64 + lane]));
            if (b < p.B && ud < H) {
                const int64_t si = (int64_t)b * H + ud;
                const float c0 = cprev ? cprev[si] : 0.f;
                const float cn = fast_sigmoid(g4[1]) * c0 + fast_sigmoid(g4[0]) * fast_tanh(g4[2]);
                const float hn = fast_sigmoid(g4[3]) * fast_tanh(cn);
                p.cnext[ch][si] = cn;
                p.hnext[ch][si] = hn;
                const _Float16 a = (_Float16)hn;                 // the next step's B operand: the two fp16 terms
                _Float16* d = p.h16next[ch] + ((int64_t)b * H8 + (ud >> 3)) * 16 + (ud & 7);
                d[0] = a;
                d[8] = (_Float16)((hn - (float)a) * 2048.0f);
            }
        }
        __syncthreads();
    }
    (void)udv;
}

// W_hh [4H, H] -> [H/4 unit groups][H/32 k-blocks][2 terms][64 lanes][8]: lane (i, g) of unit group ug holds k = 32 kb + 8 g .. + 7 of weight row
// (i & 3) * H + 4 ug + (i >> 2) -- the A fragment of lstm_step16_kernel.  err |= 2 when a weight is outside the split's range (|w| >= 2^15 or NaN).
__global__ __launch_bounds__(64) void lstm_step_whh_frag_kernel(const float* __restrict__ whh, int H, _Float16* __restrict__ out, int* __restrict__ err) {
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const int ug = blockIdx.x, kb = blockIdx.y, KB = H >> 5;
    const float* wr = whh + ((int64_t)(i & 3) * H + 4 * ug + (i >> 2)) * H + 32 * kb + 8 * g;
    _Float16* o = out + (((int64_t)ug * KB + kb) * 2 * 64 + lane) * 8;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float w = wr[j];
        const _Float16 a = (_Float16)w;
        o[j] = a;
        o[64 * 8 + j] = (_Float16)((w - (float)a) * 2048.0f);
        bad |= !(fabsf(w) < 32768.0f);
    }
    if (bad && err) atomicOr(err, 2);
}

int launch_lstm_step(const LstmStepArgs& a, int nchains, hipStream_t st) {
    NIR_REQUIRE(a.I % 4 == 0 && a.H % 4 == 0 && a.B >= 0, "lstm_step: I and H must be multiples of 4");
    if (a.B == 0) return 0;
    const dim3 grid((unsigned)((a.H + 3) / 4), (unsigned)nchains);
    bool f16 = a.H % 32 == 0 && !tun(g_tun.exact_f32);
    for (int c = a.chain0; c < a.chain0 + nchains; ++c)
        f16 = f16 && a.whh_frag[c] && a.gx[c] && a.h16next[c] && (a.hprev[c] == nullptr || a.h16prev[c] != nullptr);
    if (f16) {
        ProfScope ps("lstm_step16_kernel", st);
        // unit groups per workgroup / batch tiles per workgroup: measured at the session shapes (HS = 512, B = 64 / 128): more, smaller workgroups win --
        // the step is a latency chain (one L2 round trip, a few dozen MFMAs, an LDS reduction), not a bandwidth problem: UG 1 / 2 / 4 = 12.5 / 13.8 /
        // 18.4 us per bench step at B = 128.  Tunables lstm_step_ug / lstm_step_nb override (tools, tests).
        const int ugsel = tun(g_tun.lstm_step_ug), nbsel = tun(g_tun.lstm_step_nb);
        const int NUG = a.H / 4;
        int NBv = nbsel ? nbsel : (a.B > 32 ? 4 : (a.B > 16 ? 2 : 1));
        NBv = NBv >= 4 ? 4 : (NBv >= 2 ? 2 : 1);
        // (from 256 rows on -- the greedy decoders' 768 -- the [rows, H] state operand is re-read once per workgroup column: four unit groups per workgroup)
        const int ugd = ugsel ? ugsel : (a.B >= 256 ? 4 : 1);
        const int UGv = ugd == 4 ? 4 : (ugd == 2 ? 2 : 1);
        const dim3 gridu((unsigned)((NUG + UGv - 1) / UGv), (unsigned)nchains, (unsigned)((a.B + 16 * NBv - 1) / (16 * NBv)));
        if (NBv == 4) {
            if (UGv == 4) hipLaunchKernelGGL((lstm_step16_kernel<4, 4, 2>), gridu, dim3(256), 0, st, a);
            else if (UGv == 2) hipLaunchKernelGGL((lstm_step16_kernel<4, 2, 4>), gridu, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((lstm_step16_kernel<4, 1, 4>), gridu, dim3(256), 0, st, a);
        } else if (NBv == 2) {
            if (UGv >= 2) hipLaunchKernelGGL((lstm_step16_kernel<2, 2, 4>), gridu, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((lstm_step16_kernel<2, 1, 8>), gridu, dim3(256), 0, st, a);
        } else {
            if (UGv >= 2) hipLaunchKernelGGL((lstm_step16_kernel<1, 2, 4>), gridu, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((lstm_step16_kernel<1, 1, 8>), gridu, dim3(256), 0, st, a);
        }
        NIR_CHECK_LAUNCH("lstm_step16_kernel");
        return 0;
    }
    ProfScope ps("lstm_step_kernel", st);
    {
        const int nbv = a.B > 32 ? 4 : (a.B > 16 ? 2 : 1);
        const dim3 gridz(grid.x, grid.y, (unsigned)((a.B + 16 * nbv - 1) / (16 * nbv)));
        if (a.B >= 256 && a.H % 8 == 0) {               // many row slabs: two unit groups per workgroup (half the re-reads of the row operand)
            const dim3 grid2((unsigned)((a.H + 7) / 8), grid.y, gridz.z);
            hipLaunchKernelGGL((lstm_step_kernel<4, 2>), grid2, dim3(256), 0, st, a);
        }
        else if (a.B > 32) hipLaunchKernelGGL(lstm_step_kernel<4>, gridz, dim3(256), 0, st, a);
        else if (a.B > 16) hipLaunchKernelGGL(lstm_step_kernel<2>, gridz, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(lstm_step_kernel<1>, gridz, dim3(256), 0, st, a);
        NIR_CHECK_LAUNCH("lstm_step_kernel");
        return 0;
    }
}

// Cross attention over the session states (cars.py:348-366) for one (session b, step t) per workgroup, projection folded
// onto the query side: U[bt] = [W_sq^T q | W_sd^T q | b_sq.q | b_sd.q]; logit_k = s_k . U_chain + bias term, k = 0..t with
// s_0 = 0; out = sum_k softmax(logit)_k s_k.  Writes xcat[bt] = [q ; sq ; sd] (only the chains that are on).
__global__ __launch_bounds__(256) void session_attend2_kernel(const float* __restrict__ U, int NU, const float* __restrict__ Qs,
                                                              const float* __restrict__ Ds, const float* __restrict__ q,
                                                              int B, int S, int D, int HS, int q_on, int d_on,
                                                              float* __restrict__ xcat) {
    extern __shared__ float lgs[];                       // [2 chains][S] logits of the <= S previous states
    float* lg[2] = {lgs, lgs + S};
    const int bt = blockIdx.x, b = bt / S, t = bt % S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = q_on + d_on;
    const int HSQ = q_on ? HS : 0;
    const float* ur = U + (int64_t)bt * NU;
    const float* qv = q + (int64_t)bt * D;
    float* orow = xcat + (int64_t)bt * (D + nch * HS);
    // chain slot c (0..nch-1): states pointer, U column offset, bias column
    for (int e = wave; e < nch * t; e += 4) {
        const int c = e / t, k = e % t + 1;
        const float* st = (c == 0 && q_on) ? Qs : Ds;
        const float* sv = st + ((int64_t)k * B + b) * HS;
        const float* uv = ur + c * HS;
        float sacc = 0.f;
        for (int f = 4 * lane; f < HS; f += 256) {
            const float4 a = *reinterpret_cast<const float4*>(sv + f), w = *reinterpret_cast<const float4*>(uv + f);
            sacc += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
        }
        sacc = wave_sum(sacc);
        if (lane == 0) lg[c][k] = sacc + ur[nch * HS + c];
    }
    if (threadIdx.x < nch) lg[threadIdx.x][0] = ur[nch * HS + threadIdx.x];
    for (int f = threadIdx.x; f < D; f += 256) orow[f] = qv[f];
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const float* st = (c == 0 && q_on) ? Qs : Ds;
        float mx = -INFINITY;
        for (int k = 0; k <= t; ++k) mx = fmaxf(mx, lg[c][k]);
        float den = 0.f;
        for (int k = 0; k <= t; ++k) den += expf(lg[c][k] - mx);
        for (int f = threadIdx.x; f < HS; f += 256) {
            float acc = 0.f;
            for (int k = 1; k <= t; ++k) acc = fmaf(expf(lg[c][k] - mx) / den, st[((int64_t)k * B + b) * HS + f], acc);
            orow[D + c * HS + f] = acc;
        }
    }
    (void)HSQ;
}

// Pack-time weights: wrank[o,:] = [W_q[o,:] | W_shared[o,:] + W_priv1[o,:]]  ([D, D + KS]);
//                    ut = [W_sq^T ; W_sd^T ; b_sq ; b_sd]                      ([KS + nch, D]) for the U GEMM.
__global__ void session_pack_kernel(const float* wq, const float* wshared, const float* wpriv, const float* sqw, const float* sqb,
                                    const float* sdw, const float* sdb, int D, int HS, int q_on, int d_on, float* wrank, float* ut) {
    const int nch = q_on + d_on, KS = nch * HS, KR = D + KS;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)D * KR) {
        const int o = (int)(i / KR), k = (int)(i % KR);
        float v;
        if (k < D) v = wq[(int64_t)o * D + k];
        else v = wshared[(int64_t)o * KS + (k - D)] + wpriv[(int64_t)o * KS + (k - D)];
        wrank[i] = v;
    }
    if (i < (int64_t)(KS + nch) * D) {
        const int j = (int)(i / D), f = (int)(i % D);
        float v;
        if (j < KS) {
            const int c = j / HS, jj = j % HS;
            const float* w = (c == 0 && q_on) ? sqw : sdw;        // [D, HS]
            v = w[(int64_t)f * HS + jj];
        } else {
            const int c = j - KS;
            v = ((c == 0 && q_on) ? sqb : sdb)[f];
        }
        ut[i] = v;
    }
}

// feats[(b,t,n)] = [q', d, |q'-d|, q'*d]   (cars.py:514-518); q' row = (b,t), d row = (b,t,n)
__global__ void rank_feats_kernel(const float* qp, const float* docs, int N, int D, int64_t rows, float* feats) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i < rows * D) {
        const int64_t r = i / D;
        const int f = (int)(i % D);
        const float4 q = *reinterpret_cast<const float4*>(qp + (r / N) * D + f), d = *reinterpret_cast<const float4*>(docs + r * D + f);
        float* o = feats + r * 4 * D + f;
        *reinterpret_cast<float4*>(o) = q;
        *reinterpret_cast<float4*>(o + D) = d;
        *reinterpret_cast<float4*>(o + 2 * D) = make_float4(fabsf(q.x - d.x), fabsf(q.y - d.y), fabsf(q.z - d.z), fabsf(q.w - d.w));
        *reinterpret_cast<float4*>(o + 3 * D) = make_float4(q.x * d.x, q.y * d.y, q.z * d.z, q.w * d.w);
    }
}

// inner self-attention pools over the states produced so far (cars.py:385-388, 407-410), suggestion side only:
// inner[b,t] = sum_{k=1..t+1} softmax_k(l_k) s_k,  l_k = sum of the NP epilogue partials of state row (k,b) + b3;
// `states` and `lpart` both start at state 1
__global__ __launch_bounds__(256) void session_inner_pool_kernel(const float* __restrict__ states, const float* __restrict__ lpart,
                                                                 int NP, const float* __restrict__ b3, int B, int S, int HS,
                                                                 float* __restrict__ out /*[B,S,HS]*/) {
    extern __shared__ float lg[];                         // [S]
    const int bt = blockIdx.x, b = bt / S, t = bt % S;
    const int n = t + 1;                                  // states 1..t+1
    for (int k = threadIdx.x; k < n; k += 256) {
        const float* lp = lpart + ((int64_t)k * B + b) * NP;               // lpart row 0 = state 1
        float s = b3[0];
        for (int j = 0; j < NP; ++j) s += lp[j];
        lg[k] = s;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int k = 0; k < n; ++k) mx = fmaxf(mx, lg[k]);
    float den = 0.f;
    for (int k = 0; k < n; ++k) den += expf(lg[k] - mx);
    for (int f = threadIdx.x; f < HS; f += 256) {
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc = fmaf(expf(lg[k] - mx) / den, states[((int64_t)k * B + b) * HS + f], acc);
        out[(int64_t)bt * HS + f] = acc;
    }
}

// cat[(t,b)] = [hq_{t+1}[b] ; hd_{t+1}[b]]  for t = 0..S-2  (the reference concatenates the per-step states along the batch
// axis in step-major order, cars.py:431-436 -- kept as is, including the resulting (step, session) row order)
__global__ void session_cat_states_kernel(const float* a, const float* b2, int rows, int HSa, int HSb, float* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int W = HSa + HSb;
    if (i < (int64_t)rows * W) {
        const int r = (int)(i / W), f = (int)(i % W);
        out[i] = f < HSa ? a[(int64_t)r * HSa + f] : b2[(int64_t)r * HSb + (f - HSa)];
    }
}

struct SessPlan {
    float *epart, *clicks, *Qs, *Ds, *Cq, *Cd, *U, *xcat, *qp, *feats, *y0, *y1, *lin, *cat, *gx, *Qs16, *Ds16;
    size_t bytes;
};
static SessPlan sess_plan(void* ws, size_t cap, int B, int S, int N, int D, int HS, int nch, bool rank_on, bool want_states) {
    Workspace a(ws, cap);
    SessPlan p;
    const size_t R = (size_t)B * S * N, BS = (size_t)B * S;
    p.epart = a.take<float>(R * (D / 16));
    p.clicks = a.take<float>(BS * D);
    p.Qs = a.take<float>((size_t)(S + 1) * B * HS);
    p.Ds = a.take<float>((size_t)(S + 1) * B * HS);
    p.Cq = a.take<float>((size_t)(S + 1) * B * HS);
    p.Cd = a.take<float>((size_t)(S + 1) * B * HS);
    p.U = a.take<float>(BS * (size_t)(nch * HS + nch + 4));
    p.xcat = a.take<float>(BS * (size_t)(D + nch * HS));
    p.qp = a.take<float>(BS * D);
    p.feats = a.take<float>(rank_on ? R * 4 * D : 0);
    p.y0 = a.take<float>(rank_on ? R * 256 : 0);
    p.y1 = a.take<float>(rank_on ? R * 128 : 0);
    p.lin = a.take<float>(want_states ? (size_t)(S + 1) * B * (HS / 16) : 0);
    p.cat = a.take<float>(want_states ? (size_t)S * B * nch * HS : 0);
    p.gx = a.take<float>((size_t)nch * BS * 4 * HS);            // input side of the session LSTM gates, every step at once
    p.Qs16 = a.take<float>((size_t)(S + 1) * B * HS);          // the session states again as fp16 term pairs (4 B per element)
    p.Ds16 = a.take<float>((size_t)(S + 1) * B * HS);
    p.bytes = align_up(a.off, 256);
    return p;
}

static inline dim3 g1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace nir

extern "C" size_t nir_lstm_step_whh_frag_bytes(int H) { return (H > 0 && H % 32 == 0) ? (size_t)4 * H * H * 2 * sizeof(_Float16) : 0; }

extern "C" int nir_lstm_step_pack_whh_frag(const float* w_hh, int H, void* frag, int* err_flag, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(w_hh && frag && H > 0 && H % 32 == 0, "lstm_step_pack_whh_frag: H must be a positive multiple of 32");
    hipLaunchKernelGGL(lstm_step_whh_frag_kernel, dim3((unsigned)(H / 4), (unsigned)(H / 32)), dim3(64), 0, (hipStream_t)stream, w_hh, H, (_Float16*)frag, err_flag);
    NIR_CHECK_LAUNCH("lstm_step_whh_frag_kernel");
    return 0;
}

extern "C" size_t nir_cars_session_pack_floats(const nir_cars_session_weights* w, size_t* wrank_floats, size_t* ut_floats) {
    if (!w) return 0;
    const int nch = (w->q_on ? 1 : 0) + (w->d_on ? 1 : 0);
    const size_t a = (size_t)w->D * (w->D + nch * w->HS), b = (size_t)(nch * w->HS + nch) * w->D;
    if (wrank_floats) *wrank_floats = a;
    if (ut_floats) *ut_floats = b;
    return a + b;
}

extern "C" int nir_cars_session_pack(const nir_cars_session_weights* w, float* wrank, float* ut, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(w && wrank && ut, "cars_session_pack: null pointer");
    const int q_on = w->q_on ? 1 : 0, d_on = w->d_on ? 1 : 0, nch = q_on + d_on;
    NIR_REQUIRE(w->qproj_w, "cars_session_pack: q_projection missing (ranker off?)");
    NIR_REQUIRE(nch == 0 || (w->shared_w && w->priv1_w), "cars_session_pack: session projectors missing");
    const int64_t n = std::max((int64_t)w->D * (w->D + nch * w->HS), (int64_t)(nch * w->HS + nch) * w->D);
    hipLaunchKernelGGL(session_pack_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, w->qproj_w, w->shared_w, w->priv1_w, w->sq_attn_w,
                       w->sq_attn_b, w->sd_attn_w, w->sd_attn_b, w->D, w->HS, q_on, d_on, wrank, ut);
    NIR_CHECK_LAUNCH("session_pack_kernel");
    return 0;
}

extern "C" size_t nir_cars_session_workspace_bytes(int B, int S, int N, const nir_cars_session_weights* w) {
    if (!w || B < 0 || S <= 0 || N <= 0) return 0;
    const int nch = (w->q_on ? 1 : 0) + (w->d_on ? 1 : 0);
    return nir::sess_plan(nullptr, 0, B, S, N, w->D, w->HS, nch, true, true).bytes;
}

extern "C" int nir_cars_rank_session(const float* pooled_q, const float* pooled_docs, const float* labels, int B, int S, int N,
                                     const nir_cars_session_weights* w, void* workspace, size_t workspace_bytes,
                                     float* click_scores, float* clicks_out, const nir_cars_session_outputs* extra,
                                     nir_stream_t stream) {
    return nir_cars_rank_session_rows(pooled_q, pooled_docs, labels, B, S, N, w, workspace, workspace_bytes, click_scores, clicks_out, extra,
                                      nullptr, 0, nullptr, 0, nullptr, 0, stream);
}

extern "C" int nir_cars_rank_session_shard(const float* pooled_q, const float* pooled_docs, const float* labels, int B, int S, int N,
                                           const nir_cars_session_weights* w, void* workspace, size_t workspace_bytes,
                                           float* click_scores, float* clicks_out, const nir_cars_session_outputs* extra,
                                           const float* rank_docs, int NR, nir_stream_t stream) {
    return nir_cars_rank_session_rows(pooled_q, pooled_docs, labels, B, S, N, w, workspace, workspace_bytes, click_scores, clicks_out, extra,
                                      rank_docs, NR, nullptr, 0, nullptr, 0, stream);
}

extern "C" int nir_cars_click_max(const float* labels, int groups, int rows, int N, int* m_out, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(labels && m_out && groups > 0 && rows > 0 && N > 0, "cars_click_max: bad args");
    hipLaunchKernelGGL(click_max_kernel, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream, labels, rows, N, m_out);
    NIR_CHECK_LAUNCH("click_max_kernel");
    return 0;
}

extern "C" int nir_cars_rank_session_rows(const float* pooled_q, const float* pooled_docs, const float* labels, int B, int S, int N,
                                          const nir_cars_session_weights* w, void* workspace, size_t workspace_bytes,
                                          float* click_scores, float* clicks_out, const nir_cars_session_outputs* extra,
                                          const float* rank_docs, int NR, const float* labels_all, int64_t rows_all, const int* m_groups,
                                          int sessions_per_group, nir_stream_t stream) {
    return nir_cars_rank_session_pre(pooled_q, pooled_docs, labels, B, S, N, w, workspace, workspace_bytes, click_scores, clicks_out, extra, rank_docs, NR,
                                     labels_all, rows_all, m_groups, sessions_per_group, nullptr, nullptr, stream);
}

// The two GEMMs of the tail that read the pooled QUERIES only (cars.py:346-361 keys the session attention by the query; :364-378 feeds it to the query
// chain): U = pooled_q [W_sq^T | W_sd^T | b_sq | b_sd] and the query chain's hoisted input projection gq.  A caller that encodes the queries on a side
// stream next to the document encoder (wrappers.Multitask._rank under capture) issues them there and hands the results to
// nir_cars_rank_session_pre: 10.7 us off the one-batch critical path, no additional cross-stream edge.
extern "C" int nir_cars_session_query_side(const float* pooled_q, int B, int S, const nir_cars_session_weights* w, float* U, float* gq,
                                           nir_stream_t stream) {
    using namespace nir;
    hipStream_t st = (hipStream_t)stream;
    NIR_REQUIRE(pooled_q && w && B >= 0 && S > 0, "cars_session_query_side: bad arguments");
    if (B == 0) return 0;
    const bool q_on = w->q_on != 0, d_on = w->d_on != 0, rank_on = w->rank_on != 0;
    const int nch = (q_on ? 1 : 0) + (d_on ? 1 : 0), D = w->D, HS = w->HS, NU = nch * HS + nch;
    const int64_t BS = (int64_t)B * S;
    const int bnd = (w->rank_bounded & 8) ? ACT_BOUNDED : 0;
    NIR_REQUIRE(!(nch && rank_on) || (U && w->attn_ut), "cars_session_query_side: U / packed attention weights missing");
    NIR_REQUIRE(!(nch && q_on) || gq, "cars_session_query_side: gq missing");
    if (nch && rank_on)
        NIR_PROPAGATE(launch_linear(pooled_q, D, nullptr, nullptr, 0, 0, 0, w->attn_ut, D, nullptr, nullptr, U, NU, BS, NU, D, NIR_ACT_NONE, st));
    if (nch && q_on)
        NIR_PROPAGATE(launch_linear_ex(pooled_q, D, nullptr, nullptr, 0, 0, 0, w->sq_wih, D, w->sq_bih, w->sq_bhh, gq, 4 * HS, BS, 4 * HS, D, NIR_ACT_NONE | bnd, nullptr, 0, st));
    return 0;
}

extern "C" int nir_cars_rank_session_pre(const float* pooled_q, const float* pooled_docs, const float* labels, int B, int S, int N,
                                         const nir_cars_session_weights* w, void* workspace, size_t workspace_bytes,
                                         float* click_scores, float* clicks_out, const nir_cars_session_outputs* extra,
                                         const float* rank_docs, int NR, const float* labels_all, int64_t rows_all, const int* m_groups,
                                         int sessions_per_group, const float* pre_U, const float* pre_gq, nir_stream_t stream) {
    using namespace nir;
    hipStream_t st = (hipStream_t)stream;
    NIR_REQUIRE(pooled_q && w, "cars_rank_session: null pointer");
    const bool q_on = w->q_on != 0, d_on = w->d_on != 0, rank_on = w->rank_on != 0;
    const int nch = (q_on ? 1 : 0) + (d_on ? 1 : 0);
    NIR_REQUIRE(!rank_on || (pooled_docs && click_scores && w->wrank), "cars_rank_session: ranker on needs documents, scores and packed weights");
    NIR_REQUIRE(!d_on || (pooled_docs && labels), "cars_rank_session: the document session needs documents and labels");
    NIR_REQUIRE(nch == 0 || !rank_on || w->attn_ut, "cars_rank_session: packed attention weights missing (nir_cars_session_pack)");
    NIR_REQUIRE(B >= 0 && S > 0 && N > 0, "cars_rank_session: bad dims");
    NIR_REQUIRE(N <= 2048, "cars_rank_session: %d candidates > 2048 unsupported", N);
    NIR_REQUIRE(!labels_all || (rows_all >= (int64_t)B * S && rows_all < (1 << 30)), "cars_rank_session: labels_all must hold at least the B*S rows of this call");
    NIR_REQUIRE(!rank_docs || (NR > 0 && NR <= N), "cars_rank_session: the ranked candidate slice must hold 1..N candidates (got %d)", NR);
    NIR_REQUIRE(S <= 4096, "cars_rank_session: session length %d > 4096 unsupported", S);
    NIR_REQUIRE(!m_groups || (sessions_per_group > 0 && !labels_all), "cars_rank_session: m_groups needs sessions_per_group > 0 and excludes labels_all");
    NIR_REQUIRE(w->D % 64 == 0 && w->HS % 16 == 0 && w->D % 16 == 0, "cars_rank_session: D %% 64 / HS %% 16 required");
    if (B == 0) return 0;
    const int D = w->D, HS = w->HS, NP = D / 16;
    const bool want_states = extra != nullptr;
    SessPlan p = sess_plan(workspace, workspace_bytes, B, S, N, D, HS, nch, rank_on, want_states);
    if (!workspace || p.bytes > workspace_bytes) {
        set_error("cars_rank_session: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
        return NIR_ERR_WORKSPACE;
    }
    const int64_t BS = (int64_t)B * S, R = BS * N;
    float* clicks = clicks_out ? clicks_out : p.clicks;
    const int NU = nch * HS + nch;
    const float* gq = (pre_gq && q_on) ? pre_gq : p.gx;
    float* gd = p.gx + (q_on ? BS * 4 * (int64_t)HS : 0);
    const float* Uq = (pre_U && nch && rank_on) ? pre_U : p.U;
    const int bnd = (w->rank_bounded & 8) ? ACT_BOUNDED : 0;
    // (Round 6, measured and NOT kept: the two GEMMs that read the pooled QUERIES only -- the attention projection U and the query chain's hoisted
    // input projection -- on a forked stream beside the click MLP / click pooling / the document chain's projection.  Inside a hipGraph every
    // cross-queue edge costs ~9 us (fork 9.3 + join 8.7 us in tools/iter_timeline.py) against the ~10 us the two launches take: 120 -> 123 us.)
    {
        hipStream_t sq_ = st;
        // ---- U = pooled_q [W_sq^T | W_sd^T | b_sq | b_sd]  (independent of the chains)
        if (nch && rank_on && Uq == p.U)
            NIR_PROPAGATE(launch_linear(pooled_q, D, nullptr, nullptr, 0, 0, 0, w->attn_ut, D, nullptr, nullptr, p.U, NU, BS, NU, D, NIR_ACT_NONE, sq_));
        if (nch && q_on && gq == p.gx) NIR_PROPAGATE(launch_linear_ex(pooled_q, D, nullptr, nullptr, 0, 0, 0, w->sq_wih, D, w->sq_bih, w->sq_bhh, p.gx, 4 * HS, BS, 4 * HS, D, NIR_ACT_NONE | bnd, nullptr, 0, sq_));
    }
    // ---- encode_clicks (cars.py:262-304)
    if (d_on) {
        NIR_PROPAGATE(launch_linear_ex(pooled_docs, D, nullptr, nullptr, 0, 0, 0, w->click0_w, D, w->click0_b, nullptr, p.epart, NP, R, D, D,
                                       ACT_TANH_ROWDOT16 | ((w->rank_bounded & 4) ? ACT_BOUNDED : 0), w->click3_w, 0, st));
        {
            ProfScope ps("click_pool2_kernel", st);
            const float* lall = labels_all ? labels_all : labels;
            const int rall = labels_all ? (int)rows_all : (int)BS;
            if (N <= 64)
                hipLaunchKernelGGL(click_pool2_kernel, dim3((unsigned)((BS + 3) / 4)), dim3(256), 0, st, pooled_docs, p.epart, NP, w->click3_b, labels,
                                   lall, rall, m_groups, sessions_per_group * S, (int)BS, N, D, clicks);
            else
                hipLaunchKernelGGL(click_pool_big_kernel, dim3((unsigned)((BS + 3) / 4)), dim3(256), (size_t)4 * 2 * N * sizeof(float), st, pooled_docs,
                                   p.epart, NP, w->click3_b, labels, lall, rall, m_groups, sessions_per_group * S, (int)BS, N, D, clicks);
        }
        NIR_CHECK_LAUNCH("click_pool2_kernel");
    }
    // ---- session LSTM chains: state t+1 = LSTM(x_t, state t); the ranking path needs states 1..S-1, the decoder S as well
    const int nsteps = want_states ? S : S - 1;
    if (nch) {
        LstmStepArgs a;
        a.wih[0] = w->sq_wih; a.whh[0] = w->sq_whh; a.bih[0] = w->sq_bih; a.bhh[0] = w->sq_bhh;
        a.wih[1] = w->sd_wih; a.whh[1] = w->sd_whh; a.bih[1] = w->sd_bih; a.bhh[1] = w->sd_bhh;
        a.xid[0] = a.xid[1] = nullptr;
        a.xstride[0] = a.xstride[1] = (int64_t)S * D;
        a.chain0 = q_on ? 0 : 1;
        a.B = B; a.I = D; a.H = HS;
        // The chains' inputs -- the pooled query / the click-pooled documents of step t -- are known before the loop: their gate contribution
        // x W_ih^T + b_ih + b_hh is ONE GEMM per chain over all B*S rows (split-precision matrix-core kernels, M = B*S rows at once) instead of a
        // K = 256 slice of every sequential step on the fp32 MFMA; the steps then walk W_hh (K = HS) only.
        {
            // (inputs are pooled encoder states / softmax-weighted sums of them, inside (-1, 1); bit 3 of rank_bounded: |W_ih| < 2^15 host-checked -> fp16 two-term split)
            // (the query chain's projection was issued with U above, on the side stream)
            if (d_on) NIR_PROPAGATE(launch_linear_ex(clicks, D, nullptr, nullptr, 0, 0, 0, w->sd_wih, D, w->sd_bih, w->sd_bhh, gd, 4 * HS, BS, 4 * HS, D, NIR_ACT_NONE | bnd, nullptr, 0, st));
            a.gx[0] = gq; a.gx[1] = gd;
            a.gxstride = (int64_t)S * 4 * HS;
            a.whh_frag[0] = w->sq_whh_frag; a.whh_frag[1] = w->sd_whh_frag;
        }
        const float* gx0[2] = {a.gx[0], a.gx[1]};
        const int64_t slot = (int64_t)B * HS;
        for (int t = 0; t < nsteps; ++t) {      // slot 0 = the initial zero state: never read (NULL previous state), never written
            a.gx[0] = gx0[0] + (int64_t)t * 4 * HS; a.gx[1] = gx0[1] + (int64_t)t * 4 * HS;
            a.x[0] = pooled_q + (int64_t)t * D; a.x[1] = clicks + (int64_t)t * D;
            a.hprev[0] = t ? p.Qs + t * slot : nullptr; a.cprev[0] = t ? p.Cq + t * slot : nullptr;
            a.hprev[1] = t ? p.Ds + t * slot : nullptr; a.cprev[1] = t ? p.Cd + t * slot : nullptr;
            a.hnext[0] = p.Qs + (t + 1) * slot; a.cnext[0] = p.Cq + (t + 1) * slot;
            a.hnext[1] = p.Ds + (t + 1) * slot; a.cnext[1] = p.Cd + (t + 1) * slot;
            a.h16prev[0] = t ? reinterpret_cast<const _Float16*>(p.Qs16 + t * slot) : nullptr;
            a.h16prev[1] = t ? reinterpret_cast<const _Float16*>(p.Ds16 + t * slot) : nullptr;
            a.h16next[0] = reinterpret_cast<_Float16*>(p.Qs16 + (t + 1) * slot);
            a.h16next[1] = reinterpret_cast<_Float16*>(p.Ds16 + (t + 1) * slot);
            NIR_PROPAGATE(launch_lstm_step(a, nch, st));
        }
    }
    if (rank_on) {
        const int KR = D + nch * HS;
        const float* xrows = pooled_q;
        if (nch) {
            {
                ProfScope ps("session_attend2_kernel", st);
                hipLaunchKernelGGL(session_attend2_kernel, dim3((unsigned)BS), dim3(256), 2 * S * sizeof(float), st, Uq, NU, p.Qs, p.Ds, pooled_q, B, S, D, HS,
                                   (int)q_on, (int)d_on, p.xcat);
            }
            NIR_CHECK_LAUNCH("session_attend2_kernel");
            xrows = p.xcat;
        }
        NIR_PROPAGATE(launch_linear(xrows, KR, nullptr, nullptr, 0, 0, 0, w->wrank, KR, w->qproj_b, nullptr, p.qp, D, BS, D, KR, NIR_ACT_NONE, st));
        // the ranker scores a candidate against the session state only: a candidate-sharded caller hands its own slice of the pooled
        // documents here (rank_docs [B,S,NR,D] -> click_scores [B,S,NR]) while clicks and sessions above saw all N candidates
        const float* rdocs = rank_docs ? rank_docs : pooled_docs;
        const int Nr = rank_docs ? NR : N;
        const int64_t Rr = BS * Nr;
        {
            ProfScope ps("rank_feats_kernel", st);
            hipLaunchKernelGGL(rank_feats_kernel, g1(Rr * D / 4), dim3(256), 0, st, p.qp, rdocs, Nr, D, Rr, p.feats);
        }
        NIR_CHECK_LAUNCH("rank_feats_kernel");
        // maxout 1024 -> 256 -> 128 -> 1 (pool 2): the pairwise max is fused into the GEMM epilogues
        NIR_PROPAGATE(launch_linear_ex(p.feats, 4 * D, nullptr, nullptr, 0, 0, 0, w->mo0_w, 4 * D, w->mo0_b, nullptr, p.y0, 256, Rr, 512, 4 * D,
                                       ACT_MAXOUT2 | ((w->rank_bounded & 1) ? ACT_BOUNDED : 0), nullptr, 0, st));
        NIR_PROPAGATE(launch_linear_ex(p.y0, 256, nullptr, nullptr, 0, 0, 0, w->mo1_w, 256, w->mo1_b, nullptr, p.y1, 128, Rr, 256, 256,
                                       ACT_MAXOUT2 | ((w->rank_bounded & 2) ? ACT_BOUNDED : 0), nullptr, 0, st));
        NIR_PROPAGATE(launch_linear_ex(p.y1, 128, nullptr, nullptr, 0, 0, 0, w->mo2_w, 128, w->mo2_b, nullptr, click_scores, 1, Rr, 2, 128, ACT_MAXOUT2, nullptr, 0, st));
    }
    if (want_states) {
        // ---- suggestion-side outputs (cars.py:382-456): inner attention pools and the decoder initial states
        NIR_REQUIRE(nch > 0, "cars_rank_session: decoder states need at least one session encoder");
        const float* st_a = q_on ? p.Qs : p.Ds;
        const float* st_b = (q_on && d_on) ? p.Ds : nullptr;
        const float* c_a = q_on ? p.Cq : p.Cd;
        const float* c_b = (q_on && d_on) ? p.Cd : nullptr;
        const int HSb = st_b ? HS : 0, W = HS + HSb;
        const int rows = (S - 1) * B;                                // hidden_states[:-1]: steps 0..S-2 -> states 1..S-1
        if (extra->inner_q && q_on) {
            NIR_PROPAGATE(launch_linear_ex(p.Qs + (int64_t)B * HS, HS, nullptr, nullptr, 0, 0, 0, w->sq_inner0_w, HS, w->sq_inner0_b, nullptr, p.lin, HS / 16,
                                           (int64_t)S * B, HS, HS, ACT_TANH_ROWDOT16, w->sq_inner3_w, 0, st));
            hipLaunchKernelGGL(session_inner_pool_kernel, dim3((unsigned)BS), dim3(256), S * sizeof(float), st, p.Qs + (int64_t)B * HS, p.lin, HS / 16, w->sq_inner3_b, B, S, HS, extra->inner_q);
        }
        if (extra->inner_d && d_on) {
            NIR_PROPAGATE(launch_linear_ex(p.Ds + (int64_t)B * HS, HS, nullptr, nullptr, 0, 0, 0, w->sd_inner0_w, HS, w->sd_inner0_b, nullptr, p.lin, HS / 16,
                                           (int64_t)S * B, HS, HS, ACT_TANH_ROWDOT16, w->sd_inner3_w, 0, st));
            hipLaunchKernelGGL(session_inner_pool_kernel, dim3((unsigned)BS), dim3(256), S * sizeof(float), st, p.Ds + (int64_t)B * HS, p.lin, HS / 16, w->sd_inner3_b, B, S, HS, extra->inner_d);
        }
        NIR_CHECK_LAUNCH("session_inner_pool_kernel");
        if (rows > 0 && extra->dec_h && w->th_w) {
            hipLaunchKernelGGL(session_cat_states_kernel, g1((int64_t)rows * W), dim3(256), 0, st, st_a + (int64_t)B * HS,
                               st_b ? st_b + (int64_t)B * HS : nullptr, rows, HS, HSb, p.cat);
            NIR_PROPAGATE(launch_linear(p.cat, W, nullptr, nullptr, 0, 0, 0, w->th_w, W, w->th_b, nullptr, extra->dec_h, w->HDEC, rows, w->HDEC, W, NIR_ACT_NONE, st));
        }
        if (rows > 0 && extra->dec_c && w->tc_w) {
            hipLaunchKernelGGL(session_cat_states_kernel, g1((int64_t)rows * W), dim3(256), 0, st, c_a + (int64_t)B * HS,
                               c_b ? c_b + (int64_t)B * HS : nullptr, rows, HS, HSb, p.cat);
            NIR_PROPAGATE(launch_linear(p.cat, W, nullptr, nullptr, 0, 0, 0, w->tc_w, W, w->tc_b, nullptr, extra->dec_c, w->HDEC, rows, w->HDEC, W, NIR_ACT_NONE, st));
        }
        NIR_CHECK_LAUNCH("session_cat_states_kernel");
    }
    return 0;
}
