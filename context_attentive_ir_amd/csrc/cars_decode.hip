// CARS.decode -- greedy query suggestion (neuroir/multitask/cars.py:706-791; RNNDecoder: decoders/decoder.py:94-168,
// decoders/rnn_decoder.py:19-88; Luong "general" attention: modules/global_attention.py:98-196).
//
// Per decoding step, for all Bd = B*(S-1) rows at once (the reference runs the same ops, one torch call each, plus a host
// round trip per step to map target-vocabulary ids back to source-vocabulary ids through two Python dicts):
//   x = emb(tgt)                      gathered inside the LSTM step kernel (no [Bd,1,E] tensor)
//   (h,c) = LSTM(x, (h,c))            lstm_step_kernel (16x16x4 MFMA, cell fused)
//   q = W_in h                        GEMM
//   a = softmax_j(mask(q . m_j)); ctx = sum_j a_j m_j; cat = [ctx ; h]      dec_attend_kernel (m = dec_attn(encoded queries))
//   o = tanh(W_out cat)               GEMM + tanh epilogue
//   p = W_p1 o + session_rep          GEMM + addend epilogue (session_rep = (W_shared + W_priv2) [inner_q ; inner_d], once)
//   logits = W_p2 p                   GEMM  [Bd, V_tgt]
//   pred = argmax logits; tgt = lut[pred]                                    argmax_map_kernel (softmax is monotone; first
//                                                                           index wins ties like torch.max)
// Everything is enqueued on the caller's stream; no host synchronisation.
#include <algorithm>
#include <mutex>
#include "common.hpp"

namespace nir {

int launch_linear(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                  int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                  int64_t ldc, int64_t M, int N, int K, int act, hipStream_t st);
int launch_linear_ex(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                     int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                     int64_t ldc, int64_t M, int N, int K, int act, const float* add, int64_t ldadd, hipStream_t st);



// one wave per decode row i: memory block = mem[rowmap[i]] ([QL, HD]), valid length = lens[rowmap[i]]
__global__ __launch_bounds__(256) void dec_attend_kernel(const float* __restrict__ qv, const float* __restrict__ h,
                                                         const float* __restrict__ mem, const float* __restrict__ memq,
                                                         const int64_t* __restrict__ rowmap, const int64_t* __restrict__ lens, int Bd, int QL, int HD,
                                                         float* __restrict__ cat) {
    extern __shared__ float pr[];                 // [4][QL]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= Bd) return;
    float* pw = pr + wave * QL;
    const int64_t r = rowmap[i];
    int len = (int)lens[r];
    len = len < 0 ? 0 : (len > QL ? QL : len);
    const float* mb = mem + r * QL * HD;
    const float* mq = memq + r * QL * HD;       // the bank the scores are taken against (== mem, or the bank with linear_in folded in)
    const float* q = qv + (int64_t)i * HD;
    float mx = -INFINITY;
    for (int j = 0; j < len; ++j) {
        float s = 0.f;
        for (int f = 4 * lane; f < HD; f += 256) {
            const float4 a = *reinterpret_cast<const float4*>(mq + (int64_t)j * HD + f), b = *reinterpret_cast<const float4*>(q + f);
            s += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        }
        s = wave_sum(s);
        if (lane == 0) pw[j] = s;
        mx = fmaxf(mx, s);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float den = 0.f;
    for (int j = 0; j < len; ++j) den += expf(pw[j] - mx);
    float* o = cat + (int64_t)i * 2 * HD;
    for (int f = 4 * lane; f < HD; f += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < len; ++j) {
            const float p = expf(pw[j] - mx) / den;
            const float4 v = *reinterpret_cast<const float4*>(mb + (int64_t)j * HD + f);
            acc.x = fmaf(p, v.x, acc.x); acc.y = fmaf(p, v.y, acc.y); acc.z = fmaf(p, v.z, acc.z); acc.w = fmaf(p, v.w, acc.w);
        }
        if (len == 0) acc = make_float4(NAN, NAN, NAN, NAN);      // softmax over an all -inf row
        *reinterpret_cast<float4*>(o + f) = acc;
        *reinterpret_cast<float4*>(o + HD + f) = *reinterpret_cast<const float4*>(h + (int64_t)i * HD + f);
    }
}

// pred[i*pstride] = argmax_v logits[i,v] (first index on ties); tgt[i] = lut ? lut[pred] : pred
__global__ __launch_bounds__(256) void argmax_map_kernel(const float* __restrict__ logits, int64_t V, const int64_t* __restrict__ lut,
                                                         int64_t* __restrict__ pred, int64_t pstride, int64_t* __restrict__ tgt,
                                                         int64_t Vsrc) {
    __shared__ float bv[256];
    __shared__ int64_t bi[256];
    const int i = blockIdx.x;
    const float* row = logits + (int64_t)i * V;
    float best = -INFINITY;
    int64_t idx = V;                                 // V = "none yet"
    for (int64_t v = threadIdx.x; v < V; v += 256) {
        const float x = row[v];
        if (x > best || (x == best && v < idx)) { best = x; idx = v; }
    }
    bv[threadIdx.x] = best;
    bi[threadIdx.x] = idx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float ob = bv[threadIdx.x + s];
            const int64_t oi = bi[threadIdx.x + s];
            if (ob > bv[threadIdx.x] || (ob == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = ob; bi[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int64_t w = bi[0] < V ? bi[0] : 0;
        pred[(int64_t)i * pstride] = w;
        const int64_t nxt = lut ? lut[w] : w;
        tgt[i] = (nxt >= 0 && nxt < Vsrc) ? nxt : 1;     // a token without a source-vocabulary row is fed back as <unk> (id 1), like src_dict[word]
    }
}

// ---- token_prob_predictor2 fused with the arg-max (cars.py:779-787): logits[b, v] = p1[b, :] . W2[v, :] never leave the chip. ----------
// W2 [VT, 256] arrives as pre-split fp16 two-term planes in MFMA A-fragment order (built once per weight version by the host):
//   frag[vt][ks][term][lane][8],  element (lane, j) = W2[16 vt + (lane & 15)][32 ks + 8 (lane >> 4) + j]      (vt = 16-row vocabulary tile)
// p1 (<= 96 decode rows per pass) is split into two fp16 planes in LDS and read as the B operand; a wave owns PA_TW vocabulary tiles at a
// time, three v_mfma_f32_16x16x32_f16 per product block (x = x1 + 2^-11 x2': fp32-class error, as everywhere else).  Each lane keeps the
// running (max, first index) of the 4 x PA_TW vocabulary rows it sees per decode row; lanes that share a decode row combine through two
// ds_bpermute hops, the four waves write one partial per decode row, argmax_finish_kernel reduces the partials of all workgroups and maps
// the winner to the next input token.  Replaces a 96 x 30 000 x 256 fp32 GEMM that wrote 11.5 MB of logits per step plus an arg-max kernel
// that read them back.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PA_K = 256, PA_KS = PA_K / 32, PA_NBT = 6, PA_ROWS = 16 * PA_NBT, PA_LD = PA_K + 8;     // 96 decode rows per pass
constexpr int PA_TW = 2;                                                                              // vocabulary tiles per wave and pass
constexpr size_t PA_LDS = (size_t)2 * PA_ROWS * PA_LD * 2;

// Grid (round 5) = row blocks of 96 decode rows x `nvr` vocabulary ranges: a workgroup stages ITS row block once and walks its vocabulary range
// (at 768 rows: 8 x 32 workgroups of ~15 tiles per wave).  Before, every workgroup took all row blocks in turn over 1/256 of the vocabulary:
// eight re-stagings of 96 rows and ~2 tiles per wave and pass -- 16 us of MFMA work in a 124 us launch.
__global__ __launch_bounds__(256, 1) void pred_argmax_kernel(const float* __restrict__ p1, const _Float16* __restrict__ wfrag, int64_t VT,
                                                             int64_t ntiles, int64_t Bd, float* __restrict__ pval, int* __restrict__ pidx, int nvr,
                                                             unsigned long long* __restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) _Float16 pa_sm[];          // [2 terms][96 rows][PA_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    const int vr = (int)(blockIdx.x % nvr);
    const int64_t per_wg = (ntiles + nvr - 1) / nvr;
    const int64_t t_lo = (int64_t)vr * per_wg, t_hi = min(ntiles, t_lo + per_wg);
    {
        const int64_t b0 = (int64_t)(blockIdx.x / nvr) * PA_ROWS;
        // stage + split this workgroup's decode rows (zero rows past Bd).  ALL 24 float4 loads of a thread are issued before the first is
        // converted (unconditional, from a clamped row: a branch around a load puts an s_waitcnt vmcnt(0) at its join): written as one
        // load -> convert -> ds_write loop the iterations stayed in order, 24 dependent L2 round trips per workgroup -- with a third of the
        // MFMAs, no epilogue and no W stream the launch still took 31 of its 40 us.  (The requests of the first W fragment sets in front of
        // the staging loads: no change.)
        constexpr int PA_SIT = PA_ROWS * (PA_K / 4) / 256;
        static_assert(PA_ROWS * (PA_K / 4) % 256 == 0, "staging: whole trips");
        float4 sv[PA_SIT];
#pragma unroll
        for (int q = 0; q < PA_SIT; ++q) {
            const int e = tid + 256 * q;
            const int r = e / (PA_K / 4), k4 = (e - r * (PA_K / 4)) * 4;
            const int64_t b = b0 + r;
            sv[q] = *reinterpret_cast<const float4*>(p1 + (b < Bd ? b : Bd - 1) * PA_K + k4);
        }
        __builtin_amdgcn_sched_barrier(0);         // (without it the scheduler pairs the loads with their conversions again, six in flight)
#pragma unroll
        for (int q = 0; q < PA_SIT; ++q) {
            const int e = tid + 256 * q;
            const int r = e / (PA_K / 4), k4 = (e - r * (PA_K / 4)) * 4;
            float4 v = sv[q];
            if (b0 + r >= Bd) v = make_float4(0.f, 0.f, 0.f, 0.f);
            const fp16x2_t a01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), a23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
            const fp16x2_t b01 = __builtin_amdgcn_cvt_pkrtz((v.x - (float)a01[0]) * 2048.0f, (v.y - (float)a01[1]) * 2048.0f);
            const fp16x2_t b23 = __builtin_amdgcn_cvt_pkrtz((v.z - (float)a23[0]) * 2048.0f, (v.w - (float)a23[1]) * 2048.0f);
            _Float16* d = pa_sm + r * PA_LD + k4;
            *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, a23));
            *reinterpret_cast<uint2*>(d + PA_ROWS * PA_LD) = make_uint2(__builtin_bit_cast(unsigned, b01), __builtin_bit_cast(unsigned, b23));
        }
        __syncthreads();
        float best[PA_NBT];
        int bidx[PA_NBT];
#pragma unroll
        for (int bt = 0; bt < PA_NBT; ++bt) { best[bt] = -INFINITY; bidx[bt] = 0x7FFFFFFF; }
        // all of a tile pair's W fragments are requested at once (2 tiles x 8 k-steps x 2 terms x 16 B per lane = 128 VGPRs; one workgroup per
        // CU has them to spare): ONE memory round trip per tile pair instead of one per k-step (first version: 32 us per launch = 0.96 TB/s
        // on a 30.7 MB weight stream, latency-bound on 8 dependent round trips).  Round 5: THREE fragment sets of one tile each -- a round trip
        // (~4 000 cycles) runs under the 2 x 144 MFMAs (~2 300 each) of the two tiles in front of it instead of in front of its own tile (one wave
        // per SIMD: nothing else hides it; two sets of two tiles: 512 registers, 40 spilled, 160 us).
        constexpr int TW = 1;                                          // tiles per fragment set (three sets in flight)
        auto load_w = [&](int64_t t0, f16x8 (&wf)[PA_KS][TW][2]) {
#pragma unroll
            for (int ks = 0; ks < PA_KS; ++ks)
#pragma unroll
                for (int u = 0; u < TW; ++u) {
                    int64_t t = t0 + u < t_hi ? t0 + u : t_hi - 1;            // clamped: a duplicate tile's results are discarded below
                    t = t < 0 ? 0 : t;
                    const _Float16* wp = wfrag + ((t * PA_KS + ks) * 2 * 64 + lane) * 8;
                    wf[ks][u][0] = *reinterpret_cast<const f16x8*>(wp);
                    wf[ks][u][1] = *reinterpret_cast<const f16x8*>(wp + 512);
                }
        };
        auto compute = [&](int64_t t0, const f16x8 (&wf)[PA_KS][TW][2]) {
            f32x4 acc[TW][PA_NBT], acx[TW][PA_NBT];
#pragma unroll
            for (int u = 0; u < TW; ++u)
#pragma unroll
                for (int bt = 0; bt < PA_NBT; ++bt) { acc[u][bt] = (f32x4){0.f, 0.f, 0.f, 0.f}; acx[u][bt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            // The decode rows' fragments of k-step ks + 1 are read from LDS while the MFMAs of k-step ks run (two register sets), and a k-step's
            // MFMAs go in three passes over the six row tiles so that no MFMA waits for the one issued just before it.  (Before: two
            // ds_read_b128 -> wait -> three MFMAs per row tile, the middle one dependent on the first: with ONE wave per SIMD every LDS round
            // trip was exposed -- 48 of them per vocabulary tile, ~45 of the launch's 57 us; removing the W stream entirely changed nothing.)
            const _Float16* bp0 = pa_sm + c16 * PA_LD + 8 * g4;
            f16x8 bq[2][PA_NBT][2];
            auto ldb = [&](int ks, f16x8 (&b)[PA_NBT][2]) {
#pragma unroll
                for (int bt = 0; bt < PA_NBT; ++bt) {
                    b[bt][0] = *reinterpret_cast<const f16x8*>(bp0 + 32 * ks + bt * 16 * PA_LD);
                    b[bt][1] = *reinterpret_cast<const f16x8*>(bp0 + 32 * ks + bt * 16 * PA_LD + PA_ROWS * PA_LD);
                }
            };
            ldb(0, bq[0]);
#pragma unroll
            for (int ks = 0; ks < PA_KS; ++ks) {
                if (ks + 1 < PA_KS) ldb(ks + 1, bq[(ks + 1) & 1]);
                const f16x8 (&b)[PA_NBT][2] = bq[ks & 1];
#pragma unroll
                for (int u = 0; u < TW; ++u) {
#pragma unroll
                    for (int bt = 0; bt < PA_NBT; ++bt) acx[u][bt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][u][1], b[bt][0], acx[u][bt], 0, 0, 0);
#pragma unroll
                    for (int bt = 0; bt < PA_NBT; ++bt) acc[u][bt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][u][0], b[bt][0], acc[u][bt], 0, 0, 0);
#pragma unroll
                    for (int bt = 0; bt < PA_NBT; ++bt) acx[u][bt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][u][0], b[bt][1], acx[u][bt], 0, 0, 0);
                }
                // one LDS read between MFMAs: the 12 reads of the next k-step spread under this k-step's 18 MFMAs
#pragma unroll
                for (int i = 0; i < 2 * PA_NBT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * PA_NBT * TW - 2 * PA_NBT, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                if (t0 + u >= t_hi) continue;                              // wave-uniform
#pragma unroll
                for (int bt = 0; bt < PA_NBT; ++bt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t v = (t0 + u) * 16 + 4 * g4 + r;         // ascending in (u, r): '>' keeps the first index on ties
                        const float x = fmaf(acx[u][bt][r], 1.0f / 2048.0f, acc[u][bt][r]);
                        if (v < VT && x > best[bt]) { best[bt] = x; bidx[bt] = (int)v; }
                    }
            }
        };
        {
            constexpr int64_t ST = 4 * TW;
            f16x8 wfA[PA_KS][TW][2], wfB[PA_KS][TW][2], wfC[PA_KS][TW][2];
            int64_t t0 = t_lo + wave * TW;
            if (t0 < t_hi) load_w(t0, wfA);
            if (t0 + ST < t_hi) load_w(t0 + ST, wfB);
            for (; t0 < t_hi; t0 += 3 * ST) {
                if (t0 + 2 * ST < t_hi) load_w(t0 + 2 * ST, wfC);
                __builtin_amdgcn_sched_barrier(0);
                compute(t0, wfA);
                if (t0 + ST >= t_hi) break;
                if (t0 + 3 * ST < t_hi) load_w(t0 + 3 * ST, wfA);
                __builtin_amdgcn_sched_barrier(0);
                compute(t0 + ST, wfB);
                if (t0 + 2 * ST >= t_hi) break;
                if (t0 + 4 * ST < t_hi) load_w(t0 + 4 * ST, wfB);
                __builtin_amdgcn_sched_barrier(0);
                compute(t0 + 2 * ST, wfC);
            }
        }
        // lanes l, l+16, l+32, l+48 hold the same decode column: combine (first index wins ties), then one partial per wave and column
#pragma unroll
        for (int bt = 0; bt < PA_NBT; ++bt) {
#pragma unroll
            for (int sh = 16; sh <= 32; sh <<= 1) {
                const float ov = __shfl_xor(best[bt], sh);
                const int oi = __shfl_xor(bidx[bt], sh);
                if (ov > best[bt] || (ov == best[bt] && oi < bidx[bt])) { best[bt] = ov; bidx[bt] = oi; }
            }
            const int64_t b = b0 + bt * 16 + c16;
            if (g4 == 0 && b < Bd && !keys) {
                const int64_t slot = ((int64_t)vr * 4 + wave) * Bd + b;
                pval[slot] = best[bt];
                pidx[slot] = bidx[bt];
            }
        }
        if (keys) {
            // (round 6) keyed form: the four waves' winners meet in LDS (the staged rows are dead by now), ONE 64-bit atomic max per workgroup and decode
            // row, spread over ARGMAX_KEY_BUCKETS words per row (all 235 workgroups finish together: a single word per row serialised ~900 atomics)
            unsigned long long* kl = reinterpret_cast<unsigned long long*>(pa_sm);      // [4 waves][96]
            __syncthreads();
            if (g4 == 0) {
#pragma unroll
                for (int bt = 0; bt < PA_NBT; ++bt) kl[wave * PA_ROWS + bt * 16 + c16] = argmax_key(best[bt], bidx[bt]);
            }
            __syncthreads();
            if (tid < PA_ROWS) {
                unsigned long long k = kl[tid];
#pragma unroll
                for (int w_ = 1; w_ < 4; ++w_) { const unsigned long long o = kl[w_ * PA_ROWS + tid]; k = o > k ? o : k; }
                const int64_t b = b0 + tid;
                if (b < Bd) atomicMax(keys + b * ARGMAX_KEY_BUCKETS + (vr % ARGMAX_KEY_BUCKETS), k);
            }
        }
    }
}

// one wave per decode row: reduce the per-(workgroup, wave) partials, first index on ties; pred[i*pstride] = winner, tgt[i] = lut[winner]
__global__ __launch_bounds__(64) void argmax_finish_kernel(const float* __restrict__ pval, const int* __restrict__ pidx, int nparts, int64_t Bd,
                                                           const int64_t* __restrict__ lut, int64_t* __restrict__ pred, int64_t pstride,
                                                           int64_t* __restrict__ tgt, int64_t Vsrc) {
    const int64_t i = blockIdx.x;
    const int lane = threadIdx.x;
    float best = -INFINITY;
    int idx = 0x7FFFFFFF;
    for (int p = lane; p < nparts; p += 64) {
        const float v = pval[(int64_t)p * Bd + i];
        const int vi = pidx[(int64_t)p * Bd + i];
        if (v > best || (v == best && vi < idx)) { best = v; idx = vi; }
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        const float ov = __shfl_xor(best, sh);
        const int oi = __shfl_xor(idx, sh);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (lane == 0) {
        const int64_t w = idx != 0x7FFFFFFF ? idx : 0;
        pred[i * pstride] = w;
        const int64_t nxt = lut ? lut[w] : w;
        tgt[i] = (nxt >= 0 && nxt < Vsrc) ? nxt : 1;
    }
}

// keys [max_len][Bd] of a whole greedy decode -> pred [Bd][max_len] (target-vocabulary ids): ONE launch behind the last step
__global__ void keys_to_pred_kernel(const unsigned long long* __restrict__ keys, int64_t Bd, int max_len, int64_t* __restrict__ pred) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Bd * max_len) return;
    const int64_t step = e / Bd, row = e - step * Bd;
    unsigned long long k = 0ull;
#pragma unroll
    for (int q = 0; q < ARGMAX_KEY_BUCKETS; ++q) { const unsigned long long o = keys[e * ARGMAX_KEY_BUCKETS + q]; k = o > k ? o : k; }
    pred[row * max_len + step] = argmax_key_index(k);
}

__global__ void fill_i64_kernel(int64_t* p, int64_t v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// wsum = w1 + w2  (shared_session_projector + private_session_projector2, packed once)
__global__ void add_weights_kernel(const float* a, const float* b, float* o, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}

// h [n = rows * H] fp32 -> the fp16 term pairs of lstm_step16_kernel's B operand: [row][H/8][2 terms][8]  (H % 8 == 0)
__global__ void h16_pack_kernel(const float* __restrict__ h, int64_t n, _Float16* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float v = h[e];
    const _Float16 a = (_Float16)v;
    _Float16* d = out + (e >> 3) * 16 + (e & 7);
    d[0] = a;
    d[8] = (_Float16)((v - (float)a) * 2048.0f);
}

struct DecPlan {
    float *mem, *memq, *sess, *h[2], *c[2], *h16[2], *qv, *cat, *ah, *p1, *logits, *pval;
    int* pidx;
    int64_t* tgt;
    size_t bytes;
};
constexpr int PA_MAX_WGS = 256;
static int cu_count() {
    static int n = 0;
    static std::once_flag once;
    std::call_once(once, [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    });
    return n;
}
static DecPlan dec_plan(void* ws, size_t cap, int64_t rows_src, int64_t Bd, int QL, int HD, int P, int64_t VT, bool fused_argmax, bool foldq) {
    Workspace a(ws, cap);
    DecPlan p;
    p.mem = a.take<float>((size_t)rows_src * QL * HD);
    p.memq = a.take<float>(foldq ? (size_t)rows_src * QL * HD : 0);
    p.sess = a.take<float>((size_t)Bd * P);
    for (int k = 0; k < 2; ++k) { p.h[k] = a.take<float>((size_t)Bd * HD); p.c[k] = a.take<float>((size_t)Bd * HD); }
    for (int k = 0; k < 2; ++k) p.h16[k] = a.take<float>((size_t)Bd * HD);      // the state as fp16 term pairs [Bd][HD/8][2][8] (fp16-term decoder step)
    p.qv = a.take<float>((size_t)Bd * HD);
    p.cat = a.take<float>((size_t)Bd * 2 * HD);
    p.ah = a.take<float>((size_t)Bd * HD);
    p.p1 = a.take<float>((size_t)Bd * P);
    p.logits = a.take<float>(fused_argmax ? 0 : (size_t)Bd * VT);
    p.pval = a.take<float>(fused_argmax ? (size_t)PA_MAX_WGS * 4 * Bd : 0);
    p.pidx = a.take<int>(fused_argmax ? (size_t)PA_MAX_WGS * 4 * Bd : 0);
    p.tgt = a.take<int64_t>((size_t)Bd);
    p.bytes = align_up(a.off, 256);
    return p;
}

}  // namespace nir

extern "C" int nir_add_f32(const float* a, const float* b, float* out, int64_t n, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(a && b && out && n >= 0, "add_f32: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(add_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    NIR_CHECK_LAUNCH("add_weights_kernel");
    return 0;
}

extern "C" size_t nir_cars_decode_workspace_bytes(int64_t rows_src, int64_t Bd, int QL, const nir_cars_decoder_weights* w) {
    if (!w || rows_src < 0 || Bd < 0 || QL <= 0) return 0;
    return nir::dec_plan(nullptr, 0, rows_src, Bd, QL, w->HD, w->P, w->VT, w->pred2_frag != nullptr && w->P == nir::PA_K, w->attn_q_w != nullptr).bytes;
}

extern "C" int nir_cars_decode_greedy(const float* dec_h, const float* dec_c, const float* encoded_source, const int64_t* source_len,
                                      int64_t rows_src, int QL, const int64_t* rowmap, int64_t Bd, const float* session_cat,
                                      const float* table, int64_t V, int E, const int64_t* tgt2src, int64_t bos, int max_len,
                                      const nir_cars_decoder_weights* w, void* workspace, size_t workspace_bytes,
                                      int64_t* predictions, nir_stream_t stream) {
    using namespace nir;
    hipStream_t st = (hipStream_t)stream;
    NIR_REQUIRE(dec_h && dec_c && encoded_source && source_len && rowmap && table && w && predictions, "cars_decode: null pointer");
    NIR_REQUIRE(rows_src > 0 && Bd >= 0 && QL > 0 && max_len > 0 && V > 0 && E > 0, "cars_decode: bad dims");
    NIR_REQUIRE(w->HD % 4 == 0 && E % 4 == 0 && w->DQ % 4 == 0 && w->P % 4 == 0, "cars_decode: dims must be multiples of 4");
    NIR_REQUIRE(w->KS == 0 || (session_cat && w->sess_w), "cars_decode: session representation / packed projector missing");
    NIR_REQUIRE(bos >= 0 && bos < V, "cars_decode: BOS id outside the vocabulary");
    if (Bd == 0) return 0;
    const int HD = w->HD, P = w->P;
    const bool fused_argmax = w->pred2_frag != nullptr && P == PA_K && w->VT < 0x7FFFFFF0LL;
    const bool foldq = w->attn_q_w != nullptr;
    DecPlan p = dec_plan(workspace, workspace_bytes, rows_src, Bd, QL, HD, P, w->VT, fused_argmax, foldq);
    const int64_t ntiles = (w->VT + 15) / 16;
    // vocabulary ranges per row block: enough workgroups for the chip, at least ~4 tiles per wave
    const int64_t pa_rb = (Bd + PA_ROWS - 1) / PA_ROWS;
    const int pa_wgs = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(PA_MAX_WGS, (cu_count() + pa_rb - 1) / pa_rb), (ntiles + 4 * PA_TW - 1) / (4 * PA_TW)));
    if (fused_argmax && PA_LDS > 64 * 1024) {
        static std::once_flag once;
        std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)pred_argmax_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PA_LDS); });
    }
    if (!workspace || p.bytes > workspace_bytes) {
        set_error("cars_decode: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
        return NIR_ERR_WORKSPACE;
    }
    // memory bank = dec_attn(encoded queries)  (cars.py:757-767), all (session, query) rows; rowmap picks [:, :-1]
    NIR_PROPAGATE(launch_linear(encoded_source, w->DQ, nullptr, nullptr, 0, 0, 0, w->dec_attn_w, w->DQ, nullptr, nullptr, p.mem, HD,
                                rows_src * QL, HD, w->DQ, NIR_ACT_NONE, st));
    // attn.linear_in folded into a second bank: score_j = (W_in h) . mem_j = h . (W_in^T mem_j) -- memq = encoded (W_in^T W_dec_attn)^T, one GEMM per
    // decode instead of one [Bd, HD] x [HD, HD] GEMM per step (global_attention.py:139-150 'general')
    if (foldq)
        NIR_PROPAGATE(launch_linear(encoded_source, w->DQ, nullptr, nullptr, 0, 0, 0, w->attn_q_w, w->DQ, nullptr, nullptr, p.memq, HD,
                                    rows_src * QL, HD, w->DQ, NIR_ACT_NONE, st));
    // session_rep = (shared + private2)(cat_session_rep[:, :-1])  (cars.py:775-778): rows gathered through rowmap
    if (w->KS > 0)
        NIR_PROPAGATE(launch_linear(nullptr, 0, rowmap, session_cat, w->KS, 1, 1, w->sess_w, w->KS, nullptr, nullptr, p.sess, P, Bd, P, w->KS,
                                    NIR_ACT_NONE, st));
    hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)((Bd + 255) / 256)), dim3(256), 0, st, p.tgt, bos, Bd);
    NIR_CHECK_LAUNCH("fill_i64_kernel");
    LstmStepArgs a;
    a.x[0] = table; a.xid[0] = p.tgt; a.xstride[0] = E;
    a.wih[0] = w->rnn_wih; a.whh[0] = w->rnn_whh; a.bih[0] = w->rnn_bih; a.bhh[0] = w->rnn_bhh;
    a.x[1] = nullptr; a.xid[1] = nullptr; a.xstride[1] = 0; a.wih[1] = a.whh[1] = a.bih[1] = a.bhh[1] = nullptr;
    a.hprev[1] = a.cprev[1] = nullptr; a.hnext[1] = a.cnext[1] = nullptr;
    a.chain0 = 0; a.B = (int)Bd; a.I = E; a.H = HD;
    // fp16-term decoder step: emb(token) W_ih^T + b_ih + b_hh is a per-token row of the folded gate table (gathered by the previous step's ids),
    // the recurrent product runs on pre-split W_hh fragments and the state travels as term pairs next to its fp32 copy
    const bool step16 = w->rnn_gate_fold && w->rnn_whh_frag && HD % 32 == 0 && !tun(g_tun.exact_f32);
    if (step16) {
        a.gx[0] = w->rnn_gate_fold; a.gxid[0] = p.tgt; a.gxstride = (int64_t)4 * HD; a.gx_unit_major = 1;
        a.whh_frag[0] = w->rnn_whh_frag;
        const int64_t n = Bd * HD;
        hipLaunchKernelGGL(h16_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dec_h, n, reinterpret_cast<_Float16*>(p.h16[1]));
        NIR_CHECK_LAUNCH("h16_pack_kernel");
    }
    // Round 6: with the fused projection + arg-max AND the folded step, the winner of step s travels as a 64-bit arg-max KEY (atomic max per wave and
    // row, argmax_key) that the NEXT step's kernel decodes itself (token -> source id -> gate row): no argmax_finish launch between the steps
    // (8 us of 50 per step at 96 decode rows), one keys_to_pred launch behind the loop.  The keys live in the partial-value scratch.
    const bool keyed = fused_argmax && step16 && max_len * ARGMAX_KEY_BUCKETS <= 512;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(p.pval);
    if (keyed) NIR_PROPAGATE((int)hipMemsetAsync(keys, 0, (size_t)max_len * Bd * ARGMAX_KEY_BUCKETS * sizeof(unsigned long long), st));
    const float* hp = dec_h;
    const float* cp = dec_c;
    for (int step = 0; step < max_len; ++step) {
        float* hn = p.h[step & 1];
        float* cn = p.c[step & 1];
        a.hprev[0] = hp; a.cprev[0] = cp; a.hnext[0] = hn; a.cnext[0] = cn;
        if (step16) {
            a.h16prev[0] = reinterpret_cast<const _Float16*>(p.h16[(step + 1) & 1]);
            a.h16next[0] = reinterpret_cast<_Float16*>(p.h16[step & 1]);
        }
        if (keyed && step > 0) {
            a.gxid[0] = nullptr; a.gxkey = keys + (size_t)(step - 1) * Bd * ARGMAX_KEY_BUCKETS; a.gxmap = tgt2src; a.gxV = V;
        }
        NIR_PROPAGATE(launch_lstm_step(a, 1, st));
        if (!foldq)
            NIR_PROPAGATE(launch_linear(hn, HD, nullptr, nullptr, 0, 0, 0, w->attn_in_w, HD, nullptr, nullptr, p.qv, HD, Bd, HD, HD, NIR_ACT_NONE, st));
        {
            ProfScope ps("dec_attend_kernel", st);
            hipLaunchKernelGGL(dec_attend_kernel, dim3((unsigned)((Bd + 3) / 4)), dim3(256), (size_t)4 * QL * 4, st, foldq ? hn : p.qv, hn, p.mem, foldq ? p.memq : p.mem, rowmap,
                               source_len, (int)Bd, QL, HD, p.cat);
        }
        NIR_CHECK_LAUNCH("dec_attend_kernel");
        NIR_PROPAGATE(launch_linear(p.cat, 2 * HD, nullptr, nullptr, 0, 0, 0, w->attn_out_w, 2 * HD, nullptr, nullptr, p.ah, HD, Bd, HD, 2 * HD, NIR_ACT_TANH, st));
        NIR_PROPAGATE(launch_linear_ex(p.ah, HD, nullptr, nullptr, 0, 0, 0, w->pred1_w, HD, nullptr, nullptr, p.p1, P, Bd, P, HD, NIR_ACT_NONE,
                                       w->KS > 0 ? p.sess : nullptr, P, st));
        if (fused_argmax) {
            {
                ProfScope ps(prof_shape_name("pred_argmax_kernel", (long long)Bd, (long long)w->VT, P), st);
                hipLaunchKernelGGL(pred_argmax_kernel, dim3((unsigned)(pa_wgs * pa_rb)), dim3(256), PA_LDS, st, p.p1, (const _Float16*)w->pred2_frag, w->VT, ntiles, Bd,
                                   p.pval, p.pidx, pa_wgs, keyed ? keys + (size_t)step * Bd * ARGMAX_KEY_BUCKETS : nullptr);
            }
            NIR_CHECK_LAUNCH("pred_argmax_kernel");
            if (!keyed) {
                hipLaunchKernelGGL(argmax_finish_kernel, dim3((unsigned)Bd), dim3(64), 0, st, p.pval, p.pidx, pa_wgs * 4, Bd, tgt2src, predictions + step,
                                   (int64_t)max_len, p.tgt, V);
                NIR_CHECK_LAUNCH("argmax_finish_kernel");
            }
        } else {
            NIR_PROPAGATE(launch_linear(p.p1, P, nullptr, nullptr, 0, 0, 0, w->pred2_w, P, nullptr, nullptr, p.logits, w->VT, Bd, (int)w->VT, P, NIR_ACT_NONE, st));
            {
                ProfScope ps("argmax_map_kernel", st);
                hipLaunchKernelGGL(argmax_map_kernel, dim3((unsigned)Bd), dim3(256), 0, st, p.logits, w->VT, tgt2src, predictions + step, (int64_t)max_len, p.tgt, V);
            }
            NIR_CHECK_LAUNCH("argmax_map_kernel");
        }
        hp = hn;
        cp = cn;
    }
    if (keyed) {
        const int64_t n = Bd * max_len;
        hipLaunchKernelGGL(keys_to_pred_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keys, Bd, max_len, predictions);
        NIR_CHECK_LAUNCH("keys_to_pred_kernel");
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Greedy decoding WITHOUT attention: the suggestion side of M_MATCH_TENSOR and MNSRF (multitask/mmtensor.py:281-325,
// mnsrf.py:251-296: Decoder(attn_type='none') + generator).  Per step: embedding gather + LSTM cell in one launch (lstm_step_kernel),
// generator GEMM [Bd, V_tgt], arg-max + target -> source id map (argmax_map_kernel).  No host synchronisation.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" size_t nir_decode_greedy_plain_workspace_bytes(int64_t Bd, int H, int64_t VT) {
    if (Bd <= 0 || H <= 0 || VT <= 0) return 0;
    return ((size_t)6 * Bd * H + (size_t)Bd * VT) * sizeof(float) + (size_t)Bd * sizeof(int64_t) + 2048;
}

extern "C" int nir_decode_greedy_plain(const float* dec_h, const float* dec_c, int64_t Bd, int H, const float* table, int64_t V, int E,
                                       const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* gen_w,
                                       const float* gen_b, int64_t VT, const int64_t* tgt2src, int64_t bos, int max_len, void* workspace,
                                       size_t workspace_bytes, int64_t* predictions, nir_stream_t stream) {
    return nir_decode_greedy_plain_folded(dec_h, dec_c, Bd, H, table, V, E, w_ih, w_hh, b_ih, b_hh, gen_w, gen_b, VT, tgt2src, bos, max_len, nullptr,
                                          nullptr, workspace, workspace_bytes, predictions, stream);
}

extern "C" int nir_decode_greedy_plain_folded(const float* dec_h, const float* dec_c, int64_t Bd, int H, const float* table, int64_t V, int E,
                                              const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* gen_w,
                                              const float* gen_b, int64_t VT, const int64_t* tgt2src, int64_t bos, int max_len,
                                              const float* gate_fold, const void* whh_frag, void* workspace, size_t workspace_bytes,
                                              int64_t* predictions, nir_stream_t stream) {
    using namespace nir;
    hipStream_t st = (hipStream_t)stream;
    NIR_REQUIRE(dec_h && dec_c && table && w_ih && w_hh && b_ih && b_hh && gen_w && predictions && workspace, "decode_plain: null pointer");
    NIR_REQUIRE(Bd >= 0 && H > 0 && E > 0 && V > 0 && VT > 0 && max_len >= 0, "decode_plain: bad dims");
    NIR_REQUIRE(E % 4 == 0 && H % 4 == 0, "decode_plain: emsize and hidden size must be multiples of 4");
    NIR_REQUIRE(bos >= 0 && bos < V, "decode_plain: BOS id outside the vocabulary");
    NIR_REQUIRE(workspace_bytes >= nir_decode_greedy_plain_workspace_bytes(Bd, H, VT), "decode_plain: workspace too small");
    if (Bd == 0 || max_len == 0) return 0;
    Workspace a(workspace, workspace_bytes);
    float* hb[2] = {a.take<float>((size_t)Bd * H), a.take<float>((size_t)Bd * H)};
    float* cb[2] = {a.take<float>((size_t)Bd * H), a.take<float>((size_t)Bd * H)};
    float* h16b[2] = {a.take<float>((size_t)Bd * H), a.take<float>((size_t)Bd * H)};       // the state as fp16 term pairs (folded step)
    float* logits = a.take<float>((size_t)Bd * VT);
    int64_t* tgt = a.take<int64_t>((size_t)Bd);
    hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)((Bd + 255) / 256)), dim3(256), 0, st, tgt, bos, Bd);
    NIR_CHECK_LAUNCH("fill_i64_kernel");
    // the decoder's input is the previous token's embedding alone: with a folded gate table + W_hh fragments (both or neither) the step gathers
    // its gate rows by token id and runs the recurrent product as fp16 term pairs, like nir_cars_decode_greedy
    const bool step16 = gate_fold && whh_frag && H % 32 == 0 && !tun(g_tun.exact_f32);
    LstmStepArgs s;
    s.x[0] = table; s.xid[0] = tgt; s.xstride[0] = E;
    s.wih[0] = w_ih; s.whh[0] = w_hh; s.bih[0] = b_ih; s.bhh[0] = b_hh;
    s.x[1] = nullptr; s.xid[1] = nullptr; s.xstride[1] = 0; s.wih[1] = s.whh[1] = s.bih[1] = s.bhh[1] = nullptr;
    s.hprev[1] = s.cprev[1] = nullptr; s.hnext[1] = s.cnext[1] = nullptr;
    s.chain0 = 0; s.B = (int)Bd; s.I = E; s.H = H;
    if (step16) {
        s.gx[0] = gate_fold; s.gxid[0] = tgt; s.gxstride = (int64_t)4 * H; s.gx_unit_major = 1;
        s.whh_frag[0] = whh_frag;
        const int64_t n = Bd * H;
        hipLaunchKernelGGL(h16_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dec_h, n, reinterpret_cast<_Float16*>(h16b[1]));
        NIR_CHECK_LAUNCH("h16_pack_kernel");
    }
    const float* hp = dec_h;
    const float* cp = dec_c;
    for (int step = 0; step < max_len; ++step) {
        s.hprev[0] = hp; s.cprev[0] = cp; s.hnext[0] = hb[step & 1]; s.cnext[0] = cb[step & 1];
        if (step16) {
            s.h16prev[0] = reinterpret_cast<const _Float16*>(h16b[(step + 1) & 1]);
            s.h16next[0] = reinterpret_cast<_Float16*>(h16b[step & 1]);
        }
        NIR_PROPAGATE(launch_lstm_step(s, 1, st));
        NIR_PROPAGATE(launch_linear(hb[step & 1], H, nullptr, nullptr, 0, 0, 0, gen_w, H, gen_b, nullptr, logits, VT, Bd, (int)VT, H, NIR_ACT_NONE, st));
        {
            ProfScope ps("argmax_map_kernel", st);
            hipLaunchKernelGGL(argmax_map_kernel, dim3((unsigned)Bd), dim3(256), 0, st, logits, VT, tgt2src, predictions + step, (int64_t)max_len, tgt, V);
        }
        NIR_CHECK_LAUNCH("argmax_map_kernel");
        hp = hb[step & 1];
        cp = cb[step & 1];
    }
    return 0;
}
