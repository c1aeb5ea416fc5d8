// Folded-embedding BiLSTM (inference): RNNEncoder over an embedding lookup (neuroir/encoders/rnn_encoder.py:62-141 fed by
// neuroir/modules/embeddings.py:243-252; CARS: multitask/cars.py:193-260) with the input projection folded into the table.
//
//   gates_x[m,t,:] = emb(ids[m,t]) W_ih^T + b_ih + b_hh  ==  (table W_ih^T + b)[ids[m,t], :]
//
// In eval mode (dropout = identity) the left side only depends on the token id, so the product is computed ONCE per
// vocabulary row when the weights are packed (nir_lstm_fold_table: V x 8H x E MACs, the cost of about one C3 batch) and
// the per-batch gate GEMM (59 % of the CARS FLOPs) disappears: the recurrence gathers its gate pre-activations straight
// from the folded table by token id.  288 GB of HBM make the V x 8H table (410 MB at V = 100 000, H = 128) a non-issue.
// The [M*T, 8H] gate tensor (293 MB written + re-read per C3 batch) is never materialised.
//
// Folded-table layout: pt[v][dir][unit][gate]  (gate order i,f,g,o interleaved innermost), fp32 or bf16: the recurrence
// lane that owns (unit, sequence) reads its four gate pre-activations with ONE 16-byte (fp32) / 8-byte (bf16) load, and a
// wave covers 4*NT consecutive units = 64*NT contiguous bytes of each of its 16 rows.
//
// Recurrence kernels (16 sequences per workgroup, one direction, 16 waves; same MFMA mapping as lstm_mfma16_gin_kernel):
//   lstm16_pt_kernel<G,NT>       fp32: v_mfma_f32_16x16x4_f32, W_hh slice in VGPRs, exact-fp32 (the parity path)
//   lstm16_pt_bf16_kernel<KB,NT> bf16: v_mfma_f32_16x16x32_bf16, bf16 W_hh / h_t operands, fp32 accumulate, fp32 cell
//                                state, bf16 folded table (BASELINE config 5)
#include <type_traits>
#include "common.hpp"
#include <hip/hip_bf16.h>
#include <string>
#include <algorithm>

namespace nir {

int launch_linear(const float* a, int64_t lda, const int64_t* ids, const float* table, int E, int64_t rows_per_seq,
                  int64_t seq_stride, const float* w, int64_t ldw, const float* bias, const float* bias2, float* c,
                  int64_t ldc, int64_t M, int N, int K, int act, hipStream_t st);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// wperm[(dir*H + unit)*4 + gate][:] = wih[dir*4H + gate*H + unit][:],  bperm likewise = bih + bhh
__global__ void fold_permute_kernel(const float* __restrict__ wih, const float* __restrict__ bih, const float* __restrict__ bhh,
                                    int H, int ND, int E, float* __restrict__ wperm, float* __restrict__ bperm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows = (int64_t)ND * 4 * H;
    if (i >= rows * E) return;
    const int64_t ro = i / E;
    const int k = (int)(i % E);
    const int gate = (int)(ro & 3), unit = (int)((ro >> 2) % H), dir = (int)((ro >> 2) / H);
    const int64_t ri = (int64_t)dir * 4 * H + (int64_t)gate * H + unit;
    wperm[i] = wih[ri * E + k];
    if (k == 0) bperm[ro] = bih[ri] + bhh[ri];
}

__global__ void iota_i64_kernel(int64_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// (used by the per-batch form of the CARS encoders, cars.hip) the LSTM input weights in the folded gate order + 0, 1, 2, .. as "token ids" of a per-batch gate tensor
int launch_fold_permute(const float* w_ih, const float* b_ih, const float* b_hh, int H, int ndir, int E, float* wperm, float* bperm, int64_t* iota, int64_t niota,
                        hipStream_t st) {
    const int64_t n = (int64_t)ndir * 4 * H * E;
    hipLaunchKernelGGL(fold_permute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_ih, b_ih, b_hh, H, ndir, E, wperm, bperm);
    if (iota && niota > 0) hipLaunchKernelGGL(iota_i64_kernel, dim3((unsigned)((niota + 255) / 256)), dim3(256), 0, st, iota, niota);
    NIR_CHECK_LAUNCH("fold_permute_kernel");
    return 0;
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        ushort4 o;
        o.x = f2bf(v.x); o.y = f2bf(v.y); o.z = f2bf(v.z); o.w = f2bf(v.w);
        *reinterpret_cast<ushort4*>(dst + i) = o;
    } else {
        for (int64_t j = i; j < n; ++j) dst[j] = f2bf(src[j]);
    }
}

struct LstmPtArgs {
    const void* pt;         // folded table [V][ND][H][4]  (fp32 or bf16)
    const int64_t* ids;     // [M,T]
    const int64_t* lens;    // [M] or NULL
    const float* whh;       // [ND,4H,H]  (state-dict layout, fp32)
    float* out;             // [M,T,ND*H] fp32
    int* err;               // device flag (may be NULL): bit 0 = an id fell outside [0,V); bit 1 = |w_hh| outside the fp16 (split) range
    int64_t M, V;
    int T, H, ND;
    int out_f16;            // 1 (bf16-table kernels): `out` is [M,T,ND*H] fp16.  2 (f32 table, lstm16_pt_h2_kernel<4,4,8>): `out` keeps 4 bytes per
                            // element, but every group of 4 consecutive units holds [4 x fp16 leading term | 4 x fp16 residual x 2^11] -- the two-term
                            // split the kernel forms anyway for its own next step, in the order attn_pool_pipe_kernel stages its LDS planes
    const void* whh_frag;   // optional: W_hh pre-split into the two fp16 terms, in the lane order of lstm16_pt_h2_kernel<4,4,8> (nir_lstm_pack_whh_frag)
    float* act;             // train-mode forward (lstm16_pt_h2_kernel<4,4,8,false,true>): [M,T,ND,4H] gate activations i,f,g,o (gate-major inside a direction)
    float* cst;             //   and [M,T,ND,H] cell states of every valid step, for the backward pass (csrc/train.hip)
};

// ---------------------------------------------------------------------------------------------------------------------
// fp32 recurrence over the folded table
// ---------------------------------------------------------------------------------------------------------------------
template <int G, int NT>
__global__ __launch_bounds__(1024) void lstm16_pt_kernel(LstmPtArgs p) {
    constexpr int SEQ = 16, KP = 16 * G, ZLD = KP + 4;
    constexpr uint32_t OOB = 0x7FFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* z = smem;                                   // [2][SEQ][ZLD]
    int* lens_s = reinterpret_cast<int*>(z + 2 * SEQ * ZLD);
    int* ids_s = lens_s + SEQ;                         // [SEQ][T]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sq = lane & 15, kq = lane >> 4;          // operand view: (row/col sq, k quarter kq); result view: (sequence sq, local unit kq)
    const int dir = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * SEQ;
    const int H = p.H, T = p.T, H4 = 4 * H;
    const int nvalid = (int)min((int64_t)SEQ, p.M - m0);
    const int OW = p.ND * H;
    const int64_t GW = (int64_t)p.ND * H4;
    const int ntiles = (H + 3) / 4;

    if (tid < SEQ) {
        int l = 0;
        if (tid < nvalid) {
            l = p.lens ? (int)p.lens[m0 + tid] : T;
            l = l < 0 ? 0 : (l > T ? T : l);
        }
        lens_s[tid] = l;
    }
    {   // token ids of the 16 sequences -> LDS (int32), validated against V (nn.Embedding raises IndexError there)
        bool bad = false;
        for (int e = tid; e < SEQ * T; e += 1024) {
            const int s = e / T;
            int64_t id = 0;
            if (s < nvalid) id = p.ids[m0 * T + e];
            if (id < 0 || id >= p.V) { bad = true; id = 0; }
            ids_s[e] = (int)id;
        }
        if (bad && p.err) atomicOr(p.err, 1);
    }
    for (int e = tid; e < 2 * SEQ * ZLD; e += 1024) z[e] = 0.f;
    __syncthreads();
    int tmax = 0;
#pragma unroll
    for (int s2 = 0; s2 < SEQ; ++s2) tmax = max(tmax, lens_s[s2]);
    const int mylen = lens_s[sq];

    float wreg[NT][4 * G];
    float creg[NT], hreg[NT];
    int unit_d[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tile = NT * wave + t;
        const int unit_a = 4 * tile + (sq >> 2), gate_a = sq & 3;
        const bool av = unit_a < H;
        const float* wr = p.whh + ((int64_t)dir * H4 + (int64_t)gate_a * H + (av ? unit_a : 0)) * H;
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * q + 4 * kq + j;
                wreg[t][4 * q + j] = wr[k < H ? k : H - 1] * ((av && k < H) ? 1.f : 0.f);   // unconditional load, 0/1 mask (see the h2 kernel)
            }
        unit_d[t] = 4 * tile + kq;
        creg[t] = 0.f;
        hreg[t] = 0.f;
    }
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(p.out + m0 * T * OW, 0,
                                                                             (int)((uint32_t)nvalid * T * OW * 4u), 0x00020000);
    const float* ptf = reinterpret_cast<const float*>(p.pt) + (int64_t)dir * H4;
    auto load_g = [&](int step, f32x4 (&dst)[NT]) {
        int s_ = min(step, mylen - 1);
        s_ = s_ < 0 ? 0 : s_;
        const int t_ = dir == 0 ? s_ : mylen - 1 - s_;
        const int id = ids_s[sq * T + (t_ < 0 ? 0 : t_)];
        const float* row = ptf + (int64_t)id * GW;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int u = unit_d[t] < H ? unit_d[t] : H - 1;      // branch-free: out-of-range units re-read a valid lane's data
            dst[t] = *reinterpret_cast<const f32x4*>(row + 4 * u);
        }
    };
    f32x4 gcur[NT], gnext[NT];
    load_g(0, gcur);

    const bool has1 = NT * wave < ntiles;
    const bool has2 = NT > 1 && NT * wave + 1 < ntiles;   // wave-uniform
    for (int step = 0; step < tmax; ++step) {
        const float* zc = z + (step & 1) * SEQ * ZLD;
        float* zn = z + ((step + 1) & 1) * SEQ * ZLD;
        load_g(step + 1, gnext);                     // lands during this step's MFMAs
        const bool live = step < mylen;
        const int tt = dir == 0 ? step : mylen - 1 - step;
        auto cell_tile = [&](int t, const f32x4& acc) {
            const bool dv = unit_d[t] < H;
            const float gi = fast_sigmoid(acc[0] + gcur[t][0]);
            const float gf = fast_sigmoid(acc[1] + gcur[t][1]);
            const float gg = fast_tanh(acc[2] + gcur[t][2]);
            const float go = fast_sigmoid(acc[3] + gcur[t][3]);
            const float cn = gf * creg[t] + gi * gg;
            const float hn = go * fast_tanh(cn);
            const bool act = dv && live;
            creg[t] = act ? cn : creg[t];            // a finished sequence carries its state over
            hreg[t] = act ? hn : hreg[t];
            if (dv) zn[sq * ZLD + unit_d[t]] = hreg[t];
            const uint32_t off = act ? (uint32_t)((sq * T + tt) * OW + dir * H + unit_d[t]) * 4u : OOB;
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
#pragma unroll
        for (int t = 0; t < NT; ++t) gcur[t] = gnext[t];
        lds_barrier();
    }

    // zero the padded tail (pad_packed_sequence)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (unit_d[t] < H && sq < nvalid) {
            const int64_t m = m0 + sq;
            for (int t2 = mylen; t2 < T; ++t2) p.out[(m * T + t2) * OW + (int64_t)dir * H + unit_d[t]] = 0.f;
        }
    }
}

// LSTM cell on the four pre-activations x = (i, f, g, o) of one unit, merged fractions: sigma(i) tanh(g) = sgn(g) (1 - d) / ((1 + a)(1 + d))
// with a = e^-i, d = e^-2|g| (d in (0, 1]; a = inf gives 1 / inf = 0, the right limit), likewise o and tanh(c): 5 v_exp_f32 + 3 v_rcp_f32
// instead of 5 + 5 -- the transcendentals are quarter rate, and the VALU port (gate math of all waves of a SIMD) is as loaded as the
// matrix pipe in these recurrences.  The plain arithmetic is packed along the gate axis, (i, f) and (g, o) are adjacent accumulator
// registers: v_pk_{fma,mul,add}_f32 without any register shuffling.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lstm_cell_v(const f32x4 x, float& c, float& h) {
    constexpr float L2E = 1.4426950408889634f;
    const f32x2 e_if = (f32x2){x[0], x[1]} * (f32x2){-L2E, -L2E};
    const f32x2 e_go = (f32x2){fabsf(x[2]), x[3]} * (f32x2){-2.f * L2E, -L2E};
    const float a = __builtin_amdgcn_exp2f(e_if.x), b = __builtin_amdgcn_exp2f(e_if.y);
    const float d = __builtin_amdgcn_exp2f(e_go.x), q = __builtin_amdgcn_exp2f(e_go.y);
    const f32x2 p_ab = (f32x2){a, b} + (f32x2){1.f, 1.f};
    const f32x2 p_dq = (f32x2){d, q} + (f32x2){1.f, 1.f};
    const float r1 = __builtin_amdgcn_rcpf(p_ab.x * p_dq.x), rf = __builtin_amdgcn_rcpf(p_ab.y);
    c = fmaf(c, rf, copysignf((1.f - d) * r1, x[2]));
    const float e = __builtin_amdgcn_exp2f(fabsf(c) * (-2.f * L2E));
    h = copysignf((1.f - e) * __builtin_amdgcn_rcpf(p_dq.y * (1.f + e)), c);
}

// The same cell for N units at once, written statement by statement ACROSS the units: program order is then N independent dependence chains
// interleaved (every instruction's operand was produced N instructions earlier), which is what an in-order wave needs when this block is
// issued between the MFMAs of another sequence group -- unit by unit, each v_exp / v_rcp result was consumed by the very next instruction.
template <int N>
__device__ __forceinline__ void lstm_cell_vn(const f32x4 (&x)[N], float (&c)[N], float (&h)[N]) {
    constexpr float L2E = 1.4426950408889634f;
    f32x2 e_if[N], e_go[N], p_ab[N], p_dq[N];
    float a[N], b[N], d[N], q[N], r1[N], rf[N], e[N], t1[N], t2[N];
#pragma unroll
    for (int i = 0; i < N; ++i) e_if[i] = (f32x2){x[i][0], x[i][1]} * (f32x2){-L2E, -L2E};
#pragma unroll
    for (int i = 0; i < N; ++i) e_go[i] = (f32x2){fabsf(x[i][2]), x[i][3]} * (f32x2){-2.f * L2E, -L2E};
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = __builtin_amdgcn_exp2f(e_if[i].x);
#pragma unroll
    for (int i = 0; i < N; ++i) b[i] = __builtin_amdgcn_exp2f(e_if[i].y);
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = __builtin_amdgcn_exp2f(e_go[i].x);
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = __builtin_amdgcn_exp2f(e_go[i].y);
#pragma unroll
    for (int i = 0; i < N; ++i) p_ab[i] = (f32x2){a[i], b[i]} + (f32x2){1.f, 1.f};
#pragma unroll
    for (int i = 0; i < N; ++i) p_dq[i] = (f32x2){d[i], q[i]} + (f32x2){1.f, 1.f};
#pragma unroll
    for (int i = 0; i < N; ++i) t1[i] = p_ab[i].x * p_dq[i].x;
#pragma unroll
    for (int i = 0; i < N; ++i) r1[i] = __builtin_amdgcn_rcpf(t1[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) rf[i] = __builtin_amdgcn_rcpf(p_ab[i].y);
#pragma unroll
    for (int i = 0; i < N; ++i) t2[i] = (1.f - d[i]) * r1[i];
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = fmaf(c[i], rf[i], copysignf(t2[i], x[i][2]));
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __builtin_amdgcn_exp2f(fabsf(c[i]) * (-2.f * L2E));
#pragma unroll
    for (int i = 0; i < N; ++i) t1[i] = p_dq[i].y * (1.f + e[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) r1[i] = __builtin_amdgcn_rcpf(t1[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) h[i] = copysignf((1.f - e[i]) * r1[i], c[i]);
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 recurrence over the bf16 folded table: W_hh and h_t as bf16 MFMA operands (v_mfma_f32_16x16x32_bf16), fp32
// accumulators, fp32 gate math and cell state.  KB = ceil(H / 32) K-blocks; wave w owns tiles NT*w .. NT*w+NT-1.
// A operand (16 gate rows x 32 k): lane (row = lane & 15, k = 8*(lane >> 4) + j), B operand (32 k x 16 sequences): lane
// (col = lane & 15, k = 8*(lane >> 4) + j); C/D: col = lane & 15 (sequence), row = 4*(lane >> 4) + r (unit kq, gate r).
// ---------------------------------------------------------------------------------------------------------------------
// O16: h_t leaves as fp16 rows [M,T,ND*H] (the attention-pooling pipeline takes single fp16 terms from a bf16 encoder anyway): the fp16
// copy of h_t that the next step's MFMAs read from LDS IS the output, so wave w streams sequence w's row (256 contiguous bytes at
// H = 128) with one store at the top of the next step instead of NT scattered 4-byte stores per lane.
template <int KB, int NT, bool O16>
__global__ __launch_bounds__(1024) void lstm16_pt_bf16_kernel(LstmPtArgs p) {
    constexpr int SEQ = 16, KP = 32 * KB, ZLD = KP + 8;     // bf16 elements per h row (+8: 16-byte aligned, bank-staggered)
    constexpr uint32_t OOB = 0x7FFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* z = reinterpret_cast<unsigned short*>(smem);   // [2][SEQ][ZLD] fp16 (W_hh and h_t are fp16 MFMA operands: 11
                                                                    // mantissa bits, three more than bf16, same matrix-pipe rate)
    int* lens_s = reinterpret_cast<int*>(z + 2 * SEQ * ZLD);
    int* ids_s = lens_s + SEQ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sq = lane & 15, kq = lane >> 4;
    const int dir = blockIdx.y;
    const int H = p.H, T = p.T, H4 = 4 * H;
    const int OW = p.ND * H;
    const int64_t GW = (int64_t)p.ND * H4;
    const int ntiles = (H + 3) / 4;
    f16x8 wreg[NT][KB];
    float creg[NT];
    int unit_d[NT];
    bool wbad = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tile = NT * wave + t;
        const int unit_a = 4 * tile + (sq >> 2), gate_a = sq & 3;
        const bool av = unit_a < H;
        const float* wr = p.whh + ((int64_t)dir * H4 + (int64_t)gate_a * H + (av ? unit_a : 0)) * H;
        // the lane's 8*KB weights of this tile as 2*KB independent 16-byte loads (a scalar load -> convert loop serialises 64 memory
        // round trips: ~35 us of prologue per workgroup, a quarter of a 64-step workgroup's life at the C5 shape)
        const bool vec_ok = (H % 8) == 0 && ((reinterpret_cast<uintptr_t>(wr) & 15) == 0);
        if (vec_ok) {                                         // wave-uniform
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int k0 = 32 * kb + 8 * kq;
                const int kc = k0 + 8 <= H ? k0 : 0;           // branch-free: out-of-range chunks re-read chunk 0 and are zeroed below
                const float4 a = *reinterpret_cast<const float4*>(wr + kc), b = *reinterpret_cast<const float4*>(wr + kc + 4);
                const float m = (av && k0 + 8 <= H) ? 1.f : 0.f;
                wreg[t][kb][0] = (_Float16)(a.x * m); wreg[t][kb][1] = (_Float16)(a.y * m);
                wreg[t][kb][2] = (_Float16)(a.z * m); wreg[t][kb][3] = (_Float16)(a.w * m);
                wreg[t][kb][4] = (_Float16)(b.x * m); wreg[t][kb][5] = (_Float16)(b.y * m);
                wreg[t][kb][6] = (_Float16)(b.z * m); wreg[t][kb][7] = (_Float16)(b.w * m);
                wbad |= !(fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                                fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))) * m < 65504.0f);
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * kb + 8 * kq + j;
                    const float w = wr[k < H ? k : H - 1] * ((av && k < H) ? 1.f : 0.f);
                    wreg[t][kb][j] = (_Float16)w;
                    wbad |= !(fabsf(w) < 65504.0f);
                }
        }
        unit_d[t] = 4 * tile + kq;
    }
    if (wbad && p.err) atomicOr(p.err, 2);          // |w_hh| outside fp16's range (or NaN): flagged, never silently wrong
    // persistent over sequence tiles: W_hh is fetched and converted once per workgroup, not once per 16 sequences (the launcher sizes the
    // grid to the CU count when the tiles outnumber it)
    for (int64_t mt = blockIdx.x; mt * SEQ < p.M; mt += gridDim.x) {
        __syncthreads();                                 // the previous tile's last readers of lens_s / ids_s are done
        const int64_t m0 = mt * SEQ;
        const int nvalid = (int)min((int64_t)SEQ, p.M - m0);

        if (tid < SEQ) {
            int l = 0;
            if (tid < nvalid) {
                l = p.lens ? (int)p.lens[m0 + tid] : T;
                l = l < 0 ? 0 : (l > T ? T : l);
            }
            lens_s[tid] = l;
        }
        {
            bool bad = false;
            for (int e = tid; e < SEQ * T; e += 1024) {
                const int s = e / T;
                int64_t id = 0;
                if (s < nvalid) id = p.ids[m0 * T + e];
                if (id < 0 || id >= p.V) { bad = true; id = 0; }
                ids_s[e] = (int)id;
            }
            if (bad && p.err) atomicOr(p.err, 1);
        }
        for (int e = tid; e < SEQ * ZLD; e += 1024) reinterpret_cast<unsigned*>(z)[e] = 0u;   // both buffers (2*SEQ*ZLD bf16)
        __syncthreads();
        int tmax = 0;
#pragma unroll
        for (int s2 = 0; s2 < SEQ; ++s2) tmax = max(tmax, lens_s[s2]);
        const int mylen = lens_s[sq];

#pragma unroll
        for (int t = 0; t < NT; ++t) creg[t] = 0.f;
        const __amdgpu_buffer_rsrc_t out_rs =
            O16 ? __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<_Float16*>(p.out) + m0 * T * OW, 0, (int)((uint32_t)nvalid * T * OW * 2u), 0x00020000)
                : __builtin_amdgcn_make_buffer_rsrc(p.out + m0 * T * OW, 0, (int)((uint32_t)nvalid * T * OW * 4u), 0x00020000);
        // O16: wave w copies sequence w's row of h_{t-1} (lane = two consecutive units) out of the LDS buffer the MFMAs of step t read
        const int wlen = __builtin_amdgcn_readfirstlane(lens_s[wave & (SEQ - 1)]);
        auto copy_out = [&](int s_, const unsigned short* zsrc) {
            const bool on = s_ >= 0 && s_ < wlen && 2 * lane < H;
            const int tt_ = dir == 0 ? s_ : wlen - 1 - s_;
            const uint32_t v = *reinterpret_cast<const uint32_t*>(zsrc + wave * ZLD + 2 * lane);
            const uint32_t off = on ? (uint32_t)((wave * T + tt_) * OW + dir * H + 2 * lane) * 2u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(v, out_rs, off, 0, 0);
        };
        const unsigned short* pth = reinterpret_cast<const unsigned short*>(p.pt) + (int64_t)dir * H4;
        auto id_of = [&](int step) {
            int s_ = min(step, mylen - 1);
            s_ = s_ < 0 ? 0 : s_;
            const int t_ = dir == 0 ? s_ : mylen - 1 - s_;
            return ids_s[sq * T + (t_ < 0 ? 0 : t_)];
        };
        auto load_g = [&](int id, uint2 (&dst)[NT]) {
            const unsigned short* row = pth + (int64_t)id * GW;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int u = unit_d[t] < H ? unit_d[t] : H - 1;
                dst[t] = *reinterpret_cast<const uint2*>(row + 4 * u);
            }
        };
        // Same step structure as lstm16_pt_h2_kernel below: the gate rows of step t are handed over at the top of step t (the only vmcnt
        // wait of the loop, on a request that is two steps old -- a bf16 step is shorter than an HBM round trip) and ride in as the MFMA C
        // operand; the rows of step t+2 are requested under the MFMA phase from an id looked up one step earlier; every path issues NT stores
        // per step (out-of-range offset = dropped) so the waitcnt pass can count them; no hold registers for finished sequences.
        uint2 ga[NT], gb[NT];
        load_g(id_of(0), ga);
#pragma unroll
        for (int t = 0; t < (O16 ? 1 : NT); ++t) __builtin_amdgcn_raw_buffer_store_b32(0u, out_rs, OOB, 0, 0);
        load_g(id_of(1), gb);
#pragma unroll
        for (int t = 0; t < (O16 ? 1 : NT); ++t) __builtin_amdgcn_raw_buffer_store_b32(0u, out_rs, OOB, 0, 0);
        int id_n = id_of(2);

        for (int step = 0; step < tmax; ++step) {
            const unsigned short* zc = z + (step & 1) * SEQ * ZLD;
            unsigned short* zn = z + ((step + 1) & 1) * SEQ * ZLD;
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                asm volatile("" : "+v"(ga[t].x), "+v"(ga[t].y));     // pins the hand-over (and its wait) to this point
                acc[t] = (f32x4){bf2f((unsigned short)(ga[t].x & 0xFFFFu)), bf2f((unsigned short)(ga[t].x >> 16)),
                                 bf2f((unsigned short)(ga[t].y & 0xFFFFu)), bf2f((unsigned short)(ga[t].y >> 16))};
                ga[t] = gb[t];
            }
            const bool live = step < mylen;
            const int tt = dir == 0 ? step : mylen - 1 - step;
            const unsigned short* zr = zc + sq * ZLD + 8 * kq;
            if (O16) copy_out(step - 1, zc);             // step 0: nothing yet (a dropped store keeps the per-step store count fixed)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const f16x8 hb = *reinterpret_cast<const f16x8*>(zr + 32 * kb);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[t][kb], hb, acc[t], 0, 0, 0);
                if (kb == 0) {
                    load_g(id_n, gb);
                    id_n = id_of(step + 3);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                uint32_t off = OOB;
                float hv = 0.f;
                if (NT * wave + t < ntiles) {            // wave-uniform
                    const bool dv = unit_d[t] < H;
                    float hn;
                    lstm_cell_v(acc[t], creg[t], hn);
                    if (dv) reinterpret_cast<_Float16*>(zn)[sq * ZLD + unit_d[t]] = (_Float16)hn;
                    if (dv && live) off = (uint32_t)((sq * T + tt) * OW + dir * H + unit_d[t]) * 4u;
                    hv = hn;
                }
                if (!O16) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hv), out_rs, off, 0, 0);
            }
            lds_barrier();
        }
        if (O16 && tmax > 0) copy_out(tmax - 1, z + (tmax & 1) * SEQ * ZLD);
        // zero the padded steps of this direction's half: one wave per (sequence, step) row, coalesced (ragged batches are the normal case)
        for (int s_ = 0; s_ < nvalid; ++s_) {
            const int64_t ob = (m0 + s_) * T * OW + (int64_t)dir * H;
            for (int t2 = lens_s[s_] + wave; t2 < T; t2 += 16)
                for (int col = lane; col < H; col += 64) {
                    if (O16) reinterpret_cast<_Float16*>(p.out)[ob + (int64_t)t2 * OW + col] = (_Float16)0.f;
                    else p.out[ob + (int64_t)t2 * OW + col] = 0.f;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32-accurate recurrence on the fp16 matrix cores (the default parity path for H % 32 == 0):
//   w = w1 + 2^-11 w2',  h = h1 + 2^-11 h2'   with  w1 = fp16(w), w2' = fp16(2^11 (w - w1))  (both terms carry 11 mantissa bits;
//   the 2^11 scaling keeps the residual in fp16's normal range), so   w.h = w1.h1 + 2^-11 (w1.h2' + w2'.h1) + O(2^-22 |w.h|)
// Three v_mfma_f32_16x16x32_f16 per 32-wide k-block (two accumulators: leading term, scaled cross terms) replace the 32
// v_mfma_f32_16x16x4_f32 of the fp32 kernel: 12 x 16 cycles instead of 32 x 32 cycles of matrix pipe per tile and step, with
// the same fp32-class error (measured against fp64: 2.9e-6 vs 3.4e-6 for the fp32 chain at K = 128).  Gate math, cell state,
// the folded table and the output stay fp32.  W_hh terms live in VGPRs (2 x 16 per tile), the two h terms in LDS.
// ---------------------------------------------------------------------------------------------------------------------

#ifdef NIR_PT_TRACE   // tools/recur_micro.py --trace: per-wave phase clocks of workgroup (0, 0), summed over the steps
__device__ unsigned long long* g_pt_trace_dev;
#define PT_T(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#endif
// NW waves per workgroup, NT gate tiles per wave (NW * NT * 4 >= H units).  16 waves x 2 tiles is the latency form (H = 128); for
// H <= 80 four waves x 5 tiles leave room for three workgroups per CU, which fill each other's per-step bubbles when several
// batches are in flight.
// H1 (round 5, the opt-in "split2" precision tier, <4,4,8,true> only): h enters the recurrent product as ONE fp16 term -- w.h = w1.h1 + 2^-11 w2'.h1:
// two MFMAs per k-block instead of three, one LDS h plane instead of two -- and leaves as fp16 rows [M,T,ND*H] (out_f16 == 3) for
// attn_pool_pipe_kernel<false,1>.  Measured error and where it holds: DESIGN.md section 10; never selected by default.
// TR (round 5, <4,4,8,false,true>, H = 128): the train-mode forward -- the "table" is the batch's own gate tensor (ids = 0, 1, 2, ..), and the
// gate activations and the cell state of every valid step go to memory for the backward (the cell is then written gate by gate: the
// backward needs i, f, g, o themselves, not the merged fractions of lstm_cell_v).
template <int KB, int NT, int NW, bool H1 = false, bool TR = false>
__global__ __launch_bounds__(64 * NW, NW <= 4 ? 2 : 1) void lstm16_pt_h2_kernel(LstmPtArgs p) {   // (second argument: waves per SIMD)
    static_assert(!TR || !H1, "train-mode stores go with the fp32 output");
    constexpr int NTH = 64 * NW;
    constexpr int SEQ = 16, KP = 32 * KB, ZLD = KP + 8;     // fp16 elements per h row
    constexpr uint32_t OOB = 0x7FFFFFF0u;
    constexpr float SC = 2048.0f, ISC = 1.0f / 2048.0f;
    // LDS layout of an h row (round 6).  The B fragments are read with ds_read_b128, whose lane groups are {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
    // (MI355X_MICROARCH.md, LDS): a group mixes sequences {0-3,12-15} of k-quarter kq = 2j with sequences {4-11} of kq = 2j + 1.  In the plain
    // [sequence][k] layout (row pitch 17 x 16 B) the two sets land one 16-byte slot apart and overlap in one slot: 2-way in every group, +4 LDS
    // cycles on each of the 64 fragment reads a CU issues right behind the step's barrier (SQ_LDS_BANK_CONFLICT 3.4 per LDS instruction in
    // round 5's capture).  KB == 4 (16 pieces of 16 B per row and term): piece (kb, kq) of sequence s sits in row s ^ 8 (kq & 1) at position
    // 8 (kq & 1) + 2 kb + (kq >> 1) -- the odd k-quarters are shifted by eight slots AND eight rows, so a group's 16 lanes cover 16 distinct slots.
    constexpr bool SWZ = KB == 4;
    constexpr int KBS = SWZ ? 16 : 32;                      // halves between the pieces of consecutive k-blocks of one lane
    auto zoff = [](int s_, int k_) -> int {                 // element offset of (sequence s_, k index k_) inside a term plane
        if constexpr (SWZ) {
            const int q_ = (k_ >> 3) & 3, b_ = k_ >> 5;
            return (s_ ^ (8 * (q_ & 1))) * ZLD + (8 * (q_ & 1) + 2 * b_ + (q_ >> 1)) * 8 + (k_ & 7);
        } else {
            return s_ * ZLD + k_;
        }
    };
#ifdef NIR_X_NOPIPE
    constexpr bool PIPE = false;
#else
    constexpr bool PIPE = NT >= 4 && NW == 8;               // in-wave software pipeline over the tiles (see the loop); 4 waves x 5 tiles: slower with it
#endif
    constexpr bool DEFER = NW >= 8;                         // output stores one step late, in front of the row requests (see store_prev)
    const int TP = p.T + 3;                                 // id columns per sequence: the lookup runs two steps ahead
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* z = reinterpret_cast<_Float16*>(smem);        // [2 buffers][2 terms][SEQ][ZLD]
    int* lens_s = reinterpret_cast<int*>(z + 4 * SEQ * ZLD);
    int* simd_s = lens_s + SEQ;                             // [16] SIMD of every wave (issue-priority ranking below)
    int* ids_s = simd_s + 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sq = lane & 15, kq = lane >> 4;
    const int dir = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * SEQ;
    const int H = p.H, T = p.T, H4 = 4 * H;
    const int nvalid = (int)min((int64_t)SEQ, p.M - m0);
    const int OW = p.ND * H;
    const int64_t GW = (int64_t)p.ND * H4;
    f16x8 w1[NT][KB], w2[NT][KB];
    const bool use_frag = KB == 4 && NT == 4 && NW == 8 && p.whh_frag != nullptr && H == 128;   // wave-uniform
    if (use_frag) {
        // W_hh arrives pre-split in this kernel's lane order (packed once per weight version): 32 independent 16-byte loads per lane, no
        // conversion arithmetic (the in-kernel split below costs ~1 000 VALU instructions per wave and workgroup), requested before anything
        // else so that their round trip runs under the id staging (the barriers of the prologue are LDS-only: no vmcnt wait)
        const f16x8* fp = reinterpret_cast<const f16x8*>(p.whh_frag) + ((size_t)(dir * NW + wave) * NT * KB * 2) * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                w1[t][kb] = fp[((t * KB + kb) * 2 + 0) * 64];
                w2[t][kb] = fp[((t * KB + kb) * 2 + 1) * 64];
            }
    }
    if (NW > 4 && lane == 0) simd_s[wave] = (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));   // HW_REG_HW_ID bits [5:4] = SIMD_ID
    if (tid < SEQ) {
        int l = 0;
        if (tid < nvalid) {
            l = p.lens ? (int)p.lens[m0 + tid] : T;
            l = l < 0 ? 0 : (l > T ? T : l);
        }
        lens_s[tid] = l;
    }
    lds_barrier();
    {
        // ids_s[s][k] = the id sequence s consumes at STEP k (reverse direction: from its last valid token down; past its end the last
        // valid id repeats -- those gate rows feed nothing that is stored): the loop's lookup is one LDS read at base + 4 * step.
        bool bad = false;
        for (int e = tid; e < SEQ * TP; e += NTH) {
            const int s_ = e / TP, k = e - s_ * TP;
            int64_t id = 0;
            if (s_ < nvalid) {
                const int l = lens_s[s_];
                int kk = k < l - 1 ? k : l - 1;
                kk = kk < 0 ? 0 : kk;
                int t_ = dir == 0 ? kk : l - 1 - kk;
                t_ = t_ < 0 ? 0 : t_;
                id = p.ids[(m0 + s_) * T + t_];
                if (k < T) {                                   // every id of the padded row is validated, like the reference's nn.Embedding
                    const int64_t raw = p.ids[(m0 + s_) * T + k];
                    bad |= raw < 0 || raw >= p.V;
                }
            }
            if (id < 0 || id >= p.V) id = 0;
            ids_s[e] = (int)id;
        }
        if (bad && p.err) atomicOr(p.err, 1);
    }
    for (int e = tid; e < 2 * SEQ * ZLD; e += NTH) reinterpret_cast<unsigned*>(z)[e] = 0u;   // 4*SEQ*ZLD halves
    lds_barrier();
    int tmax = 0;
#pragma unroll
    for (int s2 = 0; s2 < SEQ; ++s2) tmax = max(tmax, lens_s[s2]);
    const int mylen = lens_s[sq];
    if (NW > 4) {
        // The waves that share a SIMD get DISTINCT issue priorities.  With equal priorities the arbiter interleaves their MFMAs one by
        // one, all of them leave the matrix phase together and then queue for the VALU with the matrix pipe idle (the gate math is
        // ~2400 VALU cycles per SIMD and step against ~1500 of MFMA).  Ranked, wave A's MFMAs go back to back, and its gate math
        // runs under wave B's MFMAs, and so on down the ranks.
        const int mine = simd_s[wave];
        int rank = 0;
#pragma unroll
        for (int w2_ = 0; w2_ < NW; ++w2_) rank += (w2_ < wave && simd_s[w2_] == mine) ? 1 : 0;
        rank = __builtin_amdgcn_readfirstlane(rank);
        if (rank == 0) __builtin_amdgcn_s_setprio(3);
        else if (rank == 1) __builtin_amdgcn_s_setprio(2);
        else if (rank == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }

    // Unit mapping: lane (sq, kq) of wave w owns the NT CONSECUTIVE units u0 .. u0 + NT - 1, u0 = NT * (4 w + kq), of sequence sq -- tile t of
    // the wave holds unit u0 + t of each of its four unit groups (A rows 4 g + gate <-> unit NT * (4 w + g) + t).  The lane's output is then
    // one 4 NT-byte store, its h terms one 2 NT-byte LDS write each, its gate rows one 16 NT-byte piece of the folded row.
    float creg[NT];
    const int u0 = NT * (4 * wave + kq);
    // wave-uniform: a wave whose units are all past H skips the step's work (H = 70 on 16 waves x 2 tiles: 7 of 16).  Not in the pipelined form:
    // the branch around its loop body costs the H = 128 kernel 13 % (measured), and at most one of its eight waves can be idle.
    const bool wave_on = PIPE || NT * 4 * wave < H;
    const bool full = u0 + NT <= H;                           // all NT units real (the vector forms); otherwise unit by unit
    bool wbad = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (use_frag) { creg[t] = 0.f; continue; }
        const int unit_a = NT * (4 * wave + (sq >> 2)) + t, gate_a = sq & 3;
        const bool av = unit_a < H;
        const float* wr = p.whh + ((int64_t)dir * H4 + (int64_t)gate_a * H + (av ? unit_a : 0)) * H;
        // the lane's 8*KB weights of this tile: issued as 2*KB independent 16-byte loads up front (a scalar load -> convert loop
        // serialises 64 memory round trips and made the prologue cost ~35 us per launch)
        float wv[KB][8];
        const bool vec_ok = (H % 8) == 0 && ((reinterpret_cast<uintptr_t>(wr) & 15) == 0);
        if (vec_ok) {                                         // wave-uniform
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int k0 = 32 * kb + 8 * kq;
                const int kc = k0 + 8 <= H ? k0 : 0;           // branch-free: out-of-range chunks re-read chunk 0 and are zeroed below
                const float4 a = *reinterpret_cast<const float4*>(wr + kc), b = *reinterpret_cast<const float4*>(wr + kc + 4);
                const float m = (av && k0 + 8 <= H) ? 1.f : 0.f;
                wv[kb][0] = a.x * m; wv[kb][1] = a.y * m; wv[kb][2] = a.z * m; wv[kb][3] = a.w * m;
                wv[kb][4] = b.x * m; wv[kb][5] = b.y * m; wv[kb][6] = b.z * m; wv[kb][7] = b.w * m;
            }
        } else {                                              // any H: unconditional loads from a clamped index, masked afterwards
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * kb + 8 * kq + j;
                    // multiply by a 0/1 mask instead of selecting: a select lets the compiler predicate the load, and each
                    // predicated load became its own exec-masked block with an s_waitcnt vmcnt(0) behind it (64 serialised round trips)
                    wv[kb][j] = wr[k < H ? k : H - 1] * ((av && k < H) ? 1.f : 0.f);
                }
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = wv[kb][j];
                const _Float16 a = (_Float16)w;
                w1[t][kb][j] = a;
                w2[t][kb][j] = (_Float16)((w - (float)a) * SC);
                wbad |= !(fabsf(w) < 32768.0f);             // outside the fp16 split's range (or NaN): flagged, never silently wrong
            }
        creg[t] = 0.f;
    }
    if (wbad && p.err) atomicOr(p.err, 2);
    const __amdgpu_buffer_rsrc_t out_rs = H1 ? __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<_Float16*>(p.out) + m0 * T * OW, 0,
                                                                                  (int)((uint32_t)nvalid * T * OW * 2u), 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc(p.out + m0 * T * OW, 0, (int)((uint32_t)nvalid * T * OW * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t act_rs = __builtin_amdgcn_make_buffer_rsrc(TR ? p.act + m0 * T * GW : p.out, 0,
                                                                             TR ? (int)((uint32_t)nvalid * T * (uint32_t)GW * 4u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t cst_rs = __builtin_amdgcn_make_buffer_rsrc(TR ? p.cst + m0 * T * OW : p.out, 0,
                                                                             TR ? (int)((uint32_t)nvalid * T * OW * 4u) : 0, 0x00020000);
    // per-unit base of the lane's 16-byte gate groups inside a folded row (units past H re-read the last real one: never used);
    // row = base + id * GW floats
    const float* ptf = reinterpret_cast<const float*>(p.pt) + (int64_t)dir * H4;
    const float* pb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) pb[t] = ptf + 4 * (u0 + t < H ? u0 + t : H - 1);
    const uint32_t gw = (uint32_t)GW;
    auto load_g = [&](int id, f32x4 (&dst)[NT]) {
#ifdef NIR_X_NOROWS
        return;
#endif
        const uint64_t ro = (uint64_t)(uint32_t)id * gw;
#pragma unroll
        for (int t = 0; t < NT; ++t) dst[t] = *reinterpret_cast<const f32x4*>(pb[t] + ro);
    };
    f32x4 gnext[NT];
    // DEFER (8 / 16 waves): the output of step t is stored during step t+1, immediately IN FRONT of the row requests for step t+2: the vector-memory queue of a
    // step is then [stores, requests] and the wait the compiler derives for the requests (vmcnt counts in order) never covers a store
    // that is younger than they are.  (With [requests, stores] the derived wait flipped with unrelated edits between "requests only" and
    // "requests and the first store" -- a store round trip, ~1000 cycles, in front of the first MFMA of every step.)
    uint32_t hprev[NT];                                       // fp32 bit patterns, or (split output) the packed term pairs
#pragma unroll
    for (int t = 0; t < NT; ++t) hprev[t] = 0u;
    // split output (out_f16 == 2, H = 128: every lane `full`): the 16 bytes of the lane's four units carry the two fp16 terms instead of four floats
    const bool split = NT == 4 && p.out_f16 == 2;
    uint32_t poff = OOB;                                      // byte offset of (sequence, step, u0) in the output block, OOB = dropped
    auto store_prev = [&]() {
#ifdef NIR_X_NOSTORE
        return;
#endif
        if (H1) {                                             // four fp16 values = 8 bytes at half the fp32 offset
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){hprev[0], hprev[1 % NT]}, out_rs, poff == OOB ? OOB : poff >> 1, 0, 0);
            return;
        }
        if (NT == 4) {
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){hprev[0], hprev[1 % NT], hprev[2 % NT], hprev[3 % NT]}, out_rs, full ? poff : OOB, 0, 0);
        } else if (NT == 2) {
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){hprev[0], hprev[1 % NT]}, out_rs, full ? poff : OOB, 0, 0);
        }
        if ((NT != 4 && NT != 2) || !full) {                  // unit by unit (NT = 4 / 2: only the lanes that straddle H)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                __builtin_amdgcn_raw_buffer_store_b32(hprev[t], out_rs, (u0 + t < H && poff != OOB) ? poff + 4u * t : OOB, 0, 0);
        }
    };
    const int* idp = ids_s + sq * TP;
    load_g(idp[0], gnext);
    int id_n = idp[1];
    uint32_t soff = (uint32_t)(((sq * T + (dir == 0 ? 0 : mylen - 1)) * OW + dir * H + u0) * 4);
    const uint32_t sstep = (uint32_t)(dir == 0 ? OW * 4 : -(OW * 4));
    f32x4 acc[NT], acx[NT];
    float hn[NT] = {};
#ifdef NIR_X_NOGATES   // ablation: the step without its gate math (tools/recur_micro.py)
    auto gates = [&](int t) { hn[t] = (acx[t][0] + acc[t][1]) * 1e-3f; };
#else
    float tga[TR ? NT : 1][4];                               // train mode: this step's i, f, g, o of the lane's units
    auto gates = [&](int t) {
        if constexpr (TR) {
            const f32x4 x = acx[t] * ISC + acc[t];
            const float gi = fast_sigmoid(x[0]), gf = fast_sigmoid(x[1]), gg = fast_tanh(x[2]), go = fast_sigmoid(x[3]);
            const float cn = gf * creg[t] + gi * gg;
            creg[t] = cn;
            hn[t] = go * fast_tanh(cn);
            tga[t][0] = gi; tga[t][1] = gf; tga[t][2] = gg; tga[t][3] = go;
        } else {
            lstm_cell_v(acx[t] * ISC + acc[t], creg[t], hn[t]);
        }
    };
    uint32_t aoff = (uint32_t)(((int64_t)(sq * T + (dir == 0 ? 0 : mylen - 1)) * GW + dir * H4 + u0) * 4);
    const uint32_t astep = (uint32_t)(dir == 0 ? GW * 4 : -(GW * 4));
#endif
#ifdef NIR_PT_TRACE
    unsigned long long tr_a = 0, tr_b = 0, tr_c = 0, tr_t0 = 0, tr_t1 = 0, tr_t2 = 0, tr_first = 0, tr_l = 0, tr_m = 0;
#endif
    for (int step = 0; step < tmax; ++step) {
#ifdef NIR_PT_TRACE
        { unsigned long long tn; PT_T(tn); if (step > 0) tr_c += tn - tr_t2; else tr_first = tn; tr_t0 = tn; }
#endif
        const _Float16* zc = z + (step & 1) * 2 * SEQ * ZLD;
        _Float16* zn = z + ((step + 1) & 1) * 2 * SEQ * ZLD;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = gnext[t];                       // the gate rows ride in as the MFMA's C operand
            acx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const bool live = step < mylen;
        const _Float16* zr = zc + zoff(sq, 8 * kq);
        if (!wave_on) {                                // all of this wave's units are past H: it only keeps the barriers company
#ifdef NIR_X_BURST
        } else if constexpr (NT == 4 && NW == 8) {
            // Burst form: all h fragments read once, then the step's 48 MFMAs issued back to back with no dependent pair adjacent (the same
            // accumulator recurs every fourth MFMA), then the gate math of the four tiles as ONE straight-line block (four independent chains)
            f16x8 hh1[KB], hh2[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                hh1[kb] = *reinterpret_cast<const f16x8*>(zr + KBS * kb);
                hh2[kb] = *reinterpret_cast<const f16x8*>(zr + SEQ * ZLD + KBS * kb);
            }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[t][kb], hh1[kb], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[t][kb], hh2[kb], acx[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[t][kb], hh1[kb], acx[t], 0, 0, 0);
                if (kb == 0) {
                    if (DEFER) store_prev();
                    load_g(id_n, gnext);
                    id_n = idp[step + 2];
                }
            }
#ifdef NIR_PT_TRACE
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t]), "+v"(acx[t]));
            PT_T(tr_t1); tr_a += tr_t1 - tr_t0;
#endif
#pragma unroll
            for (int t = 0; t < NT; ++t) gates(t);
#endif
        } else if constexpr (!PIPE) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {            // h terms are read per k-block (8 live VGPRs instead of 8*KB)
                const f16x8 h1 = *reinterpret_cast<const f16x8*>(zr + KBS * kb);
                const f16x8 h2 = *reinterpret_cast<const f16x8*>(zr + SEQ * ZLD + KBS * kb);
#pragma unroll
                for (int t = 0; t < NT; ++t) {           // independent accumulators back to back: no dependent-MFMA stalls
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[t][kb], h1, acc[t], 0, 0, 0);
                    acx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[t][kb], h2, acx[t], 0, 0, 0);
                    acx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[t][kb], h1, acx[t], 0, 0, 0);
                }
                if (kb == 0) {
                    if (DEFER) store_prev();
                    load_g(id_n, gnext);
                    id_n = idp[step + 2];
                }
            }
#ifdef NIR_PT_TRACE
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t]), "+v"(acx[t]));
            PT_T(tr_t1); tr_a += tr_t1 - tr_t0;
#endif
            // Gate math of all NT tiles in one straight-line block (units past H compute on zero weights and are masked at the writes); no
            // hold for finished sequences either -- a sequence's column of the B operand only feeds its own gates, and nothing of a
            // finished sequence is stored again (its padded outputs are zero-filled below), in either direction.
#pragma unroll
            for (int t = 0; t < NT; ++t) gates(t);
        } else {
            // Software pipeline over the tiles inside the wave (two waves per SIMD): the MFMAs of tile k+1 are issued over the gate math of
            // tile k -- an MFMA occupies the matrix pipe for 16 cycles and the issue port for 4 -- so only the last tile's gate math of
            // the lower-priority wave is exposed.  All KB h fragments are read once (half the LDS traffic of the 16-wave form: every
            // wave reads the whole B operand, 128 KB per step and CU there, a third of the step at 128 B/clk).
            f16x8 hh1[KB], hh2[H1 ? 1 : KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                hh1[kb] = *reinterpret_cast<const f16x8*>(zr + KBS * kb);
                if (!H1) hh2[H1 ? 0 : kb] = *reinterpret_cast<const f16x8*>(zr + SEQ * ZLD + KBS * kb);
            }
#ifdef NIR_PT_TRACE
            { unsigned long long tn; PT_T(tn); tr_l += tn - tr_t0; }
#endif
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[t][kb], hh1[kb], acc[t], 0, 0, 0);
                    if (!H1) acx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[t][kb], hh2[H1 ? 0 : kb], acx[t], 0, 0, 0);
                    acx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[t][kb], hh1[kb], acx[t], 0, 0, 0);
                }
                if (t == 0) {
                    if (DEFER) store_prev();
                    load_g(id_n, gnext);
                    id_n = idp[step + 2];
                } else {
#ifdef NIR_PT_TRACE
                    if (t == 1) { asm volatile("" : "+v"(acc[0]), "+v"(acx[0])); unsigned long long tn; PT_T(tn); tr_m += tn - tr_t0; }
#endif
                    gates(t - 1);
#pragma unroll
                    for (int q = 0; q < (H1 ? 2 : 3) * KB; ++q) {       // one MFMA, then three of the previous tile's VALU instructions, ...
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    }
                }
            }
#ifdef NIR_PT_TRACE
            asm volatile("" : "+v"(acc[NT - 1]), "+v"(acx[NT - 1]));
            PT_T(tr_t1); tr_a += tr_t1 - tr_t0;
#endif
            gates(NT - 1);
        }
        // the two h terms for the next step's B operand
        if (full && (NT == 4 || NT == 2)) {
            _Float16 a[NT], r[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) a[t] = (_Float16)hn[t];
            // residual = fp16(2^11 (h - a)) from ONE FMA per value: 2^11 a is exact in fp16 (|a| <= 1: a packed fp16 multiply per pair) and
            // h 2^11 - (2^11 a) is the same exact real number as (h - a) 2^11, rounded once -- bit-identical to convert / subtract / scale /
            // convert, 12 VALU per four values instead of 16 (8 with hand-written v_fma_mix{lo,hi}_f16).  Measured (tools/recur_micro.py,
            // M = 8 960, three builds in one run): 16 / 12 / 8 VALU = 658-666 / 657-665 / 665 us -- the step is not bound by VALU issue.
            {
                _Float16 as_[NT];
#pragma unroll
                for (int t = 0; t < NT; t += 2) {
                    const f16x2 p2 = (f16x2){a[t], a[(t + 1) % NT]} * (f16x2){(_Float16)2048.0f, (_Float16)2048.0f};
                    as_[t] = p2[0]; as_[(t + 1) % NT] = p2[1];
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) r[t] = (_Float16)__builtin_fmaf(hn[t], SC, -(float)as_[t]);
            }
            if (NT == 4) {
                const f16x4 av = (f16x4){a[0], a[1 % NT], a[2 % NT], a[3 % NT]}, rv = (f16x4){r[0], r[1 % NT], r[2 % NT], r[3 % NT]};
                *reinterpret_cast<f16x4*>(zn + zoff(sq, u0)) = av;
                if (!H1) *reinterpret_cast<f16x4*>(zn + SEQ * ZLD + zoff(sq, u0)) = rv;
                if (H1) {                                     // the leading terms ARE the output: fp16 rows
                    const u32x2 au = __builtin_bit_cast(u32x2, av);
                    hprev[0] = au[0]; hprev[1 % NT] = au[1];
                } else
                if (split) {                                  // the same two vectors ARE the output (wave-uniform)
                    const u32x2 au = __builtin_bit_cast(u32x2, av), ru = __builtin_bit_cast(u32x2, rv);
                    hprev[0] = au[0]; hprev[1 % NT] = au[1]; hprev[2 % NT] = ru[0]; hprev[3 % NT] = ru[1];
                }
            } else {
                *reinterpret_cast<f16x2*>(zn + zoff(sq, u0)) = (f16x2){a[0], a[1 % NT]};
                *reinterpret_cast<f16x2*>(zn + SEQ * ZLD + zoff(sq, u0)) = (f16x2){r[0], r[1 % NT]};
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (u0 + t < H) {
                    const _Float16 a = (_Float16)hn[t];
                    zn[zoff(sq, u0 + t)] = a;
                    zn[SEQ * ZLD + zoff(sq, u0 + t)] = (_Float16)((hn[t] - (float)a) * SC);
                }
        }
        if (!split && !H1) {
#pragma unroll
            for (int t = 0; t < NT; ++t) hprev[t] = __float_as_uint(hn[t]);
        }
        poff = live ? soff : OOB;
        soff += sstep;
#ifndef NIR_X_NOGATES
        if constexpr (TR) {                        // (younger than the row requests of this step: the wait at the top of the next one does not cover them)
            const uint32_t ao = live ? aoff : OOB;
            if (NT == 4 && full) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(tga[0][r]), __float_as_uint(tga[1 % NT][r]), __float_as_uint(tga[2 % NT][r]),
                                                                   __float_as_uint(tga[3 % NT][r])}, act_rs, ao == OOB ? OOB : ao + (uint32_t)(r * H * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(creg[0]), __float_as_uint(creg[1 % NT]), __float_as_uint(creg[2 % NT]),
                                                               __float_as_uint(creg[3 % NT])}, cst_rs, poff, 0, 0);
            } else if (NT == 2 && full) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(tga[0][r]), __float_as_uint(tga[1 % NT][r])}, act_rs,
                                                          ao == OOB ? OOB : ao + (uint32_t)(r * H * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(creg[0]), __float_as_uint(creg[1 % NT])}, cst_rs, poff, 0, 0);
            } else {                               // unit by unit (lanes that straddle H, other tile counts)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bool uv = u0 + t < H && ao != OOB;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tga[t][r]), act_rs, uv ? ao + (uint32_t)((r * H + t) * 4) : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(creg[t]), cst_rs, uv ? poff + 4u * t : OOB, 0, 0);
                }
            }
            aoff += astep;
        }
#endif
        if (!DEFER) store_prev();                  // (this step's output)
#ifdef NIR_PT_TRACE
        PT_T(tr_t2); tr_b += tr_t2 - tr_t1;
#endif
        lds_barrier();
    }
    if (DEFER) store_prev();
#ifdef NIR_PT_TRACE
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && g_pt_trace_dev) {
        unsigned long long tn; PT_T(tn);
        unsigned long long* o = g_pt_trace_dev + wave * 8;
        o[0] = tr_a; o[1] = tr_b; o[2] = tr_c; o[3] = tn - tr_first; o[4] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)); o[5] = tmax; o[6] = tr_l; o[7] = tr_m;
    }
#endif
    // zero the padded steps of this direction's half: one wave per (sequence, step) row, coalesced (ragged batches are the normal case)
    for (int s_ = 0; s_ < nvalid; ++s_) {
        if (H1) {
            _Float16* orow = reinterpret_cast<_Float16*>(p.out) + (m0 + s_) * T * OW + (int64_t)dir * H;
            for (int t2 = lens_s[s_] + wave; t2 < T; t2 += NW)
                for (int col = lane; col < H; col += 64) orow[(int64_t)t2 * OW + col] = (_Float16)0.f;
            continue;
        }
        float* orow = p.out + (m0 + s_) * T * OW + (int64_t)dir * H;
        for (int t2 = lens_s[s_] + wave; t2 < T; t2 += NW)
            for (int col = lane; col < H; col += 64) orow[(int64_t)t2 * OW + col] = 0.f;
    }
}

template <int KB, int NT, int NW = 16, bool H1 = false, bool TR = false>
static int launch_pt_h2(const LstmPtArgs& p, hipStream_t st) {
    static const std::string pname = "lstm16_pt_h2_kernel<" + std::to_string(KB) + "," + std::to_string(NT) + (NW == 16 ? "" : "," + std::to_string(NW)) + (TR ? ",false,true" : H1 ? ",true" : "") + ">";
    const size_t lds = (size_t)(4 * 16 * (32 * KB + 8)) * 2 + 2 * 16 * 4 + (size_t)16 * (p.T + 3) * 4;
    ProfScope ps(prof_shape_name(pname.c_str(), (long long)p.M, p.T, p.H), st);
#ifdef NIR_PT_TRACE
    { unsigned long long* d = g_debug_buf; (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_pt_trace_dev), &d, sizeof(d), 0, hipMemcpyHostToDevice, st); }
#endif
    hipLaunchKernelGGL((lstm16_pt_h2_kernel<KB, NT, NW, H1, TR>), dim3((unsigned)((p.M + 15) / 16), (unsigned)p.ND), dim3(64 * NW), lds, st, p);
    NIR_CHECK_LAUNCH("nir_bilstm_folded_fwd[f16x2]");
    return 0;
}

template <int G, int NT>
static int launch_pt(const LstmPtArgs& p, hipStream_t st) {
    static const std::string pname = "lstm16_pt_kernel<" + std::to_string(G) + "," + std::to_string(NT) + ">";
    const size_t lds = (size_t)(2 * 16 * (16 * G + 4)) * 4 + 16 * 4 + (size_t)16 * p.T * 4;
    ProfScope ps(prof_shape_name(pname.c_str(), (long long)p.M, p.T, p.H), st);
    hipLaunchKernelGGL((lstm16_pt_kernel<G, NT>), dim3((unsigned)((p.M + 15) / 16), (unsigned)p.ND), dim3(1024), lds, st, p);
    NIR_CHECK_LAUNCH("nir_bilstm_folded_fwd[f32]");
    return 0;
}
static int cu_count() {
    static const int ncu = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }();
    return ncu;
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 recurrence, H in (96, 128], as 8 waves x 4 tiles in 128 VGPRs: TWO workgroups per CU.  A single term of W_hh is 64 registers per
// lane, so unlike the fp32-parity kernel two workgroups fit a CU's register file -- and this recurrence needs exactly that: its step is a
// latency chain (barrier -> LDS reads -> 16 MFMAs per wave -> gate math -> LDS write -> barrier) with the matrix pipe 19 % busy; a second
// workgroup's chain runs in the other one's bubbles.  Same step structure as lstm16_pt_bf16_kernel (gate rows two steps ahead as the MFMA C
// operand, persistent over sequence tiles, fp16 output rows copied from the LDS h buffer), consecutive-unit lane mapping of
// lstm16_pt_h2_kernel (8-byte LDS writes, 16-byte fp32 stores), step-ordered id lists, merged-fraction cell.
// ---------------------------------------------------------------------------------------------------------------------
template <bool O16>
__global__ __launch_bounds__(512, 4) void lstm16_pt_bf16w8_kernel(LstmPtArgs p) {
    constexpr int KB = 4, NT = 4, NW = 8, NTH = 64 * NW, SEQ = 16, KP = 32 * KB, ZLD = KP + 8;
    constexpr uint32_t OOB = 0x7FFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* z = reinterpret_cast<unsigned short*>(smem);       // [2][SEQ][ZLD] fp16
    f16x8* wl = reinterpret_cast<f16x8*>(z + 2 * SEQ * ZLD);            // [8 waves][4 tiles][64 lanes]: the W fragments of the LAST k-block (see below)
    int* lens_s = reinterpret_cast<int*>(wl + NW * NT * 64);
    int* ids_s = lens_s + SEQ;                                          // [SEQ][TP]: the id sequence s consumes at step k
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sq = lane & 15, kq = lane >> 4;
    const int dir = blockIdx.y;
    const int H = p.H, T = p.T, H4 = 4 * H, TP = p.T + 4;
    const int OW = p.ND * H;
    const int64_t GW = (int64_t)p.ND * H4;
    const int u0 = NT * (4 * wave + kq);
    const bool full = u0 + NT <= H;

    // W_hh of k-blocks 0..2 lives in registers (48 VGPRs), the fragments of k-block 3 in LDS (32 KB per workgroup, one ds_read_b128 per tile
    // and step): with all four in registers the loop spills at 128 VGPRs, and 128 is what two workgroups per CU allow
    f16x8 wreg[NT][KB - 1];
    bool wbad = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int unit_a = NT * (4 * wave + (sq >> 2)) + t, gate_a = sq & 3;
        const bool av = unit_a < H;
        const float* wr = p.whh + ((int64_t)dir * H4 + (int64_t)gate_a * H + (av ? unit_a : 0)) * H;
        const bool vec_ok = (H % 8) == 0 && ((reinterpret_cast<uintptr_t>(wr) & 15) == 0);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            float wv[8];
            const int k0 = 32 * kb + 8 * kq;
            if (vec_ok) {
                const int kc = k0 + 8 <= H ? k0 : 0;
                const float4 a = *reinterpret_cast<const float4*>(wr + kc), b = *reinterpret_cast<const float4*>(wr + kc + 4);
                const float m = (av && k0 + 8 <= H) ? 1.f : 0.f;
                wv[0] = a.x * m; wv[1] = a.y * m; wv[2] = a.z * m; wv[3] = a.w * m; wv[4] = b.x * m; wv[5] = b.y * m; wv[6] = b.z * m; wv[7] = b.w * m;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = wr[k0 + j < H ? k0 + j : H - 1] * ((av && k0 + j < H) ? 1.f : 0.f);
            }
            f16x8 wf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                wf[j] = (_Float16)wv[j];
                wbad |= !(fabsf(wv[j]) < 65504.0f);
            }
            if (kb < KB - 1) wreg[t][kb < KB - 1 ? kb : 0] = wf;
            else wl[(wave * NT + t) * 64 + lane] = wf;
        }
    }
    if (wbad && p.err) atomicOr(p.err, 2);
    const f16x8* wlp = wl + wave * NT * 64 + lane;           // (read back by the lane that wrote it: no barrier needed)
    const unsigned short* pth = reinterpret_cast<const unsigned short*>(p.pt) + (int64_t)dir * H4;
    // the lane's four units are 32 contiguous bytes of a folded bf16 row (H % 4 == 0: a lane is all real or all past H -- those re-read unit 0)
    const unsigned short* pb = pth + 4 * (full ? u0 : 0);
    const uint32_t gw = (uint32_t)GW;

    for (int64_t mt = blockIdx.x; mt * SEQ < p.M; mt += gridDim.x) {
        __syncthreads();                                 // the previous tile's last readers of lens_s / ids_s / z are done
        const int64_t m0 = mt * SEQ;
        const int nvalid = (int)min((int64_t)SEQ, p.M - m0);
        if (tid < SEQ) {
            int l = 0;
            if (tid < nvalid) {
                l = p.lens ? (int)p.lens[m0 + tid] : T;
                l = l < 0 ? 0 : (l > T ? T : l);
            }
            lens_s[tid] = l;
        }
        __syncthreads();
        {
            bool bad = false;
            for (int e = tid; e < SEQ * TP; e += NTH) {
                const int s_ = e / TP, k = e - s_ * TP;
                int64_t id = 0;
                if (s_ < nvalid) {
                    const int l = lens_s[s_];
                    int kk = k < l - 1 ? k : l - 1;
                    kk = kk < 0 ? 0 : kk;
                    int t_ = dir == 0 ? kk : l - 1 - kk;
                    t_ = t_ < 0 ? 0 : t_;
                    id = p.ids[(m0 + s_) * T + t_];
                    if (k < T) {
                        const int64_t raw = p.ids[(m0 + s_) * T + k];
                        bad |= raw < 0 || raw >= p.V;
                    }
                }
                if (id < 0 || id >= p.V) id = 0;
                ids_s[e] = (int)id;
            }
            if (bad && p.err) atomicOr(p.err, 1);
        }
        for (int e = tid; e < SEQ * ZLD; e += NTH) reinterpret_cast<unsigned*>(z)[e] = 0u;      // both buffers
        __syncthreads();
        int tmax = 0;
#pragma unroll
        for (int s2 = 0; s2 < SEQ; ++s2) tmax = max(tmax, lens_s[s2]);
        const int mylen = lens_s[sq];
        float creg[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) creg[t] = 0.f;
        const __amdgpu_buffer_rsrc_t out_rs =
            O16 ? __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<_Float16*>(p.out) + m0 * T * OW, 0, (int)((uint32_t)nvalid * T * OW * 2u), 0x00020000)
                : __builtin_amdgcn_make_buffer_rsrc(p.out + m0 * T * OW, 0, (int)((uint32_t)nvalid * T * OW * 4u), 0x00020000);
        // O16: wave w copies the rows of sequences 2w and 2w+1 of h_{t-1} (lane = two consecutive units) out of the LDS buffer the MFMAs of
        // step t read: 256 contiguous bytes per row
        int wlen[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) wlen[r] = __builtin_amdgcn_readfirstlane(lens_s[2 * wave + r]);
        auto copy_out = [&](int s_, const unsigned short* zsrc) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int sr = 2 * wave + r;
                const bool on = s_ >= 0 && s_ < wlen[r] && 2 * lane < H;
                const int tt_ = dir == 0 ? s_ : wlen[r] - 1 - s_;
                const uint32_t v = *reinterpret_cast<const uint32_t*>(zsrc + sr * ZLD + 2 * lane);
                __builtin_amdgcn_raw_buffer_store_b32(v, out_rs, on ? (uint32_t)((sr * T + tt_) * OW + dir * H + 2 * lane) * 2u : OOB, 0, 0);
            }
        };
        const int* idp = ids_s + sq * TP;
        auto load_g = [&](int id, uint4 (&dst)[2]) {
            const uint4* row = reinterpret_cast<const uint4*>(pb + (uint64_t)(uint32_t)id * gw);
            dst[0] = row[0];
            dst[1] = row[1];
        };
        uint4 ga[2];                                     // rows of the NEXT step (one step ahead: with two workgroups per CU the other one covers a late row)
        load_g(idp[0], ga);
        int id_n = idp[1];
        float hprev[NT] = {0.f, 0.f, 0.f, 0.f};
        uint32_t poff = OOB;
        uint32_t soff = (uint32_t)(((sq * T + (dir == 0 ? 0 : mylen - 1)) * OW + dir * H + u0) * 4);
        const uint32_t sstep = (uint32_t)(dir == 0 ? OW * 4 : -(OW * 4));
        auto store_prev = [&]() {                        // fp32 output of the previous step, in front of this step's row requests
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(hprev[0]), __float_as_uint(hprev[1]), __float_as_uint(hprev[2]), __float_as_uint(hprev[3])},
                                                   out_rs, full ? poff : OOB, 0, 0);
            if (!full) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hprev[t]), out_rs, (u0 + t < H && poff != OOB) ? poff + 4u * t : OOB, 0, 0);
            }
        };
        for (int step = 0; step < tmax; ++step) {
            const unsigned short* zc = z + (step & 1) * SEQ * ZLD;
            unsigned short* zn = z + ((step + 1) & 1) * SEQ * ZLD;
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint32_t lo = (t & 1) ? ga[t >> 1].z : ga[t >> 1].x, hi = (t & 1) ? ga[t >> 1].w : ga[t >> 1].y;
                acc[t] = (f32x4){__uint_as_float(lo << 16), __uint_as_float(lo & 0xFFFF0000u), __uint_as_float(hi << 16), __uint_as_float(hi & 0xFFFF0000u)};
            }
            const bool live = step < mylen;
            const unsigned short* zr = zc + sq * ZLD + 8 * kq;
            if (O16) copy_out(step - 1, zc);
            else store_prev();
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const f16x8 hb = *reinterpret_cast<const f16x8*>(zr + 32 * kb);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb < KB - 1 ? wreg[t][kb < KB - 1 ? kb : 0] : wlp[t * 64], hb, acc[t], 0, 0, 0);
                if (kb == 0) {
                    load_g(id_n, ga);
                    id_n = idp[step + 2];
                }
                __builtin_amdgcn_sched_barrier(0);       // one k-block of h fragments live at a time (128 VGPRs; the other waves of the SIMD cover the LDS latency)
            }
            float hn[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                lstm_cell_v(acc[t], creg[t], hn[t]);
                __builtin_amdgcn_sched_barrier(0);       // tile by tile: the temporaries of four interleaved cells do not fit 128 VGPRs
            }
            if (full) {
                *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(zn) + sq * ZLD + u0) = (f16x4){(_Float16)hn[0], (_Float16)hn[1], (_Float16)hn[2], (_Float16)hn[3]};
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (u0 + t < H) reinterpret_cast<_Float16*>(zn)[sq * ZLD + u0 + t] = (_Float16)hn[t];
            }
            if (!O16) {
#pragma unroll
                for (int t = 0; t < NT; ++t) hprev[t] = hn[t];
                poff = live ? soff : OOB;
                soff += sstep;
            }
            lds_barrier();
        }
        if (O16) { if (tmax > 0) copy_out(tmax - 1, z + (tmax & 1) * SEQ * ZLD); }
        else store_prev();
        for (int s_ = 0; s_ < nvalid; ++s_) {
            const int64_t ob = (m0 + s_) * T * OW + (int64_t)dir * H;
            for (int t2 = lens_s[s_] + wave; t2 < T; t2 += NW)
                for (int col = lane; col < H; col += 64) {
                    if (O16) reinterpret_cast<_Float16*>(p.out)[ob + (int64_t)t2 * OW + col] = (_Float16)0.f;
                    else p.out[ob + (int64_t)t2 * OW + col] = 0.f;
                }
        }
    }
}

static int launch_pt_bf16w8(const LstmPtArgs& p, hipStream_t st) {
    const size_t lds = (size_t)(2 * 16 * (32 * 4 + 8)) * 2 + (size_t)8 * 4 * 64 * 16 + 16 * 4 + (size_t)16 * (p.T + 4) * 4;
    ProfScope ps(prof_shape_name("lstm16_pt_bf16w8_kernel", (long long)p.M, p.T, p.H), st);
    const int64_t tiles = (p.M + 15) / 16;
    const int64_t cap = std::max(1, (2 * cu_count() + p.ND - 1) / p.ND);    // two 512-thread workgroups per CU
    if (p.out_f16) hipLaunchKernelGGL((lstm16_pt_bf16w8_kernel<true>), dim3((unsigned)std::min(tiles, cap), (unsigned)p.ND), dim3(512), lds, st, p);
    else hipLaunchKernelGGL((lstm16_pt_bf16w8_kernel<false>), dim3((unsigned)std::min(tiles, cap), (unsigned)p.ND), dim3(512), lds, st, p);
    NIR_CHECK_LAUNCH("nir_bilstm_folded_fwd[bf16, 8 waves]");
    return 0;
}

template <int KB, int NT>
static int launch_pt_bf16(const LstmPtArgs& p, hipStream_t st) {
    static const std::string pname = "lstm16_pt_bf16_kernel<" + std::to_string(KB) + "," + std::to_string(NT) + ">";
    const size_t lds = (size_t)(2 * 16 * (32 * KB + 8)) * 2 + 16 * 4 + (size_t)16 * p.T * 4;
    ProfScope ps(prof_shape_name(pname.c_str(), (long long)p.M, p.T, p.H), st);
    const int64_t tiles = (p.M + 15) / 16;
    const int64_t cap = std::max(1, (cu_count() + p.ND - 1) / p.ND);       // one 1024-thread workgroup per CU
    if (p.out_f16) hipLaunchKernelGGL((lstm16_pt_bf16_kernel<KB, NT, true>), dim3((unsigned)std::min(tiles, cap), (unsigned)p.ND), dim3(1024), lds, st, p);
    else hipLaunchKernelGGL((lstm16_pt_bf16_kernel<KB, NT, false>), dim3((unsigned)std::min(tiles, cap), (unsigned)p.ND), dim3(1024), lds, st, p);
    NIR_CHECK_LAUNCH("nir_bilstm_folded_fwd[bf16]");
    return 0;
}

// W_hh as the two fp16 terms of the split, in the lane order of lstm16_pt_h2_kernel<4,4,8>: [ND][8 waves][4 tiles][4 k-blocks][2 terms][64 lanes][8]
__global__ __launch_bounds__(64) void lstm_whh_frag_kernel(const float* __restrict__ whh, int H, _Float16* __restrict__ out, int* __restrict__ err) {
    constexpr int NT = 4, KB = 4, NW = 8;
    const int lane = threadIdx.x, wave = blockIdx.x, dir = blockIdx.y;
    const int sq = lane & 15, kq = lane >> 4;
    bool bad = false;
    for (int t = 0; t < NT; ++t) {
        const int unit_a = NT * (4 * wave + (sq >> 2)) + t, gate_a = sq & 3;
        const float* wr = whh + ((int64_t)dir * 4 * H + (int64_t)gate_a * H + unit_a) * H;
        for (int kb = 0; kb < KB; ++kb) {
            _Float16* o1 = out + ((((size_t)(dir * NW + wave) * NT + t) * KB + kb) * 2 * 64 + lane) * 8;
            _Float16* o2 = o1 + 64 * 8;
            for (int j = 0; j < 8; ++j) {
                const float w = wr[32 * kb + 8 * kq + j];
                const _Float16 hi = (_Float16)w;
                o1[j] = hi;
                o2[j] = (_Float16)((w - (float)hi) * 2048.0f);
                bad |= !(fabsf(w) < 32768.0f);
            }
        }
    }
    if (bad && err) atomicOr(err, 2);
}

// true when launch_bilstm_folded(.., out_f16 = 2) is served: the dispatch below ends in lstm16_pt_h2_kernel<4,4,8> with every lane `full`
bool bilstm_folded_split_out_ok(int pt_dtype, int H, int T) {
    (void)T;
    return pt_dtype == NIR_DTYPE_F32 && H == 128 && !tun(g_tun.exact_f32) && tun(g_tun.lstm_w16) != 1 && (tun(g_tun.lstm_w16) < 3 || tun(g_tun.lstm_w16) == 6);
}

int launch_bilstm_folded(const void* pt, int pt_dtype, const int64_t* ids, const int64_t* lens, const float* whh, float* out,
                         int* err, int64_t M, int64_t V, int T, int H, int ND, hipStream_t st, int out_f16, const void* whh_frag) {
    NIR_REQUIRE(pt && ids && whh && out, "bilstm_folded: null pointer");
    NIR_REQUIRE(M >= 0 && V > 0 && T > 0 && (ND == 1 || ND == 2), "bilstm_folded: bad dims");
    NIR_REQUIRE(H >= 4 && H <= 128, "bilstm_folded: hidden size %d per direction unsupported (4..128)", H);
    NIR_REQUIRE(T <= 512, "bilstm_folded: sequence length %d > 512 unsupported", T);
    NIR_REQUIRE((int64_t)16 * T * ND * H * 4 < 0x7FFFFFF0LL, "bilstm_folded: T*H too large for 32-bit tile offsets");
    NIR_REQUIRE(pt_dtype == NIR_DTYPE_F32 || pt_dtype == NIR_DTYPE_BF16, "bilstm_folded: unknown table dtype %d", pt_dtype);
    if (M == 0) return 0;
    NIR_REQUIRE(out_f16 != 1 || (pt_dtype == NIR_DTYPE_BF16 && H > 64 && H % 2 == 0), "bilstm_folded: fp16 output needs the bf16 table and an even H > 64");
    NIR_REQUIRE((out_f16 != 2 && out_f16 != 3) || bilstm_folded_split_out_ok(pt_dtype, H, T), "bilstm_folded: split-term / one-term output is produced by lstm16_pt_h2_kernel<4,4,8> only (f32 table, H = 128)");
    LstmPtArgs p{pt, ids, lens, whh, out, err, M, V, T, H, ND, out_f16, whh_frag, nullptr, nullptr};
    if (pt_dtype == NIR_DTYPE_BF16) {
        const int KB = (H + 31) / 32;
        if (H <= 64) return KB == 1 ? launch_pt_bf16<1, 1>(p, st) : launch_pt_bf16<2, 1>(p, st);
        if (KB == 3) return launch_pt_bf16<3, 2>(p, st);
        // H in (96, 128]: two 8-wave workgroups per CU (tunable lstm_w16 = 1: the 16-wave form, one workgroup per CU)
        return tun(g_tun.lstm_w16) == 1 ? launch_pt_bf16<4, 2>(p, st) : launch_pt_bf16w8(p, st);
    }
    if (!tun(g_tun.exact_f32) && H >= 32) {      // fp32-accurate two-term fp16 split on the fp16 matrix cores
        const int KB = (H + 31) / 32;
        if (H <= 64) return KB == 1 ? launch_pt_h2<1, 1>(p, st) : launch_pt_h2<2, 1>(p, st);
        // 4 waves x 5 tiles, two workgroups per CU: pays when the workgroups in flight (this launch x the caller's batches in flight)
        // outnumber the CUs twice over -- 32x50 MatchTensor with 4 batches in flight: 8.8 M -> 10.4 M pairs/s; at 32x10 the dispatcher
        // pairs workgroups on a CU while other CUs idle and the same kernel LOSES 18 % (tunable lstm_s: 1 = never, 2 = always)
        if (KB == 3 && H <= 80) {
            const int ncu = cu_count();
            const int64_t wgs = ((p.M + 15) / 16) * p.ND * (int64_t)std::max(1, batches_in_flight(st));
            const int sel = tun(g_tun.lstm_s);
            if (sel == 2 || (sel != 1 && wgs >= 2 * (int64_t)ncu)) return launch_pt_h2<3, 5, 4>(p, st);
        }
        if (KB == 3) return launch_pt_h2<3, 2>(p, st);
        // H in (96, 128]: 8 waves x 4 tiles with the in-wave pipeline (tunable lstm_w16 = 1: the 16-wave x 2-tile form)
        if (tun(g_tun.lstm_w16) == 1) return launch_pt_h2<4, 2>(p, st);
        // (two sequence groups per workgroup sharing the W registers, skewed wave roles, output stores on one early wave: measured in rounds 3-4,
        // all tied or lost -- DESIGN.md section 10; sources archived under tools/variants/, not built)
        if (p.out_f16 == 3) return launch_pt_h2<4, 4, 8, true>(p, st);      // the opt-in one-term-h tier (H = 128, checked by the caller)
        return launch_pt_h2<4, 4, 8>(p, st);
    }
    const int G = (H + 15) / 16;
    if (H <= 64) {
        switch (G) {
            case 1: return launch_pt<1, 1>(p, st);
            case 2: return launch_pt<2, 1>(p, st);
            case 3: return launch_pt<3, 1>(p, st);
            default: return launch_pt<4, 1>(p, st);
        }
    }
    switch (G) {
        case 5: return launch_pt<5, 2>(p, st);
        case 6: return launch_pt<6, 2>(p, st);
        case 7: return launch_pt<7, 2>(p, st);
        default: return launch_pt<8, 2>(p, st);
    }
}

// Train-mode forward on the split-fp16 recurrence (64 < H <= 128 per direction): gates_perm [M*T][ND][H][4] = x W_ih^T + b in the folded gate order
// (nir_lstm_perm_weights + one GEMM), row_ids = 0 .. M*T-1.  3 fp16 MFMAs per k-block instead of the 32 fp32 ones of lstm_mfma16_gin_kernel.
int launch_lstm_train_split(const float* gates_perm, const int64_t* row_ids, const int64_t* lens, const float* whh, float* out, float* act, float* cst,
                            int* err, int64_t M, int T, int H, int ND, hipStream_t st) {
    NIR_REQUIRE(gates_perm && row_ids && whh && out && act && cst, "lstm_train_fwd_split: null pointer");
    NIR_REQUIRE(M >= 0 && T > 0 && T <= 512 && (ND == 1 || ND == 2) && H > 64 && H <= 128, "lstm_train_fwd_split: bad dims (64 < H <= 128 per direction, T <= 512)");
    NIR_REQUIRE((int64_t)16 * T * ND * H * 16 < 0x7FFFFFF0LL, "lstm_train_fwd_split: T too large for 32-bit tile offsets");
    if (M == 0) return 0;
    LstmPtArgs p{gates_perm, row_ids, lens, whh, out, err, M, M * (int64_t)T, T, H, ND, 0, nullptr, act, cst};
    if (H <= 96) return launch_pt_h2<3, 2, 16, false, true>(p, st);        // (MatchTensor's H = 70)
    return launch_pt_h2<4, 4, 8, false, true>(p, st);
}

// per-direction nn.LSTM parameters -> the folded gate order: wperm[(dir*H + unit)*4 + gate][:] = w_ih[dir][gate*H + unit][:], bperm = b_ih + b_hh
__global__ void lstm_perm_weights_kernel(const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ bi0, const float* __restrict__ bh0,
                                         const float* __restrict__ bi1, const float* __restrict__ bh1, int H, int ND, int E, float* __restrict__ wperm,
                                         float* __restrict__ bperm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows = (int64_t)ND * 4 * H;
    if (i >= rows * E) return;
    const int64_t ro = i / E;
    const int k = (int)(i % E);
    const int gate = (int)(ro & 3), unit = (int)((ro >> 2) % H), dir = (int)((ro >> 2) / H);
    const int64_t ri = (int64_t)gate * H + unit;
    wperm[i] = (dir ? w1 : w0)[ri * E + k];
    if (k == 0) bperm[ro] = (dir ? bi1 : bi0)[ri] + (dir ? bh1 : bh0)[ri];
}

}  // namespace nir

extern "C" int nir_lstm_perm_weights(const float* w_ih_fwd, const float* b_ih_fwd, const float* b_hh_fwd, const float* w_ih_rev, const float* b_ih_rev,
                                     const float* b_hh_rev, int H, int ndir, int E, float* wperm, float* bperm, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(w_ih_fwd && b_ih_fwd && b_hh_fwd && wperm && bperm && (ndir == 1 || (ndir == 2 && w_ih_rev && b_ih_rev && b_hh_rev)), "lstm_perm_weights: null pointer");
    NIR_REQUIRE(H > 0 && E > 0, "lstm_perm_weights: bad dims");
    const int64_t n = (int64_t)ndir * 4 * H * E;
    hipLaunchKernelGGL(lstm_perm_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ih_fwd, w_ih_rev, b_ih_fwd, b_hh_fwd,
                       b_ih_rev, b_hh_rev, H, ndir, E, wperm, bperm);
    NIR_CHECK_LAUNCH("lstm_perm_weights_kernel");
    return 0;
}
extern "C" int nir_lstm_train_fwd_split(const float* gates_perm, const int64_t* row_ids, const int64_t* lengths, const float* w_hh, float* out, float* act,
                                        float* cst, int* err_flag, int64_t M, int T, int H, int ndir, nir_stream_t stream) {
    return nir::launch_lstm_train_split(gates_perm, row_ids, lengths, w_hh, out, act, cst, err_flag, M, T, H, ndir, (hipStream_t)stream);
}

extern "C" size_t nir_lstm_fold_table_bytes(int64_t V, int H, int ndir, int dtype) {
    if (V <= 0 || H <= 0 || ndir <= 0) return 0;
    return (size_t)V * ndir * 4 * H * (dtype == NIR_DTYPE_BF16 ? 2 : 4);
}

extern "C" size_t nir_lstm_fold_table_workspace_bytes(int64_t V, int E, int H, int ndir, int dtype) {
    if (V <= 0 || H <= 0 || ndir <= 0 || E <= 0) return 0;
    size_t b = nir::align_up((size_t)ndir * 4 * H * E * 4, 256) + nir::align_up((size_t)ndir * 4 * H * 4, 256);
    if (dtype == NIR_DTYPE_BF16) b += nir::align_up((size_t)std::min<int64_t>(V, 65536) * ndir * 4 * H * 4, 256);
    return b;
}

extern "C" int nir_lstm_fold_table(const float* table, int64_t V, int E, const float* w_ih, const float* b_ih, const float* b_hh,
                                   int H, int ndir, void* folded, int dtype, void* workspace, size_t workspace_bytes,
                                   nir_stream_t stream) {
    using namespace nir;
    hipStream_t st = (hipStream_t)stream;
    NIR_REQUIRE(table && w_ih && b_ih && b_hh && folded, "lstm_fold_table: null pointer");
    NIR_REQUIRE(V > 0 && E > 0 && H > 0 && (ndir == 1 || ndir == 2), "lstm_fold_table: bad dims");
    NIR_REQUIRE(dtype == NIR_DTYPE_F32 || dtype == NIR_DTYPE_BF16, "lstm_fold_table: unknown dtype %d", dtype);
    const size_t need = nir_lstm_fold_table_workspace_bytes(V, E, H, ndir, dtype);
    if (!workspace || workspace_bytes < need) {
        set_error("lstm_fold_table: workspace too small (%zu < %zu)", workspace_bytes, need);
        return NIR_ERR_WORKSPACE;
    }
    const int GW = ndir * 4 * H;
    Workspace a(workspace, workspace_bytes);
    float* wperm = a.take<float>((size_t)GW * E);
    float* bperm = a.take<float>((size_t)GW);
    {
        const int64_t n = (int64_t)GW * E;
        hipLaunchKernelGGL(fold_permute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_ih, b_ih, b_hh, H, ndir, E, wperm, bperm);
        NIR_CHECK_LAUNCH("fold_permute_kernel");
    }
    if (dtype == NIR_DTYPE_F32)
        return launch_linear(table, E, nullptr, nullptr, 0, 0, 0, wperm, E, bperm, nullptr, (float*)folded, GW, V, GW, E, NIR_ACT_NONE, st);
    float* tmp = a.take<float>((size_t)std::min<int64_t>(V, 65536) * GW);
    for (int64_t v0 = 0; v0 < V; v0 += 65536) {     // fp32 product, rounded once to bf16
        const int64_t nv = std::min<int64_t>(65536, V - v0);
        NIR_PROPAGATE(launch_linear(table + v0 * E, E, nullptr, nullptr, 0, 0, 0, wperm, E, bperm, nullptr, tmp, GW, nv, GW, E, NIR_ACT_NONE, st));
        const int64_t n = nv * GW;
        hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, st, tmp,
                           (unsigned short*)folded + v0 * GW, n);
        NIR_CHECK_LAUNCH("f32_to_bf16_kernel");
    }
    return 0;
}

extern "C" int nir_bilstm_folded_fwd(const void* folded, int dtype, const int64_t* ids, const int64_t* lengths, const float* w_hh,
                                     float* out, int* err_flag, int64_t M, int64_t V, int T, int H, int ndir, nir_stream_t stream) {
    return nir::launch_bilstm_folded(folded, dtype, ids, lengths, w_hh, out, err_flag, M, V, T, H, ndir, (hipStream_t)stream, 0, nullptr);
}

extern "C" size_t nir_lstm_whh_frag_bytes(int H, int ndir) {
    return (H == 128 && (ndir == 1 || ndir == 2)) ? (size_t)ndir * 8 * 4 * 4 * 2 * 64 * 8 * sizeof(_Float16) : 0;
}
extern "C" int nir_lstm_pack_whh_frag(const float* w_hh, int H, int ndir, void* frag, int* err_flag, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(w_hh && frag, "lstm_pack_whh_frag: null pointer");
    NIR_REQUIRE(nir_lstm_whh_frag_bytes(H, ndir) != 0, "lstm_pack_whh_frag: only the H = 128 recurrence takes pre-split fragments (got H = %d)", H);
    hipLaunchKernelGGL(lstm_whh_frag_kernel, dim3(8, (unsigned)ndir), dim3(64), 0, (hipStream_t)stream, w_hh, H, (_Float16*)frag, err_flag);
    NIR_CHECK_LAUNCH("nir_lstm_pack_whh_frag");
    return 0;
}
