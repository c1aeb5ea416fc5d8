// Shared device/host helpers for libneuroir_hip (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/neuroir_hip.h"

namespace nir {

void set_error(const char* fmt, ...);

#define NIR_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            nir::set_error(__VA_ARGS__);       \
            return NIR_ERR_BAD_ARG;            \
        }                                      \
    } while (0)

// Check the launch just enqueued; returns the hipError_t (>0) on failure.
#define NIR_CHECK_LAUNCH(name)                                                      \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            nir::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                        \
        }                                                                           \
    } while (0)

#define NIR_PROPAGATE(expr)        \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != 0) return rc__; \
    } while (0)

// Optional per-kernel timing (nir_profile_enable / nir_profile_report): HIP events recorded on the launch stream
// around every kernel launch while enabled.  Never enabled during graph capture or timed regions.
struct ProfScope {
    void* rec;
    hipStream_t st;
    ProfScope(const char* name, hipStream_t stream);
    ~ProfScope();
};

// Interned "name[M=..,N=..,K=..]" label (recurrences: M sequences, N = T steps, K = H units) while profiling is enabled --
// bench.py prices every launch by its own shape; returns `base` unchanged otherwise.  The returned pointer stays valid for the life of the process.
const char* prof_shape_name(const char* base, long long M, long long N, long long K);

// Fork/join helper: independent kernel chains of one C-ABI call (e.g. the query side and the document side of a
// ranker) run concurrently on a lazily created per-device side stream.  Event record / wait are capturable, so the
// whole call still records into a hipGraph.
struct ForkJoin {
    hipStream_t main, side;
    hipEvent_t ev_fork, ev_join;
    bool ok, open;
    explicit ForkJoin(hipStream_t main_stream);
    ~ForkJoin() { if (open) join(); }   // error returns between fork() and join() still re-join the side stream (an unjoined side
                                        // stream would invalidate an in-progress hipGraph capture)
    void fork();   // side waits for everything enqueued on main so far
    void join();   // main waits for everything enqueued on side so far
};

// Optional device debug buffer (nir_debug_set_buffer): kernels that support it drop s_memtime stamps there.
extern unsigned long long* g_debug_buf;
// Scheduling hint of the caller: how many independent batches it keeps in flight next to the calls it enqueues on THIS stream
// (nir_set_stream_batches_in_flight; no entry = 1: optimise the latency of the single call).  > 1 -> favour chip throughput.  There is no
// process-wide setting: callers that share the library cannot steer each other.  Lock-free while no hint exists, else one
// mutex-protected map lookup per C-ABI call.
int batches_in_flight(hipStream_t st);

// Tuning / debug switches.  Initialised ONCE from the environment when the library is loaded (NIR_NO_FORK, NIR_LSTM_VALU,
// NIR_LSTM_MFMA16, NIR_LSTM_MFMA_S, NIR_LSTM_S, NIR_NO_SKINNY, NIR_NO_GEMM16, NIR_ESM_WAVE_ROWS, NIR_DEBUG, NIR_EXACT_F32);
// nir_debug_set_tunable changes one at run time (tests, profilers).  Hot entry points only do relaxed atomic loads.
struct Tunables {
    std::atomic<int> no_fork, lstm_valu, lstm_mfma16, lstm_mfma_s, lstm_s, lstm_w16, no_skinny, no_gemm16, esm_wave_rows, debug, exact_f32, duet_unfused, attn_unfused, attn_unfused_pipe, duet_rows64, attn_fp32_rows, attn_io_prio, lstm_step_ug, lstm_step_nb, nofold_old, gemm3_ks, wgrad_no_lds, wgrad_lds_tiles, wgrad_min_rows, lstm_bwd_w8, cl_poll_limit;
};
extern Tunables g_tun;
inline int tun(const std::atomic<int>& a) { return a.load(std::memory_order_relaxed); }

constexpr int WAVE = 64;

// Wave (64-lane) all-reduce without LDS traffic: two quad_perm butterflies, row_half_mirror, row_mirror (DPP, VALU
// rate) leave every 16-lane row fully reduced; the four rows are combined through v_readlane (SGPRs).  The
// __shfl_xor version lowers to 6 ds_bpermute round trips (~600+ cycles per value when latency-bound).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

// sigma / tanh built on the hardware exp2 (v_exp_f32, ~1 ulp) -- absolute error ~1e-7, far inside the
// 1e-4 score tolerance while ~4x cheaper than the ocml tanhf in the 64..290-step recurrences.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }  // v_rcp_f32, 1 ulp
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
    float ax = fabsf(x);
    float e = __expf(-2.0f * ax);           // in (0,1]
    float t = (1.0f - e) * fast_rcp(1.0f + e);
    return copysignf(t, x);
}

// bf16 helpers (round-to-nearest-even on the raw bits; NaN stays NaN)
__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// 3 x bf16 split of an fp32 value: x = b0 + b1 + b2 up to 2^-27 |x| (each term takes the next 8 mantissa bits; the
// residuals x - b0, x - b0 - b1 are exact in fp32).  Products of two such triples keep the six leading cross terms
// (b0b0, b0b1, b1b0, b0b2, b1b1, b2b0): every bf16 x bf16 product is exact in fp32, so six bf16 MFMAs with fp32 accumulation
// reproduce an fp32 dot product to fp32 accuracy at 6/16 of the fp32-MFMA cost.
__device__ __forceinline__ void split3(float x, unsigned short& b0, unsigned short& b1, unsigned short& b2) {
    b0 = f2bf(x);
    const float r1 = x - bf2f(b0);
    b1 = f2bf(r1);
    const float r2 = r1 - bf2f(b1);
    b2 = f2bf(r2);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (global loads AND stores
// share that counter on gfx950), which would expose a full HBM round trip per recurrence step.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == NIR_ACT_TANH) return fast_tanh(v);
    if (act == NIR_ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over the caller-provided workspace.
struct Workspace {
    char* base;
    size_t cap, off;
    Workspace(void* p, size_t bytes) : base((char*)p), cap(bytes), off(0) {}
    template <typename T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* r = (T*)(base ? base + off : nullptr);
        off += n * sizeof(T);
        return r;
    }
    bool ok() const { return off <= cap; }
};

// One LSTM time step for up to two independent chains that share (B, I, H) (csrc/cars_session.hip: lstm_step_kernel): the CARS session
// encoders (cars.py:306-380) and the greedy decoders.  Shared by cars_session.hip and cars_decode.hip.
// Arg-max key of the fused projection + arg-max (csrc/cars_decode.hip): larger value wins, the SMALLER index on ties; 0 = no candidate yet.
__device__ __forceinline__ unsigned long long argmax_key(float x, int idx) {
    uint32_t u = __float_as_uint(x);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)idx);
}
constexpr int ARGMAX_KEY_BUCKETS = 8;      // atomics of one decode row are spread over this many words (235 workgroups -> ~30 per word)
__device__ __forceinline__ int64_t argmax_key_index(unsigned long long key) {
    const uint32_t idx = 0xFFFFFFFFu - (uint32_t)key;
    return (key == 0ull || idx == 0x7FFFFFFFu) ? 0 : (int64_t)idx;
}
struct LstmStepArgs {
    const float* x[2];        // input rows: row b at x + (xid ? xid[b] : b) * xstride   (xid: embedding gather by token id)
    const int64_t* xid[2];
    int64_t xstride[2];
    const float* wih[2];      // [4H, I]
    const float* whh[2];      // [4H, H]
    const float* bih[2];
    const float* bhh[2];
    const float* hprev[2];    // [B,H] or NULL (zero state: the recurrent product is skipped)
    const float* cprev[2];    // [B,H] or NULL
    float* hnext[2];          // [B,H]
    float* cnext[2];
    int chain0;               // blockIdx.y + chain0 = chain id
    int B, I, H;
    // gx != NULL: the input side of the gates -- x W_ih^T + b_ih + b_hh, [row b at gx + b * gxstride][4H] -- was computed for ALL time steps by
    // one batched GEMM in front of the loop (the inputs of the session chains do not depend on the recurrence); the step then walks W_hh only
    const float* gx[2] = {nullptr, nullptr};
    int64_t gxstride = 0;
    // (fp16-term step only) gxid != NULL: row b of the gate rows is gx + gxid[b] * gxstride (a folded per-token gate table gathered by the
    // previous step's token ids: the greedy decoders); gx_unit_major: a gate row is [unit][i,f,g,o] (nir_lstm_fold_table's order), not [gate][unit]
    const int64_t* gxid[2] = {nullptr, nullptr};
    int gx_unit_major = 0;
    // (round 6, chain 0 only) gxkey != NULL instead of gxid: the row index is decoded from the arg-max KEY the fused projection kernel left for row b
    // (argmax_key: value order in the high word, 0xFFFFFFFF - vocabulary index in the low word), mapped through gxmap (target -> source ids; NULL:
    // identity) and clamped to [0, gxV) like argmax_finish_kernel did -- the greedy decoder then needs no finish launch between its steps
    const unsigned long long* gxkey = nullptr;      // [B][ARGMAX_KEY_BUCKETS]: the row's key is the max over its buckets
    const int64_t* gxmap = nullptr;
    int64_t gxV = 0;
    // whh_frag != NULL (H % 32 == 0): W_hh pre-split into two fp16 terms in MFMA-fragment order (nir_lstm_step_pack_whh_frag) and the previous
    // state ALSO kept as fp16 term pairs (h16prev / h16next: [B][H/8][2 terms][8]) -- the recurrent product then runs as three
    // v_mfma_f32_16x16x32_f16 per 32-wide k-block (fp32-class, like the folded recurrences) instead of eight v_mfma_f32_16x16x4_f32
    const void* whh_frag[2] = {nullptr, nullptr};
    const _Float16* h16prev[2] = {nullptr, nullptr};
    _Float16* h16next[2] = {nullptr, nullptr};
};
int launch_lstm_step(const LstmStepArgs& a, int nchains, hipStream_t st);

}  // namespace nir
