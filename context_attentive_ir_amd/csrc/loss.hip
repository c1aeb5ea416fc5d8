// Reductions over the N candidates of a query: predict-time softmax (models/ranker.py:258,
// models/multitask.py:279), BCE-with-logits (models/ranker.py:55-69, multitask/cars.py:603) and the
// list-wise softmax NLL (models/ranker.py:79-89).  One wave per row; N <= 64 sits in one wavefront, larger N
// strides.  The two losses finish with a single-workgroup deterministic tree (no float atomics), so the
// scalar is reproducible run to run.
#include "common.hpp"

namespace nir {

__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* in, float* out, int64_t rows, int n) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* x = in + r * n;
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, x[i]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += expf(x[i] - mx);
    s = wave_sum(s);
    for (int i = lane; i < n; i += 64) out[r * n + i] = expf(x[i] - mx) / s;
}

// softmax_rows + the deferred error word (core.hip: flag_publish_kernel) in one launch: lane 0 of block 0 copies a non-zero device flag into
// the pinned host word.  Every kernel that can set the flag precedes this one in stream order.
__global__ __launch_bounds__(256) void softmax_rows_publish_kernel(const float* in, float* out, int64_t rows, int n, const int* dev_flag, int* host_flag) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int v = __builtin_nontemporal_load(dev_flag);
        if (v != 0) __hip_atomic_store(host_flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* x = in + r * n;
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, x[i]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += expf(x[i] - mx);
    s = wave_sum(s);
    for (int i = lane; i < n; i += 64) out[r * n + i] = expf(x[i] - mx) / s;
}

// Softmax over the candidates of a query straight from the rank-major all-gather buffer [world][B][per] (what
// all_gather_into_tensor leaves on every rank): candidate n of query b lives at ((n / per) * B + b) * per + n % per.
// Writes the query-major [B][N] probabilities (N <= world*per drops the padded tail); optionally also the raw scores.
__global__ __launch_bounds__(256) void softmax_gathered_kernel(const float* in, float* probs, float* scores, int world,
                                                               int64_t B, int per, int N) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    auto at = [&](int n) { return in[((int64_t)(n / per) * B + b) * per + (n % per)]; };
    float mx = -INFINITY;
    for (int i = lane; i < N; i += 64) mx = fmaxf(mx, at(i));
    mx = wave_max(mx);
    float s = 0.f;
    for (int i = lane; i < N; i += 64) s += expf(at(i) - mx);
    s = wave_sum(s);
    for (int i = lane; i < N; i += 64) {
        const float v = at(i);
        probs[b * N + i] = expf(v - mx) / s;
        if (scores) scores[b * N + i] = v;
    }
}

// mode 0: BCE with logits, per-element  max(x,0) - x*y + log1p(exp(-|x|));  mode 1: -(log_softmax(x)*y).sum()
template <int MODE>
__global__ __launch_bounds__(1024) void rank_loss_kernel(const float* sc, const float* lab, int64_t rows, int n, float* loss) {
    __shared__ float part[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int64_t r = wave; r < rows; r += 16) {
        const float* x = sc + r * n;
        const float* y = lab + r * n;
        if (MODE == 0) {
            float s = 0.f;
            for (int i = lane; i < n; i += 64) {
                float v = x[i];
                s += fmaxf(v, 0.f) - v * y[i] + log1pf(expf(-fabsf(v)));
            }
            acc += s;
        } else {
            float mx = -INFINITY;
            for (int i = lane; i < n; i += 64) mx = fmaxf(mx, x[i]);
            mx = wave_max(mx);
            float se = 0.f;
            for (int i = lane; i < n; i += 64) se += expf(x[i] - mx);
            float lse = mx + logf(wave_sum(se));
            float s = 0.f;
            for (int i = lane; i < n; i += 64) s -= (x[i] - lse) * y[i];
            acc += s;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += part[w];
        loss[0] = t / (MODE == 0 ? (float)(rows * n) : (float)rows);
    }
}

}  // namespace nir

extern "C" int nir_softmax_rows(const float* in, float* out, int64_t rows, int n, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(in && out && rows >= 0 && n > 0, "softmax_rows: bad args");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, out, rows, n);
    NIR_CHECK_LAUNCH("nir_softmax_rows");
    return 0;
}

namespace nir { int mapped_host_pointer(const void* host, void** out); }
extern "C" int nir_softmax_rows_publish(const float* in, float* out, int64_t rows, int n, const int* dev_flag, int* host_flag, nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(in && out && rows > 0 && n > 0 && dev_flag && host_flag, "softmax_rows_publish: bad args");
    void* d = nullptr;
    NIR_PROPAGATE(mapped_host_pointer(host_flag, &d));
    hipLaunchKernelGGL(softmax_rows_publish_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, out, rows, n, dev_flag, (int*)d);
    NIR_CHECK_LAUNCH("nir_softmax_rows_publish");
    return 0;
}

extern "C" int nir_softmax_gathered(const float* gathered, float* probs, float* scores, int world, int64_t B, int per, int N,
                                    nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(gathered && probs && world > 0 && B >= 0 && per > 0 && N > 0 && N <= world * per, "softmax_gathered: bad args");
    if (B == 0) return 0;
    hipLaunchKernelGGL(softmax_gathered_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, gathered, probs,
                       scores, world, B, per, N);
    NIR_CHECK_LAUNCH("nir_softmax_gathered");
    return 0;
}

extern "C" int nir_rank_loss_bce(const float* scores, const float* labels, int64_t rows, int n, float* loss,
                                 nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(scores && labels && loss && rows > 0 && n > 0, "rank_loss_bce: bad args");
    hipLaunchKernelGGL(rank_loss_kernel<0>, dim3(1), dim3(1024), 0, (hipStream_t)stream, scores, labels, rows, n, loss);
    NIR_CHECK_LAUNCH("nir_rank_loss_bce");
    return 0;
}

extern "C" int nir_rank_loss_softmax_nll(const float* scores, const float* labels, int64_t rows, int n, float* loss,
                                         nir_stream_t stream) {
    using namespace nir;
    NIR_REQUIRE(scores && labels && loss && rows > 0 && n > 0, "rank_loss_softmax_nll: bad args");
    hipLaunchKernelGGL(rank_loss_kernel<1>, dim3(1), dim3(1024), 0, (hipStream_t)stream, scores, labels, rows, n, loss);
    NIR_CHECK_LAUNCH("nir_rank_loss_softmax_nll");
    return 0;
}
