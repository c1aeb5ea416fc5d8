// What does one barrier interval of a recurrence step cost on gfx950?  512-thread workgroups, one per CU (100 KB of LDS requested), 256 of
// them; per iteration: [LDS reads of a 8 KB B operand] [48 v_mfma_f32_16x16x32_f16, no two adjacent on one accumulator] [LDS write] s_barrier.
//   hipcc --offload-arch=gfx950 -O3 tools/barrier_probe.hip -o /tmp/barrier_probe && /tmp/barrier_probe
// mode 0: barrier only;  1: MFMAs + barrier (all 8 waves);  2: LDS reads + MFMAs + LDS write + barrier (all waves);
// mode 3 / 4: as 2 / 1 but only ONE wave of each SIMD does the work, its partner only meets the barrier;
// mode 5: as 2, and the partner wave issues 200 v_fma_f32 (four chains) per interval;  6: as 5 with two intervals per iteration and the roles swapped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(unsigned long long* out, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* z = reinterpret_cast<_Float16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sq = lane & 15, kq = lane >> 4;
    constexpr int ZLD = 136;
    for (int e = tid; e < 4 * 16 * ZLD; e += 512) z[e] = (_Float16)(0.001f * (e & 63));
    __syncthreads();
    f16x8 w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) w[i][j] = (_Float16)(0.01f * (i + j + lane));
    f32x4 acc[4], acx[4];
    float v[4] = {1.f + lane, 2.f, 3.f, 4.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) { acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acx[t] = acc[t]; }
    // which waves work: the hardware places waves 0-3 and 4-7 of a 512-thread workgroup on SIMD 0-3 each (checked in tools/valu_rate.hip)
    const bool first = wave < 4;
    auto work = [&](int it) {
        const _Float16* zr = z + ((it & 1) * 2) * 16 * ZLD + sq * ZLD + 8 * kq;
        f16x8 h1[4], h2[4];
        if (MODE == 1 || MODE == 4) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) { h1[kb] = w[kb]; h2[kb] = w[4 + kb]; }
        } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                h1[kb] = *reinterpret_cast<const f16x8*>(zr + 32 * kb);
                h2[kb] = *reinterpret_cast<const f16x8*>(zr + 16 * ZLD + 32 * kb);
            }
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[4 * t + kb], h1[kb], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[4 * t + kb], h2[kb], acx[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[(4 * t + kb + 5) & 15], h1[kb], acx[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(acc[t]), "+v"(acx[t]));
        if (MODE == 2 || MODE == 3 || MODE >= 5) {
            _Float16* zn = z + (((it + 1) & 1) * 2) * 16 * ZLD;
            const f16x4 a = {(_Float16)acc[0][0], (_Float16)acc[1][0], (_Float16)acc[2][0], (_Float16)acc[3][0]};
            const f16x4 r = {(_Float16)acx[0][0], (_Float16)acx[1][0], (_Float16)acx[2][0], (_Float16)acx[3][0]};
            *reinterpret_cast<f16x4*>(zn + sq * ZLD + 4 * (4 * wave + kq)) = a;
            *reinterpret_cast<f16x4*>(zn + 16 * ZLD + sq * ZLD + 4 * (4 * wave + kq)) = r;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t] *= 1e-6f; acx[t] *= 1e-6f; }
    };
    auto valu = [&]() {
#pragma unroll
        for (int r = 0; r < 50; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(0.5f));
    };
    bar();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
        } else if (MODE == 1 || MODE == 2) {
            work(it);
        } else if (MODE == 3 || MODE == 4) {
            if (first) work(it);
        } else if (MODE == 5) {
            if (first) work(it); else valu();
        } else if (MODE == 6) {
            if (first) work(it); else valu();
            bar();
            if (!first) work(it); else valu();
        }
        bar();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && lane == 0) out[wave] = t1 - t0;
    float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
    for (int t = 0; t < 4; ++t) s += acc[t][0] + acx[t][1];
    if (s == 123.456f) sink[tid] = s;
}

template <int MODE>
static void run(const char* what, int iters) {
    unsigned long long* out;
    float* sink;
    (void)hipMalloc(&out, 64);
    (void)hipMalloc(&sink, 4096);
    (void)hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 100 * 1024, 0, out, sink, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 100 * 1024, 0, out, sink, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    const double per = MODE == 6 ? 2.0 : 1.0;
    printf("mode %d  %-62s %7.1f ns / interval   %7.1f ticks / interval (wave 0), %.0f (wave 4)   [%.2f ticks per ns]\n", MODE, what,
           ms * 1e6 / iters / per, (double)h[0] / iters / per, (double)h[4] / iters / per, (double)h[0] / (ms * 1e6));
    (void)hipFree(out); (void)hipFree(sink);
}

int main() {
    const int it = 4000;
    run<0>("barrier only", it);
    run<1>("48 MFMAs, all 8 waves", it);
    run<4>("48 MFMAs, one wave per SIMD", it);
    run<2>("8 LDS reads + 48 MFMAs + LDS write, all 8 waves", it);
    run<3>("8 LDS reads + 48 MFMAs + LDS write, one wave per SIMD", it);
    run<5>("as above, the partner wave issues 200 v_fma", it);
    run<6>("as above, roles swapped every interval", it);
    return 0;
}
