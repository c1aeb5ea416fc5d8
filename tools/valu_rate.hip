// Issue-rate probe for gfx950 VALU classes: cycles per wave64 instruction of v_mul_f32, v_pk_mul_f32, v_exp_f32, v_rcp_f32 and
// v_mfma_f32_16x16x32_f16, one wave per SIMD (1 workgroup of 256 threads), independent chains.  hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define REP 64
template <int MODE>
__global__ void probe(unsigned long long* out, float* sink, int iters) {
    float a[8];
    f32x2 p[8];
    f32x4 acc[4];
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { a[i] = 1.0f + 0.001f * (threadIdx.x + i); p[i] = (f32x2){a[i], a[i] + 0.5f}; ha[i] = (_Float16)0.01f; hb[i] = (_Float16)0.02f; }
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(1.0001f));
                if (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 4) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(ha), "v"(hb));
                if (MODE == 5) {   // 1 MFMA + 3 v_mul interleaved (can the VALU issue under a running MFMA?)
                    if ((i & 3) == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[(i >> 2) & 3]) : "v"(ha), "v"(hb));
                    else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(1.0001f));
                }
                if (MODE == 6) {   // 1 MFMA + 1 v_exp interleaved
                    if ((i & 1) == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[(i >> 1) & 3]) : "v"(ha), "v"(hb));
                    else asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                }
                if (MODE == 7) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(1.0001f));
                if (MODE == 8) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a[i]));
                // MIXED waves on one SIMD (512 threads: waves 0-3 and 4-7 share the four SIMDs): waves 0-3 issue only MFMAs, waves 4-7 only VALU
                // (MODE 9: v_mul, MODE 10: v_exp) -- do the two streams overlap, or does their time add?
                if (MODE >= 11 && MODE <= 14 && it == 0 && r == 0 && i == 0) {     // static issue priorities for the mixed test
                    const bool mma = (threadIdx.x >> 6) < 4;
                    if (MODE == 11 && !mma) __builtin_amdgcn_s_setprio(3);      // VALU wave above
                    if (MODE == 12 && mma) __builtin_amdgcn_s_setprio(3);       // MFMA wave above
                    if (MODE == 13 && !mma) __builtin_amdgcn_s_setprio(1);
                    if (MODE == 14 && mma) __builtin_amdgcn_s_setprio(1);
                }
                if (MODE >= 11 && MODE <= 14) {
                    if ((threadIdx.x >> 6) < 4) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(ha), "v"(hb));
                    else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(1.0001f));
                }
                if (MODE == 15) {      // both waves of a SIMD run the in-wave interleave 1 MFMA : 3 v_mul
                    if ((i & 3) == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[(i >> 2) & 3]) : "v"(ha), "v"(hb));
                    else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(1.0001f));
                }
                if (MODE == 9 || MODE == 10) {
                    if ((threadIdx.x >> 6) < 4) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(ha), "v"(hb));
                    else if (MODE == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(1.0001f));
                    else asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                }
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (MODE >= 9 && (threadIdx.x & 63) == 0) {          // per wave: ticks and the SIMD it ran on (HW_REG_HW_ID bits [5:4])
        out[8 + 2 * (threadIdx.x >> 6)] = t1 - t0;
        out[9 + 2 * (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));
    }
}
// MIXED waves on one SIMD, role decided ONCE per wave (a branch around every single instruction would dominate the measurement): waves 0-3 issue only
// MFMAs, waves 4-7 only VALU (VK 0: v_mul, 1: v_exp); PRIO 1: VALU waves s_setprio 3, 2: MFMA waves s_setprio 3
template <int VK, int PRIO>
__global__ void mixed(unsigned long long* out, float* sink, int iters) {
    float a[8];
    f32x4 acc[4];
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { a[i] = 1.0f + 0.001f * (threadIdx.x + i); ha[i] = (_Float16)0.01f; hb[i] = (_Float16)0.02f; }
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool mma = (threadIdx.x >> 6) < 4;
    if (PRIO == 1 && !mma) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 2 && mma) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (mma) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < REP; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(ha), "v"(hb));
    } else {
        for (int it = 0; it < 3 * iters; ++it)          // (three times the instructions: both streams stay busy for about the same time)
#pragma unroll
            for (int i = 0; i < REP; ++i) {
                if (VK == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i & 7]) : "v"(1.0001f));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(a[i & 7]));
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        out[8 + 2 * (threadIdx.x >> 6)] = t1 - t0;
        out[9 + 2 * (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));
    }
}

int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 4096 * 8); hipMalloc(&sink, 4);
    const char* names[] = {"v_mul_f32", "v_pk_mul_f32", "v_exp_f32", "v_rcp_f32", "mfma16x16x32f16", "1 mfma + 3 v_mul (per 4)", "1 mfma + 1 v_exp (per 2)", "v_fma_f32", "v_cvt_f16_f32"};
    for (int waves = 1; waves <= 4; waves *= 2)
    for (int m = 0; m < 9; ++m) {
        const int iters = 200;
        auto launch = [&](int mode) {
            dim3 g(1), b(256 * waves);
            switch (mode) {
                case 0: hipLaunchKernelGGL(probe<0>, g, b, 0, 0, d, sink, iters); break;
                case 1: hipLaunchKernelGGL(probe<1>, g, b, 0, 0, d, sink, iters); break;
                case 2: hipLaunchKernelGGL(probe<2>, g, b, 0, 0, d, sink, iters); break;
                case 3: hipLaunchKernelGGL(probe<3>, g, b, 0, 0, d, sink, iters); break;
                case 4: hipLaunchKernelGGL(probe<4>, g, b, 0, 0, d, sink, iters); break;
                case 5: hipLaunchKernelGGL(probe<5>, g, b, 0, 0, d, sink, iters); break;
                case 6: hipLaunchKernelGGL(probe<6>, g, b, 0, 0, d, sink, iters); break;
                case 7: hipLaunchKernelGGL(probe<7>, g, b, 0, 0, d, sink, iters); break;
                case 8: hipLaunchKernelGGL(probe<8>, g, b, 0, 0, d, sink, iters); break;
            }
        };
        launch(m); launch(m);
        hipDeviceSynchronize();
        unsigned long long h = 0;
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%d wave(s)/SIMD  %-28s %.2f s_memtime ticks per instruction (per wave)\n", waves, names[m], (double)h / (iters * REP));
    }
    for (int m = 0; m < 5; ++m) {
        const int iters = 200;
        for (int rep = 0; rep < 2; ++rep) {
            switch (m) {
                case 0: hipLaunchKernelGGL((mixed<0, 0>), dim3(1), dim3(512), 0, 0, d, sink, iters); break;
                case 1: hipLaunchKernelGGL((mixed<1, 0>), dim3(1), dim3(512), 0, 0, d, sink, iters); break;
                case 2: hipLaunchKernelGGL((mixed<0, 1>), dim3(1), dim3(512), 0, 0, d, sink, iters); break;
                case 3: hipLaunchKernelGGL((mixed<0, 2>), dim3(1), dim3(512), 0, 0, d, sink, iters); break;
                default: hipLaunchKernelGGL(probe<15>, dim3(1), dim3(512), 0, 0, d, sink, iters); break;
            }
        }
        hipDeviceSynchronize();
        unsigned long long h[24];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const char* mn[] = {"v_mul_f32", "v_exp_f32", "v_mul_f32, VALU waves s_setprio 3", "v_mul_f32, MFMA waves s_setprio 3", "(all eight waves: in-wave 1 MFMA : 3 v_mul)"};
        printf("MIXED 2 waves/SIMD, waves 0-3 MFMA + waves 4-7 %s:\n", mn[m]);
        for (int w = 0; w < 8; ++w)
            printf("   wave %d (SIMD %llu, %s): %.2f ticks per instruction\n", w, h[9 + 2 * w], (w < 4 || m == 4) ? "mfma" : "valu",
                   (double)h[8 + 2 * w] / (iters * REP * ((w >= 4 && m < 4) ? 3 : 1)));
    }
    // whole-chip MFMA rate under sustained load (the clock the matrix pipe really runs at): 2048 workgroups x 4 waves of independent MFMAs
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 4000;
        hipLaunchKernelGGL(probe<4>, dim3(2048), dim3(256), 0, 0, d, sink, 200);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<4>, dim3(2048), dim3(256), 0, 0, d, sink, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        const double flops = 2048.0 * 4 * iters * REP * 16384.0;
        printf("sustained v_mfma_f32_16x16x32_f16, whole chip: %.1f TFLOP/s (%.2f ms); block 0: %.2f ticks per MFMA, %.3f ticks per ns\n",
               flops / (ms * 1e-3) / 1e12, ms, (double)h / (iters * (double)REP), (double)h / (ms * 1e6) * (2048.0 / 2048.0));
    }
    return 0;
}
