// Issue-rate probe for the packed-fp16 gate-math question: cycles per wave64 instruction of the fp32 / fp16 transcendentals and
// the packed fp16 / fp32 arithmetic on gfx950, one wave per SIMD, eight independent chains.
//   hipcc --offload-arch=gfx950 -O2 tools/f16_rate.hip -o /tmp/f16_rate && /tmp/f16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE>
__global__ void probe(unsigned long long* out, float* sink, int iters) {
    float a[8];
    f32x2 p[8];
    unsigned int h[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0f + 0.001f * (threadIdx.x + i); p[i] = (f32x2){a[i], a[i] + 0.5f}; h[i] = 0x3C003C00u + i; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(1.0001f));
                if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 3) asm volatile("v_exp_f16 %0, %0" : "+v"(h[i]));
                if (MODE == 4) asm volatile("v_rcp_f16 %0, %0" : "+v"(h[i]));
                if (MODE == 5) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(h[i]) : "v"(h[(i + 1) & 7]));
                if (MODE == 6) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(h[i]) : "v"(h[(i + 1) & 7]));
                if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (MODE == 8) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(1.0001f));
                if (MODE == 9) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 10) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + (float)h[i];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) out[MODE] = t1 - t0;
}
int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 64);
    hipMemset(out, 0, 64 * 8);
    const int iters = 2000;
    probe<0><<<1, 256>>>(out, sink, iters); probe<1><<<1, 256>>>(out, sink, iters); probe<2><<<1, 256>>>(out, sink, iters);
    probe<3><<<1, 256>>>(out, sink, iters); probe<4><<<1, 256>>>(out, sink, iters); probe<5><<<1, 256>>>(out, sink, iters);
    probe<6><<<1, 256>>>(out, sink, iters); probe<7><<<1, 256>>>(out, sink, iters); probe<8><<<1, 256>>>(out, sink, iters);
    probe<9><<<1, 256>>>(out, sink, iters); probe<10><<<1, 256>>>(out, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"v_mul_f32", "v_exp_f32", "v_rcp_f32", "v_exp_f16", "v_rcp_f16", "v_pk_mul_f16", "v_pk_fma_f16", "v_pk_fma_f32", "v_fma_f32", "v_cvt_f16_f32", "v_cvt_pkrtz_f16_f32"};
    for (int i = 0; i < 11; ++i) printf("%-22s %.2f cycles per wave64 instruction\n", names[i], (double)h[i] / (iters * REP));
    return 0;
}
