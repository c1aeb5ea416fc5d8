// Probe the operand/result lane layout of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products, K=1).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
    int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
int main() {
    float ha[64], hb[64], hd[256];
    // a[l] = 1000*(l/4) + 10*(l%4) + 1 ; b[l] = 100*(l/4)... choose primes so the product identifies (la, lb)
    for (int l = 0; l < 64; ++l) { ha[l] = (float)(l + 1); hb[l] = (float)(1 << (l % 4)) * (1 + (l / 4) * 0.001f); }
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    // for each output lane/reg find which (la, lb) produced it
    for (int l = 0; l < 64; l += 1) {
        if (!(l < 8 || l >= 60)) continue;
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) {
            int fa = -1, fb = -1;
            for (int x = 0; x < 64 && fa < 0; ++x)
                for (int y = 0; y < 64; ++y)
                    if (fabsf(hd[l * 4 + r] - ha[x] * hb[y]) < 1e-4f * fabsf(hd[l * 4 + r])) { fa = x; fb = y; break; }
            printf("  reg%d = a[lane %2d] * b[lane %2d]", r, fa, fb);
        }
        printf("\n");
    }
    return 0;
}
