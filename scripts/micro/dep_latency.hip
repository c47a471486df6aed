// Single-wave issue model on gfx950 (the serial sampler wave's world): cycles per instruction of
//   (a) a chain of DEPENDENT v_fma_f32, (b) the same number of INDEPENDENT v_fma_f32 (4 chains interleaved),
//   (c) the dense walk's chain  fma -> sub -> v_readlane -> fmac  (one wave, 64 steps, lane l broadcast at step l).
// hipcc --offload-arch=gfx950 -O3 dep_latency.hip -o dep_latency && ./dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, long long* cyc, int n)
{
    float a = out[threadIdx.x], b = 1.0001f, c = 0.5f;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    }
    long long t1 = clock64();
    float x0 = a, x1 = a + 1.f, x2 = a + 2.f, x3 = a + 3.f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(b), "v"(c));
        }
    }
    long long t2 = clock64();
    // the walk's chain: an = fma(k1, x, k0); D = ao - an; Db = readlane(D, l); x = fma(Db, g, x)
    float x = a, k1 = 0.25f, k0 = 0.125f, ao = 2.f, g = 0.001f * (float)threadIdx.x;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int l = 0; l < 16; ++l) {
            const float an = fmaf(k1, x, k0);
            const float D = ao - an;
            const float Db = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(D), l));
            x = fmaf(Db, g, x);
        }
    }
    long long t3 = clock64();
    out[threadIdx.x] = a + x0 + x1 + x2 + x3 + x;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}
int main()
{
    float* o; long long* c;
    hipMalloc(&o, 64 * sizeof(float)); hipMalloc(&c, 3 * sizeof(long long));
    hipMemset(o, 0, 64 * sizeof(float));
    const int n = 4096;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, n);
    long long h[3];
    hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    printf("dependent v_fma_f32:   %.2f cycles per instruction\n", (double)h[0] / (16.0 * n));
    printf("independent v_fma_f32: %.2f cycles per instruction (4 chains interleaved)\n", (double)h[1] / (16.0 * n));
    printf("walk chain (fma, sub, v_readlane, fma): %.2f cycles per step\n", (double)h[2] / (16.0 * n));
    return 0;
}
