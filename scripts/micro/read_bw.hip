// read_bw.hip -- what a pure streaming READ of HBM reaches on this GPU (the sweep's roofline in practice).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/read_bw.hip -o scripts/micro/read_bw && scripts/micro/read_bw
// Variants: workgroups x threads, float4 loads in flight per thread (U), plain vs non-temporal loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int U, bool NT>
__global__ void k_read(const float4* __restrict__ x, size_t n4, float* __restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            if (NT) { const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(&x[i + u * stride])); v[u] = make_float4(t.x, t.y, t.z, t.w); }
            else v[u] = x[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;     // keep the loads
}

template <int U, bool NT>
static void run(const char* tag, const float4* x, size_t n4, float* out, int wgs, int threads)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_read<U, NT>), dim3(wgs), dim3(threads), 0, 0, x, n4, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_read<U, NT>), dim3(wgs), dim3(threads), 0, 0, x, n4, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s wgs=%5d thr=%4d U=%d : %7.1f GB/s\n", tag, wgs, threads, U, 3.0 * n4 * 16.0 / (ms * 1e-3) / 1e9);
}

int main()
{
    const size_t bytes = (size_t)48 << 30;                    // 48 GB: far beyond every cache
    float4* x = nullptr; float* out = nullptr;
    if (hipMalloc(&x, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 4);
    hipMemset(x, 0, bytes);
    const size_t n4 = bytes / 16;
    for (int wgs : {256, 512, 1024, 2048, 4096, 8192}) {
        run<4, false>("plain", x, n4, out, wgs, 256);
        run<8, false>("plain", x, n4, out, wgs, 256);
    }
    for (int wgs : {256, 512, 1024, 2048}) {
        run<4, false>("plain 512thr", x, n4, out, wgs, 512);
        run<8, false>("plain 512thr", x, n4, out, wgs, 512);
        run<16, false>("plain 512thr", x, n4, out, wgs, 512);
    }
    for (int wgs : {1024, 2048, 4096}) {
        run<8, true>("non-temporal", x, n4, out, wgs, 256);
        run<16, true>("non-temporal", x, n4, out, wgs, 256);
    }
    return 0;
}
