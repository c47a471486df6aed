// Micro-benchmark: cost of a stream-ordered chain of small kernels, plain launches vs one hipGraph replay.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_touch(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const int N = 2000, n = 200 * 512;
    float* d; CK(hipMalloc(&d, sizeof(float) * n)); CK(hipMemset(d, 0, sizeof(float) * n));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, s));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_touch, dim3(200), dim3(512), 0, s, d, n);
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("plain launches: %.3f us per kernel\n", 1e3 * ms / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_touch, dim3(200), dim3(512), 0, s, d, n);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("graph replay : %.3f us per kernel\n", 1e3 * ms / N);
    }
    return 0;
}
