// micro-benchmark: v_fmac_f32 throughput of ONE..FOUR waves per SIMD as a function of the number of independent accumulators
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float w, float t) {
    float a[ILP];
#pragma unroll
    for (int j = 0; j < ILP; j++) a[j] = threadIdx.x + j;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 64 / ILP; u++) {
#pragma unroll
            for (int j = 0; j < ILP; j++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[j]) : "v"(w), "v"(t));
        }
    }
    float r = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) r += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int ILP> float run(float *d, int wps, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<ILP>, dim3(256 * wps), dim3(256), 0, 0, d, iters, 0.5f, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main() {
    float *d; hipMalloc(&d, 256 * 8 * 1024 * 4);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps++) {
        const float m1 = run<1>(d, wps, iters), m2 = run<2>(d, wps, iters), m4 = run<4>(d, wps, iters), m8 = run<8>(d, wps, iters), m16 = run<16>(d, wps, iters);
        const double n = (double)iters * 64 * wps;
        printf("waves/SIMD %d  ns per instr per SIMD:  ILP1 %.3f  ILP2 %.3f  ILP4 %.3f  ILP8 %.3f  ILP16 %.3f\n", wps, m1 * 1e6 / n, m2 * 1e6 / n,
               m4 * 1e6 / n, m8 * 1e6 / n, m16 * 1e6 / n);
    }
    return 0;
}
