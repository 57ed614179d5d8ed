// micro-benchmark: v_fma_f32 vs v_pk_fma_f32 issue rate on gfx950, 1..8 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int PACKED>
__global__ void k(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float w = 0.999f, c = 0.001f;
    const v2f pw = {w, w}, pc = {c, c};
    for (int i = 0; i < iters; i++) {
        if (PACKED) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                p0 = __builtin_elementwise_fma(p0, pw, pc); p1 = __builtin_elementwise_fma(p1, pw, pc);
                p2 = __builtin_elementwise_fma(p2, pw, pc); p3 = __builtin_elementwise_fma(p3, pw, pc);
                p4 = __builtin_elementwise_fma(p4, pw, pc); p5 = __builtin_elementwise_fma(p5, pw, pc);
                p6 = __builtin_elementwise_fma(p6, pw, pc); p7 = __builtin_elementwise_fma(p7, pw, pc);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0 = fmaf(a0, w, c); a1 = fmaf(a1, w, c); a2 = fmaf(a2, w, c); a3 = fmaf(a3, w, c);
                a4 = fmaf(a4, w, c); a5 = fmaf(a5, w, c); a6 = fmaf(a6, w, c); a7 = fmaf(a7, w, c);
            }
        }
    }
    float r = PACKED ? (p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y)
                     : (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    float *d; hipMalloc(&d, 256 * 8 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {          // waves per SIMD: blocks of 256 thr = 1 wave/SIMD; grid = 256 CUs * wps
        for (int packed = 0; packed < 2; packed++) {
            dim3 grid(256 * wps), block(256);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (packed) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters, 1.0f);
                else hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters, 1.0f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_wave = (double)iters * 64;             // 64 FMA instructions per iteration (8 regs x 8 unroll)
            const double fma_per_lane = instr_per_wave * (packed ? 2 : 1);
            const double cyc = ms * 1e-3 * 2.4e9;
            printf("waves/SIMD %d  %s  %.3f ms  cycles/instr/SIMD %.2f  TFLOP/s %.1f\n", wps, packed ? "v_pk_fma_f32" : "v_fma_f32   ", ms,
                   cyc / (instr_per_wave * wps), 2.0 * fma_per_lane * 256.0 * wps * 256 / (ms * 1e-3) * 1e-12);
        }
    }
    return 0;
}
