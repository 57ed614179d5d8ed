// micro-benchmark: does v_fmac_f32 pay for VGPR operands that share a register bank (index mod 4) on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    // v0-v3 accumulators, v4-v7 weights, v8-v11 texels
    asm volatile("v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n"
                 "v_mov_b32 v4, 0.5\n v_mov_b32 v5, 0.5\n v_mov_b32 v6, 0.5\n v_mov_b32 v7, 0.5\n"
                 "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n" ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11");
    for (int i = 0; i < iters; i++) {
        if (MODE == 0)       // all three operands in the same bank
            asm volatile(REP8("v_fmac_f32 v0, v4, v8\n v_fmac_f32 v1, v5, v9\n v_fmac_f32 v2, v6, v10\n v_fmac_f32 v3, v7, v11\n") ::: "v0","v1","v2","v3");
        else if (MODE == 1)  // three different banks
            asm volatile(REP8("v_fmac_f32 v0, v5, v10\n v_fmac_f32 v1, v6, v11\n v_fmac_f32 v2, v7, v8\n v_fmac_f32 v3, v4, v9\n") ::: "v0","v1","v2","v3");
        else if (MODE == 2)  // accumulator and texel share a bank, weight elsewhere (the tricubic chain as compiled)
            asm volatile(REP8("v_fmac_f32 v0, v5, v8\n v_fmac_f32 v1, v6, v9\n v_fmac_f32 v2, v7, v10\n v_fmac_f32 v3, v4, v11\n") ::: "v0","v1","v2","v3");
        else                 // weight and texel share a bank, accumulator elsewhere
            asm volatile(REP8("v_fmac_f32 v0, v5, v9\n v_fmac_f32 v1, v6, v10\n v_fmac_f32 v2, v7, v11\n v_fmac_f32 v3, v4, v8\n") ::: "v0","v1","v2","v3");
    }
    float r;
    asm volatile("v_add_f32 %0, v0, v1\n v_add_f32 %0, %0, v2\n v_add_f32 %0, %0, v3" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    float *d; hipMalloc(&d, 256 * 8 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char *names[4] = {"all same bank", "three banks", "acc+texel same bank", "weight+texel same bank"};
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int mode = 0; mode < 4; mode++) {
            dim3 grid(256 * wps), block(256);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, iters);
                else hipLaunchKernelGGL(k<3>, grid, block, 0, 0, d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("waves/SIMD %d  %-24s %.3f ms  ns/instr/SIMD %.3f\n", wps, names[mode], ms, ms * 1e6 / ((double)iters * 32 * wps));
        }
    return 0;
}
