// micro-benchmark: would TWO rays per lane pay for the tricubic march?  Per wave and sample the kernel issues ~449 VALU
// instructions and 64 broadcast ds_read_b128 at 5 waves per SIMD; with two rays per lane sharing every texel read it
// would issue ~874 VALU instructions per 64 reads at 3 waves per SIMD (register budget).  Time per 64 ray-samples:
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NV>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    __shared__ float4 tile[4][80];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 80; i += 64) tile[wave][i] = make_float4(seed + i, seed - i, 0.5f * i, 1.f);
    __syncthreads();
    float f[16];
    for (int j = 0; j < 16; j++) f[j] = seed + lane + j;
    float wgt = 0.25f + 1e-3f * lane;
    const float decay = 0.999f + 1e-9f * seed * lane;
    constexpr int extra = NV - 4 * 64 * (NV > 600 ? 2 : 1);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 64; g++) {
            const float4 t = tile[wave][g];
            f[0] = fmaf(t.x, decay, f[0]); f[1] = fmaf(t.y, decay, f[1]); f[2] = fmaf(t.z, decay, f[2]); f[3] = fmaf(t.w, decay, f[3]);
            if (NV > 600) { f[8] = fmaf(t.x, wgt, f[8]); f[9] = fmaf(t.y, wgt, f[9]); f[10] = fmaf(t.z, wgt, f[10]); f[11] = fmaf(t.w, wgt, f[11]); }
            const int per = extra / 64 + (g < extra % 64 ? 1 : 0);
#pragma unroll
            for (int q = 0; q < per; q++) f[4 + ((q + g) & 3) + ((q & 1) ? 8 : 0)] = fmaf(f[4 + ((q + g) & 3) + ((q & 1) ? 8 : 0)], decay, wgt);
            if ((g & 3) == 3) asm volatile("" : "+v"(f[0]) : : "memory");
        }
        wgt = fmaf(wgt, 0.9999f, 1e-6f);
    }
    float r = 0;
    for (int j = 0; j < 16; j++) r += f[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NV>
static double run(int waves, float *d, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 3000;
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<NV>, dim3(256 * waves), dim3(256), 0, 0, d, iters, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms * 1e6 / ((double)waves * iters);         // ns per wave-sample per SIMD
}

int main() {
    float *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 3; w <= 6; w++) printf("one ray  per lane, 449 VALU + 64 reads, %d waves/SIMD: %.1f ns per 64 ray-samples\n", w, run<449>(w, d, e0, e1));
    for (int w = 2; w <= 4; w++) printf("two rays per lane, 874 VALU + 64 reads, %d waves/SIMD: %.1f ns per 64 ray-samples\n", w, run<874>(w, d, e0, e1) / 2);
    return 0;
}
