// micro-benchmark: v_fmac_f32_dpp with row_newbcast (lane k of each 16-lane row feeds the whole row) against plain
// v_fmac_f32, 5 waves per SIMD; plus a semantics check.  If the DPP form issues at the plain rate, a coherent wave can keep
// its 4x4x4 texel tile in 16 VGPRs (lane k of every row = texel k of a z-slab) and run the tricubic chain with NO LDS reads.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#define FMAC_DPP(acc, t, w, N) asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(w))

template <int DPP>
__global__ __launch_bounds__(256, 5) void rate(float *out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    float t[4] = {seed + lane, seed - lane, 0.5f * lane, 1.0f};
    float w = 0.25f + 1e-3f * lane;
    float a[8];
    for (int j = 0; j < 8; j++) a[j] = seed * j;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            if (DPP) {
                FMAC_DPP(a[0], t[0], w, 0); FMAC_DPP(a[1], t[1], w, 1); FMAC_DPP(a[2], t[2], w, 2); FMAC_DPP(a[3], t[3], w, 3);
                FMAC_DPP(a[4], t[0], w, 4); FMAC_DPP(a[5], t[1], w, 5); FMAC_DPP(a[6], t[2], w, 6); FMAC_DPP(a[7], t[3], w, 7);
                FMAC_DPP(a[0], t[0], w, 8); FMAC_DPP(a[1], t[1], w, 9); FMAC_DPP(a[2], t[2], w, 10); FMAC_DPP(a[3], t[3], w, 11);
                FMAC_DPP(a[4], t[0], w, 12); FMAC_DPP(a[5], t[1], w, 13); FMAC_DPP(a[6], t[2], w, 14); FMAC_DPP(a[7], t[3], w, 15);
            } else {
#pragma unroll
                for (int q = 0; q < 16; q++) asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[q & 7]) : "v"(t[q & 3]), "v"(w));
            }
        }
        w = fmaf(w, 0.9999f, 1e-6f);
    }
    float r = 0;
    for (int j = 0; j < 8; j++) r += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void sem(const float *t, const float *w, float *out) {
    const int lane = threadIdx.x;
    float tv = t[lane], wv = w[lane], acc = 1.0f;
    asm volatile("s_nop 4");
    FMAC_DPP(acc, tv, wv, 5);
    float m = 0.f;
    asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xf" : "=v"(m) : "v"(tv), "v"(wv));
    out[lane] = acc;
    out[64 + lane] = m;
}

int main() {
    float *d; (void)hipMalloc(&d, 256 * 5 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int dpp = 0; dpp < 2; dpp++) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            if (dpp) hipLaunchKernelGGL(rate<1>, dim3(256 * 5), dim3(256), 0, 0, d, iters, 1.0f);
            else hipLaunchKernelGGL(rate<0>, dim3(256 * 5), dim3(256), 0, 0, d, iters, 1.0f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-28s %.3f ms  %.3f ns per instruction per SIMD (5 waves)\n", dpp ? "v_fmac_f32_dpp row_newbcast" : "v_fmac_f32_e32", ms,
               ms * 1e6 / (5.0 * iters * 256));
    }
    std::vector<float> t(64), w(64), got(128);
    for (int i = 0; i < 64; i++) { t[i] = 1.0f + 0.37f * i; w[i] = 0.5f - 0.011f * i; }
    float *dt, *dw, *dout;
    (void)hipMalloc(&dt, 256); (void)hipMalloc(&dw, 256); (void)hipMalloc(&dout, 512);
    (void)hipMemcpy(dt, t.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dw, w.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, dt, dw, dout);
    (void)hipMemcpy(got.data(), dout, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const float want = fmaf(t[(l & ~15) + 5], w[l], 1.0f), wantm = t[(l & ~15) + 11] * w[l];
        if (memcmp(&want, &got[l], 4) || memcmp(&wantm, &got[64 + l], 4)) bad++;
    }
    printf("semantics: %d of 64 lanes differ from fmaf(t[row*16 + k], w[lane], acc)\n", bad);
    return 0;
}
