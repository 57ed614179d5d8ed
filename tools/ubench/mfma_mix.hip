// micro-benchmark: can the tricubic sample's 64-tap contraction move onto the matrix pipe?
//
// A coherent wave's sample is out[64 rays][4 channels] = W[64 rays][64 taps] . T[64 taps][4 channels].
// v_mfma_f32_4x4x1_16B_f32 does exactly 16 blocks x (4 rays x 4 channels) per tap.  This file measures, at the
// march kernel's occupancy (5 waves per SIMD, every CU busy), the time per "sample" of instruction mixes:
//   valu      497 v_fma_f32 + 64 broadcast ds_read_b128           (today's kernel, per wave and sample)
//   mix       64 MFMA 4x4x1 + NV v_fma_f32 + 16 ds_read_b128      (contraction on the matrix pipe)
//   mfma      64 MFMA only           valu257   257 v_fma_f32 only
// and checks that a chain of 4x4x1 MFMAs is bitwise the fmaf chain (denormals and signed zeros included)
// and which lane / register holds which element.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NM, int NV, int NL>
__global__ __launch_bounds__(256, 5) void mix(float *out, int iters, float seed) {
    __shared__ float4 tile[4][80];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 80; i += 64) tile[wave][i] = make_float4(seed + i, seed - i, 0.5f * i, 1.f);
    __syncthreads();
    const float4 *rd = &tile[wave][(lane & 3) * 17];            // channel-major image, 68-dword channel stride
    v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float f[8];
    for (int j = 0; j < 8; j++) f[j] = seed + lane + j;
    float wgt = 0.25f + 1e-3f * lane;
    const float decay = 0.999f + 1e-9f * seed * lane;
    for (int it = 0; it < iters; it++) {
        constexpr int G = NM ? NM / 4 : (NL ? NL : 16);         // groups per sample
#pragma unroll
        for (int g = 0; g < G; g++) {
            float4 t = make_float4(1.f, 2.f, 3.f, 4.f);
            if (NL && g < NL) t = NM ? rd[g % 16] : tile[wave][g];     // channel-major 4-tap read | broadcast texel read
            if (NM) {
                acc[g & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wgt, t.x, acc[g & 3], 0, 0, 0);
                acc[g & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wgt, t.y, acc[g & 3], 0, 0, 0);
                acc[g & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wgt, t.z, acc[g & 3], 0, 0, 0);
                acc[g & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wgt, t.w, acc[g & 3], 0, 0, 0);
            } else if (NL) {
                f[0] = fmaf(t.x, 1e-9f, f[0]); f[1] = fmaf(t.y, 1e-9f, f[1]);
                f[2] = fmaf(t.z, 1e-9f, f[2]); f[3] = fmaf(t.w, 1e-9f, f[3]);
            }
            constexpr int extra = NV - ((NM == 0 && NL) ? 4 * NL : 0);
            const int per = extra / G + (g < extra % G ? 1 : 0);
#pragma unroll
            for (int q = 0; q < per; q++) f[(q + g) & 7] = fmaf(f[(q + g) & 7], decay, wgt);
            if ((g & 3) == 3) asm volatile("" : "+v"(f[0]) : : "memory");     // keep the reads from being hoisted (spills)
        }
        wgt = fmaf(wgt, 0.9999f, 1e-6f);
    }
    float r = 0;
    for (int j = 0; j < 8; j++) r += f[j];
    for (int j = 0; j < 4; j++) r += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// numerics + layout: D = chain over k of A_k (one value per lane) x B_k (one value per lane)
__global__ void chain(const float *a, const float *b, int K, float *d) {
    const int lane = threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    for (int k = 0; k < K; k++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k * 64 + lane], b[k * 64 + lane], acc, 0, 0, 0);
    d[lane * 4 + 0] = acc.x; d[lane * 4 + 1] = acc.y; d[lane * 4 + 2] = acc.z; d[lane * 4 + 3] = acc.w;
}

template <int NM, int NV, int NL>
static void run(const char *name, float *d, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 4000;
    dim3 grid(256 * 5), block(256);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mix<NM, NV, NL>), grid, block, 0, 0, d, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // per SIMD: 5 waves x iters samples
    const double ns_per_wave_sample = ms * 1e6 / (5.0 * iters);
    printf("%-10s  MFMA %3d  VALU %3d  LDS %2d   %.3f ms   %.1f ns per wave-sample per SIMD (%.0f cycles at 2.4 GHz)\n", name, NM, NV,
           NL, ms, ns_per_wave_sample, ns_per_wave_sample * 2.4);
}

int main() {
    float *d; hipMalloc(&d, 256 * 5 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    run<0, 497, 64>("valu", d, e0, e1);
    run<64, 257, 16>("mix257", d, e0, e1);
    run<64, 192, 16>("mix192", d, e0, e1);
    run<64, 128, 16>("mix128", d, e0, e1);
    run<64, 0, 16>("mfma+lds", d, e0, e1);
    run<64, 0, 0>("mfma", d, e0, e1);
    run<0, 257, 0>("valu257", d, e0, e1);
    run<0, 497, 0>("valu497", d, e0, e1);
    run<0, 336, 64>("valu336+l", d, e0, e1);

    // ---- numerics and layout ----
    const int K = 64;
    std::vector<float> a(K * 64), b(K * 64), got(256);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    for (int i = 0; i < K * 64; i++) { a[i] = rnd(); b[i] = rnd() * 1e-3f; }
    // denormal, signed-zero and huge/tiny cases in a few (k, lane) slots
    for (int l = 0; l < 64; l++) { a[0 * 64 + l] = -0.f; b[1 * 64 + l] = (l & 1) ? 1e-41f : -1e-42f; a[2 * 64 + l] = 3e-39f; }
    for (int l = 0; l < 64; l++) if ((l & 3) == 3) for (int k = 0; k < K; k++) b[k * 64 + l] = (k & 1) ? 1e-40f : -2e-41f;   // denormal-only column
    float *da, *db, *dd;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dd, 256 * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, da, db, K, dd);
    hipMemcpy(got.data(), dd, 256 * 4, hipMemcpyDeviceToHost);
    // expected layout: lane l = 4*blk + j, register i  <->  D[blk][row i][col j] = chain_k fma(A_k[4*blk+i], B_k[4*blk+j], .)
    int bad = 0, bad_mag = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 4; i++) {
            const int blk = l >> 2, j = l & 3;
            float acc = 0.f;
            for (int k = 0; k < K; k++) acc = fmaf(a[k * 64 + 4 * blk + i], b[k * 64 + 4 * blk + j], acc);
            const float g = got[l * 4 + i];
            if (memcmp(&g, &acc, 4) != 0) {
                bad++;
                if (!(fabsf(g - acc) <= 1e-30f)) bad_mag++;
                if (bad <= 8) printf("  lane %2d reg %d: mfma %.9g (0x%08x)  fmaf chain %.9g (0x%08x)\n", l, i, g, *(unsigned *)&g, acc, *(unsigned *)&acc);
            }
        }
    printf("numerics: %d of 256 outputs differ bitwise from the fmaf chain in the assumed layout (%d by more than 1e-30)\n", bad, bad_mag);
    return 0;
}
