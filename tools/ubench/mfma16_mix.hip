// micro-benchmark: the tricubic sample of a coherent wave as 16 x v_mfma_f32_16x16x1_4B_f32.
//
//   D[ray][(c, ch)] = sum over the 16 (a, b) taps of a z-slab of (wx[a] wy[b])[ray] * T[a, b, c][ch]
//
// M = 64 rays (4 blocks x 16 rows; A = the lane's OWN weight product: lane l <-> block l/16, row l%16),
// N = 16 = 4 z-slabs x 4 channels (B: lane l <-> column l%16, the same texel dword in all four blocks),
// K = 1 per instruction, 16 instructions per sample, no wasted flops.  The z pass (4 FMAs per channel with
// the ray's own wz) runs on the VALU after an LDS transpose of D.
//
// Measures the time per wave-sample at 5 waves per SIMD with NV plain VALU instructions beside the 16
// MFMAs and the sample's LDS traffic, and checks numerics + register layout against an fmaf chain.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NM, int NV, bool LDS>
__global__ __launch_bounds__(256, 5) void mix(float *out, int iters, float seed) {
    __shared__ float tileT[4][16 * 20];             // per wave: [column j = 4c+ch][16 taps], row stride 20 dwords
    __shared__ float dT[4][64 * 17];                // per wave: D transposed back, [ray][16 values], stride 17
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 16 * 20; i += 64) tileT[wave][i] = seed + 1e-3f * i;
    __syncthreads();
    const float4 *rd = reinterpret_cast<const float4 *>(&tileT[wave][(lane & 15) * 20]);
    v16f acc;
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
    float f[8];
    for (int j = 0; j < 8; j++) f[j] = seed + lane + j;
    float wgt = 0.25f + 1e-3f * lane;
    const float decay = 0.999f + 1e-9f * seed * lane;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            float4 t = make_float4(1.f, 2.f, 3.f, 4.f);
            if (LDS) t = rd[g];
            if (NM) {
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(wgt, t.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(wgt, t.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(wgt, t.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(wgt, t.w, acc, 0, 0, 0);
            } else {
                f[0] = fmaf(t.x, 1e-9f, f[0]); f[1] = fmaf(t.y, 1e-9f, f[1]);
            }
            const int per = NV / 4 + (g < NV % 4 ? 1 : 0);
#pragma unroll
            for (int q = 0; q < per; q++) f[(q + g) & 7] = fmaf(f[(q + g) & 7], decay, wgt);
            asm volatile("" : "+v"(f[0]) : : "memory");
        }
        if (LDS && NM) {
            // D: lane = 16 g + j, register 4 b + r  <->  ray 16 b + 4 g + r, column j
            float *w = &dT[wave][(4 * (lane >> 4)) * 17 + (lane & 15)];
#pragma unroll
            for (int v = 0; v < 16; v++) w[((v >> 2) * 16 + (v & 3)) * 17] = acc[v];
            asm volatile("" : : : "memory");
            const float *r = &dT[wave][lane * 17];
#pragma unroll
            for (int v = 0; v < 16; v++) f[v & 7] = fmaf(r[v], 1e-9f, f[v & 7]);      // stands in for the z pass
#pragma unroll
            for (int j = 0; j < 16; j++) acc[j] = 0.f;
        }
        wgt = fmaf(wgt, 0.9999f, 1e-6f);
    }
    float r = 0;
    for (int j = 0; j < 8; j++) r += f[j];
    for (int j = 0; j < 16; j++) r += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void chain(const float *a, const float *b, int K, float *d) {
    const int lane = threadIdx.x;
    v16f acc;
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
    for (int k = 0; k < K; k++) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[k * 64 + lane], b[k * 64 + lane], acc, 0, 0, 0);
    for (int j = 0; j < 16; j++) d[lane * 16 + j] = acc[j];
}

template <int NM, int NV, bool LDS>
static void run(const char *name, float *d, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 4000;
    dim3 grid(256 * 5), block(256);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((mix<NM, NV, LDS>), grid, block, 0, 0, d, iters, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double ns = ms * 1e6 / (5.0 * iters);
    printf("%-12s MFMA16 %2d  VALU %3d  LDS %d   %.3f ms   %.1f ns per wave-sample per SIMD (%.0f cycles at 2.4 GHz)\n", name, NM, NV,
           (int)LDS, ms, ns, ns * 2.4);
}

int main() {
    float *d; (void)hipMalloc(&d, 256 * 5 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    run<16, 0, false>("mfma16", d, e0, e1);
    run<16, 0, true>("mfma16+lds", d, e0, e1);
    run<16, 128, true>("mix128", d, e0, e1);
    run<16, 192, true>("mix192", d, e0, e1);
    run<16, 224, true>("mix224", d, e0, e1);
    run<16, 257, true>("mix257", d, e0, e1);
    run<16, 320, true>("mix320", d, e0, e1);
    run<0, 257, false>("valu257", d, e0, e1);
    run<0, 497, false>("valu497", d, e0, e1);

    const int K = 16;
    std::vector<float> a(K * 64), b(K * 64), got(64 * 16);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    for (int i = 0; i < K * 64; i++) { a[i] = rnd(); b[i] = rnd() * 1e-3f; }
    for (int l = 0; l < 64; l++) { a[0 * 64 + l] = -0.f; b[1 * 64 + l] = (l & 1) ? 1e-41f : -1e-42f; a[2 * 64 + l] = 3e-39f; }
    for (int l = 0; l < 64; l++) if ((l & 15) == 7) for (int k = 0; k < K; k++) b[k * 64 + l] = (k & 1) ? 1e-40f : -2e-41f;
    float *da, *db, *dd;
    (void)hipMalloc(&da, a.size() * 4); (void)hipMalloc(&db, b.size() * 4); (void)hipMalloc(&dd, got.size() * 4);
    (void)hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, da, db, K, dd);
    (void)hipMemcpy(got.data(), dd, got.size() * 4, hipMemcpyDeviceToHost);
    // assumed: A lane l = block l/16, row l%16; B lane l = block l/16, column l%16;
    //          D lane l = 16 g + j, register v = 4 blk + r  <->  D[blk][row 4 g + r][column j]
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int v = 0; v < 16; v++) {
            const int g = l >> 4, j = l & 15, blk = v >> 2, r = v & 3, row = 4 * g + r;
            float acc = 0.f;
            for (int k = 0; k < K; k++) acc = fmaf(a[k * 64 + 16 * blk + row], b[k * 64 + 16 * blk + j], acc);
            const float gv = got[l * 16 + v];
            if (memcmp(&gv, &acc, 4) != 0) {
                bad++;
                if (bad <= 8) printf("  lane %2d reg %2d: mfma %.9g (0x%08x)  fmaf chain %.9g (0x%08x)\n", l, v, gv, *(unsigned *)&gv, acc, *(unsigned *)&acc);
            }
        }
    printf("numerics: %d of 1024 outputs differ bitwise from the fmaf chain in the assumed layout\n", bad);
    return 0;
}
