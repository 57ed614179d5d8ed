// photon_post.hip - sensor post-processing on the device (perform_ray_tracing_03.py:2190-2259) and
// the streaming-copy yardstick bench.py quotes next to the HBM specification.
#include <algorithm>
#include <cfloat>
#include <cmath>

#include "../../include/photon_philox.h"
#include "photon_internal.hpp"

using namespace photon;

// The streaming copy bench.py quotes as the achievable HBM rate next to the 8 TB/s specification.  Shape chosen by
// measurement (tools/ubench/copy_bw.hip, 28 shapes on one MI355X): every block owns ONE contiguous chunk, eight 16-byte
// loads in flight per lane, non-temporal loads and stores, 16 blocks per CU -- 5.5-5.6 TB/s read + write, against 4.1-4.7
// for the grid-stride form of rounds 1-2 and 5.1 for the runtime's own hipMemcpyAsync on the same box (the guide's
// 6.29 TB/s was not reached by any shape).
__global__ __launch_bounds__(256) void copy_float4_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f *s = reinterpret_cast<const v4f *>(src);
    v4f *d = reinterpret_cast<v4f *>(dst);
    constexpr int U = 8;
    const size_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per_block, b1 = b0 + per_block < n ? b0 + per_block : n;
    for (size_t i = b0 + threadIdx.x; i < b1; i += (size_t)U * 256) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + (size_t)u * 256 < b1) v[u] = __builtin_nontemporal_load(s + i + (size_t)u * 256);
#pragma unroll
        for (int u = 0; u < U; u++) if (i + (size_t)u * 256 < b1) __builtin_nontemporal_store(v[u], d + i + (size_t)u * 256);
    }
}

// ---------------------------------------------------------------------------------------------
// Sensor post-processing (perform_ray_tracing_03.py:2190-2259), the step right after the hot path, on the
// device: the raw f32 image stays in HBM and only the uint16 picture crosses the bus.  Arithmetic in f32 in
// the order numpy evaluates the reference's in-place expressions on its float32 array.
// ---------------------------------------------------------------------------------------------
// pass 1: (noise) -> clip negatives and non-finite values -> gain; leaves the scaled value in place? no: the raw
// image is only read (and, with noise, rewritten: the reference adds the noise to I_raw itself, :2196-2206).
__device__ __forceinline__ float postprocess_scaled(float v, float gain) {
    if (v < 0.0f) v = 0.0f;                     // I[I < 0] = 0 (NaN compares false and is zeroed next)
    if (!(fabsf(v) <= FLT_MAX)) v = 0.0f;       // I[~isfinite(I)] = 0
    return v * gain;                            // I *= 10 ** (pixel_gain / 20)
}
__global__ __launch_bounds__(256) void postprocess_max_kernel(float *__restrict__ image, size_t n, float gain, float noise_sigma,
                                                              unsigned long long seed, unsigned *__restrict__ max_bits) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = image[i];
        if (noise_sigma > 0.f) {                // np.random.normal(0, image_noise * 100) per pixel, seeded instead of time-seeded
            float n0, n1;
            photon_normal2(seed, (unsigned long long)i, 0u, PHOTON_STREAM_IMAGE_NOISE, &n0, &n1);
            v = v + n0 * noise_sigma;
            image[i] = v;
        }
        m = fmaxf(m, postprocess_scaled(v, gain));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(max_bits, __float_as_uint(m));      // non-negative floats order like their bits
}
// pass 2: normalise to the brightest pixel, round to the sensor's bit depth, stretch to 16 bit, crop
__global__ __launch_bounds__(256) void postprocess_quantize_kernel(const float *__restrict__ image, int W, int row0, int col0,
                                                                   int out_rows, int out_cols, float gain, float levels,
                                                                   float stretch, int rescale, const unsigned *__restrict__ max_bits,
                                                                   unsigned short *__restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (size_t)out_rows * out_cols) return;
    const int r = (int)(k / out_cols), c = (int)(k % out_cols);
    float v = image[(size_t)(row0 + r) * W + (col0 + c)];
    if (rescale) {
        v = postprocess_scaled(v, gain);
        const float mx = __uint_as_float(*max_bits);
        if (mx > 0.0f) v = (levels * v) / mx;   // (2**bits - 1) * I / max(I)
        v = rintf(v);                           // np.round: half to even
        v = v * stretch;                        // I *= (2**16 - 1) / (2**bits - 1)
    } else if (v < 0.0f) {
        v = 0.0f;
    }
    // np.uint16(I): C conversion (truncation); values beyond the range wrap like numpy's cast through int64
    out[k] = (unsigned short)(long long)v;
}

// Device-to-device float4 copy rate (read + write bytes per second, GB/s): what a trivial streaming kernel reaches on
// this GPU -- the "achievable HBM peak" bench.py prints next to the 8 TB/s specification.
extern "C" int photon_measure_copy_gbs(size_t bytes, int reps, double *gbs_out) {
    if (!gbs_out || bytes < 4096 || reps < 1) return 1;
    const size_t n = bytes / sizeof(float4);
    DeviceBuffer<float4> a, b;
    PH_CHECK(a.alloc(n));
    PH_CHECK(b.alloc(n));
    PH_CHECK(device_zero(a.p, n * sizeof(float4)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    PH_CHECK(hipEventCreate(&e0));
    PH_CHECK(hipEventCreate(&e1));
    const dim3 grid(256 * 16), block(256);
    hipLaunchKernelGGL(copy_float4_kernel, grid, block, 0, 0, a.p, b.p, n);          // warm-up
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(copy_float4_kernel, grid, block, 0, 0, a.p, b.p, n);
    (void)hipEventRecord(e1, 0);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    PH_CHECK(e);
    PH_CHECK(hipGetLastError());
    *gbs_out = ms > 0.f ? 2.0 * (double)(n * sizeof(float4)) * reps / (ms * 1e-3) * 1e-9 : 0.0;
    return 0;
}

// Sensor post-processing of perform_ray_tracing_03.py:2190-2259 on the device.
extern "C" int photon_postprocess_u16(float *d_image, int width, int height, float pixel_gain, int pixel_bit_depth,
                                      int intensity_rescaling, float image_noise, uint64_t noise_seed, int crop_rows,
                                      int crop_cols, uint16_t *d_out, int *out_rows, int *out_cols, void *stream_p) {
    if (!d_image || !d_out || width < 1 || height < 1 || pixel_bit_depth < 1 || pixel_bit_depth > 16 || crop_rows < 0 || crop_cols < 0) {
        fprintf(stderr, "photon: photon_postprocess_u16: bad arguments\n");
        return 1;
    }
    hipStream_t stream = (hipStream_t)stream_p;
    // crop window (:2250-2259): rows [nr/2 - nr_crop/2, nr/2 + nr_crop/2 - 1) with integer division -- one row and one
    // column fewer than asked for, as the reference's slice has it
    int row0 = 0, col0 = 0, rows = height, cols = width;
    if (crop_rows > 0 && crop_cols > 0) {
        row0 = height / 2 - crop_rows / 2; rows = crop_rows / 2 * 2 - 1;
        col0 = width / 2 - crop_cols / 2; cols = crop_cols / 2 * 2 - 1;
        if (row0 < 0 || col0 < 0 || rows < 1 || cols < 1 || row0 + rows > height || col0 + cols > width) {
            fprintf(stderr, "photon: photon_postprocess_u16: crop %d x %d does not fit a %d x %d image\n", crop_rows, crop_cols, height, width);
            return 1;
        }
    }
    if (out_rows) *out_rows = rows;
    if (out_cols) *out_cols = cols;
    const size_t n = (size_t)width * height;
    const float gain = (float)pow(10.0, (double)pixel_gain / 20.0);                     // python float, cast to the array's f32
    const float levels = (float)((1 << pixel_bit_depth) - 1);
    const float stretch = (float)(65535.0 / ((double)(1 << pixel_bit_depth) - 1.0));
    DeviceBuffer<unsigned> d_max;
    PH_CHECK(d_max.alloc(1));
    PH_CHECK(hipMemsetAsync(d_max.p, 0, sizeof(unsigned), stream));
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(postprocess_max_kernel, dim3(blocks), dim3(256), 0, stream, d_image, n, gain, image_noise > 0.f ? image_noise * 100.0f : 0.f,
                       (unsigned long long)noise_seed, d_max.p);
    PH_CHECK(hipGetLastError());
    const size_t nout = (size_t)rows * cols;
    hipLaunchKernelGGL(postprocess_quantize_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, stream, d_image, width, row0, col0,
                       rows, cols, gain, levels, stretch, intensity_rescaling ? 1 : 0, d_max.p, (unsigned short *)d_out);
    PH_CHECK(hipGetLastError());
    PH_CHECK(hipStreamSynchronize(stream));         // d_max dies here
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Self-test hook: the march loops' normal-range division / reciprocal / square root (device_vec.hpp: the compiler's correctly
// rounded sequences without their range scaling) evaluated on caller-supplied operands, so that a test can hold them against
// IEEE division and square root bit for bit (tests/test_parity_gpu.py::test_normal_range_division_and_sqrt_are_exact).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normal_range_math_kernel(int n, const float *__restrict__ a, const float *__restrict__ b,
                                                                float *__restrict__ quot, float *__restrict__ rcp, float *__restrict__ root) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    quot[i] = div_nr(a[i], b[i]);
    rcp[i] = rcp_nr(b[i]);
    root[i] = sqrt_nr(a[i]);
}

extern "C" int photon_selftest_normal_range_math(int n, const float *a, const float *b, float *quot, float *rcp, float *root) {
    if (n < 0 || !a || !b || !quot || !rcp || !root) return 1;
    if (n == 0) return 0;
    DeviceBuffer<float> d;
    PH_CHECK(d.alloc((size_t)n * 5));
    float *da = d.p, *db = d.p + n, *dq = d.p + 2 * (size_t)n, *dr = d.p + 3 * (size_t)n, *ds = d.p + 4 * (size_t)n;
    PH_CHECK(hipMemcpy(da, a, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    PH_CHECK(hipMemcpy(db, b, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(normal_range_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, da, db, dq, dr, ds);
    PH_CHECK(hipGetLastError());
    PH_CHECK(hipMemcpy(quot, dq, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    PH_CHECK(hipMemcpy(rcp, dr, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    PH_CHECK(hipMemcpy(root, ds, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
