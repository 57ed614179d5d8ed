// photon_core.hip - libparallel_ray_tracing.so for AMD Instinct MI355X (gfx950).
//
// Hand-written HIP implementation of photon's ray-tracing core behind the reference's C-ABI
// (include/parallel_ray_tracing.h).  What runs where:
//
//   host   start_ray_tracing / photon_*      argument marshalling, NRRD parse, chunk loop, dumps
//   GPU    build_volume_kernel               density -> (grad n, n-1) float4 texels
//          prefilter_lines_kernel x3         cubic B-spline prefilter, per channel (x, y, z lines)
//          raygen_kernel                     ray generation + Mie lookup + world transform -> SoA state
//          march_kernel<ALGO,INTERP>         Euler/RK4 march through the volume, in place on the state
//          sensor_kernel<FROM_STATE>         (ray generation |) lens / aperture / apparent image
//                                            + erf or 4-pixel splat, f64 atomics into a private
//                                            accumulator
//          finalize_image_kernel             image = (float)(image + accumulator)
//
// The product has no CPU compute path: if HIP reports an error the call fails loudly
// (message on stderr, non-zero return / untouched image).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (photon_amd/build.py).

#include <hip/hip_runtime.h>

#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/parallel_ray_tracing.h"
#include "device_optics.hpp"
#include "device_vec.hpp"
#include "device_volume.hpp"
#include "device_volume_coop.hpp"
#include "device_volume_extra.hpp"
#include "photon_sort.hpp"

using namespace photon;

// =============================================================================================
// error handling
// =============================================================================================
#define PH_CHECK(expr)                                                                          \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            fprintf(stderr, "photon: HIP error %d (%s) at %s:%d: %s\n", (int)_e,                \
                    hipGetErrorString(_e), __FILE__, __LINE__, #expr);                          \
            return (int)_e;                                                                     \
        }                                                                                       \
    } while (0)

static bool verbose() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("PHOTON_VERBOSE"); v = (e && atoi(e) > 0) ? 1 : 0; }
    return v == 1;
}

// a device allocation that is released on every return path
template <typename T>
struct DeviceBuffer {
    T *p = nullptr;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)); }
};

// No C++ exception may cross the C boundary (a ctypes caller would be terminated): every extern "C" body that
// allocates host memory runs inside this guard.
template <typename F>
static int guarded(const char *what, F &&body) {
    try {
        return body();
    } catch (const std::exception &e) {
        fprintf(stderr, "photon: %s failed: %s\n", what, e.what());
    } catch (...) {
        fprintf(stderr, "photon: %s failed: unknown exception\n", what);
    }
    return 100;
}

// =============================================================================================
// a cache of freed device blocks
// =============================================================================================
// photon's unchanged Python builds everything anew for every start_ray_tracing call: per call ~25 hipMalloc / hipFree
// pairs, among them the ray-state workspace (320 MB for the 1e7-ray job) -- measured, the frees alone take 0.8-1.3 ms of a
// call (PHOTON_VERBOSE), 10 % of one GPU's eighth of the headline job, most of a small PIV frame.  Scene-lifetime blocks are
// therefore handed back to this cache instead of the runtime and the next call of the same shape takes them from it
// (exact size match, per device); the cache holds at most PHOTON_POOL_MAX_MB (default 4096; 0 = off), evicting its largest
// blocks first; photon_trim_caches() empties it.  Recycled memory is not zeroed -- neither is hipMalloc'd memory: every
// buffer that needs a defined start is cleared where it is allocated or used.
namespace {
struct PoolKey {
    int device; size_t bytes;
    bool operator<(const PoolKey &o) const { return device != o.device ? device < o.device : bytes < o.bytes; }
};
struct DevicePool {
    std::mutex lock;
    std::multimap<PoolKey, void *> idle;
    std::map<void *, PoolKey> live;
    size_t idle_bytes = 0;
};
DevicePool &device_pool() { static DevicePool *p = new DevicePool; return *p; }      // never destroyed: the runtime may be gone by then
size_t pool_cap_bytes() {
    static const size_t cap = [] { const char *e = getenv("PHOTON_POOL_MAX_MB"); return (size_t)(e ? strtoull(e, nullptr, 10) : 4096ull) << 20; }();
    return cap;
}
void pool_trim(size_t keep_bytes) {                              // caller holds no lock
    DevicePool &p = device_pool();
    std::vector<void *> victims;
    {
        std::lock_guard<std::mutex> g(p.lock);
        while (p.idle_bytes > keep_bytes && !p.idle.empty()) {
            auto big = p.idle.begin();
            for (auto it = p.idle.begin(); it != p.idle.end(); ++it) if (it->first.bytes > big->first.bytes) big = it;
            p.idle_bytes -= big->first.bytes;
            victims.push_back(big->second);
            p.idle.erase(big);
        }
    }
    for (void *v : victims) (void)hipFree(v);
}
hipError_t pool_malloc(void **out, size_t bytes) {
    if (bytes == 0) bytes = 1;
    int device = 0;
    (void)hipGetDevice(&device);
    DevicePool &p = device_pool();
    {
        std::lock_guard<std::mutex> g(p.lock);
        auto it = p.idle.find(PoolKey{device, bytes});
        if (it != p.idle.end()) {
            *out = it->second;
            p.idle.erase(it);
            p.idle_bytes -= bytes;
            p.live[*out] = PoolKey{device, bytes};
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); pool_trim(0); e = hipMalloc(out, bytes); }
    if (e == hipSuccess) { std::lock_guard<std::mutex> g(p.lock); p.live[*out] = PoolKey{device, bytes}; }
    return e;
}
void pool_free(void *ptr) {
    if (!ptr) return;
    DevicePool &p = device_pool();
    bool keep = false;
    {
        std::lock_guard<std::mutex> g(p.lock);
        auto it = p.live.find(ptr);
        if (it != p.live.end()) {
            const PoolKey k = it->second;
            p.live.erase(it);
            if (k.bytes <= pool_cap_bytes()) { p.idle.emplace(k, ptr); p.idle_bytes += k.bytes; keep = true; }
        }
    }
    if (!keep) { (void)hipFree(ptr); return; }
    if (device_pool().idle_bytes > pool_cap_bytes()) pool_trim(pool_cap_bytes());
}
}  // namespace

extern "C" void photon_trim_caches(void) { pool_trim(0); }

#if PHOTON_PATH_STATS
// debug builds only: read (and clear) the sampler-path counters of device_volume_coop.hpp
extern "C" int photon_debug_path_stats(unsigned long long out[8]) {
    PH_CHECK(hipDeviceSynchronize());
    PH_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(photon::g_path_stats), 8 * sizeof(unsigned long long)));
    unsigned long long zero[8] = {};
    PH_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(photon::g_path_stats), zero, sizeof zero));
    return 0;
}
#endif

// =============================================================================================
// device-side aggregates
// =============================================================================================
struct RayStateDev {                // SoA ray state between the march and the sensor stage
    float *px, *py, *pz, *dx, *dy, *dz;
    double *radiance;
    // what a ray carries between the segments of a segmented march (device_volume_coop.hpp, MarchResume), per ray:
    unsigned *ctr;                  // bit 31: still marching; bits 24-30: the segment that wrote the word; bits 0-23: completed iterations
    unsigned *spins;
    float *vprev;                   // [4][rays]: the last value sampled (trilinear branches)
    unsigned *seg_flag;             // per 64-ray group: (launch epoch << 8) | segments completed (0xff: every ray has left)
    unsigned stride;                // rays the arrays were allocated for (distance between the four planes of vprev)
};

struct DumpDev {                    // ray dumps (save_lightrays), indexed by chunk-global ray id
    float *final_pos;               // [num_save][3] or nullptr
    float *final_dir;
    int num_save;
    float *inter_pos;               // [num_save][inter_slots][3] or nullptr (save_intermediate_ray_data)
    float *inter_dir;
    int inter_slots;
};

enum { CNT_ON_SENSOR = 0, CNT_ITER = 1, CNT_SAMPLES = 2, CNT_TAPS = 3, CNT_MARCHED = 4, CNT_CLK = 5, CNT_REAL = 6, CNT_N = 7 };
// Statistics counters are kept in kCounterSlots copies (one 64-byte line each) and summed on the host: with one
// copy every wave of a launch ends on an atomic to the SAME address, and 1.6e5 same-address device-scope atomics
// serialise into ~2 ms -- more than the rest of the sensor stage (measured).
constexpr int kCounterSlots = 1024;
constexpr int kCounterStride = 8;       // u64 per slot: CNT_N used, padded to a cache line
__device__ __forceinline__ unsigned long long *counter_slot(unsigned long long *counters) {
    return counters + (size_t)(blockIdx.x % kCounterSlots) * kCounterStride;
}

// Blocks b and b+8 share an XCD (round-robin dispatch).  Rays that walk the same voxels should meet in one L2,
// so each XCD gets whole CHUNKS of PHOTON_XCD_CHUNK consecutive logical blocks (8 K rays: some dozen neighbouring
// sources); the chunks themselves are dealt round-robin, which keeps the eight XCDs evenly loaded when the work
// per ray varies along the launch (lens-major launches: whole lens samples may be skipped as doomed).
// Bijective for any grid size (the ragged tail is left in place).  Speed only -- never correctness.
#ifndef PHOTON_XCD_CHUNK
#define PHOTON_XCD_CHUNK 32
#endif
#ifndef PHOTON_MARCH_BLOCK
#define PHOTON_MARCH_BLOCK 256          // threads per workgroup of the march (a multiple of 64)
#endif
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nb) {
    constexpr unsigned B = PHOTON_XCD_CHUNK * (256 / PHOTON_MARCH_BLOCK);
    const unsigned full = nb / (8u * B) * (8u * B);            // blocks that form complete rounds of 8 chunks
    if (bid >= full) return bid;
    const unsigned xcd = bid & 7u, q = bid >> 3;                // q-th block this XCD receives
    return ((q / B) * 8u + xcd) * B + q % B;
}

__device__ __forceinline__ void wave_add(unsigned long long *dst, unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}

// =============================================================================================
// volume construction kernels
// =============================================================================================

// setData (trace_rays_through_density_gradients.h:1820-2002) with loadNRRD's Gladstone-Dale
// scaling (.h:1729-1748) folded in: one thread per voxel; edges use the double-precision
// one-sided stencils, the interior the f32 (x, z) / f64-divisor (y) central differences.
__global__ __launch_bounds__(256) void build_volume_kernel(const float *__restrict__ rho, int W, int H, int D,
                                                           float gx, float gy, float gz, f4 *__restrict__ out,
                                                           float *__restrict__ block_min) {
    const size_t n = (size_t)W * H * D;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float K = 0.225e-3;
    float mine = FLT_MAX, gmag = 0.f;
    if (i < n) {
        const int x = (int)(i % W), y = (int)((i / W) % H), z = (int)(i / ((size_t)W * H));
        const size_t WH = (size_t)W * H;
        auto d = [&](int xx, int yy, int zz) { return K * (rho[zz * WH + (size_t)yy * W + xx] * 1.0f); };
        float nxv, nyv, nzv, s1, s2, s3;
        if (x < 1) {
            s1 = d(x, y, z); s2 = d(x + 1, y, z); s3 = d(x + 2, y, z);
            nxv = (float)((-3.0 / 2 * s1 + 2 * s2 - 1.0 / 2 * s3) / gx);
        } else if (x >= W - 1) {
            s1 = d(x, y, z); s2 = d(x - 1, y, z); s3 = d(x - 2, y, z);
            nxv = (float)((3.0 / 2 * s1 - 2 * s2 + 1.0 / 2 * s3) / gx);
        } else {
            s1 = d(x - 1, y, z); s2 = d(x + 1, y, z);
            nxv = (s2 - s1) / (2 * gx);
        }
        if (y < 1) {
            s1 = d(x, y, z); s2 = d(x, y + 1, z); s3 = d(x, y + 2, z);
            nyv = (float)((-3.0 / 2 * s1 + 2 * s2 - 1.0 / 2 * s3) / gy);
        } else if (y >= H - 1) {
            s1 = d(x, y, z); s2 = d(x, y - 1, z); s3 = d(x, y - 2, z);
            nyv = (float)((3.0 / 2 * s1 - 2 * s2 + 1.0 / 2 * s3) / gy);
        } else {
            s1 = d(x, y - 1, z); s2 = d(x, y + 1, z);
            nyv = (float)((s2 - s1) / (2.0 * gy));
        }
        if (z < 1) {
            s1 = d(x, y, z); s2 = d(x, y, z + 1); s3 = d(x, y, z + 2);
            nzv = (float)((-3.0 / 2 * s1 + 2 * s2 - 1.0 / 2 * s3) / gz);
        } else if (z >= D - 1) {
            s1 = d(x, y, z); s2 = d(x, y, z - 1); s3 = d(x, y, z - 2);
            nzv = (float)((3.0 / 2 * s1 - 2 * s2 + 1.0 / 2 * s3) / gz);
        } else {
            s1 = d(x, y, z - 1); s2 = d(x, y, z + 1);
            nzv = (s2 - s1) / (2 * gz);
        }
        const float w = d(x, y, z);
        *reinterpret_cast<float4 *>(out + i) = make_float4(nxv, nyv, nzv, w);
        mine = w;
        gmag = sqrtf(nxv * nxv + nyv * nyv + nzv * nzv);
    }
    // block minimum of n-1 (data_min, .h:1868-1869)
    __shared__ float red[256];
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fminf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) block_min[blockIdx.x] = red[0];
    // block maximum of |grad n| (bounds how far the volume can bend a ray: launch_chunk's doom margin)
    __syncthreads();
    red[threadIdx.x] = gmag == gmag ? gmag : 0.f;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) block_min[gridDim.x + blockIdx.x] = red[0];
}

// rho[k][j][i] = rho0 + amp * gz[k] * (gy[j] * gx[i]) in double, rounded once to f32 (the order of the
// numpy expression a host-side generator would use): the density of photon_volume_gaussian.
__global__ __launch_bounds__(256) void separable_density_kernel(const double *__restrict__ gx, const double *__restrict__ gy,
                                                                const double *__restrict__ gz, int W, int H, int D, double rho0,
                                                                double amp, float *__restrict__ rho) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)W * H * D) return;
    const int i = (int)(idx % W), j = (int)((idx / W) % H), k = (int)(idx / ((size_t)W * H));
    rho[idx] = (float)(rho0 + amp * gz[k] * (gy[j] * gx[i]));
}

// Light-field sources generated in HBM.
// BOS target (generate_bos_lightfield_data, run_simulation_02.py:1328-1551): source (dot g, point j) sits at
// (dot_x[g] + tmpl_x[j], dot_y[g] + tmpl_y[j], z); sums in double, cast to f32 like the ctypes marshalling.
__global__ __launch_bounds__(256) void sources_bos_kernel(const double *__restrict__ dot_x, const double *__restrict__ dot_y,
                                                          long long n_dots, const double *__restrict__ tx,
                                                          const double *__restrict__ ty, int n_tmpl, double z, double radiance,
                                                          float *sx, float *sy, float *sz, double *srad, int *sdia) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_dots * n_tmpl) return;
    const long long g = i / n_tmpl;
    const int j = (int)(i % n_tmpl);
    sx[i] = (float)(dot_x[g] + tx[j]);
    sy[i] = (float)(dot_y[g] + ty[j]);
    sz[i] = (float)z;
    srad[i] = radiance;
    sdia[i] = 1;                                                        // run_simulation_02.py:1544
}

// PIV particle field (run_simulation_02.py:774-996): X, Y, Z uniform in the box, radiance = the laser
// sheet's Gaussian profile in Z, Z shifted to the object plane.  The reference draws from numpy's unseeded
// generator; here particle i takes the four words of Philox(seed, i) -- any particle can be regenerated.
struct PivFieldDev {
    double lo[3], hi[3];
    double z_object, coef, two_sigma2;      // coef = irradiance_constant / (sigma sqrt(2 pi))
    int n_diameters;                        // 0: diameter_index = 1 (run_simulation_02.py:992)
};
__global__ __launch_bounds__(256) void sources_piv_kernel(unsigned long long seed, long long n, PivFieldDev f,
                                                          const double *__restrict__ diameter_cdf, float *sx, float *sy,
                                                          float *sz, double *srad, int *sdia) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const photon_u32x4 r = photon_philox4x32_10(seed, (unsigned long long)i, 0u, PHOTON_STREAM_SCENE);
    const double ux = ((double)r.x + 0.5) * (1.0 / 4294967296.0), uy = ((double)r.y + 0.5) * (1.0 / 4294967296.0);
    const double uz = ((double)r.z + 0.5) * (1.0 / 4294967296.0), ud = ((double)r.w + 0.5) * (1.0 / 4294967296.0);
    const double X = (f.hi[0] - f.lo[0]) * ux + f.lo[0];
    const double Y = (f.hi[1] - f.lo[1]) * uy + f.lo[1];
    const double Z = (f.hi[2] - f.lo[2]) * uz + f.lo[2];
    sx[i] = (float)X;
    sy[i] = (float)Y;
    sz[i] = (float)(Z + f.z_object);
    srad[i] = f.coef * photon_det_exp(-1.0 * (Z * Z / f.two_sigma2));
    int dia = 1;
    if (f.n_diameters > 0) {
        dia = f.n_diameters - 1;
        for (int d = 0; d < f.n_diameters; d++)
            if (ud < diameter_cdf[d]) { dia = d; break; }
    }
    sdia[i] = dia;
}

// ConvertToInterpolationCoefficients (cubicPrefilter_kernel.cu:52-112) on all four channels of one
// line of float4 texels, in place.  One thread per line; `lines_inner` lines are adjacent in
// memory by `inner_stride` texels (coalesced for the y and z passes).
__global__ __launch_bounds__(256) void prefilter_lines_kernel(f4 *vol, int len, size_t len_stride, int lines_inner,
                                                              size_t inner_stride, int lines_outer,
                                                              size_t outer_stride) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)lines_inner * lines_outer) return;
    const size_t li = t % lines_inner, lo = t / lines_inner;
    f4 *c = vol + lo * outer_stride + li * inner_stride;
    const float Pole = sqrtf(3.0f) - 2.0f;
    const float Lambda = (1.0f - Pole) * (1.0f - 1.0f / Pole);
    const int horizon = len < 12 ? len : 12;
    float zn = Pole;
    f4 first = c[0];
    f4 sum = first;
    for (int k = 0; k < horizon; k++) {
        const f4 v = c[k * len_stride];
        sum.x += zn * v.x; sum.y += zn * v.y; sum.z += zn * v.z; sum.w += zn * v.w;
        zn *= Pole;
    }
    f4 prev = f4{Lambda * sum.x, Lambda * sum.y, Lambda * sum.z, Lambda * sum.w};
    c[0] = prev;
    for (int k = 1; k < len; k++) {
        const f4 v = c[k * len_stride];
        prev = f4{Lambda * v.x + Pole * prev.x, Lambda * v.y + Pole * prev.y, Lambda * v.z + Pole * prev.z,
                  Lambda * v.w + Pole * prev.w};
        c[k * len_stride] = prev;
    }
    const float g = Pole / (Pole - 1.0f);
    const f4 last = c[(size_t)(len - 1) * len_stride];
    prev = f4{g * last.x, g * last.y, g * last.z, g * last.w};
    c[(size_t)(len - 1) * len_stride] = prev;
    for (int k = len - 2; k >= 0; k--) {
        const f4 v = c[k * len_stride];
        prev = f4{Pole * (prev.x - v.x), Pole * (prev.y - v.y), Pole * (prev.z - v.z), Pole * (prev.w - v.w)};
        c[k * len_stride] = prev;
    }
}

__global__ __launch_bounds__(256) void sample_kernel(VolumeDev v, int n, const float *__restrict__ coords,
                                                     float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = coords[3 * i], y = coords[3 * i + 1], z = coords[3 * i + 2];
    const f4 r = v.interpolation == 2 ? tex3d_cubic(v, x, y, z) : tex3d_linear(v, x, y, z);
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

template <int ALGO, int INTERP>
__global__ __launch_bounds__(256) void march_rays_kernel(VolumeDev v, const f4 *__restrict__ tex, int n,
                                                         float *__restrict__ pos, float *__restrict__ dir,
                                                         int *__restrict__ steps) {
    __shared__ f4 tiles[4][kWaveLdsTexels];                     // per wave: 4x4x4 tile + 8x8x4 brick, rows padded (device_volume_coop.hpp)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool has_ray = i < n;
    f3 p = mk3(0, 0, 0), d = mk3(0, 0, -1);
    if (has_ray) {
        p = mk3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        d = mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    }
    MarchCount mc{0, 0};
    const GradNoise no_noise{0, 0.f, 0ull, 0ull};
    const InterDump no_dump{nullptr, nullptr, 0, 0, 0u};
    MarchResume rs = resume_fresh();
    if (INTERP == 1 && v.weight_scale > 0.f)                    // kernel-uniform: texture-unit weights
        trace_volume_coop<ALGO, INTERP, false, false, true, MarchCount>(has_ray, p, d, v, tex, tiles[threadIdx.x >> 6], mc, no_noise, no_dump, rs);
    else
        trace_volume_coop<ALGO, INTERP, false, false, false, MarchCount>(has_ray, p, d, v, tex, tiles[threadIdx.x >> 6], mc, no_noise, no_dump, rs);
    if (has_ray) {
        pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z;
        dir[3 * i] = d.x; dir[3 * i + 1] = d.y; dir[3 * i + 2] = d.z;
        if (steps) steps[i] = mc.iterations;
    }
}

// Algorithms 3 (rk45), 4 (adams_bashforth) and "anything else" (the reference's `default: break`,
// .h:1537: the ray is only moved to its entry point): per-lane code, trilinear gathers of the raw volume.
template <int ALGO>
__global__ __launch_bounds__(256) void march_rays_extra_kernel(VolumeDev v, int n, float *__restrict__ pos,
                                                               float *__restrict__ dir, int *__restrict__ steps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 p = mk3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
    f3 d = mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    MarchCount mc{0, 0};
    trace_volume_extra<ALGO>(p, d, v, mc);
    pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z;
    dir[3 * i] = d.x; dir[3 * i + 1] = d.y; dir[3 * i + 2] = d.z;
    if (steps) steps[i] = mc.iterations;
}

// =============================================================================================
// the two ray-tracing kernels
// =============================================================================================

// Stage 1a (density gradients on): generate the ray and move it into the volume's world frame
// (parallel_ray_tracing.cu:2004-2082).  Kept apart from the march so that the hot kernel carries
// neither the scene description (a kilobyte of kernel arguments pinned in SGPRs) nor the
// double-precision ray-generation code in its register budget.
__global__ __launch_bounds__(256) void raygen_kernel(SceneDev sc, long long src_begin, unsigned n_rays, RayStateDev st) {
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    int source, local_ray;
    slot_to_ray(sc, src_begin, n_rays, r, source, local_ray);
    const Ray ray = generate_ray(sc, source, local_ray);
    f3 p = ray.pos, d = ray.dir;
    p.z = (float)(p.z - (sc.z_offset + 750e3));                         // .cu:2045
    p = matvec(sc.cam.inverse_rotation_matrix, p);                      // camera -> world
    d = matvec(sc.cam.inverse_rotation_matrix, d);
    if (sc.doom_margin > 0.f) {
        // aimed so far outside the first element's aperture that no deflection the volume can produce brings it
        // back (margin from the volume's largest gradient, launch_chunk): dead on arrival -- mark it, the march
        // skips it, the sensor stage drops it as it would after the lens.  Half of a full-aperture PIV cone.
        const float dist = front_axis_distance(sc.elems[0], mk3(sc.centers[0][0], sc.centers[0][1], sc.centers[0][2]),
                                               sc.planes[0], ray);
        if (dist > sc.elems[0].element_geometry.pitch / 2.0 + sc.doom_margin) p = nan3();
    }
    st.px[r] = p.x; st.py[r] = p.y; st.pz[r] = p.z;
    st.dx[r] = d.x; st.dy[r] = d.y; st.dz[r] = d.z;
    st.radiance[r] = ray.radiance;
}

// Stage 1b: march the rays through the volume, in place on the SoA state (world frame).  One lane per ray;
// the launch's ray order (source-major / lens-major, SceneDev::ray_order) decides which rays share a wave.
// Launch bound: 5 waves per SIMD for the tricubic kernels (<= 96 VGPRs; their 7.75 KiB of LDS per wave allow no more), 6
// for the RK4 trilinear ones (below).  A wave issues at most one VALU instruction per ~4 cycles, the
// SIMD one per 2, and every wave spends part of its time waiting on LDS: the more resident waves the better
// (C3 tricubic: 3 waves 100.8 ms, 4 waves 93.2 ms at the time; now 4 waves 68.8, 5 waves 67.1 ms; trilinear 28.0
// -> 25.1 ms).  What made 96 registers reachable was the out-of-line gather fallback: under the AMDGPU calling
// convention the caller's live values sit ABOVE the callee's registers, so its 81 VGPRs were part of the march
// kernels' budget until it was rewritten to need 51 (device_volume.hpp).
#ifndef PHOTON_MARCH_WAVES
#define PHOTON_MARCH_WAVES 5
#endif
#ifndef PHOTON_MARCH_WAVES_LINEAR
#define PHOTON_MARCH_WAVES_LINEAR 6     // RK4 trilinear: a sixth wave (80 VGPRs) cost 18 spilled dwords in the loop (36 in the segmented instantiation) and
#endif                                  // still won once the sampler work of round 4 had left the kernel waiting -- C3 17.19 -> 16.32 ms, C5 quarter 13.26 ->
                                        // 12.49, one GPU's eighth of C3 2.278 -> 2.270 (round 2, 197 instructions per sample: 26.6 -> 27.7 ms); with the
                                        // lean out-of-line gather (device_volume_coop.hpp) 3 / 26 spilled dwords: 16.29 and 2.21 ms.  Seven waves: 16.75
#ifndef PHOTON_MARCH_WAVES_EULER_LINEAR
#define PHOTON_MARCH_WAVES_EULER_LINEAR 6   // Euler trilinear: the whole-march kernel needs 71 VGPRs (seven waves per SIMD as it is); the segmented one 83:
#endif                                      // capped at 80 it spills 2 dwords and runs a sixth wave -- one GPU's eighth 0.849 -> 0.809 ms, a quarter 1.625 ->
                                            // 1.531 (before the lean out-of-line gather the cap cost 17 spilled dwords: 0.845 -> 0.855)
#ifndef PHOTON_MARCH_WAVES_NOISE
#define PHOTON_MARCH_WAVES_NOISE 3      // the gradient-noise instantiations (Philox + Box-Muller in f64 inside the loop) need ~130 VGPRs: at five
#endif                                  // waves per SIMD they spilled 46-70 of them into the loop (176-208 B of scratch per lane); three waves, no spill
template <int ALGO, int INTERP, bool NOISE> constexpr int march_waves() {
    return NOISE ? PHOTON_MARCH_WAVES_NOISE : INTERP == 1 ? (ALGO == 2 ? PHOTON_MARCH_WAVES_LINEAR : PHOTON_MARCH_WAVES_EULER_LINEAR) : PHOTON_MARCH_WAVES;
}
// resident march waves per SIMD of a launch (the segment planner's chip fill)
static unsigned march_waves_of(int algorithm, int interp) {
    return interp == 1 ? (algorithm == 2 ? PHOTON_MARCH_WAVES_LINEAR : 6 /* whole marches: 71 VGPRs */) : PHOTON_MARCH_WAVES;
}
// Shader-clock stamp of a wave: s_memtime ticks at the shader clock, s_memrealtime at a constant 100 MHz
// (MI355X_MICROARCH.md, "DVFS give-back" item 6).  The chip lowers its clock under load, by an amount that differs from
// device to device; the ratio of the two deltas, summed over the waves of a launch, is the clock the march actually ran
// at -- what bench.py normalises its roofline fraction with.  Two stamps per wave (a wave lives ~2 ms): no cost.
__device__ __forceinline__ void clock_stamp(unsigned long long &clk, unsigned long long &real) {
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk), "=s"(real) : : "memory");
}

// PERSISTENT WAVES (round 3).  Every ray of a BOS launch marches for the same ~1.8 ms, so the waves of a conventional
// launch finish generation by generation, and each time the dispatcher has a whole chip's worth of workgroups to start at
// once: measured on C3, the resident-wave slots stood empty 7.5 % of the kernel's time (156 250 waves x 1.76 ms mean
// lifetime / 5120 slots = 53.9 ms of work in a 58.2 ms kernel; the same 7.7 % on the 1.2 s C4 launch, and a launch of G
// generations lasted about G + 0.9 lifetimes -- one GPU's eighth of C3, 3.8 generations, ran at 81 % occupancy).
// Here the grid is just large enough to fill the chip ONCE and every wave takes 64-ray groups from a queue until the
// launch is served: a wave that finishes a group loads the next one itself, no slot waits for the dispatcher.
// 32 queues, four per XCD (workgroup i runs on XCD i % 8), each handing out the groups of its chunks (16 or 128 consecutive groups: kChunkShift*) in order, so
// rays that walk the same voxels still meet in one L2 (what xcd_remap did for the one-shot launch); the visiting order and
// why four are at the loop.  Every wave leaves as soon as its eleven queues are past their ends.
#ifndef PHOTON_MARCH_PERSISTENT
#define PHOTON_MARCH_PERSISTENT 1
#endif
#ifndef PHOTON_MARCH_SEGMENTS
#define PHOTON_MARCH_SEGMENTS 32        // most segments a ray's march is cut into in launches of several chip fills (launch_march picks)
#endif
constexpr unsigned kQueueStride = 16;                           // u32 per queue counter: one 64-byte line each
#ifndef PHOTON_SUBQUEUES
#define PHOTON_SUBQUEUES 4              // work queues per XCD (a power of two, <= 8); measured 1 / 2 / 4 / 8, see march_kernel
#endif
constexpr unsigned kSubQueues = PHOTON_SUBQUEUES;
constexpr unsigned kQueues = 64;                                // room for 8 XCDs x 8 sub-queues
// Consecutive 64-ray groups an XCD's queue owns as one CHUNK: 2^shift.  Large chunks keep the rays of neighbouring sources
// in one L2; small ones balance the XCDs' queues at the end of a launch.  Measured on C3 with the segmented march (HBM
// traffic does not care: 4.0-4.15 GB): tricubic RK4 march with chunks of 128 / 32 / 16 / 8 groups 57.70 / 57.59 / 57.53 /
// 57.55 ms (one GPU's eighth 7.60 / 7.55 / 7.53 / 7.51), trilinear RK4 19.83 / 19.89 / 19.92: 16 for the tricubic kernels in
// source-major launches through volumes of up to 256^3 texels, 128 otherwise (launch_march says why).
#ifndef PHOTON_CHUNK_SHIFT_CUBIC
#define PHOTON_CHUNK_SHIFT_CUBIC 4
#endif
constexpr unsigned kChunkShiftCubic = PHOTON_CHUNK_SHIFT_CUBIC, kChunkShiftLinear = 7;
// The k-th group handed out by sub-queue `sub` of XCD `xcd`, C = 2^shift groups per chunk: chunk ((k / C) * 4 + sub) * 8 + xcd, group k % C of it.  Grows
// with k, so the first k whose group lies past the launch ends the queue; every group belongs to exactly one (xcd, sub).
__host__ __device__ inline unsigned march_queue_group(unsigned k, unsigned xcd, unsigned sub, unsigned shift) {
    return ((((k >> shift) * kSubQueues + sub) * 8u + xcd) << shift) + (k & ((1u << shift) - 1u));
}

// The march kernel's arguments, read from the kernel-argument segment WHERE THEY ARE USED (scalar loads through a pointer
// the optimiser cannot see through) instead of being held in SGPRs from the prologue on: the persistent loop needs them
// again for every group, and ~55 argument SGPRs live across the march loop -- whose own constants, masks and tile ids take
// ~60 -- overflowed the 102 a wave has (15-55 SGPRs spilled into VGPR lanes, and VGPRs into scratch).
constexpr unsigned kMaxSegments = 64;
struct MarchArgs {
    VolumeDev vol;
    const f4 *tex;
    unsigned n_rays;
    RayStateDev st;
    unsigned long long *counters;
    NoiseDev noise;
    unsigned long long ray_base;
    InterDump idump;
    unsigned *queue;
    unsigned long long *profile;        // this launch's wave-timing slots (photon_scene_set_march_profile), or nullptr
    unsigned segments;                  // segments every ray's march is cut into (1: whole marches, the state arrays below unused)
    unsigned seg_begin[kMaxSegments + 1];   // segment s covers the trips [seg_begin[s], seg_begin[s + 1]) of the march loop (the last
                                        // one runs until every ray has left): equal, halving or tapered pieces (plan_segments)
    unsigned epoch;                     // tag of this launch in RayStateDev::seg_flag
    unsigned *error;                    // waves that gave a segment up (zero unless the hand-off between segments is broken)
    unsigned chunk_shift;               // log2 of the groups per queue chunk (launch_march)
};
typedef const __attribute__((address_space(4))) MarchArgs *MarchArgsPtr;
template <class T>
__device__ __forceinline__ T load_arg(const __attribute__((address_space(4))) T *p) {      // scalar loads from the argument segment
    T out;
    __builtin_memcpy(&out, p, sizeof(T));
    return out;
}
__device__ __forceinline__ MarchArgsPtr march_args() {
    MarchArgsPtr p = (MarchArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                                 // a fresh pointer each time: loads through it are neither hoisted nor kept
    return p;
}

// Wave timing of a march launch (photon_scene_set_march_profile; off by default): when the first wave entered, when each
// wave started its first group and when it left, on the constant 100 MHz clock -- what tells a launch's start-up cost
// (dispatch, cold caches) from its drain (the last groups finishing one by one while the rest of the chip idles).
// kProfileSub copies per launch (a cache line each, chosen by workgroup) so that the stamps of a chip's worth of waves
// do not serialise on one address; minima are kept as maxima of the complement, so a slot starts from zeros.
enum { PF_ENTER_NEGMIN = 0, PF_START_NEGMIN, PF_START_SUM, PF_START_MAX, PF_END_NEGMIN, PF_END_SUM, PF_END_MAX, PF_WAVES, PF_N };
constexpr unsigned kProfileLaunches = 64, kProfileSub = 64;
__device__ __forceinline__ unsigned long long real_time() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
__device__ __forceinline__ unsigned long long *profile_slot() {
    unsigned long long *p = march_args()->profile;
    return p ? p + (size_t)(blockIdx.x % kProfileSub) * PF_N : nullptr;
}

// What a march wave accumulates over the groups it serves (wave-uniform: SGPRs) and adds to the counters once, at its end.
struct WaveTotals {
    WaveCount mc{0u, 0u};
    unsigned n_marched = 0;                                     // rays that entered the march (not skipped as doomed)
    unsigned groups = 0;                                        // groups served
    unsigned long long clk_sum = 0, real_sum = 0;               // shader-clock / 100 MHz ticks spent in groups
};

// Groups of a launch of n_groups that belong to queue (xcd, sub): its items are k = 0 .. that many - 1 (march_queue_group).
__host__ __device__ inline unsigned march_queue_size(unsigned n_groups, unsigned xcd, unsigned sub, unsigned shift) {
    constexpr unsigned Q = 8u * kSubQueues;                     // chunk c belongs to queue c % Q = sub * 8 + xcd
    const unsigned q = sub * 8u + xcd, full = n_groups >> shift, rem = n_groups & ((1u << shift) - 1u);
    return ((full / Q + (full % Q > q ? 1u : 0u)) << shift) + (full % Q == q ? rem : 0u);
}

// Agent-scope relaxed accesses (global_load / global_store ... sc1): the loads bypass this CU's L1, the stores write
// through the XCD's L2 -- how the ray state travels from the wave that marched one segment of a group to the wave, on any
// CU of any XCD, that marches the next (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 payload, the storing wave's
// own vmcnt(0), an sc1 flag; the reader polls the flag with an sc1 load, then loads the payload with sc1 loads only).
template <class T> __device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#ifndef PHOTON_SEG_ACQUIRE
#define PHOTON_SEG_ACQUIRE 0            // 1: an agent-scope acquire (buffer_inv sc1) after the flag poll of every segment start
#endif
static_assert(kLoopMax < (1 << 24) - 1, "completed iterations travel in 24 bits of RayStateDev::ctr");
constexpr int kSegPollMax = 1 << 20;                            // polls (~2 us each) before a wave gives a segment up: the exit every wave reaches
constexpr unsigned kSegDone = 0xffu;                            // seg_flag: every ray of the group has left the volume

// One item of the launch: segment `seg` of 64-ray group `group` -- load the state, march, store it back.  Must be called
// by all 64 lanes of the wave.  With MarchArgs::segments == 1 (seg = 0) this is the whole march of the group.
//
// SEGMENTS (round 4).  A group marches for ~1.9 ms whatever the launch, and a launch ends when its LAST group does: the
// waves finish one by one over the final ~0.8 group times while the rest of the chip idles (measured with the wave-timing
// profile: span - mean end; 1.4 ms of a 8.8 ms launch of one GPU's eighth of the headline job, the same 1.4 ms of the full
// job's 59).  Cutting every march into S segments handed out breadth-first (all first segments, then all second ones, ...)
// makes the quantum smaller and the drain with it (how many pieces, how long: plan_segments); the state a ray carries between segments is the loops' own
// (MarchResume), so the bits do not change.  A segment's wave may have to wait for the wave still marching the previous
// one (only when a launch has fewer groups than the chip holds waves: the host does not segment those): it polls the
// group's flag, bounded -- a wave that gives up counts itself in MarchArgs::error and leaves (march_error_check).
// SEG: this instantiation handles segmented launches (MarchArgs::segments > 1); the whole-march instantiations carry none of
// the resume code -- the trilinear RK4 kernel, which sits on its 96-register budget, spilled 15 VGPRs into its loop with it.
template <int ALGO, int INTERP, bool SAVE, bool NOISE, bool SEG>
__device__ __forceinline__ void march_group(unsigned group, unsigned seg, unsigned n_rays, f4 *tile, WaveTotals &tot) {
    unsigned long long clk0, real0, clk1, real1;
    clock_stamp(clk0, real0);
    const unsigned lane = threadIdx.x & 63u;
    if (tot.groups++ == 0) {                                    // wave-uniform: this wave's first group
        unsigned long long *pf = profile_slot();
        if (pf && lane == 0) {
            atomicMax(&pf[PF_START_NEGMIN], ~real0); atomicAdd(&pf[PF_START_SUM], real0); atomicMax(&pf[PF_START_MAX], real0);
        }
    }
    const unsigned r = group * 64u + lane;
    const bool has_ray = r < n_rays;
    f3 p = mk3(0, 0, 0), d = mk3(0, 0, -1);
    bool marching = has_ray;
    MarchArgsPtr a = march_args();
    MarchResume rs = resume_fresh();
    {
        const bool fresh = !SEG || seg == 0;                    // wave-uniform
        const RayStateDev st = load_arg(&a->st);
        if (!fresh) {
            // the previous segment of this group: handed out before this one, to a wave that is running -- normally long done
            const unsigned want = a->epoch;
            unsigned flag = 0;
            int polls = 0;
            while (true) {
                unsigned f = 0;
                if (lane == 0) f = ld_agent(&st.seg_flag[group]);
                flag = (unsigned)__builtin_amdgcn_readfirstlane((int)f);
                if ((flag >> 8) == want && (flag & 0xffu) >= seg) break;
                if (++polls > kSegPollMax) {                    // never seen; an exit every wave reaches
                    if (lane == 0) atomicAdd(a->error, 1u);
                    return;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if ((flag & 0xffu) == kSegDone) return;             // wave-uniform: no ray of this group is still in the volume
#if PHOTON_SEG_ACQUIRE
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // belt and braces (off: see below)
#endif
            // No agent-scope acquire here: every load of the handed-off words below is an sc1 load (bypasses this CU's L1),
            // every one of them was stored sc1 and drained before the flag, and the flag itself was polled sc1 -- the guide's
            // conditions for leaving the buffer_inv out, checked for exactly this pattern (lines shared between groups, five
            // workgroups per CU, uneven arrivals) by tools/ubench/xcd_handoff.hip: 0 stale words of 7.4e7 with or without it.
            // An acquire per segment start invalidates the L1 under the CU's nineteen other waves' texel blocks.
        }
        if (has_ray) {
            p = mk3(ld_agent(&st.px[r]), ld_agent(&st.py[r]), ld_agent(&st.pz[r]));
            d = mk3(ld_agent(&st.dx[r]), ld_agent(&st.dy[r]), ld_agent(&st.dz[r]));
            if (fresh) {
                marching = !isnan3(p);                          // rays marked dead by raygen_kernel stay out of the march
            } else {
                const unsigned c = ld_agent(&st.ctr[r]);
                marching = (c >> 31) != 0u;
                rs.loop_ctr = (int)(c & 0xffffffu);
                // a ray still marching was written by the previous segment: its word says so (bits 24-30).  Anything else is
                // a STALE word -- the hand-off broken -- and the render must not be returned (march_error_check)
                if (marching && ((c >> 24) & 0x7fu) != ((seg - 1u) & 0x7fu)) atomicAdd(a->error, 1u);
                rs.spins = (int)ld_agent(&st.spins[r]);
                if (INTERP == 1) {
                    const size_t n = st.stride;
                    rs.val_prev = f4{ld_agent(&st.vprev[r]), ld_agent(&st.vprev[n + r]), ld_agent(&st.vprev[2 * n + r]), ld_agent(&st.vprev[3 * n + r])};
                }
            }
        }
        if (SEG) {
            const unsigned b0 = a->seg_begin[seg];              // wave-uniform index: scalar loads from the argument segment
            if (!fresh) { rs.fresh = false; rs.trips_base = b0; }
            if (seg + 1u < a->segments) rs.max_trips = a->seg_begin[seg + 1u] - b0;
        }
        tot.n_marched += fresh ? (unsigned)__popcll(ballot(marching)) : 0u;
    }
    const VolumeDev vol = load_arg(&a->vol);
    const f4 *tex = a->tex;
    GradNoise gn{0, 0.f, 0ull, 0ull};
    if (NOISE) { const NoiseDev nz = load_arg(&a->noise); gn = GradNoise{nz.add_ngrad, nz.ngrad_std, nz.seed, a->ray_base + r}; }
    InterDump idump{nullptr, nullptr, 0, 0, 0u};
    if (SAVE) idump = load_arg(&a->idump);
    idump.ray = r;                                              // chunk-global ray id, like the final dumps
    unsigned long long still;                                   // lanes whose rays are still in the volume when the segment ends
    if (INTERP == 1 && vol.weight_scale > 0.f)                  // kernel-uniform: the texture unit's 8-bit weights (the default) / exact f32
        still = trace_volume_coop<ALGO, INTERP, SAVE, NOISE, true, WaveCount>(marching, p, d, vol, tex, tile, tot.mc, gn, idump, rs);   // all 64 lanes
    else
        still = trace_volume_coop<ALGO, INTERP, SAVE, NOISE, false, WaveCount>(marching, p, d, vol, tex, tile, tot.mc, gn, idump, rs);
    {
        MarchArgsPtr b = march_args();
        const RayStateDev st = load_arg(&b->st);                // loaded again: not carried through the march in SGPRs
        const bool fresh = !SEG || seg == 0, last = !SEG || seg + 1u >= b->segments;      // likewise
        const unsigned group = (unsigned)__builtin_amdgcn_readfirstlane((int)r) >> 6;
        if (last) {
            if (marching) {
                st.px[r] = p.x; st.py[r] = p.y; st.pz[r] = p.z;
                st.dx[r] = d.x; st.dy[r] = d.y; st.dz[r] = d.z;
            }
        } else {
            if (marching) {                                     // where the rays that marched this segment stand now
                st_agent(&st.px[r], p.x); st_agent(&st.py[r], p.y); st_agent(&st.pz[r], p.z);
                st_agent(&st.dx[r], d.x); st_agent(&st.dy[r], d.y); st_agent(&st.dz[r], d.z);
            }
            if (fresh ? has_ray : marching) {
                st_agent(&st.ctr[r], (lane_of(still) ? 0x80000000u : 0u) | ((seg & 0x7fu) << 24) | ((unsigned)rs.loop_ctr & 0xffffffu));
                st_agent(&st.spins[r], (unsigned)rs.spins);
                if (INTERP == 1) {
                    const size_t n = st.stride;
                    st_agent(&st.vprev[r], rs.val_prev.x); st_agent(&st.vprev[n + r], rs.val_prev.y);
                    st_agent(&st.vprev[2 * n + r], rs.val_prev.z); st_agent(&st.vprev[3 * n + r], rs.val_prev.w);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's stores have reached memory ...
            if (lane == 0) st_agent(&st.seg_flag[group], (b->epoch << 8) | (still != 0 ? seg + 1u : kSegDone));      // ... before its flag
        }
    }
    clock_stamp(clk1, real1);
    tot.clk_sum += clk1 - clk0; tot.real_sum += real1 - real0;
}

template <int ALGO, int INTERP, bool SAVE, bool NOISE, bool SEG>
__global__ __launch_bounds__(PHOTON_MARCH_BLOCK, (march_waves<ALGO, INTERP, NOISE>())) void march_kernel(MarchArgs) {
    __shared__ f4 tiles[PHOTON_MARCH_BLOCK / 64][wave_lds_texels<INTERP>()];           // per wave: tile + brick, rows padded (device_volume_coop.hpp)
    f4 *const tile = tiles[threadIdx.x >> 6];
    const unsigned lane = threadIdx.x & 63u;
    WaveTotals tot;
    {
        unsigned long long *pf = profile_slot();
        if (pf && lane == 0) atomicMax(&pf[PF_ENTER_NEGMIN], ~real_time());
    }
#if PHOTON_MARCH_PERSISTENT
    // 32 queues: XCD x (workgroup i runs on XCD i % 8) owns the chunks c (16 or 128 consecutive groups) with c % 8 == x, dealt over its four
    // sub-queues by (c / 8) % 4; one counter per queue, a cache line apart.  A wave serves its home sub-queue, then the
    // other three of its XCD (between them the XCD's waves drain all four: every group is taken), then the same sub-queue
    // of the seven other XCDs (balance at the end of the launch).  How many sub-queues (same box, 1 / 2 / 4 / 8 per XCD):
    //   * few queues = returning atomics to few addresses: a launch whose groups are no work (every ray misses the
    //     volume: the reference's sample BOS case, ~1e6 groups) takes 9.3 / 8.2 / 7.7 / 7.7 ms per call (one-shot: 7.6);
    //   * many queues = each served by few waves, so the eight groups that carry ONE source's rays start further apart in
    //     time, walk the volume at different depths and share fewer L2 lines: HBM traffic of the C3 launch 1.9 / 2.0 /
    //     2.6 / 4.8 GB (one-shot, where a whole generation walks in lockstep: 1.3); the march time does not care (62.1 /
    //     61.9 / 62.3 / 62.1 ms).
    // Taking several groups per access instead of adding queues was tried and dropped: wherever trivial and real groups
    // mix (doomed lens samples of a PIV launch) a wave ends up holding dozens of real groups while the chip drains (C5
    // quarter 38.6 -> 92 ms).
    // A queue of Gq groups hands out Gq x S items, segment-major: item k = segment k / Gq of its (k % Gq)-th group.
    const unsigned home_x = blockIdx.x & 7u, home_s = (blockIdx.x >> 3) & (kSubQueues - 1u);
    for (unsigned step = 0; step < kSubQueues + 7u; step++) {
        const unsigned x = step < kSubQueues ? home_x : ((home_x + step - (kSubQueues - 1u)) & 7u);
        const unsigned sub = step < kSubQueues ? ((home_s + step) & (kSubQueues - 1u)) : home_s;
        while (true) {
            unsigned k = 0;
            if (lane == 0) k = atomicAdd(&march_args()->queue[(sub * 8u + x) * kQueueStride], 1u);
            k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
            const unsigned n_rays = march_args()->n_rays;
            const unsigned shift = march_args()->chunk_shift;
            const unsigned gq = march_queue_size((n_rays + 63u) / 64u, x, sub, shift);
            const unsigned n_seg = SEG ? march_args()->segments : 1u;
            if (k >= gq * n_seg) break;                         // this queue is served (k < 2^26 / 64 * 255: no overflow)
            const unsigned seg = SEG ? k / gq : 0u;
            march_group<ALGO, INTERP, SAVE, NOISE, SEG>(march_queue_group(k - seg * gq, x, sub, shift), seg, n_rays, tile, tot);
            if ((tot.mc.samples | tot.mc.iterations) >> 31) {     // wave-uniform: the 32-bit wave totals go out before they can wrap
                if (lane == 0) {
                    unsigned long long *slot = counter_slot(march_args()->counters);
                    atomicAdd(&slot[CNT_ITER], (unsigned long long)tot.mc.iterations);
                    atomicAdd(&slot[CNT_SAMPLES], (unsigned long long)tot.mc.samples);
                }
                tot.mc.iterations = tot.mc.samples = 0u;
            }
        }
    }
#else
    {                                                           // one-shot grid (A/B builds): one group per wave
        const unsigned n_rays = march_args()->n_rays;
        const unsigned group = xcd_remap(blockIdx.x, gridDim.x) * (PHOTON_MARCH_BLOCK / 64) + (threadIdx.x >> 6);
        if (group < (n_rays + 63u) / 64u) march_group<ALGO, INTERP, SAVE, NOISE, false>(group, 0u, n_rays, tile, tot);
    }
#endif
    if (tot.groups) {                                           // wave-uniform
        unsigned long long *pf = profile_slot();
        if (pf && lane == 0) {                                  // the wave leaves a few queue visits after its last group
            const unsigned long long t = real_time();
            atomicMax(&pf[PF_END_NEGMIN], ~t); atomicAdd(&pf[PF_END_SUM], t); atomicMax(&pf[PF_END_MAX], t);
            atomicAdd(&pf[PF_WAVES], 1ull);
        }
    }
    if (lane == 0) {
        unsigned long long *slot = counter_slot(march_args()->counters);
        if (tot.mc.iterations) atomicAdd(&slot[CNT_ITER], (unsigned long long)tot.mc.iterations);
        if (tot.mc.samples) atomicAdd(&slot[CNT_SAMPLES], (unsigned long long)tot.mc.samples);
        if (tot.n_marched) atomicAdd(&slot[CNT_MARCHED], (unsigned long long)tot.n_marched);
        if (tot.real_sum) {
            atomicAdd(&slot[CNT_CLK], tot.clk_sum);
            atomicAdd(&slot[CNT_REAL], tot.real_sum);
        }
    }
}

// Stage 1b for ray_tracing_algorithm 3, 4 and the reference's no-op default (see march_rays_extra_kernel).
template <int ALGO>
__global__ __launch_bounds__(256) void march_extra_kernel(VolumeDev vol, unsigned n_rays, RayStateDev st,
                                                          unsigned long long *__restrict__ counters) {
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    MarchCount mc{0, 0};
    unsigned marched = 0;
    if (r < n_rays) {
        f3 p = mk3(st.px[r], st.py[r], st.pz[r]);
        f3 d = mk3(st.dx[r], st.dy[r], st.dz[r]);
        if (!isnan3(p)) {
            marched = 1;
            trace_volume_extra<ALGO>(p, d, vol, mc);
            st.px[r] = p.x; st.py[r] = p.y; st.pz[r] = p.z;
            st.dx[r] = d.x; st.dy[r] = d.y; st.dz[r] = d.z;
        }
    }
    wave_add(&counter_slot(counters)[CNT_ITER], (unsigned long long)mc.iterations);
    wave_add(&counter_slot(counters)[CNT_SAMPLES], (unsigned long long)mc.samples);
    wave_add(&counter_slot(counters)[CNT_MARCHED], (unsigned long long)marched);
}

// Stage 2: everything after the volume (parallel_ray_tracing.cu:2136-2241).  FROM_STATE=false
// (no density gradients) generates the ray in place, so that path is one fused kernel.
#ifndef PHOTON_SPLIT_SENSOR
#define PHOTON_SPLIT_SENSOR 1
#endif
#ifndef PHOTON_SENSOR_WAVES
#define PHOTON_SENSOR_WAVES 5           // the cooperative splats park 8 KiB per wave in LDS: five blocks per CU
#endif
template <bool FROM_STATE, bool TRAIN, bool SPLIT>
__global__ __launch_bounds__(256, PHOTON_SENSOR_WAVES) void sensor_kernel(SceneDev sc, long long src_begin, unsigned n_rays, RayStateDev st,
                                                     double *image, DumpDev dump, unsigned long long *counters) {
    __shared__ SplatLds splat_lds[4];                                   // per wave: the parked rays of the cooperative splats
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    int taps = 0;
    unsigned on_sensor = 0;
    SplatReq req;                                                       // erf splat, done wave-cooperatively below
    req.valid = false;
    req.X = req.Y = req.D = req.rfD = 0.f; req.scale = 0.0; req.c0 = req.c1 = req.r0 = req.r1 = 0;
    TapReq tap;                                                         // 4-pixel splat, likewise
    tap.valid = false;
    tap.ii_ul = tap.jj_ul = 0; tap.inc[0] = tap.inc[1] = tap.inc[2] = tap.inc[3] = 0.f;
    if (r < n_rays) {
        Ray ray;
        bool alive = true;
        int source, local_ray;
        slot_to_ray(sc, src_begin, n_rays, r, source, local_ray);
        if (FROM_STATE) {                                              // back to the camera frame (.cu:2100-2122)
            f3 p = mk3(st.px[r], st.py[r], st.pz[r]);
            f3 d = mk3(st.dx[r], st.dy[r], st.dz[r]);
            p = matvec(sc.cam.rotation_matrix, p);
            d = normalize(matvec(sc.cam.rotation_matrix, d));
            p.z = (float)(p.z + (sc.z_offset + 750e3));                 // .cu:2119
            ray.pos = p;
            ray.dir = d;
            ray.radiance = st.radiance[r];
            ray.wavelength = sc.beam_wavelength;
            alive = !(isnan3(ray.dir) || isnan3(ray.pos));              // .cu:2125-2129
        } else {
            ray = generate_ray(sc, source, local_ray);
        }
        const bool dumping = dump.final_pos != nullptr && r < (unsigned)dump.num_save;
        // the ray's identity for the noise generator: independent of the launch order
        const unsigned long long ray_id = (unsigned long long)(sc.source_base + source) * (unsigned)sc.rays_per_source + (unsigned)local_ray;
        f3 fin = nan3();
        bool have_fin = false;
        if (alive) {
            if (dumping) {                                              // .cu:2136-2141
                dump.final_dir[3 * r] = ray.dir.x; dump.final_dir[3 * r + 1] = ray.dir.y;
                dump.final_dir[3 * r + 2] = ray.dir.z;
            }
            if (sc.elems[0].element_type == 'n') {                      // .cu:2143-2158
                const float z_obj = sc.object_distance + sc.z_offset;
                fin = apparent_image(ray, sc.cam, z_obj, sc.z_offset, sc.elems[0], req, sc.noise, ray_id);
                have_fin = true;
                on_sensor = !isnan(fin.x);
            } else {
                ray = optical_system<TRAIN>(sc, ray);
                if (!(isnan3(ray.dir) || isnan3(ray.pos))) {            // .cu:2172-2176
                    if (sc.cam.implement_diffraction) {
                        fin = sensor_diffraction(ray, sc.cam, req, sc.noise, ray_id);
                        have_fin = true;
                        on_sensor = !isnan(fin.x);
                    } else {
                        fin = sensor_bilinear(ray, sc.cam, tap, sc.noise, ray_id);
                        have_fin = !(isnan(fin.x) || isnan(fin.y));     // .cu:2196
                        on_sensor = have_fin;
                    }
                }
            }
        }
        if (dumping && have_fin) {
            dump.final_pos[3 * r] = fin.x; dump.final_pos[3 * r + 1] = fin.y; dump.final_pos[3 * r + 2] = fin.z;
        }
    }
    if (SPLIT) {
        // hand the erf splat to splat_kernel through the (now consumed) state arrays: the optics above and the
        // wave-cooperative splat below each want the register file to themselves
        if (r < n_rays) {
            st.px[r] = req.X; st.py[r] = req.Y; st.pz[r] = req.valid ? req.D : -1.f; st.dx[r] = req.rfD;
            st.radiance[r] = req.scale;
        }
    } else {
        taps += erf_splat_wave(image, sc.cam.x_pixel_number, sc.cam.y_pixel_number, req, splat_lds[threadIdx.x >> 6]);  // all 64 lanes
    }
    taps += bilinear_splat_wave(image, sc.cam.x_pixel_number, sc.cam.y_pixel_number, tap, splat_lds[threadIdx.x >> 6]);     // all 64 lanes
    wave_add(&counter_slot(counters)[CNT_TAPS], (unsigned long long)taps);
    wave_add(&counter_slot(counters)[CNT_ON_SENSOR], (unsigned long long)on_sensor);
}

// Second half of the sensor stage for erf splats coming from the march (sensor_kernel<.., SPLIT=true>).
__global__ __launch_bounds__(256, PHOTON_SENSOR_WAVES) void splat_kernel(unsigned n_rays, RayStateDev st, double *image, int W, int H,
                                                                         unsigned long long *counters) {
    __shared__ SplatLds splat_lds[4];
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    SplatReq req;
    req.valid = false;
    req.X = req.Y = req.D = req.rfD = 0.f; req.scale = 0.0; req.c0 = req.c1 = req.r0 = req.r1 = 0;
    if (r < n_rays) {
        const float D = st.pz[r];
        if (D >= 0.f) {
            req.valid = true;
            req.X = st.px[r]; req.Y = st.py[r]; req.D = D; req.rfD = st.dx[r];
            req.scale = st.radiance[r];
            req.c0 = (int)floorf(req.X - req.rfD); req.c1 = (int)ceilf(req.X + req.rfD);     // erf_splat_prepare's window
            req.r0 = (int)floorf(req.Y - req.rfD); req.r1 = (int)ceilf(req.Y + req.rfD);
        }
    }
    const int taps = erf_splat_wave(image, W, H, req, splat_lds[threadIdx.x >> 6]);     // all 64 lanes
    wave_add(&counter_slot(counters)[CNT_TAPS], (unsigned long long)taps);
}

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

// image_array is read-modify-write (parallel_ray_tracing.cu:3309,3675): fold the f64 accumulator of
// this call into the caller's f32 image, one rounding per pixel.
__global__ __launch_bounds__(256) void finalize_image_kernel(float *__restrict__ image, const double *__restrict__ acc,
                                                             size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) image[i] = (float)((double)image[i] + acc[i]);
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

// =============================================================================================
// host: handles
// =============================================================================================
struct photon_volume {
    float grad_max = 0.f;               // largest |grad n| of the texels (per micron)
    VolumeDev dev{};
    photon_volume_info_t info{};
    f4 *d_texels = nullptr;
    f4 *d_coeffs = nullptr;
};

struct photon_sources {                 // light-field sources generated in HBM (SoA, like lightfield_source_t)
    long long n = 0;
    float *x = nullptr, *y = nullptr, *z = nullptr;
    double *radiance = nullptr;
    int *diameter_index = nullptr;
};

struct PermEntry { long long begin = -1, end = -1; int *d_perm = nullptr; size_t capacity = 0; unsigned long long stamp = 0; };

struct photon_scene {
    SceneDev dev{};
    std::vector<void *> allocs;         // device buffers owned by the scene
    RayStateDev ws{};                   // march -> sensor state, grown on demand
    size_t ws_rays = 0;
    unsigned long long *d_counters = nullptr;
    unsigned *d_queue = nullptr;        // the march's work queues: room for 64 counters a cache line apart, 8 XCDs x kSubQueues (4) in use
    int num_cus = 256;                  // compute units of the scene's device (size of the persistent march grid)
    unsigned *d_error = nullptr;        // march waves that gave a segment up (march_error_check)
    unsigned march_epoch = 0;           // tag of the last segmented march launch in ws.seg_flag
    int march_segments = -1;            // photon_scene_set_march_segments: -1 the library's choice, 1 whole marches, n segments
    unsigned long long *d_profile = nullptr;    // wave-timing slots of the march launches (photon_scene_set_march_profile), or nullptr
    unsigned prof_next = 0;             // march launches since the slots were last zeroed
    double *d_acc = nullptr;            // f64 sensor accumulator, W*H
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // statistics window (photon_scene_stats_begin / _end): traces inside it record their events and leave the counters
    // running instead of synchronising per call -- a timed loop then has no host sync and no D2H copy inside it
    bool win_open = false;
    std::vector<hipEvent_t> win_events;        // created on demand, reused by the next window
    size_t win_used = 0;
    std::vector<std::pair<size_t, size_t>> win_march, win_total;      // (begin, end) event indices
    uint64_t win_rays = 0;
    uint32_t win_traces = 0;
    bool win_have_volume = false;
    hipStream_t win_stream = nullptr;   // the stream the window was opened on: its traces must run there (the counters were zeroed there)
    int ray_order_mode = 2;             // 0 source-major, 1 lens-major, 2 auto (photon_scene_set_ray_order)
    bool skip_doomed = true;            // photon_scene_set_skip_doomed
    float lens_z = 0.f;                 // element 0's centre, for the auto rule
    PermEntry perms[4];                 // spatial (Morton) orders of the lens-major launch ranges seen last
    unsigned long long perm_clock = 0;
    photon_sort_scratch sort_scratch;   // keys / indices / radix-sort temporaries, grown on demand (photon_sort.hip)
};

static void free_resume_state(photon_scene *s);
static int march_error_check(photon_scene *scene);

template <typename T>
static int upload(photon_scene *s, const T *host, size_t n, const T **dev_out) {
    T *d = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    PH_CHECK(pool_malloc((void **)&d, bytes));
    s->allocs.push_back(d);
    if (n) PH_CHECK(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
    *dev_out = d;
    return 0;
}

template <typename T>
static int copy_device(photon_scene *s, const T *dev_src, size_t n, const T **dev_out) {
    T *d = nullptr;
    PH_CHECK(pool_malloc((void **)&d, std::max<size_t>(n, 1) * sizeof(T)));
    s->allocs.push_back(d);
    if (n) PH_CHECK(hipMemcpy(d, dev_src, n * sizeof(T), hipMemcpyDeviceToDevice));
    *dev_out = d;
    return 0;
}

// glibc rand()/srand() sequence (TYPE_3 additive-feedback generator r[i] = r[i-3] + r[i-31]),
// re-implemented so the lens-sample table of parallel_ray_tracing.cu:3228-3235 is reproduced
// without touching the caller's process-wide rand() state.
static void glibc_rand_sequence(unsigned seed, int count, std::vector<int> &out) {
    std::vector<int32_t> r(344 + count);
    r[0] = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
        const int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
        int64_t word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        r[i] = (int32_t)word;
    }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
    for (int i = 34; i < 344 + count; i++) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
    out.resize(count);
    for (int i = 0; i < count; i++) out[i] = (int)((uint32_t)r[344 + i] >> 1);
}

static bool parse_nrrd(const char *path, std::vector<float> &rho, int dims[3], double spacing[3], double origin[3],
                       std::string &why) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { why = "cannot open file"; return false; }
    std::string line;
    if (!std::getline(f, line) || line.rfind("NRRD", 0) != 0) { why = "missing NRRD magic"; return false; }
    std::string type, encoding = "raw", endian = "little";
    int dimension = 0;
    bool sizes_ok = false;
    for (int a = 0; a < 3; a++) { spacing[a] = 1.0; origin[a] = 0.0; }
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) break;                        // blank line ends the header
        if (line[0] == '#') continue;
        const size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        const std::string key = line.substr(0, colon);
        size_t vs = colon + 1;
        if (vs < line.size() && line[vs] == '=') vs++;  // "key:=value" pairs
        while (vs < line.size() && line[vs] == ' ') vs++;
        const std::string val = line.substr(vs);
        if (key == "type") type = val;
        else if (key == "dimension") dimension = atoi(val.c_str());
        else if (key == "encoding") encoding = val;
        else if (key == "endian") endian = val;
        else if (key == "sizes") sizes_ok = sscanf(val.c_str(), "%d %d %d", &dims[0], &dims[1], &dims[2]) == 3;
        else if (key == "spacings") sscanf(val.c_str(), "%lf %lf %lf", &spacing[0], &spacing[1], &spacing[2]);
        else if (key == "space origin") sscanf(val.c_str(), " (%lf,%lf,%lf)", &origin[0], &origin[1], &origin[2]);
        else if (key == "space directions") {
            double m[9];
            if (sscanf(val.c_str(), " (%lf,%lf,%lf) (%lf,%lf,%lf) (%lf,%lf,%lf)", &m[0], &m[1], &m[2], &m[3], &m[4],
                       &m[5], &m[6], &m[7], &m[8]) == 9)
                for (int a = 0; a < 3; a++)
                    spacing[a] = std::sqrt(m[3 * a] * m[3 * a] + m[3 * a + 1] * m[3 * a + 1] + m[3 * a + 2] * m[3 * a + 2]);
        }
    }
    if (dimension != 3 || !sizes_ok) { why = "need dimension 3 with three sizes"; return false; }
    if (type != "float") { why = "type must be float (single precision)"; return false; }
    if (encoding != "raw" || endian != "little") { why = "only raw little-endian encoding is supported"; return false; }
    if (dims[0] < 3 || dims[1] < 3 || dims[2] < 3) { why = "each axis needs at least 3 samples"; return false; }
    // a corrupt header must not drive the allocation: the payload has to be in the file
    if (dims[0] > 65536 || dims[1] > 65536 || dims[2] > 65536) { why = "sizes beyond 65536 per axis"; return false; }
    const unsigned long long count = (unsigned long long)dims[0] * dims[1] * dims[2];
    const std::streamoff here = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streamoff total = f.tellg();
    f.seekg(here, std::ios::beg);
    if (here < 0 || total < here || (unsigned long long)(total - here) < count * sizeof(float)) { why = "payload shorter than sizes"; return false; }
    rho.resize((size_t)count);
    f.read(reinterpret_cast<char *>(rho.data()), (std::streamsize)(rho.size() * sizeof(float)));
    if ((size_t)f.gcount() != rho.size() * sizeof(float)) { why = "payload shorter than sizes"; return false; }
    return true;
}

// =============================================================================================
// C-ABI: extension entry points
// =============================================================================================
extern "C" {

const char *photon_version(void) { return "photon-amd 0.4 (gfx950, HIP)"; }

int photon_set_device(int device) {
    PH_CHECK(hipSetDevice(device));
    return 0;
}

int photon_device_pci_bus_id(char *buf, int len) {
    if (!buf || len < 16) return 1;
    int dev = 0;
    PH_CHECK(hipGetDevice(&dev));
    PH_CHECK(hipDeviceGetPCIBusId(buf, len, dev));
    return 0;
}

int photon_rand_table(int n, float *r1, float *r2) {
    if (n < 0) return 1;
    std::vector<int> seq;
    glibc_rand_sequence(10u, 2 * n, seq);
    for (int k = 0; k < n; k++) {                       // RAND_MAX = 2147483647
        r1[k] = (float)((double)seq[2 * k] / 2147483647);
        r2[k] = (float)((double)seq[2 * k + 1] / 2147483647);
    }
    return 0;
}

void photon_volume_free(photon_volume_t *vol) {
    if (!vol) return;
    if (vol->d_texels) (void)hipFree(vol->d_texels);
    if (vol->d_coeffs) (void)hipFree(vol->d_coeffs);
    delete vol;
}

// Where the density comes from: a host array (NRRD / caller) or a field evaluated on the device
// (photon_volume_gaussian): rho0 + amp * gz[k] * (gy[j] * gx[i]) from three device-resident axis profiles.
struct DensitySource {
    const float *host_rho = nullptr;
    const double *d_gx = nullptr, *d_gy = nullptr, *d_gz = nullptr;
    double rho0 = 0, amp = 0;
};

static int volume_build(const DensitySource &src, int nx, int ny, int nz, const double spacing[3],
                        const double origin[3], int interpolation, photon_volume_t **out);

int photon_volume_from_density(const float *rho, int nx, int ny, int nz, const double spacing[3],
                               const double origin[3], int interpolation, photon_volume_t **out) {
    if (!rho) {
        fprintf(stderr, "photon: photon_volume_from_density: bad arguments\n");
        return 1;
    }
    DensitySource src;
    src.host_rho = rho;
    return guarded("photon_volume_from_density", [&]() -> int { return volume_build(src, nx, ny, nz, spacing, origin, interpolation, out); });
}

// Synthetic density field evaluated on the device: rho = rho0 + amp * exp(-|r - centre|^2 / (2 sigma^2)),
// separable, so the host prepares three axis profiles (O(n) work, photon_det_exp) and a kernel fills the
// n^3 grid in HBM -- no host array, no file, no upload (BASELINE C3 / C4's volume).
// the three axis profiles of the separable Gaussian, on the device
static int gaussian_profiles(int nx, int ny, int nz, const double spacing[3], const double origin[3],
                             const double centre[3], double sigma, double *d_prof[3]) {
    const int dims[3] = {nx, ny, nz};
    for (int a = 0; a < 3; a++) {
        std::vector<double> prof(dims[a]);
        for (int i = 0; i < dims[a]; i++) {
            const double x = origin[a] + spacing[a] * (double)i;
            prof[i] = photon_det_exp(-((x - centre[a]) * (x - centre[a])) / (2 * (sigma * sigma)));
        }
        if (hipMalloc((void **)&d_prof[a], dims[a] * sizeof(double)) != hipSuccess ||
            hipMemcpy(d_prof[a], prof.data(), dims[a] * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 3;
    }
    return 0;
}

int photon_volume_gaussian(int nx, int ny, int nz, const double spacing[3], const double origin[3], double rho0,
                           double amp, const double centre[3], double sigma, int interpolation,
                           photon_volume_t **out) {
    if (!spacing || !origin || !centre || !(sigma > 0) || nx < 3 || ny < 3 || nz < 3) {
        fprintf(stderr, "photon: photon_volume_gaussian: bad arguments\n");
        return 1;
    }
    double *d_prof[3] = {nullptr, nullptr, nullptr};
    int rc = guarded("photon_volume_gaussian", [&]() -> int { return gaussian_profiles(nx, ny, nz, spacing, origin, centre, sigma, d_prof); });
    if (!rc) {
        DensitySource src;
        src.d_gx = d_prof[0]; src.d_gy = d_prof[1]; src.d_gz = d_prof[2];
        src.rho0 = rho0; src.amp = amp;
        rc = guarded("photon_volume_gaussian", [&]() -> int { return volume_build(src, nx, ny, nz, spacing, origin, interpolation, out); });
    } else {
        fprintf(stderr, "photon: photon_volume_gaussian: device allocation failed\n");
    }
    for (double *p : d_prof) if (p) (void)hipFree(p);
    return rc;
}

// The same field written as an NRRD file (what nrrd_functions.py:14-57 writes with pynrrd and loadNRRD reads
// back: type float, dimension 3, raw, little endian, sizes / spacings / space origin): evaluated on the device,
// streamed to disk.  For feeding synthetic volumes to code that wants a file -- photon itself included.
int photon_density_gaussian_write_nrrd(const char *path, int nx, int ny, int nz, const double spacing[3],
                                       const double origin[3], double rho0, double amp, const double centre[3],
                                       double sigma) {
    if (!path || !spacing || !origin || !centre || !(sigma > 0) || nx < 1 || ny < 1 || nz < 1) {
        fprintf(stderr, "photon: photon_density_gaussian_write_nrrd: bad arguments\n");
        return 1;
    }
    double *d_prof[3] = {nullptr, nullptr, nullptr};
    float *d_rho = nullptr;
    const size_t n = (size_t)nx * ny * nz;
    std::vector<float> rho;
    int rc = guarded("photon_density_gaussian_write_nrrd", [&]() -> int {
        rho.resize(n);
        return gaussian_profiles(nx, ny, nz, spacing, origin, centre, sigma, d_prof);
    });
    if (!rc && hipMalloc((void **)&d_rho, n * sizeof(float)) != hipSuccess) rc = 3;
    if (!rc) {
        hipLaunchKernelGGL(separable_density_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_prof[0], d_prof[1],
                           d_prof[2], nx, ny, nz, rho0, amp, d_rho);
        if (hipGetLastError() != hipSuccess || hipMemcpy(rho.data(), d_rho, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = 4;
    }
    for (double *p : d_prof) if (p) (void)hipFree(p);
    if (d_rho) (void)hipFree(d_rho);
    if (rc) {
        fprintf(stderr, "photon: photon_density_gaussian_write_nrrd: HIP error\n");
        return rc;
    }
    std::ofstream f(path, std::ios::out | std::ios::binary);
    if (!f) { fprintf(stderr, "photon: cannot write %s\n", path); return 2; }
    char header[512];
    snprintf(header, sizeof header,
             "NRRD0005\n# written by photon_density_gaussian_write_nrrd\ntype: float\ndimension: 3\nspace: 3D-left-handed\n"
             "sizes: %d %d %d\nendian: little\nencoding: raw\nspacings: %.17g %.17g %.17g\nspace origin: (%.17g,%.17g,%.17g)\n\n",
             nx, ny, nz, spacing[0], spacing[1], spacing[2], origin[0], origin[1], origin[2]);
    f.write(header, (std::streamsize)strlen(header));
    f.write(reinterpret_cast<const char *>(rho.data()), (std::streamsize)(n * sizeof(float)));
    return f ? 0 : 2;
}

static int volume_build(const DensitySource &src, int nx, int ny, int nz, const double spacing[3],
                        const double origin[3], int interpolation, photon_volume_t **out) {
    if (!out || !spacing || !origin || nx < 3 || ny < 3 || nz < 3 || (interpolation != 1 && interpolation != 2)) {
        fprintf(stderr, "photon: volume: bad arguments\n");
        return 1;
    }
    // bounds from the file's own size (loadNRRD, .h:1696-1706), then the 1024-slice cap (.h:1714-1717)
    const double xmin = origin[0], ymin = origin[1], zmin = origin[2] - 750e3;
    const double xmax = xmin + (nx - 1) * spacing[0], ymax = ymin + (ny - 1) * spacing[1];
    const double zmax = zmin + (nz - 1) * spacing[2];
    if (nz > 1024) nz = 1024;
    if ((unsigned long long)(nx + 1) * (ny + 1) * (nz + 1) >= (1ull << 31)) {
        fprintf(stderr, "photon: volume of %d x %d x %d texels exceeds the 2^31-texel limit of the samplers\n", nx, ny, nz);
        return 1;
    }
    photon_volume *v = new photon_volume();
    const size_t n = (size_t)nx * ny * nz;
    float *d_rho = nullptr, *d_min = nullptr;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    auto fail = [&](int code) { if (d_rho) (void)hipFree(d_rho); if (d_min) (void)hipFree(d_min); photon_volume_free(v); return code; };
#define PH_VCHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { fprintf(stderr, "photon: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return fail((int)_e); } } while (0)
    PH_VCHECK(hipMalloc((void **)&v->d_texels, n * sizeof(f4)));
    PH_VCHECK(hipMalloc((void **)&d_rho, n * sizeof(float)));
    PH_VCHECK(hipMalloc((void **)&d_min, 2 * (size_t)blocks * sizeof(float)));      // block minima of n-1 | block maxima of |grad n|
    if (src.host_rho) {
        PH_VCHECK(hipMemcpy(d_rho, src.host_rho, n * sizeof(float), hipMemcpyHostToDevice));
    } else {
        hipLaunchKernelGGL(separable_density_kernel, dim3(blocks), dim3(256), 0, 0, src.d_gx, src.d_gy, src.d_gz, nx, ny, nz,
                           src.rho0, src.amp, d_rho);
        PH_VCHECK(hipGetLastError());
    }
    const float gx = (float)spacing[0], gy = (float)spacing[1], gz = (float)spacing[2];
    hipLaunchKernelGGL(build_volume_kernel, dim3(blocks), dim3(256), 0, 0, d_rho, nx, ny, nz, gx, gy, gz, v->d_texels,
                       d_min);
    PH_VCHECK(hipGetLastError());
    std::vector<float> mins(2 * (size_t)blocks);
    PH_VCHECK(hipMemcpy(mins.data(), d_min, mins.size() * sizeof(float), hipMemcpyDeviceToHost));
    float data_min = FLT_MAX, grad_max = 0.f;
    for (unsigned k = 0; k < blocks; k++) {
        if (mins[k] < data_min) data_min = mins[k];
        if (mins[blocks + k] > grad_max) grad_max = mins[blocks + k];
    }
    v->grad_max = grad_max;
    if (interpolation == 2) {
        PH_VCHECK(hipMalloc((void **)&v->d_coeffs, n * sizeof(f4)));
        PH_VCHECK(hipMemcpy(v->d_coeffs, v->d_texels, n * sizeof(f4), hipMemcpyDeviceToDevice));
        const size_t sx = 1, sy = (size_t)nx, sz = (size_t)nx * ny;
        auto nblk = [](size_t lines) { return dim3((unsigned)((lines + 255) / 256)); };
        // x lines: (y inner, z outer); y lines: (x inner, z outer); z lines: (x inner, y outer)
        hipLaunchKernelGGL(prefilter_lines_kernel, nblk((size_t)ny * nz), dim3(256), 0, 0, v->d_coeffs, nx, sx, ny, sy, nz, sz);
        hipLaunchKernelGGL(prefilter_lines_kernel, nblk((size_t)nx * nz), dim3(256), 0, 0, v->d_coeffs, ny, sy, nx, sx, nz, sz);
        hipLaunchKernelGGL(prefilter_lines_kernel, nblk((size_t)nx * ny), dim3(256), 0, 0, v->d_coeffs, nz, sz, nx, sx, ny, sy);
        PH_VCHECK(hipGetLastError());
    }
    PH_VCHECK(hipDeviceSynchronize());
    (void)hipFree(d_rho); d_rho = nullptr;
    (void)hipFree(d_min); d_min = nullptr;
#undef PH_VCHECK
    float step = (float)fmin(spacing[0], spacing[1]);                   // .h:2086-2098
    step = step < spacing[2] ? step : (float)spacing[2];
    VolumeDev &d = v->dev;
    d.min_bound = f3{(float)xmin, (float)ymin, (float)zmin};
    d.max_bound = f3{(float)xmax, (float)ymax, (float)zmax};
    d.nx = nx; d.ny = ny; d.nz = nz;
    d.step_size = step;
    d.data_min = data_min;
    d.interpolation = interpolation;
    d.weight_inv = 1.0f / 256.f;
    d.weight_scale = 256.f;             // trilinear weights as the reference's texture unit holds them (photon_volume_set_weight_bits)
    d.weight_fast = (double)std::max(nx, std::max(ny, nz)) * 256.0 <= 2097152.0;
    d.texels = v->d_texels;
    d.coeffs = v->d_coeffs;
    photon_volume_info_t &info = v->info;
    info.min_bound[0] = d.min_bound.x; info.min_bound[1] = d.min_bound.y; info.min_bound[2] = d.min_bound.z;
    info.max_bound[0] = d.max_bound.x; info.max_bound[1] = d.max_bound.y; info.max_bound[2] = d.max_bound.z;
    info.nx = nx; info.ny = ny; info.nz = nz;
    info.grid_spacing[0] = gx; info.grid_spacing[1] = gy; info.grid_spacing[2] = gz;
    info.step_size = step; info.data_min = data_min; info.interpolation = interpolation;
    *out = v;
    return 0;
}

int photon_volume_load_nrrd(const char *path, int interpolation, photon_volume_t **out) {
  return guarded("photon_volume_load_nrrd", [&]() -> int {
    if (!path || !out) { fprintf(stderr, "photon: photon_volume_load_nrrd: null argument\n"); return 1; }
    std::vector<float> rho;
    int dims[3];
    double spacing[3], origin[3];
    std::string why;
    if (!parse_nrrd(path, rho, dims, spacing, origin, why)) {
        fprintf(stderr, "photon: failed to read NRRD \"%s\": %s\n", path ? path : "(null)", why.c_str());
        return 2;
    }
    if (verbose())
        printf("photon: NRRD %s  sizes %d %d %d  spacings %g %g %g  origin (%g,%g,%g)\n", path, dims[0], dims[1], dims[2],
               spacing[0], spacing[1], spacing[2], origin[0], origin[1], origin[2]);
    return photon_volume_from_density(rho.data(), dims[0], dims[1], dims[2], spacing, origin, interpolation, out);
  });
}

int photon_volume_set_weight_bits(photon_volume_t *vol, int bits) {
    if (!vol || bits < 0 || bits > 23) return 1;
    vol->dev.weight_scale = bits ? (float)(1 << bits) : 0.f;
    vol->dev.weight_inv = bits ? 1.0f / (float)(1 << bits) : 0.f;
    vol->dev.weight_fast = bits && (double)std::max(vol->dev.nx, std::max(vol->dev.ny, vol->dev.nz)) * (double)(1 << bits) <= 2097152.0;
    return 0;
}

int photon_volume_info(const photon_volume_t *vol, photon_volume_info_t *info) {
    if (!vol || !info) return 1;
    *info = vol->info;
    return 0;
}

int photon_volume_download(const photon_volume_t *vol, int coefficients, float *out) {
    if (!vol || !out) return 1;
    const f4 *src = (coefficients && vol->d_coeffs) ? vol->d_coeffs : vol->d_texels;
    const size_t n = (size_t)vol->dev.nx * vol->dev.ny * vol->dev.nz;
    PH_CHECK(hipMemcpy(out, src, n * sizeof(f4), hipMemcpyDeviceToHost));
    return 0;
}

int photon_volume_sample(const photon_volume_t *vol, int n, const float *coords, float *out) {
    if (!vol || n < 0) return 1;
    if (n == 0) return 0;
    DeviceBuffer<float> d_c, d_o;                       // freed on every return path
    PH_CHECK(d_c.alloc((size_t)n * 3));
    PH_CHECK(d_o.alloc((size_t)n * 4));
    PH_CHECK(hipMemcpy(d_c.p, coords, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sample_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, vol->dev, n, d_c.p, d_o.p);
    PH_CHECK(hipGetLastError());
    PH_CHECK(hipMemcpy(out, d_o.p, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int photon_trace_volume_rays(const photon_volume_t *vol, int ray_tracing_algorithm, int n, float *pos, float *dir,
                             int *steps) {
    if (!vol || n < 0) {
        fprintf(stderr, "photon: photon_trace_volume_rays: bad arguments\n");
        return 1;
    }
    if (n == 0) return 0;
    DeviceBuffer<float> d_p, d_d;                       // freed on every return path
    DeviceBuffer<int> d_s;
    const size_t b3 = (size_t)n * 3 * sizeof(float);
    PH_CHECK(d_p.alloc((size_t)n * 3));
    PH_CHECK(d_d.alloc((size_t)n * 3));
    PH_CHECK(d_s.alloc((size_t)n));
    PH_CHECK(hipMemcpy(d_p.p, pos, b3, hipMemcpyHostToDevice));
    PH_CHECK(hipMemcpy(d_d.p, dir, b3, hipMemcpyHostToDevice));
    const dim3 grid((n + 255) / 256), block(256);
    const int interp = vol->dev.interpolation;
    const f4 *tex = interp == 2 ? vol->d_coeffs : vol->d_texels;
    if (ray_tracing_algorithm == 3) hipLaunchKernelGGL((march_rays_extra_kernel<3>), grid, block, 0, 0, vol->dev, n, d_p.p, d_d.p, d_s.p);
    else if (ray_tracing_algorithm == 4) hipLaunchKernelGGL((march_rays_extra_kernel<4>), grid, block, 0, 0, vol->dev, n, d_p.p, d_d.p, d_s.p);
    else if (ray_tracing_algorithm != 1 && ray_tracing_algorithm != 2) hipLaunchKernelGGL((march_rays_extra_kernel<0>), grid, block, 0, 0, vol->dev, n, d_p.p, d_d.p, d_s.p);
    else if (ray_tracing_algorithm == 1 && interp == 1) hipLaunchKernelGGL((march_rays_kernel<1, 1>), grid, block, 0, 0, vol->dev, tex, n, d_p.p, d_d.p, d_s.p);
    else if (ray_tracing_algorithm == 1) hipLaunchKernelGGL((march_rays_kernel<1, 2>), grid, block, 0, 0, vol->dev, tex, n, d_p.p, d_d.p, d_s.p);
    else if (interp == 1) hipLaunchKernelGGL((march_rays_kernel<2, 1>), grid, block, 0, 0, vol->dev, tex, n, d_p.p, d_d.p, d_s.p);
    else hipLaunchKernelGGL((march_rays_kernel<2, 2>), grid, block, 0, 0, vol->dev, tex, n, d_p.p, d_d.p, d_s.p);
    PH_CHECK(hipGetLastError());
    PH_CHECK(hipMemcpy(pos, d_p.p, b3, hipMemcpyDeviceToHost));
    PH_CHECK(hipMemcpy(dir, d_d.p, b3, hipMemcpyDeviceToHost));
    if (steps) PH_CHECK(hipMemcpy(steps, d_s.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

void photon_scene_free(photon_scene_t *s) {
    if (!s) return;
    for (void *p : s->allocs) pool_free(p);
    pool_free(s->ws.px);
    pool_free(s->ws.radiance);
    free_resume_state(s);
    pool_free(s->d_counters);
    pool_free(s->d_queue);
    pool_free(s->d_profile);
    pool_free(s->d_error);
    pool_free(s->d_acc);
    for (auto &p : s->perms) pool_free(p.d_perm);
    photon_sort_scratch_free(&s->sort_scratch);
    for (auto &e : s->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : s->win_events) if (e) (void)hipEventDestroy(e);
    delete s;
}

// ---------------------------------------------------------------------------------------------
// light-field sources generated on the device
// ---------------------------------------------------------------------------------------------
void photon_sources_free(photon_sources_t *src) {
    if (!src) return;
    if (src->x) (void)hipFree(src->x);
    if (src->y) (void)hipFree(src->y);
    if (src->z) (void)hipFree(src->z);
    if (src->radiance) (void)hipFree(src->radiance);
    if (src->diameter_index) (void)hipFree(src->diameter_index);
    delete src;
}

static int sources_alloc(long long n, photon_sources **out) {
    photon_sources *src = new photon_sources();
    src->n = n;
    const size_t m = (size_t)std::max<long long>(n, 1);
    if (hipMalloc((void **)&src->x, m * sizeof(float)) != hipSuccess || hipMalloc((void **)&src->y, m * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&src->z, m * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&src->radiance, m * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&src->diameter_index, m * sizeof(int)) != hipSuccess) {
        fprintf(stderr, "photon: sources: device allocation failed\n");
        photon_sources_free(src);
        return 3;
    }
    *out = src;
    return 0;
}

int photon_sources_bos(const double *dot_x, const double *dot_y, int n_dots, const double *tmpl_x, const double *tmpl_y,
                       int n_tmpl, double z, double radiance, photon_sources_t **out) {
    if (!out || n_dots < 0 || n_tmpl < 1 || (n_dots && (!dot_x || !dot_y)) || !tmpl_x || !tmpl_y ||
        (long long)n_dots * n_tmpl > 0x7fffffffLL) {
        fprintf(stderr, "photon: photon_sources_bos: bad arguments\n");
        return 1;
    }
    const long long n = (long long)n_dots * n_tmpl;
    photon_sources *src = nullptr;
    int rc = sources_alloc(n, &src);
    if (rc) return rc;
    double *d_in = nullptr;                                             // dot_x | dot_y | tmpl_x | tmpl_y
    const size_t total = 2 * (size_t)n_dots + 2 * (size_t)n_tmpl;
    auto fail = [&](int code) { if (d_in) (void)hipFree(d_in); photon_sources_free(src); return code; };
    if (hipMalloc((void **)&d_in, total * sizeof(double)) != hipSuccess) return fail(3);
    double *d_dx = d_in, *d_dy = d_in + n_dots, *d_tx = d_in + 2 * (size_t)n_dots, *d_ty = d_tx + n_tmpl;
    if ((n_dots && (hipMemcpy(d_dx, dot_x, n_dots * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(d_dy, dot_y, n_dots * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)) ||
        hipMemcpy(d_tx, tmpl_x, n_tmpl * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ty, tmpl_y, n_tmpl * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(4);
    if (n) {
        hipLaunchKernelGGL(sources_bos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_dx, d_dy, (long long)n_dots,
                           d_tx, d_ty, n_tmpl, z, radiance, src->x, src->y, src->z, src->radiance, src->diameter_index);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(4);
    }
    (void)hipFree(d_in);
    *out = src;
    return 0;
}

int photon_sources_piv(uint64_t seed, long long n, const double box_min[3], const double box_max[3], double z_object,
                       double beam_fwhm, double irradiance_constant, const double *diameter_cdf, int n_diameters,
                       photon_sources_t **out) {
    if (!out || n < 0 || n > 0x7fffffffLL || !box_min || !box_max || !(beam_fwhm > 0) || n_diameters < 0 ||
        (n_diameters > 0 && !diameter_cdf)) {
        fprintf(stderr, "photon: photon_sources_piv: bad arguments\n");
        return 1;
    }
    photon_sources *src = nullptr;
    int rc = sources_alloc(n, &src);
    if (rc) return rc;
    PivFieldDev f;
    for (int a = 0; a < 3; a++) { f.lo[a] = box_min[a]; f.hi[a] = box_max[a]; }
    const double sigma = beam_fwhm / (2.0 * sqrt(2.0 * log(2.0)));     // run_simulation_02.py:961
    f.z_object = z_object;
    f.coef = irradiance_constant * (1.0 / (sigma * sqrt(2.0 * PHOTON_PI)));
    f.two_sigma2 = 2.0 * (sigma * sigma);
    f.n_diameters = n_diameters;
    double *d_cdf = nullptr;
    auto fail = [&](int code) { if (d_cdf) (void)hipFree(d_cdf); photon_sources_free(src); return code; };
    if (n_diameters > 0) {
        if (hipMalloc((void **)&d_cdf, n_diameters * sizeof(double)) != hipSuccess) return fail(3);
        if (hipMemcpy(d_cdf, diameter_cdf, n_diameters * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(4);
    }
    if (n) {
        hipLaunchKernelGGL(sources_piv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (unsigned long long)seed, n, f,
                           d_cdf, src->x, src->y, src->z, src->radiance, src->diameter_index);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(4);
    }
    if (d_cdf) (void)hipFree(d_cdf);
    *out = src;
    return 0;
}

long long photon_sources_count(const photon_sources_t *src) { return src ? src->n : -1; }

int photon_sources_download(const photon_sources_t *src, float *x, float *y, float *z, double *radiance,
                            int *diameter_index) {
    if (!src) return 1;
    const size_t n = (size_t)src->n;
    if (!n) return 0;
    if (x) PH_CHECK(hipMemcpy(x, src->x, n * sizeof(float), hipMemcpyDeviceToHost));
    if (y) PH_CHECK(hipMemcpy(y, src->y, n * sizeof(float), hipMemcpyDeviceToHost));
    if (z) PH_CHECK(hipMemcpy(z, src->z, n * sizeof(float), hipMemcpyDeviceToHost));
    if (radiance) PH_CHECK(hipMemcpy(radiance, src->radiance, n * sizeof(double), hipMemcpyDeviceToHost));
    if (diameter_index) PH_CHECK(hipMemcpy(diameter_index, src->diameter_index, n * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

static int scene_create_impl(float lens_pitch, float image_distance, const scattering_data_t *sdp,
                             const char *scattering_type_str, const lightfield_source_t *lsp,
                             const photon_sources *generated, int lightray_number_per_particle, float beam_wavelength,
                             float aperture_f_number, int num_elements, const double (*element_center)[3],
                             const element_data_t *edp, const double (*element_plane_parameters)[4],
                             const int *element_system_index, const camera_design_t *cam, float ray_cone_pitch_ratio,
                             photon_scene_t **out);

int photon_scene_create(float lens_pitch, float image_distance, const scattering_data_t *sdp,
                        const char *scattering_type_str, const lightfield_source_t *lsp,
                        int lightray_number_per_particle, float beam_wavelength, float aperture_f_number,
                        int num_elements, const double (*element_center)[3], const element_data_t *edp,
                        const double (*element_plane_parameters)[4], const int *element_system_index,
                        const camera_design_t *cam, float ray_cone_pitch_ratio, photon_scene_t **out) {
    return guarded("photon_scene_create", [&]() -> int {
        return scene_create_impl(lens_pitch, image_distance, sdp, scattering_type_str, lsp, nullptr,
                                 lightray_number_per_particle, beam_wavelength, aperture_f_number, num_elements, element_center,
                                 edp, element_plane_parameters, element_system_index, cam, ray_cone_pitch_ratio, out);
    });
}

int photon_scene_create_from_sources(float lens_pitch, float image_distance, const scattering_data_t *sdp,
                                     const char *scattering_type_str, const lightfield_source_t *lsp,
                                     const photon_sources_t *sources, int lightray_number_per_particle,
                                     float beam_wavelength, float aperture_f_number, int num_elements,
                                     const double (*element_center)[3], const element_data_t *edp,
                                     const double (*element_plane_parameters)[4], const int *element_system_index,
                                     const camera_design_t *cam, float ray_cone_pitch_ratio, photon_scene_t **out) {
    if (!sources) {
        fprintf(stderr, "photon: photon_scene_create_from_sources: null sources\n");
        return 1;
    }
    return guarded("photon_scene_create_from_sources", [&]() -> int {
        return scene_create_impl(lens_pitch, image_distance, sdp, scattering_type_str, lsp, sources,
                                 lightray_number_per_particle, beam_wavelength, aperture_f_number, num_elements, element_center,
                                 edp, element_plane_parameters, element_system_index, cam, ray_cone_pitch_ratio, out);
    });
}

static int scene_create_impl(float lens_pitch, float image_distance, const scattering_data_t *sdp,
                             const char *scattering_type_str, const lightfield_source_t *lsp,
                             const photon_sources *generated, int lightray_number_per_particle, float beam_wavelength,
                             float aperture_f_number, int num_elements, const double (*element_center)[3],
                             const element_data_t *edp, const double (*element_plane_parameters)[4],
                             const int *element_system_index, const camera_design_t *cam, float ray_cone_pitch_ratio,
                             photon_scene_t **out) {
    if (!sdp || !scattering_type_str || !lsp || !edp || !cam || !out || !element_center || !element_plane_parameters ||
        !element_system_index) {
        fprintf(stderr, "photon: photon_scene_create: null argument\n");
        return 1;
    }
    if (num_elements < 1 || num_elements > 65536) {
        fprintf(stderr, "photon: %d optical elements given, 1..65536 supported\n", num_elements);
        return 1;
    }
    const long long n_sources = generated ? generated->n : (long long)lsp->num_particles;
    if (lightray_number_per_particle < 1 || n_sources < 0 || n_sources > 0x7fffffffLL) {
        fprintf(stderr, "photon: bad ray / source counts\n");
        return 1;
    }
    photon_scene *s = new photon_scene();
    SceneDev &d = s->dev;
    int rc = 0;
    auto bail = [&](int code) { photon_scene_free(s); return code; };
    d.lens_pitch = lens_pitch; d.image_distance = image_distance; d.beam_wavelength = beam_wavelength;
    d.f_number = aperture_f_number; d.ratio = ray_cone_pitch_ratio;
    d.scattering_type = strcmp(scattering_type_str, "mie") == 0 ? 1 : 0;       // .cu:3192
    d.rays_per_source = lightray_number_per_particle;
    const size_t ns = (size_t)n_sources;
    d.num_sources = (int)ns;
    if (generated) {                                    // already in HBM: device-to-device, no host arrays
        if ((rc = copy_device<float>(s, generated->x, ns, &d.sx))) return bail(rc);
        if ((rc = copy_device<float>(s, generated->y, ns, &d.sy))) return bail(rc);
        if ((rc = copy_device<float>(s, generated->z, ns, &d.sz))) return bail(rc);
        if ((rc = copy_device<double>(s, generated->radiance, ns, &d.sradiance))) return bail(rc);
        if ((rc = copy_device<int>(s, generated->diameter_index, ns, &d.sdia))) return bail(rc);
    } else {
        if ((rc = upload(s, lsp->x, ns, &d.sx))) return bail(rc);
        if ((rc = upload(s, lsp->y, ns, &d.sy))) return bail(rc);
        if ((rc = upload(s, lsp->z, ns, &d.sz))) return bail(rc);
        if ((rc = upload(s, lsp->radiance, ns, &d.sradiance))) return bail(rc);
        if ((rc = upload(s, lsp->diameter_index, ns, &d.sdia))) return bail(rc);
    }
    d.z_offset = lsp->z_offset; d.object_distance = lsp->object_distance;
    memcpy(d.mie_inv_rot, sdp->inverse_rotation_matrix, sizeof d.mie_inv_rot);
    memcpy(d.beam, sdp->beam_propagation_vector, sizeof d.beam);
    d.num_angles = sdp->num_angles; d.num_diameters = sdp->num_diameters;
    if (d.scattering_type) {
        if (sdp->num_angles < 2 || sdp->num_diameters < 1 || !sdp->scattering_angle || !sdp->scattering_irradiance) {
            fprintf(stderr, "photon: \"mie\" scattering needs an angle/irradiance table\n");
            return bail(1);
        }
        if ((rc = upload(s, sdp->scattering_angle, (size_t)sdp->num_angles, &d.mie_angle))) return bail(rc);
        if ((rc = upload(s, sdp->scattering_irradiance, (size_t)sdp->num_angles * sdp->num_diameters, &d.mie_irr)))
            return bail(rc);
    }
    std::vector<float> r1(lightray_number_per_particle), r2(lightray_number_per_particle);
    photon_rand_table(lightray_number_per_particle, r1.data(), r2.data());
    if ((rc = upload(s, r1.data(), r1.size(), &d.r1))) return bail(rc);
    if ((rc = upload(s, r2.data(), r2.size(), &d.r2))) return bail(rc);
    d.num_elements = num_elements;
    {
        std::vector<float> centers(3 * (size_t)num_elements), planes(4 * (size_t)num_elements);
        for (int k = 0; k < num_elements; k++) {                               // .cu:3256-3260 (f64 -> f32)
            for (int j = 0; j < 3; j++) centers[3 * k + j] = (float)element_center[k][j];
            for (int j = 0; j < 4; j++) planes[4 * k + j] = (float)element_plane_parameters[k][j];
            if (k < kMaxElements) {                                            // the reference path reads these
                d.elems[k] = edp[k];
                for (int j = 0; j < 3; j++) d.centers[k][j] = centers[3 * k + j];
                for (int j = 0; j < 4; j++) d.planes[k][j] = planes[4 * k + j];
                d.sys_index[k] = element_system_index[k];
            }
        }
        d.train_mode = 0;
        d.ray_order = 0;
        d.src_perm = nullptr;
        d.source_base = 0;
        d.doom_margin = 0.f;
        s->lens_z = (float)element_center[0][2];
        if ((rc = upload(s, edp, (size_t)num_elements, &d.all_elems))) return bail(rc);
        if ((rc = upload(s, centers.data(), centers.size(), &d.all_centers))) return bail(rc);
        if ((rc = upload(s, planes.data(), planes.size(), &d.all_planes))) return bail(rc);
        if ((rc = upload(s, element_system_index, (size_t)num_elements, &d.all_sys_index))) return bail(rc);
    }
    d.cam = *cam;
    d.noise = NoiseDev{0, 0, 0.f, 0.f, 0ull};
    if (cam->x_pixel_number < 1 || cam->y_pixel_number < 1) {
        fprintf(stderr, "photon: sensor needs at least one pixel\n");
        return bail(1);
    }
    hipError_t e = pool_malloc((void **)&s->d_counters, (size_t)kCounterSlots * kCounterStride * sizeof(unsigned long long));
    if (e != hipSuccess) { fprintf(stderr, "photon: hipMalloc failed: %s\n", hipGetErrorString(e)); return bail((int)e); }
    e = pool_malloc((void **)&s->d_acc, (size_t)cam->x_pixel_number * cam->y_pixel_number * sizeof(double));
    if (e != hipSuccess) { fprintf(stderr, "photon: hipMalloc failed: %s\n", hipGetErrorString(e)); return bail((int)e); }
    e = pool_malloc((void **)&s->d_queue, kQueues * kQueueStride * sizeof(unsigned));
    if (e != hipSuccess) { fprintf(stderr, "photon: hipMalloc failed: %s\n", hipGetErrorString(e)); return bail((int)e); }
    e = pool_malloc((void **)&s->d_error, sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(s->d_error, 0, sizeof(unsigned));
    if (e != hipSuccess) { fprintf(stderr, "photon: hipMalloc failed: %s\n", hipGetErrorString(e)); return bail((int)e); }
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            s->num_cus = cus;
    }
    for (auto &ev : s->ev) {
        e = hipEventCreate(&ev);
        if (e != hipSuccess) { fprintf(stderr, "photon: hipEventCreate failed: %s\n", hipGetErrorString(e)); return bail((int)e); }
    }
    *out = s;
    return 0;
}

int photon_scene_set_noise(photon_scene_t *scene, int add_pos_noise, float pos_noise_std, int add_ngrad_noise,
                           float ngrad_noise_std, uint64_t seed) {
    if (!scene) return 1;
    scene->dev.noise = NoiseDev{add_pos_noise ? 1 : 0, add_ngrad_noise ? 1 : 0, pos_noise_std, ngrad_noise_std,
                                (unsigned long long)seed};
    return 0;
}

unsigned photon_march_queue_count(void) { return 8u * kSubQueues; }
unsigned photon_march_queue_chunk(int interpolation) { return 1u << (interpolation == 2 ? kChunkShiftCubic : kChunkShiftLinear); }
static unsigned chunk_shift_of(unsigned groups_per_chunk) {    // log2 of a power of two in [1, 2^16]; 32 otherwise
    for (unsigned s = 0; s <= 16; s++) if (groups_per_chunk == (1u << s)) return s;
    return 32u;
}
unsigned photon_march_queue_group(unsigned k, unsigned xcd, unsigned sub, unsigned groups_per_chunk) {
    const unsigned shift = chunk_shift_of(groups_per_chunk);
    return xcd < 8u && sub < kSubQueues && shift < 32u ? march_queue_group(k, xcd, sub, shift) : ~0u;
}
unsigned photon_march_queue_size(unsigned n_groups, unsigned xcd, unsigned sub, unsigned groups_per_chunk) {
    const unsigned shift = chunk_shift_of(groups_per_chunk);
    return xcd < 8u && sub < kSubQueues && shift < 32u ? march_queue_size(n_groups, xcd, sub, shift) : ~0u;
}

int photon_scene_set_source_base(photon_scene_t *s, int64_t first_source) {
    if (!s || first_source < 0) return 1;
    s->dev.source_base = (long long)first_source;
    return 0;
}

int photon_scene_set_element_train(photon_scene_t *s, int mode) {
    if (!s || (mode != 0 && mode != 1)) return 1;
    s->dev.train_mode = mode;
    return 0;
}

int photon_scene_set_ray_order(photon_scene_t *s, int mode) {
    if (!s || mode < 0 || mode > 2) return 1;
    s->ray_order_mode = mode;
    return 0;
}

int photon_scene_set_skip_doomed(photon_scene_t *s, int on) {
    if (!s) return 1;
    s->skip_doomed = on != 0;
    return 0;
}

}  // extern "C"

// rays per launch: bounded so that 32-bit ray ids suffice and the state stays a few GB
static const unsigned kMaxRaysPerLaunch = 1u << 26;

static void free_resume_state(photon_scene *s) {
    if (s->ws.ctr) { pool_free(s->ws.ctr); s->ws.ctr = nullptr; }
    if (s->ws.vprev) { pool_free(s->ws.vprev); s->ws.vprev = nullptr; }
    s->ws.spins = nullptr; s->ws.seg_flag = nullptr;
}

static int ensure_workspace(photon_scene *s, size_t rays) {
    if (s->ws_rays >= rays) return 0;
    if (s->ws.px) { pool_free(s->ws.px); s->ws.px = nullptr; }
    if (s->ws.radiance) { pool_free(s->ws.radiance); s->ws.radiance = nullptr; }
    free_resume_state(s);
    s->ws_rays = 0;
    float *f = nullptr;
    PH_CHECK(pool_malloc((void **)&f, rays * 6 * sizeof(float)));
    s->ws.px = f; s->ws.py = f + rays; s->ws.pz = f + 2 * rays;
    s->ws.dx = f + 3 * rays; s->ws.dy = f + 4 * rays; s->ws.dz = f + 5 * rays;
    PH_CHECK(pool_malloc((void **)&s->ws.radiance, rays * sizeof(double)));
    s->ws_rays = rays;
    s->ws.stride = (unsigned)rays;
    return 0;
}

// What a segmented march keeps per ray between segments (MarchResume) and the per-group flags; allocated with the first
// segmented launch of a workspace size.  The flags carry the launch's epoch, so they are zeroed once, here (and when the
// 24-bit epoch wraps), not per launch.
static int ensure_resume_state(photon_scene *s, bool linear, hipStream_t stream) {
    const size_t rays = s->ws_rays, groups = (rays + 63) / 64;
    if (!s->ws.ctr) {
        unsigned *u = nullptr;
        PH_CHECK(pool_malloc((void **)&u, (2 * rays + groups) * sizeof(unsigned)));
        s->ws.ctr = u; s->ws.spins = u + rays; s->ws.seg_flag = u + 2 * rays;
        PH_CHECK(hipMemsetAsync(s->ws.seg_flag, 0, groups * sizeof(unsigned), stream));
        s->march_epoch = 0;
    }
    if (linear && !s->ws.vprev) PH_CHECK(pool_malloc((void **)&s->ws.vprev, 4 * rays * sizeof(float)));
    if (++s->march_epoch >= (1u << 24)) {
        PH_CHECK(hipMemsetAsync(s->ws.seg_flag, 0, groups * sizeof(unsigned), stream));
        s->march_epoch = 1;
    }
    return 0;
}

static int begin_accumulate(photon_scene *s, hipStream_t stream) {
    const size_t npix = (size_t)s->dev.cam.x_pixel_number * s->dev.cam.y_pixel_number;
    PH_CHECK(hipMemsetAsync(s->d_acc, 0, npix * sizeof(double), stream));
    return 0;
}
static int end_accumulate(photon_scene *s, float *d_image, hipStream_t stream) {
    const size_t npix = (size_t)s->dev.cam.x_pixel_number * s->dev.cam.y_pixel_number;
    hipLaunchKernelGGL(finalize_image_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream, d_image,
                       s->d_acc, npix);
    PH_CHECK(hipGetLastError());
    return 0;
}

// Spatial order of the sources for lens-major launches: Morton code of (x, y) on a 2^16 grid over the bounding
// box of the LAUNCHED range [src_begin, src_end), sorted on the device (photon_sort.hip) -- start_ray_tracing
// builds a new scene per call, so this sits on the per-image path of every PIV-through-volume frame (1e6 sources:
// a host sort cost a D2H of the coordinates, ~0.1 s of std::stable_sort and an H2D per call).  The permutation
// covers exactly the launched range, so [src_begin, src_end) always counts sources in the CALLER's order,
// whatever order the lanes then use; it is kept for the next launch of the same range.

// The permutation of a launched range is kept (a few ranges: a job's chunks, a caller alternating shards), and the sort's
// scratch lives in the scene: a lens-major launch of a range seen before costs nothing, a new range costs the sort's
// kernels on the stream -- no allocation, no host wait, so photon_trace without stats stays asynchronous.
static int ensure_source_order(photon_scene *s, long long src_begin, long long src_end, hipStream_t stream, const int **perm_out) {
    const size_t n = (size_t)(src_end - src_begin);
    s->perm_clock++;
    PermEntry *slot = nullptr;
    for (auto &p : s->perms)
        if (p.d_perm && p.begin == src_begin && p.end == src_end) { p.stamp = s->perm_clock; *perm_out = p.d_perm; return 0; }
    for (auto &p : s->perms)                                            // least recently used (an empty one first)
        if (!slot || (!p.d_perm && slot->d_perm) || (!!p.d_perm == !!slot->d_perm && p.stamp < slot->stamp)) slot = &p;
    slot->begin = slot->end = -1;
    if (slot->capacity < n || !slot->d_perm) {
        if (slot->d_perm) { pool_free(slot->d_perm); slot->d_perm = nullptr; }
        slot->capacity = 0;
        PH_CHECK(pool_malloc((void **)&slot->d_perm, std::max<size_t>(n, 1) * sizeof(int)));
        slot->capacity = n;
    }
    const int rc = photon_morton_order(s->dev.sx, s->dev.sy, (int)src_begin, (long long)n, slot->d_perm, stream, &s->sort_scratch);
    if (rc) return rc;
    slot->begin = src_begin; slot->end = src_end; slot->stamp = s->perm_clock;
    *perm_out = slot->d_perm;
    return 0;
}

// Which order a launch uses.  Lens-major pays off when the ray cone of a source is wider than the volume's
// texels where it crosses the volume (then the 64 rays of ONE source fan out over many texel blocks, while
// 64 neighbouring sources aimed at one lens point stay together); source-major otherwise (BOS: the cone is a
// micron wide) and whenever something indexes rays by the reference's launch order (ray dumps) or the march
// needs per-ray ids (gradient noise).
static bool use_lens_major(const photon_scene *s, const photon_volume *vol, const DumpDev &dump) {
    if (!vol || dump.final_pos || dump.inter_pos || s->dev.noise.add_ngrad || s->dev.rays_per_source < 2) return false;
    if (s->ray_order_mode != 2) return s->ray_order_mode == 1;
    const double z_obj = (double)s->dev.object_distance + s->dev.z_offset;             // camera frame
    const double z_face = (double)vol->dev.min_bound.z + s->dev.z_offset + 750e3;      // the volume's lens-side face
    const double span = z_obj - s->lens_z;
    if (!(span > 0)) return false;
    double frac = (z_obj - z_face) / span;
    frac = frac < 0 ? 0 : (frac > 1 ? 1 : frac);
    const double cone = (double)s->dev.ratio * s->dev.lens_pitch * frac;               // cone diameter at that face
    const photon_volume_info_t &i = vol->info;
    const double texel = std::min((double)i.grid_spacing[0], std::min((double)i.grid_spacing[1], (double)i.grid_spacing[2]));
    return cone > texel;
}

// Rays that cannot reach the sensor need not be marched.  The reference kills a ray whose intersection with the
// first element's front surface lies more than pitch/2 from the axis (.cu:447, 560-566) -- for a full-aperture
// cone that is half of all rays, because the lens-sample radius goes up to pitch, not pitch/2 (.cu:123-124).
// The volume only bends a ray by a bounded angle: |d(n t)/ds| = |grad n| <= G, so after a path of length L inside
// the volume its direction is off by at most G L / n_min, and its footprint on the lens by at most that angle times
// the distance still to go (plus the walk-off inside the volume).  Returns that bound, times a safety factor
// that also covers the tricubic sampler's overshoot and the integrator's error, plus a thousandth of the
// aperture; 0 when the skip does not apply.
static float doom_margin(const photon_scene *s, const photon_volume *vol, int algorithm, const DumpDev &dump) {
    if (!s->skip_doomed || !vol || (algorithm != 1 && algorithm != 2) || dump.final_pos || dump.inter_pos) return 0.f;
    if (s->dev.train_mode != 0 || s->dev.noise.add_ngrad) return 0.f;
    const char type = s->dev.elems[0].element_type;
    if (type != 'l' && type != 't') return 0.f;
    // the reference path applies element 0 once per single-member group: there must be one
    bool applied = false;
    const int n = std::min(s->dev.num_elements, kMaxElements);
    int seq = 0;
    for (int k = 0; k < n; k++) seq = std::max(seq, s->dev.sys_index[k]);
    for (int idx = 0; idx < seq && !applied; idx++) {
        int count = 0;
        for (int k = 0; k < n; k++) count += (seq - s->dev.sys_index[k] == idx);
        applied = count == 1;
    }
    if (!applied) return 0.f;
    const VolumeDev &v = vol->dev;
    const double ex = (double)v.max_bound.x - v.min_bound.x, ey = (double)v.max_bound.y - v.min_bound.y,
                 ez = (double)v.max_bound.z - v.min_bound.z;
    const double L = sqrt(ex * ex + ey * ey + ez * ez);
    const double n_min = 1.0 + std::min(0.0, (double)v.data_min);
    const double angle = (double)vol->grad_max * L / n_min;
    const double z_obj = (double)s->dev.object_distance + s->dev.z_offset;
    const double to_lens = fabs(z_obj - s->lens_z) + L;                 // generous: the whole object-lens distance
    const double pitch = s->dev.elems[0].element_geometry.pitch;
    const double margin = 8.0 * angle * (to_lens + L) + 1e-3 * pitch;
    if (!(margin == margin) || !(pitch > 0)) return 0.f;
    return (float)margin;
}

// Segments per march of a launch large enough to be segmented: PHOTON_MARCH_SEGMENTS=<n> (1 = whole marches), or
// PHOTON_MARCH_SEGMENTS=force:<n> to segment launches of any size (tests of the hand-off between segments).
static int march_segments_default(bool *forced) {
    const char *e = getenv("PHOTON_MARCH_SEGMENTS");            // read per launch: tests switch it between calls
    if (e && !strncmp(e, "force:", 6)) { *forced = true; e += 6; }
    const int v = (e && atoi(e) > 0) ? atoi(e) : PHOTON_MARCH_SEGMENTS;
    return v > 64 ? 64 : v;
}

// Shape of the pieces of a segmented march.  Equal pieces; HALVING pieces (1/2, 1/4, ... of the depth, the last two equal):
// a third of the hand-offs for the same final piece, but every pass then runs twice as fast as the one that feeds it -- in a
// launch of few chip fills its front catches up with the pieces it depends on and waves stand polling (measured, one GPU's
// eighth of C3, 3.8 fills: 8.07-8.15 ms halving against 7.53-7.62 equal; the full job, 30.5 fills: 56.95 against 57.28; the
// front stays clear while r / 2 <= R - 2 for every round r <= R of a pass: halving from 12 fills on); TAPERED pieces: equal
// ones, the last of them halved t times (.., u, u/2, u/4, u/4 for t = 2) -- a short final pass without the long chain of
// ever faster passes.  PHOTON_MARCH_SEGMENT_SHAPE=uniform|halving|taper:<t> overrides the choice (A/B runs, tests).
enum SegShape { SEG_UNIFORM = 0, SEG_HALVING = 1, SEG_TAPER = 2 };
static SegShape segment_shape(double fills, unsigned *taper) {
    const char *e = getenv("PHOTON_MARCH_SEGMENT_SHAPE");
    *taper = 0;
    if (e && !strcmp(e, "uniform")) return SEG_UNIFORM;
    if (e && !strcmp(e, "halving")) return SEG_HALVING;
    if (e && !strncmp(e, "taper:", 6)) { *taper = (unsigned)std::max(1, std::min(atoi(e + 6), 8)); return SEG_TAPER; }
    return fills >= 12.0 ? SEG_HALVING : SEG_UNIFORM;
}

// How many pieces, and how long each: fills `begin` (begin[s] = first trip of piece s; begin[S] = depth) and returns S.
// Every hand-off costs c (flag poll, state round trip, tile refetch); the launch's drain is 0.75 of its LAST pieces.  Equal
// pieces: a launch of R chip fills of groups that march for L each costs R (S - 1) c + 0.75 L / S -- measured on C3
// (tools/segments_sweep.sh; tricubic / trilinear RK4, full job R = 30.5, one GPU's eighth R = 3.8): optima S = 4 / 2-3 and
// 12-16 / 6-8, the model's 4.0 / 2.3 and 11.3 / 6.5 with c = 2.9 us and L = 0.82 us per unit of work (one trilinear sample per
// texel of depth; x3 for RK4's three samples, x3 for the 64-tap sampler: RK4 tricubic through 256 texels = 2304 units =
// 1.9 ms).  Only the last pass's pieces need to be short: R (S - 1) c + 0.75 (last piece), minimised over S for the shape
// in use.  At most `cap` pieces; `forced` takes the cap itself (tests); the shortest piece is 4 trips.
static unsigned plan_segments(unsigned groups, unsigned slots, unsigned depth, int algorithm, int interp, unsigned cap, bool forced,
                              unsigned *begin, int *shape_out) {
    const double fills = (double)groups / (double)std::max(slots, 1u);
    unsigned taper = 0;
    const SegShape shape = segment_shape(fills, &taper);
    cap = std::max(1u, std::min(cap, kMaxSegments));
    // lengths (as fractions of the depth) of the S pieces of a shape
    auto lengths = [&](unsigned S) {
        std::vector<double> len;
        if (shape == SEG_HALVING) {
            for (unsigned k = 1; k < S; k++) len.push_back(1.0 / (double)(1ull << std::min(k, 40u)));
            len.push_back(S > 1 ? len.back() : 1.0);
        } else {
            const unsigned t = shape == SEG_TAPER ? std::min(taper, S - 1) : 0, base = S - t;
            for (unsigned k = 0; k + 1 < base; k++) len.push_back(1.0 / base);
            double u = 1.0 / base;
            for (unsigned k = 0; k < t; k++) { u *= 0.5; len.push_back(u); }
            len.push_back(u);
        }
        return len;
    };
    unsigned S = cap;
    if (!forced) {
        // Per sampler (refitted for the trilinear kernels after the sampler work of round 4: same sweep, full job 1 / 2 / 3 / 4
        // pieces 17.22 / 17.33 / 17.39 / 17.48 ms, one GPU's eighth 1 / 3 / 4 / 6 / 8 / 12 pieces 2.359 / 2.276 / 2.288 / 2.298 /
        // 2.315 / 2.376: a hand-off costs them 4.2 us -- they carry the last sampled value along -- and their launches drain
        // over 0.31 of a last piece, in 0.72 us per unit: the full job runs whole, the eighth in 3 pieces).
        const double units = (double)depth * (algorithm == 2 ? 3.0 : 1.0) * (interp == 2 ? 3.0 : 1.0);
        const double L = (interp == 2 ? 0.82 : 0.72) * units, c = interp == 2 ? 2.9 : 4.2, drain = interp == 2 ? 0.75 : 0.31;
        double best_cost = drain * L;
        S = 1;
        for (unsigned k = 2; k <= cap; k++) {
            const double cost = fills * (k - 1) * c + drain * L * lengths(k).back();
            if (cost < best_cost) { best_cost = cost; S = k; }
        }
    }
    // boundaries in trips; pieces shorter than 4 trips are merged into their predecessor
    for (;; S--) {
        const std::vector<double> len = lengths(S);
        double at = 0.0;
        bool ok = true;
        begin[0] = 0;
        for (unsigned k = 0; k < S; k++) {
            at += len[k];
            begin[k + 1] = k + 1 == S ? depth : (unsigned)(at * depth + 0.5);
            if (begin[k + 1] < begin[k] + 4u) ok = false;
        }
        if (ok || S == 1) break;
    }
    if (S == 1) { begin[0] = 0; begin[1] = depth; }
    if (shape_out) *shape_out = S > 1 ? (int)shape : (int)SEG_UNIFORM;
    return S;
}

// The library's choice for a launch of n_rays through a volume of `depth` texels on a device of num_cus compute units
// (host restatement for tests and documentation; PHOTON_MARCH_SEGMENT_SHAPE is honoured, PHOTON_MARCH_SEGMENTS is not).
extern "C" int photon_march_segments_plan(unsigned n_rays, int depth, int ray_tracing_algorithm, int interpolation, int num_cus, int *halving) {
    if (depth < 1 || num_cus < 1 || (ray_tracing_algorithm != 1 && ray_tracing_algorithm != 2)) return 0;
    const unsigned groups = (n_rays + 63u) / 64u, slots = (unsigned)num_cus * 4u * march_waves_of(ray_tracing_algorithm, interpolation);
    int shape = 0;
    unsigned s = 1, begin[kMaxSegments + 1];
    if (PHOTON_MARCH_PERSISTENT && groups >= slots + slots / 4) s = plan_segments(groups, slots, (unsigned)depth, ray_tracing_algorithm, interpolation, PHOTON_MARCH_SEGMENTS, false, begin, &shape);
    if (halving) *halving = shape == SEG_HALVING ? 1 : 0;
    return (int)s;
}

// The march launch of n rays whose state sits in the scene's workspace (stage 1b): persistent grid, work queues, segments.
static int launch_march(photon_scene *s, const photon_volume *vol, int algorithm, unsigned n, unsigned long long ray_base,
                        const InterDump &idump, bool save, hipStream_t stream, hipEvent_t ev_march_begin) {
    const dim3 block(256), grid((n + 255) / 256);
    const int interp = vol->dev.interpolation;
    const f4 *tex = interp == 2 ? vol->d_coeffs : vol->d_texels;
    // persistent waves: a grid that fills the chip once (more workgroups than fit only find empty queues and leave)
    const unsigned all_blocks = (n + PHOTON_MARCH_BLOCK - 1) / PHOTON_MARCH_BLOCK;
    const unsigned fill_blocks = (unsigned)s->num_cus * 8u * (256 / PHOTON_MARCH_BLOCK);
    const dim3 mblock(PHOTON_MARCH_BLOCK), mgrid(PHOTON_MARCH_PERSISTENT ? std::min(all_blocks, fill_blocks) : all_blocks);
    if (PHOTON_MARCH_PERSISTENT && (algorithm == 1 || algorithm == 2))
        PH_CHECK(hipMemsetAsync(s->d_queue, 0, kQueues * kQueueStride * sizeof(unsigned), stream));
    if (ev_march_begin) PH_CHECK(hipEventRecord(ev_march_begin, stream));
    unsigned long long *profile = nullptr;                  // wave timing of this launch, while there are free slots
    if (s->d_profile && s->prof_next < kProfileLaunches && (algorithm == 1 || algorithm == 2))
        profile = s->d_profile + (size_t)(s->prof_next++) * kProfileSub * PF_N;
    // Segments: only where the launch is several times what the chip holds at once (a segment's wave then finds the
    // previous segment of its group long done) and nothing indexes a ray's iterations (dumps, gradient noise).
    unsigned segments = 1;
    MarchArgs margs{};
    if (PHOTON_MARCH_PERSISTENT && (algorithm == 1 || algorithm == 2) && !save && !s->dev.noise.add_ngrad) {
        const unsigned groups = (n + 63u) / 64u;
        // resident march waves: five or six per SIMD (the launch bounds of the march kernels)
        const unsigned slots = (unsigned)s->num_cus * 4u * march_waves_of(algorithm, interp);
        bool forced = s->march_segments > 1;                // an explicit count segments launches of any size (tests)
        const int want = s->march_segments >= 0 ? s->march_segments : march_segments_default(&forced);
        if (want > 1 && (forced || groups >= slots + slots / 4)) {
            const unsigned depth = (unsigned)std::max(vol->dev.nx, std::max(vol->dev.ny, vol->dev.nz));
            segments = plan_segments(groups, slots, depth, algorithm, interp, (unsigned)std::min(want, 64), forced, margs.seg_begin, nullptr);
            if (segments > 1) { const int rc = ensure_resume_state(s, interp == 1, stream); if (rc) return rc; }
        }
    }
    margs.vol = vol->dev; margs.tex = tex; margs.n_rays = n; margs.st = s->ws; margs.counters = s->d_counters; margs.noise = s->dev.noise;
    margs.ray_base = ray_base; margs.idump = idump; margs.queue = s->d_queue; margs.profile = profile; margs.segments = segments;
    margs.epoch = s->march_epoch; margs.error = s->d_error;
    // queue chunks: small ones (tail balance) for the tricubic kernels where neighbouring groups are neighbouring SOURCES and
    // the volume is small enough for every L2 to hold what its waves touch; lens-major launches (neighbouring groups share
    // a lens tile, their rays fan out over the whole volume) and large volumes keep the L2-friendly 128 -- C5 at a
    // quarter: 11.0 GB of HBM traffic per launch with 16-group chunks against 3.8 GB with 128, 38.03 against 37.94 ms
    margs.chunk_shift = interp == 2 && s->dev.ray_order == 0 && (size_t)vol->dev.nx * vol->dev.ny * vol->dev.nz <= ((size_t)1 << 24)
                            ? kChunkShiftCubic : kChunkShiftLinear;
#define PH_MARCH(A, I, S, N) do { if (!S && !N && segments > 1) hipLaunchKernelGGL((march_kernel<A, I, false, false, true>), mgrid, mblock, 0, stream, margs); \
                              else hipLaunchKernelGGL((march_kernel<A, I, S, N, false>), mgrid, mblock, 0, stream, margs); } while (0)
    if (algorithm == 3) hipLaunchKernelGGL((march_extra_kernel<3>), grid, block, 0, stream, vol->dev, n, s->ws, s->d_counters);
    else if (algorithm == 4) hipLaunchKernelGGL((march_extra_kernel<4>), grid, block, 0, stream, vol->dev, n, s->ws, s->d_counters);
    else if (algorithm != 1 && algorithm != 2) hipLaunchKernelGGL((march_extra_kernel<0>), grid, block, 0, stream, vol->dev, n, s->ws, s->d_counters);
    else if (algorithm == 1 && interp == 1) {              // the gradient-noise hook exists in this branch only (.h:853-863)
        const bool ngrad = s->dev.noise.add_ngrad != 0;
        if (save) { if (ngrad) PH_MARCH(1, 1, true, true); else PH_MARCH(1, 1, true, false); }
        else { if (ngrad) PH_MARCH(1, 1, false, true); else PH_MARCH(1, 1, false, false); }
    }
    else if (algorithm == 1) PH_MARCH(1, 2, false, false);
    else if (interp == 1) { if (save) PH_MARCH(2, 1, true, false); else PH_MARCH(2, 1, false, false); }
    else PH_MARCH(2, 2, false, false);
#undef PH_MARCH
    PH_CHECK(hipGetLastError());
    return 0;
}

static int launch_chunk(photon_scene *s, const photon_volume *vol, int algorithm, long long src_begin,
                        long long src_end, DumpDev dump, hipStream_t stream, hipEvent_t ev_march_begin, hipEvent_t ev_march_end) {
    double *d_image = s->d_acc;
    const unsigned long long n64 = (unsigned long long)(src_end - src_begin) * (unsigned)s->dev.rays_per_source;
    if (n64 == 0) return 0;
    if (n64 > kMaxRaysPerLaunch) {
        fprintf(stderr, "photon: a launch of %llu rays (sources [%lld, %lld) x %d) exceeds the %u-ray limit per launch\n", n64,
                src_begin, src_end, s->dev.rays_per_source, kMaxRaysPerLaunch);
        return 1;
    }
    const unsigned n = (unsigned)n64;
    const dim3 block(256), grid((n + 255) / 256);
    s->dev.doom_margin = doom_margin(s, vol, algorithm, dump);
    s->dev.ray_order = 0;
    s->dev.src_perm = nullptr;
    if (use_lens_major(s, vol, dump)) {
        const int *perm = nullptr;
        const int rc = ensure_source_order(s, src_begin, src_end, stream, &perm);
        if (rc) return rc;
        s->dev.ray_order = 1;
        s->dev.src_perm = perm;
    }
    if (vol) {
        int rc = ensure_workspace(s, n);
        if (rc) return rc;
        hipLaunchKernelGGL(raygen_kernel, grid, block, 0, stream, s->dev, src_begin, n, s->ws);
        PH_CHECK(hipGetLastError());
        const int interp = vol->dev.interpolation;
        const unsigned long long ray_base = (unsigned long long)(s->dev.source_base + src_begin) * (unsigned)s->dev.rays_per_source;
        const InterDump idump{dump.inter_pos, dump.inter_dir, dump.inter_slots, dump.num_save, 0u};
        const bool save = dump.inter_pos != nullptr && interp == 1;     // only the trilinear branches record
        rc = launch_march(s, vol, algorithm, n, ray_base, idump, save, stream, ev_march_begin);
        if (rc) return rc;
        if (ev_march_end) PH_CHECK(hipEventRecord(ev_march_end, stream));
        // erf splats: optics and splat as two kernels (each gets the register file to itself); the 4-pixel
        // splat is done in place by the first
        const bool erf = PHOTON_SPLIT_SENSOR && (s->dev.cam.implement_diffraction || s->dev.elems[0].element_type == 'n');
#define PH_SENSOR(T, S) hipLaunchKernelGGL((sensor_kernel<true, T, S>), grid, block, 0, stream, s->dev, src_begin, n, s->ws, d_image, dump, s->d_counters)
        if (s->dev.train_mode) { if (erf) PH_SENSOR(true, true); else PH_SENSOR(true, false); }
        else { if (erf) PH_SENSOR(false, true); else PH_SENSOR(false, false); }
#undef PH_SENSOR
        if (erf) {
            PH_CHECK(hipGetLastError());
            hipLaunchKernelGGL(splat_kernel, grid, block, 0, stream, n, s->ws, d_image, s->dev.cam.x_pixel_number,
                               s->dev.cam.y_pixel_number, s->d_counters);
        }
    } else {
        if (s->dev.train_mode) hipLaunchKernelGGL((sensor_kernel<false, true, false>), grid, block, 0, stream, s->dev, src_begin, n, s->ws, d_image, dump, s->d_counters);
        else hipLaunchKernelGGL((sensor_kernel<false, false, false>), grid, block, 0, stream, s->dev, src_begin, n, s->ws, d_image, dump, s->d_counters);
    }
    PH_CHECK(hipGetLastError());
    return 0;
}

// March-only entry point THROUGH the render path's march launch (persistent waves, work queues, segments) for arbitrary
// rays: what the adversarial parity tests drive (photon_trace_volume_rays runs a plain one-thread-per-ray grid instead).
extern "C" int photon_trace_volume_rays_queued(const photon_volume_t *vol, int ray_tracing_algorithm, int n, float *pos, float *dir,
                                               int segments) {
    if (!vol || !pos || !dir || n < 0 || (unsigned)n > kMaxRaysPerLaunch || segments == 0 || segments < -1 || segments > 64 ||
        (ray_tracing_algorithm != 1 && ray_tracing_algorithm != 2)) return 1;
    if (n == 0) return 0;
    return guarded("photon_trace_volume_rays_queued", [&]() -> int {
        photon_scene sc;                                        // a bare scene: only what the march launch touches
        struct Cleanup { photon_scene *s; ~Cleanup() {
            pool_free(s->ws.px); pool_free(s->ws.radiance); free_resume_state(s);
            pool_free(s->d_counters); pool_free(s->d_queue); pool_free(s->d_error);
        } } cleanup{&sc};
        sc.march_segments = segments;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) sc.num_cus = cus;
        PH_CHECK(pool_malloc((void **)&sc.d_counters, (size_t)kCounterSlots * kCounterStride * sizeof(unsigned long long)));
        PH_CHECK(hipMemset(sc.d_counters, 0, (size_t)kCounterSlots * kCounterStride * sizeof(unsigned long long)));
        PH_CHECK(pool_malloc((void **)&sc.d_queue, kQueues * kQueueStride * sizeof(unsigned)));
        PH_CHECK(pool_malloc((void **)&sc.d_error, sizeof(unsigned)));
        PH_CHECK(hipMemset(sc.d_error, 0, sizeof(unsigned)));
        { const int rc = ensure_workspace(&sc, (size_t)n); if (rc) return rc; }
        std::vector<float> soa((size_t)n * 6);
        for (int i = 0; i < n; i++)
            for (int c = 0; c < 3; c++) { soa[(size_t)c * n + i] = pos[3 * i + c]; soa[(size_t)(3 + c) * n + i] = dir[3 * i + c]; }
        float *arrays[6] = {sc.ws.px, sc.ws.py, sc.ws.pz, sc.ws.dx, sc.ws.dy, sc.ws.dz};
        for (int c = 0; c < 6; c++) PH_CHECK(hipMemcpy(arrays[c], soa.data() + (size_t)c * n, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        const InterDump no_dump{nullptr, nullptr, 0, 0, 0u};
        { const int rc = launch_march(&sc, vol, ray_tracing_algorithm, (unsigned)n, 0ull, no_dump, false, nullptr, nullptr); if (rc) return rc; }
        PH_CHECK(hipDeviceSynchronize());
        { const int rc = march_error_check(&sc); if (rc) return rc; }
        for (int c = 0; c < 6; c++) PH_CHECK(hipMemcpy(soa.data() + (size_t)c * n, arrays[c], (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++)
            for (int c = 0; c < 3; c++) { pos[3 * i + c] = soa[(size_t)c * n + i]; dir[3 * i + c] = soa[(size_t)(3 + c) * n + i]; }
        return 0;
    });
}

constexpr unsigned kWindowMaxTraces = 1u << 16;        // traces per statistics window (each keeps a few HIP events alive)

// An event of the open statistics window (created on first use, kept for the next window).
static int window_event(photon_scene *s, size_t *index_out) {
    if (s->win_used == s->win_events.size()) {
        hipEvent_t e = nullptr;
        PH_CHECK(hipEventCreate(&e));
        s->win_events.push_back(e);
    }
    *index_out = s->win_used++;
    return 0;
}

// The launch loop for sources [src_begin, src_end) into the scene's private f64 accumulator (zeroed first); the
// caller folds the accumulator into an image (end_accumulate) -- or, when several devices share one call, sums the
// accumulators first.  timed: 0 no events; 1 immediate (the march of every launch is timed with ev[1], ev[2] and the host
// waits for it: photon_trace with a stats pointer); 2 deferred (events of the open statistics window, no host wait).
static int trace_accumulate(photon_scene *scene, const photon_volume *vol, int ray_tracing_algorithm, long long src_begin,
                            long long src_end, hipStream_t stream, int timed, float *march_ms_out) {
    const unsigned rps = (unsigned)scene->dev.rays_per_source;
    if (rps > kMaxRaysPerLaunch) { fprintf(stderr, "photon: too many rays per source\n"); return 1; }
    const long long max_sources = std::max<long long>(1, kMaxRaysPerLaunch / rps);
    float march_ms = 0.f;
    const DumpDev no_dump{nullptr, nullptr, 0, nullptr, nullptr, 0};
    { const int rc = begin_accumulate(scene, stream); if (rc) return rc; }
    for (long long b = src_begin; b < src_end; b += max_sources) {
        const long long e = std::min<long long>(src_end, b + max_sources);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timed == 1 && vol) { e0 = scene->ev[1]; e1 = scene->ev[2]; }
        size_t i0 = 0, i1 = 0;
        if (timed == 2 && vol) {
            { const int rc = window_event(scene, &i0); if (rc) return rc; }
            { const int rc = window_event(scene, &i1); if (rc) return rc; }
            e0 = scene->win_events[i0]; e1 = scene->win_events[i1];
        }
        const int rc = launch_chunk(scene, vol, ray_tracing_algorithm, b, e, no_dump, stream, e0, e1);
        if (rc) return rc;
        if (timed == 2 && vol) scene->win_march.emplace_back(i0, i1);      // only pairs whose events were recorded
        if (timed == 1 && vol) {
            PH_CHECK(hipEventSynchronize(scene->ev[2]));
            float ms = 0.f;
            PH_CHECK(hipEventElapsedTime(&ms, scene->ev[1], scene->ev[2]));
            march_ms += ms;
        }
    }
    if (march_ms_out) *march_ms_out = march_ms;
    return 0;
}

// Wave timing of the march launches (off by default): the slots are zeroed where the statistics counters are, and every
// march launch after that takes the next one.
static int profile_reset(photon_scene *s, hipStream_t stream) {
    s->prof_next = 0;
    if (s->d_profile) PH_CHECK(hipMemsetAsync(s->d_profile, 0, (size_t)kProfileLaunches * kProfileSub * PF_N * sizeof(unsigned long long), stream));
    return 0;
}

extern "C" int photon_scene_set_march_segments(photon_scene_t *scene, int segments) {
    if (!scene || segments < -1 || segments == 0 || segments > 64) return 1;
    scene->march_segments = segments;
    return 0;
}

extern "C" int photon_scene_set_march_profile(photon_scene_t *scene, int on) {
    if (!scene) return 1;
    return guarded("photon_scene_set_march_profile", [&]() -> int {
        if (on && !scene->d_profile) {
            PH_CHECK(pool_malloc((void **)&scene->d_profile, (size_t)kProfileLaunches * kProfileSub * PF_N * sizeof(unsigned long long)));
            PH_CHECK(hipMemset(scene->d_profile, 0, (size_t)kProfileLaunches * kProfileSub * PF_N * sizeof(unsigned long long)));
        } else if (!on && scene->d_profile) {
            PH_CHECK(hipDeviceSynchronize());
            pool_free(scene->d_profile);
            scene->d_profile = nullptr;
        }
        scene->prof_next = 0;
        return 0;
    });
}

extern "C" int photon_scene_march_profile(photon_scene_t *scene, photon_march_profile_t *out) {
    if (!scene || !out || out->struct_size < sizeof(photon_march_profile_t)) {
        fprintf(stderr, "photon: photon_scene_march_profile: bad arguments (set struct_size = sizeof(photon_march_profile_t))\n");
        return 1;
    }
    return guarded("photon_scene_march_profile", [&]() -> int {
        const uint32_t size = out->struct_size;
        memset(out, 0, sizeof *out);
        out->struct_size = size;
        if (!scene->d_profile || scene->prof_next == 0) return 0;
        const unsigned launches = std::min(scene->prof_next, kProfileLaunches);
        std::vector<unsigned long long> h((size_t)launches * kProfileSub * PF_N);
        PH_CHECK(hipDeviceSynchronize());
        PH_CHECK(hipMemcpy(h.data(), scene->d_profile, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double span = 0, start_mean = 0, start_max = 0, end_min = 0, end_mean = 0, waves_sum = 0;
        unsigned used = 0;
        for (unsigned l = 0; l < launches; l++) {
            unsigned long long enter_min = ~0ull, start_min = ~0ull, start_max_t = 0, end_min_t = ~0ull, end_max_t = 0, waves = 0;
            unsigned long long start_sum = 0, end_sum = 0;       // sums of absolute stamps: modulo 2^64, differences below are exact
            for (unsigned k = 0; k < kProfileSub; k++) {
                const unsigned long long *q = &h[((size_t)l * kProfileSub + k) * PF_N];
                if (q[PF_ENTER_NEGMIN]) enter_min = std::min(enter_min, ~q[PF_ENTER_NEGMIN]);
                if (!q[PF_WAVES]) continue;
                start_min = std::min(start_min, ~q[PF_START_NEGMIN]);
                start_max_t = std::max(start_max_t, q[PF_START_MAX]);
                end_min_t = std::min(end_min_t, ~q[PF_END_NEGMIN]);
                end_max_t = std::max(end_max_t, q[PF_END_MAX]);
                start_sum += q[PF_START_SUM]; end_sum += q[PF_END_SUM]; waves += q[PF_WAVES];
            }
            if (!waves) continue;
            const double tick_ms = 1e-5;                         // 100 MHz
            used++;
            waves_sum += (double)waves;
            span += (double)(end_max_t - enter_min) * tick_ms;
            start_mean += (double)(long long)(start_sum - waves * enter_min) / (double)waves * tick_ms;
            start_max += (double)(start_max_t - enter_min) * tick_ms;
            end_min += (double)(end_min_t - enter_min) * tick_ms;
            end_mean += (double)(long long)(end_sum - waves * enter_min) / (double)waves * tick_ms;
        }
        if (!used) return 0;
        out->launches = used;
        out->waves = (uint32_t)(waves_sum / used + 0.5);
        out->span_ms = (float)(span / used);
        out->start_mean_ms = (float)(start_mean / used);
        out->start_max_ms = (float)(start_max / used);
        out->end_min_ms = (float)(end_min / used);
        out->end_mean_ms = (float)(end_mean / used);
        return 0;
    });
}

// Did any march wave give a segment up (march_group)?  Read wherever the host waits for the device anyway: with the
// statistics, and at the end of start_ray_tracing.  Never seen; a render it happened in is incomplete and is not returned.
static int march_error_check(photon_scene *scene) {
    unsigned e = 0;
    PH_CHECK(hipMemcpy(&e, scene->d_error, sizeof e, hipMemcpyDeviceToHost));
    if (!e) return 0;
    fprintf(stderr, "photon: %u hand-off errors between the segments of a march (a wave gave up waiting for the previous segment of its group, "
                    "or read a stale ray state): this render is not valid\n", e);
    PH_CHECK(hipMemset(scene->d_error, 0, sizeof e));
    return 1;
}

// The raw wave-timing slots of one profiled launch (64 sub-slots x 8 words: PF_*; sub-slot = workgroup index % 64, so
// sub-slot & 7 is the XCD the workgroup ran on): for tools that look at the launch's end per XCD.
extern "C" int photon_scene_march_profile_raw(photon_scene_t *scene, unsigned launch, unsigned long long *out) {
    if (!scene || !out || !scene->d_profile || launch >= std::min(scene->prof_next, kProfileLaunches)) return 1;
    PH_CHECK(hipDeviceSynchronize());
    PH_CHECK(hipMemcpy(out, scene->d_profile + (size_t)launch * kProfileSub * PF_N, (size_t)kProfileSub * PF_N * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

// Sum the counter slots into stats (the caller has made sure the device is done with them).
static int read_counters(photon_scene *scene, bool have_volume, photon_trace_stats_t *stats) {
    std::vector<unsigned long long> slots((size_t)kCounterSlots * kCounterStride);
    PH_CHECK(hipMemcpy(slots.data(), scene->d_counters, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long c[CNT_N] = {};
    for (int k = 0; k < kCounterSlots; k++)
        for (int j = 0; j < CNT_N; j++) c[j] += slots[(size_t)k * kCounterStride + j];
    { const int rc = march_error_check(scene); if (rc) return rc; }
    stats->rays_on_sensor = c[CNT_ON_SENSOR];
    stats->rk_iterations = c[CNT_ITER];
    stats->volume_samples = c[CNT_SAMPLES];
    stats->sensor_taps = c[CNT_TAPS];
    stats->rays_marched = have_volume ? c[CNT_MARCHED] : 0;
    // s_memtime ticks per s_memrealtime tick (100 MHz), over all waves of the march: the clock the kernel ran at
    stats->shader_clock_mhz = c[CNT_REAL] ? (float)((double)c[CNT_CLK] / (double)c[CNT_REAL] * 100.0) : 0.f;
    // mean time a wave spends on one 64-ray group: with 5 waves per SIMD a launch lasts about (groups / 5120) of these
    stats->march_wave_ms = c[CNT_MARCHED] ? (float)((double)c[CNT_REAL] * 1e-5 / ((double)(c[CNT_MARCHED] + 63) / 64.0)) : 0.f;
    return 0;
}

extern "C" int photon_trace(photon_scene_t *scene, const photon_volume_t *vol, int ray_tracing_algorithm,
                            int64_t src_begin, int64_t src_end, float *d_image, void *stream_p,
                            photon_trace_stats_t *stats) {
    if (!scene || !d_image || src_begin < 0 || src_end < src_begin || src_end > scene->dev.num_sources) {
        fprintf(stderr, "photon: photon_trace: bad arguments (sources [%lld,%lld) of %d)\n", (long long)src_begin,
                (long long)src_end, scene ? scene->dev.num_sources : -1);
        return 1;
    }
    if (stats && scene->win_open) {
        fprintf(stderr, "photon: photon_trace: per-call stats inside an open statistics window (photon_scene_stats_begin); "
                        "pass stats = NULL and read them with photon_scene_stats_end\n");
        return 1;
    }
    if (scene->win_open && (hipStream_t)stream_p != scene->win_stream) {
        fprintf(stderr, "photon: photon_trace: a statistics window is open on another stream (its counters were zeroed there)\n");
        return 1;
    }
    if (scene->win_open && scene->win_traces >= kWindowMaxTraces) {
        fprintf(stderr, "photon: photon_trace: more than %u traces in one statistics window; close it with photon_scene_stats_end\n", kWindowMaxTraces);
        return 1;
    }
    return guarded("photon_trace", [&]() -> int {
        hipStream_t stream = (hipStream_t)stream_p;
        const unsigned rps = (unsigned)scene->dev.rays_per_source;
        size_t w0 = 0, w1 = 0;
        if (stats) {
            PH_CHECK(hipMemsetAsync(scene->d_counters, 0, (size_t)kCounterSlots * kCounterStride * sizeof(unsigned long long), stream));
            { const int rc = profile_reset(scene, stream); if (rc) return rc; }
            PH_CHECK(hipEventRecord(scene->ev[0], stream));
        } else if (scene->win_open) {
            { const int rc = window_event(scene, &w0); if (rc) return rc; }
            { const int rc = window_event(scene, &w1); if (rc) return rc; }
            PH_CHECK(hipEventRecord(scene->win_events[w0], stream));
        }
        float march_ms = 0.f;
        const int timed = stats ? 1 : (scene->win_open ? 2 : 0);
        { const int rc = trace_accumulate(scene, vol, ray_tracing_algorithm, src_begin, src_end, stream, timed, &march_ms); if (rc) return rc; }
        { const int rc = end_accumulate(scene, d_image, stream); if (rc) return rc; }
        if (stats) {
            PH_CHECK(hipEventRecord(scene->ev[3], stream));
            PH_CHECK(hipEventSynchronize(scene->ev[3]));
            memset(stats, 0, sizeof *stats);
            { const int rc = read_counters(scene, vol != nullptr, stats); if (rc) return rc; }
            stats->rays_launched = (uint64_t)(src_end - src_begin) * rps;
            stats->march_ms = march_ms;
            stats->traces = 1;
            PH_CHECK(hipEventElapsedTime(&stats->total_ms, scene->ev[0], scene->ev[3]));
        } else if (scene->win_open) {
            PH_CHECK(hipEventRecord(scene->win_events[w1], stream));
            scene->win_total.emplace_back(w0, w1);
            scene->win_rays += (uint64_t)(src_end - src_begin) * rps;
            scene->win_traces += 1;
            scene->win_have_volume = scene->win_have_volume || vol != nullptr;
        }
        return 0;
    });
}

// Statistics over a WINDOW of photon_trace calls without a host synchronisation inside it: _begin zeroes the counters (on
// the stream), every photon_trace(stats = NULL) of this scene up to _end records its events on its stream and lets the
// counters run; _end waits for the stream and returns the sums (march_ms, total_ms: summed over the traces; counters:
// summed over the traces; shader_clock_mhz: over all march waves of the window).
extern "C" int photon_scene_stats_begin(photon_scene_t *scene, void *stream_p) {
    if (!scene) return 1;
    return guarded("photon_scene_stats_begin", [&]() -> int {
        hipStream_t stream = (hipStream_t)stream_p;
        PH_CHECK(hipMemsetAsync(scene->d_counters, 0, (size_t)kCounterSlots * kCounterStride * sizeof(unsigned long long), stream));
        { const int rc = profile_reset(scene, stream); if (rc) return rc; }
        scene->win_used = 0;
        scene->win_march.clear();
        scene->win_total.clear();
        scene->win_rays = 0;
        scene->win_traces = 0;
        scene->win_have_volume = false;
        scene->win_stream = stream;
        scene->win_open = true;
        return 0;
    });
}

extern "C" int photon_scene_stats_end(photon_scene_t *scene, void *stream_p, photon_trace_stats_t *stats) {
    if (!scene || !stats || !scene->win_open) {
        fprintf(stderr, "photon: photon_scene_stats_end: no open statistics window\n");
        return 1;
    }
    return guarded("photon_scene_stats_end", [&]() -> int {
        scene->win_open = false;
        PH_CHECK(hipStreamSynchronize((hipStream_t)stream_p));
        memset(stats, 0, sizeof *stats);
        { const int rc = read_counters(scene, scene->win_have_volume, stats); if (rc) return rc; }
        double march = 0.0, total = 0.0;
        for (const auto &pr : scene->win_march) {
            float ms = 0.f;
            PH_CHECK(hipEventElapsedTime(&ms, scene->win_events[pr.first], scene->win_events[pr.second]));
            march += ms;
        }
        for (const auto &pr : scene->win_total) {
            float ms = 0.f;
            PH_CHECK(hipEventElapsedTime(&ms, scene->win_events[pr.first], scene->win_events[pr.second]));
            total += ms;
        }
        stats->march_ms = (float)march;
        stats->total_ms = (float)total;
        stats->rays_launched = scene->win_rays;
        stats->traces = scene->win_traces;
        return 0;
    });
}


// Device-to-device float4 copy rate (read + write bytes per second, GB/s): what a trivial streaming kernel reaches on
// this GPU -- the "achievable HBM peak" bench.py prints next to the 8 TB/s specification.
extern "C" int photon_measure_copy_gbs(size_t bytes, int reps, double *gbs_out) {
    if (!gbs_out || bytes < 4096 || reps < 1) return 1;
    const size_t n = bytes / sizeof(float4);
    DeviceBuffer<float4> a, b;
    PH_CHECK(a.alloc(n));
    PH_CHECK(b.alloc(n));
    PH_CHECK(hipMemset(a.p, 0, n * sizeof(float4)));
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

// =============================================================================================
// C-ABI: the reference's entry point
// =============================================================================================
namespace {

struct VolumeCache {                // the library stays loaded between photon's calls: keep the
    std::string path;               // uploaded volume, keyed by (file, mtime, size, sampler)
    long long mtime_ns = 0;
    long long size = 0;
    int interpolation = 0;
    int device = -1;
    photon_volume *vol = nullptr;
};
std::mutex g_cache_mutex;                       // guards the map; each entry has its own lock for the (slow) load
struct DeviceCache { std::mutex lock; VolumeCache entry; };
std::map<int, DeviceCache> g_cache;             // one cached volume per device (PHOTON_DEVICES renders on several)

int interpolation_from_env() {
    const char *e = getenv("PHOTON_INTERP");
    if (e && (strcmp(e, "cubic") == 0 || strcmp(e, "2") == 0)) return 2;
    return 1;                       // the reference hard-codes interpolation_scheme = 1 (.cu:3330)
}

// PHOTON_ELEMENT_TRAIN=sequential: the working multi-element train instead of the reference's
// "element 0 for every single-member group, nothing for the others" (.cu:1331-1333, 1049-1272)
int element_train_from_env() {
    const char *e = getenv("PHOTON_ELEMENT_TRAIN");
    return e && (strcmp(e, "sequential") == 0 || strcmp(e, "1") == 0) ? 1 : 0;
}

// PHOTON_SKIP_DOOMED=0 marches every ray like the reference does (photon_scene_set_skip_doomed)
int skip_doomed_from_env() {
    const char *e = getenv("PHOTON_SKIP_DOOMED");
    return !(e && strcmp(e, "0") == 0);
}

// PHOTON_RAY_ORDER=source|lens|auto (photon_scene_set_ray_order)
int ray_order_from_env() {
    const char *e = getenv("PHOTON_RAY_ORDER");
    if (e && strcmp(e, "source") == 0) return 0;
    if (e && strcmp(e, "lens") == 0) return 1;
    return 2;
}

// PHOTON_TEX_WEIGHTS=fixed8|exact: trilinear weights as the reference's texture unit holds them (8 fractional bits:
// the documented arithmetic of the tex3D() the reference calls; default) or as exact f32
int weight_bits_from_env() {
    const char *e = getenv("PHOTON_TEX_WEIGHTS");
    return e && (strcmp(e, "exact") == 0 || strcmp(e, "0") == 0) ? 0 : 8;
}

// The NRRD of one call, parsed at most once on the host however many devices need it (PHOTON_DEVICES): the
// first device thread whose cache misses reads the file, the others build their volume from the same array.
struct SharedDensity {
    std::once_flag once;
    bool ok = false;
    std::string why;
    std::vector<float> rho;
    int dims[3] = {0, 0, 0};
    double spacing[3] = {1, 1, 1}, origin[3] = {0, 0, 0};
};

int cached_volume(const char *path, int interpolation, photon_volume **out, SharedDensity *shared = nullptr) {
    struct stat st;
    if (stat(path, &st) != 0) {
        fprintf(stderr, "photon: failed to open \"%s\"\n", path);
        return 2;
    }
    int device = 0;
    (void)hipGetDevice(&device);
    const long long mt = (long long)st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec;
    DeviceCache *dc;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        dc = &g_cache[device];                  // std::map: references stay valid
    }
    std::lock_guard<std::mutex> lock(dc->lock);
    VolumeCache &c = dc->entry;
    if (c.vol && c.path == path && c.mtime_ns == mt && c.size == (long long)st.st_size && c.interpolation == interpolation) {
        *out = c.vol;
        return 0;
    }
    if (c.vol) { photon_volume_free(c.vol); c.vol = nullptr; }
    photon_volume *v = nullptr;
    int rc;
    if (shared) {
        std::call_once(shared->once, [&]() { shared->ok = parse_nrrd(path, shared->rho, shared->dims, shared->spacing, shared->origin, shared->why); });
        if (!shared->ok) {
            fprintf(stderr, "photon: failed to read NRRD \"%s\": %s\n", path, shared->why.c_str());
            return 2;
        }
        rc = photon_volume_from_density(shared->rho.data(), shared->dims[0], shared->dims[1], shared->dims[2], shared->spacing,
                                        shared->origin, interpolation, &v);
    } else {
        rc = photon_volume_load_nrrd(path, interpolation, &v);
    }
    if (rc) return rc;
    c.path = path; c.mtime_ns = mt; c.size = (long long)st.st_size;
    c.interpolation = interpolation; c.device = device; c.vol = v;
    *out = v;
    return 0;
}

// PHOTON_DEVICES: "all", or a comma-separated list of device ordinals (repeats allowed: "0,0" renders two
// shards side by side on device 0).  Empty = the calling thread's current device only.
std::vector<int> devices_from_env() {
    std::vector<int> out;
    const char *e = getenv("PHOTON_DEVICES");
    if (!e || !*e) return out;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return out;
    if (strcmp(e, "all") == 0) {
        for (int d = 0; d < count; d++) out.push_back(d);
        return out;
    }
    const char *p = e;
    while (*p) {
        char *end = nullptr;
        const long d = strtol(p, &end, 10);
        if (end == p) break;
        if (d < 0 || d >= count) {
            fprintf(stderr, "photon: PHOTON_DEVICES names device %ld, %d present; using the current device\n", d, count);
            out.clear();
            return out;
        }
        out.push_back((int)d);
        p = *end == ',' ? end + 1 : end;
        if (*end && *end != ',') break;
    }
    return out;
}

bool write_dump(const char *dir, const char *prefix, int k, const std::vector<float> &v) {
    char name[64];
    snprintf(name, sizeof name, "%s%04d.bin", prefix, k);               // .cu:3574
    const std::string full = std::string(dir) + "/" + name;
    std::ofstream f(full.c_str(), std::ios::out | std::ios::binary);
    if (!f) { fprintf(stderr, "photon: cannot write %s\n", full.c_str()); return false; }
    f.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(float)));
    f.flush();
    if (!f) { fprintf(stderr, "photon: short write to %s\n", full.c_str()); return false; }
    return true;
}

}  // namespace

// acc[i] += other[i] (f64 sensor accumulators of two shards of one call)
__global__ __launch_bounds__(256) void add_accumulator_kernel(double *__restrict__ acc, const double *__restrict__ other, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += other[i];
}

namespace {

// Arguments of one start_ray_tracing call, as the multi-device path hands them to its workers.
struct CallArgs {
    float lens_pitch, image_distance;
    scattering_data_t *sdp; char *scattering_type_str; lightfield_source_t *lsp;
    int rays_per_source; float beam_wavelength, f_number; int num_elements;
    double (*element_center)[3]; element_data_t *edp; double (*element_planes)[4]; int *sys_index;
    camera_design_t *cam; bool density; char *density_path; int algorithm;
    bool add_pos_noise; float pos_noise_std; bool add_ngrad_noise; float ngrad_noise_std; float ratio;
};

// PHOTON_DEVICES (SURVEY 8e inside ONE call, for photon's single Python process): the sources are cut into
// contiguous, count-balanced blocks, one per listed device; each device thread uploads ONLY its block (plus the
// replicated tables, optics and volume -- the NRRD is parsed once, SharedDensity), renders into its scene's private
// f64 accumulator, and the accumulators are then summed ON THE DEVICES: device k's accumulator travels to the first
// device by hipMemcpyPeer (xGMI when peer access is available) and is added there by a kernel, in device-list order;
// one finalize folds the sum into the caller's image.  Summation stays f64 end to end, one rounding per pixel.
int render_on_devices(const std::vector<int> &devices, const CallArgs &a, float *image_array) {
    const char *e = getenv("PHOTON_NOISE_SEED");
    const uint64_t seed = e ? strtoull(e, nullptr, 0) : 0x5eedULL;
    const long long n_src = a.lsp->num_particles;
    const size_t npix = (size_t)a.cam->x_pixel_number * a.cam->y_pixel_number;
    const size_t K = devices.size();
    std::vector<photon_scene *> scenes(K, nullptr);
    std::vector<int> rcs(K, 0);
    SharedDensity shared;
    std::vector<std::thread> workers;
    for (size_t k = 0; k < K; k++) {
        workers.emplace_back([&, k]() {
            rcs[k] = guarded("start_ray_tracing (device worker)", [&]() -> int {
                const long long b = n_src * (long long)k / (long long)K, e2 = n_src * (long long)(k + 1) / (long long)K;
                if (hipSetDevice(devices[k]) != hipSuccess) return 1;
                lightfield_source_t shard = *a.lsp;                     // this device's block of the caller's arrays
                shard.x += b; shard.y += b; shard.z += b; shard.radiance += b; shard.diameter_index += b;
                shard.num_particles = (int)(e2 - b);
                photon_scene *sc = nullptr;
                if (photon_scene_create(a.lens_pitch, a.image_distance, a.sdp, a.scattering_type_str, &shard, a.rays_per_source,
                                        a.beam_wavelength, a.f_number, a.num_elements, a.element_center, a.edp, a.element_planes,
                                        a.sys_index, a.cam, a.ratio, &sc)) return 2;
                scenes[k] = sc;
                sc->dev.source_base = b;
                photon_scene_set_noise(sc, a.add_pos_noise, a.pos_noise_std, a.density && a.add_ngrad_noise, a.ngrad_noise_std, seed);
                photon_scene_set_element_train(sc, element_train_from_env());
                photon_scene_set_ray_order(sc, ray_order_from_env());
                photon_scene_set_skip_doomed(sc, skip_doomed_from_env());
                photon_volume *v = nullptr;
                int rc = 0;
                if (a.density) rc = cached_volume(a.density_path, interpolation_from_env(), &v, &shared);
                if (!rc && v) photon_volume_set_weight_bits(v, weight_bits_from_env());
                const auto tw = std::chrono::steady_clock::now();
                if (!rc) rc = trace_accumulate(sc, v, a.algorithm, 0, e2 - b, nullptr, false, nullptr);
                if (!rc && hipDeviceSynchronize() != hipSuccess) rc = 4;
                if (!rc) rc = march_error_check(sc);
                if (!rc && verbose())
                    fprintf(stderr, "photon: device %d: sources [%lld, %lld) traced in %.3f ms\n", devices[k], b, e2,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count());
                return rc;
            });
        });
    }
    for (auto &w : workers) w.join();
    int rc = 0;
    for (size_t k = 0; k < K && !rc; k++)
        if (rcs[k]) { fprintf(stderr, "photon: device %d failed (%d); image left untouched\n", devices[k], rcs[k]); rc = rcs[k]; }
    // reduce onto the first device, fold into the caller's image there
    {
        DeviceBuffer<double> d_peer;
        DeviceBuffer<float> d_img;
        auto check = [&](hipError_t err, int line) {
            if (err != hipSuccess && !rc) {
                fprintf(stderr, "photon: HIP error %d (%s) at %s:%d; image left untouched\n", (int)err, hipGetErrorString(err), __FILE__, line);
                rc = (int)err;
            }
            return rc == 0;
        };
        if (!rc && check(hipSetDevice(devices[0]), __LINE__)) {
            const dim3 grid((unsigned)((npix + 255) / 256)), block(256);
            for (size_t k = 1; k < K && !rc; k++) {
                const double *other = scenes[k]->d_acc;
                if (devices[k] != devices[0]) {
                    if (!d_peer.p && !check(d_peer.alloc(npix), __LINE__)) break;
                    // direct xGMI copy when the first device may map the other's memory; otherwise the runtime stages the
                    // copy through the host -- correct, slower, and said out loud
                    int can = 0;
                    bool direct = false;
                    const hipError_t ce = hipDeviceCanAccessPeer(&can, devices[0], devices[k]);
                    if (ce == hipSuccess && can) {
                        const hipError_t pe = hipDeviceEnablePeerAccess(devices[k], 0);
                        direct = pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled;
                        if (pe != hipSuccess) (void)hipGetLastError();
                        if (!direct)
                            fprintf(stderr, "photon: hipDeviceEnablePeerAccess(device %d from device %d) failed: %s; that accumulator is copied through host staging\n",
                                    devices[k], devices[0], hipGetErrorString(pe));
                    } else {
                        if (ce != hipSuccess) (void)hipGetLastError();
                        fprintf(stderr, "photon: device %d cannot access device %d as a peer (%s); that accumulator is copied through host staging\n",
                                devices[0], devices[k], ce == hipSuccess ? "hipDeviceCanAccessPeer: no" : hipGetErrorString(ce));
                    }
                    const auto tc = std::chrono::steady_clock::now();
                    if (!check(hipMemcpyPeer(d_peer.p, devices[0], other, devices[k], npix * sizeof(double)), __LINE__)) break;
                    if (verbose())
                        fprintf(stderr, "photon: accumulator of device %d -> device %d: %s copy of %.1f MiB, %.3f ms\n", devices[k], devices[0],
                                direct ? "direct peer" : "staged", npix * sizeof(double) / 1048576.0,
                                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc).count());
                    other = d_peer.p;
                }
                hipLaunchKernelGGL(add_accumulator_kernel, grid, block, 0, 0, scenes[0]->d_acc, other, npix);
                if (!check(hipGetLastError(), __LINE__)) break;
                if (!check(hipDeviceSynchronize(), __LINE__)) break;        // d_peer is reused by the next device
            }
            if (!rc && check(d_img.alloc(npix), __LINE__) &&
                check(hipMemcpy(d_img.p, image_array, npix * sizeof(float), hipMemcpyHostToDevice), __LINE__)) {     // .cu:3309
                const int frc = end_accumulate(scenes[0], d_img.p, nullptr);
                if (frc && !rc) rc = frc;
                if (!rc && check(hipDeviceSynchronize(), __LINE__))
                    check(hipMemcpy(image_array, d_img.p, npix * sizeof(float), hipMemcpyDeviceToHost), __LINE__);      // .cu:3675
            }
        }
    }
    for (size_t k = 0; k < K; k++)
        if (scenes[k]) { (void)hipSetDevice(devices[k]); photon_scene_free(scenes[k]); }
    return rc;
}

}  // namespace

static void start_ray_tracing_impl(float lens_pitch, float image_distance, scattering_data_t *scattering_data_p,
                                  char *scattering_type_str, lightfield_source_t *lightfield_source_p,
                                  int lightray_number_per_particle, float beam_wavelength, float aperture_f_number,
                                  int num_elements, double (*element_center)[3], element_data_t *element_data_p,
                                  double (*element_plane_parameters)[4], int *element_system_index,
                                  camera_design_t *camera_design_p, float *image_array,
                                  bool simulate_density_gradients, char *density_grad_filename, bool save_lightrays,
                                  char *lightray_position_save_path, char *lightray_direction_save_path,
                                  int num_lightrays_save, int ray_tracing_algorithm, bool add_pos_noise,
                                  float pos_noise_std, bool add_ngrad_noise, float ngrad_noise_std,
                                  float ray_cone_pitch_ratio, bool save_intermediate_ray_data,
                                  int num_intermediate_positions_save) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!image_array || !camera_design_p || !lightfield_source_p) {
        fprintf(stderr, "photon: start_ray_tracing: null argument; image left untouched\n");
        return;
    }
    const bool dumping = save_lightrays && num_lightrays_save > 0;
    int caller_device = 0;                                              // the caller's current device is restored on every path
    const bool have_caller_device = hipGetDevice(&caller_device) == hipSuccess;
    struct RestoreDevice { bool on; int dev; ~RestoreDevice() { if (on) (void)hipSetDevice(dev); } } restore{have_caller_device, caller_device};
    {   // PHOTON_DEVICES: shard the sources of one call over several GPUs.  Ray dumps keep the
        // reference's chunk -> file mapping and stay on one device.
        const std::vector<int> devices = devices_from_env();
        if (devices.size() > 1 && !dumping) {
            const CallArgs a{lens_pitch, image_distance, scattering_data_p, scattering_type_str, lightfield_source_p,
                             lightray_number_per_particle, beam_wavelength, aperture_f_number, num_elements, element_center,
                             element_data_p, element_plane_parameters, element_system_index, camera_design_p,
                             simulate_density_gradients, density_grad_filename, ray_tracing_algorithm, add_pos_noise,
                             pos_noise_std, add_ngrad_noise, ngrad_noise_std, ray_cone_pitch_ratio};
            const int rc = render_on_devices(devices, a, image_array);
            if (!rc && verbose()) {
                const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                const long long n_src = lightfield_source_p->num_particles;
                printf("photon: %lld sources x %d rays on %zu devices in %.3f s (%.2f Mrays/s incl. transfers)\n", n_src,
                       lightray_number_per_particle, devices.size(), sec, n_src * (double)lightray_number_per_particle / sec * 1e-6);
            }
            return;
        }
        if (!devices.empty() && hipSetDevice(devices[0]) != hipSuccess) {
            fprintf(stderr, "photon: cannot select device %d; image left untouched\n", devices[0]);
            return;
        }
    }
    photon_scene *scene = nullptr;
    float *d_image = nullptr, *d_fpos = nullptr, *d_fdir = nullptr, *d_ipos = nullptr, *d_idir = nullptr;
    // PHOTON_VERBOSE: where a call's time goes beside the trace itself (scene upload, volume, image in / out, frees)
    auto t_prev = t0;
    double t_scene = 0, t_volume = 0, t_image_in = 0, t_trace = 0, t_image_out = 0;
    auto lap = [&](double &acc) { const auto now = std::chrono::steady_clock::now(); acc += std::chrono::duration<double, std::milli>(now - t_prev).count(); t_prev = now; };
    auto cleanup = [&]() {
        if (scene) photon_scene_free(scene);
        pool_free(d_image);
        pool_free(d_fpos);
        pool_free(d_fdir);
        pool_free(d_ipos);
        pool_free(d_idir);
    };
#define PH_VOID(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { fprintf(stderr, "photon: HIP error %d (%s) at %s:%d; image left untouched\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); cleanup(); return; } } while (0)
    if (photon_scene_create(lens_pitch, image_distance, scattering_data_p, scattering_type_str, lightfield_source_p,
                            lightray_number_per_particle, beam_wavelength, aperture_f_number, num_elements,
                            element_center, element_data_p, element_plane_parameters, element_system_index,
                            camera_design_p, ray_cone_pitch_ratio, &scene)) {
        fprintf(stderr, "photon: scene upload failed; image left untouched\n");
        return;
    }
    lap(t_scene);
    {   // noise hooks: same switches as the reference; seed from the environment instead of time(NULL)
        const char *e = getenv("PHOTON_NOISE_SEED");
        const uint64_t seed = e ? strtoull(e, nullptr, 0) : 0x5eedULL;
        // gradient noise only exists inside the volume march (Euler, .h:853-863)
        photon_scene_set_noise(scene, add_pos_noise, pos_noise_std, simulate_density_gradients && add_ngrad_noise,
                               ngrad_noise_std, seed);
        photon_scene_set_element_train(scene, element_train_from_env());
        photon_scene_set_ray_order(scene, ray_order_from_env());
        photon_scene_set_skip_doomed(scene, skip_doomed_from_env());
    }
    photon_volume *vol = nullptr;
    if (simulate_density_gradients) {
        if (cached_volume(density_grad_filename, interpolation_from_env(), &vol)) { cleanup(); return; }
        photon_volume_set_weight_bits(vol, weight_bits_from_env());
    }
    lap(t_volume);
    const int W = camera_design_p->x_pixel_number, H = camera_design_p->y_pixel_number;
    const size_t npix = (size_t)W * H;
    PH_VOID(pool_malloc((void **)&d_image, npix * sizeof(float)));
    PH_VOID(hipMemcpy(d_image, image_array, npix * sizeof(float), hipMemcpyHostToDevice));     // .cu:3309
    lap(t_image_in);

    const long long num_particles = lightfield_source_p->num_particles;
    const long long rps = lightray_number_per_particle;
    int rc = 0;
    if (dumping) {
        // the reference's chunking decides which rays land in which pos_/dir_ file (.cu:3366-3372,
        // 3515-3611): chunks of source_point_number sources, one file pair per chunk
        long long chunk = lightfield_source_p->source_point_number;
        if (num_particles < chunk) chunk = num_particles;
        if (chunk < 1) chunk = 1;
        if ((unsigned long long)(chunk * rps) > kMaxRaysPerLaunch) {
            fprintf(stderr, "photon: source_point_number*rays exceeds %u rays per launch; image left untouched\n", kMaxRaysPerLaunch);
            cleanup();
            return;
        }
        const size_t nsave = (size_t)num_lightrays_save * 3;
        PH_VOID(pool_malloc((void **)&d_fpos, nsave * sizeof(float)));
        PH_VOID(pool_malloc((void **)&d_fdir, nsave * sizeof(float)));
        std::vector<float> host(nsave);
        // intermediate dumps ride on the same chunking (.cu:3484-3492, 3535-3546, 3613-3670)
        const bool inter = simulate_density_gradients && save_intermediate_ray_data && num_intermediate_positions_save > 0;
        const size_t ninter = inter ? nsave * (size_t)num_intermediate_positions_save : 0;
        std::vector<float> host_inter(ninter);
        if (inter) {
            PH_VOID(pool_malloc((void **)&d_ipos, ninter * sizeof(float)));
            PH_VOID(pool_malloc((void **)&d_idir, ninter * sizeof(float)));
        }
        const long long kmax = (num_particles + chunk - 1) / chunk;
        rc = begin_accumulate(scene, nullptr);
        for (long long k = 0; k < kmax && rc == 0; k++) {
            PH_VOID(hipMemset(d_fpos, 0xFF, nsave * sizeof(float)));    // all-ones = NaN (.cu:3527-3533)
            PH_VOID(hipMemset(d_fdir, 0xFF, nsave * sizeof(float)));
            if (inter) {
                PH_VOID(hipMemset(d_ipos, 0xFF, ninter * sizeof(float)));
                PH_VOID(hipMemset(d_idir, 0xFF, ninter * sizeof(float)));
            }
            const DumpDev dump{d_fpos, d_fdir, num_lightrays_save, d_ipos, d_idir, inter ? num_intermediate_positions_save : 0};
            rc = launch_chunk(scene, vol, ray_tracing_algorithm, k * chunk, std::min(num_particles, (k + 1) * chunk),
                              dump, nullptr, nullptr, nullptr);
            if (rc) break;
            bool wrote = true;                                          // a dump that cannot be written fails the call
            PH_VOID(hipMemcpy(host.data(), d_fpos, nsave * sizeof(float), hipMemcpyDeviceToHost));
            wrote = write_dump(lightray_position_save_path, "pos_", (int)k, host) && wrote;
            PH_VOID(hipMemcpy(host.data(), d_fdir, nsave * sizeof(float), hipMemcpyDeviceToHost));
            wrote = write_dump(lightray_direction_save_path, "dir_", (int)k, host) && wrote;
            if (inter) {
                PH_VOID(hipMemcpy(host_inter.data(), d_ipos, ninter * sizeof(float), hipMemcpyDeviceToHost));
                wrote = write_dump(lightray_position_save_path, "intermediate_pos_", (int)k, host_inter) && wrote;
                PH_VOID(hipMemcpy(host_inter.data(), d_idir, ninter * sizeof(float), hipMemcpyDeviceToHost));
                wrote = write_dump(lightray_direction_save_path, "intermediate_dir_", (int)k, host_inter) && wrote;
            }
            if (!wrote) rc = 5;
        }
        if (rc == 0) rc = end_accumulate(scene, d_image, nullptr);
    } else {
        if (simulate_density_gradients && save_intermediate_ray_data)
            fprintf(stderr, "photon: warning: save_intermediate_ray_data needs save_lightrays with num_lightrays_save > 0 "
                            "(the reference sizes the intermediate buffers by it, .cu:3488); nothing recorded\n");
        rc = photon_trace(scene, vol, ray_tracing_algorithm, 0, num_particles, d_image, nullptr, nullptr);
    }
    if (rc) {
        fprintf(stderr, "photon: trace failed (%d); image left untouched\n", rc);
        cleanup();
        return;
    }
    PH_VOID(hipDeviceSynchronize());
    lap(t_trace);
    if (march_error_check(scene)) {
        fprintf(stderr, "photon: trace failed; image left untouched\n");
        cleanup();
        return;
    }
    PH_VOID(hipMemcpy(image_array, d_image, npix * sizeof(float), hipMemcpyDeviceToHost));     // .cu:3675
    lap(t_image_out);
#undef PH_VOID
    cleanup();
    if (verbose()) {
        double t_free = 0;
        lap(t_free);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("photon: %lld sources x %lld rays in %.3f s (%.2f Mrays/s incl. transfers)\n", num_particles, rps, s,
               num_particles * rps / s * 1e-6);
        printf("photon:   scene upload %.2f ms, volume %.2f, image in %.2f, trace (launches + wait) %.2f, image out %.2f, frees %.2f\n",
               t_scene, t_volume, t_image_in, t_trace, t_image_out, t_free);
    }
}

// The exported symbol: no C++ exception crosses the C boundary.
extern "C" void start_ray_tracing(float lens_pitch, float image_distance, scattering_data_t *scattering_data_p,
                                  char *scattering_type_str, lightfield_source_t *lightfield_source_p,
                                  int lightray_number_per_particle, float beam_wavelength, float aperture_f_number,
                                  int num_elements, double (*element_center)[3], element_data_t *element_data_p,
                                  double (*element_plane_parameters)[4], int *element_system_index,
                                  camera_design_t *camera_design_p, float *image_array,
                                  bool simulate_density_gradients, char *density_grad_filename, bool save_lightrays,
                                  char *lightray_position_save_path, char *lightray_direction_save_path,
                                  int num_lightrays_save, int ray_tracing_algorithm, bool add_pos_noise,
                                  float pos_noise_std, bool add_ngrad_noise, float ngrad_noise_std,
                                  float ray_cone_pitch_ratio, bool save_intermediate_ray_data,
                                  int num_intermediate_positions_save) {
    (void)guarded("start_ray_tracing", [&]() -> int {
        start_ray_tracing_impl(lens_pitch, image_distance, scattering_data_p, scattering_type_str, lightfield_source_p,
                               lightray_number_per_particle, beam_wavelength, aperture_f_number, num_elements, element_center,
                               element_data_p, element_plane_parameters, element_system_index, camera_design_p, image_array,
                               simulate_density_gradients, density_grad_filename, save_lightrays, lightray_position_save_path,
                               lightray_direction_save_path, num_lightrays_save, ray_tracing_algorithm, add_pos_noise,
                               pos_noise_std, add_ngrad_noise, ngrad_noise_std, ray_cone_pitch_ratio,
                               save_intermediate_ray_data, num_intermediate_positions_save);
        return 0;
    });
}
