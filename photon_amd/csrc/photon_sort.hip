// photon_sort.hip - spatial (Morton) order of a range of light-field sources, built on the device.
//
// Lens-major launches (photon_trace.hip, use_lens_major) put 64 neighbouring sources aimed at one lens point into
// a wave; "neighbouring" = consecutive in the Morton order of (x, y) on a 2^16 x 2^16 grid over the range's
// bounding box.  start_ray_tracing builds a new scene on every call, so the order is on the per-image path of
// every PIV-through-volume frame: it is computed where the sources already are (HBM) -- bounding box by a
// block reduce + ordered-integer atomics, keys, then a stable LSD radix sort of (key, index) pairs (rocPRIM via
// hipCUB: a utility off the hot path; stable, so equal keys keep the caller's order and the permutation is
// deterministic) -- with no host round trip of the coordinates.  Speed only: the image is a sum over sources.
//
// Own translation unit: the sort's templates do not ride in the march kernels' compile.
#include "photon_sort.hpp"
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cstdint>
#include <cstdio>

namespace {

// order-preserving map float -> uint32 (NaN never enters: filtered by the caller of enc)
__device__ __forceinline__ unsigned enc(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec(unsigned e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

// box[0..3] = enc(min x), enc(max x), enc(min y), enc(max y); initialised to {~0, 0, ~0, 0}
__global__ __launch_bounds__(256) void bbox_kernel(const float *__restrict__ x, const float *__restrict__ y, long long n,
                                                   unsigned *box) {
    unsigned lo_x = 0xffffffffu, hi_x = 0u, lo_y = 0xffffffffu, hi_y = 0u;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float xv = x[i], yv = y[i];
        if (xv == xv) { const unsigned e = enc(xv); lo_x = min(lo_x, e); hi_x = max(hi_x, e); }
        if (yv == yv) { const unsigned e = enc(yv); lo_y = min(lo_y, e); hi_y = max(hi_y, e); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo_x = min(lo_x, (unsigned)__shfl_xor((int)lo_x, o, 64)); hi_x = max(hi_x, (unsigned)__shfl_xor((int)hi_x, o, 64));
        lo_y = min(lo_y, (unsigned)__shfl_xor((int)lo_y, o, 64)); hi_y = max(hi_y, (unsigned)__shfl_xor((int)hi_y, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&box[0], lo_x); atomicMax(&box[1], hi_x);
        atomicMin(&box[2], lo_y); atomicMax(&box[3], hi_y);
    }
}

__device__ __forceinline__ unsigned spread16(unsigned v) {              // 16 bits -> every other bit of 32
    v = (v | (v << 8)) & 0x00FF00FFu; v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u;
    return v;
}

__global__ __launch_bounds__(256) void morton_keys_kernel(const float *__restrict__ x, const float *__restrict__ y, long long n,
                                                          const unsigned *__restrict__ box, int first, unsigned *keys, int *idx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x0 = dec(box[0]), x1 = dec(box[1]), y0 = dec(box[2]), y1 = dec(box[3]);
    // ONE scale for both axes (the longer side of the box spans the 2^16 cells): the Z-order's cells are squares in the
    // sources' own units whatever the shape of the launched range.  Scaled per axis -- round 3's form -- a range that is a
    // strip of the field (one GPU's shard of a tile-sorted particle list: 60 mm x 7.5 mm) got cells eight times longer than
    // high, its waves' 64 neighbouring sources spanned patches eight times wider, and the eighth of C5 marched at 64 % of the
    // whole job's rate per ray.
    const double ext = fmax((double)x1 - x0, (double)y1 - y0);
    const double fx = ext > 0.0 ? 65535.0 / ext : 0.0, fy = fx;
    const double qx = ((double)x[i] - x0) * fx, qy = ((double)y[i] - y0) * fy;
    const unsigned ix = qx == qx ? (unsigned)qx : 0u, iy = qy == qy ? (unsigned)qy : 0u;    // NaN -> 0
    keys[i] = spread16(ix & 0xffffu) | (spread16(iy & 0xffffu) << 1);
    idx[i] = first + (int)i;
}


__global__ void init_box_kernel(unsigned *box) {
    if (threadIdx.x < 4) box[threadIdx.x] = (threadIdx.x & 1) ? 0u : 0xffffffffu;     // {min x, max x, min y, max y} encoded
}

}  // namespace

void photon_sort_scratch_free(photon_sort_scratch *s) {
    if (!s) return;
    if (s->box) (void)hipFree(s->box);
    if (s->keys) (void)hipFree(s->keys);
    if (s->idx) (void)hipFree(s->idx);
    if (s->tmp) (void)hipFree(s->tmp);
    *s = photon_sort_scratch{};
}

int photon_morton_order(const float *d_x, const float *d_y, int first, long long n, int *d_perm_out, hipStream_t stream,
                        photon_sort_scratch *sc) {
    if (n <= 0) return 0;
    hipError_t e = hipSuccess;
#define PS_CHECK(expr) do { e = (expr); if (e != hipSuccess) { fprintf(stderr, "photon: HIP error %d (%s) at %s:%d\n", (int)e, hipGetErrorString(e), __FILE__, __LINE__); return (int)e; } } while (0)
    size_t need_tmp = 0;
    PS_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, need_tmp, (unsigned *)nullptr, (unsigned *)nullptr, (int *)nullptr, (int *)nullptr, (int)n, 0, 32, stream));
    if (!sc->box) PS_CHECK(hipMalloc((void **)&sc->box, 4 * sizeof(unsigned)));
    if (sc->capacity < (size_t)n) {
        if (sc->keys) { (void)hipFree(sc->keys); sc->keys = nullptr; }
        if (sc->idx) { (void)hipFree(sc->idx); sc->idx = nullptr; }
        sc->capacity = 0;
        PS_CHECK(hipMalloc((void **)&sc->keys, 2 * (size_t)n * sizeof(unsigned)));
        PS_CHECK(hipMalloc((void **)&sc->idx, (size_t)n * sizeof(int)));
        sc->capacity = (size_t)n;
    }
    if (sc->tmp_bytes < need_tmp || !sc->tmp) {
        if (sc->tmp) { (void)hipFree(sc->tmp); sc->tmp = nullptr; }
        sc->tmp_bytes = 0;
        PS_CHECK(hipMalloc(&sc->tmp, need_tmp ? need_tmp : 16));
        sc->tmp_bytes = need_tmp ? need_tmp : 16;
    }
    unsigned *keys_out = sc->keys + sc->capacity;
    hipLaunchKernelGGL(init_box_kernel, dim3(1), dim3(64), 0, stream, sc->box);
    PS_CHECK(hipGetLastError());
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(bbox_kernel, dim3(blocks < 1024u ? blocks : 1024u), dim3(256), 0, stream, d_x + first, d_y + first, n, sc->box);
    PS_CHECK(hipGetLastError());
    hipLaunchKernelGGL(morton_keys_kernel, dim3(blocks), dim3(256), 0, stream, d_x + first, d_y + first, n, sc->box, first, sc->keys, sc->idx);
    PS_CHECK(hipGetLastError());
    size_t tmp_bytes = sc->tmp_bytes;
    PS_CHECK(hipcub::DeviceRadixSort::SortPairs(sc->tmp, tmp_bytes, sc->keys, keys_out, sc->idx, d_perm_out, (int)n, 0, 32, stream));
#undef PS_CHECK
    return 0;
}
