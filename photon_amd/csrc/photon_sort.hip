// photon_sort.hip - spatial (Morton) order of a range of light-field sources, built on the device.
//
// Lens-major launches (photon_trace.hip, use_lens_major) put 64 neighbouring sources aimed at one lens point into
// a wave; "neighbouring" = consecutive in the Morton order of (x, y) on a 2^16 x 2^16 grid over the range's
// bounding box.  start_ray_tracing builds a new scene on every call, so the order is on the per-image path of
// every PIV-through-volume frame: it is computed where the sources already are (HBM) -- bounding box by a
// block reduce + ordered-integer atomics, keys, then a STABLE least-significant-digit radix sort of (key, index) pairs,
// hand-written since round 6 (rounds 2-5 called rocPRIM through hipCUB): four passes of eight bits, each a histogram
// kernel, a row scan of the (digit, tile) counts and a scatter in which ONE WAVE owns a tile of 256 .. 4096 pairs and ranks it 64
// pairs at a time with eight ballots per pair (the lanes that share a digit), so equal keys keep the caller's order and
// the permutation is deterministic -- with no host round trip of the coordinates.  Speed only: the image is a sum over
// sources.
//
// Own translation unit: nothing of it rides in the march kernels' compile.
#include "photon_sort.hpp"
#include <hip/hip_runtime.h>

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
    // one set of atomics per BLOCK (the four waves meet in LDS first) and at most 128 blocks: with one set per wave of 1024
    // blocks the 16 000 same-address atomics were all this kernel did (91 us at 1.25e5 sources, 180 us at 2.5e5)
    __shared__ unsigned part[4][4];
    if ((threadIdx.x & 63) == 0) { unsigned *p = part[threadIdx.x >> 6]; p[0] = lo_x; p[1] = hi_x; p[2] = lo_y; p[3] = hi_y; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { lo_x = min(lo_x, part[w][0]); hi_x = max(hi_x, part[w][1]); lo_y = min(lo_y, part[w][2]); hi_y = max(hi_y, part[w][3]); }
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

// ---- stable LSD radix sort of (key, index) pairs, 8 bits per pass ----------------------------------------------------------
// A TILE is `tile` consecutive pairs (radix_tile), owned by one 64-lane workgroup in both kernels of a pass.
constexpr int kRadixBits = 8, kRadix = 1 << kRadixBits, kTileMax = 4096;
// pairs per tile: a multiple of 64 chosen per sort so that a launch has ~500 tiles (one wave each) whatever n -- at 4096 a
// shard of 1.25e5 sources kept 31 waves busy for 64 chunks each (scatter 40 us per pass); at 320 it is 391 waves x 5 chunks
inline int radix_tile(long long n) {
    long long t = (n / 480 + 63) / 64 * 64;
    return (int)(t < 256 ? 256 : (t > kTileMax ? kTileMax : t));
}

// counts[digit * n_tiles + tile] = pairs of the tile whose key has that digit (digit-major: the scan below runs over it as it lies)
__global__ __launch_bounds__(64) void radix_hist_kernel(const unsigned *__restrict__ keys, long long n, int shift, unsigned n_tiles, int tile,
                                                        unsigned *__restrict__ counts) {
    __shared__ unsigned h[kRadix];
    for (int d = threadIdx.x; d < kRadix; d += 64) h[d] = 0u;
    __syncthreads();
    const long long base = (long long)blockIdx.x * tile;
    for (int j = threadIdx.x; j < tile; j += 64) {
        const long long i = base + j;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & (kRadix - 1)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < kRadix; d += 64) counts[(size_t)d * n_tiles + blockIdx.x] = h[d];
}

// The (digit, tile) counts become offsets in two steps.  Here: every ROW (one digit's tiles) is scanned on its own by one wave
// -- 256 waves, coalesced loads, a shuffle scan per 64 counts with a running carry -- and its total written behind the counts;
// the scatter kernel's prologue then scans the 256 row totals itself (four loads and four wave scans per tile: cheaper than
// a third launch, and than the single-workgroup scan of the first version: 31 - 85 us per pass of shuffle latency on one CU).
__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v) {
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = (unsigned)__shfl_up((int)v, o, 64);
        if (lane >= (unsigned)o) v += u;
    }
    return v;
}
__global__ __launch_bounds__(256) void radix_rowscan_kernel(unsigned *__restrict__ counts, unsigned n_tiles) {
    const unsigned lane = threadIdx.x & 63u, d = blockIdx.x * 4u + (threadIdx.x >> 6);        // 64 blocks x 4 waves = 256 rows
    unsigned *row = counts + (size_t)d * n_tiles;
    unsigned carry = 0;
    for (unsigned t0 = 0; t0 < n_tiles; t0 += 64u) {
        const unsigned t = t0 + lane;
        const unsigned c = t < n_tiles ? row[t] : 0u;
        const unsigned inc = wave_inclusive_scan(c);
        if (t < n_tiles) row[t] = carry + inc - c;
        carry += (unsigned)__shfl((int)inc, 63, 64);
    }
    if (lane == 0) counts[(size_t)kRadix * n_tiles + d] = carry;       // the row totals live behind the counts
}

// offsets[digit * n_tiles + tile] = where the tile's first pair with that digit goes.  The wave walks its tile 64 pairs at a
// time IN ORDER; a pair's rank among the tile's pairs of its digit = those of earlier chunks (next[digit], kept in LDS) + the
// lanes BELOW it in this chunk that share its digit (eight ballots tell which lanes do): stable.
__global__ __launch_bounds__(64) void radix_scatter_kernel(const unsigned *__restrict__ keys_in, const int *__restrict__ vals_in,
                                                           unsigned *__restrict__ keys_out, int *__restrict__ vals_out, long long n,
                                                           int shift, unsigned n_tiles, int tile, const unsigned *__restrict__ offsets) {
    __shared__ unsigned next[kRadix];
    const unsigned lane = threadIdx.x;
    {   // where digit d starts in the output = exclusive scan of the row totals; + where this tile's share of the row starts
        unsigned run = 0;
        for (int d0 = 0; d0 < kRadix; d0 += 64) {
            const unsigned c = offsets[(size_t)kRadix * n_tiles + d0 + lane];
            const unsigned inc = wave_inclusive_scan(c);
            next[d0 + lane] = run + inc - c + offsets[(size_t)(d0 + lane) * n_tiles + blockIdx.x];
            run += (unsigned)__shfl((int)inc, 63, 64);
        }
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    const long long base = (long long)blockIdx.x * tile;
    for (int j = 0; j < tile; j += 64) {
        const long long i = base + j + lane;
        const bool have = i < n;
        const unsigned key = have ? keys_in[i] : 0u;
        const int val = have ? vals_in[i] : 0;
        const unsigned digit = (key >> shift) & (kRadix - 1);
        unsigned long long peers = __ballot(have);                      // lanes of this chunk with the same digit
#pragma unroll
        for (int b = 0; b < kRadixBits; b++) {
            const unsigned long long m = __ballot((digit >> b) & 1u);
            peers &= ((digit >> b) & 1u) ? m : ~m;
        }
        unsigned dst = 0;
        if (have) dst = next[digit] + (unsigned)__popcll(peers & below);
        __syncthreads();                                                // every lane has read next[] before the leaders move it on
        if (have && (peers & below) == 0ull) next[digit] += (unsigned)__popcll(peers);     // the lowest lane of each digit
        __syncthreads();
        if (have) { keys_out[dst] = key; vals_out[dst] = val; }
        if (base + j + 64 >= n) break;                                  // wave-uniform: the tile's last chunk
    }
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
    const int tile = radix_tile(n);
    const unsigned n_tiles = (unsigned)((n + tile - 1) / tile);
    const size_t need_tmp = ((size_t)kRadix * n_tiles + kRadix) * sizeof(unsigned);        // the (digit, tile) counts of one pass + the row totals
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
        PS_CHECK(hipMalloc(&sc->tmp, need_tmp));
        sc->tmp_bytes = need_tmp;
    }
    // two buffers of pairs: A = (sc->keys, sc->idx), B = (sc->keys + capacity, d_perm_out).  The keys are written into B and the
    // four passes go B -> A -> B -> A -> B: the sorted indices end where the caller wants them.
    unsigned *keys_a = sc->keys, *keys_b = sc->keys + sc->capacity, *counts = (unsigned *)sc->tmp;
    int *vals_a = sc->idx, *vals_b = d_perm_out;
    hipLaunchKernelGGL(init_box_kernel, dim3(1), dim3(64), 0, stream, sc->box);
    PS_CHECK(hipGetLastError());
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(bbox_kernel, dim3(blocks < 128u ? blocks : 128u), dim3(256), 0, stream, d_x + first, d_y + first, n, sc->box);
    PS_CHECK(hipGetLastError());
    hipLaunchKernelGGL(morton_keys_kernel, dim3(blocks), dim3(256), 0, stream, d_x + first, d_y + first, n, sc->box, first, keys_b, vals_b);
    PS_CHECK(hipGetLastError());
    for (int pass = 0; pass < 32 / kRadixBits; pass++) {
        const unsigned *kin = (pass & 1) ? keys_a : keys_b;
        const int *vin = (pass & 1) ? vals_a : vals_b;
        unsigned *kout = (pass & 1) ? keys_b : keys_a;
        int *vout = (pass & 1) ? vals_b : vals_a;
        const int shift = pass * kRadixBits;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(n_tiles), dim3(64), 0, stream, kin, n, shift, n_tiles, tile, counts);
        PS_CHECK(hipGetLastError());
        hipLaunchKernelGGL(radix_rowscan_kernel, dim3(kRadix / 4), dim3(256), 0, stream, counts, n_tiles);
        PS_CHECK(hipGetLastError());
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(n_tiles), dim3(64), 0, stream, kin, vin, kout, vout, n, shift, n_tiles, tile, counts);
        PS_CHECK(hipGetLastError());
    }
#undef PS_CHECK
    return 0;
}

// Self-test hook (include/parallel_ray_tracing.h): the Morton order of host arrays, through the device path above.
extern "C" int photon_selftest_morton_order(const float *x, const float *y, long long n_total, long long first, long long n, int *perm_out) {
    if (!x || !y || !perm_out || n_total <= 0 || first < 0 || n <= 0 || first + n > n_total || n_total > 0x7fffffffLL) return 1;
    float *dx = nullptr, *dy = nullptr;
    int *dp = nullptr;
    photon_sort_scratch sc;
    int rc = 1;
    if (hipMalloc((void **)&dx, (size_t)n_total * sizeof(float)) == hipSuccess && hipMalloc((void **)&dy, (size_t)n_total * sizeof(float)) == hipSuccess &&
        hipMalloc((void **)&dp, (size_t)n * sizeof(int)) == hipSuccess &&
        hipMemcpy(dx, x, (size_t)n_total * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(dy, y, (size_t)n_total * sizeof(float), hipMemcpyHostToDevice) == hipSuccess) {
        rc = photon_morton_order(dx, dy, (int)first, n, dp, nullptr, &sc);
        if (!rc) rc = hipMemcpy(perm_out, dp, (size_t)n * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;      // waits for the null stream
    }
    (void)hipDeviceSynchronize();
    photon_sort_scratch_free(&sc);
    if (dx) (void)hipFree(dx);
    if (dy) (void)hipFree(dy);
    if (dp) (void)hipFree(dp);
    return rc;
}
