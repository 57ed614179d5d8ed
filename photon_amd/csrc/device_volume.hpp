// device_volume.hpp - refractive-index-gradient volume: samplers and ray integrators (device).
//
// Replaces, for gfx950, the CUDA texture path of the reference:
//   tex3D(tex_data, ...) trilinear fetch      trace_rays_through_density_gradients.h:77,1052,1612-1646
//   cubicTex3D(coeffs3D, ...) tricubic fetch  CubicInterpolationCUDA/code/internal/cubicTex3D_kernel.cu:48-81
//   IntersectWithVolume / lookup / bounds     trace_rays_through_density_gradients.h:100-277
//   euler / rk4                               trace_rays_through_density_gradients.h:743-1291
// There is no texture unit on the path: texels are float4 in HBM (x fastest), addressed and
// filtered in software with a fixed operation order.
#pragma once
#include <float.h>
#include "device_vec.hpp"

namespace photon {

// density_grad_params_t (cuda_codes/parallel_ray_tracing.h:213-252) as the kernels see it
struct VolumeDev {
    f3 min_bound, max_bound;
    int nx, ny, nz;
    float step_size;
    float data_min;
    int interpolation;          // 1 trilinear, 2 tricubic
    float weight_scale;         // trilinear weights: 0 = exact f32, 256 = NVIDIA texture-unit emulation (8 fractional bits)
    float weight_inv;           // 1 / weight_scale (a power of two: exact), 0 when weight_scale is 0
    const f4 *texels;           // grad n (xyz), n-1 (w)   [nz][ny][nx]
    const f4 *coeffs;           // B-spline coefficients    [nz][ny][nx] (interpolation == 2)
};

struct MarchCount { int iterations; int samples; };

__device__ __forceinline__ float lerpf(float a, float b, float t) { return fmaf(t, b - a, a); }
__device__ __forceinline__ f4 lerp4(f4 a, f4 b, float t) {
    return f4{lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t), lerpf(a.w, b.w, t)};
}
// lerp4 with the difference d = b - a already formed (the coherent trilinear tile parks it): fmaf(t, d, a), the same bits
__device__ __forceinline__ f4 lerp4d(f4 a, f4 d, float t) {
    return f4{fmaf(t, d.x, a.x), fmaf(t, d.y, a.y), fmaf(t, d.z, a.z), fmaf(t, d.w, a.w)};
}
// CUDA's linear texture filter keeps the interpolation weights in 9-bit fixed point with 8 fractional bits
// (CUDA C Programming Guide, "Linear Filtering"): that is the arithmetic the reference's tex3D() calls run
// with on its own hardware.  weight_scale = 256 reproduces it (round to nearest), 0 keeps exact f32 weights.
// scale = 2^bits: the division is an exact multiplication by 2^-bits (no f32 divide sequence in the sampler)
// floorf(a * scale + 0.5f) is written as floorf(fmaf(a, scale, 0.5f)): a * 2^bits is exact (a in [0, 1), no
// overflow; a denormal scales exactly too), so the fused form rounds once exactly where the two-step form does --
// the same bits as the CPU checker's two-step form, one instruction fewer per weight.
__device__ __forceinline__ float quant_weight(float a, float scale, float inv) {
    return scale > 0.f ? floorf(fmaf(a, scale, 0.5f)) * inv : a;
}
__device__ __forceinline__ f4 ldtexel(const f4 *p) {
    const float4 v = *reinterpret_cast<const float4 *>(p);      // one global_load_dwordx4
    return f4{v.x, v.y, v.z, v.w};
}

// Trilinear fetch at unnormalised coordinates, clamp addressing: sample point x-0.5, texels
// floor and floor+1 (CUDA "linear filtering" semantics, exact f32 weights).
__device__ __forceinline__ f4 tex3d_linear(const VolumeDev &v, float x, float y, float z) {
    const float xb = x - 0.5f, yb = y - 0.5f, zb = z - 0.5f;
    const float fi = floorf(xb), fj = floorf(yb), fk = floorf(zb);
    const float a = quant_weight(xb - fi, v.weight_scale, v.weight_inv), b = quant_weight(yb - fj, v.weight_scale, v.weight_inv),
                c = quant_weight(zb - fk, v.weight_scale, v.weight_inv);
    const int i0 = clampi((int)fi, 0, v.nx - 1), i1 = clampi((int)fi + 1, 0, v.nx - 1);
    const int j0 = clampi((int)fj, 0, v.ny - 1), j1 = clampi((int)fj + 1, 0, v.ny - 1);
    const int k0 = clampi((int)fk, 0, v.nz - 1), k1 = clampi((int)fk + 1, 0, v.nz - 1);
    const size_t W = v.nx, WH = (size_t)v.nx * v.ny;
    const f4 *t = v.texels;
    const f4 v000 = ldtexel(t + k0 * WH + j0 * W + i0), v100 = ldtexel(t + k0 * WH + j0 * W + i1);
    const f4 v010 = ldtexel(t + k0 * WH + j1 * W + i0), v110 = ldtexel(t + k0 * WH + j1 * W + i1);
    const f4 v001 = ldtexel(t + k1 * WH + j0 * W + i0), v101 = ldtexel(t + k1 * WH + j0 * W + i1);
    const f4 v011 = ldtexel(t + k1 * WH + j1 * W + i0), v111 = ldtexel(t + k1 * WH + j1 * W + i1);
    const f4 c00 = lerp4(v000, v100, a), c10 = lerp4(v010, v110, a);
    const f4 c01 = lerp4(v001, v101, a), c11 = lerp4(v011, v111, a);
    const f4 c0 = lerp4(c00, c10, b), c1 = lerp4(c01, c11, b);
    return lerp4(c0, c1, c);
}

// bspline_weights (CubicInterpolationCUDA/code/internal/bspline_kernel.cu:83-94): the uniform cubic B-spline's four
// weights at fraction f.  w0 and w3 as the reference spells them; the two middle weights 2/3 - f^2 (2 - f) / 2 in the
// Horner form fmaf(f^2, fmaf(0.5, f, -1), 2/3) -- 11 instructions per axis instead of 15, rounded twice instead of four
// times.  The DEFINED form: the CPU checker evaluates the same expressions; both are pinned to an f64
// evaluation of the 64-tap sum (tests/test_oracle_golden.py).  The reference itself never executes this code path
// (interpolation_scheme is hard-wired to trilinear), so there are no reference bits to match.
__device__ __forceinline__ void bspline_weights(float f, float &w0, float &w1, float &w2, float &w3) {
    const float one_frac = 1.0f - f;
    const float squared = f * f;
    const float one_sqd = one_frac * one_frac;
    w0 = 1.0f / 6.0f * one_sqd * one_frac;
    w1 = fmaf(squared, fmaf(0.5f, f, -1.0f), 2.0f / 3.0f);
    w2 = fmaf(one_sqd, fmaf(0.5f, one_frac, -1.0f), 2.0f / 3.0f);
    w3 = 1.0f / 6.0f * squared * f;
}

// Tricubic B-spline fetch on the prefiltered coefficients: the 64-tap sum over texels floor(x-0.5)-1 .. +2
// (clamped) in the SLAB order -- the defined evaluation order of the 64-tap sum, also the CPU checker's: wxy[b][a] = wx[a] * wy[b]; per z-slab one
// 16-tap chain (a product, then 15 fmaf, x innermost); then the z pass.  288 multiply-adds per sample.  This is
// the exact form of what the reference evaluates with 8 hardware trilinear fetches (cubicTex3D_kernel.cu:48-81;
// 64-tap equivalent: cubicTex3D.cu:63-90).
__device__ __forceinline__ f4 tex3d_cubic(const VolumeDev &v, float x, float y, float z) {
    const float xg = x - 0.5f, yg = y - 0.5f, zg = z - 0.5f;
    const float fi = floorf(xg), fj = floorf(yg), fk = floorf(zg);
    float wx[4], wy[4], wz[4];
    bspline_weights(xg - fi, wx[0], wx[1], wx[2], wx[3]);
    bspline_weights(yg - fj, wy[0], wy[1], wy[2], wy[3]);
    bspline_weights(zg - fk, wz[0], wz[1], wz[2], wz[3]);
    int ix[4], iy[4], iz[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        ix[a] = clampi((int)fi - 1 + a, 0, v.nx - 1);
        iy[a] = clampi((int)fj - 1 + a, 0, v.ny - 1);
        iz[a] = clampi((int)fk - 1 + a, 0, v.nz - 1);
    }
    const size_t W = v.nx, WH = (size_t)v.nx * v.ny;
    f4 acc = f4{0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        f4 s = f4{0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const f4 *row = v.coeffs + iz[c] * WH + iy[b] * W;
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const f4 t = ldtexel(row + ix[a]);
                const float w = wx[a] * wy[b];
                if (a == 0 && b == 0) s = f4{w * t.x, w * t.y, w * t.z, w * t.w};
                else s = f4{fmaf(w, t.x, s.x), fmaf(w, t.y, s.y), fmaf(w, t.z, s.z), fmaf(w, t.w, s.w)};
            }
        }
        if (c == 0) acc = f4{wz[0] * s.x, wz[0] * s.y, wz[0] * s.z, wz[0] * s.w};
        else acc = f4{fmaf(wz[c], s.x, acc.x), fmaf(wz[c], s.y, acc.y), fmaf(wz[c], s.z, acc.z), fmaf(wz[c], s.w, acc.w)};
    }
    return acc;
}

// Slab test, restated with the reference's asymmetric z handling (.h:158-179).
__device__ __forceinline__ bool intersect_with_volume(f3 &pos, f3 dir, f3 p1, f3 p2) {
    float tnear = -(FLT_MAX - 1);
    float tfar = FLT_MAX;
    float t1 = (p1.x - pos.x) / dir.x, t2 = (p2.x - pos.x) / dir.x;
    if (t1 > t2) { const float t = t1; t1 = t2; t2 = t; }
    if (t1 > tnear) tnear = t1;
    if (t2 < tfar) tfar = t2;
    if (tnear > tfar) return false;
    if (tfar < 0.0) return false;
    t1 = (p1.y - pos.y) / dir.y; t2 = (p2.y - pos.y) / dir.y;
    if (t1 > t2) { const float t = t1; t1 = t2; t2 = t; }
    if (t1 > tnear) tnear = t1;
    if (t2 < tfar) tfar = t2;
    if (tnear > tfar) return false;
    if (tfar < 0.0) return false;
    t1 = (p1.z - pos.z) / dir.z; t2 = (p2.z - pos.z) / dir.z;
    if (t1 > t2) { const float t = t1; t1 = t2; t2 = t; }
    float t;
    if (t1 >= 0 && t1 > tnear) tnear = t1;
    if (t2 < tfar) tfar = t2;
    if (tnear > tfar) return false;
    if (tfar < 0.0) return false;
    else if (tnear < 0) t = tfar;
    else t = tnear;
    pos.x += dir.x * t; pos.y += dir.y * t; pos.z += dir.z * t;
    return true;
}

__device__ __forceinline__ f3 lookup_index(f3 pos, const VolumeDev &v, f3 scale) {     // .h:195-215
    const f3 off = pos - v.min_bound;
    const f3 fn = mk3(scale.x * off.x, scale.y * off.y, scale.z * off.z);
    return mk3(1 + fn.x * (v.nx - 2), 1 + fn.y * (v.ny - 2), 1 + fn.z * (v.nz - 2));
}
__device__ __forceinline__ bool inside_box(f3 p, const VolumeDev &v, f3 l) {            // .h:217-251
    if (p.x < v.min_bound.x || p.y < v.min_bound.y || p.z < v.min_bound.z ||
        p.x >= v.max_bound.x || p.y >= v.max_bound.y || p.z >= v.max_bound.z) return false;
    if (l.x < 0 || l.y < 0 || l.z < 0 || l.x >= v.nx || l.y >= v.ny || l.z >= v.nz) return false;
    return true;
}
__device__ __forceinline__ bool can_access(const VolumeDev &v, f3 l) {                  // .h:253-277
    return !(l.x < 0 || l.y < 0 || l.z < 0 || l.x >= v.nx || l.y >= v.ny || l.z >= v.nz);
}

// Intermediate ray dumps (save_intermediate_ray_data): position / direction at the start of each
// of the first `slots` iterations, [ray][slot] float3, world frame.  Like the reference only the
// trilinear branches record them (.h:784-790, 1004-1008); unlike it the ray index is bounds-checked
// (the reference indexes a num_lightrays_save-sized buffer with the unchecked thread id).
struct InterDump { float *pos, *dir; int slots, num_save; unsigned ray; };
__device__ __forceinline__ void record_intermediate(const InterDump &d, int loop_ctr, f3 p, f3 q) {
    if (d.pos != nullptr && loop_ctr < d.slots && d.ray < (unsigned)d.num_save) {
        const size_t o = ((size_t)d.ray * d.slots + loop_ctr) * 3;
        d.pos[o] = p.x; d.pos[o + 1] = p.y; d.pos[o + 2] = p.z;
        d.dir[o] = q.x; d.dir[o + 1] = q.y; d.dir[o + 2] = q.z;
    }
}

constexpr int kLoopMax = 10000000;      // loop_ctr_max (.h:765,981)
constexpr int kSpinMax = 1 << 20;       // bound on the reference's uncounted `continue` spins

// linear-branch fetch with the reference's "n-1 below data_min" repair (.h:1052-1065)
__device__ __forceinline__ f4 fetch_linear(const VolumeDev &v, f3 l, const f4 &prev, float ambient,
                                           MarchCount &mc) {
    f4 val = tex3d_linear(v, l.x, l.y, l.z);
    mc.samples++;
    if (val.w < v.data_min) {
        if (prev.w == 0) {
            const f4 t = tex3d_linear(v, l.x, l.y, l.z - 1);
            mc.samples++;
            val = f4{t.x, t.y, t.z, ambient - 1};
        } else {
            val = prev;
        }
    }
    return val;
}

// Sharma-Kumar-Ghatak RK4 of the ray equation (.h:952-1291).  INTERP is a compile-time branch.
template <int INTERP>
__device__ __forceinline__ void rk4(f3 &rpos, f3 &rdir, const VolumeDev &v, f3 scale, MarchCount &mc) {
    const float ambient = 1.000277;
    int loop_ctr = 0, spins = 0;
    f3 pos, lookup, R_n, T_n, A, B, C, D;
    f4 val, val_prev = f4{0, 0, 0, 0};
    float delta_t, current_n;
    while (true) {
        if (loop_ctr > kLoopMax) break;
        pos = rpos;
        lookup = lookup_index(pos, v, scale);
        if (!inside_box(pos, v, lookup) && loop_ctr != 0) break;
        if (!can_access(v, lookup)) {
            pos = pos + v.step_size / (1 + v.data_min) * rdir;
            rpos = pos;
            if (++spins > kSpinMax) break;
            continue;
        }
        if (INTERP == 1) {
            val = fetch_linear(v, lookup, val_prev, ambient, mc);
        } else {
            val = tex3d_cubic(v, lookup.x, lookup.y, lookup.z);
            mc.samples++;
            if (val.w < v.data_min) {
                pos = pos + v.step_size / (1 + v.data_min) * rdir;
                rpos = pos;
                if (++spins > kSpinMax) break;
                continue;
            }
        }
        loop_ctr += 1;
        val.w += 1;
        current_n = val.w;
        R_n = pos;
        delta_t = v.step_size / val.w;
        T_n = val.w * rdir;
        D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
        A = delta_t * D;
        // reference: delta_t/2.0 and 1/8.0*delta_t in double, narrowed to float (.h:1088) --
        // exact powers of two, so the f32 products below are the same values
        pos = R_n + (0.5f * delta_t) * T_n + (0.125f * delta_t) * A;
        lookup = lookup_index(pos, v, scale);
        if (!inside_box(pos, v, lookup)) break;
        if (INTERP == 1) {
            val_prev = val; val_prev.w -= 1;
            val = fetch_linear(v, lookup, val_prev, ambient, mc);
        } else {
            val = tex3d_cubic(v, lookup.x, lookup.y, lookup.z);
            mc.samples++;
        }
        val.w += 1;
        D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
        B = delta_t * D;
        pos = R_n + delta_t * T_n + (0.5f * delta_t) * B;                  // .h:1131
        lookup = lookup_index(pos, v, scale);
        if (!inside_box(pos, v, lookup)) break;
        if (INTERP == 1) {
            val_prev = val; val_prev.w -= 1;
            val = fetch_linear(v, lookup, val_prev, ambient, mc);
        } else {
            val = tex3d_cubic(v, lookup.x, lookup.y, lookup.z);
            mc.samples++;
        }
        val.w += 1;
        D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
        C = delta_t * D;
        R_n = R_n + delta_t * (T_n + (float)(1 / 6.0) * (A + 2.0f * B));
        T_n = T_n + (float)(1 / 6.0) * (A + 4.0f * B + C);
        if (INTERP == 1) { val_prev = val; val_prev.w -= 1; }
        rpos = R_n;
        // linear branch divides by the first sample's n (.h:1178), cubic by the last (.h:1276)
        rdir = normalize(T_n / (INTERP == 1 ? current_n : val.w));
        mc.iterations++;
    }
}

// Euler integrator (.h:743-950), noise hook not built.
template <int INTERP>
__device__ __forceinline__ void euler(f3 &rpos, f3 &rdir, const VolumeDev &v, f3 scale, MarchCount &mc) {
    const float ambient = 1.000277;
    int loop_ctr = 0, spins = 0;
    f3 pos, dir, lookup, normal;
    f4 val, val_prev = f4{0, 0, 0, 0};
    while (true) {
        if (loop_ctr > kLoopMax) break;
        pos = rpos; dir = rdir;
        lookup = lookup_index(pos, v, scale);
        if (!inside_box(pos, v, lookup) && loop_ctr != 0) break;
        if (INTERP == 1) {
            if (!can_access(v, lookup)) {
                pos = pos + v.step_size / (1 + v.data_min) * dir;
                rpos = pos;
                if (++spins > kSpinMax) break;
                continue;
            }
            val = fetch_linear(v, lookup, val_prev, ambient, mc);
            const float current_n = 1 + val.w;
            normal = mk3(val.x, val.y, val.z);
            dir = dir + v.step_size * normal;
            pos = pos + v.step_size / current_n * dir;
            rpos = pos; rdir = dir;
            val_prev = val;
            loop_ctr += 1;
        } else {
            val = tex3d_cubic(v, lookup.x, lookup.y, lookup.z);
            mc.samples++;
            if (val.w < v.data_min) {
                pos = pos + v.step_size / (1 + v.data_min) * dir;
                rpos = pos;
                if (++spins > kSpinMax) break;
                continue;
            }
            loop_ctr += 1;
            normal = mk3(val.x, val.y, val.z);
            dir = dir + v.step_size * normal;
            dir = normalize(dir);
            const float n = 1 + val.w;
            pos = pos + dir * v.step_size / n;
            rpos = pos; rdir = dir;
        }
        mc.iterations++;
    }
}

// =============================================================================================
// Per-lane building blocks shared with the wave-cooperative samplers (device_volume_coop.hpp).
// =============================================================================================

// The 64-tap separable sum with per-lane, clamped addressing (stragglers of incoherent waves).  Deliberately
// NOT inlined: it is the rare path, and inlined at three call sites it would set the register budget (and so
// the occupancy) of the whole march kernel.  It recomputes the B-spline weights from the coordinate -- same
// operations, same values -- so a call site only has to pass three floats.
// One of the four B-spline weights of bspline_weights(f, ...), selected at run time (same operations, same
// values): lets the rolled loops below keep a single weight in registers instead of three arrays.
__device__ __forceinline__ float bspline_weight_at(float f, int idx) {
    float w0, w1, w2, w3;
    bspline_weights(f, w0, w1, w2, w3);
    return idx == 0 ? w0 : idx == 1 ? w1 : idx == 2 ? w2 : w3;
}

// Written for the SMALLEST register footprint, not for speed: every loop rolled, one texel in flight, weights
// recomputed where they are used.  Under the AMDGPU calling convention a kernel's values that live across a
// call sit above the callee's registers, so this function's count adds to the caller's: at 81 VGPRs (the first
// version) it alone pushed the march kernels past the 96 of five waves per SIMD.
__device__ __attribute__((noinline)) f4 cubic_gather_fn(const f4 *__restrict__ tex, int nx, int ny, int nz, float x,
                                                        float y, float z) {
    const float xg = x - 0.5f, yg = y - 0.5f, zg = z - 0.5f;
    const float fi = floorf(xg), fj = floorf(yg), fk = floorf(zg);
    const float fx = xg - fi, fy = yg - fj, fz = zg - fk;
    const int i = (int)fi, j = (int)fj, k = (int)fk;
    f4 acc = f4{0, 0, 0, 0};
#pragma unroll 1
    for (int c = 0; c < 4; c++) {
        const unsigned slab = (unsigned)clampi(k - 1 + c, 0, nz - 1) * (unsigned)ny;
        f4 s = f4{0, 0, 0, 0};
#pragma unroll 1
        for (int b = 0; b < 4; b++) {
            const unsigned row = (slab + (unsigned)clampi(j - 1 + b, 0, ny - 1)) * (unsigned)nx;
            const float wyb = bspline_weight_at(fy, b);
#pragma unroll 1
            for (int a = 0; a < 4; a++) {
                const f4 t = ldtexel(tex + (row + (unsigned)clampi(i - 1 + a, 0, nx - 1)));
                const float w = bspline_weight_at(fx, a) * wyb;        // wxy[b][a] of the slab order
                if (a == 0 && b == 0) s = f4{w * t.x, w * t.y, w * t.z, w * t.w};
                else s = f4{fmaf(w, t.x, s.x), fmaf(w, t.y, s.y), fmaf(w, t.z, s.z), fmaf(w, t.w, s.w)};
            }
        }
        const float wzc = bspline_weight_at(fz, c);
        if (c == 0) acc = f4{wzc * s.x, wzc * s.y, wzc * s.z, wzc * s.w};
        else acc = f4{fmaf(wzc, s.x, acc.x), fmaf(wzc, s.y, acc.y), fmaf(wzc, s.z, acc.z), fmaf(wzc, s.w, acc.w)};
    }
    return acc;
}

template <bool CLAMP>
__device__ __forceinline__ f4 linear_taps(const f4 *__restrict__ t, int nx, int ny, int nz, int i, int j, int k,
                                          float a, float b, float c) {
    const int i0 = CLAMP ? clampi(i, 0, nx - 1) : i, i1 = CLAMP ? clampi(i + 1, 0, nx - 1) : i + 1;
    const int j0 = CLAMP ? clampi(j, 0, ny - 1) : j, j1 = CLAMP ? clampi(j + 1, 0, ny - 1) : j + 1;
    const int k0 = CLAMP ? clampi(k, 0, nz - 1) : k, k1 = CLAMP ? clampi(k + 1, 0, nz - 1) : k + 1;
    const size_t W = nx, WH = (size_t)nx * ny;
    const f4 v000 = ldtexel(t + k0 * WH + j0 * W + i0), v100 = ldtexel(t + k0 * WH + j0 * W + i1);
    const f4 v010 = ldtexel(t + k0 * WH + j1 * W + i0), v110 = ldtexel(t + k0 * WH + j1 * W + i1);
    const f4 v001 = ldtexel(t + k1 * WH + j0 * W + i0), v101 = ldtexel(t + k1 * WH + j0 * W + i1);
    const f4 v011 = ldtexel(t + k1 * WH + j1 * W + i0), v111 = ldtexel(t + k1 * WH + j1 * W + i1);
    const f4 c00 = lerp4(v000, v100, a), c10 = lerp4(v010, v110, a);
    const f4 c01 = lerp4(v001, v101, a), c11 = lerp4(v011, v111, a);
    const f4 c0 = lerp4(c00, c10, b), c1 = lerp4(c01, c11, b);
    return lerp4(c0, c1, c);
}

// trace_rays_through_density_gradients (.h:1455-1544): entry test + integrator dispatch.
template <int ALGO, int INTERP>
__device__ __forceinline__ void trace_volume(f3 &pos_io, f3 &dir_io, const VolumeDev &v, MarchCount &mc) {
    const f3 mn = v.min_bound, mx = v.max_bound;
    const f3 scale = mk3(1.0f / (mx.x - mn.x), 1.0f / (mx.y - mn.y), 1.0f / (mx.z - mn.z));
    f3 pos = pos_io;
    const f3 dir = dir_io;
    if (pos.x <= mn.x || pos.y <= mn.y || pos.z <= mn.z || pos.x >= mx.x || pos.y >= mx.y || pos.z >= mx.z) {
        if (!intersect_with_volume(pos, dir, mn, mx)) return;       // miss: ray left unchanged
    }
    pos_io = pos;
    if (ALGO == 1) euler<INTERP>(pos_io, dir_io, v, scale, mc);
    else rk4<INTERP>(pos_io, dir_io, v, scale, mc);
}

}  // namespace photon
