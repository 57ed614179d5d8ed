// device_volume_extra.hpp - the two remaining values of ray_tracing_algorithm: 3 = rk45
// (trace_rays_through_density_gradients.h:304-718) and 4 = adams_bashforth (.h:1293-1453).
//
// Per-lane code with per-lane trilinear gathers of the RAW gradient volume (the reference has no
// cubic branch for them): these integrators are restated literally, oddities included -- .w (n-1)
// used as the refractive index, the inside-box test before the very first step that turns both into
// no-ops for rays entering through a max face (DESIGN.md section 5 has the list) -- so they
// are about completing the enum and its wire behaviour, not about speed.  Operation order is that of
// the C expressions in the reference (double literals fold to float factors before they meet a
// float3); powf(x, 0.25f) is two correctly rounded square roots on both sides of the parity check.
#pragma once
#include "device_volume.hpp"

namespace photon {

__device__ __forceinline__ f4 fetch_raw(const VolumeDev &v, f3 l, MarchCount &mc) {
    mc.samples++;
    return tex3d_linear(v, l.x, l.y, l.z);
}
__device__ __forceinline__ f3 scaled_grad(float s, const f4 &val) { return s * mk3(val.x, val.y, val.z); }

// h /= 10 when a stage point left the volume; keep going while h >= floor_frac * step (.h:397-419)
__device__ __forceinline__ bool rk45_shrink(float &h, double floor_frac, float step) {
    h = (float)(h / 10.0);
    return (double)h >= floor_frac * step;
}

__device__ __noinline__ void rk45(f3 &rpos, f3 &rdir, const VolumeDev &v, f3 scale, MarchCount &mc) {
    const float tol = 1e-3;
    float refractive_index = 1.000277;
    float h = v.step_size / refractive_index;
    f3 pos = rpos, dir = rdir;
    int loop_ctr = 0;
    while (true) {
        loop_ctr += 1;
        if (loop_ctr > 100000) break;
        const f3 T0 = refractive_index * dir;
        f3 R_n = pos, T_n = T0;
        const f3 k1 = h * T_n;
        f3 lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) { if (rk45_shrink(h, 0.1, v.step_size)) continue; else break; }
        f4 val = fetch_raw(v, lookup, mc);
        if (val.w < v.data_min) {
            pos = pos + v.step_size / v.data_min * dir;
            rpos = pos;
            continue;
        }
        const f3 l1 = scaled_grad(h * val.w, val);
        R_n = pos + k1 / (float)4.0;
        T_n = refractive_index * dir + l1 / (float)4.0;
        const f3 k2 = h * T_n;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) { if (rk45_shrink(h, 0.1, v.step_size)) continue; else break; }
        val = fetch_raw(v, lookup, mc);
        const f3 l2 = scaled_grad(h * val.w, val);
        R_n = pos + (float)(3.0 / 32.0) * k1 + (float)(9.0 / 32.0) * k2;
        T_n = refractive_index * dir + (float)(3.0 / 32.0) * l1 + (float)(9.0 / 32.0) * l2;
        const f3 k3 = h * T_n;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) { if (rk45_shrink(h, 0.1, v.step_size)) continue; else break; }
        val = fetch_raw(v, lookup, mc);
        const f3 l3 = scaled_grad(h * val.w, val);
        R_n = pos + (float)(1932.0 / 2197.0) * k1 - (float)(7200.0 / 2197.0) * k2 + (float)(7296.0 / 2197.0) * k3;
        T_n = refractive_index * dir + (float)(1932.0 / 2197.0) * l1 - (float)(7200.0 / 2197.0) * l2 +
              (float)(7296.0 / 2197.0) * l3;
        const f3 k4 = h * T_n;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) { if (rk45_shrink(h, 0.1, v.step_size)) continue; else break; }
        val = fetch_raw(v, lookup, mc);
        const f3 l4 = scaled_grad(h * val.w, val);
        R_n = pos + (float)(439.0 / 216.0) * k1 - (float)8.0 * k2 + (float)(3680.0 / 513.0) * k3 -
              (float)(845.0 / 4104.0) * k4;
        T_n = refractive_index * dir + (float)(439.0 / 216.0) * l1 - (float)8.0 * l2 + (float)(3680.0 / 513.0) * l3 -
              (float)(845.0 / 4104.0) * l4;
        const f3 k5 = h * T_n;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) { if (rk45_shrink(h, 0.1, v.step_size)) continue; else break; }
        val = fetch_raw(v, lookup, mc);
        const f3 l5 = scaled_grad(h * val.w, val);
        R_n = pos - (float)(8.0 / 27.0) * k1 + (float)2.0 * k2 - (float)(3544.0 / 2565.0) * k3 +
              (float)(1859.0 / 4104.0) * k4 - (float)(11.0 / 40.0) * k5;
        T_n = refractive_index * dir - (float)(8.0 / 27.0) * l1 + (float)2.0 * l2 - (float)(3544.0 / 2565.0) * l3 +
              (float)(1859.0 / 4104.0) * l4 - (float)(11.0 / 40.0) * l5;
        const f3 k6 = h * T_n;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) { if (rk45_shrink(h, 0.01, v.step_size)) continue; else break; }   // .h:606
        val = fetch_raw(v, lookup, mc);
        const f3 l6 = scaled_grad(h * val.w, val);
        const f3 y4 = pos + (float)(25.0 / 216.0) * k1 + (float)(1408.0 / 2565.0) * k3 + (float)(2197.0 / 4104.0) * k4 -
                      (float)(1.0 / 5.0) * k5;
        const f3 y5 = pos + (float)(16.0 / 135.0) * k1 + (float)(6656.0 / 12825.0) * k3 +
                      (float)(28561.0 / 56430.0) * k4 - (float)(9.0 / 50.0) * k5 + (float)(2.0 / 55.0) * k6;
        const f3 z4 = refractive_index * dir + (float)(25.0 / 216.0) * l1 + (float)(1408.0 / 2565.0) * l3 +
                      (float)(2197.0 / 4104.0) * l4 - (float)(1.0 / 5.0) * l5;
        const f3 z5 = refractive_index * dir + (float)(16.0 / 135.0) * l1 + (float)(6656.0 / 12825.0) * l3 +
                      (float)(28561.0 / 56430.0) * l4 - (float)(9.0 / 50.0) * l5 + (float)(2.0 / 55.0) * l6;
        const f3 dy = y4 - y5, dz = z4 - z5;
        const float ih = 1 / h;
        const f3 R0 = ih * mk3(fabsf(dy.x), fabsf(dy.y), fabsf(dy.z));
        const f3 R1 = ih * mk3(fabsf(dz.x), fabsf(dz.y), fabsf(dz.z));
        const float a = R0.x > R1.x ? R0.x : R1.x;
        const float b = R0.y > R1.y ? R0.y : R1.y;
        const float c = R0.z > R1.z ? R0.z : R1.z;
        const float R_max = (a > b ? a : b) > c ? (a > b ? a : b) : c;
        float s = (float)(0.84 * (double)sqrtf(sqrtf(tol / R_max)));
        if (R_max <= tol) {
            pos = y4;
            dir = (1 / refractive_index) * z4;
            dir = normalize(dir);
            rpos = pos;
            rdir = dir;
            mc.iterations++;
            lookup = lookup_index(pos, v, scale);
            if (!inside_box(pos, v, lookup)) return;
            val = fetch_raw(v, lookup, mc);
            refractive_index = val.w;
            if ((double)s > 5.00) s = (float)5.00;
            h *= s;
        } else {
            if ((double)s < 0.1) s = (float)0.1;
            h *= s;
        }
    }
}

__device__ __noinline__ void adams_bashforth(f3 &rpos, f3 &rdir, const VolumeDev &v, f3 scale, MarchCount &mc) {
    f3 pos = rpos, dir = rdir, lookup;
    f4 val = f4{0, 0, 0, 0};
    int loop_ctr = 0, spins = 0;
    f3 R_n = mk3(0, 0, 0), T_n = mk3(0, 0, 0), D = mk3(0, 0, 0);
    float delta_t;
    // history of the start-up steps; zero where the reference leaves it uninitialised
    f3 T0 = mk3(0, 0, 0), T1 = mk3(0, 0, 0), T2 = mk3(0, 0, 0), D0 = mk3(0, 0, 0), D1 = mk3(0, 0, 0), D2 = mk3(0, 0, 0);
    while (loop_ctr < 3) {                                              // RK4 start-up, .h:1316-1398
        pos = rpos;
        dir = rdir;
        lookup = lookup_index(pos, v, scale);
        if (!inside_box(pos, v, lookup)) break;
        if (!can_access(v, lookup)) {
            pos = pos + v.step_size / v.data_min * dir;
            rpos = pos;
            if (++spins > kSpinMax) break;
            continue;
        }
        val = fetch_raw(v, lookup, mc);
        if (val.w < v.data_min) {
            pos = pos + v.step_size / v.data_min * dir;
            rpos = pos;
            if (++spins > kSpinMax) break;
            continue;
        }
        R_n = pos;
        delta_t = v.step_size / val.w;
        T_n = val.w * rdir;
        D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
        const f3 A = delta_t * D;
        pos = R_n + (float)(delta_t / 2.0) * T_n + (float)(1 / 8.0 * delta_t) * A;
        lookup = lookup_index(pos, v, scale);
        if (!inside_box(pos, v, lookup)) break;
        val = fetch_raw(v, lookup, mc);
        D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
        const f3 B = delta_t * D;
        pos = R_n + delta_t * T_n + (float)(1 / 2.0 * delta_t) * B;
        lookup = lookup_index(pos, v, scale);
        if (!inside_box(pos, v, lookup)) break;
        val = fetch_raw(v, lookup, mc);
        D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
        const f3 C = delta_t * D;
        R_n = R_n + delta_t * (T_n + (float)(1 / 6.0) * (A + 2.0f * B));
        T_n = T_n + (float)(1 / 6.0) * (A + 4.0f * B + C);
        rpos = R_n;
        rdir = normalize(T_n / val.w);
        if (loop_ctr == 0) { T0 = T_n; D0 = D; }
        else if (loop_ctr == 1) { T1 = T_n; D1 = D; }
        else { T2 = T_n; D2 = D; }
        loop_ctr += 1;
        mc.iterations++;
    }
    loop_ctr = 0;                                                       // predictor, .h:1402-1450
    const float refractive_index = val.w;
    pos = rpos;
    dir = rdir;
    while (true) {
        loop_ctr += 1;
        if (loop_ctr > kLoopMax) break;
        R_n = rpos;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) break;
        val = fetch_raw(v, lookup, mc);
        if (val.w < v.data_min) {
            pos = pos + v.step_size / refractive_index * dir;
            rpos = pos;
            continue;
        }
        delta_t = v.step_size / val.w;
        D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
        const f3 R_n_1 = R_n + (delta_t / 24) * (55.0f * T_n - 59.0f * T2 + 37.0f * T1 - 9.0f * T0);
        const f3 T_n_1 = T_n + (delta_t / 24) * (55.0f * D - 59.0f * D2 + 37.0f * D1 - 9.0f * D0);
        T0 = T1; D0 = D1;
        T1 = T2; D1 = D2;
        T2 = T_n; D2 = D;
        R_n = R_n_1;
        T_n = T_n_1;
        rpos = R_n;
        rdir = normalize(T_n / val.w);
        mc.iterations++;
    }
}

// Entry test + dispatch for algorithms 3, 4 and the no-op default (the prologue of trace_volume, .h:1455-1544).
template <int ALGO>
__device__ __forceinline__ void trace_volume_extra(f3 &pos_io, f3 &dir_io, const VolumeDev &v, MarchCount &mc) {
    const f3 mn = v.min_bound, mx = v.max_bound;
    const f3 scale = mk3(1.0f / (mx.x - mn.x), 1.0f / (mx.y - mn.y), 1.0f / (mx.z - mn.z));
    f3 pos = pos_io;
    const f3 dir = dir_io;
    if (pos.x <= mn.x || pos.y <= mn.y || pos.z <= mn.z || pos.x >= mx.x || pos.y >= mx.y || pos.z >= mx.z) {
        if (!intersect_with_volume(pos, dir, mn, mx)) return;
    }
    pos_io = pos;
    if (ALGO == 3) rk45(pos_io, dir_io, v, scale, mc);
    else if (ALGO == 4) adams_bashforth(pos_io, dir_io, v, scale, mc);
    // anything else: `default: break` (.h:1537) -- the ray only moved to its entry point
}

}  // namespace photon
