// device_optics.hpp - per-ray optics on the device: light-field ray generation with Mie lookup,
// lens / aperture propagation, sensor intersection and the two splat models.
//
// Replaces these device functions of cuda_codes/parallel_ray_tracing.cu:
//   generate_lightfield_angular_data   :71-237     ray_sphere_intersection        :239-343
//   measure_distance_to_optical_axis   :345-380    propagate_rays_through_single_element :383-1011
//   propagate_rays_through_optical_system :1274-1381
//   intersect_sensor_02 :1383-1543   create_apparent_image :1545-1733   intersect_sensor :1735-1895
// Float/double placement follows the reference expression by expression; elementary functions
// come from include/photon_det_math.h (bit-reproducible, see that header).
#pragma once
#include "../../include/parallel_ray_tracing.h"
#include "../../include/photon_det_math.h"
#include "../../include/photon_philox.h"
#include "device_vec.hpp"

namespace photon {

constexpr int kMaxElements = 5;         // MAX_CURRENT_ELEMENTS (.cu:38)

// the four noise arguments of start_ray_tracing + the seed (reference: cuRAND states seeded with
// time(NULL), .cu:3405-3445; here Philox keyed by PHOTON_NOISE_SEED, include/photon_philox.h)
struct NoiseDev {
    int add_pos, add_ngrad;
    float pos_std, ngrad_std;
    unsigned long long seed;
};

// Everything start_ray_tracing uploads before its launch loop (.cu:3132-3314), by value.
struct SceneDev {
    float lens_pitch, image_distance, beam_wavelength, f_number, ratio;
    int scattering_type;                // 1 = Mie table, 0 = diffuse
    int rays_per_source;
    const float *sx, *sy, *sz;          // light-field sources, SoA
    const double *sradiance;
    const int *sdia;
    int num_sources;
    float z_offset, object_distance;
    float mie_inv_rot[9];
    float beam[3];
    const float *mie_angle;             // [num_angles]
    const float *mie_irr;               // [num_angles][num_diameters]
    int num_angles, num_diameters;
    const float *lens_x, *lens_y;       // where lens sample k meets the lens: the table every source shares (.cu:2006), its double-precision
                                        // product (.cu:123-124) evaluated once per scene on the host instead of once per ray (photon_scene.hip)
    int num_elements;
    element_data_t elems[kMaxElements];
    float centers[kMaxElements][3];
    float planes[kMaxElements][4];
    int sys_index[kMaxElements];
    // the working element train (train_mode 1): every element, in device memory (lenslet arrays have
    // far more than kMaxElements members)
    int train_mode;
    const element_data_t *all_elems;
    const float *all_centers;           // [num_elements][3]
    const float *all_planes;            // [num_elements][4]
    const int *all_sys_index;
    camera_design_t cam;
    NoiseDev noise;
    // launch-slot -> (source, lens sample) mapping.  0 = source-major, the reference's order (.cu:1988-2006):
    // slot r is ray r % rays_per_source of source src_begin + r / rays_per_source -- a wave holds the narrow cone
    // of one BOS source.  1 = lens-major: slot r is lens sample r / S of the (r % S)-th source of this launch,
    // sources taken in the spatial order src_perm -- a wave holds 64 neighbouring particles aimed at the SAME
    // point of the lens (every source uses the same lens-sample table, .cu:2006), which is what stays
    // coherent when the cone is as wide as the aperture (PIV through a volume).
    int ray_order;
    // slots per source of THIS launch and what lens sample each stands for: rays_per_source and the identity (nullptr), or -- the
    // volume-free path -- only the lens samples that can reach the first element's aperture from any source (photon_scene.hip)
    int slot_rays;
    const int *slot_map;
    const int *src_perm;                // spatial order of THIS launch's sources (lens-major only); nullptr = identity
    const int *src_list;                // source-major, volume-free path: the launch's sources that can reach the sensor (already offset
                                        // to the launch's first one); nullptr = every source of [src_begin, src_end)
    // index of this scene's source 0 in the caller's source list (PHOTON_DEVICES uploads each device only its
    // shard): keeps the noise generator's per-ray key independent of the sharding
    long long source_base;
    // > 0: rays whose UNDEFLECTED path meets element 0's front surface more than pitch/2 + doom_margin from the
    // axis are not marched (they are killed there whatever the volume does, .cu:560-566 / :447); 0 = off
    float doom_margin;
};

// slot of this launch -> source index and lens-sample index; n_src = sources in this launch
__device__ __forceinline__ void slot_to_ray(const SceneDev &sc, long long src_begin, unsigned n_rays, unsigned r, int &source,
                                            int &local_ray) {
    const unsigned rps = (unsigned)sc.slot_rays;
    if (sc.ray_order == 0) {
        source = sc.src_list ? sc.src_list[r / rps] : (int)(src_begin + r / rps);
        local_ray = (int)(r % rps);
    } else {
        const unsigned n_src = n_rays / rps;
        local_ray = (int)(r / n_src);
        const unsigned k = r % n_src;                                   // src_perm covers exactly this launch's range
        source = sc.src_perm ? sc.src_perm[k] : (int)(src_begin + k);
    }
    if (sc.slot_map) local_ray = sc.slot_map[local_ray];
}

struct Ray {                            // light_ray_data_t
    f3 pos, dir;
    float wavelength;
    double radiance;
};

__device__ __forceinline__ void kill(Ray &r) {
    r.pos = nan3();
    r.dir = nan3();
    r.wavelength = nanf32();
    r.radiance = (double)nanf32();
}

// generate_lightfield_angular_data (.cu:71-237)
__device__ __forceinline__ Ray generate_ray(const SceneDev &sc, int source, int local_ray) {
    const float x_current = sc.sx[source], y_current = sc.sy[source], z_current = sc.sz[source];
    const double src_radiance = sc.sradiance[source];
    float x_lens, y_lens;
    if (sc.rays_per_source == 1) {
        x_lens = 0.0f; y_lens = 0.0f;
    } else {                            // .cu:123-124, a function of the lens sample only: tabulated per scene
        x_lens = sc.lens_x[local_ray];
        y_lens = sc.lens_y[local_ray];
    }
    const float theta = photon_det_atanf(-(x_lens - x_current) / (sc.image_distance - z_current));
    const float phi = photon_det_atanf(-(y_lens - y_current) / (sc.image_distance - z_current));
    const f3 d0 = normalize(mk3(photon_det_tanf(theta), photon_det_tanf(phi), -1.0f));
    double irradiance_current;
    if (sc.scattering_type) {
        const float del = sc.mie_angle[1] - sc.mie_angle[0];
        const f3 beam = mk3(sc.beam[0], sc.beam[1], sc.beam[2]);
        f3 d = normalize(matvec(sc.mie_inv_rot, d0));
        const float dp = beam.x * d.x + beam.y * d.y + beam.z * d.z;
        const float deg = (float)(photon_det_acosf(dp) * 180.0 / M_PI);     // angleBetween -> degrees
        const float ray_angle = (float)(deg * M_PI / 180.0);                // .cu:184
        const float angle = (ray_angle - sc.mie_angle[0]) / del;
        const int angle_l = isnan(angle) ? 0 : (int)angle;
        const int angle_u = angle_l + 1;
        // rows clamped into the table: the reference reads one row past the end when the angle
        // lands on the last row (.cu:194-198)
        const int rl = clampi(angle_l, 0, sc.num_angles - 1), ru = clampi(angle_u, 0, sc.num_angles - 1);
        const int col = clampi(sc.sdia[source], 0, sc.num_diameters - 1);
        const float il = sc.mie_irr[rl * sc.num_diameters + col];
        const float iu = sc.mie_irr[ru * sc.num_diameters + col];
        const float irr = il + (angle - angle_l) / (angle_u - angle_l) * (iu - il);
        irradiance_current = irr * src_radiance;
    } else {
        irradiance_current = src_radiance;
    }
    Ray r;
    r.pos = mk3(x_current, y_current, z_current);
    r.dir = d0;
    r.wavelength = sc.beam_wavelength;
    r.radiance = 1 / (sc.f_number * sc.f_number) * irradiance_current;      // .cu:233
    return r;
}

// ray_sphere_intersection (.cu:239-343).  The f32 cancellation in gamma is reproduced, not fixed.
__device__ __forceinline__ f3 ray_sphere_intersection(f3 pos_c, float R, f3 dir_i, f3 pos_i) {
    const float alpha = dot(dir_i, dir_i);
    const float beta = 2 * dot(dir_i, (pos_i - pos_c));
    const float gamma = dot(pos_i - pos_c, pos_i - pos_c) - R * R;
    const float sq = (float)(beta * beta - 4.0 * alpha * gamma);
    if (sq < 0.0) return nan3();
    const float t1 = (float)((-beta + sqrtf(sq)) / (2.0 * alpha));
    const float t2 = (float)((-beta - sqrtf(sq)) / (2.0 * alpha));
    // front and back surface make the same choice (.cu:298-336): R>0 -> smaller root
    const float t = (R > 0) ? (t1 <= t2 ? t1 : t2) : (t1 >= t2 ? t1 : t2);
    return pos_i + dir_i * t;
}

// measure_distance_to_optical_axis (.cu:345-380)
__device__ __forceinline__ float axis_distance(f3 pos_i, f3 pos_0, const float *plane) {
    const float a = plane[0], b = plane[1], c = plane[2];
    const float tmin = dot(mk3(a, b, c), pos_i - pos_0) / (a * a + b * b + c * c);
    const f3 p = pos_0 + mk3(a, b, c) * tmin;
    return sqrtf(dot(pos_i - p, pos_i - p));
}

// propagate_rays_through_single_element (.cu:383-1011)
__device__ __forceinline__ Ray single_element(const element_data_t &e, f3 center, const float *plane, Ray ray) {
    const char type = e.element_type;
    f3 dir = ray.dir, src = ray.pos;
    const float wavelength = ray.wavelength;
    double radiance = ray.radiance;
    const float a = plane[0], b = plane[1], c = plane[2], d = plane[3];
    const float pitch = e.element_geometry.pitch;
    const double vertex_distance = e.element_geometry.vertex_distance;
    if (type == 't') {                                                  // thin lens :416-503
        const float focal = e.element_properties.thin_lens_focal_length;
        const float t = -(dot(mk3(a, b, c), src) + d) / dot(mk3(a, b, c), dir);
        const f3 hit = src + dir * t;
        const float dist = axis_distance(hit, center, plane);
        if (dist > pitch / 2.0) { kill(ray); return ray; }
        src = hit;
        dir = -(src - center) / focal + dir;
        dir = normalize(dir);
    } else if (type == 'l') {                                           // thick lens :507-864
        const float Rf = e.element_geometry.front_surface_radius;
        const float Rb = e.element_geometry.back_surface_radius;
        const double n_lens = e.element_properties.refractive_index;
        const float abbe = e.element_properties.abbe_number;
        const float transmission = e.element_properties.transmission_ratio;
        const float absorbance = e.element_properties.absorbance_rate;
        const float nmag = sqrtf(a * a + b * b + c * c);
        float ds = (float)(+vertex_distance / 2.0 - Rf);
        const f3 c_front = center + mk3(a, b, c) * ds / nmag;
        f3 hit = ray_sphere_intersection(c_front, Rf, dir, src);
        float dist = axis_distance(hit, center, plane);
        if (dist > pitch / 2.0) { kill(ray); return ray; }
        f3 normal = normalize(hit - c_front);
        float eta;
        const float lambda_D = 589.3, lambda_F = 486.1, lambda_C = 656.3;
        if (!isnan(abbe)) {                                             // Cauchy dispersion :622-636
            eta = (float)(1.0 / (n_lens + (1. / (wavelength * wavelength) - 1 / (lambda_D * lambda_D)) *
                                              ((n_lens - 1) / (abbe * (1 / (lambda_F * lambda_F) -
                                                                       1 / (lambda_C * lambda_C))))));
        } else {
            eta = (float)(1.0 / n_lens);
        }
        float cosi = -dot(dir, normal);
        float radicand = (float)(1.0 - (eta * eta) * (1.0 - cosi * cosi));
        if (radicand < 0.0) { kill(ray); return ray; }                  // total internal reflection
        dir = dir * eta + (eta * cosi - sqrtf(radicand)) * normal;
        dir = normalize(dir);
        src = hit;
        ds = (float)(-vertex_distance / 2 - Rb);
        const f3 c_back = center + mk3(a, b, c) * ds / nmag;
        hit = ray_sphere_intersection(c_back, Rb, dir, src);
        dist = axis_distance(hit, center, plane);
        if (dist > pitch / 2.0) { kill(ray); return ray; }
        normal = normalize(-(hit - c_back));
        if (!isnan(abbe)) {
            eta = (float)(n_lens + (1.0 / (wavelength * wavelength) - 1.0 / (lambda_D * lambda_D)) *
                                       ((n_lens - 1) / (abbe * (1 / (lambda_F * lambda_F) -
                                                                1 / (lambda_C * lambda_C)))));
        } else {
            eta = (float)n_lens;
        }
        cosi = -dot(dir, normal);
        radicand = (float)(1.0 - (eta * eta) * (1.0 - cosi * cosi));
        if (radicand < 0.0) { kill(ray); return ray; }
        dir = eta * dir + (eta * cosi - sqrtf(radicand)) * normal;
        dir = normalize(dir);
        if (absorbance != 0) {
            const float dist_in = sqrtf(dot(hit - src, hit - src));
            radiance = (1.0 - absorbance) * radiance * dist_in;
        } else {
            radiance = transmission * radiance;
        }
        src = hit;
    } else {                                                            // aperture stop :868-992
        const float nmag = sqrtf(a * a + b * b + c * c);
        float ds = (float)(-vertex_distance / 2.0);
        float d_temp = d - ds * nmag;
        float t = -(dot(mk3(a, b, c), src) + d_temp) / dot(mk3(a, b, c), dir);
        f3 hit = src + dir * t;
        float dist = axis_distance(hit, center, plane);
        if (dist > pitch / 2.0) { kill(ray); return ray; }
        ds = (float)(+vertex_distance / 2);
        d_temp = d - ds * nmag;
        t = -(dot(mk3(a, b, c), src) + d_temp) / dot(mk3(a, b, c), dir);
        hit = src + dir * t;
        dist = axis_distance(hit, center, plane);
        if (dist > pitch / 2.0) { kill(ray); return ray; }
        src = hit;
    }
    ray.dir = dir; ray.pos = src; ray.wavelength = wavelength; ray.radiance = radiance;
    return ray;
}

// Distance from the optical axis at which a ray meets the front of element `e` -- the quantity the reference
// tests against pitch/2 before anything else happens to the ray (thin lens .cu:440-447, thick lens .cu:540-566).
// NaN when there is no such test for this element type (or no intersection): callers must treat NaN as
// "cannot tell".
__device__ __forceinline__ float front_axis_distance(const element_data_t &e, f3 center, const float *plane, const Ray &ray) {
    const float a = plane[0], b = plane[1], c = plane[2], d = plane[3];
    if (e.element_type == 't') {
        const float t = -(dot(mk3(a, b, c), ray.pos) + d) / dot(mk3(a, b, c), ray.dir);
        return axis_distance(ray.pos + ray.dir * t, center, plane);
    }
    if (e.element_type == 'l') {
        const float Rf = e.element_geometry.front_surface_radius;
        const double vertex_distance = e.element_geometry.vertex_distance;
        const float nmag = sqrtf(a * a + b * b + c * c);
        const float ds = (float)(+vertex_distance / 2.0 - Rf);
        const f3 c_front = center + mk3(a, b, c) * ds / nmag;
        return axis_distance(ray_sphere_intersection(c_front, Rf, ray.dir, ray.pos), center, plane);
    }
    return nanf("");
}

// Stage 1a for one launch slot (parallel_ray_tracing.cu:2004-2082): generate the ray and move it into the volume's world frame;
// a ray aimed so far outside the first element's aperture that no deflection the volume can produce brings it back
// (SceneDev::doom_margin, launch_chunk) is marked dead (NaN position): the march skips it, the sensor stage drops it as it
// would after the lens.  ONE body for raygen_kernel and for the march kernels' own prologue (march_kernel.hpp): same bits.
struct RayPD { f3 p, d; };
__device__ __forceinline__ RayPD generate_state(const SceneDev &sc, long long src_begin, unsigned n_rays, unsigned r, double &radiance) {
    int source, local_ray;
    slot_to_ray(sc, src_begin, n_rays, r, source, local_ray);
    const Ray ray = generate_ray(sc, source, local_ray);
    f3 p = ray.pos, d = ray.dir;
    p.z = (float)(p.z - (sc.z_offset + 750e3));                         // .cu:2045
    p = matvec(sc.cam.inverse_rotation_matrix, p);                      // camera -> world
    d = matvec(sc.cam.inverse_rotation_matrix, d);
    if (sc.doom_margin > 0.f) {
        const float dist = front_axis_distance(sc.elems[0], mk3(sc.centers[0][0], sc.centers[0][1], sc.centers[0][2]),
                                               sc.planes[0], ray);
        if (dist > sc.elems[0].element_geometry.pitch / 2.0 + sc.doom_margin) p = nan3();
    }
    radiance = ray.radiance;
    return RayPD{p, d};
}

// The WORKING element train (train_mode 1; the reference advertises it, its device code is a stub; the
// sequential branch of its numpy ancestor, perform_ray_tracing_03.py:1419-1485, runs and pins this one
// through tests/golden/train_f64.npz, the simultaneous-elements branch :1254-1417 does not): groups in
// decreasing system index; a single-member group goes through ITS element; a group of simultaneous elements (lenslet
// array) is split by element plane, planes visited in the order the entering ray meets them, and on
// each plane the ray goes through the member whose centre is nearest to its intersection point.
constexpr int kMaxGroupPlanes = 8;
__device__ __forceinline__ bool same_plane(const float *p, const float *q) {
    return p[0] == q[0] && p[1] == q[1] && p[2] == q[2] && p[3] == q[3];
}
__device__ __forceinline__ float plane_time(const float *pl, const Ray &ray) {
    return -(pl[0] * ray.pos.x + pl[1] * ray.pos.y + pl[2] * ray.pos.z + pl[3]) /
           (pl[0] * ray.dir.x + pl[1] * ray.dir.y + pl[2] * ray.dir.z);
}
__device__ __noinline__ Ray element_train(const SceneDev &sc, Ray ray) {
    const int n = sc.num_elements;
    int seq = 0;
    for (int k = 0; k < n; k++)
        if (seq <= sc.all_sys_index[k]) seq = sc.all_sys_index[k];
    for (int idx = 0; idx < seq; idx++) {
        int count = 0, only = 0;
        for (int k = 0; k < n; k++)
            if (seq - sc.all_sys_index[k] == idx) { only = k; count++; }
        if (count == 0) continue;
        if (count == 1) {
            const float *c = sc.all_centers + 3 * only;
            ray = single_element(sc.all_elems[only], mk3(c[0], c[1], c[2]), sc.all_planes + 4 * only, ray);
            continue;
        }
        int uplane[kMaxGroupPlanes], nu = 0;
        float ut[kMaxGroupPlanes];
        for (int k = 0; k < n; k++) {
            if (seq - sc.all_sys_index[k] != idx) continue;
            const float *pl = sc.all_planes + 4 * k;
            bool seen = false;
            for (int u = 0; u < nu; u++) seen = seen || same_plane(sc.all_planes + 4 * uplane[u], pl);
            if (seen || nu == kMaxGroupPlanes) continue;
            uplane[nu] = k;
            ut[nu] = plane_time(pl, ray);
            nu++;
        }
        for (int a = 1; a < nu; a++) {                                  // stable insertion sort by time
            const int pk = uplane[a];
            const float tk = ut[a];
            int b = a - 1;
            while (b >= 0 && ut[b] > tk) { uplane[b + 1] = uplane[b]; ut[b + 1] = ut[b]; b--; }
            uplane[b + 1] = pk; ut[b + 1] = tk;
        }
        for (int u = 0; u < nu; u++) {
            const float *pl = sc.all_planes + 4 * uplane[u];
            const float t = plane_time(pl, ray);
            const f3 hit = ray.pos + t * ray.dir;
            int best = -1;
            float best_d2 = 0;
            for (int k = 0; k < n; k++) {
                if (seq - sc.all_sys_index[k] != idx || !same_plane(sc.all_planes + 4 * k, pl)) continue;
                const float *c = sc.all_centers + 3 * k;
                const f3 dc = hit - mk3(c[0], c[1], c[2]);
                const float d2 = dot(dc, dc);
                if (best < 0 || d2 < best_d2) { best = k; best_d2 = d2; }
            }
            const float *c = sc.all_centers + 3 * best;
            ray = single_element(sc.all_elems[best], mk3(c[0], c[1], c[2]), sc.all_planes + 4 * best, ray);
        }
    }
    return ray;
}

// propagate_rays_through_optical_system (.cu:1274-1381): sequential groups; a group with exactly
// one member is propagated through element 0 (.cu:1331-1333); larger groups reach the
// reference's empty multi-element stub and leave the ray unchanged.
template <bool TRAIN>
__device__ __forceinline__ Ray optical_system(const SceneDev &sc, Ray ray) {
    if (TRAIN) return element_train(sc, ray);           // separate kernel instantiation: keeps the call (and its
                                                        // scratch frame) out of the default sensor kernel
    int seq = 0;
    const int n = sc.num_elements < kMaxElements ? sc.num_elements : kMaxElements;
    for (int k = 0; k < n; k++)
        if (seq <= sc.sys_index[k]) seq = sc.sys_index[k];
    for (int idx = 0; idx < seq; idx++) {
        int count = 0;
        for (int k = 0; k < n; k++)
            if (seq - sc.sys_index[k] == idx) count++;
        if (count == 1)
            ray = single_element(sc.elems[0], mk3(sc.centers[0][0], sc.centers[0][1], sc.centers[0][2]),
                                 sc.planes[0], ray);
    }
    return ray;
}

// ---------------------------------------------------------------------------------------------
// Gaussian-spot (erf) splat: I0*pi/32 * d_erf(col) * d_erf(row) over the pixels within
// render_fraction*D of the centroid (.cu:1477-1540 / :1660-1730).
//
// Each rendered pixel receives the reference's f32 increment; the running sum is kept in f64: the
// reference's atomicAdd(float) in arbitrary order loses up to N*2^-25 relative on a pixel that
// receives N near-identical increments (BOS dots: N ~ 1e4), far above the 1e-5 parity bar.
//
// The per-lane form below (erf_splat_lane) is the fallback; the wave-cooperative form follows it.
// ---------------------------------------------------------------------------------------------
struct SplatReq {
    bool valid;                 // this lane has a ray to splat
    float X, Y, D, rfD;         // centroid (pixel units), spot diameter, render_fraction * D
    double scale;               // I0 * pi / 32
    int c0, c1, r0, r1;         // the reference's render window
};

__device__ __forceinline__ SplatReq erf_splat_prepare(bool valid, float d_x, float d_y, double radiance, f3 dir, float D,
                                                      float render_fraction) {
    SplatReq q;
    q.valid = valid;
    const double pi = 3.141592653589793;
    const float alpha = photon_det_atanf(sqrtf((dir.x / dir.z) * (dir.x / dir.z) + (dir.y / dir.z) * (dir.y / dir.z)));
    const float ca = photon_det_cosf(alpha);
    const double cos4 = ca * ca * ca * ca;
    q.X = d_x - 0.5; q.Y = d_y - 0.5;
    const float I0 = (float)(radiance * cos4 * 8.0 / pi);
    q.D = D;
    q.rfD = render_fraction * D;
    q.c0 = (int)floorf(q.X - render_fraction * D); q.c1 = (int)ceilf(q.X + render_fraction * D);
    q.r0 = (int)floorf(q.Y - render_fraction * D); q.r1 = (int)ceilf(q.Y + render_fraction * D);
    q.scale = I0 * pi / 32.0;
    return q;
}

__device__ __forceinline__ double erf_edge_diff(float sqrt8, int idx, float centre, float D) {
    return photon_det_erf(sqrt8 * (idx - centre - 0.5) / D) - photon_det_erf(sqrt8 * (idx - centre + 0.5) / D);
}

// per-lane form (fallback): one atomic per rendered pixel
__device__ __forceinline__ int erf_splat_lane(double *image, int W, int H, const SplatReq &q) {
    const float sqrt8 = sqrtf(8.0f);
    int taps = 0;
    for (int col = q.c0; col <= q.c1; col++) {
        if (col < 0 || col > W - 1) continue;
        const double sx = q.scale * erf_edge_diff(sqrt8, col, q.X, q.D);
        for (int row = q.r0; row <= q.r1; row++) {
            const float rad = sqrtf((col - q.X) * (col - q.X) + (row - q.Y) * (row - q.Y));
            if (!(row >= 0 && row <= H - 1 && rad <= q.rfD)) continue;
            const float inc = (float)(sx * erf_edge_diff(sqrt8, row, q.Y, q.D));
            atomicAdd(&image[(size_t)row * W + col], (double)inc);
            taps++;
        }
    }
    return taps;
}

// Wave-wide integer min / max on the DPP network (row shifts, then the two row broadcasts; the total lands in lane 63
// and is handed to every lane as a scalar): six dependent VALU steps, where the shuffle-based butterfly paid six
// ds_bpermute round trips through the LDS crossbar.
template <bool MAX>
__device__ __forceinline__ int wave_minmax_i(int v) {
    auto op = [](int a, int b) { return MAX ? max(a, b) : min(a, b); };
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));       // row_shr:1
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));       // row_shr:2
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));       // row_shr:4
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));       // row_shr:8 -> lane 15 of each row holds the row's result
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));       // row_bcast:15 into rows 1 and 3
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));       // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i(int v) { return wave_minmax_i<false>(v); }
__device__ __forceinline__ int wave_max_i(int v) { return wave_minmax_i<true>(v); }

// ---------------------------------------------------------------------------------------------
// Wave-cooperative splats.  The 64 rays of a wave land on a handful of neighbouring pixels (BOS: the narrow cone of
// one source; PIV without a volume: the rays of one particle converge on its image; lens-major PIV: 64 neighbouring
// particles), so issued per lane their atomics all collide.  Instead the wave TRANSPOSES the work through its own LDS
// area: every lane (ray) parks what it contributes -- for the erf splat its separable per-column / per-row factors,
// 16 erf evaluations, for the 4-pixel splat its four increments -- then every lane becomes a PIXEL of an 8x8 tile of
// the wave's window, walks the parked rays (broadcast LDS reads), sums what they add to its pixel in f64, and the wave
// issues ONE atomic instruction with 64 distinct addresses per tile.  Values are the reference's, term for term; only
// the (f64) summation order differs.  Waves whose window needs more than kSplatTiles tiles, or whose spots are too
// large for the parked layout, fall back to per-lane atomics.
// ---------------------------------------------------------------------------------------------
constexpr int kSplatTiles = 6;
constexpr int kSplatSlots = 7;              // columns / rows of one ray's window the parked layout carries (D = 3: 6 or 7)
constexpr unsigned kSplatStride = 2 * kSplatSlots * sizeof(double);      // bytes between the parked factors of consecutive rays

struct SplatLds {                           // one wave's area: 8 KiB (five 256-thread blocks per CU)
    float4 head[64];                        // 4-pixel splat: bits(ii), bits(jj), inc0, inc1 (the erf splat reads its rays' windows from registers)
    double f[64][2 * kSplatSlots];          // erf: scale * d_erf(column c0 + j), j < 7 | d_erf(row r0 + j)   4-pixel: [0] = inc2, inc3
    __device__ __forceinline__ float2 &tail(int r) { return *reinterpret_cast<float2 *>(&f[r][0]); }
};
struct TapLds {                             // the 4-pixel splat on its own (a camera without diffraction: sensor_kernel's SPLAT = 2): 1.5 KiB per wave
    float4 head[64];
    float2 more[64];                        // inc2, inc3
    __device__ __forceinline__ float2 &tail(int r) { return more[r]; }
};

// largest f32 t with sqrtf(t) <= r: the radius test sqrtf(s) <= r is then s <= t, exactly (sqrtf is correctly rounded
// and monotone), without a square root per pixel and ray
__device__ __forceinline__ float sqrt_threshold(float r) {
    float t = r * r;
    while (sqrtf(t) > r) t = __uint_as_float(__float_as_uint(t) - 1u);
    while (sqrtf(__uint_as_float(__float_as_uint(t) + 1u)) <= r) t = __uint_as_float(__float_as_uint(t) + 1u);
    return t;
}

__device__ __forceinline__ bool lane_of_mask(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
// (bits << 1) | (a <= b) in two instructions: the comparison into VCC, then bits + bits + carry-in (the compiler's own form
// of "or in a bit where the test holds" is a compare, a select and an or)
__device__ __forceinline__ unsigned shift_in_le(unsigned bits, float a, float b) {
    asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(a), "v"(b) : "vcc");
    return bits;
}

// Must be called by all 64 lanes of the wave.  Returns this lane's share of the number of rendered pixels (the wave's
// total is what the callers accumulate).
__device__ __forceinline__ int erf_splat_wave(double *image, int W, int H, const SplatReq &q, SplatLds &lds) {
    const unsigned long long any = ballot(q.valid);
    if (any == 0) return 0;
    const int lane = threadIdx.x & 63;
    const int nw = q.c1 - q.c0 + 1, nh = q.r1 - q.r0 + 1;
    const int big = 0x3fffffff;
    const int cmin = wave_min_i(q.valid ? q.c0 : big), cmax = wave_max_i(q.valid ? q.c1 : -big);
    const int rmin = wave_min_i(q.valid ? q.r0 : big), rmax = wave_max_i(q.valid ? q.r1 : -big);
    const int tiles_x = (cmax - cmin) / 8 + 1, tiles_y = (rmax - rmin) / 8 + 1;
    // the render radius is a camera constant (render_fraction * D): wave-uniform; r0 travels in 20 bits
    const float rfD = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(__shfl(q.rfD, __ffsll((long long)any) - 1, 64))));
    const bool fits = !q.valid || (nw <= kSplatSlots && nh <= kSplatSlots && q.rfD == rfD && q.r0 > -(1 << 18) && q.r0 < (1 << 18));
    if (ballot(!fits) != 0 || tiles_x * tiles_y > kSplatTiles || !(rfD > 0.f && rfD < 1.0e4f))     // wave-uniform branch
        return q.valid ? erf_splat_lane(image, W, H, q) : 0;
    const float rad2_max = sqrt_threshold(rfD);
    const float sqrt8 = sqrtf(8.0f);
    // ---- park this ray's factors.  The reference evaluates erf at both edges of every pixel; neighbouring pixels share
    // an edge, and (idx - X) +- 0.5 is the same double from either side whenever the f32 difference idx - X is exact --
    // it is for X >= 8 (|idx - X| < 8 then needs no more bits than X has).  Rays nearer the image's first columns /
    // rows evaluate both edges like the reference does.
    if (q.valid) {
        const bool exact_x = q.X >= 8.0f, exact_y = q.Y >= 8.0f;
        double *fx = lds.f[lane], *fy = lds.f[lane] + kSplatSlots;
        // the reference's sqrt8 * (idx - X -+ 0.5) / D: the quotients through the reciprocal of D (photon_det_div_rcp: the same
        // correctly rounded values as the divisions, three operations each instead of eleven)
        const double Dd = q.D, rD = 1.0 / Dd;
        if (exact_x) {
            double lo = photon_det_erf(photon_det_div_rcp(sqrt8 * (q.c0 - q.X - 0.5), Dd, rD));
#pragma unroll
            for (int j = 0; j < kSplatSlots; j++) {
                if (j < nw) {
                    const double hi = photon_det_erf(photon_det_div_rcp(sqrt8 * (q.c0 + j - q.X + 0.5), Dd, rD));
                    fx[j] = q.scale * (lo - hi);
                    lo = hi;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kSplatSlots; j++)
                if (j < nw) fx[j] = q.scale * erf_edge_diff(sqrt8, q.c0 + j, q.X, q.D);
        }
        if (exact_y) {
            double lo = photon_det_erf(photon_det_div_rcp(sqrt8 * (q.r0 - q.Y - 0.5), Dd, rD));
#pragma unroll
            for (int j = 0; j < kSplatSlots; j++) {
                if (j < nh) {
                    const double hi = photon_det_erf(photon_det_div_rcp(sqrt8 * (q.r0 + j - q.Y + 0.5), Dd, rD));
                    fy[j] = lo - hi;
                    lo = hi;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kSplatSlots; j++)
                if (j < nh) fy[j] = erf_edge_diff(sqrt8, q.r0 + j, q.Y, q.D);
        }
    }
    // ---- which pixels of its window this ray renders, as a bit pattern: bit 8 jy + jx <-> pixel (c0 + jx, r0 + jy), set when
    // the pixel lies in the window, in the image and within the render radius -- sqrtf((col - X)^2 + (row - Y)^2) <= rfD,
    // the reference's test (.cu:1504), as rad2 <= rad2_max.  49 tests per RAY, here, instead of one per ray AND pixel lane in
    // the loop below.
    unsigned long long pattern = 0;
    if (q.valid) {
        float dx2[kSplatSlots];
        unsigned col_ok = 0;
#pragma unroll
        for (int jx = 0; jx < kSplatSlots; jx++) {
            const int col = q.c0 + jx;
            dx2[jx] = (col - q.X) * (col - q.X);
            col_ok |= (jx < nw && col >= 0 && col <= W - 1) ? 1u << jx : 0u;
        }
#pragma unroll
        for (int jy = 0; jy < kSplatSlots; jy++) {
            const int row = q.r0 + jy;
            const float dy2 = (row - q.Y) * (row - q.Y);
            unsigned bits = 0;
#pragma unroll
            for (int jx = kSplatSlots - 1; jx >= 0; jx--) bits = shift_in_le(bits, dx2[jx] + dy2, rad2_max);      // bit jx <- pixel (jx, jy)
            bits = (jy < nh && row >= 0 && row <= H - 1) ? bits & col_ok : 0u;
            pattern |= (unsigned long long)bits << (8 * jy);
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- every lane is now BOTH a ray (what it parked) and a pixel of an 8x8 tile of the wave's window: lane l <-> pixel
    // (tc + (l & 7), tr + (l >> 3)).  Per tile, first as a ray: the lane moves its pattern to tile coordinates -- a 64-bit
    // shift by 8 dy + dx, (dx, dy) its window's origin relative to the tile, columns that would wrap into a neighbouring row
    // masked off first -- which makes it THE LANE MASK of the pixels this ray renders in this tile, and works out where in its
    // parked factors pixel (0, 0) of the tile would sit.  Then as a pixel: a wave-uniform loop over the rays with a non-empty
    // mask reads ray r's mask and offsets out of lane r's registers (v_readlane -> scalar registers), and under that mask the
    // vector unit multiplies the two factors, rounds the increment to f32 like the reference (.cu:1519-1528: float inc,
    // atomicAdd(float)) and adds it: 2 LDS reads + 6 vector instructions per ray and tile where the per-pixel form of rounds
    // 2-4 (unpack, clamp, radius test, selects) took ~25.
    const int lx = lane & 7, ly = lane >> 3;
    const unsigned f_base = (unsigned)(size_t)&lds.f[0][0];              // LDS byte address of the parked factors
    const unsigned px_addr = f_base + 8u * (unsigned)lx, py_addr = f_base + 8u * (unsigned)(kSplatSlots + ly);
    int taps = 0;
    for (int ty = 0; ty < tiles_y; ty++)
        for (int tx = 0; tx < tiles_x; tx++) {
            const int tc = cmin + 8 * tx, tr = rmin + 8 * ty;           // tile origin (wave-uniform)
            // as a ray
            const int dx = q.c0 - tc, dy = q.r0 - tr;                   // in [-6, 7] for a window that meets the tile
            const bool touch = q.valid && dx <= 7 && dx >= -(kSplatSlots - 1) && dy <= 7 && dy >= -(kSplatSlots - 1);
            const int lo = dx < 0 ? -dx : 0, hi = dx > 0 ? 7 - dx : 7;  // columns jx of the pattern with 0 <= jx + dx <= 7
            const unsigned row_bits = (0xffu >> (7 - hi)) & (0xffu << lo) & 0xffu;
            const unsigned cols = row_bits * 0x01010101u;
            const unsigned long long kept = pattern & (((unsigned long long)cols << 32) | cols);
            const int sh = dy * 8 + dx;                                 // in [-54, 63]: rows that leave the tile drop out of the word
            const unsigned long long mine = touch ? (sh >= 0 ? kept << sh : kept >> -sh) : 0ull;
            const int m_lo = (int)(unsigned)mine, m_hi = (int)(unsigned)(mine >> 32);
            taps += __popcll(mine);
            unsigned long long rays = ballot(mine != 0);                // rays that render something in this tile
            if (rays == 0) continue;
            // as a pixel
            double sum = 0.0;
            const int first = __ffsll((long long)rays) - 1;
            const int dx0 = __builtin_amdgcn_readlane(dx, first), dy0 = __builtin_amdgcn_readlane(dy, first);
            if (ballot(mine != 0 && (dx != dx0 || dy != dy0)) == 0) {       // wave-uniform
                // EVERY ray that renders here has the same window (one light source per wave: seven BOS waves in eight): the
                // factors of ray r sit at a compile-time offset from one per-lane address -- the loop over r is unrolled, no
                // offset travels through v_readlane, no address is formed per ray: 2 v_readlane + 2 LDS reads + 4 f64
                // instructions per ray.
                const unsigned ax = px_addr - 8u * (unsigned)dx0, ay = py_addr - 8u * (unsigned)dy0;
#pragma unroll
                for (int r = 0; r < 64; r++) {
                    const unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(m_hi, r) << 32) |
                                                 (unsigned)__builtin_amdgcn_readlane(m_lo, r);
                    if (lane_of_mask(m)) {
                        const double fx = *reinterpret_cast<const __attribute__((address_space(3))) double *>(ax + r * kSplatStride);
                        const double fy = *reinterpret_cast<const __attribute__((address_space(3))) double *>(ay + r * kSplatStride);
                        const float inc = (float)(fx * fy);
                        sum += (double)inc;
                    }
                }
            } else {
                // windows differ (a wave across two sources, a cone across a pixel boundary): ray r's offsets travel with its
                // mask (both in one word: they fit 16 bits each)
                const int offs = ((lane * (int)kSplatStride - 8 * dx) & 0xffff) | ((lane * (int)kSplatStride - 8 * dy) << 16);
                while (rays != 0) {
                    const int r = __ffsll((long long)rays) - 1;
                    rays &= rays - 1;
                    const unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(m_hi, r) << 32) |
                                                 (unsigned)__builtin_amdgcn_readlane(m_lo, r);
                    const int o = __builtin_amdgcn_readlane(offs, r);
                    const unsigned ax = px_addr + (unsigned)(int)(short)(o & 0xffff), ay = py_addr + (unsigned)(o >> 16);
                    if (lane_of_mask(m)) {
                        const double fx = *reinterpret_cast<const __attribute__((address_space(3))) double *>(ax);
                        const double fy = *reinterpret_cast<const __attribute__((address_space(3))) double *>(ay);
                        const float inc = (float)(fx * fy);
                        sum += (double)inc;
                    }
                }
            }
            if (sum != 0.0) atomicAdd(&image[(size_t)(tr + ly) * W + (tc + lx)], sum);      // pixels outside the image never receive anything
        }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();                                    // the area may be re-parked by the caller's next use
    return taps;
}

struct Pixel { float d_x, d_y; bool inside; };

__device__ __forceinline__ f3 sensor_hit(const Ray &ray, float a, float b, float c, float d, f3 dir) {
    const float t = -(dot(mk3(a, b, c), ray.pos) + d) / dot(mk3(a, b, c), dir);
    return ray.pos + dir * t;
}

// Gaussian position noise on the sensor hit, sigma in pixels (.cu:1424-1434, 1607-1616, 1773-1783)
__device__ __forceinline__ void add_position_noise(f3 &hit, const camera_design_t &cam, const NoiseDev &nz,
                                                   unsigned long long ray_id) {
    if (!nz.add_pos) return;
    float n0, n1;
    photon_normal2(nz.seed, ray_id, 0, PHOTON_STREAM_POS_NOISE, &n0, &n1);
    hit.x += n0 * nz.pos_std * cam.pixel_pitch;
    hit.y += n1 * nz.pos_std * cam.pixel_pitch;
}

// intersect_sensor_02 (.cu:1383-1543): sensor hit, x axis flipped; fills the splat request.
// Returns the final position (NaN = outside the sensor, no splat).
__device__ __forceinline__ f3 sensor_diffraction(const Ray &ray, const camera_design_t &cam, SplatReq &req,
                                                 const NoiseDev &nz, unsigned long long ray_id) {
    f3 hit = sensor_hit(ray, 0.0f, 0.0f, 1.0f, -cam.z_sensor, ray.dir);
    add_position_noise(hit, cam, nz, ray_id);
    const float p1x = (float)(-cam.pixel_pitch * (cam.x_pixel_number - 1) / 2.0);
    const float p1y = (float)(-cam.pixel_pitch * (cam.y_pixel_number - 1) / 2.0);
    const float d_x = cam.x_pixel_number - 1 - (hit.x - p1x) / cam.pixel_pitch;
    const float d_y = (hit.y - p1y) / cam.pixel_pitch;
    if (d_x >= cam.x_pixel_number || d_y >= cam.y_pixel_number || d_x < 0 || d_y < 0) return nan3();
    req = erf_splat_prepare(true, d_x, d_y, ray.radiance, ray.dir, cam.diffraction_diameter, 0.75f);
    return hit;
}

// create_apparent_image (.cu:1545-1733): back-project to the object plane, scale by the
// thin-lens magnification, splat with render_fraction 1.
__device__ __forceinline__ f3 apparent_image(const Ray &ray, const camera_design_t &cam, float z_object, float z_offset,
                                             const element_data_t &e, SplatReq &req, const NoiseDev &nz,
                                             unsigned long long ray_id) {
    const f3 dir = -ray.dir;
    f3 hit = sensor_hit(ray, 0.0f, 0.0f, -1.0f, z_object, dir);
    const float focal = e.element_properties.thin_lens_focal_length;
    const float M = focal / (z_object - z_offset - focal);
    hit.x = -hit.x * M;
    hit.y = -hit.y * M;
    add_position_noise(hit, cam, nz, ray_id);
    const float p1x = (float)(-cam.pixel_pitch * (cam.x_pixel_number - 1) / 2.0);
    const float p1y = (float)(-cam.pixel_pitch * (cam.y_pixel_number - 1) / 2.0);
    const float d_x = cam.x_pixel_number - 1 - (hit.x - p1x) / cam.pixel_pitch;
    const float d_y = (hit.y - p1y) / cam.pixel_pitch;
    if (d_x >= cam.x_pixel_number || d_y >= cam.y_pixel_number || d_x < 0 || d_y < 0) return nan3();
    req = erf_splat_prepare(true, d_x, d_y, ray.radiance, dir, cam.diffraction_diameter, 1.0f);
    return hit;
}

// intersect_sensor + 4-pixel area-weighted splat (.cu:1735-1895, :2199-2234).  The increments are handed back in
// `req` (bilinear_splat_wave adds them to the image); returns the sensor hit (NaN = outside the sensor, nothing to add).
struct TapReq {
    bool valid;
    int ii_ul, jj_ul;           // upper-left pixel of the four (row, column)
    float inc[4];               // the reference's f32 increments for (ii, jj), (ii, jj+1), (ii+1, jj), (ii+1, jj+1)
};

__device__ __forceinline__ f3 sensor_bilinear(const Ray &ray, const camera_design_t &cam, TapReq &req, const NoiseDev &nz,
                                              unsigned long long ray_id) {
    f3 hit = sensor_hit(ray, 0.0f, 0.0f, 1.0f, -cam.z_sensor, ray.dir);
    add_position_noise(hit, cam, nz, ray_id);
    const f3 dir = ray.dir;
    const float alpha = photon_det_atanf(sqrtf((dir.x / dir.z) * (dir.x / dir.z) + (dir.y / dir.z) * (dir.y / dir.z)));
    const float ca = photon_det_cosf(alpha);
    const double cos4 = ca * ca * ca * ca;
    const float p1x = (float)(-cam.pixel_pitch * (cam.x_pixel_number - 1) / 2.0);
    const float p1y = (float)(-cam.pixel_pitch * (cam.y_pixel_number - 1) / 2.0);
    const float d_x = (hit.x - p1x) / cam.pixel_pitch;
    const float d_y = (hit.y - p1y) / cam.pixel_pitch;
    if (d_x >= cam.x_pixel_number || d_y >= cam.y_pixel_number || d_x < 0 || d_y < 0) return nan3();
    const float d_y_lower = d_y - 0.5, d_x_lower = d_x - 0.5;
    const double d_ii_ul = ceilf(d_y_lower) - d_y_lower;
    const double d_jj_ul = ceilf(d_x_lower) - d_x_lower;
    const double w[4] = {d_ii_ul * d_jj_ul, d_ii_ul * (1 - d_jj_ul), (1 - d_ii_ul) * d_jj_ul,
                         (1 - d_ii_ul) * (1 - d_jj_ul)};
    req.valid = true;
    req.ii_ul = (int)(ceilf(d_y_lower) - 1);
    req.jj_ul = (int)(ceilf(d_x_lower) - 1);
#pragma unroll
    for (int k = 0; k < 4; k++) req.inc[k] = (float)(w[k] * ray.radiance * cos4);
    return hit;
}

// which of a hit's four taps the reference actually adds (.cu:2223-2233): pixel inside the sensor, and its index
// (ii-1)*W + jj-1 not before the image (the reference writes there; skipped here and in the oracle)
__device__ __forceinline__ bool tap_lands(int ii, int jj, int W, int H) {
    return !(ii < 0 || ii >= H || jj < 0 || jj >= W) && ((long)(ii - 1) * W + jj - 1) >= 0;
}

// per-lane form (fallback): one atomic per tap
__device__ __forceinline__ int bilinear_splat_lane(double *image, int W, int H, const TapReq &q) {
    int taps = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int ii = q.ii_ul + (k >> 1), jj = q.jj_ul + (k & 1);
        if (!tap_lands(ii, jj, W, H)) continue;
        atomicAdd(&image[(long)(ii - 1) * W + jj - 1], (double)q.inc[k]);
        taps++;
    }
    return taps;
}

// Wave-cooperative form (see erf_splat_wave): rays park (ii_ul, jj_ul, four increments), lanes become the pixels of
// 8x8 tiles of the wave's window and sum what the parked rays add to them; one atomic per pixel per wave.  Must be
// called by all 64 lanes.
template <class LDS>
__device__ __forceinline__ int bilinear_splat_wave(double *image, int W, int H, const TapReq &q, LDS &lds) {
    const unsigned long long any = ballot(q.valid);
    if (any == 0) return 0;
    const int lane = threadIdx.x & 63;
    const int big = 0x3fffffff;
    const int cmin = wave_min_i(q.valid ? q.jj_ul : big), cmax = wave_max_i(q.valid ? q.jj_ul + 1 : -big);
    const int rmin = wave_min_i(q.valid ? q.ii_ul : big), rmax = wave_max_i(q.valid ? q.ii_ul + 1 : -big);
    const int tiles_x = (cmax - cmin) / 8 + 1, tiles_y = (rmax - rmin) / 8 + 1;
    if (tiles_x * tiles_y > kSplatTiles) return q.valid ? bilinear_splat_lane(image, W, H, q) : 0;  // wave-uniform
    if (q.valid) {
        lds.head[lane] = make_float4(__int_as_float(q.ii_ul), __int_as_float(q.jj_ul), q.inc[0], q.inc[1]);
        lds.tail(lane) = make_float2(q.inc[2], q.inc[3]);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int taps = 0;
    for (int ty = 0; ty < tiles_y; ty++)
        for (int tx = 0; tx < tiles_x; tx++) {
            const int tc = cmin + 8 * tx, tr = rmin + 8 * ty;
            const bool touch = q.valid && q.jj_ul <= tc + 7 && q.jj_ul + 1 >= tc && q.ii_ul <= tr + 7 && q.ii_ul + 1 >= tr;
            unsigned long long rays = ballot(touch);
            if (rays == 0) continue;
            const int jj = tc + (lane & 7), ii = tr + (lane >> 3);
            const bool lands = tap_lands(ii, jj, W, H);
            double sum = 0.0;
            while (rays != 0) {
                const int r = __ffsll((long long)rays) - 1;
                rays &= rays - 1;
                const float4 h = lds.head[r];
                const float2 h2 = lds.tail(r);
                const int di = ii - __float_as_int(h.x), dj = jj - __float_as_int(h.y);
                if (lands && (unsigned)di <= 1u && (unsigned)dj <= 1u) {
                    const float inc = di ? (dj ? h2.y : h2.x) : (dj ? h.w : h.z);
                    sum += (double)inc;
                    taps++;
                }
            }
            if (lands && sum != 0.0) atomicAdd(&image[(long)(ii - 1) * W + jj - 1], sum);
        }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return taps;
}

}  // namespace photon
