// photon_sensor.hip - the kernels either side of the volume march: ray generation (stage 1a) and the sensor stage
// (stage 2: lens / aperture / apparent image + erf or 4-pixel splat into the scene's private f64 accumulator, one kernel),
// and the fold of that accumulator into the caller's image.
#include "photon_internal.hpp"

using namespace photon;

// Stage 1a (density gradients on): generate the ray and move it into the volume's world frame
// (parallel_ray_tracing.cu:2004-2082).  Kept apart from the march so that the hot kernel carries
// neither the scene description (a kilobyte of kernel arguments pinned in SGPRs) nor the
// double-precision ray-generation code in its register budget.
__global__ __launch_bounds__(256) void raygen_kernel(SceneDev sc, long long src_begin, unsigned n_rays, RayStateDev st) {
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    double radiance;
    const RayPD g = generate_state(sc, src_begin, n_rays, r, radiance);
    const f3 p = g.p, d = g.d;
    st.px[r] = p.x; st.py[r] = p.y; st.pz[r] = p.z;
    st.dx[r] = d.x; st.dy[r] = d.y; st.dz[r] = d.z;
    st.radiance[r] = radiance;
}

// Stage 2: everything after the volume (parallel_ray_tracing.cu:2136-2241): back to the camera frame, lens / aperture /
// apparent image, and the wave-cooperative splats (device_optics.hpp), in ONE kernel.  FROM_STATE=false (no density
// gradients) generates the ray in place, so that path is a single kernel altogether.  (Rounds 2-4 ran the erf splat as a
// second kernel, splat_kernel, fed through the consumed state arrays: optics and splat together wanted 94 VGPRs.  With the
// splat's pixel loop on the scalar unit -- round 5 -- the fused kernel needs 64, and one launch and a 24 B/ray round trip
// fewer are worth 0.12 ms per 1e7 rays: C3 trilinear 16.79 -> 16.67 ms per step, its eighth 2.285 -> 2.260, same box.)
#ifndef PHOTON_SENSOR_WAVES
#define PHOTON_SENSOR_WAVES 5           // the cooperative splats park 8 KiB per wave in LDS: five blocks per CU
#endif
// SPLAT: 0 = which splat a ray gets is decided where it lands (apparent image / diffraction spot / 4 pixels: the camera's
// switches); 2 = a camera WITHOUT diffraction behind a real first element -- every ray takes the 4-pixel splat (photon's PIV
// frames): the instantiation carries neither the erf code nor its 8 KiB of parked factors per wave, so more waves fit
// (PHOTON_SENSOR_WAVES_TAPS) under the f64 generation chains it waits on.
#ifndef PHOTON_SENSOR_WAVES_TAPS
#define PHOTON_SENSOR_WAVES_TAPS 6
#endif
template <int SPLAT> struct SplatArea { typedef SplatLds type; };
template <> struct SplatArea<2> { typedef TapLds type; };
template <bool FROM_STATE, bool TRAIN, int SPLAT>
__global__ __launch_bounds__(256, (SPLAT == 2 ? PHOTON_SENSOR_WAVES_TAPS : PHOTON_SENSOR_WAVES)) void sensor_kernel(SceneDev sc, long long src_begin, unsigned n_rays, RayStateDev st,
                                                     double *image, DumpDev dump, unsigned long long *counters) {
    __shared__ typename SplatArea<SPLAT>::type splat_lds[4];            // per wave: the parked rays of the cooperative splats
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
            if (SPLAT != 2 && sc.elems[0].element_type == 'n') {        // .cu:2143-2158
                const float z_obj = sc.object_distance + sc.z_offset;
                fin = apparent_image(ray, sc.cam, z_obj, sc.z_offset, sc.elems[0], req, sc.noise, ray_id);
                have_fin = true;
                on_sensor = !isnan(fin.x);
            } else {
                ray = optical_system<TRAIN>(sc, ray);
                if (!(isnan3(ray.dir) || isnan3(ray.pos))) {            // .cu:2172-2176
                    if (SPLAT != 2 && sc.cam.implement_diffraction) {
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
    if constexpr (SPLAT != 2) taps += erf_splat_wave(image, sc.cam.x_pixel_number, sc.cam.y_pixel_number, req, splat_lds[threadIdx.x >> 6]);      // all 64 lanes
    taps += bilinear_splat_wave(image, sc.cam.x_pixel_number, sc.cam.y_pixel_number, tap, splat_lds[threadIdx.x >> 6]);     // all 64 lanes
    wave_add(&counter_slot(counters)[CNT_TAPS], (unsigned long long)taps);
    wave_add(&counter_slot(counters)[CNT_ON_SENSOR], (unsigned long long)on_sensor);
}

// image_array is read-modify-write (parallel_ray_tracing.cu:3309,3675): fold the f64 accumulator of
// this call into the caller's f32 image, one rounding per pixel.
// The accumulator is left zeroed for the next trace of the scene (one launch fewer per trace than a memset would be).
__global__ __launch_bounds__(256) void finalize_image_kernel(float *__restrict__ image, double *__restrict__ acc, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        image[i] = (float)((double)image[i] + acc[i]);
        acc[i] = 0.0;
    }
}

namespace photon {

int launch_raygen(photon_scene *s, long long src_begin, unsigned n, hipStream_t stream) {
    hipLaunchKernelGGL(raygen_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, s->dev, src_begin, n, s->ws);
    PH_CHECK(hipGetLastError());
    return 0;
}

int launch_sensor(photon_scene *s, bool from_state, long long src_begin, unsigned n, const DumpDev &dump, hipStream_t stream) {
    const dim3 block(256), grid((n + 255) / 256);
    double *d_image = s->d_acc;
    // a camera without diffraction behind a real first element: the 4-pixel-only instantiations (the default element path; after a march: C5)
    const bool taps_only = !s->dev.train_mode && !s->dev.cam.implement_diffraction && s->dev.elems[0].element_type != 'n';
    if (taps_only && !from_state) hipLaunchKernelGGL((sensor_kernel<false, false, 2>), grid, block, 0, stream, s->dev, src_begin, n, s->ws, d_image, dump, s->d_counters);
    else if (taps_only) hipLaunchKernelGGL((sensor_kernel<true, false, 2>), grid, block, 0, stream, s->dev, src_begin, n, s->ws, d_image, dump, s->d_counters);
    else if (from_state) {
        if (s->dev.train_mode) hipLaunchKernelGGL((sensor_kernel<true, true, 0>), grid, block, 0, stream, s->dev, src_begin, n, s->ws, d_image, dump, s->d_counters);
        else hipLaunchKernelGGL((sensor_kernel<true, false, 0>), grid, block, 0, stream, s->dev, src_begin, n, s->ws, d_image, dump, s->d_counters);
    } else {
        if (s->dev.train_mode) hipLaunchKernelGGL((sensor_kernel<false, true, 0>), grid, block, 0, stream, s->dev, src_begin, n, s->ws, d_image, dump, s->d_counters);
        else hipLaunchKernelGGL((sensor_kernel<false, false, 0>), grid, block, 0, stream, s->dev, src_begin, n, s->ws, d_image, dump, s->d_counters);
    }
    PH_CHECK(hipGetLastError());
    return 0;
}

int launch_finalize(photon_scene *s, float *d_image, hipStream_t stream) {
    const size_t npix = (size_t)s->dev.cam.x_pixel_number * s->dev.cam.y_pixel_number;
    hipLaunchKernelGGL(finalize_image_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream, d_image, s->d_acc, npix);
    PH_CHECK(hipGetLastError());
    s->acc_clean = true;
    return 0;
}

}  // namespace photon
