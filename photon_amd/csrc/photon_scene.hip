// photon_scene.hip - scene and source handles: what start_ray_tracing uploads before its launch loop
// (parallel_ray_tracing.cu:3132-3314) as a device-resident scene, the glibc srand(10) lens-sample table, and the
// on-device scene generators.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "photon_internal.hpp"

using namespace photon;

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

// A scene's small host arrays (tables, optics, a shard's sources) travel in ONE block and one host-to-device copy: a scene
// is built per start_ray_tracing call -- per device and call with PHOTON_DEVICES -- and thirteen synchronous copies of a
// few kilobytes each cost more than the bytes (measured: 0.15 ms of a call).  Arrays beyond kPackLimit are copied straight
// from the caller's memory (staging 24 MB of source coordinates through another host buffer would cost more than it saves).
constexpr size_t kPackLimit = 256 << 10, kPackAlign = 256;
struct UploadPack {
    struct Item { size_t offset; const void **slot; };
    std::vector<char> host;
    std::vector<Item> items;
};
template <typename T>
static int upload(photon_scene *s, UploadPack &pack, const T *host, size_t n, const T **dev_out) {
    const size_t bytes = n * sizeof(T);
    if (bytes > kPackLimit) {
        T *d = nullptr;
        PH_CHECK(pool_malloc((void **)&d, bytes));
        s->allocs.push_back(d);
        PH_CHECK(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
        *dev_out = d;
        return 0;
    }
    const size_t offset = (pack.host.size() + kPackAlign - 1) / kPackAlign * kPackAlign;
    pack.host.resize(offset + std::max<size_t>(bytes, sizeof(T)));           // an empty array still gets a valid address
    if (bytes) memcpy(pack.host.data() + offset, host, bytes);
    pack.items.push_back({offset, reinterpret_cast<const void **>(dev_out)});
    *dev_out = nullptr;
    return 0;
}
// a zeroed region of the block (the statistics counters, the work queues): no fill kernel, no second block
template <typename T>
static void reserve_zeroed(UploadPack &pack, size_t n, T **dev_out) {
    const size_t offset = (pack.host.size() + kPackAlign - 1) / kPackAlign * kPackAlign;
    pack.host.resize(offset + n * sizeof(T));                           // std::vector value-initialises: zeros
    pack.items.push_back({offset, reinterpret_cast<const void **>(const_cast<const T **>(dev_out))});
    *dev_out = nullptr;
}
static int flush_uploads(photon_scene *s, UploadPack &pack) {
    if (pack.items.empty()) return 0;
    char *d = nullptr;
    PH_CHECK(pool_malloc((void **)&d, pack.host.size()));
    s->allocs.push_back(d);
    PH_CHECK(hipMemcpy(d, pack.host.data(), pack.host.size(), hipMemcpyHostToDevice));
    for (const auto &it : pack.items) *it.slot = d + it.offset;
    return 0;
}

template <typename T>
static int copy_device(photon_scene *s, const T *dev_src, size_t n, const T **dev_out) {
    T *d = nullptr;
    PH_CHECK(pool_malloc((void **)&d, std::max<size_t>(n, 1) * sizeof(T)));
    s->allocs.push_back(d);
    if (n) PH_CHECK(hipMemcpyAsync(d, dev_src, n * sizeof(T), hipMemcpyDeviceToDevice, nullptr));     // the caller waits for the null stream
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

// =============================================================================================
// C-ABI: extension entry points
// =============================================================================================
extern "C" {

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

void photon_scene_free(photon_scene_t *s) {
    if (!s) return;
    // The blocks below go back to the CACHE, not to the runtime (whose hipFree would wait for the device): the next scene of
    // the same shape may be handed them at once and overwrite them with copies on the null stream, which does not wait for
    // kernels of this scene still running on a non-blocking stream.  So wait here; microseconds on an idle device.
    photon::DeviceScope on_scene_device(s->device);             // the caller may have another device current: wait on, and free into, the scene's
    scene_quiesce(s);
    for (void *p : s->allocs) pool_free(p);
    pool_free(s->ws.px);
    pool_free(s->ws.radiance);
    free_resume_state(s);
    pool_free(s->d_profile);                                    // (d_counters and d_queue live in the upload block: allocs)
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
    if (device_malloc((void **)&src->x, m * sizeof(float)) != hipSuccess || device_malloc((void **)&src->y, m * sizeof(float)) != hipSuccess ||
        device_malloc((void **)&src->z, m * sizeof(float)) != hipSuccess ||
        device_malloc((void **)&src->radiance, m * sizeof(double)) != hipSuccess ||
        device_malloc((void **)&src->diameter_index, m * sizeof(int)) != hipSuccess) {
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
    if (device_malloc((void **)&d_in, total * sizeof(double)) != hipSuccess) return fail(3);
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
        if (device_malloc((void **)&d_cdf, n_diameters * sizeof(double)) != hipSuccess) return fail(3);
        if (hipMemcpy(d_cdf, diameter_cdf, n_diameters * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(4);
    }
    if (n) {
        hipLaunchKernelGGL(sources_piv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (unsigned long long)seed, n, f,
                           d_cdf, src->x, src->y, src->z, src->radiance, src->diameter_index);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(4);
    }
    if (d_cdf) (void)hipFree(d_cdf);
    {   // the box the particles were drawn from (sources_piv_kernel: X, Y uniform in the box, z = Z + z_object), a rounding to float wider
        const double ax = std::max(fabs(box_min[0]), fabs(box_max[0])), ay = std::max(fabs(box_min[1]), fabs(box_max[1]));
        const double z0 = std::min(box_min[2], box_max[2]) + z_object, z1 = std::max(box_min[2], box_max[2]) + z_object;
        src->rmax = sqrt(ax * ax + ay * ay) * (1 + 1e-6);
        src->zmin = z0 - 1e-6 * fabs(z0) - 1e-3;
        src->zmax = z1 + 1e-6 * fabs(z1) + 1e-3;
        src->have_extent = src->rmax == src->rmax && src->zmin == src->zmin && src->zmax == src->zmax;
    }
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

// The lens samples that CAN reach the aperture of the first element from some source of this scene.  The reference kills a
// ray whose hit on that element's front surface (a sphere, 'l', or a plane, 't') lies more than pitch / 2 from the axis
// (.cu:447, 560-566), and aims ray k of every source at the SAME point P_k = (x_lens, y_lens) of the plane z = image_distance
// (.cu:123-141: x(z) = x_s + tan(theta) (z_s - z) with tan(theta) = -(x_lens - x_s) / (image_distance - z_s)), with
// |P_k| up to ratio x pitch: a full-aperture cone loses half its rays there, the same ones for every source.  A ray through P_k
// that hit the surface at axis distance rho <= pitch / 2 and height z_h would have |P_k| <= rho + |z_h - z_a| slope, its slope
// at most (|P_k| + R) / D (R: largest axis distance of a source, D: smallest source-plane distance along z) and z_h within dz
// of the plane (the element's vertex plane +- its sag at pitch / 2).  So sample k is DEAD for every source when
//     |P_k| (1 - dz / D) - dz R / D > pitch / 2 + slack        (slack: a thousandth of the pitch, for the f32 rounding of the aim)
// and only the others are launched where nothing else needs the dead rays (launch_chunk: no volume, no dumps, reference element
// path).  Everything in double, from the caller's arrays -- or, for sources generated on the device, from the box the generator
// drew them from; geometries this does not cover (tilted or off-axis element, generated BOS patterns, a degenerate sphere)
// return every sample.
static std::vector<int> live_lens_samples(const std::vector<float> &lx, const std::vector<float> &ly, const lightfield_source_t *lsp,
                                          const photon_sources *generated, size_t n_sources, float image_distance, int num_elements,
                                          const element_data_t *edp, const double (*center)[3], const double (*plane)[4]) {
    std::vector<int> all(lx.size());
    for (size_t k = 0; k < all.size(); k++) all[k] = (int)k;
    if ((generated && !generated->have_extent) || n_sources == 0 || num_elements < 1 || lx.size() < 2) return all;
    const char type = edp[0].element_type;
    const double pitch = edp[0].element_geometry.pitch;
    if ((type != 'l' && type != 't') || !(pitch > 0)) return all;
    // (the dz bound below puts the front vertex on the +z side of the centre: plane normal c > 0, what the reference's Python always
    // emits, perform_ray_tracing_03.py:49; a flipped normal moves the front sphere, .cu:557 -- leave that to the kernels)
    if (plane[0][0] != 0.0 || plane[0][1] != 0.0 || !(plane[0][2] > 0.0) || center[0][0] != 0.0 || center[0][1] != 0.0) return all;
    const double za = image_distance;
    double dz;
    if (type == 't') {
        dz = fabs(-plane[0][3] / plane[0][2] - za);
    } else {
        const double R = fabs((double)edp[0].element_geometry.front_surface_radius), t = fabs(edp[0].element_geometry.vertex_distance);
        if (!(R > pitch / 2) || !(t == t)) return all;
        const double sag = R - sqrt(R * R - pitch * pitch / 4);
        dz = fabs(center[0][2] - za) + t / 2 + sag;
    }
    double rmax = 0, dmin = HUGE_VAL;
    if (generated) {                                                    // sources made on the device: the generator's box stands in for them
        rmax = generated->rmax;
        dmin = za < generated->zmin ? generated->zmin - za : (za > generated->zmax ? za - generated->zmax : 0.0);
    } else
    {                                                                   // (one sqrt at the end: this loop runs per start_ray_tracing call)
        double r2max = 0, nan_probe = 0;
        for (size_t i = 0; i < n_sources; i++) {
            const double x = lsp->x[i], y = lsp->y[i], r2 = x * x + y * y, dd = fabs(za - (double)lsp->z[i]);
            nan_probe += r2 * 0.0 + dd * 0.0;                           // NaN (or infinity) anywhere -> NaN
            r2max = r2 > r2max ? r2 : r2max;
            dmin = dd < dmin ? dd : dmin;
        }
        if (!(nan_probe == 0.0)) return all;                           // a NaN source: leave everything to the kernels
        rmax = sqrt(r2max);
    }
    if (!(dz == dz) || !(dmin > 16 * dz)) return all;
    std::vector<int> live;
    for (size_t k = 0; k < lx.size(); k++) {
        const double r = sqrt((double)lx[k] * lx[k] + (double)ly[k] * ly[k]);
        const bool dead = r * (1 - dz / dmin) - dz * rmax / dmin > pitch / 2 + 1e-3 * pitch + 1e-4 * r;
        if (!dead) live.push_back((int)k);
    }
    if (live.empty()) live.push_back(0);                                // a launch of zero rays per source is nobody's friend
    return live;
}

// The sources whose image CANNOT fall on the sensor, whatever lens sample the ray is aimed at: they need not be launched on the
// volume-free path.  photon's sample PIV frame draws its particles over a field 1.5 x wider than the camera sees
// (run_simulation_02.py:956-958): more than half of them image beside the sensor.
//
// One biconvex thick lens ('l', on the z axis, normal +z) -- or one thin lens ('t': lens_cull_setup) --, then the sensor plane z = z_sensor.  Everything a surviving ray
// does is, in the xy plane, a linear combination of two vectors -- its aim point P on the plane z = image_distance (.cu:123-141)
// and its source's S = (x_s, y_s) -- with SCALAR coefficients, because the lens is rotationally symmetric and every surface
// normal's xy part is the hit point's over the radius:
//     H1 = (1 + e) P - e S                     front hit;  e = (z_H1 - z_a) / (z_a - z_s), z_H1 within the front sag of the vertex
//     n v = u - a1 H1,   u = q (P - S)         Snell in vector form (.cu:652-682); q = 1 / |P - S| (3-D), a1 = G1 / R1,
//                                              G1 = n cos(t') - cos(t) = sqrt(n^2 - sin^2 t) - cos t: n - 1 at normal incidence, growing with t
//     H2 = H1 + s2 v                           back hit; s2 = glass path = (z_H1 - z_H2) / |v_z|
//     w  = n v - a2 H2                         a2 = G2 / |R2|, G2 = n cos(t) - cos(t') likewise from n - 1 upwards (.cu:797-827)
//     h  = H2 + tau w                          sensor hit; tau = (z_H2 - z_sensor) / |w_z|
// so h = A P + B S with A, B polynomials in (e, q, a1, a2, s2, tau).  Each of the six lies in an interval that follows from the
// aperture tests alone (both hits within pitch / 2 of the axis, .cu:560-566, 737-743: a ray that fails one is dead anyway):
// sin(incidence) <= |u_xy| + (pitch / 2) / R, the sags of the two caps, |v_xy| and |w_xy| from the same sums.  A and B are
// evaluated in interval arithmetic; |P| <= the largest lens sample, and for a ray that passes the front aperture also
// <= (pitch / 2 + |e| r_s) / (1 - |e|).  The source is OFF when the box  B S +- |A|max |P|max +- slack  misses the rectangle of
// sensor hits that reach a pixel (.cu:1440-1452, 1803-1815: half a pixel beyond the array either side; one more pixel here).
// slack: 10 um + 1e-5 of the ray's length for the kernels' f32 arithmetic (its cancellation in the sphere intersection is worth
// 1.5 um along the ray, tests/test_oracle_golden.py).  Held against exact f64 ray tracing of every ray of every culled source in
// tests/test_parity_gpu.py::test_culled_sources_against_exact_geometry; geometries this does not cover keep every source.
}  // extern "C"  (overloaded helpers below)
namespace {
#define PH_HD __host__ __device__ inline
PH_HD double dmin2(double a, double b) { return a < b ? a : b; }
PH_HD double dmax2(double a, double b) { return a > b ? a : b; }
struct Ivl {
    double lo, hi;
};
PH_HD Ivl iv(double a) { return Ivl{a, a}; }
PH_HD Ivl iv(double a, double b) { return a <= b ? Ivl{a, b} : Ivl{b, a}; }
PH_HD Ivl operator+(Ivl a, Ivl b) { return Ivl{a.lo + b.lo, a.hi + b.hi}; }
PH_HD Ivl operator-(Ivl a) { return Ivl{-a.hi, -a.lo}; }
PH_HD Ivl operator-(Ivl a, Ivl b) { return a + (-b); }
PH_HD Ivl operator*(Ivl a, Ivl b) {
    const double c0 = a.lo * b.lo, c1 = a.lo * b.hi, c2 = a.hi * b.lo, c3 = a.hi * b.hi;
    return Ivl{dmin2(dmin2(c0, c1), dmin2(c2, c3)), dmax2(dmax2(c0, c1), dmax2(c2, c3))};
}
PH_HD Ivl operator*(Ivl a, double b) { return a * iv(b); }
PH_HD double mag(Ivl a) { return dmax2(fabs(a.lo), fabs(a.hi)); }

}  // namespace

static photon::LensCull lens_cull_setup(const std::vector<float> &lx, const std::vector<float> &ly, float image_distance, float beam_wavelength,
                                int num_elements, const element_data_t *edp, const double (*center)[3], const double (*plane)[4],
                                const int *sys_index, const camera_design_t *cam) {
    photon::LensCull c;
    if (num_elements < 1 || (edp[0].element_type != 'l' && edp[0].element_type != 't')) return c;
    // the reference's element path sends the ray through element 0 once per single-member group (.cu:1331-1333): exactly once here
    {
        const int n = std::min(num_elements, kMaxElements);
        int seq = 0, applications = 0;
        for (int k = 0; k < n; k++) seq = std::max(seq, sys_index[k]);
        for (int idx = 0; idx < seq; idx++) {
            int count = 0;
            for (int k = 0; k < n; k++) count += (seq - sys_index[k] == idx);
            applications += count == 1;
        }
        if (applications != 1) return c;
    }
    if (plane[0][0] != 0.0 || plane[0][1] != 0.0 || !(plane[0][2] > 0.0) || center[0][0] != 0.0 || center[0][1] != 0.0) return c;
    const element_data_t &e = edp[0];
    if (e.element_type == 't') {
        // Thin lens: the ray meets the element's plane at H, within pitch / 2 of the axis, and leaves along u - (H - C) / f
        // (.cu:447-503).  With the centre ON the plane the z component of that is u's, so the sensor hit is
        //     h = H (1 - s / f) + s u_xy,   s = (z_plane - z_sensor) / |u_z|,   H = (1 + e) P - e S,  e = (z_plane - z_a) / (z_a - z_s):
        // exact up to the one quantity that depends on P, |P - S| in s / f (source_misses_sensor: an interval again).
        c.thin = true;
        c.focal = (double)(float)e.element_properties.thin_lens_focal_length;
        c.hp = (double)(float)e.element_geometry.pitch / 2.0;
        const double z_plane = -plane[0][3] / plane[0][2];
        if (!(c.focal > 0) || !(c.hp > 0) || !(fabs(z_plane - center[0][2]) <= 1e-9 * fabs(z_plane) + 1e-9)) return c;
        c.za = image_distance;
        c.zf = c.zb = z_plane;
        c.z_sen = cam->z_sensor;
        c.t = c.sag1 = c.sag2 = 0; c.R1 = c.R2a = 0; c.n = 1;
        if (!(c.zb > c.z_sen) || !(cam->pixel_pitch > 0)) return c;
        double rp = 0;
        for (size_t k = 0; k < lx.size(); k++) rp = std::max(rp, sqrt((double)lx[k] * lx[k] + (double)ly[k] * ly[k]));
        c.rp_all = rp * (1 + 1e-6);
        c.half_x = (double)cam->pixel_pitch * (cam->x_pixel_number + 1) / 2.0 + cam->pixel_pitch;
        c.half_y = (double)cam->pixel_pitch * (cam->y_pixel_number + 1) / 2.0 + cam->pixel_pitch;
        c.ok = c.za == c.za && c.zf == c.zf && c.half_x == c.half_x && c.half_y == c.half_y;
        return c;
    }
    c.R1 = e.element_geometry.front_surface_radius;
    c.R2a = -(double)e.element_geometry.back_surface_radius;
    c.hp = (double)(float)e.element_geometry.pitch / 2.0;                // the kernels compare against the f32 pitch
    c.t = e.element_geometry.vertex_distance;
    double n = e.element_properties.refractive_index;
    const double abbe = (float)e.element_properties.abbe_number;
    if (abbe == abbe) {                                                 // .cu:622-636: the index at the beam's wavelength
        const double lD = 589.3, lF = 486.1, lC = 656.3, w = beam_wavelength;
        n = n + (1.0 / (w * w) - 1.0 / (lD * lD)) * ((n - 1) / (abbe * (1 / (lF * lF) - 1 / (lC * lC))));
    }
    c.n = n;
    if (!(c.R1 > 0) || !(c.R2a > 0) || !(n > 1.0) || !(n < 4.0) || !(c.hp > 0) || !(c.t >= 0)) return c;
    if (!(c.hp < 0.95 * c.R1) || !(c.hp < 0.95 * c.R2a)) return c;
    c.za = image_distance;
    c.zf = center[0][2] + c.t / 2;
    c.zb = center[0][2] - c.t / 2;
    c.z_sen = cam->z_sensor;
    c.sag1 = c.R1 - sqrt(c.R1 * c.R1 - c.hp * c.hp);
    c.sag2 = c.R2a - sqrt(c.R2a * c.R2a - c.hp * c.hp);
    if (!(c.zb > c.z_sen) || !(cam->pixel_pitch > 0)) return c;
    // the two caps must not meet inside the aperture (the glass path of a surviving ray is then >= 0, which the bound on H2 uses);
    // photon's own lens has t = sag1 + sag2 exactly (run_simulation_02.py: zero edge thickness), hence the tolerance
    if (!(c.t >= 0.999 * (c.sag1 + c.sag2))) return c;
    double rp = 0;
    for (size_t k = 0; k < lx.size(); k++) rp = std::max(rp, sqrt((double)lx[k] * lx[k] + (double)ly[k] * ly[k]));
    c.rp_all = rp * (1 + 1e-6);
    c.half_x = (double)cam->pixel_pitch * (cam->x_pixel_number + 1) / 2.0 + cam->pixel_pitch;
    c.half_y = (double)cam->pixel_pitch * (cam->y_pixel_number + 1) / 2.0 + cam->pixel_pitch;
    c.ok = c.za == c.za && c.zf == c.zf && c.half_x == c.half_x && c.half_y == c.half_y;
    return c;
}

// true: no ray of the source (xs, ys, zs) that passes both apertures of the lens can reach a pixel
__host__ __device__ static bool source_misses_sensor(const photon::LensCull &c, double xs, double ys, double zs) {
    const double Ds = zs - c.za;
    if (!(Ds > 0) || !(zs > c.zf + (c.zf - c.zb))) return false;
    const double rs = sqrt(xs * xs + ys * ys);
    if (!(rs == rs)) return false;
    if (c.thin) {
        if (!(zs > c.zf + 1e-3 * Ds)) return false;
        const double e = -(c.zf - c.za) / Ds;                           // exact: the hit lies ON the plane
        if (!(fabs(e) < 0.25)) return false;
        const double rp = dmin2(c.rp_all, (c.hp + fabs(e) * rs) / (1 - fabs(e)));
        const double gmax = rs + rp, gmin = dmax2(0.0, rs - rp);
        // s / f = (z_plane - z_sensor) |P - S| / ((z_s - z_a) f): the flight to the sensor in units of the focal length
        const double k0 = (c.zf - c.z_sen) / (Ds * c.focal);
        const Ivl sig = iv(k0 * sqrt(gmin * gmin + Ds * Ds), k0 * sqrt(gmax * gmax + Ds * Ds));
        const double m1 = (c.zf - c.z_sen) / Ds;                        // s u_xy = m1 (P - S), exactly
        const Ivl one_sig = iv(1.0) - sig;
        const Ivl A = one_sig * (1 + e) + iv(m1), B = one_sig * (-e) - iv(m1);
        const double blur = mag(A) * rp + 1e-5 * (Ds + (c.zf - c.z_sen)) + 10.0;
        if (!(blur == blur)) return false;
        const Ivl bx = B * xs, by = B * ys;
        return bx.lo - blur > c.half_x || bx.hi + blur < -c.half_x || by.lo - blur > c.half_y || by.hi + blur < -c.half_y;
    }
    const Ivl e = iv(-(c.zf - c.za) / Ds, -(c.zf - c.sag1 - c.za) / Ds);
    const double em = mag(e);
    if (!(em < 0.25)) return false;
    const double rp = dmin2(c.rp_all, (c.hp + em * rs) / (1 - em));
    const double gmax = rs + rp, gmin = dmax2(0.0, rs - rp);
    const double lmax = sqrt(gmax * gmax + Ds * Ds);
    const Ivl q = iv(1.0 / lmax, 1.0 / sqrt(gmin * gmin + Ds * Ds));
    const double su = gmax / lmax;                                      // |u_xy| at most
    const double s1 = su + c.hp / c.R1;                                 // sin(incidence at the front) at most
    if (!(s1 < 0.9)) return false;
    const Ivl a1 = iv(c.n - 1, sqrt(c.n * c.n - s1 * s1) - sqrt(1 - s1 * s1)) * (1.0 / c.R1);
    const double sv = (su + a1.hi * c.hp) / c.n;                        // |v_xy| at most
    const double s2m = sv + c.hp / c.R2a;                               // sin(incidence at the back, in the glass) at most
    if (!(c.n * s2m < 0.9)) return false;
    const Ivl a2 = iv(c.n - 1, sqrt(c.n * c.n - c.n * c.n * s2m * s2m) - sqrt(1 - c.n * c.n * s2m * s2m)) * (1.0 / c.R2a);
    const Ivl s2 = iv(dmax2(0.0, c.t - c.sag1 - c.sag2), c.t / sqrt(1 - sv * sv));
    const double sw = c.n * sv + a2.hi * c.hp;                          // |w_xy| at most
    if (!(sw < 0.9)) return false;
    const Ivl tau = iv(c.zb - c.z_sen, (c.zb + c.sag2 - c.z_sen) / sqrt(1 - sw * sw));
    const Ivl one_e = iv(1.0) + e;
    const Ivl cP = q - a1 * one_e, cS = a1 * e - q;                     // n v = cP P + cS S
    const Ivl hP = one_e + s2 * cP * (1.0 / c.n), hS = s2 * cS * (1.0 / c.n) - e;      // H2 = hP P + hS S
    const Ivl k = iv(1.0) - tau * a2;
    const Ivl A = k * hP + tau * cP, B = k * hS + tau * cS;
    const double blur = mag(A) * rp + 1e-5 * (Ds + tau.hi) + 10.0;
    if (!(blur == blur)) return false;
    const Ivl bx = B * xs, by = B * ys;
    return bx.lo - blur > c.half_x || bx.hi + blur < -c.half_x || by.lo - blur > c.half_y || by.hi + blur < -c.half_y;
}

// One thread per source: the same bound on the device, over the scene's uploaded (or generated) source arrays -- 120 000 sources
// cost the host 5 ms per start_ray_tracing call (more than the BOS sample image's trace), the device a few microseconds.
__global__ __launch_bounds__(256) void source_cull_kernel(photon::LensCull c, const float *__restrict__ x, const float *__restrict__ y,
                                                          const float *__restrict__ z, long long n, unsigned char *__restrict__ off) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) off[i] = source_misses_sensor(c, x[i], y[i], z[i]) ? 1 : 0;
}

extern "C" {

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
    (void)hipGetDevice(&s->device);                             // first: the failure paths below free into this device's block cache
    SceneDev &d = s->dev;
    UploadPack pack;
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
        if (hipStreamSynchronize(nullptr) != hipSuccess) return bail(4);        // complete before the scene is handed out: its launches may use any stream
    } else {
        if ((rc = upload(s, pack, lsp->x, ns, &d.sx))) return bail(rc);
        if ((rc = upload(s, pack, lsp->y, ns, &d.sy))) return bail(rc);
        if ((rc = upload(s, pack, lsp->z, ns, &d.sz))) return bail(rc);
        if ((rc = upload(s, pack, lsp->radiance, ns, &d.sradiance))) return bail(rc);
        if ((rc = upload(s, pack, lsp->diameter_index, ns, &d.sdia))) return bail(rc);
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
        if ((rc = upload(s, pack, sdp->scattering_angle, (size_t)sdp->num_angles, &d.mie_angle))) return bail(rc);
        if ((rc = upload(s, pack, sdp->scattering_irradiance, (size_t)sdp->num_angles * sdp->num_diameters, &d.mie_irr)))
            return bail(rc);
    }
    photon::LensCull source_cull;
    std::vector<float> r1(lightray_number_per_particle), r2(lightray_number_per_particle);
    photon_rand_table(lightray_number_per_particle, r1.data(), r2.data());
    // x_lens = ratio * 1.0 * pitch * r1 * cos(2 pi r2), the whole product in double, then to float (.cu:123-124): the same for
    // every source (the table is indexed by the ray's number within its source, .cu:2006), so it is evaluated HERE, once per
    // lens sample, with the function the kernels used per ray (photon_det_sincos: the same bits on host and device, which is
    // what the CPU oracle relies on) -- a double-precision sincos and six double multiplies per ray less: ray generation
    // 0.205 -> 0.184 ms per 1e7 rays (135 -> 121 M VALU instructions per launch), the volume-free PIV frame 25.9 -> 25.5 ms
    {
        std::vector<float> lx(r1.size()), ly(r1.size());
        for (size_t k = 0; k < r1.size(); k++) {
            double sn, cs;
            photon_det_sincos(2 * M_PI * r2[k], &sn, &cs);
            lx[k] = (float)(d.ratio * 1.0 * d.lens_pitch * r1[k] * cs);
            ly[k] = (float)(d.ratio * 1.0 * d.lens_pitch * r1[k] * sn);
        }
        if ((rc = upload(s, pack, lx.data(), lx.size(), &d.lens_x))) return bail(rc);
        if ((rc = upload(s, pack, ly.data(), ly.size(), &d.lens_y))) return bail(rc);
        // Which lens samples can reach the first element's aperture at all (live_lens_samples below): the rest need not be
        // launched on the volume-free path -- half of a full-aperture cone.
        std::vector<int> live = live_lens_samples(lx, ly, lsp, generated, ns, image_distance, num_elements, edp, element_center,
                                                  element_plane_parameters);
        s->live_count = (int)live.size();
        s->live_host = live;
        if (live.size() < lx.size() && (rc = upload(s, pack, live.data(), live.size(), &s->d_live))) return bail(rc);
        // ... and which SOURCES can reach the sensor at all (source_misses_sensor above): decided on the device once the sources are there
        s->live_sources_known = false;
        if (ns > 0 && lx.size() >= 2)
            source_cull = lens_cull_setup(lx, ly, image_distance, beam_wavelength, num_elements, edp, element_center,
                                          element_plane_parameters, element_system_index, cam);
    }
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
        d.slot_rays = lightray_number_per_particle;
        d.slot_map = nullptr;
        d.src_perm = nullptr;
        d.source_base = 0;
        d.doom_margin = 0.f;
        s->lens_z = (float)element_center[0][2];
        if ((rc = upload(s, pack, edp, (size_t)num_elements, &d.all_elems))) return bail(rc);
        if ((rc = upload(s, pack, centers.data(), centers.size(), &d.all_centers))) return bail(rc);
        if ((rc = upload(s, pack, planes.data(), planes.size(), &d.all_planes))) return bail(rc);
        if ((rc = upload(s, pack, element_system_index, (size_t)num_elements, &d.all_sys_index))) return bail(rc);
    }
    // The statistics counters (the march's error word among them: scene_error_word) and the work queues start at zero --
    // every march launch leaves the queues so -- and are zeroed HERE, as part of the one host-to-device copy, which is
    // complete when hipMemcpy returns.  hipMemset is not: it returns as soon as its fill kernel is queued on the null
    // stream (9 us, with the fill itself 200 ms away behind a full chip: tools/ubench/null_stream_memset.hip), and a march
    // launched on a non-blocking stream is not ordered behind the null stream -- with eight shards side by side on one device
    // the fill of one shard's queues waited for wave slots next to that shard's own march and, once in a dozen calls, ran
    // AFTER the march had started handing out groups (pieces handed out again: 3 of 60 C4 calls refused with hand-off errors, one
    // image off by 6e-5 with none).
    reserve_zeroed(pack, kCounterBytes / sizeof(unsigned long long), &s->d_counters);
    reserve_zeroed(pack, (size_t)kQueues * kQueueStride, &s->d_queue);
    if ((rc = flush_uploads(s, pack))) return bail(rc);
    s->source_cull = source_cull;                               // the device pass runs with the first volume-free launch (ensure_live_sources)
    d.cam = *cam;
    d.noise = NoiseDev{0, 0, 0.f, 0.f, 0ull};
    if (cam->x_pixel_number < 1 || cam->y_pixel_number < 1) {
        fprintf(stderr, "photon: sensor needs at least one pixel\n");
        return bail(1);
    }
    hipError_t e = pool_malloc((void **)&s->d_acc, (size_t)cam->x_pixel_number * cam->y_pixel_number * sizeof(double));
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

int photon_scene_live_rays(const photon_scene_t *s) { return s ? s->live_count : -1; }
int photon_scene_live_samples(const photon_scene_t *s, int *out, int capacity) {
    if (!s || !out || capacity < s->live_count) return -1;
    for (int k = 0; k < s->live_count; k++) out[k] = s->live_host[(size_t)k];
    return s->live_count;
}

// The bound behind photon_scene_live_sources on its own, host only (no device call): off[i] = 1 when source i cannot reach the
// sensor through lens samples (lens_x[k], lens_y[k]).  Returns 0, or 1 when the geometry is not covered (off is all zeros).
int photon_sources_missing_sensor(const float *lens_x, const float *lens_y, int n_samples, float image_distance, float beam_wavelength,
                                  int num_elements, const element_data_t *edp, const double (*element_center)[3],
                                  const double (*element_plane_parameters)[4], const int *element_system_index,
                                  const camera_design_t *cam, const float *x, const float *y, const float *z, long long n,
                                  unsigned char *off) {
    if (!lens_x || !lens_y || n_samples < 1 || !edp || !element_center || !element_plane_parameters || !element_system_index || !cam ||
        n < 0 || (n > 0 && (!x || !y || !z || !off))) return 2;
    for (long long i = 0; i < n; i++) off[i] = 0;
    const std::vector<float> lx(lens_x, lens_x + n_samples), ly(lens_y, lens_y + n_samples);
    const photon::LensCull cull = lens_cull_setup(lx, ly, image_distance, beam_wavelength, num_elements, edp, element_center,
                                                  element_plane_parameters, element_system_index, cam);
    if (!cull.ok) return 1;
    for (long long i = 0; i < n; i++) off[i] = source_misses_sensor(cull, x[i], y[i], z[i]) ? 1 : 0;
    return 0;
}

// The sources the volume-free path launches (ascending indices), or -1 when every source is (nothing could be ruled out)
long long photon_scene_live_sources(const photon_scene_t *s, int *out, long long capacity) {
    if (!s) return -2;
    if (photon::ensure_live_sources(const_cast<photon_scene_t *>(s))) return -2;
    if (!s->live_sources_known) return -1;
    const long long n = (long long)s->live_sources.size();
    if (out) {
        if (capacity < n) return -2;
        memcpy(out, s->live_sources.data(), (size_t)n * sizeof(int));
    }
    return n;
}

int photon_scene_set_skip_doomed(photon_scene_t *s, int on) {
    if (!s) return 1;
    s->skip_doomed = on != 0;
    return 0;
}

}  // extern "C"

namespace photon {

// The sources whose image can fall on the sensor (source_misses_sensor), decided ONCE per scene, with its first volume-free
// launch (a scene that only ever marches through a volume -- C3, C5, every shard of a PHOTON_DEVICES call -- never pays the
// kernel and the two small copies): flags on the device (null stream; the sources were uploaded when the scene was created),
// back to the host, compacted there (ascending: launches take slices of the list), the list up again.
int ensure_live_sources(photon_scene *s) {
    if (s->live_sources_tried) return 0;
    s->live_sources_tried = true;
    s->live_sources_known = false;
    const size_t ns = (size_t)s->dev.num_sources;
    if (!s->source_cull.ok || ns == 0) return 0;
    DeviceScope on_scene_device(s->device);
    unsigned char *d_off = nullptr;
    PH_CHECK(pool_malloc((void **)&d_off, ns));
    hipLaunchKernelGGL(source_cull_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, nullptr, s->source_cull, s->dev.sx, s->dev.sy, s->dev.sz,
                       (long long)ns, d_off);
    std::vector<unsigned char> off(ns);
    hipError_t he = hipGetLastError();
    if (he == hipSuccess) he = hipMemcpy(off.data(), d_off, ns, hipMemcpyDeviceToHost);
    pool_free(d_off);
    PH_CHECK(he);
    std::vector<int> keep;
    keep.reserve(ns);
    for (size_t i = 0; i < ns; i++)
        if (!off[i]) keep.push_back((int)i);
    if (keep.size() == ns) return 0;
    if (keep.empty()) keep.push_back(0);                                // a launch of zero rays is nobody's friend
    int *d_keep = nullptr;
    PH_CHECK(pool_malloc((void **)&d_keep, keep.size() * sizeof(int)));
    s->allocs.push_back(d_keep);
    PH_CHECK(hipMemcpy(d_keep, keep.data(), keep.size() * sizeof(int), hipMemcpyHostToDevice));
    s->d_live_sources = d_keep;
    s->live_sources = std::move(keep);
    s->live_sources_known = true;
    return 0;
}

void scene_quiesce(photon_scene *s) {
    if (!s->launched) return;
    DeviceScope on_scene_device(s->device);                     // hipDeviceSynchronize waits for the CURRENT device
    (void)hipDeviceSynchronize();
    s->launched = false;
}

void free_resume_state(photon_scene *s) {
    if (s->ws.ctr) { pool_free(s->ws.ctr); s->ws.ctr = nullptr; }
    if (s->ws.vprev) { pool_free(s->ws.vprev); s->ws.vprev = nullptr; }
    s->ws.spins = nullptr; s->ws.seg_flag = nullptr;
}

int ensure_workspace(photon_scene *s, size_t rays) {
    if (s->ws_rays >= rays) return 0;
    scene_quiesce(s);                                           // a smaller launch of this scene may still be using the old blocks
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

}  // namespace photon
