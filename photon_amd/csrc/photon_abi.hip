// photon_abi.hip - the reference's entry point: start_ray_tracing (cuda_codes/parallel_ray_tracing.cu:3078-3775) on
// host arrays, as photon's unmodified Python calls it through ctypes (perform_ray_tracing_03.py:1888-1938), and
// PHOTON_DEVICES: the sources of ONE call sharded over several devices, their accumulators summed by one kernel.
#include <chrono>
#include <cstring>
#include <fstream>
#include <map>
#include <condition_variable>
#include <thread>

#include "photon_internal.hpp"

using namespace photon;

namespace {

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


// ---------------------------------------------------------------------------------------------
// PHOTON_DEVICES: one call, several devices
// ---------------------------------------------------------------------------------------------
// The sum of the per-device f64 accumulators, on the first device, by ONE kernel: every thread reads its pixel of up to
// kGatherPeers other accumulators THROUGH THEIR PEER-MAPPED POINTERS (xGMI is point to point: the seven links of the
// first device are read concurrently, 8 MiB each for a 1024^2 sensor) and adds them in device-list order -- f64 end to
// end, one rounding per pixel, the same bits whatever the number of devices -- and, in the last launch of a call, folds
// the sum into the caller's image (image_array is read-modify-write: parallel_ray_tracing.cu:3309, 3675).  No staging
// buffer, no host synchronisation per peer.  Two pixels per thread: 16-byte loads across the links.
constexpr int kGatherPeers = 15;
struct GatherArgs { const double *peer[kGatherPeers]; int n; };
__global__ __launch_bounds__(256) void gather_sum_kernel(double *__restrict__ acc, GatherArgs g, float *__restrict__ image, size_t n) {
    const size_t i = 2 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 < n) {
        double2 s = *reinterpret_cast<const double2 *>(acc + i);
#pragma unroll
        for (int k = 0; k < kGatherPeers; k++)
            if (k < g.n) { const double2 p = *reinterpret_cast<const double2 *>(g.peer[k] + i); s.x += p.x; s.y += p.y; }
        if (image) { image[i] = (float)((double)image[i] + s.x); image[i + 1] = (float)((double)image[i + 1] + s.y); }
        else *reinterpret_cast<double2 *>(acc + i) = s;
    } else if (i < n) {                                                 // odd pixel count: the last one alone
        double s = acc[i];
#pragma unroll
        for (int k = 0; k < kGatherPeers; k++) if (k < g.n) s += g.peer[k][i];
        if (image) image[i] = (float)((double)image[i] + s); else acc[i] = s;
    }
}

// A non-blocking stream per (device, worker slot), created once per process: workers that share a device (PHOTON_DEVICES
// with repeats: tests, rehearsals) then run side by side instead of queueing on the null stream.
hipStream_t worker_stream(int device, int slot) {
    static std::mutex lock;
    static std::map<std::pair<int, int>, hipStream_t> *streams = new std::map<std::pair<int, int>, hipStream_t>;
    std::lock_guard<std::mutex> g(lock);
    auto it = streams->find({device, slot});
    if (it != streams->end()) return it->second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); s = nullptr; }     // the null stream still works
    (*streams)[{device, slot}] = s;
    return s;
}

// Arguments of one start_ray_tracing call, as the multi-device path hands them to its workers.
struct CallArgs {
    float lens_pitch, image_distance;
    scattering_data_t *sdp; char *scattering_type_str; lightfield_source_t *lsp;
    int rays_per_source; float beam_wavelength, f_number; int num_elements;
    double (*element_center)[3]; element_data_t *edp; double (*element_planes)[4]; int *sys_index;
    camera_design_t *cam; bool density; char *density_path; int algorithm;
    bool add_pos_noise; float pos_noise_std; bool add_ngrad_noise; float ngrad_noise_std; float ratio;
};

// PHOTON_DEVICES (SURVEY 8e inside ONE call, for photon's single Python process).  What is distributed is the reference's
// chunk loop over light-field sources (parallel_ray_tracing.cu:3505-3558): the sources are cut into contiguous,
// count-balanced blocks, one per listed device; each device's host thread uploads ONLY its block (plus the replicated
// tables, optics and volume -- the NRRD is parsed once, SharedDensity); when every thread has done so (a rendezvous) each
// renders into its scene's private f64 accumulator on a stream of its own, while the calling thread uploads the caller's image to the first device (peer access
// from the first device to the others has been settled before: once per pair and process).  When the workers are done ONE kernel on the first device sums the
// accumulators through their peer-mapped pointers and folds the sum into the image (gather_sum_kernel).  A device the
// first one cannot map (no xGMI / PCIe peer path) has its accumulator copied into a block of the cache first
// (hipMemcpyPeerAsync, all such copies in flight together) -- said on stderr when the pair is first seen, and per call
// under PHOTON_VERBOSE.
int render_on_devices(const std::vector<int> &devices, const CallArgs &a, float *image_array) {
    const char *e = getenv("PHOTON_NOISE_SEED");
    const uint64_t seed = e ? strtoull(e, nullptr, 0) : 0x5eedULL;
    const long long n_src = a.lsp->num_particles;
    const size_t npix = (size_t)a.cam->x_pixel_number * a.cam->y_pixel_number;
    const size_t K = devices.size();
    std::vector<photon_scene *> scenes(K, nullptr);
    std::vector<int> rcs(K, 0), slot(K, 0);
    for (size_t k = 0; k < K; k++)                                      // k-th worker of its device
        for (size_t j = 0; j < k; j++) slot[k] += devices[j] == devices[k];
    SharedDensity shared;
    std::vector<std::thread> workers;
    const auto t_start = std::chrono::steady_clock::now();
    // peer access from the first device to the others, BEFORE any worker allocates its accumulator: what the sum's kernel
    // dereferences must have been allocated under the mapping (photon_pool.hpp); once per pair and process, then a table look-up
    // PHOTON_PEER_READS=0: never dereference another device's memory, always stage (for a node whose peer mappings misbehave)
    const char *pr = getenv("PHOTON_PEER_READS");
    const bool allow_direct = !(pr && strcmp(pr, "0") == 0);
    std::vector<char> direct(K, 1);
    for (size_t k = 1; k < K; k++) direct[k] = devices[k] == devices[0] || (allow_direct && peer_access(devices[0], devices[k])) ? 1 : 0;
    // Two phases with a rendezvous between them: every worker first builds its scene (uploads) and gets its volume, THEN all
    // start tracing.  With distinct devices the rendezvous costs the spread of eight equal uploads; with a device listed more
    // than once (rehearsals, tests: all eight on one) it keeps one shard's uploads -- blit kernels -- from queueing behind
    // another shard's march, whose persistent waves hold every wave slot of the device until they are done.
    struct Rendezvous {
        std::mutex m; std::condition_variable cv; size_t waiting = 0, total;
        explicit Rendezvous(size_t n) : total(n) {}
        void arrive_and_wait() {
            std::unique_lock<std::mutex> g(m);
            if (++waiting >= total) cv.notify_all();
            else cv.wait(g, [&] { return waiting >= total; });
        }
        void expect(size_t n) {                                         // fewer parties after all (a thread could not be started)
            std::lock_guard<std::mutex> g(m);
            total = n;
            if (waiting >= total) cv.notify_all();
        }
    } rendezvous(K);
    std::vector<photon_volume *> volumes(K, nullptr);
    bool all_started = true;
    for (size_t k = 0; k < K && all_started; k++) {
        try {
        workers.emplace_back([&, k]() {
            const long long b = n_src * (long long)k / (long long)K, e2 = n_src * (long long)(k + 1) / (long long)K;
            hipStream_t stream = nullptr;
            const int rc_setup = guarded("start_ray_tracing (device worker, setup)", [&]() -> int {
                if (hipSetDevice(devices[k]) != hipSuccess) return 1;
                stream = worker_stream(devices[k], slot[k]);
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
                int rc = 0;
                if (a.density) rc = cached_volume(a.density_path, interpolation_from_env(), &volumes[k], &shared);
                if (!rc && volumes[k]) photon_volume_set_weight_bits(volumes[k], weight_bits_from_env());
                return rc;
            });
            rendezvous.arrive_and_wait();                               // on every path: a worker that failed still arrives
            if (rc_setup) { rcs[k] = rc_setup; return; }
            rcs[k] = guarded("start_ray_tracing (device worker)", [&]() -> int {
                const auto tw = std::chrono::steady_clock::now();
                int rc = trace_accumulate(scenes[k], volumes[k], a.algorithm, 0, e2 - b, stream, 0, nullptr);
                if (!rc && hipStreamSynchronize(stream) != hipSuccess) rc = 4;
                if (!rc) rc = march_error_check(scenes[k]);
                if (!rc && verbose())
                    fprintf(stderr, "photon: device %d: sources [%lld, %lld) traced in %.3f ms\n", devices[k], b, e2,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count());
                return rc;
            });
        });
        } catch (const std::exception &ex) {                            // no thread: the ones already running must not wait for it
            fprintf(stderr, "photon: cannot start the worker thread of device %d (%s); image left untouched\n", devices[k], ex.what());
            rendezvous.expect(workers.size());
            all_started = false;
        }
    }
    // meanwhile, on the calling thread: the caller's image onto the first device
    int rc = all_started ? 0 : 1;
    auto check = [&](hipError_t err, int line) {
        if (err != hipSuccess && !rc) {
            fprintf(stderr, "photon: HIP error %d (%s) at %s:%d; image left untouched\n", (int)err, hipGetErrorString(err), __FILE__, line);
            rc = (int)err;
        }
        return rc == 0;
    };
    PoolBuffer<float> d_img;
    if (check(hipSetDevice(devices[0]), __LINE__) && check(d_img.alloc(npix), __LINE__))
        check(hipMemcpy(d_img.p, image_array, npix * sizeof(float), hipMemcpyHostToDevice), __LINE__);      // .cu:3309
    for (auto &w : workers) w.join();
    const auto t_traced = std::chrono::steady_clock::now();
    for (size_t k = 0; k < K && !rc; k++)
        if (rcs[k]) { fprintf(stderr, "photon: device %d failed (%d); image left untouched\n", devices[k], rcs[k]); rc = rcs[k]; }
    // ---- the sum, on the first device ----
    std::vector<PoolBuffer<double>> staged(K);
    size_t n_staged = 0;
    if (!rc && check(hipSetDevice(devices[0]), __LINE__)) {
        std::vector<const double *> peers;
        for (size_t k = 1; k < K && !rc; k++) {
            const double *other = scenes[k]->d_acc;
            if (!direct[k]) {                                           // no peer mapping: the runtime stages the copy through the host
                if (!check(staged[k].alloc(npix), __LINE__)) break;
                if (!check(hipMemcpyPeerAsync(staged[k].p, devices[0], other, devices[k], npix * sizeof(double), nullptr), __LINE__)) break;
                other = staged[k].p;
                n_staged++;
            }
            peers.push_back(other);
        }
        const dim3 grid((unsigned)((npix / 2 + 1 + 255) / 256)), block(256);
        for (size_t at = 0; !rc; at += kGatherPeers) {
            GatherArgs g{};
            g.n = (int)std::min<size_t>(kGatherPeers, peers.size() - at);
            for (int j = 0; j < g.n; j++) g.peer[j] = peers[at + j];
            const bool last = at + g.n >= peers.size();
            hipLaunchKernelGGL(gather_sum_kernel, grid, block, 0, nullptr, scenes[0]->d_acc, g, last ? d_img.p : nullptr, npix);
            if (!check(hipGetLastError(), __LINE__) || last) break;
        }
        if (!rc) check(hipMemcpy(image_array, d_img.p, npix * sizeof(float), hipMemcpyDeviceToHost), __LINE__);      // .cu:3675 (waits for the kernel)
    }
    if (verbose()) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "photon: %zu devices: shards traced in %.3f ms (uploads included); sum of %zu accumulators on device %d (%zu by direct peer reads, "
                        "%zu staged) + fold + image out: %.3f ms\n", K, std::chrono::duration<double, std::milli>(t_traced - t_start).count(), K, devices[0],
                K - 1 - n_staged, n_staged, std::chrono::duration<double, std::milli>(t_end - t_traced).count());
    }
    for (size_t k = 0; k < K; k++)
        if (scenes[k]) { (void)hipSetDevice(devices[k]); photon_scene_free(scenes[k]); }       // waits for the device first
    (void)hipSetDevice(devices[0]);
    if (rc) (void)hipDeviceSynchronize();                               // a failed call may have left copies or the sum in flight: the
                                                                        // staged blocks and the image block go back to the cache on return
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
            PH_VOID(hipMemsetAsync(d_fpos, 0xFF, nsave * sizeof(float), nullptr));    // all-ones = NaN (.cu:3527-3533); the null stream, like the chunk's launches
            PH_VOID(hipMemsetAsync(d_fdir, 0xFF, nsave * sizeof(float), nullptr));
            if (inter) {
                PH_VOID(hipMemsetAsync(d_ipos, 0xFF, ninter * sizeof(float), nullptr));
                PH_VOID(hipMemsetAsync(d_idir, 0xFF, ninter * sizeof(float), nullptr));
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
        if (rc == 0) rc = launch_finalize(scene, d_image, nullptr);
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
