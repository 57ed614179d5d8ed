// photon_internal.hpp - what the translation units of libparallel_ray_tracing.so share on the host side: error
// handling, the handle types behind include/parallel_ray_tracing.h, and the functions one unit calls in another.
//
//   photon_pool.hip          block cache, allocation helper, peer-access record
//   photon_volume.hip        NRRD parser, gradient-volume build + B-spline prefilter kernels, volume handle API, volume cache
//   photon_scene.hip         scene / source handles, on-device scene generation, the glibc rand table
//   photon_march.hip         host side of a march launch: segment planner, work queues, wave-timing profile
//   photon_march_{linear,cubic,extra}.hip   the march kernels (march_kernel.hpp), one unit per sampler
//   photon_sensor.hip        ray generation, sensor stage (lens / aperture / splats), finalize
//   photon_trace.hip         launch loop of a trace, photon_trace, statistics
//   photon_post.hip          sensor post-processing, the streaming-copy yardstick
//   photon_abi.hip           start_ray_tracing, PHOTON_DEVICES (several devices inside one call)
//   photon_sort.hip          Morton order of a range of sources
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/parallel_ray_tracing.h"
#include "device_optics.hpp"
#include "device_vec.hpp"
#include "device_volume.hpp"
#include "march_args.hpp"
#include "photon_pool.hpp"
#include "photon_sort.hpp"

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

namespace photon {

inline bool verbose() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("PHOTON_VERBOSE"); v = (e && atoi(e) > 0) ? 1 : 0; }
    return v == 1;
}

// No C++ exception may cross the C boundary (a ctypes caller would be terminated): every extern "C" body that
// allocates host memory runs inside this guard.
template <typename F>
int guarded(const char *what, F &&body) {
    try {
        return body();
    } catch (const std::exception &e) {
        fprintf(stderr, "photon: %s failed: %s\n", what, e.what());
    } catch (...) {
        fprintf(stderr, "photon: %s failed: unknown exception\n", what);
    }
    return 100;
}

}  // namespace photon

// =============================================================================================
// handles (opaque in the public header)
// =============================================================================================
struct photon_volume {
    float grad_max = 0.f;               // largest |grad n| of the texels (per micron)
    photon::VolumeDev dev{};
    photon_volume_info_t info{};
    photon::f4 *d_texels = nullptr;
    photon::f4 *d_coeffs = nullptr;
};

struct photon_sources {                 // light-field sources generated in HBM (SoA, like lightfield_source_t)
    long long n = 0;
    float *x = nullptr, *y = nullptr, *z = nullptr;
    double *radiance = nullptr;
    int *diameter_index = nullptr;
    // where the generator put them, when it can say (the PIV field's box): the largest distance from the z axis and the z range --
    // what the static skip of dead lens samples needs to know about sources it cannot read (photon_scene.hip, live_lens_samples)
    bool have_extent = false;
    double rmax = 0, zmin = 0, zmax = 0;
};

struct PermEntry { long long begin = -1, end = -1; int *d_perm = nullptr; size_t capacity = 0; unsigned long long stamp = 0; };

namespace photon {
// What the source cull of the volume-free path needs beside a source's coordinates (photon_scene.hip, source_misses_sensor)
struct LensCull {
    bool ok = false;
    bool thin = false;                  // element 't': one refraction on the element's plane (.cu:416-503); focal = its focal length
    double focal = 0;
    double za, zf, zb, z_sen, R1, R2a, n, hp, t, sag1, sag2, rp_all, half_x, half_y;
};
}  // namespace photon

struct photon_scene {
    photon::SceneDev dev{};
    std::vector<void *> allocs;         // device buffers owned by the scene
    photon::RayStateDev ws{};           // march -> sensor state, grown on demand
    size_t ws_rays = 0;
    unsigned long long *d_counters = nullptr;   // kCounterSlots x kCounterStride statistics words; the last word of slot 0 is the march's
                                        // hand-off error count (scene_error_word), so whatever zeroes the statistics zeroes it too
    unsigned *d_queue = nullptr;        // the march's work queues: room for 64 counters a cache line apart, 8 XCDs x kSubQueues (4) in use + the
                                        // ticket of the waves that have left; zeroed at creation, re-armed by every launch's last wave
    int device = 0;                     // the device the scene was created on: every entry point that launches, allocates, frees or waits
                                        // for this scene makes it current first (DeviceScope) and hands the caller's device back
    int num_cus = 256;                  // compute units of the scene's device (size of the persistent march grid)
    unsigned march_epoch = 0;           // tag of the last segmented march launch in ws.seg_flag
    int march_segments = -1;            // photon_scene_set_march_segments: -1 the library's choice, 1 whole marches, n segments
    unsigned long long *d_profile = nullptr;    // wave-timing slots of the march launches (photon_scene_set_march_profile), or nullptr
    unsigned prof_next = 0;             // march launches since the slots were last zeroed
    double *d_acc = nullptr;            // f64 sensor accumulator, W*H
    bool acc_clean = false;             // the accumulator is all zeros (finalize_image_kernel leaves it so): the next trace needs no memset
    bool launched = false;              // kernels of this scene may be in flight: its blocks go back to the cache only after a device sync
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
    const int *d_live = nullptr;        // the lens samples that can reach element 0's aperture from ANY source of this scene, ascending
    int live_count = 0;                 // (part of the upload block); == rays_per_source when none can be ruled out (or nothing is known)
    std::vector<int> live_host;         // the same list on the host (photon_scene_live_samples: tests hold the bound against exact geometry)
    const int *d_live_sources = nullptr;    // the sources whose image can fall on the sensor (photon_scene.hip, source_misses_sensor), ascending;
    std::vector<int> live_sources;      // part of the upload block, and the same list on the host; used by the volume-free path only
    bool live_sources_known = false;    // false: nothing could be ruled out (or the geometry is not covered): every source is launched
    bool live_sources_tried = false;    // the device pass has run (ensure_live_sources: with the scene's first volume-free launch)
    photon::LensCull source_cull;       // set at creation (host arithmetic only)
    PermEntry perms[4];                 // spatial (Morton) orders of the lens-major launch ranges seen last
    unsigned long long perm_clock = 0;
    photon_sort_scratch sort_scratch;   // keys / indices / radix-sort temporaries, grown on demand (photon_sort.hip)
};

namespace photon {

// Makes `device` current for the scope and restores the caller's device afterwards (no HIP call at all when it already is).
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// rays per launch: bounded so that 32-bit ray ids suffice and the state stays a few GB
constexpr unsigned kMaxRaysPerLaunch = 1u << 26;

// March waves that gave a segment up or read a stale ray state (march_group) count themselves here: the spare word of
// statistics slot 0.  Zeroed with the statistics (photon_trace with stats, photon_scene_stats_begin) and when read.
inline unsigned *scene_error_word(const photon_scene *s) { return reinterpret_cast<unsigned *>(s->d_counters + CNT_N); }
constexpr size_t kCounterBytes = (size_t)kCounterSlots * kCounterStride * sizeof(unsigned long long);

// ---- photon_volume.hip ----
// the volume of `path` on the current device, uploaded once and kept while the file does not change (one per device)
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
int cached_volume(const char *path, int interpolation, photon_volume **out, SharedDensity *shared = nullptr);

// ---- photon_scene.hip ----
// Wait for the device before blocks of this scene go back to the cache (its kernels may still be using them); no-op for a
// scene that never launched anything.
void scene_quiesce(photon_scene *s);
int ensure_live_sources(photon_scene *s);
void free_resume_state(photon_scene *s);
int ensure_workspace(photon_scene *s, size_t rays);

// ---- photon_march.hip ----
// The march launch of n rays whose state sits in the scene's workspace (stage 1b): persistent grid, work queues, segments.
// gen_src_begin >= 0: no raygen_kernel has run; the march generates the rays of sources [gen_src_begin, ...) itself (algorithms 1, 2)
int launch_march(photon_scene *s, const photon_volume *vol, int algorithm, unsigned n, unsigned long long ray_base,
                 const InterDump &idump, bool save, hipStream_t stream, hipEvent_t ev_march_begin, long long gen_src_begin = -1);
// Did any march wave give a segment up?  Reads (and clears) the scene's error word; the caller has synchronised.
int march_error_check(photon_scene *scene);
int profile_reset(photon_scene *s, hipStream_t stream);

// ---- the march kernels, one unit per sampler (photon_march_linear.hip / _cubic.hip / _extra.hip) ----
int march_launch_linear(int algorithm, bool save, bool noise, bool segmented, dim3 grid, dim3 block, hipStream_t stream, const MarchArgs &a);
int march_launch_cubic(int algorithm, bool segmented, dim3 grid, dim3 block, hipStream_t stream, const MarchArgs &a);
int march_launch_extra(int algorithm, dim3 grid, dim3 block, hipStream_t stream, const VolumeDev &vol, unsigned n_rays, const RayStateDev &st,
                       unsigned long long *counters);
// plain one-thread-per-ray grids over [n][3] position / direction arrays (photon_trace_volume_rays)
int march_rays_launch_linear(int algorithm, const VolumeDev &vol, const f4 *tex, int n, float *pos, float *dir, int *steps);
int march_rays_launch_cubic(int algorithm, const VolumeDev &vol, const f4 *tex, int n, float *pos, float *dir, int *steps);
int march_rays_launch_extra(int algorithm, const VolumeDev &vol, int n, float *pos, float *dir, int *steps);

// ---- photon_sensor.hip ----
int launch_raygen(photon_scene *s, long long src_begin, unsigned n, hipStream_t stream);
// the sensor stage of a launch of n rays: from the marched state (from_state) or generating its rays in place
int launch_sensor(photon_scene *s, bool from_state, long long src_begin, unsigned n, const DumpDev &dump, hipStream_t stream);
// image = (float)(image + accumulator): image_array is read-modify-write (parallel_ray_tracing.cu:3309, 3675)
int launch_finalize(photon_scene *s, float *d_image, hipStream_t stream);

// ---- photon_trace.hip ----
int begin_accumulate(photon_scene *s, hipStream_t stream);
int launch_chunk(photon_scene *s, const photon_volume *vol, int algorithm, long long src_begin, long long src_end, DumpDev dump,
                 hipStream_t stream, hipEvent_t ev_march_begin, hipEvent_t ev_march_end);
// The launch loop for sources [src_begin, src_end) into the scene's private f64 accumulator (zeroed first); timed: 0 no
// events, 1 immediate (host waits per launch), 2 deferred (events of the open statistics window)
int trace_accumulate(photon_scene *scene, const photon_volume *vol, int ray_tracing_algorithm, long long src_begin,
                     long long src_end, hipStream_t stream, int timed, float *march_ms_out);

}  // namespace photon
