// photon_trace.hip - the launch loop of a trace (the reference's chunk loop, parallel_ray_tracing.cu:3505-3558, on
// inputs resident in HBM): ray order and doomed-ray rules per launch, raygen -> march -> sensor stage, photon_trace and
// the statistics window.  Host code only: every kernel is launched through the unit that defines it.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "photon_internal.hpp"

using namespace photon;

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
        if (slot->d_perm) { scene_quiesce(s); pool_free(slot->d_perm); slot->d_perm = nullptr; }     // an earlier launch may still read it
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
// the reference's element path (optical_system without the working train) applies element 0 once per single-member group of
// the sequence: is there one, and is element 0 a lens with an aperture test?
static bool first_aperture_applies(const photon_scene *s) {
    if (s->dev.train_mode != 0) return false;
    const char type = s->dev.elems[0].element_type;
    if (type != 'l' && type != 't') return false;
    bool applied = false;
    const int n = std::min(s->dev.num_elements, kMaxElements);
    int seq = 0;
    for (int k = 0; k < n; k++) seq = std::max(seq, s->dev.sys_index[k]);
    for (int idx = 0; idx < seq && !applied; idx++) {
        int count = 0;
        for (int k = 0; k < n; k++) count += (seq - s->dev.sys_index[k] == idx);
        applied = count == 1;
    }
    return applied;
}

static float doom_margin(const photon_scene *s, const photon_volume *vol, int algorithm, const DumpDev &dump) {
    if (!s->skip_doomed || !vol || (algorithm != 1 && algorithm != 2) || dump.final_pos || dump.inter_pos) return 0.f;
    if (s->dev.noise.add_ngrad || !first_aperture_applies(s)) return 0.f;
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

namespace photon {

int begin_accumulate(photon_scene *s, hipStream_t stream) {
    const size_t npix = (size_t)s->dev.cam.x_pixel_number * s->dev.cam.y_pixel_number;
    s->launched = true;                                     // the fill and, later, the finalize kernel use d_acc even when no source is traced
    if (!s->acc_clean) PH_CHECK(hipMemsetAsync(s->d_acc, 0, npix * sizeof(double), stream));      // else: left zeroed by the last finalize
    s->acc_clean = false;
    return 0;
}

// Without a volume only the lens samples that can reach the first aperture are launched (photon_scene.hip, live_lens_samples):
// the dead ones would be generated, meet the element's front surface and be dropped -- half of a full-aperture PIV cone.
static bool launches_live_samples_only(const photon_scene *s, const photon_volume *vol, const DumpDev &dump) {
    return !vol && s->skip_doomed && s->d_live && s->live_count < s->dev.rays_per_source && !dump.final_pos && !dump.inter_pos &&
           first_aperture_applies(s);
}
// ... and only the sources whose image can fall on the sensor (photon_scene.hip, source_misses_sensor): same conditions, no
// sensor-position noise (unbounded), the scene's source list as it was created
static bool launches_live_sources_only(photon_scene *s, const photon_volume *vol, const DumpDev &dump) {
    if (vol || !s->skip_doomed || dump.final_pos || dump.inter_pos || s->dev.noise.add_pos || !first_aperture_applies(s)) return false;
    if (ensure_live_sources(s)) return false;               // the scene's first volume-free launch decides the list (a failure: everything is launched)
    return s->live_sources_known;
}

// PHOTON_RAYGEN=kernel|fold (read once): where the rays of a launch through a volume are generated
static bool raygen_folded() {
    static const bool fold = [] { const char *e = getenv("PHOTON_RAYGEN"); return !(e && strcmp(e, "kernel") == 0); }();
    return fold;
}

int launch_chunk(photon_scene *s, const photon_volume *vol, int algorithm, long long src_begin,
                        long long src_end, DumpDev dump, hipStream_t stream, hipEvent_t ev_march_begin, hipEvent_t ev_march_end) {
    const bool live_only = launches_live_samples_only(s, vol, dump);
    s->dev.slot_rays = live_only ? s->live_count : s->dev.rays_per_source;
    s->dev.slot_map = live_only ? s->d_live : nullptr;
    s->dev.src_list = nullptr;
    long long n_sources = src_end - src_begin;
    if (launches_live_sources_only(s, vol, dump)) {
        const auto lo = std::lower_bound(s->live_sources.begin(), s->live_sources.end(), (int)src_begin);
        const auto hi = std::lower_bound(lo, s->live_sources.end(), (int)src_end);
        n_sources = hi - lo;
        s->dev.src_list = s->d_live_sources + (lo - s->live_sources.begin());
    }
    const unsigned long long n64 = (unsigned long long)n_sources * (unsigned)s->dev.slot_rays;
    if (n64 == 0) return 0;
    if (n64 > kMaxRaysPerLaunch) {
        fprintf(stderr, "photon: a launch of %llu rays (sources [%lld, %lld) x %d) exceeds the %u-ray limit per launch\n", n64,
                src_begin, src_end, s->dev.rays_per_source, kMaxRaysPerLaunch);
        return 1;
    }
    const unsigned n = (unsigned)n64;
    s->dev.doom_margin = doom_margin(s, vol, algorithm, dump);
    s->dev.ray_order = 0;
    s->dev.src_perm = nullptr;
    if (use_lens_major(s, vol, dump)) {
        const int *perm = nullptr;
        const int rc = ensure_source_order(s, src_begin, src_end, stream, &perm);
        if (rc) return rc;
        s->launched = true;                                     // the sort's kernels
        s->dev.ray_order = 1;
        s->dev.src_perm = perm;
    }
    if (vol) {
        int rc = ensure_workspace(s, n);
        if (rc) return rc;
        s->launched = true;                                     // from here on kernels of this scene may be in flight (scene_quiesce)
        // ray generation: a kernel of its own (PHOTON_RAYGEN=kernel), or the prologue of the march's first piece (fold)
        const bool fold = raygen_folded() && (algorithm == 1 || algorithm == 2);
        if (!fold) {
            rc = launch_raygen(s, src_begin, n, stream);
            if (rc) return rc;
        }
        const int interp = vol->dev.interpolation;
        const unsigned long long ray_base = (unsigned long long)(s->dev.source_base + src_begin) * (unsigned)s->dev.rays_per_source;
        const InterDump idump{dump.inter_pos, dump.inter_dir, dump.inter_slots, dump.num_save, 0u};
        const bool save = dump.inter_pos != nullptr && interp == 1;     // only the trilinear branches record
        rc = launch_march(s, vol, algorithm, n, ray_base, idump, save, stream, ev_march_begin, fold ? src_begin : -1);
        if (rc) return rc;
        if (ev_march_end) PH_CHECK(hipEventRecord(ev_march_end, stream));
        return launch_sensor(s, true, src_begin, n, dump, stream);
    }
    s->launched = true;
    return launch_sensor(s, false, src_begin, n, dump, stream);
}

}  // namespace photon

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

namespace photon {

// The launch loop for sources [src_begin, src_end) into the scene's private f64 accumulator (zeroed first); the
// caller folds the accumulator into an image (end_accumulate) -- or, when several devices share one call, sums the
// accumulators first.  timed: 0 no events; 1 immediate (the march of every launch is timed with ev[1], ev[2] and the host
// waits for it: photon_trace with a stats pointer); 2 deferred (events of the open statistics window, no host wait).
int trace_accumulate(photon_scene *scene, const photon_volume *vol, int ray_tracing_algorithm, long long src_begin,
                            long long src_end, hipStream_t stream, int timed, float *march_ms_out) {
    const unsigned rps = (unsigned)scene->dev.rays_per_source;
    if (rps > kMaxRaysPerLaunch) { fprintf(stderr, "photon: too many rays per source\n"); return 1; }
    float march_ms = 0.f;
    const DumpDev no_dump{nullptr, nullptr, 0, nullptr, nullptr, 0};
    // a launch holds at most kMaxRaysPerLaunch rays: of those it really launches (the volume-free path leaves out dead lens samples
    // and sources that miss the sensor -- the sample PIV frame's 5e8 rays go in two launches, not eight)
    const unsigned slot_rays = launches_live_samples_only(scene, vol, no_dump) ? (unsigned)scene->live_count : rps;
    const long long max_sources = std::max<long long>(1, kMaxRaysPerLaunch / slot_rays);
    const bool listed = launches_live_sources_only(scene, vol, no_dump);
    { const int rc = begin_accumulate(scene, stream); if (rc) return rc; }
    for (long long b = src_begin, e = src_begin; b < src_end; b = e) {
        e = std::min<long long>(src_end, b + max_sources);
        if (listed) {                                           // up to max_sources LISTED sources: the range ends before the next one
            const auto &ls = scene->live_sources;
            const auto lo = std::lower_bound(ls.begin(), ls.end(), (int)b);
            e = (ls.end() - lo) > max_sources ? (long long)lo[max_sources] : src_end;
            e = std::min<long long>(e, src_end);
            if (e <= b) e = src_end;                            // (cannot happen: lo[max_sources] > *lo >= b)
        }
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

}  // namespace photon

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
        photon::DeviceScope on_scene_device(scene->device);
        hipStream_t stream = (hipStream_t)stream_p;
        const unsigned rps = (unsigned)scene->dev.rays_per_source;
        size_t w0 = 0, w1 = 0;
        if (stats) {
            PH_CHECK(hipMemsetAsync(scene->d_counters, 0, kCounterBytes, stream));
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
        { const int rc = launch_finalize(scene, d_image, stream); if (rc) return rc; }
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
        photon::DeviceScope on_scene_device(scene->device);
        hipStream_t stream = (hipStream_t)stream_p;
        PH_CHECK(hipMemsetAsync(scene->d_counters, 0, kCounterBytes, stream));
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

extern "C" int photon_scene_check(photon_scene_t *scene, void *stream_p) {
    if (!scene) return 1;
    PH_CHECK(hipStreamSynchronize((hipStream_t)stream_p));
    return march_error_check(scene);
}

extern "C" int photon_scene_stats_end(photon_scene_t *scene, void *stream_p, photon_trace_stats_t *stats) {
    if (!scene || !stats || !scene->win_open) {
        fprintf(stderr, "photon: photon_scene_stats_end: no open statistics window\n");
        return 1;
    }
    return guarded("photon_scene_stats_end", [&]() -> int {
        photon::DeviceScope on_scene_device(scene->device);
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
