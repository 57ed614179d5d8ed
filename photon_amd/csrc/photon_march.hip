// photon_march.hip - host side of a march launch (stage 1b): how many pieces a launch's marches are cut into and how
// long each is (plan_segments), the persistent grid and its work queues, the per-ray resume state, the wave-timing
// profile, and the march-only entry points the parity tests drive.  The kernels themselves are instantiated in
// photon_march_{linear,cubic,extra}.hip (march_kernel.hpp).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "photon_internal.hpp"

using namespace photon;

#if PHOTON_PATH_STATS
namespace photon { int march_path_stats_linear(unsigned long long out[8]); int march_path_stats_cubic(unsigned long long out[8]); }
// debug builds only: read (and clear) the sampler-path counters of device_volume_coop.hpp, summed over the march units
extern "C" int photon_debug_path_stats(unsigned long long out[8]) {
    unsigned long long a[8] = {}, b[8] = {};
    if (int rc = march_path_stats_linear(a)) return rc;
    if (int rc = march_path_stats_cubic(b)) return rc;
    for (int k = 0; k < 8; k++) out[k] = a[k] + b[k];
    return 0;
}
#endif

extern "C" {

unsigned photon_march_queue_count(void) { return 8u * kSubQueues; }
unsigned photon_march_queue_chunk(int interpolation) { return 1u << (interpolation == 2 ? kChunkShiftCubic : kChunkShiftLinear); }
static unsigned chunk_shift_of(unsigned groups_per_chunk) {    // log2 of a power of two in [1, 2^16]; 32 otherwise
    for (unsigned s = 0; s <= 16; s++) if (groups_per_chunk == (1u << s)) return s;
    return 32u;
}
unsigned photon_march_queue_group(unsigned k, unsigned xcd, unsigned sub, unsigned groups_per_chunk) {
    const unsigned shift = chunk_shift_of(groups_per_chunk);
    return xcd < 8u && sub < kSubQueues && shift < 32u ? march_queue_group(k, xcd, sub, shift) : ~0u;
}
unsigned photon_march_queue_size(unsigned n_groups, unsigned xcd, unsigned sub, unsigned groups_per_chunk) {
    const unsigned shift = chunk_shift_of(groups_per_chunk);
    return xcd < 8u && sub < kSubQueues && shift < 32u ? march_queue_size(n_groups, xcd, sub, shift) : ~0u;
}

int photon_trace_volume_rays(const photon_volume_t *vol, int ray_tracing_algorithm, int n, float *pos, float *dir,
                             int *steps) {
    if (!vol || n < 0) {
        fprintf(stderr, "photon: photon_trace_volume_rays: bad arguments\n");
        return 1;
    }
    if (n == 0) return 0;
    DeviceBuffer<float> d_p, d_d;                       // freed on every return path
    DeviceBuffer<int> d_s;
    const size_t b3 = (size_t)n * 3 * sizeof(float);
    PH_CHECK(d_p.alloc((size_t)n * 3));
    PH_CHECK(d_d.alloc((size_t)n * 3));
    PH_CHECK(d_s.alloc((size_t)n));
    PH_CHECK(hipMemcpy(d_p.p, pos, b3, hipMemcpyHostToDevice));
    PH_CHECK(hipMemcpy(d_d.p, dir, b3, hipMemcpyHostToDevice));
    const int interp = vol->dev.interpolation;
    const f4 *tex = interp == 2 ? vol->d_coeffs : vol->d_texels;
    int rc;
    if (ray_tracing_algorithm != 1 && ray_tracing_algorithm != 2) rc = march_rays_launch_extra(ray_tracing_algorithm, vol->dev, n, d_p.p, d_d.p, d_s.p);
    else if (interp == 1) rc = march_rays_launch_linear(ray_tracing_algorithm, vol->dev, tex, n, d_p.p, d_d.p, d_s.p);
    else rc = march_rays_launch_cubic(ray_tracing_algorithm, vol->dev, tex, n, d_p.p, d_d.p, d_s.p);
    if (rc) return rc;
    PH_CHECK(hipGetLastError());
    PH_CHECK(hipMemcpy(pos, d_p.p, b3, hipMemcpyDeviceToHost));
    PH_CHECK(hipMemcpy(dir, d_d.p, b3, hipMemcpyDeviceToHost));
    if (steps) PH_CHECK(hipMemcpy(steps, d_s.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"

// What a segmented march keeps per ray between segments (MarchResume) and the per-group flags; allocated with the first
// segmented launch of a workspace size.  The flags carry the launch's epoch, so they are zeroed once, here (and when the
// 24-bit epoch wraps), not per launch.
static int ensure_resume_state(photon_scene *s, bool linear, hipStream_t stream) {
    const size_t rays = s->ws_rays, groups = (rays + 63) / 64;
    if (!s->ws.ctr) {
        unsigned *u = nullptr;
        PH_CHECK(pool_malloc((void **)&u, (2 * rays + groups) * sizeof(unsigned)));
        s->ws.ctr = u; s->ws.spins = u + rays; s->ws.seg_flag = u + 2 * rays;
        PH_CHECK(hipMemsetAsync(s->ws.seg_flag, 0, groups * sizeof(unsigned), stream));
        s->march_epoch = 0;
    }
    if (linear && !s->ws.vprev) PH_CHECK(pool_malloc((void **)&s->ws.vprev, 4 * rays * sizeof(float)));
    if (++s->march_epoch >= (1u << 24)) {
        PH_CHECK(hipMemsetAsync(s->ws.seg_flag, 0, groups * sizeof(unsigned), stream));
        s->march_epoch = 1;
    }
    return 0;
}

// Segments per march of a launch large enough to be segmented: PHOTON_MARCH_SEGMENTS=<n> (1 = whole marches), or
// PHOTON_MARCH_SEGMENTS=force:<n> to segment launches of any size (tests of the hand-off between segments).
static int march_segments_default(bool *forced) {
    const char *e = getenv("PHOTON_MARCH_SEGMENTS");            // read per launch: tests switch it between calls
    if (e && !strncmp(e, "force:", 6)) { *forced = true; e += 6; }
    const int v = (e && atoi(e) > 0) ? atoi(e) : PHOTON_MARCH_SEGMENTS;
    return v > 64 ? 64 : v;
}

// Shape of the pieces of a segmented march.  Equal pieces; HALVING pieces (1/2, 1/4, ... of the depth, the last two equal):
// a third of the hand-offs for the same final piece, but every pass then runs twice as fast as the one that feeds it -- in a
// launch of few chip fills its front catches up with the pieces it depends on and waves stand polling (measured, one GPU's
// eighth of C3, 3.8 fills: 8.07-8.15 ms halving against 7.53-7.62 equal; the full job, 30.5 fills: 56.95 against 57.28; the
// front stays clear while r / 2 <= R - 2 for every round r <= R of a pass: halving from 12 fills on); TAPERED pieces: equal
// ones, the last of them halved t times (.., u, u/2, u/4, u/4 for t = 2) -- a short final pass without the long chain of
// ever faster passes.  PHOTON_MARCH_SEGMENT_SHAPE=uniform|halving|taper:<t> overrides the choice (A/B runs, tests).
enum SegShape { SEG_UNIFORM = 0, SEG_HALVING = 1, SEG_TAPER = 2 };
static SegShape segment_shape(double fills, unsigned *taper) {
    const char *e = getenv("PHOTON_MARCH_SEGMENT_SHAPE");
    *taper = 0;
    if (e && !strcmp(e, "uniform")) return SEG_UNIFORM;
    if (e && !strcmp(e, "halving")) return SEG_HALVING;
    if (e && !strncmp(e, "taper:", 6)) { *taper = (unsigned)std::max(1, std::min(atoi(e + 6), 8)); return SEG_TAPER; }
    return fills >= 12.0 ? SEG_HALVING : SEG_UNIFORM;
}

// How many pieces, and how long each: fills `begin` (begin[s] = first trip of piece s; begin[S] = depth) and returns S.
// Every hand-off costs c (flag poll, state round trip, tile refetch); the launch's drain is 0.75 of its LAST pieces.  Equal
// pieces: a launch of R chip fills of groups that march for L each costs R (S - 1) c + 0.75 L / S -- measured on C3
// (tools/segments_sweep.sh; tricubic / trilinear RK4, full job R = 30.5, one GPU's eighth R = 3.8): optima S = 4 / 2-3 and
// 12-16 / 6-8, the model's 4.0 / 2.3 and 11.3 / 6.5 with c = 2.9 us and L = 0.82 us per unit of work (one trilinear sample per
// texel of depth; x3 for RK4's three samples, x3 for the 64-tap sampler: RK4 tricubic through 256 texels = 2304 units =
// 1.9 ms).  Only the last pass's pieces need to be short: R (S - 1) c + 0.75 (last piece), minimised over S for the shape
// in use.  At most `cap` pieces; `forced` takes the cap itself (tests); the shortest piece is 4 trips.
static unsigned plan_segments(unsigned groups, unsigned slots, unsigned depth, int algorithm, int interp, unsigned cap, bool forced,
                              unsigned *begin, int *shape_out) {
    const double fills = (double)groups / (double)std::max(slots, 1u);
    unsigned taper = 0;
    const SegShape shape = segment_shape(fills, &taper);
    cap = std::max(1u, std::min(cap, kMaxSegments));
    // lengths (as fractions of the depth) of the S pieces of a shape
    auto lengths = [&](unsigned S) {
        std::vector<double> len;
        if (shape == SEG_HALVING) {
            for (unsigned k = 1; k < S; k++) len.push_back(1.0 / (double)(1ull << std::min(k, 40u)));
            len.push_back(S > 1 ? len.back() : 1.0);
        } else {
            const unsigned t = shape == SEG_TAPER ? std::min(taper, S - 1) : 0, base = S - t;
            for (unsigned k = 0; k + 1 < base; k++) len.push_back(1.0 / base);
            double u = 1.0 / base;
            for (unsigned k = 0; k < t; k++) { u *= 0.5; len.push_back(u); }
            len.push_back(u);
        }
        return len;
    };
    unsigned S = cap;
    if (!forced) {
        // Per sampler (refitted for the trilinear kernels after the sampler work of round 4: same sweep, full job 1 / 2 / 3 / 4
        // pieces 17.22 / 17.33 / 17.39 / 17.48 ms, one GPU's eighth 1 / 3 / 4 / 6 / 8 / 12 pieces 2.359 / 2.276 / 2.288 / 2.298 /
        // 2.315 / 2.376: a hand-off costs them 4.2 us -- they carry the last sampled value along -- and their launches drain
        // over 0.31 of a last piece, in 0.72 us per unit: the full job runs whole, the eighth in 3 pieces).
        const double units = (double)depth * (algorithm == 2 ? 3.0 : 1.0) * (interp == 2 ? 3.0 : 1.0);
        // (Round 5: the tricubic kernels march 10 % faster -- 0.74 us per unit --, a hand-off costs what it did: the whole C3 job
        // moves from five halving pieces to four -- measured 51.53 against 51.50-51.60 ms, 3.29 instead of 3.95 GB of HBM
        // traffic per launch; three pieces +0.4 %, two +1.0 %; one GPU's eighth stays at eleven: profiles/r05_h_pieces_traffic.txt.)
        const double L = (interp == 2 ? 0.74 : 0.72) * units, c = interp == 2 ? 2.9 : 4.2, drain = interp == 2 ? 0.75 : 0.31;
        double best_cost = drain * L;
        S = 1;
        for (unsigned k = 2; k <= cap; k++) {
            const double cost = fills * (k - 1) * c + drain * L * lengths(k).back();
            if (cost < best_cost) { best_cost = cost; S = k; }
        }
    }
    // boundaries in trips; pieces shorter than 4 trips are merged into their predecessor
    for (;; S--) {
        const std::vector<double> len = lengths(S);
        double at = 0.0;
        bool ok = true;
        begin[0] = 0;
        for (unsigned k = 0; k < S; k++) {
            at += len[k];
            begin[k + 1] = k + 1 == S ? depth : (unsigned)(at * depth + 0.5);
            if (begin[k + 1] < begin[k] + 4u) ok = false;
        }
        if (ok || S == 1) break;
    }
    if (S == 1) { begin[0] = 0; begin[1] = depth; }
    if (shape_out) *shape_out = S > 1 ? (int)shape : (int)SEG_UNIFORM;
    return S;
}

// The library's choice for a launch of n_rays through a volume of `depth` texels on a device of num_cus compute units
// (host restatement for tests and documentation; PHOTON_MARCH_SEGMENT_SHAPE is honoured, PHOTON_MARCH_SEGMENTS is not).
extern "C" int photon_march_segments_plan(unsigned n_rays, int depth, int ray_tracing_algorithm, int interpolation, int num_cus, int *halving) {
    if (depth < 1 || num_cus < 1 || (ray_tracing_algorithm != 1 && ray_tracing_algorithm != 2)) return 0;
    const unsigned groups = (n_rays + 63u) / 64u, slots = (unsigned)num_cus * 4u * march_waves_of(ray_tracing_algorithm, interpolation);
    int shape = 0;
    unsigned s = 1, begin[kMaxSegments + 1];
    if (groups >= slots + slots / 4) s = plan_segments(groups, slots, (unsigned)depth, ray_tracing_algorithm, interpolation, PHOTON_MARCH_SEGMENTS, false, begin, &shape);
    if (halving) *halving = shape == SEG_HALVING ? 1 : 0;
    return (int)s;
}

namespace photon {

// The march launch of n rays whose state sits in the scene's workspace (stage 1b): persistent grid, work queues, segments.
int launch_march(photon_scene *s, const photon_volume *vol, int algorithm, unsigned n, unsigned long long ray_base,
                        const InterDump &idump, bool save, hipStream_t stream, hipEvent_t ev_march_begin, long long gen_src_begin) {
    const dim3 block(256), grid((n + 255) / 256);
    const int interp = vol->dev.interpolation;
    const f4 *tex = interp == 2 ? vol->d_coeffs : vol->d_texels;
    // persistent waves: a grid that fills the chip once (more workgroups than fit only find empty queues and leave)
    const unsigned all_blocks = (n + PHOTON_MARCH_BLOCK - 1) / PHOTON_MARCH_BLOCK;
    const unsigned fill_blocks = (unsigned)s->num_cus * 8u * (256 / PHOTON_MARCH_BLOCK);
    const dim3 mblock(PHOTON_MARCH_BLOCK), mgrid(std::min(all_blocks, fill_blocks));
    if (ev_march_begin) PH_CHECK(hipEventRecord(ev_march_begin, stream));
    unsigned long long *profile = nullptr;                  // wave timing of this launch, while there are free slots
    if (s->d_profile && s->prof_next < kProfileLaunches && (algorithm == 1 || algorithm == 2))
        profile = s->d_profile + (size_t)(s->prof_next++) * kProfileSub * PF_N;
    // Segments: only where the launch is several times what the chip holds at once (a segment's wave then finds the
    // previous segment of its group long done) and nothing indexes a ray's iterations (dumps, gradient noise).
    unsigned segments = 1;
    MarchArgs margs{};
    if ((algorithm == 1 || algorithm == 2) && !save && !s->dev.noise.add_ngrad) {
        const unsigned groups = (n + 63u) / 64u;
        // resident march waves: five or six per SIMD (the launch bounds of the march kernels)
        const unsigned slots = (unsigned)s->num_cus * 4u * march_waves_of(algorithm, interp);
        bool forced = s->march_segments > 1;                // an explicit count segments launches of any size (tests)
        const int want = s->march_segments >= 0 ? s->march_segments : march_segments_default(&forced);
        if (want > 1 && (forced || groups >= slots + slots / 4)) {
            const unsigned depth = (unsigned)std::max(vol->dev.nx, std::max(vol->dev.ny, vol->dev.nz));
            segments = plan_segments(groups, slots, depth, algorithm, interp, (unsigned)std::min(want, 64), forced, margs.seg_begin, nullptr);
            if (segments > 1) { const int rc = ensure_resume_state(s, interp == 1, stream); if (rc) return rc; }
        }
    }
    margs.vol = vol->dev; margs.tex = tex; margs.n_rays = n; margs.st = s->ws; margs.counters = s->d_counters; margs.noise = s->dev.noise;
    margs.ray_base = ray_base; margs.idump = idump; margs.queue = s->d_queue; margs.profile = profile; margs.segments = segments;
    margs.epoch = s->march_epoch; margs.error = scene_error_word(s);
    if (gen_src_begin >= 0 && (algorithm == 1 || algorithm == 2)) { margs.gen = 1u; margs.src_begin = gen_src_begin; margs.scene = s->dev; }
    // queue chunks: small ones (tail balance) for the tricubic kernels where neighbouring groups are neighbouring SOURCES and
    // the volume is small enough for every L2 to hold what its waves touch; lens-major launches (neighbouring groups share
    // a lens tile, their rays fan out over the whole volume) and large volumes keep the L2-friendly 128 -- C5 at a
    // quarter: 11.0 GB of HBM traffic per launch with 16-group chunks against 3.8 GB with 128, 38.03 against 37.94 ms
    margs.chunk_shift = interp == 2 && s->dev.ray_order == 0 && (size_t)vol->dev.nx * vol->dev.ny * vol->dev.nz <= ((size_t)1 << 24)
                            ? kChunkShiftCubic : kChunkShiftLinear;
    int rc;
    if (algorithm != 1 && algorithm != 2) rc = march_launch_extra(algorithm, grid, block, stream, vol->dev, n, s->ws, s->d_counters);
    else if (interp == 1) rc = march_launch_linear(algorithm, save, algorithm == 1 && s->dev.noise.add_ngrad != 0, segments > 1, mgrid, mblock, stream, margs);
    else rc = march_launch_cubic(algorithm, segments > 1, mgrid, mblock, stream, margs);
    if (rc) return rc;
    PH_CHECK(hipGetLastError());
    return 0;
}

// Did any march wave give a segment up (march_group)?  Read wherever the host waits for the device anyway: with the
// statistics, and at the end of start_ray_tracing.  Never seen; a render it happened in is incomplete and is not returned.
int march_error_check(photon_scene *scene) {
    unsigned e = 0;
    PH_CHECK(hipMemcpy(&e, scene_error_word(scene), sizeof e, hipMemcpyDeviceToHost));
    if (!e) return 0;
    fprintf(stderr, "photon: %u hand-off errors between the segments of a march (a wave gave up waiting for the previous segment of its group, "
                    "or read a stale ray state): this render is not valid\n", e);
    PH_CHECK(device_zero(scene_error_word(scene), sizeof e));
    return 1;
}

// Wave timing of the march launches (off by default): the slots are zeroed where the statistics counters are, and every
// march launch after that takes the next one.
int profile_reset(photon_scene *s, hipStream_t stream) {
    s->prof_next = 0;
    if (s->d_profile) PH_CHECK(hipMemsetAsync(s->d_profile, 0, (size_t)kProfileLaunches * kProfileSub * PF_N * sizeof(unsigned long long), stream));
    return 0;
}

}  // namespace photon

// March-only entry point THROUGH the render path's march launch (persistent waves, work queues, segments) for arbitrary
// rays: what the adversarial parity tests drive (photon_trace_volume_rays runs a plain one-thread-per-ray grid instead).
extern "C" int photon_trace_volume_rays_queued(const photon_volume_t *vol, int ray_tracing_algorithm, int n, float *pos, float *dir,
                                               int segments) {
    if (!vol || !pos || !dir || n < 0 || (unsigned)n > kMaxRaysPerLaunch || segments == 0 || segments < -1 || segments > 64 ||
        (ray_tracing_algorithm != 1 && ray_tracing_algorithm != 2)) return 1;
    if (n == 0) return 0;
    return guarded("photon_trace_volume_rays_queued", [&]() -> int {
        photon_scene sc;                                        // a bare scene: only what the march launch touches
        struct Cleanup { photon_scene *s; ~Cleanup() {
            scene_quiesce(s);
            pool_free(s->ws.px); pool_free(s->ws.radiance); free_resume_state(s);
            pool_free(s->d_counters); pool_free(s->d_queue);
        } } cleanup{&sc};
        sc.march_segments = segments;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            sc.device = dev;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) sc.num_cus = cus;
        }
        PH_CHECK(pool_malloc((void **)&sc.d_counters, (size_t)kCounterSlots * kCounterStride * sizeof(unsigned long long)));
        PH_CHECK(device_zero(sc.d_counters, (size_t)kCounterSlots * kCounterStride * sizeof(unsigned long long)));
        PH_CHECK(pool_malloc((void **)&sc.d_queue, kQueues * kQueueStride * sizeof(unsigned)));
        PH_CHECK(device_zero(sc.d_queue, kQueues * kQueueStride * sizeof(unsigned)));      // zero once: every march launch leaves them zero
        { const int rc = ensure_workspace(&sc, (size_t)n); if (rc) return rc; }
        std::vector<float> soa((size_t)n * 6);
        for (int i = 0; i < n; i++)
            for (int c = 0; c < 3; c++) { soa[(size_t)c * n + i] = pos[3 * i + c]; soa[(size_t)(3 + c) * n + i] = dir[3 * i + c]; }
        float *arrays[6] = {sc.ws.px, sc.ws.py, sc.ws.pz, sc.ws.dx, sc.ws.dy, sc.ws.dz};
        for (int c = 0; c < 6; c++) PH_CHECK(hipMemcpy(arrays[c], soa.data() + (size_t)c * n, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        const InterDump no_dump{nullptr, nullptr, 0, 0, 0u};
        sc.launched = true;
        { const int rc = launch_march(&sc, vol, ray_tracing_algorithm, (unsigned)n, 0ull, no_dump, false, nullptr, nullptr); if (rc) return rc; }
        PH_CHECK(hipDeviceSynchronize());
        { const int rc = march_error_check(&sc); if (rc) return rc; }
        for (int c = 0; c < 6; c++) PH_CHECK(hipMemcpy(soa.data() + (size_t)c * n, arrays[c], (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++)
            for (int c = 0; c < 3; c++) { pos[3 * i + c] = soa[(size_t)c * n + i]; dir[3 * i + c] = soa[(size_t)(3 + c) * n + i]; }
        return 0;
    });
}

extern "C" int photon_scene_set_march_segments(photon_scene_t *scene, int segments) {
    if (!scene || segments < -1 || segments == 0 || segments > 64) return 1;
    scene->march_segments = segments;
    return 0;
}

extern "C" int photon_scene_set_march_profile(photon_scene_t *scene, int on) {
    if (!scene) return 1;
    return guarded("photon_scene_set_march_profile", [&]() -> int {
        DeviceScope on_scene_device(scene->device);
        if (on && !scene->d_profile) {
            PH_CHECK(pool_malloc((void **)&scene->d_profile, (size_t)kProfileLaunches * kProfileSub * PF_N * sizeof(unsigned long long)));
            PH_CHECK(device_zero(scene->d_profile, (size_t)kProfileLaunches * kProfileSub * PF_N * sizeof(unsigned long long)));
        } else if (!on && scene->d_profile) {
            PH_CHECK(hipDeviceSynchronize());
            pool_free(scene->d_profile);
            scene->d_profile = nullptr;
        }
        scene->prof_next = 0;
        return 0;
    });
}

extern "C" int photon_scene_march_profile(photon_scene_t *scene, photon_march_profile_t *out) {
    if (!scene || !out || out->struct_size < sizeof(photon_march_profile_t)) {
        fprintf(stderr, "photon: photon_scene_march_profile: bad arguments (set struct_size = sizeof(photon_march_profile_t))\n");
        return 1;
    }
    return guarded("photon_scene_march_profile", [&]() -> int {
        const uint32_t size = out->struct_size;
        memset(out, 0, sizeof *out);
        out->struct_size = size;
        if (!scene->d_profile || scene->prof_next == 0) return 0;
        const unsigned launches = std::min(scene->prof_next, kProfileLaunches);
        std::vector<unsigned long long> h((size_t)launches * kProfileSub * PF_N);
        PH_CHECK(hipDeviceSynchronize());
        PH_CHECK(hipMemcpy(h.data(), scene->d_profile, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double span = 0, start_mean = 0, start_max = 0, end_min = 0, end_mean = 0, waves_sum = 0;
        unsigned used = 0;
        for (unsigned l = 0; l < launches; l++) {
            unsigned long long enter_min = ~0ull, start_min = ~0ull, start_max_t = 0, end_min_t = ~0ull, end_max_t = 0, waves = 0;
            unsigned long long start_sum = 0, end_sum = 0;       // sums of absolute stamps: modulo 2^64, differences below are exact
            for (unsigned k = 0; k < kProfileSub; k++) {
                const unsigned long long *q = &h[((size_t)l * kProfileSub + k) * PF_N];
                if (q[PF_ENTER_NEGMIN]) enter_min = std::min(enter_min, ~q[PF_ENTER_NEGMIN]);
                if (!q[PF_WAVES]) continue;
                start_min = std::min(start_min, ~q[PF_START_NEGMIN]);
                start_max_t = std::max(start_max_t, q[PF_START_MAX]);
                end_min_t = std::min(end_min_t, ~q[PF_END_NEGMIN]);
                end_max_t = std::max(end_max_t, q[PF_END_MAX]);
                start_sum += q[PF_START_SUM]; end_sum += q[PF_END_SUM]; waves += q[PF_WAVES];
            }
            if (!waves) continue;
            const double tick_ms = 1e-5;                         // 100 MHz
            used++;
            waves_sum += (double)waves;
            span += (double)(end_max_t - enter_min) * tick_ms;
            start_mean += (double)(long long)(start_sum - waves * enter_min) / (double)waves * tick_ms;
            start_max += (double)(start_max_t - enter_min) * tick_ms;
            end_min += (double)(end_min_t - enter_min) * tick_ms;
            end_mean += (double)(long long)(end_sum - waves * enter_min) / (double)waves * tick_ms;
        }
        if (!used) return 0;
        out->launches = used;
        out->waves = (uint32_t)(waves_sum / used + 0.5);
        out->span_ms = (float)(span / used);
        out->start_mean_ms = (float)(start_mean / used);
        out->start_max_ms = (float)(start_max / used);
        out->end_min_ms = (float)(end_min / used);
        out->end_mean_ms = (float)(end_mean / used);
        return 0;
    });
}

// The raw wave-timing slots of one profiled launch (64 sub-slots x 8 words: PF_*; sub-slot = workgroup index % 64, so
// sub-slot & 7 is the XCD the workgroup ran on): for tools that look at the launch's end per XCD.
extern "C" int photon_scene_march_profile_raw(photon_scene_t *scene, unsigned launch, unsigned long long *out) {
    if (!scene || !out || !scene->d_profile || launch >= std::min(scene->prof_next, kProfileLaunches)) return 1;
    PH_CHECK(hipDeviceSynchronize());
    PH_CHECK(hipMemcpy(out, scene->d_profile + (size_t)launch * kProfileSub * PF_N, (size_t)kProfileSub * PF_N * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
