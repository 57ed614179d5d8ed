// march_kernel.hpp - the march kernels (stage 1b): persistent waves over per-XCD work queues, the segmented
// breadth-first march and its hand-off between waves, and the plain one-thread-per-ray grid of photon_trace_volume_rays.
// Included by the translation units that instantiate them (photon_march_linear.hip, photon_march_cubic.hip); the host
// side of a launch (planner, queues, segments) is photon_march.hip.
#pragma once
#include "device_volume_coop.hpp"
#include "march_args.hpp"

using namespace photon;

// Shader-clock stamp of a wave: s_memtime ticks at the shader clock, s_memrealtime at a constant 100 MHz
// (MI355X_MICROARCH.md, "DVFS give-back" item 6).  The chip lowers its clock under load, by an amount that differs from
// device to device; the ratio of the two deltas, summed over the waves of a launch, is the clock the march actually ran
// at -- what bench.py normalises its roofline fraction with.  Two stamps per wave (a wave lives ~2 ms): no cost.
__device__ __forceinline__ void clock_stamp(unsigned long long &clk, unsigned long long &real) {
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk), "=s"(real) : : "memory");
}

// PERSISTENT WAVES (round 3).  Every ray of a BOS launch marches for the same ~1.8 ms, so the waves of a conventional
// launch finish generation by generation, and each time the dispatcher has a whole chip's worth of workgroups to start at
// once: measured on C3, the resident-wave slots stood empty 7.5 % of the kernel's time (156 250 waves x 1.76 ms mean
// lifetime / 5120 slots = 53.9 ms of work in a 58.2 ms kernel; the same 7.7 % on the 1.2 s C4 launch, and a launch of G
// generations lasted about G + 0.9 lifetimes -- one GPU's eighth of C3, 3.8 generations, ran at 81 % occupancy).
// Here the grid is just large enough to fill the chip ONCE and every wave takes 64-ray groups from a queue until the
// launch is served: a wave that finishes a group loads the next one itself, no slot waits for the dispatcher.
// 32 queues, four per XCD (workgroup i runs on XCD i % 8), each handing out the groups of its chunks (16 or 128 consecutive groups: kChunkShift*) in order, so
// rays that walk the same voxels still meet in one L2; the visiting order and
// why four are at the loop.  Every wave leaves as soon as its eleven queues are past their ends.
template <class T>
__device__ __forceinline__ T load_arg(const __attribute__((address_space(4))) T *p) {      // scalar loads from the argument segment
    T out;
    __builtin_memcpy(&out, p, sizeof(T));
    return out;
}
__device__ __forceinline__ MarchArgsPtr march_args() {
    MarchArgsPtr p = (MarchArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                                 // a fresh pointer each time: loads through it are neither hoisted nor kept
    return p;
}

__device__ __forceinline__ unsigned long long real_time() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
__device__ __forceinline__ unsigned long long *profile_slot() {
    unsigned long long *p = march_args()->profile;
    return p ? p + (size_t)(blockIdx.x % kProfileSub) * PF_N : nullptr;
}

// What a march wave accumulates over the groups it serves (wave-uniform: SGPRs) and adds to the counters once, at its end.
struct WaveTotals {
    WaveCount mc{0u, 0u};
    unsigned n_marched = 0;                                     // rays that entered the march (not skipped as doomed)
    unsigned groups = 0;                                        // groups served
    unsigned long long clk_sum = 0, real_sum = 0;               // shader-clock / 100 MHz ticks spent in groups
};

// Agent-scope relaxed accesses (global_load / global_store ... sc1): the loads bypass this CU's L1, the stores write
// through the XCD's L2 -- how the ray state travels from the wave that marched one segment of a group to the wave, on any
// CU of any XCD, that marches the next (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 payload, the storing wave's
// own vmcnt(0), an sc1 flag; the reader polls the flag with an sc1 load, then loads the payload with sc1 loads only).
template <class T> __device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static_assert(kLoopMax < (1 << 24) - 1, "completed iterations travel in 24 bits of RayStateDev::ctr");
constexpr int kSegPollMax = 1 << 20;                            // polls (~2 us each) before a wave gives a segment up: the exit every wave reaches
constexpr unsigned kSegDone = 0xffu;                            // seg_flag: every ray of the group has left the volume
constexpr unsigned kSegPoison = 0xfeu;                          // seg_flag: a wave gave a segment of this group up (counted in MarchArgs::error)

// Ray generation inside the march (MarchArgs::gen): out of line, so that the march loop's register allocation sees a call
// with six values coming back and nothing of the f64 generation code or the scene description.  The scene is read through
// a generic pointer into the argument segment (global loads, cached; once per ray).
__device__ __attribute__((noinline)) RayPD raygen_in_march(const SceneDev *sc, long long src_begin, unsigned n_rays, unsigned r,
                                                           double *radiance_out) {
    double radiance;
    const RayPD g = generate_state(*sc, src_begin, n_rays, r, radiance);
    radiance_out[r] = radiance;
    return g;
}

// One item of the launch: segment `seg` of 64-ray group `group` -- load the state, march, store it back.  Must be called
// by all 64 lanes of the wave.  With MarchArgs::segments == 1 (seg = 0) this is the whole march of the group.
//
// SEGMENTS (round 4).  A group marches for ~1.9 ms whatever the launch, and a launch ends when its LAST group does: the
// waves finish one by one over the final ~0.8 group times while the rest of the chip idles (measured with the wave-timing
// profile: span - mean end; 1.4 ms of a 8.8 ms launch of one GPU's eighth of the headline job, the same 1.4 ms of the full
// job's 59).  Cutting every march into S segments handed out breadth-first (all first segments, then all second ones, ...)
// makes the quantum smaller and the drain with it (how many pieces, how long: plan_segments); the state a ray carries between segments is the loops' own
// (MarchResume), so the bits do not change.  A segment's wave may have to wait for the wave still marching the previous
// one (only when a launch has fewer groups than the chip holds waves: the host does not segment those): it polls the
// group's flag, bounded -- a wave that gives up counts itself in MarchArgs::error and leaves (march_error_check).
// SEG: this instantiation handles segmented launches (MarchArgs::segments > 1); the whole-march instantiations carry none of
// the resume code -- the trilinear RK4 kernel, which sits on its 96-register budget, spilled 15 VGPRs into its loop with it.
template <int ALGO, int INTERP, bool SAVE, bool NOISE, bool SEG>
__device__ __forceinline__ void march_group(unsigned group, unsigned seg, unsigned n_rays, f4 *tile, WaveTotals &tot) {
    unsigned long long clk0, real0, clk1, real1;
    clock_stamp(clk0, real0);
    const unsigned lane = threadIdx.x & 63u;
    if (tot.groups++ == 0) {                                    // wave-uniform: this wave's first group
        unsigned long long *pf = profile_slot();
        if (pf && lane == 0) {
            atomicMax(&pf[PF_START_NEGMIN], ~real0); atomicAdd(&pf[PF_START_SUM], real0); atomicMax(&pf[PF_START_MAX], real0);
        }
    }
    const unsigned r = group * 64u + lane;
    const bool has_ray = r < n_rays;
    f3 p = mk3(0, 0, 0), d = mk3(0, 0, -1);
    bool marching = has_ray;
    MarchArgsPtr a = march_args();
    MarchResume rs = resume_fresh();
    {
        const bool fresh = !SEG || seg == 0;                    // wave-uniform
        const RayStateDev st = load_arg(&a->st);
        if (!fresh) {
            // the previous segment of this group: handed out before this one, to a wave that is running -- normally long done
            const unsigned want = a->epoch;
            unsigned flag = 0;
            int polls = 0;
            while (true) {
                unsigned f = 0;
                if (lane == 0) f = ld_agent(&st.seg_flag[group]);
                flag = (unsigned)__builtin_amdgcn_readfirstlane((int)f);
                if ((flag >> 8) == want && (flag & 0xffu) >= seg) break;
                if (++polls > kSegPollMax) {                    // never seen; an exit every wave reaches
                    // poison the group: the waves that take its later segments then return at once instead of polling
                    // for the full bound each (64 segments x ~2 s would stall the launch for minutes before the host sees
                    // the error count)
                    if (lane == 0) { atomicAdd(a->error, 1u); st_agent(&st.seg_flag[group], (want << 8) | kSegPoison); }
                    return;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if ((flag & 0xffu) == kSegDone) return;             // wave-uniform: no ray of this group is still in the volume
            if ((flag & 0xffu) == kSegPoison) {                 // an earlier segment of this group was given up: so is this one
                if (lane == 0) atomicAdd(a->error, 1u);
                return;
            }
            // No agent-scope acquire here: every load of the handed-off words below is an sc1 load (bypasses this CU's L1),
            // every one of them was stored sc1 and drained before the flag, and the flag itself was polled sc1 -- the guide's
            // conditions for leaving the buffer_inv out, checked for exactly this pattern (lines shared between groups, five
            // workgroups per CU, uneven arrivals) by tools/ubench/xcd_handoff.hip: 0 stale words of 7.4e7 with or without it.
            // An acquire per segment start invalidates the L1 under the CU's nineteen other waves' texel blocks.
        }
        const bool gen = fresh && a->gen != 0u;                 // wave-uniform: this wave generates the group's rays itself
        if (has_ray && gen) {
            const RayPD g = raygen_in_march((const SceneDev *)&a->scene, a->src_begin, n_rays, r, st.radiance);
            p = g.p; d = g.d;
            marching = !isnan3(p);
            if (!marching) { st.px[r] = p.x; st.py[r] = p.y; st.pz[r] = p.z; st.dx[r] = d.x; st.dy[r] = d.y; st.dz[r] = d.z; }     // what the sensor stage reads of a ray that is not marched
        } else if (has_ray) {
            // segmented launches: sc1 loads (the words may have been stored by a wave on another XCD a moment ago); whole
            // marches read what raygen_kernel wrote before this kernel started: plain loads
            p = SEG ? mk3(ld_agent(&st.px[r]), ld_agent(&st.py[r]), ld_agent(&st.pz[r])) : mk3(st.px[r], st.py[r], st.pz[r]);
            d = SEG ? mk3(ld_agent(&st.dx[r]), ld_agent(&st.dy[r]), ld_agent(&st.dz[r])) : mk3(st.dx[r], st.dy[r], st.dz[r]);
            if (fresh) {
                marching = !isnan3(p);                          // rays marked dead by raygen_kernel stay out of the march
            } else {
                const unsigned c = ld_agent(&st.ctr[r]);
                marching = (c >> 31) != 0u;
                rs.loop_ctr = (int)(c & 0xffffffu);
                // a ray still marching was written by the previous segment: its word says so (bits 24-30).  Anything else is
                // a STALE word -- the hand-off broken -- and the render must not be returned (march_error_check)
                if (marching && ((c >> 24) & 0x7fu) != ((seg - 1u) & 0x7fu)) atomicAdd(a->error, 1u);
                rs.spins = (int)ld_agent(&st.spins[r]);
                if (INTERP == 1) {
                    const size_t n = st.stride;
                    rs.val_prev = f4{ld_agent(&st.vprev[r]), ld_agent(&st.vprev[n + r]), ld_agent(&st.vprev[2 * n + r]), ld_agent(&st.vprev[3 * n + r])};
                }
            }
        }
        if (SEG) {
            const unsigned b0 = a->seg_begin[seg];              // wave-uniform index: scalar loads from the argument segment
            if (!fresh) { rs.fresh = false; rs.trips_base = b0; }
            if (seg + 1u < a->segments) rs.max_trips = a->seg_begin[seg + 1u] - b0;
        }
        tot.n_marched += fresh ? (unsigned)__popcll(ballot(marching)) : 0u;
    }
    const VolumeDev vol = load_arg(&a->vol);
    const f4 *tex = a->tex;
    GradNoise gn{0, 0.f, 0ull, 0ull};
    if (NOISE) { const NoiseDev nz = load_arg(&a->noise); gn = GradNoise{nz.add_ngrad, nz.ngrad_std, nz.seed, a->ray_base + r}; }
    InterDump idump{nullptr, nullptr, 0, 0, 0u};
    if (SAVE) idump = load_arg(&a->idump);
    idump.ray = r;                                              // chunk-global ray id, like the final dumps
    const unsigned long long marching_mask = ballot(marching);  // scalar from here on
    unsigned long long still;                                   // lanes whose rays are still in the volume when the segment ends
    if (INTERP == 1 && vol.weight_scale > 0.f)                  // kernel-uniform: the texture unit's 8-bit weights (the default) / exact f32
        still = trace_volume_coop<ALGO, INTERP, SAVE, NOISE, true, WaveCount>(marching, p, d, vol, tex, tile, tot.mc, gn, idump, rs);   // all 64 lanes
    else
        still = trace_volume_coop<ALGO, INTERP, SAVE, NOISE, false, WaveCount>(marching, p, d, vol, tex, tile, tot.mc, gn, idump, rs);
    {
        MarchArgsPtr b = march_args();
        const RayStateDev st = load_arg(&b->st);                // loaded again: not carried through the march in SGPRs
        const bool fresh = !SEG || seg == 0, last = !SEG || seg + 1u >= b->segments;      // likewise
        // the ray's index and predicates are formed again from the (scalar) group number rather than carried through the march
        // loop in vector registers the loop has no room for
        unsigned g2 = group;
        asm volatile("" : "+s"(g2));
        const unsigned r = g2 * 64u + (threadIdx.x & 63u);
        const bool has_ray = r < b->n_rays;
        const bool marching = lane_of(marching_mask);
        if (last) {
            if (marching) {
                st.px[r] = p.x; st.py[r] = p.y; st.pz[r] = p.z;
                st.dx[r] = d.x; st.dy[r] = d.y; st.dz[r] = d.z;
            }
        } else {
            if (marching) {                                     // where the rays that marched this segment stand now
                st_agent(&st.px[r], p.x); st_agent(&st.py[r], p.y); st_agent(&st.pz[r], p.z);
                st_agent(&st.dx[r], d.x); st_agent(&st.dy[r], d.y); st_agent(&st.dz[r], d.z);
            }
            if (fresh ? has_ray : marching) {
                st_agent(&st.ctr[r], (lane_of(still) ? 0x80000000u : 0u) | ((seg & 0x7fu) << 24) | ((unsigned)rs.loop_ctr & 0xffffffu));
                st_agent(&st.spins[r], (unsigned)rs.spins);
                if (INTERP == 1) {
                    const size_t n = st.stride;
                    st_agent(&st.vprev[r], rs.val_prev.x); st_agent(&st.vprev[n + r], rs.val_prev.y);
                    st_agent(&st.vprev[2 * n + r], rs.val_prev.z); st_agent(&st.vprev[3 * n + r], rs.val_prev.w);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's stores have reached memory ...
            if (lane == 0) st_agent(&st.seg_flag[group], (b->epoch << 8) | (still != 0 ? seg + 1u : kSegDone));      // ... before its flag
        }
    }
    clock_stamp(clk1, real1);
    tot.clk_sum += clk1 - clk0; tot.real_sum += real1 - real0;
}

template <int ALGO, int INTERP, bool SAVE, bool NOISE, bool SEG>
__global__ __launch_bounds__(PHOTON_MARCH_BLOCK, (march_waves<ALGO, INTERP, NOISE>())) void march_kernel(MarchArgs) {
    __shared__ f4 tiles[PHOTON_MARCH_BLOCK / 64][wave_lds_texels<INTERP>()];           // per wave: tile + brick, rows padded (device_volume_coop.hpp)
    f4 *const tile = tiles[threadIdx.x >> 6];
    const unsigned lane = threadIdx.x & 63u;
    WaveTotals tot;
    {
        unsigned long long *pf = profile_slot();
        if (pf && lane == 0) atomicMax(&pf[PF_ENTER_NEGMIN], ~real_time());
    }
    // 32 queues: XCD x (workgroup i runs on XCD i % 8) owns the chunks c (16 or 128 consecutive groups) with c % 8 == x, dealt over its four
    // sub-queues by (c / 8) % 4; one counter per queue, a cache line apart.  A wave serves its home sub-queue, then the
    // other three of its XCD (between them the XCD's waves drain all four: every group is taken), then the same sub-queue
    // of the seven other XCDs (balance at the end of the launch).  How many sub-queues (same box, 1 / 2 / 4 / 8 per XCD):
    //   * few queues = returning atomics to few addresses: a launch whose groups are no work (every ray misses the
    //     volume: the reference's sample BOS case, ~1e6 groups) takes 9.3 / 8.2 / 7.7 / 7.7 ms per call (one-shot: 7.6);
    //   * many queues = each served by few waves, so the eight groups that carry ONE source's rays start further apart in
    //     time, walk the volume at different depths and share fewer L2 lines: HBM traffic of the C3 launch 1.9 / 2.0 /
    //     2.6 / 4.8 GB (one-shot, where a whole generation walks in lockstep: 1.3); the march time does not care (62.1 /
    //     61.9 / 62.3 / 62.1 ms).
    // Taking several groups per access instead of adding queues was tried and dropped: wherever trivial and real groups
    // mix (doomed lens samples of a PIV launch) a wave ends up holding dozens of real groups while the chip drains (C5
    // quarter 38.6 -> 92 ms).
    // A queue of Gq groups hands out Gq x S items, segment-major: item k = segment k / Gq of its (k % Gq)-th group.
    const unsigned home_x = blockIdx.x & 7u, home_s = (blockIdx.x >> 3) & (kSubQueues - 1u);
    for (unsigned step = 0; step < kSubQueues + 7u; step++) {
        const unsigned x = step < kSubQueues ? home_x : ((home_x + step - (kSubQueues - 1u)) & 7u);
        const unsigned sub = step < kSubQueues ? ((home_s + step) & (kSubQueues - 1u)) : home_s;
        while (true) {
            unsigned k = 0;
            if (lane == 0) k = atomicAdd(&march_args()->queue[(sub * 8u + x) * kQueueStride], 1u);
            k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
            const unsigned n_rays = march_args()->n_rays;
            const unsigned shift = march_args()->chunk_shift;
            const unsigned gq = march_queue_size((n_rays + 63u) / 64u, x, sub, shift);
            const unsigned n_seg = SEG ? march_args()->segments : 1u;
            if (k >= gq * n_seg) break;                         // this queue is served (k < 2^26 / 64 * 255: no overflow)
            const unsigned seg = SEG ? k / gq : 0u;
            march_group<ALGO, INTERP, SAVE, NOISE, SEG>(march_queue_group(k - seg * gq, x, sub, shift), seg, n_rays, tile, tot);
            if ((tot.mc.samples | tot.mc.iterations) >> 31) {     // wave-uniform: the 32-bit wave totals go out before they can wrap
                if (lane == 0) {
                    unsigned long long *slot = counter_slot(march_args()->counters);
                    atomicAdd(&slot[CNT_ITER], (unsigned long long)tot.mc.iterations);
                    atomicAdd(&slot[CNT_SAMPLES], (unsigned long long)tot.mc.samples);
                }
                tot.mc.iterations = tot.mc.samples = 0u;
            }
        }
    }
    if (tot.groups) {                                           // wave-uniform
        unsigned long long *pf = profile_slot();
        if (pf && lane == 0) {                                  // the wave leaves a few queue visits after its last group
            const unsigned long long t = real_time();
            atomicMax(&pf[PF_END_NEGMIN], ~t); atomicAdd(&pf[PF_END_SUM], t); atomicMax(&pf[PF_END_MAX], t);
            atomicAdd(&pf[PF_WAVES], 1ull);
        }
    }
    if (lane == 0) {
        // The launch's last wave re-arms the work queues for the next launch (no memset between launches: at one GPU's eighth
        // of the trilinear job every tiny launch is 0.2 % of the step).  A wave takes its ticket after its last queue access
        // -- whose returned value it has consumed -- so the wave that draws the last ticket finds every counter final.
        unsigned *q = march_args()->queue;
        if (atomicAdd(&q[kQueueDoneSlot * kQueueStride], 1u) == gridDim.x * (PHOTON_MARCH_BLOCK / 64u) - 1u) {
            for (unsigned k = 0; k < 8u * kSubQueues; k++) q[k * kQueueStride] = 0u;
            q[kQueueDoneSlot * kQueueStride] = 0u;
        }
        unsigned long long *slot = counter_slot(march_args()->counters);
        if (tot.mc.iterations) atomicAdd(&slot[CNT_ITER], (unsigned long long)tot.mc.iterations);
        if (tot.mc.samples) atomicAdd(&slot[CNT_SAMPLES], (unsigned long long)tot.mc.samples);
        if (tot.n_marched) atomicAdd(&slot[CNT_MARCHED], (unsigned long long)tot.n_marched);
        if (tot.real_sum) {
            atomicAdd(&slot[CNT_CLK], tot.clk_sum);
            atomicAdd(&slot[CNT_REAL], tot.real_sum);
        }
    }
}

template <int ALGO, int INTERP>
__global__ __launch_bounds__(256) void march_rays_kernel(VolumeDev v, const f4 *__restrict__ tex, int n,
                                                         float *__restrict__ pos, float *__restrict__ dir,
                                                         int *__restrict__ steps) {
    __shared__ f4 tiles[4][kWaveLdsTexels];                     // per wave: 4x4x4 tile + 8x8x4 brick, rows padded (device_volume_coop.hpp)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool has_ray = i < n;
    f3 p = mk3(0, 0, 0), d = mk3(0, 0, -1);
    if (has_ray) {
        p = mk3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        d = mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    }
    MarchCount mc{0, 0};
    const GradNoise no_noise{0, 0.f, 0ull, 0ull};
    const InterDump no_dump{nullptr, nullptr, 0, 0, 0u};
    MarchResume rs = resume_fresh();
    if (INTERP == 1 && v.weight_scale > 0.f)                    // kernel-uniform: texture-unit weights
        trace_volume_coop<ALGO, INTERP, false, false, true, MarchCount>(has_ray, p, d, v, tex, tiles[threadIdx.x >> 6], mc, no_noise, no_dump, rs);
    else
        trace_volume_coop<ALGO, INTERP, false, false, false, MarchCount>(has_ray, p, d, v, tex, tiles[threadIdx.x >> 6], mc, no_noise, no_dump, rs);
    if (has_ray) {
        pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z;
        dir[3 * i] = d.x; dir[3 * i + 1] = d.y; dir[3 * i + 2] = d.z;
        if (steps) steps[i] = mc.iterations;
    }
}
