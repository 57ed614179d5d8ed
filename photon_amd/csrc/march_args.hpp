// march_args.hpp - what the host side of a march launch (photon_march.hip: planner, launch) and the march kernels
// (march_kernel.hpp) share: the ray state between the stages, the kernel-argument block, the work-queue arithmetic and
// the statistics counters.
#pragma once
#include <hip/hip_runtime.h>

#include "device_optics.hpp"
#include "device_vec.hpp"
#include "device_volume.hpp"

namespace photon {

struct RayStateDev {                // SoA ray state between the march and the sensor stage
    float *px, *py, *pz, *dx, *dy, *dz;
    double *radiance;
    // what a ray carries between the segments of a segmented march (device_volume_coop.hpp, MarchResume), per ray:
    unsigned *ctr;                  // bit 31: still marching; bits 24-30: the segment that wrote the word; bits 0-23: completed iterations
    unsigned *spins;
    float *vprev;                   // [4][rays]: the last value sampled (trilinear branches)
    unsigned *seg_flag;             // per 64-ray group: (launch epoch << 8) | segments completed (0xff: every ray has left)
    unsigned stride;                // rays the arrays were allocated for (distance between the four planes of vprev)
};

struct DumpDev {                    // ray dumps (save_lightrays), indexed by chunk-global ray id
    float *final_pos;               // [num_save][3] or nullptr
    float *final_dir;
    int num_save;
    float *inter_pos;               // [num_save][inter_slots][3] or nullptr (save_intermediate_ray_data)
    float *inter_dir;
    int inter_slots;
};

enum { CNT_ON_SENSOR = 0, CNT_ITER = 1, CNT_SAMPLES = 2, CNT_TAPS = 3, CNT_MARCHED = 4, CNT_CLK = 5, CNT_REAL = 6, CNT_N = 7 };
// Statistics counters are kept in kCounterSlots copies (one 64-byte line each) and summed on the host: with one
// copy every wave of a launch ends on an atomic to the SAME address, and 1.6e5 same-address device-scope atomics
// serialise into ~2 ms -- more than the rest of the sensor stage (measured).
constexpr int kCounterSlots = 1024;
constexpr int kCounterStride = 8;       // u64 per slot: CNT_N used, padded to a cache line
__device__ __forceinline__ unsigned long long *counter_slot(unsigned long long *counters) {
    return counters + (size_t)(blockIdx.x % kCounterSlots) * kCounterStride;
}
__device__ __forceinline__ void wave_add(unsigned long long *dst, unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}


#ifndef PHOTON_MARCH_BLOCK
#define PHOTON_MARCH_BLOCK 256          // threads per workgroup of the march (a multiple of 64)
#endif
// Stage 1b: march the rays through the volume, in place on the SoA state (world frame).  One lane per ray;
// the launch's ray order (source-major / lens-major, SceneDev::ray_order) decides which rays share a wave.
// Launch bound: 5 waves per SIMD for the tricubic kernels (<= 96 VGPRs; their 7.75 KiB of LDS per wave allow no more), 6
// for the RK4 trilinear ones (below).  A wave issues at most one VALU instruction per ~4 cycles, the
// SIMD one per 2, and every wave spends part of its time waiting on LDS: the more resident waves the better
// (C3 tricubic: 3 waves 100.8 ms, 4 waves 93.2 ms at the time; now 4 waves 68.8, 5 waves 67.1 ms; trilinear 28.0
// -> 25.1 ms).  What made 96 registers reachable was the out-of-line gather fallback: under the AMDGPU calling
// convention the caller's live values sit ABOVE the callee's registers, so its 81 VGPRs were part of the march
// kernels' budget until it was rewritten to need 51 (device_volume.hpp).
#ifndef PHOTON_MARCH_WAVES
#define PHOTON_MARCH_WAVES 5
#endif
#ifndef PHOTON_MARCH_WAVES_LINEAR
#define PHOTON_MARCH_WAVES_LINEAR 6     // RK4 trilinear: a sixth wave (80 VGPRs) cost 18 spilled dwords in the loop (36 in the segmented instantiation) and
#endif                                  // still won once the sampler work of round 4 had left the kernel waiting -- C3 17.19 -> 16.32 ms, C5 quarter 13.26 ->
                                        // 12.49, one GPU's eighth of C3 2.278 -> 2.270 (round 2, 197 instructions per sample: 26.6 -> 27.7 ms); with the
                                        // lean out-of-line gather (device_volume_coop.hpp) 3 / 26 spilled dwords: 16.29 and 2.21 ms.  Seven waves: 16.75
#ifndef PHOTON_MARCH_WAVES_EULER_LINEAR
#define PHOTON_MARCH_WAVES_EULER_LINEAR 7   // Euler trilinear: the whole-march kernel needs 70 VGPRs (seven waves per SIMD as it is); the segmented one 83:
#endif                                      // capped at 72 it spills 17 dwords (48 B of scratch per lane) and, with the blend below the march's base priority
                                            // (device_volume_coop.hpp), still wins with the seventh wave -- C3 Euler march 5.327 -> 5.147 ms, one GPU's eighth
                                            // 0.791 -> 0.786 (round 5; six waves before that: 0.849 -> 0.809 over five)
#ifndef PHOTON_MARCH_WAVES_NOISE
#define PHOTON_MARCH_WAVES_NOISE 3      // the gradient-noise instantiations (Philox + Box-Muller in f64 inside the loop) need ~130 VGPRs: at five
#endif                                  // waves per SIMD they spilled 46-70 of them into the loop (176-208 B of scratch per lane); three waves, no spill
template <int ALGO, int INTERP, bool NOISE> constexpr int march_waves() {
    return NOISE ? PHOTON_MARCH_WAVES_NOISE : INTERP == 1 ? (ALGO == 2 ? PHOTON_MARCH_WAVES_LINEAR : PHOTON_MARCH_WAVES_EULER_LINEAR) : PHOTON_MARCH_WAVES;
}
// resident march waves per SIMD of a launch (the segment planner's chip fill)
inline unsigned march_waves_of(int algorithm, int interp) {
    return interp == 1 ? (algorithm == 2 ? PHOTON_MARCH_WAVES_LINEAR : PHOTON_MARCH_WAVES_EULER_LINEAR) : PHOTON_MARCH_WAVES;
}
#ifndef PHOTON_MARCH_SEGMENTS
#define PHOTON_MARCH_SEGMENTS 32        // most segments a ray's march is cut into in launches of several chip fills (launch_march picks)
#endif
constexpr unsigned kQueueStride = 16;                           // u32 per queue counter: one 64-byte line each
#ifndef PHOTON_SUBQUEUES
#define PHOTON_SUBQUEUES 4              // work queues per XCD (a power of two, <= 8); measured 1 / 2 / 4 / 8, see march_kernel
#endif
constexpr unsigned kSubQueues = PHOTON_SUBQUEUES;
constexpr unsigned kQueues = 64;                                // room for 8 XCDs x 8 sub-queues
constexpr unsigned kQueueDoneSlot = 63;                         // the line that counts the waves that have left (march_kernel re-arms the queues itself)
// Consecutive 64-ray groups an XCD's queue owns as one CHUNK: 2^shift.  Large chunks keep the rays of neighbouring sources
// in one L2; small ones balance the XCDs' queues at the end of a launch.  Measured on C3 with the segmented march (HBM
// traffic does not care: 4.0-4.15 GB): tricubic RK4 march with chunks of 128 / 32 / 16 / 8 groups 57.70 / 57.59 / 57.53 /
// 57.55 ms (one GPU's eighth 7.60 / 7.55 / 7.53 / 7.51), trilinear RK4 19.83 / 19.89 / 19.92: 16 for the tricubic kernels in
// source-major launches through volumes of up to 256^3 texels, 128 otherwise (launch_march says why).
#ifndef PHOTON_CHUNK_SHIFT_CUBIC
#define PHOTON_CHUNK_SHIFT_CUBIC 4
#endif
constexpr unsigned kChunkShiftCubic = PHOTON_CHUNK_SHIFT_CUBIC, kChunkShiftLinear = 7;
// The k-th group handed out by sub-queue `sub` of XCD `xcd`, C = 2^shift groups per chunk: chunk ((k / C) * 4 + sub) * 8 + xcd, group k % C of it.  Grows
// with k, so the first k whose group lies past the launch ends the queue; every group belongs to exactly one (xcd, sub).
__host__ __device__ inline unsigned march_queue_group(unsigned k, unsigned xcd, unsigned sub, unsigned shift) {
    return ((((k >> shift) * kSubQueues + sub) * 8u + xcd) << shift) + (k & ((1u << shift) - 1u));
}

// The march kernel's arguments, read from the kernel-argument segment WHERE THEY ARE USED (scalar loads through a pointer
// the optimiser cannot see through) instead of being held in SGPRs from the prologue on: the persistent loop needs them
// again for every group, and ~55 argument SGPRs live across the march loop -- whose own constants, masks and tile ids take
// ~60 -- overflowed the 102 a wave has (15-55 SGPRs spilled into VGPR lanes, and VGPRs into scratch).
constexpr unsigned kMaxSegments = 64;
struct MarchArgs {
    VolumeDev vol;
    const f4 *tex;
    unsigned n_rays;
    RayStateDev st;
    unsigned long long *counters;
    NoiseDev noise;
    unsigned long long ray_base;
    InterDump idump;
    unsigned *queue;
    unsigned long long *profile;        // this launch's wave-timing slots (photon_scene_set_march_profile), or nullptr
    unsigned segments;                  // segments every ray's march is cut into (1: whole marches, the state arrays below unused)
    unsigned seg_begin[kMaxSegments + 1];   // segment s covers the trips [seg_begin[s], seg_begin[s + 1]) of the march loop (the last
                                        // one runs until every ray has left): equal, halving or tapered pieces (plan_segments)
    unsigned epoch;                     // tag of this launch in RayStateDev::seg_flag
    unsigned *error;                    // waves that gave a segment up (zero unless the hand-off between segments is broken)
    unsigned chunk_shift;               // log2 of the groups per queue chunk (launch_march)
    // Ray generation folded into the march (round 6): gen != 0 = no raygen_kernel ran; the wave that takes the FIRST piece of a
    // group generates its rays itself (generate_state: the kernel's own body) from this copy of the scene description, read
    // through the argument segment where it is used, and writes only what the sensor stage needs beside the marched state:
    // the radiance, and the NaN position of a ray that is not marched.
    unsigned gen;
    long long src_begin;
    SceneDev scene;
};
typedef const __attribute__((address_space(4))) MarchArgs *MarchArgsPtr;

// Groups of a launch of n_groups that belong to queue (xcd, sub): its items are k = 0 .. that many - 1 (march_queue_group).
__host__ __device__ inline unsigned march_queue_size(unsigned n_groups, unsigned xcd, unsigned sub, unsigned shift) {
    constexpr unsigned Q = 8u * kSubQueues;                     // chunk c belongs to queue c % Q = sub * 8 + xcd
    const unsigned q = sub * 8u + xcd, full = n_groups >> shift, rem = n_groups & ((1u << shift) - 1u);
    return ((full / Q + (full % Q > q ? 1u : 0u)) << shift) + (full % Q == q ? rem : 0u);
}

// Wave timing of a march launch (photon_scene_set_march_profile; off by default): when the first wave entered, when each
// wave started its first group and when it left, on the constant 100 MHz clock -- what tells a launch's start-up cost
// (dispatch, cold caches) from its drain (the last groups finishing one by one while the rest of the chip idles).
// kProfileSub copies per launch (a cache line each, chosen by workgroup) so that the stamps of a chip's worth of waves
// do not serialise on one address; minima are kept as maxima of the complement, so a slot starts from zeros.
enum { PF_ENTER_NEGMIN = 0, PF_START_NEGMIN, PF_START_SUM, PF_START_MAX, PF_END_NEGMIN, PF_END_SUM, PF_END_MAX, PF_WAVES, PF_N };
constexpr unsigned kProfileLaunches = 64, kProfileSub = 64;

}  // namespace photon
