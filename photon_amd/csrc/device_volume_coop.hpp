// device_volume_coop.hpp - wave-cooperative, LDS-staged volume samplers and the RK4 march built
// on them (the hot loop of the library).
//
// Why: a launch lays its rays out so that the 64 lanes of a wave travel together -- source-major for
// narrow cones (BOS: ray_cone_pitch_ratio 1e-4, the rays of one source stay within a fraction of a texel
// of each other), lens-major over spatially sorted sources for full-aperture cones (a patch a few texels
// wide).  Nearly every sample of a coherent wave needs the SAME 4x4x4 (tricubic) / 2x2x2 (trilinear)
// texel block, only the weights differ per lane; an incoherent one needs a small neighbourhood.  A
// per-lane gather issues 64 x 64 16-byte loads for it and is bound by the vector-memory address
// path (measured: 273 ms for 1e7 rays through 256^3).  Here instead:
//
//   * the march is WAVE-SYNCHRONOUS: all 64 lanes stay in the loop until the last ray of the wave
//     has left the volume (finished lanes are predicated off), so every lane is available for
//     cooperative work at each sample;
//   * COHERENT waves (every sampling lane wants the same 4x4x4 / 2x2x2 block: BOS, always, except where a
//     wave straddles two sources or a texel boundary): the whole wave fetches the block with ONE load
//     instruction -- lane l loads texel (l&3, (l>>2)&3, l>>4), clamp-to-edge per texel, 64 lanes x 16 B --
//     parks it in the wave's 1 KiB LDS tile, and the lanes run the separable 64-tap fmaf chain reading
//     the texels back with broadcast ds_read_b128 (all lanes read the same address: conflict-free);
//   * INCOHERENT waves (full-aperture cones): the wave parks an 8x8x4 (8x8x2) BRICK of texels around the
//     first unserved lane's block -- four (two) loads per lane -- and every lane whose block lies within
//     two (three) texels of it runs its chain from the brick in one pass, reading its own addresses; up to
//     PHOTON_BRICK_PASSES bricks per sample, then the per-lane gather for the stragglers.
//
// Every path evaluates the same fmaf chain in the same order: results are bit-identical to the
// per-lane samplers in device_volume.hpp and to the CPU oracle.
#pragma once
#include "../../include/photon_philox.h"
#include "device_volume.hpp"

namespace photon {

// Debug build only (-DPHOTON_PATH_STATS=1, tools/path_stats.py): how often a wave's sample takes which sampler path.
#ifndef PHOTON_PATH_STATS
#define PHOTON_PATH_STATS 0
#endif
#if PHOTON_PATH_STATS
__device__ unsigned long long g_path_stats[8];      // 0 coherent samples, 1 tile fetches, 2 incoherent samples, 3 brick passes, 4 brick fetches, 5 gathered lanes, 6 lanes served per pass (sum)
__device__ __forceinline__ void path_stat(int k, unsigned long long n = 1) { if ((threadIdx.x & 63) == 0) atomicAdd(&g_path_stats[k], n); }
#else
__device__ __forceinline__ void path_stat(int, unsigned long long = 1) {}
#endif

#ifndef PHOTON_BRICK_PASSES
#define PHOTON_BRICK_PASSES 6       // bricks parked per sample before the remaining lanes fall back to the per-lane gather
#endif
#ifndef PHOTON_LINEAR_TILE_LAYERS
#define PHOTON_LINEAR_TILE_LAYERS 16        // layers of a trilinear sampler's coherent tile: 4 / 8 / 16 (3 / 7 / 15 cells of a column)
#endif
#ifndef PHOTON_LINEAR_TILES
#define PHOTON_LINEAR_TILES 2       // tiles the trilinear sampler parks per wave: 2 = a wave across two columns (two light sources) is served like a coherent one
#endif
#ifndef PHOTON_TILE_RETRY_MASK
#define PHOTON_TILE_RETRY_MASK 15  // an incoherent wave tries its tiles again on the trips of the march loop whose number & mask == 0
#endif
// Row pitch of the brick in LDS, in texels.  The brick is 8 texels wide; with a pitch of 8 the 16-byte reads of lanes
// whose blocks sit two rows apart land on the same four banks (a ds_read_b128 serves 16 lanes per LDS cycle from 64
// banks: texel index mod 16 names the bank group, and 8 * dj mod 16 only takes two values).  A pitch of 12 makes
// (di + 12 dj) mod 16 distinct over any 4 x 4 window of block offsets: the patch a lens-major wave covers.  Immediate
// offsets stay immediate (a padded pitch, not an XOR swizzle, whose per-tap addresses would each cost an instruction).
#ifndef PHOTON_BRICK_PITCH
#define PHOTON_BRICK_PITCH 12
#endif
constexpr int kBrickPitch = PHOTON_BRICK_PITCH;                  // texels between consecutive rows of the brick
constexpr int kBrickSlab = 8 * kBrickPitch;                     // texels between consecutive z-slabs
// per wave: the tile(s) -- 4x4x7 = 112 texels for the tricubic sampler, two of 2x2x16 = 128 for the trilinear one -- + the
// brick: 8x8x4 texels (tricubic), 8x8x2 (trilinear), rows padded -- 7.75 KiB against 5 KiB, i.e. at most 5 against 8
// workgroups of four waves in a CU's 160 KiB
// Layers of the tricubic sampler's coherent tile: 4 (one cell: rounds 1-3) ... 8 (five cells of the column the wave travels
// along).  C3, same box, march ms twice each: 4 layers 58.40 / 58.45, 6: 58.17 / 58.39, 7: 58.10 / 58.12, 8: 58.79 / 58.84 --
// 8 KiB of LDS per wave, four waves per SIMD instead of five (and only 0.7 % slower for it: the kernel is not latency-bound).
#ifndef PHOTON_CUBIC_TILE_LAYERS
#define PHOTON_CUBIC_TILE_LAYERS 7
#endif
constexpr int kCubicTileLayers = PHOTON_CUBIC_TILE_LAYERS;
static_assert(kCubicTileLayers >= 4 && kCubicTileLayers <= 8, "PHOTON_CUBIC_TILE_LAYERS");
template <int INTERP> constexpr int tile_texels() { return INTERP == 2 ? 16 * kCubicTileLayers : PHOTON_LINEAR_TILES * PHOTON_LINEAR_TILE_LAYERS * 4; }
#ifndef PHOTON_SPINS_IN_LDS
#define PHOTON_SPINS_IN_LDS 1       // the RK4 trilinear march keeps its rays' `continue` counters in LDS (touched on a rare path only)
#endif
constexpr int kSpinSlotTexels = PHOTON_SPINS_IN_LDS ? 32 : 0;               // 64 lanes x 2 x 4 bytes behind the trilinear brick
template <int INTERP> constexpr int wave_lds_texels() { return tile_texels<INTERP>() + (INTERP == 2 ? 4 : 2) * kBrickSlab + (INTERP == 1 ? kSpinSlotTexels : 0); }
constexpr int kWaveLdsTexels = wave_lds_texels<2>();

// 64-tap sum (slab order) over the block parked in LDS: blk[c*16 + b*4 + a] = texel (a,b,c); each texel
// is one broadcast ds_read_b128.  Plain (unpacked) f32 FMAs on purpose: on gfx950 v_pk_fma_f32 issues
// in 4 cycles against 2 for v_fma_f32 (measured, tools/ubench/fma_rate.hip), so packing buys no
// throughput and costs the (w,w) operand splats; the library is built with -fno-slp-vectorize.
//
// Written as a rolling pipeline over the 16 texel rows (C3: 83.9 -> 80.0 ms when it was introduced); LDS returns
// data in order, so the compiler's s_waitcnt lgkmcnt(N) lets a tap start while the later reads are still in flight.
// RS / SS: texels between consecutive rows / z-slabs of the parked data (4 / 16 for the 4x4x4 tile, 8 / 64
// for the 8x8x4 brick of incoherent waves).
template <int RS, int SS>
__device__ __forceinline__ f4 cubic_taps_lds(const f4 *blk, const float (&wx)[4], const float (&wy)[4],
                                             const float (&wz)[4]) {
    // slab order (device_volume.hpp, tex3d_cubic): 16 products wxy[b][a] = wx[a] * wy[b], each z-slab ONE 16-tap
    // chain (a product, then 15 fmaf per channel), then the z pass: 16 + 256 + 16 = 288 VALU instructions per sample
    // (the fully separable x, y, z order of round 1 took 336: C3 march 66.3 -> 63.5 ms)
    float wxy[4][4];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int a = 0; a < 4; a++) wxy[b][a] = wx[a] * wy[b];
    f4 acc = f4{0, 0, 0, 0}, s = f4{0, 0, 0, 0};
    float w0 = wxy[0][0];
    // ONE row of texels in registers (16 VGPRs): texel (r+1, a) is read into the slot of texel (r, a) right after that
    // one's four FMAs have been issued -- in-order issue: they have read their operands before the later read lands --
    // so every read runs three taps (12 FMAs) ahead of its use and the LDS pipe and the VALU stay busy together
    // instead of alternating "read a slab / wait / FMAs" (round 1 kept a two-row ring, 32 VGPRs, reads of row r+1
    // before the FMAs of row r: 63.0 -> 62.6 ms on C3 and 4 spilled dwords instead of 8).  The asm statements pin the
    // order -- left alone the scheduler hoists all 64 reads to the top (256 VGPRs of texels, one wave per SIMD).
    f4 t[4];
#pragma unroll
    for (int a = 0; a < 4; a++) t[a] = ldtexel(blk + a);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int b = r & 3, c = r >> 2;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const f4 ta = t[a];
            const float w = (a == 0 && b == 0) ? w0 : wxy[b][a];
            if (a == 0 && b == 0) s = f4{w * ta.x, w * ta.y, w * ta.z, w * ta.w};
            else s = f4{fmaf(w, ta.x, s.x), fmaf(w, ta.y, s.y), fmaf(w, ta.z, s.z), fmaf(w, ta.w, s.w)};
            asm volatile("" : "+v"(s.x), "+v"(s.y), "+v"(s.z), "+v"(s.w) : : "memory");
            if (r < 15) t[a] = ldtexel(blk + ((r + 1) >> 2) * SS + ((r + 1) & 3) * RS + a);
            asm volatile("" : "+v"(w0) : : "memory");
        }
        if (b == 3) {
            if (c == 0) acc = f4{wz[0] * s.x, wz[0] * s.y, wz[0] * s.z, wz[0] * s.w};
            else acc = f4{fmaf(wz[c], s.x, acc.x), fmaf(wz[c], s.y, acc.y), fmaf(wz[c], s.z, acc.z), fmaf(wz[c], s.w, acc.w)};
            asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w) : : "memory");
        }
    }
    return acc;
}

// The coherent tile's 64-tap sum with z-slabs served from REGISTERS instead of LDS: `reg` (slab 0) and `more` (slabs 1 ..
// kDppSlabs - 1) hold, in every 16-lane row of the wave, the 16 texels of their slab (lane l: texel l & 15), and a slab's 16
// taps are v_fmac_f32_dpp with row_newbcast:k -- every lane multiplies ITS weight with lane k's texel.  Sixteen of the
// sample's 64 broadcast ds_read_b128 disappear per slab; a DPP multiply-add costs more issue time than a plain one (round
// 2's micro-benchmark: ~1.35x) -- what it buys is power: the chip, which holds its clock down under this kernel's load,
// runs the version with fewer LDS reads faster (kDppSlabs below).
// Same taps, same order, same fmaf chain: same bits.
// The FIRST slab's asm block also issues the four reads of the first LDS-fed row: the 64 register taps run while they are in
// flight, and the block ends on the wait for them (long since landed) -- so the compiler, which does not see LDS
// traffic inside inline asm, finds nothing outstanding after it.  The DPP operand registers are not written
// inside the blocks; the s_nop covers the EXEC-write / VALU-write -> DPP hazards of whatever precedes the first (the compiler's
// hazard recogniser does not look into inline asm).
#define PH_DPP_ROW0 \
    "v_mul_f32_dpp %0, %8, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
    "v_mul_f32_dpp %1, %9, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
    "v_mul_f32_dpp %2, %10, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
    "v_mul_f32_dpp %3, %11, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %0, %8, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %0, %8, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %0, %8, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define PH_DPP_ROW1 \
    "v_fmac_f32_dpp %0, %8, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %0, %8, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %0, %8, %18 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %18 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %18 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %18 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %0, %8, %19 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %19 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %19 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %19 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
#define PH_DPP_ROW2 \
    "v_fmac_f32_dpp %0, %8, %20 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %20 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %20 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %20 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %0, %8, %21 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %21 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %21 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %21 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %0, %8, %22 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %22 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %22 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %22 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %0, %8, %23 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %23 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %23 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %23 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
#define PH_DPP_ROW3 \
    "v_fmac_f32_dpp %0, %8, %24 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %24 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %24 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %24 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %0, %8, %25 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %25 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %25 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %25 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %0, %8, %26 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %26 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %26 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %26 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %0, %8, %27 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %1, %9, %27 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %10, %27 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %3, %11, %27 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
#define PH_DPP_ROWS_TEXT PH_DPP_ROW0 PH_DPP_ROW1 PH_DPP_ROW2 PH_DPP_ROW3
// the four texels the asm block reads ahead: row 0 of slab 1
#define PH_DS_OFF1 "16"
#define PH_DS_OFF2 "32"
#define PH_DS_OFF3 "48"
// WAVE PRIORITY inside the march (s_setprio).  A SIMD holds five or six waves of the march; the arbiter picks among the
// ready ones oldest-first, whatever they are about to do.  Four levels are used:
//   PHOTON_PRIO_BASE 1   everything not named below (stage algebra, weights, tile tests, exits), set when the march starts;
//   PHOTON_PRIO_TAPS_DPP 3   the DPP blocks of the tricubic chain's register slabs -- what holds the tile's LDS rows and
//     the wave's weight registers live; with ONE slab in registers raising them cut the busy cycles of the headline march by
//     12 % (levels then, DPP block / LDS-fed slabs over a base of 0: 3/0, 2/0, 1/0 all 54.5 ms; 3/1, 2/1 53.8-54.1; 1/1 55.2;
//     0/3 58.8; none 57.1; profiles/r05_d_priority.txt); the card answered with a lower clock (its power limit), which is
//     what moving two more slabs into registers then addressed (kDppSlabs below).  With THREE register slabs the chain is
//     mostly DPP forms and cannot do without: none 67.95 ms, 1/1 57.10, 3/3 55.89, 2/1 53.47, 3/1 53.40, 3/0 53.17;
//   PHOTON_PRIO_TAPS_LDS 0, PHOTON_PRIO_BLEND 0   the stretches that are FED BY LDS READS and wait on them: the tricubic
//     chain's last slab and the eight-texel trilinear blend run BELOW the base level -- a wave that is about to wait anyway
//     (tile and brick alike: a quarter of C5 through the trilinear sampler 12.95 ms with the brick's blend at the base
//     level, 12.53 below it, 13.25 above) yields its issue slots.  RK4 trilinear march 15.47 -> 15.09 ms (its eighth 2.064 -> 2.012), Euler trilinear 5.557 ->
//     5.32; tricubic 52.71 -> 52.41, its eighth 6.930 -> 6.835 (base 1 against base 0; base 2 the same).  RAISING the
//     trilinear blend above the rest had lost 1 % (15.47 -> 15.62);
//   PHOTON_PRIO_BRICK 2   the brick chain of the lanes outside the tiles (per-lane LDS reads, 64 of them): one above the
//     base -- a quarter of C5 36.5 -> 35.7 ms when it was introduced; at the base level 36.6, below it 37.3.
// Results do not depend on any of it: priority orders issue, not arithmetic (profiles/r05_e_register_slabs.txt, r05_f_priority_base.txt).
#ifndef PHOTON_PRIO_BASE
#define PHOTON_PRIO_BASE 1
#endif
#ifndef PHOTON_PRIO_BLEND
#define PHOTON_PRIO_BLEND 0
#endif
#ifndef PHOTON_PRIO_TAPS_DPP
#define PHOTON_PRIO_TAPS_DPP 3
#endif
#ifndef PHOTON_PRIO_TAPS_LDS
#define PHOTON_PRIO_TAPS_LDS 0
#endif
#ifndef PHOTON_PRIO_BRICK
#define PHOTON_PRIO_BRICK 2
#endif
// How many of the cell's four z-slabs are served from registers (1 .. 4; the rest from the tile in LDS).  The DPP form is
// the dearer instruction to issue, a broadcast ds_read_b128 the dearer one in watts -- and this kernel runs at the board's
// power limit.  With the chains at a raised wave priority (above), same box, C3 tricubic march | one GPU's eighth | clock:
//   1 slab 53.88 ms | 6.948 | 2113 MHz     2 slabs 52.53 | 6.818 | 2172     3 slabs 51.84 | 6.838 | 2288     4 slabs 57.47 | 7.579 | 2376
// (cycles: 1.00, 1.00, 1.04, 1.20).  By rows of four taps on another box -- 8: 51.96, 10: 51.73, 11: 51.60, 12: 51.80, 13: 52.20,
// 14: 52.89 -- a flat optimum; whole slabs keep the code one loop, and three leave the kernel close to the clock's ceiling,
// i.e. least exposed to what a particular card's power budget allows (profiles/r05_e_register_slabs.txt).  Round 3 had
// measured one slab against none without the wave priority: same cycles, +2.7 % clock.
#ifndef PHOTON_DPP_SLABS
#define PHOTON_DPP_SLABS 3
#endif
constexpr int kDppSlabs = PHOTON_DPP_SLABS;
static_assert(kDppSlabs >= 1 && kDppSlabs <= 4, "PHOTON_DPP_SLABS");
#define PH_DPP_OPERANDS(S, T) \
            : "=&v"(S.x), "=&v"(S.y), "=&v"(S.z), "=&v"(S.w), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) \
            : "v"(T.x), "v"(T.y), "v"(T.z), "v"(T.w), "v"(wxy[0][0]), "v"(wxy[0][1]), "v"(wxy[0][2]), "v"(wxy[0][3]), \
              "v"(wxy[1][0]), "v"(wxy[1][1]), "v"(wxy[1][2]), "v"(wxy[1][3]), "v"(wxy[2][0]), "v"(wxy[2][1]), "v"(wxy[2][2]), \
              "v"(wxy[2][3]), "v"(wxy[3][0]), "v"(wxy[3][1]), "v"(wxy[3][2]), "v"(wxy[3][3]) \
            : "memory"
// one z-slab's sixteen taps from the registers `tex` (its texels, one per lane of every 16-lane row) into s
__device__ __forceinline__ void dpp_slab(f4 &s, const f4 &tex, const float (&wxy)[4][4]) {
    float d0, d1, d2, d3;                                           // operand slots 4-7: the read-ahead of the first block, unused here
    // the s_nop: the compiler's hazard recogniser does not look into inline asm -- should it ever write one of these operands
    // (or EXEC) in the instruction before, the DPP reads would come too early (2 / 5 wait states); non-VALU slots are free here
    asm volatile("s_nop 4\n\t" PH_DPP_ROWS_TEXT PH_DPP_OPERANDS(s, tex));
}
__device__ __forceinline__ f4 cubic_taps_hybrid(const f4 *blk, const f4 &reg, const f4 (&more)[3], const float (&wx)[4], const float (&wy)[4],
                                                const float (&wz)[4]) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    constexpr int R = 4 * kDppSlabs;                                // rows (of four taps) served from registers
    float wxy[4][4];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int a = 0; a < 4; a++) wxy[b][a] = wx[a] * wy[b];
    v4f t0, t1, t2, t3;
    f4 s;
    const unsigned lds = (unsigned)(size_t)blk + 64u * R;           // LDS byte address of the first row read from the tile
#if PHOTON_PRIO_TAPS_DPP != PHOTON_PRIO_BASE
    __builtin_amdgcn_s_setprio(PHOTON_PRIO_TAPS_DPP);
#endif
    if (R < 16)
    asm volatile(
            "ds_read_b128 %4, %28\n\t"
            "ds_read_b128 %5, %28 offset:" PH_DS_OFF1 "\n\t"
            "ds_read_b128 %6, %28 offset:" PH_DS_OFF2 "\n\t"
            "ds_read_b128 %7, %28 offset:" PH_DS_OFF3 "\n\t"
            "s_nop 4\n\t"
            PH_DPP_ROWS_TEXT
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(s.x), "=&v"(s.y), "=&v"(s.z), "=&v"(s.w), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
            : "v"(reg.x), "v"(reg.y), "v"(reg.z), "v"(reg.w), "v"(wxy[0][0]), "v"(wxy[0][1]), "v"(wxy[0][2]), "v"(wxy[0][3]),
              "v"(wxy[1][0]), "v"(wxy[1][1]), "v"(wxy[1][2]), "v"(wxy[1][3]), "v"(wxy[2][0]), "v"(wxy[2][1]), "v"(wxy[2][2]),
              "v"(wxy[2][3]), "v"(wxy[3][0]), "v"(wxy[3][1]), "v"(wxy[3][2]), "v"(wxy[3][3]), "v"(lds)
            : "memory");
    else {                                                          // every row from registers: nothing to read ahead
        dpp_slab(s, reg, wxy);
        t0 = t1 = t2 = t3 = v4f{0, 0, 0, 0};
    }
    f4 acc = f4{wz[0] * s.x, wz[0] * s.y, wz[0] * s.z, wz[0] * s.w};                 // slab 0 complete
#pragma unroll
    for (int z = 1; z < kDppSlabs; z++) {                           // the next slabs the same way, from their registers
        dpp_slab(s, more[z - 1], wxy);
        acc = f4{fmaf(wz[z], s.x, acc.x), fmaf(wz[z], s.y, acc.y), fmaf(wz[z], s.z, acc.z), fmaf(wz[z], s.w, acc.w)};
        asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w) : : "memory");
    }
#if PHOTON_PRIO_TAPS_DPP != PHOTON_PRIO_TAPS_LDS
    __builtin_amdgcn_s_setprio(PHOTON_PRIO_TAPS_LDS);
#endif
    f4 t[4] = {f4{t0.x, t0.y, t0.z, t0.w}, f4{t1.x, t1.y, t1.z, t1.w}, f4{t2.x, t2.y, t2.z, t2.w}, f4{t3.x, t3.y, t3.z, t3.w}};
    float w0 = wxy[0][0];
#pragma unroll
    for (int r = R; r < 16; r++) {                                  // the rest of the tile from LDS, as in cubic_taps_lds
        const int b = r & 3, c = r >> 2;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const f4 ta = t[a];
            const float w = (a == 0 && b == 0) ? w0 : wxy[b][a];
            if (a == 0 && b == 0) s = f4{w * ta.x, w * ta.y, w * ta.z, w * ta.w};
            else s = f4{fmaf(w, ta.x, s.x), fmaf(w, ta.y, s.y), fmaf(w, ta.z, s.z), fmaf(w, ta.w, s.w)};
            asm volatile("" : "+v"(s.x), "+v"(s.y), "+v"(s.z), "+v"(s.w) : : "memory");
            if (r < 15) t[a] = ldtexel(blk + ((r + 1) >> 2) * 16 + ((r + 1) & 3) * 4 + a);
            asm volatile("" : "+v"(w0) : : "memory");
        }
        if (b == 3) {
            if (c == 0) acc = f4{wz[0] * s.x, wz[0] * s.y, wz[0] * s.z, wz[0] * s.w};
            else acc = f4{fmaf(wz[c], s.x, acc.x), fmaf(wz[c], s.y, acc.y), fmaf(wz[c], s.z, acc.z), fmaf(wz[c], s.w, acc.w)};
            asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w) : : "memory");
        }
    }
#if PHOTON_PRIO_TAPS_LDS != PHOTON_PRIO_BASE
    __builtin_amdgcn_s_setprio(PHOTON_PRIO_BASE);
#endif
    return acc;
}

// Lane predicates of the march are kept as WAVE MASKS (64-bit, wave-uniform, SGPR pairs): a mask comes out of
// ballot(one comparison) -- a single v_cmp writing an SGPR pair -- and masks combine with integer & | ~ on the scalar
// unit.  A predicate that is an AND / OR of i1 values reaches ballot() as a 0/1 VGPR instead (v_cndmask + v_cmp per use),
// and through nested branches it drags copies of the ray state along: rounds 1-2's form, ~45 VALU instructions per sample.
// lane_of() turns a mask back into this lane's predicate where the code really is per lane (commits, rare paths).
__device__ __forceinline__ bool lane_of(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

// What is parked in the wave's LDS area (wave-uniform, kept by the marchers across samples): consecutive
// samples of a ray advance by half a texel, so about every other sample finds its block / brick still there
// and skips the fetch.  A block is named by the BIT PATTERNS of its three floor() values (readlane'd from its
// leader): comparing those needs no float -> int conversion on the per-sample path.
// ti, tj, tk: the tricubic sampler's current CELL (the one the register slab belongs to), the trilinear sampler's column and
// base layer (tile A; ui, uj, uk: tile B).  ci, cj, k0, coff (tricubic, tiles deeper than one cell): the tile's column (bit patterns), the k of its first
// cell, and the current cell's texel offset in it.
struct Parked { int ti, tj, tk; int bi, bj, bk; f4 reg; int ci, cj, k0, coff; int ui, uj, uk; f4 more[3]; };      // reg: z-slab 0 of the current cell in registers (per lane); more: z-slabs 1 .. kDppSlabs - 1 likewise
__device__ __forceinline__ Parked parked_none() {                // 0x7fffffff: a NaN pattern no floor() of a sampled coordinate has
    return Parked{0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, f4{0, 0, 0, 0}, 0x7fffffff, 0x7fffffff, 0, 0, 0x7fffffff, 0x7fffffff, 0x7fffffff, {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}}};
}

// Block coherence of a wave's sample: the first sampling lane leads; the wave is coherent when every sampling lane
// wants the leader's block.  The floor values are compared as bit patterns (each integer-valued float has one pattern;
// +0 is the only zero a floor of x - 0.5 produces): three v_readlane and three integer compares per sample, the rest on
// the scalar unit.
struct Lead { int i, j, k; };                                   // float bit patterns, wave-uniform (SGPRs)
__device__ __forceinline__ Lead lead_of(unsigned long long todo, float fi, float fj, float fk) {
    const int leader = __ffsll((long long)todo) - 1;
    return Lead{__builtin_amdgcn_readlane(__float_as_int(fi), leader), __builtin_amdgcn_readlane(__float_as_int(fj), leader),
                __builtin_amdgcn_readlane(__float_as_int(fk), leader)};
}
__device__ __forceinline__ unsigned long long same_block(const Lead &c, float fi, float fj, float fk) {
    return ballot(__float_as_int(fi) == c.i) & ballot(__float_as_int(fj) == c.j) & ballot(__float_as_int(fk) == c.k);
}

// Must be called by ALL 64 lanes of the wave (wave-uniform control flow); `need` = mask of the lanes that want a
// sample (the value returned to the other lanes is unspecified).  blk = this wave's LDS area: a 64-texel tile followed
// by a 256-texel brick.
__device__ __forceinline__ f4 tex3d_cubic_coop(const VolumeDev &v, const f4 *__restrict__ tex, f4 *blk, unsigned long long need,
                                               float x, float y, float z, Parked &parked) {
    f4 *const brick = blk + tile_texels<2>();                   // the wave's 8x8x4 brick follows its tile
    const float xg = x - 0.5f, yg = y - 0.5f, zg = z - 0.5f;
    const float fi = floorf(xg), fj = floorf(yg), fk = floorf(zg);
    float wx[4], wy[4], wz[4];
    bspline_weights(xg - fi, wx[0], wx[1], wx[2], wx[3]);
    bspline_weights(yg - fj, wy[0], wy[1], wy[2], wy[3]);
    bspline_weights(zg - fk, wz[0], wz[1], wz[2], wz[3]);
    const int lane = threadIdx.x & 63;
    if (need == 0) return f4{0, 0, 0, 0};                       // wave-uniform
    {
        // COHERENT wave -- every sampling lane in ONE cell (BOS: always, except where a wave straddles two sources or a
        // cone straddles a texel boundary).  The parked tile is 4 x 4 texels wide and TL layers deep: TL - 3 cells of the
        // column the wave travels along (a ray advances a cell per RK4 iteration: the one-cell tile of rounds 1-3 was
        // fetched 0.35 times per sample), lane l <-> texel (l&3, (l>>2)&3, l>>4) of every four layers, clamp-to-edge per
        // texel.  The chain reads its cell with broadcast reads at the cell's offset in the tile; the first z-slabs of the CURRENT
        // cell sit in registers (parked.reg, parked.more) and are re-read from LDS when the wave moves on to the next cell of the
        // tile.  The test is against the current cell first (no leader, no readlane); a wave that went to the bricks
        // forgets its cell and skips that test while it stays incoherent.
        constexpr int TL = kCubicTileLayers;
        bool hit = false;
        if (parked.ti != 0x7fffffff) {                          // wave-uniform
            asm volatile("");
            hit = ((ballot(__float_as_int(fi) == parked.ti) & ballot(__float_as_int(fj) == parked.tj) & ballot(__float_as_int(fk) == parked.tk)) & need) == need;
        }
        if (!hit) {
            const Lead c = lead_of(need, fi, fj, fk);
            if ((same_block(c, fi, fj, fk) & need) == need) {   // wave-uniform: one cell
                const int ck = (int)__int_as_float(c.k);
                int dz = ck - parked.k0;
                if (c.i != parked.ci || c.j != parked.cj || (unsigned)dz > (unsigned)(TL - 4)) {      // not in the parked tile
                    path_stat(1);
                    const int ci = (int)__int_as_float(c.i), cj = (int)__int_as_float(c.j);
                    // the tile starts at the wave's cell when it travels upwards (or nothing tells), ends there when downwards
                    const bool down = c.i == parked.ci && c.j == parked.cj && dz < 0;
                    const int k0 = down ? ck - (TL - 4) : ck;
                    __builtin_amdgcn_wave_barrier();
                    int l = lane;
                    asm volatile("" : "+v"(l));                 // keep the tile's lane offsets out of the march loop's live registers
                    const int tx = clampi(ci - 1 + (l & 3), 0, v.nx - 1), ty = clampi(cj - 1 + ((l >> 2) & 3), 0, v.ny - 1);
#pragma unroll
                    for (int j = 0; j < (TL + 3) / 4; j++) {
                        if (TL % 4 == 0 || j < TL / 4 || l < 16 * (TL % 4)) {
                            const int tz = clampi(k0 - 1 + (l >> 4) + 4 * j, 0, v.nz - 1);
                            const f4 t = ldtexel(tex + (unsigned)((tz * v.ny + ty) * v.nx + tx));      // < 2^31 texels (checked on the host)
                            *reinterpret_cast<float4 *>(blk + l + 64 * j) = make_float4(t.x, t.y, t.z, t.w);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    parked.ci = c.i; parked.cj = c.j; parked.k0 = k0;
                    dz = ck - k0;
                }
                parked.coff = dz * 16;
                parked.reg = ldtexel(blk + parked.coff + (lane & 15));   // slab 0 of the cell, into every 16-lane row
#pragma unroll
                for (int z = 1; z < kDppSlabs; z++) parked.more[z - 1] = ldtexel(blk + parked.coff + 16 * z + (lane & 15));
                parked.ti = c.i; parked.tj = c.j; parked.tk = c.k;
                hit = true;
            }
        }
        if (hit) {
            path_stat(0);
            const f4 *cell = blk + parked.coff;
            const f4 acc = cubic_taps_hybrid(cell, parked.reg, parked.more, wx, wy, wz);
            __builtin_amdgcn_wave_barrier();
            return acc;
        }
        parked.ti = 0x7fffffff;                                 // to the bricks: the cell is forgotten (the tile stays)
    }
    // INCOHERENT wave (full-aperture cones, source boundaries: the lanes' blocks form a patch a few texels wide
    // in one z-slab).  Serving it group by group would cost the whole wave one chain per distinct block.
    // Instead: park the 8x8x4 BRICK around the first unserved lane's block -- four loads per lane -- and let
    // every lane whose block lies within +-2 texels of it in x and y (same z) run its chain from there in ONE
    // pass, each lane reading its own addresses.  Same chain, same order, same bits.
    const int bi = (int)fi, bj = (int)fj, bk = (int)fk;         // lanes that do not sample never lead and never match
    f4 acc = f4{0, 0, 0, 0};
    bool done = !lane_of(need);
    unsigned long long todo = need;
    path_stat(2);
#pragma unroll 1
    for (int pass = 0; pass < PHOTON_BRICK_PASSES && todo != 0; pass++) {
        const int leader = __ffsll((long long)todo) - 1;
        const int ci = __builtin_amdgcn_readlane(bi, leader), cj = __builtin_amdgcn_readlane(bj, leader),
                  ck = __builtin_amdgcn_readlane(bk, leader);
        const int di = bi - ci + 2, dj = bj - cj + 2;
        const bool in_brick = !done && bk == ck && (unsigned)di <= 4u && (unsigned)dj <= 4u;
        path_stat(3);
        path_stat(6, (unsigned long long)__popcll(ballot(in_brick)));
        if (ci != parked.bi || cj != parked.bj || ck != parked.bk) {        // wave-uniform: the brick is not parked yet
            path_stat(4);
            __builtin_amdgcn_wave_barrier();
            int l = lane;
            asm volatile("" : "+v"(l));                         // keep the brick's lane offsets out of the march loop's live registers
            const int slot = (l & 7) + ((l >> 3) & 7) * kBrickPitch;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int t = l + 64 * j;
                const int tx = clampi(ci - 3 + (t & 7), 0, v.nx - 1), ty = clampi(cj - 3 + ((t >> 3) & 7), 0, v.ny - 1),
                          tz = clampi(ck - 1 + (t >> 6), 0, v.nz - 1);
                const f4 tv = ldtexel(tex + (unsigned)((tz * v.ny + ty) * v.nx + tx));
                *reinterpret_cast<float4 *>(brick + slot + j * kBrickSlab) = make_float4(tv.x, tv.y, tv.z, tv.w);
            }
            __builtin_amdgcn_wave_barrier();
            parked.bi = ci; parked.bj = cj; parked.bk = ck;
        }
        if (in_brick) {
#if PHOTON_PRIO_BRICK != PHOTON_PRIO_BASE
            __builtin_amdgcn_s_setprio(PHOTON_PRIO_BRICK);
#endif
            acc = cubic_taps_lds<kBrickPitch, kBrickSlab>(brick + (dj * kBrickPitch + di), wx, wy, wz);
#if PHOTON_PRIO_BRICK != PHOTON_PRIO_BASE
            __builtin_amdgcn_s_setprio(PHOTON_PRIO_BASE);
#endif
            done = true;
        }
        __builtin_amdgcn_wave_barrier();
        todo = ballot(!done);
    }
    path_stat(5, (unsigned long long)__popcll(ballot(!done)));
    if (!done) acc = cubic_gather_fn(tex, v.nx, v.ny, v.nz, x, y, z);                       // stragglers
    return acc;
}

// The trilinear blend with per-lane, clamped addressing (stragglers of an incoherent wave, the rare repair sample).  Not
// inlined, for the same reason as cubic_gather_fn: it is the rare path and would otherwise sit, with its eight address
// computations, at every sampler call site of the march.  And LEAN on purpose: under the AMDGPU calling convention what
// the caller keeps live across the call sits ABOVE the callee's registers, so this function's VGPR count is part of the
// march kernels' budget at every call site -- with all eight texels in flight it needed 52, which left the RK4 kernel (80
// VGPRs, six waves per SIMD) 28 for its ray state and made it spill into its loop.  One x-pair of texels at a time (the
// empty asm statements keep the compiler from hoisting the later loads): same lerp tree, same bits.
__device__ __attribute__((noinline)) f4 linear_gather_fn(const f4 *__restrict__ tex, int nx, int ny, int nz, float x,
                                                         float y, float z, float weight_scale, float weight_inv) {
    const float xb = x - 0.5f, yb = y - 0.5f, zb = z - 0.5f;
    const float fi = floorf(xb), fj = floorf(yb), fk = floorf(zb);
    const float a = quant_weight(xb - fi, weight_scale, weight_inv), b = quant_weight(yb - fj, weight_scale, weight_inv),
                c = quant_weight(zb - fk, weight_scale, weight_inv);
    const int i = (int)fi, j = (int)fj, k = (int)fk;
    const int i0 = clampi(i, 0, nx - 1), i1 = clampi(i + 1, 0, nx - 1);
    const int j0 = clampi(j, 0, ny - 1), j1 = clampi(j + 1, 0, ny - 1);
    const int k0 = clampi(k, 0, nz - 1), k1 = clampi(k + 1, 0, nz - 1);
    auto row = [&](int kk, int jj) {                            // first-level lerp of one x-pair
        const f4 *r = tex + (unsigned)((kk * ny + jj) * nx);    // < 2^31 texels (checked on the host)
        f4 v = lerp4(ldtexel(r + i0), ldtexel(r + i1), a);
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        return v;
    };
    const f4 c00 = row(k0, j0), c10 = row(k0, j1);
    f4 c0 = lerp4(c00, c10, b);
    asm volatile("" : "+v"(c0.x), "+v"(c0.y), "+v"(c0.z), "+v"(c0.w));
    const f4 c01 = row(k1, j0), c11 = row(k1, j1);
    const f4 c1 = lerp4(c01, c11, b);
    return lerp4(c0, c1, c);
}

// Inline asm on purpose: the builtin is folded away when the compiler can prove its operand uniform,
// which leaves a VALU-computed float (cvt, div) in a VGPR for the whole loop -- or in scratch.
// The s_nops cover the gfx940+ hazards the compiler cannot see through inline asm (VALU writes VGPR ->
// readlane reads it: 1 wait state; VALU writes SGPR -> VALU / VMEM reads it: 2 / 5 wait states); this
// runs once per kernel, in the prologue.
__device__ __forceinline__ float uniformf(float x) {
    float s;
    asm("s_nop 0\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(s) : "v"(x));
    return s;
}

// v of the neighbouring lane of the pair (lane ^ 1): one DPP move, no LDS round trip
__device__ __forceinline__ float pair_swap(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, false));
}

// QUANT: the volume uses the texture unit's fixed-point weights (v.weight_scale > 0; the march kernels branch ONCE on
// it -- as a run-time select per weight it cost three v_cndmask per sample).
template <bool QUANT, bool PIN = false>
__device__ __forceinline__ f4 tex3d_linear_coop(const VolumeDev &v, const f4 *__restrict__ tex, f4 *blk, unsigned long long need,
                                                float x, float y, float z, Parked &parked, unsigned tick = 0) {
    f4 *const brick = blk + tile_texels<1>();                   // 8x8x2 texels around the leader for incoherent waves
    const float xb = x - 0.5f, yb = y - 0.5f, zb = z - 0.5f;
    const float fi = floorf(xb), fj = floorf(yb), fk = floorf(zb);
    float a = xb - fi, b = yb - fj, c = zb - fk;
    if (QUANT) {                                                // texture-unit weights (8 fractional bits)
        a = floorf(fmaf(a, v.weight_scale, 0.5f)) * v.weight_inv;           // quant_weight() with the scale known positive

        b = floorf(fmaf(b, v.weight_scale, 0.5f)) * v.weight_inv;
        c = floorf(fmaf(c, v.weight_scale, 0.5f)) * v.weight_inv;
    }
    const int lane = threadIdx.x & 63;
    if (need == 0) return f4{0, 0, 0, 0};                       // wave-uniform
    {
        constexpr int TL = PHOTON_LINEAR_TILE_LAYERS;                // layers of a tile: TL - 1 cells of a column
        constexpr int NT = PHOTON_LINEAR_TILES;                      // tiles parked per wave
        static_assert(TL == 4 || TL == 8 || TL == 16, "PHOTON_LINEAR_TILE_LAYERS");
        static_assert(NT == 1 || NT == 2, "PHOTON_LINEAR_TILES");
        // coherent wave: ONE column of cells serves everybody.  A parked tile is 2 x 2 texels wide and TL layers deep --
        // TL - 1 cells of the column the wave is travelling along (a ray advances one cell per RK4 iteration, so the 2x2x2
        // tile of rounds 1-3 was fetched about once per iteration, 0.34 fetches per sample on C3, and the kernel waits on
        // each: 0.05 with sixteen layers) -- fetched by lanes 0 .. 4 TL - 1 with one load instruction and parked as
        // (texel, x-difference) pairs: slot 4 L + 2 b = texel (0, b, L), slot 4 L + 2 b + 1 = texel (1, b, L) - texel
        // (0, b, L) -- the subtraction every lane's first-level lerp fmaf(a, t1 - t0, t0) would repeat on identical operands
        // is done once per tile by the fetching lane (same f32 operation, same bits).  A lane whose cell is dz layers above
        // the tile's base blends from slot 4 dz on, each lane reading its own addresses: the lanes of a wave that is
        // crossing a layer boundary are served together.  The test is against the PARKED tile (no leader, no readlane, on
        // the common path): same column -- the bit patterns of two floor() values -- and dz = fk - base in 0 .. TL - 2, by
        // conversion (a negative difference converts to a large unsigned value).
        // TWO tiles are parked (A, and B right behind it in LDS): a wave whose lanes sit in two columns -- the last rays of
        // one light source and the first of the next (one wave in eight on C3), a cone across a cell boundary -- is served
        // like a coherent one, each lane blending from its column's tile; only when the first test fails is the second made.
        // A wave that has to go to the bricks (three columns or more) forgets its tiles and marks itself incoherent (ui =
        // kIncoherent): it then goes straight to the bricks, trying its tiles again only on every sixteenth trip of the
        // march loop (`tick`: a counter the loop keeps anyway -- the bookkeeping takes no register, and the trilinear
        // kernels have none to spare: with two more scalars for a pause counter RK4 on C3 ran 4 % slower).  A wave of
        // full-aperture cones (C5: never coherent) pays a scalar compare and branch per sample for the tiles it cannot
        // use.  (The empty asm keeps the compiler from turning the scalar branch around the first test into
        // unconditional vector code.)
        constexpr unsigned kDzMax = (unsigned)(TL - 2);
        constexpr int kIncoherent = 0x7ffffffe;                 // another NaN pattern
        const int fib = __float_as_int(fi), fjb = __float_as_int(fj);
        unsigned dzq = 0;                                       // the lane's layer offset in the tile area (tile B: + TL)
        unsigned long long ok_a = 0, ok_b = 0;                  // lanes whose cell lies in tile A / B
        bool hit = false;
        if (parked.ti != 0x7fffffff) {                          // wave-uniform
            asm volatile("");
            dzq = (unsigned)(int)(fk - __int_as_float(parked.tk));
            ok_a = ballot(fib == parked.ti) & ballot(fjb == parked.tj) & ballot(dzq <= kDzMax);
            hit = (ok_a & need) == need;
            if (NT == 2 && !hit && parked.ui < kIncoherent) {   // wave-uniform
                const unsigned dzu = (unsigned)(int)(fk - __int_as_float(parked.uk));
                ok_b = ballot(fib == parked.ui) & ballot(fjb == parked.uj) & ballot(dzu <= kDzMax);
                if (((ok_a | ok_b) & need) == need) {
                    hit = true;
                    dzq = lane_of(ok_a) ? dzq : dzu + TL;
                }
            }
        }
        if (!hit && (parked.ui != kIncoherent || (tick & (unsigned)PHOTON_TILE_RETRY_MASK) == 0)) {      // wave-uniform: some sampling lanes have no tile
            // the columns of the lanes without a tile: at most as many as there are tiles that serve nobody
            const unsigned long long rest = need & ~(ok_a | ok_b);
            const Lead l1 = lead_of(rest, fi, fj, fk);
            const unsigned long long col1 = rest & ballot(fib == l1.i) & ballot(fjb == l1.j), rest2 = rest & ~col1;
            Lead l2 = l1;
            unsigned long long col2 = 0;
            int n_new = 1;
            if (rest2 != 0) {
                l2 = lead_of(rest2, fi, fj, fk);
                col2 = rest2 & ballot(fib == l2.i) & ballot(fjb == l2.j);
                n_new = (rest2 & ~col2) != 0 ? 3 : 2;
            }
            const bool free_a = (ok_a & need) == 0, free_b = NT == 2 && (ok_b & need) == 0;
            if (n_new <= (free_a ? 1 : 0) + (free_b ? 1 : 0)) {
#pragma unroll 1
                for (int j = 0; j < n_new; j++) {
                    path_stat(1);
                    const bool to_a = j == 0 && free_a;         // the second new tile always goes to B
                    const Lead ld = j == 0 ? l1 : l2;
                    const unsigned long long lanes = j == 0 ? col1 : col2;
                    // where the tile starts: the leader's layer when the wave travels upwards (or nothing tells: first fetch,
                    // new column), TL - 2 below it when downwards -- the tile this one replaces says which --, shifted by
                    // one when lanes of the column sit on the other side of the leader (a wave crossing a layer boundary)
                    const int oi = to_a ? parked.ti : parked.ui, oj = to_a ? parked.tj : parked.uj, ok = to_a ? parked.tk : parked.uk;
                    const float klf = __int_as_float(ld.k);
                    const bool behind = (lanes & ballot(fk < klf)) != 0, ahead = (lanes & ballot(fk > klf)) != 0;
                    const bool down = ld.i == oi && ld.j == oj && klf < __int_as_float(ok);
                    const int ci = (int)__int_as_float(ld.i), cj = (int)__int_as_float(ld.j);
                    const int base = (int)klf - (down ? (ahead ? TL - 3 : TL - 2) : (behind ? 1 : 0));
                    f4 *const tile = blk + (to_a ? 0 : 4 * TL);
                    __builtin_amdgcn_wave_barrier();
                    if (lane < 4 * TL) {
                        int l = lane;
                        if (PIN) asm volatile("" : "+v"(l));   // keep the tile's lane offsets out of the march loop's live
                                                                // registers: hoisted as loop invariants they were spilled, every fetch reloading them
                        const int tx = clampi(ci + (l & 1), 0, v.nx - 1), ty = clampi(cj + ((l >> 1) & 1), 0, v.ny - 1),
                                  tz = clampi(base + (l >> 2), 0, v.nz - 1);
                        const f4 t = ldtexel(tex + (unsigned)((tz * v.ny + ty) * v.nx + tx));
                        const f4 o = f4{pair_swap(t.x), pair_swap(t.y), pair_swap(t.z), pair_swap(t.w)};     // the pair's other texel
                        const bool hi = (l & 1) != 0;
                        *reinterpret_cast<float4 *>(tile + l) = make_float4(hi ? t.x - o.x : t.x, hi ? t.y - o.y : t.y,
                                                                            hi ? t.z - o.z : t.z, hi ? t.w - o.w : t.w);
                    }
                    __builtin_amdgcn_wave_barrier();
                    const int basef = __float_as_int(uniformf((float)base));
                    if (to_a) { parked.ti = ld.i; parked.tj = ld.j; parked.tk = basef; }
                    else { parked.ui = ld.i; parked.uj = ld.j; parked.uk = basef; }
                }
                // every sampling lane's column is parked now; its layer may still be out of reach (lanes of one column
                // more than TL - 1 layers apart): then the bricks serve the wave
                dzq = (unsigned)(int)(fk - __int_as_float(parked.tk));
                ok_a = ballot(fib == parked.ti) & ballot(fjb == parked.tj) & ballot(dzq <= kDzMax);
                if (NT == 2 && parked.ui < kIncoherent) {
                    const unsigned dzu = (unsigned)(int)(fk - __int_as_float(parked.uk));
                    ok_b = ballot(fib == parked.ui) & ballot(fjb == parked.uj) & ballot(dzu <= kDzMax);
                    dzq = lane_of(ok_a) ? dzq : dzu + TL;
                }
                hit = ((ok_a | ok_b) & need) == need;
                if (hit && parked.ui == kIncoherent) parked.ui = 0x7fffffff;     // coherent again
            }
        }
        if (hit) {
            path_stat(0);
            // q[tc*4 + tb*2 + {0: texel, 1: x-difference}], q = the lane's cell in its tile; same lerp tree as tex3d_linear
            const f4 *q = blk + ((dzq & (unsigned)(NT * TL - 1)) << 2);         // (the mask: lanes that do not sample stay inside the wave's LDS area)
#if PHOTON_PRIO_BLEND != PHOTON_PRIO_BASE
            __builtin_amdgcn_s_setprio(PHOTON_PRIO_BLEND);
#endif
            const f4 c00 = lerp4d(ldtexel(q), ldtexel(q + 1), a), c10 = lerp4d(ldtexel(q + 2), ldtexel(q + 3), a);
            const f4 c01 = lerp4d(ldtexel(q + 4), ldtexel(q + 5), a), c11 = lerp4d(ldtexel(q + 6), ldtexel(q + 7), a);
            const f4 c0 = lerp4(c00, c10, b), c1 = lerp4(c01, c11, b);
            const f4 acc = lerp4(c0, c1, c);
#if PHOTON_PRIO_BLEND != PHOTON_PRIO_BASE
            __builtin_amdgcn_s_setprio(PHOTON_PRIO_BASE);
#endif
            __builtin_amdgcn_wave_barrier();
            return acc;
        }
        parked.ti = 0x7fffffff;                                 // to the bricks: the tiles are forgotten
        parked.ui = kIncoherent;
    }
    // incoherent wave: bricks of 8x8x2 texels around the first unserved lane (two loads per lane); every lane
    // whose block starts within [-3, +3] texels of it in x and y (same z) blends from there in one pass
    const int bi = (int)fi, bj = (int)fj, bk = (int)fk;
    f4 acc = f4{0, 0, 0, 0};
    bool done = !lane_of(need);
    unsigned long long todo = need;
    path_stat(2);
#pragma unroll 1
    for (int pass = 0; pass < PHOTON_BRICK_PASSES && todo != 0; pass++) {
        const int leader = __ffsll((long long)todo) - 1;
        const int ci = __builtin_amdgcn_readlane(bi, leader), cj = __builtin_amdgcn_readlane(bj, leader),
                  ck = __builtin_amdgcn_readlane(bk, leader);
        const int di = bi - ci + 3, dj = bj - cj + 3;
        const bool in_brick = !done && bk == ck && (unsigned)di <= 6u && (unsigned)dj <= 6u;
        path_stat(3);
        path_stat(6, (unsigned long long)__popcll(ballot(in_brick)));
        if (ci != parked.bi || cj != parked.bj || ck != parked.bk) {        // wave-uniform: the brick is not parked yet
            path_stat(4);
            __builtin_amdgcn_wave_barrier();
            int l = lane;
            asm volatile("" : "+v"(l));
            const int slot = (l & 7) + ((l >> 3) & 7) * kBrickPitch;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int t = l + 64 * j;
                const int tx = clampi(ci - 3 + (t & 7), 0, v.nx - 1), ty = clampi(cj - 3 + ((t >> 3) & 7), 0, v.ny - 1),
                          tz = clampi(ck + (t >> 6), 0, v.nz - 1);
                const f4 tv = ldtexel(tex + (unsigned)((tz * v.ny + ty) * v.nx + tx));
                *reinterpret_cast<float4 *>(brick + slot + j * kBrickSlab) = make_float4(tv.x, tv.y, tv.z, tv.w);
            }
            __builtin_amdgcn_wave_barrier();
            parked.bi = ci; parked.bj = cj; parked.bk = ck;
        }
        if (in_brick) {
#if PHOTON_PRIO_BLEND != PHOTON_PRIO_BASE
            __builtin_amdgcn_s_setprio(PHOTON_PRIO_BLEND);     // LDS-fed, like the tile's blend: below the march's base level
#endif
            const f4 *q = brick + (dj * kBrickPitch + di);
            const f4 c00 = lerp4(ldtexel(q), ldtexel(q + 1), a), c10 = lerp4(ldtexel(q + kBrickPitch), ldtexel(q + kBrickPitch + 1), a);
            const f4 c01 = lerp4(ldtexel(q + kBrickSlab), ldtexel(q + kBrickSlab + 1), a),
                     c11 = lerp4(ldtexel(q + kBrickSlab + kBrickPitch), ldtexel(q + kBrickSlab + kBrickPitch + 1), a);
            const f4 c0 = lerp4(c00, c10, b), c1 = lerp4(c01, c11, b);
            acc = lerp4(c0, c1, c);
#if PHOTON_PRIO_BLEND != PHOTON_PRIO_BASE
            __builtin_amdgcn_s_setprio(PHOTON_PRIO_BASE);
#endif
            done = true;
        }
        __builtin_amdgcn_wave_barrier();
        todo = ballot(!done);
    }
    path_stat(5, (unsigned long long)__popcll(ballot(!done)));
    if (!done) acc = linear_gather_fn(tex, v.nx, v.ny, v.nz, x, y, z, v.weight_scale, v.weight_inv);
    return acc;
}

// Loop-invariant, wave-uniform quantities of the march, pinned into SGPRs (readfirstlane): f32
// arithmetic has no scalar ALU on gfx950, so without this the compiler keeps them in VGPRs -- ~20
// registers per lane that cost a whole wave of occupancy (or get spilled into the hot loop).
struct MarchU {
    float minx, miny, minz, maxx, maxy, maxz;   // box
    float sx, sy, sz;                           // 1 / (max - min)
    float nxm2, nym2, nzm2;                     // (float)(n - 2)
    float fnx, fny, fnz;                        // (float)n
    float step, data_min, spin_step;            // h, min(n-1), h / (1 + data_min)
    int nx, ny, nz;
};
__device__ __forceinline__ MarchU make_march_consts(const VolumeDev &v, f3 scale) {
    MarchU u;
    u.minx = uniformf(v.min_bound.x); u.miny = uniformf(v.min_bound.y); u.minz = uniformf(v.min_bound.z);
    u.maxx = uniformf(v.max_bound.x); u.maxy = uniformf(v.max_bound.y); u.maxz = uniformf(v.max_bound.z);
    u.sx = uniformf(scale.x); u.sy = uniformf(scale.y); u.sz = uniformf(scale.z);
    u.nxm2 = uniformf((float)(v.nx - 2)); u.nym2 = uniformf((float)(v.ny - 2)); u.nzm2 = uniformf((float)(v.nz - 2));
    u.fnx = uniformf((float)v.nx); u.fny = uniformf((float)v.ny); u.fnz = uniformf((float)v.nz);
    u.step = uniformf(v.step_size); u.data_min = uniformf(v.data_min);
    u.spin_step = uniformf(v.step_size / (1 + v.data_min));
    u.nx = v.nx; u.ny = v.ny; u.nz = v.nz;
    return u;
}
// calculate_lookup_index / ray_inside_box / access_refractive_index (.h:195-277) on the constants
__device__ __forceinline__ f3 lookup_index_u(f3 pos, const MarchU &u) {
    const f3 off = mk3(pos.x - u.minx, pos.y - u.miny, pos.z - u.minz);
    const f3 fn = mk3(u.sx * off.x, u.sy * off.y, u.sz * off.z);
    return mk3(1 + fn.x * u.nxm2, 1 + fn.y * u.nym2, 1 + fn.z * u.nzm2);
}
// lanes whose lookup index can be sampled (.h:253-277)
__device__ __forceinline__ unsigned long long access_mask(const MarchU &u, f3 l) {
    return ~(ballot(l.x < 0) | ballot(l.y < 0) | ballot(l.z < 0) | ballot(l.x >= u.fnx) | ballot(l.y >= u.fny) | ballot(l.z >= u.fnz));
}
// Lanes inside the box, min <= p < max (.h:217-251; a NaN position passes, as it does in the reference: every
// comparison is false).  The reference's second test (0 <= lookup < n, .h:236-248) cannot fail once the first has passed:
// (p - min) * scale lies in [0, 1 + 2^-23], so lookup = 1 + that * (n - 2) lies in [1, n - 1 + (n - 2) * 2^-23] --
// inside [0, n) for every n the samplers accept.  (The CPU oracle evaluates both.)  For the same reason
// access_refractive_index -- that very test -- can only fail for a ray that is NOT inside: the march evaluates it only for
// rays on their first iteration, whose inside test the reference skips (loop_ctr != 0, .h:1021).
__device__ __forceinline__ unsigned long long inside_mask(f3 p, const MarchU &u) {
    return ~(ballot(p.x < u.minx) | ballot(p.y < u.miny) | ballot(p.z < u.minz) | ballot(p.x >= u.maxx) | ballot(p.y >= u.maxy) | ballot(p.z >= u.maxz));
}

// Statistics.  MarchCount (device_volume.hpp) counts per lane -- photon_trace_volume_rays reports
// the steps of each ray.  The render kernels only need totals: WaveCount keeps them wave-uniform (SALU
// popcount of the lane mask, SGPR accumulators): no VGPR, no VALU instruction in the hot loop.
struct WaveCount { unsigned iterations; unsigned samples; };
__device__ __forceinline__ void count_samples(MarchCount &mc, unsigned long long m) { if (lane_of(m)) mc.samples++; }
__device__ __forceinline__ void count_iterations(MarchCount &mc, unsigned long long m) { if (lane_of(m)) mc.iterations++; }
__device__ __forceinline__ void count_samples(WaveCount &mc, unsigned long long m) { mc.samples += (unsigned)__popcll(m); }
__device__ __forceinline__ void count_iterations(WaveCount &mc, unsigned long long m) { mc.iterations += (unsigned)__popcll(m); }

// val_prev -- the last value a lane sampled, which the trilinear branches fall back on when a blend comes out below the
// volume's minimum (.h:1056-1065): four VGPRs per lane across the march loop (a per-lane LDS slot instead was measured in
// round 4: RK4 461 against 463-468 Mrays/s on C3; dropped).
struct PrevVal {
    f4 v;
    __device__ __forceinline__ void init(f4 *, f4 x) { v = x; }
    __device__ __forceinline__ void set(f4 x) { v = x; }
    __device__ __forceinline__ f4 get() const { return v; }
};

// One cooperative sample + the linear branch's "n-1 below data_min" repair (.h:1056-1065).
template <int INTERP, bool QUANT, class CNT, bool PIN = false>
__device__ __forceinline__ f4 sample_coop(const VolumeDev &v, const f4 *__restrict__ tex, f4 *blk, unsigned long long need, f3 lookup,
                                          const PrevVal &prev, float data_min, CNT &mc, Parked &parked, unsigned tick = 0) {
    f4 val = INTERP == 1 ? tex3d_linear_coop<QUANT, PIN>(v, tex, blk, need, lookup.x, lookup.y, lookup.z, parked, tick)
                         : tex3d_cubic_coop(v, tex, blk, need, lookup.x, lookup.y, lookup.z, parked);
    count_samples(mc, need);
    if (INTERP == 1) {
        const unsigned long long low = need & ballot(val.w < data_min);
        if (low != 0) {                                         // wave-uniform, rare: a blend below the volume's minimum
            const float ambient = 1.000277;
            const f4 val_prev = prev.get();                     // the lane's last sampled value
            const unsigned long long repair = low & ballot(val_prev.w == 0);
            if (repair != 0) {                                  // per lane, out of line: the same blend (the cooperative sampler's
                count_samples(mc, repair);                      // own fallback), not a fourth copy of that sampler in the loop
                if (lane_of(repair)) {
                    const f4 t = linear_gather_fn(tex, v.nx, v.ny, v.nz, lookup.x, lookup.y, lookup.z - 1, v.weight_scale, v.weight_inv);
                    val = f4{t.x, t.y, t.z, ambient - 1};
                }
            }
            if (lane_of(low & ~repair)) val = val_prev;
        }
    }
    return val;
}

// SEGMENTED MARCH (round 4).  A launch may cut every ray's march into segments of at most `max_trips` trips of the loops
// below, handled by different waves at different times (march_kernel.hpp): what a ray carries from one
// segment to the next, besides its position and direction, is exactly the loops' per-lane state -- the completed
// iterations (loop_ctr; "first" = none yet), the `continue` spins, and for the trilinear branches the last value sampled
// (val_prev, the repair of .h:1056-1065).  What is wave-level -- the parked tile, the trip counter -- starts afresh, and
// none of it enters a ray's arithmetic: a segmented march returns the bits of an unsegmented one.
struct MarchResume {
    int loop_ctr, spins;            // per lane
    f4 val_prev;                    // per lane; the trilinear branches only
    unsigned max_trips;             // wave-uniform: trips this call may run (~0u: until every ray has left)
    unsigned trips_base;            // wave-uniform: an upper bound of the trips earlier segments ran (the kLoopMax cap)
    bool fresh;                     // wave-uniform: first segment (the rays enter the volume here)
};
__device__ __forceinline__ MarchResume resume_fresh() { return MarchResume{0, 0, f4{0, 0, 0, 0}, ~0u, 0u, true}; }

// Wave-synchronous RK4 (reference: trace_rays_through_density_gradients.h:952-1291).  All 64
// lanes call it; active_lane = this lane carries a ray that is inside (or entering) the volume.  One
// trip of the loop is one RK4 iteration: Sharma's three samples A, B, C in straight-line code with
// a cooperative sampler call each.  A lane whose ray leaves the box drops out of `active` (the
// reference's `break`); a lane that has to step forward without sampling (the reference's
// `continue`) sits out the rest of the trip and retries on the next one.  The per-ray operation
// order is that of rk4<> in device_volume.hpp.
//
// Shape of the code (round 3): the stage algebra runs UNPREDICATED on every lane -- a lane that does not take part
// computes garbage nobody reads -- and the predicates (active, go, spin, first) are wave masks; only the commit of a
// finished iteration and the rare paths are predicated.
// Returns the mask of the lanes still marching when the call ran out of trips (0 unless rs.max_trips cut it short).
template <int INTERP, bool SAVE, bool QUANT, class CNT>
__device__ __forceinline__ unsigned long long rk4_coop(bool active_lane, f3 &rpos, f3 &rdir, const VolumeDev &v,
                                                       const f4 *__restrict__ tex, f4 *blk, f3 scale, CNT &mc,
                                                       const InterDump &idump, MarchResume &rs) {
    const MarchU u = make_march_consts(v, scale);
    int loop_ctr = rs.loop_ctr;
    constexpr bool kSpinLds = PHOTON_SPINS_IN_LDS && INTERP == 1;
    int spins = rs.spins;
    int *const spin_slot = reinterpret_cast<int *>(blk + tile_texels<1>() + 2 * kBrickSlab) + (threadIdx.x & 63);
    if (kSpinLds) *spin_slot = spins;
    int *const ctr_slot = spin_slot + 64;
    if (kSpinLds) *ctr_slot = loop_ctr;
    unsigned trips = rs.trips_base;                             // wave-uniform; no lane's loop_ctr exceeds it
    const unsigned trips_end = rs.max_trips == ~0u ? ~0u : trips + rs.max_trips;
    Parked parked = parked_none();
    // val_prev: the last value a lane sampled (n-1 form), for the linear branch's repair (PrevVal).  Updated
    // unpredicated; the only live lanes that sit samples out are spinning ones, which have not
    // sampled yet (a ray spins only on its first iteration, see inside_mask) -- theirs is put back to its initial zeros at
    // the end of such a trip.
    PrevVal prev{};
    if (INTERP == 1) prev.init(blk, rs.val_prev);               // the tricubic branches have no such fallback
    unsigned long long active = ballot(active_lane);
    unsigned long long first = active & ballot(loop_ctr == 0);  // lanes with loop_ctr == 0
    while (active != 0 && trips < trips_end) {                  // wave-uniform loop
        // ---------------- sample A at R_n (= rpos) ----------------
        f3 lookup = lookup_index_u(rpos, u);
        const unsigned long long in_a = inside_mask(rpos, u);
        unsigned long long not_over = ~0ull;
        if (trips > (unsigned)kLoopMax) not_over = ~ballot((kSpinLds ? *ctr_slot : loop_ctr) > kLoopMax);       // wave-uniform; a safety cap, never reached
        trips++;
        if (SAVE && INTERP == 1) { if (lane_of(active & not_over)) record_intermediate(idump, kSpinLds ? *ctr_slot : loop_ctr, rpos, rdir); }
        const unsigned long long alive = active & not_over & (in_a | first);           // else the reference's `break`
        unsigned long long go = alive, spin = 0;                // go: lane samples A and -- if that succeeds -- B and C
        if ((alive & ~in_a) != 0) {                             // wave-uniform: rays that start outside the box (.h:1043-1049)
            const unsigned long long access = access_mask(u, lookup);
            go = alive & access;
            spin = alive & ~access;                             // the reference's `continue`: step forward, retry next trip
        }
        f4 val = sample_coop<INTERP, QUANT, CNT, true>(v, tex, blk, go, lookup, prev, u.data_min, mc, parked, trips);
        if (INTERP == 2) {                                      // .h:1220-1227
            const unsigned long long low = go & ballot(val.w < u.data_min);
            spin |= low;
            go &= ~low;
        }
        active = alive;
        first &= ~go;
        if (spin != 0) {                                        // wave-uniform, rare
            if (lane_of(spin)) {
                rpos = rpos + u.spin_step * rdir;
                if (kSpinLds) { spins = *spin_slot + 1; *spin_slot = spins; } else spins++;
            }
            active &= ~(spin & ballot(spins > kSpinMax));
        }
        if (kSpinLds) { if (lane_of(go)) atomicAdd(ctr_slot, 1); } else loop_ctr += lane_of(go) ? 1 : 0;
        const float n_a = val.w + 1;
        const float delta_t = div_nr(u.step, n_a);              // correctly rounded, normal-range form (device_vec.hpp)
        f3 T_n = n_a * rdir;
        const f3 A = delta_t * mk3(n_a * val.x, n_a * val.y, n_a * val.z);
        f3 spos = rpos + (0.5f * delta_t) * T_n + (0.125f * delta_t) * A;       // .h:1088
        if (INTERP == 1) prev.set(f4{val.x, val.y, val.z, n_a - 1});
        // ---------------- sample B ----------------
        lookup = lookup_index_u(spos, u);
        unsigned long long in = inside_mask(spos, u);           // .h:1094-1101: outside = `break`, nothing committed
        active &= ~go | in;
        go &= in;
        val = sample_coop<INTERP, QUANT, CNT, true>(v, tex, blk, go, lookup, prev, u.data_min, mc, parked, trips);
        const float n_b = val.w + 1;
        const f3 B = delta_t * mk3(n_b * val.x, n_b * val.y, n_b * val.z);
        spos = rpos + delta_t * T_n + (0.5f * delta_t) * B;                     // .h:1131
        if (INTERP == 1) prev.set(f4{val.x, val.y, val.z, n_b - 1});
        // ---------------- sample C ----------------
        lookup = lookup_index_u(spos, u);
        in = inside_mask(spos, u);                              // .h:1135-1141
        active &= ~go | in;
        go &= in;
        val = sample_coop<INTERP, QUANT, CNT, true>(v, tex, blk, go, lookup, prev, u.data_min, mc, parked, trips);
        const float n_c = val.w + 1;
        const f3 C = delta_t * mk3(n_c * val.x, n_c * val.y, n_c * val.z);
        if (INTERP == 1) prev.set(f4{val.x, val.y, val.z, n_c - 1});
        if (lane_of(go)) {                                      // the iteration completed: commit
            rpos = rpos + delta_t * (T_n + (float)(1 / 6.0) * (A + 2.0f * B));         // .h:1169
            T_n = T_n + (float)(1 / 6.0) * (A + 4.0f * B + C);                         // .h:1170
            rdir = normalize_nr(div_nr(T_n, INTERP == 1 ? n_a : n_c));                 // .h:1178 / 1276
        }
        count_iterations(mc, go);
        if (INTERP == 1 && spin != 0) { if (lane_of(spin)) prev.set(f4{0, 0, 0, 0}); }
    }
    rs.loop_ctr = kSpinLds ? *ctr_slot : loop_ctr; rs.spins = kSpinLds ? *spin_slot : spins;
    if (INTERP == 1) rs.val_prev = prev.get();
    return active;
}

// Wave-synchronous Euler integrator (reference: .h:743-950): one cooperative sample per trip.  Per-ray operation
// order is that of euler<> in device_volume.hpp.
// Gradient noise of the Euler integrator (.h:853-863): N(0,1)*sigma added to dn/dx, dn/dy.  NOISE is a template
// parameter: with the Philox generator behind a run-time flag the default instantiation spilled 38 VGPRs into its loop.
struct GradNoise { int on; float std; unsigned long long seed, ray_id; };

template <int INTERP, bool SAVE, bool NOISE, bool QUANT, class CNT>
__device__ __forceinline__ unsigned long long euler_coop(bool active_lane, f3 &rpos, f3 &rdir, const VolumeDev &v,
                                                         const f4 *__restrict__ tex, f4 *blk, f3 scale, CNT &mc,
                                                         const GradNoise &gn, const InterDump &idump, MarchResume &rs) {
    const MarchU u = make_march_consts(v, scale);
    int loop_ctr = rs.loop_ctr, spins = rs.spins;
    unsigned trips = rs.trips_base;
    const unsigned trips_end = rs.max_trips == ~0u ? ~0u : trips + rs.max_trips;
    Parked parked = parked_none();
    PrevVal prev{};                                             // val_prev, as in rk4_coop
    if (INTERP == 1) prev.init(blk, rs.val_prev);
    unsigned long long active = ballot(active_lane);
    unsigned long long first = active & ballot(loop_ctr == 0);
    while (active != 0 && trips < trips_end) {
        const f3 lookup = lookup_index_u(rpos, u);
        const unsigned long long in_a = inside_mask(rpos, u);
        unsigned long long not_over = ~0ull;
        if (trips > (unsigned)kLoopMax) not_over = ~ballot(loop_ctr > kLoopMax);
        trips++;
        if (SAVE && INTERP == 1) { if (lane_of(active & not_over)) record_intermediate(idump, loop_ctr, rpos, rdir); }
        const unsigned long long alive = active & not_over & (in_a | first);
        unsigned long long go = alive, spin = 0;
        if (INTERP == 1 && (alive & ~in_a) != 0) {              // only the linear branch guards (.h:821)
            const unsigned long long access = access_mask(u, lookup);
            go = alive & access;
            spin = alive & ~access;
        }
        f4 val = sample_coop<INTERP, QUANT, CNT>(v, tex, blk, go, lookup, prev, u.data_min, mc, parked, trips);
        if (INTERP == 2) {                                      // .h:916-923
            const unsigned long long low = go & ballot(val.w < u.data_min);
            spin |= low;
            go &= ~low;
        }
        active = alive;
        first &= ~go;
        if (spin != 0) {                                        // wave-uniform, rare
            if (lane_of(spin)) {
                rpos = rpos + u.spin_step * rdir;
                spins++;
            }
            active &= ~(spin & ballot(spins > kSpinMax));
        }
        if (INTERP == 1) {
            if (NOISE) {
                float n0, n1;
                photon_normal2(gn.seed, gn.ray_id, (unsigned)loop_ctr, PHOTON_STREAM_NGRAD_NOISE, &n0, &n1);
                val.x += n0 * gn.std;
                val.y += n1 * gn.std;
            }
            if (lane_of(go)) {
                const float current_n = 1 + val.w;
                rdir = rdir + u.step * mk3(val.x, val.y, val.z);               // .h:869 (not renormalised)
                rpos = rpos + div_nr(u.step, current_n) * rdir;                // .h:875
                prev.set(val);
            }
        } else {
            if (lane_of(go)) {
                rdir = normalize_nr(rdir + u.step * mk3(val.x, val.y, val.z));   // .h:931-933
                const float n = 1 + val.w;
                rpos = rpos + div_nr(rdir * u.step, n);                        // .h:939
            }
        }
        loop_ctr += lane_of(go) ? 1 : 0;
        count_iterations(mc, go);
    }
    rs.loop_ctr = loop_ctr; rs.spins = spins;
    if (INTERP == 1) rs.val_prev = prev.get();
    return active;
}

// trace_rays_through_density_gradients (.h:1455-1544), wave-synchronous.  has_ray = this lane
// carries a ray at all (tail lanes of the last workgroup do not).
// rs: where the rays stand (resume_fresh(): at their start, the whole march in one call); for a later segment has_ray =
// this lane's ray was still marching when the previous one ended.  Returns the mask of the lanes still marching.
template <int ALGO, int INTERP, bool SAVE, bool NOISE, bool QUANT, class CNT>
__device__ __forceinline__ unsigned long long trace_volume_coop(bool has_ray, f3 &pos_io, f3 &dir_io, const VolumeDev &v,
                                                                const f4 *__restrict__ tex, f4 *blk, CNT &mc,
                                                                const GradNoise &gn, const InterDump &idump, MarchResume &rs) {
    const f3 mn = v.min_bound, mx = v.max_bound;
    const f3 scale = mk3(1.0f / (mx.x - mn.x), 1.0f / (mx.y - mn.y), 1.0f / (mx.z - mn.z));
    bool active = has_ray;
    if (has_ray && rs.fresh) {
        f3 pos = pos_io;
        const f3 dir = dir_io;
        if (pos.x <= mn.x || pos.y <= mn.y || pos.z <= mn.z || pos.x >= mx.x || pos.y >= mx.y || pos.z >= mx.z) {
            if (!intersect_with_volume(pos, dir, mn, mx)) active = false;      // miss: ray unchanged
        }
        if (active) pos_io = pos;
    }
#if PHOTON_PRIO_BASE > 0
    __builtin_amdgcn_s_setprio(PHOTON_PRIO_BASE);               // the march's own level: its LDS-fed stretches run BELOW it (see PHOTON_PRIO_*)
#endif
    unsigned long long still;
    if (ALGO == 1) still = euler_coop<INTERP, SAVE, NOISE, QUANT, CNT>(active, pos_io, dir_io, v, tex, blk, scale, mc, gn, idump, rs);
    else still = rk4_coop<INTERP, SAVE, QUANT, CNT>(active, pos_io, dir_io, v, tex, blk, scale, mc, idump, rs);
#if PHOTON_PRIO_BASE > 0
    __builtin_amdgcn_s_setprio(0);
#endif
    return still;
}

}  // namespace photon
