/*
 * photon_philox.h - counter-based Gaussian noise for the optional noise hooks (host C++ and HIP).
 *
 * The reference draws its position / gradient noise from cuRAND XORWOW states seeded with
 * time(NULL) (cuda_codes/parallel_ray_tracing.cu:3405-3445, curand_normal2 at :1428,1610,1777 and
 * trace_rays_through_density_gradients.h:857), i.e. it is irreproducible by construction.  Here
 * the same hooks are fed from Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel Random Numbers:
 * As Easy as 1, 2, 3", SC'11 -- public algorithm, constants below are the paper's), keyed by a user
 * seed and addressed by (ray id, draw index): no per-ray state in memory, any ray can be replayed,
 * and the CPU oracle and the GPU kernels draw the same numbers.
 *
 * photon_normal2(): two N(0,1) variates by Box-Muller on two 32-bit uniforms in (0,1).  The only
 * libm call is log(); its last-ulp differences between glibc and ROCm's ocml move a noise sample
 * by ~1e-16 relative, far below anything the sensor stage can resolve (the noise is applied after
 * the lens, or to a gradient that is integrated over hundreds of steps).
 */
#ifndef PHOTON_PHILOX_H_
#define PHOTON_PHILOX_H_

#include <stdint.h>

#include "photon_det_math.h"

struct photon_u32x4 { uint32_t x, y, z, w; };

PHOTON_HD void photon_philox_round(photon_u32x4 *c, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c->x;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c->z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const photon_u32x4 n = {hi1 ^ c->y ^ k0, lo1, hi0 ^ c->w ^ k1, lo0};
    *c = n;
}

/* Philox4x32-10: counter (ray id lo/hi, draw index, stream id), key = 64-bit seed. */
PHOTON_HD photon_u32x4 photon_philox4x32_10(uint64_t seed, uint64_t ray_id, uint32_t draw, uint32_t stream) {
    photon_u32x4 c = {(uint32_t)ray_id, (uint32_t)(ray_id >> 32), draw, stream};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; r++) {
        photon_philox_round(&c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

/* Two standard normal variates for (seed, ray, draw, stream). */
PHOTON_HD void photon_normal2(uint64_t seed, uint64_t ray_id, uint32_t draw, uint32_t stream, float *n0, float *n1) {
    const photon_u32x4 r = photon_philox4x32_10(seed, ray_id, draw, stream);
    const double u1 = ((double)r.x + 0.5) * (1.0 / 4294967296.0);      /* (0,1): log() is finite */
    const double u2 = ((double)r.y + 0.5) * (1.0 / 4294967296.0);
    const double rad = sqrt(-2.0 * log(u1));
    double s, c;
    photon_det_sincos(2.0 * PHOTON_PI * u2, &s, &c);
    *n0 = (float)(rad * c);
    *n1 = (float)(rad * s);
}

enum { PHOTON_STREAM_POS_NOISE = 1, PHOTON_STREAM_NGRAD_NOISE = 2, PHOTON_STREAM_SCENE = 3, PHOTON_STREAM_IMAGE_NOISE = 4 };

#endif /* PHOTON_PHILOX_H_ */
