/*
 * photon_det_math.h - bit-reproducible elementary functions (host C++ and HIP device code)
 *
 * Why this exists.  photon's per-ray arithmetic is float32 with catastrophic cancellation in
 * the ray/sphere solve (cuda_codes/parallel_ray_tracing.cu:271-289: gamma = |p-c|^2 - R^2 with
 * both terms ~1e10).  Perturbing a ray by ONE ulp before the lens moves its sensor position by
 * ~1e-3 pixel, coherently for all rays of a source, and changes the rendered image by ~2e-4
 * relative L2 (measured with the oracle: BOS sample, image 1 vs image 2).  The acceptance bar is
 * 1e-5.  Two implementations can therefore only agree if every operation that feeds a ray's
 * position is the same IEEE-754 operation in the same order on both sides.  +,-,*,/,sqrt,fma,
 * floor, conversions are (hipcc keeps f32 divide/sqrt correctly rounded by default; both builds
 * use -ffp-contract=off and spell every fused multiply-add as fma()/fmaf()).  libm functions are
 * not: glibc, ROCm's ocml and CUDA's libdevice round differently.  The reference uses five of
 * them on the ray path (atanf, tanf, cos, sin in ray generation, .cu:123-130,228; acosf for the
 * Mie angle, float3_operators.h:84-90; atan/cos for cos^4, .cu:1467-1472).  This header defines
 * them once, as straight-line double-precision sequences of IEEE operations, so the CPU oracle
 * and the HIP kernels produce identical bits.  Accuracy: double results within a few 1e-16
 * relative, float results = the correctly rounded value in all but ~1e-7 of cases (<= 1 ulp
 * always) -- i.e. at least as accurate as any of the three libm's; tests/test_det_math.py pins
 * them against numpy/glibc.  Neither side of the parity check "owns" this file: it is part of
 * the interface contract, like parallel_ray_tracing.h.
 *
 * All functions propagate NaN, are branch-light and use only: + - * / sqrt fma rint compare.
 */
#ifndef PHOTON_DET_MATH_H_
#define PHOTON_DET_MATH_H_

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PHOTON_HD __host__ __device__ inline
#else
#define PHOTON_HD inline
#endif

#define PHOTON_PI      3.141592653589793238462643383279502884   /* M_PI  */
#define PHOTON_PI_2    1.570796326794896619231321691639751442   /* M_PI_2 */
#define PHOTON_PI_4    0.785398163397448309615660845819875721
/* pi/2 split for Cody-Waite reduction: HI has 33 significant bits, HI+LO = pi/2 to ~1e-22 */
#define PHOTON_PIO2_HI 1.57079632673412561417e+00
#define PHOTON_PIO2_LO 6.07710050650619224932e-11
#define PHOTON_2_PI    0.636619772367581343075535053490057448   /* 2/pi */

/* sin(y), cos(y) for |y| <= pi/4 by Taylor series (truncation < 1e-20), Horner with fma. */
PHOTON_HD void photon_det_sincos_kernel(double y, double *s, double *c) {
    const double z = y * y;
    double ps = -1.0 / 51090942171709440000.0;             /* -1/21! */
    ps = fma(ps, z, 1.0 / 121645100408832000.0);           /*  1/19! */
    ps = fma(ps, z, -1.0 / 355687428096000.0);             /* -1/17! */
    ps = fma(ps, z, 1.0 / 1307674368000.0);                /*  1/15! */
    ps = fma(ps, z, -1.0 / 6227020800.0);                  /* -1/13! */
    ps = fma(ps, z, 1.0 / 39916800.0);                     /*  1/11! */
    ps = fma(ps, z, -1.0 / 362880.0);                      /* -1/9!  */
    ps = fma(ps, z, 1.0 / 5040.0);                         /*  1/7!  */
    ps = fma(ps, z, -1.0 / 120.0);                         /* -1/5!  */
    ps = fma(ps, z, 1.0 / 6.0);                            /*  1/3!  */
    *s = fma(-(y * z), ps, y);                             /* y - y^3 * (...) */
    double pc = 1.0 / 2432902008176640000.0;               /*  1/20! */
    pc = fma(pc, z, -1.0 / 6402373705728000.0);            /* -1/18! */
    pc = fma(pc, z, 1.0 / 20922789888000.0);               /*  1/16! */
    pc = fma(pc, z, -1.0 / 87178291200.0);                 /* -1/14! */
    pc = fma(pc, z, 1.0 / 479001600.0);                    /*  1/12! */
    pc = fma(pc, z, -1.0 / 3628800.0);                     /* -1/10! */
    pc = fma(pc, z, 1.0 / 40320.0);                        /*  1/8!  */
    pc = fma(pc, z, -1.0 / 720.0);                         /* -1/6!  */
    pc = fma(pc, z, 1.0 / 24.0);                           /*  1/4!  */
    pc = fma(pc, z, -0.5);                                 /* -1/2!  */
    *c = fma(z, pc, 1.0);
}

/* sin and cos of a double angle (accurate for |t| up to ~1e5; the callers pass [0, 2*pi]). */
PHOTON_HD void photon_det_sincos(double t, double *s, double *c) {
    const double k = rint(t * PHOTON_2_PI);
    double y = fma(-k, PHOTON_PIO2_HI, t);
    y = fma(-k, PHOTON_PIO2_LO, y);
    double sy, cy;
    photon_det_sincos_kernel(y, &sy, &cy);
    const long long q = (long long)k & 3;
    double ss = sy, cc = cy;
    if (q == 1) { ss = cy; cc = -sy; }
    else if (q == 2) { ss = -sy; cc = -cy; }
    else if (q == 3) { ss = -cy; cc = sy; }
    *s = ss;
    *c = cc;
}
PHOTON_HD double photon_det_sin(double t) { double s, c; photon_det_sincos(t, &s, &c); return s; }
PHOTON_HD double photon_det_cos(double t) { double s, c; photon_det_sincos(t, &s, &c); return c; }

/* tan(x), any finite x (callers pass atanf results, |x| < pi/2). */
PHOTON_HD double photon_det_tan(double x) {
    const double k = rint(x * PHOTON_2_PI);
    double y = fma(-k, PHOTON_PIO2_HI, x);
    y = fma(-k, PHOTON_PIO2_LO, y);
    double sy, cy;
    photon_det_sincos_kernel(y, &sy, &cy);
    const long long q = (long long)k & 1;
    return q ? -(cy / sy) : (sy / cy);
}

/* atan(x): |x|>1 -> pi/2 - atan(1/x); two half-angle steps atan(a) = 2 atan(a/(1+sqrt(1+a^2)))
 * bring a below tan(pi/16) = 0.199; odd Taylor series to a^29 (truncation < 1e-21). */
PHOTON_HD double photon_det_atan(double x) {
    const bool neg = x < 0.0;
    double a = neg ? -x : x;
    const bool inv = a > 1.0;
    if (inv) a = 1.0 / a;
    a = a / (1.0 + sqrt(fma(a, a, 1.0)));
    a = a / (1.0 + sqrt(fma(a, a, 1.0)));
    const double z = a * a;
    double p = 1.0 / 29.0;
    p = fma(p, -z, 1.0 / 27.0);
    p = fma(p, -z, 1.0 / 25.0);
    p = fma(p, -z, 1.0 / 23.0);
    p = fma(p, -z, 1.0 / 21.0);
    p = fma(p, -z, 1.0 / 19.0);
    p = fma(p, -z, 1.0 / 17.0);
    p = fma(p, -z, 1.0 / 15.0);
    p = fma(p, -z, 1.0 / 13.0);
    p = fma(p, -z, 1.0 / 11.0);
    p = fma(p, -z, 1.0 / 9.0);
    p = fma(p, -z, 1.0 / 7.0);
    p = fma(p, -z, 1.0 / 5.0);
    p = fma(p, -z, 1.0 / 3.0);
    p = fma(p, -z, 1.0);
    double r = 4.0 * (a * p);
    if (inv) r = PHOTON_PI_2 - r;
    return neg ? -r : r;
}

/* acos(x) = 2 atan(sqrt((1-x)/(1+x))) on [-1,1]; NaN outside. */
PHOTON_HD double photon_det_acos(double x) {
    if (!(x >= -1.0 && x <= 1.0)) return (double)NAN;
    return 2.0 * photon_det_atan(sqrt((1.0 - x) / (1.0 + x)));
}

/* float-in / float-out forms: evaluate in double, round once. */
/* exp(x), used by the on-device scene generators (laser-sheet profile, synthetic density fields):
 * x = k ln2 + r with |r| <= ln2/2 (Cody-Waite, ln2 split so that k*LN2_HI is exact for |k| < 2^20),
 * exp(r) by its Taylor series to r^14/14! (truncation < 2e-19 relative), scaled by 2^k through the
 * exponent field.  Underflows to 0 below -708, overflows to +inf above 709; NaN propagates. */
#define PHOTON_LN2_HI 6.93147180369123816490e-01
#define PHOTON_LN2_LO 1.90821492927058770002e-10
#define PHOTON_1_LN2  1.44269504088896338700e+00
PHOTON_HD double photon_det_exp(double x) {
    if (x != x) return x;
    if (x > 709.0) return 1.0 / 0.0;
    if (x < -708.0) return 0.0;
    const double k = rint(x * PHOTON_1_LN2);
    double r = fma(-k, PHOTON_LN2_HI, x);
    r = fma(-k, PHOTON_LN2_LO, r);
    double p = 1.0 / 87178291200.0;                         /* 1/14! */
    p = fma(p, r, 1.0 / 6227020800.0);                     /* 1/13! */
    p = fma(p, r, 1.0 / 479001600.0);                      /* 1/12! */
    p = fma(p, r, 1.0 / 39916800.0);                       /* 1/11! */
    p = fma(p, r, 1.0 / 3628800.0);                        /* 1/10! */
    p = fma(p, r, 1.0 / 362880.0);                         /* 1/9!  */
    p = fma(p, r, 1.0 / 40320.0);                          /* 1/8!  */
    p = fma(p, r, 1.0 / 5040.0);                           /* 1/7!  */
    p = fma(p, r, 1.0 / 720.0);                            /* 1/6!  */
    p = fma(p, r, 1.0 / 120.0);                            /* 1/5!  */
    p = fma(p, r, 1.0 / 24.0);                             /* 1/4!  */
    p = fma(p, r, 1.0 / 6.0);                              /* 1/3!  */
    p = fma(p, r, 0.5);                                    /* 1/2!  */
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    /* 2^k: two factors so that the scaling itself never over/underflows prematurely */
    const long long ki = (long long)k;
    const long long k1 = ki / 2, k2 = ki - k1;
    union { unsigned long long u; double d; } a, b;
    a.u = (unsigned long long)(k1 + 1023) << 52;
    b.u = (unsigned long long)(k2 + 1023) << 52;
    return p * a.d * b.d;
}

PHOTON_HD float photon_det_atanf(float x) { return (float)photon_det_atan((double)x); }
PHOTON_HD float photon_det_tanf(float x) { return (float)photon_det_tan((double)x); }
PHOTON_HD float photon_det_acosf(float x) { return (float)photon_det_acos((double)x); }
PHOTON_HD float photon_det_cosf(float x) { return (float)photon_det_cos((double)x); }
PHOTON_HD float photon_det_sinf(float x) { return (float)photon_det_sin((double)x); }

#endif /* PHOTON_DET_MATH_H_ */
