// device_vec.hpp - float3/float4 arithmetic for the HIP kernels.
//
// The operation ORDER of every helper is part of the numerical contract: a ray must take the
// same IEEE-754 steps here as in the parity oracle (see include/photon_det_math.h for why).
// The order follows what the reference's helpers do
// (CubicInterpolationCUDA/code/internal/cutil_math_bugfixes.h:300-410): division by a scalar
// is multiplication by its reciprocal, normalize is v * (1/sqrt(v.v)), dot is left to right.
// The library is compiled with -ffp-contract=off; fused multiply-adds appear only where written.
#pragma once
#include <hip/hip_runtime.h>

namespace photon {

struct f3 { float x, y, z; };
struct alignas(16) f4 { float x, y, z, w; };     // 16-byte aligned: one dwordx4 / ds_read_b128 per texel

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { const float inv = 1.0f / s; return a * inv; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 normalize(f3 v) { const float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
__device__ __forceinline__ bool isnan3(f3 v) { return isnan(v.x) || isnan(v.y) || isnan(v.z); }
__device__ __forceinline__ float nanf32() { return __int_as_float(0x7fc00000); }
__device__ __forceinline__ f3 nan3() { const float n = nanf32(); return mk3(n, n, n); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// Lane mask of a predicate.  The builtin takes the i1 itself; HIP's __ballot(int) widens the predicate to an int and
// compares it with zero again -- a v_cndmask + v_cmp pair per call that the march loops issue several times per sample.
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// ---------------------------------------------------------------------------------------------
// Correctly rounded f32 division, reciprocal and square root for operands in the NORMAL range -- the march loops' forms.
// The compiler's expansion of a / b and sqrtf(x) (this library is built with correctly rounded f32 divide and sqrt: the ray's
// rounding sequence is part of the parity contract) wraps its Newton / Markstein core in range scaling: v_div_scale on both
// operands + v_div_fmas (11 instructions per division), a scale test, an unscale and a class test around the square root
// (16).  The scaling only acts when an operand or the result leaves [2^-96, 2^96]; a marching ray divides a step length by a
// refractive index and normalises a direction of length ~n: its operands sit within a few binades of 1.  These forms are the
// compiler's own sequences with the scaling left out -- the SAME instructions on the same values whenever no scaling would
// have happened, hence the same (correctly rounded) bits; v_div_fixup is kept, so zeros, infinities and NaNs come out as the
// full sequence has them.  9 / 8 / 8 instructions; per RK4 iteration (three divisions, one root) 16 fewer: 5.3 per sample.
// NOT for general use: operands or quotients below 2^-96 or above 2^96 in magnitude lose the guarantee.
__device__ __forceinline__ float div_nr(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    r = fmaf(fmaf(-b, r, 1.0f), r, r);
    float q = a * r;
    q = fmaf(fmaf(-b, q, a), r, q);
    q = fmaf(fmaf(-b, q, a), r, q);
    return __builtin_amdgcn_div_fixupf(q, b, a);
}
__device__ __forceinline__ float rcp_nr(float b) {              // 1.0f / b: div_nr with its product a * r = r
    float r = __builtin_amdgcn_rcpf(b);
    r = fmaf(fmaf(-b, r, 1.0f), r, r);
    float q = r;
    q = fmaf(fmaf(-b, q, 1.0f), r, q);
    q = fmaf(fmaf(-b, q, 1.0f), r, q);
    return __builtin_amdgcn_div_fixupf(q, b, 1.0f);
}
__device__ __forceinline__ float sqrt_nr(float x) {             // v_sqrt_f32 (1 ulp), then the neighbour whose square brackets x
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __int_as_float(__float_as_int(s) - 1), up = __int_as_float(__float_as_int(s) + 1);
    const float rd = fmaf(-dn, s, x), ru = fmaf(-up, s, x);
    float t = (0.0f >= rd) ? dn : s;
    t = (0.0f < ru) ? up : t;
    return t;
}
__device__ __forceinline__ f3 div_nr(f3 a, float s) { const float inv = rcp_nr(s); return a * inv; }                 // operator/ above
__device__ __forceinline__ f3 normalize_nr(f3 v) { const float inv = rcp_nr(sqrt_nr(dot(v, v))); return v * inv; }    // normalize above

// 3x3 row-major matrix times vector, each row a left-to-right dot product
__device__ __forceinline__ f3 matvec(const float *m, f3 v) {
    return mk3(dot(mk3(m[0], m[1], m[2]), v), dot(mk3(m[3], m[4], m[5]), v), dot(mk3(m[6], m[7], m[8]), v));
}

}  // namespace photon
