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

// 3x3 row-major matrix times vector, each row a left-to-right dot product
__device__ __forceinline__ f3 matvec(const float *m, f3 v) {
    return mk3(dot(mk3(m[0], m[1], m[2]), v), dot(mk3(m[3], m[4], m[5]), v), dot(mk3(m[6], m[7], m[8]), v));
}

}  // namespace photon
