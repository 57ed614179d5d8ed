// photon_oracle.cpp - CPU restatement of photon's ray-tracing core.
//
// TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the CPU baseline for the
// MI355X build of libparallel_ray_tracing.so.  Nothing in photon_amd/ links, loads or
// calls it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// It restates, in scalar C++ with the same float/double placement, the arithmetic of the
// reference CUDA code (paths relative to /root/reference):
//   cuda_codes/parallel_ray_tracing.cu                  (ray generation, lens, sensor, host loop)
//   cuda_codes/trace_rays_through_density_gradients.h   (volume build, bounds, euler, rk4)
//   cuda_codes/float3_operators.h, CubicInterpolationCUDA/code/internal/*  (helpers, B-spline)
// Each function cites the lines it follows.
//
// Pinning: the reference has no tests or golden vectors and its GPU
// path cannot be built here (no nvcc, texture references, cutil.h, libteem).  The oracle is
// pinned by (1) fixtures generated in this container by the reference's own Python
// (tests/golden/make_golden.py: ABI inputs, float64 numpy ancestors of the lens functions),
// (2) known answers implied by the reference's formulas (glibc srand(10) sequence,
// constant-gradient deflection, interpolation-at-knots, thin-lens magnification).
// Two pieces of third-party arithmetic are NOT in the repository and stay "parity unpinned":
// NVIDIA texture-unit filtering (we use exact f32 lerps; PHOTON_TEX_FRAC_BITS=8 emulates the
// documented 8-bit weights) and teem's nrrdLoad (we parse the NRRD header ourselves).
//
// Deliberate deviations from the literal CUDA text (same in the product, listed in DESIGN.md):
//   * reads the reference performs out of bounds are clamped / skipped (Mie row past the end,
//     negative pixel index in the 4-pixel splat);
//   * loops that can spin forever in the reference (ray never enters the box) are capped;
//   * sensor accumulation is summed in double per pixel (order-free expectation of the
//     reference's float atomicAdd in arbitrary order) and rounded to float once;
//   * the noise hooks draw from a seeded counter-based generator (include/photon_philox.h)
//     instead of time(NULL)-seeded cuRAND states: same hooks, reproducible numbers.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp, no fast-math).

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../include/parallel_ray_tracing.h"
#include "../include/photon_det_math.h"   // bit-reproducible atan/tan/sin/cos/acos (see its header)
#include "../include/photon_philox.h"     // counter-based N(0,1) for the optional noise hooks

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------
// float3 / float4 helpers with the exact operation order of cutil_math
// (CubicInterpolationCUDA/code/internal/cutil_math_bugfixes.h:300-410) and
// cuda_codes/float3_operators.h:46-90.
// ---------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

inline f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
inline f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
inline f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
inline f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
inline f3 operator/(f3 a, float s) { float inv = 1.0f / s; return a * inv; }   // :349-353
inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }     // :387
inline f3 normalize(f3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }  // :405
inline bool isnan3(f3 v) { return std::isnan(v.x) || std::isnan(v.y) || std::isnan(v.z); }
const float NANF = std::nanf("");

struct Ray {            // light_ray_data_t, parallel_ray_tracing.h:102-112
    f3 pos, dir;
    float wavelength;
    double radiance;
};

inline void kill(Ray &r) {
    r.pos = mk3(NANF, NANF, NANF);
    r.dir = mk3(NANF, NANF, NANF);
    r.wavelength = NANF;
    r.radiance = std::nan("");
}

// ---------------------------------------------------------------------------------------
// Volume (density_grad_params_t + the texture it is bound to)
// ---------------------------------------------------------------------------------------
struct Volume {
    f3 min_bound, max_bound;
    int nx = 0, ny = 0, nz = 0;
    f3 grid_spacing;
    float step_size = 0, data_min = 0;
    int interpolation = 1;          // 1 linear, 2 cubic
    int frac_bits = 0;              // 0 = exact f32 weights, 8 = NVIDIA texture-unit emulation
    std::vector<f4> data;           // grad n (x,y,z), n-1 (w); x fastest
    std::vector<f4> coeffs;         // B-spline coefficients when interpolation == 2
};

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline float lerpf(float a, float b, float t) { return fmaf(t, b - a, a); }
inline f4 lerp4(f4 a, f4 b, float t) {
    return f4{lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t), lerpf(a.w, b.w, t)};
}
inline float quant(float a, int bits) {
    if (bits <= 0) return a;
    const float s = (float)(1 << bits);
    return floorf(a * s + 0.5f) / s;
}

// tex3D(tex_data, x, y, z): unnormalised coordinates, linear filter, clamp addressing
// (trace_rays_through_density_gradients.h:1628-1631).  CUDA programming guide, "Linear
// filtering": xB = x - 0.5, i = floor(xB), alpha = frac(xB), texels i and i+1 clamped.
inline f4 tex3d_linear(const Volume &v, const std::vector<f4> &t, float x, float y, float z) {
    const float xb = x - 0.5f, yb = y - 0.5f, zb = z - 0.5f;
    const float fi = floorf(xb), fj = floorf(yb), fk = floorf(zb);
    const float a = quant(xb - fi, v.frac_bits), b = quant(yb - fj, v.frac_bits),
                c = quant(zb - fk, v.frac_bits);
    const int i0 = clampi((int)fi, 0, v.nx - 1), i1 = clampi((int)fi + 1, 0, v.nx - 1);
    const int j0 = clampi((int)fj, 0, v.ny - 1), j1 = clampi((int)fj + 1, 0, v.ny - 1);
    const int k0 = clampi((int)fk, 0, v.nz - 1), k1 = clampi((int)fk + 1, 0, v.nz - 1);
    const size_t W = v.nx, WH = (size_t)v.nx * v.ny;
    auto at = [&](int i, int j, int k) -> const f4 & { return t[k * WH + j * W + i]; };
    const f4 c00 = lerp4(at(i0, j0, k0), at(i1, j0, k0), a);
    const f4 c10 = lerp4(at(i0, j1, k0), at(i1, j1, k0), a);
    const f4 c01 = lerp4(at(i0, j0, k1), at(i1, j0, k1), a);
    const f4 c11 = lerp4(at(i0, j1, k1), at(i1, j1, k1), a);
    const f4 c0 = lerp4(c00, c10, b), c1 = lerp4(c01, c11, b);
    return lerp4(c0, c1, c);
}

// bspline_weights, CubicInterpolationCUDA/code/internal/bspline_kernel.cu:83-94.  w0, w3 as spelled there; the middle
// weights 2/3 - f^2 (2 - f) / 2 in the Horner form 2/3 + f^2 (f / 2 - 1) with two fused multiply-adds (round 3; rounds
// 1-2 spelled them literally: four roundings instead of two).  This is the DEFINED form the HIP kernels reproduce bit
// for bit; it is pinned against an f64 evaluation of the same 64-tap sum in tests/test_oracle_golden.py.  The
// reference never executes its tricubic path (interpolation_scheme hard-wired to 1), so no reference bits exist.
inline void bspline_weights(float f, float &w0, float &w1, float &w2, float &w3) {
    const float one_frac = 1.0f - f;
    const float squared = f * f;
    const float one_sqd = one_frac * one_frac;
    w0 = 1.0f / 6.0f * one_sqd * one_frac;
#ifdef PHOTON_ORACLE_LITERAL_BSPLINE
    // the reference's own spelling (bspline_kernel.cu:90-91), four roundings per middle weight: the `literal` build of this
    // file (oracle/Makefile), a sensitivity probe for tests/test_oracle_golden.py::test_bspline_weight_form_sensitivity --
    // never the parity reference
    w1 = 2.0f / 3.0f - 0.5f * squared * (2.0f - f);
    w2 = 2.0f / 3.0f - 0.5f * one_sqd * (2.0f - one_frac);
#else
    w1 = fmaf(squared, fmaf(0.5f, f, -1.0f), 2.0f / 3.0f);
    w2 = fmaf(one_sqd, fmaf(0.5f, one_frac, -1.0f), 2.0f / 3.0f);
#endif
    w3 = 1.0f / 6.0f * squared * f;
}

// cubicTex3D: tricubic B-spline lookup on the prefiltered coefficient texture.
// Reference fast form = 8 trilinear fetches through the texture unit (cubicTex3D_kernel.cu:48-81); its exact
// equivalent is the 64-tap sum over texels index-1..index+2 with clamp addressing, weights B(x) B(y) B(z)
// (cubicTex3D.cu:63-90, examples/referenceCubicTexture3D/cubicFilter3D_kernel.hpp).
// We evaluate that 64-tap sum in a DEFINED order, so that the HIP kernels reproduce it bit for bit -- the slab order:
//   wxy[b][a] = wx[a] * wy[b]                                   (16 products)
//   s_c       = sum over b, then a, of wxy[b][a] * T[a, b, c]   (per z-slab: a product, then 15 fmaf; x innermost)
//   result    = wz[0] * s_0, then fmaf(wz[c], s_c, result)      (z pass)
// (round 1 used the fully separable x, y, z order: 336 multiply-adds per sample against 288 here; the two differ by
// reassociation only, a few ulp -- cubicTex3D.cu's own simple form associates as bx * (by * bz) per tap.)
inline f4 tex3d_cubic(const Volume &v, float x, float y, float z) {
    const float xg = x - 0.5f, yg = y - 0.5f, zg = z - 0.5f;
    const float fi = floorf(xg), fj = floorf(yg), fk = floorf(zg);
    float wx[4], wy[4], wz[4];
    bspline_weights(xg - fi, wx[0], wx[1], wx[2], wx[3]);
    bspline_weights(yg - fj, wy[0], wy[1], wy[2], wy[3]);
    bspline_weights(zg - fk, wz[0], wz[1], wz[2], wz[3]);
    int ix[4], iy[4], iz[4];
    for (int a = 0; a < 4; a++) {
        ix[a] = clampi((int)fi - 1 + a, 0, v.nx - 1);
        iy[a] = clampi((int)fj - 1 + a, 0, v.ny - 1);
        iz[a] = clampi((int)fk - 1 + a, 0, v.nz - 1);
    }
    const size_t W = v.nx, WH = (size_t)v.nx * v.ny;
    float wxy[4][4];
    for (int b = 0; b < 4; b++)
        for (int a = 0; a < 4; a++) wxy[b][a] = wx[a] * wy[b];
    float acc[4] = {0, 0, 0, 0};
    for (int c = 0; c < 4; c++) {
        float s[4] = {0, 0, 0, 0};
        for (int b = 0; b < 4; b++) {
            const f4 *row = &v.coeffs[iz[c] * WH + iy[b] * W];
            for (int a = 0; a < 4; a++) {
                const f4 &t = row[ix[a]];
                const float w = wxy[b][a];
                if (a == 0 && b == 0) { s[0] = w * t.x; s[1] = w * t.y; s[2] = w * t.z; s[3] = w * t.w; }
                else { s[0] = fmaf(w, t.x, s[0]); s[1] = fmaf(w, t.y, s[1]); s[2] = fmaf(w, t.z, s[2]); s[3] = fmaf(w, t.w, s[3]); }
            }
        }
        for (int q = 0; q < 4; q++) acc[q] = (c == 0) ? wz[0] * s[q] : fmaf(wz[c], s[q], acc[q]);
    }
    return f4{acc[0], acc[1], acc[2], acc[3]};
}

// ConvertToInterpolationCoefficients on one line of one channel, in place.
// CubicInterpolationCUDA/code/internal/cubicPrefilter_kernel.cu:52-112.
void prefilter_line(float *c, int n, size_t stride /* in floats */) {
    const float Pole = sqrtf(3.0f) - 2.0f;
    const float Lambda = (1.0f - Pole) * (1.0f - 1.0f / Pole);
    // causal initialisation (:56-72): horizon min(12, n), clamping boundary
    const int horizon = n < 12 ? n : 12;
    float zn = Pole;
    float sum = c[0];
    for (int k = 0; k < horizon; k++) {
        sum += zn * c[k * stride];
        zn *= Pole;
    }
    float prev = Lambda * sum;
    c[0] = prev;
    for (int k = 1; k < n; k++) {                       // causal recursion (:99-102)
        prev = Lambda * c[k * stride] + Pole * prev;
        c[k * stride] = prev;
    }
    prev = (Pole / (Pole - 1.0f)) * c[(size_t)(n - 1) * stride];   // anticausal init (:75-83)
    c[(size_t)(n - 1) * stride] = prev;
    for (int k = n - 2; k >= 0; k--) {                  // anticausal recursion (:106-110)
        prev = Pole * (prev - c[k * stride]);
        c[k * stride] = prev;
    }
}

// CubicBSplinePrefilter3D (cubicPrefilter3D.cu:54-153): x lines, then y, then z -- applied
// to each of the four channels of the float4 volume.  (The reference wires the float4 array
// through the scalar filter, trace_rays_through_density_gradients.h:1654-1655, which filters
// interleaved channels; SURVEY.md section 2a: we build the intended per-channel filter.)
void prefilter_volume(Volume &v) {
    v.coeffs = v.data;
    float *base = reinterpret_cast<float *>(v.coeffs.data());
    const int nx = v.nx, ny = v.ny, nz = v.nz;
    const size_t sx = 4, sy = 4 * (size_t)nx, sz = 4 * (size_t)nx * ny;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int ch = 0; ch < 4; ch++) prefilter_line(base + k * sz + j * sy + ch, nx, sx);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz; k++)
        for (int i = 0; i < nx; i++)
            for (int ch = 0; ch < 4; ch++) prefilter_line(base + k * sz + i * sx + ch, ny, sy);
#pragma omp parallel for collapse(2) schedule(static)
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++)
            for (int ch = 0; ch < 4; ch++) prefilter_line(base + j * sy + i * sx + ch, nz, sz);
}

inline f4 sample(const Volume &v, f3 lookup) {
    return v.interpolation == 2 ? tex3d_cubic(v, lookup.x, lookup.y, lookup.z)
                                : tex3d_linear(v, v.data, lookup.x, lookup.y, lookup.z);
}

// setData, trace_rays_through_density_gradients.h:1820-2002: refractive-index gradient by
// one-sided (edges, double arithmetic) / central (interior) differences; .w = n - 1.
// `n1` is K*rho as produced by loadNRRD (:1729-1748).
void build_gradient_volume(Volume &v, const std::vector<float> &n1) {
    const int W = v.nx, H = v.ny, D = v.nz;
    const float gx = v.grid_spacing.x, gy = v.grid_spacing.y, gz = v.grid_spacing.z;
    v.data.assign((size_t)W * H * D, f4{0, 0, 0, 0});
    float data_min = FLT_MAX;
    for (size_t i = 0; i < n1.size(); i++)
        if (n1[i] < data_min) data_min = n1[i];
    v.data_min = data_min;
    const size_t WH = (size_t)W * H;
#pragma omp parallel for schedule(static)
    for (int z = 0; z < D; z++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                auto d = [&](int xx, int yy, int zz) { return n1[zz * WH + (size_t)yy * W + xx]; };
                float nxv, nyv, nzv, s1, s2, s3;
                if (x < 1) {                                              // :1874-1884
                    s1 = d(x, y, z); s2 = d(x + 1, y, z); s3 = d(x + 2, y, z);
                    nxv = (float)((-3.0 / 2 * s1 + 2 * s2 - 1.0 / 2 * s3) / gx);
                } else if (x >= W - 1) {                                  // :1885-1895
                    s1 = d(x, y, z); s2 = d(x - 1, y, z); s3 = d(x - 2, y, z);
                    nxv = (float)((3.0 / 2 * s1 - 2 * s2 + 1.0 / 2 * s3) / gx);
                } else {                                                  // :1896-1904
                    s1 = d(x - 1, y, z); s2 = d(x + 1, y, z);
                    nxv = (s2 - s1) / (2 * gx);
                }
                if (y < 1) {                                              // :1909-1919
                    s1 = d(x, y, z); s2 = d(x, y + 1, z); s3 = d(x, y + 2, z);
                    nyv = (float)((-3.0 / 2 * s1 + 2 * s2 - 1.0 / 2 * s3) / gy);
                } else if (y >= H - 1) {
                    s1 = d(x, y, z); s2 = d(x, y - 1, z); s3 = d(x, y - 2, z);
                    nyv = (float)((3.0 / 2 * s1 - 2 * s2 + 1.0 / 2 * s3) / gy);
                } else {                                                  // :1931-1939 (2.0: double)
                    s1 = d(x, y - 1, z); s2 = d(x, y + 1, z);
                    nyv = (float)((s2 - s1) / (2.0 * gy));
                }
                if (z < 1) {                                              // :1945-1957
                    s1 = d(x, y, z); s2 = d(x, y, z + 1); s3 = d(x, y, z + 2);
                    nzv = (float)((-3.0 / 2 * s1 + 2 * s2 - 1.0 / 2 * s3) / gz);
                } else if (z >= D - 1) {
                    s1 = d(x, y, z); s2 = d(x, y, z - 1); s3 = d(x, y, z - 2);
                    nzv = (float)((3.0 / 2 * s1 - 2 * s2 + 1.0 / 2 * s3) / gz);
                } else {
                    s1 = d(x, y, z - 1); s2 = d(x, y, z + 1);
                    nzv = (s2 - s1) / (2 * gz);
                }
                v.data[z * WH + (size_t)y * W + x] = f4{nxv, nyv, nzv, d(x, y, z)};
            }
}

// readDatafromFile + loadNRRD (.h:1663-1817, :2004-2105) given the parsed header values.
void setup_volume(Volume &v, const float *rho, int nx, int ny, int nz, const double spacing[3],
                  const double origin[3], int interpolation, int frac_bits) {
    const double xmin = origin[0], ymin = origin[1], zmin = origin[2] - 750e3;     // :1696-1706
    const double xmax = xmin + (nx - 1) * spacing[0];
    const double ymax = ymin + (ny - 1) * spacing[1];
    const double zmax = zmin + (nz - 1) * spacing[2];
    // "not sure what these statements do" (:1714-1717): depth is capped at 1024 slices,
    // after the bounds were computed from the file's own size
    int data_max = 1024;
    if (data_max > nz) data_max = nz;
    if (nz > data_max) nz = data_max;
    v.min_bound = mk3((float)xmin, (float)ymin, (float)zmin);                      // :2065-2066
    v.max_bound = mk3((float)xmax, (float)ymax, (float)zmax);
    v.nx = nx; v.ny = ny; v.nz = nz;
    v.grid_spacing = mk3((float)spacing[0], (float)spacing[1], (float)spacing[2]);
    v.interpolation = interpolation;
    v.frac_bits = frac_bits;
    const float K = 0.225e-3;                                                      // :1729
    std::vector<float> n1((size_t)nx * ny * nz);
    for (size_t i = 0; i < n1.size(); i++) n1[i] = K * (rho[i] * 1.0f);            // :1743-1748
    build_gradient_volume(v, n1);
    float step = (float)fmin(spacing[0], spacing[1]);                              // :2086-2098
    step = step < spacing[2] ? step : (float)spacing[2];
    v.step_size = step;
    if (interpolation == 2) prefilter_volume(v);
}

// Minimal NRRD reader: what teem's nrrdLoad yields for the fields loadNRRD reads
// (.h:1687-1706): sizes, spacings, space origin, raw little-endian float payload.
bool read_nrrd(const char *path, std::vector<float> &rho, int dims[3], double spacing[3],
               double origin[3]) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string line;
    if (!std::getline(f, line) || line.compare(0, 4, "NRRD") != 0) return false;
    bool have_sizes = false;
    spacing[0] = spacing[1] = spacing[2] = 1.0;
    origin[0] = origin[1] = origin[2] = 0.0;
    std::string type, encoding = "raw", endian = "little";
    int dimension = 0;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) break;
        if (line[0] == '#') continue;
        const size_t c = line.find(':');
        if (c == std::string::npos) continue;
        std::string key = line.substr(0, c), val = line.substr(c + 1);
        while (!val.empty() && (val[0] == ' ' || val[0] == '=')) val.erase(0, 1);
        if (key == "type") type = val;
        else if (key == "dimension") dimension = atoi(val.c_str());
        else if (key == "sizes") have_sizes = sscanf(val.c_str(), "%d %d %d", &dims[0], &dims[1], &dims[2]) == 3;
        else if (key == "spacings") sscanf(val.c_str(), "%lf %lf %lf", &spacing[0], &spacing[1], &spacing[2]);
        else if (key == "space origin") sscanf(val.c_str(), " (%lf,%lf,%lf)", &origin[0], &origin[1], &origin[2]);
        else if (key == "encoding") encoding = val;
        else if (key == "endian") endian = val;
        else if (key == "space directions") {
            double m[9];
            if (sscanf(val.c_str(), " (%lf,%lf,%lf) (%lf,%lf,%lf) (%lf,%lf,%lf)", &m[0], &m[1], &m[2],
                       &m[3], &m[4], &m[5], &m[6], &m[7], &m[8]) == 9) {
                spacing[0] = sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
                spacing[1] = sqrt(m[3] * m[3] + m[4] * m[4] + m[5] * m[5]);
                spacing[2] = sqrt(m[6] * m[6] + m[7] * m[7] + m[8] * m[8]);
            }
        }
    }
    if (!have_sizes || dimension != 3 || type != "float" || encoding != "raw" || endian != "little")
        return false;
    rho.resize((size_t)dims[0] * dims[1] * dims[2]);
    f.read(reinterpret_cast<char *>(rho.data()), rho.size() * sizeof(float));
    return (size_t)f.gcount() == rho.size() * sizeof(float);
}

// ---------------------------------------------------------------------------------------
// Volume traversal
// ---------------------------------------------------------------------------------------

// IntersectWithVolume, trace_rays_through_density_gradients.h:100-186 (restated literally,
// including the z-slab asymmetry at :168 and the "ray starts in box" branch at :176-177).
bool intersect_with_volume(f3 &pos, f3 dir, f3 p1, f3 p2) {
    float tnear = -(FLT_MAX - 1);
    float tfar = FLT_MAX;
    float t1 = (p1.x - pos.x) / dir.x, t2 = (p2.x - pos.x) / dir.x;
    if (t1 > t2) { float t = t1; t1 = t2; t2 = t; }
    if (t1 > tnear) tnear = t1;
    if (t2 < tfar) tfar = t2;
    if (tnear > tfar) return false;
    if (tfar < 0.0) return false;
    t1 = (p1.y - pos.y) / dir.y; t2 = (p2.y - pos.y) / dir.y;
    if (t1 > t2) { float t = t1; t1 = t2; t2 = t; }
    if (t1 > tnear) tnear = t1;
    if (t2 < tfar) tfar = t2;
    if (tnear > tfar) return false;
    if (tfar < 0.0) return false;
    t1 = (p1.z - pos.z) / dir.z; t2 = (p2.z - pos.z) / dir.z;
    if (t1 > t2) { float t = t1; t1 = t2; t2 = t; }
    float t;
    if (t1 >= 0 && t1 > tnear) tnear = t1;
    if (t2 < tfar) tfar = t2;
    if (tnear > tfar) return false;
    if (tfar < 0.0) return false;
    else if (tnear < 0) t = tfar;
    else t = tnear;
    pos.x += dir.x * t; pos.y += dir.y * t; pos.z += dir.z * t;
    return true;
}

// calculate_lookup_index, .h:195-215 ("1 + f*(N-2)": int literals -> all-float arithmetic)
inline f3 lookup_index(f3 pos, const Volume &v, f3 scale) {
    const f3 off = pos - v.min_bound;
    const f3 fn = mk3(scale.x * off.x, scale.y * off.y, scale.z * off.z);
    return mk3(1 + fn.x * (v.nx - 2), 1 + fn.y * (v.ny - 2), 1 + fn.z * (v.nz - 2));
}
// ray_inside_box, .h:217-251
inline bool inside_box(f3 p, const Volume &v, f3 l) {
    if (p.x < v.min_bound.x || p.y < v.min_bound.y || p.z < v.min_bound.z ||
        p.x >= v.max_bound.x || p.y >= v.max_bound.y || p.z >= v.max_bound.z) return false;
    if (l.x < 0 || l.y < 0 || l.z < 0 || l.x >= v.nx || l.y >= v.ny || l.z >= v.nz) return false;
    return true;
}
// access_refractive_index, .h:253-277
inline bool can_access(const Volume &v, f3 l) {
    return !(l.x < 0 || l.y < 0 || l.z < 0 || l.x >= v.nx || l.y >= v.ny || l.z >= v.nz);
}

const int LOOP_MAX = 10000000;          // loop_ctr_max = 1e7 (.h:765,981)
const int SPIN_MAX = 1 << 20;           // our cap on the reference's uncounted `continue` spins

struct MarchCount { int iterations = 0; int samples = 0; };

struct Noise {                      // the four noise arguments of start_ray_tracing + our seed
    bool add_pos = false, add_ngrad = false;
    float pos_std = 0.f, ngrad_std = 0.f;
    uint64_t seed = 0;
};
uint64_t g_noise_seed = 0;          // oracle_set_noise_seed()

// Intermediate dumps (save_intermediate_ray_data): one ray's row of `slots` positions / directions.
// Only the trilinear branches of euler / rk4 record them (.h:784-790, 1004-1008).
struct InterRec { f3 *pos = nullptr, *dir = nullptr; int slots = 0; };
inline void record_intermediate(const InterRec *ir, int loop_ctr, const f3 &p, const f3 &d) {
    if (ir && ir->pos && loop_ctr < ir->slots) { ir->pos[loop_ctr] = p; ir->dir[loop_ctr] = d; }
}

// The "val.w < data_min" repair used by the linear branches (.h:834-845, 1056-1065 ...)
inline f4 fetch_linear(const Volume &v, f3 l, const f4 &prev, float ambient, MarchCount &mc) {
    f4 val = tex3d_linear(v, v.data, l.x, l.y, l.z);
    mc.samples++;
    if (val.w < v.data_min) {
        if (prev.w == 0) {
            const f4 t = tex3d_linear(v, v.data, l.x, l.y, l.z - 1);
            mc.samples++;
            val = f4{t.x, t.y, t.z, ambient - 1};
        } else {
            val = prev;
        }
    }
    return val;
}

// rk4, trace_rays_through_density_gradients.h:952-1291 (Sharma et al. 1982)
void rk4(f3 &rpos, f3 &rdir, const Volume &v, f3 scale, MarchCount &mc, const InterRec *ir = nullptr) {
    const float ambient = 1.000277;
    int loop_ctr = 0, spins = 0;
    f3 pos, lookup, R_n, T_n, A, B, C, D;
    f4 val, val_prev = f4{0, 0, 0, 0};
    float delta_t, current_n;
    if (v.interpolation == 1) {                                         // :992-1181
        while (true) {
            if (loop_ctr > LOOP_MAX) break;
            record_intermediate(ir, loop_ctr, rpos, rdir);              // :1004-1008
            pos = rpos;
            lookup = lookup_index(pos, v, scale);
            if (!inside_box(pos, v, lookup) && loop_ctr != 0) break;   // :1021
            if (!can_access(v, lookup)) {                               // :1043-1049
                pos = pos + v.step_size / (1 + v.data_min) * rdir;
                rpos = pos;
                if (++spins > SPIN_MAX) break;
                continue;
            }
            val = fetch_linear(v, lookup, val_prev, ambient, mc);       // :1052-1065
            loop_ctr += 1;
            val.w += 1;
            current_n = val.w;
            R_n = pos;
            delta_t = v.step_size / val.w;
            T_n = val.w * rdir;
            D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
            A = delta_t * D;
            pos = R_n + (float)(delta_t / 2.0) * T_n + (float)(1 / 8.0 * delta_t) * A;     // :1088
            lookup = lookup_index(pos, v, scale);
            if (!inside_box(pos, v, lookup)) break;
            val_prev = val; val_prev.w -= 1;
            val = fetch_linear(v, lookup, val_prev, ambient, mc);       // :1108-1119
            val.w += 1;
            D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
            B = delta_t * D;
            pos = R_n + delta_t * T_n + (float)(1 / 2.0 * delta_t) * B;                    // :1131
            lookup = lookup_index(pos, v, scale);
            if (!inside_box(pos, v, lookup)) break;
            val_prev = val; val_prev.w -= 1;
            val = fetch_linear(v, lookup, val_prev, ambient, mc);       // :1147-1158
            val.w += 1;
            D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
            C = delta_t * D;
            R_n = R_n + delta_t * (T_n + (float)(1 / 6.0) * (A + 2.0f * B));               // :1169
            T_n = T_n + (float)(1 / 6.0) * (A + 4.0f * B + C);                             // :1170
            val_prev = val; val_prev.w -= 1;
            rpos = R_n;
            rdir = normalize(T_n / current_n);                                              // :1178
            mc.iterations++;
        }
    } else {                                                            // :1186-1279
        while (true) {
            if (loop_ctr > LOOP_MAX) break;
            pos = rpos;
            lookup = lookup_index(pos, v, scale);
            if (!inside_box(pos, v, lookup) && loop_ctr != 0) break;
            if (!can_access(v, lookup)) {
                pos = pos + v.step_size / (1 + v.data_min) * rdir;
                rpos = pos;
                if (++spins > SPIN_MAX) break;
                continue;
            }
            val = tex3d_cubic(v, lookup.x, lookup.y, lookup.z); mc.samples++;               // :1216
            if (val.w < v.data_min) {                                                       // :1220-1227
                pos = pos + v.step_size / (1 + v.data_min) * rdir;
                rpos = pos;
                if (++spins > SPIN_MAX) break;
                continue;
            }
            loop_ctr += 1;
            val.w += 1;
            R_n = pos;
            delta_t = v.step_size / val.w;
            T_n = val.w * rdir;
            D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
            A = delta_t * D;
            pos = R_n + (float)(delta_t / 2.0) * T_n + (float)(1 / 8.0 * delta_t) * A;     // :1243
            lookup = lookup_index(pos, v, scale);
            if (!inside_box(pos, v, lookup)) break;
            val = tex3d_cubic(v, lookup.x, lookup.y, lookup.z); mc.samples++;
            val.w += 1;
            D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
            B = delta_t * D;
            pos = R_n + delta_t * T_n + (float)(1 / 2.0 * delta_t) * B;                    // :1256
            lookup = lookup_index(pos, v, scale);
            if (!inside_box(pos, v, lookup)) break;
            val = tex3d_cubic(v, lookup.x, lookup.y, lookup.z); mc.samples++;
            val.w += 1;
            D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
            C = delta_t * D;
            R_n = R_n + delta_t * (T_n + (float)(1 / 6.0) * (A + 2.0f * B));
            T_n = T_n + (float)(1 / 6.0) * (A + 4.0f * B + C);
            rpos = R_n;
            rdir = normalize(T_n / val.w);                                                  // :1276
            mc.iterations++;
        }
    }
}

// euler, trace_rays_through_density_gradients.h:743-950 (noise hook omitted)
void euler(f3 &rpos, f3 &rdir, const Volume &v, f3 scale, MarchCount &mc, const Noise &nz, uint64_t ray_id,
           const InterRec *ir = nullptr) {
    const float ambient = 1.000277;
    int loop_ctr = 0, spins = 0;
    f3 pos, dir, lookup, normal;
    f4 val, val_prev = f4{0, 0, 0, 0};
    if (v.interpolation == 1) {                                         // :770-894
        while (true) {
            if (loop_ctr > LOOP_MAX) break;
            record_intermediate(ir, loop_ctr, rpos, rdir);              // :784-790
            pos = rpos; dir = rdir;
            lookup = lookup_index(pos, v, scale);
            if (!inside_box(pos, v, lookup) && loop_ctr != 0) break;
            if (!can_access(v, lookup)) {
                pos = pos + v.step_size / (1 + v.data_min) * dir;
                rpos = pos;
                if (++spins > SPIN_MAX) break;
                continue;
            }
            val = fetch_linear(v, lookup, val_prev, ambient, mc);       // :830-845
            const float current_n = 1 + val.w;
            if (nz.add_ngrad) {                                         // :853-863
                float n0, n1;
                photon_normal2(nz.seed, ray_id, (uint32_t)loop_ctr, PHOTON_STREAM_NGRAD_NOISE, &n0, &n1);
                val.x += n0 * nz.ngrad_std;
                val.y += n1 * nz.ngrad_std;
            }
            normal = mk3(val.x, val.y, val.z);
            dir = dir + v.step_size * normal;                           // :869 (not renormalised)
            pos = pos + v.step_size / current_n * dir;                  // :875
            rpos = pos; rdir = dir;
            val_prev = val;
            loop_ctr += 1;
            mc.iterations++;
        }
    } else {                                                            // :897-945
        while (true) {
            if (loop_ctr > LOOP_MAX) break;
            pos = rpos; dir = rdir;
            lookup = lookup_index(pos, v, scale);
            if (!inside_box(pos, v, lookup) && loop_ctr != 0) break;
            // the cubic euler branch has no access_refractive_index guard (:906-912);
            // the sampler clamps addresses, so the fetch itself is always defined
            val = tex3d_cubic(v, lookup.x, lookup.y, lookup.z); mc.samples++;
            if (val.w < v.data_min) {
                pos = pos + v.step_size / (1 + v.data_min) * dir;
                rpos = pos;
                if (++spins > SPIN_MAX) break;
                continue;
            }
            loop_ctr += 1;
            normal = mk3(val.x, val.y, val.z);
            dir = dir + v.step_size * normal;
            dir = normalize(dir);                                       // :933
            const float n = 1 + val.w;
            pos = pos + dir * v.step_size / n;                          // :939  (dir*h)/n
            rpos = pos; rdir = dir;
            mc.iterations++;
        }
    }
}

// rk45, trace_rays_through_density_gradients.h:304-718: Runge-Kutta-Fehlberg 4(5) with step-size
// control on trilinear fetches of the raw volume (there is no cubic branch).  Restated literally,
// including what looks unintended: the texture's .w is n-1, but after the first accepted step it is
// used as the refractive index itself (:689), and a ray that starts ON a max face (where
// IntersectWithVolume puts every ray entering through one) fails the first inside test, shrinks h
// once to 0.09997 step and stops (:397-419) -- for such rays the integrator is a no-op.
// One deliberate difference: powf(x, 0.25f) (:668, not correctly rounded in any libm) is evaluated
// as sqrtf(sqrtf(x)), which is, so that the CPU and the GPU agree bit for bit.
void rk45(f3 &rpos, f3 &rdir, const Volume &v, f3 scale, MarchCount &mc) {
    const float tol = 1e-3;
    float refractive_index = 1.000277;
    float h = v.step_size / refractive_index;
    f3 pos = rpos, dir = rdir;
    f3 lookup, R_n, T_n;
    f4 val;
    f3 k1, k2, k3, k4, k5, k6, y4, y5, l1, l2, l3, l4, l5, l6, z4, z5;
    int loop_ctr = 0;
    // a stage point outside the volume: h /= 10, retry while h >= floor * step, else stop (:397-419 ...)
    auto shrink = [&](double floor_frac) {
        h = (float)(h / 10.0);
        return (double)h >= floor_frac * v.step_size;
    };
    auto fetch = [&](f3 l) { mc.samples++; return tex3d_linear(v, v.data, l.x, l.y, l.z); };
    while (true) {
        loop_ctr += 1;
        if (loop_ctr > 100000) break;                                   // :358
        R_n = pos;
        T_n = refractive_index * dir;
        k1 = h * T_n;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) { if (shrink(0.1)) continue; else break; }
        val = fetch(lookup);
        if (val.w < v.data_min) {                                       // :425-430
            pos = pos + v.step_size / v.data_min * dir;
            rpos = pos;
            continue;
        }
        l1 = (h * val.w) * mk3(val.x, val.y, val.z);
        R_n = pos + k1 / (float)4.0;
        T_n = refractive_index * dir + l1 / (float)4.0;
        k2 = h * T_n;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) { if (shrink(0.1)) continue; else break; }
        val = fetch(lookup);
        l2 = (h * val.w) * mk3(val.x, val.y, val.z);
        R_n = pos + (float)(3.0 / 32.0) * k1 + (float)(9.0 / 32.0) * k2;
        T_n = refractive_index * dir + (float)(3.0 / 32.0) * l1 + (float)(9.0 / 32.0) * l2;
        k3 = h * T_n;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) { if (shrink(0.1)) continue; else break; }
        val = fetch(lookup);
        l3 = (h * val.w) * mk3(val.x, val.y, val.z);
        R_n = pos + (float)(1932.0 / 2197.0) * k1 - (float)(7200.0 / 2197.0) * k2 + (float)(7296.0 / 2197.0) * k3;
        T_n = refractive_index * dir + (float)(1932.0 / 2197.0) * l1 - (float)(7200.0 / 2197.0) * l2 +
              (float)(7296.0 / 2197.0) * l3;
        k4 = h * T_n;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) { if (shrink(0.1)) continue; else break; }
        val = fetch(lookup);
        l4 = (h * val.w) * mk3(val.x, val.y, val.z);
        R_n = pos + (float)(439.0 / 216.0) * k1 - (float)8.0 * k2 + (float)(3680.0 / 513.0) * k3 -
              (float)(845.0 / 4104.0) * k4;
        T_n = refractive_index * dir + (float)(439.0 / 216.0) * l1 - (float)8.0 * l2 + (float)(3680.0 / 513.0) * l3 -
              (float)(845.0 / 4104.0) * l4;
        k5 = h * T_n;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) { if (shrink(0.1)) continue; else break; }
        val = fetch(lookup);
        l5 = (h * val.w) * mk3(val.x, val.y, val.z);
        R_n = pos - (float)(8.0 / 27.0) * k1 + (float)2.0 * k2 - (float)(3544.0 / 2565.0) * k3 +
              (float)(1859.0 / 4104.0) * k4 - (float)(11.0 / 40.0) * k5;
        T_n = refractive_index * dir - (float)(8.0 / 27.0) * l1 + (float)2.0 * l2 - (float)(3544.0 / 2565.0) * l3 +
              (float)(1859.0 / 4104.0) * l4 - (float)(11.0 / 40.0) * l5;
        k6 = h * T_n;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) { if (shrink(0.01)) continue; else break; }    // :606: 0.01 here
        val = fetch(lookup);
        l6 = (h * val.w) * mk3(val.x, val.y, val.z);
        y4 = pos + (float)(25.0 / 216.0) * k1 + (float)(1408.0 / 2565.0) * k3 + (float)(2197.0 / 4104.0) * k4 -
             (float)(1.0 / 5.0) * k5;
        y5 = pos + (float)(16.0 / 135.0) * k1 + (float)(6656.0 / 12825.0) * k3 + (float)(28561.0 / 56430.0) * k4 -
             (float)(9.0 / 50.0) * k5 + (float)(2.0 / 55.0) * k6;
        z4 = refractive_index * dir + (float)(25.0 / 216.0) * l1 + (float)(1408.0 / 2565.0) * l3 +
             (float)(2197.0 / 4104.0) * l4 - (float)(1.0 / 5.0) * l5;
        z5 = refractive_index * dir + (float)(16.0 / 135.0) * l1 + (float)(6656.0 / 12825.0) * l3 +
             (float)(28561.0 / 56430.0) * l4 - (float)(9.0 / 50.0) * l5 + (float)(2.0 / 55.0) * l6;
        const f3 dy = y4 - y5, dz = z4 - z5;
        const float ih = 1 / h;
        const f3 R0 = ih * mk3(fabsf(dy.x), fabsf(dy.y), fabsf(dy.z));     // :656-657
        const f3 R1 = ih * mk3(fabsf(dz.x), fabsf(dz.y), fabsf(dz.z));
        const float a = R0.x > R1.x ? R0.x : R1.x;
        const float b = R0.y > R1.y ? R0.y : R1.y;
        const float c = R0.z > R1.z ? R0.z : R1.z;
        const float R_max = (a > b ? a : b) > c ? (a > b ? a : b) : c;
        float s = (float)(0.84 * (double)sqrtf(sqrtf(tol / R_max)));        // :668 (powf -> two square roots)
        if (R_max <= tol) {                                                 // accept the 4th-order result
            pos = y4;
            dir = (1 / refractive_index) * z4;
            dir = normalize(dir);
            rpos = pos;
            rdir = dir;
            mc.iterations++;
            lookup = lookup_index(pos, v, scale);
            if (!inside_box(pos, v, lookup)) return;
            val = fetch(lookup);
            refractive_index = val.w;                                       // :689 (n-1, as stored)
            if ((double)s > 5.00) s = (float)5.00;
            h *= s;
        } else {
            if ((double)s < 0.1) s = (float)0.1;
            h *= s;
        }
    }
}

// adams_bashforth, .h:1293-1453: three RK4 start-up steps, then the 4-step Adams-Bashforth
// predictor, on trilinear fetches of the raw volume.  Restated literally: .w (n-1) is used as the
// refractive index throughout, the start-up loop tests ray_inside_box before its first step (so a ray
// that enters through a max face is returned untouched), its retreat step is step/data_min, and the
// main loop's retreat uses the position/direction captured before the loop.  Two definitions where
// the reference has undefined behaviour or no bound: T_n_prev / D_n_prev / val are zero when the
// start-up loop ends early (uninitialised there), and both loops are capped (LOOP_MAX, SPIN_MAX).
void adams_bashforth(f3 &rpos, f3 &rdir, const Volume &v, f3 scale, MarchCount &mc) {
    f3 pos = rpos, dir = rdir, lookup;
    f4 val = f4{0, 0, 0, 0};
    int loop_ctr = 0, spins = 0;
    f3 R_n = mk3(0, 0, 0), T_n = mk3(0, 0, 0), A, B, C, D;
    float delta_t;
    f3 D_prev[3] = {mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0)}, T_prev[3] = {mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0)};
    auto fetch = [&](f3 l) { mc.samples++; return tex3d_linear(v, v.data, l.x, l.y, l.z); };
    while (loop_ctr < 3) {                                              // :1316-1398
        pos = rpos;
        dir = rdir;
        lookup = lookup_index(pos, v, scale);
        if (!inside_box(pos, v, lookup)) break;
        if (!can_access(v, lookup)) {
            pos = pos + v.step_size / v.data_min * dir;
            rpos = pos;
            if (++spins > SPIN_MAX) break;
            continue;
        }
        val = fetch(lookup);
        if (val.w < v.data_min) {
            pos = pos + v.step_size / v.data_min * dir;
            rpos = pos;
            if (++spins > SPIN_MAX) break;
            continue;
        }
        R_n = pos;
        delta_t = v.step_size / val.w;
        T_n = val.w * rdir;
        D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
        A = delta_t * D;
        pos = R_n + (float)(delta_t / 2.0) * T_n + (float)(1 / 8.0 * delta_t) * A;
        lookup = lookup_index(pos, v, scale);
        if (!inside_box(pos, v, lookup)) break;
        val = fetch(lookup);
        D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
        B = delta_t * D;
        pos = R_n + delta_t * T_n + (float)(1 / 2.0 * delta_t) * B;
        lookup = lookup_index(pos, v, scale);
        if (!inside_box(pos, v, lookup)) break;
        val = fetch(lookup);
        D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
        C = delta_t * D;
        R_n = R_n + delta_t * (T_n + (float)(1 / 6.0) * (A + 2.0f * B));
        T_n = T_n + (float)(1 / 6.0) * (A + 4.0f * B + C);
        rpos = R_n;
        rdir = normalize(T_n / val.w);
        T_prev[loop_ctr] = T_n;
        D_prev[loop_ctr] = D;
        loop_ctr += 1;
        mc.iterations++;
    }
    loop_ctr = 0;                                                       // :1402-1450
    const float refractive_index = val.w;
    pos = rpos;
    dir = rdir;
    while (true) {
        loop_ctr += 1;
        if (loop_ctr > LOOP_MAX) break;
        R_n = rpos;
        lookup = lookup_index(R_n, v, scale);
        if (!inside_box(R_n, v, lookup)) break;
        val = fetch(lookup);
        if (val.w < v.data_min) {
            pos = pos + v.step_size / refractive_index * dir;
            rpos = pos;
            continue;
        }
        delta_t = v.step_size / val.w;
        D = mk3(val.w * val.x, val.w * val.y, val.w * val.z);
        const f3 R_n_1 = R_n + (delta_t / 24) * (55.0f * T_n - 59.0f * T_prev[2] + 37.0f * T_prev[1] - 9.0f * T_prev[0]);
        const f3 T_n_1 = T_n + (delta_t / 24) * (55.0f * D - 59.0f * D_prev[2] + 37.0f * D_prev[1] - 9.0f * D_prev[0]);
        T_prev[0] = T_prev[1]; D_prev[0] = D_prev[1];
        T_prev[1] = T_prev[2]; D_prev[1] = D_prev[2];
        T_prev[2] = T_n; D_prev[2] = D;
        R_n = R_n_1;
        T_n = T_n_1;
        rpos = R_n;
        rdir = normalize(T_n / val.w);
        mc.iterations++;
    }
}

// trace_rays_through_density_gradients, .h:1455-1544
void trace_volume(f3 &pos_io, f3 &dir_io, const Volume &v, int algorithm, MarchCount &mc, const Noise &nz = Noise(),
                  uint64_t ray_id = 0, const InterRec *ir = nullptr) {
    const f3 mn = v.min_bound, mx = v.max_bound;
    const f3 scale = mk3(1.0f / (mx.x - mn.x), 1.0f / (mx.y - mn.y), 1.0f / (mx.z - mn.z));
    f3 pos = pos_io;
    const f3 dir = dir_io;
    if (pos.x <= mn.x || pos.y <= mn.y || pos.z <= mn.z || pos.x >= mx.x || pos.y >= mx.y ||
        pos.z >= mx.z) {
        if (!intersect_with_volume(pos, dir, mn, mx)) return;           // miss: ray unchanged
    }
    pos_io = pos;
    switch (algorithm) {
        case 1: euler(pos_io, dir_io, v, scale, mc, nz, ray_id, ir); break;
        case 2: rk4(pos_io, dir_io, v, scale, mc, ir); break;
        case 3: rk45(pos_io, dir_io, v, scale, mc); break;
        case 4: adams_bashforth(pos_io, dir_io, v, scale, mc); break;
        default: break;
    }
}

// ---------------------------------------------------------------------------------------
// Ray generation, optics, sensor
// ---------------------------------------------------------------------------------------

struct Source { float x, y, z; double radiance; int diameter_index; };

// generate_lightfield_angular_data, parallel_ray_tracing.cu:71-237
Ray generate_ray(float lens_pitch, float image_distance, const scattering_data_t &sd, int scattering_type,
                 const Source &s, int rays_per_source, float beam_wavelength, float f_number,
                 float r1, float r2, float ratio) {
    const float x_current = s.x, y_current = s.y, z_current = s.z;
    float x_lens, y_lens;
    if (rays_per_source == 1) {
        x_lens = 0.0; y_lens = 0.0;
    } else {                                                            // :123-124 (double math)
        x_lens = (float)(ratio * 1.0 * lens_pitch * r1 * photon_det_cos(2 * M_PI * r2));
        y_lens = (float)(ratio * 1.0 * lens_pitch * r1 * photon_det_sin(2 * M_PI * r2));
    }
    const float theta = photon_det_atanf(-(x_lens - x_current) / (image_distance - z_current));
    const float phi = photon_det_atanf(-(y_lens - y_current) / (image_distance - z_current));
    double irradiance_current;
    if (scattering_type) {
        const float del = sd.scattering_angle[1] - sd.scattering_angle[0];
        const f3 beam = mk3(sd.beam_propagation_vector[0], sd.beam_propagation_vector[1],
                            sd.beam_propagation_vector[2]);
        f3 d = normalize(mk3(photon_det_tanf(theta), photon_det_tanf(phi), -1.0f));
        float dv[3];
        for (int i = 0; i < 3; i++) {
            const f3 row = mk3(sd.inverse_rotation_matrix[i * 3 + 0], sd.inverse_rotation_matrix[i * 3 + 1],
                               sd.inverse_rotation_matrix[i * 3 + 2]);
            dv[i] = dot(row, d);
        }
        d = normalize(mk3(dv[0], dv[1], dv[2]));
        // angleBetween (float3_operators.h:84-90) returns degrees as float; :184 converts back
        const float dp = beam.x * d.x + beam.y * d.y + beam.z * d.z;
        const float deg = (float)(photon_det_acosf(dp) * 180.0 / M_PI);
        const float ray_angle = (float)(deg * M_PI / 180.0);
        const float angle = (ray_angle - sd.scattering_angle[0]) / del;          // :190
        // (int)angle, clamped into the table (the reference reads row num_angles when the
        // angle lands on the last row, :194-198; NaN -> 0 as CUDA's cvt does)
        int angle_l = std::isnan(angle) ? 0 : (int)angle;
        int angle_u = angle_l + 1;
        const int rl = clampi(angle_l, 0, sd.num_angles - 1), ru = clampi(angle_u, 0, sd.num_angles - 1);
        const int col = clampi(s.diameter_index, 0, sd.num_diameters - 1);
        const float il = sd.scattering_irradiance[rl * sd.num_diameters + col];
        const float iu = sd.scattering_irradiance[ru * sd.num_diameters + col];
        const float irr = il + (angle - angle_l) / (angle_u - angle_l) * (iu - il);   // :201
        irradiance_current = irr * s.radiance;
    } else {
        irradiance_current = s.radiance;
    }
    Ray r;
    r.pos = mk3(x_current, y_current, z_current);
    r.dir = normalize(mk3(photon_det_tanf(theta), photon_det_tanf(phi), -1.0f));
    r.wavelength = beam_wavelength;
    r.radiance = 1 / (f_number * f_number) * irradiance_current;        // :233
    return r;
}

// ray_sphere_intersection, parallel_ray_tracing.cu:239-343
f3 ray_sphere_intersection(f3 pos_c, float R, f3 dir_i, f3 pos_i, char surface) {
    const float alpha = dot(dir_i, dir_i);
    const float beta = 2 * dot(dir_i, (pos_i - pos_c));
    const float gamma = dot(pos_i - pos_c, pos_i - pos_c) - R * R;
    const float sq = (float)(beta * beta - 4.0 * alpha * gamma);        // :276 (double product)
    if (sq < 0.0) return mk3(NANF, NANF, NANF);
    const float t1 = (float)((-beta + sqrtf(sq)) / (2.0 * alpha));
    const float t2 = (float)((-beta - sqrtf(sq)) / (2.0 * alpha));
    float t;
    if (surface == 'f') t = (R > 0) ? (t1 <= t2 ? t1 : t2) : (t1 >= t2 ? t1 : t2);
    else t = (R > 0) ? (t1 <= t2 ? t1 : t2) : (t1 >= t2 ? t1 : t2);   // :318-336 (same picks)
    return pos_i + dir_i * t;
}

// measure_distance_to_optical_axis, parallel_ray_tracing.cu:345-380
float axis_distance(f3 pos_i, f3 pos_0, const float plane[4]) {
    const float a = plane[0], b = plane[1], c = plane[2];
    const float tmin = dot(mk3(a, b, c), pos_i - pos_0) / (a * a + b * b + c * c);
    const f3 p = pos_0 + mk3(a, b, c) * tmin;
    return sqrtf(dot(pos_i - p, pos_i - p));
}

// propagate_rays_through_single_element, parallel_ray_tracing.cu:383-1011
Ray single_element(const element_data_t &e, f3 center, const float plane[4], Ray ray) {
    const char type = e.element_type;
    f3 dir = ray.dir, src = ray.pos;
    const float wavelength = ray.wavelength;
    double radiance = ray.radiance;
    const float a = plane[0], b = plane[1], c = plane[2], d = plane[3];
    const float pitch = e.element_geometry.pitch;
    const double vertex_distance = e.element_geometry.vertex_distance;
    if (type == 't') {                                                  // :416-503
        const float focal = e.element_properties.thin_lens_focal_length;
        const float t = -(dot(mk3(a, b, c), src) + d) / dot(mk3(a, b, c), dir);
        const f3 hit = src + dir * t;
        const float dist = axis_distance(hit, center, plane);
        if (dist > pitch / 2.0) { kill(ray); return ray; }
        src = hit;
        dir = -(src - center) / focal + dir;
        dir = normalize(dir);
    } else if (type == 'l') {                                           // :507-864
        const float Rf = e.element_geometry.front_surface_radius;
        const float Rb = e.element_geometry.back_surface_radius;
        const double n_lens = e.element_properties.refractive_index;
        const float abbe = e.element_properties.abbe_number;
        const float transmission = e.element_properties.transmission_ratio;
        const float absorbance = e.element_properties.absorbance_rate;
        const float nmag = sqrtf(a * a + b * b + c * c);
        float ds = (float)(+vertex_distance / 2.0 - Rf);                // :557
        const f3 c_front = center + mk3(a, b, c) * ds / nmag;           // (v*ds)/nmag
        f3 hit = ray_sphere_intersection(c_front, Rf, dir, src, 'f');
        float dist = axis_distance(hit, center, plane);
        if (dist > pitch / 2.0) { kill(ray); return ray; }
        f3 normal = normalize(hit - c_front);
        float eta;
        const float lambda_D = 589.3, lambda_F = 486.1, lambda_C = 656.3;
        if (!std::isnan(abbe)) {                                        // :622-636
            eta = (float)(1.0 / (n_lens + (1. / (wavelength * wavelength) - 1 / (lambda_D * lambda_D)) *
                                              ((n_lens - 1) / (abbe * (1 / (lambda_F * lambda_F) -
                                                                       1 / (lambda_C * lambda_C))))));
        } else {
            eta = (float)(1.0 / n_lens);
        }
        float cosi = -dot(dir, normal);
        float radicand = (float)(1.0 - (eta * eta) * (1.0 - cosi * cosi));          // :652
        if (radicand < 0.0) { kill(ray); return ray; }
        dir = dir * eta + (eta * cosi - sqrtf(radicand)) * normal;      // :682
        dir = normalize(dir);
        src = hit;
        ds = (float)(-vertex_distance / 2 - Rb);                        // :704
        const f3 c_back = center + mk3(a, b, c) * ds / nmag;
        hit = ray_sphere_intersection(c_back, Rb, dir, src, 'b');
        dist = axis_distance(hit, center, plane);
        if (dist > pitch / 2.0) { kill(ray); return ray; }
        normal = normalize(-(hit - c_back));
        if (!std::isnan(abbe)) {                                        // :770-782
            eta = (float)(n_lens + (1.0 / (wavelength * wavelength) - 1.0 / (lambda_D * lambda_D)) *
                                       ((n_lens - 1) / (abbe * (1 / (lambda_F * lambda_F) -
                                                                1 / (lambda_C * lambda_C)))));
        } else {
            eta = (float)n_lens;
        }
        cosi = -dot(dir, normal);
        radicand = (float)(1.0 - (eta * eta) * (1.0 - cosi * cosi));    // :797
        if (radicand < 0.0) { kill(ray); return ray; }
        dir = eta * dir + (eta * cosi - sqrtf(radicand)) * normal;      // :827
        dir = normalize(dir);
        if (absorbance != 0) {                                          // :838-848
            const float dist_in = sqrtf(dot(hit - src, hit - src));
            radiance = (1.0 - absorbance) * radiance * dist_in;
        } else {
            radiance = transmission * radiance;
        }
        src = hit;
    } else {                                                            // aperture, :868-992
        const float nmag = sqrtf(a * a + b * b + c * c);
        float ds = (float)(-vertex_distance / 2.0);
        float d_temp = d - ds * nmag;
        float t = -(dot(mk3(a, b, c), src) + d_temp) / dot(mk3(a, b, c), dir);
        f3 hit = src + dir * t;
        float dist = axis_distance(hit, center, plane);
        if (dist > pitch / 2.0) { kill(ray); return ray; }
        ds = (float)(+vertex_distance / 2);
        d_temp = d - ds * nmag;
        t = -(dot(mk3(a, b, c), src) + d_temp) / dot(mk3(a, b, c), dir);
        hit = src + dir * t;
        dist = axis_distance(hit, center, plane);
        if (dist > pitch / 2.0) { kill(ray); return ray; }
        src = hit;
    }
    ray.dir = dir; ray.pos = src; ray.wavelength = wavelength; ray.radiance = radiance;
    return ray;
}

// propagate_rays_through_optical_system, parallel_ray_tracing.cu:1274-1381.  Only the
// single-element branch does anything in the reference (and it always uses element 0,
// :1331-1333); groups with more than one simultaneous element reach a stub.
//
// train_mode 1 is the WORKING train the reference advertises, after the design of
// its numpy ancestor (perform_ray_tracing_03.py:1254-1485): groups in decreasing system index; a
// single-member group goes through ITS element; a group of simultaneous elements (a lenslet array)
// is split by element plane, the planes are visited in the order the ray meets them, and on each
// plane the ray goes through the member whose centre is nearest to its intersection point.
// Pinning: the numpy ancestor's SEQUENTIAL branch (propogate_rays_through_optical_system, :1419-1485,
// trains of single-member groups) runs in this image and pins the group order and the per-element
// chaining (tests/golden/train_f64.npz, made by make_golden.py::train_golden; CPU test
// test_element_train_vs_reference_numpy, GPU twin in test_parity_gpu.py).  Only its
// simultaneous-elements branch (:1254-1417, entered through a 1x1 object array at :1436-1446) cannot
// run, so the lenslet-group walk is pinned by optics (tests/) and GPU-vs-oracle parity only.
int g_element_train = 0;            // oracle_set_element_train()

Ray optical_system(const element_data_t *elems, const float (*centers)[3], const float (*planes)[4],
                   const int *sys_index, int num_elements, Ray ray, int train_mode = 0) {
    int seq = 0;
    if (train_mode == 0) {
        const int n = num_elements < 5 ? num_elements : 5;              // MAX_CURRENT_ELEMENTS, .cu:38
        for (int k = 0; k < n; k++)
            if (seq <= sys_index[k]) seq = sys_index[k];
        for (int idx = 0; idx < seq; idx++) {
            int count = 0;
            for (int k = 0; k < n; k++)
                if (seq - sys_index[k] == idx) count++;
            if (count == 1)
                ray = single_element(elems[0], mk3(centers[0][0], centers[0][1], centers[0][2]), planes[0], ray);
        }
        return ray;
    }
    auto same_plane = [](const float *p, const float *q) { return p[0] == q[0] && p[1] == q[1] && p[2] == q[2] && p[3] == q[3]; };
    auto plane_time = [](const float *pl, const Ray &r) {
        return -(pl[0] * r.pos.x + pl[1] * r.pos.y + pl[2] * r.pos.z + pl[3]) /
               (pl[0] * r.dir.x + pl[1] * r.dir.y + pl[2] * r.dir.z);
    };
    for (int k = 0; k < num_elements; k++)
        if (seq <= sys_index[k]) seq = sys_index[k];
    for (int idx = 0; idx < seq; idx++) {
        int count = 0, only = 0;
        for (int k = 0; k < num_elements; k++)
            if (seq - sys_index[k] == idx) { only = k; count++; }
        if (count == 0) continue;
        if (count == 1) {
            ray = single_element(elems[only], mk3(centers[only][0], centers[only][1], centers[only][2]), planes[only], ray);
            continue;
        }
        const int kMaxGroupPlanes = 8;
        int uplane[kMaxGroupPlanes], nu = 0;
        float ut[kMaxGroupPlanes];
        for (int k = 0; k < num_elements; k++) {                        // distinct planes, first appearance first
            if (seq - sys_index[k] != idx) continue;
            bool seen = false;
            for (int u = 0; u < nu; u++) seen = seen || same_plane(planes[uplane[u]], planes[k]);
            if (seen || nu == kMaxGroupPlanes) continue;
            uplane[nu] = k;
            ut[nu] = plane_time(planes[k], ray);
            nu++;
        }
        for (int a = 1; a < nu; a++) {                                  // stable insertion sort by time
            const int pk = uplane[a];
            const float tk = ut[a];
            int b = a - 1;
            while (b >= 0 && ut[b] > tk) { uplane[b + 1] = uplane[b]; ut[b + 1] = ut[b]; b--; }
            uplane[b + 1] = pk; ut[b + 1] = tk;
        }
        for (int u = 0; u < nu; u++) {
            const float *pl = planes[uplane[u]];
            const float t = plane_time(pl, ray);
            const f3 hit = ray.pos + t * ray.dir;
            int best = -1;
            float best_d2 = 0;
            for (int k = 0; k < num_elements; k++) {
                if (seq - sys_index[k] != idx || !same_plane(planes[k], pl)) continue;
                const f3 dc = hit - mk3(centers[k][0], centers[k][1], centers[k][2]);
                const float d2 = dot(dc, dc);
                if (best < 0 || d2 < best_d2) { best = k; best_d2 = d2; }
            }
            ray = single_element(elems[best], mk3(centers[best][0], centers[best][1], centers[best][2]), planes[best], ray);
        }
    }
    return ray;
}

struct alignas(64) Image {          // per-thread double accumulator (a cache line of its own: taps is bumped per splat)
    int W, H;
    std::vector<double> acc;
    uint64_t taps = 0;
};

// Gaussian-spot (erf) splat shared by intersect_sensor_02 (.cu:1383-1543, render_fraction
// 0.75) and create_apparent_image (.cu:1545-1733, render_fraction 1.0).
// erf: the reference calls CUDA's erf(double).  Oracle and product share photon_det_erf (include/photon_det_math.h: absolute
// error <= 4e-16) since round 5 -- before, the oracle used glibc's and the product ocml's.  The `libm_erf` build of this file
// (oracle/Makefile) keeps glibc's: tests/test_oracle_golden.py::test_erf_form_sensitivity bounds one against the other.
#ifdef PHOTON_ORACLE_LIBM_ERF
#define ORACLE_ERF(x) erf(x)
#else
#define ORACLE_ERF(x) photon_det_erf(x)
#endif
void erf_splat(Image &img, float d_x, float d_y, double radiance, f3 dir, float D, float render_fraction) {
    const double pi = 3.141592653589793;
    const float alpha = photon_det_atanf(sqrtf((dir.x / dir.z) * (dir.x / dir.z) + (dir.y / dir.z) * (dir.y / dir.z)));
    const double cos4 = photon_det_cosf(alpha) * photon_det_cosf(alpha) * photon_det_cosf(alpha) * photon_det_cosf(alpha);     // float product
    const float X = d_x - 0.5, Y = d_y - 0.5;
    const float I0 = (float)(radiance * cos4 * 8.0 / pi);
    const float sqrt8 = sqrtf(8.0);
    const int c0 = (int)floorf(X - render_fraction * D), c1 = (int)ceilf(X + render_fraction * D);
    const int r0 = (int)floorf(Y - render_fraction * D), r1 = (int)ceilf(Y + render_fraction * D);
    for (int col = c0; col <= c1; col++)
        for (int row = r0; row <= r1; row++) {
            const float rad = sqrtf((col - X) * (col - X) + (row - Y) * (row - Y));
            const bool render = col >= 0 && col <= img.W - 1 && row >= 0 && row <= img.H - 1 &&
                                rad <= render_fraction * D;
            if (!render) continue;
            const float inc = (float)(I0 * pi / 32.0 *
                                      (ORACLE_ERF(sqrt8 * (col - X - 0.5) / D) - ORACLE_ERF(sqrt8 * (col - X + 0.5) / D)) *
                                      (ORACLE_ERF(sqrt8 * (row - Y - 0.5) / D) - ORACLE_ERF(sqrt8 * (row - Y + 0.5) / D)));
            img.acc[(size_t)row * img.W + col] += inc;
            img.taps++;
        }
}

inline float sensor_time(f3 src, f3 dir, float a, float b, float c, float d) {
    return -(dot(mk3(a, b, c), src) + d) / dot(mk3(a, b, c), dir);
}

// intersect_sensor_02, parallel_ray_tracing.cu:1383-1543.  Returns final position (NaN = lost).
inline void add_position_noise(f3 &hit, const camera_design_t &cam, const Noise &nz, uint64_t ray_id) {
    if (!nz.add_pos) return;                                            // .cu:1424-1434
    float n0, n1;
    photon_normal2(nz.seed, ray_id, 0, PHOTON_STREAM_POS_NOISE, &n0, &n1);
    hit.x += n0 * nz.pos_std * cam.pixel_pitch;
    hit.y += n1 * nz.pos_std * cam.pixel_pitch;
}

f3 sensor_diffraction(Image &img, const Ray &ray, const camera_design_t &cam, const Noise &nz, uint64_t ray_id) {
    const float t = sensor_time(ray.pos, ray.dir, 0.0f, 0.0f, 1.0f, -cam.z_sensor);
    f3 hit = ray.pos + ray.dir * t;
    add_position_noise(hit, cam, nz, ray_id);
    const float p1x = (float)(-cam.pixel_pitch * (cam.x_pixel_number - 1) / 2.0);
    const float p1y = (float)(-cam.pixel_pitch * (cam.y_pixel_number - 1) / 2.0);
    const float d_x = cam.x_pixel_number - 1 - (hit.x - p1x) / cam.pixel_pitch;     // x flipped (:1446)
    const float d_y = (hit.y - p1y) / cam.pixel_pitch;
    if (d_x >= cam.x_pixel_number || d_y >= cam.y_pixel_number || d_x < 0 || d_y < 0)
        return mk3(NANF, NANF, NANF);
    erf_splat(img, d_x, d_y, ray.radiance, ray.dir, cam.diffraction_diameter, 0.75f);
    return hit;
}

// create_apparent_image, parallel_ray_tracing.cu:1545-1733
f3 apparent_image(Image &img, const Ray &ray, const camera_design_t &cam, float z_object, float z_offset,
                  const element_data_t &e, const Noise &nz, uint64_t ray_id) {
    const f3 dir = -ray.dir;
    const float t = sensor_time(ray.pos, dir, 0.0f, 0.0f, -1.0f, z_object);
    f3 hit = ray.pos + dir * t;
    const float focal = e.element_properties.thin_lens_focal_length;
    const float M = focal / (z_object - z_offset - focal);
    hit.x = -hit.x * M;
    hit.y = -hit.y * M;
    add_position_noise(hit, cam, nz, ray_id);                           // .cu:1607-1616
    const float p1x = (float)(-cam.pixel_pitch * (cam.x_pixel_number - 1) / 2.0);
    const float p1y = (float)(-cam.pixel_pitch * (cam.y_pixel_number - 1) / 2.0);
    const float d_x = cam.x_pixel_number - 1 - (hit.x - p1x) / cam.pixel_pitch;
    const float d_y = (hit.y - p1y) / cam.pixel_pitch;
    if (d_x >= cam.x_pixel_number || d_y >= cam.y_pixel_number || d_x < 0 || d_y < 0)
        return mk3(NANF, NANF, NANF);
    erf_splat(img, d_x, d_y, ray.radiance, dir, cam.diffraction_diameter, 1.0f);
    return hit;
}

// The four pixels a sensor hit is shared between and their area weights, parallel_ray_tracing.cu:1803-1880
// (no x flip, no MATLAB-style +1.5 offset: that is the numpy ancestor's, perform_ray_tracing_03.py:1505-1506).
struct PixelTaps { bool inside; int ii[4], jj[4]; double w[4]; };
PixelTaps pixel_taps(f3 hit, const camera_design_t &cam) {
    PixelTaps t{};
    const float p1x = (float)(-cam.pixel_pitch * (cam.x_pixel_number - 1) / 2.0);
    const float p1y = (float)(-cam.pixel_pitch * (cam.y_pixel_number - 1) / 2.0);
    const float d_x = (hit.x - p1x) / cam.pixel_pitch;
    const float d_y = (hit.y - p1y) / cam.pixel_pitch;
    t.inside = !(d_x >= cam.x_pixel_number || d_y >= cam.y_pixel_number || d_x < 0 || d_y < 0);
    if (!t.inside) return t;
    const float d_y_lower = d_y - 0.5, d_x_lower = d_x - 0.5;
    const double d_ii_ul = ceilf(d_y_lower) - d_y_lower;                // float op stored in double
    const double d_jj_ul = ceilf(d_x_lower) - d_x_lower;
    const double w[4] = {d_ii_ul * d_jj_ul, d_ii_ul * (1 - d_jj_ul), (1 - d_ii_ul) * d_jj_ul,
                         (1 - d_ii_ul) * (1 - d_jj_ul)};
    const int ii_ul = (int)(ceilf(d_y_lower) - 1), jj_ul = (int)(ceilf(d_x_lower) - 1);
    const int ii[4] = {ii_ul, ii_ul, ii_ul + 1, ii_ul + 1};
    const int jj[4] = {jj_ul, jj_ul + 1, jj_ul, jj_ul + 1};
    for (int k = 0; k < 4; k++) { t.ii[k] = ii[k]; t.jj[k] = jj[k]; t.w[k] = w[k]; }
    return t;
}

// intersect_sensor + the 4-pixel splat loop, parallel_ray_tracing.cu:1735-1895, :2199-2234
f3 sensor_bilinear(Image &img, const Ray &ray, const camera_design_t &cam, const Noise &nz, uint64_t ray_id) {
    const float t = sensor_time(ray.pos, ray.dir, 0.0f, 0.0f, 1.0f, -cam.z_sensor);
    f3 hit = ray.pos + ray.dir * t;
    add_position_noise(hit, cam, nz, ray_id);                           // .cu:1773-1783
    const f3 dir = ray.dir;
    const float alpha = photon_det_atanf(sqrtf((dir.x / dir.z) * (dir.x / dir.z) + (dir.y / dir.z) * (dir.y / dir.z)));
    const double cos4 = photon_det_cosf(alpha) * photon_det_cosf(alpha) * photon_det_cosf(alpha) * photon_det_cosf(alpha);
    const PixelTaps px = pixel_taps(hit, cam);
    if (!px.inside) return mk3(NANF, NANF, NANF);
    const int *ii = px.ii, *jj = px.jj;
    const double *w = px.w;
    const int W = cam.x_pixel_number, H = cam.y_pixel_number;
    for (int k = 0; k < 4; k++) {
        if (ii[k] < 0 || ii[k] >= H || jj[k] < 0 || jj[k] >= W) continue;          // :2223
        const long idx = (long)(ii[k] - 1) * W + jj[k] - 1;                        // :2228
        if (idx < 0) continue;      // reference writes before the buffer here (UB) -> skipped
        const double inc = w[k] * ray.radiance * cos4;
        img.acc[idx] += (float)inc;
        img.taps++;
    }
    return hit;
}

// glibc rand() (TYPE_3 additive feedback) is what the reference calls (.cu:3228-3235).
void rand_table(int n, float *r1, float *r2) {
    srand(10);
    for (int k = 0; k < n; k++) {
        r1[k] = (float)((double)rand() / (RAND_MAX));
        r2[k] = (float)((double)rand() / (RAND_MAX));
    }
}

struct Scene {
    float lens_pitch, image_distance, beam_wavelength, f_number, ratio;
    scattering_data_t sd;
    int scattering_type;
    lightfield_source_t ls;
    int rays_per_source;
    int num_elements;
    std::vector<element_data_t> elems;
    std::vector<float> centers;   // [n][3]
    std::vector<float> planes;    // [n][4]
    std::vector<int> sys_index;
    camera_design_t cam;
    std::vector<float> r1, r2;
    Noise noise;
};

struct RayOut { f3 pos, dir; };     // what the reference dumps for one ray

// Body of the master kernel for one ray, parallel_ray_tracing.cu:1923-2243.
void trace_one(const Scene &sc, const Volume *vol, int algorithm, int64_t source, int local_ray,
               Image &img, MarchCount &mc, RayOut *dump, uint64_t &on_sensor, const InterRec *ir = nullptr) {
    Source s{sc.ls.x[source], sc.ls.y[source], sc.ls.z[source], sc.ls.radiance[source],
             sc.ls.diameter_index[source]};
    Ray ray = generate_ray(sc.lens_pitch, sc.image_distance, sc.sd, sc.scattering_type, s, sc.rays_per_source,
                           sc.beam_wavelength, sc.f_number, sc.r1[local_ray], sc.r2[local_ray], sc.ratio);
    const camera_design_t &cam = sc.cam;
    const uint64_t ray_id = (uint64_t)source * (uint64_t)sc.rays_per_source + (uint64_t)local_ray;
    if (vol) {                                                          // :2033-2131
        f3 p = ray.pos, d = ray.dir;
        p.z = (float)(p.z - (sc.ls.z_offset + 750e3));
        float pv[3], dv[3];
        for (int i = 0; i < 3; i++) {
            const f3 row = mk3(cam.inverse_rotation_matrix[i * 3], cam.inverse_rotation_matrix[i * 3 + 1],
                               cam.inverse_rotation_matrix[i * 3 + 2]);
            pv[i] = dot(row, p); dv[i] = dot(row, d);
        }
        p = mk3(pv[0], pv[1], pv[2]); d = mk3(dv[0], dv[1], dv[2]);
        trace_volume(p, d, *vol, algorithm, mc, sc.noise, ray_id, ir);
        for (int i = 0; i < 3; i++) {
            const f3 row = mk3(cam.rotation_matrix[i * 3], cam.rotation_matrix[i * 3 + 1], cam.rotation_matrix[i * 3 + 2]);
            pv[i] = dot(row, p); dv[i] = dot(row, d);
        }
        p = mk3(pv[0], pv[1], pv[2]);
        d = normalize(mk3(dv[0], dv[1], dv[2]));
        p.z = (float)(p.z + (sc.ls.z_offset + 750e3));
        ray.pos = p; ray.dir = d;
        if (isnan3(ray.dir) || isnan3(ray.pos)) return;
    }
    if (dump) dump->dir = ray.dir;                                      // :2136-2141
    if (sc.elems[0].element_type == 'n') {                              // :2143-2158
        const float z_obj = sc.ls.object_distance + sc.ls.z_offset;
        const f3 fin = apparent_image(img, ray, cam, z_obj, sc.ls.z_offset, sc.elems[0], sc.noise, ray_id);
        if (dump) dump->pos = fin;
        if (!std::isnan(fin.x)) on_sensor++;
        return;
    }
    ray = optical_system(sc.elems.data(), reinterpret_cast<const float(*)[3]>(sc.centers.data()),
                         reinterpret_cast<const float(*)[4]>(sc.planes.data()), sc.sys_index.data(),
                         sc.num_elements, ray, g_element_train);
    if (isnan3(ray.dir) || isnan3(ray.pos)) return;                     // :2172-2176
    if (cam.implement_diffraction) {
        const f3 fin = sensor_diffraction(img, ray, cam, sc.noise, ray_id);
        if (dump) dump->pos = fin;
        if (!std::isnan(fin.x)) on_sensor++;
    } else {
        const f3 fin = sensor_bilinear(img, ray, cam, sc.noise, ray_id);
        if (std::isnan(fin.x) || std::isnan(fin.y)) return;             // :2196
        if (dump) dump->pos = fin;
        on_sensor++;
    }
}

void build_scene(Scene &sc, float lens_pitch, float image_distance, scattering_data_t *sdp, char *stype,
                 lightfield_source_t *lsp, int rays_per_source, float beam_wavelength, float f_number,
                 int num_elements, double (*element_center)[3], element_data_t *edp,
                 double (*element_plane_parameters)[4], int *element_system_index, camera_design_t *cam,
                 float ratio) {
    sc.lens_pitch = lens_pitch; sc.image_distance = image_distance; sc.beam_wavelength = beam_wavelength;
    sc.f_number = f_number; sc.ratio = ratio;
    sc.sd = *sdp; sc.ls = *lsp; sc.rays_per_source = rays_per_source;
    sc.scattering_type = strcmp(stype, "mie") == 0 ? 1 : 0;             // .cu:3192
    sc.num_elements = num_elements;
    sc.elems.assign(edp, edp + num_elements);
    sc.centers.resize(3 * num_elements); sc.planes.resize(4 * num_elements);
    for (int k = 0; k < num_elements; k++) {                            // .cu:3256-3260 (f64 -> f32)
        for (int j = 0; j < 3; j++) sc.centers[3 * k + j] = (float)element_center[k][j];
        for (int j = 0; j < 4; j++) sc.planes[4 * k + j] = (float)element_plane_parameters[k][j];
    }
    sc.sys_index.assign(element_system_index, element_system_index + num_elements);
    sc.cam = *cam;
    sc.r1.resize(rays_per_source); sc.r2.resize(rays_per_source);
    rand_table(rays_per_source, sc.r1.data(), sc.r2.data());
}

void write_dump(const char *dir, const char *prefix, int k, const std::vector<f3> &v) {
    char name[32];
    snprintf(name, sizeof name, "%s%04d.bin", prefix, k);               // .cu:3574
    const std::string full = std::string(dir) + "/" + name;
    std::ofstream f(full.c_str(), std::ios::out | std::ios::binary);
    f.write(reinterpret_cast<const char *>(v.data()), v.size() * sizeof(f3));
}

}  // namespace

// =========================================================================================
// exported test entry points
// =========================================================================================
extern "C" {

struct oracle_stats_t {
    uint64_t rays_launched, rays_on_sensor, rk_iterations, volume_samples, sensor_taps;
};

// Shared body: the launch loop of parallel_ray_tracing.cu:3366-3675 on the CPU.
static void render_core(Scene &sc, const Volume *volp, float *image_array, bool save_lightrays,
                        char *lightray_position_save_path, char *lightray_direction_save_path,
                        int num_lightrays_save, int ray_tracing_algorithm, oracle_stats_t *stats,
                        int inter_slots = 0) {
    const int W = sc.cam.x_pixel_number, H = sc.cam.y_pixel_number;
    const int64_t num_particles = sc.ls.num_particles;
    int64_t chunk = sc.ls.source_point_number;                          // .cu:3366-3372
    if (num_particles < chunk) chunk = num_particles;
    if (chunk <= 0) return;
    const int64_t kmax = (num_particles + chunk - 1) / chunk;           // .cu:3506-3510
    const int rps = sc.rays_per_source;
    const int64_t num_rays = chunk * rps;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    std::vector<Image> imgs(nthreads);
    for (auto &im : imgs) { im.W = W; im.H = H; im.acc.assign((size_t)W * H, 0.0); }
    std::vector<MarchCount> mcs(nthreads);
    std::vector<uint64_t> on_sensor(nthreads, 0);
    std::vector<f3> fpos, fdir, ipos, idir;
    const bool inter = save_lightrays && volp && inter_slots > 0;      // .cu:3484
    for (int64_t k = 0; k < kmax; k++) {
        const int64_t n_min = k * chunk;
        if (save_lightrays) {
            fpos.assign(num_lightrays_save, mk3(NANF, NANF, NANF));
            fdir.assign(num_lightrays_save, mk3(NANF, NANF, NANF));
        }
        if (inter) {                                                    // .cu:3535-3546
            ipos.assign((size_t)num_lightrays_save * inter_slots, mk3(NANF, NANF, NANF));
            idir.assign((size_t)num_lightrays_save * inter_slots, mk3(NANF, NANF, NANF));
        }
        // per-thread counters live on the thread's own stack inside the parallel region and are merged once: as adjacent
        // elements of a vector they shared cache lines, and a counter bumped at every texel sample made sixteen threads
        // run barely faster than three (bench.py's cpu_baseline, round 3)
#pragma omp parallel
        {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        MarchCount mc_local{0, 0};
        uint64_t on_local = 0;
#pragma omp for schedule(dynamic, 64)
        for (int64_t gid = 0; gid < num_rays; gid++) {                  // one iteration = one GPU thread
            const int64_t lp = gid / rps;
            const int lr = (int)(gid % rps);
            const int64_t source = n_min + lp;
            if (source >= num_particles) continue;                      // .cu:1967
            RayOut out{mk3(NANF, NANF, NANF), mk3(NANF, NANF, NANF)};
            const bool dump = save_lightrays && gid < num_lightrays_save;
            InterRec ir;
            if (inter && dump) ir = InterRec{&ipos[(size_t)gid * inter_slots], &idir[(size_t)gid * inter_slots], inter_slots};
            trace_one(sc, volp, ray_tracing_algorithm, source, lr, imgs[tid], mc_local, dump ? &out : nullptr,
                      on_local, &ir);
            if (dump) { fpos[gid] = out.pos; fdir[gid] = out.dir; }
        }
        mcs[tid].iterations += mc_local.iterations; mcs[tid].samples += mc_local.samples;
        on_sensor[tid] += on_local;
        }
        if (save_lightrays) {
            write_dump(lightray_position_save_path, "pos_", (int)k, fpos);
            write_dump(lightray_direction_save_path, "dir_", (int)k, fdir);
        }
        if (inter) {                                                    // .cu:3613-3670
            write_dump(lightray_position_save_path, "intermediate_pos_", (int)k, ipos);
            write_dump(lightray_direction_save_path, "intermediate_dir_", (int)k, idir);
        }
    }
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < (size_t)W * H; p++) {
        double s = 0;
        for (int t = 0; t < nthreads; t++) s += imgs[t].acc[p];
        image_array[p] = (float)((double)image_array[p] + s);
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->rays_launched = (uint64_t)num_particles * rps;
        for (int t = 0; t < nthreads; t++) {
            stats->rays_on_sensor += on_sensor[t];
            stats->rk_iterations += mcs[t].iterations;
            stats->volume_samples += mcs[t].samples;
            stats->sensor_taps += imgs[t].taps;
        }
    }
}

// The reference's start_ray_tracing (parallel_ray_tracing.cu:3078-3775) on the CPU, plus two
// trailing knobs the reference hard-codes: interpolation (1 linear = reference, 2 cubic) and
// tex_frac_bits (0 exact, 8 = NVIDIA texture-unit weights).  stats may be NULL.
void oracle_start_ray_tracing(float lens_pitch, float image_distance, scattering_data_t *scattering_data_p,
                              char *scattering_type_str, lightfield_source_t *lightfield_source_p,
                              int lightray_number_per_particle, float beam_wavelength, float aperture_f_number,
                              int num_elements, double (*element_center)[3], element_data_t *element_data_p,
                              double (*element_plane_parameters)[4], int *element_system_index,
                              camera_design_t *camera_design_p, float *image_array,
                              bool simulate_density_gradients, char *density_grad_filename, bool save_lightrays,
                              char *lightray_position_save_path, char *lightray_direction_save_path,
                              int num_lightrays_save, int ray_tracing_algorithm, bool add_pos_noise,
                              float pos_noise_std, bool add_ngrad_noise, float ngrad_noise_std,
                              float ray_cone_pitch_ratio, bool save_intermediate_ray_data,
                              int num_intermediate_positions_save, int interpolation, int tex_frac_bits,
                              oracle_stats_t *stats) {
    Scene sc;
    build_scene(sc, lens_pitch, image_distance, scattering_data_p, scattering_type_str, lightfield_source_p,
                lightray_number_per_particle, beam_wavelength, aperture_f_number, num_elements, element_center,
                element_data_p, element_plane_parameters, element_system_index, camera_design_p,
                ray_cone_pitch_ratio);
    sc.noise.add_pos = add_pos_noise; sc.noise.pos_std = pos_noise_std;
    sc.noise.add_ngrad = add_ngrad_noise; sc.noise.ngrad_std = ngrad_noise_std;
    sc.noise.seed = g_noise_seed;
    Volume vol;
    const Volume *volp = nullptr;
    if (simulate_density_gradients) {
        std::vector<float> rho; int dims[3]; double sp[3], org[3];
        if (!read_nrrd(density_grad_filename, rho, dims, sp, org)) {
            fprintf(stderr, "oracle: cannot read NRRD '%s'\n", density_grad_filename);
            return;
        }
        setup_volume(vol, rho.data(), dims[0], dims[1], dims[2], sp, org, interpolation, tex_frac_bits);
        volp = &vol;
    }
    render_core(sc, volp, image_array, save_lightrays, lightray_position_save_path, lightray_direction_save_path,
                num_lightrays_save, ray_tracing_algorithm, stats,
                save_intermediate_ray_data ? num_intermediate_positions_save : 0);
}

// Same, with a volume built beforehand by oracle_volume_* (NULL = no density gradients): lets the
// CPU baseline time the ray loop without the volume build, like photon_trace on the GPU side.
void oracle_render_with_volume(float lens_pitch, float image_distance, scattering_data_t *scattering_data_p,
                               char *scattering_type_str, lightfield_source_t *lightfield_source_p,
                               int lightray_number_per_particle, float beam_wavelength, float aperture_f_number,
                               int num_elements, double (*element_center)[3], element_data_t *element_data_p,
                               double (*element_plane_parameters)[4], int *element_system_index,
                               camera_design_t *camera_design_p, float *image_array,
                               bool simulate_density_gradients, char *density_grad_filename, bool save_lightrays,
                               char *lightray_position_save_path, char *lightray_direction_save_path,
                               int num_lightrays_save, int ray_tracing_algorithm, bool add_pos_noise,
                               float pos_noise_std, bool add_ngrad_noise, float ngrad_noise_std,
                               float ray_cone_pitch_ratio, bool save_intermediate_ray_data,
                               int num_intermediate_positions_save, void *volume, oracle_stats_t *stats) {
    Scene sc;
    build_scene(sc, lens_pitch, image_distance, scattering_data_p, scattering_type_str, lightfield_source_p,
                lightray_number_per_particle, beam_wavelength, aperture_f_number, num_elements, element_center,
                element_data_p, element_plane_parameters, element_system_index, camera_design_p,
                ray_cone_pitch_ratio);
    sc.noise.add_pos = add_pos_noise; sc.noise.pos_std = pos_noise_std;
    sc.noise.add_ngrad = add_ngrad_noise; sc.noise.ngrad_std = ngrad_noise_std;
    sc.noise.seed = g_noise_seed;
    render_core(sc, simulate_density_gradients ? static_cast<const Volume *>(volume) : nullptr, image_array,
                save_lightrays, lightray_position_save_path, lightray_direction_save_path, num_lightrays_save,
                ray_tracing_algorithm, stats, save_intermediate_ray_data ? num_intermediate_positions_save : 0);
}

void oracle_rand_table(int n, float *r1, float *r2) { rand_table(n, r1, r2); }

// ---- volume: build / sample / march -------------------------------------------------------
void *oracle_volume_from_density(const float *rho, int nx, int ny, int nz, const double spacing[3],
                                 const double origin[3], int interpolation, int tex_frac_bits) {
    Volume *v = new Volume();
    setup_volume(*v, rho, nx, ny, nz, spacing, origin, interpolation, tex_frac_bits);
    return v;
}
void *oracle_volume_load_nrrd(const char *path, int interpolation, int tex_frac_bits) {
    std::vector<float> rho; int dims[3]; double sp[3], org[3];
    if (!read_nrrd(path, rho, dims, sp, org)) return nullptr;
    return oracle_volume_from_density(rho.data(), dims[0], dims[1], dims[2], sp, org, interpolation, tex_frac_bits);
}
void oracle_volume_info(void *vp, photon_volume_info_t *info) {
    const Volume &v = *static_cast<Volume *>(vp);
    info->min_bound[0] = v.min_bound.x; info->min_bound[1] = v.min_bound.y; info->min_bound[2] = v.min_bound.z;
    info->max_bound[0] = v.max_bound.x; info->max_bound[1] = v.max_bound.y; info->max_bound[2] = v.max_bound.z;
    info->nx = v.nx; info->ny = v.ny; info->nz = v.nz;
    info->grid_spacing[0] = v.grid_spacing.x; info->grid_spacing[1] = v.grid_spacing.y;
    info->grid_spacing[2] = v.grid_spacing.z;
    info->step_size = v.step_size; info->data_min = v.data_min; info->interpolation = v.interpolation;
}
void oracle_volume_download(void *vp, int coefficients, float *out) {
    const Volume &v = *static_cast<Volume *>(vp);
    const std::vector<f4> &src = (coefficients && v.interpolation == 2) ? v.coeffs : v.data;
    memcpy(out, src.data(), src.size() * sizeof(f4));
}
void oracle_volume_sample(void *vp, int n, const float *coords, float *out) {
    const Volume &v = *static_cast<Volume *>(vp);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        const f4 r = sample(v, mk3(coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]));
        out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
    }
}
void oracle_trace_volume_rays(void *vp, int algorithm, int n, float *pos, float *dir, int *steps) {
    const Volume &v = *static_cast<Volume *>(vp);
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        f3 p = mk3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        f3 d = mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
        MarchCount mc;
        trace_volume(p, d, v, algorithm, mc);
        pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z;
        dir[3 * i] = d.x; dir[3 * i + 1] = d.y; dir[3 * i + 2] = d.z;
        if (steps) steps[i] = mc.iterations;
    }
}
void oracle_volume_free(void *vp) { delete static_cast<Volume *>(vp); }

// ---- optics unit entry points (arrays of n rays, [n][3] layout) ---------------------------
void oracle_ray_sphere_intersection(int n, const float *center, float R, const float *dir, const float *pos,
                                    char surface, float *out) {
    for (int i = 0; i < n; i++) {
        const f3 r = ray_sphere_intersection(mk3(center[0], center[1], center[2]), R,
                                             mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]),
                                             mk3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), surface);
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
void oracle_axis_distance(int n, const float *pts, const float *center, const float *plane, float *out) {
    for (int i = 0; i < n; i++)
        out[i] = axis_distance(mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), mk3(center[0], center[1], center[2]), plane);
}
void oracle_single_element(int n, const element_data_t *e, const float *center, const float *plane, float *pos,
                           float *dir, float wavelength, double *radiance) {
    for (int i = 0; i < n; i++) {
        Ray r;
        r.pos = mk3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        r.dir = mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
        r.wavelength = wavelength; r.radiance = radiance[i];
        r = single_element(*e, mk3(center[0], center[1], center[2]), plane, r);
        pos[3 * i] = r.pos.x; pos[3 * i + 1] = r.pos.y; pos[3 * i + 2] = r.pos.z;
        dir[3 * i] = r.dir.x; dir[3 * i + 1] = r.dir.y; dir[3 * i + 2] = r.dir.z;
        radiance[i] = r.radiance;
    }
}
// whole element train (optical_system above) on n rays; train_mode as oracle_set_element_train
void oracle_optical_system(int n, const element_data_t *elems, const float *centers, const float *planes,
                           const int *sys_index, int num_elements, int train_mode, float *pos, float *dir,
                           float wavelength, double *radiance) {
    for (int i = 0; i < n; i++) {
        Ray r;
        r.pos = mk3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        r.dir = mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
        r.wavelength = wavelength; r.radiance = radiance[i];
        r = optical_system(elems, reinterpret_cast<const float(*)[3]>(centers), reinterpret_cast<const float(*)[4]>(planes),
                           sys_index, num_elements, r, train_mode);
        pos[3 * i] = r.pos.x; pos[3 * i + 1] = r.pos.y; pos[3 * i + 2] = r.pos.z;
        dir[3 * i] = r.dir.x; dir[3 * i + 1] = r.dir.y; dir[3 * i + 2] = r.dir.z;
        radiance[i] = r.radiance;
    }
}
// one generated ray per (source 0, lens sample k): pos/dir [n][3], radiance [n]
void oracle_generate_rays(float lens_pitch, float image_distance, scattering_data_t *sd, int scattering_type,
                          float sx, float sy, float sz, double sradiance, int diameter_index, int n,
                          float beam_wavelength, float f_number, const float *r1, const float *r2, float ratio,
                          float *pos, float *dir, double *radiance) {
    for (int k = 0; k < n; k++) {
        Source s{sx, sy, sz, sradiance, diameter_index};
        const Ray r = generate_ray(lens_pitch, image_distance, *sd, scattering_type, s, n, beam_wavelength,
                                   f_number, r1[k], r2[k], ratio);
        pos[3 * k] = r.pos.x; pos[3 * k + 1] = r.pos.y; pos[3 * k + 2] = r.pos.z;
        dir[3 * k] = r.dir.x; dir[3 * k + 1] = r.dir.y; dir[3 * k + 2] = r.dir.z;
        radiance[k] = r.radiance;
    }
}

// pixel indices and area weights of n sensor hits (x, y): ii/jj int[n][4], w double[n][4], inside int[n]
void oracle_pixel_taps(int n, const float *x, const float *y, const camera_design_t *cam, int *ii, int *jj, double *w,
                       int *inside) {
    for (int k = 0; k < n; k++) {
        const PixelTaps t = pixel_taps(mk3(x[k], y[k], cam->z_sensor), *cam);
        inside[k] = t.inside;
        for (int q = 0; q < 4; q++) { ii[4 * k + q] = t.ii[q]; jj[4 * k + q] = t.jj[q]; w[4 * k + q] = t.w[q]; }
    }
}

// photon_det_math.h evaluated on the host, for tests/test_det_math.py.
// fn: 0 sin 1 cos 2 tan 3 atan 4 acos (double);  10 atanf 11 tanf 12 acosf 13 cosf (float in/out as double)
void oracle_det_eval(int fn, int n, const double *x, double *y) {
    for (int i = 0; i < n; i++) {
        switch (fn) {
            case 0: y[i] = photon_det_sin(x[i]); break;
            case 1: y[i] = photon_det_cos(x[i]); break;
            case 2: y[i] = photon_det_tan(x[i]); break;
            case 3: y[i] = photon_det_atan(x[i]); break;
            case 4: y[i] = photon_det_acos(x[i]); break;
            case 10: y[i] = photon_det_atanf((float)x[i]); break;
            case 11: y[i] = photon_det_tanf((float)x[i]); break;
            case 12: y[i] = photon_det_acosf((float)x[i]); break;
            case 13: y[i] = photon_det_cosf((float)x[i]); break;
            default: y[i] = 0; break;
        }
    }
}

// ---- scene generation (the product's Section 3 entry points, restated on the CPU) -----------
// BOS target: generate_bos_lightfield_data, run_simulation_02.py:1328-1551 (x = dot + template, in double,
// stored f32 by the ctypes marshalling).
void oracle_sources_bos(const double *dot_x, const double *dot_y, int n_dots, const double *tx, const double *ty,
                        int n_tmpl, double z, double radiance, float *x, float *y, float *zz, double *rad, int *dia) {
    for (long long g = 0; g < n_dots; g++)
        for (int j = 0; j < n_tmpl; j++) {
            const long long i = g * n_tmpl + j;
            x[i] = (float)(dot_x[g] + tx[j]);
            y[i] = (float)(dot_y[g] + ty[j]);
            zz[i] = (float)z;
            rad[i] = radiance;
            dia[i] = 1;
        }
}
// PIV particle field: run_simulation_02.py:774-996 with Philox(seed, i) in place of numpy's unseeded generator.
void oracle_sources_piv(uint64_t seed, long long n, const double lo[3], const double hi[3], double z_object,
                        double beam_fwhm, double irradiance_constant, const double *cdf, int n_diameters, float *x,
                        float *y, float *z, double *rad, int *dia) {
    const double sigma = beam_fwhm / (2.0 * sqrt(2.0 * log(2.0)));     // :961
    const double coef = irradiance_constant * (1.0 / (sigma * sqrt(2.0 * PHOTON_PI)));
    const double two_sigma2 = 2.0 * (sigma * sigma);
    for (long long i = 0; i < n; i++) {
        const photon_u32x4 r = photon_philox4x32_10(seed, (uint64_t)i, 0u, PHOTON_STREAM_SCENE);
        const double ux = ((double)r.x + 0.5) * (1.0 / 4294967296.0), uy = ((double)r.y + 0.5) * (1.0 / 4294967296.0);
        const double uz = ((double)r.z + 0.5) * (1.0 / 4294967296.0), ud = ((double)r.w + 0.5) * (1.0 / 4294967296.0);
        const double X = (hi[0] - lo[0]) * ux + lo[0];                  // :949-951
        const double Y = (hi[1] - lo[1]) * uy + lo[1];
        const double Z = (hi[2] - lo[2]) * uz + lo[2];
        x[i] = (float)X;
        y[i] = (float)Y;
        z[i] = (float)(Z + z_object);                                   // :965
        rad[i] = coef * photon_det_exp(-1.0 * (Z * Z / two_sigma2));    // :962
        int d = 1;                                                      // :992
        if (n_diameters > 0) {
            d = n_diameters - 1;
            for (int q = 0; q < n_diameters; q++)
                if (ud < cdf[q]) { d = q; break; }
        }
        dia[i] = d;
    }
}
// Synthetic Gaussian density field -> volume (what nrrd_functions.py:14-57 would write and loadNRRD read back).
void *oracle_volume_gaussian(int nx, int ny, int nz, const double spacing[3], const double origin[3], double rho0,
                             double amp, const double centre[3], double sigma, int interpolation, int tex_frac_bits) {
    const int dims[3] = {nx, ny, nz};
    std::vector<double> prof[3];
    for (int a = 0; a < 3; a++) {
        prof[a].resize(dims[a]);
        for (int i = 0; i < dims[a]; i++) {
            const double x = origin[a] + spacing[a] * (double)i;
            prof[a][i] = photon_det_exp(-((x - centre[a]) * (x - centre[a])) / (2 * (sigma * sigma)));
        }
    }
    std::vector<float> rho((size_t)nx * ny * nz);
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++)
                rho[((size_t)k * ny + j) * nx + i] = (float)(rho0 + amp * prof[2][k] * (prof[1][j] * prof[0][i]));
    Volume *v = new Volume();
    setup_volume(*v, rho.data(), nx, ny, nz, spacing, origin, interpolation, tex_frac_bits);
    return v;
}

void oracle_set_noise_seed(uint64_t seed) { g_noise_seed = seed; }
void oracle_set_element_train(int mode) { g_element_train = mode; }

// include/photon_philox.h on the host: n pairs of N(0,1) for rays 0..n-1 (tests)
void oracle_normal2(uint64_t seed, int n, uint32_t draw, uint32_t stream, float *out) {
    for (int i = 0; i < n; i++) photon_normal2(seed, (uint64_t)i, draw, stream, &out[2 * i], &out[2 * i + 1]);
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
}

// photon_det_div_rcp against the division it replaces in the product's erf splat (tests/test_det_math.py): number of
// quotients a[i] / b that differ
long long oracle_det_div_rcp_mismatches(long long n, const double *a, double b) {
    const double rb = 1.0 / b;
    long long bad = 0;
    for (long long i = 0; i < n; i++) bad += !(photon_det_div_rcp(a[i], b, rb) == a[i] / b);
    return bad;
}

// photon_det_erf as this build of the header evaluates it (tests/test_det_math.py)
void oracle_det_erf(int n, const double *x, double *out) {
    for (int i = 0; i < n; i++) out[i] = photon_det_erf(x[i]);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
