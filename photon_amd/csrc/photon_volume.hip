// photon_volume.hip - the refractive-index-gradient volume: NRRD parser, the kernels that build the float4 texels
// (grad n, n-1) and their cubic B-spline coefficients, the volume handle API and the per-device volume cache of
// start_ray_tracing.  HBM-bound one-off work (0.15 + 0.7 ms at 256^3), cached across calls.
#include <zlib.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>

#include "photon_internal.hpp"

using namespace photon;

// =============================================================================================
// volume construction kernels
// =============================================================================================

// setData (trace_rays_through_density_gradients.h:1820-2002) with loadNRRD's Gladstone-Dale
// scaling (.h:1729-1748) folded in: one thread per voxel; edges use the double-precision
// one-sided stencils, the interior the f32 (x, z) / f64-divisor (y) central differences.
__global__ __launch_bounds__(256) void build_volume_kernel(const float *__restrict__ rho, int W, int H, int D,
                                                           float gx, float gy, float gz, f4 *__restrict__ out,
                                                           float *__restrict__ block_min) {
    const size_t n = (size_t)W * H * D;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float K = 0.225e-3;
    float mine = FLT_MAX, gmag = 0.f;
    if (i < n) {
        const int x = (int)(i % W), y = (int)((i / W) % H), z = (int)(i / ((size_t)W * H));
        const size_t WH = (size_t)W * H;
        auto d = [&](int xx, int yy, int zz) { return K * (rho[zz * WH + (size_t)yy * W + xx] * 1.0f); };
        float nxv, nyv, nzv, s1, s2, s3;
        if (x < 1) {
            s1 = d(x, y, z); s2 = d(x + 1, y, z); s3 = d(x + 2, y, z);
            nxv = (float)((-3.0 / 2 * s1 + 2 * s2 - 1.0 / 2 * s3) / gx);
        } else if (x >= W - 1) {
            s1 = d(x, y, z); s2 = d(x - 1, y, z); s3 = d(x - 2, y, z);
            nxv = (float)((3.0 / 2 * s1 - 2 * s2 + 1.0 / 2 * s3) / gx);
        } else {
            s1 = d(x - 1, y, z); s2 = d(x + 1, y, z);
            nxv = (s2 - s1) / (2 * gx);
        }
        if (y < 1) {
            s1 = d(x, y, z); s2 = d(x, y + 1, z); s3 = d(x, y + 2, z);
            nyv = (float)((-3.0 / 2 * s1 + 2 * s2 - 1.0 / 2 * s3) / gy);
        } else if (y >= H - 1) {
            s1 = d(x, y, z); s2 = d(x, y - 1, z); s3 = d(x, y - 2, z);
            nyv = (float)((3.0 / 2 * s1 - 2 * s2 + 1.0 / 2 * s3) / gy);
        } else {
            s1 = d(x, y - 1, z); s2 = d(x, y + 1, z);
            nyv = (float)((s2 - s1) / (2.0 * gy));
        }
        if (z < 1) {
            s1 = d(x, y, z); s2 = d(x, y, z + 1); s3 = d(x, y, z + 2);
            nzv = (float)((-3.0 / 2 * s1 + 2 * s2 - 1.0 / 2 * s3) / gz);
        } else if (z >= D - 1) {
            s1 = d(x, y, z); s2 = d(x, y, z - 1); s3 = d(x, y, z - 2);
            nzv = (float)((3.0 / 2 * s1 - 2 * s2 + 1.0 / 2 * s3) / gz);
        } else {
            s1 = d(x, y, z - 1); s2 = d(x, y, z + 1);
            nzv = (s2 - s1) / (2 * gz);
        }
        const float w = d(x, y, z);
        *reinterpret_cast<float4 *>(out + i) = make_float4(nxv, nyv, nzv, w);
        mine = w;
        gmag = sqrtf(nxv * nxv + nyv * nyv + nzv * nzv);
    }
    // block minimum of n-1 (data_min, .h:1868-1869)
    __shared__ float red[256];
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fminf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) block_min[blockIdx.x] = red[0];
    // block maximum of |grad n| (bounds how far the volume can bend a ray: launch_chunk's doom margin)
    __syncthreads();
    red[threadIdx.x] = gmag == gmag ? gmag : 0.f;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) block_min[gridDim.x + blockIdx.x] = red[0];
}

// rho[k][j][i] = rho0 + amp * gz[k] * (gy[j] * gx[i]) in double, rounded once to f32 (the order of the
// numpy expression a host-side generator would use): the density of photon_volume_gaussian.
__global__ __launch_bounds__(256) void separable_density_kernel(const double *__restrict__ gx, const double *__restrict__ gy,
                                                                const double *__restrict__ gz, int W, int H, int D, double rho0,
                                                                double amp, float *__restrict__ rho) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)W * H * D) return;
    const int i = (int)(idx % W), j = (int)((idx / W) % H), k = (int)(idx / ((size_t)W * H));
    rho[idx] = (float)(rho0 + amp * gz[k] * (gy[j] * gx[i]));
}

// ConvertToInterpolationCoefficients (cubicPrefilter_kernel.cu:52-112) on all four channels of one
// line of float4 texels, in place.  One thread per line; `lines_inner` lines are adjacent in
// memory by `inner_stride` texels (coalesced for the y and z passes).
__global__ __launch_bounds__(256) void prefilter_lines_kernel(f4 *vol, int len, size_t len_stride, int lines_inner,
                                                              size_t inner_stride, int lines_outer,
                                                              size_t outer_stride) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)lines_inner * lines_outer) return;
    const size_t li = t % lines_inner, lo = t / lines_inner;
    f4 *c = vol + lo * outer_stride + li * inner_stride;
    const float Pole = sqrtf(3.0f) - 2.0f;
    const float Lambda = (1.0f - Pole) * (1.0f - 1.0f / Pole);
    const int horizon = len < 12 ? len : 12;
    float zn = Pole;
    f4 first = c[0];
    f4 sum = first;
    for (int k = 0; k < horizon; k++) {
        const f4 v = c[k * len_stride];
        sum.x += zn * v.x; sum.y += zn * v.y; sum.z += zn * v.z; sum.w += zn * v.w;
        zn *= Pole;
    }
    f4 prev = f4{Lambda * sum.x, Lambda * sum.y, Lambda * sum.z, Lambda * sum.w};
    c[0] = prev;
    for (int k = 1; k < len; k++) {
        const f4 v = c[k * len_stride];
        prev = f4{Lambda * v.x + Pole * prev.x, Lambda * v.y + Pole * prev.y, Lambda * v.z + Pole * prev.z,
                  Lambda * v.w + Pole * prev.w};
        c[k * len_stride] = prev;
    }
    const float g = Pole / (Pole - 1.0f);
    const f4 last = c[(size_t)(len - 1) * len_stride];
    prev = f4{g * last.x, g * last.y, g * last.z, g * last.w};
    c[(size_t)(len - 1) * len_stride] = prev;
    for (int k = len - 2; k >= 0; k--) {
        const f4 v = c[k * len_stride];
        prev = f4{Pole * (prev.x - v.x), Pole * (prev.y - v.y), Pole * (prev.z - v.z), Pole * (prev.w - v.w)};
        c[k * len_stride] = prev;
    }
}

__global__ __launch_bounds__(256) void sample_kernel(VolumeDev v, int n, const float *__restrict__ coords,
                                                     float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = coords[3 * i], y = coords[3 * i + 1], z = coords[3 * i + 2];
    const f4 r = v.interpolation == 2 ? tex3d_cubic(v, x, y, z) : tex3d_linear(v, x, y, z);
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

static bool parse_nrrd(const char *path, std::vector<float> &rho, int dims[3], double spacing[3], double origin[3],
                       std::string &why) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { why = "cannot open file"; return false; }
    std::string line;
    if (!std::getline(f, line) || line.rfind("NRRD", 0) != 0) { why = "missing NRRD magic"; return false; }
    std::string type, encoding = "raw", endian = "little";
    int dimension = 0;
    bool sizes_ok = false;
    for (int a = 0; a < 3; a++) { spacing[a] = 1.0; origin[a] = 0.0; }
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) break;                        // blank line ends the header
        if (line[0] == '#') continue;
        const size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        const std::string key = line.substr(0, colon);
        size_t vs = colon + 1;
        if (vs < line.size() && line[vs] == '=') vs++;  // "key:=value" pairs
        while (vs < line.size() && line[vs] == ' ') vs++;
        const std::string val = line.substr(vs);
        if (key == "type") type = val;
        else if (key == "dimension") dimension = atoi(val.c_str());
        else if (key == "encoding") encoding = val;
        else if (key == "endian") endian = val;
        else if (key == "sizes") sizes_ok = sscanf(val.c_str(), "%d %d %d", &dims[0], &dims[1], &dims[2]) == 3;
        else if (key == "spacings") sscanf(val.c_str(), "%lf %lf %lf", &spacing[0], &spacing[1], &spacing[2]);
        else if (key == "space origin") sscanf(val.c_str(), " (%lf,%lf,%lf)", &origin[0], &origin[1], &origin[2]);
        else if (key == "space directions") {
            double m[9];
            if (sscanf(val.c_str(), " (%lf,%lf,%lf) (%lf,%lf,%lf) (%lf,%lf,%lf)", &m[0], &m[1], &m[2], &m[3], &m[4],
                       &m[5], &m[6], &m[7], &m[8]) == 9)
                for (int a = 0; a < 3; a++)
                    spacing[a] = std::sqrt(m[3 * a] * m[3 * a] + m[3 * a + 1] * m[3 * a + 1] + m[3 * a + 2] * m[3 * a + 2]);
        }
    }
    if (dimension != 3 || !sizes_ok) { why = "need dimension 3 with three sizes"; return false; }
    if (type != "float") { why = "type must be float (single precision)"; return false; }
    // payload encodings teem's nrrdLoad (what the reference reads its volume with, .h:1687) takes and people's tools write: raw
    // (photon's own writer, nrrd_functions.py:49), gzip (pynrrd's default), ascii; either byte order
    const bool is_raw = encoding == "raw", is_gzip = encoding == "gzip" || encoding == "gz",
               is_ascii = encoding == "ascii" || encoding == "text" || encoding == "txt";
    if (!is_raw && !is_gzip && !is_ascii) { why = "encoding must be raw, gzip or ascii"; return false; }
    if (endian != "little" && endian != "big") { why = "endian must be little or big"; return false; }
    if (dims[0] < 3 || dims[1] < 3 || dims[2] < 3) { why = "each axis needs at least 3 samples"; return false; }
    // a corrupt header must not drive the allocation: the payload has to be in the file
    if (dims[0] > 65536 || dims[1] > 65536 || dims[2] > 65536) { why = "sizes beyond 65536 per axis"; return false; }
    const unsigned long long count = (unsigned long long)dims[0] * dims[1] * dims[2];
    const std::streamoff here = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streamoff total = f.tellg();
    f.seekg(here, std::ios::beg);
    if (here < 0 || total < here) { why = "payload shorter than sizes"; return false; }
    const unsigned long long in_file = (unsigned long long)(total - here);
    if (is_raw) {
        if (in_file < count * sizeof(float)) { why = "payload shorter than sizes"; return false; }
        rho.resize((size_t)count);
        f.read(reinterpret_cast<char *>(rho.data()), (std::streamsize)(rho.size() * sizeof(float)));
        if ((size_t)f.gcount() != rho.size() * sizeof(float)) { why = "payload shorter than sizes"; return false; }
    } else if (is_ascii) {
        // at least two characters per value (a digit and a separator) have to be there before anything is allocated
        if (in_file < 2 * count - 1) { why = "payload shorter than sizes"; return false; }
        rho.resize((size_t)count);
        for (size_t i = 0; i < rho.size(); i++)
            if (!(f >> rho[i])) { why = "payload shorter than sizes"; return false; }
    } else {
        // gzip: the compressed bytes are in the file; deflate's best ratio is ~1032 : 1, so sizes a file cannot hold are refused
        // before the allocation, and the stream is never inflated beyond what the sizes ask for
        if (in_file < 18 || count * sizeof(float) / 1032ull > in_file) { why = "payload shorter than sizes"; return false; }
        std::vector<unsigned char> z((size_t)in_file);
        f.read(reinterpret_cast<char *>(z.data()), (std::streamsize)z.size());
        if ((size_t)f.gcount() != z.size()) { why = "cannot read the compressed payload"; return false; }
        rho.resize((size_t)count);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 15 + 32) != Z_OK) { why = "zlib initialisation failed"; return false; }     // zlib or gzip wrapper
        const unsigned long long want = count * sizeof(float);
        unsigned long long got = 0, fed = 0;
        int zr = Z_OK;
        while (zr == Z_OK && got < want) {                              // in pieces of < 4 GiB: zlib counts in 32 bits
            if (zs.avail_in == 0) {
                const unsigned long long piece = std::min<unsigned long long>(z.size() - fed, 1ull << 30);
                if (piece == 0) break;
                zs.next_in = z.data() + fed; zs.avail_in = (uInt)piece; fed += piece;
            }
            const unsigned long long room = std::min<unsigned long long>(want - got, 1ull << 30);
            zs.next_out = reinterpret_cast<unsigned char *>(rho.data()) + got; zs.avail_out = (uInt)room;
            zr = inflate(&zs, Z_NO_FLUSH);
            got += room - zs.avail_out;
        }
        inflateEnd(&zs);
        if ((zr != Z_OK && zr != Z_STREAM_END) || got != want) { why = "compressed payload is damaged or shorter than sizes"; return false; }
    }
    if (endian == "big" && !is_ascii) {
        unsigned *u = reinterpret_cast<unsigned *>(rho.data());
        for (size_t i = 0; i < rho.size(); i++) u[i] = __builtin_bswap32(u[i]);
    }
    return true;
}

extern "C" {

void photon_volume_free(photon_volume_t *vol) {
    if (!vol) return;
    if (vol->d_texels) (void)hipFree(vol->d_texels);
    if (vol->d_coeffs) (void)hipFree(vol->d_coeffs);
    delete vol;
}

// Where the density comes from: a host array (NRRD / caller) or a field evaluated on the device
// (photon_volume_gaussian): rho0 + amp * gz[k] * (gy[j] * gx[i]) from three device-resident axis profiles.
struct DensitySource {
    const float *host_rho = nullptr;
    const double *d_gx = nullptr, *d_gy = nullptr, *d_gz = nullptr;
    double rho0 = 0, amp = 0;
};

static int volume_build(const DensitySource &src, int nx, int ny, int nz, const double spacing[3],
                        const double origin[3], int interpolation, photon_volume_t **out);

int photon_volume_from_density(const float *rho, int nx, int ny, int nz, const double spacing[3],
                               const double origin[3], int interpolation, photon_volume_t **out) {
    if (!rho) {
        fprintf(stderr, "photon: photon_volume_from_density: bad arguments\n");
        return 1;
    }
    DensitySource src;
    src.host_rho = rho;
    return guarded("photon_volume_from_density", [&]() -> int { return volume_build(src, nx, ny, nz, spacing, origin, interpolation, out); });
}

// Synthetic density field evaluated on the device: rho = rho0 + amp * exp(-|r - centre|^2 / (2 sigma^2)),
// separable, so the host prepares three axis profiles (O(n) work, photon_det_exp) and a kernel fills the
// n^3 grid in HBM -- no host array, no file, no upload (BASELINE C3 / C4's volume).
// the three axis profiles of the separable Gaussian, on the device
static int gaussian_profiles(int nx, int ny, int nz, const double spacing[3], const double origin[3],
                             const double centre[3], double sigma, double *d_prof[3]) {
    const int dims[3] = {nx, ny, nz};
    for (int a = 0; a < 3; a++) {
        std::vector<double> prof(dims[a]);
        for (int i = 0; i < dims[a]; i++) {
            const double x = origin[a] + spacing[a] * (double)i;
            prof[i] = photon_det_exp(-((x - centre[a]) * (x - centre[a])) / (2 * (sigma * sigma)));
        }
        if (device_malloc((void **)&d_prof[a], dims[a] * sizeof(double)) != hipSuccess ||
            hipMemcpy(d_prof[a], prof.data(), dims[a] * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 3;
    }
    return 0;
}

int photon_volume_gaussian(int nx, int ny, int nz, const double spacing[3], const double origin[3], double rho0,
                           double amp, const double centre[3], double sigma, int interpolation,
                           photon_volume_t **out) {
    if (!spacing || !origin || !centre || !(sigma > 0) || nx < 3 || ny < 3 || nz < 3) {
        fprintf(stderr, "photon: photon_volume_gaussian: bad arguments\n");
        return 1;
    }
    double *d_prof[3] = {nullptr, nullptr, nullptr};
    int rc = guarded("photon_volume_gaussian", [&]() -> int { return gaussian_profiles(nx, ny, nz, spacing, origin, centre, sigma, d_prof); });
    if (!rc) {
        DensitySource src;
        src.d_gx = d_prof[0]; src.d_gy = d_prof[1]; src.d_gz = d_prof[2];
        src.rho0 = rho0; src.amp = amp;
        rc = guarded("photon_volume_gaussian", [&]() -> int { return volume_build(src, nx, ny, nz, spacing, origin, interpolation, out); });
    } else {
        fprintf(stderr, "photon: photon_volume_gaussian: device allocation failed\n");
    }
    for (double *p : d_prof) if (p) (void)hipFree(p);
    return rc;
}

// The same field written as an NRRD file (what nrrd_functions.py:14-57 writes with pynrrd and loadNRRD reads
// back: type float, dimension 3, raw, little endian, sizes / spacings / space origin): evaluated on the device,
// streamed to disk.  For feeding synthetic volumes to code that wants a file -- photon itself included.
int photon_density_gaussian_write_nrrd(const char *path, int nx, int ny, int nz, const double spacing[3],
                                       const double origin[3], double rho0, double amp, const double centre[3],
                                       double sigma) {
    if (!path || !spacing || !origin || !centre || !(sigma > 0) || nx < 1 || ny < 1 || nz < 1) {
        fprintf(stderr, "photon: photon_density_gaussian_write_nrrd: bad arguments\n");
        return 1;
    }
    double *d_prof[3] = {nullptr, nullptr, nullptr};
    float *d_rho = nullptr;
    const size_t n = (size_t)nx * ny * nz;
    std::vector<float> rho;
    int rc = guarded("photon_density_gaussian_write_nrrd", [&]() -> int {
        rho.resize(n);
        return gaussian_profiles(nx, ny, nz, spacing, origin, centre, sigma, d_prof);
    });
    if (!rc && device_malloc((void **)&d_rho, n * sizeof(float)) != hipSuccess) rc = 3;
    if (!rc) {
        hipLaunchKernelGGL(separable_density_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_prof[0], d_prof[1],
                           d_prof[2], nx, ny, nz, rho0, amp, d_rho);
        if (hipGetLastError() != hipSuccess || hipMemcpy(rho.data(), d_rho, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = 4;
    }
    for (double *p : d_prof) if (p) (void)hipFree(p);
    if (d_rho) (void)hipFree(d_rho);
    if (rc) {
        fprintf(stderr, "photon: photon_density_gaussian_write_nrrd: HIP error\n");
        return rc;
    }
    std::ofstream f(path, std::ios::out | std::ios::binary);
    if (!f) { fprintf(stderr, "photon: cannot write %s\n", path); return 2; }
    char header[512];
    snprintf(header, sizeof header,
             "NRRD0005\n# written by photon_density_gaussian_write_nrrd\ntype: float\ndimension: 3\nspace: 3D-left-handed\n"
             "sizes: %d %d %d\nendian: little\nencoding: raw\nspacings: %.17g %.17g %.17g\nspace origin: (%.17g,%.17g,%.17g)\n\n",
             nx, ny, nz, spacing[0], spacing[1], spacing[2], origin[0], origin[1], origin[2]);
    f.write(header, (std::streamsize)strlen(header));
    f.write(reinterpret_cast<const char *>(rho.data()), (std::streamsize)(n * sizeof(float)));
    return f ? 0 : 2;
}

static int volume_build(const DensitySource &src, int nx, int ny, int nz, const double spacing[3],
                        const double origin[3], int interpolation, photon_volume_t **out) {
    if (!out || !spacing || !origin || nx < 3 || ny < 3 || nz < 3 || (interpolation != 1 && interpolation != 2)) {
        fprintf(stderr, "photon: volume: bad arguments\n");
        return 1;
    }
    // bounds from the file's own size (loadNRRD, .h:1696-1706), then the 1024-slice cap (.h:1714-1717)
    const double xmin = origin[0], ymin = origin[1], zmin = origin[2] - 750e3;
    const double xmax = xmin + (nx - 1) * spacing[0], ymax = ymin + (ny - 1) * spacing[1];
    const double zmax = zmin + (nz - 1) * spacing[2];
    if (nz > 1024) nz = 1024;
    if ((unsigned long long)(nx + 1) * (ny + 1) * (nz + 1) >= (1ull << 31)) {
        fprintf(stderr, "photon: volume of %d x %d x %d texels exceeds the 2^31-texel limit of the samplers\n", nx, ny, nz);
        return 1;
    }
    photon_volume *v = new photon_volume();
    const size_t n = (size_t)nx * ny * nz;
    float *d_rho = nullptr, *d_min = nullptr;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    auto fail = [&](int code) { if (d_rho) (void)hipFree(d_rho); if (d_min) (void)hipFree(d_min); photon_volume_free(v); return code; };
#define PH_VCHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { fprintf(stderr, "photon: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return fail((int)_e); } } while (0)
    PH_VCHECK(device_malloc((void **)&v->d_texels, n * sizeof(f4)));
    PH_VCHECK(device_malloc((void **)&d_rho, n * sizeof(float)));
    PH_VCHECK(device_malloc((void **)&d_min, 2 * (size_t)blocks * sizeof(float)));      // block minima of n-1 | block maxima of |grad n|
    if (src.host_rho) {
        PH_VCHECK(hipMemcpy(d_rho, src.host_rho, n * sizeof(float), hipMemcpyHostToDevice));
    } else {
        hipLaunchKernelGGL(separable_density_kernel, dim3(blocks), dim3(256), 0, 0, src.d_gx, src.d_gy, src.d_gz, nx, ny, nz,
                           src.rho0, src.amp, d_rho);
        PH_VCHECK(hipGetLastError());
    }
    const float gx = (float)spacing[0], gy = (float)spacing[1], gz = (float)spacing[2];
    hipLaunchKernelGGL(build_volume_kernel, dim3(blocks), dim3(256), 0, 0, d_rho, nx, ny, nz, gx, gy, gz, v->d_texels,
                       d_min);
    PH_VCHECK(hipGetLastError());
    std::vector<float> mins(2 * (size_t)blocks);
    PH_VCHECK(hipMemcpy(mins.data(), d_min, mins.size() * sizeof(float), hipMemcpyDeviceToHost));
    float data_min = FLT_MAX, grad_max = 0.f;
    for (unsigned k = 0; k < blocks; k++) {
        if (mins[k] < data_min) data_min = mins[k];
        if (mins[blocks + k] > grad_max) grad_max = mins[blocks + k];
    }
    v->grad_max = grad_max;
    if (interpolation == 2) {
        PH_VCHECK(device_malloc((void **)&v->d_coeffs, n * sizeof(f4)));
        PH_VCHECK(hipMemcpyAsync(v->d_coeffs, v->d_texels, n * sizeof(f4), hipMemcpyDeviceToDevice, nullptr));     // null stream, like the prefilter passes behind it
        const size_t sx = 1, sy = (size_t)nx, sz = (size_t)nx * ny;
        auto nblk = [](size_t lines) { return dim3((unsigned)((lines + 255) / 256)); };
        // x lines: (y inner, z outer); y lines: (x inner, z outer); z lines: (x inner, y outer)
        hipLaunchKernelGGL(prefilter_lines_kernel, nblk((size_t)ny * nz), dim3(256), 0, 0, v->d_coeffs, nx, sx, ny, sy, nz, sz);
        hipLaunchKernelGGL(prefilter_lines_kernel, nblk((size_t)nx * nz), dim3(256), 0, 0, v->d_coeffs, ny, sy, nx, sx, nz, sz);
        hipLaunchKernelGGL(prefilter_lines_kernel, nblk((size_t)nx * ny), dim3(256), 0, 0, v->d_coeffs, nz, sz, nx, sx, ny, sy);
        PH_VCHECK(hipGetLastError());
    }
    PH_VCHECK(hipDeviceSynchronize());
    (void)hipFree(d_rho); d_rho = nullptr;
    (void)hipFree(d_min); d_min = nullptr;
#undef PH_VCHECK
    float step = (float)fmin(spacing[0], spacing[1]);                   // .h:2086-2098
    step = step < spacing[2] ? step : (float)spacing[2];
    VolumeDev &d = v->dev;
    d.min_bound = f3{(float)xmin, (float)ymin, (float)zmin};
    d.max_bound = f3{(float)xmax, (float)ymax, (float)zmax};
    d.nx = nx; d.ny = ny; d.nz = nz;
    d.step_size = step;
    d.data_min = data_min;
    d.interpolation = interpolation;
    d.weight_inv = 1.0f / 256.f;
    d.weight_scale = 256.f;             // trilinear weights as the reference's texture unit holds them (photon_volume_set_weight_bits)
    d.texels = v->d_texels;
    d.coeffs = v->d_coeffs;
    photon_volume_info_t &info = v->info;
    info.min_bound[0] = d.min_bound.x; info.min_bound[1] = d.min_bound.y; info.min_bound[2] = d.min_bound.z;
    info.max_bound[0] = d.max_bound.x; info.max_bound[1] = d.max_bound.y; info.max_bound[2] = d.max_bound.z;
    info.nx = nx; info.ny = ny; info.nz = nz;
    info.grid_spacing[0] = gx; info.grid_spacing[1] = gy; info.grid_spacing[2] = gz;
    info.step_size = step; info.data_min = data_min; info.interpolation = interpolation;
    *out = v;
    return 0;
}

int photon_volume_load_nrrd(const char *path, int interpolation, photon_volume_t **out) {
  return guarded("photon_volume_load_nrrd", [&]() -> int {
    if (!path || !out) { fprintf(stderr, "photon: photon_volume_load_nrrd: null argument\n"); return 1; }
    std::vector<float> rho;
    int dims[3];
    double spacing[3], origin[3];
    std::string why;
    if (!parse_nrrd(path, rho, dims, spacing, origin, why)) {
        fprintf(stderr, "photon: failed to read NRRD \"%s\": %s\n", path ? path : "(null)", why.c_str());
        return 2;
    }
    if (verbose())
        printf("photon: NRRD %s  sizes %d %d %d  spacings %g %g %g  origin (%g,%g,%g)\n", path, dims[0], dims[1], dims[2],
               spacing[0], spacing[1], spacing[2], origin[0], origin[1], origin[2]);
    return photon_volume_from_density(rho.data(), dims[0], dims[1], dims[2], spacing, origin, interpolation, out);
  });
}

int photon_volume_set_weight_bits(photon_volume_t *vol, int bits) {
    if (!vol || bits < 0 || bits > 23) return 1;
    vol->dev.weight_scale = bits ? (float)(1 << bits) : 0.f;
    vol->dev.weight_inv = bits ? 1.0f / (float)(1 << bits) : 0.f;
    return 0;
}

int photon_volume_info(const photon_volume_t *vol, photon_volume_info_t *info) {
    if (!vol || !info) return 1;
    *info = vol->info;
    return 0;
}

int photon_volume_download(const photon_volume_t *vol, int coefficients, float *out) {
    if (!vol || !out) return 1;
    const f4 *src = (coefficients && vol->d_coeffs) ? vol->d_coeffs : vol->d_texels;
    const size_t n = (size_t)vol->dev.nx * vol->dev.ny * vol->dev.nz;
    PH_CHECK(hipMemcpy(out, src, n * sizeof(f4), hipMemcpyDeviceToHost));
    return 0;
}

int photon_volume_sample(const photon_volume_t *vol, int n, const float *coords, float *out) {
    if (!vol || n < 0) return 1;
    if (n == 0) return 0;
    DeviceBuffer<float> d_c, d_o;                       // freed on every return path
    PH_CHECK(d_c.alloc((size_t)n * 3));
    PH_CHECK(d_o.alloc((size_t)n * 4));
    PH_CHECK(hipMemcpy(d_c.p, coords, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sample_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, vol->dev, n, d_c.p, d_o.p);
    PH_CHECK(hipGetLastError());
    PH_CHECK(hipMemcpy(out, d_o.p, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"

// =============================================================================================
// the volume cache of start_ray_tracing
// =============================================================================================
namespace {

struct VolumeCache {                // the library stays loaded between photon's calls: keep the
    std::string path;               // uploaded volume, keyed by (file, mtime, size, sampler)
    long long mtime_ns = 0;
    long long size = 0;
    int interpolation = 0;
    int device = -1;
    photon_volume *vol = nullptr;
};
std::mutex g_cache_mutex;                       // guards the map; each entry has its own lock for the (slow) load
struct DeviceCache { std::mutex lock; VolumeCache entry; };
std::map<int, DeviceCache> g_cache;             // one cached volume per device (PHOTON_DEVICES renders on several)

}  // namespace

namespace photon {

int cached_volume(const char *path, int interpolation, photon_volume **out, SharedDensity *shared) {
    struct stat st;
    if (stat(path, &st) != 0) {
        fprintf(stderr, "photon: failed to open \"%s\"\n", path);
        return 2;
    }
    int device = 0;
    (void)hipGetDevice(&device);
    const long long mt = (long long)st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec;
    DeviceCache *dc;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        dc = &g_cache[device];                  // std::map: references stay valid
    }
    std::lock_guard<std::mutex> lock(dc->lock);
    VolumeCache &c = dc->entry;
    if (c.vol && c.path == path && c.mtime_ns == mt && c.size == (long long)st.st_size && c.interpolation == interpolation) {
        *out = c.vol;
        return 0;
    }
    if (c.vol) { photon_volume_free(c.vol); c.vol = nullptr; }
    photon_volume *v = nullptr;
    int rc;
    if (shared) {
        std::call_once(shared->once, [&]() { shared->ok = parse_nrrd(path, shared->rho, shared->dims, shared->spacing, shared->origin, shared->why); });
        if (!shared->ok) {
            fprintf(stderr, "photon: failed to read NRRD \"%s\": %s\n", path, shared->why.c_str());
            return 2;
        }
        rc = photon_volume_from_density(shared->rho.data(), shared->dims[0], shared->dims[1], shared->dims[2], shared->spacing,
                                        shared->origin, interpolation, &v);
    } else {
        rc = photon_volume_load_nrrd(path, interpolation, &v);
    }
    if (rc) return rc;
    c.path = path; c.mtime_ns = mt; c.size = (long long)st.st_size;
    c.interpolation = interpolation; c.device = device; c.vol = v;
    *out = v;
    return 0;
}

}  // namespace photon
