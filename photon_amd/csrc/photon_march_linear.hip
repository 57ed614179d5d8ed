// photon_march_linear.hip - the march kernels of the TRILINEAR sampler (the one the reference executes,
// parallel_ray_tracing.cu:3330): Euler and RK4, with and without intermediate dumps, gradient noise, segments.
// One translation unit per sampler: a kernel edit recompiles one unit.
#include "march_kernel.hpp"
#include "photon_internal.hpp"

namespace photon {

#define PH_MARCH(A, S, N) do { if (!S && !N && segmented) hipLaunchKernelGGL((march_kernel<A, 1, false, false, true>), grid, block, 0, stream, a); \
                              else hipLaunchKernelGGL((march_kernel<A, 1, S, N, false>), grid, block, 0, stream, a); } while (0)
int march_launch_linear(int algorithm, bool save, bool noise, bool segmented, dim3 grid, dim3 block, hipStream_t stream, const MarchArgs &a) {
    if (algorithm == 1) {                                   // the gradient-noise hook exists in this branch only (.h:853-863)
        if (save) { if (noise) PH_MARCH(1, true, true); else PH_MARCH(1, true, false); }
        else { if (noise) PH_MARCH(1, false, true); else PH_MARCH(1, false, false); }
    } else {
        if (save) PH_MARCH(2, true, false); else PH_MARCH(2, false, false);
    }
    PH_CHECK(hipGetLastError());
    return 0;
}
#undef PH_MARCH

int march_rays_launch_linear(int algorithm, const VolumeDev &vol, const f4 *tex, int n, float *pos, float *dir, int *steps) {
    const dim3 grid((n + 255) / 256), block(256);
    if (algorithm == 1) hipLaunchKernelGGL((march_rays_kernel<1, 1>), grid, block, 0, 0, vol, tex, n, pos, dir, steps);
    else hipLaunchKernelGGL((march_rays_kernel<2, 1>), grid, block, 0, 0, vol, tex, n, pos, dir, steps);
    PH_CHECK(hipGetLastError());
    return 0;
}

#if PHOTON_PATH_STATS
int march_path_stats_linear(unsigned long long out[8]) {   // debug builds only: read (and clear) this unit's sampler-path counters
    PH_CHECK(hipDeviceSynchronize());
    PH_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(photon::g_path_stats), 8 * sizeof(unsigned long long)));
    unsigned long long zero[8] = {};
    PH_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(photon::g_path_stats), zero, sizeof zero));
    return 0;
}
#endif

}  // namespace photon
