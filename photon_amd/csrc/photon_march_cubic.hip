// photon_march_cubic.hip - the march kernels of the TRICUBIC B-spline sampler (the headline's): Euler and RK4, whole
// and segmented marches.  One translation unit per sampler: a kernel edit recompiles one unit.
#include "march_kernel.hpp"
#include "photon_internal.hpp"

namespace photon {

int march_launch_cubic(int algorithm, bool segmented, dim3 grid, dim3 block, hipStream_t stream, const MarchArgs &a) {
    if (algorithm == 1) {
        if (segmented) hipLaunchKernelGGL((march_kernel<1, 2, false, false, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((march_kernel<1, 2, false, false, false>), grid, block, 0, stream, a);
    } else {
        if (segmented) hipLaunchKernelGGL((march_kernel<2, 2, false, false, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((march_kernel<2, 2, false, false, false>), grid, block, 0, stream, a);
    }
    PH_CHECK(hipGetLastError());
    return 0;
}

int march_rays_launch_cubic(int algorithm, const VolumeDev &vol, const f4 *tex, int n, float *pos, float *dir, int *steps) {
    const dim3 grid((n + 255) / 256), block(256);
    if (algorithm == 1) hipLaunchKernelGGL((march_rays_kernel<1, 2>), grid, block, 0, 0, vol, tex, n, pos, dir, steps);
    else hipLaunchKernelGGL((march_rays_kernel<2, 2>), grid, block, 0, 0, vol, tex, n, pos, dir, steps);
    PH_CHECK(hipGetLastError());
    return 0;
}

#if PHOTON_PATH_STATS
int march_path_stats_cubic(unsigned long long out[8]) {    // debug builds only: read (and clear) this unit's sampler-path counters
    PH_CHECK(hipDeviceSynchronize());
    PH_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(photon::g_path_stats), 8 * sizeof(unsigned long long)));
    unsigned long long zero[8] = {};
    PH_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(photon::g_path_stats), zero, sizeof zero));
    return 0;
}
#endif

}  // namespace photon
