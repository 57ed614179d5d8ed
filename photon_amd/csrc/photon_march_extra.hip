// photon_march_extra.hip - ray_tracing_algorithm 3 (rk45), 4 (adams_bashforth) and "anything else" (the reference's
// `default: break`, .h:1537: the ray is only moved to its entry point): per-lane code, trilinear gathers of the raw volume
// (device_volume_extra.hpp).  Correct, untuned, rarely used -- and 211-214 VGPRs: kept out of the hot kernels' units.
#include "device_volume_extra.hpp"
#include "photon_internal.hpp"

using namespace photon;

template <int ALGO>
__global__ __launch_bounds__(256) void march_rays_extra_kernel(VolumeDev v, int n, float *__restrict__ pos,
                                                               float *__restrict__ dir, int *__restrict__ steps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 p = mk3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
    f3 d = mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    MarchCount mc{0, 0};
    trace_volume_extra<ALGO>(p, d, v, mc);
    pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z;
    dir[3 * i] = d.x; dir[3 * i + 1] = d.y; dir[3 * i + 2] = d.z;
    if (steps) steps[i] = mc.iterations;
}

// Stage 1b for ray_tracing_algorithm 3, 4 and the reference's no-op default (see march_rays_extra_kernel).
template <int ALGO>
__global__ __launch_bounds__(256) void march_extra_kernel(VolumeDev vol, unsigned n_rays, RayStateDev st,
                                                          unsigned long long *__restrict__ counters) {
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    MarchCount mc{0, 0};
    unsigned marched = 0;
    if (r < n_rays) {
        f3 p = mk3(st.px[r], st.py[r], st.pz[r]);
        f3 d = mk3(st.dx[r], st.dy[r], st.dz[r]);
        if (!isnan3(p)) {
            marched = 1;
            trace_volume_extra<ALGO>(p, d, vol, mc);
            st.px[r] = p.x; st.py[r] = p.y; st.pz[r] = p.z;
            st.dx[r] = d.x; st.dy[r] = d.y; st.dz[r] = d.z;
        }
    }
    wave_add(&counter_slot(counters)[CNT_ITER], (unsigned long long)mc.iterations);
    wave_add(&counter_slot(counters)[CNT_SAMPLES], (unsigned long long)mc.samples);
    wave_add(&counter_slot(counters)[CNT_MARCHED], (unsigned long long)marched);
}

namespace photon {

int march_launch_extra(int algorithm, dim3 grid, dim3 block, hipStream_t stream, const VolumeDev &vol, unsigned n_rays, const RayStateDev &st,
                       unsigned long long *counters) {
    if (algorithm == 3) hipLaunchKernelGGL((march_extra_kernel<3>), grid, block, 0, stream, vol, n_rays, st, counters);
    else if (algorithm == 4) hipLaunchKernelGGL((march_extra_kernel<4>), grid, block, 0, stream, vol, n_rays, st, counters);
    else hipLaunchKernelGGL((march_extra_kernel<0>), grid, block, 0, stream, vol, n_rays, st, counters);
    PH_CHECK(hipGetLastError());
    return 0;
}

int march_rays_launch_extra(int algorithm, const VolumeDev &vol, int n, float *pos, float *dir, int *steps) {
    const dim3 grid((n + 255) / 256), block(256);
    if (algorithm == 3) hipLaunchKernelGGL((march_rays_extra_kernel<3>), grid, block, 0, 0, vol, n, pos, dir, steps);
    else if (algorithm == 4) hipLaunchKernelGGL((march_rays_extra_kernel<4>), grid, block, 0, 0, vol, n, pos, dir, steps);
    else hipLaunchKernelGGL((march_rays_extra_kernel<0>), grid, block, 0, 0, vol, n, pos, dir, steps);
    PH_CHECK(hipGetLastError());
    return 0;
}

}  // namespace photon
