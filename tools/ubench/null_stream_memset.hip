// null_stream_memset.hip - are hipMemset / hipMemcpy (the "synchronous" ones, null stream) complete when they return, and ordered
// before a kernel launched afterwards on a NON-BLOCKING stream?  The chip is kept full by a spinning kernel on another
// non-blocking stream (every wave slot taken), so a fill KERNEL cannot run until that one ends.
//     hipcc --offload-arch=gfx950 -O2 -o null_stream_memset null_stream_memset.hip && ./null_stream_memset
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void spin(unsigned long long ticks) {                 // 100 MHz ticks
    unsigned long long t0, t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    do { asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); __builtin_amdgcn_s_sleep(32); } while (t - t0 < ticks);
}
__global__ void probe(const unsigned *p, unsigned *out) { out[0] = p[0]; out[1] = p[1023]; }

static double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

int main() {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned *d = nullptr, *out = nullptr, *dz = nullptr;
    CK(hipMalloc(&d, 4096)); CK(hipMalloc(&out, 8)); CK(hipMalloc(&dz, 4096));
    std::vector<unsigned> ones(1024, 0xffffffffu), zeros(1024, 0u);
    CK(hipMemcpy(dz, zeros.data(), 4096, hipMemcpyHostToDevice));
    for (int form = 0; form < 4; form++) {
        CK(hipMemcpy(d, ones.data(), 4096, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(spin, dim3(512), dim3(1024), 0, a, 20000000ull);             // 200 ms, 2 x 1024 threads on each of 256 CUs
        CK(hipGetLastError());
        const auto t0 = std::chrono::steady_clock::now();
        if (form == 0) CK(hipMemset(d, 0, 4096));
        if (form == 1) CK(hipMemcpy(d, zeros.data(), 4096, hipMemcpyHostToDevice));
        if (form == 2) { CK(hipMemset(d, 0, 4096)); CK(hipStreamSynchronize(nullptr)); }
        if (form == 3) CK(hipMemcpy(d, dz, 4096, hipMemcpyDeviceToDevice));
        const double t_call = ms_since(t0);
        hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, b, d, out);                      // no ordering with the null stream
        CK(hipGetLastError());
        CK(hipStreamSynchronize(b));
        const double t_probe = ms_since(t0);
        unsigned h[2];
        CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
        CK(hipDeviceSynchronize());
        printf("%-34s returned after %8.3f ms; a kernel on another non-blocking stream then read %08x %08x (probe done at %.3f ms; whole at %.3f ms)\n",
               form == 0 ? "hipMemset(4 KB)" : form == 1 ? "hipMemcpy(4 KB, host zeros)" : form == 2 ? "hipMemset + hipStreamSynchronize(0)" : "hipMemcpy(4 KB, device to device)", t_call, h[0], h[1], t_probe,
               ms_since(t0));
    }
    return 0;
}
