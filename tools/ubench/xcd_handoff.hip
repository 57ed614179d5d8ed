// micro-test: the wave-to-wave hand-off the segmented march relies on (march_kernel.hpp, march_group), stressed.
//
// N groups of 64 words x A arrays; a persistent grid takes items k from one atomic queue, item k = pass k / N of group
// k % N: wait for the group's flag to reach the pass (sc1 poll), [acquire], load the group's words, CHECK that they are
// what pass - 1 wrote, store what this pass writes, drain (s_waitcnt vmcnt(0)), raise the flag (sc1 store).  Items go to
// whichever wave asks next, so a group's consecutive passes run on different CUs of different XCDs, every L2 has held
// earlier versions of the lines it is asked for again (L2-warm), and a per-item busy loop of pseudo-random length keeps
// the arrivals uneven.  `misalign` puts the arrays' bases off the 128-byte line grid, so that neighbouring groups --
// handled by other XCDs at other times -- share cache lines.
//   variant 0: sc1 stores, sc1 loads, agent acquire after the poll      (what march_group does)
//   variant 1: sc1 stores, sc1 loads, no acquire
//   variant 2: sc1 stores, PLAIN loads, agent acquire
//   variant 3: PLAIN stores + agent release, plain loads, agent acquire  (the memory model's textbook form)
// Prints stale words per variant; 0 expected for the forms in use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <class T> __device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ unsigned value_of(unsigned group, unsigned lane, unsigned array, unsigned pass) {
    return (group * 64u + lane) * 2654435761u + array * 40503u + pass * 0x9e3779b9u;
}

constexpr int A = 8;

template <int VARIANT>
__global__ __launch_bounds__(256) void handoff(unsigned *data, size_t stride, unsigned *flags, unsigned *queue, unsigned n_groups,
                                               unsigned passes, unsigned long long *stale, unsigned *gave_up) {
    const unsigned lane = threadIdx.x & 63u;
    unsigned long long bad = 0;
    while (true) {
        unsigned k = 0;
        if (lane == 0) k = atomicAdd(queue, 1u);
        k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
        if (k >= n_groups * passes) break;
        const unsigned pass = k / n_groups, g = k - pass * n_groups;
        if (pass > 0) {
            int polls = 0;
            while (true) {
                unsigned f = 0;
                if (lane == 0) f = ld_agent(&flags[g]);
                f = (unsigned)__builtin_amdgcn_readfirstlane((int)f);
                if (f >= pass) break;
                if (++polls > (1 << 22)) { if (lane == 0) atomicAdd(gave_up, 1u); return; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (VARIANT != 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        asm volatile("" ::: "memory");
        unsigned v[A];
#pragma unroll
        for (int a = 0; a < A; a++) {
            const unsigned *p = data + a * stride + g * 64u + lane;
            v[a] = (VARIANT == 2 || VARIANT == 3) ? *p : ld_agent(p);
        }
#pragma unroll
        for (int a = 0; a < A; a++) bad += v[a] != value_of(g, lane, a, pass);      // pass 0 reads what the host wrote: value_of(.., 0)
        // uneven arrivals: a busy loop whose length depends on the item
        unsigned spin = (k * 2246822519u >> 22) & 1023u, acc = v[0];
        for (unsigned i = 0; i < spin; i++) acc = acc * 1664525u + 1013904223u;
        if (acc == 0x12345u) bad += 1u << 30;               // keeps the loop alive
#pragma unroll
        for (int a = 0; a < A; a++) {
            unsigned *p = data + a * stride + g * 64u + lane;
            if (VARIANT == 3) *p = value_of(g, lane, a, pass + 1);
            else st_agent(p, value_of(g, lane, a, pass + 1));
        }
        if (VARIANT == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) st_agent(&flags[g], pass + 1);
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o, 64);
    if (lane == 0 && bad) atomicAdd(stale, bad);
}

int main(int argc, char **argv) {
    const unsigned n_groups = argc > 1 ? atoi(argv[1]) : 6000, passes = argc > 2 ? atoi(argv[2]) : 24;
    for (int misalign = 0; misalign < 2; misalign++) {
        const size_t stride = (size_t)n_groups * 64 + (misalign ? 16 : 0);      // +16 words: array bases 64 bytes off the line grid
        unsigned *data, *flags, *queue, *gave_up;
        unsigned long long *stale;
        hipMalloc(&data, A * stride * 4 + 256); hipMalloc(&flags, n_groups * 4); hipMalloc(&queue, 4); hipMalloc(&gave_up, 4); hipMalloc(&stale, 8);
        unsigned *base = data + (misalign ? 16 : 0);
        std::vector<unsigned> h(A * stride);
        for (int variant = 0; variant < 4; variant++) {
            for (unsigned a = 0; a < (unsigned)A; a++)
                for (unsigned g = 0; g < n_groups; g++)
                    for (unsigned l = 0; l < 64; l++) h[a * stride + g * 64 + l] = (g * 64u + l) * 2654435761u + a * 40503u;
            hipMemcpy(base, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            hipMemset(flags, 0, n_groups * 4); hipMemset(queue, 0, 4); hipMemset(gave_up, 0, 4); hipMemset(stale, 0, 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            const dim3 grid(256 * 5), block(256);
            switch (variant) {
                case 0: hipLaunchKernelGGL(handoff<0>, grid, block, 0, 0, base, stride, flags, queue, n_groups, passes, stale, gave_up); break;
                case 1: hipLaunchKernelGGL(handoff<1>, grid, block, 0, 0, base, stride, flags, queue, n_groups, passes, stale, gave_up); break;
                case 2: hipLaunchKernelGGL(handoff<2>, grid, block, 0, 0, base, stride, flags, queue, n_groups, passes, stale, gave_up); break;
                default: hipLaunchKernelGGL(handoff<3>, grid, block, 0, 0, base, stride, flags, queue, n_groups, passes, stale, gave_up); break;
            }
            hipEventRecord(e1);
            if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed\n"); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long s = 0; unsigned gu = 0;
            hipMemcpy(&s, stale, 8, hipMemcpyDeviceToHost); hipMemcpy(&gu, gave_up, 4, hipMemcpyDeviceToHost);
            static const char *names[] = {"sc1 stores, sc1 loads, acquire", "sc1 stores, sc1 loads, no acquire", "sc1 stores, plain loads, acquire",
                                          "plain stores + release, plain loads, acquire"};
            printf("%s lines, %-46s: %llu stale words of %llu checked, %u waves gave up, %.2f ms\n", misalign ? "shared " : "aligned", names[variant], s,
                   (unsigned long long)n_groups * passes * 64 * A, gu, ms);
            fflush(stdout);
        }
        hipFree(data); hipFree(flags); hipFree(queue); hipFree(gave_up); hipFree(stale);
    }
    return 0;
}
