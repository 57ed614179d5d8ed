// photon_sort.hpp - what photon_trace.hip and photon_sort.hip share: the sort's scratch (owned by the scene, grown on demand)
// and the two entry points.  ONE definition of the struct for both translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

// Scratch of the sort, owned by the caller (the scene) and grown on demand: with it a sort allocates nothing and never
// waits for the host -- photon_trace stays asynchronous on its stream when a lens-major launch has to order a new range.
struct photon_sort_scratch {
    unsigned *box = nullptr, *keys = nullptr;       // keys: 2 x capacity (in, out)
    int *idx = nullptr;
    void *tmp = nullptr;
    size_t capacity = 0, tmp_bytes = 0;
};

void photon_sort_scratch_free(photon_sort_scratch *scratch);

// perm_out[k] (device, n entries) = index, in the CALLER's source numbering, of the k-th source of
// [first, first + n) in Morton order.  x, y: device pointers to the whole source arrays.  Asynchronous on
// `stream`; allocates only when the scratch has to grow.  Returns 0 or a HIP error code.
int photon_morton_order(const float *d_x, const float *d_y, int first, long long n, int *d_perm_out, hipStream_t stream,
                        photon_sort_scratch *scratch);
