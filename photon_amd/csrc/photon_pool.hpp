// photon_pool.hpp - device memory of the library: the cache of freed blocks, the allocation helper every other
// hipMalloc site goes through, and the process-wide record of peer access between devices (photon_pool.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace photon {

// A block of exactly `bytes` on the current device: from the cache of freed blocks when one of that size waits there,
// from hipMalloc otherwise (the cache is emptied and the call repeated once when the device is out of memory).
// Recycled memory is not zeroed -- neither is hipMalloc'd memory.  The CALLER makes sure the device is done with a
// block before it hands it back (photon_scene_free and the regrow paths synchronise first).
hipError_t pool_malloc(void **out, size_t bytes);
void pool_free(void *ptr);
// hand idle blocks back to the runtime, largest first, until at most keep_bytes remain cached
void pool_trim(size_t keep_bytes);
// hipMalloc for blocks that do not go through the cache (volumes, generated sources, sort scratch, one-shot buffers):
// on hipErrorOutOfMemory the cache is emptied and the call repeated once -- what the cache holds must never be the
// reason another allocation of this library fails.
hipError_t device_malloc(void **out, size_t bytes);

// Zero `bytes` at `p` on the current device, COMPLETE when the call returns.  hipMemset is not: it queues a fill kernel on
// the null stream and returns (measured: 9 us, with the fill still 200 ms away behind a full chip), and work launched
// afterwards on a non-blocking stream is not ordered behind the null stream (hipMemcpy device-to-device behaves the same;
// host-to-device and device-to-host copies are complete on return: tools/ubench/null_stream_memset.hip).  Every zeroing in
// this library that is not a hipMemsetAsync on the stream of its consumer goes through here (tests/test_sources_lint.py holds the line).
hipError_t device_zero(void *p, size_t bytes);

// a device allocation that is released on every return path
template <typename T>
struct DeviceBuffer {
    T *p = nullptr;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return device_malloc((void **)&p, (n ? n : 1) * sizeof(T)); }
};
// the same for a block of the cache
template <typename T>
struct PoolBuffer {
    T *p = nullptr;
    PoolBuffer() = default;
    PoolBuffer(const PoolBuffer &) = delete;
    PoolBuffer &operator=(const PoolBuffer &) = delete;
    ~PoolBuffer() { pool_free(p); }
    hipError_t alloc(size_t n) { return pool_malloc((void **)&p, (n ? n : 1) * sizeof(T)); }
};

// Peer access from device `dev` to the memory of device `peer`, decided ONCE per ordered pair and process
// (hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess, then remembered): true = kernels running on `dev` may dereference
// pointers into `peer`'s memory (xGMI).  A pair that cannot is reported on stderr once, never silently.  Changes the
// calling thread's current device to `dev`.  When a pair is enabled the block cache is emptied: call this BEFORE allocating
// what the peer kernels will read (blocks that idled in the cache predate the mapping).
bool peer_access(int dev, int peer);

}  // namespace photon
