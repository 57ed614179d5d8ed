// photon_pool.hip - device memory of the library (photon_pool.hpp): a cache of freed device blocks, the allocation
// helper of every other hipMalloc site, the process-wide peer-access record.  Host code only.
//
// photon's unchanged Python builds everything anew for every start_ray_tracing call: per call ~25 hipMalloc / hipFree
// pairs, among them the ray-state workspace (320 MB for the 1e7-ray job) -- measured, the frees alone take 0.8-1.3 ms of a
// call (PHOTON_VERBOSE), 10 % of one GPU's eighth of the headline job, most of a small PIV frame.  Scene-lifetime blocks are
// therefore handed back to this cache instead of the runtime and the next call of the same shape takes them from it
// (exact size match, per device); the cache holds at most PHOTON_POOL_MAX_MB (default 4096; 0 = off), evicting its largest
// blocks first; photon_trim_caches() empties it, and so does any allocation of this library that finds the device out of
// memory (device_malloc).  Recycled memory is not zeroed -- neither is hipMalloc'd memory: every buffer that needs a
// defined start is cleared where it is allocated or used.
#include "photon_pool.hpp"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace photon {
namespace {

struct PoolKey {
    int device; size_t bytes;
    bool operator<(const PoolKey &o) const { return device != o.device ? device < o.device : bytes < o.bytes; }
};
struct DevicePool {
    std::mutex lock;
    std::multimap<PoolKey, void *> idle;        // ordered by (device, size): a device's largest block is the last of its run
    std::map<void *, PoolKey> live;
    size_t idle_bytes = 0;
};
DevicePool &device_pool() { static DevicePool *p = new DevicePool; return *p; }      // never destroyed: the runtime may be gone by then
size_t pool_cap_bytes() {
    static const size_t cap = [] { const char *e = getenv("PHOTON_POOL_MAX_MB"); return (size_t)(e ? strtoull(e, nullptr, 10) : 4096ull) << 20; }();
    return cap;
}

// the largest idle block of any device (caller holds the lock): the last entry of each device's run
std::multimap<PoolKey, void *>::iterator largest_idle(DevicePool &p) {
    auto best = p.idle.end();
    for (auto it = p.idle.begin(); it != p.idle.end();) {
        auto next = p.idle.upper_bound(PoolKey{it->first.device, (size_t)-1});      // first entry of the next device
        auto last = std::prev(next);
        if (best == p.idle.end() || last->first.bytes > best->first.bytes) best = last;
        it = next;
    }
    return best;
}

}  // namespace

void pool_trim(size_t keep_bytes) {                              // caller holds no lock
    DevicePool &p = device_pool();
    std::vector<void *> victims;
    {
        std::lock_guard<std::mutex> g(p.lock);
        while (p.idle_bytes > keep_bytes && !p.idle.empty()) {
            auto big = largest_idle(p);
            p.idle_bytes -= big->first.bytes;
            victims.push_back(big->second);
            p.idle.erase(big);
        }
    }
    for (void *v : victims) (void)hipFree(v);                   // outside the lock: hipFree synchronises the device
}

hipError_t device_malloc(void **out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        pool_trim(0);
        e = hipMalloc(out, bytes ? bytes : 1);
    }
    return e;
}

hipError_t device_zero(void *p, size_t bytes) {
    if (!bytes) return hipSuccess;
    const hipError_t e = hipMemsetAsync(p, 0, bytes, nullptr);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

hipError_t pool_malloc(void **out, size_t bytes) {
    if (bytes == 0) bytes = 1;
    int device = 0;
    (void)hipGetDevice(&device);
    DevicePool &p = device_pool();
    {
        std::lock_guard<std::mutex> g(p.lock);
        auto it = p.idle.find(PoolKey{device, bytes});
        if (it != p.idle.end()) {
            *out = it->second;
            p.idle.erase(it);
            p.idle_bytes -= bytes;
            p.live[*out] = PoolKey{device, bytes};
            return hipSuccess;
        }
    }
    const hipError_t e = device_malloc(out, bytes);
    if (e == hipSuccess) { std::lock_guard<std::mutex> g(p.lock); p.live[*out] = PoolKey{device, bytes}; }
    return e;
}

void pool_free(void *ptr) {
    if (!ptr) return;
    DevicePool &p = device_pool();
    bool keep = false, over = false;
    {
        std::lock_guard<std::mutex> g(p.lock);
        auto it = p.live.find(ptr);
        if (it != p.live.end()) {
            const PoolKey k = it->second;
            p.live.erase(it);
            if (k.bytes <= pool_cap_bytes()) { p.idle.emplace(k, ptr); p.idle_bytes += k.bytes; keep = true; }
        }
        over = p.idle_bytes > pool_cap_bytes();                 // decided under the lock
    }
    if (!keep) { (void)hipFree(ptr); return; }
    if (over) pool_trim(pool_cap_bytes());
}

// ---------------------------------------------------------------------------------------------
// peer access, once per ordered pair of devices and process
// ---------------------------------------------------------------------------------------------
bool peer_access(int dev, int peer) {
    static std::mutex lock;
    static std::map<std::pair<int, int>, bool> *known = new std::map<std::pair<int, int>, bool>;
    (void)hipSetDevice(dev);
    if (dev == peer) return true;
    std::lock_guard<std::mutex> g(lock);
    auto it = known->find({dev, peer});
    if (it != known->end()) return it->second;
    int can = 0;
    bool direct = false;
    const hipError_t ce = hipDeviceCanAccessPeer(&can, dev, peer);
    if (ce == hipSuccess && can) {
        const hipError_t pe = hipDeviceEnablePeerAccess(peer, 0);
        direct = pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled;
        if (pe != hipSuccess) (void)hipGetLastError();
        // Kernels on `dev` will dereference pointers into `peer`'s memory.  The runtime maps what is allocated from now on;
        // blocks idling in the cache were allocated before.  Hand them back so that whatever a scene gets from here on is a
        // fresh allocation (once per pair and process).
        if (direct) pool_trim(0);
        if (!direct)
            fprintf(stderr, "photon: hipDeviceEnablePeerAccess(device %d from device %d) failed: %s; its accumulators are copied through host staging\n",
                    peer, dev, hipGetErrorString(pe));
    } else {
        if (ce != hipSuccess) (void)hipGetLastError();
        fprintf(stderr, "photon: device %d cannot access device %d as a peer (%s); its accumulators are copied through host staging\n", dev, peer,
                ce == hipSuccess ? "hipDeviceCanAccessPeer: no" : hipGetErrorString(ce));
    }
    (*known)[{dev, peer}] = direct;
    return direct;
}

}  // namespace photon

extern "C" void photon_trim_caches(void) { photon::pool_trim(0); }
