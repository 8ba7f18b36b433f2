// agx_mem.h — device and pinned-host memory of the engine (host side, HIP runtime API only).
//
// What the MI355X box measures (tests/tools/membench.cpp, profiles/r02_membench.txt): hipMalloc 0.03-0.7 ms per call whatever the size,
// hipFree 0.3 ms but it waits for the device; hipHostMalloc 0.22 ms per MB (4.5 GB/s!) and hipHostFree 0.13 ms per MB; registering
// ordinary memory (hipHostRegister) 0.04 ms per MB and copies from it run at the same 57 GB/s as from hipHostMalloc'ed memory;
// pageable copies 20-35 GB/s.  So:
//   * a unit takes its HBM as ONE block (DevArena: a bump allocator over a block from the device's cache) instead of ~70 hipMallocs,
//     and gives the block back to a per-device cache when it is released — the next unit of the run takes it without a driver call;
//   * pinned host memory is anonymous mmap + hipHostRegister, cached per process the same way (HostBlocks): the packed inputs are staged
//     in it when they are handed over, so an upload is nothing but asynchronous copies at PCIe rate.
// Nothing here is on a kernel's path; it only decides what a new unit pays before its first kernel can start.
#pragma once
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>

#include <cstddef>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "agx_host.h"

namespace agx {

#define AGX_HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw ::agx::Error{::agx::E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)}; } while (0)

struct MemBlock { void *p = nullptr; size_t n = 0; };

// free blocks of one kind, best fit: the smallest cached block that holds the request and is not more than twice as large
class BlockCache {
public:
    bool take(size_t need, MemBlock &out) {
        std::lock_guard<std::mutex> g(m_);
        auto it = free_.lower_bound(need);
        if (it == free_.end() || it->first > 2 * need + (64u << 20)) return false;
        out = MemBlock{it->second, it->first}; held_ -= it->first; free_.erase(it);
        return true;
    }
    void give(const MemBlock &b) { if (!b.p) return; std::lock_guard<std::mutex> g(m_); free_.emplace(b.n, b.p); held_ += b.n; }
    std::vector<MemBlock> drain() { std::lock_guard<std::mutex> g(m_); std::vector<MemBlock> v; for (auto &kv : free_) v.push_back(MemBlock{kv.second, kv.first}); free_.clear(); held_ = 0; return v; }
    size_t held() const { std::lock_guard<std::mutex> g(m_); return held_; }
private:
    mutable std::mutex m_; std::multimap<size_t, void *> free_; size_t held_ = 0;
};

inline size_t round_block(size_t n) { const size_t g = n < (64u << 20) ? (2u << 20) : (16u << 20); return (n + g - 1) / g * g; }

// ---- HBM ------------------------------------------------------------------------------------------------------------------------
inline BlockCache &dev_cache(int device) { static BlockCache c[64]; return c[device & 63]; }

inline MemBlock dev_block(int device, size_t need) {
    MemBlock b; need = round_block(need ? need : 1);
    if (dev_cache(device).take(need, b)) return b;
    AGX_HIP_OK(hipSetDevice(device));
    hipError_t e = hipMalloc(&b.p, need);
    if (e != hipSuccess) {                       // out of HBM: give the cached blocks back to the driver and try once more
        (void)hipGetLastError();
        for (const MemBlock &c : dev_cache(device).drain()) (void)hipFree(c.p);
        AGX_HIP_OK(hipMalloc(&b.p, need));
    }
    b.n = need;
    return b;
}
inline void dev_trim(int device) { (void)hipSetDevice(device); for (const MemBlock &c : dev_cache(device).drain()) (void)hipFree(c.p); }

// A unit's device memory: blocks taken from the device's cache, handed out front to back (256-byte aligned), returned together.
class DevArena {
public:
    int device = 0;
    ~DevArena() { reset(); }
    void reserve(size_t bytes) { if (room() < bytes) add(bytes); }
    void *take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (room() < bytes) add(bytes > (256u << 20) ? bytes : (256u << 20));
        void *p = (char *)blocks_.back().p + at_; at_ += bytes; used_ += bytes;
        return p;
    }
    void reset() { for (const MemBlock &b : blocks_) dev_cache(device).give(b); blocks_.clear(); at_ = 0; used_ = 0; }
    size_t used() const { return used_; }
    size_t capacity() const { size_t s = 0; for (const MemBlock &b : blocks_) s += b.n; return s; }
private:
    std::vector<MemBlock> blocks_; size_t at_ = 0, used_ = 0;
    size_t room() const { return blocks_.empty() ? 0 : blocks_.back().n - at_; }
    void add(size_t bytes) { blocks_.push_back(dev_block(device, bytes)); at_ = 0; }
};

// typed view of arena memory.  alloc() only ever grows; memory that a regrow leaves behind stays in the arena until the unit is released.
template <class T> struct DBuf {
    T *p = nullptr; size_t n = 0;
    void alloc(DevArena &a, size_t count) { if (count <= n && p) return; p = count ? (T *)a.take(count * sizeof(T)) : nullptr; n = count; }
    void release() { p = nullptr; n = 0; }
};

// ---- pinned host memory -----------------------------------------------------------------------------------------------------------
inline BlockCache &host_cache() { static BlockCache c; return c; }

inline MemBlock host_block(size_t need) {
    MemBlock b; need = round_block(need ? need : 1);
    if (host_cache().take(need, b)) return b;
    void *p = mmap(nullptr, need, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) throw Error{E_ARG, "out of host memory"};
    (void)madvise(p, need, MADV_HUGEPAGE);
    hipError_t e = hipHostRegister(p, need, hipHostRegisterPortable | hipHostRegisterMapped);
    if (e != hipSuccess) { (void)hipGetLastError(); munmap(p, need); throw Error{E_DEVICE, std::string("hipHostRegister: ") + hipGetErrorString(e)}; }
    return MemBlock{p, need};
}
inline std::vector<MemBlock> &host_retired() { static std::vector<MemBlock> v; return v; }
inline std::mutex &host_retired_mutex() { static std::mutex m; return m; }
// retire: the cached blocks are never handed out again but stay mapped until the next host_trim — a measurement loop that wants every
// job to map, fault and register fresh buffers (like a new process) without also paying for unmapping the previous job's inside its clock
inline void host_retire() { std::vector<MemBlock> v = host_cache().drain(); std::lock_guard<std::mutex> g(host_retired_mutex()); host_retired().insert(host_retired().end(), v.begin(), v.end()); }
inline void host_trim() {
    std::vector<MemBlock> v = host_cache().drain();
    { std::lock_guard<std::mutex> g(host_retired_mutex()); v.insert(v.end(), host_retired().begin(), host_retired().end()); host_retired().clear(); }
    for (const MemBlock &c : v) { (void)hipHostUnregister(c.p); munmap(c.p, c.n); }
}

template <class T> struct PBuf {            // pinned host buffer: one cached block each
    T *p = nullptr; size_t n = 0; MemBlock blk;
    void alloc(size_t count) {
        if (count <= n && p) return;
        release();
        if (!count) return;
        blk = host_block(count * sizeof(T)); p = (T *)blk.p; n = blk.n / sizeof(T);
    }
    void release() { host_cache().give(blk); blk = MemBlock(); p = nullptr; n = 0; }
    void borrow(T *host, size_t count) { release(); p = host; n = count; }      // a piece of another buffer's block (pinned by its owner, who outlives the loan)
    size_t block_bytes() const { return blk.n; }
    void *dev() const { void *d = nullptr; if (p) AGX_HIP_OK(hipHostGetDevicePointer(&d, p, 0)); return d; }      // the address kernels use for this buffer
    ~PBuf() { release(); }
    PBuf() = default; PBuf(const PBuf &) = delete; PBuf &operator=(const PBuf &) = delete;
};

}  // namespace agx
