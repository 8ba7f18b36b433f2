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
#include <condition_variable>
#include <chrono>
#include <cstdlib>
#include <string>
#include <vector>

#include "agx_host.h"

namespace agx {

#define AGX_HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw ::agx::Error{::agx::E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)}; } while (0)

struct MemBlock { void *p = nullptr; size_t n = 0; };

// free blocks of one kind, best fit: the smallest cached block that holds the request and is not more than need / waste_div (+ 64 MB) larger — pinned host blocks: up to twice
// the request; HBM blocks: a sixteenth more at most, because units are admitted to a device by what they NEED (agx_unit_hbm_needed, shard.run_job): a 32 GB unit that takes
// over the 57 GB block of the chromosome before it holds 25 GB that nobody accounted for (r04: the first whole-human job ran out of HBM that way)
class BlockCache {
public:
    bool take(size_t need, MemBlock &out, size_t waste_div = 1) {
        std::lock_guard<std::mutex> g(m_);
        auto it = free_.lower_bound(need);
        if (it == free_.end() || it->first > need + need / waste_div + (64u << 20)) return false;
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

// Units of very different sizes on one device (a whole-human job: 57 GB for chr1 down to 11 GB for chr21, eight in flight) defeat a cache of whole blocks: nothing fits
// what the last unit left, and HBM that was just given back to the driver stalls the next hipMalloc for seconds (profiles/r02_recycle.txt; r04: the first whole-human job
// spent 1-3 s per unit there, 19.5 s for a job whose copies, kernels and walks add up to 2 s).  Blocks from AGX_REGION_MIN bytes on are therefore cut from ONE region per device,
// taken from the driver once (85 % of what is free then) and never given back while anything lives in it: first fit over a list of free ranges that merges neighbours.  A
// request that finds no range waits for a unit to finish (units are admitted by their summed needs — shard.run_job — so room comes; the largest first, so what a finished unit
// leaves holds whoever comes next); with nothing else in the region it fails at once.
#define AGX_REGION_MIN ((size_t)8 << 30)
class DevRegion {
public:
    bool owns(const void *p) const { std::lock_guard<std::mutex> g(m_); return base_ && (const char *)p >= base_ && (const char *)p < base_ + size_; }      // (under the lock: another unit's thread may be making or trimming the region)
    // false: this device has no region and none could be made (the caller takes the driver's path).  `large` = false (a block below AGX_REGION_MIN, or ANY block of an arena
    // that already holds one — a unit that grows a capacity: dev_block's `may_wait`): no region is made for it and it does not wait for room either — a unit that holds a
    // block and waits for more room waits for itself, or for peers that wait the same way; it falls through to the block cache and the driver instead.
    // The region takes AGX_REGION_PERCENT (default 85) of what is free when it is made: ONE process per device is assumed while it exists (agx.h); a host that runs several
    // processes on a device sets the percentage per process, or AGX_NO_REGION=1.
    bool alloc(int device, size_t need, MemBlock &out, bool large = true) {
        std::unique_lock<std::mutex> g(m_);
        if (!base_) {
            if (tried_ || !large) return false;
            tried_ = true;
            size_t fr = 0, tot = 0;
            if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return false; }
            size_t pct = 85; if (const char *e = getenv("AGX_REGION_PERCENT")) { const long v = atol(e); if (v >= 1 && v <= 95) pct = (size_t)v; }
            size_t want = fr / 100 * pct / ((size_t)16 << 20) * ((size_t)16 << 20);
            if (const char *e = getenv("AGX_REGION_GB")) want = (size_t)atoll(e) << 30;      // (tests)
            void *p = nullptr;
            if (want < need || hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return false; }
            base_ = (char *)p; size_ = want; free_[0] = want;
        }
        if (need > size_) return false;
        for (;;) {
            for (auto it = free_.begin(); it != free_.end(); ++it) if (it->second >= need) {
                const size_t off = it->first, len = it->second;
                free_.erase(it);
                if (len > need) free_[off + need] = len - need;
                in_use_++; out = MemBlock{base_ + off, need};
                return true;
            }
            if (in_use_ == 0 || !large) return false;         // (the first cannot happen: an empty region is one range)
            if (cv_.wait_for(g, std::chrono::seconds(120)) == std::cv_status::timeout) throw Error{E_DEVICE, "no room in the device's memory region for two minutes: units in flight exceed the device"};
        }
    }
    void give(const MemBlock &b) {
        { std::lock_guard<std::mutex> g(m_);
          size_t off = (size_t)((char *)b.p - base_), len = b.n;
          auto nx = free_.lower_bound(off);
          if (nx != free_.end() && off + len == nx->first) { len += nx->second; nx = free_.erase(nx); }
          if (nx != free_.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second == off) { off = pv->first; len += pv->second; free_.erase(pv); } }
          free_[off] = len; in_use_--; }
        cv_.notify_all();
    }
    // the tail of a block that was cut from the region goes back to it: [b.p + keep, b.p + b.n) becomes a free range (merged with its neighbours), b keeps its first `keep`
    // bytes (rounded up to 2 MB).  false: not a block of this region, or nothing to give
    bool shrink(MemBlock &b, size_t keep) {
        keep = (keep + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        { std::lock_guard<std::mutex> g(m_);
          if (!base_ || (const char *)b.p < base_ || (const char *)b.p >= base_ + size_ || keep == 0 || keep >= b.n) return false;
          size_t off = (size_t)((char *)b.p - base_) + keep, len = b.n - keep;
          auto nx = free_.lower_bound(off);
          if (nx != free_.end() && off + len == nx->first) { len += nx->second; nx = free_.erase(nx); }
          free_[off] = len; b.n = keep; }
        cv_.notify_all();
        return true;
    }
    void trim() {      // back to the driver, if nothing lives in it
        std::lock_guard<std::mutex> g(m_);
        if (base_ && in_use_ == 0) { (void)hipFree(base_); base_ = nullptr; size_ = 0; free_.clear(); tried_ = false; }
    }
private:
    mutable std::mutex m_; std::condition_variable cv_; char *base_ = nullptr; size_t size_ = 0; std::map<size_t, size_t> free_; size_t in_use_ = 0; bool tried_ = false;
};
inline DevRegion &dev_region(int device) { static DevRegion r[64]; return r[device & 63]; }
inline void dev_give(int device, const MemBlock &b) { if (!b.p) return; if (dev_region(device).owns(b.p)) dev_region(device).give(b); else dev_cache(device).give(b); }

// may_wait: the caller holds nothing on the device yet (an arena's first block), so it may wait for room in the region; a later block of the same arena never waits
inline MemBlock dev_block(int device, size_t need, bool may_wait = true) {
    MemBlock b; need = round_block(need ? need : 1);
    const bool region = !getenv("AGX_NO_REGION");
    if (need >= AGX_REGION_MIN && may_wait && region && dev_region(device).alloc(device, need, b)) return b;
    if (dev_cache(device).take(need, b, 16)) return b;
    // A device whose region exists (some unit was large enough to make it) has handed most of its memory to it: smaller blocks that the cache cannot serve come out of
    // the region as well.  (r04: the quarter-size human job — 8 units of 8-15 GB beside 16 of 3-8 GB — ran out of the 15 % the region had left, gave its cached blocks back to
    // the driver and then waited in hipMalloc: 2.4 s per job instead of 0.2.)  A device that never sees a large unit keeps recycling whole blocks, as before.
    if (region && dev_region(device).alloc(device, need, b, false)) return b;
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
inline void dev_trim(int device) { (void)hipSetDevice(device); for (const MemBlock &c : dev_cache(device).drain()) (void)hipFree(c.p); dev_region(device).trim(); }

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
    void reset() { for (const MemBlock &b : blocks_) dev_give(device, b); blocks_.clear(); at_ = 0; used_ = 0; }
    size_t used() const { return used_; }
    size_t capacity() const { size_t s = 0; for (const MemBlock &b : blocks_) s += b.n; return s; }
    // Everything behind the first `keep` bytes of the arena's only block goes back to the device's region (the caller no longer uses what lies there); what is taken from
    // the arena afterwards comes from a new block.  Returns the bytes given back: 0 if the arena holds several blocks, or its block is not the region's.
    size_t shrink_to(size_t keep) {
        if (blocks_.size() != 1) return 0;
        const size_t before = blocks_[0].n;
        if (!dev_region(device).shrink(blocks_[0], keep)) return 0;
        at_ = blocks_[0].n;
        return before - blocks_[0].n;
    }
    const void *base() const { return blocks_.empty() ? nullptr : blocks_[0].p; }
private:
    std::vector<MemBlock> blocks_; size_t at_ = 0, used_ = 0;
    size_t room() const { return blocks_.empty() ? 0 : blocks_.back().n - at_; }
    void add(size_t bytes) { blocks_.push_back(dev_block(device, bytes, blocks_.empty())); at_ = 0; }
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
    Scratch plain;                           // alloc_plain: ordinary memory from the loaders' scratch cache instead (r06: the file-order forms of a unit that uploads tile-ordered ones are never copied from by the device)
    void alloc(size_t count) {
        if (count <= n && p && !plain.p) return;      // (also a loan from another buffer's block: borrow)
        release();
        if (!count) return;
        blk = host_block(count * sizeof(T)); p = (T *)blk.p; n = blk.n / sizeof(T);
    }
    void alloc_plain(size_t count) {
        if (count <= n && p && plain.p) return;
        release();
        if (!count) return;
        plain.take(count * sizeof(T)); p = (T *)plain.p; n = plain.n / sizeof(T);
    }
    void release() { host_cache().give(blk); blk = MemBlock(); plain.give(); p = nullptr; n = 0; }
    void borrow(T *host, size_t count) { release(); p = host; n = count; }      // a piece of another buffer's block (pinned by its owner, who outlives the loan)
    size_t block_bytes() const { return blk.n; }
    void *dev() const { void *d = nullptr; if (p) AGX_HIP_OK(hipHostGetDevicePointer(&d, p, 0)); return d; }      // the address kernels use for this buffer
    ~PBuf() { release(); }
    PBuf() = default; PBuf(const PBuf &) = delete; PBuf &operator=(const PBuf &) = delete;
};

}  // namespace agx
