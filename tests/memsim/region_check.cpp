// TEST-ONLY: the device-memory policy of aligngraph_amd/csrc/agx_mem.h (block cache, per-device region, DevArena) against a made-up HIP runtime — the functions
// below stand in for libamdhip64, hand out addresses that are never touched and count the calls.  Run by tests/test_mem_region.py; prints "ok" or the first failed check.
#include "../../aligngraph_amd/csrc/agx_mem.h"

#include <atomic>
#include <cstdio>
#include <thread>
#include <unistd.h>

static std::atomic<size_t> g_free_bytes{0}, g_mallocs{0}, g_frees{0}, g_malloc_fail_above{~(size_t)0};
static std::atomic<uintptr_t> g_next{(uintptr_t)1 << 40};
static std::mutex g_m; static std::map<void *, size_t> g_live;

extern "C" {
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "made-up runtime: out of memory"; }
hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *fr = g_free_bytes; *tot = (size_t)288 << 30; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) {
    g_mallocs++;
    if (n > g_free_bytes || n > g_malloc_fail_above) return hipErrorOutOfMemory;
    g_free_bytes -= n; *p = (void *)g_next.fetch_add((n + 4095) & ~(size_t)4095);
    std::lock_guard<std::mutex> l(g_m); g_live[*p] = n; return hipSuccess;
}
hipError_t hipFree(void *p) { g_frees++; std::lock_guard<std::mutex> l(g_m); auto it = g_live.find(p); if (it == g_live.end()) return hipErrorInvalidValue; g_free_bytes += it->second; g_live.erase(it); return hipSuccess; }
hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
}

using namespace agx;
#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); fflush(stdout); _exit(1); } } while (0)      // (_exit: a thread may still be waiting)
static const size_t GB = (size_t)1 << 30;

int main() {
    unsetenv("AGX_NO_REGION"); unsetenv("AGX_REGION_GB");
    // --- device 0: a large unit makes the region; everything the cache cannot serve then comes out of it ---
    g_free_bytes = 280 * GB;
    MemBlock a = dev_block(0, 57 * GB);
    CHECK(g_mallocs == 1 && dev_region(0).owns(a.p) && a.n == 57 * GB);                     // one driver call: 85 % of what was free
    CHECK(g_free_bytes == 280 * GB - 280 * GB / 100 * 85 / (16u << 20) * (16u << 20));
    MemBlock b = dev_block(0, 3 * GB), c = dev_block(0, 300u << 20);
    CHECK(g_mallocs == 1 && dev_region(0).owns(b.p) && dev_region(0).owns(c.p));             // small blocks too, without a driver call
    CHECK((char *)b.p == (char *)a.p + a.n && (char *)c.p == (char *)b.p + b.n);             // first fit, front to back
    dev_give(0, b);
    MemBlock d = dev_block(0, 2 * GB);
    CHECK(d.p == b.p);                                                                       // the hole is used again
    MemBlock e = dev_block(0, 1 * GB);
    CHECK((char *)e.p == (char *)d.p + d.n);                                                 // and what is left of it
    dev_give(0, d); dev_give(0, e); dev_give(0, c);                                          // neighbours merge, in any order
    MemBlock f = dev_block(0, 150 * GB);
    CHECK((char *)f.p == (char *)a.p + a.n);
    // the region is nearly full: a LARGE block waits for room, a small one does not (it goes to the driver)
    MemBlock g = dev_block(0, 30 * GB);
    CHECK(dev_region(0).owns(g.p));                                                          // 57 + 150 + 30 = 237 of 238 GB
    const size_t before = g_mallocs;
    MemBlock s = dev_block(0, 2 * GB);
    CHECK(!dev_region(0).owns(s.p) && g_mallocs == before + 1);
    std::atomic<int> got{0}; MemBlock h;
    std::thread waiter([&] { h = dev_block(0, 40 * GB); got = 1; });
    std::this_thread::sleep_for(std::chrono::milliseconds(200));
    CHECK(got == 0);                                                                         // waiting, not failing, not calling the driver
    CHECK(g_mallocs == before + 1);
    dev_give(0, a);                                                                          // a unit finishes
    waiter.join();
    CHECK(got == 1 && h.p == a.p && dev_region(0).owns(h.p));
    dev_give(0, s);                                                                          // a driver block goes to the cache of whole blocks ...
    MemBlock s2 = dev_block(0, 2 * GB);
    CHECK(s2.p == s.p && g_mallocs == before + 1);                                           // ... and is served from it before the region is asked
    dev_give(0, s2); dev_give(0, f); dev_give(0, g); dev_give(0, h);
    dev_trim(0);                                                                             // nothing lives in it: back to the driver, cache and region
    CHECK(g_free_bytes == 280 * GB && g_live.empty());
    MemBlock again = dev_block(0, 9 * GB);                                                   // and it can be made again
    CHECK(dev_region(0).owns(again.p));
    dev_give(0, again); dev_trim(0);
    // --- device 1: never a large unit: whole blocks, recycled; no region ---
    const size_t m1 = g_mallocs;
    MemBlock u1 = dev_block(1, 6 * GB), u2 = dev_block(1, 5 * GB);
    CHECK(g_mallocs == m1 + 2 && !dev_region(1).owns(u1.p));
    dev_give(1, u1); dev_give(1, u2);
    MemBlock u3 = dev_block(1, 6 * GB);
    CHECK(u3.p == u1.p && g_mallocs == m1 + 2);
    MemBlock u4 = dev_block(1, 3 * GB);                                                      // 5 GB cached is more than a sixteenth (+ 64 MB) above 3 GB: a new block
    CHECK(u4.p != u2.p && g_mallocs == m1 + 3);
    dev_give(1, u3); dev_give(1, u4); dev_trim(1);
    CHECK(g_live.empty());
    // --- device 2: the driver runs out: the cached blocks go back and the call is repeated once ---
    g_free_bytes = 10 * GB;
    MemBlock v1 = dev_block(2, 6 * GB); dev_give(2, v1);
    MemBlock v2 = dev_block(2, 7 * GB);                                                      // 6 GB cached + 4 GB free: only after the cache is drained
    CHECK(v2.p && g_live.size() == 1);
    bool threw = false;
    try { (void)dev_block(2, 7 * GB); } catch (const Error &er) { threw = er.code == E_DEVICE; }
    CHECK(threw);
    dev_give(2, v2); dev_trim(2);
    // --- device 3: a unit larger than what a region could be: no region, the driver's answer ---
    g_free_bytes = 20 * GB;
    threw = false;
    try { (void)dev_block(3, 30 * GB); } catch (const Error &) { threw = true; }
    CHECK(threw && g_live.empty());
    MemBlock w = dev_block(3, 9 * GB);                                                       // (a region is tried once per device until the next trim)
    CHECK(!dev_region(3).owns(w.p));
    dev_give(3, w); dev_trim(3);
    // --- device 4: AGX_NO_REGION ---
    g_free_bytes = 280 * GB; setenv("AGX_NO_REGION", "1", 1);
    MemBlock x = dev_block(4, 57 * GB);
    CHECK(!dev_region(4).owns(x.p) && x.n == 57 * GB);
    dev_give(4, x); dev_trim(4); unsetenv("AGX_NO_REGION");
    // --- a unit's arena: one block for its plan, further blocks when a capacity grows, all of them back at reset ---
    {   DevArena ar; ar.device = 5; g_free_bytes = 280 * GB;
        ar.reserve(20 * GB);
        void *p1 = ar.take(1000), *p2 = ar.take(1);
        CHECK((char *)p2 == (char *)p1 + 1024 && ar.capacity() == 20 * GB && ar.used() == 1024 + 256);
        (void)ar.take(20 * GB);                                                              // does not fit what is left of the first block
        CHECK(ar.capacity() == 40 * GB);
        ar.reset();
        CHECK(ar.capacity() == 0 && ar.used() == 0);
        dev_trim(5);
        CHECK(g_live.empty() && g_free_bytes == 280 * GB);
    }
    // --- a unit that holds a region block and grows a capacity by a LARGE block while the region is full: it must not wait (for itself, or for peers that wait the same
    // way: ADVICE r04) but take the driver's path at once; AGX_REGION_PERCENT sizes the region ---
    {   g_free_bytes = 100 * GB; setenv("AGX_REGION_PERCENT", "50", 1);
        DevArena ar; ar.device = 6;
        ar.reserve(40 * GB);                                                                 // makes the region: 50 GB
        CHECK(g_free_bytes == 50 * GB);
        (void)ar.take(30 * GB);
        const size_t m0 = g_mallocs;
        std::atomic<int> done{0};
        std::thread grow([&] { (void)ar.take(20 * GB); done = 1; });                         // 40 + 20 > 50: no room in the region, and the arena's own block is what fills it
        for (int i = 0; i < 100 && !done; i++) std::this_thread::sleep_for(std::chrono::milliseconds(20));
        CHECK(done == 1);                                                                    // (before r05: waited two minutes for itself, then threw)
        grow.join();
        CHECK(g_mallocs == m0 + 1 && ar.capacity() == 60 * GB);                              // the driver served it
        ar.reset(); dev_trim(6); unsetenv("AGX_REGION_PERCENT");
        CHECK(g_live.empty());
    }
    // --- a unit gives the tail of its block back before it is done (agx_unit_trim after the download): the room is there for the next unit at once, what the arena takes
    // afterwards comes from a new block, and everything returns at reset ---
    {   g_free_bytes = 100 * GB;
        DevArena ar; ar.device = 7;
        ar.reserve(60 * GB);                                                                 // makes the region: 85 GB
        const char *b0 = (const char *)ar.take(10 * GB);
        (void)ar.take(45 * GB);
        MemBlock other; std::atomic<int> got{0};
        std::thread waiter([&] { other = dev_block(7, 40 * GB); got = 1; });                 // 60 + 40 > 85: waits
        std::this_thread::sleep_for(std::chrono::milliseconds(150));
        CHECK(got == 0);
        const size_t freed = ar.shrink_to(10 * GB + 12345);                                  // keep the front (rounded up to 2 MB), give the rest back
        CHECK(freed == 60 * GB - (10 * GB + (2u << 20)) && ar.capacity() == 10 * GB + (2u << 20));
        waiter.join();
        CHECK(got == 1 && dev_region(7).owns(other.p) && (const char *)other.p == b0 + 10 * GB + (2u << 20));      // right behind what was kept
        void *late = ar.take(1000);                                                          // (a record fetch after the trim: a new block, never the room that was given back)
        CHECK(late != nullptr);
        CHECK((const char *)late < b0 || (const char *)late >= (const char *)other.p + other.n);
        CHECK(ar.shrink_to(1 * GB) == 0);                                                    // several blocks now: nothing more to give
        dev_give(7, other); ar.reset(); dev_trim(7);
        CHECK(g_live.empty() && g_free_bytes == 100 * GB);
        DevArena small; small.device = 8; g_free_bytes = 100 * GB;                           // a unit below the region's threshold on a device without a region: nothing to give
        small.reserve(2 * GB); (void)small.take(1 * GB);
        CHECK(small.shrink_to(1 * GB) == 0 && small.capacity() == 2 * GB);
        small.reset(); dev_trim(8);
        CHECK(g_live.empty());
    }
    printf("ok\n");
    return 0;
}
