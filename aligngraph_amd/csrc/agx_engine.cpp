// agx_engine.cpp — unit object, device memory, kernel sequencing and the C-ABI of libagx.so (include/agx.h).
//
// One agx_unit = one reference unit (chromosome or --part slice) = the body of the reference's unit loop
// (AG:4765-4783).  In the application every unit is NEW data and is built once, so what a unit costs is the whole way
// from its packed arrays in host memory to its output bytes in host memory (SURVEY §8d's T_core):
//   stage    (when the arrays are handed over, outside T_core) the arrays the device wants are packed into pinned memory: hits and runs
//            as they are, the read bases as 4-bit classes, the conti-mer keys without the walk's fields;
//   upload   one block of HBM for everything the unit will ever hold (agx_mem.h), asynchronous copies at PCIe rate on the unit's own
//            stream — beside other units' kernels —, two small kernels (conti-mer heads, vote codes), an event;
//   build    all kernels queued back to back on the device's two build streams behind that event, ONE synchronisation, capacities
//            chosen so that a first build does not have to be repeated;
//   download the walk graph into cached pinned buffers; finish = the sequential host walk.
// No CPU fallback exists: a host without a HIP device gets AGX_E_NOGPU from agx_unit_create.
#include <hip/hip_runtime_api.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <mutex>
#include <system_error>
#include <thread>
#include <pthread.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <vector>

#include "agx_host.h"
#include "agx_mem.h"

#include "agx_kargs.h"

using namespace agx;

namespace {

#define HIP_OK(expr) AGX_HIP_OK(expr)

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// AGX_TRACE=1: one stderr line per stage of every unit with host times (ms since the first line): who waited for whom in a pipelined job
static const bool g_trace = getenv("AGX_TRACE") != nullptr;
void trace(const void *unit, const char *what, double from_ms, size_t n_pos) {
    if (!g_trace) return;
    static const double t00 = now_ms();
    // (+ the CPU time of the calling thread since its previous line: a stage that waits should cost none)
    static thread_local double cpu_last = 0;
    timespec ts{}; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); const double cpu = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    fprintf(stderr, "[agx trace] unit %p (%zu pos) %-22s %9.2f -> %9.2f ms   (thread cpu +%.2f ms)\n", unit, n_pos, what, from_ms - t00, now_ms() - t00, cpu - cpu_last);
    cpu_last = cpu;
}

// Wait for an event asleep.  hipEventSynchronize spins whatever the event's flags say (ROCm 7.2: the five unit threads of a cfg3 job burned 25 ms of CPU each waiting
// for uploads and builds — a third of the job's CPU time, beside the walkers of the units in front of them, on a box whose control group grants 16 CPUs): a short
// spin for what is about to finish, then queries between short sleeps (a sleep of 20 us takes 70-80 with the default timer slack: that is the price of a wake-up).
hipError_t wait_event(hipEvent_t e) {
    static const bool spin = getenv("AGX_SPIN_WAIT") != nullptr;
    if (spin) return hipEventSynchronize(e);
    for (int i = 0; i < 64; i++) { const hipError_t r = hipEventQuery(e); if (r != hipErrorNotReady) return r; }
    for (;;) {
        const timespec ts{0, 20000}; nanosleep(&ts, nullptr);
        const hipError_t r = hipEventQuery(e); if (r != hipErrorNotReady) return r;
    }
}

// Section boundaries of a build on the build streams: the end of one section is the start of the next (one record instead of two: an
// event record costs the stream about as much as a small kernel).
enum { B_START = 0, B_PREP, B_BIN, B_NODE, B_BIG, B_EDGE, B_SLOW, B_COMPACT, B_N };
// (two more events bracket the whole build: Boundaries::first / last)
struct Boundaries {
    hipEvent_t e[B_N] = {}, first = nullptr, last = nullptr; bool at[B_N] = {}; bool all = false;      // at[b]: boundary b was marked in the last build; all: mark every boundary (AGX_FLAG_TIME_SECTIONS)
    void init() { for (auto &x : e) HIP_OK(hipEventCreate(&x)); HIP_OK(hipEventCreate(&first)); HIP_OK(hipEventCreate(&last)); }
    double span() const { float f = 0; return hipEventElapsedTime(&f, first, last) == hipSuccess ? f : 0; }
    void destroy() { for (auto &x : e) { if (x) (void)hipEventDestroy(x); x = nullptr; } if (first) (void)hipEventDestroy(first); if (last) (void)hipEventDestroy(last); first = last = nullptr; }
    void begin() { for (bool &x : at) x = false; }
    void mark(int b, hipStream_t st) { if (!all && b != B_BIN && b != B_NODE) return; HIP_OK(hipEventRecord(e[b], st)); at[b] = true; }      // the node sweep is always timed
    double ms(int b) const { if (!at[b - 1] || !at[b]) return 0; float f = 0; (void)hipEventElapsedTime(&f, e[b - 1], e[b]); return f; }      // section that ends at boundary b
};

template <class F> void on_threads(unsigned threads, F fn) {      // (as in agx_host.cpp: nothing leaves a worker thread as an exception)
    std::vector<std::thread> th; std::vector<std::exception_ptr> ex(threads);
    auto guarded = [&](unsigned t) { try { fn(t); } catch (...) { ex[t] = std::current_exception(); } };
    std::vector<unsigned> mine{0u};
    for (unsigned t = 1; t < threads; t++) { try { th.emplace_back(guarded, t); } catch (const std::system_error &) { mine.push_back(t); } }
    for (unsigned t : mine) guarded(t);
    for (auto &x : th) x.join();
    for (auto &e : ex) if (e) std::rethrow_exception(e);
}

}  // namespace

// Downloads through the HSA runtime's asynchronous copy, i.e. the SDMA engines.  hipMemcpyAsync does a device-to-host copy into registered
// memory with a copy KERNEL of its own (rocprofv3: __amd_rocclr_copyBuffer), and a kernel that stores to host memory slows what runs beside
// it — under rocprofv3 the next unit's agx_k_tile_sort ran 2.6-4.2 ms instead of 0.8 and the device window of a cfg3 job was 49 ms instead of
// 42 (profiles/r02_f_timeline.txt vs r02_g_timeline.txt; un-profiled the job takes the same 59 ms either way: the builds are not what the
// last walks wait for).  The engines move the same bytes at the same 56 GB/s (tests/tools/sdma_d2h.cpp) and leave the CUs alone.  Set up
// once per process; if anything about it fails (AGX_NO_SDMA_DOWNLOAD forces that) the copies go through HIP.
struct HsaCopy {
    bool ok = false; hsa_agent_t cpu{}; std::vector<hsa_agent_t> gpu; std::vector<std::string> uuid;      // uuid: "GPU-<16 hex digits>" of each GPU agent
    static hsa_status_t on_agent(hsa_agent_t a, void *p) {
        HsaCopy *H = (HsaCopy *)p; hsa_device_type_t t;
        if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
        if (t == HSA_DEVICE_TYPE_CPU && !H->cpu.handle) H->cpu = a;
        if (t == HSA_DEVICE_TYPE_GPU) {
            char id[32] = {0};
            (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_UUID, id);
            H->gpu.push_back(a); H->uuid.push_back(std::string(id, strnlen(id, 21)));
        }
        return HSA_STATUS_SUCCESS;
    }
    HsaCopy() {
        if (getenv("AGX_NO_SDMA_DOWNLOAD")) return;
        if (hsa_init() != HSA_STATUS_SUCCESS) return;
        if (hsa_iterate_agents(on_agent, this) != HSA_STATUS_SUCCESS) return;
        ok = cpu.handle != 0 && !gpu.empty();
    }
    // the agent of HIP device `device`, by its UUID (the two runtimes need not number the devices alike, and report different PCI
    // addresses inside a container); one device on either side is that device; false: not found
    bool agent_of(int device, hsa_agent_t &out) const {
        static std::mutex m; static int known[64]; static hsa_agent_t agent[64];      // known: 0 not asked yet, 1 found, -1 not found
        std::lock_guard<std::mutex> l(m);
        int &k = known[device & 63];
        if (!k) {
            k = -1;
            hipDeviceProp_t prop;
            if (ok && hipGetDeviceProperties(&prop, device) == hipSuccess) {
                const std::string want = "GPU-" + std::string(prop.uuid.bytes, strnlen(prop.uuid.bytes, 16));
                int n_hip = 0;
                for (size_t i = 0; i < gpu.size(); i++) if (uuid[i] == want && want.size() > 4) { agent[device & 63] = gpu[i]; k = 1; break; }
                if (k != 1 && gpu.size() == 1 && hipGetDeviceCount(&n_hip) == hipSuccess && n_hip == 1) { agent[device & 63] = gpu[0]; k = 1; }
            }
        }
        if (k == 1) out = agent[device & 63];
        return k == 1;
    }
};
const HsaCopy &hsa_copy() { static const HsaCopy h; return h; }

// positions from which a unit's host walk is split between two walkers (= walk_split's default in agx_walk.cpp; AGX_WALK_SPLIT_MIN, read at every download, overrides both: tests)
inline size_t two_walkers_min() { const char *e = getenv("AGX_WALK_SPLIT_MIN"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)4000000; }
#define AGX_TWO_WALKERS_MIN two_walkers_min()
// walkers for a unit of n_pos positions (walkers_for, agx_host.h)
inline int walkers_wanted(size_t n_pos) { return n_pos < AGX_TWO_WALKERS_MIN ? 1 : walkers_for(n_pos); }

// One helper thread per unit, started with the unit and asleep until it is handed work: what a unit can prepare while its upload and
// build run (the pinned download buffers, the output buffers) without its worker waiting for it.  Not started on demand: creating a
// thread maps a stack, and that waits for the address-space lock that another unit's hipHostRegister holds for milliseconds.
struct UnitHelper {
    enum { DL = 0, OUT = 1, WALK = 2, SLOTS = 3 };      // (taken in this order)
    std::thread th; std::mutex m; std::condition_variable cv;
    std::function<void()> job[SLOTS]; bool queued[SLOTS] = {false, false, false}, running[SLOTS] = {false, false, false}; bool stop = false, started = false;
    void start() {
        if (started) return;
        th = std::thread([this] {
            pthread_setname_np(pthread_self(), "agx-helper");
            std::unique_lock<std::mutex> l(m);
            for (;;) {
                cv.wait(l, [this] { return stop || queued[DL] || queued[OUT] || queued[WALK]; });
                if (stop) return;
                const int s = queued[DL] ? DL : queued[OUT] ? OUT : WALK;
                std::function<void()> f = std::move(job[s]); queued[s] = false; running[s] = true;
                l.unlock();
                try { f(); } catch (...) { }
                l.lock(); running[s] = false; cv.notify_all();
            }
        });
        started = true;
    }
    bool submit(int s, std::function<void()> f) {       // false: no helper (the caller does the work itself, later)
        if (!started) return false;
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !queued[s] && !running[s]; });
        job[s] = std::move(f); queued[s] = true; cv.notify_all();
        return true;
    }
    void wait(int s) { if (!started) return; std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !queued[s] && !running[s]; }); }
    ~UnitHelper() { if (started) { { std::lock_guard<std::mutex> l(m); stop = true; } cv.notify_all(); th.join(); } }
};

// The further walkers of large units (agx_walk.cpp: walk_split; the first extra one is the unit's own helper): a small pool for the process, made
// when the first large unit is finished.  A unit takes what is free and walks with fewer walkers if that is less than it wanted.
struct WalkerPool {
    enum { N = 64 };
    struct Slot { std::thread th; std::mutex m; std::condition_variable cv; std::function<void()> job; bool queued = false, running = false, taken = false, stop = false; };
    Slot slot[N]; std::mutex take_m; bool started = false;
    void start() {
        if (started) return;
        started = true;
        for (Slot &s : slot) {
            try {
                s.th = std::thread([&s] {
                    pthread_setname_np(pthread_self(), "agx-walker");
                    std::unique_lock<std::mutex> l(s.m);
                    for (;;) {
                        s.cv.wait(l, [&s] { return s.stop || s.queued; });
                        if (s.stop) return;
                        std::function<void()> f = std::move(s.job); s.queued = false; s.running = true;
                        l.unlock();
                        try { f(); } catch (...) { }
                        l.lock(); s.running = false; s.cv.notify_all();
                    }
                });
            } catch (...) { s.taken = true; }        // (never handed out)
        }
    }
    int take() { std::lock_guard<std::mutex> l(take_m); start(); for (int i = 0; i < N; i++) if (!slot[i].taken) { slot[i].taken = true; return i; } return -1; }
    void give(int i) { std::lock_guard<std::mutex> l(take_m); slot[i].taken = false; }
    void run(int i, std::function<void()> f) { Slot &s = slot[i]; std::unique_lock<std::mutex> l(s.m); s.cv.wait(l, [&s] { return !s.queued && !s.running; }); s.job = std::move(f); s.queued = true; s.cv.notify_all(); }
    void wait(int i) { Slot &s = slot[i]; std::unique_lock<std::mutex> l(s.m); s.cv.wait(l, [&s] { return !s.queued && !s.running; }); }
    ~WalkerPool() { for (Slot &s : slot) if (s.th.joinable()) { { std::lock_guard<std::mutex> l(s.m); s.stop = true; } s.cv.notify_all(); s.th.join(); } }
};
WalkerPool &walker_pool() { static WalkerPool p; return p; }
// Walkers for the unit whose walk begins now: what its size asks for (walkers_wanted), one per CPU this process may use at most — its share of them when it is
// one of several ranks on the host (LOCAL_WORLD_SIZE: torch.distributed's launcher sets it), four at least.  The walks of a pipelined job overlap two or three
// at a time, so more walkers than CPUs are runnable for a millisecond or two; a CPU quota (cgroup cpu.max) bounds CPU time per period, not threads, and a job's
// walks use well under a third of it (bench.py: host_cpu_ms_per_step).  r03 first gave a walk half the CPUs (8 of 16): cfg3 jobs of 36.2 ms instead of 35.1.
inline int walkers_now(size_t n_pos) {
    const int want = walkers_wanted(n_pos);
    if (want <= 4 || getenv("AGX_WALK_SPLIT_WALKERS")) return want;
    static const int ranks = [] { const char *e = getenv("LOCAL_WORLD_SIZE"); const int n = e ? atoi(e) : 1; return n < 1 ? 1 : n; }();
    const int cpus = (int)usable_cpus() / ranks;
    return want <= cpus ? want : cpus < 4 ? 4 : cpus;
}

// units of this process that are uploaded and not yet walked: the walk of the LAST one runs alone (the tail of a job) and may take every walker there is
static std::atomic<int> g_walks_pending{0};
// units that exist in this process: when the last one is destroyed the output buffers parked for "the next unit" (agx_host.h: out_cache) go back to the C library — an embedder
// that never calls agx_pool_trim(-1) then keeps at most what it has itself given back since (up to 16 GB stayed resident until exit)
static std::atomic<int> g_units_alive{0};

struct agx_unit {
    agx_params prm{};
    std::string err;
    Threads T; Pairs P;                 // what the loaders / the packed-array calls filled (empty when the unit came out of a cache file)
    UnitView V;                         // what everything downstream reads: into T / P, or into the mapped cache file
    struct Mapped { void *p = nullptr; size_t n = 0; void reset() { if (p) munmap(p, n); p = nullptr; n = 0; } ~Mapped() { reset(); } } cache_map;
    agx_u32 n_seg0 = 0, stride = 0, n_slots = 0, n_rows = 0; unsigned long long pairs_in_file = 0, sam_pairs = 0;
    bool have_ref = false, have_threads = false, staged = false, uploaded = false, built = false, downloaded = false;
    bool consumed = false;             // AGX_FLAG_ONE_SHOT: the download has overwritten the staged inputs
    bool pending_walk = false;         // counted in g_walks_pending: uploaded, not yet walked (or released)
    bool expanded = false;             // the conti-mer tables and vote codes have been made from what was uploaded (opens the unit's first build)
    hipEvent_t ev_built = nullptr;     // this unit's build commands are done (waited for on the host; ev_dl: its download)
    UnitOutput out; OutBuf out_initial; bool out_ready = false;      // output buffers of the next finish, reserved and touched by a helper thread while the unit is uploaded and built (prepare_outputs)
    UnitHelper helper;
    hsa_signal_t dl_signal{}; hsa_agent_t dl_agent{}; bool dl_sdma = false;      // downloads by the SDMA engines (HsaCopy)
    // streamed download (begin_streamed_download): piece 0 = what the walk needs before it can plan (side positions, overflow edges), 1 .. n = position windows from the front,
    // n + 1 = the bases; cut_main / cut_side: the ids in front of window w (entry n: all of them)
    enum { DL_SIGNALS = AGX_DL_PIECES + 2 };
    hsa_signal_t dl_piece[DL_SIGNALS] = {}; bool dl_piece_made = false, dl_streaming = false; agx_cut_args cuts{}; agx_u32 cut_main[AGX_DL_PIECES + 1] = {}, cut_side[AGX_DL_PIECES + 1] = {};
    double dl_t0 = 0, dl_est_ms = 0; size_t dl_stream_bytes = 0; std::atomic<bool> dl_timed{false};
    DevArena arena;
    // staged inputs: what the device wants of T and P, in pinned memory (stage_inputs)
    PBuf<agx_whit> s_hits; PBuf<agx_wside> s_sides; PBuf<agx_wrun> s_runs; PBuf<agx_u8> s_codes; PBuf<unsigned long long> s_other; PBuf<agx_u32> s_jump; size_t n_other = 0, n_sides = 0, n_jump = 0;      // the read alignments in the wire formats of agx_core.h
    PBuf<agx_u8> s_ref; PBuf<agx_refx> s_refx; size_t n_refx = 0; bool ref_packed = false;      // the unit sequence: 2 bits per base + the stretches of other bytes (ref_packed), or the bytes as they are
    PBuf<agx_u32> s_chain_end, s_region_off; PBuf<agx_cmseg> s_segs; size_t n_segs = 0;
    // r06, tile-ordered upload (stage_tiled): the wire records and the read rows in the order of the hits' first tiles — what the device is sent instead of s_hits / s_perm / s_codes / s_other
    PBuf<agx_whit> s_hits_t; PBuf<agx_u8> s_codes_t; PBuf<unsigned long long> s_other_t; size_t n_other_t = 0; bool tiled = false; std::vector<agx_u32> slot_row;      // slot_row[i] = staged row of the i-th hit's left mate (the walk's k-mer tails)
    agx_u32 n_win = 1, win_tile[9] = {};      // a unit's first build sweeps windows of tiles as their rows land: window w = tiles [win_tile[w], win_tile[w + 1])
    hipEvent_t ev_rows[8] = {}, ev_win[8] = {}, ev_sw0[8] = {}, ev_sw1[8] = {};      // rows of window w in HBM / expanded; around window w's sweep (its time is the sum over the windows)
    PBuf<agx_u32> s_perm, s_tfirst, s_jump_at; agx_u32 lookback = 2;      // the hits in the order of their first tile (stage_order): perm[i] = the i-th hit of that order, tile_first[t] = hits in front of tile t's own; pass J's hits as places in it
    PBuf<char> s_landing;               // one-shot units: what the download needs beyond the dead staged inputs it lands in, pinned when the unit is staged (not inside T_core)
    PBuf<agx_cntrun> s_cntruns; PBuf<agx_chunk> s_cntchunks, s_segchunks; PBuf<agx_u32> s_segindex; size_t n_cntruns = 0, n_cntchunks = 0, n_segchunks = 0, n_segindex = 0;      // what the device builds the conti-mer tables from (build_cm_layout)
    // the read rows in their upload form (agx_core.h "read rows relative to the reference"; made at the end of staging: stage_rows): count byte per row, offsets of the
    // 64-row blocks in the stream, the stream of 16-bit units, one bit per hit (its row's anchor).  rows_diffed = false: the rows cross as their 2-bit classes (s_codes)
    PBuf<agx_u16> s_units; PBuf<agx_u8> s_rowcnt; PBuf<agx_u32> s_blockoff, s_blockfirst, s_anchor; size_t n_units = 0, n_rowcnt = 0, n_blockoff = 0, n_blockfirst = 0, n_anchor = 0, n_rows_explicit = 0; bool rows_diffed = false;
    size_t nh = 0, n_runs = 0, n_cm = 0, n_codes = 0; agx_u32 maxlen = 0;      // n_codes: bytes of packed classes (four bases each); n_other: listed bases that are not A, C, G, T
    std::vector<agx_u32> row_slot;      // staged read bases: one row per (pair, a mate) that some hit uses; row -> read slot (general loader / agx_unit_push_pairs)
    std::vector<uint64_t> row_off;      // fast loader: row -> where the read's bases start in the mapped reads file (the bases are never copied: the walk reads the k-mer tails of written records there)
    std::shared_ptr<agx::ReadsIndex> reads_keep;      // the reads file that row_off points into (shared with the caller's agx_reads, or the unit's own)
    std::unique_ptr<agx::FileView> reads_map;          // the same for a unit that came out of its cache file: only the mapping, no index
    bool pairs_staged = false;          // the fast loader has written hits, runs, codes and the list of other bases straight into the staged buffers
    agx::PairsFile pairs_file;          // the unit's read alignments were handed over staged (tmp/_agx_pairs.<u>.bin, agx_host.h): the mapped file, which the walk takes its k-mer tails from
    // inputs on the device
    DBuf<agx_u32> d_cm_start, d_cm_cnt; DBuf<agx_cmkey> d_cm; DBuf<agx_cmhead> d_cm_head; DBuf<char> d_ref; DBuf<agx_cmseg> d_segs; DBuf<unsigned long long> d_up_desc;
    DBuf<agx_run> d_runs; DBuf<agx_u8> d_codes, d_vcodes; DBuf<unsigned long long> d_other;
    DBuf<agx_u16> d_units; DBuf<agx_u8> d_rowcnt; DBuf<agx_u32> d_blockoff, d_blockfirst, d_anchor;
    DBuf<agx_cntrun> d_cntruns; DBuf<agx_chunk> d_cntchunks, d_segchunks; DBuf<agx_u32> d_jump, d_segindex;
    DBuf<agx_whit> d_whits; DBuf<agx_wside> d_wsides; DBuf<agx_wrun> d_wruns; DBuf<agx_u8> d_wref; DBuf<agx_refx> d_refx;      // what was uploaded, until the first build has expanded it
    // derived
    DBuf<agx_dhit> d_dhit; DBuf<agx_u32> d_tile_cnt, d_tile_off, d_cursor, d_unsorted, d_tile_recs, d_scan_tmp, d_words; DBuf<unsigned long long> d_scan_desc; size_t scan_desc_n = 0;      // descriptors of the three one-launch scans   // d_words: counters/status
    // node table
    agx_u32 pool_cap = 0, spill_lo = 0, ovf_cap = 0, list_cap = 0, sp_cap = 0;
    DBuf<agx_u32> d_pool_cnt, d_region_off; agx_u32 n_regions = 0;      // the node pool's slices (AGX_REGION_TILES tiles each) and their counters
    DBuf<agx_u32> d_node_start, d_slow_list, d_perm, d_tfirst, d_ckey, d_long; DBuf<agx_u16> d_node_cnt; DBuf<agx_u8> d_pos_succ;
    DBuf<agx_u32> d_cid, d_coff, d_cid0, d_coff0, d_off0, d_next; DBuf<agx_u8> d_base, d_flags; DBuf<agx_sref> d_sref; DBuf<int> d_counts;
    DBuf<agx_edge_ovf> d_ovf; DBuf<agx_u32> d_mid_list, d_big_list, d_scratch, d_huge_list, d_scratch_huge; bool huge = false, dense = false;      // dense: the scatter fallback of the tile lists is queued (a build met more than AGX_LONG_MAX long hits); huge: pass 3 of the node sweep is queued (a build met a position beyond AGX_MAXV_BIG variants)
    // walk graph (agx_core.h "walk preparation")
    agx_u32 n_ids = 0;
    DBuf<agx_u32> d_side_pk, d_tile_side, d_tile_side_start, d_aid_of; DBuf<char> d_a_str;
    DBuf<agx_u8> d_a_meta, d_a_mark; DBuf<agx_walknode> d_fetch, d_sp_node; DBuf<agx_u32> d_a_nid; DBuf<agx_edge_ovf> d_a_ovf; agx_compact_args walk_args{};      // walk_args: the last build's, for record fetches
    DBuf<agx_u32> d_chain_end, d_side_xpos, d_sp_cnt, d_sp_rank; DBuf<unsigned long long> d_sp_bits;
    agx_u32 n_chain_end = 0, n_special = 0, n_words = 0;
    // downloaded: the walk graph with its sparse record table (agx_core.h); the full record table stays on the device
    PBuf<char> h_a_str; PBuf<agx_u8> h_a_meta; PBuf<agx_u32> h_side_xpos, h_sp_rank; PBuf<unsigned long long> h_sp_bits;
    PBuf<agx_walknode> h_sp_node, h_fetch; PBuf<agx_edge_ovf> h_a_ovf; PBuf<agx_hop> h_sp_hop; DBuf<agx_hop> d_sp_hop;
    PBuf<agx_u32> h_words;
    agx_u32 n_nodes = 0, n_ovf = 0, n_tiles = 0, n_tile_entries = 0, n_big = 0, n_mid = 0;
    Boundaries ev; hipEvent_t ev_front = nullptr, ev_passA = nullptr, ev_passJ = nullptr, ev_up0 = nullptr, ev_uploaded = nullptr, ev_dl = nullptr, ev_hits = nullptr;      // ev_hits: everything but the read bases is in HBM
    bool up_timed = false;
    agx_stats stats{};
    ~agx_unit() {
        if (pending_walk) g_walks_pending.fetch_sub(1);
        ev.destroy();
        for (hipEvent_t e : {ev_front, ev_passA, ev_passJ, ev_up0, ev_uploaded, ev_dl, ev_built, ev_hits}) if (e) (void)hipEventDestroy(e);
        for (auto *arr : {ev_rows, ev_win, ev_sw0, ev_sw1}) for (int i = 0; i < 8; i++) if (arr[i]) (void)hipEventDestroy(arr[i]);
        if (dl_signal.handle) (void)hsa_signal_destroy(dl_signal);
        if (dl_piece_made) for (hsa_signal_t g : dl_piece) if (g.handle) (void)hsa_signal_destroy(g);
    }
};

namespace {

// One build at a time per device, on the GPU as well as in the launch order.  All builds of a device are queued on TWO streams, under a
// mutex that a host thread holds only while it enqueues its unit's kernel chain: `main` carries the node sweep and everything behind it
// (edge passes, walk preparation), `front` what comes before the sweep (zeroing, hit preparation, binning).  Units built from several
// host threads run their chains back to back — no host round trip in between — and the FRONT of build n+1 runs beside the BACK of build n:
// it starts when sweep n is done (an event) and sweep n+1 waits for it (another).  The node sweep itself never shares the device, so its
// HIP-event time is that of an exclusive GPU; the other sections are timed in exclusive builds (AGX_FLAG_TIME_SECTIONS serialises the two
// streams).
// Four streams per device serve all of its units: `up` carries the units' upload copies and nothing else, one unit after the other in the
// order they were queued (PCIe is one pipe: five uploads that share it all finish late; first in, first built, and its host walk runs
// beside the uploads of the rest); `front` and `main` carry the builds (see do_build); `down` the downloads.  Not a stream per unit: HIP
// maps its streams onto four hardware queues, and two streams that share one run their commands in the order they were queued — a
// build's first kernels sat 6 ms behind ANOTHER unit's upload copies, downloads 6-11 ms behind other units' kernels (profiles/r02_timeline_*.txt).
struct DeviceTurn { std::mutex m; hipStream_t main = nullptr, front = nullptr, up = nullptr, down = nullptr;
                    hipEvent_t sweep_done[2] = {nullptr, nullptr}, build_done[2] = {nullptr, nullptr};
                    unsigned long long n = 0; bool prev_exclusive = false; hipEvent_t prev_node = nullptr;      // n: builds queued so far; events alternate between two handles
                    std::mutex up_m, down_m, make_m;
                    void make() {       // (device current)
                        std::lock_guard<std::mutex> l(make_m);
                        if (main) return;
                        HIP_OK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking)); HIP_OK(hipStreamCreateWithFlags(&front, hipStreamNonBlocking));
                        HIP_OK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
                        for (auto &e : sweep_done) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                        for (auto &e : build_done) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                        HIP_OK(hipStreamCreateWithFlags(&main, hipStreamNonBlocking));
                    } };
DeviceTurn &turn_of(int device) { static DeviceTurn turns[64]; return turns[device & 63]; }

// AGX_TRACE_GAP=1: diagnostic for loops that rebuild long-lived units (see do_build)
static const bool g_trace_gap = getenv("AGX_TRACE_GAP") != nullptr;
// AGX_SCAN_LEGACY=1: the scans of a build as three launches each instead of one (decoupled look-back)
static const bool g_scan1 = getenv("AGX_SCAN_LEGACY") == nullptr;
// AGX_DEBUG_SYNC=1: synchronise after every launch group of a build and name it on stderr — a memory fault then points at its kernel
static const bool g_debug_sync = getenv("AGX_DEBUG_SYNC") != nullptr;
// AGX_TEST_SMALL_CAPS=1 (tests; read at every upload): every capacity starts absurdly small, so that every regrow path runs
#define g_tiny (getenv("AGX_TEST_SMALL_CAPS") != nullptr)
#define AGX_CHECKPOINT(name) do { if (g_debug_sync) { hipError_t e_ = hipStreamSynchronize(st); fprintf(stderr, "[agx debug] %s: %s\n", name, hipGetErrorString(e_)); } } while (0)

void join_dl_helper(agx_unit *u);
void drop_outputs(agx_unit *u);     // before a unit's inputs change: the helper that prepares the output buffers reads them
void start_helper(agx_unit *u);

enum { W_CUT = 34 /* [2 * (AGX_DL_PIECES + 1)] the cuts of a streamed download: agx_cut_args */, W_TOTAL = W_CUT + 2 * (AGX_DL_PIECES + 1) };
enum { W_ERR = 0, W_POOL = 1, W_BIGCOUNT = 2, W_STATUS = 3, W_OVFCOUNT = 4, W_SLOWCOUNT = 5, W_LONGCOUNT = 6 /* hits that span more tiles than a list's window looks back over */, W_UNUSED7 = 7, W_MIDCOUNT = 8, W_JUMPCOUNT = 9, W_SPILL = 10, W_HUGECOUNT = 11, W_N = 12 };

// Where a streamed download cuts a unit of n_pos positions (r06): windows of about 4 M positions, 2 to AGX_DL_PIECES of them, at multiples of 64 ids (a word of the special-id bitmap,
// a tile of the side-id prefix).  None for units that one walker walks (nothing could begin earlier) or on request.  AGX_STREAM_PIECES=n forces n (tests: small units).
agx_cut_args stream_cuts(agx_u32 n_pos) {
    agx_cut_args C; memset(&C, 0, sizeof C);
    const char *force = getenv("AGX_STREAM_PIECES");
    if (getenv("AGX_NO_STREAM_DOWNLOAD") || n_pos < 4096) return C;
    size_t m = force ? (size_t)atoi(force) : n_pos < AGX_TWO_WALKERS_MIN ? 0 : n_pos / 4000000u;
    if (!force && m && m < 2) m = 2;
    if (m > AGX_DL_PIECES) m = AGX_DL_PIECES;
    if (m > n_pos / 1024u) m = n_pos / 1024u;
    for (size_t w = 0; w < m; w++) C.word[w] = (agx_u32)(((unsigned long long)n_pos * w / m) >> 6);
    C.word[m] = n_pos >> 6; C.n = (agx_u32)m;
    return C;
}

void fill_sweep_args(agx_unit *u, agx_sweep_args &S) {
    memset(&S, 0, sizeof S);
    S.cm_start = u->d_cm_start.p; S.cm = u->d_cm.p; S.cm_head = u->d_cm_head.p; S.ref = u->d_ref.p;
    S.dhit = u->d_dhit.p; S.runs = u->d_runs.p; S.vcodes = u->d_vcodes.p; S.stride = u->stride;
    S.tile_off = u->d_tile_off.p; S.tile_recs = (decltype(S.tile_recs))u->d_tile_recs.p;
    S.n_pos = (agx_u32)u->V.n_pos; S.n_tiles = u->n_tiles; S.k = u->prm.k; S.iv = (int)u->prm.insert_variation; S.coverage = (int)u->prm.coverage;
    S.node_start = u->d_node_start.p; S.node_cnt = u->d_node_cnt.p; S.pos_succ = u->d_pos_succ.p; S.side_pk = u->d_side_pk.p; S.tile_side = u->d_tile_side.p;
    S.nk_cid = u->d_cid.p; S.nk_coff = u->d_coff.p; S.nk_cid0 = u->d_cid0.p; S.nk_coff0 = u->d_coff0.p; S.nk_off0 = u->d_off0.p;
    S.n_base = u->d_base.p; S.n_flags = u->d_flags.p; S.n_sref = u->d_sref.p; S.n_next = u->d_next.p;
    S.n_counts = (u->prm.flags & AGX_FLAG_KEEP_COUNTS) ? u->d_counts.p : nullptr;
    S.pool_cap = u->pool_cap;
    S.sweep_stats = nullptr;
}

// ---- staging: the packed arrays a unit was handed, in the form and the memory the upload wants -------------------------------------
// Runs when the arrays are handed over (end of agx_unit_load_files, or agx_unit_stage after the last agx_unit_push_pairs; implied by an
// upload that finds nothing staged).  The read bases cross PCIe as 2-bit classes plus a list of the few bases that are not A, C, G or T
// (agx_pack_classes2): a quarter of the bytes of the largest array.  The read alignments are staged by stage_pairs() (agx_load.cpp) —
// or were written into the staged buffers by the fast loader while it parsed (u->pairs_staged).
// The file-order wire records and read rows of a unit that will upload their tile-ordered forms (stage_tiled) are only ever read by the host — the gather into those forms, the
// cache file — so they live in ordinary memory: pinning a second copy of a unit's two largest arrays cost its load 0.2 ms per MB.  r05's forms (AGX_NO_TILED_UPLOAD) are uploaded
// from them and keep them pinned.
inline bool want_tiled() { return !getenv("AGX_NO_TILED_UPLOAD"); }
template <class B> inline void alloc_file_order(B &buf, size_t count) { if (want_tiled()) buf.alloc_plain(count); else buf.alloc(count); }
struct UnitSink : StageSink {
    agx_unit *u; explicit UnitSink(agx_unit *x) : u(x) {}
    void *take(int which, size_t bytes) override {
        switch (which) {
        case SA_HITS:  alloc_file_order(u->s_hits, bytes / sizeof(agx_whit) + 1); return u->s_hits.p;
        case SA_SIDES: u->s_sides.alloc(bytes / sizeof(agx_wside) + 1); return u->s_sides.p;
        case SA_RUNS:  u->s_runs.alloc(bytes / sizeof(agx_wrun) + 1); return u->s_runs.p;
        case SA_CODES: alloc_file_order(u->s_codes, bytes); return u->s_codes.p;
        case SA_JUMP:  u->s_jump.alloc(bytes / 4 + 1); return u->s_jump.p;
        default:       u->s_other.alloc(bytes / 8 + 1); return u->s_other.p;
        }
    }
};
void adopt_pairs(agx_unit *u, StagedPairs &S) {      // the staged read alignments' counts and the host-side row table
    u->nh = S.nh; u->n_runs = S.n_runs; u->n_sides = S.n_sides; u->n_jump = S.n_jump; u->n_codes = S.n_codes; u->n_other = S.n_other; u->stride = S.stride; u->maxlen = S.maxlen;
    u->pairs_in_file = S.n_pairs_in_file; u->sam_pairs = S.n_sam_pairs;
    u->row_off.swap(S.row_off); u->row_slot.swap(S.row_slot); u->n_rows = S.n_rows;
}
void stage_cm_layout(agx_unit *u, const agx_u8 *cm_cnt, size_t n_pos, const agx_cmseg *segs, size_t n_segs) {
    CmLayout L; build_cm_layout(cm_cnt, n_pos, segs, n_segs, L);
    u->n_cntruns = L.cnt_runs.size(); u->n_cntchunks = L.cnt_chunks.size(); u->n_segchunks = L.seg_chunks.size();
    u->s_cntruns.alloc(u->n_cntruns + 1); u->s_cntchunks.alloc(u->n_cntchunks + 1); u->s_segchunks.alloc(u->n_segchunks + 1);
    if (u->n_cntruns) memcpy(u->s_cntruns.p, L.cnt_runs.data(), u->n_cntruns * sizeof(agx_cntrun));
    if (u->n_cntchunks) memcpy(u->s_cntchunks.p, L.cnt_chunks.data(), u->n_cntchunks * sizeof(agx_chunk));
    if (u->n_segchunks) memcpy(u->s_segchunks.p, L.seg_chunks.data(), u->n_segchunks * sizeof(agx_chunk));
    build_seg_index(segs, u->n_seg0, n_pos, L.seg_index);
    u->n_segindex = L.seg_index.size(); u->s_segindex.alloc(u->n_segindex + 1); memcpy(u->s_segindex.p, L.seg_index.data(), u->n_segindex * 4);
}
// the unit sequence (+ appended positions) for the upload: 2 bits per base and the stretches of other bytes, or — a soft-masked sequence — the bytes themselves
void stage_reference(agx_unit *u, const char *ref, size_t n_pos, unsigned threads) {
    std::vector<agx_refx> others;
    u->s_ref.alloc((n_pos + 3) / 4 + 32);
    u->ref_packed = getenv("AGX_REF_RAW") == nullptr && pack_reference(ref, n_pos, std::min(threads, 8u), u->s_ref.p, others);
    if (u->ref_packed) { u->n_refx = others.size(); u->s_refx.alloc(u->n_refx + 1); if (u->n_refx) memcpy(u->s_refx.p, others.data(), u->n_refx * sizeof(agx_refx)); return; }
    u->n_refx = 0; u->s_ref.alloc(n_pos + 16);
    const unsigned T = std::max(1u, std::min(threads, 8u));
    on_threads(T, [&](unsigned t) { const size_t lo = n_pos * t / T, hi = n_pos * (t + 1) / T; memcpy(u->s_ref.p + lo, ref + lo, hi - lo); });
}
// A one-shot unit's download lands in the pinned memory of its staged inputs, which are dead once they are in HBM (do_download).  Since r03 those are packed
// (wire formats) and can be smaller than the walk graph: what is missing is pinned here, by estimate (walk ids ~ 1.06 x positions, special
// ids ~ 8 % of them), while the unit is staged — inside T_core mapping and registering it cost 1 ms per unit.
void reserve_landing(agx_unit *u) {
    if (!(u->prm.flags & AGX_FLAG_ONE_SHOT)) { u->s_landing.release(); return; }
    const size_t n_pos = u->V.n_pos, ni = n_pos + n_pos / 16 + 4096, ns = ni / 10 + 4096;
    const size_t need = 2 * (ni + 512) + ns * (sizeof(agx_walknode) + sizeof(agx_hop)) + ni / 4 + (4u << 20);
    const size_t have = u->s_codes.block_bytes() + u->s_hits.block_bytes() + u->s_runs.block_bytes() + u->s_sides.block_bytes() + u->s_other.block_bytes() + u->s_codes_t.block_bytes() + u->s_hits_t.block_bytes();
    // (a buffer must fit one block: count the blocks at 85 %)
    if (need > have * 85 / 100) u->s_landing.alloc(need - have * 85 / 100 + (ni + 512)); else u->s_landing.release();
}
// The read rows for the upload as differences against the reference (agx_core.h; build_row_diffs in agx_load.cpp): AGX_ROW_DIFF=1 asks for it.  It is NOT the default:
// it takes 36-44 % off a unit's upload (66 -> 36 bytes per pair at 2x150) but no measured job is bound by its uploads — cfg3 36.4 against 36.5 ms, the 24-unit human
// shapes 50.0 against 49.4 ms (1/16) and 187 against 183 ms (1/4) per job — while making the form costs the loader 30-60 ns per row (T_unit + 0.1 s at cfg3).  For hosts
// whose ranks share a link or memory bandwidth.  Needs the packed reference.  The 2-bit rows stay where they are either way: the walk of a unit handed over staged reads
// k-mer tails out of them, and a one-shot unit's download lands in their memory.
void stage_rows(agx_unit *u, unsigned threads) {
    u->rows_diffed = false; u->n_units = u->n_rowcnt = u->n_blockoff = u->n_blockfirst = u->n_anchor = u->n_rows_explicit = 0;
    const char *on = getenv("AGX_ROW_DIFF");
    if (want_tiled()) return;      // (r06: the tile-ordered forms have their own: stage_rows_tiled)
    if (!on || atoi(on) == 0 || !u->ref_packed || u->n_rows == 0 || u->nh == 0) return;
    const double t0 = now_ms();
    RowDiffs D;
    if (!build_row_diffs(u->s_hits.p, u->nh, u->s_sides.p, u->n_sides, u->s_runs.p, u->n_runs, u->s_codes.p, u->n_rows, u->stride, (const agx_u32 *)u->s_ref.p, u->V.n_pos ? u->V.n_pos : u->T.ref.size(), threads, D)) return;
    if (D.n_units * 2 + D.cnt.size() + (D.block_off.size() + D.block_first.size() + D.anchor_bits.size()) * 4 >= u->n_codes) return;      // nothing gained (reads that do not resemble the reference)
    u->n_units = D.n_units; u->n_rowcnt = D.cnt.size(); u->n_blockoff = D.block_off.size(); u->n_blockfirst = D.block_first.size(); u->n_anchor = D.anchor_bits.size(); u->n_rows_explicit = D.n_explicit;
    u->s_units.alloc(u->n_units + 2); u->s_rowcnt.alloc(u->n_rowcnt); u->s_blockoff.alloc(u->n_blockoff); u->s_blockfirst.alloc(u->n_blockfirst); u->s_anchor.alloc(u->n_anchor);
    const unsigned T = std::max(1u, std::min(threads, 8u));
    on_threads(T, [&](unsigned t) { const size_t lo = u->n_units * t / T, hi = u->n_units * (t + 1) / T; if (hi > lo) memcpy(u->s_units.p + lo, D.units.data() + lo, (hi - lo) * 2); });
    memcpy(u->s_rowcnt.p, D.cnt.data(), u->n_rowcnt); memcpy(u->s_blockoff.p, D.block_off.data(), u->n_blockoff * 4); memcpy(u->s_blockfirst.p, D.block_first.data(), u->n_blockfirst * 4); memcpy(u->s_anchor.p, D.anchor_bits.data(), u->n_anchor * 4);
    u->rows_diffed = true;
    if (getenv("AGX_LOAD_TIMING")) fprintf(stderr, "[agx load] read rows against the reference: %.1f ms, %zu rows (%zu as they are), %.1f -> %.1f bytes per row\n", now_ms() - t0, (size_t)u->n_rows, u->n_rows_explicit,
                                          (double)u->n_codes / u->n_rows, (double)(u->n_units * 2 + u->n_rowcnt + (u->n_blockoff + u->n_blockfirst + u->n_anchor) * 4) / u->n_rows);
}
// The hits in the order of the tile their first arrival falls in (agx_core.h "hits in tile order"; order_hits in agx_load.cpp): 4 bytes per hit + 4 per tile more to upload.
void stage_order(agx_unit *u, unsigned threads) {
    const double t0 = now_ms();
    const size_t n_pos = u->V.n_pos ? u->V.n_pos : u->T.ref.size(), n_tiles = (n_pos + AGX_TILE - 1) / AGX_TILE;
    u->lookback = agx_tile_lookback(u->maxlen, u->prm.k);
    u->s_perm.alloc(u->nh + 1); u->s_tfirst.alloc(n_tiles + 2); u->s_jump_at.alloc(u->n_jump + 1);
    order_hits(u->s_hits.p, u->nh, u->s_sides.p, u->s_runs.p, u->s_jump.p, u->n_jump, n_pos, threads, u->s_perm.p, u->s_tfirst.p, u->s_jump_at.p);
    if (getenv("AGX_LOAD_TIMING")) fprintf(stderr, "[agx load] hits in tile order: %.1f ms (%zu hits, %zu tiles, window of %u tiles)\n", now_ms() - t0, u->nh, n_tiles, u->lookback);
}
// r06: what the device is sent of the read alignments, in the order of the hits' first tiles (stage_order's permutation applied on the host).
//   * the wire records: s_hits_t[i] = hit perm[i], its `row` field carrying the hit's NUMBER (the key the tile lists are ranked by) — the 4 bytes per hit of the permutation no
//     longer cross PCIe, the device reads the records in order instead of gathering them — and AGX_WF_DUP what the rule of AG:1650-1655 says about it (it needs the hit's FILE
//     neighbours, which are not its neighbours here);
//   * the read rows: row i = the left-mate row of the i-th hit (a pair with a second hit sends its row twice: 5 % of the rows), so the rows that a WINDOW of tiles reads are one
//     contiguous piece of the upload, in front of which lie the rows of every earlier tile: the unit's first build sweeps window w when its piece has landed, beside the
//     upload of the pieces behind it (do_upload / do_build) — chr1 of configs[4]: 22 ms of node sweep that used to wait for the last read row;
//   * the list of the bases that are not A, C, G, T, re-indexed to those rows.
// The k-mer string references of the nodes then name places in the tile order: slot_row maps them back for the walk (UnitView::slot_row).
// Not when the rows cross as differences from the reference (their codec numbers the rows by their anchors' file order): AGX_ROW_DIFF units keep r05's forms.
void stage_tiled(agx_unit *u, unsigned threads) {
    u->tiled = false; u->n_other_t = 0; u->slot_row.clear(); u->n_win = 1;
    const size_t nh = u->nh, s4 = u->stride / 4;
    // (rows that were to cross as differences and do not — nothing gained — cross tile-ordered)
    if (u->rows_diffed || getenv("AGX_NO_TILED_UPLOAD") || nh == 0 || u->n_rows == 0) { u->s_hits_t.release(); u->s_codes_t.release(); u->s_other_t.release(); return; }
    const double t0 = now_ms();
    u->s_hits_t.alloc(nh + 1); u->s_codes_t.alloc(nh * s4 + 16); u->slot_row.resize(nh);
    const agx_whit *wh = u->s_hits.p; const agx_wside *sd = u->s_sides.p; const agx_wrun *wr = u->s_runs.p; const agx_u32 *perm = u->s_perm.p;
    // rows that carry a listed base (few): one bit per row
    const size_t n_rows = u->n_rows; std::vector<agx_u8> has_other((n_rows + 7) / 8, 0);
    for (size_t j = 0; j < u->n_other; j++) { const size_t r = (size_t)(u->s_other.p[j] / u->stride); if (r < n_rows) has_other[r >> 3] |= (agx_u8)(1u << (r & 7)); }
    auto idx0 = [&](const agx_whit &w) -> agx_u32 {      // positionSets[hit][0] of mate1 (agx_idx0_pos on the wire forms)
        if (!(w.flags & AGX_WF_RUNS1)) return w.a;
        const agx_wside &g = sd[w.a]; const agx_wrun &r = wr[g.runs1];
        return r.q == 0 ? r.t : AGX_NONE;
    };
    const unsigned T = std::max(1u, std::min(threads, 16u));
    std::vector<std::vector<unsigned long long>> others(T);
    on_threads(T, [&](unsigned t) {
        // (two dependent random reads per hit — its record, then its row: asked for a few hits ahead, or every hit costs two cache misses in a row: 0.16 s of cfg3's load)
        const size_t AHEAD = 48, NEAR = 24;
        for (size_t i = nh * t / T, hi = nh * (t + 1) / T; i < hi; i++) {
            if (i + AHEAD < hi) __builtin_prefetch(wh + perm[i + AHEAD]);
            if (i + NEAR < hi) { const char *c = (const char *)(u->s_codes.p + (size_t)wh[perm[i + NEAR]].row * s4); __builtin_prefetch(c); if (s4 > 40) __builtin_prefetch(c + 64); }
            const agx_u32 h = perm[i]; agx_whit w = wh[h];
            const agx_u32 row = w.row;
            u->slot_row[i] = row;
            memcpy(u->s_codes_t.p + i * s4, u->s_codes.p + (size_t)row * s4, s4);
            bool dup = false;
            if (w.back) { const agx_u32 me = idx0(w); for (agx_u32 e = 1; e <= w.back && !dup; e++) dup = agx_absdiff(me, idx0(wh[h - e])) < (int)w.len; }      // agx_hit_dup_by
            w.row = h; if (dup) w.flags |= (agx_u8)AGX_WF_DUP;
            u->s_hits_t.p[i] = w;
            if (has_other[row >> 3] & (1u << (row & 7))) {
                const unsigned long long lo = (unsigned long long)row * u->stride, *b = u->s_other.p, *e = b + u->n_other;
                for (const unsigned long long *at = std::lower_bound(b, e, lo); at != e && *at < lo + u->stride; ++at) others[t].push_back((unsigned long long)i * u->stride + (*at - lo));
            }
        }
    });
    size_t no = 0; for (auto &v : others) no += v.size();
    u->s_other_t.alloc(no + 1); u->n_other_t = no;
    { size_t at = 0; for (auto &v : others) { if (!v.empty()) memcpy(u->s_other_t.p + at, v.data(), v.size() * 8); at += v.size(); } }
    u->tiled = true;
    // windows of the first build's sweep: about 8 M positions each, eight at most (AGX_UPLOAD_WINDOWS forces a number: tests), cut at tiles
    const size_t n_pos = u->V.n_pos ? u->V.n_pos : u->T.ref.size(), n_tiles = (n_pos + AGX_TILE - 1) / AGX_TILE;
    size_t W = getenv("AGX_UPLOAD_WINDOWS") ? (size_t)atoi(getenv("AGX_UPLOAD_WINDOWS")) : n_pos / 8000000u;
    W = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(W, 8), n_tiles));
    u->n_win = (agx_u32)W;
    for (size_t w = 0; w <= W; w++) u->win_tile[w] = (agx_u32)(n_tiles * w / W);
    if (getenv("AGX_LOAD_TIMING")) fprintf(stderr, "[agx load] tile-ordered upload forms: %.1f ms (%zu hits, %zu listed bases, %u windows)\n", now_ms() - t0, nh, no, u->n_win);
}

// r06: the read rows of the tile-ordered forms as their differences from the reference (agx_core.h "read rows relative to the reference"): row i is hit i's left mate, so every
// row's anchor is its own hit — no anchor search, no order condition — and the encoder and the device's decoder are r04's (build_row_diffs with rows_are_hits; agx_k_expand_rows
// over anchor bits that are all ones).  The stream of differences is in row order = tile order: a window's rows are one piece of it (do_upload), expanded when it has landed.
// WHEN ("by a rule in code ... or delete the lines"): measured on the round's last tree (profiles/r06_l_*), the rule is NEVER by itself.  The form takes 35-38 %
// off a unit's upload (cfg3 1.13 -> 0.74 GB per job, rank 0 of 8 of configs[4] 3.76 -> 2.34 GB) and costs the unit's first build the slower expansion (agx_k_expand_rows: a
// lane per row decodes its row into LDS; ~10 ms for chr1's 33 M rows of 150 bases against 1 ms of agx_k_expand_codes) and the loader 30-60 ns per row (t_unit_s 0.42 -> 0.62 s at
// cfg3).  Since the windows, a unit's kernels run beside its upload and the two are about as long: with a third of the bytes gone the kernels bind, and they are the longer for
// it — chr1 in the emulated rank is built at 66.9 ms instead of 61.9, the rank done at 132.6 instead of 128.7 ms; cfg3's builds end 0.6 ms later each.  It stays what it was
// in r04/r05: an option for hosts whose link is scarcer than this one's (AGX_ROW_DIFF=1; =0 or unset: the 2-bit rows), parity-tested both ways in both upload forms.
void stage_rows_tiled(agx_unit *u, unsigned threads) {
    if (!u->tiled || !u->ref_packed || u->stride > AGX_ROW_MAXSTRIDE) return;
    const char *env = getenv("AGX_ROW_DIFF");
    const size_t n_pos = u->V.n_pos ? u->V.n_pos : u->T.ref.size();
    if (!env || atoi(env) == 0) return;
    const double t0 = now_ms();
    RowDiffs D;
    if (!build_row_diffs(u->s_hits_t.p, u->nh, u->s_sides.p, u->n_sides, u->s_runs.p, u->n_runs, u->s_codes_t.p, u->nh, u->stride, (const agx_u32 *)u->s_ref.p, n_pos, threads, D, true)) return;
    if (D.n_units * 2 + D.cnt.size() + (D.block_off.size() + D.block_first.size() + D.anchor_bits.size()) * 4 >= u->nh * (u->stride / 4)) return;      // nothing gained (reads that do not resemble the reference)
    u->n_units = D.n_units; u->n_rowcnt = D.cnt.size(); u->n_blockoff = D.block_off.size(); u->n_blockfirst = D.block_first.size(); u->n_anchor = D.anchor_bits.size(); u->n_rows_explicit = D.n_explicit;
    u->s_units.alloc(u->n_units + 2); u->s_rowcnt.alloc(u->n_rowcnt); u->s_blockoff.alloc(u->n_blockoff); u->s_blockfirst.alloc(u->n_blockfirst); u->s_anchor.alloc(u->n_anchor);
    const unsigned T = std::max(1u, std::min(threads, 8u));
    on_threads(T, [&](unsigned t) { const size_t lo = u->n_units * t / T, hi = u->n_units * (t + 1) / T; if (hi > lo) memcpy(u->s_units.p + lo, D.units.data() + lo, (hi - lo) * 2); });
    memcpy(u->s_rowcnt.p, D.cnt.data(), u->n_rowcnt); memcpy(u->s_blockoff.p, D.block_off.data(), u->n_blockoff * 4); memcpy(u->s_blockfirst.p, D.block_first.data(), u->n_blockfirst * 4); memcpy(u->s_anchor.p, D.anchor_bits.data(), u->n_anchor * 4);
    u->rows_diffed = true;
    if (getenv("AGX_LOAD_TIMING")) fprintf(stderr, "[agx load] tile-ordered rows against the reference: %.1f ms, %zu rows (%zu as they are), %.1f -> %.1f bytes per row\n", now_ms() - t0, u->nh, u->n_rows_explicit,
                                          (double)(u->stride / 4), (double)(u->n_units * 2 + u->n_rowcnt + (u->n_blockoff + u->n_blockfirst + u->n_anchor) * 4) / u->nh);
}

void stage_inputs(agx_unit *u) {
    if (!u->have_ref || !u->have_threads) throw Error{E_ARG, "reference and contig threads must be set before upload"};
    // A one-shot unit's download lands in its staged buffers.  The general loader's pairs can be staged again from P; the fast loader wrote the hits, runs and
    // codes straight into those buffers and kept nothing else: after the download there is nothing to stage from, and clearing `consumed` would send the walk
    // graph's bytes to the device as wire records.
    if (u->consumed && u->pairs_staged) throw Error{E_ARG, "one-shot unit: its staged read alignments were overwritten by the download; load the unit again"};
    const double t0 = now_ms();
    HIP_OK(hipSetDevice(u->prm.device));            // (registering host memory needs a current device)
    u->cache_map.reset(); u->reads_map.reset();
    const size_t n_pos = u->T.ref.size();
    if (n_pos == 0 || n_pos >= 0xFFFFFF00ull) throw Error{E_ARG, "unit sequence is empty or too long"};
    if (u->T.cm_cnt.size() != n_pos) throw Error{E_ARG, "conti-mer chains were not built"};
    { size_t el = 0; for (const agx_cmseg &g : u->T.segs) el += g.len; if (el != u->T.n_cm) throw Error{E_ARG, "conti-mer runs were not built"}; }
    u->n_seg0 = u->T.n_seg0; u->n_cm = u->T.n_cm;
    u->n_segs = u->T.segs.size(); u->s_segs.alloc(u->n_segs + 1);
    if (u->n_segs) memcpy(u->s_segs.p, u->T.segs.data(), u->n_segs * sizeof(agx_cmseg));
    std::vector<agx_u32> ce(u->T.chain_end_pos); std::sort(ce.begin(), ce.end()); ce.erase(std::unique(ce.begin(), ce.end()), ce.end());      // positions where a conti-mer chain
    u->n_chain_end = (agx_u32)ce.size(); u->s_chain_end.alloc(ce.size() + 1);                                                                 // ends: their main walk ids are special
    if (!ce.empty()) memcpy(u->s_chain_end.p, ce.data(), ce.size() * 4);
    const unsigned threads = loader_threads(u->P.bases.size() + n_pos + u->nh * 64);
    stage_reference(u, u->T.ref.data(), n_pos, threads);
    stage_cm_layout(u, u->T.cm_cnt.data(), n_pos, u->T.segs.data(), u->n_segs);
    if (!u->pairs_staged) {
        UnitSink sink(u); StagedPairs S;
        u->n_slots = u->P.n_slots;
        stage_pairs(u->P, u->prm.k, threads, sink, S);
        adopt_pairs(u, S);
    }
    UnitView V = view_of(u->T, u->P);
    if (u->pairs_staged && u->pairs_file.fv) {      // staged pairs from their file: no read bases but the 2-bit rows
        const agx::PairsFile &F = u->pairs_file;
        V.bases = nullptr; V.row_off = nullptr; V.stride = u->stride; V.codes2 = (const agx_u8 *)F.sec(pairsfile::S_CODES);
        V.other_idx = (const unsigned long long *)F.sec(pairsfile::S_OTHER); V.other_byte = (const agx_u8 *)F.sec(pairsfile::S_OTHERB); V.n_other = (size_t)F.H.n_other;
    } else if (u->pairs_staged) { V.bases = u->reads_keep ? u->reads_keep->fv.p : nullptr; V.stride = u->stride; V.row_off = u->row_off.data(); }
    u->V = V;
    stage_rows(u, threads);
    stage_order(u, threads);
    stage_tiled(u, threads);
    stage_rows_tiled(u, threads);
    u->V.slot_row = u->tiled ? u->slot_row.data() : nullptr;
    reserve_landing(u);
    u->staged = true; u->consumed = false; u->uploaded = false; u->built = false; u->downloaded = false;
    u->stats.ms_stage = now_ms() - t0;
}

// ---- unit cache file ----------------------------------------------------------------------------------------------------------------
// tmp/_agx_unit.<u>.bin: a unit's STAGED form, written once (AlignGraph_amd does it when it distributes the alignments, AG:3545-3579;
// agx_unit_cache_build) so that a unit loop that finds it neither reads nor parses text: the arrays the upload wants are read straight into
// pinned memory, what only the host walk looks at (conti-mer counts, chain suffixes, read bases for the k-mer tails of written records) is
// mapped and paged in where it is touched.  Valid for one BATCH size, one k and for exactly the five text files it was made from (size and
// modification time of each are in the header): anything else and the loader falls back to the text.
namespace cache {
enum { S_HITS = 0, S_RUNS, S_CODES, S_SEGS, S_CHAIN_END, S_ROWS, S_REF, S_CM_CNT, S_CHAIN_STR, S_INITIAL, S_BASES, S_OTHER, S_SIDES, S_JUMP, S_OTHERB, S_N };
struct Header {
    char magic[8]; agx_u32 version, batch; unsigned long long stamp[6][2];      // (size, modification time) of the five text files and of tmp/_agx_pairs.<u>.bin
    unsigned long long n_pos, n_ref, nh, n_runs, n_cm, n_segs, n_seg0, n_rows, n_chain_end, n_codes, pairs_in_file, sam_pairs;
    agx_u32 stride, maxlen, n_slots, k;          // k: the staged hits name their left mate, which depends on it (agx_hit_left_is_mate2)
    agx_u32 rows_in_reads;                       // 1: S_ROWS holds 64-bit offsets into tmp/_reads.fa (whose size and time are part of the stamp), S_BASES is empty; 0: S_ROWS holds the
                                                 // read slot of every row and S_BASES the slots' bases (units that the general loader parsed); 2: both empty — the k-mer tails come out of
                                                 // S_CODES (2-bit rows) and S_OTHER / S_OTHERB (units whose alignments were handed over staged: UnitView::codes2)
    agx_u32 slot_stride;                         // bases per read slot in S_BASES (rows_in_reads == 0)
    unsigned long long n_sides, n_jump;
    agx_u32 sizes[5];                            // sizeof agx_whit, agx_wrun, agx_cmseg, Header, agx_wside: a file written by another layout is not this one
    unsigned long long off[S_N], len[S_N];
};
const char MAGIC[8] = {'A', 'G', 'X', 'U', 'N', 'I', 'T', '7'};
void stamps(const std::string &d, int unit, unsigned long long st[6][2]) {
    const std::string s = std::to_string(unit);
    const std::string f[6] = {d + "/_genome." + s + ".fa", d + "/_contigs.fa", d + "/_contigs_genome." + s + ".psl", d + "/_reads.fa", d + "/_reads_genome." + s + ".bowtie", pairsfile::path_of(d, unit)};
    for (int i = 0; i < 6; i++) { struct stat sb; if (stat(f[i].c_str(), &sb) != 0) { st[i][0] = st[i][1] = ~0ull; continue; }
        st[i][0] = (unsigned long long)sb.st_size; st[i][1] = (unsigned long long)sb.st_mtim.tv_sec * 1000000000ull + (unsigned long long)sb.st_mtim.tv_nsec; }
}
std::string path_of(const std::string &d, int unit) { return d + "/_agx_unit." + std::to_string(unit) + ".bin"; }
}  // namespace cache

void save_cache(agx_unit *u, const std::string &dir, int unit) {
    using namespace cache;
    if (!u->staged || u->cache_map.p) throw Error{E_ARG, "nothing staged from text"};
    Header H; memset(&H, 0, sizeof H); memcpy(H.magic, MAGIC, 8); H.version = 2; H.batch = u->prm.batch;
    stamps(dir, unit, H.stamp);
    const bool from_codes = u->pairs_staged && u->pairs_file.fv, in_reads = u->pairs_staged && !from_codes;
    H.n_pos = u->V.n_pos; H.n_ref = u->V.n_ref; H.nh = u->nh; H.n_runs = u->n_runs; H.n_cm = u->n_cm; H.n_segs = u->n_segs; H.n_seg0 = u->n_seg0; H.n_rows = u->n_rows;
    H.n_chain_end = u->n_chain_end; H.n_codes = u->n_codes; H.pairs_in_file = u->pairs_in_file; H.sam_pairs = u->sam_pairs; H.stride = u->stride; H.maxlen = u->maxlen; H.n_slots = u->n_slots; H.k = u->prm.k;
    H.rows_in_reads = from_codes ? 2u : in_reads ? 1u : 0u; H.sizes[0] = sizeof(agx_whit); H.sizes[1] = sizeof(agx_wrun); H.sizes[2] = sizeof(agx_cmseg); H.sizes[3] = sizeof(Header); H.sizes[4] = sizeof(agx_wside);
    H.slot_stride = u->V.stride; H.n_sides = u->n_sides; H.n_jump = u->n_jump;
    const void *ptr[S_N] = {u->s_hits.p, u->s_runs.p, u->s_codes.p, u->s_segs.p, u->s_chain_end.p, from_codes ? nullptr : in_reads ? (const void *)u->row_off.data() : (const void *)u->row_slot.data(), u->V.ref, u->V.cm_cnt,
                            u->V.chain_str, u->V.initial, (in_reads || from_codes) ? nullptr : u->V.bases, u->s_other.p, u->s_sides.p, u->s_jump.p, from_codes ? (const void *)u->V.other_byte : nullptr};
    const unsigned long long len[S_N] = {u->nh * sizeof(agx_whit), u->n_runs * sizeof(agx_wrun), u->n_codes, u->n_segs * sizeof(agx_cmseg), (unsigned long long)u->n_chain_end * 4, from_codes ? 0ull : (unsigned long long)u->n_rows * (in_reads ? 8 : 4),
                                         u->V.n_pos, u->V.n_pos, u->T.chain_str.size(), u->V.n_initial, (in_reads || from_codes) ? 0ull : (unsigned long long)u->n_slots * u->V.stride, (unsigned long long)u->n_other * 8,
                                         u->n_sides * sizeof(agx_wside), (unsigned long long)u->n_jump * 4, from_codes ? (unsigned long long)u->n_other : 0ull};
    unsigned long long at = (sizeof(Header) + 4095) & ~4095ull;
    for (int i = 0; i < S_N; i++) { H.off[i] = at; H.len[i] = len[i]; at = (at + len[i] + 4095) & ~4095ull; }
    const std::string path = path_of(dir, unit), part = path + ".part";
    const int fd = open(part.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) throw Error{E_IO, "CANNOT OPEN FILE! (" + part + ")"};
    bool ok = ftruncate(fd, (off_t)at) == 0 && pwrite(fd, &H, sizeof H, 0) == (ssize_t)sizeof H;
    if (ok) {      // the sections, large pieces on a few threads
        struct Piece { const char *src; unsigned long long off, len; };
        std::vector<Piece> pieces;
        for (int i = 0; i < S_N; i++) for (unsigned long long a = 0; a < len[i]; a += 32ull << 20) pieces.push_back(Piece{(const char *)ptr[i] + a, H.off[i] + a, std::min<unsigned long long>(32ull << 20, len[i] - a)});
        const unsigned threads = (unsigned)std::min<size_t>(8, pieces.size() ? pieces.size() : 1);
        std::vector<int> bad(threads, 0);
        on_threads(threads, [&](unsigned t) {
            for (size_t i = t; i < pieces.size(); i += threads) {
                size_t done = 0;
                while (done < pieces[i].len) { const ssize_t w = pwrite(fd, pieces[i].src + done, pieces[i].len - done, (off_t)(pieces[i].off + done)); if (w <= 0) { bad[t] = 1; return; } done += (size_t)w; }
            }
        });
        for (int b : bad) ok = ok && !b;
    }
    ok = (close(fd) == 0) && ok;
    if (!ok || rename(part.c_str(), path.c_str()) != 0) { (void)remove(part.c_str()); throw Error{E_IO, "cannot write " + path}; }
}

// The staged read alignments in u->s_* (read from a cache file or a staged-pairs file): everything the device and the walk index with must be in range — a file whose lengths
// are intact but whose content is not is a reason to refuse it, not to read out of bounds.
bool staged_pairs_fine(agx_unit *u, unsigned long long n_rows, agx_u32 stride, agx_u32 maxlen, unsigned long long n_runs, unsigned long long n_sides, unsigned threads) {
    bool fine = true;
    struct { unsigned long long n_rows, n_runs, n_sides; agx_u32 stride, maxlen; } H{n_rows, n_runs, n_sides, stride, maxlen};
    {
        const unsigned long long n_bases = H.n_rows * (unsigned long long)H.stride;
        for (size_t i = 0; i < u->n_other && fine; i++) fine = u->s_other.p[i] < n_bases;
        std::vector<int> bad2(threads, 0); std::vector<size_t> jumps(threads, 0);
        on_threads(threads, [&](unsigned t) {
            size_t nj = 0;
            for (size_t i = u->nh * t / threads, hi = u->nh * (t + 1) / threads; i < hi; i++) {
                const agx_whit &w = u->s_hits.p[i];
                if (w.row >= H.n_rows || w.len == 0 || w.len > H.maxlen || w.back > i) { bad2[t] = 1; return; }
                if (!(w.flags & (AGX_WF_RUNS1 | AGX_WF_RUNS2))) continue;                 // five hits in eight: nothing else to look at
                if (((w.flags & AGX_WF_RUNS1) ? w.a : w.b) >= H.n_sides) { bad2[t] = 1; return; }
                const agx_hit h = agx_unpack_hit(w, u->s_sides.p);
                if ((h.nruns1 && (unsigned long long)h.runs1 + h.nruns1 > H.n_runs) || (h.nruns2 && (unsigned long long)h.runs2 + h.nruns2 > H.n_runs)) { bad2[t] = 1; return; }
                nj += ((h.pad[0] & 1u) ? h.nruns2 : h.nruns1) >= 2;
            }
            jumps[t] = nj;
        });
        for (int b : bad2) fine = fine && !b;
        { size_t want = 0; for (size_t v : jumps) want += v; fine = fine && want == u->n_jump; }      // pass J's list names exactly the hits whose left mate has several runs: as many, ...
        for (size_t i = 0; i < u->n_jump && fine; i++) {      // ... ascending, each one of them
            const agx_u32 h = u->s_jump.p[i];
            fine = h < u->nh && (i == 0 || u->s_jump.p[i - 1] < h);
            if (fine) { const agx_whit &w = u->s_hits.p[h]; fine = (w.flags & ((w.flags & AGX_WF_LEFT2) ? AGX_WF_RUNS2 : AGX_WF_RUNS1)) != 0; if (fine) { const agx_hit x = agx_unpack_hit(w, u->s_sides.p); fine = ((x.pad[0] & 1u) ? x.nruns2 : x.nruns1) >= 2; } }
        }
    }
    return fine;
}

// A unit's read alignments out of tmp/_agx_pairs.<u>.bin (agx_host.h: pairsfile): the arrays into the pinned upload buffers, checked like a cache file's; the mapped file stays
// with the unit (the walk takes the k-mer tails of written records from its 2-bit rows).
void load_staged_pairs(agx_unit *u, agx::PairsFile &F) {
    using namespace pairsfile;
    const Header &H = F.H;
    if (H.k != u->prm.k || H.batch != u->prm.batch) throw Error{E_ARG, "tmp/_agx_pairs: staged for another k or another BATCH than this unit's"};
    u->nh = H.nh; u->n_runs = H.n_runs; u->n_sides = H.n_sides; u->n_jump = H.n_jump; u->n_codes = H.n_codes; u->n_other = H.n_other; u->stride = H.stride; u->maxlen = H.maxlen; u->n_rows = H.n_rows;
    u->pairs_in_file = H.pairs_in_file; u->sam_pairs = H.sam_pairs; u->n_slots = 0; u->row_off.clear(); u->row_slot.clear(); u->reads_keep.reset(); u->reads_map.reset();
    alloc_file_order(u->s_hits, u->nh + 1); u->s_sides.alloc(u->n_sides + 1); u->s_jump.alloc(u->n_jump + 1); u->s_runs.alloc(u->n_runs + 1); alloc_file_order(u->s_codes, u->n_codes + 16); u->s_other.alloc(u->n_other + 1);
    struct Piece { void *dst; const char *src; size_t len; };
    std::vector<Piece> pieces;
    auto cut = [&](void *dst, int sec) { for (unsigned long long a = 0; a < H.len[sec]; a += 32ull << 20) pieces.push_back(Piece{(char *)dst + a, F.sec(sec) + a, (size_t)std::min<unsigned long long>(32ull << 20, H.len[sec] - a)}); };
    cut(u->s_hits.p, S_HITS); cut(u->s_sides.p, S_SIDES); cut(u->s_jump.p, S_JUMP); cut(u->s_runs.p, S_RUNS); cut(u->s_codes.p, S_CODES); cut(u->s_other.p, S_OTHER);
    const unsigned threads = (unsigned)std::min<size_t>(std::min<unsigned>(8u, std::max(1u, usable_cpus())), pieces.size() ? pieces.size() : 1);
    on_threads(threads, [&](unsigned t) { for (size_t i = t; i < pieces.size(); i += threads) memcpy(pieces[i].dst, pieces[i].src, pieces[i].len); });
    bool fine = staged_pairs_fine(u, H.n_rows, H.stride, H.maxlen, H.n_runs, H.n_sides, threads);
    for (size_t i = 1; i < u->n_other && fine; i++) fine = u->s_other.p[i - 1] < u->s_other.p[i];      // (the walk bisects the list)
    if (!fine) throw Error{E_FORMAT, "tmp/_agx_pairs: the staged arrays are inconsistent"};
    u->pairs_staged = true; u->pairs_file = std::move(F);
}

// true: the unit is staged from the cache file.  false: no usable cache (missing, another batch size, older than its sources, damaged).
bool load_cache(agx_unit *u, const std::string &dir, int unit) {
    using namespace cache;
    if (getenv("AGX_NO_CACHE")) return false;
    const std::string path = path_of(dir, unit);
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    Header H; struct stat sb;
    if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof H || pread(fd, &H, sizeof H, 0) != (ssize_t)sizeof H) return false;
    unsigned long long st[6][2]; stamps(dir, unit, st);
    if (memcmp(H.magic, MAGIC, 8) != 0 || H.version != 2 || H.batch != u->prm.batch || H.k != u->prm.k || memcmp(st, H.stamp, sizeof st) != 0) return false;
    if (H.sizes[0] != sizeof(agx_whit) || H.sizes[1] != sizeof(agx_wrun) || H.sizes[2] != sizeof(agx_cmseg) || H.sizes[3] != sizeof(Header) || H.sizes[4] != sizeof(agx_wside)) return false;
    for (int i = 0; i < S_N; i++) if (H.off[i] > (unsigned long long)sb.st_size || H.len[i] > (unsigned long long)sb.st_size - H.off[i]) return false;
    const bool in_reads = H.rows_in_reads == 1, from_codes = H.rows_in_reads == 2;
    if (H.rows_in_reads > 2) return false;
    if (H.n_pos == 0 || H.n_pos >= 0xFFFFFF00ull || H.n_ref > H.n_pos || H.len[S_REF] != H.n_pos || H.len[S_CM_CNT] != H.n_pos || H.len[S_HITS] != H.nh * sizeof(agx_whit) || H.len[S_RUNS] != H.n_runs * sizeof(agx_wrun) ||
        H.len[S_SIDES] != H.n_sides * sizeof(agx_wside) || H.n_sides > H.nh || H.len[S_JUMP] != H.n_jump * 4 || H.n_jump > H.nh ||
        H.len[S_CODES] != H.n_codes || H.len[S_SEGS] != H.n_segs * sizeof(agx_cmseg) || H.n_seg0 > H.n_segs || H.len[S_ROWS] != (from_codes ? 0ull : H.n_rows * (in_reads ? 8 : 4)) || (H.stride & 3u) || H.len[S_OTHERB] != (from_codes ? H.len[S_OTHER] / 8 : 0ull) ||
        H.len[S_BASES] != ((in_reads || from_codes) ? 0ull : (unsigned long long)H.n_slots * H.slot_stride) || H.len[S_CHAIN_END] != H.n_chain_end * 4 || H.n_codes != H.n_rows * (H.stride / 4) || (H.len[S_OTHER] & 7u) ||
        H.nh >= 0xFFFFFFFFull || H.n_runs >= 0xFFFFFFFFull || H.n_rows >= 0x7FFFFFFFull || H.maxlen > H.stride || (!in_reads && !from_codes && H.maxlen > H.slot_stride)) return false;
    HIP_OK(hipSetDevice(u->prm.device));
    const double t0 = now_ms();
    std::unique_ptr<FileView> reads_map;
    if (in_reads) { try { reads_map.reset(new FileView(dir + "/_reads.fa")); } catch (const Error &) { return false; } if (reads_map->n != st[3][0]) return false; }
    void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) return false;
    drop_outputs(u);
    u->T = Threads(); u->P = Pairs(); u->cache_map.reset(); u->cache_map.p = m; u->cache_map.n = (size_t)sb.st_size; u->reads_keep.reset(); u->reads_map.reset(); u->pairs_file = agx::PairsFile();
    const char *base = (const char *)m;
    u->nh = H.nh; u->n_runs = H.n_runs; u->n_sides = H.n_sides; u->n_jump = H.n_jump; u->n_cm = H.n_cm; u->n_segs = H.n_segs; u->n_seg0 = (agx_u32)H.n_seg0; u->n_chain_end = (agx_u32)H.n_chain_end; u->n_codes = H.n_codes; u->n_other = H.len[S_OTHER] / 8;
    u->pairs_in_file = H.pairs_in_file; u->sam_pairs = H.sam_pairs; u->stride = H.stride; u->maxlen = H.maxlen; u->n_slots = H.n_slots; u->n_rows = (agx_u32)H.n_rows;
    alloc_file_order(u->s_hits, u->nh + 1); u->s_sides.alloc(u->n_sides + 1); u->s_jump.alloc(u->n_jump + 1); u->s_runs.alloc(u->n_runs + 1); alloc_file_order(u->s_codes, u->n_codes + 16); u->s_other.alloc(u->n_other + 1); u->s_segs.alloc(u->n_segs + 1); u->s_chain_end.alloc((size_t)u->n_chain_end + 1);
    // the staged arrays: read into the pinned buffers, a few threads, large pieces
    struct Piece { void *dst; unsigned long long off, len; };
    std::vector<Piece> pieces;
    auto cut = [&](void *dst, int sec) { for (unsigned long long a = 0; a < H.len[sec]; a += 32ull << 20) pieces.push_back(Piece{(char *)dst + a, H.off[sec] + a, std::min<unsigned long long>(32ull << 20, H.len[sec] - a)}); };
    cut(u->s_hits.p, S_HITS); cut(u->s_sides.p, S_SIDES); cut(u->s_jump.p, S_JUMP); cut(u->s_runs.p, S_RUNS); cut(u->s_codes.p, S_CODES); cut(u->s_other.p, S_OTHER); cut(u->s_segs.p, S_SEGS); cut(u->s_chain_end.p, S_CHAIN_END);
    const unsigned threads = (unsigned)std::min<size_t>(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())), pieces.size() ? pieces.size() : 1);
    std::vector<int> bad(threads, 0);
    on_threads(threads, [&](unsigned t) {
        for (size_t i = t; i < pieces.size(); i += threads) {
            size_t done = 0;
            while (done < pieces[i].len) { const ssize_t r = pread(fd, (char *)pieces[i].dst + done, pieces[i].len - done, (off_t)(pieces[i].off + done)); if (r <= 0) { bad[t] = 1; return; } done += (size_t)r; }
        }
    });
    bool fine = true;
    for (int b : bad) fine = fine && !b;
    // what the device and the walk index with must be in range: a file whose lengths are intact but whose content is not is a reason to parse the text again, not to read out of bounds
    if (fine) {
        fine = staged_pairs_fine(u, H.n_rows, H.stride, H.maxlen, H.n_runs, H.n_sides, threads);
        unsigned long long el = 0;
        for (size_t i = 0; i < u->n_segs && fine; i++) { const agx_cmseg &g = u->s_segs.p[i]; fine = (unsigned long long)g.pos0 + g.len <= H.n_pos && g.hop_end < H.n_pos && (unsigned long long)g.hop_str0 + g.hop_len0 <= H.len[S_CHAIN_STR] + 1 && g.elem0 == el; el += g.len; }
        fine = fine && el == H.n_cm;
        if (fine) {      // the device derives cm_start from the counts and writes every run element at cm[cm_start[x] + rank] (agx_k_cm_fill): the counts must add up to the
            // conti-mers, and every element's rank must lie below its position's count — or a file with intact lengths writes out of bounds in HBM
            const agx_u8 *cc = (const agx_u8 *)(base + H.off[S_CM_CNT]);
            std::vector<unsigned long long> part(threads, 0); std::vector<int> bad3(threads, 0);
            on_threads(threads, [&](unsigned t) {
                unsigned long long sum = 0;
                for (size_t x = (size_t)H.n_pos * t / threads, hi = (size_t)H.n_pos * (t + 1) / threads; x < hi; x++) sum += cc[x];
                part[t] = sum;
                for (size_t i = u->n_segs * t / threads, hi = u->n_segs * (t + 1) / threads; i < hi; i++) { const agx_cmseg &g = u->s_segs.p[i]; if (g.rank > 254) { bad3[t] = 1; return; }
                    for (agx_u32 j = 0; j < g.len; j++) if (cc[(size_t)g.pos0 + j] <= g.rank) { bad3[t] = 1; return; } }
            });
            unsigned long long total = 0; for (unsigned t = 0; t < threads; t++) { total += part[t]; fine = fine && !bad3[t]; }
            fine = fine && total == H.n_cm;
        }
        for (size_t i = 0; i < u->n_chain_end && fine; i++) fine = u->s_chain_end.p[i] < H.n_pos;
        if (fine && in_reads) { const uint64_t *ro = (const uint64_t *)(base + H.off[S_ROWS]); for (size_t r = 0; r < H.n_rows && fine; r++) fine = ro[r] + H.maxlen <= reads_map->n && ro[r] < reads_map->n; }      // (a row is read up to the longest read's length)
        if (fine && !in_reads && !from_codes) { const agx_u32 *rs = (const agx_u32 *)(base + H.off[S_ROWS]); for (size_t r = 0; r < H.n_rows && fine; r++) fine = rs[r] < H.n_slots; }
        for (size_t i = 1; i < u->n_other && fine && from_codes; i++) fine = u->s_other.p[i - 1] < u->s_other.p[i];      // (the walk bisects the list)
    }
    if (!fine) { u->cache_map.reset(); return false; }
    const double tv = now_ms();
    stage_reference(u, base + H.off[S_REF], H.n_pos, threads);
    const double tr = now_ms();
    stage_cm_layout(u, (const agx_u8 *)(base + H.off[S_CM_CNT]), H.n_pos, u->s_segs.p, u->n_segs);
    if (getenv("AGX_LOAD_TIMING")) fprintf(stderr, "[agx load] cache: buffers + read + checks %.1f ms, reference %.1f ms, conti-mer layout %.1f ms\n", tv - t0, tr - tv, now_ms() - tr);
    UnitView V; V.ref = base + H.off[S_REF]; V.n_pos = H.n_pos; V.n_ref = (agx_u32)H.n_ref; V.cm_cnt = (const agx_u8 *)(base + H.off[S_CM_CNT]); V.chain_str = base + H.off[S_CHAIN_STR];
    V.hop = nullptr; V.segs = (const agx_cmseg *)(base + H.off[S_SEGS]); V.n_seg0 = (agx_u32)H.n_seg0; V.stride = (in_reads || from_codes) ? H.stride : H.slot_stride; V.initial = base + H.off[S_INITIAL]; V.n_initial = H.len[S_INITIAL];
    u->row_off.clear(); u->row_slot.clear();
    if (from_codes) { V.bases = nullptr; V.codes2 = (const agx_u8 *)(base + H.off[S_CODES]); V.other_idx = (const unsigned long long *)(base + H.off[S_OTHER]); V.other_byte = (const agx_u8 *)(base + H.off[S_OTHERB]); V.n_other = u->n_other; }
    else if (in_reads) { u->reads_map = std::move(reads_map); V.bases = u->reads_map->p; V.row_off = (const uint64_t *)(base + H.off[S_ROWS]); }
    else { V.bases = base + H.off[S_BASES]; u->row_slot.assign((const agx_u32 *)(base + H.off[S_ROWS]), (const agx_u32 *)(base + H.off[S_ROWS]) + H.n_rows); }
    u->V = V; u->pairs_staged = false;
    stage_rows(u, std::max(threads, std::min(8u, usable_cpus())));
    stage_order(u, std::max(threads, std::min(16u, usable_cpus())));
    stage_tiled(u, std::max(threads, std::min(16u, usable_cpus())));
    stage_rows_tiled(u, std::max(threads, std::min(16u, usable_cpus())));
    u->V.slot_row = u->tiled ? u->slot_row.data() : nullptr;
    reserve_landing(u);
    u->have_ref = u->have_threads = true; u->staged = true; u->consumed = false; u->uploaded = false; u->built = false; u->downloaded = false;
    u->stats.ms_stage = now_ms() - t0; u->stats.ms_parse = 0; u->stats.ms_thread = 0; u->stats.from_cache = 1;
    return true;
}

// ---- capacities ---------------------------------------------------------------------------------------------------------------------
// First guesses, made so that a unit's first build is normally its only one (every capacity is still checked on the device and grown
// from the device-side counters if it proves too small: tests force that with AGX_TEST_SMALL_CAPS).
agx_u32 spill_min(const agx_unit *u) { const size_t n_pos = u->V.n_pos; return (agx_u32)std::min<size_t>(g_tiny ? 64 : n_pos / 8 + 65536, 0x10000000u); }

// Slices of the node pool, one per region, and the spill area behind them.  Without a measurement every region gets the same share of
// `main_cap` ids; after a build in which the pool ran out, `demand` holds what every region asked for (the counters keep counting) and the
// slices are cut to that plus slack.  Returns the ids the layout needs; applies it (queues the copy on `st`) if they fit.
unsigned long long layout_regions(agx_unit *u, const agx_u32 *demand, agx_u32 main_cap, bool apply, hipStream_t st) {
    const agx_u32 R = u->n_regions;
    u->s_region_off.alloc((size_t)R + 1);
    agx_u32 *off = u->s_region_off.p;
    unsigned long long at = 0;
    for (agx_u32 r = 0; r < R; r++) {
        off[r] = (agx_u32)std::min<unsigned long long>(at, 0xFFFFFFFFull);
        at += demand ? (unsigned long long)demand[r] + demand[r] / 8 + 64 : main_cap / R;
    }
    const unsigned long long need = at + spill_min(u);
    if (!apply || need > u->pool_cap) return need;
    off[R] = (agx_u32)at; u->spill_lo = (agx_u32)at;
    HIP_OK(hipMemcpyAsync(u->d_region_off.p, off, ((size_t)R + 1) * 4, hipMemcpyHostToDevice, st));
    return need;
}

void alloc_pool(agx_unit *u, agx_u32 cap) {
    u->pool_cap = cap;
    DevArena &a = u->arena;
    const size_t kcap = (size_t)cap + AGX_SLOW_V;      // slack: the edge pass reads whole AGX_SLOW_V-row batches of keys (agx_edge_slow_ctx)
    // First what the host walk may still ask the device for after the download (agx_walk_record through fetch_records: the records of ids outside the sparse table): the walk id
    // and node of every id, and four node arrays — 40 bytes per node slot + 4 per id, a quarter of the unit's block.  They lie at the FRONT of the block so that everything
    // behind them can go back to the device when the download is done (agx_unit_trim), not when the walk is (r05: on the whole-human job a unit's walk takes longer than its
    // upload and build, and the units waiting for room on the device were waiting for walks).
    const size_t n_pos = u->V.n_pos, ids_cap = n_pos + cap;
    u->d_aid_of.release(); u->d_aid_of.alloc(a, (size_t)cap + 1); u->d_a_nid.release(); u->d_a_nid.alloc(a, ids_cap + 1);
    u->d_off0.release(); u->d_off0.alloc(a, kcap); u->d_sref.release(); u->d_sref.alloc(a, cap); u->d_next.release(); u->d_next.alloc(a, (size_t)cap * AGX_MAXE);
    u->d_fetch.release(); u->d_fetch.alloc(a, 65536);
    for (auto *b : {&u->d_cid, &u->d_coff, &u->d_cid0, &u->d_coff0}) { b->release(); b->alloc(a, kcap); }
    u->d_base.release(); u->d_base.alloc(a, cap); u->d_flags.release(); u->d_flags.alloc(a, cap);
    if (u->prm.flags & AGX_FLAG_KEEP_COUNTS) { u->d_counts.release(); u->d_counts.alloc(a, (size_t)cap * 6); }
    // walk-graph arrays indexed by walk id: side variants <= nodes <= pool_cap
    u->d_a_str.release(); u->d_a_str.alloc(a, ids_cap + 1); u->d_a_meta.release(); u->d_a_meta.alloc(a, ids_cap + 16);
    u->d_a_mark.release(); u->d_a_mark.alloc(a, ids_cap + 2); u->d_side_xpos.release(); u->d_side_xpos.alloc(a, (size_t)cap + 1);
    u->n_words = (agx_u32)(ids_cap / 64 + 1);
    u->d_sp_bits.release(); u->d_sp_bits.alloc(a, (size_t)u->n_words + 1); u->d_sp_cnt.release(); u->d_sp_cnt.alloc(a, (size_t)u->n_words + 1);
    u->d_sp_rank.release(); u->d_sp_rank.alloc(a, (size_t)u->n_words + 2);
    const size_t nb = ((size_t)std::max<size_t>(n_pos, u->n_words) + 1 + 1023) / 1024;
    u->d_scan_tmp.release(); u->d_scan_tmp.alloc(a, 2 * (nb + 1) + 2 * ((nb + 1023) / 1024 + 1) + 16);
    u->scan_desc_n = (std::max<size_t>(u->n_tiles, u->n_words) + 1) / 4096 + 2; u->d_scan_desc.release(); u->d_scan_desc.alloc(a, 3 * u->scan_desc_n);
}
void alloc_lists(agx_unit *u, agx_u32 cap) { u->list_cap = cap; u->d_unsorted.release(); u->d_unsorted.alloc(u->arena, (size_t)cap + 1); u->d_tile_recs.release(); u->d_tile_recs.alloc(u->arena, ((size_t)cap + 4) * 8); }
void alloc_ovf(agx_unit *u, agx_u32 cap) { u->ovf_cap = cap; u->d_ovf.release(); u->d_ovf.alloc(u->arena, cap); u->d_a_ovf.release(); u->d_a_ovf.alloc(u->arena, (size_t)cap + 1); }
void alloc_sparse(agx_unit *u, agx_u32 cap) { u->sp_cap = cap; u->d_sp_node.release(); u->d_sp_node.alloc(u->arena, (size_t)cap + 1); u->d_sp_hop.release(); u->d_sp_hop.alloc(u->arena, (size_t)cap + 2); }

void do_release(agx_unit *u);

// Capacities of a unit's first build and the HBM they add up to (what do_upload reserves as one block; AlignGraph_amd admits a unit to a device by it:
// agx_unit_hbm_needed).  From the staged counts: positions, hits, runs, conti-mers, read rows.
struct Plan { agx_u32 pool_cap, list_cap, ovf_cap, sp_cap; size_t total; };
// first row of window w's piece of a tile-ordered upload (w = n_win: all rows): the place in the tile order of the first hit of the window's first tile, rounded up to 16 rows —
// a piece then begins on a 16-byte boundary of the packed classes and on a multiple of 16 bases (agx_k_expand_codes takes 16 bases per thread); the few rows of the next window's
// hits that ride in this piece only arrive early
inline size_t win_row(const agx_unit *u, agx_u32 w) { const size_t a = u->rows_diffed ? 63 : 15;      // (rows as differences: whole 64-row blocks of the stream)
    return w == 0 ? 0 : w >= u->n_win ? u->nh : std::min<size_t>(u->nh, ((size_t)u->s_tfirst.p[u->win_tile[w]] + a) & ~a); }
inline size_t codes_bytes(const agx_unit *u) { return u->tiled ? u->nh * (u->stride / 4) : u->n_codes; }      // packed read rows as they are uploaded (tile-ordered: one row per hit)
inline size_t others_up(const agx_unit *u) { return u->tiled ? u->n_other_t : u->n_other; }
Plan plan_capacities(const agx_unit *u) {
    const size_t n_pos = u->V.n_pos, nh = u->nh;
    const agx_u32 n_tiles = (agx_u32)((n_pos + AGX_TILE - 1) / AGX_TILE), n_regions = (n_tiles + AGX_REGION_TILES - 1) / AGX_REGION_TILES;
    const size_t n_bases = codes_bytes(u) * 4;
    Plan P;
    const agx_u32 main_cap = (agx_u32)std::min<size_t>(g_tiny ? n_pos / 8 + 64 : n_pos + n_pos / 4 + 4096, 0xE0000000ull);
    P.pool_cap = u->pool_cap ? u->pool_cap : main_cap + spill_min(u);
    // a hit's arrivals span len - k + 1 positions = 1 + (span - 1) / 64 tiles on average
    const double per_hit = 1.0 + (u->maxlen > u->prm.k ? (double)(u->maxlen - u->prm.k) : 0.0) / AGX_TILE;
    P.list_cap = u->list_cap ? u->list_cap : (agx_u32)std::min<double>(g_tiny ? (double)nh / 2 + 16 : (double)nh * per_hit * 1.1 + 4096, 4.0e9);
    P.ovf_cap = u->ovf_cap ? u->ovf_cap : (g_tiny ? 4u : 1u << 16);
    const size_t ids_cap = n_pos + P.pool_cap;
    P.sp_cap = u->sp_cap ? u->sp_cap : (agx_u32)std::min<size_t>(g_tiny ? 32 : ids_cap / 4 + 4096, 0xFFFFFF00ull);      // special ids: 8 % on the bench unit
    // what the takes of do_upload add up to, plus the alignment of ~90 buffers
    const size_t per_pos = 4 + 16 + 1 + 4 + 2 + 1 + 4 + 4, per_tile = 4 * 4 + 4 * 2 + 4 * 2, per_hit_b = sizeof(agx_dhit) + 4 + 4 + 4;      // per hit: derived record, order, last-tile key, (pass J / long list)
    const size_t per_slot = 5 * 4 + 4 + 4 * AGX_MAXE + 1 + 1 + sizeof(agx_sref) + ((u->prm.flags & AGX_FLAG_KEEP_COUNTS) ? 24 : 0) + 4 + 4, per_id = 1 + 1 + 4 + 1 + 3.0 * 8 / 64 + 1;
    const size_t wire = nh * sizeof(agx_whit) + u->n_sides * sizeof(agx_wside) + u->n_runs * sizeof(agx_wrun) + u->n_jump * 4 + (u->ref_packed ? n_pos / 4 + u->n_refx * sizeof(agx_refx) : 0) + 4096;
    const size_t total = wire + n_pos * per_pos + n_tiles * per_tile + nh * per_hit_b + u->n_runs * sizeof(agx_run) + u->n_cm * sizeof(agx_cmkey) + (u->rows_diffed ? u->n_units * 2 + u->n_rowcnt + (u->n_blockoff + u->n_blockfirst + u->n_anchor) * 4 : codes_bytes(u)) + n_bases + others_up(u) * 8 +
                         (size_t)P.pool_cap * per_slot + ids_cap * per_id + (size_t)P.list_cap * 36 + (size_t)P.ovf_cap * 16 + (size_t)P.sp_cap * (sizeof(agx_walknode) + sizeof(agx_hop)) +
                         (size_t)AGX_BIG_WAVES * AGX_NF * AGX_MAXV_BIG * 64 * 4 + (size_t)n_regions * AGX_REGION_PAD * 4 + (64u << 10) * 100;
    P.total = total + total / 64;
    return P;
}

// Everything a unit holds on the device is taken here, from ONE block of HBM sized by the sum; then the inputs are copied (asynchronously,
// from the staged pinned arrays, on the device's upload stream: behind the copies of the units queued before, beside whatever kernels run on
// the device) and the unit's helper thread is handed what can be prepared meanwhile.  Nothing waits on the host; the unit's first build
// waits for ev_uploaded (on the host) and then expands what was copied.
void do_upload(agx_unit *u) {
    if (u->consumed) throw Error{E_ARG, "one-shot unit: hand its inputs over again before another upload"};
    if (!u->staged) stage_inputs(u);
    if (u->arena.used()) do_release(u);              // uploaded before: start over (the unit's blocks go through the cache)
    const double t0 = now_ms();
    HIP_OK(hipSetDevice(u->prm.device));
    const size_t n_pos = u->V.n_pos, nh = u->nh;
    u->arena.device = u->prm.device;
    u->n_tiles = (agx_u32)((n_pos + AGX_TILE - 1) / AGX_TILE);
    u->n_regions = (u->n_tiles + AGX_REGION_TILES - 1) / AGX_REGION_TILES;
    const size_t n_bases = codes_bytes(u) * 4;
    const Plan plan = plan_capacities(u);
    const agx_u32 pool_cap = plan.pool_cap, list_cap = plan.list_cap, ovf_cap = plan.ovf_cap, sp_cap = plan.sp_cap;
    u->arena.reserve(plan.total);                        // one block for all of it
    DevArena &a = u->arena;
    alloc_pool(u, pool_cap);                             // (first: what outlives the download lies at the front of the block, agx_unit_trim)
    u->d_cm_start.alloc(a, n_pos + 2); u->d_cm.alloc(a, u->n_cm + 1); u->d_ref.alloc(a, n_pos + 16); u->d_cm_head.alloc(a, n_pos + 1);
    u->d_segs.alloc(a, u->n_segs + 1); u->d_cntruns.alloc(a, u->n_cntruns + 1); u->d_cntchunks.alloc(a, u->n_cntchunks + 1); u->d_segchunks.alloc(a, u->n_segchunks + 1); u->d_segindex.alloc(a, u->n_segindex + 1);
    u->d_runs.alloc(a, u->n_runs + 1); u->d_vcodes.alloc(a, n_bases + 16); u->d_other.alloc(a, others_up(u) + 1);
    if (u->rows_diffed) { u->d_units.alloc(a, u->n_units + 2); u->d_rowcnt.alloc(a, u->n_rowcnt); u->d_blockoff.alloc(a, u->n_blockoff); u->d_blockfirst.alloc(a, u->n_blockfirst); u->d_anchor.alloc(a, u->n_anchor); }
    else u->d_codes.alloc(a, codes_bytes(u) + 16);
    u->d_whits.alloc(a, nh + 1); u->d_wsides.alloc(a, u->n_sides + 1); u->d_wruns.alloc(a, u->n_runs + 1); u->d_jump.alloc(a, u->n_jump + 1);
    if (u->ref_packed) { u->d_wref.alloc(a, (n_pos + 3) / 4 + 32); u->d_refx.alloc(a, u->n_refx + 1); }
    u->d_dhit.alloc(a, nh + 1);
    u->d_perm.alloc(a, nh + 1); u->d_tfirst.alloc(a, (size_t)u->n_tiles + 2); u->d_ckey.alloc(a, nh + 1); u->d_long.alloc(a, AGX_LONG_MAX);
    u->d_tile_cnt.alloc(a, (size_t)u->n_tiles + 1); u->d_tile_off.alloc(a, (size_t)u->n_tiles + 2); u->d_cursor.alloc(a, (size_t)u->n_tiles + 1);
    u->d_words.alloc(a, W_TOTAL); u->h_words.alloc(W_TOTAL);
    u->d_chain_end.alloc(a, (size_t)u->n_chain_end + 1);
    u->d_node_start.alloc(a, n_pos); u->d_node_cnt.alloc(a, n_pos); u->d_pos_succ.alloc(a, n_pos); u->d_slow_list.alloc(a, n_pos + 64);
    u->d_side_pk.alloc(a, n_pos + 2); u->d_tile_side.alloc(a, (size_t)u->n_tiles + 2); u->d_tile_side_start.alloc(a, (size_t)u->n_tiles + 2);
    u->d_big_list.alloc(a, (size_t)u->n_tiles + 1); u->d_mid_list.alloc(a, (size_t)u->n_tiles + 1);
    u->d_scratch.alloc(a, (size_t)AGX_BIG_WAVES * AGX_NF * AGX_MAXV_BIG * 64);
    u->d_region_off.alloc(a, (size_t)u->n_regions + 1); u->d_pool_cnt.alloc(a, (size_t)u->n_regions * AGX_REGION_PAD);
    alloc_lists(u, list_cap); alloc_ovf(u, ovf_cap); alloc_sparse(u, sp_cap);
    // copies: all of them on the device's upload stream, behind those of the units queued before.  The kernels that expand what was copied
    // (conti-mer tables, vote codes) open the unit's first build instead of following the copies here: a kernel on this stream would wait for
    // CUs while another unit's node sweep holds them all, and every later unit's copies with it.
    DeviceTurn &turn = turn_of(u->prm.device);
    turn.make();
    hipStream_t st = turn.up;
    trace(u, "upload: allocate", t0, n_pos);
    const double tq = now_ms();
    try {
        std::lock_guard<std::mutex> up_turn(turn.up_m);
        u->up_timed = u->ev.all;
        if (u->up_timed) HIP_OK(hipEventRecord(u->ev_up0, st));
        // (AGX_UP_CHUNK_MB: experiment knob — copies cut into pieces of that size)
        static const size_t chunk = getenv("AGX_UP_CHUNK_MB") ? (size_t)atoi(getenv("AGX_UP_CHUNK_MB")) << 20 : 0;
        auto up = [&](void *dst, const void *src, size_t bytes) {
            for (size_t at = 0; at < bytes;) { const size_t m = chunk ? std::min(chunk, bytes - at) : bytes; HIP_OK(hipMemcpyAsync((char *)dst + at, (const char *)src + at, m, hipMemcpyHostToDevice, st)); at += m; }
        };
        up(u->d_segs.p, u->s_segs.p, u->n_segs * sizeof(agx_cmseg)); up(u->d_cntruns.p, u->s_cntruns.p, u->n_cntruns * sizeof(agx_cntrun));
        up(u->d_cntchunks.p, u->s_cntchunks.p, u->n_cntchunks * sizeof(agx_chunk)); up(u->d_segchunks.p, u->s_segchunks.p, u->n_segchunks * sizeof(agx_chunk)); up(u->d_segindex.p, u->s_segindex.p, u->n_segindex * 4);
        up(u->d_whits.p, u->tiled ? u->s_hits_t.p : u->s_hits.p, nh * sizeof(agx_whit)); up(u->d_wsides.p, u->s_sides.p, u->n_sides * sizeof(agx_wside)); up(u->d_wruns.p, u->s_runs.p, u->n_runs * sizeof(agx_wrun)); up(u->d_jump.p, u->s_jump_at.p, u->n_jump * 4);
        if (!u->tiled) up(u->d_perm.p, u->s_perm.p, nh * 4);      // (tile-ordered records carry their hit numbers: agx_k_hit_prep writes the array)
        up(u->d_tfirst.p, u->s_tfirst.p, ((size_t)u->n_tiles + 2) * 4);
        HIP_OK(hipEventRecord(u->ev_hits, st));         // what the front of the build needs (conti-mer tables, hit preparation, binning) is there: it starts while the rest still travels
        if (u->tiled) {      // the small things first, the read rows last and window by window: a window's sweep starts when its rows are in (do_build)
            up(u->d_other.p, u->s_other_t.p, u->n_other_t * 8);
            if (u->ref_packed) { up(u->d_wref.p, u->s_ref.p, (n_pos + 3) / 4); up(u->d_refx.p, u->s_refx.p, u->n_refx * sizeof(agx_refx)); } else up(u->d_ref.p, u->s_ref.p, n_pos);
            up(u->d_chain_end.p, u->s_chain_end.p, (size_t)u->n_chain_end * 4);
            layout_regions(u, nullptr, pool_cap - spill_min(u), true, st);
            const size_t s4 = u->stride / 4;
            if (u->rows_diffed) { up(u->d_blockoff.p, u->s_blockoff.p, u->n_blockoff * 4); up(u->d_blockfirst.p, u->s_blockfirst.p, u->n_blockfirst * 4); up(u->d_anchor.p, u->s_anchor.p, u->n_anchor * 4); }
            for (agx_u32 w = 0; w < u->n_win; w++) {      // rows of the hits whose first tile lies in window w (a tile's list also names hits that begin in earlier tiles: earlier pieces)
                const size_t r_lo = win_row(u, w), r_hi = win_row(u, w + 1);
                if (u->rows_diffed) {                     // the window's count bytes and its piece of the stream of differences (whole 64-row blocks)
                    const size_t b_lo = r_lo / 64, b_hi = (r_hi + 63) / 64, c_hi = std::min<size_t>(u->n_rowcnt, b_hi * 64);
                    up(u->d_rowcnt.p + r_lo, u->s_rowcnt.p + r_lo, c_hi - r_lo);
                    up(u->d_units.p + u->s_blockoff.p[b_lo], u->s_units.p + u->s_blockoff.p[b_lo], ((size_t)u->s_blockoff.p[b_hi] - u->s_blockoff.p[b_lo]) * 2);
                } else up(u->d_codes.p + r_lo * s4, u->s_codes_t.p + r_lo * s4, (r_hi - r_lo) * s4);
                HIP_OK(hipEventRecord(u->ev_rows[w], st));
            }
        } else {
            if (u->rows_diffed) { up(u->d_units.p, u->s_units.p, u->n_units * 2); up(u->d_rowcnt.p, u->s_rowcnt.p, u->n_rowcnt); up(u->d_blockoff.p, u->s_blockoff.p, u->n_blockoff * 4); up(u->d_blockfirst.p, u->s_blockfirst.p, u->n_blockfirst * 4); up(u->d_anchor.p, u->s_anchor.p, u->n_anchor * 4); }
            else up(u->d_codes.p, u->s_codes.p, u->n_codes);
            up(u->d_other.p, u->s_other.p, u->n_other * 8);      // first needed by the sweep
            if (u->ref_packed) { up(u->d_wref.p, u->s_ref.p, (n_pos + 3) / 4); up(u->d_refx.p, u->s_refx.p, u->n_refx * sizeof(agx_refx)); } else up(u->d_ref.p, u->s_ref.p, n_pos);
            up(u->d_chain_end.p, u->s_chain_end.p, (size_t)u->n_chain_end * 4);      // first needed by the walk preparation
            layout_regions(u, nullptr, pool_cap - spill_min(u), true, st);
        }
        HIP_OK(hipEventRecord(u->ev_uploaded, st));
    } catch (...) { (void)hipStreamSynchronize(st); throw; }      // (copies that were queued before the failure must not outlive the unit's HBM block)
    u->expanded = false;
    trace(u, "upload: queue copies", tq, n_pos);
    // While the copies run: the download's pinned buffers, by estimate (walk ids ~ 1.05 x positions, special ids ~ 8 % of them), on a helper
    // thread.  Mapping and registering them costs 5-13 ms for a cold cache.  Between the build and the download that was on the unit's critical
    // path; on the unit's worker right here it kept the NEXT unit's upload from starting (a job hands its units out one at a time); beside the
    // build's kernels, or at the start of do_build, it held up the other units' HIP calls.  do_download joins the helper and re-sizes what is too small.
    auto dl_buffers = [u, n_pos] {
        static std::mutex one_at_a_time;      // five units pinning memory at once take five times as long each, and stall everything else that touches the address space
        std::lock_guard<std::mutex> l(one_at_a_time);
        const double th0 = now_ms();
        try {
            if (hipSetDevice(u->prm.device) != hipSuccess) return;
            const size_t ni = n_pos + n_pos / 8 + 4096, nw = ni / 64 + 1, ns = ni / 8 + 4096;
            u->h_a_str.alloc(ni + 1); u->h_a_meta.alloc(ni + 64); u->h_side_xpos.alloc(n_pos / 8 + 4097);
            u->h_sp_bits.alloc(nw + 1); u->h_sp_rank.alloc(nw + 1); u->h_sp_node.alloc(ns + 1); u->h_sp_hop.alloc(ns + 2); u->h_a_ovf.alloc(64);
        } catch (...) { }                               // do_download allocates what is missing and reports
        trace(u, "helper: download buffers", th0, n_pos);
    };
    if (!(u->prm.flags & AGX_FLAG_ONE_SHOT)) u->helper.submit(UnitHelper::DL, dl_buffers);      // (a one-shot unit's download borrows the staged inputs' memory: nothing to pin)
    start_helper(u);                                     // then the output buffers

    u->uploaded = true; u->built = false; u->downloaded = false;
    if (!u->pending_walk) { u->pending_walk = true; g_walks_pending.fetch_add(1); }
    u->stats.ms_upload = now_ms() - t0;
    u->stats.upload_bytes = u->n_segs * sizeof(agx_cmseg) + u->n_cntruns * sizeof(agx_cntrun) + (u->n_cntchunks + u->n_segchunks) * sizeof(agx_chunk) + (u->ref_packed ? (n_pos + 3) / 4 + u->n_refx * sizeof(agx_refx) : n_pos) + nh * sizeof(agx_whit) + u->n_sides * sizeof(agx_wside) + u->n_runs * sizeof(agx_wrun) + u->n_jump * 4 + nh * 4 + ((size_t)u->n_tiles + 2) * 4 +
                            (size_t)u->n_chain_end * 4 + (u->rows_diffed ? u->n_units * 2 + u->n_rowcnt + (u->n_blockoff + u->n_blockfirst + u->n_anchor) * 4 : codes_bytes(u)) + others_up(u) * 8 + ((size_t)u->n_regions + 1) * 4 - (u->tiled ? nh * 4 : 0);
    u->stats.device_bytes = u->arena.capacity();
    u->stats.rows_by_reference = u->rows_diffed ? (uint32_t)((u->tiled ? u->nh : u->n_rows) - u->n_rows_explicit) : 0u;
}

// The three output buffers of a unit: reserved by estimate (the walk grows what is too small), every page touched, the initial contigs
// copied.  A 30 Mb unit's outputs are 100 MB of fresh memory: faulted in inside the walk and after it they cost 8-10 ms of the unit's
// critical path, and 20 ms on the unit's own worker before the build (five workers faulting at once).  So a helper thread makes them while
// the unit is uploaded and built; agx_unit_finish joins it.
void prepare_outputs(agx_unit *u) {
    if (u->out_ready) return;
    // a few at a time: with huge pages five threads touching fresh memory at once cost each other 20 % (tests/tools/faultbench.cpp), but 24 units
    // taking strict turns made the last small units of a cfg5s job wait 3 ms for buffers that take 1 ms to make
    struct Slots { std::mutex m; std::condition_variable cv; int free = 4; } static slots;
    { std::unique_lock<std::mutex> l(slots.m); slots.cv.wait(l, [] { return slots.free > 0; }); slots.free--; }
    struct Give { ~Give() { { std::lock_guard<std::mutex> l(slots.m); slots.free++; } slots.cv.notify_one(); } } give;
    try {
        const size_t n_pos = u->V.n_pos;
        u->out.pre_extended.n = 0; u->out.extended.n = 0; u->out_initial.n = 0;
        u->out.pre_extended.reserve(n_pos + n_pos / 8 + 8192); u->out.pre_extended.prefault();
        u->out.extended.reserve(n_pos + n_pos / 16 + 4096); u->out.extended.prefault();
        u->out_initial.reserve(u->V.n_initial);                      // (one copy: a 30 Mb unit's initial contigs are 30 MB)
        if (u->V.n_initial) u->out_initial.append(u->V.initial, u->V.n_initial);
        u->out_ready = true;
    } catch (...) { u->out_ready = false; }                              // out of memory: agx_unit_finish tries again on its own thread and reports
}
void join_helper(agx_unit *u) { u->helper.wait(UnitHelper::OUT); }
void join_dl_helper(agx_unit *u) { u->helper.wait(UnitHelper::DL); }
void drop_outputs(agx_unit *u) { join_helper(u); u->out_ready = false; }
void start_helper(agx_unit *u) {
    join_helper(u);
    if (u->out_ready) return;
    u->helper.submit(UnitHelper::OUT, [u] { const double th0 = now_ms(); prepare_outputs(u); trace(u, "helper: output buffers", th0, u->V.n_pos); });      // (no helper: finish prepares the outputs itself)
}

// All kernels of one build are queued back to back with the current buffer capacities; the counters they produce (tile-list
// entries, nodes, overflowed tiles, edge overflow, side variants) are read after ONE synchronisation.  A capacity that proved too
// small is grown and the build repeats — the first guesses (do_upload) are made so that this is rare.
void do_build(agx_unit *u) {
    if (!u->uploaded) do_upload(u);
    HIP_OK(hipSetDevice(u->prm.device));
    const agx_u32 n_pos = (agx_u32)u->V.n_pos, nh = (agx_u32)u->nh;
    hipStream_t st = nullptr;              // the device's build stream, taken with the turn
    u->stats.node_sweep_launches = u->stats.edge_sweep_launches = 0;
    agx_u32 swept_windows = 1; bool swept_timed = false;      // of the last attempt
    for (int attempt = 0;; attempt++) {
        if (attempt > 8) throw Error{E_DEVICE, "build did not converge"};
        const size_t ids_cap = (size_t)n_pos + u->pool_cap;                       // side variants <= nodes <= pool_cap

        // The build streams are shared by all units of the device: nothing is queued on them that could wait long.  The unit's upload is
        // awaited here, on the host, before the turn is taken.
        const double tb0 = now_ms();
        const bool early = !u->expanded && attempt == 0 && !u->ev.all;      // a unit's first build starts on the arrays that arrive first; the read bases are waited for on the device, in front of their first use (not in builds that time their sections: those would time the wait)
        HIP_OK(wait_event(early ? u->ev_hits : u->ev_uploaded));
        trace(u, "build: wait for upload", tb0, n_pos);
        const double tb1 = now_ms();
        DeviceTurn &turn = turn_of(u->prm.device);
        std::unique_lock<std::mutex> my_turn(turn.m);
        // ---- front (its own stream): may run beside the previous build's edge passes and walk preparation, not beside its sweep ----
        st = turn.front;
        if (turn.n) HIP_OK(hipStreamWaitEvent(st, (u->ev.all || turn.prev_exclusive) ? turn.build_done[(turn.n - 1) & 1] : turn.sweep_done[(turn.n - 1) & 1], 0));
        if (u->ev.all) HIP_OK(hipEventRecord(u->ev.first, st));      // (every event record costs the stream a few microseconds: untimed builds record only what orders them)
        if (!u->expanded) {   // the unit's first build: the hits and runs out of their wire forms, then the conti-mer tables from their runs (agx_cmseg: count per position, scan, keys, heads); the read bases follow behind the binning
            agx_launch_expand_runs(u->d_wruns.p, u->d_runs.p, (agx_u32)u->n_runs, st);
            agx_launch_cm_tables(u->d_cntruns.p, u->d_cntchunks.p, (agx_u32)u->n_cntchunks, u->d_segs.p, u->d_segchunks.p, (agx_u32)u->n_segchunks, u->d_cm_start.p, u->d_cm.p, u->d_cm_head.p, n_pos, (agx_u32)u->n_cm, st);
        }
        {   // everything a build counts into or marks, zeroed by one kernel and one fill (every command on the stream costs a few microseconds)
            agx_zero_args Z; memset(&Z, 0, sizeof Z);
            auto seg = [&](int i, agx_u32 *ptr, size_t words) { Z.p[i] = ptr; Z.n[i] = (agx_u32)words; };
            seg(0, u->d_words.p, W_TOTAL); seg(1, u->d_tile_cnt.p, (size_t)u->n_tiles + 1); seg(2, u->d_cursor.p, (size_t)u->n_tiles + 1);
            seg(3, u->d_pool_cnt.p, (size_t)u->n_regions * AGX_REGION_PAD); seg(4, u->d_tile_side.p + u->n_tiles, 1); seg(5, u->d_sp_cnt.p, (size_t)u->n_words + 1);
            seg(6, reinterpret_cast<agx_u32 *>(u->d_scan_desc.p), 6 * u->scan_desc_n); seg(7, reinterpret_cast<agx_u32 *>(u->d_sp_bits.p), 2 * ((size_t)u->n_words + 1));      // (the special-id kernels write and read the words of the live ids only)
            agx_launch_zero(&Z, st);
            HIP_OK(hipMemsetAsync(u->d_a_mark.p, 0, ids_cap + 2, st));
        }
        // ---- hit_prep + tile histogram ----
        u->ev.begin(); u->ev.mark(B_START, st);
        agx_prep_args PA{(const agx_whit *)u->d_whits.p, (const agx_wside *)u->d_wsides.p, u->d_runs.p, u->d_dhit.p, nh, u->prm.k, n_pos, u->d_tile_cnt.p, u->d_words.p + W_ERR,
                         u->d_perm.p, u->d_tfirst.p, u->d_ckey.p, u->lookback, u->d_long.p, u->d_words.p + W_LONGCOUNT, u->tiled ? 1u : 0u, u->d_perm.p};
        agx_launch_hit_prep(&PA, st);
        AGX_CHECKPOINT("hit_prep");
        u->ev.mark(B_PREP, st);
        // ---- tile lists ----
        if (g_scan1) agx_launch_exclusive_scan1(u->d_tile_cnt.p, u->d_tile_off.p, u->n_tiles, u->d_scan_desc.p, st);
        else agx_launch_exclusive_scan(u->d_tile_cnt.p, u->d_tile_off.p, u->n_tiles, u->d_scan_tmp.p, st);
        agx_fill_args FA{u->d_tile_off.p, u->d_tfirst.p, u->d_perm.p, u->d_ckey.p, u->d_dhit.p, u->d_runs.p, u->d_tile_recs.p, u->d_unsorted.p, u->n_tiles, u->list_cap, u->prm.k, u->lookback,
                         u->d_long.p, u->d_words.p + W_LONGCOUNT, u->d_words.p + W_ERR, u->d_words.p + W_STATUS, u->dense ? 1u : 0u};
        agx_launch_tile_fill(&FA, st);                  // every tile's list from its window of the tile order
        if (u->dense) {                                 // the fallback for a unit with more long hits than a list's window scan takes: queued once a build has met them (an idle launch of a kernel per tile costs 0.1 ms)
            agx_bin_args BA{u->d_dhit.p, nh, u->d_tile_off.p, u->d_cursor.p, u->d_unsorted.p, u->list_cap, u->d_words.p + W_LONGCOUNT};
            agx_launch_bin_fill(&BA, st);
            agx_launch_tile_sort(&FA, st);
        }
        AGX_CHECKPOINT("tile_sort");
        // r06: a tile-ordered unit's first build expands its read rows and sweeps its tiles WINDOW BY WINDOW, each when its piece of the upload has landed (the rows of a window's
        // hits are one piece, in front of which lie the rows of every earlier tile: stage_tiled) — the sweep of the unit's front runs beside the upload of its back
        const bool windows = !u->expanded && u->tiled;
        const agx_u32 n_win = (windows && early) ? u->n_win : 1u;      // (a build that has waited for the whole upload on the host — timed sections — has nothing to overlap: one window, no boundaries to pay for)
        agx_u32 wt[9]; size_t wr[9];                                   // the windows' tiles and rows
        for (agx_u32 w = 0; w <= n_win; w++) { wt[w] = n_win == 1 ? (w ? u->n_tiles : 0u) : u->win_tile[w]; wr[w] = n_win == 1 ? (w ? (size_t)nh : 0) : win_row(u, w); }
        swept_windows = windows ? n_win : 1u; swept_timed = windows;
        if (!u->expanded) {   // the vote codes (and the region layout, the last copy of the upload) are first needed by the sweep
            if (!windows) {
                const size_t n_bases = codes_bytes(u) * 4;
                if (early) HIP_OK(hipStreamWaitEvent(st, u->ev_uploaded, 0));      // (at most the tail of this unit's own upload: nothing else is ever waited for on a build stream)
                if (u->rows_diffed) agx_launch_expand_rows(u->d_whits.p, (agx_u32)nh, u->d_wsides.p, u->d_wruns.p, u->d_anchor.p, u->d_blockfirst.p, u->d_rowcnt.p, u->d_blockoff.p, u->d_units.p, u->d_wref.p, u->d_vcodes.p, u->n_rows, u->stride, u->d_other.p, u->n_other, st);
                else agx_launch_expand_codes(u->d_codes.p, u->d_vcodes.p, (n_bases + 15) / 16 * 16, u->d_other.p, u->n_other, st);
                if (u->ref_packed) agx_launch_expand_ref(u->d_wref.p, u->d_ref.p, ((size_t)n_pos + 15) / 16 * 16, u->d_refx.p, (agx_u32)u->n_refx, st);      // (the letters are first read by the sweep's write-out)
            } else {
                if (early) HIP_OK(hipStreamWaitEvent(st, u->ev_rows[0], 0));       // (the small arrays travel in front of the first piece of rows: with it the reference and the region layout are in)
                if (u->ref_packed) agx_launch_expand_ref(u->d_wref.p, u->d_ref.p, ((size_t)n_pos + 15) / 16 * 16, u->d_refx.p, (agx_u32)u->n_refx, st);
            }
            u->expanded = true;
        }
        HIP_OK(hipEventRecord(u->ev_front, st));
        if (windows) {      // still on the front stream, behind ev_front: the rows of window w out of their 2-bit form when they are in, then the listed bases among them
            const size_t s4 = u->stride / 4;
            for (agx_u32 w = 0; w < n_win; w++) {
                const size_t r_lo = wr[w], r_hi = wr[w + 1];
                if (early && w) HIP_OK(hipStreamWaitEvent(st, u->ev_rows[w], 0));
                const unsigned long long *ob = u->s_other_t.p, *oe = ob + u->n_other_t;
                const size_t o_lo = (size_t)(std::lower_bound(ob, oe, (unsigned long long)r_lo * u->stride) - ob), o_hi = (size_t)(std::lower_bound(ob, oe, (unsigned long long)r_hi * u->stride) - ob);
                // (pieces begin at multiples of 16 rows: whole 16-base groups, 16-byte aligned stores; the last piece is padded like the whole array was)
                // (the piece that holds the LAST row — not always the last window's: a unit with few hits has windows without rows — is padded like the whole array was; r06's fuzz found the
                // unpadded form: the last row's last bases stayed unexpanded where an earlier window already reached the last row)
                const size_t b_lo = r_lo * s4 * 4, b_hi = r_hi == (size_t)nh ? (codes_bytes(u) * 4 + 15) / 16 * 16 : r_hi * s4 * 4;
                if (r_hi <= r_lo) { HIP_OK(hipEventRecord(u->ev_win[w], st)); continue; }
                if (u->rows_diffed) agx_launch_expand_rows(u->d_whits.p, (agx_u32)nh, u->d_wsides.p, u->d_wruns.p, u->d_anchor.p, u->d_blockfirst.p + r_lo / 64, u->d_rowcnt.p + r_lo, u->d_blockoff.p + r_lo / 64, u->d_units.p, u->d_wref.p,
                                                           u->d_vcodes.p + b_lo, (agx_u32)(r_hi - r_lo), u->stride, nullptr, 0, st);      // (every row's anchor is its own hit: the anchor bits are all ones)
                else agx_launch_expand_codes(u->d_codes.p + r_lo * s4, u->d_vcodes.p + b_lo, b_hi - b_lo, nullptr, 0, st);
                agx_launch_patch_codes(u->d_other.p + o_lo, o_hi - o_lo, u->d_vcodes.p, st);
                HIP_OK(hipEventRecord(u->ev_win[w], st));
            }
        }
        // ---- main stream ----
        st = turn.main;
        HIP_OK(hipStreamWaitEvent(st, u->ev_front, 0));
        u->ev.mark(B_BIN, st);
        // ---- node sweep: every tile with small LDS buckets, then the tiles that overflowed with wider ones, then with global scratch (device-side lists) ----
        agx_node_kargs K; fill_sweep_args(u, K.S);
#ifdef AGX_SWEEP_STATS
        K.S.sweep_stats = u->d_words.p + W_N + 6;
#endif
        K.pool_cnt = u->d_pool_cnt.p; K.region_off = u->d_region_off.p; K.spill_lo = u->spill_lo; K.spill_cnt = u->d_words.p + W_SPILL;
        K.mid_count = u->d_words.p + W_MIDCOUNT; K.mid_list = u->d_mid_list.p; K.mid_n = u->d_words.p + W_MIDCOUNT;
        K.big_count = u->d_words.p + W_BIGCOUNT; K.big_list = u->d_big_list.p; K.status = u->d_words.p + W_STATUS;
        K.list_cap = u->list_cap; K.big_n = u->d_words.p + W_BIGCOUNT; K.scratch = u->d_scratch.p;
        K.slow_list = u->d_slow_list.p; K.slow_count = u->d_words.p + W_SLOWCOUNT; K.fallback_queued = 1u;
        K.huge_count = u->d_words.p + W_HUGECOUNT; K.huge_n = u->d_words.p + W_HUGECOUNT; K.huge_list = u->d_huge_list.p; K.scratch_huge = u->d_scratch_huge.p; K.huge_queued = u->huge ? 1u : 0u;
        for (agx_u32 w = 0; w < n_win; w++) {
            K.tile_lo = windows ? wt[w] : 0u; K.tile_hi = windows ? wt[w + 1] : u->n_tiles;
            if (windows) { HIP_OK(hipStreamWaitEvent(st, u->ev_win[w], 0)); HIP_OK(hipEventRecord(u->ev_sw0[w], st)); }      // (behind the wait: the sweep's time must not hold the rows' journey)
            agx_launch_node_sweep(&K, st);
            if (windows) HIP_OK(hipEventRecord(u->ev_sw1[w], st));
        }
        AGX_CHECKPOINT("node_sweep");
        u->ev.mark(B_NODE, st); u->stats.node_sweep_launches++;
        hipEvent_t trace_from = turn.prev_node; turn.prev_node = u->ev.e[B_NODE];
        HIP_OK(hipEventRecord(turn.sweep_done[turn.n & 1], st));
        agx_launch_node_sweep_big(&K, st);      // the two fallback passes: they find their (usually empty) tile lists on the device.  A unit is built once, so they
        if (u->huge) agx_launch_node_sweep_huge(&K, st);
        AGX_CHECKPOINT("node_sweep_big");       // are always queued: two idle launches cost less than repeating the build of a unit that turns out to need them
        u->ev.mark(B_BIG, st);
        // ---- edge sweep ----
        agx_edge_kargs E; fill_sweep_args(u, E.S); E.ovf = u->d_ovf.p; E.ovf_count = u->d_words.p + W_OVFCOUNT; E.ovf_cap = u->ovf_cap; E.list_cap = u->list_cap;
        E.n_hits = nh; E.jump_list = u->d_jump.p; E.n_jump = (agx_u32)u->n_jump; E.abort = u->d_words.p + W_STATUS; E.big_list = u->d_big_list.p; E.big_n = u->d_words.p + W_BIGCOUNT; E.slow_list = u->d_slow_list.p; E.slow_count = u->d_words.p + W_SLOWCOUNT;
        agx_launch_edge_sweep(&E, st);
        AGX_CHECKPOINT("edge_sweep");
        // Passes J and B insert edges out of different sources (positions with one variant / with several) and both wait on memory more than
        // they compute: J goes to the front stream, behind pass A and in front of the next build's front, and runs beside B.  (Timed builds
        // keep everything on the main stream.)
        const bool side_j = !u->ev.all && !g_debug_sync;
        if (side_j) {
            HIP_OK(hipEventRecord(u->ev_passA, st));
            HIP_OK(hipStreamWaitEvent(turn.front, u->ev_passA, 0));
            agx_launch_edge_jump(&E, turn.front);
            HIP_OK(hipEventRecord(u->ev_passJ, turn.front));
        } else {
            agx_launch_edge_jump(&E, st);
            AGX_CHECKPOINT("edge_jump");
        }
        u->ev.mark(B_EDGE, st);
        agx_launch_edge_slow(&E, st);
        AGX_CHECKPOINT("edge_slow");
        u->ev.mark(B_SLOW, st);
        if (side_j) HIP_OK(hipStreamWaitEvent(st, u->ev_passJ, 0));
        // ---- walk preparation: side counts -> scan -> walk ids, node records, rewritten edges, forced-run flags ----
        agx_compact_args C; memset(&C, 0, sizeof C);
        C.node_start = u->d_node_start.p; C.node_cnt = u->d_node_cnt.p; C.n_flags = u->d_flags.p; C.n_base = u->d_base.p;
        C.nk_off0 = u->d_off0.p; C.n_sref = u->d_sref.p; C.n_next = u->d_next.p; C.ref = u->d_ref.p; C.n_pos = n_pos;
        C.side_pk = u->d_side_pk.p; C.tile_side_start = u->d_tile_side_start.p; C.aid_of = u->d_aid_of.p;
        C.a_str = u->d_a_str.p; C.a_meta = u->d_a_meta.p; C.a_nid = u->d_a_nid.p; C.ovf = u->d_ovf.p; C.n_ovf = 0; C.a_ovf = u->d_a_ovf.p;
        C.abort = u->d_words.p + W_STATUS;
        C.a_mark = u->d_a_mark.p; C.side_xpos = u->d_side_xpos.p; C.sparse_min = (u->prm.flags & AGX_FLAG_SPARSE_MIN) ? 1u : 0u;
        C.sp_bits = u->d_sp_bits.p; C.sp_cnt = u->d_sp_cnt.p; C.sp_rank = u->d_sp_rank.p; C.sp_node = u->d_sp_node.p; C.sp_cap = u->sp_cap;
        C.segs = u->d_segs.p; C.n_seg0 = u->n_seg0; C.cm_start = u->d_cm_start.p; C.sp_hop = u->d_sp_hop.p; C.seg_index = u->d_segindex.p;
        if (g_scan1) agx_launch_exclusive_scan1(u->d_tile_side.p, u->d_tile_side_start.p, u->n_tiles, u->d_scan_desc.p + u->scan_desc_n, st);
        else agx_launch_exclusive_scan(u->d_tile_side.p, u->d_tile_side_start.p, u->n_tiles, u->d_scan_tmp.p, st);      // per tile: the sweep has scanned inside the tiles
        agx_launch_compact(&C, u->d_chain_end.p, u->n_chain_end, u->d_words.p + W_OVFCOUNT, u->ovf_cap, st);
        AGX_CHECKPOINT("compact");
        u->cuts = stream_cuts(n_pos); u->cuts.cut_out = u->d_words.p + W_CUT;
        agx_launch_special(&C, u->n_words, u->d_sp_rank.p, u->d_scan_tmp.p, g_scan1 ? u->d_scan_desc.p + 2 * u->scan_desc_n : nullptr,
                           u->d_words.p + W_N, u->d_tile_off.p + u->n_tiles, u->d_tile_side_start.p + u->n_tiles, u->d_pool_cnt.p, u->n_regions, u->d_words.p + W_POOL, u->cuts.n ? &u->cuts : nullptr, st);
        u->walk_args = C;
        AGX_CHECKPOINT("special");
        u->ev.mark(B_COMPACT, st);
        // ---- the one synchronisation ----
        u->stats.edge_sweep_launches++;
        if (u->ev.all) HIP_OK(hipEventRecord(u->ev.last, st));
        HIP_OK(hipEventRecord(turn.build_done[turn.n & 1], st));               // (a stream wait binds to the record that precedes it: the handle may be recorded again later)
        {   void *dst = u->h_words.dev(); const void *src = u->d_words.p; const size_t bytes = W_TOTAL * 4;      // the counters, by a kernel at the end of the chain (a copy
            agx_launch_copy_out(&dst, &src, &bytes, 1, st); }                                                              // command would queue behind the uploads on the copy engines)
        HIP_OK(hipEventRecord(u->ev_built, st));
        turn.n++; turn.prev_exclusive = u->ev.all;
        my_turn.unlock();
        trace(u, "build: queue kernels", tb1, n_pos);
        const double tb2 = now_ms();
        HIP_OK(wait_event(u->ev_built));
        trace(u, "build: wait for kernels", tb2, n_pos);
        HIP_OK(hipGetLastError());
        if (g_trace_gap && trace_from && trace_from != u->ev.e[B_NODE]) {      // diagnostic (units must outlive each other's builds): end of the previous sweep -> start of this one
            float f = 0; if (hipEventElapsedTime(&f, trace_from, u->ev.e[B_BIN]) == hipSuccess) fprintf(stderr, "[agx gap] %.3f ms between sweeps, %.3f ms sweep\n", f, u->ev.ms(B_NODE)); else (void)hipGetLastError(); }
        const agx_u32 *w = u->h_words.p;
        if (w[W_ERR] & 1u) throw Error{E_ALIGNMENT, "BOWTIE ALIGNMENT ERROR"};
        if (w[W_ERR] & 2u) throw Error{E_UNSUPPORTED, "read alignment beyond the end of the unit sequence"};
        if (w[W_ERR] & (4u | 8u)) throw Error{E_DEVICE, (w[W_ERR] & 4u) ? "internal: the staged order of the hits is not the order of their first tiles" : "internal: a tile's list does not hold what its histogram counted"};
        u->n_tile_entries = w[W_N];
        // a capacity that was too small: take a larger buffer (the arena keeps the old one until the unit is released) and build again.
        // Whatever a retry changes on the device goes through the download stream and a fresh ev_uploaded, which the next attempt waits for.
        bool again = false;
        if ((w[W_STATUS] & 16u) && !u->dense) { u->dense = true; HIP_OK(hipEventRecord(u->ev_uploaded, turn.down)); continue; }      // more long hits than the window scan takes: again, with the scatter fallback queued
        if (u->n_tile_entries > u->list_cap) { alloc_lists(u, u->n_tile_entries + u->n_tile_entries / 8 + 1024); again = true; }
        else {
            if ((w[W_STATUS] & 2u) && !u->huge) {     // a position beyond the 64 variants of pass 2 (deep repeats under a wide --distanceHigh): queue pass 3 and build again
                u->huge = true; u->d_huge_list.alloc(u->arena, (size_t)u->n_tiles + 1); u->d_scratch_huge.alloc(u->arena, (size_t)AGX_HUGE_WAVES * AGX_NF * AGX_MAXV_HUGE * 64);
                HIP_OK(hipEventRecord(u->ev_uploaded, turn.down)); continue;
            }
            if (w[W_STATUS] & 2u) throw Error{E_OVERFLOW, "more than " + std::to_string(AGX_MAXV_HUGE) + " node variants at one position"};
            if (w[W_STATUS] & 1u) {                  // the node pool ran out: cut the slices to what the regions asked for
                std::vector<agx_u32> padded((size_t)u->n_regions * AGX_REGION_PAD), demand(u->n_regions);
                HIP_OK(hipMemcpyAsync(padded.data(), u->d_pool_cnt.p, padded.size() * 4, hipMemcpyDeviceToHost, turn.down)); HIP_OK(hipStreamSynchronize(turn.down));
                for (agx_u32 r = 0; r < u->n_regions; r++) demand[r] = padded[(size_t)r * AGX_REGION_PAD];
                const unsigned long long need = layout_regions(u, demand.data(), 0, false, turn.down);
                if (need >= 0xFFFFFF00ull) throw Error{E_OVERFLOW, "node table exceeds 2^32 entries"};
                if (need > u->pool_cap) alloc_pool(u, (agx_u32)need);
                layout_regions(u, demand.data(), 0, true, turn.down);
                again = true;
            } else if (w[W_OVFCOUNT] > u->ovf_cap) { alloc_ovf(u, w[W_OVFCOUNT] + w[W_OVFCOUNT] / 2 + 1024); again = true; }
            else if (w[W_N + 2] > u->sp_cap) { alloc_sparse(u, w[W_N + 2] + w[W_N + 2] / 8 + 1024); again = true; }
        }
        if (again) { HIP_OK(hipEventRecord(u->ev_uploaded, turn.down)); continue; }
        u->n_nodes = w[W_POOL]; u->n_big = w[W_BIGCOUNT]; u->n_mid = w[W_MIDCOUNT]; u->n_ovf = w[W_OVFCOUNT];
        const unsigned long long ids = (unsigned long long)n_pos + w[W_N + 1];
        if (ids >= 0xFFFFFF00ull) throw Error{E_OVERFLOW, "walk graph exceeds 2^32 ids"};
        u->n_ids = (agx_u32)ids; u->n_special = w[W_N + 2];
        u->stats.build_attempts = (uint32_t)attempt + 1; u->stats.n_spilled = w[W_SPILL]; u->stats.dense_lists = w[W_LONGCOUNT] > AGX_LONG_MAX ? 2u : w[W_LONGCOUNT] ? 1u : 0u;
#ifdef AGX_SWEEP_STATS
        {   const agx_u32 *c = w + W_N + 6;
            fprintf(stderr, "[agx sweep stats] wave-entries %u (lanes with an arrival %u = %.1f per entry); leave the fast path: %u entries / %u lanes; of those not a first store: %u / %u; "
                            "variant 0 incompatible: %u / %u; several candidate keys: %u / %u; entries with a multi-run record %u\n",
                    c[0], c[3], c[0] ? (double)c[3] / c[0] : 0.0, c[4], c[5], c[6], c[7], c[8], c[9], c[10], c[11], c[12]); }
#endif
        break;
    }
    u->built = true; u->downloaded = false;
    u->stats.ms_build_span = u->ev.all ? u->ev.span() : 0.0;
    u->stats.ms_prep = u->ev.ms(B_PREP); u->stats.ms_bin = u->ev.ms(B_BIN); u->stats.ms_node_sweep = u->ev.ms(B_NODE);
    if (swept_windows >= 1 && swept_timed) {      // a windowed first build: the sweep's time is the sum over its windows (between them the stream may have waited for rows that were still travelling)
        double sum = 0; for (agx_u32 w = 0; w < swept_windows; w++) { float f = 0; if (hipEventElapsedTime(&f, u->ev_sw0[w], u->ev_sw1[w]) == hipSuccess) sum += f; else (void)hipGetLastError(); }
        u->stats.ms_node_sweep = sum;
    }
    u->stats.ms_node_big = u->ev.ms(B_BIG); u->stats.ms_edge_fast = u->ev.ms(B_EDGE); u->stats.ms_edge_slow = u->ev.ms(B_SLOW); u->stats.ms_edge_sweep = u->stats.ms_edge_fast + u->stats.ms_edge_slow; u->stats.ms_compact = u->ev.ms(B_COMPACT);
    if (u->up_timed) { float f = 0; if (hipEventElapsedTime(&f, u->ev_up0, u->ev_uploaded) == hipSuccess) u->stats.ms_upload_dev = f; else (void)hipGetLastError(); }
}

// the pinned buffers a download lands in (a one-shot unit: cut from its dead staged inputs)
void download_buffers(agx_unit *u) {
    const size_t n_pos = u->V.n_pos, ni = u->n_ids;
    const size_t nw = ni / 64 + 1, ns = u->n_special, nside = ni - n_pos;
    join_dl_helper(u);
    if (u->prm.flags & AGX_FLAG_ONE_SHOT) {
        // The inputs are in HBM and will not be uploaded again: their staged copies are dead pinned memory.  The download's arrays are cut
        // from the largest of those blocks, largest array first; what does not fit (thin read sets) gets a buffer of its own below.
        struct Room { char *at; size_t left; } room[8] = {{(char *)u->s_codes.p, u->s_codes.block_bytes()}, {(char *)u->s_hits.p, u->s_hits.block_bytes()}, {(char *)u->s_landing.p, u->s_landing.block_bytes()},
                                                          {(char *)u->s_codes_t.p, u->s_codes_t.block_bytes()}, {(char *)u->s_hits_t.p, u->s_hits_t.block_bytes()},
                                                          {(char *)u->s_runs.p, u->s_runs.block_bytes()}, {(char *)u->s_sides.p, u->s_sides.block_bytes()}, {(char *)u->s_other.p, u->s_other.block_bytes()}};
        // (a loan from an earlier download of this unit object must not survive into alloc() below: the memory it names has been handed out again)
        u->h_sp_node.release(); u->h_a_meta.release(); u->h_a_str.release(); u->h_sp_hop.release(); u->h_side_xpos.release(); u->h_sp_bits.release(); u->h_sp_rank.release(); u->h_a_ovf.release();
        auto cut = [&](auto &buf, size_t count) {
            using T = typename std::remove_reference<decltype(*buf.p)>::type;
            const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
            for (Room &r : room) if (r.at && r.left >= bytes) { buf.borrow((T *)r.at, count); r.at += bytes; r.left -= bytes; return; }
        };
        u->consumed = true; u->staged = false;
        cut(u->h_sp_node, ns + 1); cut(u->h_a_meta, ni + 64); cut(u->h_a_str, ni + 1); cut(u->h_sp_hop, ns + 2); cut(u->h_side_xpos, nside + 1);
        cut(u->h_sp_bits, nw + 1); cut(u->h_sp_rank, nw + 1); cut(u->h_a_ovf, (size_t)u->n_ovf + 1);
    }
    u->h_a_str.alloc(ni + 1); u->h_a_meta.alloc(ni + 64); u->h_side_xpos.alloc(nside + 1);
    u->h_sp_bits.alloc(nw + 1); u->h_sp_rank.alloc(nw + 1); u->h_sp_node.alloc(ns + 1); u->h_sp_hop.alloc(ns + 2);
    u->h_a_ovf.alloc((size_t)u->n_ovf + 1);
}

void do_download(agx_unit *u) {
    if (!u->built) do_build(u);
    HIP_OK(hipSetDevice(u->prm.device));             // the calling thread may never have touched this device
    const double t0 = now_ms();
    const size_t n_pos = u->V.n_pos, ni = u->n_ids;
    DeviceTurn &turn = turn_of(u->prm.device);
    const size_t nw = ni / 64 + 1, ns = u->n_special, nside = ni - n_pos;
    download_buffers(u);
    u->dl_streaming = false;
    // the walk graph into the pinned buffers: plain copy commands on the device's download stream.  (r02 first used a kernel of its own for
    // this — copy commands seemed to queue behind other units' uploads; that was the hardware queue the streams shared.  Its stores to host
    // memory slowed whatever ran beside it, the next unit's binning most of all: the five builds of a cfg3 job ended at 45 ms with it, at
    // 42-45 ms with grids of 16-128 blocks, at 35 ms with the runtime's copies.)
    {
        void *dst[24]; const void *src[24]; size_t bytes[24]; int n = 0;
        auto add = [&](void *h, const void *d, size_t b) { if (b) { dst[n] = h; src[n] = d; bytes[n] = b; n++; } };
        if (ni) { add(u->h_sp_bits.p, u->d_sp_bits.p, nw * 8); add(u->h_sp_rank.p, u->d_sp_rank.p, nw * 4); add(u->h_a_str.p, u->d_a_str.p, ni); add(u->h_a_meta.p, u->d_a_meta.p, ni); }
        add(u->h_side_xpos.p, u->d_side_xpos.p, nside * 4);
        add(u->h_sp_node.p, u->d_sp_node.p, ns * sizeof(agx_walknode)); add(u->h_sp_hop.p, u->d_sp_hop.p, ns * sizeof(agx_hop)); add(u->h_a_ovf.p, u->d_a_ovf.p, (size_t)u->n_ovf * sizeof(agx_edge_ovf));
        bool by_engines = u->dl_sdma && n > 0;
        if (by_engines) {                            // (the build is complete and visible: this thread has waited for its last command)
            hsa_signal_store_relaxed(u->dl_signal, n);
            int queued = 0;
            for (; queued < n; queued++) if (hsa_amd_memory_async_copy(dst[queued], hsa_copy().cpu, src[queued], u->dl_agent, bytes[queued], 0, nullptr, u->dl_signal) != HSA_STATUS_SUCCESS) break;
            if (queued < n) hsa_signal_subtract_relaxed(u->dl_signal, n - queued);      // what was queued still counts down
            hsa_signal_value_t v;
            do v = hsa_signal_wait_scacquire(u->dl_signal, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED); while (v >= 1);      // (the wait may return early: only a value below 1 means the copies are done)
            if (queued < n || v < 0) { u->dl_sdma = false; by_engines = false; }        // the engines refused: this and every later download of the unit through HIP
        }
        if (!by_engines) {
            std::lock_guard<std::mutex> l(turn.down_m);
            for (int i = 0; i < n; i++) HIP_OK(hipMemcpyAsync(dst[i], src[i], bytes[i], hipMemcpyDeviceToHost, turn.down));
            HIP_OK(hipEventRecord(u->ev_dl, turn.down));
        }
    }
    const double t1 = now_ms();
    HIP_OK(wait_event(u->ev_dl));
    if (getenv("AGX_DL_TIMING")) fprintf(stderr, "[agx download] buffers %.2f ms, copies %.2f ms (%zu ids, %zu records)\n", t1 - t0, now_ms() - t1, ni, ns);
    memset(u->h_a_meta.p + ni, 0, 64);
    u->stats.n_walk_ids = ni; u->stats.n_special = ns;
    u->stats.download_bytes = 2 * ni + nw * 12 + nside * 4 + ns * (sizeof(agx_walknode) + sizeof(agx_hop)) + (size_t)u->n_ovf * sizeof(agx_edge_ovf);
    u->downloaded = true;
    u->stats.ms_download = now_ms() - t0;
    trace(u, u->dl_sdma ? "download (copy engines)" : "download (hipMemcpyAsync)", t0, n_pos);
}

// ---- streamed download (r06) ------------------------------------------------------------------------------------------------------------
// A unit's chain used to be upload -> build -> download -> walk in series; the walk of a large unit is split between walkers that each look at a WINDOW of the walk graph
// only (agx_walk.cpp: walk_split).  So the download goes out in position windows from the front — every window's meta bytes, bitmap words, ranks, records and hop entries of
// its main ids and of its positions' side ids; the bases last: the walk only takes byte ranges of them — each with its own completion signal, and agx_unit_finish starts
// the walk when the small head (side positions, overflow edges) is in.  A walker waits for its window (GraphView::wait_landed), the first walker — whose walks may lead anywhere
// — for all of them, and the stretches are cut by when the windows land.  What it takes off a unit's chain: about half its download (chr1 of configs[4]: 29 ms of 149).
// The cuts' ranks come with the build's counters (agx_cut_args).  Not for units that are trimmed after their download (agx_unit_download + agx_unit_trim: the caller wants the
// HBM back before the walk) — agx_unit_finish streams when nothing has been downloaded yet.
// (asleep: hsa_signal_wait spins whatever wait state it is asked for — sixteen walkers waiting for their windows burned 150 CPU-ms per cfg3 job — so: a short spin for what is about
// to land, then looks between short sleeps, like wait_event)
inline void wait_signal(hsa_signal_t g) {
    for (int i = 0; i < 64; i++) if (hsa_signal_load_scacquire(g) < 1) return;
    for (;;) { const timespec ts{0, 20000}; nanosleep(&ts, nullptr); if (hsa_signal_load_scacquire(g) < 1) return; }
}
std::atomic<double> &download_rate() { static std::atomic<double> r{40e6}; return r; }      // bytes per millisecond the engines have delivered (40 GB/s until measured)
void stream_wait_landed(void *ctx, agx_u32 main_hi, agx_u32 side_hi) {
    agx_unit *u = (agx_unit *)ctx; const int n = (int)u->cuts.n; int need = 0;
    while (need < n && (u->cut_main[need] < main_hi || u->cut_side[need] < side_hi)) need++;
    for (int p = 1; p <= need; p++) wait_signal(u->dl_piece[p]);
    if (need == n && !u->dl_timed.exchange(true)) {      // the last window is in: what the engines delivered per millisecond, for the next unit's estimate
        const double ms = now_ms() - u->dl_t0; u->stats.ms_download = ms;
        if (ms > 0.05 && u->dl_stream_bytes > (4u << 20)) download_rate().store(0.5 * download_rate().load() + 0.5 * std::min(std::max((double)u->dl_stream_bytes / ms, 35e6), 60e6));      // (clamped to what the link does: a download that queued behind another unit's, or whose last window was only looked at late, says little about it — chr1's 20 ms were once taken for 43)
        trace(u, "download (streamed): last window", u->dl_t0, u->V.n_pos);
    }
}
void stream_wait_str(void *ctx) { agx_unit *u = (agx_unit *)ctx; wait_signal(u->dl_piece[u->cuts.n + 1]); }
void stream_wait_all(agx_unit *u) { if (!u->dl_streaming) return; for (agx_u32 p = 0; p <= u->cuts.n + 1; p++) wait_signal(u->dl_piece[p]); u->dl_streaming = false; }

// false: this unit's download is not streamed (no cuts, no copy engines, counters that do not add up): the caller downloads the usual way
bool begin_streamed_download(agx_unit *u) {
    if (!u->built) do_build(u);
    const agx_u32 M = u->cuts.n;
    if (!M || !u->dl_sdma || getenv("AGX_NO_STREAM_DOWNLOAD")) return false;
    HIP_OK(hipSetDevice(u->prm.device));
    const double t0 = now_ms();
    const size_t n_pos = u->V.n_pos, ni = u->n_ids, ns = u->n_special, nside = ni - n_pos;
    const agx_u32 *rank = u->h_words.p + W_CUT, *side = rank + (AGX_DL_PIECES + 1);
    // the cuts must be what the tables say: ranks and side counts ascending, every side id special, the last cut the whole table
    if (rank[0] != 0 || side[0] != 0 || side[M] != nside || (size_t)rank[M] + nside != ns) return false;
    for (agx_u32 w = 0; w < M; w++) if (rank[w] > rank[w + 1] || side[w] > side[w + 1]) return false;
    if (!u->dl_piece_made) { for (hsa_signal_t &g : u->dl_piece) if (hsa_signal_create(0, 0, nullptr, &g) != HSA_STATUS_SUCCESS) { u->dl_sdma = false; return false; } u->dl_piece_made = true; }
    download_buffers(u);
    memset(u->h_a_meta.p + ni, 0, 64);
    for (agx_u32 w = 0; w <= M; w++) { u->cut_main[w] = w == M ? (agx_u32)n_pos : u->cuts.word[w] * 64u; u->cut_side[w] = (agx_u32)n_pos + side[w]; }
    struct Copy { void *dst; const void *src; size_t bytes; int piece; };
    std::vector<Copy> copies; copies.reserve(12 * (size_t)M + 4);
    auto add = [&](int piece, void *h, const void *d, size_t b) { if (b) copies.push_back(Copy{h, d, b, piece}); };
    add(0, u->h_side_xpos.p, u->d_side_xpos.p, nside * 4); add(0, u->h_a_ovf.p, u->d_a_ovf.p, (size_t)u->n_ovf * sizeof(agx_edge_ovf));
    add(0, u->h_sp_bits.p, u->d_sp_bits.p, (ni / 64 + 1) * 8); add(0, u->h_sp_rank.p, u->d_sp_rank.p, (ni / 64 + 1) * 4);      // (the bitmap and its ranks whole: 0.2 bytes per id, and two copies instead of four per window — a copy command costs the engines ~10 us)
    size_t main_bytes = 0;
    auto ids = [&](int piece, size_t lo, size_t hi, size_t r_lo, size_t r_hi) {      // everything about walk ids [lo, hi), whose records are [r_lo, r_hi) of the sparse table
        if (lo >= hi) return;
        add(piece, u->h_a_meta.p + lo, u->d_a_meta.p + lo, hi - lo);
        add(piece, u->h_sp_node.p + r_lo, u->d_sp_node.p + r_lo, (r_hi - r_lo) * sizeof(agx_walknode)); add(piece, u->h_sp_hop.p + r_lo, u->d_sp_hop.p + r_lo, (r_hi - r_lo) * sizeof(agx_hop));
        main_bytes += (hi - lo) + (r_hi - r_lo) * (sizeof(agx_walknode) + sizeof(agx_hop));
    };
    for (agx_u32 w = 0; w < M; w++) {
        ids(1 + (int)w, u->cut_main[w], u->cut_main[w + 1], rank[w], rank[w + 1]);
        ids(1 + (int)w, u->cut_side[w], u->cut_side[w + 1], (size_t)rank[M] + side[w], (size_t)rank[M] + side[w + 1]);
    }
    add((int)M + 1, u->h_a_str.p, u->d_a_str.p, ni);
    int count[agx_unit::DL_SIGNALS] = {};
    for (const Copy &c : copies) count[c.piece]++;
    for (agx_u32 p = 0; p <= M + 1; p++) hsa_signal_store_relaxed(u->dl_piece[p], count[p]);
    u->dl_streaming = true; u->dl_timed.store(false); u->dl_t0 = now_ms(); u->dl_stream_bytes = main_bytes; u->dl_est_ms = (double)main_bytes / download_rate().load();
    size_t queued = 0;
    for (; queued < copies.size(); queued++) { const Copy &c = copies[queued]; if (hsa_amd_memory_async_copy(c.dst, hsa_copy().cpu, c.src, u->dl_agent, c.bytes, 0, nullptr, u->dl_piece[c.piece]) != HSA_STATUS_SUCCESS) break; }
    if (queued < copies.size()) {      // the engines refused: what was queued still counts down; then the usual way (through HIP from now on)
        for (size_t i = queued; i < copies.size(); i++) hsa_signal_subtract_relaxed(u->dl_piece[copies[i].piece], 1);
        stream_wait_all(u); u->dl_sdma = false; return false;
    }
    wait_signal(u->dl_piece[0]);
    u->stats.n_walk_ids = ni; u->stats.n_special = ns;
    u->stats.download_bytes = 2 * ni + (ni / 64 + 1) * 12 + nside * 4 + ns * (sizeof(agx_walknode) + sizeof(agx_hop)) + (size_t)u->n_ovf * sizeof(agx_edge_ovf);
    trace(u, "download (streamed): queued, head in", t0, n_pos);
    return true;
}

// After the download: everything on the device that the walk cannot ask for (all but the arrays agx_walk_record reads, which lie at the front of the unit's block: alloc_pool)
// goes back to the device's memory region.  The unit is no longer built: another build uploads it again.  Returns the bytes given back (0: the block is not the region's — a
// unit below the region's threshold on a device without one —, or a capacity grew during the build and the arrays are no longer at the front).
size_t do_trim(agx_unit *u) {
    if (!u->downloaded) throw Error{E_ARG, "trim: the unit's walk graph has not been downloaded"};
    HIP_OK(hipSetDevice(u->prm.device));
    const char *base = (const char *)u->arena.base();
    if (!base) return 0;
    size_t keep = 0;
    auto end_of = [&](const void *p, size_t bytes) { if (p) { const size_t e = (size_t)((const char *)p - base) + bytes; if (e > keep) keep = e; } };
    end_of(u->d_aid_of.p, u->d_aid_of.n * 4); end_of(u->d_a_nid.p, u->d_a_nid.n * 4); end_of(u->d_off0.p, u->d_off0.n * 4);
    end_of(u->d_sref.p, u->d_sref.n * sizeof(agx_sref)); end_of(u->d_next.p, u->d_next.n * 4); end_of(u->d_fetch.p, u->d_fetch.n * sizeof(agx_walknode));
    if (keep > u->arena.capacity()) return 0;            // (an array in a later block: nothing is given back)
    const size_t freed = u->arena.shrink_to(keep);
    if (freed) { u->built = false; u->uploaded = false; }      // what lay behind the kept arrays is gone: the build's buffers, the uploaded inputs
    return freed;
}

// Gives back everything the unit holds on the device and its download buffers (to the caches of agx_mem.h: the next unit of the run takes
// them without a driver call).  The inputs stay staged: the unit can be uploaded again as if it were new.
void do_release(agx_unit *u) {
    const double tr0 = now_ms();
    if (u->pending_walk) { u->pending_walk = false; g_walks_pending.fetch_sub(1); }
    struct Tr { agx_unit *u; double t; ~Tr() { trace(u, "release", t, u->V.n_pos); } } tr{u, tr0};
    join_dl_helper(u);                                 // (it fills the download buffers released below)
    stream_wait_all(u);
    if (u->uploaded) { (void)hipSetDevice(u->prm.device); (void)hipEventSynchronize(u->ev_uploaded); (void)hipEventSynchronize(u->ev_built); (void)hipEventSynchronize(u->ev_dl); }      // (its commands are done before its memory goes)
    for (auto *b : {&u->d_cm_start, &u->d_tile_cnt, &u->d_tile_off, &u->d_cursor, &u->d_unsorted, &u->d_tile_recs, &u->d_scan_tmp, &u->d_words, &u->d_pool_cnt, &u->d_region_off, &u->d_node_start,
                    &u->d_slow_list, &u->d_perm, &u->d_tfirst, &u->d_ckey, &u->d_long, &u->d_cid, &u->d_coff, &u->d_cid0, &u->d_coff0, &u->d_off0, &u->d_next, &u->d_mid_list, &u->d_big_list, &u->d_scratch,
                    &u->d_side_pk, &u->d_tile_side, &u->d_tile_side_start, &u->d_aid_of, &u->d_a_nid, &u->d_chain_end, &u->d_side_xpos, &u->d_sp_cnt, &u->d_sp_rank}) b->release();
    u->d_node_cnt.release();
    for (auto *b : {&u->d_pos_succ, &u->d_base, &u->d_flags, &u->d_a_meta, &u->d_a_mark, &u->d_codes, &u->d_vcodes}) b->release();
    u->d_units.release(); u->d_rowcnt.release(); u->d_blockoff.release(); u->d_blockfirst.release(); u->d_anchor.release();
    u->d_other.release(); u->d_whits.release(); u->d_wsides.release(); u->d_wruns.release(); u->d_wref.release(); u->d_refx.release();
    u->d_cm.release(); u->d_cm_head.release(); u->d_ref.release(); u->d_cm_cnt.release(); u->d_segs.release(); u->d_up_desc.release(); u->d_cntruns.release(); u->d_cntchunks.release(); u->d_segchunks.release(); u->d_jump.release(); u->d_segindex.release(); u->d_sp_hop.release(); u->d_runs.release(); u->d_dhit.release(); u->d_scan_desc.release(); u->d_sref.release(); u->d_counts.release();
    u->d_ovf.release(); u->d_a_ovf.release(); u->d_huge_list.release(); u->d_scratch_huge.release(); u->huge = false; u->dense = false; u->d_a_str.release(); u->d_fetch.release(); u->d_sp_node.release(); u->d_sp_bits.release();
    u->arena.reset();
    u->h_a_str.release(); u->h_a_meta.release(); u->h_side_xpos.release(); u->h_sp_rank.release(); u->h_sp_bits.release(); u->h_sp_node.release(); u->h_fetch.release(); u->h_a_ovf.release();
    u->h_sp_hop.release();
    u->pool_cap = u->spill_lo = u->ovf_cap = u->list_cap = u->sp_cap = 0;
    u->uploaded = u->built = u->downloaded = false;
    join_helper(u);
    u->out.pre_extended.clear(); u->out.extended.clear(); u->out_initial.clear(); u->out_ready = false;
}

// records of non-special walk ids: built on the device from the node table, which stays in HBM (agx_walk_record), then one copy
void fetch_records(void *ctx, agx_u32 first, agx_u32 stride, agx_u32 rows, agx_u32 width, agx_walknode *out) {
    agx_unit *u = (agx_unit *)ctx;
    const size_t n = (size_t)rows * width;
    if (!n) return;
    if ((size_t)first + (size_t)(rows - 1) * stride + width > u->n_ids || n > 0x7FFFFFFFull) throw Error{E_ARG, "record fetch beyond the walk graph"};
    HIP_OK(hipSetDevice(u->prm.device));
    DeviceTurn &turn = turn_of(u->prm.device);
    std::lock_guard<std::mutex> l(turn.down_m);      // (also: the walkers of a large unit share the unit's fetch buffers)
    u->h_fetch.alloc(n); u->d_fetch.alloc(u->arena, n);
    agx_compact_args C = u->walk_args; C.n_ids = u->n_ids;
    agx_launch_fetch_records(&C, first, stride, rows, width, u->d_fetch.p, turn.down);
    HIP_OK(hipMemcpyAsync(u->h_fetch.p, u->d_fetch.p, n * sizeof(agx_walknode), hipMemcpyDeviceToHost, turn.down));
    HIP_OK(hipStreamSynchronize(turn.down));
    memcpy(out, u->h_fetch.p, n * sizeof(agx_walknode));
}

GraphView view_of(agx_unit *u) {
    GraphView G; G.n_pos = (agx_u32)u->V.n_pos; G.n_ids = u->n_ids;
    G.meta = u->h_a_meta.p; G.meta_rw = u->h_a_meta.p; G.str = u->h_a_str.p; G.side_xpos = u->h_side_xpos.p;
    G.sp_bits = u->h_sp_bits.p; G.sp_rank = u->h_sp_rank.p; G.sp_node = u->h_sp_node.p; G.sp_hop = u->h_sp_hop.p; G.n_special = u->n_special;
    G.fetch = fetch_records; G.fetch_ctx = u;
    G.ovf = u->h_a_ovf.p; G.n_ovf = u->n_ovf; G.row_slot = u->row_slot.empty() ? nullptr : u->row_slot.data();
    return G;
}

template <class F> int guarded(agx_unit *u, F f) {
    try { f(); return AGX_OK; }
    catch (const Error &e) { if (u) u->err = e.msg; return e.code ? e.code : AGX_E_ARG; }
    catch (const std::bad_alloc &) { if (u) u->err = "out of host memory"; return AGX_E_ARG; }
    catch (const std::exception &e) { if (u) u->err = e.what(); return AGX_E_ARG; }
}

void write_file(const std::string &path, const std::string &data) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw Error{E_IO, "CANNOT OPEN FILE! (" + path + ")"};
    if (!data.empty() && fwrite(data.data(), 1, data.size(), f) != data.size()) { fclose(f); throw Error{E_IO, "short write to " + path}; }
    fclose(f);
}

}  // namespace

extern "C" {

const char *agx_version(void) { return "aligngraph_amd 0.1 (gfx950)"; }

int agx_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

int agx_selftest_scan(int device, uint32_t n, uint32_t seed) {
    try {
        HIP_OK(hipSetDevice(device));
        std::vector<agx_u32> in((size_t)n + 1, 0), want((size_t)n + 1), got((size_t)n + 1);
        agx_u32 x = seed * 2654435761u + 1u, acc = 0;
        for (uint32_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; in[i] = (x % 5u == 0) ? x % 97u : 0u; }       // mostly zeros, like the side-id counts
        for (uint32_t i = 0; i <= n; i++) { want[i] = acc; acc += in[i]; }
        const size_t nb = ((size_t)n + 1 + 4095) / 4096;
        DevArena arena; arena.device = device;
        DBuf<agx_u32> d_in, d_out; DBuf<unsigned long long> d_desc;
        d_in.alloc(arena, (size_t)n + 1); d_out.alloc(arena, (size_t)n + 1); d_desc.alloc(arena, nb + 1);
        HIP_OK(hipMemcpy(d_in.p, in.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemset(d_desc.p, 0, (nb + 1) * 8)); HIP_OK(hipMemset(d_out.p, 0xFF, ((size_t)n + 1) * 4));
        agx_launch_exclusive_scan1(d_in.p, d_out.p, n, d_desc.p, nullptr);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(got.data(), d_out.p, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
        return got == want ? AGX_OK : AGX_E_DEVICE;
    } catch (const Error &) { return AGX_E_DEVICE; }
}

int agx_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
    size_t f = 0, t = 0;
    if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) return AGX_E_DEVICE;
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return AGX_OK;
}

int agx_unit_create(const agx_params *p, agx_unit **out) {
    if (!p || !out) return AGX_E_ARG;
    *out = nullptr;
    if (p->k == 0 || p->k >= 32768) return AGX_E_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return AGX_E_NOGPU;
    if (p->device < 0 || p->device >= n) return AGX_E_ARG;
    agx_unit *u = new (std::nothrow) agx_unit();
    if (!u) return AGX_E_ARG;
    u->prm = *p; if (u->prm.batch == 0) u->prm.batch = 1000000;
    const int rc = guarded(u, [&] {
        try { u->helper.start(); } catch (...) { }      // (without it the unit prepares its buffers on the caller's threads)
        HIP_OK(hipSetDevice(p->device));
        u->ev.init(); u->ev.all = (u->prm.flags & AGX_FLAG_TIME_SECTIONS) != 0;
        for (hipEvent_t *e : {&u->ev_front, &u->ev_passA, &u->ev_passJ}) HIP_OK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        for (hipEvent_t *e : {&u->ev_dl, &u->ev_built, &u->ev_hits}) HIP_OK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        HIP_OK(hipEventCreate(&u->ev_up0)); HIP_OK(hipEventCreate(&u->ev_uploaded));
        for (int i = 0; i < 8; i++) { HIP_OK(hipEventCreateWithFlags(&u->ev_rows[i], hipEventDisableTiming)); HIP_OK(hipEventCreateWithFlags(&u->ev_win[i], hipEventDisableTiming)); HIP_OK(hipEventCreate(&u->ev_sw0[i])); HIP_OK(hipEventCreate(&u->ev_sw1[i])); }
        u->dl_sdma = hsa_copy().agent_of(p->device, u->dl_agent) && hsa_signal_create(0, 0, nullptr, &u->dl_signal) == HSA_STATUS_SUCCESS;
    });
    if (rc != AGX_OK) { delete u; return rc; }
    g_units_alive.fetch_add(1);
    *out = u;
    return AGX_OK;
}

void agx_unit_destroy(agx_unit *u) { if (u) { (void)hipSetDevice(u->prm.device); do_release(u); delete u; if (g_units_alive.fetch_sub(1) == 1) out_cache_trim(); } }

const char *agx_unit_error(const agx_unit *u) { return u ? u->err.c_str() : "null unit"; }

int agx_unit_set_reference(agx_unit *u, const char *bases, uint32_t n) {
    if (!u || (!bases && n)) return AGX_E_ARG;
    return guarded(u, [&] { drop_outputs(u); u->T = Threads(); u->T.ref.assign(bases, n); u->T.n_ref = n; u->have_ref = true; u->have_threads = false; u->staged = false; u->uploaded = false; u->built = false; });
}

int agx_unit_set_contig_threads(agx_unit *u, const char *appended, uint32_t n_appended, const uint32_t *cm_start, const agx_contimer *cm, uint32_t n_cm,
                                const char *initial_contigs, size_t initial_len) {
    if (!u || !cm_start || (!cm && n_cm) || (!appended && n_appended)) return AGX_E_ARG;
    return guarded(u, [&] {
        if (!u->have_ref) throw Error{E_ARG, "set the reference first"};
        drop_outputs(u);
        u->T.ref.resize(u->T.n_ref); u->T.ref.append(appended ? appended : "", n_appended);
        const size_t n_pos = u->T.ref.size();
        u->T.cm_start.assign(cm_start, cm_start + n_pos + 1);
        if (u->T.cm_start[n_pos] != n_cm) throw Error{E_ARG, "cm_start does not end at n_cm"};
        u->T.cm.resize(n_cm);
        for (uint32_t i = 0; i < n_cm; i++) {
            if (cm[i].next_off != AGX_NONE && cm[i].next_off >= n_pos) throw Error{E_ARG, "conti-mer link beyond the position array"};
            u->T.cm[i] = ContiMer{cm[i].nuc, cm[i].cid, cm[i].coff, cm[i].next_off, cm[i].next_item};
        }
        u->T.initial_contigs.assign(initial_contigs ? initial_contigs : "", initial_len);
        build_chains(u->T);
        u->have_threads = true; u->staged = false; u->uploaded = false; u->built = false;
    });
}

int agx_unit_push_pairs(agx_unit *u, const agx_pair_batch *b) {
    if (!u || !b) return AGX_E_ARG;
    return guarded(u, [&] {
        drop_outputs(u);
        if (u->pairs_staged) { u->pairs_staged = false; u->row_off.clear(); u->reads_keep.reset(); u->P = Pairs(); }
        if (u->P.hits.empty()) { u->P = Pairs(); u->P.stride = b->stride; }
        if (b->stride != u->P.stride) throw Error{E_ARG, "all batches of a unit must use one read stride"};
        const agx_u32 slot0 = u->P.n_slots, run0 = (agx_u32)u->P.runs.size();
        for (uint64_t i = 0; i < b->n_hits; i++) {
            agx_hit h = b->hits[i];
            if (h.len == 0 || h.len > b->stride || (uint64_t)h.slot1 + 1 >= b->n_slots) throw Error{E_ARG, "hit names a read slot outside the batch"};
            if ((h.nruns1 && (uint64_t)h.runs1 + h.nruns1 > b->n_runs) || (h.nruns2 && (uint64_t)h.runs2 + h.nruns2 > b->n_runs)) throw Error{E_ARG, "hit names runs outside the batch"};
            if (h.back > i) throw Error{E_ARG, "hit.back reaches before the batch"};
            h.slot1 += slot0; if (h.nruns1) h.runs1 += run0; if (h.nruns2) h.runs2 += run0;
            u->P.hits.push_back(h);
        }
        u->P.runs.insert(u->P.runs.end(), b->runs, b->runs + b->n_runs);
        u->P.bases.append(b->bases, (size_t)b->n_slots * b->stride);
        u->P.n_slots += b->n_slots;
        u->staged = false; u->uploaded = false; u->built = false;
    });
}

struct agx_reads { std::shared_ptr<agx::ReadsIndex> idx; };

int agx_reads_open(const char *reads_fa, agx_reads **out, char *err, size_t err_len) {
    if (!reads_fa || !out) return AGX_E_ARG;
    *out = nullptr;
    try { std::unique_ptr<agx_reads> r(new agx_reads); r->idx.reset(reads_index_open(reads_fa), reads_index_close); *out = r.release(); return AGX_OK; }
    catch (const Error &e) { if (err && err_len) snprintf(err, err_len, "%s", e.msg.c_str()); return e.code ? e.code : AGX_E_ARG; }
    catch (const std::exception &e) { if (err && err_len) snprintf(err, err_len, "%s", e.what()); return AGX_E_ARG; }
}

void agx_reads_close(agx_reads *reads) { delete reads; }      // (units that were loaded through it keep the mapping alive for as long as they need it)

int agx_unit_load_files(agx_unit *u, const char *tmp_dir, int unit) { return agx_unit_load_files_shared(u, tmp_dir, unit, nullptr); }

// The five text files of a unit -> staged arrays.  Three independent pieces of work — the unit sequence + contig threading, the read alignments, and
// (without a shared agx_reads) the index of tmp/_reads.fa — of which the first runs on a thread of its own beside the others.  Each piece first
// tries the fast loader (agx_load.cpp) and takes the general one (agx_host.cpp) when that declines; AGX_NO_FAST_LOAD=1 turns the fast ones off.
int agx_unit_load_files_shared(agx_unit *u, const char *tmp_dir, int unit, const agx_reads *reads) {
    if (!u || !tmp_dir) return AGX_E_ARG;
    return guarded(u, [&] {
        const std::string d = tmp_dir, s = std::to_string(unit);
        u->stats.from_cache = 0;
        if (load_cache(u, d, unit)) return;               // the unit's staged form, written when the alignments were distributed (agx_unit_cache_build)
        const double t0 = now_ms();
        const bool fast = getenv("AGX_NO_FAST_LOAD") == nullptr;
        drop_outputs(u);
        HIP_OK(hipSetDevice(u->prm.device));              // (the fast loader writes into pinned memory, and registering it needs a current device)
        u->T = Threads(); u->P = Pairs(); u->pairs_staged = false; u->row_off.clear(); u->row_slot.clear(); u->reads_keep.reset(); u->reads_map.reset(); u->cache_map.reset(); u->pairs_file = agx::PairsFile();
        u->staged = false; u->uploaded = false; u->built = false; u->consumed = false;      // (new inputs: whatever a download did to the old staged ones no longer matters)
        double ms_thread = 0;
        std::exception_ptr thread_err;
        auto threads_job = [&] {
            try {
                const double a = now_ms();
                load_unit_reference(d + "/_genome." + s + ".fa", u->T.ref);
                if (!(fast && thread_contigs_fast(d + "/_contigs.fa", d + "/_contigs_genome." + s + ".psl", u->T)))
                    thread_contigs_from_files(d + "/_contigs.fa", d + "/_contigs_genome." + s + ".psl", u->T);
                ms_thread = now_ms() - a;
            } catch (...) { thread_err = std::current_exception(); }
        };
        std::thread side; bool side_started = false;
        try { side = std::thread(threads_job); side_started = true; } catch (const std::system_error &) { }
        struct Join { std::thread &t; bool on; ~Join() { if (on && t.joinable()) t.join(); } } join{side, side_started};
        if (!side_started) threads_job();
        if (thread_err && !side_started) std::rethrow_exception(thread_err);      // (the reference loads the genome and the contig alignment first: their errors come first)
        const double tp0 = now_ms();
        std::exception_ptr pairs_err;
        try {
            const std::string sam = d + "/_reads_genome." + s + ".bowtie";
            bool done = false;
            {   // the alignments handed over staged (tmp/_agx_pairs.<u>.bin) stand in for the two text files
                agx::PairsFile F;
                if (open_pairs_file(pairsfile::path_of(d, unit), F)) { load_staged_pairs(u, F); done = true; }
            }
            std::shared_ptr<ReadsIndex> idx = done ? std::shared_ptr<ReadsIndex>() : reads ? reads->idx : std::shared_ptr<ReadsIndex>(reads_index_open(d + "/_reads.fa"), reads_index_close);
            if (!done && fast) {
                struct stat sb; const size_t sam_bytes = stat(sam.c_str(), &sb) == 0 ? (size_t)sb.st_size : 0;
                UnitSink sink(u); StagedPairs S;
                if (load_pairs_fast(*idx, sam, (long)u->prm.batch, u->prm.k, loader_threads(sam_bytes), sink, S)) { adopt_pairs(u, S); u->pairs_staged = true; u->reads_keep = idx; u->n_slots = 0; done = true; }
            }
            if (!done) load_pairs_from_files(d + "/_reads.fa", sam, (long)u->prm.batch, u->prm.k, u->P, idx.get());
        } catch (...) { pairs_err = std::current_exception(); }
        if (side_started) { side.join(); join.on = false; }
        if (thread_err) std::rethrow_exception(thread_err);
        if (pairs_err) std::rethrow_exception(pairs_err);
        u->stats.ms_thread = ms_thread; u->stats.ms_parse = now_ms() - tp0;
        u->have_ref = u->have_threads = true;
        stage_inputs(u);
        u->stats.ms_stage = now_ms() - t0 - u->stats.ms_parse;      // (what the load took beyond the read alignments: staging, and whatever of the contig threading was not hidden behind them)
    });
}

int agx_unit_cache_build(const agx_params *p, const char *tmp_dir, int unit, const agx_reads *reads, char *err, size_t err_len) {
    if (err && err_len) err[0] = 0;
    if (!p || !tmp_dir) return AGX_E_ARG;
    agx_unit *u = nullptr;
    agx_params once; if (p) { once = *p; once.flags |= AGX_FLAG_ONE_SHOT; }      // loaded, built and thrown away here
    int rc = agx_unit_create(p ? &once : nullptr, &u);
    if (rc != AGX_OK) { if (err && err_len) snprintf(err, err_len, "%s", rc == AGX_E_NOGPU ? "no HIP device" : "bad parameters"); return rc; }
    rc = agx_unit_load_files_shared(u, tmp_dir, unit, reads);      // (takes a current cache file if there is one)
    if (rc == AGX_OK && !u->stats.from_cache) rc = guarded(u, [&] { save_cache(u, tmp_dir, unit); });
    if (rc != AGX_OK && err && err_len) snprintf(err, err_len, "%s", agx_unit_error(u));
    agx_unit_destroy(u);
    return rc;
}

int agx_unit_cache_save(agx_unit *u, const char *tmp_dir, int unit) { if (!u || !tmp_dir) return AGX_E_ARG; return guarded(u, [&] { save_cache(u, tmp_dir, unit); }); }

int agx_unit_hbm_needed(agx_unit *u, uint64_t *bytes) {
    if (!u || !bytes) return AGX_E_ARG;
    return guarded(u, [&] { if (!u->staged) stage_inputs(u); *bytes = round_block(plan_capacities(u).total); });
}

int agx_unit_stage(agx_unit *u) { if (!u) return AGX_E_ARG; return guarded(u, [&] { stage_inputs(u); }); }
int agx_unit_upload(agx_unit *u) { if (!u) return AGX_E_ARG; return guarded(u, [&] { do_upload(u); }); }
int agx_unit_release(agx_unit *u) { if (!u) return AGX_E_ARG; return guarded(u, [&] { do_release(u); }); }
int agx_unit_trim(agx_unit *u, uint64_t *freed) { if (!u) return AGX_E_ARG; if (freed) *freed = 0; return guarded(u, [&] { const size_t f = do_trim(u); if (freed) *freed = f; }); }
void agx_pool_trim(int device) { if (device >= 0) dev_trim(device); else if (device == -1) { host_trim(); scratch_trim(); out_cache_trim(); } else host_retire(); }      // (-1 also unmaps the loaders' cached scratch memory: up to 16 GB of touched pages per process otherwise stay until exit)
int agx_unit_build(agx_unit *u) { if (!u) return AGX_E_ARG; return guarded(u, [&] { do_build(u); }); }
int agx_unit_download(agx_unit *u) { if (!u) return AGX_E_ARG; return guarded(u, [&] { do_download(u); }); }

int agx_unit_finish(agx_unit *u, agx_result *r) {
    if (!u || !r) return AGX_E_ARG;
    memset(r, 0, sizeof *r);
    return guarded(u, [&] {
        // nothing downloaded yet: the download is streamed where it can be, and the walk begins on what has landed (begin_streamed_download)
        const bool streamed = !u->downloaded && begin_streamed_download(u);
        if (!u->downloaded && !streamed) do_download(u);
        struct Landed { agx_unit *u; ~Landed() { stream_wait_all(u); } } landed{u};      // (also if the walk throws: the copies write into buffers that are released afterwards)
        const double t0 = now_ms();
        join_helper(u);
        prepare_outputs(u);               // (a unit whose helper could not make them, or that is finished a second time)
        if (!u->out_ready) throw Error{E_ARG, "out of host memory"};
        u->downloaded = false;            // the walk marks the downloaded meta bytes: another finish downloads again
        u->out_ready = false;             // (an error below leaves the buffers to the next prepare_outputs)
        // Walkers for this walk: sixteen while few units are on the way (five at most) and at most two other walks are running, or nobody else is on the way — a job's tail: the
        // walkers of two or three walks on 16 CPUs fill each other's waits (cfg3: 33.2 -> 32.2 ms per job at +15 % CPU time); else eight, or what the walker threads already
        // running leave of this process's CPUs: a job with eight units in flight is bound by its CPU time (the whole-human job: 572 ms with this rule, 589 ms with sixteen walkers
        // whenever two walks or fewer were running).
        // (r04 gave sixteen to the last walk only; since r05 a job's builds end closer together than its walks take.)
        static std::atomic<int> walkers_busy{0}, walks_running{0};
        {   static const int ranks = [] { const char *e = getenv("LOCAL_WORLD_SIZE"); const int n = e ? atoi(e) : 1; return n < 1 ? 1 : n; }();
            const int free_cpus = (int)usable_cpus() / ranks - walkers_busy.load();
            agx::walkers_cap = (g_walks_pending.load() <= 1 || (g_walks_pending.load() <= 5 && walks_running.load() <= 2)) ? (int)GraphView::MAX_WALKERS : std::max(8, std::min((int)GraphView::MAX_WALKERS, free_cpus)); }
        struct Busy { std::atomic<int> &n, &w; int k; Busy(std::atomic<int> &c, std::atomic<int> &r, int walkers) : n(c), w(r), k(walkers) { n.fetch_add(k); w.fetch_add(1); } ~Busy() { n.fetch_sub(k); w.fetch_sub(1); } } busy{walkers_busy, walks_running, walkers_now(u->V.n_pos)};
        struct Walked { agx_unit *u; ~Walked() { if (u->pending_walk) { u->pending_walk = false; g_walks_pending.fetch_sub(1); } } } walked{u};
        struct Helpers : Assistant {           // helper 0: the unit's own thread (formats the written records while the walk goes on, or walks a stretch); 1..: pool threads for further walkers
            agx_unit *u; int pool[GraphView::MAX_WALKERS] = {}; int n_pool = 0;
            Helpers(agx_unit *x, int extra) : u(x) { for (int i = 0; i < extra && i < GraphView::MAX_WALKERS - 2; i++) { const int t = walker_pool().take(); if (t < 0) break; pool[n_pool++] = t; } }
            ~Helpers() override { for (int i = 0; i < n_pool; i++) { walker_pool().wait(pool[i]); walker_pool().give(pool[i]); } }
            int helpers() const override { return 1 + n_pool; }
            void run(std::function<void()> f, int who) override {
                if (who == 0) { if (!u->helper.submit(UnitHelper::WALK, std::move(f))) throw Error{E_ARG, "no helper thread"}; }
                else walker_pool().run(pool[who - 1], std::move(f));
            }
            void wait(int who) override { if (who == 0) u->helper.wait(UnitHelper::WALK); else walker_pool().wait(pool[who - 1]); }
        } second(u, walkers_now(u->V.n_pos) - 2);      // one thread per further walker: the unit's helper + pool threads
        GraphView G = view_of(u);
        if (streamed) { G.wait_landed = stream_wait_landed; G.wait_str = stream_wait_str; G.land_ctx = u; G.land_ms = std::max(0.01, u->dl_est_ms - (now_ms() - u->dl_t0)); }
        walk_join_scaffold(u->V, G, u->out, u->helper.started ? &second : nullptr);
        u->stats.ms_walk = now_ms() - t0; u->stats.n_fetched = u->out.n_fetched;
        trace(u, "walk", t0, u->V.n_pos);
        r->initial_len = u->out_initial.n; r->initial_contigs = u->out_initial.release();
        r->pre_len = u->out.pre_extended.n; r->pre_extended = u->out.pre_extended.release();
        r->extended_len = u->out.extended.n; r->extended = u->out.extended.release();
    });
}

void agx_result_free(agx_result *r) { if (!r) return; out_cache_give(r->initial_contigs); out_cache_give(r->pre_extended); out_cache_give(r->extended); memset(r, 0, sizeof *r); }      // (kept for the next unit's outputs: agx_host.h)

int agx_unit_stats(const agx_unit *u, agx_stats *s) {
    if (!u || !s) return AGX_E_ARG;
    *s = u->stats;
    const bool st = u->staged || u->consumed;      // (a one-shot unit's staged inputs are gone after its download; their counts are not)
    s->n_pos = st ? u->V.n_pos : u->T.ref.size(); s->n_ref = st ? u->V.n_ref : u->T.n_ref; s->n_hits = st ? u->nh : u->P.hits.size(); s->n_runs = st ? u->n_runs : u->P.runs.size(); s->pinned_bytes_cached = host_cache().held(); s->device_bytes_cached = dev_cache(u->prm.device).held(); s->n_nodes = u->n_nodes;
    s->n_tiles = u->n_tiles; s->n_tile_entries = u->n_tile_entries; s->n_big_tiles = u->n_big; s->n_mid_tiles = u->n_mid; s->n_edge_overflow = u->n_ovf;
    s->pairs_in_file = u->pairs_in_file; s->sam_line_pairs = u->sam_pairs;
    return AGX_OK;
}

int agx_unit_graph(agx_unit *u, agx_graph *g) {
    if (!u || !g) return AGX_E_ARG;
    memset(g, 0, sizeof *g);
    return guarded(u, [&] {
        if (!u->built) do_build(u);
        HIP_OK(hipSetDevice(u->prm.device));
        HIP_OK(wait_event(u->ev_built));
        // the pool has unused slots (one slice per region): the arrays come down whole, nodes are reached through node_start / node_cnt
        const agx_u32 n_pos = (agx_u32)u->V.n_pos, nn = u->n_nodes, cap = u->pool_cap;
        std::vector<agx_u32> node_start(n_pos), cid(cap), coff(cap), cid0(cap), coff0(cap), off0(cap), next((size_t)cap * AGX_MAXE); std::vector<agx_u16> node_cnt(n_pos);
        std::vector<agx_sref> sref(cap); std::vector<int> counts; std::vector<agx_edge_ovf> ovf(u->n_ovf);
        HIP_OK(hipMemcpy(node_start.data(), u->d_node_start.p, (size_t)n_pos * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(node_cnt.data(), u->d_node_cnt.p, (size_t)n_pos * 2, hipMemcpyDeviceToHost));
        if (cap) {
            HIP_OK(hipMemcpy(cid.data(), u->d_cid.p, (size_t)cap * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(coff.data(), u->d_coff.p, (size_t)cap * 4, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(cid0.data(), u->d_cid0.p, (size_t)cap * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(coff0.data(), u->d_coff0.p, (size_t)cap * 4, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(off0.data(), u->d_off0.p, (size_t)cap * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(next.data(), u->d_next.p, (size_t)cap * AGX_MAXE * 4, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(sref.data(), u->d_sref.p, (size_t)cap * sizeof(agx_sref), hipMemcpyDeviceToHost));
            if (u->prm.flags & AGX_FLAG_KEEP_COUNTS) { counts.resize((size_t)cap * 6); HIP_OK(hipMemcpy(counts.data(), u->d_counts.p, (size_t)cap * 24, hipMemcpyDeviceToHost)); }
        }
        if (u->n_ovf) HIP_OK(hipMemcpy(ovf.data(), u->d_ovf.p, (size_t)u->n_ovf * sizeof(agx_edge_ovf), hipMemcpyDeviceToHost));
        g->n_pos = n_pos; g->n_nodes = nn;
        g->node_start = (uint32_t *)malloc(4 * ((size_t)n_pos + 1)); g->node_key = (uint32_t *)malloc(24 * ((size_t)nn + 1)); g->node_cnt = (int32_t *)malloc(24 * ((size_t)nn + 1));
        g->node_slen = (uint32_t *)malloc(4 * ((size_t)nn + 1)); g->edge_start = (uint32_t *)malloc(4 * ((size_t)nn + 1));
        std::vector<agx_u32> canon(cap, AGX_NONE), slot_of(nn); agx_u32 id = 0;
        for (agx_u32 x = 0; x < n_pos; x++) {
            g->node_start[x] = id;
            for (agx_u32 v = 0; v < node_cnt[x]; v++) {
                if ((size_t)node_start[x] + v >= cap || id >= nn) throw Error{E_DEVICE, "node table is inconsistent (count mismatch)"};
                slot_of[id] = node_start[x] + v; canon[node_start[x] + v] = id++;
            }
        }
        g->node_start[n_pos] = id;
        if (id != nn) throw Error{E_DEVICE, "node table is inconsistent (count mismatch)"};
        std::vector<std::vector<agx_u32> > adj(nn);
        for (agx_u32 c = 0; c < nn; c++) for (agx_u32 e = 0; e < AGX_MAXE; e++) { const agx_u32 d = next[(size_t)slot_of[c] * AGX_MAXE + e]; if (d != AGX_NONE) adj[c].push_back(canon[d]); }
        for (agx_u32 i = 0; i < u->n_ovf; i++) adj[canon[ovf[i].src]].push_back(canon[ovf[i].dst]);
        size_t ne = 0;
        for (auto &a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); ne += a.size(); }
        g->n_edges = (uint32_t)ne; g->edge_dst = (uint32_t *)malloc(4 * (ne + 1));
        for (agx_u32 c = 0; c < nn; c++) {
            const agx_u32 v = slot_of[c]; uint32_t *kk = g->node_key + 6 * (size_t)c;
            kk[0] = cid[v]; kk[1] = coff[v]; kk[2] = cid0[v]; kk[3] = coff0[v]; kk[4] = off0[v] == AGX_NONE ? AGX_NONE : 0; kk[5] = off0[v];
            if (!counts.empty()) memcpy(g->node_cnt + 6 * (size_t)c, counts.data() + 6 * (size_t)v, 24); else for (int j = 0; j < 6; j++) g->node_cnt[6 * (size_t)c + j] = -1;
            g->node_slen[c] = (sref[v].qlen >> 16) & 0x7FFF;
        }
        size_t eo = 0;
        for (agx_u32 c = 0; c < nn; c++) { g->edge_start[c] = (uint32_t)eo; for (agx_u32 d : adj[c]) g->edge_dst[eo++] = d; }
        g->edge_start[nn] = (uint32_t)eo;
    });
}

void agx_graph_free(agx_graph *g) {
    if (!g) return;
    free(g->node_start); free(g->node_key); free(g->node_cnt); free(g->node_slen); free(g->edge_start); free(g->edge_dst); memset(g, 0, sizeof *g);
}

int agx_run_unit(const agx_params *p, const char *tmp_dir, int unit, int write_files, agx_result *r, char *err, size_t err_len) {
    return agx_run_unit_shared(p, tmp_dir, unit, write_files, nullptr, r, err, err_len);
}

int agx_run_unit_shared(const agx_params *p, const char *tmp_dir, int unit, int write_files, const agx_reads *reads, agx_result *r, char *err, size_t err_len) {
    if (err && err_len) err[0] = 0;
    agx_unit *u = nullptr;
    agx_params once; if (p) { once = *p; once.flags |= AGX_FLAG_ONE_SHOT; }      // created, loaded, uploaded once, finished and destroyed here: the download may land in the dead staged inputs
    int rc = agx_unit_create(p ? &once : nullptr, &u);
    if (rc != AGX_OK) { if (err && err_len) snprintf(err, err_len, "%s", rc == AGX_E_NOGPU ? "no HIP device" : "bad parameters"); return rc; }
    rc = agx_unit_load_files_shared(u, tmp_dir, unit, reads);
    if (rc == AGX_OK) rc = agx_unit_upload(u);
    if (rc == AGX_OK) rc = agx_unit_build(u);
    if (rc == AGX_OK) rc = agx_unit_finish(u, r);
    if (rc == AGX_OK && write_files)
        rc = guarded(u, [&] {
            const std::string d = tmp_dir, s = std::to_string(unit);
            write_file(d + "/_initial_contigs." + s + ".fa", std::string(r->initial_contigs, r->initial_len));
            write_file(d + "/_pre_extended_contigs." + s + ".fa", std::string(r->pre_extended, r->pre_len));
            write_file(d + "/_extended_contigs." + s + ".fa", std::string(r->extended, r->extended_len));
        });
    if (rc != AGX_OK && err && err_len) snprintf(err, err_len, "%s", agx_unit_error(u));
    if (rc != AGX_OK && r) agx_result_free(r);          // (a failed write of the three files leaves nothing allocated behind)
    agx_unit_destroy(u);
    return rc;
}

}  // extern "C"
