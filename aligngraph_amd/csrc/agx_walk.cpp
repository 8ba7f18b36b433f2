// agx_walk.cpp — coverage-pruned path walk, contig join and mate-guided scaffolding on the packed graph.
//
// Follows extdContigs1 (AG:1954-2204), extdContigs2 (AG:2296-2380) and scaffoldContigs (AG:2396-2464) of
// /root/reference/AlignGraph/AlignGraph.cpp.  The walk is sequential BY SPECIFICATION: every branch decision
// depends on which nodes earlier walks already consumed (AG:2027-2033, 2096-2113), the position scan skips ahead
// inside long records (AG:2194-2202) and a record is suppressed against the previously WRITTEN one (AG:2176).
// It therefore runs on the host over the flat node table the kernels produced: prune flags and consensus bases
// were fused into the node sweep, so the walk touches 1 byte of state per node plus its out-edge slots.  Large units are walked by
// several walkers whose stretches are checked against each other where they meet (walk_split): the result is the sequential walk's.
#include "agx_host.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <memory_resource>
#include <string>
#include <vector>
#include <unordered_map>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

// -DAGX_WALK_PROF: cycle counts per section of the walk, printed with AGX_WALK_TIMING (development aid, compiled out by default)
#if defined(AGX_WALK_PROF) && defined(__x86_64__)
#include <x86intrin.h>
static unsigned long long g_prof[12];
// (lfence: without it the out-of-order core reads the counter early and a section's cache misses are billed to a later one)
#define AGX_PT(i) do { _mm_lfence(); const unsigned long long t_ = __rdtsc(); g_prof[i] += t_ - tp_; tp_ = t_; } while (0)
#define AGX_PT_START _mm_lfence(); unsigned long long tp_ = __rdtsc()
#else
#define AGX_PT(i) do { } while (0)
#define AGX_PT_START do { } while (0)
#endif

namespace agx {
namespace {

// Bases of a record are kept as a list of byte ranges — runs of node bases in the downloaded walk graph, conti-mer chain suffixes, reference
// stretches, trailing k-mers — from the walk to the last output byte: most walks end up contained in the previously written record
// (AG:2176) and are dropped without ever being copied, and the join and the scaffolder concatenate lists instead of re-copying 30 MB
// of strings (their string appends were a quarter of a large unit's host time).  Every range outlives the outputs: the walk graph and
// the unit's tables belong to the caller, the k-mer tails to walk_join_scaffold's arena.
struct Seg { const char *p; size_t n; };
// (the lists live in walk_join_scaffold's monotonic arena: a few large blocks instead of one small allocation per record — a thread's
// malloc arena grows a page at a time, through mprotect, and that waits for the address-space lock whenever another unit is pinning memory)
typedef std::pmr::memory_resource Arena;
struct Bases {
    std::pmr::vector<Seg> segs; size_t len = 0;
    explicit Bases(Arena *a) : segs(a) {}
    void add(const char *p, size_t n) { if (n) { segs.push_back(Seg{p, n}); len += n; } }
    void add_from(const Bases &o, size_t from) {          // o's bases from index `from` on (std::string::append(o, from, npos))
        for (const Seg &g : o.segs) { if (from >= g.n) { from -= g.n; continue; } add(g.p + from, g.n - from); from = 0; }
    }
};
struct Rec {                 // Contig, AG:123-139
    int extended;
    agx_u32 sID, sOff, eID, eOff, sID0, sOff0, eID0, eOff0;
    Bases nuc;
    explicit Rec(Arena *a) : nuc(a) {}
};

inline char *put_u32(char *w, agx_u32 v) {      // decimal, no padding
    char t[10]; int n = 0;
    do { t[n++] = (char)('0' + v % 10u); v /= 10u; } while (v);
    while (n) *w++ = t[--n];
    return w;
}

// FASTA body, 60 columns (AG:2184-2188), written across range boundaries
struct LineWriter {
    char *w; unsigned col = 0;
    explicit LineWriter(char *at) : w(at) {}
    void put(const char *p, size_t n) {
        if (col) {                                   // finish the open line
            const size_t m = n < 60u - col ? n : 60u - col;
            memcpy(w, p, m); w += m; p += m; n -= m; col += (unsigned)m;
            if (col < 60u) return;
            *w++ = '\n'; col = 0;
        }
        // whole lines: a 64-byte move per line (its last four bytes are overwritten by the newline and the next line) instead of a
        // length-dispatching memcpy of 60 — half a million lines per 30 MB output
        for (; n >= 64; n -= 60, p += 60, w += 61) { memcpy(w, p, 64); w[60] = '\n'; }
        if (n >= 60) { memcpy(w, p, 60); w[60] = '\n'; w += 61; p += 60; n -= 60; }
        memcpy(w, p, n); w += n; col = (unsigned)n;
    }
    void put(const Bases &b) { for (const Seg &g : b.segs) put(g.p, g.n); }
    char *end() { if (col) { *w++ = '\n'; col = 0; } return w; }
};

// One written record of the walk as the pre-extended output needs it: the ten header numbers (AG:2180-2183) and the byte ranges of its bases as
// they were when it was written (a snapshot: the join appends to the record's list later, and a list that grows moves to new arena memory
// while the old stays valid).  Formatting them is the assistant thread's work while the walk goes on (PreFormat), or this thread's.
struct PreJob { agx_u32 f[10]; const Seg *segs; size_t n_segs, total; };
inline void format_pre(OutBuf &out, const PreJob &j) {
    char hdr[256];                        // ">%u, %d, %u, %u, %u, %u, %u, %u, %u, %u \n" without going through printf
    char *h = hdr; *h++ = '>';
    for (int i = 0; i < 10; i++) { h = put_u32(h, j.f[i]); if (i < 9) { *h++ = ','; *h++ = ' '; } }
    *h++ = ' '; *h++ = '\n';
    const size_t hl = (size_t)(h - hdr), lines = (j.total + 59) / 60;
    char *w = out.grow(hl + j.total + lines); memcpy(w, hdr, hl);
    LineWriter lw(w + hl); for (size_t i = 0; i < j.n_segs; i++) lw.put(j.segs[i].p, j.segs[i].n); lw.end();
}
// The records the walk publishes, formatted in order by the assistant while the walk is still going.  The job array never moves (records
// beyond its capacity are left to the walk's own thread, after the walk); `published` hands a job over, `finished` ends the assistant's loop.
struct PreFormat {
    OutBuf &out; Assistant *assistant;
    std::vector<PreJob> jobs, late; size_t cap = 0;
    std::atomic<size_t> published{0}; std::atomic<bool> finished{false}, failed{false};
    PreFormat(OutBuf &o, Assistant *a, size_t capacity) : out(o), assistant(a), cap(a ? capacity : 0) {
        if (!assistant) return;
        jobs.reserve(cap);
        assistant->run([this] {
            try {
                size_t at = 0; unsigned idle = 0;
                for (;;) {
                    const size_t n = published.load(std::memory_order_acquire);
                    if (at < n) { while (at < n) format_pre(out, jobs[at++]); idle = 0; continue; }
                    if (finished.load(std::memory_order_acquire) && at == published.load(std::memory_order_acquire)) return;
                    if (++idle < 64) {
#if defined(__x86_64__)
                        _mm_pause();
#endif
                    } else std::this_thread::sleep_for(std::chrono::microseconds(20));
                }
            } catch (...) { failed.store(true); }
        });
    }
    void add(const PreJob &j) {
        if (!assistant) { format_pre(out, j); return; }
        if (jobs.size() < cap) { jobs.push_back(j); published.store(jobs.size(), std::memory_order_release); }
        else late.push_back(j);
    }
    void finish() {                       // everything is in `out` when this returns
        if (!assistant) return;
        finished.store(true, std::memory_order_release);
        assistant->wait();
        if (failed.load()) throw Error{E_ARG, "out of host memory"};
        for (const PreJob &j : late) format_pre(out, j);
        late.clear();
    }
    ~PreFormat() { if (assistant && !finished.load()) { finished.store(true, std::memory_order_release); assistant->wait(); } }      // (the walk threw)
};

inline bool contains(agx_u32 sID1, agx_u32 sOff1, agx_u32 eID1, agx_u32 eOff1, agx_u32 sID2, agx_u32 sOff2, agx_u32 eID2, agx_u32 eOff2) {
    return sID1 == sID2 && eID1 == eID2 && sOff1 <= sOff2 && eOff1 >= eOff2;      // AG:1897-1902
}


// The walk's "traversed" flag of a node is bit 7 of its meta byte: the device sets it on ids without a node (AGX_WM_ABSENT: they count as
// visited), the walk sets it on the nodes it passes.  One byte array serves the run scan, the marks and the position scan.
#define AGX_WM_VISITED AGX_WM_ABSENT
// End of the forced run that starts at node `cur`: the first j >= cur whose cont bit is clear or whose successor j+1 is already
// visited (the reference steps cur -> cur+1 while its unique live successor is unvisited, AG:2020-2046).  `seen` collects the meta
// bits of the nodes passed.  meta is padded by 64 zero bytes, so whole-vector reads past j are safe (and stop there: no cont bit).
inline agx_u32 run_end_scalar(const agx_u8 *meta, agx_u32 j, agx_u8 &seen) {
    for (;;) {
        uint64_t mw, dw; memcpy(&mw, meta + j, 8); memcpy(&dw, meta + j + 1, 8);
        if ((mw & 0x0101010101010101ull) == 0x0101010101010101ull && (dw & 0x8080808080808080ull) == 0) { mw |= mw >> 32; mw |= mw >> 16; mw |= mw >> 8; seen |= (agx_u8)mw; j += 8; continue; }
        for (int b8 = 0; b8 < 8; b8++) { const agx_u8 m = meta[j]; seen |= m; if (!(m & AGX_WM_CONT) || (meta[j + 1] & AGX_WM_VISITED)) return j; j++; }
    }
}
// marks nodes [a, b] visited
inline void mark_scalar(agx_u8 *meta, agx_u32 a, agx_u32 b) { for (agx_u32 i = a; i <= b; i++) meta[i] |= AGX_WM_VISITED; }
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline agx_u32 run_end_avx2(const agx_u8 *meta, agx_u32 j, agx_u8 &seen) {
    unsigned contig = 0;                                                        // lanes (nodes passed) whose AGX_WM_CONTIG bit is set
    for (;;) {
        const __m256i m = _mm256_loadu_si256((const __m256i *)(meta + j)), d = _mm256_loadu_si256((const __m256i *)(meta + j + 1));
        // a node lets the run pass iff its cont bit (bit 0 -> the byte's sign bit) is set and its successor's visited bit (the sign bit) is clear
        const unsigned pass = (unsigned)_mm256_movemask_epi8(_mm256_slli_epi16(m, 7)) & ~(unsigned)_mm256_movemask_epi8(d);
        const unsigned stop = ~pass;
        const unsigned cbits = (unsigned)_mm256_movemask_epi8(_mm256_slli_epi16(m, 6));       // bit 1 of every byte -> its sign bit
        if (stop == 0) { contig |= cbits; j += 32; continue; }
        const unsigned k = (unsigned)__builtin_ctz(stop);                       // the run ends ON node j+k
        contig |= cbits & (k == 31 ? 0xFFFFFFFFu : ((2u << k) - 1u));
        // the walk only asks whether a node with a contig offset was passed (AG:2004): that is the one bit reported
        if (contig) seen |= AGX_WM_CONTIG;
        return j + k;
    }
}
__attribute__((target("avx2"))) inline void mark_avx2(agx_u8 *meta, agx_u32 a, agx_u32 b) {
    const agx_u32 n = b - a + 1;
    if (n >= 32) {                                   // whole vectors from the front, then one that ends on b (setting a bit twice is setting it)
        const __m256i v = _mm256_set1_epi8((char)AGX_WM_VISITED);
        for (agx_u32 i = a; i + 31 < b; i += 32) _mm256_storeu_si256((__m256i *)(meta + i), _mm256_or_si256(_mm256_loadu_si256((const __m256i *)(meta + i)), v));
        agx_u8 *t = meta + b - 31;
        _mm256_storeu_si256((__m256i *)t, _mm256_or_si256(_mm256_loadu_si256((const __m256i *)t), v));
    } else if (n >= 8) {                             // two overlapping 8-byte words per step
        agx_u32 i = a;
        for (; i + 7 < b; i += 8) { uint64_t w; memcpy(&w, meta + i, 8); w |= 0x8080808080808080ull; memcpy(meta + i, &w, 8); }
        uint64_t w; memcpy(&w, meta + b - 7, 8); w |= 0x8080808080808080ull; memcpy(meta + b - 7, &w, 8);
    } else for (agx_u32 i = a; i <= b; i++) meta[i] |= AGX_WM_VISITED;
}
#endif
// first unvisited index i in [from, n) (n if none)
inline agx_u32 next_zero_scalar(const agx_u8 *meta, agx_u32 from, agx_u32 n) {
    agx_u32 i = from;
    while (i < n && (i & 7u)) { if (!(meta[i] & AGX_WM_VISITED)) return i; i++; }
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, &meta[i], 8); if ((w & 0x8080808080808080ull) != 0x8080808080808080ull) break; }
    while (i < n && (meta[i] & AGX_WM_VISITED)) i++;
    return i < n ? i : n;
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline agx_u32 next_zero_avx2(const agx_u8 *meta, agx_u32 from, agx_u32 n) {
    agx_u32 i = from;
    for (; i + 32 <= n; i += 32) {
        const unsigned m = ~(unsigned)_mm256_movemask_epi8(_mm256_loadu_si256((const __m256i *)(meta + i)));
        if (m) return i + (agx_u32)__builtin_ctz(m);
    }
    while (i < n && (meta[i] & AGX_WM_VISITED)) i++;
    return i < n ? i : n;
}
#endif
typedef agx_u32 (*next_zero_fn)(const agx_u8 *, agx_u32, agx_u32);
inline next_zero_fn pick_next_zero() {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) return next_zero_avx2;
#endif
    return next_zero_scalar;
}
typedef agx_u32 (*run_end_fn)(const agx_u8 *, agx_u32, agx_u8 &);
typedef void (*mark_fn)(agx_u8 *, agx_u32, agx_u32);
inline mark_fn pick_mark() {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) return mark_avx2;
#endif
    return mark_scalar;
}
inline run_end_fn pick_run_end() {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) return run_end_avx2;
#endif
    return run_end_scalar;
}

#ifdef AGX_WALK_CHECK
#define AGX_WCHK(W_, v, where) (W_).wchk((v), (where))
#else
#define AGX_WCHK(W_, v, where) do { } while (0)
#endif
struct Walker {
    const UnitView &V; const GraphView &G;
    // meta bytes with the traversed flags in bit 7 (AGX_WM_VISITED): the caller's own array if it may be written (GraphView::meta_rw: the
    // engine's download buffer — no copy, no first-touch faults on the unit's critical path), else a copy
    agx_u8 *m = nullptr; agx_u8 *owned = nullptr;
    ~Walker() { free(owned); }
    bool visited(agx_u32 v) const { AGX_WCHK(*this, v, "visited"); return (m[v] & AGX_WM_VISITED) != 0; }
    // A speculative walker (walk_split below) may only look at nodes of positions [look_lo, look_hi): what it finds anywhere else is not
    // what the sequential walk would find there.  It gives up (invalid) the moment an edge or a conti-mer chain leads outside.  (The record fetch hook serves both walkers: it must be callable from two threads.)
    bool spec = false; agx_u32 look_lo = 0, look_hi = 0xFFFFFFFFu; mutable bool invalid = false; mutable agx_u32 gave_up_at = AGX_NONE;
#ifdef AGX_WALK_CHECK      // development aid (tests/hostsim builds it on request): every byte a decision rests on must lie in what the walker copied
    agx_u32 chk_lo = 0, chk_hi = 0xFFFFFFFFu, chk_slo = 0, chk_shi = 0xFFFFFFFFu;
    void wchk(agx_u32 v, const char *where) const {
        if (!spec) return;
        const bool ok = v < G.n_pos ? (v >= chk_lo && v < chk_hi) : (v >= chk_slo && v < chk_shi);
        if (!ok && !invalid) { fprintf(stderr, "[agx walk check] %s looks at id %u outside the window [%u, %u) + [%u, %u)\n", where, v, chk_lo, chk_hi, chk_slo, chk_shi); abort(); }
    }
#endif
    agx_u32 scan_hi = 0xFFFFFFFFu;               // (a walker behind a window: its visited bytes are only real below this position — a scan for the next unvisited node that ends at or behind it has told nothing)
    // The first walker of walk_split marks in the array the other walkers copy their windows from.  It used to wait until every window was taken (1.3-1.5 ms in front of every
    // large unit's walk, r05); now it walks at once and only a mark that reaches the first byte any window holds (guard_main / guard_side: the lowest window's start) waits for
    // the copies — by then they are long done.
    const std::atomic<int> *copies = nullptr; int copies_want = 0; agx_u32 guard_main = 0xFFFFFFFFu, guard_side = 0xFFFFFFFFu;
    void before_mark(agx_u32 last) {             // `last`: the highest id about to be marked (a run lies in the main block or in the side block)
        if (last < G.n_pos ? last < guard_main : last < guard_side) return;
        while (copies->load(std::memory_order_acquire) < copies_want) {
#if defined(__x86_64__)
            _mm_pause();
#endif
        }
        copies = nullptr;
    }
    bool may_look(agx_u32 v) const { if (!spec) return true; const agx_u32 x = pos_of(v); if (x >= look_lo && x < look_hi) return true; invalid = true; gave_up_at = x; return false; }
    std::vector<agx_edge_ovf> ovf;                  // sorted, unique
    Walker(const UnitView &v, const GraphView &g) : V(v), G(g) {
        if (g.meta_rw) { if (g.meta_rw != g.meta) throw Error{E_ARG, "meta_rw must be the meta array"}; m = g.meta_rw; }
        else {
            const size_t n = (size_t)g.n_ids + 64;
            owned = (agx_u8 *)malloc(n); if (!owned) throw Error{E_ARG, "out of host memory"};
            advise_huge(owned, n); memcpy(owned, g.meta, n); m = owned;
        }
        for (size_t i = 0; i < g.n_ovf; i++) if (g.ovf[i].src != AGX_NONE) ovf.push_back(g.ovf[i]);
        std::sort(ovf.begin(), ovf.end(), [](const agx_edge_ovf &a, const agx_edge_ovf &b) { return a.src != b.src ? a.src < b.src : a.dst < b.dst; });
        ovf.erase(std::unique(ovf.begin(), ovf.end(), [](const agx_edge_ovf &a, const agx_edge_ovf &b) { return a.src == b.src && a.dst == b.dst; }), ovf.end());
    }
    // record of node v: from the sparse table of special ids, else (a walk started inside a forced run after the +1000 skip,
    // agx_core.h) from the full table through the fetch hook
    mutable unsigned long long n_fetched = 0;
    mutable std::unordered_map<agx_u32, agx_walknode> extra;      // records fetched so far (a few per 1000 positions of long records)
    mutable std::vector<agx_walknode> rows;
    mutable agx_u32 last_v = AGX_NONE, last_at = AGX_NONE;      // a walk asks for the same id several times (start, successors, hop entry, end)
    agx_u32 rank_of(agx_u32 v) const {              // index of v in the sparse table, NONE if v is not a special id
        if (v == last_v) return last_at;
        const unsigned long long w = G.sp_bits[v >> 6], bit = 1ull << (v & 63u);
        last_v = v;
        return last_at = (w & bit) ? G.sp_rank[v >> 6] + (agx_u32)__builtin_popcountll(w & (bit - 1)) : AGX_NONE;
    }
    // hop entry of node v at position x: next to v's record in the sparse table (one cache line away from what the walk just read)
    // instead of a random access into the per-position table
    agx_hop hop_of(agx_u32 v, agx_u32 x) const {
        const agx_u32 r = G.sp_hop ? rank_of(v) : AGX_NONE;
        if (r != AGX_NONE) return G.sp_hop[r];
        if (V.hop) return V.hop[x];
        // no per-position table (units from the fast loader or a cache file): the entry from the conti-mer runs, as the device computes it for the special ids
        agx_hop h{0, 0, 0};
        if (!V.segs || V.n_seg0 == 0 || V.cm_count(x) != 1) return h;
        agx_u32 lo = 0, hi = V.n_seg0;                          // last rank-0 run with pos0 <= x
        while (hi - lo > 1) { const agx_u32 mid = lo + (hi - lo) / 2; if (V.segs[mid].pos0 <= x) lo = mid; else hi = mid; }
        const agx_cmseg &g = V.segs[lo];
        const agx_u32 j = x - g.pos0;
        if (x < g.pos0 || j >= g.len || j >= g.hop_len0) return h;
        h.str_off = g.hop_str0 + j; h.len = g.hop_len0 - j; h.end_pos = g.hop_end;
        return h;
    }
    agx_walknode node(agx_u32 v) const {
        const agx_u32 at = rank_of(v);
        if (at != AGX_NONE) {
            // the walk asks for records in nearly increasing rank order (two streams: main ids and side ids), and for the hop entry of the same
            // rank a few hundred cycles later: ask for that line now, and for the records and hop entries a few ranks ahead
            if (G.sp_hop) { __builtin_prefetch(G.sp_hop + at); __builtin_prefetch(G.sp_hop + at + 8); }
            __builtin_prefetch(G.sp_node + at + 6);
            return G.sp_node[at];
        }
        const auto it = extra.find(v);
        if (it != extra.end()) return it->second;
        if (!G.fetch) throw Error{E_ARG, "walk graph without a record fetch hook"};
        agx_walknode r; G.fetch(G.fetch_ctx, v, 1, 1, 1, &r); n_fetched++;
        extra.emplace(v, r);
        return r;
    }
    // A record longer than 100 kb was just written while the scan stands at cp: the scan will now sample every 1000th position up to
    // its end (AG:2194-2202) and may start walks in the middle of forced runs there.  Those main ids, and the ids right before them
    // (where a later run stops in front of them), are not in the sparse table: get them in one strided copy.
    void prefetch_skip_positions(agx_u32 cp, agx_u32 end) const {
        if (!G.fetch || cp + 1000 >= end) return;
        const agx_u32 n = (end - cp - 1) / 1000;                   // cp + 1000*i < end for i = 1..n
        rows.resize((size_t)n * 2);
        G.fetch(G.fetch_ctx, cp + 999, 1000, n, 2, rows.data());
        for (agx_u32 i = 0; i < n; i++) { extra.emplace(cp + 999 + i * 1000, rows[2 * (size_t)i]); extra.emplace(cp + 1000 + i * 1000, rows[2 * (size_t)i + 1]); }
    }
    agx_u32 pos_of(agx_u32 v) const { return v < G.n_pos ? v : G.side_xpos[v - G.n_pos]; }      // main ids are positions
    // side ids of position x: [lo, hi)
    void side_range(agx_u32 x, agx_u32 &lo, agx_u32 &hi) const {
        AGX_WCHK(*this, x, "side_range");
        if (!(m[x] & AGX_WM_SIDE)) { lo = hi = G.n_pos; return; }
        const agx_u32 *b = G.side_xpos, *e = G.side_xpos + (G.n_ids - G.n_pos);
        const agx_u32 *l = std::lower_bound(b, e, x), *h = l;
        while (h < e && *h == x) h++;
        lo = G.n_pos + (agx_u32)(l - b); hi = G.n_pos + (agx_u32)(h - b);
    }
    // number of live (unvisited) successors of node v, capped at 2; target = the last one seen (AG:2020-2033)
    int live_successors(agx_u32 v, agx_u32 &target) const {
        AGX_WCHK(*this, v, "live_successors");
        if (m[v] & AGX_WM_CONT) { if (spec && v >= G.n_pos && !may_look(v + 1)) return 2; if (visited(v + 1)) return 0; target = v + 1; return 1; }      // its only alive successor is v+1
        int n = 0;
        const agx_walknode rec = node(v);
        const agx_u32 *s = rec.next;
        for (agx_u32 e = 0; e < AGX_MAXE && s[e] != AGX_NONE; e++) { if (!may_look(s[e])) return 2; if (!visited(s[e])) { target = s[e]; if (++n > 1) return n; } }
        if (!ovf.empty()) {                             // nodes with more than AGX_MAXE out-edges (rare): the rest is in the overflow list
            auto it = std::lower_bound(ovf.begin(), ovf.end(), v, [](const agx_edge_ovf &a, agx_u32 key) { return a.src < key; });
            for (; it != ovf.end() && it->src == v; ++it) {
                bool inl = false; for (agx_u32 e = 0; e < AGX_MAXE; e++) inl |= s[e] == it->dst;
                if (!inl) { if (!may_look(it->dst)) return 2; if (!visited(it->dst)) { target = it->dst; if (++n > 1) return n; } }
            }
        }
        return n;
    }
    // k-mer string of node v from its read reference (agx_sref)
    void kmer_string(agx_u32 v, std::string &out) const {
        agx_sref r = node(v).sref;
        if (V.slot_row) r.slot = V.slot_row[r.slot];      // (tile-ordered upload: the device's row numbers are places in the tile order)
        const agx_u32 first = r.qlen & 0xFFFFu, len = (r.qlen >> 16) & 0x7FFFu; const bool rev = (r.qlen >> 31) != 0;
        out.clear();
        if (!V.bases) {      // the staged 2-bit rows are all there is (agx_host.h: UnitView::codes2): classes back to letters, the listed other bases by their bytes
            if (!V.codes2) throw Error{E_ARG, "no read bases for the k-mer strings"};
            const agx_u8 *row = V.codes2 + (size_t)r.slot * (V.stride / 4);
            const unsigned long long base = (unsigned long long)r.slot * V.stride;
            for (agx_u32 i = 0; i < len; i++) {
                const agx_u32 j = rev ? first - i : first + i;
                char c = "ACGT"[(row[j >> 2] >> (2u * (j & 3u))) & 3u];
                if (c == 'A' && V.n_other) {         // (class 0 is also what a listed base was packed as)
                    const unsigned long long *e = V.other_idx + V.n_other, *at = std::lower_bound(V.other_idx, e, base + j);
                    if (at != e && *at == base + j) c = (char)V.other_byte[at - V.other_idx];
                }
                out.push_back(!rev ? c : c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c);
            }
            return;
        }
        const char *p = V.row_off ? V.bases + V.row_off[r.slot] : V.bases + (size_t)(G.row_slot ? G.row_slot[r.slot] : r.slot) * V.stride;
        for (agx_u32 i = 0; i < len; i++) {
            if (!rev) out.push_back(p[first + i]);
            else { const char c = p[first - i]; out.push_back(c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c); }
        }
    }
    // first unvisited node with id in [from, n) (n if none): the position scan of AG:1972-1978 in walk-id space
    next_zero_fn next_zero = pick_next_zero();
    agx_u32 next_live(agx_u32 from, agx_u32 n) const { return from < n ? next_zero(m, from, n) : n; }
};

// extdContigs1, AG:1954-2204, replayed on the alive-compacted graph.  Alive ids are position-major, so "for every
// position, for every variant, if untraversed" (AG:1972-1978) is "for every alive id in order, if not done".
//
// Everything the position scan carries from one position to the next (the function-scope variables of the reference) is in WalkState, so
// the scan can be stopped at a position and taken up again — by the same walker, or by another one that arrives at the same state (walk_split).
struct WalkState {
    agx_u32 cp = 0;                              // the position the scan stands at
    agx_u32 seqID = 0, sIDBak = AGX_NONE, sOffBak = AGX_NONE, eIDBak = AGX_NONE, eOffBak = AGX_NONE;      // records written so far; the last one written (AG:2176)
    agx_u32 pos_bak = 0;                         // cppBak of the reference (function scope)
    bool same_scan(const WalkState &o) const { return cp == o.cp && sIDBak == o.sIDBak && sOffBak == o.sOffBak && eIDBak == o.eIDBak && eOffBak == o.eOffBak && pos_bak == o.pos_bak; }
};
typedef std::vector<std::pair<agx_u32, agx_u32>> MarkLog;      // id ranges [a, b] marked visited

struct WalkRun {
    Walker &W; Arena *arena;
    WalkState st;
    std::vector<Rec> &written; PreFormat *pre;   // pre: format the written records as they come; else their jobs are kept (jobs) for later
    std::vector<PreJob> jobs;
    bool keep = true;                            // false: a warm-up — records are decided and remembered as "the last one written", not kept
    MarkLog *log = nullptr; agx_u32 log_main = 0, log_side = 0;      // log the marks that reach main ids >= log_main or side ids >= log_side
    const std::atomic<bool> *cancel = nullptr;
    unsigned long long n_walks = 0, n_hops = 0, n_runs = 0, n_general = 0, run_nodes = 0;
    WalkRun(Walker &w, Arena *a, std::vector<Rec> &out, PreFormat *p) : W(w), arena(a), written(out), pre(p) {}
    void note_marks(agx_u32 a, agx_u32 b) {      // (a run lies in the main block or in the side block)
        if (a < W.G.n_pos) { if (b >= log_main) log->push_back({a < log_main ? log_main : a, b}); }
        else if (b >= log_side) log->push_back({a < log_side ? log_side : a, b});
    }
    // the scan from st.cp until it stands at a position >= stop (or the walker gave up / was cancelled): st is where it stands then
    void go(agx_u32 stop) {
        const GraphView &G = W.G; const UnitView &V = W.V;
        agx_u32 &seqID = st.seqID, &sIDBak = st.sIDBak, &sOffBak = st.sOffBak, &eIDBak = st.eIDBak, &eOffBak = st.eOffBak, &pos_bak = st.pos_bak;
        std::string kmer; agx_u32 klen = 0, klast = 0;
        agx_hop hcur{0, 0, 0};                       // hop entry of the position the walk is about to leave the k-mer graph at
        std::vector<Seg> segs;
        agx_u8 *const m = W.m;
        auto done = [m, this](agx_u32 v) { AGX_WCHK(W, v, "done"); (void)this; return (m[v] & AGX_WM_VISITED) != 0; };
        const mark_fn mark = pick_mark();
        AGX_PT_START;
        const run_end_fn run_end = pick_run_end();
        const agx_u32 n_side = G.n_ids - G.n_pos;
        if (stop > G.n_pos) stop = G.n_pos;
        agx_u32 sc = (agx_u32)(std::lower_bound(G.side_xpos, G.side_xpos + n_side, st.cp) - G.side_xpos);      // side index cursor: every side id before it lies at a position < cp
        agx_u32 side_live = G.n_pos + sc, main_live = st.cp;  // cached "first unvisited side / main id" of the position scan (see below; recomputed when stale)
        agx_u32 cp = st.cp;
        for (; cp < stop;) {
            if (W.invalid || (cancel && cancel->load(std::memory_order_relaxed))) break;
            // variants of position cp in order: its main slot, then its side range
            while (sc < n_side && G.side_xpos[sc] < cp) sc++;
            agx_u32 sh = sc; while (sh < n_side && G.side_xpos[sh] == cp) sh++;
            const agx_u32 s_lo = G.n_pos + sc, s_hi = G.n_pos + sh;
            for (agx_u32 vi = 0, start = cp; vi <= s_hi - s_lo; vi++, start = s_lo + vi - 1) {
                if (done(start)) continue;
                AGX_PT(8);
                Rec C(arena); C.sID = 0; C.sOff = cp; C.extended = 0;
                segs.clear(); n_walks++;
                agx_u32 cur = start;                 // current k-mer node (mode 1)
                // (the start node's mate offset — C.sID0 / C.sOff0, AG:1988 — is looked up when the record turns out to be written: thirteen walks in fourteen are dropped
                // below, AG:2176, and the look-up into the sparse table was one of a dropped walk's two)
                AGX_PT(0);
                agx_u32 cpp = cp; int mode = 1;               // mode = kMerTag
                agx_u32 last = cur;
                while ((mode == 1 && !done(cur)) || mode == 0) {
                    if (mode == 0) {                            // on a conti-mer, AG:2061-2138
                        // the whole conti-mer chain in one segment (the reference steps through it one base at a time), then its end:
                        // hop back onto the k-mer graph only through the single live node there and its single live edge (AG:2093-2136)
                        AGX_PT(7); const agx_hop h = hcur;
                        segs.push_back(Seg{V.chain_str + h.str_off, h.len}); C.extended = 1; n_hops++;
                        pos_bak = h.end_pos; cpp = h.end_pos;
                        if (!W.may_look(cpp)) { mode = -2; break; }      // (a speculative walker: the chain lands outside what it may look at)
                        agx_u32 live = 0, item = 0;
                        if (!done(cpp)) { live++; item = cpp; }
                        agx_u32 h_lo, h_hi; W.side_range(cpp, h_lo, h_hi);
                        for (agx_u32 v = h_lo; v < h_hi; v++) if (!done(v)) { live++; item = v; }
                        agx_u32 tgt = 0; int ns = 0;
                        if (live == 1) ns = W.live_successors(item, tgt);
                        if (ns == 1) { cur = tgt; pos_bak = W.pos_of(tgt); cpp = pos_bak; mode = done(cur) ? -2 : 1; }
                        else mode = -2;
                        AGX_PT(1);
                    } else {                                    // on a k-mer node, AG:1995-2060
                        // forced run: while the cont bit holds and the next node is unvisited the reference steps cur -> cur+1 (its unique live
                        // successor).
                        agx_u8 seen = 0;
                        AGX_WCHK(W, cur, "run start");
                        const agx_u32 j = run_end(m, cur, seen);
                        const agx_u32 xj = j < G.n_ids ? W.pos_of(j) : 0xFFFFFFFFu;      // (a run cannot leave the table: the bytes behind it are zero, and a walker's window ends in a byte that stops it)
                        // (a run over side ids ends on what the NEXT side id's byte says, and that id may lie many positions further)
                        if (W.spec && (xj < W.look_lo || xj >= W.look_hi || (j >= G.n_pos && j + 1 < G.n_ids && W.pos_of(j + 1) >= W.look_hi))) { W.invalid = true; W.gave_up_at = xj; mode = -2; break; }      // (the run left what this walker may look at: nothing is marked — the bytes out there may be another walker's)
                        AGX_PT(2);                             // most walks leave the k-mer graph here, onto a conti-mer chain
                        AGX_WCHK(W, j, "run end"); if (j + 1 < G.n_ids && (j + 1 < G.n_pos) == (j < G.n_pos)) AGX_WCHK(W, j + 1, "behind the run");
                        segs.push_back(Seg{G.str + cur, (size_t)j - cur + 1}); n_runs++; run_nodes += j - cur + 1; if (!(m[j] & AGX_WM_CONT)) n_general++;
                        if (seen & AGX_WM_CONTIG) C.extended = 1;
                        if (W.copies) W.before_mark(j);
                        mark(m, cur, j);
                        if (log && j >= log_main) note_marks(cur, j);
                        if (j > cur) pos_bak = xj;
                        cur = j; last = j; cpp = xj;
                        AGX_PT(3);
                        agx_u32 tgt = 0;
                        const int ns = (m[cur] & AGX_WM_CONT) ? 0 : W.live_successors(cur, tgt);      // cont && stopped: its only alive successor is already visited
                        if (ns == 1) { cur = tgt; pos_bak = W.pos_of(tgt); cpp = pos_bak; }
                        else if ((hcur = W.hop_of(j, xj)).len) mode = 0;            // exactly one conti-mer here and it has a next (AG:2047-2057)
                        else mode = -1;
                        AGX_PT(4);
                    }
                }
                // end bookkeeping, AG:2142-2173
                C.eID = 0; C.eOff = mode == 1 ? pos_bak : cpp;
                if (mode == 1 || mode == -1) {
                    { const agx_u32 o = W.node(cur).off0; C.eID0 = o == AGX_NONE ? AGX_NONE : 0; C.eOff0 = o; }
                    // the record ends with the last node's k-mer string minus its first base; only its LENGTH matters until the record is known
                    // to be written (most are dropped below), so the read bases are not touched yet
                    klen = (W.node(last).sref.qlen >> 16) & 0x7FFFu; klast = last;
                    C.eOff = C.eOff + klen - 1; C.eOff0 = C.eOff0 + klen - 1;
                } else { C.eID0 = AGX_NONE; C.eOff0 = AGX_NONE; klen = 0; }
                AGX_PT(5);
                if (!contains(sIDBak, sOffBak, eIDBak, eOffBak, C.sID, C.sOff, C.eID, C.eOff)) {        // AG:2176-2189
                    { const agx_u32 o = W.node(start).off0; C.sID0 = o == AGX_NONE ? AGX_NONE : 0; C.sOff0 = o; }
                    if (klen > 1) {                  // the record's trailing k-mer: its bytes go to the arena (every other range lives in the caller's tables)
                        W.kmer_string(klast, kmer);
                        char *t = (char *)arena->allocate(kmer.size() - 1, 1); memcpy(t, kmer.data() + 1, kmer.size() - 1);
                        segs.push_back(Seg{t, kmer.size() - 1});
                    }
                    C.nuc.segs.reserve(segs.size());
                    for (const Seg &g : segs) C.nuc.add(g.p, g.n);
                    sIDBak = C.sID; sOffBak = C.sOff; eIDBak = C.eID; eOffBak = C.eOff;
                    if (eOffBak - sOffBak > 100000) W.prefetch_skip_positions(cp, eOffBak < G.n_pos ? eOffBak : G.n_pos);
                    if (keep) {
                        PreJob j; const agx_u32 f[10] = {seqID, (agx_u32)C.extended, C.sID, C.sOff, C.eID, C.eOff, C.sID0, C.sOff0, C.eID0, C.eOff0};
                        memcpy(j.f, f, sizeof f); j.segs = C.nuc.segs.data(); j.n_segs = C.nuc.segs.size(); j.total = C.nuc.len;
                        if (pre) pre->add(j); else jobs.push_back(j);
                        written.push_back(std::move(C));
                    }
                    seqID++;
                }
                AGX_PT(6);
            }
            AGX_PT(9);
            // AG:2194-2202: inside a written record longer than 100 kb the scan jumps 1000 positions at a time; otherwise it moves to the
            // next position — and positions without an unvisited node do nothing, so jump straight to the next unvisited node's position
            if (eOffBak - sOffBak > 100000 && eIDBak == 0 && cp + 1000 < eOffBak) cp += 1000;
            else {
                // the +1000 rule is re-evaluated at every position on the way, but it can only switch ON again after a new record is
                // written, which needs an unvisited node: skipping node-less positions one by one or at once is the same
                // first unvisited main id (= position) after cp and first unvisited side id from s_hi on.  Visited nodes stay visited and both
                // bounds only grow, so a previous answer is still the answer unless it has been passed or visited since: each block is scanned
                // once over the whole walk, not once per step (a long record leaves thousands of side nodes behind, each a step of its own)
                // (behind a walker's window these two look at bytes that are not the table's: whatever they answer lies at or behind scan_hi, and is refused below)
                auto stale = [m](agx_u32 v) { return (m[v] & AGX_WM_VISITED) != 0; };
                if (main_live <= cp || (main_live < G.n_pos && stale(main_live))) main_live = W.next_live(main_live > cp + 1 ? main_live : cp + 1, G.n_pos);
                const agx_u32 m = main_live;                                                     // main slot id == position
                if (side_live < s_hi || (side_live < G.n_ids && stale(side_live))) side_live = W.next_live(side_live > s_hi ? side_live : s_hi, G.n_ids);
                const agx_u32 sd = side_live;
                const agx_u32 sp = sd < G.n_ids ? G.side_xpos[sd - G.n_pos] : G.n_pos;
                cp = m < sp ? m : sp;
                // (a walker behind a window: no unvisited node left inside it — where the scan stands next is not in its bytes.  Whichever of the two answers lies
                // inside the window is true: the scan met it before it left the real bytes)
                if (cp >= W.scan_hi) { W.invalid = true; W.gave_up_at = cp; }
            }
            AGX_PT(10);
    }
        st.cp = cp;
    }
};

inline void walk_report(const WalkRun &r) {
#if defined(AGX_WALK_PROF) && defined(__x86_64__)
    if (getenv("AGX_WALK_TIMING")) {
        fprintf(stderr, "[agx walk] Mcycles: start %.1f, hop %.1f, run scan %.1f, mark %.1f, successors %.1f, end %.1f, contain+write %.1f, between %.1f, position scan %.1f (+ %.1f after the last variant, %.1f next position)\n",
                g_prof[0] / 1e6, g_prof[1] / 1e6, g_prof[2] / 1e6, g_prof[3] / 1e6, g_prof[4] / 1e6, g_prof[5] / 1e6, g_prof[6] / 1e6, g_prof[7] / 1e6, g_prof[8] / 1e6, g_prof[9] / 1e6, g_prof[10] / 1e6);
        memset(g_prof, 0, sizeof g_prof);
    }
#endif
    if (getenv("AGX_WALK_TIMING")) fprintf(stderr, "[agx walk] walks %llu, contig hops %llu, runs %llu (%llu nodes), general evaluations %llu\n", r.n_walks, r.n_hops, r.n_runs, r.run_nodes, r.n_general);
}

// The whole scan by one walker (the written records are formatted by the assistant while it goes on, if there is one).
void walk(Walker &W, OutBuf &pre_out, std::vector<Rec> &written, Arena *arena, Assistant *assistant) {
    if (W.G.wait_landed) W.G.wait_landed(W.G.land_ctx, W.G.n_pos, W.G.n_ids);      // (a streamed download: one walker looks everywhere, and its records are formatted as they come)
    if (W.G.wait_str) W.G.wait_str(W.G.land_ctx);
    pre_out.reserve((size_t)W.G.n_ids + W.G.n_ids / 32 + 4096);
    PreFormat pre(pre_out, assistant, (size_t)W.G.n_pos / 256 + 65536);
    WalkRun run(W, arena, written, &pre);
    run.go(W.G.n_pos);
    pre.finish();
    walk_report(run);
}

// ---- the scan by several walkers ---------------------------------------------------------------------------------------------------
// The walk is sequential by specification, but what it carries from position to position is small (WalkState) and what a walk touches is
// near where it starts: records end at the next branch, a conti-mer chain lands a contig's length further.  So a second walker B starts a
// warm-up stretch before the middle position c on a pristine copy of the visited bytes, and when the first walker A — the sequential walk
// itself — arrives at c, the two states are compared: where the scan stands, the last record written, and every node at or behind c that
// either has visited.  Equal states have equal futures: B's records from c on ARE the sequential walk's, provided B never looked at a node
// in front of c or in the appended positions behind the reference (which A may have visited and B not) — B checks that as it goes and
// gives up if an edge or a conti-mer chain leads there.  If anything differs A simply walks on through B's half: the
// result is the sequential walk's either way, only the time differs.  The appended positions are walked last, by A, on the merged marks.
// positions a walker's stretch covers per millisecond (measured by every split walk; the first guess is a 2 GHz server core's)
inline std::atomic<double> &walker_speed() { static std::atomic<double> v{700000.0}; return v; }
inline void clip_marks(const MarkLog &log, agx_u32 main_lo, agx_u32 main_hi, agx_u32 n_pos, agx_u32 side_lo, agx_u32 side_hi, MarkLog &out) {
    out.clear();
    for (const auto &r : log) {
        agx_u32 a = r.first, b = r.second;
        if (b < n_pos) { if (a < main_lo) a = main_lo; if (b >= main_hi) { if (main_hi == 0) continue; b = main_hi - 1; } }
        else { if (a < side_lo) a = side_lo; if (b >= side_hi) { if (side_hi == 0) continue; b = side_hi - 1; } }
        if (a <= b) out.push_back({a, b});
    }
    std::sort(out.begin(), out.end());
    size_t k = 0;
    for (size_t i = 0; i < out.size(); i++) {
        if (k && out[i].first <= out[k - 1].second + 1) { if (out[i].second > out[k - 1].second) out[k - 1].second = out[i].second; }
        else out[k++] = out[i];
    }
    out.resize(k);
}

// One speculative walker: its own visited bytes (a window of the meta bytes, copied while they are pristine), its own arena, its stretch [c, c_next) of the reference
struct SpecWalker {
    agx_u32 c = 0, c_next = 0, w0 = 0, side_c = 0, side_next = 0, warm_lo = 0, warm_hi = 0xFFFFFFFFu, look_hi = 0, win_lo = 0, win_hi = 0, side_win_hi = 0, copy_hi = 0, side_copy_lo = 0, side_copy_hi = 0;
    Scratch bytes;                               // its visited bytes: as long as the table, only the window filled in
    GraphView G; std::unique_ptr<Walker> W; std::unique_ptr<WalkRun> R; std::vector<Rec> recs;
    MarkLog log; size_t n_warm = 0;              // every mark of the warm-up, then the marks of the stretch that reach c_next or further
    WalkState at_c;                              // where it stood when it arrived at c
    bool ok = false; std::string error;
    double t_run = 0, t_copy = 0, t_warm = 0, t_done = 0;     // (AGX_WALK_TIMING) when its thread began, had its window, arrived at c, arrived at c_next
};

// false: not split (too small, no assistant): the caller walks the usual way
bool walk_split(Walker &WA, const UnitView &V, const GraphView &G, OutBuf &pre_out, std::vector<Rec> &written, Arena *arena, Arena *const *more_arenas, Assistant *assistant) {
    const agx_u32 min_ref = getenv("AGX_WALK_SPLIT_MIN") ? (agx_u32)strtoul(getenv("AGX_WALK_SPLIT_MIN"), nullptr, 10) : 4000000u;
    // The warm-up in front of a walker's stretch must hold every walk that can reach into the stretch: walks are local, and what reaches furthest is a conti-mer chain, which lands
    // where its contig's placement ends.  So: three times the longest reach of a chain in this unit (+ a margin), 100 k to 400 k positions (r03/r04: 400 k whatever the unit; a
    // warm-up goes at half the speed of a stretch and the first walker's stretch is two warm-ups longer than the others': a 30 Mb unit's eight walkers spent a fifth of the
    // walk's CPU time warming up).  Too short a warm-up is never wrong: the stretch then does not stand and the first walker walks it.
    agx_u32 warm = 400000u;
    if (const char *e = getenv("AGX_WALK_SPLIT_WARMUP")) warm = (agx_u32)strtoul(e, nullptr, 10);
    else if (V.segs) {
        agx_u32 reach = 0;
        for (agx_u32 i = 0; i < V.n_seg0; i++) { const agx_cmseg &g = V.segs[i]; if (g.hop_len0 && g.hop_end > g.pos0 && g.hop_end - g.pos0 > reach) reach = g.hop_end - g.pos0; }
        const unsigned long long w = 3ull * reach + 20000ull;
        warm = (agx_u32)(w < 100000ull ? 100000ull : w > 400000ull ? 400000ull : w);
    }
    const agx_u32 n_ref = V.n_ref < G.n_pos ? V.n_ref : G.n_pos;
    if (!assistant || n_ref < min_ref || n_ref < 16 || getenv("AGX_WALK_NO_SPLIT")) return false;
    // walkers: one per 1.2 M positions, two to eight (sixteen on request: walkers_for), as many as there are helper threads.  Every further walker works on visited bytes of its own: a
    // WINDOW of the meta bytes that it copies itself, before anybody walks, from the first walker's still pristine array — two warm-ups (one stretch at
    // most) back from its stretch, the stretch, one stretch ahead — and it may only look at nodes inside the window (walks are local: a record ends
    // at the next branch, a conti-mer chain lands a contig's length further; one that does lead further makes the walker give up, as a look in front of
    // its stretch always did).  r02/r03a downloaded one whole copy of the meta bytes per walker instead: 0.4-0.6 ms of PCIe each in front of the walk,
    // 37 % of a unit's download with three of them; a window is 3/K of the bytes, copied by K - 1 threads side by side in 0.2-0.4 ms.
    int K = walkers_for(n_ref);
    if (K > 1 + assistant->helpers()) K = 1 + assistant->helpers();
    const agx_u32 slack = 64;                           // a walker reads a few bytes past the node it stands on (the end of a run, the cont successor)
    if (K > (int)(n_ref / (8 * slack))) K = (int)(n_ref / (8 * slack));
    if (K < 2) return false;
    // the first walker has no window to copy and no warm-up to walk (which goes at half the speed of a stretch: cold memory, every mark logged): its stretch is
    // longer by two warm-ups, so that all arrive at about the same time
    const agx_u32 lead = n_ref / (unsigned)K > 4 * warm ? 2 * warm : 0;
    std::vector<agx_u32> cuts((size_t)K + 1, n_ref); cuts[0] = 0;
    for (int i = 1; i < K; i++) cuts[(size_t)i] = lead + (agx_u32)((unsigned long long)(n_ref - lead) * (unsigned)i / (unsigned)K);
    // A streamed download (GraphView::wait_landed): the table arrives from the front while the walk goes on.  Walker i begins when its window is in place — the part of the
    // download in front of the window's end, then its copy and its warm-up —, the first walker, whose walks may lead anywhere, when everything is.  The stretches are cut so that
    // all of them END together: len_i = speed * (T - ready_i) with T from sum(len) = n_ref; the window's end depends on the cuts, so a few rounds from the even cuts.  Any cuts
    // give the sequential walk's result (that is what the meeting points check): the model only decides who waits for whom.
    // The same model cuts the stretches of a download that is complete (D = 0; r05: even cuts with the first walker's stretch two warm-ups longer): the first walker has no window to
    // copy and no warm-up to walk — the others finished 0.5-0.8 ms behind it once it no longer waited for their copies.
    const bool streamed = G.wait_landed != nullptr && G.land_ms > 0;
    if (!getenv("AGX_WALK_EVEN_CUTS")) {
        const double v = walker_speed().load(), D = streamed ? G.land_ms : 0.0, copy_ms = 0.15 + 3.2 * ((double)n_ref / K) / 12e6, warm_ms = 2.0 * (double)warm / v, floor_len = std::max(1024.0, (double)n_ref / (16.0 * K));
        std::vector<double> len((size_t)K, (double)n_ref / K), ready((size_t)K, 0.0), at((size_t)K + 1, 0.0);
        for (int round = 0; round < 6; round++) {
            for (int i = 0; i < K; i++) at[(size_t)i + 1] = at[(size_t)i] + len[(size_t)i];
            double sum = ready[0] = D;
            for (int i = 1; i < K; i++) { const double hi = i + 2 >= K ? (double)G.n_pos : std::max(at[(size_t)i + 2], at[(size_t)i + 1] + (double)warm); sum += ready[(size_t)i] = D * std::min(1.0, hi / (double)G.n_pos) + copy_ms + warm_ms; }
            const double T = ((double)n_ref / v + sum) / K;
            double total = 0;
            for (int i = 0; i < K; i++) total += len[(size_t)i] = std::max(v * (T - ready[(size_t)i]), floor_len);
            for (int i = 0; i < K; i++) len[(size_t)i] *= (double)n_ref / total;
        }
        double acc = 0;
        for (int i = 1; i < K; i++) { acc += len[(size_t)i - 1]; const agx_u32 lo = cuts[(size_t)i - 1] + 8 * slack, hi = n_ref - (agx_u32)(K - i) * 8 * slack; const agx_u32 c = (agx_u32)acc; cuts[(size_t)i] = c < lo ? lo : c > hi ? hi : c; }
    }
    auto cut_at = [&](int i) { return i >= K ? n_ref : i <= 0 ? 0u : cuts[(size_t)i]; };
    const agx_u32 n_side = G.n_ids - G.n_pos;
    auto side_of = [&](agx_u32 x) { return G.n_pos + (agx_u32)(std::lower_bound(G.side_xpos, G.side_xpos + n_side, x) - G.side_xpos); };
    const agx_u32 side_ref = side_of(n_ref);
    pre_out.reserve((size_t)G.n_ids + G.n_ids / 32 + 4096);
    const bool timing = getenv("AGX_WALK_TIMING") != nullptr;
    auto clock = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

    const double t_enter = clock();
    std::atomic<bool> cancel{false};
    std::atomic<int> copied{0};
    const agx_u8 *const pristine = WA.m;                // (nobody marks it before every window is copied)
    std::vector<SpecWalker> B((size_t)K - 1);
    for (int i = 1; i < K; i++) {
        SpecWalker &b = B[(size_t)i - 1];
        b.c = cut_at(i); b.c_next = cut_at(i + 1);
        b.side_c = side_of(b.c); b.side_next = side_of(b.c_next);
        // its window: main ids [win_lo, win_hi), or to the end of the table (the last two walkers: the appended positions behind the reference too); what it may
        // look at while it warms up / on its stretch lies `slack` inside
        // (uneven cuts: a short stretch's neighbour is short too — the window then reaches a warm-up's length, three times the longest chain's reach, beyond the stretch at least)
        int ahead = i + 2; while (ahead < K && cut_at(ahead) < b.c_next + warm) ahead++;
        const bool open_end = ahead >= K;
        // (behind it: the warm-up and as much again, one stretch at most)
        const agx_u32 back = 2 * warm + slack < b.c - cut_at(i - 1) ? 2 * warm + slack : b.c - cut_at(i - 1);
        b.win_lo = b.c - back; b.win_hi = open_end ? n_ref : cut_at(ahead); b.side_win_hi = open_end ? side_ref : side_of(b.win_hi);
        b.copy_hi = open_end ? G.n_pos : b.win_hi; b.side_copy_lo = side_of(b.win_lo); b.side_copy_hi = open_end ? G.n_ids : b.side_win_hi;
        b.warm_lo = b.win_lo ? b.win_lo + slack : 0; b.warm_hi = open_end ? 0xFFFFFFFFu : b.win_hi - slack; b.look_hi = open_end ? n_ref : b.win_hi - slack;
        b.w0 = b.c > warm ? b.c - warm : 0; if (b.w0 < b.warm_lo) b.w0 = b.warm_lo;
        b.bytes.take((size_t)G.n_ids + 64);
        b.G = G; b.G.meta = (const agx_u8 *)b.bytes.p; b.G.meta_rw = (agx_u8 *)b.bytes.p;
        b.W.reset(new Walker(V, b.G));
        b.recs.reserve((size_t)G.n_pos / 2048 + 1024);
        b.R.reset(new WalkRun(*b.W, more_arenas[i - 1], b.recs, nullptr));      // (its lists and k-mer tails live in one of the caller's arenas: they outlive this function, and an arena serves one thread)
    }
    // they start: copy the window, warm-up from w0 to c, then the stretch [c, c_next)
    struct Join { Assistant *a; int n; std::atomic<bool> &c; std::vector<bool> joined; void now(int i) { if (!joined[(size_t)i]) { a->wait(i); joined[(size_t)i] = true; } } ~Join() { c.store(true); for (int i = 0; i < n; i++) now(i); } } join{assistant, K - 1, cancel, std::vector<bool>((size_t)K - 1, false)};
    for (int i = 1; i < K; i++) {
        SpecWalker *bp = &B[(size_t)i - 1];
        assistant->run([bp, &cancel, &copied, pristine, &G, n_ref, side_ref] {
            SpecWalker &b = *bp; Walker &W = *b.W; WalkRun &R = *b.R;
            auto clk = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            b.t_run = clk();
            struct Copied { std::atomic<int> &n; bool done = false; void now() { if (!done) { done = true; n.fetch_add(1, std::memory_order_release); } } ~Copied() { now(); } } copied_mark{copied};      // (the first walker waits for this count whatever happens here)
            try {
                if (const char *pz = getenv("AGX_WALK_POISON")) { const long v = strtol(pz, nullptr, 0); memset(W.m, v > 1 ? (int)(v & 0xFF) : 0xA5, (size_t)G.n_ids + 64); }      // test hook (1 or a byte value): whatever the last unit left outside the window must not matter
                if (G.wait_landed) G.wait_landed(G.land_ctx, b.copy_hi, b.side_copy_hi);      // (a streamed download: the window — all this walker will ever look at — is in place)
                memcpy(W.m + b.win_lo, pristine + b.win_lo, (size_t)b.copy_hi - b.win_lo);
                memcpy(W.m + b.side_copy_lo, pristine + b.side_copy_lo, (size_t)b.side_copy_hi - b.side_copy_lo + (b.side_copy_hi == G.n_ids ? 64 : 0));      // (+ the padding behind the table)
                // what lies behind the window is whatever the last unit left there: a forced run that reaches the window's end must stop AT it (a visited byte without a
                // cont bit ends every run in front of it) and so gives itself away — the run's last node then lies behind what the walker may look at
                if (b.copy_hi < G.n_pos) W.m[b.copy_hi] = AGX_WM_VISITED;
                if (b.side_copy_hi < G.n_ids) W.m[b.side_copy_hi] = AGX_WM_VISITED;
                copied_mark.now(); b.t_copy = clk();
                W.spec = true; W.look_lo = b.warm_lo; W.look_hi = b.warm_hi; W.scan_hi = b.warm_hi;
#ifdef AGX_WALK_CHECK
                W.chk_lo = b.win_lo; W.chk_hi = b.copy_hi; W.chk_slo = b.side_copy_lo; W.chk_shi = b.side_copy_hi;
#endif
                R.cancel = &cancel; R.st.cp = b.w0; R.keep = false; R.log = &b.log; R.log_main = 0; R.log_side = G.n_pos;
                R.go(b.c);                                  // warm-up: decides records, keeps none; every mark is logged
                b.at_c = R.st; b.n_warm = b.log.size(); b.t_warm = clk();
                // What the warm-up marked behind the reference (appended positions) is forgotten: this walker does not know what the sequential
                // walk has visited there, so it must not find anything visited there that it marked itself — it finds those nodes unvisited, at
                // most, and gives up when a walk leads to them.  (Its marks in front of c are never looked at again: whatever leads there makes it give up.)
                for (const auto &r : b.log) for (agx_u32 v = r.first; v <= r.second; v++) if (v < G.n_pos ? v >= n_ref : v >= side_ref) W.m[v] &= (agx_u8)~AGX_WM_VISITED;
                R.keep = true; R.st.seqID = 0;
                if (b.c_next < n_ref) { R.log_main = b.c_next; R.log_side = b.side_next; } else R.log = nullptr;      // (the next walker is checked against what reaches its stretch)
                W.look_lo = b.c; W.look_hi = b.look_hi;
                if (const char *e = getenv("AGX_WALK_SPLIT_LOOK")) { const unsigned long long hi = (unsigned long long)b.c + strtoull(e, nullptr, 10); if (hi < W.look_hi) W.look_hi = (agx_u32)hi; }      // test hook: a narrow view makes the walker give up
                if (!W.invalid && !cancel.load()) R.go(b.c_next);
                b.ok = !W.invalid && !cancel.load() && R.st.cp >= b.c_next; b.t_done = clk();
            } catch (const Error &e) { b.error = e.msg; } catch (const std::exception &e) { b.error = e.what(); }
        }, i - 1);
    }
    // (the first walker's bytes are the source of the windows: what it marks at or behind the lowest window's start waits until they are all taken — Walker::before_mark;
    // r05 waited here, 1.3-1.5 ms in front of every large unit's walk)
    WA.copies = &copied; WA.copies_want = K - 1; WA.guard_main = B[0].win_lo; WA.guard_side = B[0].side_copy_lo;
    struct Unguard { Walker &w; ~Unguard() { w.copies = nullptr; } } unguard{WA};      // (`copied` lives in this frame)
    if (G.wait_landed) G.wait_landed(G.land_ctx, G.n_pos, G.n_ids);      // a streamed download: the first walker may look anywhere

    const double tw0 = clock();
    // A: the sequential walk up to the first meeting point, its marks at and behind it logged
    MarkLog log_a;
    WalkRun A(WA, arena, written, nullptr);
    A.log = &log_a; A.log_main = B[0].c; A.log_side = B[0].side_c;
    A.go(B[0].c);
    A.log = nullptr;
    const size_t a_first = A.jobs.size();               // A's records in front of the first meeting point
    const double tw1 = clock();
    // the walkers' stretches, one after the other: each stands if the walker arrived at its c in the state the sequential walk is in there
    const MarkLog *truth = &log_a; WalkState at = A.st; int stood = 0; MarkLog ma, mb;
    for (int i = 1; i < K; i++) {
        SpecWalker &b = B[(size_t)i - 1];
        join.now(i - 1);
        bool same = b.ok && b.error.empty() && at.same_scan(b.at_c);
        if (same) {
            MarkLog warm_marks(b.log.begin(), b.log.begin() + (long)b.n_warm);
            clip_marks(*truth, b.c, n_ref, G.n_pos, b.side_c, side_ref, ma); clip_marks(warm_marks, b.c, n_ref, G.n_pos, b.side_c, side_ref, mb);
            same = ma == mb;
        }
        if (timing) {
            if (same) fprintf(stderr, "[agx walk] walker %d of %d, [%u, %u) after a warm-up from %u: its stretch stands (began at %.2f ms, window %.2f ms, warm-up %.2f ms, stretch %.2f ms)\n", i + 1, K, b.c, b.c_next, b.w0, b.t_run - t_enter, b.t_copy - b.t_run, b.t_warm - b.t_copy, b.t_done - b.t_warm);
            else if (!b.error.empty()) fprintf(stderr, "[agx walk] walker %d of %d failed (%s): walked on by the first walker from %u\n", i + 1, K, b.error.c_str(), b.c);
            else if (b.W->invalid) fprintf(stderr, "[agx walk] walker %d of %d gave up (a walk led to position %u, outside [%u, %u)): walked on by the first walker from %u\n", i + 1, K, b.W->gave_up_at, b.W->look_lo, b.W->look_hi, b.c);
            else fprintf(stderr, "[agx walk] walker %d of %d, [%u, %u) after a warm-up from %u: states differ at the meeting point (scan at %u / %u, last record %u..%u / %u..%u, %zu / %zu mark ranges): walked on by the first walker\n",
                         i + 1, K, b.c, b.c_next, b.w0, at.cp, b.at_c.cp, at.sOffBak, at.eOffBak, b.at_c.sOffBak, b.at_c.eOffBak, ma.size(), mb.size());
        }
        if (!same) break;
        // its stretch is the sequential walk's: the state where it ends is the sequential walk's there, and so is every mark that reaches further
        const agx_u32 shift = at.seqID;
        for (PreJob &j : b.R->jobs) j.f[0] += shift;
        at = b.R->st; at.seqID = shift + b.R->st.seqID;
        truth = &b.log; stood = i;
    }
    cancel.store(true);                                 // (walkers behind one that does not stand walk for nothing)
    for (int i = stood; i < K - 1; i++) join.now(i);
    WA.copies = nullptr;                                // (every walker's thread is done: every window was taken)
    {   // what a stretch costs, for the next streamed walk's cuts: positions per millisecond of the stretches that stood
        double pos = 0, ms = 0;
        for (int i = 1; i <= stood; i++) { const SpecWalker &b = B[(size_t)i - 1]; if (b.t_done > b.t_warm) { pos += (double)(b.c_next - b.c); ms += b.t_done - b.t_warm; } }
        if (ms > 0.05 && pos > 1e5) { const double seen = pos / ms, old = walker_speed().load(); walker_speed().store(0.7 * old + 0.3 * std::min(std::max(seen, 1e5), 5e6)); }
    }
    const double tw2 = clock();
    // what stands: the records behind A's, numbered on; the marks into A's bytes (the appended positions — and whatever did not stand — are walked on those).
    // The walkers' threads are idle now: each merges its own walker's marks (disjoint ranges of A's bytes), this thread the last one's.  (Nothing is
    // merged when nobody will look: every stretch stood and no unvisited node is left at the appended positions behind the reference — only the first
    // walker's bytes can say so: the others never mark there.)
    if (!(stood == K - 1 && WA.next_live(n_ref, G.n_pos) >= G.n_pos && WA.next_live(side_ref, G.n_ids) >= G.n_ids)) {
        auto merge = [&WA](const agx_u8 *src, agx_u32 lo, agx_u32 hi) {
            agx_u32 v = lo;
            for (; v < hi && (v & 7u); v++) WA.m[v] |= (agx_u8)(src[v] & AGX_WM_VISITED);
            for (; v + 8 <= hi; v += 8) { uint64_t x, y; memcpy(&x, WA.m + v, 8); memcpy(&y, src + v, 8); x |= y & 0x8080808080808080ull; memcpy(WA.m + v, &x, 8); }
            for (; v < hi; v++) WA.m[v] |= (agx_u8)(src[v] & AGX_WM_VISITED);
        };
        // (what a walker marked behind its own stretch is in the next walker's bytes as well — that is what was compared — unless this is the last one that
        // stands; walkers that share bytes: the bytes behind that one's window are another walker's)
        auto merge_of = [&](int i) {
            const SpecWalker &b = B[(size_t)i - 1]; const agx_u8 *src = b.W->m;
            if (i < stood) { merge(src, b.c, b.c_next); merge(src, b.side_c, b.side_next); } else { merge(src, b.c, b.win_hi); merge(src, b.side_c, b.side_win_hi); }
        };
        struct Wait { Assistant *a; int n; ~Wait() { for (int i = 0; i < n; i++) a->wait(i); } } wait{assistant, stood > 1 ? stood - 1 : 0};
        for (int i = 1; i < stood; i++) assistant->run([&merge_of, i] { merge_of(i); }, i - 1);
        if (stood) merge_of(stood);
    }
    for (int i = 1; i <= stood; i++) {
        SpecWalker &b = B[(size_t)i - 1];
        for (Rec &r : b.recs) written.push_back(std::move(r));
        WA.n_fetched += b.W->n_fetched;
        A.n_walks += b.R->n_walks; A.n_hops += b.R->n_hops; A.n_runs += b.R->n_runs; A.n_general += b.R->n_general; A.run_nodes += b.R->run_nodes;
    }
    if (stood) A.st = at;
    const double tw3 = clock();
    A.go(G.n_pos);                                      // the appended positions — and everything from the first stretch that did not stand
    const double tw4 = clock();
    // the pre-extended output: every record's place is known now, so the helpers format their shares while this thread formats the last one
    std::vector<PreJob> all;
    all.insert(all.end(), A.jobs.begin(), A.jobs.begin() + (long)a_first);      // A's in front of the first meeting point, the stretches that stand, A's behind them
    for (int i = 1; i <= stood; i++) all.insert(all.end(), B[(size_t)i - 1].R->jobs.begin(), B[(size_t)i - 1].R->jobs.end());
    all.insert(all.end(), A.jobs.begin() + (long)a_first, A.jobs.end());
    std::vector<size_t> place(all.size() + 1, 0);
    auto header_len = [](const PreJob &j) { size_t n = 1 + 9 * 2 + 2; for (int i = 0; i < 10; i++) { agx_u32 v = j.f[i]; do { n++; v /= 10u; } while (v); } return n; };
    for (size_t i = 0; i < all.size(); i++) place[i + 1] = place[i] + header_len(all[i]) + all[i].total + (all[i].total + 59) / 60;
    const size_t total = place[all.size()];
    if (G.wait_str) G.wait_str(G.land_ctx);             // (a streamed download: the bases come last — nothing has read them so far, the records hold byte ranges)
    pre_out.n = 0; char *base = pre_out.grow(total);
    auto format = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const PreJob &j = all[i];
            char *h = base + place[i]; *h++ = '>';
            for (int f = 0; f < 10; f++) { h = put_u32(h, j.f[f]); if (f < 9) { *h++ = ','; *h++ = ' '; } }
            *h++ = ' '; *h++ = '\n';
            LineWriter lw(h); for (size_t g = 0; g < j.n_segs; g++) lw.put(j.segs[g].p, j.segs[g].n); lw.end();
        }
    };
    {
        const int shares = total > (1u << 16) ? K : 1;
        std::vector<size_t> cut((size_t)shares + 1, all.size()); cut[0] = 0;
        for (int t = 1; t < shares; t++) { size_t m = cut[(size_t)t - 1]; while (m < all.size() && place[m] < total / (unsigned)shares * (unsigned)t) m++; cut[(size_t)t] = m; }
        struct Wait { Assistant *a; int n; ~Wait() { for (int i = 0; i < n; i++) a->wait(i); } } wait{assistant, shares - 1};
        for (int t = 0; t + 1 < shares; t++) { const size_t lo = cut[(size_t)t], hi = cut[(size_t)t + 1]; assistant->run([&format, lo, hi] { format(lo, hi); }, t); }
        format(cut[(size_t)shares - 1], all.size());
    }
    if (timing) fprintf(stderr, "[agx walk] first walker began at %.2f ms; %u appended positions behind the reference%s\n", tw0 - t_enter, G.n_pos - n_ref, streamed ? " (streamed download: stretches cut by when their windows land)" : "");
    if (timing && streamed) { fprintf(stderr, "[agx walk] cuts (%.1f ms of download left at the start, %.0f positions per ms):", G.land_ms, walker_speed().load()); for (int i = 1; i < K; i++) fprintf(stderr, " %u", cuts[(size_t)i]); fprintf(stderr, "\n"); }
    if (timing) fprintf(stderr, "[agx walk] %d walkers, %d stretches stood: first stretch %.1f ms, waited %.1f ms for the others, merge %.1f ms, rest %.1f ms, formatting %.1f ms\n", K, stood, tw1 - tw0, tw2 - tw1, tw3 - tw2, tw4 - tw3, clock() - tw4);
    walk_report(A);
    return true;
}

}  // namespace

namespace {

// extdContigs2, AG:2296-2380
void join(std::vector<Rec> &c) {
    const int n = (int)c.size();
    for (int cp = 0; cp < n; cp++) if (c[cp].extended == 1)
        for (int q = cp + 1; q < n; q++) {
            if (contains(c[cp].sID, c[cp].sOff, c[cp].eID, c[cp].eOff, c[q].sID, c[q].sOff, c[q].eID, c[q].eOff)) c[q].extended = 2;
            else if (c[cp].eID != c[q].sID || c[cp].eOff < c[q].sOff) break;
        }
    for (int cp = n - 1; cp >= 0; cp--) if (c[cp].extended == 1)
        for (int q = cp - 1; q >= 0; q--) {
            if (contains(c[cp].sID, c[cp].sOff, c[cp].eID, c[cp].eOff, c[q].sID, c[q].sOff, c[q].eID, c[q].eOff)) c[q].extended = 2;
            else if (c[q].eID != c[cp].sID || c[q].eOff < c[cp].sOff) break;
        }
    for (int cp = 0; cp < n; cp++) {
        while (c[cp].extended == 1) {
            int cand = -1, ncand = 0;
            for (int q = cp + 1; q < n; q++) if (c[q].extended != 2) { if (c[cp].eOff >= c[q].sOff) { cand = q; ncand++; } else break; }
            if (ncand != 1) break;
            Rec &d = c[cand]; d.extended = 2;
            const size_t from = (size_t)(int)(c[cp].eOff - d.sOff + 1);      // (int) -> size_t as in the reference's loop test (AG:2368-2370)
            if (from < d.nuc.len) c[cp].nuc.add_from(d.nuc, from);
            c[cp].eID = d.eID; c[cp].eOff = d.eOff; c[cp].eID0 = d.eID0; c[cp].eOff0 = d.eOff0;
        }
    }
}

inline int overlaps(agx_u32 x1, agx_u32 y1, agx_u32 x2, agx_u32 y2) {      // AG:2388-2394
    return (x1 <= x2 && x2 <= y1 && y1 <= y2 && (int)y1 - (int)x2 > 0) || (x2 <= x1 && x1 <= y2 && y2 <= y1 && (int)y2 - (int)x1 > 0) ||
           (x1 <= x2 && x2 <= y2 && y2 <= y1 && (int)y2 - (int)x2 > 0) || (x2 <= x1 && x1 <= y1 && y1 <= y2 && (int)y1 - (int)x1 > 0);
}

// scaffoldContigs, AG:2396-2464
void scaffold(const UnitView &V, const GraphView &G, std::vector<Rec> &c, OutBuf &out, Arena *arena, Assistant *assistant) {
    std::vector<Bases> sc; sc.reserve(c.size());
    const agx_u32 n = (agx_u32)c.size();
    // The reference looks at EVERY later record for one that overlaps the mate range [sOff0, eOff0] of the current one (AG:2415-2447): 52 k
    // records of a 62 Mb unit made that 25-40 ms.  Every clause of overlap() needs sOff(q) <= eOff0 and sOff0 <= eOff(q); the records are
    // in scan order (sOff never decreases), so the candidates are the stretch from the first record that can still reach sOff0 (none is
    // longer than `longest`) to the last one that starts at or before eOff0 — the same first match, found without the rest.
    std::vector<agx_u32> s_off(n); agx_u32 longest = 0; bool ordered = true;
    for (agx_u32 i = 0; i < n; i++) {
        s_off[i] = c[i].sOff;
        if (i && s_off[i] < s_off[i - 1]) ordered = false;
        if (c[i].eOff < c[i].sOff) longest = 0xFFFFFFFFu; else if (c[i].eOff - c[i].sOff > longest) longest = c[i].eOff - c[i].sOff;
    }
    for (agx_u32 cp = 0; cp < n; cp++) {
        if (!(c[cp].sID != AGX_NONE && c[cp].extended == 1)) continue;
        sc.push_back(std::move(c[cp].nuc)); c[cp].nuc = Bases(arena); c[cp].sID = AGX_NONE;      // a record is used at most once (sID = -1 marks it, AG:2411)
        bool cont = true;
        while (c[cp].sID0 == c[cp].eID0 && cont) {
            cont = false;
            agx_u32 q = cp + 1;
            if (ordered && longest != 0xFFFFFFFFu && c[cp].sOff0 > longest) {      // records that end in front of sOff0 cannot overlap
                const agx_u32 from = (agx_u32)(std::lower_bound(s_off.begin(), s_off.end(), c[cp].sOff0 - longest) - s_off.begin());
                if (from > q) q = from;
            }
            for (; q < n; q++) {
                if (ordered && c[q].sOff > c[cp].eOff0) break;      // nor can any record from here on
                if (!(c[cp].eID0 == c[q].sID && c[q].sID == c[q].eID && overlaps(c[cp].sOff0, c[cp].eOff0, c[q].sOff, c[q].eOff) && c[q].extended == 1)) continue;
                if (c[q].sOff > c[cp].eOff) {
                    const agx_u32 gap = c[q].sOff - c[cp].eOff - 1; agx_u32 covered = 0;
                    for (agx_u32 i = 0; i < gap; i++) { const agx_u32 x = c[cp].eOff + i + 1; if ((G.meta[x] & AGX_WM_ANY) || V.has_cm(x)) covered++; }
                    if (gap == 0 || (double)(int)covered / gap >= 0.5) sc.back().add(V.ref + c[cp].eOff + 1, gap);
                    else continue;
                }
                sc.back().add_from(c[q].nuc, 0); c[q].sID = AGX_NONE; cp = q; cont = true;
                break;
            }
        }
    }
    const double t_dec = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    // output: every scaffold's place is known before a byte is written, so the assistant formats the first half while this thread formats the second
    std::vector<size_t> at(sc.size() + 1, 0);
    auto header = [](char *hdr, size_t i) { char *h = hdr; *h++ = '>'; char t[24]; int n = 0; do { t[n++] = (char)('0' + i % 10); i /= 10; } while (i); while (n) *h++ = t[--n]; *h++ = '\n'; return (size_t)(h - hdr); };
    for (size_t i = 0; i < sc.size(); i++) { char hdr[32]; at[i + 1] = at[i] + header(hdr, i) + sc[i].len + (sc[i].len + 59) / 60; }
    const size_t total = at[sc.size()];
    out.n = 0; char *base = out.grow(total);
    auto format = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            char hdr[32]; const size_t hl = header(hdr, i);
            char *w = base + at[i]; memcpy(w, hdr, hl);
            LineWriter lw(w + hl); lw.put(sc[i]); lw.end();
        }
    };
    // (r02: the assistant formatted the first half; r03: up to eight shares, one per helper thread and one for this thread — 20 MB of scaffolds are 0.5 ms of copying for one)
    const int shares = assistant && total > (1u << 16) ? std::min(8, 1 + assistant->helpers()) : 1;
    if (shares > 1) {
        std::vector<size_t> cut((size_t)shares + 1, sc.size()); cut[0] = 0;
        for (int t = 1; t < shares; t++) { size_t m = cut[(size_t)t - 1]; while (m < sc.size() && at[m] < total / (unsigned)shares * (unsigned)t) m++; cut[(size_t)t] = m; }
        struct Wait { Assistant *a; int n; ~Wait() { for (int i = 0; i < n; i++) a->wait(i); } } wait{assistant, shares - 1};      // (also if this share throws)
        for (int t = 0; t + 1 < shares; t++) { const size_t lo = cut[(size_t)t], hi = cut[(size_t)t + 1]; assistant->run([&format, lo, hi] { format(lo, hi); }, t); }
        format(cut[(size_t)shares - 1], sc.size());
    } else format(0, sc.size());
    if (getenv("AGX_WALK_TIMING")) fprintf(stderr, "[agx walk] scaffold: output %.1f ms of it (%zu scaffolds, %zu bytes)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_dec, sc.size(), total);
}

}  // namespace

// chains of conti-mers: heads are the conti-mers nobody links to
void build_chains(Threads &T) {
    const size_t n = T.cm.size(), n_pos = T.ref.size();
    T.chain_off.assign(1, 0); T.chain_end_pos.clear(); T.chain_str.clear();
    T.chain_str.reserve(n);
    T.hop.assign(n_pos, agx_hop{0, 0, 0});
    std::vector<agx_u8> state(n, 0);                     // 1: some conti-mer links to it (not a chain head); 2: visited
    auto index_of = [&](agx_u32 off, agx_u32 item) -> size_t {
        if (off >= n_pos || T.cm_start[off] + item >= T.cm_start[off + 1]) throw Error{E_ARG, "conti-mer link to a missing entry"};
        return (size_t)T.cm_start[off] + item;
    };
    for (size_t i = 0; i < n; i++) if (T.cm[i].next_off != AGX_NONE) state[index_of(T.cm[i].next_off, T.cm[i].next_item)] = 1;
    // One traversal per chain, heads in conti-mer order.  A position with exactly one conti-mer that has a next gets its hop entry (the walk
    // continues ON the next conti-mer, AG:2049-2055: the chain's bases from there on, and where the chain ends) while the chain is
    // followed; length and landing position are filled in when its end is known.
    std::vector<agx_u32> pending;                        // positions of this chain whose hop entry waits for the chain's end
    size_t visited = 0;
    T.segs.clear(); T.n_seg0 = 0;
    for (size_t x = 0; x < n_pos; x++) for (size_t h = T.cm_start[x]; h < T.cm_start[x + 1]; h++) {
        if (state[h]) continue;                          // linked to (or, never true here, visited): not a head
        pending.clear();
        size_t c = h; agx_u32 pos = (agx_u32)x;
        const size_t seg_first = T.segs.size(), str_base = T.chain_str.size();
        for (agx_u32 i = 0;; i++) {
            if (state[c] == 2) throw Error{E_ARG, "conti-mer chains are not simple lists"};
            state[c] = 2; visited++;
            T.chain_str.push_back(T.cm[c].nuc);
            const ContiMer &m = T.cm[c];
            {   // the chain as runs (agx_cmseg): extend the open run or start one.  hop_len0 holds the element's index until the chain's end is known
                const agx_u32 rank = (agx_u32)(c - T.cm_start[pos]);
                agx_cmseg *g = T.segs.size() > seg_first ? &T.segs.back() : nullptr;
                const bool joins = g && pos == g->pos0 + g->len && rank == g->rank && m.cid == g->cid && (g->len == 1 || m.coff == g->coff0 + g->len * g->dcoff);
                if (joins) { if (g->len == 1) g->dcoff = m.coff - g->coff0; g->len++; }
                else T.segs.push_back(agx_cmseg{pos, 1u, m.cid, m.coff, 0u, rank, (agx_u32)(str_base + i + 1), i, 0u, 0u});
            }
            if (m.next_off == AGX_NONE) break;
            if (T.cm_start[pos + 1] - T.cm_start[pos] == 1) { T.hop[pos].str_off = (agx_u32)T.chain_str.size(); pending.push_back(pos); }     // (the next conti-mer's base is pushed next)
            c = index_of(m.next_off, m.next_item); pos = m.next_off;
        }
        if (T.chain_str.size() >= 0xFFFFFFFFull) throw Error{E_ARG, "conti-mer chains exceed 2^32 bases"};
        for (agx_u32 p : pending) { T.hop[p].len = (agx_u32)(T.chain_str.size() - T.hop[p].str_off); T.hop[p].end_pos = pos; }
        {   const agx_u32 last = (agx_u32)(T.chain_str.size() - str_base - 1);       // index of the chain's last conti-mer
            for (size_t g = seg_first; g < T.segs.size(); g++) { T.segs[g].hop_len0 = last - T.segs[g].hop_len0; T.segs[g].hop_end = pos; } }
        T.chain_end_pos.push_back(pos); T.chain_off.push_back(T.chain_str.size());
    }
    if (visited != n) throw Error{E_ARG, "conti-mer cycle"};
    // rank-0 runs first, by position (the device finds the run of a position by bisection); then elem0 = running element count
    std::stable_sort(T.segs.begin(), T.segs.end(), [](const agx_cmseg &a, const agx_cmseg &b) { return (a.rank != 0) != (b.rank != 0) ? a.rank == 0 : (a.rank == 0 && a.pos0 < b.pos0); });
    agx_u32 e = 0;
    for (agx_cmseg &g : T.segs) { g.elem0 = e; e += g.len; if (g.rank == 0) T.n_seg0++; }
    if (e != n) throw Error{E_ARG, "conti-mer runs do not cover the conti-mers"};
    T.n_cm = n; T.cm_cnt.resize(n_pos);
    for (size_t x = 0; x < n_pos; x++) { const agx_u32 c = T.cm_start[x + 1] - T.cm_start[x]; if (c > 255) throw Error{E_UNSUPPORTED, "more than 255 conti-mers at one position"}; T.cm_cnt[x] = (agx_u8)c; }
}

void walk_join_scaffold(const UnitView &V, const GraphView &G, UnitOutput &out, Assistant *assistant) {
    if (!V.hop && !G.sp_hop && !V.segs) throw Error{E_ARG, "conti-mer chains were not built"};
    const bool timing = getenv("AGX_WALK_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double ts = now();
    Walker W(V, G);
    std::pmr::monotonic_buffer_resource arena((size_t)8 << 20);      // byte-range lists and trailing k-mers of the written records
    std::vector<std::unique_ptr<std::pmr::monotonic_buffer_resource>> others;      // the other walkers' (walk_split): an arena serves one thread (nothing is allocated before its first use)
    Arena *more_arenas[GraphView::MAX_WALKERS - 1];
    for (int i = 0; i < GraphView::MAX_WALKERS - 1; i++) { others.emplace_back(new std::pmr::monotonic_buffer_resource((size_t)1 << 20)); more_arenas[i] = others.back().get(); }
    std::vector<Rec> recs; recs.reserve((size_t)G.n_pos / 1024 + 1024);
    double t0 = now();
    if (!walk_split(W, V, G, out.pre_extended, recs, &arena, more_arenas, assistant)) walk(W, out.pre_extended, recs, &arena, assistant);
    double t1 = now();
    join(recs);
    double t2 = now();
    scaffold(V, G, recs, out.extended, &arena, assistant);
    double t3 = now();
    out.n_fetched = W.n_fetched;
    if (timing) fprintf(stderr, "[agx walk] set-up %.2f ms, walk %.2f ms, join %.2f ms, scaffold %.1f ms, %zu records, %u special ids of %u, %llu records fetched\n", t0 - ts, t1 - t0, t2 - t1, t3 - t2, recs.size(), G.n_special, G.n_ids, W.n_fetched);
}

}  // namespace agx
