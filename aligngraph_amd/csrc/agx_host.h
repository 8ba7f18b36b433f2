// agx_host.h — host-side containers shared by the loader, the engine and the walk.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>
#include "agx_core.h"

namespace agx {

struct Error { int code; std::string msg; };   // thrown inside the library, converted to a return code at the C-ABI

// error codes (include/agx.h mirrors these)
enum {
    E_OK = 0, E_IO = -1, E_FORMAT = -2, E_UNSUPPORTED = -3, E_ALIGNMENT = -4, E_DEVICE = -5, E_ARG = -6, E_OVERFLOW = -7, E_NOGPU = -8
};

// full conti-mer (ContiMer, AG:51-62): key for node build + what the walk follows
struct ContiMer { char nuc; agx_u32 cid, coff, next_off, next_item; };   // next_off == NONE: chain end

// Result of contig threading (updateGenomeWithContig, AG:884-1217) for one unit
struct Threads {
    std::string ref;                       // reference bases followed by the positions appended for contig insertions (AG:1016, 1036)
    agx_u32 n_ref = 0;                     // length of the original unit sequence
    std::vector<agx_u32> cm_start;         // [n_pos+1]   (general loader only; the fast loader keeps counts and runs)
    std::vector<ContiMer> cm;              //             (general loader only)
    std::vector<agx_u8> cm_cnt;            // [n_pos] conti-mers per position
    size_t n_cm = 0;                       // conti-mers in all (= cm.size() where cm is kept)
    std::string initial_contigs;           // bytes of tmp/_initial_contigs.<u>.fa
    // Conti-mer chains (one per contig placement): following `next` from a conti-mer never looks at mutable state
    // (AG:2064-2072), so the walk appends a precomputed suffix instead of chasing pointers.  Built by build_chains().
    std::vector<size_t> chain_off;          // [n_chains+1] chain c's bases are chain_str[chain_off[c] .. chain_off[c+1])
    std::vector<agx_u32> chain_end_pos;     // position of the chain's terminal conti-mer
    std::string chain_str;
    // per-position hop table (agx_hop, agx_core.h)
    std::vector<agx_hop> hop;               // [n_pos]
    // the same chains as runs (agx_cmseg): what the device gets instead of cm_start / cm / hop.  The first n_seg0 are the rank-0 runs, by position.
    std::vector<agx_cmseg> segs; agx_u32 n_seg0 = 0;
};
void build_chains(Threads &T);

// Packed read alignments of one unit (loadSeq/loadReadAli, AG:361-404, 1233-1277)
struct Pairs {
    std::vector<agx_hit> hits;             // processing order
    std::vector<agx_run> runs;
    std::string bases;                     // read slot s occupies [s*stride, s*stride+len)
    agx_u32 stride = 0;
    agx_u32 n_slots = 0;
    unsigned long long n_pairs_in_file = 0, n_sam_pairs = 0, n_kept = 0;
};

// What the host walk (and the engine's staging) reads of a loaded unit, wherever it lives: in the loader's containers (Threads / Pairs) or in
// a mapped unit cache file (agx_engine.cpp).
struct UnitView {
    const char *ref = nullptr; size_t n_pos = 0; agx_u32 n_ref = 0;   // unit sequence + appended positions
    const agx_u32 *cm_start = nullptr;                                // [n_pos + 1], or null: then cm_cnt
    const agx_u8 *cm_cnt = nullptr;                                   // [n_pos] conti-mers per position (the fast loader and the unit cache keep counts, not a prefix)
    const char *chain_str = nullptr;                                  // conti-mer chain suffixes (agx_hop::str_off points in here)
    const agx_hop *hop = nullptr;                                     // per-position hop table, or null: then GraphView::sp_hop, and for ids outside it the runs below
    const agx_cmseg *segs = nullptr; agx_u32 n_seg0 = 0;              // the conti-mer chains as runs (rank-0 runs first, by position): hop entries by bisection (agx_seg_hop)
    const char *bases = nullptr; agx_u32 stride = 0;                  // read bases, slot s at bases + s * stride (k-mer strings of written records)
    const uint64_t *row_off = nullptr;                                // or: row r of the staged read bases starts at bases + row_off[r] (the mapped reads file; GraphView::row_slot is then not consulted)
    // or (bases == nullptr): the k-mer tails come out of the staged 2-bit rows themselves — row r = codes2 + r * (stride / 4), base j in bits 2 * (j & 3) of byte j >> 2 (agx_pack_classes2) —
    // and the list of the bases that are not A, C, G, T (index r * stride + j, ascending) with their bytes.  Units whose read alignments were handed over staged (tmp/_agx_pairs.<u>.bin) have
    // nothing else; the arrays live in the mapped file (the pinned copies are gone once a one-shot unit has been downloaded).
    const agx_u8 *codes2 = nullptr; const unsigned long long *other_idx = nullptr; const agx_u8 *other_byte = nullptr; size_t n_other = 0;
    // r06, tile-ordered upload: the device numbers the read rows by the hit's place in the tile order (every hit's left-mate row travels at that place); slot_row[place] = the row
    // of the staged arrays above (null: the k-mer string references name those rows themselves)
    const agx_u32 *slot_row = nullptr;
    bool has_cm(size_t x) const { return cm_cnt ? cm_cnt[x] != 0 : cm_start[x + 1] > cm_start[x]; }
    agx_u32 cm_count(size_t x) const { return cm_cnt ? cm_cnt[x] : cm_start[x + 1] - cm_start[x]; }
    const char *initial = nullptr; size_t n_initial = 0;              // bytes of tmp/_initial_contigs.<u>.fa
};
inline UnitView view_of(const Threads &T, const Pairs &P) {
    UnitView V; V.ref = T.ref.data(); V.n_pos = T.ref.size(); V.n_ref = T.n_ref; V.cm_start = T.cm_start.size() == T.ref.size() + 1 ? T.cm_start.data() : nullptr;
    V.cm_cnt = T.cm_cnt.size() == T.ref.size() ? T.cm_cnt.data() : nullptr; V.chain_str = T.chain_str.data();
    V.hop = T.hop.size() == T.ref.size() ? T.hop.data() : nullptr; V.segs = T.segs.data(); V.n_seg0 = T.n_seg0; V.bases = P.bases.data(); V.stride = P.stride; V.initial = T.initial_contigs.data(); V.n_initial = T.initial_contigs.size();
    return V;
}

// Walk graph as downloaded from the device (or produced by the test executor); see agx_core.h "walk preparation".
// Walk ids: [0, n_pos) = first alive variant of each position (AGX_WM_ABSENT where there is none), [n_pos, n_ids) = further variants.
// Node records come as the sparse table of the special ids; any other id (only reachable after the +1000 skip) goes through fetch().
struct GraphView {
    agx_u32 n_pos = 0, n_ids = 0;
    const agx_u8 *meta = nullptr;                           // [n_ids + 64] AGX_WM_* bits (padding reads as 0)
    agx_u8 *meta_rw = nullptr;                              // null, or == meta: the walk may keep its visited marks in meta's bit 7 (the array is consumed by the walk)
    enum { MAX_WALKERS = 16 };                              // a large unit is walked by several walkers (agx_walk.cpp: walk_split), each of the further ones on a window of meta that it copies for itself
    const char *str = nullptr;                              // [n_ids] base a node emits
    const agx_u32 *side_xpos = nullptr;                     // [n_ids - n_pos] position of each side id, non-decreasing
    const unsigned long long *sp_bits = nullptr;            // [n_ids/64 + 1] special-id bitmap
    const agx_u32 *sp_rank = nullptr;                       // [n_ids/64 + 1] special ids before each 64-id word
    const agx_walknode *sp_node = nullptr; agx_u32 n_special = 0;   // records of the special ids, id order
    const agx_hop *sp_hop = nullptr;                        // [n_special] hop entry of each special id's position (optional: else UnitView::hop)
    // records of non-special ids: `rows` groups of `width` consecutive ids, group r starting at first + r*stride, into out[rows*width]
    void (*fetch)(void *ctx, agx_u32 first, agx_u32 stride, agx_u32 rows, agx_u32 width, agx_walknode *out) = nullptr; void *fetch_ctx = nullptr;
    const agx_edge_ovf *ovf = nullptr; size_t n_ovf = 0;     // walk ids; NONE/NONE entries and duplicates are ignored
    const agx_u32 *row_slot = nullptr;                      // k-mer string references name rows of the staged read bases: row -> read slot (null: they name read slots)
    // Streamed download (r06; agx_engine.cpp: begin_streamed_download): meta, sp_bits, sp_rank, sp_node and sp_hop fill in behind the walk's back, a position window at a
    // time from the front — main ids [0, x) together with the side ids of the positions below x —, str last; side_xpos and ovf are in place from the start.
    // wait_landed returns when everything about main ids [0, main_hi) and side ids [n_pos, side_hi) is in place; wait_str when str is.  The walk never looks at a byte it has
    // not waited for: the further walkers of walk_split wait for their windows and are confined to them (as they always were), the first walker — the only one whose walks may lead
    // anywhere — waits for everything and gets the shortest stretch in exchange (land_ms: what is left of the download when the walk begins, for that balance).
    void (*wait_landed)(void *ctx, agx_u32 main_hi, agx_u32 side_hi) = nullptr; void (*wait_str)(void *ctx) = nullptr; void *land_ctx = nullptr; double land_ms = 0;
};

// Megabyte-sized buffers that are written once, front to back (outputs, the walk's visited bytes): ask for transparent huge pages where
// the kernel offers them on request (THP "madvise" mode), so first-touch costs a few faults instead of one per 4 KB.  Advisory only.
void advise_huge(void *p, size_t n);

// malloc-backed output buffer whose storage can be handed to the C caller without another copy
// Output buffers that were handed to a caller and given back (agx_result_free) are kept for the next unit instead of going back to the C library: a unit's three outputs
// are 2.2 bytes per position of FRESH memory otherwise — page faults and the kernel's zeroing, a sixth of a whole-human job's host CPU time (r05) — and a run's units
// follow each other.  malloc'd memory throughout (a caller may free() an output itself); at most 128 buffers / 16 GB are kept (a whole-human job's 24 units hand back 72 buffers, 6.8 GB); agx_pool_trim(-1) frees them.
void *out_cache_take(size_t need, size_t &cap);      // a kept buffer of at least `need` bytes and at most twice that (cap = its usable size); nullptr: none
void out_cache_give(void *p);                        // (nullptr is fine)
void out_cache_trim();
struct OutBuf {
    char *p = nullptr; size_t n = 0, cap = 0;
    OutBuf() = default; OutBuf(const OutBuf &) = delete; OutBuf &operator=(const OutBuf &) = delete;
    ~OutBuf() { out_cache_give(p); }
    void reserve(size_t c) {
        if (c + 1 <= cap) return;
        if (!p && c >= ((size_t)1 << 20)) { size_t got = 0; if (void *q = out_cache_take(c + 1, got)) { p = (char *)q; cap = got; return; } }
        char *q = (char *)realloc(p, c + 1); if (!q) throw Error{E_ARG, "out of host memory"};
        p = q; cap = c + 1; advise_huge(p, cap);
    }
    char *grow(size_t add) { if (n + add + 1 > cap) reserve((n + add) + (n + add) / 2 + 64); char *w = p + n; n += add; return w; }
    void append(const char *s, size_t len) { memcpy(grow(len), s, len); }
    char *release() { if (!p) reserve(0); p[n] = 0; char *r = p; p = nullptr; n = cap = 0; return r; }
    // touches every page of the capacity: the first-touch faults (and the kernel's zeroing) of a 30 MB output happen where the caller has
    // time to spare instead of inside the walk
    void prefault() { for (size_t i = 0; i < cap; i += 4096) p[i] = 0; }
    void clear() { out_cache_give(p); p = nullptr; n = cap = 0; }
};
struct UnitOutput { OutBuf pre_extended, extended; unsigned long long n_fetched = 0; };

// read-only view of a whole file (mmap)
struct FileView {
    const char *p = nullptr; size_t n = 0; int fd = -1; bool mapped = false;
    explicit FileView(const std::string &path);
    ~FileView();
    FileView(const FileView &) = delete; FileView &operator=(const FileView &) = delete;
};

// Threads of a loader: `threads` workers that live as long as the team and run one function after another (a phase of a loader = one run()).
// Nothing leaves a worker as an exception: what one throws is rethrown by run() on the caller's thread.
class Team {
public:
    explicit Team(unsigned threads);
    ~Team();
    unsigned size() const { return n_; }
    void run(const std::function<void(unsigned)> &fn);      // fn(t) for t in [0, size()), t = 0 on the calling thread
private:
    struct Impl; Impl *impl_; unsigned n_;
};
// Scratch memory of the loaders: anonymous mappings, huge pages asked for, kept in a process-wide cache when they are given back (a unit's
// loader touches as much temporary memory as it produces output; fresh 4 KB pages cost it more than the parsing itself, and 160 threads faulting
// them in at once wait for each other on the address-space lock — the next unit finds the pages of the last one already there).
struct Scratch {
    void *p = nullptr; size_t n = 0;
    Scratch() = default; explicit Scratch(size_t bytes) { take(bytes); }
    ~Scratch() { give(); }
    Scratch(const Scratch &) = delete; Scratch &operator=(const Scratch &) = delete;
    Scratch(Scratch &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    Scratch &operator=(Scratch &&o) noexcept { if (this != &o) { give(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    void take(size_t bytes); void give();
};
void scratch_trim();      // unmaps what the cache holds
// fixed-capacity array in scratch memory (the capacity is an upper bound known before the fill; grow() for the rare case that it was a guess)
template <class T> struct SBuf {
    Scratch s; T *p = nullptr; size_t n = 0, cap = 0;
    SBuf() = default;
    SBuf(SBuf &&o) noexcept : s(std::move(o.s)), p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    SBuf &operator=(SBuf &&o) noexcept { if (this != &o) { s = std::move(o.s); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; } return *this; }
    void reserve(size_t c) { if (c <= cap) return; Scratch m(c * sizeof(T) + 64); if (n) memcpy(m.p, p, n * sizeof(T)); s = std::move(m); p = (T *)s.p; cap = c; }
    void push_back(const T &v) { if (n == cap) reserve(cap ? cap * 2 : 1024); p[n++] = v; }
    void append(const T *src, size_t k) { if (n + k > cap) reserve(std::max(cap * 2, n + k)); memcpy(p + n, src, k * sizeof(T)); n += k; }
    T &operator[](size_t i) { return p[i]; } const T &operator[](size_t i) const { return p[i]; }
    T &back() { return p[n - 1]; } const T &back() const { return p[n - 1]; } size_t size() const { return n; } bool empty() const { return n == 0; } void resize_down(size_t k) { n = k; }
    T *begin() { return p; } T *end() { return p + n; } const T *begin() const { return p; } const T *end() const { return p + n; } T *data() { return p; } const T *data() const { return p; }
};

unsigned cgroup_cpu_quota(const char *proc_cgroup, const char *sys_root);   // tightest CPU quota over the process's control groups (v1 and v2, nested), 0 = none
unsigned usable_cpus();                   // CPUs this process can keep busy: its affinity mask capped by the CPU quota of its control group
unsigned loader_threads(size_t bytes);      // AGX_LOAD_THREADS, else by the size of the input and the cores this process may use
// walkers of a unit of n positions at most (agx_walk.cpp: walk_split): AGX_WALK_SPLIT_WALKERS (two to sixteen), else one per 1.2 M positions, two to eight.  The engine hands
// out one per CPU this process may use at most (agx_engine.cpp: walkers_now).
inline thread_local int walkers_cap = 8;      // set by the engine for the walk that begins on this thread: sixteen when no other unit is on its way (the tail of a job), else eight
inline int walkers_for(size_t n) {
    const char *e = getenv("AGX_WALK_SPLIT_WALKERS");
    // (r04: eight by default, sixteen on request.  Measured on the cfg3 job: 35.43 ms and 386 CPU-ms per job with up to sixteen, 35.58 ms and 299 CPU-ms with eight, 38.98 ms with
    // four — every walker beyond what the walk's length needs is a window copy, a warm-up and a thread that spins up for nothing)
    const int k = e ? atoi(e) : (int)std::min<size_t>(n / 1200000u, (size_t)walkers_cap);
    return k < 2 ? 2 : k > GraphView::MAX_WALKERS ? (int)GraphView::MAX_WALKERS : k;
}

// agx_load.cpp — the fast loaders.  Each returns false when the input is anything but the well-formed common case (an '@' line, an empty line in the
// middle, ids out of order, blocks that overlap, whatever would be an error): the caller then takes the general loader of agx_host.cpp, which follows
// the reference line by line and reports errors in its order.  What they return is byte for byte what the general loader + staging produce.
struct ReadsIndex;
enum { SA_HITS = 0, SA_SIDES, SA_RUNS, SA_CODES, SA_OTHER, SA_JUMP, SA_N };
struct StageSink { virtual void *take(int which, size_t bytes) = 0; virtual ~StageSink() {} };      // where the staged arrays live (the engine: pinned memory)
struct StagedPairs {        // what the upload wants of a unit's read alignments, in the wire formats of agx_core.h (agx_engine.cpp: stage)
    agx_whit *hits = nullptr; size_t nh = 0;             // row = ROW of the left mate's bases, AGX_WF_LEFT2 = which mate that is
    agx_wside *sides = nullptr; size_t n_sides = 0;      // one per hit with a multi-run mate
    agx_wrun *runs = nullptr; size_t n_runs = 0;
    agx_u8 *codes = nullptr; size_t n_codes = 0;         // 2-bit classes, stride / 4 bytes per row
    unsigned long long *other = nullptr; size_t n_other = 0;   // bases that are not A, C, G, T: row * stride + index, ascending
    agx_u32 *jump = nullptr; size_t n_jump = 0;          // the hits whose LEFT mate has two or more runs, ascending: the only ones pass J of the edge build has to look at (agx_edge_jump_hit)
    agx_u32 stride = 0, maxlen = 0, n_rows = 0;          // stride: bases per row = the longest read rounded up to 4
    std::vector<uint64_t> row_off;                       // fast loader: where each row's bases start in the reads file
    std::vector<agx_u32> row_slot;                       // general loader: the read slot (Pairs::bases) of each row
    unsigned long long n_pairs_in_file = 0, n_sam_pairs = 0;
};
bool thread_contigs_fast(const std::string &contigs_fa, const std::string &psl, Threads &T);      // T.ref must hold the unit sequence (left as it was on false)
bool load_pairs_fast(const ReadsIndex &reads, const std::string &sam, long batch, agx_u32 k, unsigned threads, StageSink &sink, StagedPairs &S);
// the same arrays from what the general loader (or agx_unit_push_pairs) holds; staged_out (tests): the hits as the device will unpack them
void stage_pairs(const Pairs &P, agx_u32 k, unsigned threads, StageSink &sink, StagedPairs &S, std::vector<agx_hit> *staged_out = nullptr);
// The staged hits in the order of the tile their first arrival falls in (agx_core.h "hits in tile order"): perm[i] = the i-th hit of that order (hit numbers ascend inside a
// tile), tile_first[t] = hits in front of tile t's own (n_tiles + 2 entries, the last two = nh), jump_at = pass J's list (`jump`, hit numbers) as places in the order, ascending
void order_hits(const agx_whit *hits, size_t nh, const agx_wside *sides, const agx_wrun *runs, const agx_u32 *jump, size_t n_jump, size_t n_pos, unsigned threads,
                agx_u32 *perm, agx_u32 *tile_first, agx_u32 *jump_at);
// reference bases as 2 bits each + the stretches of other bytes (agx_core.h); false: too many such stretches (soft-masked sequence): the bases cross as they are
// what the device builds the conti-mer tables from besides the runs themselves (agx_core.h: agx_cntrun, agx_chunk): the counts as runs, and both run lists in chunks
struct CmLayout { std::vector<agx_cntrun> cnt_runs; std::vector<agx_chunk> cnt_chunks, seg_chunks; std::vector<agx_u32> seg_index; };
void build_seg_index(const agx_cmseg *segs, size_t n_seg0, size_t n_pos, std::vector<agx_u32> &index);      // agx_compact_args::seg_index
void build_cm_layout(const agx_u8 *cm_cnt, size_t n_pos, const agx_cmseg *segs, size_t n_segs, CmLayout &L);      // (seg_index is built separately: it needs n_seg0)
bool pack_reference(const char *ref, size_t n, unsigned threads, agx_u8 *packed /* (n + 3) / 4 + 16 bytes */, std::vector<agx_refx> &others);
// The read rows in their upload form (agx_core.h "read rows relative to the reference").  cnt: one count byte per row, padded with zeros to whole 64-row blocks + 64;
// block_off[i]: units in front of row 64 i (and the total behind the last block); anchor_bits: bit h = hit h is its row's anchor; units: the stream.
// block_first[i]: the anchor of row 64 i.
struct RowDiffs { std::vector<agx_u8> cnt; std::vector<agx_u32> block_off, block_first, anchor_bits; std::vector<agx_u16> units; size_t n_units = 0, n_explicit = 0; };
// wref: the packed reference (pack_reference) as 32-bit words, readable one word beyond position n_pos + 15.  false: the form does not apply.
bool build_row_diffs(const agx_whit *hits, size_t nh, const agx_wside *sides, size_t n_sides, const agx_wrun *runs, size_t n_runs, const agx_u8 *codes2, size_t n_rows, agx_u32 stride,
                     const agx_u32 *wref, size_t n_pos, unsigned threads, RowDiffs &D, bool rows_are_hits = false);

// tmp/_agx_pairs.<u>.bin — a unit's read alignments handed over STAGED (the wire formats of agx_core.h) instead of as SAM text + tmp/_reads.fa: what an aligner that is
// linked with the engine (or a generator of synthetic alignments: tools/agx_synth.cpp) writes where the reference's flow distributes SAM lines (AG:3545-3579).  The arrays are
// exactly what the loaders + staging produce from the text (tests/test_staged_pairs.py compares them byte for byte); they depend on k (which mate is the left one) and on
// BATCH (which line pairs the batch boundaries drop), both in the header.  agx_unit_load_files takes the file when it is there and does not look for the two text files.
namespace pairsfile {
enum { S_HITS = 0, S_SIDES, S_RUNS, S_CODES, S_OTHER, S_OTHERB, S_JUMP, S_N };
struct Header {
    char magic[8]; agx_u32 version, k, batch, stride, maxlen, n_rows;
    unsigned long long nh, n_sides, n_runs, n_codes, n_other, n_jump, pairs_in_file, sam_pairs;
    agx_u32 sizes[3];                            // sizeof agx_whit, agx_wside, agx_wrun
    unsigned long long off[S_N], len[S_N];
};
const char MAGIC[8] = {'A', 'G', 'X', 'P', 'A', 'I', 'R', '1'};
inline std::string path_of(const std::string &dir, int unit) { return dir + "/_agx_pairs." + std::to_string(unit) + ".bin"; }
}  // namespace pairsfile
// other_bytes[i] = the byte of listed base i (S.other[i]).  Throws Error{E_IO} if the file cannot be written.
void write_pairs_file(const std::string &path, const StagedPairs &S, const agx_u8 *other_bytes, agx_u32 k, agx_u32 batch);
std::vector<agx_u8> other_bytes_of(const Pairs &P, const StagedPairs &S);      // from the general loader's containers (S.row_slot names P's read slots)
// maps and checks the header and the section lengths (not the content); false: no such file.  Throws Error{E_FORMAT} on a file that is not one.
struct PairsFile { std::unique_ptr<FileView> fv; pairsfile::Header H; const char *sec(int i) const { return fv->p + H.off[i]; } };
bool open_pairs_file(const std::string &path, PairsFile &F);

// agx_host.cpp
void load_unit_reference(const std::string &path, std::string &ref);
void thread_contigs_from_files(const std::string &contigs_fa, const std::string &psl, Threads &T);   // T.ref must hold the unit sequence
// tmp/_reads.fa mapped once with the byte offset of every record: the units of one run (AG:1880 re-reads the whole file for each of
// them) then only touch the reads their own SAM names.  Immutable after open(); shared by any number of threads.
struct ReadsIndex {
    FileView fv;
    SBuf<uint64_t> rec_off;                     // offset of the first line of record r (a record = two lines)
    unsigned long long headers = 0;             // header lines up to the first empty line
    explicit ReadsIndex(const std::string &path);
};
ReadsIndex *reads_index_open(const std::string &reads_fa);
void reads_index_close(ReadsIndex *);
// reads: optional index of reads_fa (nullptr: the file is scanned here)
void load_pairs_from_files(const std::string &reads_fa, const std::string &sam, long batch, agx_u32 k, Pairs &P, const ReadsIndex *reads = nullptr);

// agx_walk.cpp
// A second thread the walk may hand work to (the engine: the unit's helper thread): run(f) starts f there, wait() returns when it is done.
// The walk itself is sequential by specification; copying the written records' bases into the outputs is not.
struct Assistant {          // (helpers(): how many threads there are; `who` picks one)
    virtual void run(std::function<void()> f, int who = 0) = 0; virtual void wait(int who = 0) = 0; virtual int helpers() const { return 1; } virtual ~Assistant() {}
};
void walk_join_scaffold(const UnitView &V, const GraphView &G, UnitOutput &out, Assistant *assistant = nullptr);

}  // namespace agx
