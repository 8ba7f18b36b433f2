// agx_host.cpp — host-side loaders: the reference's tmp/ text files -> packed arrays for the device.
//
// Mirrors, for one unit, loadGenome (AG:287-320), loadContigAlignment (AG:1219-1231: loadSeq AG:322-359,
// loadContiAli AG:817-852 with parseBLAT AG:406-522 and updateContig AG:763-815, updateGenomeWithContig
// AG:884-1217) and the parsing half of loadReadAlignment (loadSeq AG:361-404, loadReadAli AG:1233-1277 with
// parseBOWTIE AG:181-285).  AG = /root/reference/AlignGraph/AlignGraph.cpp.
//
// Contig threading stays on the host on purpose: it is ~1 % of the reference's time, inherently ordered
// (a placement is skipped when it meets a position that already carries two conti-mers, AG:908-920) and its
// result is a static, read-only input of the kernels.
//
// Inputs the reference would mis-handle by reading out of bounds are rejected here with an error instead
// (listed in DESIGN.md "Rejected inputs"): SAM not sorted by read id, CIGAR length != read length,
// a RNAME / tName that does not resolve to the unit sequence, alignments beyond the unit sequence.
#include "agx_host.h"
#include <malloc.h>
#include <map>
#include <mutex>
#include "agx_parse.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <fstream>
#include <map>
#include <memory>
#include <system_error>
#include <thread>
#include <pthread.h>
#include <mutex>
#include <condition_variable>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace agx {

FileView::FileView(const std::string &path) {
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error{E_IO, "CANNOT OPEN FILE! (" + path + ")"};
    struct stat st; if (fstat(fd, &st) != 0) { ::close(fd); throw Error{E_IO, "cannot stat " + path}; }
    n = (size_t)st.st_size;
    if (n) {
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); throw Error{E_IO, "cannot map " + path}; }
        p = (const char *)m; mapped = true;
        madvise(m, n, MADV_SEQUENTIAL);
    }
}
FileView::~FileView() { if (mapped) munmap((void *)p, n); if (fd >= 0) ::close(fd); }

namespace {

struct ContigSeq {
    std::string nuc; int real_id = 0; int placed = 0;
    std::vector<std::vector<agx_u32> > sets;   // per placement: per-base reference offset or NONE
    std::vector<int> fr;
};

// keepPositions (AG:731-748) on the last placement of contig `id`
bool keeps_last(const std::vector<ContigSeq> &c, agx_u32 id, double thr) {
    if (id == AGX_NONE || c[id].sets.empty()) return true;
    const std::vector<agx_u32> &last = c[id].sets.back();
    size_t m = 0; for (agx_u32 v : last) m += v != AGX_NONE;
    return (double)m / (double)last.size() >= thr;
}

}  // namespace

// ---- output buffers kept between units (agx_host.h: OutBuf) ----------------------------------------------------------------------------
namespace {
struct OutCache { std::mutex m; std::multimap<size_t, void *> kept; size_t bytes = 0; };
OutCache &out_cache() { static OutCache c; return c; }
}
void *out_cache_take(size_t need, size_t &cap) {
    OutCache &C = out_cache(); std::lock_guard<std::mutex> g(C.m);
    auto it = C.kept.lower_bound(need);
    if (it == C.kept.end() || it->first > 2 * need + ((size_t)16 << 20)) return nullptr;
    void *p = it->second; cap = it->first; C.bytes -= it->first; C.kept.erase(it);
    return p;
}
void out_cache_give(void *p) {
    if (!p) return;
    const size_t n = malloc_usable_size(p);
    OutCache &C = out_cache();
    { std::lock_guard<std::mutex> g(C.m);
      if (n >= ((size_t)1 << 20) && C.kept.size() < 128 && C.bytes + n <= ((size_t)16 << 30) && !getenv("AGX_NO_OUT_CACHE")) { C.kept.emplace(n, p); C.bytes += n; return; } }
    free(p);
}
void out_cache_trim() {
    OutCache &C = out_cache(); std::multimap<size_t, void *> all;
    { std::lock_guard<std::mutex> g(C.m); all.swap(C.kept); C.bytes = 0; }
    for (auto &kv : all) free(kv.second);
}

void advise_huge(void *p, size_t n) {
#if defined(__linux__) && defined(MADV_HUGEPAGE)
    const size_t huge = (size_t)2 << 20;
    const uintptr_t lo = ((uintptr_t)p + huge - 1) & ~(uintptr_t)(huge - 1), hi = ((uintptr_t)p + n) & ~(uintptr_t)(huge - 1);
    if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
#else
    (void)p; (void)n;
#endif
}

// loadGenome, AG:287-320
void load_unit_reference(const std::string &path, std::string &ref) {
    FileView fv(path); LineReader in(fv.p, fv.n);
    const char *s; size_t n; int records = 0;
    { std::string fresh; fresh.reserve(fv.n + fv.n / 64 + 4096); advise_huge(&fresh[0], fresh.capacity()); ref.swap(fresh); }      // (room for the positions contig threading appends; huge pages: 30 MB of fresh 4 KB pages cost more than reading the file)
    while (in.next(s, n)) {
        if (s[0] == '>') { if (++records > 1) throw Error{E_UNSUPPORTED, "unit genome file holds more than one record"}; continue; }
        if (!records) throw Error{E_FORMAT, "unit genome file has no header line"};
        ref.append(s, n);
    }
}

// loadContigAlignment, AG:1219-1231
void thread_contigs_from_files(const std::string &contigs_fa, const std::string &psl_path, Threads &T) {
    std::vector<ContigSeq> cs;
    const bool lt = getenv("AGX_LOAD_TIMING") != nullptr; auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }; double tt[8]; tt[0] = tnow();
    {   // loadSeq, AG:322-359
        FileView fv(contigs_fa); LineReader in(fv.p, fv.n); const char *s; size_t n;
        while (in.next(s, n)) {
            if (s[0] == '>') {
                size_t i = 0; while (i < n && s[i] != '.') i++;
                if (i == n) throw Error{E_FORMAT, "contig header without '.' in " + contigs_fa};
                cs.emplace_back(); cs.back().real_id = to_int(s + i + 1, n - i - 1);
            } else { if (cs.empty()) throw Error{E_FORMAT, "sequence before header in " + contigs_fa}; cs.back().nuc.append(s, n); }
        }
    }
    const agx_u32 n_ref = (agx_u32)T.ref.size();
    T.n_ref = n_ref; tt[1] = tnow();
    {   // loadContiAli, AG:817-852 (sourceIDBak starts at -1 for every unit, AG:4781)
        FileView fv(psl_path); LineReader in(fv.p, fv.n); const char *s; size_t n;
        Psl r; std::vector<agx_run> seg; agx_u32 bak = AGX_NONE, last_parsed = AGX_NONE;
        while (in.next(s, n)) {
            parse_psl_line(s, n, r, seg); last_parsed = r.sID;
            const bool keep = (double)(agx_u32)(r.sEnd - r.sStart - r.sGap) / r.sSize >= 0.5 &&
                              (double)(agx_u32)(r.tEnd - r.tStart - r.tGap) / (agx_u32)(r.tEnd - r.tStart) >= 0.5 && r.sSize > 200;
            if (!keep) continue;
            if (r.tID != 0) throw Error{E_UNSUPPORTED, "PSL target is not the unit sequence"};
            if (r.sID >= cs.size()) throw Error{E_FORMAT, "PSL names a contig that is not in _contigs.fa"};
            ContigSeq &q = cs[r.sID];
            for (const agx_run &g : seg) {
                if ((size_t)g.q + g.n > q.nuc.size() || g.q == AGX_NONE) throw Error{E_FORMAT, "PSL block beyond the contig"};
                if ((unsigned long long)g.t + g.n > n_ref) throw Error{E_FORMAT, "PSL block beyond the unit sequence"};
            }
            auto open_set = [&]() { q.sets.emplace_back(q.nuc.size(), AGX_NONE); q.fr.push_back((int)r.fr); };
            if (r.sID != bak) {                                                       // updateContig, AG:772-785
                if (!keeps_last(cs, bak, 0.5)) { cs[bak].sets.pop_back(); cs[bak].fr.pop_back(); }
                open_set(); bak = r.sID;
            } else {                                                                  // AG:786-806: a block that meets a filled base opens a new placement
                bool hit = false;
                for (const agx_run &g : seg) { for (agx_u32 i = g.q; i < g.q + g.n && !hit; i++) hit = q.sets.back()[i] != AGX_NONE; if (hit) break; }
                if (hit) { if (!keeps_last(cs, r.sID, 0.5)) { q.sets.pop_back(); q.fr.pop_back(); } open_set(); }
            }
            std::vector<agx_u32> &set = q.sets.back();
            for (const agx_run &g : seg) for (agx_u32 i = 0; i < g.n; i++) set[g.q + i] = g.t + i;
        }
        // AG:830-836: reached only through an empty line, i.e. when the file is empty or ends with '\n'
        const bool through_empty_line = fv.n == 0 || fv.p[fv.n - 1] == '\n' || in.p < in.e;
        if (through_empty_line && last_parsed != AGX_NONE && last_parsed < cs.size() && !keeps_last(cs, last_parsed, 0.5)) cs[last_parsed].sets.pop_back();
    }

    tt[2] = tnow();
    // updateGenomeWithContig, AG:884-1217 — per-position conti-mer lists kept as index-linked pools while threading
    struct Cell { ContiMer m; int next; };
    std::vector<Cell> pool;
    { size_t bases = 0; for (const ContigSeq &q : cs) bases += q.nuc.size() * (q.sets.empty() ? 0 : 1); pool.reserve(bases + cs.size() + 16); }     // about one conti-mer per placed contig base
    struct Slot { int head, tail; agx_u32 count; };      // one line per position instead of three arrays: threading touches all three together
    std::vector<Slot> slot(n_ref, Slot{-1, -1, 0});
    auto push_cm = [&](agx_u32 x, const ContiMer &m) {
        const int id = (int)pool.size(); pool.push_back(Cell{m, -1});
        Slot &s = slot[x];
        if (s.tail < 0) s.head = id; else pool[s.tail].next = id;
        s.tail = id; s.count++;
    };
    auto push_pos = [&](char nuc) { T.ref.push_back(nuc); slot.push_back(Slot{-1, -1, 0}); };
    agx_u32 off = 0, next_off = AGX_NONE; bool has_next = false;     // function-scope in the reference: survive from one placement to the next
    for (size_t sp = 0; sp < cs.size(); sp++) {
        ContigSeq &q = cs[sp];
        size_t pp = 0;
    again:
        for (; pp < q.sets.size(); pp++) {
            const std::vector<agx_u32> &set = q.sets[pp];
            const size_t len = set.size();
            for (size_t e = 0; e < pp; e++) if (agx_absdiff(set[0], q.sets[e][0]) < (int)q.nuc.size()) { pp++; goto again; }      // AG:902-907
            for (size_t i = 0; i + 1 < len; i++) if (set[i] != AGX_NONE && slot[set[i]].count >= 2) { pp++; goto again; }              // AG:908-920
            if (pp >= q.fr.size()) throw Error{E_FORMAT, "contig placement without strand"};
            const bool rc = q.fr[pp] == 1;
            if (rc) rc_inplace(q.nuc);
            q.placed = 1;
            size_t i;
            for (i = 0; i + 1 < len; i++) {
                if (set[i] == AGX_NONE) continue;
                off = set[i]; next_off = set[i + 1]; has_next = next_off != AGX_NONE;
                const char nuc = q.nuc[i];
                if (!has_next) {                                                        // contig bases missing from the reference, AG:940-1045 ("large insertion" arm; SI=0)
                    for (size_t np = i + 2; np < len; np++) {
                        if (set[np] == AGX_NONE) continue;
                        has_next = true; next_off = set[np];
                        push_cm(off, ContiMer{nuc, (agx_u32)sp, (agx_u32)i, (agx_u32)T.ref.size(), 0});
                        for (size_t j = 0; j + 2 < np - i; j++) {
                            const char b = q.nuc[i + 1 + j];
                            push_pos(b);
                            push_cm((agx_u32)T.ref.size() - 1, ContiMer{b, (agx_u32)sp, (agx_u32)(i + 1 + j), (agx_u32)T.ref.size(), 0});
                        }
                        const char b = q.nuc[np - 1];
                        push_pos(b);
                        push_cm((agx_u32)T.ref.size() - 1, ContiMer{b, (agx_u32)sp, (agx_u32)(np - 1), next_off, slot[next_off].count});
                        i = np - 1;
                        break;
                    }
                } else {
                    push_cm(off, ContiMer{nuc, (agx_u32)sp, (agx_u32)i, next_off, slot[next_off].count});       // ordinary AG:1099-1118 and deletion AG:1075-1097 (SD=0)
                }
            }
            const agx_u32 at = has_next ? next_off : off;                                                  // terminal conti-mer carries the reference base, AG:1121-1148
            push_cm(at, ContiMer{T.ref[at], (agx_u32)sp, (agx_u32)i, AGX_NONE, AGX_NONE});
            if (rc) rc_inplace(q.nuc);
        }
    }
    tt[3] = tnow();
    const size_t n_pos = T.ref.size();
    T.cm_start.assign(n_pos + 1, 0); T.cm.clear(); T.cm.reserve(pool.size());
    for (size_t x = 0; x < n_pos; x++) { T.cm_start[x] = (agx_u32)T.cm.size(); for (int c = slot[x].head; c >= 0; c = pool[c].next) T.cm.push_back(pool[c].m); }
    T.cm_start[n_pos] = (agx_u32)T.cm.size(); tt[4] = tnow();
    build_chains(T); tt[5] = tnow();

    // tmp/_initial_contigs.<u>.fa, AG:1179-1216: runs of equal realID form one real contig; it is written when >= 50 % of its
    // chunks were placed on this unit
    T.initial_contigs.clear();
    std::vector<std::string> real; std::vector<int> placed, total; int id_bak = -1;
    for (const ContigSeq &q : cs) {
        if (q.real_id != id_bak) { real.emplace_back(); placed.push_back(0); total.push_back(0); id_bak = q.real_id; }
        total.back()++; placed.back() += q.placed; real.back() += q.nuc;
    }
    for (size_t g = 0; g < real.size(); g++)
        if ((double)placed[g] / (double)total[g] >= 0.5) {
            T.initial_contigs += ">" + std::to_string(g) + "\n";
            fasta_body(T.initial_contigs, real[g].data(), real[g].size());
        }
    if (lt) fprintf(stderr, "[agx load] contigs: fasta %.1f ms, psl + sets %.1f ms, threading %.1f ms, flatten %.1f ms, chains %.1f ms, initial %.1f ms\n", tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3], tt[5] - tt[4], tnow() - tt[5]);
}

// loadReadAlignment's parsing half: batches of `batch` pairs (AG:37, 361-404), SAM line pairs (AG:1233-1277)
// One pass over tmp/_reads.fa: where every line pair (header, sequence) starts, up to the first empty line (which ends the file for
// the reference's getline loops), and how many header lines there are (what decides the batch boundaries, AG:361-404).
// CPUs this process can keep busy: its affinity mask, capped by the CPU quota of its control group (a container on a 256-thread host may be
// allowed 16 CPUs' worth of time per period — threads beyond that only take turns being throttled).
// cgroup_cpu_quota: the tightest quota between the process's own group and the root of the hierarchy, in CPUs rounded up; 0 when there is
// none.  `proc_cgroup` is /proc/self/cgroup ("0::/a/b" for the unified hierarchy; "N:cpu,cpuacct:/a/b" for version 1), `sys_root` is
// /sys/fs/cgroup.  Version 2 keeps "quota period" (or "max period") in cpu.max; version 1 keeps cpu.cfs_quota_us (-1 = none) and
// cpu.cfs_period_us under the cpu controller's own mount.  A group path the mount does not show (a container sees its own group as the
// root) falls back to the files at the mount's root.
static long long read_ll(const std::string &path, bool *is_max = nullptr, long long *second = nullptr) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return -2;
    char q[64] = {0}; long long b = 0;
    const int got = fscanf(f, "%63s %lld", q, &b);
    fclose(f);
    if (got < 1) return -2;
    if (is_max) *is_max = strcmp(q, "max") == 0;
    if (second) *second = got == 2 ? b : 0;
    return atoll(q);
}
unsigned cgroup_cpu_quota(const char *proc_cgroup, const char *sys_root) {
    std::ifstream in(proc_cgroup);
    std::string line; long long best = 0;
    auto take = [&](long long quota, long long period) { if (quota > 0 && period > 0) { const long long c = std::max<long long>(1, (quota + period - 1) / period); if (!best || c < best) best = c; } };
    auto walk_up = [&](const std::string &mount, std::string group, auto &&read_one) {           // the group itself, then every ancestor up to the mount
        bool any = false;
        for (;;) {
            any |= read_one(mount + group);
            if (group.empty() || group == "/") break;
            const size_t cut = group.rfind('/');
            group = cut == std::string::npos || cut == 0 ? std::string() : group.substr(0, cut);
        }
        return any;
    };
    while (std::getline(in, line)) {
        const size_t a = line.find(':'), b = a == std::string::npos ? a : line.find(':', a + 1);
        if (b == std::string::npos) continue;
        const std::string ctl = "," + line.substr(a + 1, b - a - 1) + ",";
        std::string group = line.substr(b + 1);
        if (group.find("..") != std::string::npos) group = "/";
        const std::string root = sys_root;
        if (ctl == ",,") {                                                                          // unified hierarchy: at the mount itself, or under "unified" beside version 1
            auto v2 = [&](const std::string &dir) { bool mx = false; long long period = 0; const long long q = read_ll(dir + "/cpu.max", &mx, &period); if (q == -2) return false; if (!mx) take(q, period); return true; };
            if (!walk_up(root, group, v2)) v2(root);
        } else if (ctl.find(",cpu,") != std::string::npos) {
            auto v1 = [&](const std::string &dir) { const long long q = read_ll(dir + "/cpu.cfs_quota_us"); if (q == -2) return false; take(q, read_ll(dir + "/cpu.cfs_period_us")); return true; };
            bool any = false;
            for (const char *mnt : {"/cpu", "/cpu,cpuacct"}) any |= walk_up(root + mnt, group, v1);
            if (!any) for (const char *mnt : {"/cpu", "/cpu,cpuacct"}) v1(root + mnt);
        }
    }
    return (unsigned)std::min<long long>(best, 1 << 20);
}
unsigned usable_cpus() {
    static const unsigned cached = [] {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set; CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = (unsigned)c; }
        if (const unsigned q = cgroup_cpu_quota("/proc/self/cgroup", "/sys/fs/cgroup")) n = std::min(n, q);
        return n;
    }();
    return cached;
}
// A unit's loader runs beside the loaders of the other units in flight (AlignGraph_amd: four; bench.py: all of a rank's units): a quarter of the
// CPUs this process can keep busy each, between 4 and 32, and no more than one per 2 MB of input.
unsigned loader_threads(size_t bytes) {
    if (const char *e = getenv("AGX_LOAD_THREADS")) return (unsigned)std::min(64, std::max(1, atoi(e)));      // tests force the multi-thread paths on small files
    const unsigned cores = usable_cpus(), share = std::min(32u, std::max(4u, cores / 4));
    return (unsigned)std::min<size_t>(std::min(share, cores), bytes / (2u << 20) + 1);
}
// fn(t) on `threads` threads.  Nothing may leave a worker thread as an exception (it would terminate the process behind a C ABI that promises
// return codes): whatever a worker throws is carried to the caller and rethrown there; a thread that cannot be started just leaves its
// share to be done here.
template <class F> void on_threads(unsigned threads, F fn) {
    std::vector<std::thread> th; std::vector<std::exception_ptr> ex(threads);
    auto guarded = [&](unsigned t) { try { fn(t); } catch (...) { ex[t] = std::current_exception(); } };
    std::vector<unsigned> mine{0u};
    for (unsigned t = 1; t < threads; t++) { try { th.emplace_back(guarded, t); } catch (const std::system_error &) { mine.push_back(t); } }
    for (unsigned t : mine) guarded(t);
    for (auto &x : th) x.join();
    for (auto &e : ex) if (e) std::rethrow_exception(e);
}

// One record = a header line and a sequence line; the index holds where every record starts, up to the first empty line (which ends the file for
// the reference's getline loops), and how many header lines there are (what decides the batch boundaries, AG:361-404).  Two passes at SIMD speed
// (count the lines of every byte range, then write every other line start to its place) on the cores this process can keep busy.
ReadsIndex::ReadsIndex(const std::string &path) : fv(path) {
    const char *b = fv.p, *e = b + fv.n;
    const unsigned threads = (unsigned)std::min<size_t>(getenv("AGX_LOAD_THREADS") ? loader_threads(fv.n) : usable_cpus(), fv.n / (1u << 20) + 1);
    bool done = false;
    if (threads > 1 && fv.n) {                // byte ranges cut at line starts; a file with an empty line takes the one-thread scan
        Team team(threads);
        const unsigned n_thr = team.size(), nr = n_thr * 4;
        struct alignas(64) Range { const char *lo, *hi; size_t lines = 0, heads = 0; bool clean = true; };
        std::vector<Range> R(nr);
        auto line_start = [&](const char *c) -> const char * { if (c <= b) return b; const char *nl = (const char *)memchr(c - 1, '\n', (size_t)(e - (c - 1))); return nl ? nl + 1 : e; };
        for (unsigned t = 0; t < nr; t++) R[t].lo = line_start(b + fv.n / nr * t);
        for (unsigned t = 0; t < nr; t++) R[t].hi = t + 1 < nr ? R[t + 1].lo : e;
        std::atomic<unsigned> next{0};
        team.run([&](unsigned) {
            for (unsigned i; (i = next.fetch_add(1)) < nr;) {
                Range &r = R[i]; size_t n = 0, h = 0;
                const char *stop = scan_lines(r.lo, r.hi, [&](const char *c) { n++; h += *c == '>'; });
                // (scan_lines also stops at a line that begins with a NUL byte; the one-thread scan below decides what that means here)
                r.lines = n; r.heads = h; r.clean = stop == r.hi;
            }
        });
        bool clean = true; size_t total = 0;
        std::vector<size_t> before(nr, 0);
        for (unsigned t = 0; t < nr; t++) { clean &= R[t].clean; before[t] = total; total += R[t].lines; headers += R[t].heads; }
        if (clean) {
            rec_off.reserve((total + 1) / 2 + 1); rec_off.n = (total + 1) / 2;
            next.store(0);
            team.run([&](unsigned) {
                for (unsigned i; (i = next.fetch_add(1)) < nr;) {
                    size_t line = before[i]; uint64_t *out = rec_off.p;
                    scan_lines(R[i].lo, R[i].hi, [&](const char *c) { if ((line & 1) == 0) out[line >> 1] = (uint64_t)(c - b); line++; });
                }
            });
            done = true;
        } else headers = 0;
    }
    if (!done) {
        const char *c = b; unsigned long long line = 0;
        while (c < e) {
            const char *nl = (const char *)memchr(c, '\n', (size_t)(e - c));
            if (nl == c) break;                  // empty line
            if (*c == '>') headers++;
            if ((line & 1ull) == 0) rec_off.push_back((uint64_t)(c - b));
            line++;
            if (!nl) break;
            c = nl + 1;
        }
    }
    if (fv.mapped) madvise((void *)fv.p, fv.n, MADV_RANDOM);
}

ReadsIndex *reads_index_open(const std::string &reads_fa) { return new ReadsIndex(reads_fa); }
void reads_index_close(ReadsIndex *r) { delete r; }

void load_pairs_from_files(const std::string &reads_fa, const std::string &sam_path, long batch, agx_u32 k, Pairs &P, const ReadsIndex *reads) {
    P = Pairs();
    std::unique_ptr<ReadsIndex> own_reads(reads ? nullptr : new ReadsIndex(reads_fa));       // no shared index: build one for this unit
    if (!reads) reads = own_reads.get();
    const FileView &rf = reads->fv;
    FileView sf(sam_path);
    if (batch <= 0) batch = 1000000;

    // ---- pass 1 over the SAM: kept hits, in order, with the batch-boundary rule applied (PairRules, agx_parse.h) ----------------
    PairRules rules(P, reads->headers / 2, batch);      // (reads->headers: header lines up to the first empty line — decides where the last batch ends)
    std::vector<agx_u32> &hit_id = rules.hit_id;
    auto consume = [&](const Mate &m1, const Mate &m2, const agx_run *src) -> bool { return rules.consume(m1, m2, src); };

    // Parsing the text (field splitting, CIGAR walk) is the expensive part and independent per line pair; the rules above are cheap but
    // sequential.  Large, clean files (no '@' lines, no empty line before the end) are therefore cut into byte ranges aligned to line
    // pairs, parsed on several threads into per-range buffers and consumed in order.  Anything else takes the one-thread path below.
    struct Range { const char *lo = nullptr, *hi = nullptr; size_t lines = 0; bool odd = false, at = false, empty = false;
                   std::vector<std::pair<Mate, Mate> > pairs; std::vector<agx_run> runs; bool broken = false, failed = false; Error err{0, ""}; };
    unsigned threads = 1;
    threads = loader_threads(sf.n);
    bool parallel_done = false;
    if (threads > 1 && sf.n > 0) {
        std::vector<Range> R(threads);
        const char *fb = sf.p, *fe = sf.p + sf.n;
        auto line_start_at_or_after = [&](const char *c) -> const char * {       // first line start >= c
            if (c <= fb) return fb;
            const char *nl = (const char *)memchr(c - 1, '\n', (size_t)(fe - (c - 1)));
            return nl ? nl + 1 : fe;
        };
        for (unsigned t = 0; t < threads; t++) R[t].lo = line_start_at_or_after(fb + sf.n / threads * t);
        for (unsigned t = 0; t < threads; t++) R[t].hi = t + 1 < threads ? R[t + 1].lo : fe;
        auto for_all = [&](auto fn) { on_threads(threads, fn); };
        const bool lt = getenv("AGX_LOAD_TIMING") != nullptr; auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }; const double ta = tnow();
        for_all([&](unsigned t) {                                               // phase A: lines per range; anything that needs the one-thread path
            Range &r = R[t];
            for (const char *c = r.lo; c < r.hi;) {
                const char *nl = (const char *)memchr(c, '\n', (size_t)(fe - c));
                if (nl == c) { r.empty = true; return; }
                if (*c == '@') { r.at = true; return; }
                r.lines++;
                if (!nl) break;
                c = nl + 1;
            }
        });
        const double tb = tnow();
        bool clean = true; size_t before = 0;
        for (Range &r : R) { clean &= !r.at && !r.empty; r.odd = (before & 1) != 0; before += r.lines; }
        if (clean) {
            for_all([&](unsigned t) {                                           // phase B: parse the pairs that START in this range
                Range &r = R[t];
                auto next = [&](const char *&c, const char *&ls, size_t &ln) -> bool {
                    if (c >= fe) return false;
                    const char *nl = (const char *)memchr(c, '\n', (size_t)(fe - c));
                    ls = c; ln = (size_t)((nl ? nl : fe) - c); c = nl ? nl + 1 : fe;
                    return true;
                };
                const char *c = r.lo, *ls = r.lo; size_t ln = 0;
                if (r.odd && !next(c, ls, ln)) return;                          // the previous range's last pair ends with this range's first line
                r.pairs.reserve(r.lines / 2 + 1);
                try {
                    while (c < r.hi) {
                        std::pair<Mate, Mate> pr;
                        next(c, ls, ln); parse_sam_line(ls, ln, pr.first, r.runs);
                        if (!next(c, ls, ln)) { r.broken = true; return; }
                        parse_sam_line(ls, ln, pr.second, r.runs);
                        r.pairs.push_back(pr);
                    }
                } catch (const Error &e) { r.failed = true; r.err = e; }
            });
            const double tc = tnow();
            bool go = true;
            for (Range &r : R) {                                                // phase C: the sequential rules, in file order
                for (const auto &pr : r.pairs) if (go && !consume(pr.first, pr.second, r.runs.data())) go = false;
                if (!go) break;
                if (r.failed) throw r.err;                                      // the line pair after the last parsed one could not be parsed
                if (r.broken) throw Error{E_FORMAT, "BROKEN BOWTIE FILE"};
                std::vector<std::pair<Mate, Mate> >().swap(r.pairs); std::vector<agx_run>().swap(r.runs);
            }
            parallel_done = true;
            if (lt) fprintf(stderr, "[agx load] SAM on %u threads: line count %.1f ms, parse %.1f ms, rules %.1f ms\n", threads, tb - ta, tc - tb, tnow() - tc);
        }
    }
    if (!parallel_done) {
        LineReader in(sf.p, sf.n); const char *s; size_t n;
        Mate m1, m2; std::vector<agx_run> tmp;
        for (;;) {
            if (!in.next(s, n)) break;
            if (s[0] == '@') continue;
            tmp.clear();
            parse_sam_line(s, n, m1, tmp);
            if (!in.next(s, n)) throw Error{E_FORMAT, "BROKEN BOWTIE FILE"};
            parse_sam_line(s, n, m2, tmp);
            if (!consume(m1, m2, tmp.data())) break;
        }
    }
    P.n_kept = P.hits.size();

    // ---- pass 2 over the reads file: copy the bases of pairs that have kept hits ------------------------
    agx_u32 maxlen = 0;
    for (const agx_hit &h : P.hits) maxlen = std::max<agx_u32>(maxlen, h.len);
    P.stride = (maxlen + 15u) & ~15u;
    if (P.hits.empty()) return;
    if (k >= 32768) throw Error{E_ARG, "k too large"};
    // distinct ids in order
    size_t n_ids = 0; { agx_u32 last = 0; for (size_t i = 0; i < hit_id.size(); i++) if (i == 0 || hit_id[i] != last) { n_ids++; last = hit_id[i]; } }
    P.n_slots = (agx_u32)(2 * n_ids);
    P.bases.assign((size_t)P.n_slots * P.stride, 'N');
    // slots in hit order (sequential, cheap), then the bases of every slot through the record index (independent per slot)
    struct Slot { agx_u32 id; agx_u32 len; };
    std::vector<Slot> slots; slots.reserve(n_ids);
    for (size_t i = 0; i < P.hits.size(); i++) {
        if (i == 0 || hit_id[i] != hit_id[i - 1]) slots.push_back(Slot{hit_id[i], P.hits[i].len});
        if (P.hits[i].len != P.hits[i - P.hits[i].back].len) throw Error{E_UNSUPPORTED, "hits of one pair disagree on the read length"};
        P.hits[i].slot1 = (agx_u32)(2 * (slots.size() - 1));
    }
    const char *fb = rf.p, *fe = rf.p + rf.n;
    const unsigned copy_threads = loader_threads(slots.size() * 64);
    std::vector<size_t> bad(copy_threads, (size_t)-1); std::vector<Error> bad_err(copy_threads, Error{0, ""});
    on_threads(copy_threads, [&](unsigned t) {
        const size_t s0 = slots.size() * t / copy_threads, s1 = slots.size() * (t + 1) / copy_threads;
        for (size_t si = s0; si < s1; si++) {
            const unsigned long long r0 = 2ull * slots[si].id;
            try {
                if (r0 >= reads->rec_off.size()) throw Error{E_FORMAT, "reads file ends before a read named by the SAM"};
                for (int mate = 0; mate < 2; mate++) {
                    if (r0 + mate >= reads->rec_off.size()) throw Error{E_FORMAT, "reads file: header expected"};
                    const char *c = fb + reads->rec_off[r0 + mate];
                    if (c >= fe || *c != '>') throw Error{E_FORMAT, "reads file: header expected"};
                    const char *nl = (const char *)memchr(c, '\n', (size_t)(fe - c));
                    if (!nl || nl + 1 >= fe) throw Error{E_FORMAT, "reads file: sequence expected"};
                    const char *ls = nl + 1, *nl2 = (const char *)memchr(ls, '\n', (size_t)(fe - ls));
                    const size_t ln = (size_t)((nl2 ? nl2 : fe) - ls);
                    if (ln == 0) throw Error{E_FORMAT, "reads file: sequence expected"};
                    if (ln > P.stride || ln != slots[si].len) throw Error{E_UNSUPPORTED, "CIGAR length differs from the read length"};
                    memcpy(&P.bases[(2 * si + (size_t)mate) * P.stride], ls, ln);
                }
            } catch (const Error &e) { bad[t] = si; bad_err[t] = e; return; }
        }
    });
    for (unsigned t = 0; t < copy_threads; t++) if (bad[t] != (size_t)-1) throw bad_err[t];      // ranges are in slot order: the first failure in file order
}

// ---- Scratch ------------------------------------------------------------------------------------------------------------------------
namespace {
struct ScratchCache {
    std::mutex m; std::multimap<size_t, void *> free_; size_t held = 0;
    static size_t limit() { static const size_t l = getenv("AGX_SCRATCH_CACHE_MB") ? (size_t)atoll(getenv("AGX_SCRATCH_CACHE_MB")) << 20 : (size_t)16 << 30; return l; }
};
ScratchCache &scratch_cache() { static ScratchCache *c = new ScratchCache; return *c; }      // (never destroyed: worker threads may still give memory back at exit)
inline size_t scratch_round(size_t n) { size_t c = (size_t)2 << 20; while (c < n) c <<= 1; return c; }
}  // namespace
void Scratch::take(size_t bytes) {
    give();
    const size_t need = scratch_round(bytes ? bytes : 1);
    ScratchCache &C = scratch_cache();
    { std::lock_guard<std::mutex> l(C.m); auto it = C.free_.find(need); if (it != C.free_.end()) { p = it->second; n = need; C.held -= need; C.free_.erase(it); return; } }
    void *m = mmap(nullptr, need, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) throw Error{E_ARG, "out of host memory"};
    advise_huge(m, need);
    p = m; n = need;
}
void Scratch::give() {
    if (!p) return;
    ScratchCache &C = scratch_cache();
    bool keep;
    { std::lock_guard<std::mutex> l(C.m); keep = C.held + n <= ScratchCache::limit(); if (keep) { C.free_.emplace(n, p); C.held += n; } }
    if (!keep) munmap(p, n);
    p = nullptr; n = 0;
}
void scratch_trim() {
    ScratchCache &C = scratch_cache(); std::vector<std::pair<size_t, void *>> v;
    { std::lock_guard<std::mutex> l(C.m); for (auto &kv : C.free_) v.push_back(kv); C.free_.clear(); C.held = 0; }
    for (auto &kv : v) munmap(kv.second, kv.first);
}

// ---- Team ---------------------------------------------------------------------------------------------------------------------------
struct Team::Impl {
    std::vector<std::thread> th; std::mutex m; std::condition_variable cv_go, cv_done;
    const std::function<void(unsigned)> *fn = nullptr; unsigned long long gen = 0; unsigned pending = 0; bool stop = false;
    std::vector<std::exception_ptr> ex;
};
Team::Team(unsigned threads) : impl_(new Impl), n_(threads ? threads : 1) {
    Impl &I = *impl_; I.ex.resize(n_);
    unsigned started = 1;
    for (unsigned t = 1; t < n_; t++) {
        try {
            I.th.emplace_back([&I, t] {
                pthread_setname_np(pthread_self(), "agx-team");
                unsigned long long seen = 0;
                for (;;) {
                    const std::function<void(unsigned)> *f;
                    { std::unique_lock<std::mutex> l(I.m); I.cv_go.wait(l, [&] { return I.stop || I.gen != seen; }); if (I.stop) return; seen = I.gen; f = I.fn; }
                    try { (*f)(t); } catch (...) { I.ex[t] = std::current_exception(); }
                    { std::lock_guard<std::mutex> l(I.m); if (--I.pending == 0) I.cv_done.notify_all(); }
                }
            });
            started++;
        } catch (const std::system_error &) { break; }      // fewer threads than asked for: the team is as large as what could be started
    }
    n_ = started;
}
Team::~Team() { { std::lock_guard<std::mutex> l(impl_->m); impl_->stop = true; } impl_->cv_go.notify_all(); for (auto &t : impl_->th) t.join(); delete impl_; }
void Team::run(const std::function<void(unsigned)> &fn) {
    Impl &I = *impl_;
    for (auto &e : I.ex) e = nullptr;
    { std::lock_guard<std::mutex> l(I.m); I.fn = &fn; I.pending = n_ - 1; I.gen++; }
    I.cv_go.notify_all();
    try { fn(0); } catch (...) { I.ex[0] = std::current_exception(); }
    { std::unique_lock<std::mutex> l(I.m); I.cv_done.wait(l, [&] { return I.pending == 0; }); }
    for (auto &e : I.ex) if (e) std::rethrow_exception(e);
}

}  // namespace agx
