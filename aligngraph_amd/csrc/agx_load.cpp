// agx_load.cpp — the fast host loaders: a unit's text files -> the arrays the upload wants, at memory speed.
//
// Rows a2-a5 and a15 of the path: parseBOWTIE AG:181-285, loadSeq AG:361-404, updateContig/keepPositions AG:731-815,
// loadReadAli AG:1233-1277, and the feeders parseBLAT AG:406-522, loadContiAli AG:817-852, updateGenomeWithContig AG:884-1217.  The general
// loaders of agx_host.cpp follow the reference line by line (and base by base) and stay the definition: they take every input, and
// report every error in the reference's order.  What is here takes the well-formed common case — bowtie2 --reorder output without '@'
// lines, BLAT blocks that advance, files that end with one newline — on all the cores the process may use, and produces byte for byte
// what the general loader + the engine's staging produce (tests/test_fast_loader.py compares them on every fixture and on random
// generator settings); whatever it does not recognise makes it return false, and the caller takes the general path.
//
//   contig threading   A placement of a contig is a list of PSL blocks.  The general loader expands it to one reference offset per contig base,
//                      pushes one conti-mer per base into per-position lists, and build_chains() then recovers RUNS (agx_cmseg) from the
//                      lists.  Here the runs come straight from the blocks: the only per-position state threading needs is how many
//                      conti-mers a position already carries (the rank of the next one, AG:1106, and the two-conti-mer rule, AG:908-920) —
//                      one byte per position, scanned and bumped a block at a time.
//   read alignments    the SAM file cut into byte ranges at line-pair boundaries, every range parsed on its own thread into hits in their
//                      final form; the rules that look across lines (batch boundaries AG:1258-1259, hits per pair, rows of read bases) are
//                      prefix computations over the ranges; the left mate of every hit (AG:1672-1679) is decided where it is parsed and only
//                      its bases are fetched from the mapped reads file, straight into 2-bit classes in the upload buffer.
#include "agx_host.h"
#include <sys/stat.h>
#include <unistd.h>
#include "agx_parse.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace agx {
namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
const bool g_timing = getenv("AGX_LOAD_TIMING") != nullptr;

// length of the prefix of p[0..n) whose bytes all equal p[0]
inline size_t run_same(const agx_u8 *p, size_t n) {
    const agx_u8 v = p[0]; size_t i = 1;
    const uint64_t pat = 0x0101010101010101ull * v;
    while (i < n && ((uintptr_t)(p + i) & 7u)) { if (p[i] != v) return i; i++; }
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); if (w != pat) break; }
    while (i < n && p[i] == v) i++;
    return i;
}
inline bool any_at_least(const agx_u8 *p, size_t n, agx_u8 v) { agx_u8 m = 0; for (size_t i = 0; i < n; i++) m = p[i] > m ? p[i] : m; return m >= v; }      // (vectorises)

// ---- contig threading -------------------------------------------------------------------------------------------------------------------
struct ContigRec {
    const char *hdr = nullptr, *body = nullptr, *end = nullptr;     // header line, first byte behind it, end of the record's lines
    int real_id = 0; size_t len = (size_t)-1;                       // bases (body bytes that are not '\n'), counted when first asked for
    const char *nuc = nullptr;                                      // the bases without the line breaks (in the caller's arena), made when first asked for
    int placed = 0;
    size_t length() { if (len == (size_t)-1) len = (size_t)(end - body) - count_newlines(body, end); return len; }
    const char *bases(SBuf<char> &arena) {      // (the arena is sized for every contig that can be asked for: it never moves)
        if (!nuc) {
            char *w = arena.p + arena.n; nuc = w; arena.n += length();
            if (wrapped60()) {          // 60 bases and a line break, over and over: copies of a fixed size instead of a search for every line end
                const size_t n = length(), full = n / 60; const char *c = body;
                for (size_t l = 0; l < full; l++, c += 61, w += 60) memcpy(w, c, 60);
                if (n % 60) memcpy(w, c, n % 60);
            } else
            for (const char *c = body; c < end;) { const char *nl = (const char *)memchr(c, '\n', (size_t)(end - c)); const size_t m = (size_t)((nl ? nl : end) - c); memcpy(w, c, m); w += m; c = nl ? nl + 1 : end; }
        }
        return nuc;
    }
    // the record's lines are already what fasta_body() would write for its bases: 60 per line, every line ended
    int w60 = -1;
    bool wrapped60() {
        if (w60 >= 0) return w60 != 0;
        const size_t n = length(), lines = (n + 59) / 60;
        w60 = 0;
        if ((size_t)(end - body) != n + lines) return false;
        for (size_t l = 0; l < lines; l++) { const size_t at = l + 1 < lines ? (l + 1) * 61 - 1 : n + lines - 1; if (body[at] != '\n') return false; }
        w60 = 1;
        return true;
    }
};
struct Place { std::vector<agx_run> blk; int fr = 0; size_t filled = 0; };

inline char comp_base(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c; }

}  // namespace

bool thread_contigs_fast(const std::string &contigs_fa, const std::string &psl_path, Threads &T) {
    double tt[8]; tt[0] = now_ms();
    const agx_u32 n_ref = (agx_u32)T.ref.size();
    if (n_ref == 0) return false;
    FileView cf(contigs_fa), pf(psl_path);
    // ---- loadSeq, AG:322-359: where every record of tmp/_contigs.fa is (no base is copied yet) ----
    std::vector<ContigRec> cs;
    {
        std::vector<const char *> heads; bool body_first = false; bool first = true;
        const char *stop = scan_lines(cf.p, cf.p + cf.n, [&](const char *c) { if (*c == '>') heads.push_back(c); else if (first) body_first = true; first = false; });
        if (body_first) return false;                        // "sequence before header": the general loader reports it
        cs.resize(heads.size());
        for (size_t i = 0; i < heads.size(); i++) {
            ContigRec &r = cs[i]; r.hdr = heads[i]; r.end = i + 1 < heads.size() ? heads[i + 1] : stop;
            const char *nl = (const char *)memchr(r.hdr, '\n', (size_t)(r.end - r.hdr));
            const char *he = nl ? nl : r.end; r.body = nl ? nl + 1 : r.end;
            const char *dot = (const char *)memchr(r.hdr, '.', (size_t)(he - r.hdr));
            if (!dot) return false;
            r.real_id = to_int(dot + 1, (size_t)(he - dot - 1));
        }
    }
    tt[1] = now_ms();
    // ---- loadContiAli, AG:817-852 with updateContig, AG:763-815: placements as block lists ----
    std::vector<std::vector<Place>> sets(cs.size());
    {
        LineReader in(pf.p, pf.n); const char *s; size_t n;
        Psl r; std::vector<agx_run> seg; agx_u32 bak = AGX_NONE, last_parsed = AGX_NONE;
        auto keeps_last = [&](agx_u32 id) -> bool {           // keepPositions, AG:731-748
            if (id == AGX_NONE || sets[id].empty()) return true;
            return (double)sets[id].back().filled / (double)cs[id].length() >= 0.5;
        };
        while (in.next(s, n)) {
            parse_psl_line(s, n, r, seg); last_parsed = r.sID;
            const bool keep = (double)(agx_u32)(r.sEnd - r.sStart - r.sGap) / r.sSize >= 0.5 &&
                              (double)(agx_u32)(r.tEnd - r.tStart - r.tGap) / (agx_u32)(r.tEnd - r.tStart) >= 0.5 && r.sSize > 200;
            if (!keep) continue;
            if (r.tID != 0 || r.sID >= cs.size()) return false;
            const size_t len = cs[r.sID].length();
            if (len < 2 || len >= 0x7FFFFFFFu) return false;
            size_t q_at = 0, filled = 0;
            for (const agx_run &g : seg) {                    // blocks that advance on the contig and lie inside contig and unit (anything else: the general loader decides)
                if (g.q == AGX_NONE || g.t == AGX_NONE || g.q < q_at || (size_t)g.q + g.n > len || (unsigned long long)g.t + g.n > n_ref) return false;
                q_at = (size_t)g.q + g.n; filled += g.n;
            }
            std::vector<Place> &ps = sets[r.sID];
            auto open_set = [&]() { ps.emplace_back(); ps.back().fr = (int)r.fr; };
            if (r.sID != bak) {
                if (!keeps_last(bak)) sets[bak].pop_back();
                open_set(); bak = r.sID;
            } else {                                          // AG:786-806: a block that meets a filled base opens a new placement
                bool hit = false;
                for (const agx_run &g : seg) { if (!g.n) continue; for (const agx_run &o : ps.back().blk) if (g.q < o.q + o.n && o.q < g.q + g.n) { hit = true; break; } if (hit) break; }
                if (hit) { if (!keeps_last(r.sID)) ps.pop_back(); open_set(); }
            }
            Place &pl = ps.back();
            for (const agx_run &g : seg) if (g.n) pl.blk.push_back(g);
            pl.filled += filled;
        }
        const bool through_empty_line = pf.n == 0 || pf.p[pf.n - 1] == '\n' || in.p < in.e;       // AG:830-836
        if (through_empty_line && last_parsed != AGX_NONE && last_parsed < cs.size() && !keeps_last(last_parsed)) sets[last_parsed].pop_back();
    }
    tt[2] = now_ms();
    // ---- updateGenomeWithContig, AG:884-1217, a block at a time ----
    struct Chain { agx_u32 head_pos, head_rank, end_pos, n_elem, sp, i0; bool rc; size_t seg_first, seg_n; };
    std::vector<Chain> chains; std::vector<agx_cmseg> segs;
    size_t placed_bases = 0, group_bases = 0;
    for (size_t sp = 0; sp < cs.size(); sp++) if (!sets[sp].empty()) placed_bases += cs[sp].length();
    // conti-mers per position, in place in T.cm_cnt (fresh memory, huge pages asked for; room for the positions that contig insertions append)
    std::vector<agx_u8> &cnt = T.cm_cnt;
    { std::vector<agx_u8> fresh; fresh.reserve((size_t)n_ref + placed_bases / 8 + 4096); advise_huge(fresh.data(), fresh.capacity()); fresh.assign(n_ref, 0); cnt.swap(fresh); }
    struct Gap { agx_u32 sp, q0, n; bool rc; };            // contig bases that became new positions, in the order the positions were appended
    std::vector<Gap> gaps; size_t n_appended = 0;
    size_t n_cm = 0;
    bool failed = false;
    for (size_t sp = 0; sp < cs.size() && !failed; sp++) {
        std::vector<Place> &ps = sets[sp];
        if (ps.empty()) continue;
        ContigRec &q = cs[sp];
        const size_t len = q.length();
        for (Place &pl : ps) std::sort(pl.blk.begin(), pl.blk.end(), [](const agx_run &a, const agx_run &b) { return a.q < b.q; });
        auto set0 = [&](const Place &pl) -> agx_u32 { return !pl.blk.empty() && pl.blk[0].q == 0 ? pl.blk[0].t : AGX_NONE; };
        for (size_t pp = 0; pp < ps.size(); pp++) {
            const Place &pl = ps[pp];
            if (pl.blk.empty()) { failed = true; break; }
            bool skip = false;
            for (size_t e = 0; e < pp && !skip; e++) skip = agx_absdiff(set0(pl), set0(ps[e])) < (int)len;                       // AG:902-907
            for (size_t b = 0; b < pl.blk.size() && !skip; b++) {                                                                // AG:908-920: bases 0 .. len-2
                const agx_run &g = pl.blk[b]; const size_t n = (size_t)g.q + g.n == len ? g.n - 1 : g.n;
                skip = n && any_at_least(&cnt[g.t], n, 2);
            }
            if (skip) continue;
            if (pl.blk[0].q + 1 >= len) { failed = true; break; }      // no aligned base below len-1: the reference then works with what the previous placement left behind
            if (pl.fr != 0 && pl.fr != 1) { failed = true; break; }
            const bool rc = pl.fr == 1;
            q.placed = 1;
            const agx_u32 i0 = pl.blk[0].q, i_last = pl.blk.back().q + pl.blk.back().n - 1;
            const bool trailing = (size_t)i_last + 1 < len;      // the contig's last bases are unaligned: the terminal conti-mer sits on the last aligned base's position
            Chain ch; ch.seg_first = segs.size(); ch.n_elem = i_last - i0 + 1; ch.sp = (agx_u32)sp; ch.i0 = i0; ch.rc = rc; ch.head_pos = ch.head_rank = ch.end_pos = 0;
            // one element of the chain: the join rule of build_chains()
            auto add_elem = [&](agx_u32 pos, agx_u32 coff, agx_u32 rank, agx_u32 idx) {
                agx_cmseg *g = segs.size() > ch.seg_first ? &segs.back() : nullptr;
                const bool joins = g && pos == g->pos0 + g->len && rank == g->rank && (g->len == 1 || coff == g->coff0 + g->len * g->dcoff);
                if (joins) { if (g->len == 1) g->dcoff = coff - g->coff0; g->len++; }
                else segs.push_back(agx_cmseg{pos, 1u, (agx_u32)sp, coff, 0u, rank, idx + 1u, idx, 0u, 0u});      // hop_str0, hop_len0: relative to the chain until it is placed
            };
            // n elements on consecutive positions with consecutive contig offsets, all of rank `rank`
            auto add_run = [&](agx_u32 pos0, agx_u32 n, agx_u32 coff0, agx_u32 rank, agx_u32 idx) {
                agx_u32 j = 0;
                for (; j < n; j++) {
                    const agx_cmseg *g = segs.size() > ch.seg_first ? &segs.back() : nullptr;
                    if (g && g->len >= 2 && g->dcoff == 1 && g->rank == rank && g->pos0 + g->len == pos0 + j && g->coff0 + g->len == coff0 + j) break;
                    add_elem(pos0 + j, coff0 + j, rank, idx + j);
                }
                if (j < n) segs.back().len += n - j;
            };
            bool first = true;
            for (size_t b = 0; b < pl.blk.size() && !failed; b++) {
                const agx_run &g = pl.blk[b];
                const bool last_blk = b + 1 == pl.blk.size();
                const agx_u32 n_run = last_blk && trailing ? g.n - 1 : g.n;          // (trailing: the last aligned base's element is the terminal one, with the contig's LAST offset)
                for (agx_u32 j = 0; j < n_run;) {                                      // stretches of equal count = equal rank
                    const agx_u32 c = cnt[g.t + j]; const agx_u32 m = (agx_u32)run_same(&cnt[g.t + j], n_run - j);
                    if (c >= 250) { failed = true; break; }
                    if (first) { ch.head_pos = g.t + j; ch.head_rank = c; first = false; }
                    add_run(g.t + j, m, g.q + j, c, g.q + j - i0);
                    j += m;
                }
                if (failed) break;
                { agx_u8 *pc = &cnt[g.t]; for (agx_u32 j = 0; j < n_run; j++) pc[j]++; }
                if (last_blk) {
                    if (trailing) {
                        const agx_u32 pos = g.t + g.n - 1, c = cnt[pos];
                        if (c >= 250) { failed = true; break; }
                        if (first) { ch.head_pos = pos; ch.head_rank = c; first = false; }
                        add_elem(pos, (agx_u32)(len - 1), c, i_last - i0); cnt[pos]++;
                        ch.end_pos = pos;
                    } else ch.end_pos = g.t + g.n - 1;
                } else {
                    const agx_run &nx = pl.blk[b + 1];
                    const agx_u32 gap = nx.q - (g.q + g.n);                           // contig bases missing from the reference: new positions behind the unit (AG:974-1040; SI = 0)
                    if (gap) {
                        const size_t p0 = cnt.size();
                        if (p0 + gap >= 0xFFFFFF00ull) { failed = true; break; }
                        add_run((agx_u32)p0, gap, g.q + g.n, 0u, g.q + g.n - i0);
                        cnt.insert(cnt.end(), gap, (agx_u8)1);
                        gaps.push_back(Gap{(agx_u32)sp, g.q + g.n, gap, rc}); n_appended += gap;
                    }
                }
            }
            if (failed) break;
            ch.seg_n = segs.size() - ch.seg_first;
            n_cm += ch.n_elem;
            chains.push_back(ch);
        }
    }
    if (failed || n_cm >= 0xFFFFFFFFull) { T.cm_cnt.clear(); return false; }
    tt[3] = now_ms();
    // ---- the bases: contigs that were placed (chains, appended positions) and the members of real contigs that will be written, without their line breaks ----
    struct Group { size_t first, n; int placed; };
    std::vector<Group> groups;
    { int id_bak = -1; for (size_t i = 0; i < cs.size(); i++) { if (groups.empty() || cs[i].real_id != id_bak) { groups.push_back(Group{i, 0, 0}); id_bak = cs[i].real_id; } groups.back().n++; groups.back().placed += cs[i].placed; } }
    for (const Group &g : groups) if (g.n > 1 && (double)g.placed / (double)g.n >= 0.5) for (size_t i = g.first; i < g.first + g.n; i++) if (sets[i].empty()) group_bases += cs[i].length();
    SBuf<char> arena; arena.reserve(placed_bases + group_bases + 64);
    // a contig's bases [from, from + n) in the placement's orientation (rc: reverse complement of the stored sequence, AG:854-865)
    auto oriented = [&](agx_u32 sp, bool rc, size_t from, size_t n, char *dst) {
        const char *nuc = cs[sp].bases(arena); const size_t len = cs[sp].length();
        if (!rc) memcpy(dst, nuc + from, n);
        else for (size_t j = 0; j < n; j++) dst[j] = comp_base(nuc[len - 1 - (from + j)]);
    };
    // ---- the chains in build_chains()' order (by head position, then rank), their runs, then the runs in the device's order ----
    std::vector<size_t> order(chains.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return chains[a].head_pos != chains[b].head_pos ? chains[a].head_pos < chains[b].head_pos : chains[a].head_rank < chains[b].head_rank; });
    T.chain_off.assign(1, 0); T.chain_end_pos.clear();
    { std::string fresh; fresh.reserve(n_cm + 16); advise_huge(&fresh[0], fresh.capacity()); fresh.resize(n_cm); T.chain_str.swap(fresh); }
    T.segs.clear(); T.segs.reserve(segs.size()); T.n_seg0 = 0;
    size_t str_base = 0;
    for (size_t oi : order) {
        const Chain &ch = chains[oi];
        // the contig's bases i0 .. i_last, the last one replaced by the reference base under the terminal conti-mer (AG:1129, 1143)
        oriented(ch.sp, ch.rc, ch.i0, ch.n_elem, &T.chain_str[str_base]);
        T.chain_str[str_base + ch.n_elem - 1] = T.ref[ch.end_pos];
        for (size_t g = ch.seg_first; g < ch.seg_first + ch.seg_n; g++) {
            agx_cmseg sgm = segs[g];
            sgm.hop_str0 += (agx_u32)str_base; sgm.hop_len0 = (ch.n_elem - 1) - sgm.hop_len0; sgm.hop_end = ch.end_pos;
            T.segs.push_back(sgm);
        }
        str_base += ch.n_elem;
        T.chain_end_pos.push_back(ch.end_pos); T.chain_off.push_back(str_base);
    }
    std::stable_sort(T.segs.begin(), T.segs.end(), [](const agx_cmseg &a, const agx_cmseg &b) { return (a.rank != 0) != (b.rank != 0) ? a.rank == 0 : (a.rank == 0 && a.pos0 < b.pos0); });
    { agx_u32 e = 0; for (agx_cmseg &g : T.segs) { g.elem0 = e; e += g.len; if (g.rank == 0) T.n_seg0++; } if (e != n_cm) { T.cm_cnt.clear(); return false; } }
    T.n_ref = n_ref;
    { const size_t at = T.ref.size(); T.ref.resize(at + n_appended); size_t w = at; for (const Gap &g : gaps) { oriented(g.sp, g.rc, g.q0, g.n, &T.ref[w]); w += g.n; } }      // appended positions carry the contig bases of the gaps
    T.n_cm = n_cm; T.cm_start.clear(); T.cm.clear(); T.hop.clear();
    tt[4] = now_ms();
    // ---- tmp/_initial_contigs.<u>.fa, AG:1179-1216 ----
    T.initial_contigs.clear();
    {
        size_t total_bytes = 0;
        for (const Group &g : groups) if ((double)g.placed / (double)g.n >= 0.5) for (size_t i = g.first; i < g.first + g.n; i++) total_bytes += (size_t)(cs[i].end - cs[i].body) + 16;
        { std::string fresh; fresh.reserve(total_bytes + 64); advise_huge(&fresh[0], fresh.capacity()); T.initial_contigs.swap(fresh); }
        std::string joined;
        for (size_t gi = 0; gi < groups.size(); gi++) {
            const Group &g = groups[gi];
            if ((double)g.placed / (double)g.n < 0.5) continue;
            T.initial_contigs += ">" + std::to_string(gi) + "\n";
            if (g.n == 1 && cs[g.first].wrapped60()) { T.initial_contigs.append(cs[g.first].body, (size_t)(cs[g.first].end - cs[g.first].body)); continue; }
            joined.clear();
            for (size_t i = g.first; i < g.first + g.n; i++) {
                if (g.n > 1 || !sets[i].empty()) joined.append(cs[i].bases(arena), cs[i].length());
                else { ContigRec &r = cs[i]; for (const char *c = r.body; c < r.end;) { const char *nl = (const char *)memchr(c, '\n', (size_t)(r.end - c)); joined.append(c, (size_t)((nl ? nl : r.end) - c)); c = nl ? nl + 1 : r.end; } }
            }
            fasta_body(T.initial_contigs, joined.data(), joined.size());
        }
    }
    if (g_timing) fprintf(stderr, "[agx load] contigs (fast): scan %.1f ms, psl %.1f ms, threading %.1f ms, bases + order %.1f ms, initial %.1f ms\n", tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3], now_ms() - tt[4]);
    return true;
}

// ---- read alignments ----------------------------------------------------------------------------------------------------------------------
namespace {

// 2-bit classes of one row: `len` bases at src (file orientation), quarter = stride / 4 bytes at dst; bases that are not A, C, G, T are listed
inline void pack_row_scalar(const char *src, size_t j, size_t len, agx_u8 *dst, size_t quarter, unsigned long long row_base, std::vector<unsigned long long> &other) {
    for (; j + 4 <= len; j += 4) {
        const agx_u32 c0 = agx_base_class((agx_u8)src[j]), c1 = agx_base_class((agx_u8)src[j + 1]), c2 = agx_base_class((agx_u8)src[j + 2]), c3 = agx_base_class((agx_u8)src[j + 3]);
        dst[j >> 2] = (agx_u8)((c0 & 3u) | ((c1 & 3u) << 2) | ((c2 & 3u) << 4) | ((c3 & 3u) << 6));
        if ((c0 | c1 | c2 | c3) & 4u) { if (c0 == 4u) other.push_back(row_base + j); if (c1 == 4u) other.push_back(row_base + j + 1); if (c2 == 4u) other.push_back(row_base + j + 2); if (c3 == 4u) other.push_back(row_base + j + 3); }
    }
    if (j < len) {
        agx_u32 b = 0;
        for (size_t i = j; i < len; i++) { const agx_u32 c = agx_base_class((agx_u8)src[i]); b |= (c & 3u) << (2 * (i - j)); if (c == 4u) other.push_back(row_base + i); }
        dst[j >> 2] = (agx_u8)b; j += 4;
    }
    for (size_t b = j >> 2; b < quarter; b++) dst[b] = 0;                 // (the general path pads the row with 'N': class 4, packed as 0)
}
#if defined(__x86_64__)
// 32 bases per step: agx_base_class() on byte lanes ((c >> 1) & 3 picks the expected letter of "ACTG" by table look-up; equal = that class, else "other"),
// then four 2-bit classes per byte through two multiply-adds
__attribute__((target("avx2"))) inline void pack_row_avx2(const char *src, size_t len, agx_u8 *dst, size_t quarter, unsigned long long row_base, std::vector<unsigned long long> &other) {
    const __m256i letters = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i three = _mm256_set1_epi8(3), one = _mm256_set1_epi8(1), m14 = _mm256_set1_epi16(0x0401), m116 = _mm256_set1_epi32(0x00100001);
    const __m256i gather = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    size_t j = 0;
    for (; j + 32 <= len; j += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(src + j));
        const __m256i idx = _mm256_and_si256(_mm256_srli_epi16(v, 1), three);
        const __m256i ok = _mm256_cmpeq_epi8(_mm256_shuffle_epi8(letters, idx), v);
        const __m256i cls = _mm256_and_si256(_mm256_xor_si256(idx, _mm256_and_si256(_mm256_srli_epi16(idx, 1), one)), ok);
        const __m256i t32 = _mm256_madd_epi16(_mm256_maddubs_epi16(cls, m14), m116);
        const __m256i sh = _mm256_shuffle_epi8(t32, gather);
        const uint32_t lo = (uint32_t)_mm256_extract_epi32(sh, 0), hi = (uint32_t)_mm256_extract_epi32(sh, 4);
        memcpy(dst + (j >> 2), &lo, 4); memcpy(dst + (j >> 2) + 4, &hi, 4);
        for (uint32_t bad = ~(uint32_t)_mm256_movemask_epi8(ok); bad; bad &= bad - 1) other.push_back(row_base + j + (unsigned)__builtin_ctz(bad));
    }
    pack_row_scalar(src, j, len, dst, quarter, row_base, other);
}
#endif
inline void pack_row(const char *src, size_t len, agx_u8 *dst, size_t quarter, unsigned long long row_base, std::vector<unsigned long long> &other) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { pack_row_avx2(src, len, dst, quarter, row_base, other); return; }
#endif
    pack_row_scalar(src, 0, len, dst, quarter, row_base, other);
}

// One SAM line, the common shape — QNAME, FLAG and POS plain numbers, RNAME without a '.', CIGAR of M, I, D, S — in one forward scan: what parse_sam_line()
// (agx_parse.h; parseBOWTIE, AG:181-285) computes.  false: anything else; the caller then hands the line to parse_sam_line(), which knows all of it.
inline bool parse_sam_line_fast(const char *s, const char *e, Mate &m, std::vector<agx_run> &runs) {
    const char *c = s, *b = s;
    agx_u32 id = 0; while (c < e && (unsigned)(*c - '0') <= 9u) id = id * 10u + (unsigned)(*c++ - '0');
    if (c == b || c - b > 9 || c >= e || *c != '\t') return false;
    b = ++c;
    agx_u32 flag = 0; while (c < e && (unsigned)(*c - '0') <= 9u) flag = flag * 10u + (unsigned)(*c++ - '0');
    if (c == b || c - b > 9 || c >= e || *c != '\t') return false;
    b = ++c;
    m.id = id; m.fr = (flag & 0x10u) ? 1u : 0u;
    m.run0 = runs.size(); m.nruns = 0; m.total = m.ins = m.del = m.clipL = m.clipR = 0; m.pos0 = 0;
    while (c < e && *c != '\t') { if (*c == '.') return false; c++; }
    m.aligned = !(c > b && *b == '*');
    if (!m.aligned) return true;
    if (c >= e) return false;
    b = ++c;
    agx_u32 pos1 = 0; while (c < e && (unsigned)(*c - '0') <= 9u) pos1 = pos1 * 10u + (unsigned)(*c++ - '0');
    if (c == b || c - b > 9 || c >= e || *c != '\t') return false;
    ++c;
    while (c < e && *c != '\t') c++;                      // MAPQ
    if (c >= e) return false;
    ++c;
    int ins = 0, del = 0, total = 0, start = 0, end = 0, first = 1, num = 0, digits = 0;
    for (; c < e && *c != '\t'; c++) {
        const char ch = *c;
        if ((unsigned)(ch - '0') <= 9u) { num = num * 10 + (ch - '0'); if (++digits > 8) return false; continue; }
        if (ch == 'M') { if (num > 0) { runs.push_back(agx_run{(agx_u32)total, (agx_u32)((int)pos1 + total + del - start - ins - 1), (agx_u32)num}); m.nruns++; } total += num; first = 0; }
        else if (ch == 'I') { ins += num; total += num; }
        else if (ch == 'D') del += num;
        else if (ch == 'S' && first) { start = num; total += num; first = 0; }
        else if (ch == 'S') { end = num; total += num; }
        else return false;
        num = 0; digits = 0;
    }
    m.total = (agx_u32)total; m.ins = (agx_u32)ins; m.del = (agx_u32)del; m.clipL = (agx_u32)start; m.clipR = (agx_u32)end; m.pos0 = pos1 - 1u;
    return true;
}
// the identity filter (passes(), AG:1261): a mate without indels or clips has both ratios at exactly 1
inline bool passes_fast(const Mate &m) { return ((m.ins | m.del | m.clipL | m.clipR) == 0 && m.total != 0) || passes(m); }

}  // namespace

// The general path's staging of the read alignments (what agx_engine.cpp's stage_inputs did inline): every hit with the ROW of its left mate's bases
// and which mate that is, packed into the wire formats; one row of 2-bit classes per (pair, left mate), first come.
void stage_pairs(const Pairs &P, agx_u32 k, unsigned threads, StageSink &sink, StagedPairs &S, std::vector<agx_hit> *staged_out) {
    S = StagedPairs();
    const size_t nh = P.hits.size(), n_runs = P.runs.size();
    S.nh = nh; S.n_runs = n_runs; S.n_pairs_in_file = P.n_pairs_in_file; S.n_sam_pairs = P.n_sam_pairs;
    for (const agx_hit &h : P.hits) { S.maxlen = std::max<agx_u32>(S.maxlen, h.len); if (h.len > P.stride) throw Error{E_ARG, "read longer than the read stride"}; }
    S.stride = (S.maxlen + 3u) & ~3u;
    S.hits = (agx_whit *)sink.take(SA_HITS, (nh + 1) * sizeof(agx_whit)); S.runs = (agx_wrun *)sink.take(SA_RUNS, (n_runs + 1) * sizeof(agx_wrun));
    const agx_run *runs = P.runs.data();
    for (size_t i = 0; i < n_runs; i++) { if (runs[i].q > 0xFFFFu || runs[i].n > 0xFFFFu) throw Error{E_UNSUPPORTED, "alignment run beyond read index 65535"}; S.runs[i] = agx_wrun{runs[i].t, (agx_u16)runs[i].q, (agx_u16)runs[i].n}; }
    std::vector<agx_hit> st(nh);
    Team team(std::max(1u, threads));
    const unsigned T = team.size();
    team.run([&](unsigned t) {
        for (size_t i = nh * t / T, hi = nh * (t + 1) / T; i < hi; i++) {
            agx_hit h = P.hits[i];
            h.pad[0] = agx_hit_left_is_mate2(h, runs, k) ? 1 : 0; h.pad[1] = h.pad[2] = 0;
            st[i] = h;
        }
    });
    std::vector<agx_u16> row_len;                       // read length of every row: bases beyond it are never looked at
    size_t n_sides = 0;
    {
        std::vector<agx_u32> row_of((size_t)P.n_slots + 1, AGX_NONE);
        S.row_slot.reserve(P.n_slots / 2 + 16); row_len.reserve(P.n_slots / 2 + 16);
        for (size_t i = 0; i < nh; i++) {
            agx_hit &h = st[i];
            const agx_u32 sa = h.slot1 + (h.pad[0] & 1u);
            if (sa >= P.n_slots) throw Error{E_ARG, "hit names a read slot outside the unit"};
            if (row_of[sa] == AGX_NONE) { row_of[sa] = (agx_u32)S.row_slot.size(); S.row_slot.push_back(sa); row_len.push_back(h.len); }
            else if (row_len[row_of[sa]] < h.len) row_len[row_of[sa]] = h.len;
            h.slot1 = row_of[sa];
            if (h.nruns1 | h.nruns2) n_sides++;
        }
    }
    S.n_sides = n_sides; S.sides = (agx_wside *)sink.take(SA_SIDES, (n_sides + 1) * sizeof(agx_wside));
    { size_t at = 0; for (size_t i = 0; i < nh; i++) { const agx_hit &h = st[i]; const bool sd = (h.nruns1 | h.nruns2) != 0; S.hits[i] = agx_pack_hit(h, (agx_u32)at); if (sd) S.sides[at++] = agx_wside{h.runs1, h.runs2, (agx_u32)h.nruns1 | ((agx_u32)h.nruns2 << 16)}; } }
    { size_t nj = 0; for (const agx_hit &h : st) nj += ((h.pad[0] & 1u) ? h.nruns2 : h.nruns1) >= 2; S.n_jump = nj; S.jump = (agx_u32 *)sink.take(SA_JUMP, (nj + 1) * 4);
      size_t at = 0; for (size_t i = 0; i < nh; i++) if (((st[i].pad[0] & 1u) ? st[i].nruns2 : st[i].nruns1) >= 2) S.jump[at++] = (agx_u32)i; }
    const size_t quarter = S.stride / 4, n_rows = S.row_slot.size();
    S.n_rows = (agx_u32)n_rows; S.n_codes = n_rows * quarter;
    S.codes = (agx_u8 *)sink.take(SA_CODES, S.n_codes + 16);
    const char *bases = P.bases.data();
    std::vector<std::vector<unsigned long long>> other(T);
    team.run([&](unsigned t) {
        for (size_t r = n_rows * t / T, hi = n_rows * (t + 1) / T; r < hi; r++)
            pack_row(bases + (size_t)S.row_slot[r] * P.stride, row_len[r], S.codes + r * quarter, quarter, (unsigned long long)r * S.stride, other[t]);
    });
    for (const auto &o : other) S.n_other += o.size();
    S.other = (unsigned long long *)sink.take(SA_OTHER, (S.n_other + 1) * 8);
    { size_t at = 0; for (const auto &o : other) { if (!o.empty()) memcpy(S.other + at, o.data(), o.size() * 8); at += o.size(); } }      // (threads take ascending row ranges: the list is sorted)
    if (staged_out) staged_out->swap(st);
}

// ---- tmp/_agx_pairs.<u>.bin (agx_host.h) -------------------------------------------------------------------------------------------------------
std::vector<agx_u8> other_bytes_of(const Pairs &P, const StagedPairs &S) {
    std::vector<agx_u8> out(S.n_other);
    for (size_t i = 0; i < S.n_other; i++) {
        const unsigned long long x = S.other[i]; const size_t r = (size_t)(x / S.stride), j = (size_t)(x % S.stride);
        if (r >= S.row_slot.size()) throw Error{E_ARG, "listed base beyond the rows"};
        out[i] = (agx_u8)P.bases[(size_t)S.row_slot[r] * P.stride + j];
    }
    return out;
}
void write_pairs_file(const std::string &path, const StagedPairs &S, const agx_u8 *other_bytes, agx_u32 k, agx_u32 batch) {
    using namespace pairsfile;
    Header H; memset(&H, 0, sizeof H); memcpy(H.magic, MAGIC, 8); H.version = 1; H.k = k; H.batch = batch ? batch : 1000000u; H.stride = S.stride; H.maxlen = S.maxlen; H.n_rows = S.n_rows;
    H.nh = S.nh; H.n_sides = S.n_sides; H.n_runs = S.n_runs; H.n_codes = S.n_codes; H.n_other = S.n_other; H.n_jump = S.n_jump; H.pairs_in_file = S.n_pairs_in_file; H.sam_pairs = S.n_sam_pairs;
    H.sizes[0] = sizeof(agx_whit); H.sizes[1] = sizeof(agx_wside); H.sizes[2] = sizeof(agx_wrun);
    const void *ptr[S_N] = {S.hits, S.sides, S.runs, S.codes, S.other, other_bytes, S.jump};
    const unsigned long long len[S_N] = {S.nh * sizeof(agx_whit), S.n_sides * sizeof(agx_wside), S.n_runs * sizeof(agx_wrun), S.n_codes, S.n_other * 8ull, S.n_other, S.n_jump * 4ull};
    unsigned long long at = (sizeof(Header) + 4095) & ~4095ull;
    for (int i = 0; i < S_N; i++) { H.off[i] = at; H.len[i] = len[i]; at = (at + len[i] + 4095) & ~4095ull; }
    const std::string part = path + ".part";
    FILE *f = fopen(part.c_str(), "wb");
    if (!f) throw Error{E_IO, "CANNOT OPEN FILE! (" + part + ")"};
    bool ok = fwrite(&H, sizeof H, 1, f) == 1;
    for (int i = 0; i < S_N && ok; i++) { ok = fseeko(f, (off_t)H.off[i], SEEK_SET) == 0 && (len[i] == 0 || fwrite(ptr[i], 1, len[i], f) == len[i]); }
    if (ok) ok = ftruncate(fileno(f), (off_t)at) == 0;
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(part.c_str(), path.c_str()) != 0) { (void)remove(part.c_str()); throw Error{E_IO, "cannot write " + path}; }
}
bool open_pairs_file(const std::string &path, PairsFile &F) {
    using namespace pairsfile;
    struct stat sb; if (stat(path.c_str(), &sb) != 0) return false;
    F.fv.reset(new FileView(path));
    if (F.fv->n < sizeof(Header)) throw Error{E_FORMAT, path + " is not a staged-pairs file"};
    memcpy(&F.H, F.fv->p, sizeof(Header));
    const Header &H = F.H;
    bool fine = memcmp(H.magic, MAGIC, 8) == 0 && H.version == 1 && H.sizes[0] == sizeof(agx_whit) && H.sizes[1] == sizeof(agx_wside) && H.sizes[2] == sizeof(agx_wrun);
    for (int i = 0; i < S_N && fine; i++) fine = H.off[i] <= F.fv->n && H.len[i] <= F.fv->n - H.off[i];
    fine = fine && H.len[S_HITS] == H.nh * sizeof(agx_whit) && H.len[S_SIDES] == H.n_sides * sizeof(agx_wside) && H.len[S_RUNS] == H.n_runs * sizeof(agx_wrun) && H.len[S_CODES] == H.n_codes &&
           H.len[S_OTHER] == H.n_other * 8 && H.len[S_OTHERB] == H.n_other && H.len[S_JUMP] == H.n_jump * 4 && (H.stride & 3u) == 0 && H.maxlen <= H.stride &&
           H.n_codes == (unsigned long long)H.n_rows * (H.stride / 4) && H.n_sides <= H.nh && H.n_jump <= H.nh && H.nh < 0xFFFFFFFFull && H.n_runs < 0xFFFFFFFFull && H.n_rows < 0x7FFFFFFFu;
    if (!fine) throw Error{E_FORMAT, path + " is not a staged-pairs file of this layout"};
    return true;
}

void build_cm_layout(const agx_u8 *cm_cnt, size_t n_pos, const agx_cmseg *segs, size_t n_segs, CmLayout &L) {
    L.cnt_runs.clear(); L.cnt_chunks.clear(); L.seg_chunks.clear();
    unsigned long long base = 0;
    for (size_t x = 0; x < n_pos;) {
        const size_t m = run_same(cm_cnt + x, n_pos - x);
        const agx_u32 run = (agx_u32)L.cnt_runs.size();
        L.cnt_runs.push_back(agx_cntrun{(agx_u32)x, (agx_u32)m, cm_cnt[x], (agx_u32)base});
        for (size_t off = 0; off < m; off += AGX_CM_CHUNK) L.cnt_chunks.push_back(agx_chunk{run, (agx_u32)off});
        base += (unsigned long long)m * cm_cnt[x]; x += m;
    }
    if (base >= 0xFFFFFFFFull) throw Error{E_ARG, "conti-mer table exceeds 2^32 entries"};
    for (size_t g = 0; g < n_segs; g++) for (agx_u32 off = 0; off < segs[g].len; off += AGX_CM_CHUNK) L.seg_chunks.push_back(agx_chunk{(agx_u32)g, off});
}

// index of the rank-0 runs (the first n_seg0, sorted by position): entry i = the last run that starts at or before position i * AGX_SEG_INDEX, 0 if none does
void build_seg_index(const agx_cmseg *segs, size_t n_seg0, size_t n_pos, std::vector<agx_u32> &index) {
    index.assign(n_pos / AGX_SEG_INDEX + 2, 0u);
    size_t si = 0;
    for (size_t i = 0; i < index.size(); i++) {
        const unsigned long long x = (unsigned long long)i * AGX_SEG_INDEX;
        while (si + 1 < n_seg0 && segs[si + 1].pos0 <= x) si++;
        index[i] = (agx_u32)si;
    }
}

// Reference bases for the upload: 2 bits each where they are A, C, G, T, and the rest as stretches of one byte value (N runs; a soft-masked sequence
// has too many of them: false, and the bases cross as they are).  The classes are the read bases' (agx_base_class: A, C, G, T = 0..3).
bool pack_reference(const char *ref, size_t n, unsigned threads, agx_u8 *packed, std::vector<agx_refx> &others) {
    others.clear();
    const size_t chunk = (size_t)1 << 20, n_chunks = (n + chunk - 1) / chunk;
    std::vector<std::vector<unsigned long long>> other(n_chunks);
    Team team(std::max<unsigned>(1u, (unsigned)std::min<size_t>(threads, n_chunks ? n_chunks : 1)));
    std::atomic<size_t> next{0};
    team.run([&](unsigned) {
        for (size_t c; (c = next.fetch_add(1)) < n_chunks;) {
            const size_t lo = c * chunk, len = std::min(chunk, n - lo);          // (chunks start at multiples of 4: whole bytes)
            pack_row(ref + lo, len, packed + lo / 4, (len + 3) / 4, lo, other[c]);
        }
    });
    size_t total = 0; for (const auto &o : other) total += o.size();
    const size_t limit = n / 256 + 1024;                // stretches, not bases: an N run of a million is one
    agx_refx cur{0, 0, 0};
    for (const auto &o : other) for (unsigned long long x : o) {
        const agx_u32 byte = (agx_u8)ref[x];
        if (cur.len && (unsigned long long)cur.pos + cur.len == x && cur.byte == byte) { cur.len++; continue; }
        if (cur.len) { others.push_back(cur); if (others.size() > limit) return false; }
        cur = agx_refx{(agx_u32)x, 1u, byte};
    }
    if (cur.len) others.push_back(cur);
    (void)total;
    return others.size() <= limit;
}

// The read rows in their upload form (agx_core.h "read rows relative to the reference"): every row against what the reference predicts for it
// under its anchor hit.  Rows in ranges of whole 64-row blocks, one range after another handed to the threads; a range's units are gathered in a buffer of its own and
// put together in row order at the end.  false: the form does not apply (rows too long for a 14-bit index, more than 2^32 units).
// The prediction is agx_row_expected16's (the function the device calls), made 32 bases at a time here: the same reference codes, the same complement, the same zeros
// at and beyond the read's length — tests/test_row_diffs.py decodes every row with the device's function.
namespace {
inline unsigned long long ref_window32(const agx_u8 *packed, long long p) {      // codes of positions p .. p + 31, position p in the low bits; positions below 0 read as 0
    if (p <= -32) return 0ull;
    if (p < 0) { unsigned long long lo; memcpy(&lo, packed, 8); return lo << (2u * (unsigned)(-p)); }
    const size_t b = (size_t)(p >> 2); const unsigned sh = 2u * (unsigned)(p & 3);
    unsigned long long lo; memcpy(&lo, packed + b, 8);
    return sh ? (lo >> sh) | ((unsigned long long)packed[b + 8] << (64u - sh)) : lo;
}
inline unsigned long long reverse_pairs32(unsigned long long x) {
    x = __builtin_bswap64(x);
    x = ((x & 0xF0F0F0F0F0F0F0F0ull) >> 4) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return ((x & 0xCCCCCCCCCCCCCCCCull) >> 2) | ((x & 0x3333333333333333ull) << 2);
}
}
bool build_row_diffs(const agx_whit *hits, size_t nh, const agx_wside *sides, size_t n_sides, const agx_wrun *runs, size_t n_runs, const agx_u8 *codes2, size_t n_rows, agx_u32 stride,
                     const agx_u32 *wref, size_t n_pos, unsigned threads, RowDiffs &D, bool rows_are_hits) {
    D = RowDiffs();
    if (rows_are_hits && n_rows != nh) return false;
    if (stride == 0 || (stride & 3u) || stride > AGX_ROW_MAXSTRIDE || n_rows >= 0x7FFFFFFFull || nh >= 0xFFFFFFFFull) return false;
    const agx_u8 *packed = (const agx_u8 *)wref;
    const size_t row_bytes = stride / 4, n_blocks = (n_rows + 63) / 64;
    enum { W_MAX = AGX_ROW_MAXSTRIDE / 32 };
    const agx_u32 n_words = (stride + 31u) / 32u;
    const agx_u32 expl_units = agx_row_explicit_units(stride), max_diffs = std::min<agx_u32>(254u, expl_units ? expl_units - 1u : 0u);
    // anchors: the first hit that names a row
    std::vector<agx_u32> anchor(n_rows, AGX_NONE);
    D.anchor_bits.assign(nh / 32 + 2, 0u);
    // (rows_are_hits, r06: the tile-ordered forms — row i is the left-mate row of hit i, whose `row` field carries something else: every row is its own hit's)
    if (rows_are_hits) { for (size_t h = 0; h < nh; h++) { anchor[h] = (agx_u32)h; D.anchor_bits[h >> 5] |= 1u << (h & 31); } }
    else for (size_t h = 0; h < nh; h++) { const agx_u32 r = hits[h].row; if (r < n_rows && anchor[r] == AGX_NONE) { anchor[r] = (agx_u32)h; D.anchor_bits[h >> 5] |= 1u << (h & 31); } }
    // the device finds a row's anchor by counting anchor bits from its block's first anchor on (agx_anchor_select): every row needs an anchor, the anchors must come in row
    // order (the loaders number the rows as the hits first name them), and a block's anchors must lie within the eight words the device reads
    D.block_first.assign(n_blocks + 1, 0u);
    for (size_t r = 0; r < n_rows; r++) {
        if (anchor[r] == AGX_NONE || (r && anchor[r] <= anchor[r - 1])) { D = RowDiffs(); return false; }
        if ((r & 63) == 0) D.block_first[r >> 6] = anchor[r];
        else if (anchor[r] - (D.block_first[r >> 6] & ~31u) >= 256u) { D = RowDiffs(); return false; }
    }
    D.anchor_bits.resize(nh / 32 + 10, 0u);               // (eight words are read from a block's first anchor on)
    D.cnt.assign(n_blocks * 64 + 64, 0);
    const size_t per_range = 256;                         // blocks per range
    const size_t n_ranges = (n_blocks + per_range - 1) / per_range;
    std::vector<std::vector<agx_u16>> part(n_ranges);
    std::vector<size_t> part_n(n_ranges, 0), n_expl(n_ranges, 0);
    std::atomic<size_t> next{0};
    Team team(std::max<unsigned>(1u, (unsigned)std::min<size_t>(threads, n_ranges ? n_ranges : 1)));
    team.run([&](unsigned) {
        std::vector<agx_u16> buf(per_range * 64 * expl_units + 4);          // a range's units (no row takes more than it does as it is); kept exact-size per range below
        for (size_t g; (g = next.fetch_add(1)) < n_ranges;) {
            const size_t r_lo = g * per_range * 64, r_hi = std::min(n_rows, (g + 1) * per_range * 64);
            agx_u16 *const out0 = buf.data(), *out = out0;
            for (size_t r = r_lo; r < r_hi; r++) {
                const agx_u8 *row = codes2 + r * row_bytes;
                agx_u16 *const mark = out;
                bool diffed = false;
                const agx_whit w = hits[anchor[r]];
                const agx_u32 len = w.len;
                // the anchor's left mate as runs inside the read and the unit, or the row stays as it is
                agx_wrun one{agx_whit_left_t0(w), (agx_u16)0, w.len};
                const agx_wrun *rr = &one; agx_u32 nr = 1;
                bool fits = len != 0 && len <= stride;
                if (fits && !agx_whit_left_simple(w)) {
                    const size_t si = agx_whit_side(w);
                    fits = si < n_sides;
                    if (fits) { const agx_wside sd = sides[si]; const size_t first = agx_wside_left_first(w, sd); nr = agx_wside_left_count(w, sd); fits = nr >= 1 && nr <= AGX_ROW_MAXRUNS && first + nr <= n_runs; rr = runs + first; }
                }
                for (agx_u32 i = 0; i < nr && fits; i++) fits = (agx_u32)rr[i].q + rr[i].n <= len && (unsigned long long)rr[i].t + rr[i].n <= n_pos;
                if (fits) {
                    // the prediction for the whole row (agx_row_expected16's, 32 bases at a time, run by run), then the row against it
                    unsigned long long E[W_MAX + 1] = {0}, A[W_MAX] = {0};
                    const bool rev = agx_whit_left_rev(w);
                    for (agx_u32 i = 0; i < nr; i++) {
                        const agx_u32 n = rr[i].n, lo = rev ? len - rr[i].q - n : rr[i].q;
                        const long long t = rr[i].t;
                        for (agx_u32 k = 0; k < n; k += 32) {
                            unsigned long long e = rev ? ~reverse_pairs32(ref_window32(packed, t + (long long)n - 32ll - (long long)k)) : ref_window32(packed, t + (long long)k);
                            if (n - k < 32u) e &= (1ull << (2u * (n - k))) - 1ull;
                            const agx_u32 bit = 2u * (lo + k), wd = bit >> 6, sh = bit & 63u;
                            E[wd] |= e << sh;
                            if (sh) E[wd + 1] |= e >> (64u - sh);
                        }
                    }
                    memcpy(A, row, row_bytes);
                    diffed = true;
                    for (agx_u32 wd = 0; wd < n_words && diffed; wd++)
                        for (unsigned long long x = E[wd] ^ A[wd]; x;) {
                            const agx_u32 pair = (agx_u32)__builtin_ctzll(x) >> 1;
                            if ((agx_u32)(out - mark) >= max_diffs) { diffed = false; break; }
                            *out++ = (agx_u16)(((wd * 32u + pair) << 2) | (agx_u32)((A[wd] >> (2u * pair)) & 3ull));
                            x &= ~(3ull << (2u * pair));
                        }
                }
                if (diffed) { D.cnt[r] = (agx_u8)(out - mark); continue; }
                out = mark;
                out[expl_units - 1] = 0;                                    // (an odd number of bytes: the last unit's high byte)
                memcpy(out, row, row_bytes);
                out += expl_units;
                D.cnt[r] = (agx_u8)AGX_ROW_EXPLICIT; n_expl[g]++;
            }
            part_n[g] = (size_t)(out - out0);
            part[g].assign(out0, out);
        }
    });
    unsigned long long total = 0;
    for (size_t n : part_n) total += n;
    if (total >= 0xFFFFFFF0ull) { D = RowDiffs(); return false; }
    D.units.resize((size_t)total + 2);
    D.block_off.assign(n_blocks + 2, 0u);
    {   std::vector<size_t> at(n_ranges + 1, 0);
        for (size_t g = 0; g < n_ranges; g++) { at[g + 1] = at[g] + part_n[g]; D.n_explicit += n_expl[g]; }
        next = 0;
        team.run([&](unsigned) {
            for (size_t g; (g = next.fetch_add(1)) < n_ranges;) {
                if (part_n[g]) memcpy(D.units.data() + at[g], part[g].data(), part_n[g] * 2);
                std::vector<agx_u16>().swap(part[g]);
                agx_u32 run = (agx_u32)at[g];
                for (size_t b = g * per_range, hi = std::min(n_blocks, (g + 1) * per_range); b < hi; b++) { D.block_off[b] = run; for (size_t r = b * 64; r < b * 64 + 64; r++) run += agx_row_units(D.cnt[r], stride) * (r < n_rows ? 1u : 0u); }
            }
        });
        D.block_off[n_blocks] = (agx_u32)total;
    }
    D.n_units = (size_t)total;
    return true;
}

// loadReadAlignment's parsing half (loadSeq AG:361-404, loadReadAli AG:1233-1277 with parseBOWTIE AG:181-285 and updateContig AG:763-815) and the
// staging, in one go.  See the head of the file.
bool load_pairs_fast(const ReadsIndex &reads, const std::string &sam_path, long batch, agx_u32 k, unsigned threads, StageSink &sink, StagedPairs &S) {
    S = StagedPairs();
    double tt[8]; tt[0] = now_ms();
    FileView sf(sam_path);
    if (batch <= 0) batch = 1000000;
    const long long N = (long long)(reads.headers / 2), B = batch;
    if (N == 0 || sf.n == 0 || k >= 32768) return false;
    S.n_pairs_in_file = (unsigned long long)N;
    struct alignas(128) Range {      // (one per thread: no two on a cache line)
        const char *lo = nullptr, *hi = nullptr; size_t lines = 0; bool odd = false, bad = false;
        SBuf<agx_u32> ids;                                // read id of every line pair that starts here
        SBuf<agx_hit> cand; SBuf<agx_u32> cand_pair; SBuf<agx_run> runs;      // pairs that pass the identity filter: slot1 = read id, run indices local; which pair each is
        size_t pair_base = 0;                             // global index of the first pair
        // after the batch rule (phase C): what stays, in final form but local numbering
        size_t n_keep = 0, n_runs = 0, n_sides = 0, n_jump = 0, lead_n = 0, side_base = 0, jump_base = 0; agx_u32 maxlen = 0;
        agx_u32 lead_rows = 0, n_rows_local = 0;          // rows opened by the leading group (the hits of the range's first read id), and by the whole range, as if nothing came before
        agx_u32 last_id = 0, last_count = 0; agx_u32 last_row[2] = {AGX_NONE, AGX_NONE};      // the range's last group: kept hits, rows of its mates (local numbering)
        bool single_group = false, lead_fixed = false;
        size_t hit_base = 0, run_base = 0; agx_u32 row_base = 0, row_shift = 0;               // row_shift: rows the leading group did not have to open after all (C3)
        std::vector<unsigned long long> other;
    };
    Team team(std::max(1u, threads));
    const unsigned n_thr = team.size();
    // More ranges than threads, handed out as threads become free: a thread that shares its core (or lost it for a while) takes fewer of them
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)n_thr * (n_thr > 1 ? 6 : 1), sf.n / (256u << 10) + 1));
    std::vector<Range> R(T);
    std::vector<double> busy(3 * (size_t)n_thr * 16, 0.0);      // per thread and phase (padded)
    std::atomic<unsigned> next_range{0};
    auto each_range = [&](int phase, const std::function<void(Range &)> &fn) {
        next_range.store(0);
        team.run([&](unsigned t) { const double a = now_ms(); for (unsigned i; (i = next_range.fetch_add(1)) < T;) fn(R[i]); busy[((size_t)phase * n_thr + t) * 16] = now_ms() - a; });
    };
    const char *fb = sf.p, *fe = sf.p + sf.n;
    auto line_start_at_or_after = [&](const char *c) -> const char * {
        if (c <= fb) return fb;
        const char *nl = (const char *)memchr(c - 1, '\n', (size_t)(fe - (c - 1)));
        return nl ? nl + 1 : fe;
    };
    for (unsigned t = 0; t < T; t++) R[t].lo = line_start_at_or_after(fb + sf.n / T * t);
    for (unsigned t = 0; t < T; t++) R[t].hi = t + 1 < T ? R[t + 1].lo : fe;
    // ---- phase A: lines per range; '@' lines or an empty line anywhere: not the common case ----
    each_range(0, [&](Range &r) {
        size_t n = 0; bool at = false;
#if defined(__linux__)
        {   // map the range's pages in one call instead of one fault per 64 KB (MADV_POPULATE_READ, Linux 5.14; an older kernel says EINVAL and the scan faults them in)
            const uintptr_t a = (uintptr_t)r.lo & ~(uintptr_t)4095, z = ((uintptr_t)r.hi + 4095) & ~(uintptr_t)4095;
            if (z > a) (void)madvise((void *)a, z - a, 22 /* MADV_POPULATE_READ */);
        }
#endif
        const char *stop = scan_lines(r.lo, r.hi, [&](const char *c) { n++; at |= *c == '@'; });
        r.lines = n; r.bad = at || stop != r.hi;
    });
    { size_t before = 0; for (Range &r : R) { if (r.bad) return false; r.odd = (before & 1) != 0; r.pair_base = (before + 1) / 2; before += r.lines; } if (before & 1) return false; }      // (odd line count: "BROKEN BOWTIE FILE")
    tt[1] = now_ms();
    // ---- phase B: parse the line pairs that START in each range; the identity filter (AG:1261); hits in their packed form ----
    each_range(0, [&](Range &r) {
        auto next = [&](const char *&c, const char *&ls, size_t &ln) -> bool {
            if (c >= fe) return false;
            const char *nl = (const char *)memchr(c, '\n', (size_t)(fe - c));
            ls = c; ln = (size_t)((nl ? nl : fe) - c); c = nl ? nl + 1 : fe;
            return true;
        };
        const char *c = r.lo, *ls = r.lo; size_t ln = 0;
        if (r.odd && !next(c, ls, ln)) return;
        // (the arrays are filled through locals: the Range structs of neighbouring threads are neighbours in memory)
        SBuf<agx_u32> ids, cand_pair; SBuf<agx_hit> cand; SBuf<agx_run> runs;
        ids.reserve(r.lines / 2 + 2); cand.reserve(r.lines / 2 + 2); cand_pair.reserve(r.lines / 2 + 2); runs.reserve(r.lines + 1024);
        std::vector<agx_run> tmp; tmp.reserve(256); Mate m1, m2;
        bool bad = false; agx_u32 last_id = 0; bool any = false;
        try {
            while (c < r.hi) {
                tmp.clear();
                next(c, ls, ln); if (!parse_sam_line_fast(ls, ls + ln, m1, tmp)) { tmp.clear(); parse_sam_line(ls, ln, m1, tmp); }
                if (!next(c, ls, ln)) { bad = true; break; }
                { const size_t keep = tmp.size(); if (!parse_sam_line_fast(ls, ls + ln, m2, tmp)) { tmp.resize(keep); parse_sam_line(ls, ln, m2, tmp); } }
                if (any && m1.id < last_id) { bad = true; break; }             // ids must not decrease (bowtie2 --reorder)
                last_id = m1.id; any = true;
                ids.push_back(m1.id);
                if (!(m1.aligned && m2.aligned && passes_fast(m1) && passes_fast(m2))) continue;
                if (m1.id != m2.id || m1.total != m2.total || m1.total > 65000 || m1.total == 0) { bad = true; break; }
                agx_hit h; memset(&h, 0, sizeof h);
                h.slot1 = m1.id; h.len = (agx_u16)m1.total; h.rev1 = (agx_u8)m1.fr; h.rev2 = (agx_u8)m2.fr;
                bool ok = true;
                auto fill = [&](const Mate &m, agx_u32 &pos, agx_u32 &r0, agx_u16 &nr) {
                    if (m.nruns == 1 && tmp[m.run0].q == 0 && tmp[m.run0].n == m.total) { pos = tmp[m.run0].t; r0 = 0; nr = 0; return; }
                    if (m.nruns > 60000) { ok = false; return; }
                    pos = 0; r0 = (agx_u32)runs.size(); nr = (agx_u16)m.nruns;
                    runs.append(tmp.data() + m.run0, m.nruns);
                    for (agx_u32 i = 1; i < nr; i++) if (runs[r0 + i].t < runs[r0 + i - 1].t + runs[r0 + i - 1].n) ok = false;
                };
                fill(m1, h.pos1, h.runs1, h.nruns1); fill(m2, h.pos2, h.runs2, h.nruns2);
                if (!ok) { bad = true; break; }
                h.pad[0] = agx_hit_left_is_mate2(h, runs.data(), k) ? 1 : 0;                      // the left mate (AG:1672-1679): decided here, where the runs are at hand
                cand.push_back(h); cand_pair.push_back((agx_u32)(ids.size() - 1));
            }
        } catch (const Error &) { bad = true; }
        r.ids = std::move(ids); r.cand = std::move(cand); r.cand_pair = std::move(cand_pair); r.runs = std::move(runs); r.bad = bad;
    });
    for (Range &r : R) if (r.bad) return false;
    tt[2] = now_ms();
    // ---- phase C1: the batch rule (AG:361-404, 1258-1259) over the ids, which do not decrease: binary searches instead of a scan ----
    size_t M = 0; { agx_u32 prev = 0; bool any = false; for (Range &r : R) { if (!r.ids.empty()) { if (any && r.ids[0] < prev) return false; prev = r.ids.back(); any = true; } if (r.pair_base != M) return false; M += r.ids.size(); } }
    // first pair at or behind `from` whose id is beyond `hi`
    auto first_beyond = [&](size_t from, long long hi) -> size_t {
        for (unsigned t = 0; t < T; t++) {
            const Range &r = R[t];
            if (r.ids.empty() || r.pair_base + r.ids.size() <= from) continue;
            if ((long long)r.ids.back() <= hi) continue;
            const size_t lo_i = from > r.pair_base ? from - r.pair_base : 0;
            const agx_u32 *it = hi < 0 ? r.ids.begin() + lo_i : std::upper_bound(r.ids.begin() + lo_i, r.ids.end(), (agx_u32)std::min<long long>(hi, 0xFFFFFFFFll));
            if (it != r.ids.end()) return r.pair_base + (size_t)(it - r.ids.begin());
        }
        return M;
    };
    std::vector<std::pair<size_t, size_t>> dead;          // [from, to) pair indices that leave nothing behind
    size_t consumed = M;
    {
        long long lo = 0, hi = std::min<long long>(B, N) - 1;
        bool final_batch = hi == N - 1 && N % B != 0;
        size_t pos = 0;
        for (;;) {
            const size_t j = first_beyond(pos, hi);
            if (lo > hi && j > pos) dead.emplace_back(pos, j);      // the empty last batch skips everything (AG:1258)
            if (j >= M) break;
            if (final_batch) { consumed = j + 1; dead.emplace_back(j, M); break; }      // AG:1259 in the last batch: the scan ends
            dead.emplace_back(j, j + 1);                                                  // AG:1259: this line pair is consumed and lost; the next batch begins
            lo = hi + 1; hi = std::min<long long>(lo + B, N) - 1;
            final_batch = lo == N || (hi == N - 1 && N % B != 0);
            pos = j + 1;
        }
    }
    S.n_sam_pairs = consumed;
    tt[3] = now_ms();
    // ---- phase C2: what stays of every range; per pair: hits so far (agx_hit::back), rows of read bases — one per (pair, left mate), opened by the
    //      first hit that needs it — numbered as if nothing came before the range ----
    each_range(1, [&](Range &r) {
        size_t w = 0;
        {   // drop the dead pairs' hits; their runs stay in the local pool and are simply never copied
            size_t d = 0;
            for (size_t i = 0; i < r.cand.size(); i++) {
                const size_t gp = r.pair_base + r.cand_pair[i];
                while (d < dead.size() && dead[d].second <= gp) d++;
                if (d < dead.size() && dead[d].first <= gp) continue;
                r.cand[w++] = r.cand[i];
            }
            r.cand.resize_down(w); r.cand_pair.resize_down(w);     // from here on: the hit's row (low 31 bits) and whether it opens it (bit 31)
        }
        r.n_keep = w;
        if (!w) return;
        agx_u32 cur = r.cand[0].slot1, count = 0, row[2] = {AGX_NONE, AGX_NONE}, rows = 0; bool lead = true; agx_u16 cur_len = r.cand[0].len;
        for (size_t i = 0; i < w; i++) {
            agx_hit &h = r.cand[i];
            if (h.slot1 != cur) {
                if (lead) { r.lead_n = i; r.lead_rows = rows; lead = false; }
                cur = h.slot1; count = 0; row[0] = row[1] = AGX_NONE; cur_len = h.len;
            }
            if (h.len != cur_len || count > 250) { r.bad = true; return; }      // "hits of one pair disagree on the read length", "more than 250 hits for one pair"
            h.back = (agx_u8)count; count++;
            const unsigned m = h.pad[0] & 1u;
            agx_u32 info = 0;
            if (row[m] == AGX_NONE) { row[m] = rows++; info = 0x80000000u; }
            r.cand_pair[i] = info | row[m];
            r.n_runs += (size_t)h.nruns1 + h.nruns2; r.n_sides += (h.nruns1 | h.nruns2) ? 1u : 0u; r.n_jump += (m ? h.nruns2 : h.nruns1) >= 2 ? 1u : 0u;
            r.maxlen = std::max<agx_u32>(r.maxlen, h.len);
        }
        if (lead) { r.lead_n = w; r.lead_rows = rows; r.single_group = true; }
        r.n_rows_local = rows;
        r.last_id = cur; r.last_count = count; r.last_row[0] = row[0]; r.last_row[1] = row[1];
    });
    for (Range &r : R) if (r.bad) return false;
    // ---- phase C3 (sequential, a few operations per range): where every range's hits, runs and rows go; a pair whose hits straddle a range boundary ----
    size_t nh = 0, n_runs = 0, n_sides = 0, n_jump = 0; agx_u32 n_rows = 0, maxlen = 0;
    {
        bool have = false; agx_u32 c_id = 0, c_count = 0, c_row[2] = {AGX_NONE, AGX_NONE}; agx_u16 c_len = 0;      // the pair the ranges so far ended with: its kept hits, the rows of its mates (final numbering)
        for (Range &r : R) {
            r.hit_base = nh; r.run_base = n_runs; r.side_base = n_sides; r.jump_base = n_jump; r.row_base = n_rows; r.row_shift = 0; r.lead_fixed = false;
            nh += r.n_keep; n_runs += r.n_runs; n_sides += r.n_sides; n_jump += r.n_jump; maxlen = std::max(maxlen, r.maxlen);
            if (!r.n_keep) continue;
            const bool joins = have && c_id == r.cand[0].slot1;
            agx_u32 lead_row[2] = {AGX_NONE, AGX_NONE};
            if (joins) {                                   // the leading group continues the previous ranges' last pair: its hits counted on, rows that exist are used
                lead_row[0] = c_row[0]; lead_row[1] = c_row[1];
                agx_u32 opened = 0;
                for (size_t i = 0; i < r.lead_n; i++) {
                    agx_hit &h = r.cand[i];
                    const unsigned cnt = c_count + (unsigned)i;
                    if (cnt > 250 || h.len != c_len) return false;
                    h.back = (agx_u8)cnt;
                    const unsigned m = h.pad[0] & 1u;
                    agx_u32 info = 0;
                    if (lead_row[m] == AGX_NONE) { lead_row[m] = r.row_base + opened++; info = 0x80000000u; }
                    r.cand_pair[i] = info | lead_row[m];
                }
                r.lead_fixed = true; r.row_shift = r.lead_rows - opened;
            } else {
                for (size_t i = 0; i < r.lead_n; i++) { const unsigned m = r.cand[i].pad[0] & 1u; if (lead_row[m] == AGX_NONE) lead_row[m] = r.row_base + (r.cand_pair[i] & 0x7FFFFFFFu); }
            }
            n_rows += r.n_rows_local - r.row_shift;
            if (r.single_group) { c_count = (joins ? c_count : 0) + (agx_u32)r.lead_n; c_row[0] = lead_row[0]; c_row[1] = lead_row[1]; }
            else { c_count = r.last_count; for (int m = 0; m < 2; m++) c_row[m] = r.last_row[m] == AGX_NONE ? AGX_NONE : r.row_base + r.last_row[m] - r.row_shift; }
            c_id = r.last_id; c_len = r.cand[r.n_keep - 1].len;
            have = true;
        }
    }
    if (nh == 0) {          // nothing kept: what the general loader leaves (no rows, stride 0)
        S.hits = (agx_whit *)sink.take(SA_HITS, sizeof(agx_whit)); S.runs = (agx_wrun *)sink.take(SA_RUNS, sizeof(agx_wrun)); S.sides = (agx_wside *)sink.take(SA_SIDES, sizeof(agx_wside));
        S.codes = (agx_u8 *)sink.take(SA_CODES, 16); S.other = (unsigned long long *)sink.take(SA_OTHER, 8); S.jump = (agx_u32 *)sink.take(SA_JUMP, 4);
        return true;
    }
    if (n_runs >= 0xFFFFFFFFull) return false;
    tt[5] = now_ms();
    const agx_u32 stride = (maxlen + 3u) & ~3u; const size_t quarter = stride / 4;
    S.nh = nh; S.n_runs = n_runs; S.n_sides = n_sides; S.stride = stride; S.maxlen = maxlen; S.n_rows = n_rows; S.n_codes = (size_t)n_rows * quarter;
    S.hits = (agx_whit *)sink.take(SA_HITS, (nh + 1) * sizeof(agx_whit)); S.runs = (agx_wrun *)sink.take(SA_RUNS, (n_runs + 1) * sizeof(agx_wrun));
    S.sides = (agx_wside *)sink.take(SA_SIDES, (n_sides + 1) * sizeof(agx_wside));
    S.codes = (agx_u8 *)sink.take(SA_CODES, S.n_codes + 16);
    S.n_jump = n_jump; S.jump = (agx_u32 *)sink.take(SA_JUMP, (n_jump + 1) * 4);
    S.row_off.assign(n_rows, 0);
    tt[4] = now_ms();
    // ---- phase D: hits and runs to their final places; the left mates' bases from the reads file, as 2-bit classes, by the hit that opens the row ----
    const char *rb = reads.fv.p, *re = reads.fv.p + reads.fv.n;
    each_range(2, [&](Range &r) {
        size_t run_at = r.run_base, side_at = r.side_base, jump_at = r.jump_base;
        auto put_runs = [&](agx_u32 from, agx_u32 n) { for (agx_u32 j = 0; j < n; j++) { const agx_run &x = r.runs[from + j]; S.runs[run_at + j] = agx_wrun{x.t, (agx_u16)x.q, (agx_u16)x.n}; } };
        for (size_t i = 0; i < r.n_keep; i++) {
            // the reads are spread over the whole (mapped) reads file: every row is two cache misses and a TLB miss unless it is asked for ahead of time
            if (i + 40 < r.n_keep && (r.cand_pair[i + 40] & 0x80000000u)) { const unsigned long long pr = 2ull * r.cand[i + 40].slot1; if (pr + 1 < reads.rec_off.size()) __builtin_prefetch(&reads.rec_off[pr]); }
            if (i + 20 < r.n_keep && (r.cand_pair[i + 20] & 0x80000000u)) { const unsigned long long pr = 2ull * r.cand[i + 20].slot1; if (pr + 1 < reads.rec_off.size()) { const char *pc = rb + reads.rec_off[pr]; __builtin_prefetch(pc); __builtin_prefetch(pc + 64); __builtin_prefetch(pc + 128); __builtin_prefetch(pc + 192); } }      // (both mates' records are looked at: four lines)
            agx_hit h = r.cand[i];
            const agx_u32 id = h.slot1, info = r.cand_pair[i];
            const bool opens = (info & 0x80000000u) != 0;
            const agx_u32 row = (r.lead_fixed && i < r.lead_n) ? (info & 0x7FFFFFFFu) : r.row_base + (info & 0x7FFFFFFFu) - r.row_shift;
            const unsigned m = h.pad[0] & 1u;
            if (h.nruns1) { put_runs(h.runs1, h.nruns1); h.runs1 = (agx_u32)run_at; run_at += h.nruns1; }
            if (h.nruns2) { put_runs(h.runs2, h.nruns2); h.runs2 = (agx_u32)run_at; run_at += h.nruns2; }
            h.slot1 = row; h.pad[1] = h.pad[2] = 0;
            if ((m ? h.nruns2 : h.nruns1) >= 2) S.jump[jump_at++] = (agx_u32)(r.hit_base + i);
            S.hits[r.hit_base + i] = agx_pack_hit(h, (agx_u32)side_at);
            if (h.nruns1 | h.nruns2) S.sides[side_at++] = agx_wside{h.runs1, h.runs2, (agx_u32)h.nruns1 | ((agx_u32)h.nruns2 << 16)};
            if (!opens) continue;
            // both mates' records are looked at (the general loader checks both), the left mate's bases are packed
            const unsigned long long r0 = 2ull * id;
            if (r0 + 1 >= reads.rec_off.size()) { r.bad = true; return; }
            for (unsigned mate = 0; mate < 2; mate++) {
                const char *c = rb + reads.rec_off[r0 + mate];
                if (c >= re || *c != '>') { r.bad = true; return; }
                const char *nl = (const char *)memchr(c, '\n', (size_t)(re - c));
                if (!nl || nl + 1 >= re) { r.bad = true; return; }
                const char *ls = nl + 1, *nl2 = (const char *)memchr(ls, '\n', (size_t)(re - ls));
                const size_t ln = (size_t)((nl2 ? nl2 : re) - ls);
                if (ln == 0 || ln != h.len) { r.bad = true; return; }
                if (mate == m) { S.row_off[row] = (uint64_t)(ls - rb); pack_row(ls, ln, S.codes + (size_t)row * quarter, quarter, (unsigned long long)row * stride, r.other); }
            }
        }
    });
    for (Range &r : R) if (r.bad) return false;
    for (const Range &r : R) S.n_other += r.other.size();
    S.other = (unsigned long long *)sink.take(SA_OTHER, (S.n_other + 1) * 8);
    {   // rows ascend with the ranges, except that a leading group may have filled a row that an earlier range opened: sort if that happened
        size_t at = 0; bool sorted = true; unsigned long long prev = 0;
        for (const Range &r : R) { for (unsigned long long v : r.other) { if (v < prev) sorted = false; prev = v; S.other[at++] = v; } }
        if (!sorted) std::sort(S.other, S.other + S.n_other);
    }
    if (g_timing) {
        auto stat = [&](int phase, char *out) { double lo = 1e30, hi = 0, sum = 0; for (unsigned t = 0; t < n_thr; t++) { const double v = busy[((size_t)phase * n_thr + t) * 16]; lo = std::min(lo, v); hi = std::max(hi, v); sum += v; } snprintf(out, 64, "%.1f/%.1f/%.1f", lo, sum / n_thr, hi); };
        char sb[64], sc[64], sd[64]; stat(0, sb); stat(1, sc); stat(2, sd);
        fprintf(stderr, "[agx load] SAM (fast) on %u threads: line count %.1f ms, parse %.1f ms (threads min/avg/max %s), batch rule %.1f ms, groups + buffers %.1f ms (threads %s; buffers %.1f), place + read bases %.1f ms (threads %s)\n",
                n_thr, tt[1] - tt[0], tt[2] - tt[1], sb, tt[3] - tt[2], tt[4] - tt[3], sc, tt[4] - tt[5], now_ms() - tt[4], sd);
    }
    return true;
}


// ---- the hits in tile order (r05) -------------------------------------------------------------------------------------------------------
// A stable two-level counting sort of (tile of the hit's first arrival, hit number) on the loader's threads: the upper bits of the tile first (a few thousand buckets, counted
// per thread), then every bucket by the lower ten bits.  16 bytes per hit of temporary memory.
void order_hits(const agx_whit *wh, size_t nh, const agx_wside *sd, const agx_wrun *wr, const agx_u32 *jump, size_t n_jump, size_t n_pos, unsigned threads,
                agx_u32 *perm, agx_u32 *first, agx_u32 *jump_at) {
    if (n_pos == 0 || nh >= 0xFFFFFFFFull) throw Error{E_ARG, "order_hits: empty unit or too many hits"};
    const agx_u32 n_tiles = (agx_u32)((n_pos + AGX_TILE - 1) / AGX_TILE);
    Team team((unsigned)std::max<size_t>(1, std::min<size_t>(std::min(threads, 16u), nh / 65536 + 1)));
    const unsigned T = team.size();
    const agx_u32 B = (n_tiles >> 10) + 1;                                   // level-1 buckets: 1024 tiles each
    Scratch keys_m(nh * 4 + 64), pairs_m(nh * 8 + 64);
    agx_u32 *key = (agx_u32 *)keys_m.p; unsigned long long *pr = (unsigned long long *)pairs_m.p;      // pr: tile << 32 | hit number, by level-1 bucket
    std::vector<size_t> cnt((size_t)T * B, 0);
    team.run([&](unsigned t) {
        size_t *c = cnt.data() + (size_t)t * B;
        for (size_t i = nh * t / T, hi = nh * (t + 1) / T; i < hi; i++) {
            agx_u32 tl = agx_whit_first_x(wh[i], sd, wr) / AGX_TILE; if (tl >= n_tiles) tl = n_tiles - 1;      // (an alignment beyond the unit: the build refuses the hit)
            key[i] = tl; c[tl >> 10]++;
        }
    });
    std::vector<size_t> bstart((size_t)B + 1, 0);
    { size_t at = 0; for (agx_u32 b = 0; b < B; b++) { bstart[b] = at; for (unsigned t = 0; t < T; t++) { const size_t c = cnt[(size_t)t * B + b]; cnt[(size_t)t * B + b] = at; at += c; } } bstart[B] = at; }
    team.run([&](unsigned t) {
        size_t *c = cnt.data() + (size_t)t * B;
        for (size_t i = nh * t / T, hi = nh * (t + 1) / T; i < hi; i++) pr[c[key[i] >> 10]++] = (unsigned long long)key[i] << 32 | (agx_u32)i;
    });
    std::atomic<agx_u32> next_b{0};
    team.run([&](unsigned) {
        agx_u32 local[1025];
        for (agx_u32 b; (b = next_b.fetch_add(1)) < B;) {
            const size_t lo = bstart[b], hi = bstart[b + 1];
            memset(local, 0, sizeof local);
            for (size_t j = lo; j < hi; j++) local[((agx_u32)(pr[j] >> 32) & 1023u) + 1]++;
            for (agx_u32 q = 0; q < 1024; q++) { const agx_u32 tl = (b << 10) + q; if (tl < n_tiles) first[tl] = (agx_u32)lo + local[q]; local[q + 1] += local[q]; }
            for (size_t j = lo; j < hi; j++) perm[lo + local[(agx_u32)(pr[j] >> 32) & 1023u]++] = (agx_u32)pr[j];      // stable: hit numbers ascend inside a tile
        }
    });
    first[n_tiles] = (agx_u32)nh; first[n_tiles + 1] = (agx_u32)nh;
    if (n_jump) {      // pass J's hits (those whose left mate has several runs) as places in the order
        Scratch mark_m(nh + 64); agx_u8 *mk = (agx_u8 *)mark_m.p; memset(mk, 0, nh);
        for (size_t j = 0; j < n_jump; j++) { if (jump[j] >= nh) throw Error{E_ARG, "pass J's list names a hit that does not exist"}; mk[jump[j]] = 1; }
        size_t at = 0;
        for (size_t i = 0; i < nh && at < n_jump; i++) if (mk[perm[i]]) jump_at[at++] = (agx_u32)i;
        if (at != n_jump) throw Error{E_ARG, "pass J's list names a hit twice"};
    }
}

}  // namespace agx
