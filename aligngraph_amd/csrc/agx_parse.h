// agx_parse.h — line-level parsers shared by the general loader (agx_host.cpp) and the fast loader (agx_load.cpp).  Internal.
#pragma once
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include "agx_host.h"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace agx {
namespace {

// getline + `if(buf[0]==0) break` of the reference: an empty line ends the input
struct LineReader {
    const char *p, *e; bool saw_empty = false;
    LineReader(const char *b, size_t n) : p(b), e(b + n) {}
    bool next(const char *&s, size_t &len) {
        if (p >= e) { saw_empty = true; return false; }    // (callers that care check the last byte themselves)
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        const char *le = nl ? nl : e;
        s = p; len = (size_t)(le - p); p = nl ? nl + 1 : e;
        if (len == 0 || s[0] == 0) { saw_empty = true; return false; }
        return true;
    }
};

// ---- line scanning ----------------------------------------------------------------------------------------------------------------------
// One pass over a byte range that starts at a line start: f(start_of_line) for every line, until the range ends or a line is empty or begins
// with a NUL byte (the reference's getline loops stop there: `if(buf[0]==0) break`).  Returns where the scan stopped: `hi`, or the start
// of the line that ends the input.  32 bytes at a time: newline and NUL masks, line starts = newline mask shifted by one.
template <class F> inline const char *scan_lines_scalar(const char *lo, const char *hi, F f) {
    const char *c = lo;
    while (c < hi) {
        if (*c == '\n' || *c == 0) return c;
        f(c);
        const char *nl = (const char *)memchr(c, '\n', (size_t)(hi - c));
        if (!nl) return hi;
        c = nl + 1;
    }
    return hi;
}
#if defined(__x86_64__)
template <class F> __attribute__((target("avx2"))) inline const char *scan_lines_avx2(const char *lo, const char *hi, F f) {
    const __m256i vnl = _mm256_set1_epi8('\n'), vz = _mm256_setzero_si256();
    uint32_t carry = 1;                                  // the byte before `lo` ends a line
    const char *c = lo;
    for (; c + 32 <= hi; c += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)c);
        const uint32_t nl = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, vnl)), nul = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, vz));
        uint32_t starts = (nl << 1) | carry;
        carry = nl >> 31;
        if (!starts) continue;
        const uint32_t stop = starts & (nl | nul);
        if (stop) { const uint32_t below = (stop & (0u - stop)) - 1u; for (uint32_t m = starts & below; m; m &= m - 1) f(c + __builtin_ctz(m)); return c + __builtin_ctz(stop); }
        for (uint32_t m = starts; m; m &= m - 1) f(c + __builtin_ctz(m));
    }
    // tail: the last (partial) chunk byte by byte
    if (c < hi) {
        if (carry) { if (*c == '\n' || *c == 0) return c; f(c); }
        for (const char *d = c; d + 1 < hi; d++) if (*d == '\n') { if (d[1] == '\n' || d[1] == 0) return d + 1; f(d + 1); }
    }
    return hi;
}
#endif
template <class F> inline const char *scan_lines(const char *lo, const char *hi, F f) {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) return scan_lines_avx2(lo, hi, f);
#endif
    return scan_lines_scalar(lo, hi, f);
}

// number of '\n' in [lo, hi)
inline size_t count_newlines_scalar(const char *lo, const char *hi) { size_t n = 0; for (const char *c = lo; c < hi; c++) n += *c == '\n'; return n; }
#if defined(__x86_64__)
__attribute__((target("avx2,popcnt"))) inline size_t count_newlines_avx2(const char *lo, const char *hi) {
    const __m256i vnl = _mm256_set1_epi8('\n'); size_t n = 0; const char *c = lo;
    for (; c + 32 <= hi; c += 32) n += (size_t)__builtin_popcount((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)c), vnl)));
    for (; c < hi; c++) n += *c == '\n';
    return n;
}
#endif
inline size_t count_newlines(const char *lo, const char *hi) {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) return count_newlines_avx2(lo, hi);
#endif
    return count_newlines_scalar(lo, hi);
}

inline int to_int(const char *s, size_t n) {      // atoi semantics on a bounded field
    size_t i = 0; while (i < n && (s[i] == ' ' || (s[i] >= 9 && s[i] <= 13))) i++;
    bool neg = false; if (i < n && (s[i] == '-' || s[i] == '+')) { neg = s[i] == '-'; i++; }
    long long v = 0; for (; i < n && s[i] >= '0' && s[i] <= '9'; i++) { v = v * 10 + (s[i] - '0'); if (v > 0x7fffffffffffLL) break; }
    return (int)(neg ? -v : v);
}

inline void rc_inplace(std::string &s) {
    std::reverse(s.begin(), s.end());
    for (auto &c : s) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
}

inline void fasta_body(std::string &out, const char *s, size_t n) {
    for (size_t i = 0; i < n; i += 60) { const size_t m = n - i < 60 ? n - i : 60; out.append(s + i, m); out.push_back('\n'); }
}

struct Psl { agx_u32 tID, tStart, tEnd, tGap, sID, sStart, sEnd, sGap, sSize, fr; };

void parse_psl_line(const char *s, size_t n, Psl &r, std::vector<agx_run> &seg) {
    const char *f[21]; size_t fl[21]; int nf = 0; const char *b = s, *e = s + n;
    for (const char *c = s; c <= e && nf < 21; c++) if (c == e || *c == '\t') { f[nf] = b; fl[nf] = (size_t)(c - b); nf++; b = c + 1; }
    for (; nf < 21; nf++) { f[nf] = e; fl[nf] = 0; }
    seg.clear();
    for (int col = 18; col <= 20; col++) {
        size_t sp = 0; const char *st = f[col];
        for (const char *c = f[col]; c < f[col] + fl[col]; c++) if (*c == ',') {
            const agx_u32 v = (agx_u32)to_int(st, (size_t)(c - st)); st = c + 1;
            if (col == 18) seg.push_back(agx_run{AGX_NONE, AGX_NONE, v});
            else if (sp < seg.size()) { if (col == 19) seg[sp].q = v; else seg[sp].t = v; }
            sp++;
        }
    }
    r.fr = fl[8] ? (f[8][0] == '+' ? 0u : 1u) : AGX_NONE;
    r.tID = (agx_u32)to_int(f[13], fl[13]); r.tStart = (agx_u32)to_int(f[15], fl[15]); r.tEnd = (agx_u32)to_int(f[16], fl[16]);
    r.tGap = (agx_u32)to_int(f[7], fl[7]); r.sStart = (agx_u32)to_int(f[11], fl[11]); r.sEnd = (agx_u32)to_int(f[12], fl[12]);
    r.sGap = (agx_u32)to_int(f[5], fl[5]); r.sSize = (agx_u32)to_int(f[10], fl[10]);
    size_t dot = 0; while (dot < fl[9] && f[9][dot] != '.') dot++;
    r.sID = (agx_u32)to_int(f[9], dot);
}

struct Mate { agx_u32 id; agx_u32 fr; bool aligned; agx_u32 pos0; agx_u32 total, ins, del, clipL, clipR; size_t run0; agx_u32 nruns; };

// parseBOWTIE, AG:181-285; M runs are appended to `runs`
void parse_sam_line(const char *s, size_t n, Mate &m, std::vector<agx_run> &runs) {
    const char *f[6]; size_t fl[6]; int nf = 0; const char *b = s, *e = s + n;
    for (const char *c = s; nf < 6; c++) {
        if (c >= e || *c == '\t') { f[nf] = b; fl[nf] = (size_t)((c < e ? c : e) - b); nf++; b = c + 1; if (c >= e) break; }
    }
    for (; nf < 6; nf++) { f[nf] = e; fl[nf] = 0; }
    m.id = (agx_u32)to_int(f[0], fl[0]);
    m.fr = (to_int(f[1], fl[1]) & 0x10) ? 1u : 0u;
    m.run0 = runs.size(); m.nruns = 0; m.total = m.ins = m.del = m.clipL = m.clipR = 0; m.pos0 = 0;
    m.aligned = !(fl[2] > 0 && f[2][0] == '*');
    if (!m.aligned) return;
    size_t dot = 0; while (dot < fl[2] && f[2][dot] != '.') dot++;
    if (dot != fl[2] && to_int(f[2], dot) != 0) throw Error{E_UNSUPPORTED, "SAM RNAME does not resolve to the unit sequence"};
    const int pos1 = to_int(f[3], fl[3]);
    int ins = 0, del = 0, total = 0, start = 0, end = 0, first = 1, num = 0;
    for (size_t i = 0; i < fl[5]; i++) {
        const char c = f[5][i];
        if (c >= '0' && c <= '9') { num = num * 10 + (c - '0'); continue; }
        if (c == 'I') { ins += num; total += num; }
        else if (c == 'D') del += num;
        else if (c == 'M') { if (num > 0) { runs.push_back(agx_run{(agx_u32)total, (agx_u32)(pos1 + total + del - start - ins - 1), (agx_u32)num}); m.nruns++; } total += num; first = 0; }
        else if (c == 'S' && first) { start = num; total += num; first = 0; }
        else if (c == 'S') { end = num; total += num; }
        else if (c != '*') throw Error{E_FORMAT, std::string("unknown character: ") + c};
        num = 0;
    }
    m.total = (agx_u32)total; m.ins = (agx_u32)ins; m.del = (agx_u32)del; m.clipL = (agx_u32)start; m.clipR = (agx_u32)end; m.pos0 = (agx_u32)(pos1 - 1);
}

// the identity filter of loadReadAli, AG:1261, in the reference's unsigned arithmetic
inline bool passes(const Mate &m) {
    const agx_u32 sEnd = m.total - m.clipR, tEnd = m.pos0 + (m.total + m.del - m.ins);
    return (double)(agx_u32)(sEnd - m.clipL - m.ins) / m.total >= 0.6 && (double)(agx_u32)(tEnd - m.pos0 - m.del) / (agx_u32)(tEnd - m.pos0) >= 0.6;
}

// The rules of loadReadAli that look across line pairs (AG:1233-1277), one parsed line pair at a time in file order: the batch boundaries of loadSeq
// (BATCH, AG:37 / 397: a line pair whose read id lies beyond the current batch is consumed and LOST, AG:1258-1259; a reads file of exactly m * BATCH
// pairs is followed by one empty batch that skips everything), the identity filter (AG:1261) and the hits of a pair (agx_hit::back).  Kept hits go to P.hits
// (slot1 = the read id for now) with the runs of their non-simple mates in P.runs; hit_id[i] = read id of kept hit i.  Used by the general loader and by
// whoever hands alignments over without writing SAM text (tools/agx_synth.cpp's staged-pairs mode).
struct PairRules {
    Pairs &P; std::vector<agx_u32> hit_id;
    long long N, B, lo, hi; bool final_batch; agx_u32 prev_id = 0; bool any = false; agx_u32 back = 0;
    PairRules(Pairs &p, unsigned long long pairs_in_file, long batch) : P(p), N((long long)pairs_in_file), B(batch <= 0 ? 1000000 : batch) {
        if (N == 0) throw Error{E_FORMAT, "reads file holds no pairs"};
        P.n_pairs_in_file = pairs_in_file;
        // batch n loads pairs [lo, hi]; `final_batch` = loadSeq reached the end of the reads file while loading it (AG:397).
        lo = 0; hi = std::min<long long>(B, N) - 1; final_batch = hi == N - 1 && N % B != 0;
    }
    // one parsed line pair -> at most one kept hit.  m.run0 indexes `src`.  Returns false when the final batch is over (AG:1259).
    bool consume(const Mate &m1, const Mate &m2, const agx_run *src) {
        P.n_sam_pairs++;
        const long long id = (long long)m1.id;
        if (id < lo) {
            if (lo > hi) return true;                            // the empty last batch skips everything
            throw Error{E_UNSUPPORTED, "SAM is not sorted by read id (bowtie2 --reorder output expected)"};
        }
        if (id > hi) {                                           // AG:1259: this line pair is consumed and lost; the next batch begins
            if (final_batch) return false;
            lo = hi + 1; hi = std::min<long long>(lo + B, N) - 1;
            final_batch = lo == N || (hi == N - 1 && N % B != 0);
            return true;
        }
        const bool keep = m1.aligned && m2.aligned && passes(m1) && passes(m2);
        if (!keep) return true;
        if (m1.id != m2.id) throw Error{E_UNSUPPORTED, "SAM mates of one pair are not on adjacent lines"};
        if (any && m1.id < prev_id) throw Error{E_UNSUPPORTED, "SAM is not sorted by read id (bowtie2 --reorder output expected)"};
        back = (any && m1.id == prev_id) ? back + 1 : 0;
        if (back > 250) throw Error{E_UNSUPPORTED, "more than 250 hits for one pair"};
        prev_id = m1.id; any = true;
        agx_hit h; memset(&h, 0, sizeof h);
        h.slot1 = m1.id;                                       // read id for now; turned into a slot when the bases are placed
        h.len = 0; h.rev1 = (agx_u8)m1.fr; h.rev2 = (agx_u8)m2.fr; h.back = (agx_u8)back;
        auto fill = [&](const Mate &m, agx_u32 &pos, agx_u32 &r0, agx_u16 &nr) {
            // "simple" = a single M run that covers the whole read; only the runs of non-simple mates go to the pool, contiguous per mate
            if (m.nruns == 1 && src[m.run0].q == 0 && src[m.run0].n == m.total) { pos = src[m.run0].t; r0 = 0; nr = 0; return; }
            if (m.nruns > 60000) throw Error{E_UNSUPPORTED, "CIGAR with too many runs"};
            pos = 0; r0 = (agx_u32)P.runs.size(); nr = (agx_u16)m.nruns;
            P.runs.insert(P.runs.end(), src + m.run0, src + m.run0 + m.nruns);
            for (agx_u32 i = 1; i < nr; i++)                     // runs must advance on the reference (SAM guarantees it)
                if (P.runs[r0 + i].t < P.runs[r0 + i - 1].t + P.runs[r0 + i - 1].n) throw Error{E_UNSUPPORTED, "alignment runs do not advance on the reference"};
        };
        fill(m1, h.pos1, h.runs1, h.nruns1); fill(m2, h.pos2, h.runs2, h.nruns2);
        if (m1.total != m2.total) throw Error{E_UNSUPPORTED, "mates of one pair have different CIGAR lengths"};
        if (m1.total > 65000) throw Error{E_UNSUPPORTED, "read longer than 65000"};
        h.len = (agx_u16)m1.total;
        P.hits.push_back(h);
        hit_id.push_back(m1.id);
        return true;
    }
};

}  // namespace
}  // namespace agx
