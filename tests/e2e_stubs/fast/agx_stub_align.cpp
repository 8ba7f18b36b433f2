// agx_stub_align.cpp — TEST STUBS at size: the three stand-in aligners of tests/e2e_stubs/ (bt2_contigs.py, psl_chunks.py, psl_match.py) with
// a seed index instead of str.find, so that whole runs with tens of millions of read pairs finish in minutes (tests/tools/f2_at_size.py).
// Same answers as the Python stand-ins, byte for byte (tests/test_stub_fast.py); not part of the product.
//
//   agx_stub_align bt2    contigs.fa reads_1.fa reads_2.fa      > SAM on stdout      (bowtie2 -k 1 --no-mixed --no-discordant, exact matches only)
//   agx_stub_align chunks db.fa query.fa out.psl                                      (BLAT of output contigs against the genome: 40-base chunk seeds)
//   agx_stub_align match  db.fa query.fa out.psl                                      (BLAT of initial-contig heads against extended contigs: verbatim)
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rec { std::string name, seq; };

// FASTA as the Python stand-ins read it: name = first token of the header; `stop_at_empty`: an empty line ends the file (bt2_contigs.py, psl_chunks.py),
// else empty lines are skipped (psl_match.py)
std::vector<Rec> fasta(const char *path, bool stop_at_empty) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "stub: cannot open %s\n", path); exit(1); }
    std::vector<Rec> out; bool have = false; Rec cur;
    char *line = nullptr; size_t cap = 0; ssize_t n;
    while ((n = getline(&line, &cap, f)) >= 0) {
        if (n && line[n - 1] == '\n') n--;
        if (n == 0) { if (stop_at_empty) break; continue; }
        if (line[0] == '>') {
            if (have) out.push_back(std::move(cur));
            cur = Rec(); have = true;
            size_t a = 1; while (a < (size_t)n && (line[a] == ' ' || line[a] == '\t')) a++;
            size_t b = a; while (b < (size_t)n && line[b] != ' ' && line[b] != '\t') b++;
            cur.name.assign(line + a, b - a);
        } else cur.seq.append(line, (size_t)n);
    }
    if (have) out.push_back(std::move(cur));
    free(line); fclose(f);
    return out;
}

inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c; }
std::string revcomp(const std::string &s) { std::string r(s.size(), 0); for (size_t i = 0; i < s.size(); i++) r[i] = comp(s[s.size() - 1 - i]); return r; }

inline uint64_t mix(uint64_t h) { h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33; return h; }
inline uint64_t seed_hash(const char *p, unsigned S) {
    uint64_t h = 0x9e3779b97f4a7c15ull;
    unsigned i = 0;
    for (; i + 8 <= S; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = mix(h ^ w); }
    if (i < S) { uint64_t w = 0; memcpy(&w, p + i, S - i); h = mix(h ^ w); }
    return h;
}

// every S-byte window of every record, sorted by (hash, record, position); a directory on the hash's top bits finds a window's bucket
struct Index {
    struct E { uint64_t h; uint32_t rec, pos; };
    const std::vector<Rec> &db; unsigned S; std::vector<E> e; std::vector<uint32_t> dir; unsigned dbits = 0;
    Index(const std::vector<Rec> &d, unsigned s, unsigned threads) : db(d), S(s) {
        size_t total = 0; std::vector<size_t> at(db.size() + 1, 0);
        for (size_t r = 0; r < db.size(); r++) { at[r] = total; if (db[r].seq.size() >= S) total += db[r].seq.size() - S + 1; }
        at[db.size()] = total;
        if (total >= 0xFFFFFFFFull) { fprintf(stderr, "stub: database too large for the index\n"); exit(1); }
        e.resize(total);
        std::atomic<size_t> next{0};
        auto fill = [&]() { for (size_t r; (r = next.fetch_add(1)) < db.size();) { const std::string &q = db[r].seq; if (q.size() < S) continue; E *w = e.data() + at[r]; for (size_t p = 0; p + S <= q.size(); p++) w[p] = E{seed_hash(q.data() + p, S), (uint32_t)r, (uint32_t)p}; } };
        if (db.size() >= threads) { std::vector<std::thread> T; for (unsigned t = 0; t < threads; t++) T.emplace_back(fill); for (auto &t : T) t.join(); }
        else {      // few long records: cut each into pieces
            std::vector<std::thread> T;
            for (size_t r = 0; r < db.size(); r++) {
                const std::string &q = db[r].seq; if (q.size() < S) continue;
                const size_t n = q.size() - S + 1, step = (n + threads - 1) / threads;
                for (size_t a = 0; a < n; a += step) T.emplace_back([this, &q, r, a, step, n, &at]() { E *w = e.data() + at[r]; for (size_t p = a; p < std::min(n, a + step); p++) w[p] = E{seed_hash(q.data() + p, S), (uint32_t)r, (uint32_t)p}; });
            }
            for (auto &t : T) t.join();
        }
        // bucket by the top bits (counting sort, stable: inside a bucket the entries stay in (record, position) order), then order each bucket by hash
        dbits = 8; while (dbits < 26 && (1ull << dbits) < total / 4) dbits++;
        const size_t nb = (size_t)1 << dbits;
        dir.assign(nb + 1, 0);
        for (const E &x : e) dir[(x.h >> (64 - dbits)) + 1]++;
        for (size_t b = 0; b < nb; b++) dir[b + 1] += dir[b];
        { std::vector<E> t(total); std::vector<uint32_t> cur(dir.begin(), dir.end() - 1); for (const E &x : e) t[cur[x.h >> (64 - dbits)]++] = x; e.swap(t); }
        std::atomic<size_t> nextb{0};
        auto sortb = [&]() { for (size_t b; (b = nextb.fetch_add(4096)) < nb;) for (size_t c = b; c < std::min(nb, b + 4096); c++) std::stable_sort(e.begin() + dir[c], e.begin() + dir[c + 1], [](const E &x, const E &y) { return x.h < y.h; }); };
        { std::vector<std::thread> T; for (unsigned t = 0; t < threads; t++) T.emplace_back(sortb); for (auto &t : T) t.join(); }
    }
    // first place of the n bytes at p in every record that holds them, in record order: (record, position) pairs appended to out
    void find_all(const char *p, size_t n, std::vector<std::pair<uint32_t, uint32_t>> &out) const {
        out.clear();
        if (n < S) {        // shorter than a seed (an empty or tiny record): look the slow way
            for (size_t r = 0; r < db.size(); r++) {
                const std::string &q = db[r].seq;
                if (n == 0) { out.push_back({(uint32_t)r, 0}); continue; }
                const void *at = q.size() >= n ? memmem(q.data(), q.size(), p, n) : nullptr;
                if (at) out.push_back({(uint32_t)r, (uint32_t)((const char *)at - q.data())});
            }
            return;
        }
        const uint64_t h = seed_hash(p, S);
        const size_t b = h >> (64 - dbits);
        const E *lo = e.data() + dir[b], *hi = e.data() + dir[b + 1];
        lo = std::lower_bound(lo, hi, h, [](const E &x, uint64_t k) { return x.h < k; });
        uint32_t last = 0xFFFFFFFFu;
        for (; lo < hi && lo->h == h; lo++) {
            if (lo->rec == last) continue;                       // (record, position) order inside equal hashes: the first match of a record is its first place
            const std::string &q = db[lo->rec].seq;
            if ((size_t)lo->pos + n <= q.size() && memcmp(q.data() + lo->pos, p, n) == 0) { out.push_back({lo->rec, lo->pos}); last = lo->rec; }
        }
    }
};

unsigned n_threads() { const char *e = getenv("AGX_STUB_THREADS"); unsigned t = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency(); return t ? t : 1; }

// in order, in blocks: work(i, out) appends item i's text to out; blocks are written in order
template <class F> void ordered_blocks(size_t n, size_t block, FILE *f, unsigned threads, F work) {
    const size_t nblocks = (n + block - 1) / block;
    for (size_t b0 = 0; b0 < nblocks; b0 += threads) {
        const size_t nb = std::min<size_t>(threads, nblocks - b0);
        std::vector<std::string> bufs(nb); std::vector<std::thread> T;
        for (size_t j = 0; j < nb; j++) T.emplace_back([&, j]() { const size_t a = (b0 + j) * block, z = std::min(n, a + block); for (size_t i = a; i < z; i++) work(i, bufs[j]); });
        for (auto &t : T) t.join();
        for (size_t j = 0; j < nb; j++) fwrite(bufs[j].data(), 1, bufs[j].size(), f);
    }
}

inline void put(std::string &o, long v) { char b[24]; snprintf(b, sizeof b, "%ld", v); o += b; }

int bt2(const char *cpath, const char *p1, const char *p2) {
    const unsigned threads = n_threads();
    const std::vector<Rec> contigs = fasta(cpath, true), r1 = fasta(p1, true), r2 = fasta(p2, true);
    size_t minlen = 0xFFFFFFFFu; for (const Rec &r : r1) minlen = std::min(minlen, r.seq.size()); for (const Rec &r : r2) minlen = std::min(minlen, r.seq.size());
    const Index ix(contigs, (unsigned)std::max<size_t>(12, std::min<size_t>(32, minlen)), threads);
    std::string head = "@HD\tVN:1.0\tSO:unsorted\n";
    for (const Rec &c : contigs) { head += "@SQ\tSN:" + c.name + "\tLN:"; put(head, (long)c.seq.size()); head += "\n"; }
    fwrite(head.data(), 1, head.size(), stdout);
    const size_t n = std::min(r1.size(), r2.size());
    ordered_blocks(n, 16384, stdout, threads, [&](size_t i, std::string &o) {
        const std::string &a = r1[i].seq, &b = r2[i].seq, &q = r1[i].name;
        const std::string ar = revcomp(a), br = revcomp(b);
        std::vector<std::pair<uint32_t, uint32_t>> xf, yf, xr, yr;
        ix.find_all(a.data(), a.size(), xf); ix.find_all(br.data(), br.size(), yf);        // fwd1: mate 1 as read, mate 2 reverse-complemented
        ix.find_all(ar.data(), ar.size(), xr); ix.find_all(b.data(), b.size(), yr);
        // the first contig (in file order) that holds both strings of one orientation in FR order; forward first
        size_t i1 = 0, j1 = 0, i2 = 0, j2 = 0; bool hit = false, fwd = false; uint32_t rec = 0, p = 0, oo = 0;
        while (!hit) {
            uint32_t c = 0xFFFFFFFFu;
            if (i1 < xf.size()) c = std::min(c, xf[i1].first);
            if (j1 < yf.size()) c = std::min(c, yf[j1].first);
            if (i2 < xr.size()) c = std::min(c, xr[i2].first);
            if (j2 < yr.size()) c = std::min(c, yr[j2].first);
            if (c == 0xFFFFFFFFu) break;
            const bool hx = i1 < xf.size() && xf[i1].first == c, hy = j1 < yf.size() && yf[j1].first == c, hxr = i2 < xr.size() && xr[i2].first == c, hyr = j2 < yr.size() && yr[j2].first == c;
            if (hx && hy && xf[i1].second <= yf[j1].second) { hit = true; fwd = true; rec = c; p = xf[i1].second; oo = yf[j1].second; }
            else if (hxr && hyr && yr[j2].second <= xr[i2].second) { hit = true; fwd = false; rec = c; p = xr[i2].second; oo = yr[j2].second; }
            i1 += hx; j1 += hy; i2 += hxr; j2 += hyr;
        }
        if (!hit) { o += q; o += "\t77\t*\t0\t0\t*\t*\t0\t0\t"; o += a; o += "\t*\n"; o += q; o += "\t141\t*\t0\t0\t*\t*\t0\t0\t"; o += b; o += "\t*\n"; return; }
        const std::string &x = fwd ? a : ar, &y = fwd ? br : b;
        const std::string &cn = contigs[rec].name;
        o += q; o += '\t'; put(o, fwd ? 99 : 83); o += '\t'; o += cn; o += '\t'; put(o, (long)p + 1); o += "\t42\t"; put(o, (long)x.size()); o += "M\t=\t"; put(o, (long)oo + 1); o += "\t0\t"; o += x; o += "\t*\n";
        o += q; o += '\t'; put(o, fwd ? 147 : 163); o += '\t'; o += cn; o += '\t'; put(o, (long)oo + 1); o += "\t42\t"; put(o, (long)y.size()); o += "M\t=\t"; put(o, (long)p + 1); o += "\t0\t"; o += y; o += "\t*\n";
    });
    return 0;
}

int chunks(const char *dbp, const char *qp, const char *outp) {
    const unsigned threads = n_threads(); const long CH = 40;
    const std::vector<Rec> db = fasta(dbp, true), qs = fasta(qp, true);
    const Index ix(db, 32, threads);
    FILE *f = fopen(outp, "wb"); if (!f) { fprintf(stderr, "stub: cannot write %s\n", outp); return 1; }
    ordered_blocks(qs.size(), 64, f, threads, [&](size_t qi, std::string &o) {
        const std::string &q = qs[qi].seq; const std::string qr = revcomp(q);
        typedef std::vector<std::pair<long, long>> Hits;
        Hits best; int best_strand = -1; uint32_t best_t = 0;
        std::vector<Hits> per(db.size()); std::vector<std::pair<uint32_t, uint32_t>> found;
        for (int strand = 0; strand < 2; strand++) {
            const std::string &s = strand ? qr : q;
            for (Hits &h : per) h.clear();
            for (long i = 0; i + CH <= (long)s.size(); i += CH) { ix.find_all(s.data() + i, (size_t)CH, found); for (const auto &x : found) per[x.first].push_back({i, (long)x.second}); }
            for (size_t t = 0; t < db.size(); t++) if (!per[t].empty() && (best_strand < 0 || per[t].size() > best.size())) { best = per[t]; best_strand = strand; best_t = (uint32_t)t; }
        }
        if (best_strand < 0 || best.size() < 3) return;
        struct B { long i, p, n; };
        std::vector<B> blocks;
        for (const auto &h : best) {
            if (!blocks.empty() && h.first == blocks.back().i + blocks.back().n && h.second == blocks.back().p + blocks.back().n) blocks.back().n += CH;
            else if (blocks.empty() || (h.second >= blocks.back().p + blocks.back().n && h.first >= blocks.back().i + blocks.back().n)) blocks.push_back(B{h.first, h.second, CH});
        }
        std::vector<std::vector<B>> groups(1, std::vector<B>(1, blocks[0]));
        for (size_t b = 1; b < blocks.size(); b++) { const B &l = groups.back().back(); if (blocks[b].p - (l.p + l.n) > 2000) groups.push_back(std::vector<B>()); groups.back().push_back(blocks[b]); }
        const Rec &t = db[best_t];
        for (const std::vector<B> &g : groups) {
            long m = 0, qni = 0, qbi = 0, tni = 0, tbi = 0;
            for (const B &b : g) m += b.n;
            for (size_t j = 0; j + 1 < g.size(); j++) { const B &a = g[j], &b = g[j + 1]; if (b.i > a.i + a.n) qni++; qbi += b.i - a.i - a.n; if (b.p > a.p + a.n) tni++; tbi += b.p - a.p - a.n; }
            long qs_ = g.front().i, qe_ = g.back().i + g.back().n;
            if (best_strand) { const long a = (long)q.size() - qe_, b = (long)q.size() - qs_; qs_ = a; qe_ = b; }
            put(o, m); o += "\t0\t0\t0\t"; put(o, qni); o += '\t'; put(o, qbi); o += '\t'; put(o, tni); o += '\t'; put(o, tbi); o += '\t'; o += best_strand ? '-' : '+'; o += '\t';
            o += qs[qi].name; o += '\t'; put(o, (long)q.size()); o += '\t'; put(o, qs_); o += '\t'; put(o, qe_); o += '\t'; o += t.name; o += '\t'; put(o, (long)t.seq.size()); o += '\t';
            put(o, g.front().p); o += '\t'; put(o, g.back().p + g.back().n); o += '\t'; put(o, (long)g.size()); o += '\t';
            for (const B &b : g) { put(o, b.n); o += ','; } o += '\t';
            for (const B &b : g) { put(o, b.i); o += ','; } o += '\t';
            for (const B &b : g) { put(o, b.p); o += ','; } o += '\n';
        }
    });
    fclose(f);
    return 0;
}

int match(const char *dbp, const char *qp, const char *outp) {
    const unsigned threads = n_threads();
    const std::vector<Rec> db = fasta(dbp, false), qs = fasta(qp, false);
    const Index ix(db, 32, threads);
    FILE *f = fopen(outp, "wb"); if (!f) { fprintf(stderr, "stub: cannot write %s\n", outp); return 1; }
    ordered_blocks(qs.size(), 64, f, threads, [&](size_t qi, std::string &o) {
        const std::string &q = qs[qi].seq; if (q.empty()) return;
        const std::string qr = revcomp(q);
        std::vector<std::pair<uint32_t, uint32_t>> found;
        for (int strand = 0; strand < 2; strand++) {
            const std::string &s = strand ? qr : q;
            ix.find_all(s.data(), s.size(), found);
            for (const auto &x : found) {
                const long n = (long)q.size(), at = (long)x.second;
                put(o, n); o += "\t0\t0\t0\t0\t0\t0\t0\t"; o += strand ? '-' : '+'; o += '\t'; o += qs[qi].name; o += '\t'; put(o, n); o += "\t0\t"; put(o, n); o += '\t';
                o += db[x.first].name; o += '\t'; put(o, (long)db[x.first].seq.size()); o += '\t'; put(o, at); o += '\t'; put(o, at + n); o += "\t1\t"; put(o, n); o += ",\t0,\t"; put(o, at); o += ",\n";
            }
        }
    });
    fclose(f);
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc == 5 && !strcmp(argv[1], "bt2")) return bt2(argv[2], argv[3], argv[4]);
    if (argc == 5 && !strcmp(argv[1], "chunks")) return chunks(argv[2], argv[3], argv[4]);
    if (argc == 5 && !strcmp(argv[1], "match")) return match(argv[2], argv[3], argv[4]);
    fprintf(stderr, "usage: agx_stub_align bt2|chunks|match ...\n");
    return 2;
}
