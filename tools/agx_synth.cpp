// agx_synth — seeded synthetic input generator for the AlignGraph graph-build+extend path.
//
// Writes a run directory that both the reference binary (through `--resume`) and this
// repo's engine can consume:
//
//   <out>/genome.fa  <out>/contigs.fa              the "user" inputs named on the command line
//   <out>/tmp/_command.txt  <out>/tmp/_checkpoint.txt
//   <out>/tmp/_genome.fa    <out>/tmp/_genome.<u>.fa      (format of formalizeGenome, AG:3347-3418)
//   <out>/tmp/_contigs.fa   <out>/tmp/_chaff.fa           (format of formalizeInput,  AG:3228-3345)
//   <out>/tmp/_contigs_genome.<u>.psl                      (21-column PSL as `blat -noHead`)
//   <out>/tmp/_reads.fa                                    (format of formalizeInput PE, AG:3420-3518)
//   <out>/tmp/_reads_genome.<u>.bowtie                     (headerless SAM line pairs, AG:3545-3579)
//
// No aligner is run: a "target" genome is derived from the reference by SNPs and small indels,
// contigs and reads are cut from the target, and the PSL / SAM records are derived from the
// known target->reference coordinate map.  Everything is driven by one 64-bit seed.
//
// This is test/bench infrastructure; it contains no code from the reference.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>
#include <unistd.h>

// -DAGX_SYNTH_WITH_ENGINE (build/agx_synth_bin: compiled together with the engine's host-side loader sources): the --pairs-bin mode below hands every unit's read alignments
// over STAGED (tmp/_agx_pairs.<u>.bin, aligngraph_amd/csrc/agx_host.h) instead of as SAM text + tmp/_reads.fa — 58 bytes per pair instead of 420, which is what lets
// BASELINE configs[4] (whole human: 400 M pairs of 2x150 bp) exist on a box at all.  Nothing about the alignments is decided here: every pair's two SAM lines are
// formatted in memory exactly as the text mode writes them and go through the engine's own line parser (parse_sam_line), its rules across line pairs (PairRules: batch
// boundaries, identity filter, hits per pair) and its staging (stage_pairs); `--pairs-bin 2` writes the text files of the same stream as well, and
// tests/test_staged_pairs.py compares what the loaders make of those with the staged file, byte for byte.
#ifdef AGX_SYNTH_WITH_ENGINE
#include "../aligngraph_amd/csrc/agx_host.h"
#include "../aligngraph_amd/csrc/agx_parse.h"
#endif

namespace {

struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { for (auto &v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
    int64_t range(int64_t lo, int64_t hi) { return lo + (int64_t)below((uint64_t)(hi - lo + 1)); }  // inclusive
    bool coin(double p) { return uni() < p; }
    double normal() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

const char ACGT[4] = {'A', 'C', 'G', 'T'};
inline char comp(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return c; }
}
std::string revcomp(const std::string &s) {
    std::string r(s.rbegin(), s.rend());
    for (auto &c : r) c = comp(c);
    return r;
}

struct Params {
    std::string out = "synth_run";
    uint64_t seed = 1;
    std::vector<int64_t> chroms = {50000};
    int part = 1;
    int64_t pairs = 10000;
    int L = 100;
    int k = 5, coverage = 5, insert_variation = 50;
    double snp = 0.01, indel = 0.001;
    int64_t contig_min = 2000, contig_max = 50000, gap_min = 500, gap_max = 2000;
    double contig_minus = 0.3, contig_split = 0.15, contig_dup = 0.05, contig_overlap = 0.1, contig_lowid = 0.03;
    double short_contig = 0.05;
    double chimeric = 0.0;   // probability that a contig is a misassembly: piece + 300 random bases + a piece from elsewhere (exercises misassembly removal)
    double frag_mean = 500, frag_sd = 30;
    double read_err = 0.002, read_indel = 0.10, read_clip = 0.05, read_badclip = 0.01, read_n = 0.0005;
    double multi = 0.05, multi_near = 0.3, unaligned = 0.02;
    double mate1_left = 0.5;
    double mixed_len = 0.0;  // probability that a pair is TRIMMED: both mates cut to one shorter length (60 %, 75 % or 90 % of L), as formalizeInput equalises the mates of a pair (AG:3454).  Single-stream mode only
    int sam_seq = 1;  // write SEQ/QUAL columns like bowtie2 does
    int shuffle_units = 1;
    int e2e = 0;       // also write the user-level inputs of a fresh run: reads_1.fa, reads_2.fa and stub/ (what the aligner stubs replay)
    // threads > 0: reads and SAM records are generated in chunks of 65536 pairs, each from its own generator seeded by (seed, chunk), on
    // that many threads; the files are the same for every thread count, but differ from the ones the default single-stream mode writes
    // (which the committed golden fixtures came from).  Large bench / test inputs use it; not with e2e.
    int threads = 0;
    // 1: staged pairs per unit instead of reads + SAM text (needs --threads; its own stream: every pair from a generator seeded by (seed, read id), so that a unit's pairs can be
    // made without making everybody else's); 2: both, from the same stream.  Only in build/agx_synth_bin.
    int pairs_bin = 0;
    int batch = 1000000;   // BATCH of the engine that will read the staged pairs (AG:37)
    std::vector<int64_t> oracle_units;   // --pairs-bin: for these units ALSO the text a checker needs, under <out>/oracle_<u>/tmp: the unit's SAM lines, and a reads file that holds the unit's
                                         // reads where they belong and a one-base placeholder record for everybody else's (4 bytes per read: 3.2 GB at 400 M pairs instead of 130)
    std::vector<int64_t> only_units;     // --pairs-bin: make ONLY these units' files (sequence, contigs, PSL, staged pairs, checker text), each exactly as long as in the whole job and with the whole job's
                                         // read numbering — a pair belongs to a unit by its own generator (seed, read id), so a unit's share of the --pairs pairs keeps its ids and its batch boundaries
                                         // (AG:1258-1259) without anybody else's reads being made.  The contigs of a unit then come from a generator of their own (seed, unit): the files are NOT those
                                         // of the whole job's run, they are a unit of the same shape.  tests/test_gpu_big.py: configs[4]'s chromosomes one at a time against the oracle
    int lean = 0;          // 1: do not write the user-level copies genome.fa, contigs.fa and tmp/_genome.fa (9 GB at whole-human size); the unit loop reads none of them
};

void die(const char *m) { std::fprintf(stderr, "agx_synth: %s\n", m); std::exit(2); }

std::vector<int64_t> parse_list(const char *s) {
    std::vector<int64_t> v;
    const char *p = s;
    while (*p) {
        char *e; long long x = std::strtoll(p, &e, 10);
        if (e == p) die("bad list");
        v.push_back(x);
        p = (*e == ',') ? e + 1 : e;
    }
    return v;
}

Params parse_args(int argc, char **argv) {
    Params P;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string a = argv[i]; const char *v = argv[i + 1];
#define OPT_S(name, field) if (a == name) { P.field = v; continue; }
#define OPT_I(name, field) if (a == name) { P.field = std::strtoll(v, nullptr, 10); continue; }
#define OPT_D(name, field) if (a == name) { P.field = std::strtod(v, nullptr); continue; }
        OPT_S("--out", out) OPT_I("--seed", seed) OPT_I("--part", part) OPT_I("--pairs", pairs) OPT_I("--L", L)
        OPT_I("--k", k) OPT_I("--coverage", coverage) OPT_I("--insert-variation", insert_variation)
        OPT_D("--snp", snp) OPT_D("--indel", indel)
        OPT_I("--contig-min", contig_min) OPT_I("--contig-max", contig_max) OPT_I("--gap-min", gap_min) OPT_I("--gap-max", gap_max)
        OPT_D("--contig-minus", contig_minus) OPT_D("--contig-split", contig_split) OPT_D("--contig-dup", contig_dup)
        OPT_D("--contig-overlap", contig_overlap) OPT_D("--contig-lowid", contig_lowid) OPT_D("--short-contig", short_contig) OPT_D("--chimeric", chimeric)
        OPT_D("--frag-mean", frag_mean) OPT_D("--frag-sd", frag_sd)
        OPT_D("--read-err", read_err) OPT_D("--read-indel", read_indel) OPT_D("--read-clip", read_clip)
        OPT_D("--read-badclip", read_badclip) OPT_D("--read-n", read_n)
        OPT_D("--multi", multi) OPT_D("--multi-near", multi_near) OPT_D("--unaligned", unaligned)
        OPT_D("--mate1-left", mate1_left) OPT_D("--mixed-len", mixed_len) OPT_I("--sam-seq", sam_seq) OPT_I("--shuffle-units", shuffle_units) OPT_I("--e2e", e2e) OPT_I("--threads", threads) OPT_I("--pairs-bin", pairs_bin) OPT_I("--batch", batch) OPT_I("--lean", lean)
        if (a == "--chroms") { P.chroms = parse_list(v); continue; }
        if (a == "--oracle-units") { P.oracle_units = parse_list(v); continue; }
        if (a == "--only-units") { P.only_units = parse_list(v); continue; }
        std::fprintf(stderr, "agx_synth: unknown option %s\n", a.c_str()); std::exit(2);
    }
    if (P.part < 1 || P.part > 10) die("--part must be 1..10");
    if (P.threads > 0 && P.e2e) die("--threads is not available with --e2e");
    if (P.threads > 256) P.threads = 256;
    if (P.pairs_bin && (P.threads <= 0 || P.e2e || P.mixed_len > 0)) die("--pairs-bin needs --threads and excludes --e2e / --mixed-len");
    if (!P.only_units.empty() && P.pairs_bin != 1) die("--only-units needs --pairs-bin 1");
#ifndef AGX_SYNTH_WITH_ENGINE
    if (P.pairs_bin) die("--pairs-bin is only available in build/agx_synth_bin (tools/agx_data.py builds it)");
#endif
    return P;
}

// 60-column FASTA body exactly as the reference writes it: newline after every 60th base and after the last.
void put_fasta_body(FILE *f, const char *s, size_t n) {
    for (size_t i = 0; i < n; i += 60) {
        size_t m = std::min<size_t>(60, n - i);
        std::fwrite(s + i, 1, m, f); std::fputc('\n', f);
    }
}

struct Unit {
    std::string ref;             // reference bases of this unit
    std::string tgt;             // target ("true sample") bases
    std::vector<int32_t> t2r;    // target index -> reference index, -1 for bases inserted in the target
};

void make_unit(Unit &U, int64_t G, const Params &P, Rng &R) {
    U.ref.resize(G);
    for (auto &c : U.ref) c = ACGT[R.below(4)];
    U.tgt.clear(); U.t2r.clear();
    U.tgt.reserve(G + G / 64); U.t2r.reserve(G + G / 64);
    for (int64_t r = 0; r < G;) {
        if (r > 50 && r + 50 < G && R.coin(P.indel)) {
            int n = (int)R.range(1, 3);
            if (R.coin(0.5)) { for (int i = 0; i < n; i++) { U.tgt.push_back(ACGT[R.below(4)]); U.t2r.push_back(-1); } }
            else { r += n; continue; }
        }
        char c = U.ref[r];
        if (R.coin(P.snp)) { char d; do d = ACGT[R.below(4)]; while (d == c); c = d; }
        U.tgt.push_back(c); U.t2r.push_back((int32_t)r);
        r++;
    }
}

struct Block { int64_t q, t, n; };

// gap-free blocks of seq[q0,q1) (indices into the target) against the reference
std::vector<Block> blocks_of(const Unit &U, int64_t q0, int64_t q1) {
    std::vector<Block> b;
    for (int64_t i = q0; i < q1; i++) {
        int32_t r = U.t2r[i];
        if (r < 0) continue;
        if (!b.empty() && b.back().q + b.back().n == i - q0 && b.back().t + b.back().n == r) b.back().n++;
        else b.push_back({i - q0, r, 1});
    }
    return b;
}

struct ContigRec { std::string name; std::string seq; };  // as written to contigs.fa

void psl_line(FILE *f, const std::vector<Block> &b, size_t i0, size_t i1, char strand, const std::string &qname,
              int64_t qsize, int64_t tsize, int64_t extra_qgap, int64_t extra_tgap) {
    int64_t matches = 0, qni = 0, qbi = 0, tni = 0, tbi = 0;
    for (size_t i = i0; i < i1; i++) {
        matches += b[i].n;
        if (i > i0) {
            int64_t qg = b[i].q - (b[i - 1].q + b[i - 1].n), tg = b[i].t - (b[i - 1].t + b[i - 1].n);
            if (qg > 0) { qni++; qbi += qg; }
            if (tg > 0) { tni++; tbi += tg; }
        }
    }
    qbi += extra_qgap; tbi += extra_tgap;   // lets a test dial a record below the identity thresholds
    int64_t qs = b[i0].q, qe = b[i1 - 1].q + b[i1 - 1].n, ts = b[i0].t, te = b[i1 - 1].t + b[i1 - 1].n;
    int64_t qs_out = qs, qe_out = qe;
    if (strand == '-') { qs_out = qsize - qe; qe_out = qsize - qs; }
    std::fprintf(f, "%lld\t0\t0\t0\t%lld\t%lld\t%lld\t%lld\t%c\t%s\t%lld\t%lld\t%lld\t0\t%lld\t%lld\t%lld\t%zu\t",
                 (long long)matches, (long long)qni, (long long)qbi, (long long)tni, (long long)tbi, strand, qname.c_str(),
                 (long long)qsize, (long long)qs_out, (long long)qe_out, (long long)tsize, (long long)ts, (long long)te, i1 - i0);
    for (size_t i = i0; i < i1; i++) std::fprintf(f, "%lld,", (long long)b[i].n);
    std::fputc('\t', f);
    for (size_t i = i0; i < i1; i++) std::fprintf(f, "%lld,", (long long)b[i].q);
    std::fputc('\t', f);
    for (size_t i = i0; i < i1; i++) std::fprintf(f, "%lld,", (long long)b[i].t);
    std::fputc('\n', f);
}

struct Aln {           // one mate of one hit, in reference orientation
    int64_t pos1;      // 1-based leftmost reference position
    std::string cigar;
};

// Build POS/CIGAR for read bases whose per-base reference index is r[i] (-1 = inserted), with
// clipL/clipR soft-clipped bases requested at the two ends.  Returns false if nothing aligns.
bool make_aln(std::vector<int32_t> r, int clipL, int clipR, Aln &A) {
    int n = (int)r.size();
    for (int i = 0; i < clipL && i < n; i++) r[i] = -2;
    for (int i = 0; i < clipR && i < n; i++) r[n - 1 - i] = -2;
    int a = 0, b = n - 1;
    while (a < n && r[a] < 0) a++;          // leading inserted bases become soft clip
    while (b >= 0 && r[b] < 0) b--;
    if (a > b) return false;
    std::string c;
    auto emit = [&](int len, char op) { if (len > 0) { c += std::to_string(len); c += op; } };
    emit(a, 'S');
    int i = a;
    while (i <= b) {
        if (r[i] >= 0) {
            int j = i;
            while (j + 1 <= b && r[j + 1] == r[j] + 1) j++;
            emit(j - i + 1, 'M');
            if (j < b) {
                int kq = j + 1;
                while (r[kq] < 0) kq++;           // r[b] >= 0 so this stops
                int ins = kq - (j + 1);
                int del = r[kq] - r[j] - 1;
                if (del < 0) return false;
                emit(ins, 'I');
                emit(del, 'D');
                i = kq;
            } else i = j + 1;
        } else i++;
    }
    emit(n - 1 - b, 'S');
    A.pos1 = (int64_t)r[a] + 1;
    A.cigar = c;
    return true;
}

struct Mate { std::string seq_fwd; std::vector<int32_t> r; };  // in reference orientation

// cut a read of length L from the target starting at target index s (forward), applying read-level noise
Mate cut_read(const Unit &U, int64_t s, int L, const Params &P, Rng &R) {
    Mate M;
    int64_t T = (int64_t)U.tgt.size();
    bool indel = R.coin(P.read_indel);
    int at = indel ? (int)R.range(10, L - 10) : -1;
    bool ins = indel && R.coin(0.5);
    int len = indel ? (int)R.range(1, ins ? 2 : 3) : 0;
    int64_t t = s;
    while ((int)M.seq_fwd.size() < L) {
        if ((int)M.seq_fwd.size() == at && indel) {
            indel = false;
            if (ins) { for (int i = 0; i < len && (int)M.seq_fwd.size() < L; i++) { M.seq_fwd.push_back(ACGT[R.below(4)]); M.r.push_back(-1); } continue; }
            t += len;
        }
        if (t >= T) { M.seq_fwd.push_back(ACGT[R.below(4)]); M.r.push_back(-1); continue; }
        char c = U.tgt[t];
        if (R.coin(P.read_err)) { char d; do d = ACGT[R.below(4)]; while (d == c); c = d; }
        if (R.coin(P.read_n)) c = 'N';
        M.seq_fwd.push_back(c); M.r.push_back(U.t2r[t]);
        t++;
    }
    return M;
}

// The same read model for the staged-pairs mode (a stream of its own): instead of two uniform draws per base for sequencing errors and N calls — 600 per pair, half of what a
// pair costs to generate, and the whole-human configuration has 400 M of them — the distance to the next base that is hit is drawn (geometric), once per event.
struct NextHit {
    double log1m; int64_t at;
    NextHit(double p, Rng &R) : log1m(p > 0 ? std::log1p(-p) : 0), at(0) { at = skip(R) ; }
    int64_t skip(Rng &R) const { if (log1m == 0) return INT64_MAX / 2; double u = R.uni(); if (u < 1e-300) u = 1e-300; return (int64_t)std::floor(std::log(u) / log1m); }      // bases that are NOT hit in front of the next one
    bool hit(int64_t i, Rng &R) { if (i < at) return false; at = i + 1 + skip(R); return true; }      // called with ascending i
};
Mate cut_read_fast(const Unit &U, int64_t s, int L, const Params &P, Rng &R) {
    Mate M;
    int64_t T = (int64_t)U.tgt.size();
    bool indel = R.coin(P.read_indel);
    int at = indel ? (int)R.range(10, L - 10) : -1;
    bool ins = indel && R.coin(0.5);
    int len = indel ? (int)R.range(1, ins ? 2 : 3) : 0;
    NextHit err(P.read_err, R), nn(P.read_n, R);
    int64_t t = s;
    M.seq_fwd.reserve((size_t)L); M.r.reserve((size_t)L);
    while ((int)M.seq_fwd.size() < L) {
        const int i = (int)M.seq_fwd.size();
        if (i == at && indel) {
            indel = false;
            if (ins) { for (int j = 0; j < len && (int)M.seq_fwd.size() < L; j++) { M.seq_fwd.push_back(ACGT[R.below(4)]); M.r.push_back(-1); } continue; }
            t += len;
        }
        if (t >= T) { M.seq_fwd.push_back(ACGT[R.below(4)]); M.r.push_back(-1); continue; }
        char c = U.tgt[t];
        if (err.hit(i, R)) { char d; do d = ACGT[R.below(4)]; while (d == c); c = d; }
        if (nn.hit(i, R)) c = 'N';
        M.seq_fwd.push_back(c); M.r.push_back(U.t2r[t]);
        t++;
    }
    return M;
}

void mkdirs(const std::string &p) { mkdir(p.c_str(), 0777); }

}  // namespace

int main(int argc, char **argv) {
    Params P = parse_args(argc, argv);
    Rng R(P.seed);
    mkdirs(P.out); mkdirs(P.out + "/tmp");
    auto path = [&](const std::string &s) { return P.out + "/" + s; };
    auto open = [&](const std::string &s) { FILE *f = std::fopen(path(s).c_str(), "w"); if (!f) die("cannot open output"); return f; };

    // ---- units: chromosome c, part q (formalizeGenome split rule: boundaries at multiples of size/p) ----
    std::vector<int64_t> unit_len; std::vector<int> unit_chrom;
    for (size_t c = 0; c < P.chroms.size(); c++) {
        int64_t s = P.chroms[c], step = s / P.part;
        for (int q = 0; q < P.part; q++) { unit_len.push_back(q + 1 < P.part ? step : s - step * (P.part - 1)); unit_chrom.push_back((int)c); }
    }
    int NU = (int)unit_len.size();
    std::vector<Unit> units(NU);
    for (int64_t u : P.only_units) if (u < 0 || u >= NU) die("--only-units: no such unit");
    auto wanted = [&](int u) { return P.only_units.empty() || std::find(P.only_units.begin(), P.only_units.end(), (int64_t)u) != P.only_units.end(); };
    if (P.pairs_bin) {      // (the staged-pairs mode has a stream of its own anyway: every unit's sequences from a generator of their own, all units side by side — 3.1 Gb take a minute on one thread)
        std::atomic<int> next_u(0); std::vector<std::thread> th;
        for (int t = 0; t < std::min(P.threads, NU); t++) th.emplace_back([&] { for (int u; (u = next_u.fetch_add(1)) < NU;) { if (!wanted(u)) continue; Rng Ru(P.seed * 0x2545F4914F6CDD1Dull + (uint64_t)(u + 1) * 0x9E3779B97F4A7C15ull); make_unit(units[u], unit_len[u], P, Ru); } });
        for (auto &t : th) t.join();
    } else
    for (int u = 0; u < NU; u++) make_unit(units[u], unit_len[u], P, R);

    // ---- genome files ----
    {
        FILE *g = open("genome.fa"), *g0 = open("tmp/_genome.fa");
        int u = 0;
        for (size_t c = 0; c < P.chroms.size(); c++) {
            if (!P.lean) std::fprintf(g, ">chr%zu synthetic\n", c + 1);
            std::string whole;
            for (int q = 0; q < P.part; q++, u++) {
                if (!wanted(u)) continue;
                if (!P.lean) whole += units[u].ref;
                FILE *gu = open("tmp/_genome." + std::to_string(u) + ".fa");
                std::fputs(">0\n", gu); put_fasta_body(gu, units[u].ref.data(), units[u].ref.size()); std::fclose(gu);
                if (!P.lean) { std::fprintf(g0, ">%d\n", u); put_fasta_body(g0, units[u].ref.data(), units[u].ref.size()); }
            }
            if (!P.lean) put_fasta_body(g, whole.data(), whole.size());
        }
        std::fclose(g); std::fclose(g0);
    }

    // ---- contigs + PSL ----
    {
        FILE *cf = open("contigs.fa"), *tc = open("tmp/_contigs.fa"), *chaff = open("tmp/_chaff.fa");
        int64_t seqID = 0, realID = 0, nameID = 0;
        Rng &Rwhole = R;
        for (int u = 0; u < NU; u++) {
            if (!wanted(u)) continue;
            const Unit &U = units[u];
            Rng Rown(P.seed * 0xA0761D6478BD642Full + (uint64_t)(u + 1) * 0xE7037ED1A0B428DBull);
            Rng &R = P.only_units.empty() ? Rwhole : Rown;      // (--only-units: a unit's contigs do not depend on which other units are made)
            // every unit's PSL is written against the same tmp/_contigs.fa, so contigs of other units
            // simply have no line in this unit's file
            FILE *psl = open("tmp/_contigs_genome." + std::to_string(u) + ".psl");
            int64_t T = (int64_t)U.tgt.size(), t = R.range(0, P.gap_max);
            int64_t prev_s = -1, prev_e = -1;
            while (t + 300 < T) {
                int64_t len = R.range(P.contig_min, P.contig_max);
                if (t + len > T) len = T - t;
                if (len <= 250) break;
                int64_t s = t, e = t + len;
                std::string seq = U.tgt.substr(s, len);
                const bool chim = R.coin(P.chimeric) && T > 4000;
                if (chim) {                        // junk + a second piece from a random other place; only the first piece is reported to the threading PSL
                    for (int j = 0; j < 300; j++) seq.push_back(ACGT[R.below(4)]);
                    const int64_t l2 = std::min<int64_t>(len, 1500), s2 = R.range(0, T - l2 - 1);
                    seq += U.tgt.substr(s2, l2);
                }
                char strand = R.coin(P.contig_minus) ? '-' : '+';
                std::string name = "contig_" + std::to_string(nameID++);
                std::string file_seq = strand == '-' ? revcomp(seq) : seq;
                if (!P.lean) { std::fprintf(cf, ">%s\n", name.c_str()); put_fasta_body(cf, file_seq.data(), file_seq.size()); }
                std::string qname = std::to_string(seqID) + "." + std::to_string(realID);
                std::fprintf(tc, ">%s\n", qname.c_str()); put_fasta_body(tc, file_seq.data(), file_seq.size());
                seqID++; realID++;

                std::vector<Block> b = blocks_of(U, s, e);
                const int64_t qsize = (int64_t)seq.size();
                if (strand == '-' && chim) for (auto &blk : b) blk.q += qsize - len;   // blocks are in reverse-complemented query coordinates
                if (!b.empty()) {
                    if (R.coin(P.contig_lowid)) {
                        psl_line(psl, b, 0, b.size(), strand, qname, qsize, (int64_t)U.ref.size(), len, 0);   // fails identity filter
                    } else if (b.size() >= 2 && R.coin(P.contig_split)) {
                        size_t cut = 1 + R.below(b.size() - 1);
                        psl_line(psl, b, 0, cut, strand, qname, qsize, (int64_t)U.ref.size(), 0, 0);
                        psl_line(psl, b, cut, b.size(), strand, qname, qsize, (int64_t)U.ref.size(), 0, 0);
                    } else {
                        psl_line(psl, b, 0, b.size(), strand, qname, qsize, (int64_t)U.ref.size(), 0, 0);
                    }
                    if (R.coin(P.contig_dup)) {   // a second, repeat-like placement of the same contig elsewhere
                        int64_t G = (int64_t)U.ref.size();
                        int64_t sub = std::max<int64_t>(len * 6 / 10, 210); if (sub > len) sub = len;
                        if (G > sub + 10) {
                            int64_t where = R.range(0, G - sub - 1);
                            std::vector<Block> d = {{0, where, sub}};
                            psl_line(psl, d, 0, 1, strand, qname, qsize, G, 0, 0);
                        }
                    }
                }
                prev_s = s; prev_e = e; (void)prev_s;
                if (R.coin(P.contig_overlap)) t = std::max<int64_t>(s + 1, e - R.range(50, 400));
                else t = e + R.range(P.gap_min, P.gap_max);
                if (R.coin(P.short_contig)) {   // <=200 bp contigs go to chaff and get no seqID
                    int64_t sl = R.range(50, 200);
                    if (t + sl < T) {
                        std::string nm = "contig_" + std::to_string(nameID++);
                        std::string ss = U.tgt.substr(t, sl);
                        if (!P.lean) { std::fprintf(cf, ">%s\n", nm.c_str()); put_fasta_body(cf, ss.data(), ss.size()); }
                        std::fprintf(chaff, ">%s\n", nm.c_str()); put_fasta_body(chaff, ss.data(), ss.size());
                    }
                }
            }
            (void)prev_e;
            std::fclose(psl);
            if (P.e2e) {           // what the pblat/blat stub replays for this unit
                mkdirs(P.out + "/stub");
                std::string a = path("tmp/_contigs_genome." + std::to_string(u) + ".psl"), b = path("stub/_contigs_genome." + std::to_string(u) + ".psl");
                FILE *fi = std::fopen(a.c_str(), "rb"), *fo = std::fopen(b.c_str(), "wb");
                if (!fi || !fo) die("cannot copy psl");
                char buf[65536]; size_t n;
                while ((n = std::fread(buf, 1, sizeof buf, fi)) > 0) std::fwrite(buf, 1, n, fo);
                std::fclose(fi); std::fclose(fo);
            }
        }
        std::fclose(cf); std::fclose(tc); std::fclose(chaff);
    }

#ifdef AGX_SYNTH_WITH_ENGINE
    // ---- staged pairs per unit (--pairs-bin) ----
    if (P.pairs_bin) {
        const int64_t N = P.pairs, C = 65536, NC = (N + C - 1) / C;
        const int L = P.L;
        std::vector<double> cum(NU); { double tot = 0; for (auto l : unit_len) tot += (double)l; double a = 0; for (int u = 0; u < NU; u++) { a += (double)unit_len[u] / tot; cum[u] = a; } cum[NU - 1] = 2.0; }
        struct Emit { Aln l, r; bool secondary; };
        struct OnePair { int u; bool m1_left; std::string m1, m2, left_fwd, right_fwd; std::vector<Emit> emits; };
        // pair `id` from its own generator; false (and nothing else drawn) if it belongs to another unit than `want` (want < 0: whoever)
        auto gen_pair = [&](int64_t id, int want, OnePair &o) -> bool {
            Rng R(P.seed * 0x9E3779B97F4A7C15ull + (uint64_t)(id + 1) * 0xD1B54A32D192ED03ull);
            const double pick = R.uni();
            int u = 0; while (pick >= cum[u]) u++;
            if (want >= 0 && u != want) return false;
            o.u = u; o.emits.clear();
            const Unit &U = units[u];
            const int64_t T = (int64_t)U.tgt.size();
            int64_t f = (int64_t)std::llround(P.frag_mean + P.frag_sd * R.normal());
            if (f < L + 1) f = L + 1;
            if (f > T) f = T;
            const int64_t s0 = R.range(0, T - f);
            ::Mate left = cut_read_fast(U, s0, L, P, R), right = cut_read_fast(U, s0 + f - L, L, P, R);
            o.m1_left = R.coin(P.mate1_left);
            const std::string right_file = revcomp(right.seq_fwd);
            o.m1 = o.m1_left ? left.seq_fwd : right_file; o.m2 = o.m1_left ? right_file : left.seq_fwd; o.left_fwd = left.seq_fwd; o.right_fwd = right.seq_fwd;
            if (R.coin(P.unaligned)) return true;
            int clipLl = 0, clipLr = 0, clipRl = 0, clipRr = 0;
            if (R.coin(P.read_clip)) { (R.coin(0.5) ? clipLl : clipLr) = (int)R.range(1, 15); }
            if (R.coin(P.read_clip)) { (R.coin(0.5) ? clipRl : clipRr) = (int)R.range(1, 15); }
            if (R.coin(P.read_badclip)) { (R.coin(0.5) ? clipLr : clipRl) = (int)R.range(L * 45 / 100, L * 55 / 100); }
            Aln al, ar;
            if (!make_aln(left.r, clipLl, clipLr, al) || !make_aln(right.r, clipRl, clipRr, ar)) return true;
            o.emits.push_back(Emit{al, ar, false});
            if (R.coin(P.multi)) {
                const int64_t G = (int64_t)U.ref.size();
                Aln bl, br;
                const std::string allM = std::to_string(L) + "M";
                if (R.coin(P.multi_near)) {
                    const int64_t d = R.range(-(L - 1), L - 1);
                    bl.pos1 = std::max<int64_t>(1, al.pos1 + d); br.pos1 = std::max<int64_t>(1, ar.pos1 + d);
                } else {
                    const int64_t w = R.range(1, std::max<int64_t>(1, G - f - 1));
                    bl.pos1 = w; br.pos1 = w + f - L;
                }
                if (bl.pos1 + L <= G && br.pos1 + L <= G) { bl.cigar = br.cigar = allM; o.emits.push_back(Emit{bl, br, true}); }
            }
            return true;
        };
        // the two SAM lines of one emitted alignment, as the text mode writes them (m = 0: mate1's line)
        auto sam_line = [&](const OnePair &o, int64_t id, const Emit &e, int m, std::string &dst) {
            char buf[512];
            const bool is_left = (m == 0) == o.m1_left;
            const int flag = 0x1 | 0x2 | (m == 0 ? 0x40 : 0x80) | (is_left ? 0x20 : 0x10) | (e.secondary ? 0x100 : 0);
            const Aln &A = is_left ? e.l : e.r, &B = is_left ? e.r : e.l;
            const long long tlen = is_left ? (long long)(e.r.pos1 + L - e.l.pos1) : -(long long)(e.r.pos1 + L - e.l.pos1);
            int w = std::snprintf(buf, sizeof buf, "%lld\t%d\t%d\t%lld\t42\t%s\t=\t%lld\t%lld\t", (long long)id, flag, o.u, (long long)A.pos1, A.cigar.c_str(), (long long)B.pos1, tlen);
            dst.assign(buf, (size_t)w);
            if (P.sam_seq) { dst += is_left ? o.left_fwd : o.right_fwd; dst.push_back('\t'); dst.append((size_t)L, 'I'); w = std::snprintf(buf, sizeof buf, "\tAS:i:%d\tYS:i:%d\tYT:Z:CP", 2 * L - 6, 2 * L - 4); dst.append(buf, (size_t)w); }
            else dst += "*\t*";
        };
        if (P.pairs_bin == 2) {      // the text files of the same stream (small inputs: one thread)
            FILE *rf = open("tmp/_reads.fa");
            std::vector<FILE *> sam(NU);
            for (int u = 0; u < NU; u++) sam[u] = open("tmp/_reads_genome." + std::to_string(u) + ".bowtie");
            OnePair o; std::string line;
            for (int64_t id = 0; id < N; id++) {
                gen_pair(id, -1, o);
                std::fprintf(rf, ">%lld\n%s\n>%lld\n%s\n", (long long)id, o.m1.c_str(), (long long)id, o.m2.c_str());
                for (const Emit &e : o.emits) for (int m = 0; m < 2; m++) { sam_line(o, id, e, m, line); std::fwrite(line.data(), 1, line.size(), sam[o.u]); std::fputc('\n', sam[o.u]); }
            }
            std::fclose(rf); for (auto f : sam) std::fclose(f);
        }
        struct MemSink : agx::StageSink { std::vector<std::vector<char>> keep; void *take(int, size_t bytes) override { keep.emplace_back(bytes + 64); return keep.back().data(); } };
        for (int u = 0; u < NU; u++) {
            if (!wanted(u)) continue;
            struct LinePair { agx::Mate m1, m2; };
            struct Chunk { std::vector<LinePair> lp; std::vector<agx_run> runs; std::vector<int64_t> ids; std::string bases, sam; bool ready = false; };      // ids / bases: the unit's pairs of this chunk (2 x L bytes each); sam: their lines, for --oracle-units
            std::vector<Chunk> chunks((size_t)NC);
            const bool for_oracle = std::find(P.oracle_units.begin(), P.oracle_units.end(), (int64_t)u) != P.oracle_units.end();
            FILE *o_reads = nullptr, *o_sam = nullptr; int64_t o_next = 0;       // o_next: the first read id the checker's reads file does not hold yet
            std::string filler;
            if (for_oracle) {
                const std::string od = P.out + "/oracle_" + std::to_string(u), su = std::to_string(u);
                mkdirs(od); mkdirs(od + "/tmp");
                o_reads = std::fopen((od + "/tmp/_reads.fa").c_str(), "w"); o_sam = std::fopen((od + "/tmp/_reads_genome." + su + ".bowtie").c_str(), "w");
                if (!o_reads || !o_sam) die("cannot open output");
                for (const std::string &f : {"_genome." + su + ".fa", std::string("_contigs.fa"), "_contigs_genome." + su + ".psl"}) (void)!symlink(("../../tmp/" + f).c_str(), (od + "/tmp/" + f).c_str());
                for (int i = 0; i < 65536; i++) filler += ">\nN\n>\nN\n";
            }
            auto fill_to = [&](int64_t id) { while (o_next < id) { const int64_t m = std::min<int64_t>(id - o_next, 65536); std::fwrite(filler.data(), 1, (size_t)m * 8, o_reads); o_next += m; } };
            std::atomic<int64_t> next_chunk(0);
            std::mutex mu; std::condition_variable cv; int64_t consumed = 0; bool failed = false;
            auto make_chunk = [&](int64_t c) {
                Chunk &K = chunks[(size_t)c];
                OnePair o; std::string l1, l2;
                for (int64_t id = c * C; id < std::min(N, (c + 1) * C); id++) {
                    if (!gen_pair(id, u, o)) continue;
                    K.ids.push_back(id); K.bases += o.m1; K.bases += o.m2;
                    for (const Emit &e : o.emits) {
                        sam_line(o, id, e, 0, l1); sam_line(o, id, e, 1, l2);
                        LinePair lp; agx::parse_sam_line(l1.data(), l1.size(), lp.m1, K.runs); agx::parse_sam_line(l2.data(), l2.size(), lp.m2, K.runs);
                        K.lp.push_back(lp);
                        if (for_oracle) { K.sam += l1; K.sam.push_back('\n'); K.sam += l2; K.sam.push_back('\n'); }
                    }
                }
            };
            std::vector<std::thread> th;
            for (int t = 0; t < P.threads; t++) th.emplace_back([&] {
                for (;;) {
                    const int64_t c = next_chunk.fetch_add(1);
                    if (c >= NC) return;
                    { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return failed || c < consumed + 8 * (int64_t)P.threads; }); if (failed) return; }
                    try { make_chunk(c); } catch (...) { std::lock_guard<std::mutex> g(mu); failed = true; }
                    { std::lock_guard<std::mutex> g(mu); chunks[(size_t)c].ready = true; }
                    cv.notify_all();
                }
            });
            agx::Pairs PP; PP.stride = ((agx_u32)L + 15u) & ~15u;
            try {
                agx::PairRules rules(PP, (unsigned long long)N, P.batch);
                size_t placed = 0; agx_u32 n_ids = 0; bool go = true;          // hits whose read slot is set; distinct read ids among them
                for (int64_t c = 0; c < NC; c++) {
                    { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return failed || chunks[(size_t)c].ready; }); if (failed) die("staged pairs: a worker failed"); }
                    Chunk &K = chunks[(size_t)c];
                    size_t at = 0;                                               // the chunk's pair whose id the line pairs have reached
                    for (const LinePair &lp : K.lp) {
                        if (!go) break;
                        go = rules.consume(lp.m1, lp.m2, K.runs.data());
                        for (; placed < PP.hits.size(); placed++) {             // (one at most) a kept hit: its read slot, and the pair's bases the first time
                            const agx_u32 id = rules.hit_id[placed];
                            if (placed == 0 || rules.hit_id[placed - 1] != id) {
                                while (at < K.ids.size() && K.ids[at] != (int64_t)id) at++;
                                if (at == K.ids.size()) die("staged pairs: a kept hit without its reads");
                                n_ids++;
                                const size_t o0 = PP.bases.size(); PP.bases.resize(o0 + 2 * (size_t)PP.stride, 'N');
                                std::memcpy(&PP.bases[o0], K.bases.data() + at * 2 * (size_t)L, (size_t)L); std::memcpy(&PP.bases[o0 + PP.stride], K.bases.data() + (at * 2 + 1) * (size_t)L, (size_t)L);
                            }
                            if (PP.hits[placed].len != (agx_u32)L) die("staged pairs: a CIGAR that is not as long as the read");
                            PP.hits[placed].slot1 = 2 * (n_ids - 1);
                        }
                    }
                    if (for_oracle) {
                        char hdr[32];
                        for (size_t i = 0; i < K.ids.size(); i++) {
                            fill_to(K.ids[i]);
                            const int n = std::snprintf(hdr, sizeof hdr, ">%lld\n", (long long)K.ids[i]);
                            for (int m = 0; m < 2; m++) { std::fwrite(hdr, 1, (size_t)n, o_reads); std::fwrite(K.bases.data() + (i * 2 + (size_t)m) * (size_t)L, 1, (size_t)L, o_reads); std::fputc('\n', o_reads); }
                            o_next = K.ids[i] + 1;
                        }
                        std::fwrite(K.sam.data(), 1, K.sam.size(), o_sam); std::string().swap(K.sam);
                    }
                    Chunk().lp.swap(K.lp); std::vector<agx_run>().swap(K.runs); std::vector<int64_t>().swap(K.ids); std::string().swap(K.bases);
                    { std::lock_guard<std::mutex> g(mu); consumed = c + 1; }
                    cv.notify_all();
                }
                for (auto &t : th) t.join();
                if (for_oracle) { fill_to(N); std::fclose(o_reads); std::fclose(o_sam); }
                PP.n_kept = PP.hits.size(); PP.n_slots = 2 * n_ids;
                MemSink sink; agx::StagedPairs S;
                agx::stage_pairs(PP, (agx_u32)P.k, (unsigned)P.threads, sink, S);
                const std::vector<agx_u8> ob = agx::other_bytes_of(PP, S);
                agx::write_pairs_file(agx::pairsfile::path_of(P.out + "/tmp", u), S, ob.data(), (agx_u32)P.k, (agx_u32)P.batch);
            } catch (const agx::Error &e) { std::fprintf(stderr, "agx_synth: staged pairs of unit %d: %s\n", u, e.msg.c_str()); std::exit(2); }
        }
    } else
#endif
    // ---- reads + SAM, chunked mode (--threads) ----
    if (P.threads > 0) {
        const int64_t N = P.pairs, C = 65536, NC = (N + C - 1) / C;
        const int L = P.L;
        std::vector<double> cum(NU); { double tot = 0; for (auto l : unit_len) tot += (double)l; double a = 0; for (int u = 0; u < NU; u++) { a += (double)unit_len[u] / tot; cum[u] = a; } cum[NU - 1] = 2.0; }
        struct Chunk { std::string reads; std::vector<std::string> sam; bool ready = false; };
        std::vector<Chunk> chunks((size_t)NC);
        std::atomic<int64_t> next_chunk(0);
        std::mutex mu; std::condition_variable cv; int64_t written = 0;                 // chunks [0, written) are on disk; workers stay at most 4 * threads chunks ahead
        auto make_chunk = [&](int64_t c) {
            Chunk &K = chunks[(size_t)c]; K.sam.assign(NU, std::string());
            uint64_t sd = P.seed * 0x9E3779B97F4A7C15ull + (uint64_t)(c + 1) * 0xD1B54A32D192ED03ull;
            Rng R(sd);
            char buf[512];
            K.reads.reserve((size_t)C * (2 * L + 24));
            for (int64_t id = c * C; id < std::min(N, (c + 1) * C); id++) {
                const double pick = R.uni();
                int u = 0; while (pick >= cum[u]) u++;
                const Unit &U = units[u];
                const int64_t T = (int64_t)U.tgt.size();
                int64_t f = (int64_t)std::llround(P.frag_mean + P.frag_sd * R.normal());
                if (f < L + 1) f = L + 1;
                if (f > T) f = T;
                const int64_t s0 = R.range(0, T - f);
                Mate left = cut_read(U, s0, L, P, R), right = cut_read(U, s0 + f - L, L, P, R);
                const bool m1_left = R.coin(P.mate1_left);
                const std::string left_file = left.seq_fwd, right_file = revcomp(right.seq_fwd);
                const std::string &m1 = m1_left ? left_file : right_file, &m2 = m1_left ? right_file : left_file;
                int n = std::snprintf(buf, sizeof buf, ">%lld\n", (long long)id);
                K.reads.append(buf, n); K.reads += m1; K.reads.push_back('\n'); K.reads.append(buf, n); K.reads += m2; K.reads.push_back('\n');
                if (R.coin(P.unaligned)) continue;
                int clipLl = 0, clipLr = 0, clipRl = 0, clipRr = 0;
                if (R.coin(P.read_clip)) { (R.coin(0.5) ? clipLl : clipLr) = (int)R.range(1, 15); }
                if (R.coin(P.read_clip)) { (R.coin(0.5) ? clipRl : clipRr) = (int)R.range(1, 15); }
                if (R.coin(P.read_badclip)) { (R.coin(0.5) ? clipLr : clipRl) = (int)R.range(L * 45 / 100, L * 55 / 100); }
                Aln al, ar;
                if (!make_aln(left.r, clipLl, clipLr, al) || !make_aln(right.r, clipRl, clipRr, ar)) continue;
                std::string qual; if (P.sam_seq) qual.assign(L, 'I');
                auto emit = [&](bool secondary, const Aln &aL, const Aln &aR) {
                    for (int m = 0; m < 2; m++) {
                        const bool is_left = (m == 0) == m1_left;
                        const int flag = 0x1 | 0x2 | (m == 0 ? 0x40 : 0x80) | (is_left ? 0x20 : 0x10) | (secondary ? 0x100 : 0);
                        const Aln &A = is_left ? aL : aR, &B = is_left ? aR : aL;
                        const long long tlen = is_left ? (long long)(aR.pos1 + L - aL.pos1) : -(long long)(aR.pos1 + L - aL.pos1);
                        std::string &dst = K.sam[u];
                        int w = std::snprintf(buf, sizeof buf, "%lld\t%d\t%d\t%lld\t42\t%s\t=\t%lld\t%lld\t", (long long)id, flag, u, (long long)A.pos1, A.cigar.c_str(), (long long)B.pos1, tlen);
                        dst.append(buf, w);
                        if (P.sam_seq) { dst += is_left ? left.seq_fwd : right.seq_fwd; dst.push_back('\t'); dst += qual; w = std::snprintf(buf, sizeof buf, "\tAS:i:%d\tYS:i:%d\tYT:Z:CP\n", 2 * L - 6, 2 * L - 4); dst.append(buf, w); }
                        else dst += "*\t*\n";
                    }
                };
                emit(false, al, ar);
                if (R.coin(P.multi)) {
                    const int64_t G = (int64_t)U.ref.size();
                    Aln bl, br;
                    const std::string allM = std::to_string(L) + "M";
                    if (R.coin(P.multi_near)) {
                        const int64_t d = R.range(-(L - 1), L - 1);
                        bl.pos1 = std::max<int64_t>(1, al.pos1 + d); br.pos1 = std::max<int64_t>(1, ar.pos1 + d);
                    } else {
                        const int64_t w = R.range(1, std::max<int64_t>(1, G - f - 1));
                        bl.pos1 = w; br.pos1 = w + f - L;
                    }
                    if (bl.pos1 + L <= G && br.pos1 + L <= G) { bl.cigar = br.cigar = allM; emit(true, bl, br); }
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < P.threads; t++) th.emplace_back([&] {
            for (;;) {
                const int64_t c = next_chunk.fetch_add(1);
                if (c >= NC) return;
                { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return c < written + 4 * (int64_t)P.threads; }); }
                make_chunk(c);
                { std::lock_guard<std::mutex> g(mu); chunks[(size_t)c].ready = true; }
                cv.notify_all();
            }
        });
        FILE *rf = open("tmp/_reads.fa");
        std::vector<FILE *> sam(NU);
        for (int u = 0; u < NU; u++) sam[u] = open("tmp/_reads_genome." + std::to_string(u) + ".bowtie");
        for (int64_t c = 0; c < NC; c++) {
            { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return chunks[(size_t)c].ready; }); }
            Chunk &K = chunks[(size_t)c];
            std::fwrite(K.reads.data(), 1, K.reads.size(), rf);
            for (int u = 0; u < NU; u++) std::fwrite(K.sam[u].data(), 1, K.sam[u].size(), sam[u]);
            std::string().swap(K.reads); std::vector<std::string>().swap(K.sam);
            { std::lock_guard<std::mutex> g(mu); written = c + 1; }
            cv.notify_all();
        }
        for (auto &t : th) t.join();
        std::fclose(rf);
        for (auto f : sam) std::fclose(f);
    } else
    // ---- reads + SAM ----
    {
        int64_t N = P.pairs;
        std::vector<int> unit_of(N);
        {
            double tot = 0; for (auto l : unit_len) tot += (double)l;
            int64_t i = 0;
            for (int u = 0; u < NU; u++) {
                int64_t n = (u + 1 == NU) ? N - i : (int64_t)std::floor(N * (unit_len[u] / tot));
                for (int64_t j = 0; j < n && i < N; j++) unit_of[i++] = u;
            }
            if (P.shuffle_units) for (int64_t j = N - 1; j > 0; j--) std::swap(unit_of[j], unit_of[R.below(j + 1)]);
        }
        FILE *rf = open("tmp/_reads.fa");
        FILE *u1 = nullptr, *u2 = nullptr, *gsam = nullptr;
        if (P.e2e) {
            mkdirs(P.out + "/stub");
            u1 = open("reads_1.fa"); u2 = open("reads_2.fa"); gsam = open("stub/reads_genome.sam");
            std::fputs("@HD\tVN:1.0\tSO:unsorted\n", gsam);
            for (int u = 0; u < NU; u++) std::fprintf(gsam, "@SQ\tSN:%d\tLN:%lld\n", u, (long long)unit_len[u]);
            std::fputs("@PG\tID:bowtie2\tPN:bowtie2\tVN:stub\n", gsam);
        }
        std::vector<FILE *> sam(NU);
        for (int u = 0; u < NU; u++) sam[u] = open("tmp/_reads_genome." + std::to_string(u) + ".bowtie");
        for (int64_t id = 0; id < N; id++) {
            int L = P.L;
            if (P.mixed_len > 0 && R.coin(P.mixed_len)) { const int pct[3] = {60, 75, 90}; L = std::max(P.k + 6, P.L * pct[R.range(0, 2)] / 100); }      // (no draw at all when the option is off: the committed fixtures came from that stream)
            std::string qual(L, 'I');
            int u = unit_of[id];
            const Unit &U = units[u];
            int64_t T = (int64_t)U.tgt.size();
            int64_t f = (int64_t)std::llround(P.frag_mean + P.frag_sd * R.normal());
            if (f < L + 1) f = L + 1;
            if (f > T) f = T;
            int64_t s = R.range(0, T - f);
            Mate left = cut_read(U, s, L, P, R), right = cut_read(U, s + f - L, L, P, R);
            bool m1_left = R.coin(P.mate1_left);
            // sequences as they appear in the reads file: left mate forward, right mate reverse-complemented
            std::string left_file = left.seq_fwd, right_file = revcomp(right.seq_fwd);
            const std::string &m1 = m1_left ? left_file : right_file, &m2 = m1_left ? right_file : left_file;
            std::fprintf(rf, ">%lld\n%s\n>%lld\n%s\n", (long long)id, m1.c_str(), (long long)id, m2.c_str());
            if (P.e2e) { std::fprintf(u1, ">frag%lld/1 synthetic\n%s\n", (long long)id, m1.c_str()); std::fprintf(u2, ">frag%lld/2 synthetic\n%s\n", (long long)id, m2.c_str()); }
            auto unaligned_pair = [&]() {
                if (!P.e2e) return;
                std::fprintf(gsam, "%lld\t77\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\tYT:Z:UP\n%lld\t141\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\tYT:Z:UP\n",
                             (long long)id, m1.c_str(), qual.c_str(), (long long)id, m2.c_str(), qual.c_str());
            };
            if (R.coin(P.unaligned)) { unaligned_pair(); continue; }

            int clipLl = 0, clipLr = 0, clipRl = 0, clipRr = 0;
            if (R.coin(P.read_clip)) { (R.coin(0.5) ? clipLl : clipLr) = (int)R.range(1, 15); }
            if (R.coin(P.read_clip)) { (R.coin(0.5) ? clipRl : clipRr) = (int)R.range(1, 15); }
            if (R.coin(P.read_badclip)) { (R.coin(0.5) ? clipLr : clipRl) = (int)R.range(L * 45 / 100, L * 55 / 100); }
            Aln al, ar;
            if (!make_aln(left.r, clipLl, clipLr, al) || !make_aln(right.r, clipRl, clipRr, ar)) { unaligned_pair(); continue; }

            auto emit = [&](bool secondary, const Aln &aL, const Aln &aR) {
                // mate1 first, then mate2 — bowtie2 prints the pair in that order
                for (int m = 0; m < 2; m++) {
                    bool is_left = (m == 0) == m1_left;
                    int flag = 0x1 | 0x2 | (m == 0 ? 0x40 : 0x80) | (is_left ? 0x20 : 0x10) | (secondary ? 0x100 : 0);
                    const Aln &A = is_left ? aL : aR, &B = is_left ? aR : aL;
                    const std::string &seq = is_left ? left.seq_fwd : right.seq_fwd;
                    long long tlen = is_left ? (long long)(aR.pos1 + L - aL.pos1) : -(long long)(aR.pos1 + L - aL.pos1);
                    for (FILE *dst : {sam[u], gsam}) {
                        if (!dst) continue;
                        if (P.sam_seq)
                            std::fprintf(dst, "%lld\t%d\t%d\t%lld\t42\t%s\t=\t%lld\t%lld\t%s\t%s\tAS:i:%d\tYS:i:%d\tYT:Z:CP\n",
                                         (long long)id, flag, u, (long long)A.pos1, A.cigar.c_str(), (long long)B.pos1, tlen,
                                         seq.c_str(), qual.c_str(), 2 * L - 6, 2 * L - 4);
                        else
                            std::fprintf(dst, "%lld\t%d\t%d\t%lld\t42\t%s\t=\t%lld\t%lld\t*\t*\n",
                                         (long long)id, flag, u, (long long)A.pos1, A.cigar.c_str(), (long long)B.pos1, tlen);
                    }
                }
            };
            emit(false, al, ar);
            if (R.coin(P.multi)) {
                int64_t G = (int64_t)U.ref.size();
                Aln bl, br;
                std::string allM = std::to_string(L) + "M";
                if (R.coin(P.multi_near)) {       // second hit within one read length of the first: must be skipped
                    int64_t d = R.range(-(L - 1), L - 1);
                    bl.pos1 = std::max<int64_t>(1, al.pos1 + d); br.pos1 = std::max<int64_t>(1, ar.pos1 + d);
                } else {
                    int64_t w = R.range(1, std::max<int64_t>(1, G - f - 1));
                    bl.pos1 = w; br.pos1 = w + f - L;
                }
                if (bl.pos1 + L <= G && br.pos1 + L <= G) { bl.cigar = br.cigar = allM; emit(true, bl, br); }
            }
        }
        std::fclose(rf);
        if (P.e2e) { std::fclose(u1); std::fclose(u2); std::fclose(gsam); }
        for (auto f : sam) std::fclose(f);
    }

    // ---- command + checkpoint, so `AlignGraph --resume` replays exactly this tmp/ ----
    {
        FILE *c = open("tmp/_command.txt");
        std::fprintf(c, "--read1\nreads_1.fa\n--read2\nreads_2.fa\n--contig\ncontigs.fa\n--genome\ngenome.fa\n"
                        "--distanceLow\n100\n--distanceHigh\n1500\n--extendedContig\nextended.fa\n--remainingContig\nremaining.fa\n"
                        "--kMer\n%d\n--coverage\n%d\n--insertVariation\n%d\n--part\n%d\n", P.k, P.coverage, P.insert_variation, P.part);
        std::fclose(c);
        FILE *cp = open("tmp/_checkpoint.txt"); std::fputs("0\n", cp); std::fclose(cp);
        if (!P.e2e) {
            FILE *r1 = open("reads_1.fa"); std::fclose(r1);   // --resume re-opens but never reads them
            FILE *r2 = open("reads_2.fa"); std::fclose(r2);
        }
        FILE *m = open("synth_meta.txt");
        std::fprintf(m, "units %d\npairs %lld\nL %d\nk %d\ncoverage %d\ninsert_variation %d\nseed %llu\n", NU, (long long)P.pairs, P.L, P.k,
                     P.coverage, P.insert_variation, (unsigned long long)P.seed);
        std::fprintf(m, "pairs_bin %d\n", P.pairs_bin);
        std::fprintf(m, "only_units %zu\n", P.only_units.size());
        for (int u = 0; u < NU; u++) std::fprintf(m, "unit %d len %lld\n", u, (long long)unit_len[u]);
        std::fclose(m);
    }
    return 0;
}
