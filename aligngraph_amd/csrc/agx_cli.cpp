// agx_cli.cpp — `AlignGraph_amd`: the reference's command line, tmp/ contract, checkpoint/resume and final outputs around the
// MI355X engine.
//
// Mirrors main() of /root/reference/AlignGraph/AlignGraph.cpp ("AG", lines 4696-4796) stage by stage:
//   argv -> command.txt -> parameters (getParameters, AG:4329-4646)  ·  formalizeInput / formalizeGenome (AG:3228-3518)
//   bowtie2 + pblat|blat through system() with the reference's exact command strings, two threads (AG:3581-3735), distributeAlignments
//   (AG:3545-3579)  ·  tmp/_checkpoint.txt + --resume (AG:4648-4680, 4724-4760)  ·  the unit loop (AG:4765-4783) = libagx, units spread
//   over the visible GPUs  ·  refinement (AG:2864-3195) -> --extendedContig / --remainingContig (+ in.fa / ex.fa, AG:24 TEST).
// misassembly removal (AG:3821-4297, row f2).  stdout carries the reference's own progress lines.  --fastMap runs NUCMER in place of
// BLAT and converts its .delta files with delta2psl (AG:524-729).  `ps euf >> mem.txt` (AG:4778) is not run.
//
// Everything here is text plumbing around the path; the compute is agx_run_unit's (include/agx.h).
#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sched.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <ctime>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>

#include "../../include/agx.h"

using std::cout; using std::endl; using std::string; using std::vector;

namespace {

struct Options {
    string read1, read2, contig, genome, ext, rmn;
    int tagRead1 = 0, tagRead2 = 0, tagContig = 0, tagGenome = 0, tagExt = 0, tagRmn = 0, tagK = 0, tagLow = 0, tagHigh = 0, tagIV = 0, tagCov = 0, tagPart = 0;
    int fastMap = 0, ratioCheck = 0, uniqueExtension = 0, iterativeMap = 0, misassemblyRemoval = 0, resume = 0;
    int k = 5, distanceLow = 0, distanceHigh = 99999, coverage = 20, insertVariation = 50, part = 1;      // defaults of AG:4701
};

void usage() {   // AG:4304-4327, verbatim program output (user-visible contract)
    cout << "AlignGraph --read1 reads_1.fa --read2 reads_2.fa --contig contigs.fa --genome genome.fa --distanceLow distanceLow --distanceHigh distancehigh --extendedContig extendedContigs.fa --remainingContig remainingContigs.fa [--kMer k --insertVariation insertVariation --covereage coverage --part p --ratioCheck --iterativeMap --misassemblyRemoval --resume]" << endl;
    cout << "Inputs:" << endl;
    cout << "--read1 is the the first pair of PE DNA reads in fasta format" << endl;
    cout << "--read2 is the the second pair of PE DNA reads in fasta format" << endl;
    cout << "--contig is the initial contigs in fasta format" << endl;
    cout << "--genome is the reference genome in fasta format" << endl;
    cout << "--distanceLow is the lower bound of alignment distance between the first and second pairs of PE DNA reads (recommended: max{insert length - 1000, single read length})" << endl;
    cout << "--distanceHigh is the upper bound of alignment distance between the first and second pairs of PE DNA reads (recommended: insert length + 1000)" << endl;
    cout << "Outputs:" << endl;
    cout << "--extendedContig is the extended contig file in fasta format" << endl;
    cout << "--remainingContig is the not extended initial contig file in fasta format" << endl;
    cout << "Options:" << endl;
    cout << "--kMer is the k-mer size (default: 5)" << endl;
    cout << "--insertVariation is the small variation of insert length (default: 50)" << endl;
    cout << "--coverage is the minimum coverage to keep a path in de Bruijn graph (default: 20)" << endl;
    cout << "--part is the number of parts a chromosome is divided into when it is loaded to reduce memory requirement (default: 1)" << endl;
    cout << "--fastMap calls NUCMER to make fast but less sensitive and accurate contig alignment instead of BLAT (default: none)" << endl;
    cout << "--ratioCheck checks read alignment ratio to the reference beforehand and warns if the ratio is too low; may take a little more time (default: none)" << endl;
    cout << "--iterativeMap aligns reads to one chromosome and then another rather than directly to the genome, which increases sensitivity while loses precision (default: none)" << endl;
    cout << "--misassemblyRemoval detects and then breaks at or removes misassembed regions (default: none)" << endl;
    cout << "--resume resumes the previous unfinished running from several checkpoints (default: none)" << endl;
}

[[noreturn]] void die_usage() { usage(); exit(-1); }
[[noreturn]] void die(const char *msg) { cout << msg << endl; exit(-1); }

// the reference reads every text file with getline + `if(buf[0]==0) break`
vector<string> read_lines(const string &path, bool &ok) {
    vector<string> v; std::ifstream in(path.c_str());
    ok = in.is_open();
    if (!ok) return v;
    string buf;
    while (in.good()) { std::getline(in, buf); if (buf.empty() || buf[0] == 0) break; v.push_back(buf); }
    return v;
}

string itoa(long long n) { std::stringstream ss; ss << n; return ss.str(); }

bool can_read(const string &p) { std::ifstream f(p.c_str()); return f.is_open(); }

// getParameters, AG:4329-4646: one token per line; a flag may appear once; integers must round-trip through atoi
void parse_params(const string &file, Options &o) {
    bool ok; vector<string> t = read_lines(file, ok);
    if (!ok) die("CANNOT OPEN FILE!");
    const int count = (int)t.size();
    auto value = [&](int &i, int &tag) -> string { if (tag == 1 || i == count - 1) die_usage(); tag = 1; return t[++i]; };
    auto integer = [&](int &i, int &tag, int &dst) { const string v = value(i, tag); dst = atoi(v.c_str()); if (itoa(dst) != v) die_usage(); };
    auto infile = [&](int &i, int &tag, string &dst) { dst = value(i, tag); if (!can_read(dst)) { cout << "CANNOT OPEN FILE!" << endl; die_usage(); } };
    auto outfile = [&](int &i, int &tag, string &dst) { dst = value(i, tag); std::ofstream f(dst.c_str()); if (!f.is_open()) { cout << "CANNOT OPEN FILE!" << endl; die_usage(); } };
    auto flag = [&](int &tag) { if (tag == 1) die_usage(); tag = 1; };
    for (int i = 0; i < count; i++) {
        const string &b = t[i];
        if (b == "--read1") infile(i, o.tagRead1, o.read1);
        else if (b == "--read2") infile(i, o.tagRead2, o.read2);
        else if (b == "--contig") infile(i, o.tagContig, o.contig);
        else if (b == "--genome") infile(i, o.tagGenome, o.genome);
        else if (b == "--distanceLow") integer(i, o.tagLow, o.distanceLow);
        else if (b == "--distanceHigh") integer(i, o.tagHigh, o.distanceHigh);
        else if (b == "--extendedContig") outfile(i, o.tagExt, o.ext);          // opening truncates the file right away, like ofstream::open (AG:4477)
        else if (b == "--remainingContig") outfile(i, o.tagRmn, o.rmn);
        else if (b == "--kMer") integer(i, o.tagK, o.k);
        else if (b == "--insertVariation") integer(i, o.tagIV, o.insertVariation);
        else if (b == "--coverage") integer(i, o.tagCov, o.coverage);
        else if (b == "--part") integer(i, o.tagPart, o.part);
        else if (b == "--fastMap") flag(o.fastMap);
        else if (b == "--ratioCheck") flag(o.ratioCheck);
        else if (b == "--uniqueExtension") flag(o.uniqueExtension);
        else if (b == "--iterativeMap") flag(o.iterativeMap);
        else if (b == "--misassemblyRemoval") flag(o.misassemblyRemoval);
        else if (b == "--resume") { if (o.resume == 1 || count != 1) die_usage(); o.resume = 1; }
        else die_usage();
    }
}

// A text file mapped read-only and read the way the reference's `getline` loops read theirs: lines without their '\n'; past the last byte a
// read gives an empty line and the stream stops being good; a last line without '\n' is delivered and ends the stream too.  The multi-gigabyte
// inputs (reads, SAM) go through this at memchr speed instead of through ifstream::getline + a vector of strings.
struct Lines {
    const char *p = nullptr; size_t n = 0, pos = 0; bool good = false, opened = false; int fd = -1;
    explicit Lines(const string &path) {
        fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        struct stat sb;
        if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { close(fd); fd = -1; return; }
        n = (size_t)sb.st_size;
        if (n) { void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0); if (m == MAP_FAILED) { close(fd); fd = -1; n = 0; return; } p = (const char *)m; madvise(m, n, MADV_SEQUENTIAL); }
        opened = good = true;
    }
    ~Lines() { if (p) munmap((void *)p, n); if (fd >= 0) close(fd); }
    Lines(const Lines &) = delete; Lines &operator=(const Lines &) = delete;
    void next(const char *&s, size_t &len) {
        if (pos >= n) { s = p + n; len = 0; good = false; return; }
        const char *nl = (const char *)memchr(p + pos, '\n', n - pos);
        s = p + pos;
        if (nl) { len = (size_t)(nl - s); pos += len + 1; } else { len = n - pos; pos = n; good = false; }
    }
};
// Text gathered for one output file and written in 16 MB pieces; a failed write is fatal.
struct Sink {
    int fd; string buf, name;
    explicit Sink(const string &path) : fd(open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666)), name(path) { buf.reserve((size_t)17 << 20); }
    ~Sink() { flush(); if (fd >= 0) close(fd); }
    Sink(const Sink &) = delete; Sink &operator=(const Sink &) = delete;
    bool is_open() const { return fd >= 0; }
    void put(const char *s, size_t len) { buf.append(s, len); if (buf.size() >= ((size_t)16 << 20)) flush(); }
    void put(char c) { buf.push_back(c); }
    void flush() {
        if (fd < 0) { buf.clear(); return; }
        for (size_t done = 0; done < buf.size();) { const ssize_t w = write(fd, buf.data() + done, buf.size() - done); if (w <= 0) { cout << "CANNOT WRITE FILE! (" << name << ")" << endl; exit(-1); } done += (size_t)w; }
        buf.clear();
    }
};

// threads for the front end's passes over the read files (AGX_CLI_THREADS; else the CPUs this process may run on, sixteen at most)
unsigned cli_threads() {
    if (const char *e = getenv("AGX_CLI_THREADS")) { const int v = atoi(e); if (v >= 1) return (unsigned)std::min(v, 64); }
    cpu_set_t set; CPU_ZERO(&set);
    unsigned n = sched_getaffinity(0, sizeof set, &set) == 0 ? (unsigned)CPU_COUNT(&set) : std::thread::hardware_concurrency();
    return std::max(1u, std::min(n, 16u));
}
template <class F> void on_threads(unsigned T, F fn) {
    vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back([&fn, t] { fn(t); });
    fn(0);
    for (auto &x : th) x.join();
}
// [lo, hi) of piece t of T over n bytes, both ends moved forward to the next line start
inline size_t line_start_at(const char *p, size_t n, size_t at) {
    if (at == 0 || at >= n) return std::min(at, n);
    const char *nl = (const char *)memchr(p + at - 1, '\n', n - (at - 1));
    return nl ? (size_t)(nl - p) + 1 : n;
}
// where a `getline` loop that stops at the first empty line (or a line that begins with a NUL byte) stops reading: the offset of that line, n if there is none
size_t scan_end(const char *p, size_t n, unsigned T) {
    vector<size_t> first(T, n);
    on_threads(T, [&](unsigned t) {
        const size_t lo = line_start_at(p, n, n / T * t), hi = t + 1 == T ? n : line_start_at(p, n, n / T * (t + 1));
        for (size_t at = lo; at < hi;) {
            if (p[at] == '\n' || p[at] == 0) { first[t] = at; return; }
            const char *nl = (const char *)memchr(p + at, '\n', hi - at);
            if (!nl) return;
            at = (size_t)(nl - p) + 1;
        }
    });
    size_t e = n; for (size_t v : first) e = std::min(e, v);
    return e;
}

// maxReadLength, AG:3197-3226 — the longest record of a read file (the sum of its sequence lines), the scan ending at the first empty line.  r05: on all the CPUs the
// process may use (the file cut into pieces at line starts; a piece reports the sequence bytes in front of its first header, its longest whole record and the bytes behind
// its last header; the pieces are joined in order) — a 20 M-pair run's two read files are 4.4 GB, and the option check reads both before anything else happens.
int max_read_length(const string &path) {
    Lines in(path);
    if (!in.opened) die("CANNOT OPEN FILE!");
    const unsigned T = in.n > ((size_t)8 << 20) ? cli_threads() : 1u;
    const char *p = in.p; const size_t n = scan_end(p, in.n, T);
    struct Piece { long long head = 0, inside = 0, tail = 0; bool header = false; };
    vector<Piece> pc(T);
    on_threads(T, [&](unsigned t) {
        const size_t lo = line_start_at(p, n, n / T * t), hi = t + 1 == T ? n : line_start_at(p, n, n / T * (t + 1));
        Piece &P = pc[t]; long long len = 0;
        for (size_t at = lo; at < hi;) {
            const char *nl = (const char *)memchr(p + at, '\n', n - at);
            const size_t le = nl ? (size_t)(nl - p) : n, l = le - at;
            if (p[at] == '>') { if (!P.header) { P.head = len; P.header = true; } else P.inside = std::max(P.inside, len); len = 0; }
            else len += (long long)l;
            at = nl ? le + 1 : n;
        }
        if (P.header) P.tail = len; else P.head = len;
    });
    long long mx = 0, len = 0;
    for (const Piece &P : pc) {
        if (!P.header) { len += P.head; continue; }
        mx = std::max(mx, std::max(len + P.head, P.inside)); len = P.tail;
    }
    return (int)std::max(mx, len);
}

void put60(std::ostream &o, const string &s) {          // 60 columns, newline after the last base; nothing at all for an empty sequence
    for (size_t i = 0; i < s.size(); i += 60) o << s.substr(i, 60) << '\n';
}

struct Fasta { vector<string> id, seq; };
Fasta read_fasta(const string &path) {
    bool ok; vector<string> t = read_lines(path, ok);
    if (!ok) die("CANNOT OPEN FILE!");
    Fasta f;
    for (const string &b : t) { if (b[0] == '>') { f.id.push_back(b.substr(1)); f.seq.push_back(string()); } else if (!f.seq.empty()) f.seq.back() += b; }
    return f;
}

// formalizeInput (contigs), AG:3228-3319: contigs > 200 bp become ">seqID.realID" records (>= 1 Mb: 1 Mb chunks sharing realID), the rest is chaff
void formalize_contigs(const string &path, vector<string> &contigIds, const string &dest = "tmp/_contigs.fa") {
    const Fasta f = read_fasta(path);
    std::ofstream out(dest.c_str()), chaff;
    if (dest == "tmp/_contigs.fa") chaff.open("tmp/_chaff.fa");            // only the initial contigs keep their chaff (AG:3242); elsewhere short records vanish
    const size_t CHUNK = 1000000;
    unsigned long seqID = 0, realID = 0;
    contigIds.clear();
    for (size_t c = 0; c < f.seq.size(); c++) {
        const string &s = f.seq[c];
        if (s.size() > 200) {
            if (s.size() < CHUNK) { out << ">" << seqID++ << "." << realID << '\n'; put60(out, s); }
            else {
                // AG:3279-3292: a new header after every full chunk unless fewer than 61 bases remain; columns restart in each chunk
                out << ">" << seqID++ << "." << realID << '\n';
                size_t total = 0;
                for (size_t p = 0; p < s.size(); p++) {
                    out << s[p];
                    if ((p + 1) % CHUNK == 0 && p + 61 < s.size()) { total += CHUNK; out << '\n' << ">" << seqID++ << "." << realID << '\n'; continue; }
                    if ((p + 1 - total) % 60 == 0 || p == s.size() - 1) out << '\n';
                }
            }
            realID++;
            contigIds.push_back(f.id[c]);
        } else { chaff << ">" << f.id[c] << '\n'; put60(chaff, s); }
    }
}

// formalizeGenome, AG:3347-3418: one unit per (chromosome, part); part boundaries at multiples of size/p
int formalize_genome(const string &path, int p, vector<string> &genomeIds) {
    const Fasta f = read_fasta(path);
    for (const string &id : f.id) genomeIds.push_back(id);                       // NOT cleared: a --resume run appends again, like the reference (AG:3367)
    // r05: whole lines into Sinks that write 16 MB at a time (r04 put every base through two ofstreams: 1.9 s for the 119 Mb of configs[2]).  The reference's rule, base by base:
    // a newline after every 60th base OF THE CHROMOSOME (the column count does not restart in a new part), after the chromosome's last base and after a part's last base;
    // a part ends after every `step`-th base while fewer than p parts are open, and a cut on the chromosome's last base opens no new unit.
    Sink all("tmp/_genome.fa");
    int unit = 0;
    for (size_t g = 0; g < f.seq.size(); g++) {
        const string &s = f.seq[g];
        const size_t size = s.size(), step = size / (size_t)p;
        int q = 1; size_t c = 0;
        std::unique_ptr<Sink> out(new Sink("tmp/_genome." + itoa(unit) + ".fa"));
        { const string head = ">" + itoa(unit) + "\n"; out->put(">0\n", 3); all.put(head.data(), head.size()); }
        for (;;) {
            size_t end = size;                                                   // this part: bases [c, end)
            if (step != 0 && q < p) { const size_t e = (c / step + 1) * step; if (e <= size) end = e; }
            while (c < end) {
                const size_t stop = std::min(end, (c / 60 + 1) * 60);
                out->put(s.data() + c, stop - c); all.put(s.data() + c, stop - c);
                c = stop;
                if (c % 60 == 0 || c == size || c == end) { out->put('\n'); all.put('\n'); }
            }
            if (c >= size) break;
            unit++; q++;
            out.reset(new Sink("tmp/_genome." + itoa(unit) + ".fa"));
            { const string head = ">" + itoa(unit) + "\n"; out->put(">0\n", 3); all.put(head.data(), head.size()); }
        }
        out.reset(); unit++;
    }
    return unit;
}

// An output file of known size written in place by several threads: created, sized and mapped shared
struct MappedOut {
    int fd = -1; char *p = nullptr; size_t n = 0; string name;
    MappedOut(const string &path, size_t bytes) : n(bytes), name(path) {
        fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
        // (the blocks are reserved before the file is mapped: on a full disk the run ends here with the reference's message, not with a SIGBUS in the middle of a store into a
        // sparse file — ADVICE r05; file systems without fallocate (EOPNOTSUPP / EINVAL) keep the sparse file)
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { cout << "CANNOT WRITE FILE! (" << name << ")" << endl; exit(-1); }
        if (bytes) { const int rc = posix_fallocate(fd, 0, (off_t)bytes); if (rc != 0 && rc != EOPNOTSUPP && rc != EINVAL) { cout << "CANNOT WRITE FILE! (" << name << ")" << endl; exit(-1); } }
        if (bytes) { void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (m == MAP_FAILED) { cout << "CANNOT WRITE FILE! (" << name << ")" << endl; exit(-1); } p = (char *)m; }
    }
    // (written back and checked before anybody reads the file: an I/O error that only shows at writeback — NFS and the like — must not leave zero-filled alignments for the unit loop)
    // (AGX_CLI_SYNC=1: synchronously — for file systems that report a full disk or an I/O error only at writeback; the default asks for the writeback and checks that the request was taken: forcing 9 GB of a cfg3 run to the disk before the aligners start would cost the front end seconds)
    ~MappedOut() { bool ok = true; if (p) { ok = msync(p, n, getenv("AGX_CLI_SYNC") ? MS_SYNC : MS_ASYNC) == 0; munmap(p, n); } if (fd >= 0) ok = (close(fd) == 0) && ok; if (!ok) { cout << "CANNOT WRITE FILE! (" << name << ")" << endl; exit(-1); } }
};
inline unsigned dec_digits(unsigned long v) { unsigned d = 1; while (v >= 10) { v /= 10; d++; } return d; }
inline char *put_dec(char *w, unsigned long v) { char t[24]; int k = 0; do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v); while (k) *w++ = t[--k]; return w; }

// formalizeInput (reads), AG:3420-3518, for the inputs every aligner's users have — both files a header line and ONE sequence line per record, the same number of records, no
// empty line — on all the CPUs the process may use: the files are cut into pieces at record starts, the pieces' records counted, a first pass sizes every thread's share of the
// three outputs (a record's bytes follow from its id's digits and the shorter mate), a second one writes them in place.  Returns -1 (nothing written) for anything else:
// multi-line records, a header without a sequence, files of different shape — the line-by-line form below then does what the reference does with them, message for message.
long long formalize_reads_fast(const Lines &in1, const Lines &in2) {
    const unsigned T = cli_threads();
    const size_t least = getenv("AGX_CLI_FAST_MIN") ? (size_t)atoll(getenv("AGX_CLI_FAST_MIN")) : ((size_t)4 << 20);      // (tests: small files through the threaded form)
    if (T < 2 || in1.n < least || in2.n < least || in1.n < 64 * (size_t)T || in2.n < 64 * (size_t)T) return -1;
    const Lines *in[2] = {&in1, &in2};
    size_t end[2]; for (int f = 0; f < 2; f++) { end[f] = scan_end(in[f]->p, in[f]->n, T); if (end[f] != in[f]->n) return -1; }      // (an empty line ends the reference's loop early: the general form)
    // pieces at record starts; every piece: two-line records only?  how many?
    vector<size_t> lo[2], cnt[2]; bool ok[2] = {true, true};
    for (int f = 0; f < 2; f++) {
        const char *p = in[f]->p; const size_t n = in[f]->n;
        if (n == 0 || p[0] != '>' || p[n - 1] != '\n') return -1;
        lo[f].assign(T + 1, n); cnt[f].assign(T, 0);
        for (unsigned t = 0; t < T; t++) {          // the piece starts at the first header line at or behind its share's first byte
            size_t at = line_start_at(p, n, n / T * t);
            if (at < n && p[at] != '>') at = line_start_at(p, n, at + 1);
            lo[f][t] = at;
        }
        lo[f][0] = 0;
        vector<int> bad(T, 0);
        on_threads(T, [&](unsigned t) {
            const size_t a = lo[f][t], b = lo[f][t + 1]; size_t c = 0;
            for (size_t at = a; at < b;) {
                if (p[at] != '>') { bad[t] = 1; return; }
                const char *h = (const char *)memchr(p + at, '\n', n - at);
                if (!h || (size_t)(h - p) + 1 >= n) { bad[t] = 1; return; }
                const size_t sq = (size_t)(h - p) + 1;
                if (p[sq] == '>' || p[sq] == '\n') { bad[t] = 1; return; }
                const char *e = (const char *)memchr(p + sq, '\n', n - sq);
                if (!e) { bad[t] = 1; return; }
                at = (size_t)(e - p) + 1; c++;
            }
            cnt[f][t] = c;
        });
        for (unsigned t = 0; t < T; t++) if (bad[t] || lo[f][t] > lo[f][t + 1]) ok[f] = false;
    }
    if (!ok[0] || !ok[1]) return -1;
    size_t total[2] = {0, 0}; for (int f = 0; f < 2; f++) for (size_t c : cnt[f]) total[f] += c;
    if (total[0] != total[1]) return -1;                // (the line-by-line form reports what the reference reports)
    const size_t N = total[0];
    // shares of records: file 1's pieces; where the same records begin in file 2
    vector<size_t> rec0(T + 1, 0); for (unsigned t = 0; t < T; t++) rec0[t + 1] = rec0[t] + cnt[0][t];
    vector<size_t> off2(T + 1, in2.n);
    {   vector<size_t> r2(T + 1, 0); for (unsigned t = 0; t < T; t++) r2[t + 1] = r2[t] + cnt[1][t];
        on_threads(T, [&](unsigned t) {
            const size_t want = rec0[t];
            if (want >= N) { off2[t] = in2.n; return; }
            unsigned q = 0; while (q + 1 < T && r2[q + 1] <= want) q++;          // the piece of file 2 that holds record `want`
            size_t at = lo[1][q];
            for (size_t k = r2[q]; k < want; k++) { at = (size_t)((const char *)memchr(in2.p + at, '\n', in2.n - at) - in2.p) + 1; at = (size_t)((const char *)memchr(in2.p + at, '\n', in2.n - at) - in2.p) + 1; }
            off2[t] = at;
        });
    }
    // a record's two lines from its header's offset: the sequence and where the next record begins
    auto record = [](const Lines &L, size_t at, const char *&seq, size_t &len) { const char *h = (const char *)memchr(L.p + at, '\n', L.n - at); seq = h + 1; const char *e = (const char *)memchr(seq, '\n', L.n - (size_t)(seq - L.p)); len = (size_t)(e - seq); return (size_t)(e - L.p) + 1; };
    vector<size_t> bytes_all(T + 1, 0), bytes_one(T + 1, 0);
    on_threads(T, [&](unsigned t) {
        size_t a1 = lo[0][t], a2 = off2[t], all = 0, one = 0;
        for (size_t id = rec0[t]; id < rec0[t + 1]; id++) {
            const char *s1, *s2; size_t l1, l2; a1 = record(in1, a1, s1, l1); a2 = record(in2, a2, s2, l2);
            const size_t m = std::min(l1, l2), rec = 1 + dec_digits(id) + 1 + m + 1;
            one += rec; all += 2 * rec;
        }
        bytes_all[t + 1] = all; bytes_one[t + 1] = one;
    });
    for (unsigned t = 0; t < T; t++) { bytes_all[t + 1] += bytes_all[t]; bytes_one[t + 1] += bytes_one[t]; }
    MappedOut out("tmp/_reads.fa", bytes_all[T]), out1("tmp/_reads_1.fa", bytes_one[T]), out2("tmp/_reads_2.fa", bytes_one[T]);
    on_threads(T, [&](unsigned t) {
        size_t a1 = lo[0][t], a2 = off2[t]; char *w = out.p + bytes_all[t], *w1 = out1.p + bytes_one[t], *w2 = out2.p + bytes_one[t];
        for (size_t id = rec0[t]; id < rec0[t + 1]; id++) {
            const char *s1, *s2; size_t l1, l2; a1 = record(in1, a1, s1, l1); a2 = record(in2, a2, s2, l2);
            const size_t m = std::min(l1, l2);
            char head[32]; char *h = head; *h++ = '>'; h = put_dec(h, (unsigned long)id); *h++ = '\n'; const size_t hl = (size_t)(h - head);
            memcpy(w, head, hl); w += hl; memcpy(w, s1, m); w += m; *w++ = '\n'; memcpy(w, head, hl); w += hl; memcpy(w, s2, m); w += m; *w++ = '\n';
            memcpy(w1, head, hl); w1 += hl; memcpy(w1, s1, m); w1 += m; *w1++ = '\n';
            memcpy(w2, head, hl); w2 += hl; memcpy(w2, s2, m); w2 += m; *w2++ = '\n';
        }
    });
    if (getenv("AGX_CLI_TIMING")) fprintf(stderr, "[agx cli] %zu pairs of reads on %u threads: %zu + %zu bytes in, %zu + 2 x %zu out\n", N, T, in1.n, in2.n, bytes_all[T], bytes_one[T]);
    return (long long)N;
}

// formalizeInput (reads), AG:3420-3518: pairs renamed 0..N-1, mates cut to the shorter of the two
int formalize_reads(const string &p1, const string &p2) {
    Lines in1(p1), in2(p2);
    if (!in1.opened || !in2.opened) die("CANNOT OPEN FILE!");
    if (!getenv("AGX_CLI_SERIAL")) { const long long fast = formalize_reads_fast(in1, in2); if (fast >= 0) return (int)fast; }
    Sink out("tmp/_reads.fa"), out1("tmp/_reads_1.fa"), out2("tmp/_reads_2.fa");
    string r1, r2; unsigned long id = 0; char head[32];
    auto flush = [&]() {
        if (r1.empty() || r2.empty()) return;
        const size_t n = std::min(r1.size(), r2.size());
        const size_t h = (size_t)snprintf(head, sizeof head, ">%lu\n", id);
        out.put(head, h); out.put(r1.data(), n); out.put('\n'); out.put(head, h); out.put(r2.data(), n); out.put('\n');
        out1.put(head, h); out1.put(r1.data(), n); out1.put('\n');
        out2.put(head, h); out2.put(r2.data(), n); out2.put('\n');
        id++;
    };
    while (in1.good && in2.good) {
        const char *b1, *b2; size_t l1, l2;
        in1.next(b1, l1); in2.next(b2, l2);
        const bool e1 = l1 == 0 || b1[0] == 0, e2 = l2 == 0 || b2[0] == 0;
        if (e1 && e2) break;
        if (e1 != e2) die("INCONSISTENT PE FILES!");
        if (b1[0] == '>' && b2[0] == '>') { flush(); r1.clear(); r2.clear(); }
        else if (b1[0] != '>' && b2[0] != '>') { r1.append(b1, l1); r2.append(b2, l2); }
        else die("INCONSISTENT PE FILES!");
    }
    flush();
    return (int)id;
}

// distributeAlignments, AG:3545-3579 with parseBT, AG:3520-3543: a SAM line goes to the unit its RNAME (atoi of <=9 chars) names; '@' lines
// are dropped; an empty line ends the scan
// the unit of one line (not empty, not beginning with a NUL byte): -1 = none ('@' lines, a '*' anywhere in RNAME, a unit that does not exist)
inline int unit_of_line(const char *line, size_t len, int units) {
    if (line[0] == '@') return -1;
    const char *le = line + len;
    // parseBT: third tab-separated field; any '*' in it means unaligned; a missing field reads as "" -> unit 0 (atoi)
    const char *t1 = (const char *)memchr(line, '\t', len), *t2 = t1 ? (const char *)memchr(t1 + 1, '\t', (size_t)(le - t1 - 1)) : nullptr;
    const char *r0 = t2 ? t2 + 1 : le, *r1 = t2 ? (const char *)memchr(r0, '\t', (size_t)(le - r0)) : le;
    if (!r1) r1 = le;
    if (memchr(r0, '*', (size_t)(r1 - r0))) return -1;
    char num[10]; const size_t k = std::min<size_t>(9, (size_t)(r1 - r0)); memcpy(num, r0, k); num[k] = 0;
    const int u = atoi(num);
    return u >= 0 && u < units ? u : -1;
}
// r05: on all the CPUs the process may use — the mapped SAM cut into pieces at line starts, a first pass adds up every piece's bytes per unit, the units' files are made at their
// final sizes and mapped, a second pass copies every line to its place (a 20 M-pair run distributes 11 GB).  Not for thousands of units (a mapping per unit): those take the form below.
bool distribute_alignments_threaded(const char *base, size_t n_all, int units) {
    const unsigned T = cli_threads();
    const size_t least = getenv("AGX_CLI_FAST_MIN") ? (size_t)atoll(getenv("AGX_CLI_FAST_MIN")) : ((size_t)16 << 20);      // (tests: small files through this form)
    if (T < 2 || getenv("AGX_CLI_SERIAL") || n_all < least || n_all < 64 * (size_t)T || units > 4096) return false;
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now_s();
    const size_t n = scan_end(base, n_all, T);
    vector<size_t> cut(T + 1);
    for (unsigned t = 0; t <= T; t++) cut[t] = t == T ? n : line_start_at(base, n, n / T * t);
    vector<vector<size_t> > bytes(T, vector<size_t>((size_t)units, 0));
    auto lines_of = [&](unsigned t, auto &&fn) {
        for (size_t at = cut[t]; at < cut[t + 1];) {
            const char *nl = (const char *)memchr(base + at, '\n', n - at);
            const size_t len = nl ? (size_t)(nl - base) - at : n - at;
            const int u = unit_of_line(base + at, len, units);
            if (u >= 0) fn(u, base + at, len);
            at += len + 1;
        }
    };
    on_threads(T, [&](unsigned t) { vector<size_t> &b = bytes[t]; lines_of(t, [&](int u, const char *, size_t len) { b[(size_t)u] += len + 1; }); });
    const double t1 = now_s();
    // (written in place through shared mappings, like the read files: pwrite from gathered buffers — with or without the blocks allocated up front — was anything between twice as
    // fast and five times slower on the CPU container, depending on how many dirty pages the aligner had left behind; the mappings took 7 s for 6.9 GB every time)
    vector<std::unique_ptr<MappedOut> > out((size_t)units);
    for (int u = 0; u < units; u++) {
        size_t total = 0; for (unsigned t = 0; t < T; t++) { const size_t mine = bytes[t][(size_t)u]; bytes[t][(size_t)u] = total; total += mine; }      // bytes[t][u]: where piece t's lines of unit u begin
        out[(size_t)u].reset(new MappedOut("tmp/_reads_genome." + itoa(u) + ".bowtie", total));
        close(out[(size_t)u]->fd); out[(size_t)u]->fd = -1;                                                                                               // (the mapping stays; no descriptor per unit)
    }
    on_threads(T, [&](unsigned t) {
        vector<size_t> &at = bytes[t];
        lines_of(t, [&](int u, const char *line, size_t len) { char *w = out[(size_t)u]->p + at[(size_t)u]; memcpy(w, line, len); w[len] = '\n'; at[(size_t)u] += len + 1; });
    });
    for (auto &o : out) if (o->p && msync(o->p, o->n, MS_ASYNC) != 0) { cout << "CANNOT WRITE FILE! (" << o->name << ")" << endl; exit(-1); }
    if (getenv("AGX_CLI_TIMING")) fprintf(stderr, "[agx cli]   %zu bytes of SAM to %d units on %u threads: sizing pass %.3f s, copying pass %.3f s\n", n, units, T, t1 - t0, now_s() - t1);
    return true;
}
void distribute_alignments(int units) {
    // The whole SAM mapped and walked line by line with memchr, every unit's lines gathered in a buffer that goes out in 8 MB writes: the reference's
    // getline + operator<< loop (AG:3545-3579) moved ~0.2 GB/s, and a 20 M-pair run has 2.4 GB to distribute.  Same rules, same bytes.
    const int fd = open("tmp/_reads_genome.bowtie", O_RDONLY);
    if (fd < 0) die("CANNOT OPEN FILE!");
    struct stat sb; if (fstat(fd, &sb) != 0) { close(fd); die("CANNOT OPEN FILE!"); }
    const size_t n = (size_t)sb.st_size;
    const char *base = n ? (const char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    if (n && base == (const char *)MAP_FAILED) { close(fd); die("CANNOT OPEN FILE!"); }
    if (n) madvise((void *)base, n, MADV_SEQUENTIAL);
    if (n && distribute_alignments_threaded(base, n, units)) { munmap((void *)base, n); close(fd); return; }
    // Every unit's file is created and truncated up front (as the reference's ofstreams are), but only `open_max` descriptors are held at a time and the buffers
    // share one budget: a draft reference with thousands of scaffolds must neither run into the descriptor limit nor hold 8 MB per unit.  A file that cannot be
    // opened or written is fatal — silently dropped lines would make the unit run on truncated alignments and report success.
    vector<int> out(units, -1); vector<string> buf(units);
    const size_t per_unit = std::max<size_t>((size_t)64 << 10, std::min<size_t>((size_t)8 << 20, ((size_t)1 << 30) / (size_t)std::max(1, units)));
    const int open_max = 512; int n_open = 0; vector<int> lru;
    for (int u = 0; u < units; u++) { const int f = open(("tmp/_reads_genome." + itoa(u) + ".bowtie").c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666); if (f < 0) die("CANNOT OPEN FILE!"); close(f); }
    auto fd_of = [&](int u) -> int {
        if (out[u] >= 0) return out[u];
        if (n_open >= open_max) { const int v = lru.front(); lru.erase(lru.begin()); close(out[v]); out[v] = -1; n_open--; }
        out[u] = open(("tmp/_reads_genome." + itoa(u) + ".bowtie").c_str(), O_WRONLY | O_APPEND, 0666);
        if (out[u] < 0) die("CANNOT OPEN FILE!");
        n_open++; lru.push_back(u);
        return out[u];
    };
    auto flush = [&](int u) {
        if (buf[u].empty()) return;
        const int f = fd_of(u); size_t done = 0;
        while (done < buf[u].size()) { const ssize_t w = write(f, buf[u].data() + done, buf[u].size() - done); if (w <= 0) die(("CANNOT WRITE FILE! (tmp/_reads_genome." + itoa(u) + ".bowtie)").c_str()); done += (size_t)w; }
        buf[u].clear();
    };
    for (const char *c = base, *e = base + n; c < e;) {
        const char *nl = (const char *)memchr(c, '\n', (size_t)(e - c));
        const char *le = nl ? nl : e;
        const size_t len = (size_t)(le - c);
        const char *line = c; c = nl ? nl + 1 : e;
        if (len == 0 || line[0] == 0) break;
        const int u = unit_of_line(line, len, units);
        if (u >= 0) { buf[u].append(line, len); buf[u].push_back('\n'); if (buf[u].size() >= per_unit) flush(u); }
    }
    for (int u = 0; u < units; u++) { flush(u); if (out[u] >= 0) close(out[u]); }
    if (n) munmap((void *)base, n);
    close(fd);
}

std::atomic<long long> g_run_ns{0};      // time inside external commands (AGX_CLI_TIMING reports it per stage; summed over the threads that wait)
int run(const string &cmd) {
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = system(cmd.c_str());
    g_run_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// parseDelta, AG:524-586: the first four blank-separated items of a .delta line as integers.  Item 0 drops every '>'; an item of
// the form "a.b" yields a and "real" part b — except that the reference fills the second item's real part starting at the index where
// the first item's real part ended (one shared counter, AG:573), so after a dotted first item a dotted second item reads as 0.
void parse_delta(const string &buf, int &a, int &b, int &c, int &d, int &realA, int &realB) {
    string item[4];
    int it = 0;
    for (size_t i = 0; i < buf.size(); i++) {
        if (buf[i] == ' ') { it++; continue; }
        if (buf[i] == '\0') break;
        if (it == 0 && buf[i] != '>') item[0].push_back(buf[i]);
        if (it >= 1 && it <= 3) item[it].push_back(buf[i]);
    }
    auto split = [](const string &x, string &head, string &real) -> bool {    // digits before the first '.', all non-'.' characters after it
        const size_t dot = x.find('.');
        head = x.substr(0, dot);
        real.clear();
        if (dot == string::npos) return false;
        for (size_t i = dot + 1; i < x.size(); i++) if (x[i] != '.') real.push_back(x[i]);
        return true;
    };
    string ha, ra, hb, rb;
    const bool aTag = split(item[0], ha, ra), bTag = split(item[1], hb, rb);
    a = atoi(ha.c_str()); realA = aTag ? atoi(ra.c_str()) : -1;
    b = atoi(hb.c_str()); realB = bTag ? ((aTag && !ra.empty()) ? 0 : atoi(rb.c_str())) : -1;
    c = atoi(item[2].c_str()); d = atoi(item[3].c_str());
}

// delta2psl, AG:588-729: NUCMER's .delta -> the 21-column PSL the loaders read ("NA" where the reference has nothing to say).
// The output name keeps everything before the second '.' of the input name (AG:601).
void delta2psl(const string &st0) {
    std::ifstream in(st0.c_str());
    const size_t d1 = st0.find('.', 0), d2 = d1 == string::npos ? string::npos : st0.find('.', d1 + 1);
    std::ofstream out((st0.substr(0, d2) + ".psl").c_str());
    if (!in.is_open()) die("CANNOT OPEN FILE!");
    struct Seg3 { unsigned sourceStart, targetStart, size; };
    string buf;
    int targetID = 0, sourceID = 0, targetSize = 0, sourceSize = 0, realTargetID = 0, realSourceID = 0, dummy = 0;
    getline(in, buf); getline(in, buf);
    while (in.good()) {
        string align; vector<Seg3> seg; int sourceGap = 0, targetGap = 0;
        int targetStart, targetEnd, sourceStart, sourceEnd;
        getline(in, buf);
        if (buf.empty() || buf[0] == '\0') break;
        if (buf[0] == '>') { parse_delta(buf, targetID, sourceID, targetSize, sourceSize, realTargetID, realSourceID); getline(in, buf); }
        parse_delta(buf, targetStart, targetEnd, sourceStart, sourceEnd, dummy, dummy);
        char fr = '+';
        if (!(sourceStart < sourceEnd)) { fr = '-'; std::swap(sourceStart, sourceEnd); }
        getline(in, buf);
        while (buf.empty() || buf[0] != '0') {                                // one signed distance per indel, "0" ends the alignment
            if (!in.good()) die("BROKEN DELTA FILE");                         // (the reference never leaves this loop on a truncated file)
            const int b = atoi(buf.c_str());
            for (int i = 1; i < abs(b); i++) align.push_back('M');
            if (b > 0) { align.push_back('I'); targetGap++; } else { align.push_back('D'); sourceGap++; }
            getline(in, buf);
        }
        for (int i = (int)align.size(); i < (sourceEnd - sourceStart + 1) + targetGap; i++) align.push_back('M');
        const int n = (int)align.size();
        for (int i = 0, j = 0; i < n; i++) {                                   // block starts on the query ...
            if (align[i] != 'I') j++;
            if ((i == 0 && align[i] == 'M') || (i >= 1 && (align[i - 1] == 'I' || align[i - 1] == 'D') && align[i] == 'M'))
                seg.push_back(Seg3{(unsigned)(sourceStart - 1 + j), (unsigned)-1, (unsigned)-1});
        }
        for (int i = 0, j = 0, sp = 0; i < n; i++) {                           // ... on the target ...
            if (align[i] != 'D') j++;
            if ((i == 0 && align[i] == 'M') || (i >= 1 && (align[i - 1] == 'I' || align[i - 1] == 'D') && align[i] == 'M'))
                seg[sp++].targetStart = (unsigned)(targetStart - 1 + j);
        }
        for (int i = 0, j = 0, sp = 0; i < n; i++) {                           // ... and their lengths
            if (align[i] == 'M') j++;
            if (i + 1 < n && align[i] == 'M' && (align[i + 1] == 'I' || align[i + 1] == 'D')) { seg[sp++].size = (unsigned)j; j = 0; }
            if (i == n - 1 && align[i] == 'M') seg[sp].size = (unsigned)j;
        }
        sourceStart--; targetStart--;
        for (Seg3 &g : seg) { g.sourceStart--; g.targetStart--; }
        out << "NA\tNA\tNA\tNA\tNA\t" << sourceGap << "\tNA\t" << targetGap << "\t" << fr << "\t";
        if (realSourceID != -1) out << sourceID << "." << realSourceID << "\t"; else out << sourceID << "\t";
        out << sourceSize << "\t" << sourceStart << "\t" << sourceEnd << "\t";
        if (realTargetID != -1) out << targetID << "." << realTargetID << "\t"; else out << targetID << "\t";
        out << targetSize << "\t" << targetStart << "\t" << targetEnd << "\tNA\t";
        for (const Seg3 &g : seg) out << g.size << ",";
        out << "\t";
        for (const Seg3 &g : seg) out << g.sourceStart << ",";
        out << "\t";
        for (const Seg3 &g : seg) out << g.targetStart << ",";
        out << endl;
    }
}

// NUCMER in place of BLAT (--fastMap): a failed call leaves an empty .delta behind (AG:3634-3641, 2962-2969, 3833-3840)
void nucmer_to_psl(const string &ref, const string &qry, const string &prefix, const string &delta) {
    if (run("nucmer " + ref + " " + qry + " -p " + prefix + " > nucmer_doc.txt 2> nucmer_doc.txt") != 0) run("touch " + delta);
    delta2psl(delta);
}

// task0 / task1 of parallelMap, AG:3581-3735: the two aligners run side by side, command strings unchanged
void align_everything(const Options &o, int units) {
    const bool timing = getenv("AGX_CLI_TIMING") != nullptr;      // (stderr: what the aligners' share of this stage was, and the distribution's and the caches')
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_in = now_s();
    std::thread reads([&]() {
        const string lo = itoa(o.distanceLow), hi = itoa(o.distanceHigh);
        const string opts = "bowtie2 -f --no-mixed -k 5 -p 8 --local --mp 3,1 --rdg 2,1 --rfg 2,1 --score-min G,5,2 -I " + lo + " -X " + hi + " --no-discordant -x ";
        if (o.iterativeMap == 1) {
            for (int u = 0; u < units; u++) {
                run("bowtie2-build -f tmp/_genome." + itoa(u) + ".fa tmp/_genome." + itoa(u) + " > bowtie_doc.txt 2> bowtie_doc.txt");
                run(opts + "tmp/_genome." + itoa(u) + " -1 tmp/_reads_1.fa -2 tmp/_reads_2.fa --reorder > tmp/_reads_genome." + itoa(u) + ".bowtie 2> bowtie_doc.txt");
            }
        } else {
            run("bowtie2-build -f tmp/_genome.fa tmp/_genome > bowtie_doc.txt 2> bowtie_doc.txt");
            run(opts + "tmp/_genome -1 tmp/_reads_1.fa -2 tmp/_reads_2.fa --reorder > tmp/_reads_genome.bowtie 2> bowtie_doc.txt");
            const double t_al = now_s();
            distribute_alignments(units);
            if (timing) fprintf(stderr, "[agx cli]   bowtie2 (external)          %9.3f s\n[agx cli]   distribute alignments       %9.3f s\n", t_al - t_in, now_s() - t_al);
        }
    });
    std::thread contigs([&]() {
        for (int u = 0; u < units; u++) {
            if (o.fastMap == 1) { nucmer_to_psl("tmp/_genome." + itoa(u) + ".fa", "tmp/_contigs.fa", "tmp/_contigs_genome." + itoa(u), "tmp/_contigs_genome." + itoa(u) + ".delta"); continue; }
            const string io = "tmp/_genome." + itoa(u) + ".fa tmp/_contigs.fa -noHead tmp/_contigs_genome." + itoa(u) + ".psl -fastMap";
            if (run("pblat " + io + " -threads=8 > blat_doc.txt 2> blat_doc.txt") != 0)
                if (run("blat " + io + " > blat_doc.txt 2> blat_doc.txt") != 0) die("BLAT CALL FAILED!");
        }
    });
    reads.join(); contigs.join();
    const double t_joined = now_s();
    // The units' binary caches: the five text files of every unit parsed ONCE, here where the reference distributes the
    // alignments (AG:3545-3579), into tmp/_agx_unit.<u>.bin; the unit loop — of this run and of every --resume — then reads no text.  The
    // text files stay what the contract says they are.  Failures are not errors here: the unit loop falls back to the text and reports.
    if (!getenv("AGX_NO_CACHE") && agx_device_count() > 0) {
        agx_reads *rd = nullptr; { char err[512]; if (agx_reads_open("tmp/_reads.fa", &rd, err, sizeof err) != AGX_OK) rd = nullptr; }
        std::atomic<int> nextu(0);
        vector<std::thread> th;
        for (int t = 0; t < std::min(units, 4); t++) th.emplace_back([&]() {
            for (;;) { const int u = nextu.fetch_add(1); if (u >= units) return;
                agx_params p = {(uint32_t)o.k, (uint32_t)o.insertVariation, (uint32_t)o.coverage, 0, 0, 0}; char err[512];
                (void)agx_unit_cache_build(&p, "tmp", u, rd, err, sizeof err); }
        });
        for (auto &t : th) t.join();
        agx_reads_close(rd);
    }
    if (timing) fprintf(stderr, "[agx cli]   unit caches                 %9.3f s\n", now_s() - t_joined);
}

// the SAM fields checkRatio looks at (parseBOWTIE, AG:181-285) and its identity filter (AG:3790)
bool sam_pair_passes(const string &l1, const string &l2, unsigned &id) {
    auto parse = [](const string &l, unsigned &qid, bool &aligned, unsigned &total, unsigned &ins, unsigned &del, unsigned &cl, unsigned &cr) {
        vector<string> f; size_t a = 0;
        for (int i = 0; i < 6; i++) { size_t b = l.find('\t', a); f.push_back(l.substr(a, b == string::npos ? string::npos : b - a)); if (b == string::npos) break; a = b + 1; }
        f.resize(6);
        qid = (unsigned)atoi(f[0].c_str()); aligned = !(f[2].size() && f[2][0] == '*');
        total = ins = del = cl = cr = 0; bool first = true; unsigned num = 0;
        for (char c : f[5]) {
            if (c >= '0' && c <= '9') { num = num * 10 + (unsigned)(c - '0'); continue; }
            if (c == 'I') { ins += num; total += num; } else if (c == 'D') del += num; else if (c == 'M') { total += num; first = false; }
            else if (c == 'S' && first) { cl = num; total += num; first = false; } else if (c == 'S') { cr = num; total += num; }
            else if (c != '*') { cout << "unknown character: " << c << endl; exit(-1); }
            num = 0;
        }
    };
    unsigned id2, t1, i1, d1, l1c, r1c, t2, i2, d2, l2c, r2c; bool a1, a2;
    parse(l1, id, a1, t1, i1, d1, l1c, r1c); parse(l2, id2, a2, t2, i2, d2, l2c, r2c);
    auto ok = [](unsigned total, unsigned ins, unsigned del, unsigned cl, unsigned cr) {
        const unsigned sEnd = total - cr, tLen = total + del - ins;
        return (double)(unsigned)(sEnd - cl - ins) / total >= 0.6 && (double)(unsigned)(tLen - del) / tLen >= 0.6;
    };
    return a1 && a2 && ok(t1, i1, d1, l1c, r1c) && ok(t2, i2, d2, l2c, r2c);
}

// checkRatio, AG:3751-3819.  The reference re-opens one ifstream without closing it, so only unit 0's alignments are ever read (AG:3779-3783).
void check_ratio(int units) {
    bool ok; vector<string> r = read_lines("tmp/_reads_1.fa", ok);
    if (!ok) die("CANNOT OPEN FILE!");
    // NB the reference counts '>' lines with `while(good) getline` WITHOUT the empty-line break, but a reads file has no empty lines
    size_t n = 0; for (const string &b : r) n += b[0] == '>';
    vector<int> reads(n, 0);
    if (units > 0) {
        vector<string> t = read_lines("tmp/_reads_genome.0.bowtie", ok);
        if (!ok) die("CANNOT OPEN FILE!");
        for (size_t i = 0; i < t.size(); i++) {
            if (t[i][0] == '@') continue;
            if (i + 1 >= t.size()) die("BROKEN BOWTIE FILE");
            unsigned id; const bool pass = sam_pair_passes(t[i], t[i + 1], id); i++;
            if (pass && id < reads.size()) reads[id] = 1;
        }
    }
    size_t aligned = 0; for (int v : reads) aligned += v == 1;
    const double ratio = aligned == 0 ? 0 : (double)aligned / reads.size();
    cout << " - " << ratio * 100 << "% reads aligned ";
    if (ratio < 0.25) cout << "(warning: ratio below 25%; hard to guarantee good results)" << endl; else cout << endl;
}

// ---- refinement, AG:2864-3195 -------------------------------------------------------------------------------------------------
struct PslHit { unsigned tID, tStart, tEnd, tGap, sID, sStart, sEnd, sGap, sSize, tSize, realSourceSize; };
PslHit parse_psl(const string &l) {   // parseBLAT, AG:406-522 (the columns refinement uses)
    vector<string> f; size_t a = 0;
    for (;;) { size_t b = l.find('\t', a); f.push_back(l.substr(a, b == string::npos ? string::npos : b - a)); if (b == string::npos) break; a = b + 1; }
    f.resize(21);
    auto U = [&](int i) { return (unsigned)atoi(f[i].c_str()); };
    PslHit h; h.tID = U(13); h.tStart = U(15); h.tEnd = U(16); h.tGap = U(7); h.sStart = U(11); h.sEnd = U(12); h.sGap = U(5); h.sSize = U(10); h.tSize = U(14);
    const string q = f[9].substr(0, 100);                                   // 100-byte field buffer of the reference
    const size_t dot = q.find('.');
    if (dot != string::npos) { h.sID = (unsigned)atoi(q.substr(0, dot).c_str()); h.realSourceSize = (unsigned)atoi(q.substr(dot + 1).c_str()); }
    else { h.sID = (unsigned)atoi(q.c_str()); h.realSourceSize = h.sSize; }
    return h;
}

void refinement(const Options &o, int units, const vector<string> &genomeIds, const vector<string> &contigIds) {
    std::ofstream ini("in.fa"), ext("ex.fa");                                // the reference is built with TEST defined (AG:24)
    const size_t SMALL = 20000;
    for (int u = 0; u < units; u++) {                                         // first 20 kb of every initial contig, AG:2891-2953
        const Fasta f = read_fasta("tmp/_initial_contigs." + itoa(u) + ".fa");
        std::ofstream out(("tmp/_short_initial_contigs." + itoa(u) + ".fa").c_str());
        for (size_t j = 0; j < f.seq.size(); j++) {
            const int num = atoi(f.id[j].c_str());
            if (f.seq[j].size() > SMALL) { out << ">" << num << "." << f.seq[j].size() << '\n'; put60(out, f.seq[j].substr(0, SMALL)); }
            else { out << ">" << num << '\n'; put60(out, f.seq[j]); }
        }
    }
    for (int u = 0; u < units; u++) {                                         // AG:2957-2982
        if (o.fastMap == 1) {
            const string pre = "tmp/_short_initial_contigs_extended_contigs." + itoa(u);
            nucmer_to_psl("tmp/_extended_contigs." + itoa(u) + ".fa", "tmp/_short_initial_contigs." + itoa(u) + ".fa", pre, pre + ".delta");
            continue;
        }
        const string io = "tmp/_extended_contigs." + itoa(u) + ".fa tmp/_short_initial_contigs." + itoa(u) + ".fa -noHead tmp/_short_initial_contigs_extended_contigs." + itoa(u) + ".psl -fastMap";
        if (run("pblat " + io + " -threads=8 > blat_doc.txt 2> blat_doc.txt") != 0)
            if (run("blat " + io + " > blat_doc.txt 2> blat_doc.txt") != 0) die("BLAT CALL FAILED!");
    }
    // initial contigs by realID (runs of equal id after the '.'), AG:2985-3014
    vector<string> init; vector<int> initTags;
    {
        bool ok; vector<string> t = read_lines("tmp/_contigs.fa", ok);
        if (!ok) { cout << "CANNOT OPEN FILE!" << endl; return; }
        int idBak = -1;
        for (const string &b : t) {
            if (b[0] == '>') { const size_t d = b.find('.'); const int id = atoi(d == string::npos ? "" : b.substr(d + 1).c_str()); if (id != idBak) { init.push_back(string()); initTags.push_back(0); idBak = id; } }
            else if (!init.empty()) init.back() += b;
        }
    }
    std::ofstream e(o.ext.c_str(), std::ios::app), r(o.rmn.c_str(), std::ios::app);   // both were truncated when the parameters were read
    vector<vector<int> > extdInitMap;                                         // grows by every unit's records and is never cleared (AG:3035): part of the output
    int seqID = 0;
    for (int u = 0; u < units; u++) {
        const Fasta x = read_fasta("tmp/_extended_contigs." + itoa(u) + ".fa");
        vector<int> extdTags(x.seq.size(), 0);
        for (size_t j = 0; j < x.seq.size(); j++) extdInitMap.push_back(vector<int>());
        bool ok; vector<string> psl = read_lines("tmp/_short_initial_contigs_extended_contigs." + itoa(u) + ".psl", ok);
        if (!ok) { cout << "CANNOT OPEN FILE!" << endl; return; }
        int targetIDBak = -1;
        for (const string &line : psl) {
            const PslHit h = parse_psl(line);
            if (!((double)(unsigned)(h.sEnd - h.sStart - h.sGap) / h.sSize >= 0.8 && (double)(unsigned)(h.tEnd - h.tStart - h.tGap) / (double)(unsigned)(h.tEnd - h.tStart) >= 0.8 &&
                  h.tSize > h.realSourceSize + 100 && h.realSourceSize > h.tSize / 100)) continue;
            if (h.tID >= extdTags.size() || h.sID >= initTags.size() || h.tID >= extdInitMap.size()) continue;   // the reference would write out of bounds
            if (o.uniqueExtension == 1) {                                     // AG:3061-3081
                if (initTags[h.sID] > 0 && targetIDBak != -1) {
                    if (extdTags[targetIDBak] < extdTags[h.tID]) {
                        extdTags[targetIDBak] = 0; if (!extdInitMap[targetIDBak].empty()) extdInitMap[targetIDBak].pop_back();
                        extdTags[h.tID] = (int)h.tSize; initTags[h.sID] = 1; extdInitMap[h.tID].push_back((int)h.sID);
                    }
                } else { extdTags[h.tID] = (int)h.tSize; initTags[h.sID] = 1; extdInitMap[h.tID].push_back((int)h.sID); }
                targetIDBak = (int)h.tID;
            } else { extdTags[h.tID] = 1; initTags[h.sID] = 1; extdInitMap[h.tID].push_back((int)h.sID); }
        }
        for (size_t j = 0; j < extdTags.size(); j++) {
            if (extdTags[j] <= 0) continue;
            // genomeIds is indexed by UNIT in the reference (AG:3102); with --part > 1 that runs past its end (undefined there): empty here
            e << ">" << "AlignGraph" << seqID << " @ " << ((size_t)u < genomeIds.size() ? genomeIds[u] : string()) << " : ";
            for (int s : extdInitMap[j]) e << ((size_t)s < contigIds.size() ? contigIds[s] : string()) << " ; ";
            e << '\n';
            ext << ">" << u << ": " << seqID << '\n';
            seqID++;
            put60(e, x.seq[j]); put60(ext, x.seq[j]);
        }
    }
    for (size_t i = 0; i < initTags.size(); i++)
        if (initTags[i] == 0) { r << ">" << (i < contigIds.size() ? contigIds[i] : string()) << '\n'; put60(r, init[i]); }
    {
        bool ok; vector<string> chaff = read_lines("tmp/_chaff.fa", ok);
        if (!ok) { cout << "CANNOT OPEN FILE!" << endl; return; }
        for (const string &b : chaff) r << b << '\n';
    }
    for (size_t i = 0; i < initTags.size(); i++) if (initTags[i] == 1) { ini << ">" << i << '\n'; put60(ini, init[i]); }
}

// ---- misassembly removal, AG:3821-4297 (row f2) ------------------------------------------------------------------------------------
// Per-base read coverage on the output contigs + contig-to-genome alignment blocks -> unaligned, thinly covered stretches are cut
// out and contigs are split there.  Integer / ratio rules are kept literally, including the order-dependent clean-up passes.
struct CBase { char base; int cov; };
struct CPos { int tID; unsigned sStart, sEnd, tStart, tEnd; int fr; };
const CPos CP_NONE = {-1, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, -1};

inline int masb_conflict(unsigned x1, unsigned y1, unsigned x2, unsigned y2) {      // AG:3986-3993
    return (x1 <= x2 && x2 <= y1 && y1 <= y2 && (int)y1 - (int)x2 >= 100) || (x2 <= x1 && x1 <= y2 && y2 <= y1 && (int)y2 - (int)x1 >= 100) ||
           (x1 <= x2 && x2 <= y2 && y2 <= y1 && (int)y2 - (int)x2 >= 100) || (x2 <= x1 && x1 <= y1 && y1 <= y2 && (int)y1 - (int)x1 >= 100) ||
           (x1 <= x2 && y2 <= y1) || (x2 <= x1 && y1 <= y2);
}
inline int masb_close(unsigned y1, unsigned x2, unsigned threshold) { return (unsigned)abs((int)x2 - (int)y1) < threshold; }   // AG:3995-4001 (int vs unsigned compare)
inline int masb_overlap(unsigned x1, unsigned y1, unsigned x2, unsigned y2) {       // AG:2388-2394
    return (x1 <= x2 && x2 <= y1 && y1 <= y2 && (int)y1 - (int)x2 > 0) || (x2 <= x1 && x1 <= y2 && y2 <= y1 && (int)y2 - (int)x1 > 0) ||
           (x1 <= x2 && x2 <= y2 && y2 <= y1 && (int)y2 - (int)x2 > 0) || (x2 <= x1 && x1 <= y1 && y1 <= y2 && (int)y1 - (int)x1 > 0);
}

// The columns of a SAM line that parseBOWTIE (AG:181-285) turns into a place on a target record: the record (the number in front of the '.' of RNAME, 0
// without one; none — tid -1 — as soon as the column holds a '*'), the 0-based start (POS - 1, unsigned) and the end = start + the M, S and I lengths + the D
// lengths - the I lengths.  `bad`: a CIGAR character the reference stops the program at ("unknown character").
struct SamSpan { int tid; unsigned ts, te; char bad; };
inline int field_atoi(const char *s, size_t n) {
    char b[32];
    if (n < sizeof b) { memcpy(b, s, n); b[n] = 0; return atoi(b); }
    return atoi(string(s, n).c_str());
}
inline SamSpan sam_span(const char *s, size_t n) {
    const char *f[6]; size_t fl[6]; size_t a = 0; int nf = 0;
    for (; nf < 6; nf++) {
        const char *tab = a <= n ? (const char *)memchr(s + a, '\t', n - a) : nullptr;
        f[nf] = s + a; fl[nf] = tab ? (size_t)(tab - (s + a)) : n - a;
        if (!tab) { nf++; break; }
        a = (size_t)(tab - s) + 1;
    }
    for (; nf < 6; nf++) { f[nf] = s + n; fl[nf] = 0; }
    SamSpan r{0, 0, 0, 0};
    if (fl[2] && memchr(f[2], '*', fl[2])) { r.tid = -1; return r; }
    const char *dot = fl[2] ? (const char *)memchr(f[2], '.', fl[2]) : nullptr;
    r.tid = dot ? field_atoi(f[2], (size_t)(dot - f[2])) : 0;
    int total = 0, ins = 0, del = 0; unsigned num = 0;
    for (size_t i = 0; i < fl[5]; i++) {
        const char c = f[5][i];
        if (c >= '0' && c <= '9') { num = num * 10 + (unsigned)(c - '0'); continue; }
        if (c == 'I') { ins += (int)num; total += (int)num; } else if (c == 'D') del += (int)num; else if (c == 'M' || c == 'S') total += (int)num;
        else if (c == '*') continue;                   // (the digits in front of it stay in the reference's buffer)
        else { r.bad = c; return r; }
        num = 0;
    }
    r.ts = (unsigned)(field_atoi(f[3], fl[3]) - 1); r.te = r.ts + (unsigned)(total + del - ins);
    return r;
}

// loadReadAlignment of the misassembly removal, AG:3940-3984: the per-base read coverage of the chunk records (len[i] bases each) from the SAM file of the reads
// against them — +1 over [start, end) for both mates of every pair that has both mates placed; lines in pairs, '@' lines in front of a pair skipped, the file ends at
// its first empty line.  cov[off[i] + bp] = coverage of base bp of record i.  The reference reads the file (9 GB for 30 M reads) a line at a time and counts base by base;
// here every CPU the process may use parses a piece of the mapped file and adds +1 / -1 at the two ends of a span, and one running sum per record makes the counts.
// A place outside its record counts as far as the record goes (the reference writes out of bounds there); a pair that names a record that does not exist counts nothing.
void masb_read_coverage(const string &path, const vector<size_t> &len, vector<int> &cov, vector<size_t> &off) {
    const auto t_start = std::chrono::steady_clock::now();
    Lines in(path);
    if (!in.opened) die("CANNOT OPEN FILE!");
    off.assign(len.size() + 1, 0);
    for (size_t i = 0; i < len.size(); i++) off[i + 1] = off[i] + len[i] + 1;      // (one more slot per record: where a span that reaches the record's end stops)
    cov.assign(off.back() + 1, 0);
    const size_t least = getenv("AGX_CLI_FAST_MIN") ? (size_t)atoll(getenv("AGX_CLI_FAST_MIN")) : ((size_t)8 << 20);      // (tests: small files on several threads)
    const unsigned T = in.n >= least && in.n > 4096 && !getenv("AGX_CLI_SERIAL") ? cli_threads() : 1u;
    const char *p = in.p; const size_t n = scan_end(p, in.n, T);
    size_t body = 0;                                   // behind the '@' lines at the top
    while (body < n && p[body] == '@') { const char *nl = (const char *)memchr(p + body, '\n', n - body); body = nl ? (size_t)(nl - p) + 1 : n; }
    int *const d = cov.data();
    const size_t n_rec = len.size();
    auto add = [&](const SamSpan &x) {                 // (bp < targetEnd is an unsigned comparison in the reference: an empty span if start >= end)
        if (x.ts >= x.te || (size_t)x.ts >= len[(size_t)x.tid]) return;
        const size_t base = off[(size_t)x.tid], hi = std::min<size_t>(x.te, len[(size_t)x.tid]);
        __atomic_fetch_add(d + base + x.ts, 1, __ATOMIC_RELAXED); __atomic_fetch_add(d + base + hi, -1, __ATOMIC_RELAXED);
    };
    struct Fatal { size_t at; char bad; bool broken; };
    // lines [lo, hi) in pairs; the first of them is the first line of a pair.  The first thing the reference would stop at, if any.
    auto pairs = [&](size_t lo, size_t hi, size_t end, Fatal &fatal) {
        for (size_t at = lo; at < hi;) {
            const char *nl = (const char *)memchr(p + at, '\n', end - at);
            const size_t l1 = nl ? (size_t)(nl - p) - at : end - at, at2 = at + l1 + 1;
            if (p[at] == '@') { at = at2; continue; }
            const SamSpan x = sam_span(p + at, l1);
            if (x.bad) { fatal = Fatal{at, x.bad, false}; return; }
            if (at2 >= end) { fatal = Fatal{at, 0, true}; return; }
            const char *nl2 = (const char *)memchr(p + at2, '\n', end - at2);
            const size_t l2 = nl2 ? (size_t)(nl2 - p) - at2 : end - at2;
            const SamSpan y = sam_span(p + at2, l2);
            if (y.bad) { fatal = Fatal{at2, y.bad, false}; return; }
            at = at2 + l2 + 1;
            if (x.tid == -1 || y.tid == -1 || (size_t)x.tid >= n_rec || (size_t)y.tid >= n_rec) continue;
            add(x); add(y);
        }
    };
    vector<Fatal> fatal(T, Fatal{(size_t)-1, 0, false});
    bool parallel = T > 1;
    if (parallel) {
        // the pieces' first lines must be first lines of pairs: count the lines of every piece (and look for '@' lines behind the top, which would shift the pairing)
        vector<size_t> cut(T + 1), lines(T, 0); vector<char> at_sign(T, 0);
        for (unsigned t = 0; t <= T; t++) cut[t] = t == T ? n : std::max(body, line_start_at(p, n, body + (n - body) / T * t));
        on_threads(T, [&](unsigned t) {
            size_t c = 0;
            for (size_t at = cut[t]; at < cut[t + 1];) { if (p[at] == '@') at_sign[t] = 1; const char *nl = (const char *)memchr(p + at, '\n', cut[t + 1] - at); c++; if (!nl) break; at = (size_t)(nl - p) + 1; }
            lines[t] = c;
        });
        for (unsigned t = 0; t < T; t++) if (at_sign[t]) parallel = false;
        if (parallel) {
            size_t before = 0; vector<size_t> lo(cut.begin(), cut.end() - 1);
            for (unsigned t = 0; t < T; t++) {       // a piece that begins with the second line of a pair leaves that line to the piece in front of it (which reads past its own end for it)
                if ((before & 1) && cut[t] < cut[t + 1]) { const char *nl = (const char *)memchr(p + cut[t], '\n', n - cut[t]); lo[t] = std::min(nl ? (size_t)(nl - p) + 1 : n, cut[t + 1]); }
                before += lines[t];
            }
            on_threads(T, [&](unsigned t) { pairs(lo[t], cut[t + 1], n, fatal[t]); });
        }
    }
    if (!parallel) pairs(body, n, n, fatal[0]);
    if (getenv("AGX_CLI_TIMING")) fprintf(stderr, "[agx cli]   read coverage: %zu bytes of SAM on %u thread%s, %.3f s\n", n, parallel ? T : 1u, parallel ? "s" : "", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    const Fatal *first = nullptr;
    for (const Fatal &f : fatal) if (f.at != (size_t)-1 && (!first || f.at < first->at)) first = &f;
    if (first) { if (first->broken) die("BROKEN BOWTIE FILE!"); cout << "unknown character: " << first->bad << endl; exit(-1); }
    // +1 / -1 at the ends of the spans -> counts
    const unsigned TS = len.size() > 64 ? T : 1u;
    on_threads(TS, [&](unsigned t) { for (size_t i = len.size() * t / TS; i < len.size() * (t + 1) / TS; i++) { int run = 0; int *c = d + off[i]; for (size_t bp = 0; bp <= len[i]; bp++) { run += c[bp]; c[bp] = run; } } });
}

void remove_misassembly(const Options &o, const string &file, const string &id, vector<string> &contigIds) {
    const bool timing = getenv("AGX_CLI_TIMING") != nullptr;      // (stderr: the aligners' share of this step, and its own)
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_in = now_s();
    const string fa = "tmp/_" + id + "_contigs.fa";
    formalize_contigs(file, contigIds, fa);                                        // AG:4288-4290; contigIds now names THIS file's records
    // makeAlignment, AG:3821-3851
    run("bowtie2-build -f tmp/_" + id + "_contigs.fa tmp/_" + id + "_contigs > bowtie_doc.txt 2> bowtie_doc.txt");
    run("bowtie2 -f --no-mixed -k 1 -p 8 -I " + itoa(o.distanceLow) + " -X " + itoa(o.distanceHigh) + " --no-discordant -x tmp/_" + id +
        "_contigs -1 tmp/_reads_1.fa -2 tmp/_reads_2.fa --reorder > tmp/_reads_" + id + "_contigs.bowtie 2> bowtie_doc.txt");
    if (o.fastMap == 1)                            // the reference hands NUCMER a prefix that already ends in ".delta" (AG:3833), so its output is never found
        nucmer_to_psl("tmp/_genome.fa", "tmp/_" + id + "_contigs.fa", "tmp/_" + id + "_contigs_genome.delta", "tmp/_" + id + "_contigs_genome.delta");
    else {
        const string io = "tmp/_genome.fa tmp/_" + id + "_contigs.fa -noHead tmp/_" + id + "_contigs_genome.psl -fastMap";
        if (run("pblat " + io + " -threads=8 > blat_doc.txt 2> blat_doc.txt") != 0)
            if (run("blat " + io + " > blat_doc.txt 2> blat_doc.txt") != 0) die("BLAT CALL FAILED!");
    }
    const double t_aligned = now_s();
    // loadPreContigs, AG:3853-3891: one coverage counter per base of every chunk record
    bool ok; vector<string> lines = read_lines(fa, ok);
    if (!ok) die("CANNOT OPEN FILE!");
    vector<vector<CBase> > pre; vector<int> realOf;
    for (const string &b : lines) {
        if (b[0] == '>') { pre.push_back(vector<CBase>()); const size_t d = b.find('.'); realOf.push_back(atoi(d == string::npos ? "" : b.substr(d + 1, 9).c_str())); }
        else if (!pre.empty()) for (char c : b) pre.back().push_back(CBase{c, 0});
    }
    // loadReadAlignment, AG:3940-3984: +1 over [POS-1, POS-1 + CIGAR span on the contig) for both mates of every aligned pair
    {
        vector<size_t> len(pre.size()); for (size_t i = 0; i < pre.size(); i++) len[i] = pre[i].size();
        vector<int> cov; vector<size_t> off;
        masb_read_coverage("tmp/_reads_" + id + "_contigs.bowtie", len, cov, off);
        if (const char *dump = getenv("AGX_CLI_MASB_COV")) {      // (tests: the counts themselves, record after record, as 32-bit integers)
            std::ofstream w((string(dump) + "." + id + ".bin").c_str(), std::ios::binary);
            for (size_t i = 0; i < pre.size(); i++) w.write((const char *)(cov.data() + off[i]), (std::streamsize)(len[i] * sizeof(int)));
        }
        on_threads(pre.size() > 64 ? cli_threads() : 1u, [&](unsigned t) {
            const unsigned T = pre.size() > 64 ? cli_threads() : 1u;
            for (size_t i = pre.size() * t / T; i < pre.size() * (t + 1) / T; i++) { const int *c = cov.data() + off[i]; vector<CBase> &P = pre[i]; for (size_t bp = 0; bp < P.size(); bp++) P[bp].cov = c[bp]; }
        });
    }
    // loadContigs, AG:3893-3938: chunks of one real contig are concatenated (a new contig starts when realID grows)
    vector<vector<CBase> > contigs; int realBak = -1;
    for (size_t i = 0; i < pre.size(); i++) { if (realOf[i] > realBak) { contigs.push_back(vector<CBase>()); realBak = realOf[i]; } if (!contigs.empty()) contigs.back().insert(contigs.back().end(), pre[i].begin(), pre[i].end()); }
    // loadContigAlignment, AG:4003-4145
    vector<vector<CPos> > pos(contigs.size());
    {
        vector<string> psl = read_lines("tmp/_" + id + "_contigs_genome.psl", ok);
        if (!ok) die("CANNOT OPEN FILE!");
        int realBak2 = -1; unsigned sourceIDBak = 0;
        for (const string &line : psl) {
            vector<string> f; size_t a = 0;
            for (;;) { size_t b = line.find('\t', a); f.push_back(line.substr(a, b == string::npos ? string::npos : b - a)); if (b == string::npos) break; a = b + 1; }
            f.resize(21);
            const PslHit h = parse_psl(line);
            const int realSourceID = (int)h.realSourceSize;                          // parseBLAT's return value: the number after the '.', else qSize
            const int fr = f[8].empty() ? -1 : (f[8][0] == '+' ? 0 : 1);
            if (realSourceID > realBak2) { realBak2 = realSourceID; sourceIDBak = h.sID; }
            const unsigned sStart = (h.sID - sourceIDBak) * 1000000u + h.sStart, sEnd = (h.sID - sourceIDBak) * 1000000u + h.sEnd;
            if (!(sEnd - sStart >= 100 && (double)(unsigned)(sEnd - sStart - h.sGap) / (unsigned)(sEnd - sStart) >= 0.1 &&
                  (double)(unsigned)(h.tEnd - h.tStart - h.tGap) / (double)(unsigned)(h.tEnd - h.tStart) >= 0.1)) continue;
            if (realSourceID < 0 || (size_t)realSourceID >= pos.size()) continue;      // out of bounds in the reference
            vector<CPos> &P = pos[realSourceID];
            int keep = 1;
            for (size_t pp = 0; pp < P.size(); pp++)
                if (P[pp].tID != -1 && (int)h.tID == P[pp].tID && masb_conflict(sStart, sEnd, P[pp].sStart, P[pp].sEnd)) {
                    if (sEnd - sStart < P[pp].sEnd - P[pp].sStart) keep = 0; else P[pp] = CP_NONE;
                }
            if (keep) P.push_back(CPos{(int)h.tID, sStart, sEnd, h.tStart, h.tEnd, fr});
        }
    }
    for (size_t sp = 0; sp < pos.size(); sp++) {                                      // join consecutive local alignments, AG:4068-4081
        vector<CPos> &P = pos[sp];
        for (int pp = 0; pp < (int)P.size(); pp++)
            for (int q = 0; q < (int)P.size(); q++)
                if (q != pp && P[pp].tID != -1 && P[q].tID != -1 && P[pp].tID == P[q].tID &&
                    masb_close(P[pp].sEnd, P[q].sStart, (unsigned)(abs((int)P[pp].sEnd - (int)P[pp].sStart) / 10)) &&
                    masb_close(P[pp].tEnd, P[q].tStart, (unsigned)(abs((int)P[pp].tEnd - (int)P[pp].tStart) / 10)) && P[pp].fr == P[q].fr) {
                    P[pp].sEnd = P[q].sEnd; P[pp].tEnd = P[q].tEnd; P[q] = CP_NONE; q = 0;   // (the loop increment makes the rescan start at 1, as in the reference)
                }
    }
    for (size_t sp = 0; sp < pos.size(); sp++) {                                      // of two conflicting placements the longer stays, AG:4083-4090
        vector<CPos> &P = pos[sp];
        for (size_t pp = 0; pp < P.size(); pp++) for (size_t q = pp + 1; q < P.size(); q++)
            if (P[pp].tID != -1 && P[q].tID != -1 && masb_conflict(P[pp].sStart, P[pp].sEnd, P[q].sStart, P[q].sEnd)) {
                if (P[pp].sEnd - P[pp].sStart > P[q].sEnd - P[q].sStart) P[q] = CP_NONE; else P[pp] = CP_NONE;
            }
    }
    for (size_t sp = 0; sp < pos.size(); sp++) {                                      // overlapping / adjacent placements are cut at the thinnest base, AG:4093-4141
        vector<CPos> &P = pos[sp]; const vector<CBase> &C = contigs[sp];
        for (size_t pp = 0; pp < P.size(); pp++) for (size_t q = pp + 1; q < P.size(); q++) {
            if (P[pp].tID != -1 && P[q].tID != -1 && masb_overlap(P[pp].sStart, P[pp].sEnd, P[q].sStart, P[q].sEnd)) {
                int mn = 99999, mp = -1, start, end;
                if (P[pp].sStart <= P[q].sStart) { start = (int)P[q].sStart; end = (int)P[pp].sEnd - 1; } else { start = (int)P[pp].sStart; end = (int)P[q].sEnd - 1; }
                for (int bp = start; bp <= end; bp++) if (bp >= 0 && (size_t)bp < C.size() && C[bp].cov < mn) { mn = C[bp].cov; mp = bp; }
                if (P[pp].sStart <= P[q].sStart) { P[pp].sEnd = (unsigned)mp; P[q].sStart = (unsigned)(mp + 1); } else { P[q].sEnd = (unsigned)mp; P[pp].sStart = (unsigned)(mp + 1); }
            } else if (P[pp].tID != -1 && P[q].tID != -1 && P[pp].sEnd == P[q].sStart) {
                if (P[pp].sEnd - 1 < C.size() && P[q].sStart < C.size() && C[P[pp].sEnd - 1].cov < C[P[q].sStart].cov) P[pp].sEnd--; else P[q].sStart++;
            } else if (P[pp].tID != -1 && P[q].tID != -1 && P[q].sEnd == P[pp].sStart) {
                if (P[q].sEnd - 1 < C.size() && P[pp].sStart < C.size() && C[P[q].sEnd - 1].cov < C[P[pp].sStart].cov) P[q].sEnd--; else P[pp].sStart++;
            }
        }
    }
    // removeMasb, AG:4147-4279: cov becomes -1 (keep) or -2 (remove)
    for (size_t cp = 0; cp < pos.size(); cp++) {
        vector<CBase> &C = contigs[cp]; const vector<CPos> &P = pos[cp];
        bool whole = false;
        for (const CPos &p : P) if (p.tID != -1 && (double)(unsigned)(p.sEnd - p.sStart) / C.size() >= 0.8) { whole = true; break; }
        if (whole) { for (CBase &b : C) b.cov = -1; continue; }
        for (const CPos &p : P) if (p.tID != -1) for (int bp = (int)p.sStart; (unsigned)bp < p.sEnd; bp++) if ((size_t)bp < C.size()) C[bp].cov = -1;
        int start = 0, end = 0, total = 0; const int n = (int)C.size();
        for (int bp = 0; bp < n; bp++) {
            if (C[bp].cov == -1) continue;
            if (bp != 0 && bp != n - 1 && C[bp - 1].cov == -1 && C[bp + 1].cov == -1) { C[bp].cov = C[bp].cov < o.coverage ? -2 : -1; continue; }
            if (bp == 0 || C[bp - 1].cov == -1) { start = bp; total = C[bp].cov; }
            else if (bp == n - 1 || C[bp + 1].cov == -1) {
                end = bp; total += C[bp].cov;
                const int v = total / (end - start + 1) < o.coverage ? -2 : -1;
                for (int b2 = start; b2 <= end; b2++) C[b2].cov = v;
            } else total += C[bp].cov;
        }
    }
    std::ofstream out(("corrected_" + file).c_str());
    if (!out.is_open()) die("CANNOT OPEN FILE!");
    for (size_t cp = 0; cp < contigs.size(); cp++) {
        const vector<CBase> &C = contigs[cp]; vector<string> parts; const int n = (int)C.size();
        for (int bp = 0; bp < n; bp++) {
            // (the reference reads C[bp-1] at bp == 0 — one element before the array; it can only matter if that stray value were -2)
            if ((parts.empty() && C[bp].cov == -1) || (bp > 0 && C[bp - 1].cov == -2 && C[bp].cov == -1)) parts.push_back(string());
            if (C[bp].cov == -1 && !parts.empty()) parts.back().push_back(C[bp].base);
            if (bp == n - 1 || (C[bp].cov == -1 && C[bp + 1].cov == -2)) if (!parts.empty() && parts.back().size() <= 200) parts.pop_back();
        }
        for (size_t sp = 0; sp < parts.size(); sp++) {
            const string name = cp < contigIds.size() ? contigIds[cp] : string();
            if (parts.size() == 1) out << ">" << name << '\n'; else out << ">" << name << " : part" << sp << '\n';
            put60(out, parts[sp]);
        }
    }
    if (id == "remaining") { vector<string> chaff = read_lines("tmp/_chaff.fa", ok); if (ok) for (const string &b : chaff) out << b << '\n'; }
    if (timing) fprintf(stderr, "[agx cli]   %-9s contigs: aligners (external) %9.3f s, coverage + placements + cuts %9.3f s\n", id.c_str(), t_aligned - t_in, now_s() - t_aligned);
}

// ---- the unit loop on the GPUs --------------------------------------------------------------------------------------------------
// Units are independent (AG:4779-4781 clears all state between them): a work queue feeds one host thread per device slot; progress
// lines and checkpoints are emitted in unit order, as the sequential reference would.
void run_units(const Options &o, int first, int units, std::ofstream &wcp) {
    const int ndev_all = agx_device_count();
    if (ndev_all <= 0) die("NO HIP DEVICE: AlignGraph_amd needs a GPU (there is no CPU path)");
    int ndev = ndev_all;
    if (const char *e = getenv("AGX_DEVICES")) ndev = std::max(1, std::min(ndev_all, atoi(e)));
    const int per_dev = std::max(1, getenv("AGX_UNITS_PER_DEVICE") ? atoi(getenv("AGX_UNITS_PER_DEVICE")) : 4);   // >1: the text parsing and host walk of one unit overlap the kernels of others
    // every unit reads tmp/_reads.fa (AG:1880): map and index it once for all of them
    agx_reads *reads = nullptr; { char err[512]; if (agx_reads_open("tmp/_reads.fa", &reads, err, sizeof err) != AGX_OK) reads = nullptr; }   // (a missing file is reported by the first unit, as before)
    // Units of very different sizes share a device: a unit is loaded (host work only), asked what its upload will take (agx_unit_hbm_needed: the one
    // block of HBM sized from its staged counts) and admitted only while the blocks of the units in flight stay below 85 % of the device's memory
    // (a unit larger than that still runs, alone).  A build that has to grow a capacity takes more than its block: the 15 % are for that.
    auto file_bytes = [](const string &p) -> double { struct stat st; return stat(p.c_str(), &st) == 0 ? (double)st.st_size : 0.0; };
    vector<double> budget(ndev, 0.0), used(ndev, 0.0); vector<int> waiting(ndev, 0);      // waiting: units loaded and not yet admitted to the device
    for (int d = 0; d < ndev; d++) { uint64_t fr = 0, tot = 0; budget[d] = agx_device_memory(d, &fr, &tot) == AGX_OK ? 0.85 * (double)tot : 1e18; }
    std::mutex mem_mu; std::condition_variable mem_cv;
    // Longest unit first (its host walk then runs beside the uploads and kernels of the shorter ones); progress lines and
    // checkpoints still come out in unit order.  A unit that fails stops the hand-out; the message is printed by the main thread once the
    // workers are back (an exit() from a worker would tear HIP down under the other workers' feet).
    vector<int> order; for (int u = first; u < units; u++) order.push_back(u);
    vector<double> weight(units, 0.0);         // (only orders the hand-out: the unit sequence and its alignments decide how long a unit takes)
    for (int u = first; u < units; u++) weight[u] = 200.0 * file_bytes("tmp/_genome." + itoa(u) + ".fa") + 2.0 * file_bytes("tmp/_reads_genome." + itoa(u) + ".bowtie");
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return weight[a] > weight[b]; });
    std::atomic<size_t> next(0); std::atomic<bool> failed(false);
    vector<int> state(units, 0); vector<string> errors(units);
    std::mutex mu; int reported = first;
    auto report = [&]() {                                                     // under mu: flush finished units in order, up to the first failure
        while (reported < units && state[reported] > 0) {
            cout << endl << "CHROMOSOME " << reported << ": " << endl;
            cout << "(1) Chromosome loaded" << endl << "(2) Contig alignment loaded" << endl << "(3) Read alignment loaded" << endl
                 << "(4) Contigs extended" << endl << "(5) Contigs scaffolded" << endl;
            wcp << itoa(reported + 1) << endl;                                // setCheckpoint, AG:4782
            reported++;
        }
    };
    vector<std::thread> workers;
    for (int d = 0; d < ndev; d++) for (int s = 0; s < per_dev; s++)
        workers.emplace_back([&, d]() {
            for (;;) {
                const size_t at = next.fetch_add(1);
                if (at >= order.size() || failed.load()) return;
                const int u = order[at];
                // the five calls of the unit loop (AG:4768-4776) through the split entry points of agx_run_unit, with the admission between load and upload
                agx_params p = {(uint32_t)o.k, (uint32_t)o.insertVariation, (uint32_t)o.coverage, 0, d, AGX_FLAG_ONE_SHOT};
                agx_result r; memset(&r, 0, sizeof r); char err[512]; err[0] = 0;
                agx_unit *un = nullptr;
                int rc = agx_unit_create(&p, &un);
                if (rc != AGX_OK) snprintf(err, sizeof err, "%s", rc == AGX_E_NOGPU ? "no HIP device" : "bad parameters");
                double est = 0; bool admitted = false;
                if (rc == AGX_OK) rc = agx_unit_load_files_shared(un, "tmp", u, reads);
                if (rc == AGX_OK) { uint64_t need = 0; rc = agx_unit_hbm_needed(un, &need); est = (double)need; }
                if (rc == AGX_OK) { std::unique_lock<std::mutex> g(mem_mu); waiting[d]++; mem_cv.wait(g, [&] { return used[d] == 0.0 || used[d] + est <= budget[d]; }); waiting[d]--; used[d] += est; admitted = true; }
                if (rc == AGX_OK) rc = agx_unit_upload(un);
                if (rc == AGX_OK) rc = agx_unit_build(un);
                bool wanted = false; if (rc == AGX_OK) { std::lock_guard<std::mutex> g(mem_mu); wanted = waiting[d] > 0; }
                if (rc == AGX_OK && wanted) {      // r05: somebody waits for room on this device — the whole download first, then what the walk cannot ask the device for goes back: the next unit is admitted while this one is walked on the host
                    rc = agx_unit_download(un);
                    uint64_t freed = 0;
                    if (rc == AGX_OK && agx_unit_trim(un, &freed) == AGX_OK && freed) { { std::lock_guard<std::mutex> g(mem_mu); const double f = std::min(est, (double)freed); used[d] -= f; est -= f; } mem_cv.notify_all(); }
                }
                if (rc == AGX_OK) rc = agx_unit_finish(un, &r);      // (r06: nothing downloaded yet — nobody waits for this unit's HBM — : the download is streamed and the walk begins on what has landed)
                if (rc != AGX_OK && un && !err[0]) snprintf(err, sizeof err, "%s", agx_unit_error(un));
                agx_unit_destroy(un);                                          // (its HBM goes back before the next unit is admitted)
                if (admitted) { { std::lock_guard<std::mutex> g(mem_mu); used[d] -= est; if (used[d] < 1.0) used[d] = 0.0; } mem_cv.notify_all(); }
                if (rc == AGX_OK) {
                    const string su = itoa(u);
                    auto put = [&](const string &path, const char *data, size_t len) { FILE *f = fopen(path.c_str(), "wb"); bool ok = f != nullptr; if (f) { ok = len == 0 || fwrite(data, 1, len, f) == len; ok = (fclose(f) == 0) && ok; } if (!ok && rc == AGX_OK) { rc = AGX_E_IO; snprintf(err, sizeof err, "CANNOT OPEN FILE!"); } };
                    put("tmp/_initial_contigs." + su + ".fa", r.initial_contigs, r.initial_len);
                    put("tmp/_pre_extended_contigs." + su + ".fa", r.pre_extended, r.pre_len);
                    put("tmp/_extended_contigs." + su + ".fa", r.extended, r.extended_len);
                }
                agx_result_free(&r);
                std::lock_guard<std::mutex> g(mu);
                if (rc != AGX_OK) { string m = err; const size_t cut = m.find(" ("); errors[u] = cut == string::npos ? m : m.substr(0, cut); state[u] = -1; failed.store(true); }
                else state[u] = 1;
                report();
            }
        });
    for (auto &t : workers) t.join();
    agx_reads_close(reads);
    if (failed.load()) {                                                      // the first unit (in unit order) that did not finish: what the sequential reference would have stopped at
        int u = reported; while (u < units && state[u] >= 0) u++;
        if (u < units) { cout << endl << "CHROMOSOME " << u << ": " << endl << errors[u] << endl; cout.flush(); exit(-1); }
    }
}

}  // namespace

int main(int argc, char **argv) {
    cout << "AlignGraph: algorithm for secondary de novo genome assembly guided by closely related references" << endl;
    cout << "By Ergude Bao, CS Department, UC-Riverside. All Rights Reserved" << endl << endl;
    const time_t start = time(NULL);
    // AGX_CLI_TIMING=1: wall time of every stage on stderr (the reference only reports whole seconds for the run and for the aligners)
    const bool timing = getenv("AGX_CLI_TIMING") != nullptr;
    auto clock_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_stage = clock_s();
    auto stage = [&](const char *name) {
        const double t = clock_s(), ext = (double)g_run_ns.exchange(0) * 1e-9;
        if (timing) { if (ext > 0.0005) fprintf(stderr, "[agx cli] %-28s %9.3f s (%.3f s of it waiting for external commands)\n", name, t - t_stage, ext); else fprintf(stderr, "[agx cli] %-28s %9.3f s\n", name, t - t_stage); }
        t_stage = t;
    };
    {
        std::ofstream wcmd("command.txt");
        if (!wcmd.is_open()) { cout << "CANNOT OPEN FILE!" << endl; return 0; }
        for (int i = 1; i < argc; i++) wcmd << argv[i] << endl;
    }
    Options o; parse_params("command.txt", o);
    vector<string> genomeIds, contigIds;
    int units = 0, cp = 0; time_t startAlign, endAlign; std::ofstream wcp;
    if (o.resume == 0) {
        if (o.tagRead1 == 0 || o.tagRead2 == 0 || o.tagContig == 0 || o.tagGenome == 0 || o.tagExt == 0 || o.tagRmn == 0 || o.k <= 0 || o.tagLow == 0 || o.tagHigh == 0 ||
            o.distanceLow > o.distanceHigh || o.distanceLow < 0 || o.insertVariation < 0 || o.part < 1 || o.part > 10 || o.k > max_read_length(o.read1) || o.k > max_read_length(o.read2)) {
            usage(); return 0;                                                // AG:4726-4730 (exit status 0)
        }
        if (run("bowtie2 -h > bowtie_doc.txt 2> bowtie_doc.txt") != 0) die("BOWTIE2 CALL FAILED!");   // testAligners, AG:4682-4694
        if (o.fastMap == 1 && run("nucmer -h > nucmer_doc.txt 2> nucmer_doc.txt") != 0) die("NUCMER CALL FAILED!");
        mkdir("tmp", 0777);
        { std::ofstream wcmd("tmp/_command.txt"); for (int i = 1; i < argc; i++) wcmd << argv[i] << endl; }
        wcp.open("tmp/_checkpoint.txt");
        stage("start-up, aligner checks");
        formalize_reads(o.read1, o.read2); stage("formalize reads");
        formalize_contigs(o.contig, contigIds); stage("formalize contigs");
        units = formalize_genome(o.genome, o.part, genomeIds); stage("formalize genome");
        startAlign = time(NULL);
        align_everything(o, units); stage("aligners + distribute + caches");
        endAlign = time(NULL);
        cout << "(0) Alignment finished" << endl;
        wcp << "0" << endl;
    } else {
        bool ok; vector<string> c = read_lines("tmp/_checkpoint.txt", ok);
        if (!ok) die("CANNOT OPEN FILE!");
        cp = -1; for (const string &s : c) cp = atoi(s.c_str());
        if (cp == -1) die("NOT REACHED CHECKPOINT. PLEASE RERUN!");
        o = Options(); parse_params("tmp/_command.txt", o);                   // AG:4752-4753 (the tags start from the --resume parse in the reference; only resume itself carries over)
        o.resume = 1;
        cout << "RESUMED SUCCESSFULLY :-)" << endl;
        wcp.open("tmp/_checkpoint.txt", std::ios::app);
        formalize_contigs(o.contig, contigIds);
        units = formalize_genome(o.genome, o.part, genomeIds);
        startAlign = endAlign = time(NULL);
    }
    if (o.ratioCheck == 1) check_ratio(units);
    stage("resume / ratio check");
    if (cp < units) run_units(o, cp, units, wcp);
    stage("unit loop");
    refinement(o, units, genomeIds, contigIds);
    stage("refinement");
    if (o.misassemblyRemoval == 1) {                                          // AG:4787-4792
        remove_misassembly(o, o.ext, "extended", contigIds);
        remove_misassembly(o, o.rmn, "remaining", contigIds);
        cout << endl << "(6) Misassemblies removed" << endl;
        stage("misassembly removal");
    }
    const time_t end = time(NULL);
    cout << endl << "FINISHED SUCCESSFULLY for " << end - start << " seconds (" << endAlign - startAlign << " seconds for alignment) :-)" << endl;
    return 0;
}
