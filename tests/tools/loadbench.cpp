// loadbench — times the host loaders (aligngraph_amd/csrc/agx_host.cpp, agx_load.cpp) on a run's text files, without a device.
//   g++ -O3 -std=c++17 -pthread -o build/loadbench tests/tools/loadbench.cpp aligngraph_amd/csrc/agx_host.cpp aligngraph_amd/csrc/agx_walk.cpp aligngraph_amd/csrc/agx_load.cpp
//   build/loadbench <tmp_dir> <units: 0 or 0,1,2 (loaded side by side)> [k] [repeats] [general=0|1]
// AGX_LOAD_THREADS sets the threads per unit, AGX_LOAD_TIMING=1 prints the loaders' phases.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <sys/resource.h>
#include "../../aligngraph_amd/csrc/agx_host.h"
using namespace agx;
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct VecSink : StageSink { Scratch buf[SA_N]; void *take(int w, size_t b) override { buf[w].take(b + 64); return buf[w].p; } };      // (the engine: pinned memory)
int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: loadbench tmp_dir units [k] [repeats] [general]\n"); return 2; }
    const std::string d = argv[1]; const agx_u32 k = argc > 3 ? (agx_u32)atoi(argv[3]) : 5; const int reps = argc > 4 ? atoi(argv[4]) : 3; const bool general = argc > 5 && atoi(argv[5]);
    std::vector<std::string> units; { std::string s = argv[2]; size_t a = 0; while (a <= s.size()) { size_t b = s.find(',', a); if (b == std::string::npos) b = s.size(); units.push_back(s.substr(a, b - a)); a = b + 1; } }
    try {
        for (int r = 0; r < reps; r++) {
            double t0 = now();
            ReadsIndex *ri = reads_index_open(d + "/_reads.fa");
            double t1 = now();
            printf("reads index %.1f ms\n", t1 - t0);
            if (general) for (const std::string &s : units) {
                double g0 = now();
                Threads T; load_unit_reference(d + "/_genome." + s + ".fa", T.ref);
                double g1 = now();
                thread_contigs_from_files(d + "/_contigs.fa", d + "/_contigs_genome." + s + ".psl", T);
                double g2 = now();
                Pairs P; load_pairs_from_files(d + "/_reads.fa", d + "/_reads_genome." + s + ".bowtie", 1000000, k, P, ri);
                printf("general, unit %s: genome %.1f ms | contigs %.1f ms | pairs %.1f ms (%zu hits) | total %.1f ms\n", s.c_str(), g1 - g0, g2 - g1, now() - g2, P.hits.size(), now() - g0);
            }
            double f0 = now();
            struct rusage ru0; getrusage(RUSAGE_SELF, &ru0);
            std::vector<std::thread> th;
            for (const std::string &s : units) th.emplace_back([&, s] {
                VecSink sink; double a0 = now(), a1 = 0, a2 = 0;
                Threads T2; bool okc = false;
                std::thread side([&] { load_unit_reference(d + "/_genome." + s + ".fa", T2.ref); a1 = now(); okc = thread_contigs_fast(d + "/_contigs.fa", d + "/_contigs_genome." + s + ".psl", T2); a2 = now(); });
                StagedPairs S; struct stat_dummy {}; 
                const bool okp = load_pairs_fast(*ri, d + "/_reads_genome." + s + ".bowtie", 1000000, k, loader_threads((size_t)1 << 40), sink, S);
                double a3 = now();
                side.join();
                printf("fast, unit %s: genome %.1f ms | contigs %.1f ms (%s) | pairs + staging %.1f ms (%s; %zu hits, %u rows) | wall %.1f ms\n", s.c_str(), a1 - a0, a2 - a1, okc ? "ok" : "declined", a3 - a0, okp ? "ok" : "declined", S.nh, S.n_rows, now() - a0);
            });
            for (auto &t : th) t.join();
            struct rusage ru1; getrusage(RUSAGE_SELF, &ru1);
            auto tv = [](const timeval &a, const timeval &b) { return (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_usec - a.tv_usec) * 1e-3; };
            printf("fast, %zu unit(s) side by side on %u threads each: %.1f ms  [process: user %.0f ms, system %.0f ms, minor faults %ld, voluntary switches %ld, involuntary %ld]\n", units.size(), loader_threads((size_t)1 << 40), now() - f0,
                   tv(ru0.ru_utime, ru1.ru_utime), tv(ru0.ru_stime, ru1.ru_stime), ru1.ru_minflt - ru0.ru_minflt, ru1.ru_nvcsw - ru0.ru_nvcsw, ru1.ru_nivcsw - ru0.ru_nivcsw);
            reads_index_close(ri);
        }
    } catch (const Error &e) { fprintf(stderr, "error %d: %s\n", e.code, e.msg.c_str()); return 1; }
    return 0;
}
