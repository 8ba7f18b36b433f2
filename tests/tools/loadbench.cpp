// loadbench — times the host loaders (aligngraph_amd/csrc/agx_host.cpp) on one unit's text files, without a device.
//   g++ -O3 -std=c++17 -pthread -o build/loadbench tests/tools/loadbench.cpp aligngraph_amd/csrc/agx_host.cpp aligngraph_amd/csrc/agx_walk.cpp aligngraph_amd/csrc/agx_load.cpp
//   build/loadbench <tmp_dir> <unit> [k] [repeats]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include "../../aligngraph_amd/csrc/agx_host.h"
using namespace agx;
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: loadbench tmp_dir unit [k] [repeats]\n"); return 2; }
    const std::string d = argv[1], s = argv[2]; const agx_u32 k = argc > 3 ? (agx_u32)atoi(argv[3]) : 5; const int reps = argc > 4 ? atoi(argv[4]) : 3;
    try {
        for (int r = 0; r < reps; r++) {
            double t0 = now();
            ReadsIndex *ri = reads_index_open(d + "/_reads.fa");
            double t1 = now();
            Threads T; load_unit_reference(d + "/_genome." + s + ".fa", T.ref);
            double t2 = now();
            thread_contigs_from_files(d + "/_contigs.fa", d + "/_contigs_genome." + s + ".psl", T);
            double t3 = now();
            Pairs P; load_pairs_from_files(d + "/_reads.fa", d + "/_reads_genome." + s + ".bowtie", 1000000, k, P, ri);
            double t4 = now();
            printf("reads index %.1f ms | genome %.1f ms | contigs %.1f ms | pairs %.1f ms (%zu hits, %zu runs, %u slots) | total w/o index %.1f ms\n",
                   t1 - t0, t2 - t1, t3 - t2, t4 - t3, P.hits.size(), P.runs.size(), P.n_slots, t4 - t1);
            {   // the fast loaders (agx_load.cpp) on the same files
                struct VecSink : StageSink { std::vector<char> buf[SA_N]; void *take(int w, size_t b) override { buf[w].resize(b + 64); return buf[w].data(); } } sink;
                double f0 = now();
                Threads T2; load_unit_reference(d + "/_genome." + s + ".fa", T2.ref);
                double f1 = now();
                const bool okc = thread_contigs_fast(d + "/_contigs.fa", d + "/_contigs_genome." + s + ".psl", T2);
                double f2 = now();
                StagedPairs S; const bool okp = load_pairs_fast(*ri, d + "/_reads_genome." + s + ".bowtie", 1000000, k, loader_threads((size_t)1 << 40), sink, S);
                double f3 = now();
                printf("fast: genome %.1f ms | contigs %.1f ms (%s) | pairs + staging %.1f ms (%s; %zu hits, %u rows) | total %.1f ms\n", f1 - f0, f2 - f1, okc ? "ok" : "declined", f3 - f2, okp ? "ok" : "declined", S.nh, S.n_rows, f3 - f0);
            }
            reads_index_close(ri);
        }
    } catch (const Error &e) { fprintf(stderr, "error %d: %s\n", e.code, e.msg.c_str()); return 1; }
    return 0;
}
