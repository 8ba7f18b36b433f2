// fault-in cost of fresh memory on the GPU box (g++ -O2 -pthread -o build/faultbench tests/tools/faultbench.cpp; build/faultbench <MB> <threads>):
// malloc vs mmap, with and without MADV_HUGEPAGE, MADV_POPULATE_WRITE, and what OutBuf does (malloc + MADV_HUGEPAGE on the aligned inner part)
#include <sys/mman.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double touch(char *p, size_t n) { double t = now(); for (size_t i = 0; i < n; i += 4096) p[i] = 1; return now() - t; }
int main(int argc, char **argv) {
    const size_t n = (size_t)(argc > 1 ? atoi(argv[1]) : 100) << 20;
    const int threads = argc > 2 ? atoi(argv[2]) : 1;
    { FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char b[128] = {0}; if (f) { fgets(b, 127, f); fclose(f); } printf("THP enabled: %s", b); }
    { FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/defrag", "r"); char b[128] = {0}; if (f) { fgets(b, 127, f); fclose(f); } printf("THP defrag: %s", b); }
    for (int mode = 0; mode < 7; mode++) {
        std::vector<double> ms(threads);
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back([&, t] {
            char *p; double t0 = now(), d;
            if (mode == 0) { p = (char *)malloc(n); d = touch(p, n); free(p); }
            else if (mode == 5 || mode == 6) {      // what OutBuf does: malloc/realloc, MADV_HUGEPAGE on the aligned inner part, touch
                p = mode == 5 ? (char *)malloc(n) : (char *)realloc(malloc(1 << 20), n);
                const uintptr_t lo = ((uintptr_t)p + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1), hi = ((uintptr_t)p + n) & ~(uintptr_t)((2 << 20) - 1);
                int rc = madvise((void *)lo, hi - lo, 14 /* MADV_HUGEPAGE */);
                d = touch(p, n); if (rc) d = -d; free(p);
            }
            else {
                p = (char *)mmap(nullptr, n + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                char *q = (char *)(((uintptr_t)p + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
                if (mode == 2 || mode == 4) madvise(q, n, MADV_HUGEPAGE);
                if (mode >= 3) { double t1 = now(); int rc = madvise(q, n, MADV_POPULATE_WRITE); d = now() - t1; if (rc) d = -1; }
                else d = touch(q, n);
                munmap(p, n + (2 << 20));
            }
            ms[t] = d; (void)t0;
        });
        for (auto &t : th) t.join();
        const char *names[7] = {"malloc + touch", "mmap + touch", "mmap + MADV_HUGEPAGE + touch", "mmap + POPULATE_WRITE", "mmap + HUGEPAGE + POPULATE_WRITE", "malloc + inner HUGEPAGE + touch", "realloc + inner HUGEPAGE + touch"};
        printf("%-36s %zu MB x %d threads:", names[mode], n >> 20, threads);
        for (double d : ms) printf(" %.1f", d);
        printf(" ms\n");
    }
    return 0;
}
