// membench — what allocation and PCIe copies cost on this box (design input for the upload path; not part of the product).
// hipcc -O2 -o build/membench tests/tools/membench.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// recycle: what HBM costs when it goes back to the driver and comes out again (is freed memory wiped before it is handed out?)
static void recycle() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (size_t gb : {1, 8, 30}) {
        const size_t n = gb << 30; void *a = nullptr, *b = nullptr;
        double t0 = now(); CK(hipMalloc(&a, n)); double t1 = now();
        CK(hipMemsetAsync(a, 1, n, st)); CK(hipStreamSynchronize(st)); double t2 = now();
        CK(hipFree(a)); double t3 = now();
        CK(hipMalloc(&b, n)); double t4 = now();
        CK(hipMemsetAsync(b, 2, 256, st)); CK(hipStreamSynchronize(st)); double t5 = now();
        CK(hipMemsetAsync(b, 2, n, st)); CK(hipStreamSynchronize(st)); double t6 = now();
        CK(hipFree(b)); double t7 = now();
        printf("%2zu GB: hipMalloc %.2f ms, fill %.2f, hipFree %.2f, hipMalloc again %.2f, first kernel on it %.2f, fill %.2f, hipFree %.2f\n", gb, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6);
    }
    // five blocks from five threads, freed, and taken again at once (what bench.py --pool cold does between two steps)
    for (int round = 0; round < 3; round++) {
        void *p[5]; double t0 = now();
        std::vector<std::thread> T; for (int i = 0; i < 5; i++) T.emplace_back([&, i] { CK(hipSetDevice(0)); CK(hipMalloc(&p[i], (size_t)6 << 30)); }); for (auto &x : T) x.join();
        double t1 = now();
        for (int i = 0; i < 5; i++) CK(hipMemsetAsync(p[i], 1, (size_t)6 << 30, st)); CK(hipStreamSynchronize(st)); double t2 = now();
        for (int i = 0; i < 5; i++) CK(hipFree(p[i])); double t3 = now();
        printf("round %d: 5 x hipMalloc(6 GB) on 5 threads %.2f ms, fill %.2f ms, 5 x hipFree %.2f ms\n", round, t1 - t0, t2 - t1, t3 - t2);
    }
}
int main(int argc, char **argv) {
    CK(hipSetDevice(0)); CK(hipFree(0));
    if (argc > 1 && !strcmp(argv[1], "recycle")) { recycle(); return 0; }
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (size_t mb : {1, 16, 64, 256, 1024, 4096}) {
        const size_t n = mb << 20;
        void *d = nullptr, *h = nullptr;
        double t0 = now(); CK(hipMalloc(&d, n)); double t1 = now();
        CK(hipMemsetAsync(d, 0, n, st)); CK(hipStreamSynchronize(st)); double t2 = now();
        CK(hipHostMalloc(&h, n, hipHostMallocDefault)); double t3 = now();
        memset(h, 1, n); double t4 = now();
        CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double t5 = now();
        CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double t6 = now();
        CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); double t7 = now();
        char *pg = (char *)malloc(n); memset(pg, 2, n); double t8 = now();
        CK(hipMemcpyAsync(d, pg, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double t9 = now();
        CK(hipHostRegister(pg, n, hipHostRegisterDefault)); double t10 = now();
        CK(hipMemcpyAsync(d, pg, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double t11 = now();
        CK(hipHostUnregister(pg)); double t12 = now();
        memcpy(h, pg, n); double t13 = now();
        CK(hipHostFree(h)); double t14 = now();
        CK(hipFree(d)); double t15 = now();
        printf("%5zu MB: hipMalloc %.3f ms, first memset %.3f, hipHostMalloc %.2f (%.3f ms/MB), host memset %.2f, H2D pinned %.3f / %.3f ms (%.1f GB/s), D2H pinned %.3f (%.1f GB/s), "
               "H2D pageable %.2f (%.1f GB/s), hostRegister %.2f, H2D registered %.3f (%.1f GB/s), unregister %.2f, memcpy pg->pinned %.2f (%.1f GB/s), hipHostFree %.2f, hipFree %.3f\n",
               mb, t1 - t0, t2 - t1, t3 - t2, (t3 - t2) / mb, t4 - t3, t5 - t4, t6 - t5, n / (t6 - t5) / 1e6, t7 - t6, n / (t7 - t6) / 1e6,
               t9 - t8, n / (t9 - t8) / 1e6, t10 - t9, t11 - t10, n / (t11 - t10) / 1e6, t12 - t11, t13 - t12, n / (t13 - t12) / 1e6, t14 - t13, t15 - t14);
        free(pg);
    }
    // many small allocations (what a unit's ~70 buffers cost) and whether hipFree synchronises the device
    { std::vector<void *> v(70); double t0 = now(); for (auto &p : v) CK(hipMalloc(&p, 8 << 20)); double t1 = now(); for (auto &p : v) CK(hipFree(p)); double t2 = now();
      printf("70 x hipMalloc(8 MB) %.2f ms, 70 x hipFree %.2f ms\n", t1 - t0, t2 - t1); }
    // hipHostMalloc flavours
    for (unsigned fl : {(unsigned)hipHostMallocDefault, (unsigned)hipHostMallocNonCoherent, (unsigned)hipHostMallocNumaUser}) {
        void *h = nullptr; const size_t n = 512u << 20; double t0 = now(); hipError_t e = hipHostMalloc(&h, n, fl); double t1 = now();
        if (e == hipSuccess) { memset(h, 1, n); double t2 = now(); printf("hipHostMalloc flags %u: 512 MB in %.2f ms, first touch %.2f ms\n", fl, t1 - t0, t2 - t1); CK(hipHostFree(h)); } else { printf("flags %u: %s\n", fl, hipGetErrorString(e)); (void)hipGetLastError(); }
    }
    // parallel pageable -> pinned memcpy
    { const size_t n = 1024u << 20; void *h; CK(hipHostMalloc(&h, n, 0)); char *pg = (char *)malloc(n); memset(pg, 3, n); memset(h, 0, n);
      for (int th : {1, 2, 4, 8}) { double t0 = now(); std::vector<std::thread> T; for (int t = 0; t < th; t++) T.emplace_back([&, t] { memcpy((char *)h + n / th * t, pg + n / th * t, n / th); }); for (auto &x : T) x.join(); double t1 = now();
        printf("memcpy 1 GB pageable->pinned on %d threads: %.2f ms (%.1f GB/s)\n", th, t1 - t0, n / (t1 - t0) / 1e6); }
      CK(hipHostFree(h)); free(pg); }
    return 0;
}
