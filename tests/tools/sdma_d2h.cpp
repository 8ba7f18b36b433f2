// HBM -> registered host memory through the HSA runtime's asynchronous copy (the SDMA engines), next to what hipMemcpyAsync does for the same
// buffers (rocprofv3 shows the runtime's own copy KERNEL there).  hipcc -O2 -o build/sdma_d2h tests/tools/sdma_d2h.cpp -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Agents { hsa_agent_t cpu{}; bool have_cpu = false; std::vector<hsa_agent_t> gpu; std::vector<uint32_t> bdf; };
static hsa_status_t on_agent(hsa_agent_t a, void *p) {
    Agents *A = (Agents *)p; hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_CPU && !A->have_cpu) { A->cpu = a; A->have_cpu = true; }
    if (t == HSA_DEVICE_TYPE_GPU) { uint32_t b = 0; hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &b); A->gpu.push_back(a); A->bdf.push_back(b); }
    return HSA_STATUS_SUCCESS;
}
int main() {
    const size_t n = (size_t)256 << 20;
    void *d = nullptr; OK(hipMalloc(&d, n)); OK(hipMemset(d, 0x5A, n)); OK(hipDeviceSynchronize());
    char *h = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(h, n, MADV_HUGEPAGE);
    OK(hipHostRegister(h, n, hipHostRegisterPortable | hipHostRegisterMapped));
    void *hd = nullptr; OK(hipHostGetDevicePointer(&hd, h, 0));
    printf("host %p device-visible %p\n", (void *)h, hd);
    hipStream_t st; OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int r = 0; r < 3; r++) { memset(h, 0, 4096); double t = now(); OK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st)); OK(hipStreamSynchronize(st)); t = now() - t; printf("hipMemcpyAsync D2H: %.2f ms = %.1f GB/s (first byte %02x)\n", t, n / t / 1e6, (unsigned char)h[0]); }
    if (hsa_init() != HSA_STATUS_SUCCESS) { printf("hsa_init failed\n"); return 1; }
    Agents A; hsa_iterate_agents(on_agent, &A);
    printf("agents: cpu %d, gpus %zu (bdf of gpu 0: %04x)\n", (int)A.have_cpu, A.gpu.size(), A.gpu.empty() ? 0 : A.bdf[0]);
    if (!A.have_cpu || A.gpu.empty()) return 1;
    hsa_signal_t sig; if (hsa_signal_create(1, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) { printf("signal\n"); return 1; }
    for (int which = 0; which < 2; which++) {
        void *dst = which == 0 ? (void *)h : hd;
        for (int r = 0; r < 3; r++) {
            memset(h, 0, n > 4096 ? 4096 : n); h[n - 1] = 0;
            hsa_signal_store_relaxed(sig, 1);
            double t = now();
            hsa_status_t s = hsa_amd_memory_async_copy(dst, A.cpu, d, A.gpu[0], n, 0, nullptr, sig);
            if (s != HSA_STATUS_SUCCESS) { const char *m = nullptr; hsa_status_string(s, &m); printf("hsa_amd_memory_async_copy(dst = %s pointer): %s\n", which ? "device-visible" : "host", m ? m : "?"); break; }
            const hsa_signal_value_t v = hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
            t = now() - t;
            printf("hsa async copy D2H (dst = %s pointer): %.2f ms = %.1f GB/s, signal %ld, first/last byte %02x %02x\n", which ? "device-visible" : "host", t, n / t / 1e6, (long)v, (unsigned char)h[0], (unsigned char)h[n - 1]);
        }
    }
    return 0;
}
