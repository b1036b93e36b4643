// round 6 micro-benchmark: device -> pageable host copy of a large result block (the sensor series of a C3 call: 4.6 GB) into memory nobody has
// touched yet (a fresh numpy array), as ONE hipMemcpy and as T host threads copying T slices on streams of their own.
// build: hipcc -O2 --offload-arch=gfx950 -o ubench_d2h scripts/r6/ubench_d2h_threads.hip -lpthread ; run: ./ubench_d2h [GiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <thread>
#include <vector>
#include <sys/mman.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const size_t n = (size_t)(gib * 1073741824.0) & ~(size_t)4095;
    char *dev = nullptr;
    CK(hipMalloc((void **)&dev, n));
    CK(hipMemset(dev, 1, n));
    CK(hipDeviceSynchronize());
    for (int T : {1, 2, 4, 8, 1, -1, -4}) {
        const bool huge = T < 0;           // negative: the same with madvise(MADV_HUGEPAGE) on the destination first
        if (huge) T = -T;
        char *host = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);      // untouched pages, like np.zeros
        if (host == MAP_FAILED) { printf("mmap failed\n"); return 1; }
        if (huge) printf("madvise(MADV_HUGEPAGE) -> %d; ", madvise(host, n, MADV_HUGEPAGE));
        const double t0 = now();
        std::vector<std::thread> th;
        std::vector<int> rc(T, 0);
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                const size_t a = (n / T * t) & ~(size_t)4095, b = t + 1 == T ? n : (n / T * (t + 1)) & ~(size_t)4095;
                hipStream_t s;
                if (hipSetDevice(0) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { rc[t] = 1; return; }
                if (hipMemcpyAsync(host + a, dev + a, b - a, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc[t] = 1;
                hipStreamDestroy(s);
            });
        for (auto &x : th) x.join();
        const double dt = now() - t0;
        long bad = 0;
        for (size_t i = 0; i < n; i += 1 << 20) bad += host[i] != 1;
        printf("%d thread(s): %.3f s for %.2f GiB = %.1f GB/s%s\n", T, dt, gib, n / dt / 1e9, bad ? "  WRONG DATA" : "");
        munmap(host, n);
    }
    // the same through pinned staging buffers of our own: every thread moves its slice in 16 MB pieces, device -> pinned (two buffers, the next piece
    // in flight) and memcpy pinned -> pageable
    for (int T : {1, 2, 4, 8, 16, -4, -8}) {
        const bool huge = T < 0;
        if (huge) T = -T;
        const size_t PIECE = (size_t)16 << 20;
        std::vector<char *> pin(2 * T, nullptr);
        for (auto &q : pin) CK(hipHostMalloc((void **)&q, PIECE, hipHostMallocDefault));
        char *host = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (host == MAP_FAILED) { printf("mmap failed\n"); return 1; }
        if (huge) printf("madvise(MADV_HUGEPAGE) -> %d; ", madvise(host, n, MADV_HUGEPAGE));
        const double t0 = now();
        std::vector<std::thread> th;
        std::vector<int> rc(T, 0);
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                const size_t a = (n / T * t) & ~(size_t)4095, b = t + 1 == T ? n : (n / T * (t + 1)) & ~(size_t)4095;
                hipStream_t s;
                if (hipSetDevice(0) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { rc[t] = 1; return; }
                hipEvent_t ev[2];
                hipEventCreateWithFlags(&ev[0], hipEventDisableTiming); hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
                const size_t np = (b - a + PIECE - 1) / PIECE;
                auto issue = [&](size_t i) { const size_t o = a + i * PIECE, len = std::min(PIECE, b - o); hipMemcpyAsync(pin[2 * t + (i & 1)], dev + o, len, hipMemcpyDeviceToHost, s); hipEventRecord(ev[i & 1], s); };
                if (np) issue(0);
                for (size_t i = 0; i < np; i++) {
                    if (i + 1 < np) issue(i + 1);
                    hipEventSynchronize(ev[i & 1]);
                    const size_t o = a + i * PIECE, len = std::min(PIECE, b - o);
                    memcpy(host + o, pin[2 * t + (i & 1)], len);
                }
                hipEventDestroy(ev[0]); hipEventDestroy(ev[1]); hipStreamDestroy(s);
            });
        for (auto &x : th) x.join();
        const double dt = now() - t0;
        long bad = 0;
        for (size_t i = 0; i < n; i += 1 << 20) bad += host[i] != 1;
        printf("own staging, %d thread(s): %.3f s for %.2f GiB = %.1f GB/s%s\n", T, dt, gib, n / dt / 1e9, bad ? "  WRONG DATA" : "");
        munmap(host, n);
        for (auto &q : pin) hipHostFree(q);
    }
    hipFree(dev);
    return 0;
}
