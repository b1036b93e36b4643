// round 6: how long does hipMalloc take right after many GiB were freed? (the 5-second sensors() calls: profiles/r6/d2h_into_untouched_memory.txt)
// build: hipcc -O2 --offload-arch=gfx950 -o ubench_maf scripts/r6/ubench_malloc_after_free.hip ; run: ./ubench_maf [GiB freed]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const int gib = argc > 1 ? atoi(argv[1]) : 48;
    for (int touch = 0; touch < 2; touch++) {
        std::vector<void *> held;
        const size_t chunk = argc > 2 ? (size_t)atol(argv[2]) << 20 : (size_t)4 << 30;
        for (size_t got = 0; got < ((size_t)gib << 30); got += chunk) { void *p = nullptr; if (hipMalloc(&p, chunk) != hipSuccess) break; held.push_back(p); if (touch) hipMemsetAsync(p, 1, chunk, 0); }
        hipDeviceSynchronize();
        double t0 = now();
        for (void *p : held) hipFree(p);
        const double tFree = now() - t0;
        if (argc > 3) { void *w = nullptr; hipMalloc(&w, (size_t)1 << 30); const double tw = now(); while (now() - tw < atof(argv[3])) { hipMemsetAsync(w, 2, (size_t)1 << 30, 0); hipStreamSynchronize(0); } hipFree(w); }      // device work for a while, as a step loop would
        for (double sz : {0.5, 5.0, 0.5}) {
            void *p = nullptr;
            t0 = now();
            const hipError_t e = hipMalloc(&p, (size_t)(sz * 1073741824.0));
            const double tm = now() - t0;
            t0 = now();
            hipMemset(p, 0, 1 << 20); hipDeviceSynchronize();
            printf("%s %zu GiB freed in %.3f s; then hipMalloc(%.1f GiB): %.3f s (%s), first use %.3f s\n", touch ? "touched" : "untouched", held.size() * (chunk >> 20) / 1024, tFree, sz, tm, hipGetErrorString(e), now() - t0);
            hipFree(p);
        }
    }
    return 0;
}
