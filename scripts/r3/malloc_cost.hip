// what does walking through device memory cost? hipMalloc / hipFree of blocks of 0.5 ... 32 GiB (round 3, placement search)
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
int main()
{
    for (size_t gib2 : {1, 2, 8, 32, 64}) {
        const size_t bytes = gib2 << 29;
        auto t0 = std::chrono::steady_clock::now();
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { printf("%.1f GiB: hipMalloc failed\n", bytes / 1073741824.0); continue; }
        auto t1 = std::chrono::steady_clock::now();
        hipMemset(p, 0, bytes); hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        hipFree(p);
        auto t3 = std::chrono::steady_clock::now();
        printf("%5.1f GiB: hipMalloc %.1f ms, memset %.1f ms, hipFree %.1f ms\n", bytes / 1073741824.0, std::chrono::duration<double, std::milli>(t1 - t0).count(),
               std::chrono::duration<double, std::milli>(t2 - t1).count(), std::chrono::duration<double, std::milli>(t3 - t2).count());
    }
    return 0;
}
