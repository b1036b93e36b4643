// Which distances between two lock-step streams are slow? (round 3, follow-up of ubench_layout.hip)
//
// ubench_layout shows: with the engine's marching pattern every placement of the arrays inside ONE allocation runs at the
// "slow" level, separate allocations are fast now and then, and the same virtual addresses can be fast in one draw and slow
// in the next -- the physical placement decides. If the cause is two streams that advance at the same cell offset meeting in
// the same DRAM bank, then inside one physically contiguous allocation the time of a two-stream kernel is a function of the
// DISTANCE between the two arrays only. This program measures that function: array A at offset 0 of one large allocation,
// array B at offset delta, both updated in place at the same cell offset by the engine's tile march (64 x 8 tile, 16 planes,
// eight y-bands, XCD-contiguous remap); delta swept in steps of 2 MiB, of 64 KiB, and of 512 MiB.
//
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_pairmap.hip -o /tmp/ubench_pairmap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int remap_block(int bid, int nblocks)
{
    const int per = nblocks >> 3;
    if (per == 0 || bid >= (per << 3)) return bid;
    return (bid & 7) * per + (bid >> 3);
}
template <typename T> __device__ __forceinline__ T *uni(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float &F4(char *b, unsigned o) { return *(float *)(uni(b) + o); }

// NS streams at base + s*delta (s = 0..NS-1), all updated in place at the same cell offset; plane stride ps bytes
template <int NS>
__global__ __launch_bounds__(512, 8) void k_streams(char *base, long delta, long ps, int N1, const int4 *__restrict__ runs, int nblocks)
{
    const int4 run = runs[remap_block(blockIdx.x, nblocks)];
    const int i = run.x * 64 + threadIdx.x, j = run.y * 8 + threadIdx.y;
    const unsigned o = (unsigned)(j * N1 + i) * 4u;
    float v[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) v[s] = F4(base + s * delta + run.z * ps, o);
    for (int kl = run.z; kl < run.w; kl++) {
        const long ko = kl * ps;
        float n[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) n[s] = kl + 1 < run.w ? F4(base + s * delta + ko + ps, o) : 0.f;
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < NS; s++) acc += v[s];
#pragma unroll
        for (int s = 0; s < NS; s++) F4(base + s * delta + ko, o) = v[s] + acc;
#pragma unroll
        for (int s = 0; s < NS; s++) v[s] = n[s];
    }
}

static std::vector<int4> make_runs(int N1, int N2, int N3, int zrun)
{
    const int tx = N1 / 64, ty = N2 / 8, nch = N3 / zrun;
    std::vector<int4> v;
    for (int e = 0; e < 8; e++) {
        const int y0 = ty * e / 8, y1 = ty * (e + 1) / 8;
        for (int c = 0; c < nch; c++) for (int by = y0; by < y1; by++) for (int bx = 0; bx < tx; bx++) v.push_back(make_int4(bx, by, c * zrun, (c + 1) * zrun));
    }
    return v;
}

template <typename F> static float timeit(F f, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}

// mode "regions <GiB>": one block of that size, A at a few positions, B at every 1 GiB of the block: which PAIRS OF POSITIONS are fast?
static int region_map(size_t gib)
{
    const int N1 = 512, N2 = 512, N3 = 256;
    const long pl = (long)N1 * N2 * 4, arr = pl * N3;
    char *block;
    CK(hipMalloc((void **)&block, gib << 30));
    std::vector<int4> runs = make_runs(N1, N2, N3, 16);
    int4 *dr;
    CK(hipMalloc((void **)&dr, runs.size() * sizeof(int4))); CK(hipMemcpy(dr, runs.data(), runs.size() * sizeof(int4), hipMemcpyHostToDevice));
    const int n = (int)runs.size();
    printf("block of %zu GiB at %p; two arrays of 256 MiB updated in place at the same cell offset; ms per launch\n", gib, (void *)block);
    const long posA[] = {0, 20, 70, 140, 33, 100};
    for (long pa : posA) {
        if ((size_t)pa + 1 > gib) continue;
        CK(hipMemset(block + (pa << 30), 0, arr));
        printf("A at %3ld GiB:", pa);
        for (long pb = 0; pb + 1 <= (long)gib; pb++) {
            if (pb == pa) { printf("   -  "); continue; }
            char *a = block + (pa << 30), *b = block + (pb << 30);
            CK(hipMemset(b, 0, arr));
            const float t = timeit([&] { hipLaunchKernelGGL(k_streams<2>, dim3(n), dim3(64, 8), 0, 0, a, (long)(b - a), pl, N1, dr, n); }, 3);
            printf(" %.3f", t);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 2 && std::string(argv[1]) == "regions") return region_map((size_t)atol(argv[2]));
    const int N1 = 512, N2 = 512, N3 = 256;          // 256 MiB per array: deltas from 256 MiB up
    const long pl = (long)N1 * N2 * 4, arr = pl * N3;
    const size_t total = (size_t)40 << 30;
    char *block;
    CK(hipMalloc((void **)&block, total));
    CK(hipMemset(block, 0, total));
    std::vector<int4> runs = make_runs(N1, N2, N3, 16);
    int4 *dr;
    CK(hipMalloc((void **)&dr, runs.size() * sizeof(int4))); CK(hipMemcpy(dr, runs.data(), runs.size() * sizeof(int4), hipMemcpyHostToDevice));
    const int n = (int)runs.size();
    printf("block at %p, %zu GiB; arrays of %ld MiB (%dx%dx%d float32), in-place update of NS streams at the same cell offset\n", (void *)block, total >> 30, arr >> 20, N1, N2, N3);
    const double gb1 = 2.0 * arr / 1e9;               // bytes moved per stream and launch
    auto run2 = [&](long delta) { return timeit([&] { hipLaunchKernelGGL(k_streams<2>, dim3(n), dim3(64, 8), 0, 0, block, delta, pl, N1, dr, n); }, 3); };
    {
        const float t1 = timeit([&] { hipLaunchKernelGGL(k_streams<1>, dim3(n), dim3(64, 8), 0, 0, block, 0L, pl, N1, dr, n); }, 5);
        printf("one stream alone: %.4f ms  %.0f GB/s\n", t1, gb1 / t1 * 1e3);
    }
    printf("# sweep A: delta = 256 MiB + m * 2 MiB, m = 0..2047\n");
    for (int m = 0; m < 2048; m++) { const long d = arr + (long)m * (2 << 20); const float t = run2(d); printf("A %6ld MiB %.4f ms %.0f GB/s\n", d >> 20, t, 2 * gb1 / t * 1e3); }
    printf("# sweep B: delta = 300 MiB + m * 64 KiB, m = 0..127\n");
    for (int m = 0; m < 128; m++) { const long d = ((long)300 << 20) + (long)m * (64 << 10); const float t = run2(d); printf("B %9ld KiB %.4f ms %.0f GB/s\n", d >> 10, t, 2 * gb1 / t * 1e3); }
    printf("# sweep C: delta = m * 256 MiB, m = 1..150\n");
    for (int m = 1; m <= 150; m++) { const long d = (long)m * arr; const float t = run2(d); printf("C %6ld MiB %.4f ms %.0f GB/s\n", d >> 20, t, 2 * gb1 / t * 1e3); }
    printf("# sweep D: delta = 300 MiB + m * 256 B, m = 0..255 (inside one 64 KiB)\n");
    for (int m = 0; m < 256; m += 4) { const long d = ((long)300 << 20) + (long)m * 256; const float t = run2(d); printf("D %9ld B %.4f ms %.0f GB/s\n", d, t, 2 * gb1 / t * 1e3); }
    // six streams at a regular distance (the fluid kernels' count): which distances are good for all pairs at once?
    printf("# sweep E: six streams, distance = 256 MiB + m * 2 MiB, m = 0..511\n");
    for (int m = 0; m < 512; m++) {
        const long d = arr + (long)m * (2 << 20);
        const float t = timeit([&] { hipLaunchKernelGGL(k_streams<6>, dim3(n), dim3(64, 8), 0, 0, block, d, pl, N1, dr, n); }, 3);
        printf("E %6ld MiB %.4f ms %.0f GB/s\n", d >> 20, t, 6 * gb1 / t * 1e3);
    }
    return 0;
}
