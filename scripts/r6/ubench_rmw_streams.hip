// Round 6 micro-benchmark: what does the memory system give a kernel that read-modify-writes K dense streams of n = 9.6 M floats at once (the sparse
// shear kernel: 10 compact arrays + 2 read-only streams), as separate arrays, and as ONE array of K-float records?
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 scripts/r6/ubench_rmw_streams.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ptrs { float *p[12]; };
template <int K>
__global__ __launch_bounds__(256) void rmw_soa(Ptrs a, const unsigned *__restrict__ i0, const unsigned *__restrict__ i1, long n)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float add = (float)(__builtin_nontemporal_load(i0 + t) & 1u) + (float)(__builtin_nontemporal_load(i1 + t) & 1u);
    float v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = __builtin_nontemporal_load(a.p[k] + t);
#pragma unroll
    for (int k = 0; k < K; k++) __builtin_nontemporal_store(v[k] + add, a.p[k] + t);
}
// records of 10 floats as 2 x float4 + float2 per lane (40 B, contiguous over the wave)
__global__ __launch_bounds__(256) void rmw_aos10(float *__restrict__ r, const unsigned *__restrict__ i0, const unsigned *__restrict__ i1, long n)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float add = (float)(__builtin_nontemporal_load(i0 + t) & 1u) + (float)(__builtin_nontemporal_load(i1 + t) & 1u);
    float2 *q = (float2 *)(r + 10 * t);
    float2 v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = q[k];
#pragma unroll
    for (int k = 0; k < 5; k++) { v[k].x += add; v[k].y += add; q[k] = v[k]; }
}
template <typename F> float timeit(F f, int reps = 20)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main()
{
    const long n = 9600000;
    Ptrs a; unsigned *i0, *i1; float *rec; float *flush;
    for (int k = 0; k < 12; k++) { CK(hipMalloc(&a.p[k], n * 4)); CK(hipMemset(a.p[k], 0, n * 4)); }
    CK(hipMalloc(&i0, n * 4)); CK(hipMalloc(&i1, n * 4)); CK(hipMemset(i0, 0, n * 4)); CK(hipMemset(i1, 0, n * 4));
    CK(hipMalloc(&rec, n * 40)); CK(hipMemset(rec, 0, n * 40));
    CK(hipMalloc(&flush, (size_t)1 << 30));
    const int grid = (int)((n + 255) / 256);
    // between two timed launches 1 GiB is written elsewhere, as a time step does between two launches of the sparse kernel (nothing stays in the Infinity Cache)
    auto fl = [&] { hipMemsetAsync(flush, 0, (size_t)1 << 30, 0); };
    float tf = timeit([&] { fl(); });
#define RUN(K) { float ms = timeit([&] { fl(); hipLaunchKernelGGL(rmw_soa<K>, dim3(grid), dim3(256), 0, 0, a, i0, i1, n); }) - tf; \
                 printf("separate arrays, K = %2d RMW streams + 2 read streams: %.3f ms  %.2f TB/s\n", K, ms, (8.0 * K + 8.0) * n / ms / 1e9); }
    RUN(1) RUN(2) RUN(4) RUN(6) RUN(8) RUN(10) RUN(12)
    { float ms = timeit([&] { fl(); hipLaunchKernelGGL(rmw_aos10, dim3(grid), dim3(256), 0, 0, rec, i0, i1, n); }) - tf;
      printf("records of 10 floats (one stream)    + 2 read streams: %.3f ms  %.2f TB/s\n", ms, 88.0 * n / ms / 1e9); }
    return 0;
}
