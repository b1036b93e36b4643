// Streaming micro-benchmarks for MI355X: what HBM rate do access patterns like the FDTD kernels'
// reach? build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_stream.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename T>
__global__ void copy1(const T *__restrict__ x, T *__restrict__ y, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = x[i];
}
// 5 reads, 4 writes per element (velocity_fluid-like): S, Vx, Vy, Vz, acc -> Vx, Vy, Vz, acc
template <typename T>
__global__ void mix54(const T *__restrict__ s, T *__restrict__ a, T *__restrict__ b, T *__restrict__ c, T *__restrict__ d, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        T v = s[i];
        a[i] = a[i] + v; b[i] = b[i] + v; c[i] = c[i] + v; d[i] = d[i] + v;
    }
}
__device__ inline float4 operator+(float4 p, float4 q) { return make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w); }
__device__ inline float2 operator+(float2 p, float2 q) { return make_float2(p.x + q.x, p.y + q.y); }

// tile pattern: block = 64 x TY rows, marches NZ planes; each wave touches one 256-B row segment per array per plane
template <int TY, int PF>
__global__ void tile54(const float *__restrict__ s, float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, float *__restrict__ d,
                       int N1, int N2, int NZ, int tilesX, int tilesY)
{
    const int t = blockIdx.x;
    const int bx = t % tilesX, by = (t / tilesX) % tilesY, bz = t / (tilesX * tilesY);
    const long pl = (long)N1 * N2;
    const unsigned cij = (by * TY + threadIdx.y) * N1 + bx * 64 + threadIdx.x;
    const long k0 = (long)bz * NZ;
    if (PF) {   // software prefetch one plane ahead
        float v = (s + k0 * pl)[cij], va = (a + k0 * pl)[cij], vb = (b + k0 * pl)[cij], vc = (c + k0 * pl)[cij], vd = (d + k0 * pl)[cij];
        for (int k = 0; k < NZ; k++) {
            const long ko = (k0 + k) * pl, kn = (k + 1 < NZ) ? ko + pl : ko;
            float nv = (s + kn)[cij], na = (a + kn)[cij], nb = (b + kn)[cij], nc = (c + kn)[cij], nd = (d + kn)[cij];
            (a + ko)[cij] = va + v; (b + ko)[cij] = vb + v; (c + ko)[cij] = vc + v; (d + ko)[cij] = vd + v;
            v = nv; va = na; vb = nb; vc = nc; vd = nd;
        }
    } else {
        for (int k = 0; k < NZ; k++) {
            const long ko = (k0 + k) * pl;
            float v = (s + ko)[cij];
            (a + ko)[cij] = (a + ko)[cij] + v; (b + ko)[cij] = (b + ko)[cij] + v; (c + ko)[cij] = (c + ko)[cij] + v; (d + ko)[cij] = (d + ko)[cij] + v;
        }
    }
}

template <typename F>
float timeit(F f, int reps = 10)
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
    const int N1 = 512, N2 = 512, N3 = 512;
    const long n = (long)N1 * N2 * N3;
    float *p[6];
    for (int i = 0; i < 6; i++) { CK(hipMalloc(&p[i], n * 4)); CK(hipMemset(p[i], 0, n * 4)); }
    const int grid = 256 * 8;
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(copy1<float>, dim3(grid), dim3(256), 0, 0, p[0], p[1], n); });
    printf("copy  dword   : %.3f ms  %.2f TB/s\n", ms, 8.0 * n / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(copy1<float2>, dim3(grid), dim3(256), 0, 0, (float2 *)p[0], (float2 *)p[1], n / 2); });
    printf("copy  dwordx2 : %.3f ms  %.2f TB/s\n", ms, 8.0 * n / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(copy1<float4>, dim3(grid), dim3(256), 0, 0, (float4 *)p[0], (float4 *)p[1], n / 4); });
    printf("copy  dwordx4 : %.3f ms  %.2f TB/s\n", ms, 8.0 * n / ms / 1e9);
    for (int g : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        ms = timeit([&] { hipLaunchKernelGGL(mix54<float>, dim3(g), dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], n); });
        printf("mix54 dword   grid %5d: %.3f ms  %.2f TB/s\n", g, ms, 36.0 * n / ms / 1e9);
    }
    ms = timeit([&] { hipLaunchKernelGGL(mix54<float2>, dim3(grid), dim3(256), 0, 0, (float2 *)p[0], (float2 *)p[1], (float2 *)p[2], (float2 *)p[3], (float2 *)p[4], n / 2); });
    printf("mix54 dwordx2 : %.3f ms  %.2f TB/s\n", ms, 36.0 * n / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(mix54<float4>, dim3(grid), dim3(256), 0, 0, (float4 *)p[0], (float4 *)p[1], (float4 *)p[2], (float4 *)p[3], (float4 *)p[4], n / 4); });
    printf("mix54 dwordx4 : %.3f ms  %.2f TB/s\n", ms, 36.0 * n / ms / 1e9);
    {
        const int tilesX = N1 / 64;
        ms = timeit([&] { hipLaunchKernelGGL((tile54<8, 0>), dim3(tilesX * (N2 / 8) * (N3 / 32)), dim3(64, 8), 0, 0, p[0], p[1], p[2], p[3], p[4], N1, N2, 32, tilesX, N2 / 8); });
        printf("tile54 64x8x32 no-prefetch : %.3f ms  %.2f TB/s\n", ms, 36.0 * n / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((tile54<8, 1>), dim3(tilesX * (N2 / 8) * (N3 / 32)), dim3(64, 8), 0, 0, p[0], p[1], p[2], p[3], p[4], N1, N2, 32, tilesX, N2 / 8); });
        printf("tile54 64x8x32 prefetch    : %.3f ms  %.2f TB/s\n", ms, 36.0 * n / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((tile54<4, 1>), dim3(tilesX * (N2 / 4) * (N3 / 32)), dim3(64, 4), 0, 0, p[0], p[1], p[2], p[3], p[4], N1, N2, 32, tilesX, N2 / 4); });
        printf("tile54 64x4x32 prefetch    : %.3f ms  %.2f TB/s\n", ms, 36.0 * n / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((tile54<8, 1>), dim3(tilesX * (N2 / 8) * (N3 / 128)), dim3(64, 8), 0, 0, p[0], p[1], p[2], p[3], p[4], N1, N2, 128, tilesX, N2 / 8); });
        printf("tile54 64x8x128 prefetch   : %.3f ms  %.2f TB/s\n", ms, 36.0 * n / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((tile54<16, 1>), dim3(tilesX * (N2 / 16) * (N3 / 32)), dim3(64, 16), 0, 0, p[0], p[1], p[2], p[3], p[4], N1, N2, 32, tilesX, N2 / 16); });
        printf("tile54 64x16x32 prefetch   : %.3f ms  %.2f TB/s\n", ms, 36.0 * n / ms / 1e9);
    }
    return 0;
}
