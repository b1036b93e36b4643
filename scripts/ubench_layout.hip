// Layout micro-benchmark for the FDTD marching kernels on MI355X (round 3).
//
// Question: the tiled kernels stream 6-20 per-voxel arrays at the same cell offset, and where hipMalloc puts those arrays
// changes their speed by 8-12 % ("placement lottery"). Does a layout that interleaves the arrays inside
// ONE allocation make the speed deterministic, and is it then the fast or the slow case?
//
// Three proxy kernels with the engine's structure (64 x TY tile per workgroup, z-march of 16 planes, register z-queue, LDS halo
// ring, software pipelined, runs in 8 y-bands with the XCD-contiguous remap):
//   vel   : velocity_fluid-like  reads Szz (+halo), Vx Vy Vz, acc, ids   writes Vx Vy Vz acc
//   str   : stress_fluid-like    reads Vx Vy (+halo) Vz, Szz, Rzz, ids    writes Szz Rzz
//   dense : solid-run-like       reads 15 state arrays + ids + class     writes 12
// over these placements of the 18 arrays (15 float32 state, float32 accumulator, uint16 ids, uint8 class):
//   sep     one hipMalloc per array (what the engine does today), several draws with throw-away allocations in between
//   stride  one allocation, arrays back to back (array-major, regular stride)
//   plane   [k][array][N2][N1]            all arrays of a plane adjacent
//   band    [k][j/8][array][8][N1]        all arrays of an 8-row band of a plane adjacent
//   tile    [k][j/8][i/64][array][8][64]  all arrays of a 64x8 tile-plane adjacent
// each either with all 18 arrays in one block ("1g") or split into the group the fluid kernels touch (V, Szz, Rzz, acc, ids)
// and the rest ("2g").
//
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_layout.hip -o /tmp/ubench_layout ; run: /tmp/ubench_layout [draws]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NA = 18;
// array ids
enum { A_VX, A_VY, A_VZ, A_SZZ, A_RZZ, A_ACC, A_MAT, A_SXX, A_SYY, A_SXY, A_SXZ, A_SYZ, A_RXX, A_RYY, A_RXY, A_RXZ, A_RYZ, A_CLS };
__host__ __device__ constexpr int es_of(int a) { return a == A_MAT ? 2 : (a == A_CLS ? 1 : 4); }
__host__ __device__ constexpr int cls_of(int a) { return a == A_MAT ? 1 : (a == A_CLS ? 2 : 0); }   // element-size class 0: 4 B, 1: 2 B, 2: 1 B

struct Layout {
    char *base[NA];        // address of (plane 0, block 0, element 0) of each array
    long PS[NA];           // bytes from plane k to k+1
    unsigned BB[NA];       // bytes from block b to b+1 (block = 8-row band or 64x8 tile)
    int tileMode;          // 0: block = band of 8 rows, r = (j&7)*N1 + i ; 1: block = 64x8 tile, r = (j&7)*64 + (i&63)
    int N1, N2, tilesX;
};

__device__ __forceinline__ void cell(const Layout &L, int i, int j, unsigned &blk, unsigned &r)
{
    if (L.tileMode) { blk = (unsigned)((j >> 3) * L.tilesX + (i >> 6)); r = (unsigned)((j & 7) * 64 + (i & 63)); }
    else { blk = (unsigned)(j >> 3); r = (unsigned)((j & 7) * L.N1 + i); }
}
template <typename T> __device__ __forceinline__ T *uni(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float &F4(char *b, unsigned o) { return *(float *)(uni(b) + o); }
__device__ __forceinline__ unsigned U2(char *b, unsigned o) { return *(const uint16_t *)(uni(b) + o); }
__device__ __forceinline__ unsigned U1(char *b, unsigned o) { return *(const uint8_t *)(uni(b) + o); }

__device__ __forceinline__ int remap_block(int bid, int nblocks)
{
    const int per = nblocks >> 3;
    if (per == 0 || bid >= (per << 3)) return bid;
    return (bid & 7) * per + (bid >> 3);
}
__device__ __forceinline__ float dplus4(float a, float b, float c, float d) { return 1.125f * (c - b) - (1.0f / 24.0f) * (d - a); }

constexpr int TX = 64;

// halo ring of one array around a TX x TY tile: 4 rows of 64 (tasks 0..255), then 4 columns of TY (tasks 256..256+4*TY)
template <int TY>
__device__ __forceinline__ bool halo_task(int t, int i0, int j0, int N1, int N2, int &lofs, int &gi, int &gj)
{
    constexpr int LW = TX + 4;
    if (t < 256) { const int r = t >> 6, c = t & 63; const int ly = r < 2 ? r : TY + r; lofs = ly * LW + c + 2; gi = i0 + c; gj = j0 - 2 + ly; }
    else if (t < 256 + 4 * TY) { const int u = t - 256, c = u & 3, ly = (u >> 2) + 2; const int lx = c < 2 ? c : TX + c; lofs = ly * LW + lx; gi = i0 - 2 + lx; gj = j0 - 2 + ly; }
    else { lofs = -1; gi = gj = 0; return false; }
    return gi >= 0 && gi < N1 && gj >= 0 && gj < N2;
}

template <int TY>
__global__ __launch_bounds__(TX *TY, (TY == 8 ? 8 : 8)) void k_vel(Layout L, const int4 *__restrict__ runs, int nblocks)
{
    constexpr int LW = TX + 4, LH = TY + 4;
    __shared__ float sS[2][LH * LW];
    const int4 run = runs[remap_block(blockIdx.x, nblocks)];
    const int bx = run.x, by = run.y, kbeg = run.z, kend = run.w;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int i0 = bx * TX, j0 = by * TY, i = i0 + tx, j = j0 + ty;
    unsigned blk, r; cell(L, i, j, blk, r);
    const unsigned o4 = blk * L.BB[A_VX] + r * 4u, o2 = blk * L.BB[A_MAT] + r * 2u;
    const int own = (ty + 2) * LW + tx + 2;
    int lofs, gi, gj;
    const bool hok = halo_task<TY>(tid, i0, j0, L.N1, L.N2, lofs, gi, gj);
    const bool has = lofs >= 0;
    unsigned ho4 = 0;
    if (hok) { unsigned hb, hr; cell(L, gi, gj, hb, hr); ho4 = hb * L.BB[A_SZZ] + hr * 4u; }
    float *lh = &sS[0][has ? lofs : 0];
    const long ps = L.PS[A_VX], ps2 = L.PS[A_MAT];

    char *bS = L.base[A_SZZ] + kbeg * ps;
    float sm1 = F4(bS - ps, o4), s0 = F4(bS, o4), sp1 = F4(bS + ps, o4), sp2 = F4(bS + 2 * ps, o4);
    float vx = F4(L.base[A_VX] + kbeg * ps, o4), vy = F4(L.base[A_VY] + kbeg * ps, o4), vz = F4(L.base[A_VZ] + kbeg * ps, o4);
    float av = F4(L.base[A_ACC] + kbeg * ps, o4);
    unsigned m = U2(L.base[A_MAT] + kbeg * ps2, o2);
    float hv = hok ? F4(bS, ho4) : 0.f;
    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        const long ko = (long)kl * ps;
        sS[b][own] = s0;
        if (has) lh[b * (LH * LW)] = hv;
        const float rr = 1.0f + 1e-9f * (float)m;
        __syncthreads();
        float ns = 0, nh = 0, nvx = 0, nvy = 0, nvz = 0, nav = 0; unsigned nm = 0;
        if (kl + 1 < kend) {
            ns = F4(L.base[A_SZZ] + ko + 3 * ps, o4);
            nvx = F4(L.base[A_VX] + ko + ps, o4); nvy = F4(L.base[A_VY] + ko + ps, o4); nvz = F4(L.base[A_VZ] + ko + ps, o4);
            nav = F4(L.base[A_ACC] + ko + ps, o4);
            nm = U2(L.base[A_MAT] + (long)(kl + 1) * ps2, o2);
            if (hok) nh = F4(L.base[A_SZZ] + ko + ps, ho4);
        }
        const float *p = &sS[b][own];
        const float dx = dplus4(p[-1], s0, p[1], p[2]);
        const float dy = dplus4(p[-LW], s0, p[LW], p[2 * LW]);
        const float dz = dplus4(sm1, s0, sp1, sp2);
        F4(L.base[A_ACC] + ko, o4) = av + s0 * s0;
        F4(L.base[A_VX] + ko, o4) = vx + rr * dx;
        F4(L.base[A_VY] + ko, o4) = vy + rr * dy;
        F4(L.base[A_VZ] + ko, o4) = vz + rr * dz;
        sm1 = s0; s0 = sp1; sp1 = sp2; sp2 = ns;
        hv = nh; vx = nvx; vy = nvy; vz = nvz; av = nav; m = nm;
    }
}

template <int TY>
__global__ __launch_bounds__(TX *TY, 8) void k_str(Layout L, const int4 *__restrict__ runs, int nblocks)
{
    constexpr int LW = TX + 4, LH = TY + 4;
    __shared__ float sV[2][2][LH * LW];
    const int4 run = runs[remap_block(blockIdx.x, nblocks)];
    const int bx = run.x, by = run.y, kbeg = run.z, kend = run.w;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int i0 = bx * TX, j0 = by * TY, i = i0 + tx, j = j0 + ty;
    unsigned blk, r; cell(L, i, j, blk, r);
    const unsigned o4 = blk * L.BB[A_VX] + r * 4u, o2 = blk * L.BB[A_MAT] + r * 2u;
    const int own = (ty + 2) * LW + tx + 2;
    int lofs, gi, gj;
    const bool hok = halo_task<TY>(tid, i0, j0, L.N1, L.N2, lofs, gi, gj);
    const bool has = lofs >= 0;
    const int harr = __builtin_amdgcn_readfirstlane(tid < 256 ? 1 : 0);    // rows: Vy, columns: Vx
    unsigned ho4 = 0;
    if (hok) { unsigned hb, hr; cell(L, gi, gj, hb, hr); ho4 = hb * L.BB[A_VX] + hr * 4u; }
    char *ph = harr ? L.base[A_VY] : L.base[A_VX];
    float *lh = &sV[0][harr][has ? lofs : 0];
    const long ps = L.PS[A_VX], ps2 = L.PS[A_MAT];

    char *bVz = L.base[A_VZ] + kbeg * ps;
    float vx0 = F4(L.base[A_VX] + kbeg * ps, o4), vy0 = F4(L.base[A_VY] + kbeg * ps, o4);
    float vzm2 = F4(bVz - 2 * ps, o4), vzm1 = F4(bVz - ps, o4), vz0 = F4(bVz, o4), vzp1 = F4(bVz + ps, o4);
    float szz = F4(L.base[A_SZZ] + kbeg * ps, o4), rzz = F4(L.base[A_RZZ] + kbeg * ps, o4);
    unsigned m = U2(L.base[A_MAT] + kbeg * ps2, o2);
    float hv = hok ? F4(ph + kbeg * ps, ho4) : 0.f;
    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        const long ko = (long)kl * ps;
        sV[b][0][own] = vx0; sV[b][1][own] = vy0;
        if (has) lh[b * (2 * LH * LW)] = hv;
        const float AP = 1.0f + 1e-9f * (float)m, BP = 0.5f * AP;
        __syncthreads();
        float nvx = 0, nvy = 0, nvz = 0, nh = 0, nszz = 0, nrzz = 0; unsigned nm = 0;
        if (kl + 1 < kend) {
            nvx = F4(L.base[A_VX] + ko + ps, o4); nvy = F4(L.base[A_VY] + ko + ps, o4); nvz = F4(L.base[A_VZ] + ko + 2 * ps, o4);
            nszz = F4(L.base[A_SZZ] + ko + ps, o4); nrzz = F4(L.base[A_RZZ] + ko + ps, o4);
            nm = U2(L.base[A_MAT] + (long)(kl + 1) * ps2, o2);
            if (hok) nh = F4(ph + ko + ps, ho4);
        }
        const float *sx = &sV[b][0][own], *sy = &sV[b][1][own];
        const float dxVx = dplus4(sx[-2], sx[-1], vx0, sx[1]);
        const float dyVy = dplus4(sy[-2 * LW], sy[-LW], vy0, sy[LW]);
        const float dzVz = dplus4(vzm2, vzm1, vz0, vzp1);
        const float div = (dxVx + dyVy) + dzVz;
        const float rn = 0.99f * rzz - BP * div;
        F4(L.base[A_SZZ] + ko, o4) = szz + (AP * div + 0.5f * (rzz + rn));
        F4(L.base[A_RZZ] + ko, o4) = rn;
        vx0 = nvx; vy0 = nvy; vzm2 = vzm1; vzm1 = vz0; vz0 = vzp1; vzp1 = nvz;
        hv = nh; szz = nszz; rzz = nrzz; m = nm;
    }
}

// dense proxy of the solid-run kernels: every state array of the cell read, the 12 stress / memory arrays written
__global__ __launch_bounds__(512, 4) void k_dense(Layout L, const int4 *__restrict__ runs, int nblocks)
{
    const int4 run = runs[remap_block(blockIdx.x, nblocks)];
    const int bx = run.x, by = run.y, kbeg = run.z, kend = run.w;
    const int i = bx * TX + threadIdx.x, j = by * 8 + threadIdx.y;
    unsigned blk, r; cell(L, i, j, blk, r);
    const unsigned o4 = blk * L.BB[A_VX] + r * 4u, o2 = blk * L.BB[A_MAT] + r * 2u, o1 = blk * L.BB[A_CLS] + r;
    const long ps = L.PS[A_VX], ps2 = L.PS[A_MAT], ps1 = L.PS[A_CLS];
    static constexpr int W[12] = {A_SZZ, A_RZZ, A_SXX, A_SYY, A_SXY, A_SXZ, A_SYZ, A_RXX, A_RYY, A_RXY, A_RXZ, A_RYZ};
    for (int kl = kbeg; kl < kend; kl++) {
        const long ko = (long)kl * ps;
        const float v = (F4(L.base[A_VX] + ko, o4) + F4(L.base[A_VY] + ko, o4)) + F4(L.base[A_VZ] + ko, o4);
        const float c = v + 1e-9f * (float)(U2(L.base[A_MAT] + (long)kl * ps2, o2) + U1(L.base[A_CLS] + (long)kl * ps1, o1));
        float x[12];
#pragma unroll
        for (int q = 0; q < 12; q++) x[q] = F4(L.base[W[q]] + ko, o4);
#pragma unroll
        for (int q = 0; q < 12; q++) F4(L.base[W[q]] + ko, o4) = x[q] + c;
    }
}

// two streams updated in place at the same cell offset (scripts/ubench_pairmap.hip): 7-8 % slower when both lie in the same
// "colour" of physical memory
__global__ __launch_bounds__(512, 8) void k_pair(char *a, char *b, long ps, int N1, const int4 *__restrict__ runs, int nblocks)
{
    const int4 run = runs[remap_block(blockIdx.x, nblocks)];
    const int i = run.x * 64 + threadIdx.x, j = run.y * 8 + threadIdx.y;
    const unsigned o = (unsigned)(j * N1 + i) * 4u;
    float va = F4(a + run.z * ps, o), vb = F4(b + run.z * ps, o);
    for (int kl = run.z; kl < run.w; kl++) {
        const long ko = kl * ps;
        float na = 0, nb = 0;
        if (kl + 1 < run.w) { na = F4(a + ko + ps, o); nb = F4(b + ko + ps, o); }
        F4(a + ko, o) = va + vb; F4(b + ko, o) = vb + va;
        va = na; vb = nb;
    }
}

// ------------------------------------------------------------------------------------------------
struct Set {
    std::vector<void *> allocs;
    Layout L;
    std::string name;
};
static const int N1 = 512, N2 = 512, N3 = 512, G = 2;       // G ghost planes each side
static const bool fluidGroup[NA] = {1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// mode: 0 sep, 1 stride, 2 plane, 3 band, 4 tile ; groups: 1 or 2 (ignored for sep, stride)
// pad: bytes added to every plane stride (per float32 array; halved / quartered for the uint16 / uint8 arrays in sep and stride
// modes so that the same cell keeps the same relative offset); skew: array a starts a*skew bytes later (sep, stride: inside
// its allocation / slot)
static Set make_set(int mode, int groups, long pad, long skew, long strideMiB = 0)
{
    Set s;
    Layout &L = s.L;
    L.N1 = N1; L.N2 = N2; L.tilesX = N1 / 64; L.tileMode = mode == 4;
    const long pl = (long)N1 * N2, nplanes = N3 + 2 * G;
    static const char *mn[] = {"sep", "stride", "plane", "band", "tile"};
    s.name = std::string(mn[mode]) + (mode >= 2 ? (groups == 2 ? "-2g" : "-1g") : "");
    if (pad) s.name += "+p" + std::to_string(pad);
    if (skew) s.name += "+s" + std::to_string(skew);
    if (strideMiB) s.name += "@" + std::to_string(strideMiB) + "M";
    if (mode == 0 || mode == 1) {
        char *block = nullptr; size_t off = 0;
        if (mode == 1) {
            size_t tot = 0;
            for (int a = 0; a < NA; a++) tot += strideMiB ? (size_t)strideMiB << 20 : round_up((pl * es_of(a) + pad * es_of(a) / 4) * nplanes + a * skew, 2 << 20);
            CK(hipMalloc((void **)&block, tot)); CK(hipMemset(block, 0, tot)); s.allocs.push_back(block);
        }
        for (int a = 0; a < NA; a++) {
            const long psa = pl * es_of(a) + pad * es_of(a) / 4;
            const size_t bytes = psa * nplanes + a * skew;
            char *p;
            if (mode == 0) { CK(hipMalloc((void **)&p, bytes)); CK(hipMemset(p, 0, bytes)); s.allocs.push_back(p); }
            else { p = block + off; off += strideMiB ? (size_t)strideMiB << 20 : round_up(bytes, 2 << 20); }
            if (mode == 0 && getenv("UB_VERBOSE")) printf("   array %2d at %p (%% 1 GiB = %lu MiB)\n", a, (void *)p, (unsigned long)(((size_t)p >> 20) & 1023));
            L.base[a] = p + G * psa + a * skew;
            L.PS[a] = psa;
            L.BB[a] = 8u * N1 * es_of(a);
        }
        return s;
    }
    for (int g = 0; g < groups; g++) {
        // unit = elements of one array inside a block: plane: N1*N2 (one block per plane), band: 8*N1, tile: 512
        const long unit = mode == 2 ? pl : (mode == 3 ? 8L * N1 : 512L);
        const long blocksPerPlane = mode == 2 ? 1 : (mode == 3 ? N2 / 8 : (N2 / 8) * (N1 / 64));
        size_t bb = 0;
        for (int a = 0; a < NA; a++) if (groups == 1 || fluidGroup[a] == (g == 0)) bb += unit * es_of(a);
        const size_t planeBytes = bb * blocksPerPlane + pad, tot = planeBytes * nplanes;
        char *block; CK(hipMalloc((void **)&block, tot)); CK(hipMemset(block, 0, tot)); s.allocs.push_back(block);
        size_t aofs = 0;
        for (int a = 0; a < NA; a++) {
            if (!(groups == 1 || fluidGroup[a] == (g == 0))) continue;
            L.base[a] = block + G * planeBytes + aofs;
            L.PS[a] = (long)planeBytes;
            // plane mode: the "block" of the kernels' address formula is still the 8-row band, inside the array's own plane
            L.BB[a] = mode == 2 ? 8u * N1 * es_of(a) : (unsigned)bb;
            aofs += unit * es_of(a);
        }
    }
    return s;
}
static void free_set(Set &s) { for (void *p : s.allocs) hipFree(p); s.allocs.clear(); }

static std::vector<int4> make_runs(int TY, int zrun)
{
    const int tx = N1 / 64, ty = N2 / TY, nch = N3 / zrun;
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
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}

// "colours" experiment: a pool of array-sized allocations is sorted into classes by the pair test (two arrays of one class
// are slow together), then the proxy kernels run on sets drawn from ONE class and on sets spread evenly over the classes.
static int colours(int pool)
{
    const long pl = (long)N1 * N2, nplanes = N3 + 2 * G;
    const size_t bytes = (size_t)pl * nplanes * 4;
    std::vector<char *> buf(pool);
    for (int q = 0; q < pool; q++) { CK(hipMalloc((void **)&buf[q], bytes)); CK(hipMemset(buf[q], 0, bytes)); }
    std::vector<int4> rp = make_runs(8, 16), r16 = make_runs(16, 16);
    rp.resize(rp.size() / 2);            // the pair test marches half of the volume: 0.2 ms
    int4 *dp, *d8;
    CK(hipMalloc((void **)&dp, rp.size() * sizeof(int4))); CK(hipMemcpy(dp, rp.data(), rp.size() * sizeof(int4), hipMemcpyHostToDevice));
    std::vector<int4> r8 = make_runs(8, 16);
    CK(hipMalloc((void **)&d8, r8.size() * sizeof(int4))); CK(hipMemcpy(d8, r8.data(), r8.size() * sizeof(int4), hipMemcpyHostToDevice));
    const int np = (int)rp.size(), n8 = (int)r8.size();
    auto pair = [&](int x, int y) { return timeit([&] { hipLaunchKernelGGL(k_pair, dim3(np), dim3(64, 8), 0, 0, buf[x] + G * pl * 4, buf[y] + G * pl * 4, pl * 4, N1, dp, np); }, 4); };
    // classes: compare with one representative per class; "same" = the slow level. Levels from the first column.
    std::vector<float> t0(pool, 0.f);
    float lo = 1e9f, hi = 0.f;
    for (int q = 1; q < pool; q++) { t0[q] = pair(0, q); lo = std::min(lo, t0[q]); hi = std::max(hi, t0[q]); }
    {   // the raw pair times: how many levels are there?
        std::vector<float> srt(t0.begin() + 1, t0.end());
        std::sort(srt.begin(), srt.end());
        printf("pair times against allocation 0, sorted:");
        for (size_t q = 0; q < srt.size(); q++) printf(" %.4f", srt[q]);
        printf("\n");
        // the same allocation against itself, second stream half an array further on (same physical region for sure)
        std::vector<int4> rh;
        for (const int4 &r : make_runs(8, 16)) if (r.w <= N3 / 2) rh.push_back(r);
        int4 *dh;
        CK(hipMalloc((void **)&dh, rh.size() * sizeof(int4))); CK(hipMemcpy(dh, rh.data(), rh.size() * sizeof(int4), hipMemcpyHostToDevice));
        const int nh = (int)rh.size();
        for (int q = 0; q < 12; q++) {
            const float t = timeit([&] { hipLaunchKernelGGL(k_pair, dim3(nh), dim3(64, 8), 0, 0, buf[q] + G * pl * 4, buf[q] + G * pl * 4 + (N3 / 2) * pl * 4, pl * 4, N1, dh, nh); }, 4);
            const float u = timeit([&] { hipLaunchKernelGGL(k_pair, dim3(nh), dim3(64, 8), 0, 0, buf[0] + G * pl * 4, buf[q] + G * pl * 4, pl * 4, N1, dh, nh); }, 4);
            printf("allocation %d: lower half with its own upper half %.4f ms; lower half with allocation 0's lower half %.4f ms (half-volume march)\n", q, t, u);
        }
    }
    const float thr = 0.5f * (lo + hi);
    printf("pair test against allocation 0: fast level %.4f ms, slow level %.4f ms (threshold %.4f)\n", lo, hi, thr);
    std::vector<int> colour(pool, -1), rep;
    colour[0] = 0; rep.push_back(0);
    for (int q = 1; q < pool; q++) {
        for (size_t c = 0; c < rep.size() && colour[q] < 0; c++) {
            const float t = rep[c] == 0 ? t0[q] : pair(rep[c], q);
            if (t > thr) colour[q] = (int)c;
        }
        if (colour[q] < 0) { colour[q] = (int)rep.size(); rep.push_back(q); }
    }
    printf("colour of each allocation (in allocation order): ");
    for (int q = 0; q < pool; q++) printf("%d", colour[q]);
    printf("\n");
    const int nc = (int)rep.size();
    std::vector<std::vector<int>> byc(nc);
    for (int q = 0; q < pool; q++) byc[colour[q]].push_back(q);
    for (int c = 0; c < nc; c++) printf("  colour %d: %zu allocations\n", c, byc[c].size());
    // sets: the 18 arrays of the layout benchmark on pool buffers (the uint16 / uint8 arrays use the front of theirs)
    auto run_set = [&](const char *name, const std::vector<int> &pick) {
        Layout L; L.N1 = N1; L.N2 = N2; L.tilesX = N1 / 64; L.tileMode = 0;
        for (int a = 0; a < NA; a++) { L.base[a] = buf[pick[a]] + G * pl * es_of(a); L.PS[a] = pl * es_of(a); L.BB[a] = 8u * N1 * es_of(a); }
        const float v8 = timeit([&] { hipLaunchKernelGGL(k_vel<8>, dim3(n8), dim3(64, 8), 0, 0, L, d8, n8); }, 10);
        const float s8 = timeit([&] { hipLaunchKernelGGL(k_str<8>, dim3(n8), dim3(64, 8), 0, 0, L, d8, n8); }, 10);
        const float dn = timeit([&] { hipLaunchKernelGGL(k_dense, dim3(n8), dim3(64, 8), 0, 0, L, d8, n8); }, 10);
        printf("%-34s vel8 %.3f  str8 %.3f  dense %.3f   colours:", name, v8, s8, dn);
        for (int a = 0; a < NA; a++) printf("%d", colour[pick[a]]);
        printf("\n");
        fflush(stdout);
    };
    if (nc >= 2 && byc[0].size() >= NA && byc[1].size() >= NA) {
        // how many arrays of the other colour does it take? arrays Vx Vy Vz Szz Rzz acc ids Sxx ... in colour 0, except the listed ones
        auto mixed = [&](const char *name, std::vector<int> other) {
            std::vector<int> pick(NA);
            size_t u0 = 0, u1 = 0;
            for (int a = 0; a < NA; a++) pick[a] = std::find(other.begin(), other.end(), a) != other.end() ? byc[1][u1++] : byc[0][u0++];
            run_set(name, pick);
        };
        mixed("colour 1: Szz", {A_SZZ});
        mixed("colour 1: Vx", {A_VX});
        mixed("colour 1: ids", {A_MAT});
        mixed("colour 1: acc, Rzz", {A_ACC, A_RZZ});
        mixed("colour 1: Szz, acc, Rzz", {A_SZZ, A_ACC, A_RZZ});
        mixed("colour 1: Vx, Vy", {A_VX, A_VY});
        mixed("colour 1: Vx, Vy, Vz", {A_VX, A_VY, A_VZ});
        mixed("colour 1: Vx, Vy, Vz, ids", {A_VX, A_VY, A_VZ, A_MAT});
        mixed("colour 1: Vz, Szz", {A_VZ, A_SZZ});
        mixed("colour 1: Vy, Szz, acc", {A_VY, A_SZZ, A_ACC});
        mixed("colour 1: the 11 solid-only arrays", {A_SXX, A_SYY, A_SXY, A_SXZ, A_SYZ, A_RXX, A_RYY, A_RXY, A_RXZ, A_RYZ, A_CLS});
        mixed("colour 1: every second solid-only", {A_SXX, A_SXY, A_SYZ, A_RYY, A_RXZ, A_CLS});
    }
    {   // are large allocations of one colour throughout? four blocks of 8 GiB, every 512 MiB of them against one representative per colour
        const size_t big = (size_t)8 << 30;
        for (int b = 0; b < 4; b++) {
            char *blk;
            if (hipMalloc((void **)&blk, big) != hipSuccess) { (void)hipGetLastError(); break; }
            CK(hipMemset(blk, 0, big));
            printf("8 GiB allocation %d at %p, colour of every 512 MiB: ", b, (void *)blk);
            for (size_t off = 0; off + bytes <= big; off += (size_t)512 << 20) {
                int found = -1;
                for (int c = 0; c < nc && found < 0; c++) {
                    char *x = buf[rep[c]] + G * pl * 4, *y = blk + off + G * pl * 4;
                    const float t = timeit([&] { hipLaunchKernelGGL(k_pair, dim3(np), dim3(64, 8), 0, 0, x, y, pl * 4, N1, dp, np); }, 4);
                    if (t > thr) found = c;
                }
                printf("%c", found < 0 ? '?' : '0' + found);
            }
            printf("\n");
            // deliberately not freed: the next block comes from elsewhere
        }
    }
    for (int rep_ = 0; rep_ < 2; rep_++) {
        for (int c = 0; c < nc; c++) {
            if ((int)byc[c].size() < NA) continue;
            std::vector<int> pick(byc[c].begin() + (rep_ ? (int)byc[c].size() - NA : 0), byc[c].begin() + (rep_ ? (int)byc[c].size() : NA));
            char nm[64]; snprintf(nm, sizeof nm, "all arrays in colour %d (%d)", c, rep_);
            run_set(nm, pick);
        }
        if (nc >= 2) {
            // round robin over the colours in the array order Vx Vy Vz Szz Rzz acc ids | Sxx Syy Sxy Sxz Syz Rxx Ryy Rxy Rxz Ryz class:
            // both fluid kernels then see their six streams spread evenly
            std::vector<size_t> used(nc, rep_ ? 3 : 0);
            std::vector<int> pick(NA);
            bool ok = true;
            for (int a = 0; a < NA; a++) { const int c = a % nc; if (used[c] >= byc[c].size()) { ok = false; break; } pick[a] = byc[c][used[c]++]; }
            if (ok) { char nm[64]; snprintf(nm, sizeof nm, "round robin over %d colours (%d)", nc, rep_); run_set(nm, pick); }
            // two colours only
            std::fill(used.begin(), used.end(), rep_ ? 3 : 0); ok = true;
            for (int a = 0; a < NA; a++) { const int c = a % 2; if (used[c] >= byc[c].size()) { ok = false; break; } pick[a] = byc[c][used[c]++]; }
            if (ok) { char nm[64]; snprintf(nm, sizeof nm, "alternating colours 0 / 1 (%d)", rep_); run_set(nm, pick); }
        }
        {   // allocation order (what separate hipMallocs give)
            std::vector<int> pick(NA);
            for (int a = 0; a < NA; a++) pick[a] = (rep_ ? pool - NA : 0) + a;
            char nm[64]; snprintf(nm, sizeof nm, "allocation order (%d)", rep_);
            run_set(nm, pick);
        }
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 2 && std::string(argv[1]) == "colours") return colours(atoi(argv[2]));
    const int draws = argc > 1 ? atoi(argv[1]) : 6;
    const int reps = 12;
    std::vector<int4> r8 = make_runs(8, 16), r16 = make_runs(16, 16);
    int4 *d8, *d16;
    CK(hipMalloc((void **)&d8, r8.size() * sizeof(int4))); CK(hipMemcpy(d8, r8.data(), r8.size() * sizeof(int4), hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d16, r16.size() * sizeof(int4))); CK(hipMemcpy(d16, r16.data(), r16.size() * sizeof(int4), hipMemcpyHostToDevice));
    const double vox = (double)N1 * N2 * N3;
    printf("grid %dx%dx%d, %d draws per placement, %d reps; ms per launch (GB/s on algorithmic bytes: vel 38 B, str 30 B, dense 111 B per cell)\n", N1, N2, N3, draws, reps);
    printf("%-22s %4s | %8s %8s | %8s %8s | %8s %8s | %8s\n", "placement", "draw", "vel8", "vel16", "str8", "str16", "dense", "", "sum8");
    struct Cfg { int mode, groups; long pad, skew, stride; };
    std::vector<Cfg> cfgs;
    for (int a = 2; a < argc; a++) { Cfg c = {0, 1, 0, 0, 0}; sscanf(argv[a], "%d:%d:%ld:%ld:%ld", &c.mode, &c.groups, &c.pad, &c.skew, &c.stride); cfgs.push_back(c); }
    if (cfgs.empty()) cfgs = {{0, 1, 0, 0, 0}, {1, 1, 0, 0, 0}, {2, 1, 0, 0, 0}, {2, 2, 0, 0, 0}, {3, 1, 0, 0, 0}, {3, 2, 0, 0, 0}, {4, 1, 0, 0, 0}, {4, 2, 0, 0, 0}};
    srand(12345);
    for (const Cfg &c : cfgs) {
        std::vector<float> sums;
        for (int dr = 0; dr < draws; dr++) {
            // perturb the allocator between draws: a throw-away block of varying size stays allocated during the draw
            void *spacer = nullptr;
            const size_t sp = (size_t)(1 + rand() % 200) << 21;
            CK(hipMalloc(&spacer, sp));
            Set s = make_set(c.mode, c.groups, c.pad, c.skew, c.stride);
            const int n8 = (int)r8.size(), n16 = (int)r16.size();
            const float v8 = timeit([&] { hipLaunchKernelGGL(k_vel<8>, dim3(n8), dim3(64, 8), 0, 0, s.L, d8, n8); }, reps);
            const float v16 = timeit([&] { hipLaunchKernelGGL(k_vel<16>, dim3(n16), dim3(64, 16), 0, 0, s.L, d16, n16); }, reps);
            const float s8 = timeit([&] { hipLaunchKernelGGL(k_str<8>, dim3(n8), dim3(64, 8), 0, 0, s.L, d8, n8); }, reps);
            const float s16 = timeit([&] { hipLaunchKernelGGL(k_str<16>, dim3(n16), dim3(64, 16), 0, 0, s.L, d16, n16); }, reps);
            const float dn = timeit([&] { hipLaunchKernelGGL(k_dense, dim3(n8), dim3(64, 8), 0, 0, s.L, d8, n8); }, reps);
            CK(hipGetLastError());
            printf("%-22s %4d | %8.3f %8.3f | %8.3f %8.3f | %8.3f %8s | %8.3f   (vel8 %.0f, str8 %.0f, dense %.0f GB/s)\n", s.name.c_str(), dr, v8, v16, s8, s16, dn, "",
                   v8 + s8, 38.0 * vox / v8 / 1e6, 30.0 * vox / s8 / 1e6, 111.0 * vox / dn / 1e6);
            fflush(stdout);
            sums.push_back(v8 + s8);
            free_set(s);
            hipFree(spacer);
        }
        std::sort(sums.begin(), sums.end());
        printf("  => vel8+str8 over the draws: min %.3f  median %.3f  max %.3f  spread %.1f %%\n\n", sums.front(), sums[sums.size() / 2], sums.back(),
               100.0 * (sums.back() - sums.front()) / sums.front());
    }
    return 0;
}
