// Fused time step of fluid runs (kernelVariant 4): the stress and the velocity half-step of a tile in ONE pass, so that
// V, Szz (and Rzz) are read once and written once per time step instead of V read twice and Szz three times.
//
// Replaces, for the runs it takes, the two per-step device kernels of the reference's solver backends (package
// BabelViscoFDTD, absent from /root/reference; call site BabelIntegrationBASE.py:2338).
//
// Geometry (DESIGN.md "Kernels"): a workgroup of 512 threads owns 64 x 24 OUTPUT cells per plane (three 64 x 8 tiles of
// the classification grid) and marches a z-run. The velocity update of a cell needs the NEW stress at x-1 .. x+2, y-1 ..
// y+2, z-1 .. z+2 (forward differences), so the workgroup recomputes the new stress on the REGION = outputs grown by one
// cell below and two above in x and y: 67 x 27 = 1809 cells, four per thread in a fixed assignment (rows r, r+7, r+14, r+21 of
// one column: 469 of the 512 threads hold cells),
// and on three extra planes per z-run (one below, two above). Ring overhead (67 x 27) / (64 x 24) = 1.18 against
// (68 x 12) / (64 x 8) = 1.59 of the 64 x 8 form this kernel replaces.
//   per plane p:  A  stage Vx(p) (70 x 27) and Vy(p) (67 x 30) in LDS                           | barrier
//                 B  loads of plane p+1; new stress of the region cells from the OLD fields -> LDS tile, own z queue
//                                                                                              | barrier
//                 C  outputs: Vx, Vy of plane p from the LDS tile of the new stress; Vz of plane p-2 from the thread's
//                    own queue of new stresses (p-3 .. p); Pressure RMS / peak of plane p
// The z neighbours (Vz p-2 .. p+1, new stress p-3 .. p) stay in registers; every LDS tile is single-buffered (two
// barriers per plane separate its writes from its reads).
// The old fields must survive the step for the neighbours' recomputation: this variant keeps two copies of V, Szz, Rzz
// (kernels read d.X, write d.XW; swapped after the step).
// Eligible runs (bfd_api.hip, build_tile_lists): FLUID sub-tiles (no solid cell within 2 cells), no absorbing-layer cell
// within 2 cells, not the first / last sub-tile of the slab, velocity-type sources. Then every recomputed cell follows the
// same arithmetic as its owner computes for it and every index stays inside the domain. UNI: one material, no reflector
// in the grown region (coefficients are scalars); otherwise ids are loaded per region cell (runs with a reflector voxel in the
// grown region are left to the two-kernel path, classify_tiles bit8). LOSSY: some cell of the GROWN region relaxes (bit7). Same operation order as stress_fluid_body / velocity_fluid_body: bit-identical.
#include "bfd_internal.h"
#include "bfd_device.h"

namespace {

constexpr int FT_X = 64, FT_Y = BFD_FUSED_ROWS;          // output cells per plane
constexpr int FR_W = FT_X + 3, FR_H = FT_Y + 3;         // region of the new stress: x in [i0-1, i0+65], y in [j0-1, j0+FT_Y+1]
constexpr int FP = FR_W + 3;                            // LDS pitch 70: column t <-> x = i0-3+t (Vx needs x-2 .. x+1 of every region cell)
constexpr int FS_N = FR_H * FP;                         // Vx, new-stress and 1/rho tiles: region rows
constexpr int FVY_N = (FR_H + 3) * FP;                  // Vy tile: rows u <-> y = j0-3+u
#ifndef FUSED_CPT
#define FUSED_CPT 4
#endif
constexpr int NT = 512, CPT = FUSED_CPT;       // region cells per thread
constexpr int RSTEP = (FR_H + CPT - 1) / CPT;           // 7: thread (r, col) owns the region cells (r + 7 c, col), c = 0 .. 3
constexpr int NCELLT = RSTEP * FR_W;                    // 469 threads hold cells
static_assert(NCELLT <= NT && RSTEP * CPT >= FR_H, "four region cells per thread");
constexpr int NTX = 3 * FR_H;                           // extra Vx columns t = 0, 1, 69: 81 tasks (waves 0-1)
constexpr int NTY = 3 * FR_W;                           // extra Vy rows u = 0, 1, 29: 201 tasks (waves 2-5)
static_assert(NTX <= 128 && NTY <= 256, "halo tasks fit their waves");
#ifndef FUSED_WAVES_PER_SIMD
#define FUSED_WAVES_PER_SIMD 4
#endif
constexpr int FUSED_TAB_MAX = BFD_FUSED_MAX_MATERIALS;  // materials whose AP, BP, 1/rho fit the LDS table of the multi-material flavour

// Accesses of this kernel are GLOBAL instructions (wave-uniform base in SGPRs + 32-bit byte offset), not the FLAT ones of
// bfd_device.h: a flat load counts on lgkmcnt as well, so the first wait for an LDS read would also wait for every load of
// the next plane just issued -- the prefetch of this kernel has to stay in flight across its LDS phases.
#define BFD_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ BFD_GLOBAL T *guni(const T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (BFD_GLOBAL T *)(((unsigned long long)hi << 32) | lo);
}
// keeps the 32-bit offset a value of the block that uses it (hoisted out of the loop its zero-extension becomes a 64-bit VGPR address)
__device__ __forceinline__ unsigned pin(unsigned v) { asm("" : "+v"(v)); return v; }
__device__ __forceinline__ float GL4(const float *base, unsigned ofs) { return *(BFD_GLOBAL const float *)((BFD_GLOBAL const char *)guni(base) + pin(ofs)); }
__device__ __forceinline__ float GLNT(const float *base, unsigned ofs) { return __builtin_nontemporal_load((BFD_GLOBAL const float *)((BFD_GLOBAL const char *)guni(base) + pin(ofs))); }
__device__ __forceinline__ unsigned GL2(const uint16_t *base, unsigned ofs) { return *(BFD_GLOBAL const uint16_t *)((BFD_GLOBAL const char *)guni(base) + pin(ofs)); }
__device__ __forceinline__ void GS4(float *base, unsigned ofs, float v) { *(BFD_GLOBAL float *)((BFD_GLOBAL char *)guni(base) + pin(ofs)) = v; }
// Stores are plain in this kernel: non-temporal ones wrote 1.98 GB per launch where 1.78 are needed and cost 7 % (C1 512^3, same
// box: 0.953 -> 0.889 ms; profiles/r4/fused_experiments.txt). -DFUSED_EXP_NT_STORES builds the non-temporal ones.
#ifndef FUSED_EXP_NT_STORES
__device__ __forceinline__ void GSNT(float *base, unsigned ofs, float v) { GS4(base, ofs, v); }
#else
__device__ __forceinline__ void GSNT(float *base, unsigned ofs, float v) { __builtin_nontemporal_store(v, (BFD_GLOBAL float *)((BFD_GLOBAL char *)guni(base) + pin(ofs))); }
#endif

template <int K> struct Ph { static constexpr int v = K; };

// ACC: bit0 = Pressure RMS sums, bit1 = Pressure peaks
template <bool LOSSY, bool UNI, int ACC>
__device__ __forceinline__ void fused_body(const bfd_dev &d, int bx, int by, int kbeg, int kend, int tm,
                                           float *__restrict__ sVx, float *__restrict__ sVy, float *__restrict__ sS, float *__restrict__ sR,
                                           const float *__restrict__ sTab, float *__restrict__ accP, float *__restrict__ pkP)
{
    const int N1 = d.N1;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = bx * FT_X, j0 = by * BFD_TILE_Y;      // by counts the 8-row tiles of the classification grid
    const long pl = d.plane;
    const float c1 = d.c1;
    constexpr bool accA = (ACC & 1) != 0, accK = (ACC & 2) != 0;
    float APu = 0.f, BPu = 0.f, ru = 0.f;
    if (UNI) { APu = d.AP[tm]; BPu = d.BP[tm]; ru = d.invRho[tm]; }

    // region cells of this thread: rows r + 7 c of one column; LDS index l0 + 7 FP c, plane offset g0 (bytes) + 7 N1 c cells,
    // the latter folded into the wave-uniform plane base
    const bool active = tid < NCELLT;
    const int tt = active ? tid : 0, r = tt / FR_W, col = tt - r * FR_W;
    const int l0 = r * FP + col + 2;
#ifdef FUSED_EXP_NORING     // experiment (wrong results): every cell reads inside the own 64 x 24 outputs -- what the ring lines cost
    const unsigned g0 = (unsigned)((j0 + min(max(r - 1, 0), 5)) * N1 + (i0 + min(max(col - 1, 0), FT_X - 1))) * 4u;
#else
    const unsigned g0 = (unsigned)((j0 - 1 + r) * N1 + (i0 - 1 + col)) * 4u;
#endif
    const long cstep = (long)RSTEP * N1;
    const bool colOut = active && col >= 1 && col <= FT_X;
    bool has[CPT], out[CPT];
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int row = r + RSTEP * c;
        has[c] = active && row < FR_H;
        out[c] = colOut && row >= 1 && row <= FT_Y;
    }
    // one extra halo value per thread; the array of a task is uniform per wave (SGPR base)
    bool hasT = false; int lT = 0; unsigned gT = 0;
    if (wv < 2) {
        hasT = tid < NTX;
        const int u = hasT ? tid : 0, rr = u / 3, q = u - 3 * rr, t = q < 2 ? q : FP - 1;
        lT = rr * FP + t; gT = (unsigned)((j0 - 1 + rr) * N1 + (i0 - 3 + t)) * 4u;
    } else if (wv < 6) {
        const int u0 = tid - 128;
        hasT = u0 < NTY;
        const int u = hasT ? u0 : 0, rr = u / FR_W, cc = u - rr * FR_W, ur = rr < 2 ? rr : FR_H + 2;
        lT = ur * FP + cc + 2; gT = (unsigned)((j0 - 3 + ur) * N1 + (i0 - 1 + cc)) * 4u;
    } else gT = g0;
#ifdef FUSED_EXP_NORING
    gT = g0;
#endif
    const float *aT = wv < 2 ? d.Vx : d.Vy;
    float *sT = (wv < 2 ? sVx : sVy) + lT;

    const int pFirst = kbeg - 1, pLast = kend + 1;       // planes of the new stress
    // Cells a thread does not have (threads 469 .. 511; row 27 of the threads with r = 6) run the same instructions on the LDS
    // index of a cell that exists and on addressable memory; only their LDS and global stores are masked. No branch separates
    // the four cells of a thread, so their LDS reads and arithmetic interleave.
    constexpr int BATCH = UNI ? CPT : 2;
    int lc[CPT];
#pragma unroll
    for (int c = 0; c < CPT; c++) lc[c] = l0 + RSTEP * FP * c;
    if (r + RSTEP * (CPT - 1) >= FR_H) lc[CPT - 1] = l0;
    // z queues with rotating slots (the plane loop is unrolled four times, so the slot numbers are constants and nothing is
    // moved): Vz of plane q in slot (q - pFirst) & 3, own new stress likewise; Vx, Vy, halo value of plane q in slot (q - pFirst) & 1
    float vz[CPT][4], sq[CPT][4], vxq[CPT][2], vyq[CPT][2], So[CPT], Ro[CPT], tvq[2];
    unsigned mr[CPT];                                    // !UNI: material id of plane p (no reflector in these runs)
    {
        const long k0 = (long)pFirst * pl;
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            const long kc = k0 + c * cstep;
            vxq[c][0] = GL4(d.Vx + kc, g0); vyq[c][0] = GL4(d.Vy + kc, g0); vxq[c][1] = vyq[c][1] = 0.f;
            vz[c][2] = GL4(d.Vz + kc - 2 * pl, g0); vz[c][3] = GL4(d.Vz + kc - pl, g0);
            vz[c][0] = GL4(d.Vz + kc, g0); vz[c][1] = GL4(d.Vz + kc + pl, g0);
            So[c] = GL4(d.Szz + kc, g0);
            Ro[c] = 0.f; mr[c] = 0;
            if (LOSSY) Ro[c] = GL4(d.Rzz + kc, g0);
            if (!UNI) mr[c] = GL2(d.mat + kc, g0 >> 1);
            sq[c][0] = sq[c][1] = sq[c][2] = sq[c][3] = 0.f;
        }
        tvq[0] = GL4(aT + k0, gT); tvq[1] = 0.f;
    }

    // one plane; PH = (p - pFirst) & 3 as a type
    auto plane = [&](auto PH, const int p) {
        constexpr int ph = decltype(PH)::v;
        constexpr int v0 = ph & 1, v1 = v0 ^ 1;          // slots of Vx, Vy: plane p, plane p+1
        constexpr int zm2 = (ph + 2) & 3, zm1 = (ph + 3) & 3, z0 = ph & 3, zp1 = (ph + 1) & 3;      // Vz / new stress of planes p-2, p-1, p, p+1 (new stress: p-3 in zp1)
        const long ko = (long)__builtin_amdgcn_readfirstlane(p) * pl;
        const bool own = p >= kbeg && p < kend;          // the outputs of this plane belong to the run
        const bool ownZ = p - 2 >= kbeg && p - 2 < kend;
        float *sRp = sR + ((unsigned)p % 3u) * FS_N;     // !UNI: 1/rho tiles of planes p, p-1, p-2 in a ring
        // ---- A: stage the velocities of plane p ----
#pragma unroll
        for (int c = 0; c < CPT; c++) if (has[c]) { sVx[lc[c]] = vxq[c][v0]; sVy[lc[c] + 2 * FP] = vyq[c][v0]; }
        if (hasT) *sT = tvq[v0];
        __syncthreads();

        // ---- loads, first group (in the order they are needed): the Pressure sums of plane p, then Vx, Vy of plane p+1.
        // Unconditional: the plane after the last one and the cells a thread does not have are addressable, their values unused.
        float av[CPT], pv[CPT];
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            av[c] = pv[c] = 0.f;
            if (own) {
                if (accA) av[c] = GLNT(accP + ko + c * cstep, g0);
                if (accK) pv[c] = GL4(pkP + ko + c * cstep, g0);
            }
        }
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            const long kc = ko + pl + c * cstep;
            vxq[c][v1] = GL4(d.Vx + kc, g0); vyq[c][v1] = GL4(d.Vy + kc, g0);
        }
        tvq[v1] = GL4(aT + ko + pl, gT);

        // ---- B: new stress of plane p on the region (the multi-material flavour in two batches of two cells: registers) ----
        float rn[CPT];
#pragma unroll
        for (int b0 = 0; b0 < CPT; b0 += BATCH) {
            float xa[BATCH], xb[BATCH], xc[BATCH], ya[BATCH], yb[BATCH], yc[BATCH], AP[BATCH], BP[BATCH], r0[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                const int c = b0 + u;
                const float *sx = sVx + lc[c], *sy = sVy + lc[c] + 2 * FP;
                xa[u] = sx[-2]; xb[u] = sx[-1]; xc[u] = sx[1];
                ya[u] = sy[-2 * FP]; yb[u] = sy[-FP]; yc[u] = sy[FP];
                AP[u] = APu; BP[u] = BPu; r0[u] = ru;
                if (!UNI) { const int m = mr[c] & BFD_MAT_MASK; AP[u] = sTab[m]; if (LOSSY) BP[u] = sTab[FUSED_TAB_MAX + m]; r0[u] = sTab[2 * FUSED_TAB_MAX + m]; }
            }
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                const int c = b0 + u;
                const float dxVx = dminus4(xa[u], xb[u], vxq[c][v0], xc[u]);
                const float dyVy = dminus4(ya[u], yb[u], vyq[c][v0], yc[u]);
                const float dzVz = dminus4(vz[c][zm2], vz[c][zm1], vz[c][z0], vz[c][zp1]);
                const float div = (dxVx + dyVy) + dzVz;
                float val;
                rn[c] = 0.f;
                if (LOSSY) { rn[c] = c1 * Ro[c] - BP[u] * div; val = So[c] + (AP[u] * div + 0.5f * (Ro[c] + rn[c])); }
                else val = So[c] + AP[u] * div;
                sq[c][z0] = val;
                if (has[c]) { sS[lc[c]] = val; if (!UNI) sRp[lc[c]] = r0[u]; }
                if (own && out[c]) {
                    GSNT(d.SzzW + ko + c * cstep, g0, val);
                    if (LOSSY) GSNT(d.RzzW + ko + c * cstep, g0, rn[c]);
                }
            }
            if (BATCH < CPT) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();

        // ---- C: Vz of plane p-2 from the thread's own queue of new stresses (registers only; frees the slot of Vz(p-2)) ----
        if (ownZ) {
            const float *sRa = sR + ((unsigned)(p + 1) % 3u) * FS_N, *sRb = sR + ((unsigned)(p + 2) % 3u) * FS_N;      // planes p-2, p-1
#pragma unroll
            for (int c = 0; c < CPT; c++) {
                const float dz = dplus4(sq[c][zp1], sq[c][zm2], sq[c][zm1], sq[c][z0]);
                float ra = ru, rb = ru;
                if (!UNI) { ra = sRa[lc[c]]; rb = sRb[lc[c]]; }
                const float nz = vz[c][zm2] + (0.5f * (ra + rb)) * dz;
                if (out[c]) GSNT(d.VzW + ko - 2 * pl + c * cstep, g0, nz);
            }
        }
        // ---- loads, second group: Szz, Rzz, ids of plane p+1 (their registers are free since B), Vz of plane p+2 into the slot of p-2 ----
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            const long kc = ko + pl + c * cstep;
            So[c] = GL4(d.Szz + kc, g0);
            if (LOSSY) Ro[c] = GL4(d.Rzz + kc, g0);
            if (!UNI) mr[c] = GL2(d.mat + kc, g0 >> 1);
            vz[c][zm2] = GL4(d.Vz + kc + pl, g0);
        }
        // ---- Vx, Vy of plane p from the LDS tile of the new stress; Pressure sums of plane p ----
        if (own) {
#pragma unroll
            for (int b0 = 0; b0 < CPT; b0 += BATCH) {
                float rc[BATCH], rx[BATCH], ry[BATCH], xa[BATCH], xb[BATCH], xc[BATCH], ya[BATCH], yb[BATCH], yc[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; u++) {
                    const int c = b0 + u;
                    const float *ps = sS + lc[c];
                    xa[u] = ps[-1]; xb[u] = ps[1]; xc[u] = ps[2];
                    ya[u] = ps[-FP]; yb[u] = ps[FP]; yc[u] = ps[2 * FP];
                    rx[u] = ry[u] = rc[u] = ru;
                    if (!UNI) { rc[u] = sRp[lc[c]]; rx[u] = sRp[lc[c] + 1]; ry[u] = sRp[lc[c] + FP]; }
                }
#pragma unroll
                for (int u = 0; u < BATCH; u++) {
                    const int c = b0 + u;
                    const float s0 = sq[c][z0];
                    const float dx = dplus4(xa[u], s0, xb[u], xc[u]);
                    const float dy = dplus4(ya[u], s0, yb[u], yc[u]);
                    const float nx = vxq[c][v0] + (0.5f * (rc[u] + rx[u])) * dx;
                    const float ny = vyq[c][v0] + (0.5f * (rc[u] + ry[u])) * dy;
                    const float s = (s0 + s0) + s0;
                    const float pr = -s * (1.0f / 3.0f);
                    if (out[c]) {
                        GSNT(d.VxW + ko + c * cstep, g0, nx);
                        GSNT(d.VyW + ko + c * cstep, g0, ny);
                        if (accA) GSNT(accP + ko + c * cstep, g0, av[c] + pr * pr);
                        if (accK) { const float ap = fabsf(pr); if (ap > pv[c]) GS4(pkP + ko + c * cstep, g0, ap); }
                    }
                }
                if (BATCH < CPT) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    int p = pFirst;
    for (; p + 3 <= pLast; p += 4) { plane(Ph<0>(), p); plane(Ph<1>(), p + 1); plane(Ph<2>(), p + 2); plane(Ph<3>(), p + 3); }
    if (p <= pLast) plane(Ph<0>(), p);
    if (p + 1 <= pLast) plane(Ph<1>(), p + 1);
    if (p + 2 <= pLast) plane(Ph<2>(), p + 2);
}

template <int ACC>
__global__ __launch_bounds__(NT, FUSED_WAVES_PER_SIMD) void fused_fluid(bfd_dev d, int tilesX, int nblocks, const int4 *__restrict__ runs, int nMat,
                                                                       float *__restrict__ accP, float *__restrict__ pkP)
{
    __shared__ float sVx[FS_N], sVy[FVY_N], sS[FS_N], sR[3 * FS_N], sTab[3 * FUSED_TAB_MAX];
    const int4 run = runs[remap_block(blockIdx.x, nblocks)];
    const int bx = run.x % tilesX, by = run.x / tilesX, kbeg = run.y & 0xFFFF, kend = run.y >> 16, tm = run.w;
#ifdef FUSED_EXP_LDS_PAD    // experiment: one workgroup per CU
    __shared__ float sPad[FUSED_EXP_LDS_PAD];
    if (nMat < 0) sPad[threadIdx.x] = 1.f, sTab[0] = sPad[(threadIdx.x + 1) % 512];
#endif
    if (!(run.z & 4)) {              // ids per cell: AP, BP, 1/rho of every material in LDS (runs of media with more materials are not fused)
        for (int m = threadIdx.x; m < nMat; m += NT) { sTab[m] = d.AP[m]; sTab[FUSED_TAB_MAX + m] = d.BP[m]; sTab[2 * FUSED_TAB_MAX + m] = d.invRho[m]; }
        __syncthreads();
    }
    switch ((run.z >> 1) & 3) {      // bit1 lossy (grown region), bit2 UNI
    case 0: fused_body<false, false, ACC>(d, bx, by, kbeg, kend, tm, sVx, sVy, sS, sR, sTab, accP, pkP); break;
    case 1: fused_body<true, false, ACC>(d, bx, by, kbeg, kend, tm, sVx, sVy, sS, sR, sTab, accP, pkP); break;
    case 2: fused_body<false, true, ACC>(d, bx, by, kbeg, kend, tm, sVx, sVy, sS, sR, sTab, accP, pkP); break;
    default: fused_body<true, true, ACC>(d, bx, by, kbeg, kend, tm, sVx, sVy, sS, sR, sTab, accP, pkP); break;
    }
}

}  // namespace

int bfd_fused_rows(void) { return FT_Y; }
int bfd_fused_max_materials(void) { return FUSED_TAB_MAX; }

// fused time step of the eligible fluid runs (variant 4); d = the view whose d.X are the old fields and d.XW the new ones
void bfd_launch_fused(const bfd_dev &d, hipStream_t s, float *accP, float *pkP, const bfd_tiles *t, int off, int n)
{
    const int tilesX = (d.N1 + BFD_TILE_X - 1) / BFD_TILE_X;
    if (n <= 0) return;
    const int4 *runs = t->runs + t->nFluid + t->nSolid + off;
    if (t->ktimer) bfd_kmark(t->ktimer, BFD_K_FUSED, 0, s);
    switch ((accP ? 1 : 0) | (pkP ? 2 : 0)) {
    case 0: hipLaunchKernelGGL((fused_fluid<0>), dim3(n), dim3(NT), 0, s, d, tilesX, n, runs, t->nMat, accP, pkP); break;
    case 1: hipLaunchKernelGGL((fused_fluid<1>), dim3(n), dim3(NT), 0, s, d, tilesX, n, runs, t->nMat, accP, pkP); break;
    case 2: hipLaunchKernelGGL((fused_fluid<2>), dim3(n), dim3(NT), 0, s, d, tilesX, n, runs, t->nMat, accP, pkP); break;
    default: hipLaunchKernelGGL((fused_fluid<3>), dim3(n), dim3(NT), 0, s, d, tilesX, n, runs, t->nMat, accP, pkP); break;
    }
    if (t->ktimer) bfd_kmark(t->ktimer, BFD_K_FUSED, 1, s);
}
