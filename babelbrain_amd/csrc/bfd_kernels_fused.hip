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
constexpr int NT = 512, CPT = 4;
constexpr int RSTEP = (FR_H + CPT - 1) / CPT;           // 7: thread (r, col) owns the region cells (r + 7 c, col), c = 0 .. 3
constexpr int NCELLT = RSTEP * FR_W;                    // 469 threads hold cells
static_assert(NCELLT <= NT && RSTEP * CPT >= FR_H, "four region cells per thread");
constexpr int NTX = 3 * FR_H;                           // extra Vx columns t = 0, 1, 69: 81 tasks (waves 0-1)
constexpr int NTY = 3 * FR_W;                           // extra Vy rows u = 0, 1, 29: 201 tasks (waves 2-5)
static_assert(NTX <= 128 && NTY <= 256, "halo tasks fit their waves");
#ifndef FUSED_WAVES_PER_SIMD
#define FUSED_WAVES_PER_SIMD 4
#endif

template <bool LOSSY, bool UNI, bool ACC>
__device__ __forceinline__ void fused_body(const bfd_dev &d, int bx, int by, int kbeg, int kend, int tm,
                                           float *__restrict__ sVx, float *__restrict__ sVy, float *__restrict__ sS, float *__restrict__ sR,
                                           float *__restrict__ accP, float *__restrict__ pkP)
{
    const int N1 = d.N1;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = bx * FT_X, j0 = by * BFD_TILE_Y;      // by counts the 8-row tiles of the classification grid
    const long pl = d.plane;
    const float c1 = d.c1;
    const bool accA = ACC && accP != nullptr, accK = ACC && pkP != nullptr;
    float APu = 0.f, BPu = 0.f, ru = 0.f;
    if (UNI) { APu = d.AP[tm]; BPu = d.BP[tm]; ru = d.invRho[tm]; }

    // region cells of this thread: rows r + 7 c of one column; LDS index l0 + 7 FP c, plane offset g0 (bytes) + 7 N1 c cells,
    // the latter folded into the wave-uniform plane base
    const bool active = tid < NCELLT;
    const int tt = active ? tid : 0, r = tt / FR_W, col = tt - r * FR_W;
    const int l0 = r * FP + col + 2;
    const unsigned g0 = (unsigned)((j0 - 1 + r) * N1 + (i0 - 1 + col)) * 4u;
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
    }
    const float *aT = wv < 2 ? d.Vx : d.Vy;
    float *sT = (wv < 2 ? sVx : sVy) + lT;

    const int pFirst = kbeg - 1, pLast = kend + 1;       // planes of the new stress
    float vzm2[CPT], vzm1[CPT], vz0[CPT], vzp1[CPT], vx[CPT], vy[CPT], So[CPT], Ro[CPT];
    float s3[CPT], s2[CPT], s1[CPT];                     // own new stress of planes p-3, p-2, p-1
    unsigned mr[CPT];                                    // !UNI: material id of plane p (no reflector in these runs)
    float tv = 0.f;
    {
        const long k0 = (long)pFirst * pl;
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            vzm2[c] = vzm1[c] = vz0[c] = vzp1[c] = vx[c] = vy[c] = So[c] = Ro[c] = 0.f;
            s3[c] = s2[c] = s1[c] = 0.f; mr[c] = 0;
            if (has[c]) {
                const long kc = k0 + c * cstep;
                vx[c] = F4(d.Vx + kc, g0); vy[c] = F4(d.Vy + kc, g0);
                vzm2[c] = F4(d.Vz + kc - 2 * pl, g0); vzm1[c] = F4(d.Vz + kc - pl, g0);
                vz0[c] = F4(d.Vz + kc, g0); vzp1[c] = F4(d.Vz + kc + pl, g0);
                So[c] = F4(d.Szz + kc, g0);
                if (LOSSY) Ro[c] = F4(d.Rzz + kc, g0);
                if (!UNI) mr[c] = U2(d.mat + kc, g0 >> 1);
            }
        }
        if (hasT) tv = F4(aT + k0, gT);
    }

    for (int p = pFirst; p <= pLast; p++) {
        const long ko = (long)__builtin_amdgcn_readfirstlane(p) * pl;
        const bool own = p >= kbeg && p < kend;          // the outputs of this plane belong to the run
        float *sRp = sR + ((unsigned)p % 3u) * FS_N;     // !UNI: 1/rho tiles of planes p, p-1, p-2 in a ring
        // ---- A: stage the velocities of plane p ----
#pragma unroll
        for (int c = 0; c < CPT; c++) if (has[c]) { sVx[l0 + RSTEP * FP * c] = vx[c]; sVy[l0 + RSTEP * FP * c + 2 * FP] = vy[c]; }
        if (hasT) *sT = tv;
        __syncthreads();

        // ---- loads of plane p+1 (Vz: p+2) ----
        float nvx[CPT], nvy[CPT], nvz[CPT], nSo[CPT], nRo[CPT], ntv = 0.f;
        unsigned nmr[CPT];
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            nvx[c] = nvy[c] = nvz[c] = nSo[c] = nRo[c] = 0.f; nmr[c] = 0;
            if (p < pLast && has[c]) {
                const long kc = ko + pl + c * cstep;
                nvx[c] = F4(d.Vx + kc, g0); nvy[c] = F4(d.Vy + kc, g0); nvz[c] = F4(d.Vz + kc + pl, g0);
                nSo[c] = F4(d.Szz + kc, g0);
                if (LOSSY) nRo[c] = F4(d.Rzz + kc, g0);
                if (!UNI) nmr[c] = U2(d.mat + kc, g0 >> 1);
            }
        }
        if (p < pLast && hasT) ntv = F4(aT + ko + pl, gT);

        // ---- B: new stress of plane p on the region ----
        float sn[CPT], r0[CPT];
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            sn[c] = 0.f; r0[c] = ru;
            if (has[c]) {
                const int lc = l0 + RSTEP * FP * c;
                const float *sx = sVx + lc, *sy = sVy + lc + 2 * FP;
                const float dxVx = dminus4(sx[-2], sx[-1], vx[c], sx[1]);
                const float dyVy = dminus4(sy[-2 * FP], sy[-FP], vy[c], sy[FP]);
                const float dzVz = dminus4(vzm2[c], vzm1[c], vz0[c], vzp1[c]);
                float AP = APu, BP = BPu;
                if (!UNI) { const int m = mr[c] & BFD_MAT_MASK; AP = d.AP[m]; if (LOSSY) BP = d.BP[m]; r0[c] = d.invRho[m]; }
                const float div = (dxVx + dyVy) + dzVz;
                float val, rn = 0.f;
                if (LOSSY) { rn = c1 * Ro[c] - BP * div; val = So[c] + (AP * div + 0.5f * (Ro[c] + rn)); }
                else val = So[c] + AP * div;
                sn[c] = val;
                sS[lc] = val;
                if (!UNI) sRp[lc] = r0[c];
                if (own && out[c]) {
                    ST4(d.SzzW + ko + c * cstep, g0, val);
                    if (LOSSY) ST4(d.RzzW + ko + c * cstep, g0, rn);
                }
            }
        }
        __syncthreads();

        // ---- C: velocities of the outputs: Vx, Vy of plane p, Vz of plane p-2; Pressure sums of plane p ----
        const bool ownZ = p - 2 >= kbeg && p - 2 < kend;
        const float *sRa = sR + ((unsigned)(p + 1) % 3u) * FS_N, *sRb = sR + ((unsigned)(p + 2) % 3u) * FS_N;      // planes p-2, p-1
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            if (out[c]) {
                const int lc = l0 + RSTEP * FP * c;
                if (own) {
                    const float *ps = sS + lc;
                    const float s0 = sn[c];
                    const float dx = dplus4(ps[-1], s0, ps[1], ps[2]);
                    const float dy = dplus4(ps[-FP], s0, ps[FP], ps[2 * FP]);
                    float rx = ru, ry = ru;
                    if (!UNI) { rx = sRp[lc + 1]; ry = sRp[lc + FP]; }
                    ST4(d.VxW + ko + c * cstep, g0, vx[c] + (0.5f * (r0[c] + rx)) * dx);
                    ST4(d.VyW + ko + c * cstep, g0, vy[c] + (0.5f * (r0[c] + ry)) * dy);
                    if (ACC) {
                        const float s = (s0 + s0) + s0;
                        const float pr = -s * (1.0f / 3.0f);
                        if (accA) { float *pa = accP + ko + c * cstep; ST4(pa, g0, LD4(pa, g0) + pr * pr); }
                        if (accK) { float *pp = pkP + ko + c * cstep; const float ap = fabsf(pr); if (ap > F4(pp, g0)) F4(pp, g0) = ap; }
                    }
                }
                if (ownZ) {
                    const float dz = dplus4(s3[c], s2[c], s1[c], sn[c]);
                    float ra = ru, rb = ru;
                    if (!UNI) { ra = sRa[lc]; rb = sRb[lc]; }
                    ST4(d.VzW + ko - 2 * pl + c * cstep, g0, vzm2[c] + (0.5f * (ra + rb)) * dz);
                }
            }
        }
        // rotate
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            s3[c] = s2[c]; s2[c] = s1[c]; s1[c] = sn[c];
            vzm2[c] = vzm1[c]; vzm1[c] = vz0[c]; vz0[c] = vzp1[c]; vzp1[c] = nvz[c];
            vx[c] = nvx[c]; vy[c] = nvy[c]; So[c] = nSo[c]; Ro[c] = nRo[c];
            if (!UNI) mr[c] = nmr[c];
        }
        tv = ntv;
    }
}

template <bool ACC>
__global__ __launch_bounds__(NT, FUSED_WAVES_PER_SIMD) void fused_fluid(bfd_dev d, int tilesX, int nblocks, const int4 *__restrict__ runs,
                                                                       float *__restrict__ accP, float *__restrict__ pkP)
{
    __shared__ float sVx[FS_N], sVy[FVY_N], sS[FS_N], sR[3 * FS_N];
    const int4 run = runs[remap_block(blockIdx.x, nblocks)];
    const int bx = run.x % tilesX, by = run.x / tilesX, kbeg = run.y & 0xFFFF, kend = run.y >> 16, tm = run.w;
    switch ((run.z >> 1) & 3) {      // bit1 lossy (grown region), bit2 UNI
    case 0: fused_body<false, false, ACC>(d, bx, by, kbeg, kend, tm, sVx, sVy, sS, sR, accP, pkP); break;
    case 1: fused_body<true, false, ACC>(d, bx, by, kbeg, kend, tm, sVx, sVy, sS, sR, accP, pkP); break;
    case 2: fused_body<false, true, ACC>(d, bx, by, kbeg, kend, tm, sVx, sVy, sS, sR, accP, pkP); break;
    default: fused_body<true, true, ACC>(d, bx, by, kbeg, kend, tm, sVx, sVy, sS, sR, accP, pkP); break;
    }
}

}  // namespace

int bfd_fused_rows(void) { return FT_Y; }

// fused time step of the eligible fluid runs (variant 4); d = the view whose d.X are the old fields and d.XW the new ones
void bfd_launch_fused(const bfd_dev &d, hipStream_t s, float *accP, float *pkP, const bfd_tiles *t, int off, int n)
{
    const int tilesX = (d.N1 + BFD_TILE_X - 1) / BFD_TILE_X;
    if (n <= 0) return;
    const int4 *runs = t->runs + t->nFluid + t->nSolid + off;
    if (t->ktimer) bfd_kmark(t->ktimer, BFD_K_FUSED, 0, s);
    if (accP || pkP) hipLaunchKernelGGL((fused_fluid<true>), dim3(n), dim3(NT), 0, s, d, tilesX, n, runs, accP, pkP);
    else hipLaunchKernelGGL((fused_fluid<false>), dim3(n), dim3(NT), 0, s, d, tilesX, n, runs, accP, pkP);
    if (t->ktimer) bfd_kmark(t->ktimer, BFD_K_FUSED, 1, s);
}
