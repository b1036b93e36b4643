// Variant 2: LDS-staged, z-marching kernels for gfx950 (the production path).
//
// Replaces the per-step device kernels of the reference's solver backends (package
// BabelViscoFDTD, absent from /root/reference; call site BabelIntegrationBASE.py:2338).
//
// Design (DESIGN.md "Kernels"):
//  * x is the fastest axis; a wavefront is 64 consecutive x voxels of one row, so every state
//    load/store is a fully coalesced 256-B row segment.
//  * A workgroup owns a TX x TY = 64 x 8 tile (8 waves) and marches ZC planes along z. The z
//    stencil (k-2..k+2) lives in per-thread register queues, so each state value is fetched
//    from HBM once per half-step; the in-plane stencil (+-2 in x and y) is served from an LDS
//    tile of the current plane with its halo ring, double-buffered (one barrier per plane).
//  * Halo ring loads are distributed over the workgroup as two per-thread "tasks" fixed before
//    the loop; y-halo rows are full coalesced rows.
//  * Tiles are dealt to XCDs in contiguous runs (blockIdx -> tile remap) so neighbouring tiles
//    share one L2.
//  * Arithmetic is the canonical float32 sequence of oracle/fdtd_oracle.c (no contraction).
#include "bfd_internal.h"
#include "bfd_device.h"
#include <algorithm>

namespace {

constexpr int TX = BFD_TILE_X;
constexpr int TY = BFD_TILE_Y;
constexpr int LW = TX + 4;          // LDS row length (floats)
constexpr int LH = TY + 4;          // LDS rows
constexpr int NTHREADS = TX * TY;   // 512
constexpr int YT = 4 * TX;          // y-halo tasks per array (4 rows x 64)
constexpr int XT = 4 * TY;          // x-halo tasks per array (4 cols x TY)
#ifndef BFD_ZCHUNK
#define BFD_ZCHUNK 32
#endif
constexpr int ZCHUNK = BFD_ZCHUNK;  // longest z-run one workgroup marches
constexpr int SUBZ = BFD_SUBZ;             // z granularity of the fluid/solid classification (runs are merged sub-tiles)
#ifndef STRESS_WAVES_PER_SIMD
#define STRESS_WAVES_PER_SIMD 4     // 2 workgroups of 8 waves per CU (<= 128 VGPRs); 6 or 8 spill and run 1.4-2.4x slower (measured)
#endif
#ifndef FLUID_WAVES_PER_SIMD
#define FLUID_WAVES_PER_SIMD 8      // 57 / 64 VGPRs since the plane bases live in SGPRs (uni()); round 1, with 64-bit per-lane addresses: 6 -> 61 Gvoxel-steps/s, 8 (spills) -> 47
#endif
#ifndef VELOCITY_WAVES_PER_SIMD
#define VELOCITY_WAVES_PER_SIMD 4
#endif
#ifndef VELOCITY_FLUID_WAVES_PER_SIMD
#define VELOCITY_FLUID_WAVES_PER_SIMD FLUID_WAVES_PER_SIMD     // of velocity_fluid alone (64 VGPRs + one 8-byte spill at 8 waves; at 7 no spill): A/B in profiles/r4/
#endif

struct HaloTask {
    int lofs;       // offset inside one LDS tile (floats), -1 = no task
    int gofs;       // in-plane global offset j*N1+i (valid only if ok)
    int arr;        // which array of the kernel's LDS set
    bool ok;        // inside the domain (else the halo value is 0)
};

// y-type task u in [0,256): row r=u/64 -> ly = r<2 ? r : TY+r ; lx = u%64+2
__device__ __forceinline__ void ytask(int u, int arr, int i0, int j0, int N1, int N2, HaloTask &t)
{
    const int r = u >> 6, c = u & 63;
    const int ly = r < 2 ? r : TY + r, lx = c + 2;
    const int gi = i0 + c, gj = j0 - 2 + ly;
    t.arr = arr; t.lofs = ly * LW + lx; t.gofs = gj * N1 + gi;
    t.ok = (gi < N1) && (gj >= 0) && (gj < N2);
}
// x-type task u in [0,32): ly = u/4+2 ; c=u%4 -> lx = c<2 ? c : TX+c
__device__ __forceinline__ void xtask(int u, int arr, int i0, int j0, int N1, int N2, HaloTask &t)
{
    const int c = u & 3, ly = (u >> 2) + 2;
    const int lx = c < 2 ? c : TX + c;
    const int gi = i0 - 2 + lx, gj = j0 - 2 + ly;
    t.arr = arr; t.lofs = ly * LW + lx; t.gofs = gj * N1 + gi;
    t.ok = (gi >= 0) && (gi < N1) && (gj < N2);
}

// ------------------------------------------------------------------------------------------------
// stress half-step
// ------------------------------------------------------------------------------------------------
// rowFlags (may be null): per run and plane, 2 bits per tile row (wave): bit0 = every cell of the row has a
// fluid centre and no reflector -> the row takes the fluid arithmetic (one normal stress read, identical
// result written to all three, no shear work); bit1 = additionally every cell has BP == 0 (no memory variable).
__global__ __launch_bounds__(NTHREADS, STRESS_WAVES_PER_SIMD) void stress_v2(bfd_dev d, int tilesX, int nblocks,
                                                                             const int4 *__restrict__ runs,
                                                                             const unsigned short *__restrict__ rowFlags)
{
    __shared__ float sV[2][3][LH * LW];
    const int N1 = d.N1, N2 = d.N2;
    const int pos = remap_block(blockIdx.x, nblocks);
    const int4 run = runs[pos];
    const int bx = run.x % tilesX, by = run.x / tilesX;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int i0 = bx * TX, j0 = by * TY;
    const int i = i0 + tx, j = j0 + ty;
    const bool valid = (i < N1) && (j < N2);
    const long pl = d.plane;
    const int kbeg = run.y & 0xFFFF, kend = run.y >> 16;
    const int P = d.P;
    const int own = (ty + 2) * LW + tx + 2;
    const unsigned cij = valid ? (unsigned)(j * N1 + i) : 0u;    // in-plane offset shared by every array

    // halo tasks: [Vx-y, Vy-y, Vz-y] 3*256, then [Vx-x, Vy-x, Vz-x] 3*32
    HaloTask ta, tb;
    ytask(tid & 255, tid >> 8, i0, j0, N1, N2, ta);
    {
        const int t2 = tid + NTHREADS;
        if (t2 < 3 * YT) ytask(t2 - 2 * YT, 2, i0, j0, N1, N2, tb);
        else if (t2 < 3 * YT + 3 * XT) { const int u = t2 - 3 * YT; xtask(u % XT, u / XT, i0, j0, N1, N2, tb); }
        else { tb.lofs = -1; tb.ok = false; tb.arr = 0; tb.gofs = 0; }
    }
    const float *pa = (ta.arr == 0 ? d.Vx : (ta.arr == 1 ? d.Vy : d.Vz)) + (ta.ok ? ta.gofs : 0);
    const float *pb = (tb.arr == 0 ? d.Vx : (tb.arr == 1 ? d.Vy : d.Vz)) + (tb.ok ? tb.gofs : 0);
    float *la = &sV[0][ta.arr][ta.lofs];
    float *lb = &sV[0][tb.arr][tb.lofs < 0 ? 0 : tb.lofs];
    const bool hasB = tb.lofs >= 0;

    const bool zi = valid && (i < P || i >= N1 - P);
    const bool zj = valid && (j < P || j >= N2 - P);
    const float c1 = d.c1, k2 = d.k2;

    // z register queues, primed for plane kbeg (ghost planes make kbeg-2 .. always addressable)
    float vxm1 = 0, vx0 = 0, vxp1 = 0, vxp2 = 0, vym1 = 0, vy0 = 0, vyp1 = 0, vyp2 = 0, vzm2 = 0, vzm1 = 0, vz0 = 0, vzp1 = 0;
    {
        const float *bVx = d.Vx + kbeg * pl, *bVy = d.Vy + kbeg * pl, *bVz = d.Vz + kbeg * pl;
        if (valid) {
            vxm1 = F4((bVx - pl), cij * 4u); vx0 = F4(bVx, cij * 4u); vxp1 = F4((bVx + pl), cij * 4u); vxp2 = F4((bVx + 2 * pl), cij * 4u);
            vym1 = F4((bVy - pl), cij * 4u); vy0 = F4(bVy, cij * 4u); vyp1 = F4((bVy + pl), cij * 4u); vyp2 = F4((bVy + 2 * pl), cij * 4u);
            vzm2 = F4((bVz - 2 * pl), cij * 4u); vzm1 = F4((bVz - pl), cij * 4u); vz0 = F4(bVz, cij * 4u); vzp1 = F4((bVz + pl), cij * 4u);
        }
    }
    float ha = ta.ok ? pa[kbeg * pl] : 0.0f;
    float hb = tb.ok ? pb[kbeg * pl] : 0.0f;

    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        const long ko = (long)kl * pl;          // uniform: plane bases stay in SGPRs
        const int k = d.k0 + kl;
        const int bo = b * (3 * LH * LW);
        // stage plane kl in LDS
        sV[b][0][own] = vx0; sV[b][1][own] = vy0; sV[b][2][own] = vz0;
        la[bo] = ha;
        if (hasB) lb[bo] = hb;
        __syncthreads();

        // this plane's state first (needed soonest), then the prefetches for plane kl+1
        float *pSxx = d.Sxx + ko, *pSyy = d.Syy + ko, *pSzz = d.Szz + ko;
        float *pRxx = d.Rxx + ko, *pRyy = d.Ryy + ko, *pRzz = d.Rzz + ko;
        const uint16_t *pM = d.mat + ko;
        const unsigned rf = rowFlags ? (rowFlags[pos * ZCHUNK + (kl - kbeg)] >> (2 * __builtin_amdgcn_readfirstlane(ty))) & 3u : 0u;
        const bool rowFluid = rf & 1u, rowLossless = rf & 2u;       // wave-uniform
        unsigned mraw = 0;
        float sxx = 0, syy = 0, szz = 0, rxx = 0, ryy = 0, rzz = 0;
        if (valid) {
            mraw = U2(pM, cij * 2u);
            szz = F4(pSzz, cij * 4u);
            if (!rowLossless) rzz = F4(pRzz, cij * 4u);
            if (!rowFluid) { sxx = F4(pSxx, cij * 4u); syy = F4(pSyy, cij * 4u); rxx = F4(pRxx, cij * 4u); ryy = F4(pRyy, cij * 4u); }
        }
        float nvx = 0, nvy = 0, nvz = 0, nha = 0, nhb = 0;
        if (kl + 1 < kend) {
            if (valid) { nvx = F4((d.Vx + ko + 3 * pl), cij * 4u); nvy = F4((d.Vy + ko + 3 * pl), cij * 4u); nvz = F4((d.Vz + ko + 2 * pl), cij * 4u); }
            if (ta.ok) nha = pa[ko + pl];
            if (tb.ok) nhb = pb[ko + pl];
        }

        if (valid && rowFluid) {
            // fluid row inside a solid tile: same arithmetic as stress_fluid_body, all three copies written
            const float *sx = &sV[b][0][own], *sy = &sV[b][1][own];
            float dxVx = dminus4(sx[-2], sx[-1], vx0, sx[1]);
            float dyVy = dminus4(sy[-2 * LW], sy[-LW], vy0, sy[LW]);
            float dzVz = dminus4(vzm2, vzm1, vz0, vzp1);
            if (zi) {
                const int xi = i < P ? i : i - (N1 - 2 * P);
                dxVx = cpml(d.psi[0], (unsigned)((kl * N2 + j) * (2 * P) + xi), d.axI[i], d.bxI[i], dxVx);
            }
            if (zj) {
                const int yj = j < P ? j : j - (N2 - 2 * P);
                dyVy = cpml(d.psi[1], (unsigned)((kl * (2 * P) + yj) * N1 + i), d.ayI[j], d.byI[j], dyVy);
            }
            if (k < P || k >= d.N3 - P) {
                const int zk = k < P ? k : k - (d.N3 - 2 * P);
                dzVz = cpml(d.psi[2], (unsigned)(zk * d.plane) + cij, d.azI[k], d.bzI[k], dzVz);
            }
            const int m = mraw & BFD_MAT_MASK;
            const float div = (dxVx + dyVy) + dzVz;
            const float AP = d.AP[m];
            float val;
            if (rowLossless) {
                val = szz + AP * div;
            } else {
                const float rn = c1 * rzz - d.BP[m] * div;
                val = szz + (AP * div + 0.5f * (rzz + rn));
                F4(pRxx, cij * 4u) = rn; F4(pRyy, cij * 4u) = rn; F4(pRzz, cij * 4u) = rn;
            }
            F4(pSxx, cij * 4u) = val; F4(pSyy, cij * 4u) = val; F4(pSzz, cij * 4u) = val;
        } else if (valid) {
            const float *sx = &sV[b][0][own], *sy = &sV[b][1][own], *sz = &sV[b][2][own];
            float dxVx = dminus4(sx[-2], sx[-1], vx0, sx[1]);
            float dyVy = dminus4(sy[-2 * LW], sy[-LW], vy0, sy[LW]);
            float dzVz = dminus4(vzm2, vzm1, vz0, vzp1);
            float dyVx = dplus4(sx[-LW], vx0, sx[LW], sx[2 * LW]);
            float dxVy = dplus4(sy[-1], vy0, sy[1], sy[2]);
            float dzVx = dplus4(vxm1, vx0, vxp1, vxp2);
            float dxVz = dplus4(sz[-1], vz0, sz[1], sz[2]);
            float dzVy = dplus4(vym1, vy0, vyp1, vyp2);
            float dyVz = dplus4(sz[-LW], vz0, sz[LW], sz[2 * LW]);

            if (mraw & BFD_REFLECTOR_BIT) {
                F4(pSxx, cij * 4u) = 0.f; F4(pSyy, cij * 4u) = 0.f; F4(pSzz, cij * 4u) = 0.f; F4((d.Sxy + ko), cij * 4u) = 0.f; F4((d.Sxz + ko), cij * 4u) = 0.f; F4((d.Syz + ko), cij * 4u) = 0.f;
                F4(pRxx, cij * 4u) = 0.f; F4(pRyy, cij * 4u) = 0.f; F4(pRzz, cij * 4u) = 0.f; F4((d.Rxy + ko), cij * 4u) = 0.f; F4((d.Rxz + ko), cij * 4u) = 0.f; F4((d.Ryz + ko), cij * 4u) = 0.f;
            } else {
                const int m = mraw & BFD_MAT_MASK;
                if (zi) {
                    const int xi = i < P ? i : i - (N1 - 2 * P);
                    const unsigned q = (unsigned)((kl * N2 + j) * (2 * P) + xi);
                    dxVx = cpml(d.psi[0], q, d.axI[i], d.bxI[i], dxVx);
                    dxVy = cpml(d.psi[4], q, d.axH[i], d.bxH[i], dxVy);
                    dxVz = cpml(d.psi[6], q, d.axH[i], d.bxH[i], dxVz);
                }
                if (zj) {
                    const int yj = j < P ? j : j - (N2 - 2 * P);
                    const unsigned q = (unsigned)((kl * (2 * P) + yj) * N1 + i);
                    dyVy = cpml(d.psi[1], q, d.ayI[j], d.byI[j], dyVy);
                    dyVx = cpml(d.psi[3], q, d.ayH[j], d.byH[j], dyVx);
                    dyVz = cpml(d.psi[8], q, d.ayH[j], d.byH[j], dyVz);
                }
                if (k < P || k >= d.N3 - P) {
                    const int zk = k < P ? k : k - (d.N3 - 2 * P);
                    const unsigned q = (unsigned)(zk * d.plane) + cij;
                    dzVz = cpml(d.psi[2], q, d.azI[k], d.bzI[k], dzVz);
                    dzVx = cpml(d.psi[5], q, d.azH[k], d.bzH[k], dzVx);
                    dzVy = cpml(d.psi[7], q, d.azH[k], d.bzH[k], dzVy);
                }
                {
                    const float AP = d.AP[m], BP = d.BP[m], AS2 = d.AS2[m], BS2 = d.BS2[m];
                    const float sXY = dxVx + dyVy;
                    const float div = sXY + dzVz;
                    const float sYZ = dyVy + dzVz, sXZ = dxVx + dzVz;
                    float rn;
                    rn = c1 * rxx - (BP * div - BS2 * sYZ);
                    F4(pSxx, cij * 4u) = sxx + ((AP * div - AS2 * sYZ) + 0.5f * (rxx + rn)); F4(pRxx, cij * 4u) = rn;
                    rn = c1 * ryy - (BP * div - BS2 * sXZ);
                    F4(pSyy, cij * 4u) = syy + ((AP * div - AS2 * sXZ) + 0.5f * (ryy + rn)); F4(pRyy, cij * 4u) = rn;
                    rn = c1 * rzz - (BP * div - BS2 * sXY);
                    F4(pSzz, cij * 4u) = szz + ((AP * div - AS2 * sXY) + 0.5f * (rzz + rn)); F4(pRzz, cij * 4u) = rn;
                }
                const float iv0 = d.invMu[m];
                if (iv0 > 0.f) {    // shear only where the centre cell is solid
                    const float t0 = d.tauS[m];
                    const int i1 = min(i + 1, N1 - 1), j1 = min(j + 1, N2 - 1);
                    const unsigned r0 = (unsigned)(j * N1), r1 = (unsigned)(j1 * N1);
                    const uint16_t *pM1 = pM + pl;
                    const int mx = pM[r0 + i1] & BFD_MAT_MASK, my = pM[r1 + i] & BFD_MAT_MASK;
                    const int mz = pM1[r0 + i] & BFD_MAT_MASK, mxy = pM[r1 + i1] & BFD_MAT_MASK;
                    const int mxz = pM1[r0 + i1] & BFD_MAT_MASK, myz = pM1[r1 + i] & BFD_MAT_MASK;
                    const float ivx = d.invMu[mx], ivy = d.invMu[my], ivz = d.invMu[mz];
                    {
                        const float e4 = d.invMu[mxy];
                        if (ivx > 0.f && ivy > 0.f && e4 > 0.f) {
                            float *pS = d.Sxy + ko, *pR = d.Rxy + ko;
                            const float muH = 4.0f / ((iv0 + ivx) + (ivy + e4));
                            const float tau = 0.25f * ((t0 + d.tauS[mx]) + (d.tauS[my] + d.tauS[mxy]));
                            const float A = muH * (1.0f + tau), B = (muH * tau) * k2;
                            const float e = dyVx + dxVy;
                            const float r = F4(pR, cij * 4u), rn = c1 * r - B * e;
                            F4(pS, cij * 4u) = F4(pS, cij * 4u) + (A * e + 0.5f * (r + rn)); F4(pR, cij * 4u) = rn;
                        }
                    }
                    {
                        const float e4 = d.invMu[mxz];
                        if (ivx > 0.f && ivz > 0.f && e4 > 0.f) {
                            float *pS = d.Sxz + ko, *pR = d.Rxz + ko;
                            const float muH = 4.0f / ((iv0 + ivx) + (ivz + e4));
                            const float tau = 0.25f * ((t0 + d.tauS[mx]) + (d.tauS[mz] + d.tauS[mxz]));
                            const float A = muH * (1.0f + tau), B = (muH * tau) * k2;
                            const float e = dzVx + dxVz;
                            const float r = F4(pR, cij * 4u), rn = c1 * r - B * e;
                            F4(pS, cij * 4u) = F4(pS, cij * 4u) + (A * e + 0.5f * (r + rn)); F4(pR, cij * 4u) = rn;
                        }
                    }
                    {
                        const float e4 = d.invMu[myz];
                        if (ivy > 0.f && ivz > 0.f && e4 > 0.f) {
                            float *pS = d.Syz + ko, *pR = d.Ryz + ko;
                            const float muH = 4.0f / ((iv0 + ivy) + (ivz + e4));
                            const float tau = 0.25f * ((t0 + d.tauS[my]) + (d.tauS[mz] + d.tauS[myz]));
                            const float A = muH * (1.0f + tau), B = (muH * tau) * k2;
                            const float e = dzVy + dyVz;
                            const float r = F4(pR, cij * 4u), rn = c1 * r - B * e;
                            F4(pS, cij * 4u) = F4(pS, cij * 4u) + (A * e + 0.5f * (r + rn)); F4(pR, cij * 4u) = rn;
                        }
                    }
                }
            }
        }
        // rotate the z queues
        vxm1 = vx0; vx0 = vxp1; vxp1 = vxp2; vxp2 = nvx;
        vym1 = vy0; vy0 = vyp1; vyp1 = vyp2; vyp2 = nvy;
        vzm2 = vzm1; vzm1 = vz0; vz0 = vzp1; vzp1 = nvz;
        ha = nha; hb = nhb;
    }
}

// ------------------------------------------------------------------------------------------------
// velocity half-step (+ fused Pressure RMS / peak accumulation)
// ------------------------------------------------------------------------------------------------
// LDS set: 0 Sxx (x halo), 1 Syy (y halo), 2 Sxy (x and y halo), 3 Sxz (x halo), 4 Syz (y halo)
template <bool ACC>
__global__ __launch_bounds__(NTHREADS, VELOCITY_WAVES_PER_SIMD) void velocity_v2(bfd_dev d, int tilesX, int nblocks,
                                                        float *__restrict__ accP, float *__restrict__ pkP,
                                                        const int4 *__restrict__ runs)
{
    __shared__ float sS[2][5][LH * LW];
    const int N1 = d.N1, N2 = d.N2;
    const int4 run = runs[remap_block(blockIdx.x, nblocks)];
    const int bx = run.x % tilesX, by = run.x / tilesX;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int i0 = bx * TX, j0 = by * TY;
    const int i = i0 + tx, j = j0 + ty;
    const bool valid = (i < N1) && (j < N2);
    const long pl = d.plane;
    const int kbeg = run.y & 0xFFFF, kend = run.y >> 16;
    const int P = d.P;
    const int own = (ty + 2) * LW + tx + 2;
    const unsigned cij = valid ? (unsigned)(j * N1 + i) : 0u;
    const unsigned cx = valid ? (unsigned)(j * N1 + min(i + 1, N1 - 1)) : 0u;      // (i+1, j)
    const unsigned cy = valid ? (unsigned)(min(j + 1, N2 - 1) * N1 + i) : 0u;      // (i, j+1)
    const bool accA = ACC && accP != nullptr, accK = ACC && pkP != nullptr;

    // halo tasks: [Syy-y, Sxy-y] 2*256 (task A: every thread), then [Syz-y] 256, [Sxx-x, Sxy-x, Sxz-x] 3*32
    HaloTask ta, tb;
    ytask(tid & 255, (tid >> 8) ? 2 : 1, i0, j0, N1, N2, ta);
    {
        const int t2 = tid;     // second task index in [0, 256+96)
        if (t2 < YT) ytask(t2, 4, i0, j0, N1, N2, tb);
        else if (t2 < YT + 3 * XT) {
            const int u = t2 - YT;
            const int a = u / XT;
            xtask(u % XT, a == 0 ? 0 : (a == 1 ? 2 : 3), i0, j0, N1, N2, tb);
        } else { tb.lofs = -1; tb.ok = false; tb.arr = 0; tb.gofs = 0; }
    }
    const float *pa = (ta.arr == 1 ? d.Syy : d.Sxy) + (ta.ok ? ta.gofs : 0);
    const float *pb = (tb.arr == 4 ? d.Syz : (tb.arr == 0 ? d.Sxx : (tb.arr == 2 ? d.Sxy : d.Sxz))) + (tb.ok ? tb.gofs : 0);
    float *la = &sS[0][ta.arr][ta.lofs];
    float *lb = &sS[0][tb.arr][tb.lofs < 0 ? 0 : tb.lofs];
    const bool hasB = tb.lofs >= 0;

    const bool zi = valid && (i < P || i >= N1 - P);
    const bool zj = valid && (j < P || j >= N2 - P);
    const bool inner = valid && i >= d.ND && i < N1 - d.ND && j >= d.ND && j < N2 - d.ND;

    // z queues: Szz k-1..k+2 ; Sxz, Syz k-2..k+1 ; in-plane arrays one plane ahead
    float zzm1 = 0, zz0 = 0, zzp1 = 0, zzp2 = 0, xzm2 = 0, xzm1 = 0, xz0 = 0, xzp1 = 0, yzm2 = 0, yzm1 = 0, yz0 = 0, yzp1 = 0;
    float sxx = 0, syy = 0, sxy = 0;
    if (valid) {
        const float *bzz = d.Szz + kbeg * pl, *bxz = d.Sxz + kbeg * pl, *byz = d.Syz + kbeg * pl;
        zzm1 = F4((bzz - pl), cij * 4u); zz0 = F4(bzz, cij * 4u); zzp1 = F4((bzz + pl), cij * 4u); zzp2 = F4((bzz + 2 * pl), cij * 4u);
        xzm2 = F4((bxz - 2 * pl), cij * 4u); xzm1 = F4((bxz - pl), cij * 4u); xz0 = F4(bxz, cij * 4u); xzp1 = F4((bxz + pl), cij * 4u);
        yzm2 = F4((byz - 2 * pl), cij * 4u); yzm1 = F4((byz - pl), cij * 4u); yz0 = F4(byz, cij * 4u); yzp1 = F4((byz + pl), cij * 4u);
        sxx = F4((d.Sxx + kbeg * pl), cij * 4u); syy = F4((d.Syy + kbeg * pl), cij * 4u); sxy = F4((d.Sxy + kbeg * pl), cij * 4u);
    }
    // pipelined like the fluid bodies: V, accumulators and material ids of plane kl are in registers when its
    // iteration starts (own id two planes ahead: the z face needs 1/rho of plane kl+1)
    float vx = 0, vy = 0, vz = 0, av = 0, pv = 0, r0 = 0;
    unsigned mraw = 0, mraw1 = 0, mx = 0, my = 0;
    if (valid) {
        const uint16_t *bM = d.mat + kbeg * pl;
        vx = F4((d.Vx + kbeg * pl), cij * 4u); vy = F4((d.Vy + kbeg * pl), cij * 4u); vz = F4((d.Vz + kbeg * pl), cij * 4u);
        mraw = U2(bM, cij * 2u); mraw1 = U2((bM + pl), cij * 2u); mx = U2(bM, cx * 2u); my = U2(bM, cy * 2u);
        if (accA) av = F4((accP + kbeg * pl), cij * 4u);
        if (accK) pv = F4((pkP + kbeg * pl), cij * 4u);
        r0 = d.invRho[mraw & BFD_MAT_MASK];
    }
    float ha = ta.ok ? pa[kbeg * pl] : 0.0f;
    float hb = tb.ok ? pb[kbeg * pl] : 0.0f;

    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        const long ko = (long)kl * pl;
        const int k = d.k0 + kl;
        const int bo = b * (5 * LH * LW);
        sS[b][0][own] = sxx; sS[b][1][own] = syy; sS[b][2][own] = sxy; sS[b][3][own] = xz0; sS[b][4][own] = yz0;
        la[bo] = ha;
        if (hasB) lb[bo] = hb;
        float r1 = 0, rx = 0, ry = 0;
        if (valid) {
            r1 = d.invRho[mraw1 & BFD_MAT_MASK];       // plane kl+1, becomes r0 of the next iteration
            rx = d.invRho[mx & BFD_MAT_MASK];
            ry = d.invRho[my & BFD_MAT_MASK];
        }
        __syncthreads();

        const float *pVx = d.Vx + ko, *pVy = d.Vy + ko, *pVz = d.Vz + ko;
        float *wVx = d.VxW + ko, *wVy = d.VyW + ko, *wVz = d.VzW + ko;
        float nzz = 0, nxz = 0, nyz = 0, nxx = 0, nyy = 0, nxy = 0, nha = 0, nhb = 0;
        float nvx = 0, nvy = 0, nvz = 0, nav = 0, npv = 0;
        unsigned nm2 = 0, nmx = 0, nmy = 0;
        if (valid) nm2 = U2((d.mat + ko + 2 * pl), cij * 2u);              // ghost planes make kl+2 addressable
        if (kl + 1 < kend) {
            if (valid) {
                nzz = F4((d.Szz + ko + 3 * pl), cij * 4u); nxz = F4((d.Sxz + ko + 2 * pl), cij * 4u); nyz = F4((d.Syz + ko + 2 * pl), cij * 4u);
                nxx = F4((d.Sxx + ko + pl), cij * 4u); nyy = F4((d.Syy + ko + pl), cij * 4u); nxy = F4((d.Sxy + ko + pl), cij * 4u);
                nvx = F4((pVx + pl), cij * 4u); nvy = F4((pVy + pl), cij * 4u); nvz = F4((pVz + pl), cij * 4u);
                nmx = U2((d.mat + ko + pl), cx * 2u); nmy = U2((d.mat + ko + pl), cy * 2u);
                if (accA) nav = F4((accP + ko + pl), cij * 4u);
                if (accK) npv = F4((pkP + ko + pl), cij * 4u);
            }
            if (ta.ok) nha = pa[ko + pl];
            if (tb.ok) nhb = pb[ko + pl];
        }

        if (valid) {
            if (ACC) {
                // Pressure RMS / peak of this step's final stresses (stress sources were injected
                // before this kernel), outside the absorbing layer only
                if (inner && k >= d.ND && k < d.N3 - d.ND) {
                    const float s = (sxx + syy) + zz0;
                    const float p = -s * (1.0f / 3.0f);
                    if (accA) F4((accP + ko), cij * 4u) = av + p * p;
                    if (accK) { const float ap = fabsf(p); if (ap > pv) F4((pkP + ko), cij * 4u) = ap; }
                }
            }
            if (mraw & BFD_REFLECTOR_BIT) {
                F4(wVx, cij * 4u) = 0.f; F4(wVy, cij * 4u) = 0.f; F4(wVz, cij * 4u) = 0.f;
            } else {
                const float *pxx = &sS[b][0][own], *pyy = &sS[b][1][own], *pxy = &sS[b][2][own];
                const float *pxz = &sS[b][3][own], *pyz = &sS[b][4][own];
                float dxSxx = dplus4(pxx[-1], sxx, pxx[1], pxx[2]);
                float dySxy = dminus4(pxy[-2 * LW], pxy[-LW], sxy, pxy[LW]);
                float dzSxz = dminus4(xzm2, xzm1, xz0, xzp1);
                float dxSxy = dminus4(pxy[-2], pxy[-1], sxy, pxy[1]);
                float dySyy = dplus4(pyy[-LW], syy, pyy[LW], pyy[2 * LW]);
                float dzSyz = dminus4(yzm2, yzm1, yz0, yzp1);
                float dxSxz = dminus4(pxz[-2], pxz[-1], xz0, pxz[1]);
                float dySyz = dminus4(pyz[-2 * LW], pyz[-LW], yz0, pyz[LW]);
                float dzSzz = dplus4(zzm1, zz0, zzp1, zzp2);
                if (zi) {
                    const int xi = i < P ? i : i - (N1 - 2 * P);
                    const unsigned q = (unsigned)((kl * N2 + j) * (2 * P) + xi);
                    dxSxx = cpml(d.psi[9], q, d.axH[i], d.bxH[i], dxSxx);
                    dxSxy = cpml(d.psi[12], q, d.axI[i], d.bxI[i], dxSxy);
                    dxSxz = cpml(d.psi[15], q, d.axI[i], d.bxI[i], dxSxz);
                }
                if (zj) {
                    const int yj = j < P ? j : j - (N2 - 2 * P);
                    const unsigned q = (unsigned)((kl * (2 * P) + yj) * N1 + i);
                    dySxy = cpml(d.psi[10], q, d.ayI[j], d.byI[j], dySxy);
                    dySyy = cpml(d.psi[13], q, d.ayH[j], d.byH[j], dySyy);
                    dySyz = cpml(d.psi[16], q, d.ayI[j], d.byI[j], dySyz);
                }
                if (k < P || k >= d.N3 - P) {
                    const int zk = k < P ? k : k - (d.N3 - 2 * P);
                    const unsigned q = (unsigned)(zk * d.plane) + cij;
                    dzSxz = cpml(d.psi[11], q, d.azI[k], d.bzI[k], dzSxz);
                    dzSyz = cpml(d.psi[14], q, d.azI[k], d.bzI[k], dzSyz);
                    dzSzz = cpml(d.psi[17], q, d.azH[k], d.bzH[k], dzSzz);
                }
                const float bxv = 0.5f * (r0 + rx), byv = 0.5f * (r0 + ry), bzv = 0.5f * (r0 + r1);
                ST4(wVx, cij * 4u, vx + bxv * ((dxSxx + dySxy) + dzSxz));
                ST4(wVy, cij * 4u, vy + byv * ((dxSxy + dySyy) + dzSyz));
                ST4(wVz, cij * 4u, vz + bzv * ((dxSxz + dySyz) + dzSzz));
            }
        }
        zzm1 = zz0; zz0 = zzp1; zzp1 = zzp2; zzp2 = nzz;
        xzm2 = xzm1; xzm1 = xz0; xz0 = xzp1; xzp1 = nxz;
        yzm2 = yzm1; yzm1 = yz0; yz0 = yzp1; yzp1 = nyz;
        sxx = nxx; syy = nyy; sxy = nxy;
        ha = nha; hb = nhb;
        vx = nvx; vy = nvy; vz = nvz; av = nav; pv = npv;
        r0 = r1; mraw = mraw1; mraw1 = nm2; mx = nmx; my = nmy;
    }
}


// ------------------------------------------------------------------------------------------------
// QUIET runs (round 6, production calls only: bfd_dev::act != null). Ahead of the wave front every field is EXACTLY zero (float32 with
// denormals flushed: the numerical precursors of the front die out some 25 cells ahead of it), and a half-step maps an all-zero
// neighbourhood onto itself. bfd_dev::act holds one byte per 64 x 8 x SUBZ sub-tile (padded by one sub-tile on every side, border 0):
// 1 = a kernel has written a non-zero V or S value into the sub-tile at some time (set-only; the sub-tiles that hold source voxels carry
// it from the start). A run whose own sub-tiles and all their 26 neighbours are clear returns at entry: everything it would read is zero,
// everything it would write is the zero that is there already (memory variables and absorbing-layer variables included: they are driven
// by the same values). A run that does work ORs the bits of what it stores and sets the byte of its sub-tiles if anything was non-zero.
// Flags only ever get set, so reading a neighbour's byte while that neighbour sets it in the same launch errs on the side of working (the
// decision is taken once per workgroup: run_all_quiet).
// The sparse kernel has no part in this: a non-zero it writes lies within 2 cells of a non-zero V, whose sub-tile is flagged, and every run
// within reach of that cell is a neighbour of that sub-tile. In a Z-slab the runs next to a neighbour's planes always work (run_all_quiet). Bit-identical to running every run (the -0 a skipped update might have
// produced compares equal to the +0 that stays).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool run_all_quiet(const bfd_dev &d, int bx, int by, int kbeg, int kend)
{
    const int q0 = kbeg / SUBZ, q1 = (kend - 1) / SUBZ;
    const int nq = q1 - q0 + 3;                             // the run's sub-tiles, one more below and one above
    if (9 * nq > 64) return false;
    // Z-slab: the runs of the first and the last sub-tile read the ghost planes a neighbour fills: nothing here knows when those turn non-zero, so these runs
    // always work -- and set their bytes when they write a non-zero, which is how the rest of the slab learns that the wave has come in
    if (kbeg < d.actLo || kend > d.actHi) return false;
    const int l = threadIdx.x;                              // lane: a wave is one tile row
    const int dq = l / 9, r = l - 9 * dq, dy = r / 3, dx = r - 3 * dy;
    // padded coordinates: sub-tile (bx, by, q) sits at (bx + 1, by + 1, q + 1), so its lower neighbour is at (bx, by, q)
    const unsigned idx = (unsigned)(((q0 + dq) * d.actY + (by + dy)) * d.actX + (bx + dx));
    const unsigned f = (l < 9 * nq) ? (unsigned)d.act[idx] : 0u;
    // ONE answer per workgroup: every wave reads the bytes for itself, and a neighbour may set its byte between two of those reads -- a wave that
    // left while its siblings stayed would leave their LDS tile with rows nobody filled. If any wave saw a set byte, all of them work.
    return __syncthreads_or(f != 0u) == 0;
}
__device__ __forceinline__ void run_mark_active(const bfd_dev &d, int bx, int by, int kbeg, int kend, unsigned nzbits)
{
    if (__ballot((nzbits & 0x7fffffffu) != 0u) == 0ull) return;
    if (threadIdx.x == 0)
        for (int q = kbeg / SUBZ; q <= (kend - 1) / SUBZ; q++) d.act[((q + 1) * d.actY + by + 1) * d.actX + bx + 1] = 1;
}
__device__ __forceinline__ unsigned fbits(float v) { return __float_as_uint(v); }

// ------------------------------------------------------------------------------------------------
// FLUID tiles: no solid cell within the tile grown by 2 cells in every direction. There
//   * shear stresses and their memory variables are never updated (a shear update needs the 4
//     cells around an edge to be solid), so they stay exactly 0 and need not be read;
//   * the three normal stresses (and their memory variables) receive the same update
//     (AS2 = BS2 = 0 in a fluid cell), so Sxx == Syy == Szz bit for bit: one is read, the
//     identical result is written to all three (the arrays stay fully valid for neighbours).
// Values equal the canonical sequence (only the sign of an exact zero can differ).
// ------------------------------------------------------------------------------------------------
// Fluid-tile bodies are specialised per tile on two more properties found at setup (classify_tiles):
//   UNI : the tile grown by 2 cells holds ONE material and no reflector -> coefficients are per-tile
//         scalars; no id loads, table lookups or reflector tests;
//   PML : some cell of the tile lies in an absorbing-layer zone (otherwise no CPML code at all).
// One launch covers all fluid tiles of a half-step in their natural (XCD-remapped) order; the workgroup
// switches on its tile's flags (block-uniform) into the matching instantiation.
// All bodies are software pipelined: every value plane kl needs is in registers when its iteration starts
// and the iteration issues the loads for plane kl+1 (CPML memory variables included).
// SOLID (round 5, compact solid state): the run may hold solid cells. Their Szz / Rzz take the solid formula (AS2, BS2 of the cell's material, 0 in
// a fluid: the same expression serves every lane, equal to the fluid one up to the sign of an exact zero); everything else a solid cell has --
// Sxx, Syy, the shear stresses and their memory variables -- is the sparse kernel's (stress_shear_sparse, which runs after this one and reads
// the absorbing-layer memory variables this kernel has just advanced).
template <bool LOSSY, bool COLLAPSED, bool UNI, bool PML, bool SOLID = false, bool QUIET = false>
__device__ __forceinline__ void stress_fluid_body(const bfd_dev &d, int bx, int by, int kbeg, int kend, int tm,
                                                  float (*sV)[2][LH * LW])
{
    unsigned nzb = 0u;          // QUIET: bits of everything stored
    // The three normal stresses are identical in a FLUID tile; Szz is the one that is read (it is
    // also the one whose ghost planes the Z-neighbour exchange carries), all three are written.
    const int N1 = d.N1, N2 = d.N2;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int i0 = bx * TX, j0 = by * TY;
    const int i = i0 + tx, j = j0 + ty;
    const bool valid = (i < N1) && (j < N2);
    const long pl = d.plane;
    const int P = d.P;
    const int own = (ty + 2) * LW + tx + 2;
    const unsigned cij = valid ? (unsigned)(j * N1 + i) : 0u;

    // halo tasks: Vy rows above/below (256), Vx columns left/right (32)
    HaloTask t; t.lofs = -1; t.ok = false; t.arr = 0; t.gofs = 0;
    if (tid < YT) ytask(tid, 1, i0, j0, N1, N2, t);
    else if (tid < YT + XT) xtask(tid - YT, 0, i0, j0, N1, N2, t);
    const bool has = t.lofs >= 0;
    // the array of the task is uniform per wave (waves 0-3: Vy rows, wave 4: Vx columns): SGPR base + 32-bit byte offset
    const float *ph = __builtin_amdgcn_readfirstlane(t.arr) == 0 ? d.Vx : d.Vy;
    const unsigned hofs = t.ok ? (unsigned)t.gofs * 4u : 0u;
    float *lh = &sV[0][t.arr][has ? t.lofs : 0];

    const float c1 = d.c1;
    float APu = 0.f, BPu = 0.f;
    if (UNI) { APu = d.AP[tm]; BPu = d.BP[tm]; }

    // absorbing layer (PML flavours only): per-thread x/y coefficients and running zone offsets
    const bool zi = PML && valid && (i < P || i >= N1 - P);
    const bool zj = PML && valid && (j < P || j >= N2 - P);
    float ax = 0, bxc = 0, ay = 0, byc = 0, px = 0, py = 0, pz = 0;
    unsigned qx = 0, qy = 0;
    const unsigned dqx = (unsigned)(N2 * 2 * P), dqy = (unsigned)(2 * P * N1);
    if (PML) {
        if (zi) { ax = d.axI[i]; bxc = d.bxI[i]; qx = (unsigned)((kbeg * N2 + j) * (2 * P) + (i < P ? i : i - (N1 - 2 * P))); px = F4(d.psi[0], (unsigned)(qx) * 4u); }
        if (zj) { ay = d.ayI[j]; byc = d.byI[j]; qy = (unsigned)((kbeg * (2 * P) + (j < P ? j : j - (N2 - 2 * P))) * N1 + i); py = F4(d.psi[1], (unsigned)(qy) * 4u); }
        const int kg = d.k0 + kbeg;
        if (valid && (kg < P || kg >= d.N3 - P)) pz = F4(d.psi[2], (unsigned)((long)(kg < P ? kg : kg - (d.N3 - 2 * P)) * pl + cij) * 4u);
    }

    float vx0 = 0, vy0 = 0, vzm2 = 0, vzm1 = 0, vz0 = 0, vzp1 = 0, szz = 0, rzz = 0;
    unsigned mraw = 0;
    if (valid) {
        const float *bVz = d.Vz + kbeg * pl;
        vx0 = F4((d.Vx + kbeg * pl), cij * 4u); vy0 = F4((d.Vy + kbeg * pl), cij * 4u);
        vzm2 = LD4((bVz - 2 * pl), cij * 4u); vzm1 = LD4((bVz - pl), cij * 4u); vz0 = LD4(bVz, cij * 4u); vzp1 = LD4((bVz + pl), cij * 4u);
        szz = LD4((d.Szz + kbeg * pl), cij * 4u);
        if (LOSSY) rzz = LD4((d.Rzz + kbeg * pl), cij * 4u);
        if (!UNI) mraw = U2((d.mat + kbeg * pl), cij * 2u);
    }
    float hv = t.ok ? F4(ph + kbeg * pl, hofs) : 0.0f;

    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        const long ko = (long)kl * pl;
        const int k = d.k0 + kl;
        sV[b][0][own] = vx0; sV[b][1][own] = vy0;
        if (has) lh[b * (2 * LH * LW)] = hv;
        // table rows of this plane (cache-resident; short latency, overlaps the barrier)
        float AP = APu, BP = BPu, AS2 = 0.f, BS2 = 0.f;
        if (!UNI && valid) { const int m = mraw & BFD_MAT_MASK; AP = d.AP[m]; if (LOSSY) BP = d.BP[m]; if (SOLID) { AS2 = d.AS2[m]; BS2 = d.BS2[m]; } }
        __syncthreads();

        float nvx = 0, nvy = 0, nvz = 0, nh = 0, nszz = 0, nrzz = 0, npx = 0, npy = 0, npz = 0;
        unsigned nmraw = 0;
        if (kl + 1 < kend) {
            if (valid) {
                nvx = F4((d.Vx + ko + pl), cij * 4u); nvy = F4((d.Vy + ko + pl), cij * 4u); nvz = LD4((d.Vz + ko + 2 * pl), cij * 4u);
                nszz = LD4((d.Szz + ko + pl), cij * 4u);
                if (LOSSY) nrzz = LD4((d.Rzz + ko + pl), cij * 4u);
                if (!UNI) nmraw = U2((d.mat + ko + pl), cij * 2u);
            }
            if (t.ok) nh = F4(ph + ko + pl, hofs);
            if (PML) {
                if (zi) npx = F4(d.psi[0], (unsigned)(qx + dqx) * 4u);
                if (zj) npy = F4(d.psi[1], (unsigned)(qy + dqy) * 4u);
                const int kn = k + 1;
                if (valid && (kn < P || kn >= d.N3 - P)) npz = F4(d.psi[2], (unsigned)((long)(kn < P ? kn : kn - (d.N3 - 2 * P)) * pl + cij) * 4u);
            }
        }
        if (valid) {
            const float *sx = &sV[b][0][own], *sy = &sV[b][1][own];
            float dxVx = dminus4(sx[-2], sx[-1], vx0, sx[1]);
            float dyVy = dminus4(sy[-2 * LW], sy[-LW], vy0, sy[LW]);
            float dzVz = dminus4(vzm2, vzm1, vz0, vzp1);
            float val = 0.f, rn = 0.f;
            if (UNI || !(mraw & BFD_REFLECTOR_BIT)) {
                if (PML) {
                    if (zi) { const float pn = bxc * px + ax * dxVx; F4(d.psi[0], (unsigned)(qx) * 4u) = pn; dxVx = dxVx + pn; }
                    if (zj) { const float pn = byc * py + ay * dyVy; F4(d.psi[1], (unsigned)(qy) * 4u) = pn; dyVy = dyVy + pn; }
                    if (k < P || k >= d.N3 - P) {
                        const float pn = d.bzI[k] * pz + d.azI[k] * dzVz;
                        F4(d.psi[2], (unsigned)((long)(k < P ? k : k - (d.N3 - 2 * P)) * pl + cij) * 4u) = pn;
                        dzVz = dzVz + pn;
                    }
                }
                const float sXY = dxVx + dyVy;
                const float div = sXY + dzVz;
                if (SOLID) {
                    rn = c1 * rzz - (BP * div - BS2 * sXY);
                    val = szz + ((AP * div - AS2 * sXY) + 0.5f * (rzz + rn));
                } else if (LOSSY) {
                    rn = c1 * rzz - BP * div;
                    val = szz + (AP * div + 0.5f * (rzz + rn));
                } else {
                    val = szz + AP * div;
                }
            }
            // COLLAPSED (no solid tile in the slab, no per-component stress output selected): nobody
            // reads Sxx/Syy/Rxx/Ryy, so only the Szz/Rzz copy is kept (expanded on demand, bfd_api.hip)
            ST4((d.SzzW + ko), cij * 4u, val);
            if (QUIET) nzb |= fbits(val) | fbits(rn);
            if (!COLLAPSED) { F4((d.Sxx + ko), cij * 4u) = val; F4((d.Syy + ko), cij * 4u) = val; }
            if (LOSSY) {
                ST4((d.RzzW + ko), cij * 4u, rn);
                if (!COLLAPSED) { F4((d.Rxx + ko), cij * 4u) = rn; F4((d.Ryy + ko), cij * 4u) = rn; }
            }
        }
        vx0 = nvx; vy0 = nvy;
        vzm2 = vzm1; vzm1 = vz0; vz0 = vzp1; vzp1 = nvz;
        hv = nh; szz = nszz; rzz = nrzz; mraw = nmraw;
        px = npx; py = npy; pz = npz; qx += dqx; qy += dqy;
    }
    if (QUIET) run_mark_active(d, bx, by, kbeg, kend, nzb);
}

template <bool ACC, bool UNI, bool PML, bool QUIET = false>
__device__ __forceinline__ void velocity_fluid_body(const bfd_dev &d, int bx, int by, int kbeg, int kend, int tm,
                                                    float (*sS)[LH * LW], float *__restrict__ accP, float *__restrict__ pkP)
{
    const int N1 = d.N1, N2 = d.N2;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int i0 = bx * TX, j0 = by * TY;
    const int i = i0 + tx, j = j0 + ty;
    const bool valid = (i < N1) && (j < N2);
    const long pl = d.plane;
    const int P = d.P;
    const int own = (ty + 2) * LW + tx + 2;
    const unsigned cij = valid ? (unsigned)(j * N1 + i) : 0u;
    const unsigned cx = valid ? (unsigned)(j * N1 + min(i + 1, N1 - 1)) : 0u;      // (i+1, j)
    const unsigned cy = valid ? (unsigned)(min(j + 1, N2 - 1) * N1 + i) : 0u;      // (i, j+1)

    // halo ring of Szz (== Sxx == Syy here): rows (256) and columns (32)
    HaloTask t; t.lofs = -1; t.ok = false; t.arr = 0; t.gofs = 0;
    if (tid < YT) ytask(tid, 0, i0, j0, N1, N2, t);
    else if (tid < YT + XT) xtask(tid - YT, 0, i0, j0, N1, N2, t);
    const bool has = t.lofs >= 0;
    const float *ph = d.Szz;
    const unsigned hofs = t.ok ? (unsigned)t.gofs * 4u : 0u;
    float *lh = &sS[0][has ? t.lofs : 0];

    const bool inner = valid && i >= d.ND && i < N1 - d.ND && j >= d.ND && j < N2 - d.ND;
    const bool accA = ACC && accP != nullptr, accK = ACC && pkP != nullptr;
    unsigned nzb = 0u;          // QUIET: bits of everything stored
    float ru = 0.f;
    if (UNI) ru = d.invRho[tm];                       // 0.5*(r+r) == r exactly: every face of a UNI tile sees one 1/rho

    const bool zi = PML && valid && (i < P || i >= N1 - P);
    const bool zj = PML && valid && (j < P || j >= N2 - P);
    float ax = 0, bxc = 0, ay = 0, byc = 0, px = 0, py = 0, pz = 0;
    unsigned qx = 0, qy = 0;
    const unsigned dqx = (unsigned)(N2 * 2 * P), dqy = (unsigned)(2 * P * N1);
    if (PML) {
        if (zi) { ax = d.axH[i]; bxc = d.bxH[i]; qx = (unsigned)((kbeg * N2 + j) * (2 * P) + (i < P ? i : i - (N1 - 2 * P))); px = F4(d.psi[9], (unsigned)(qx) * 4u); }
        if (zj) { ay = d.ayH[j]; byc = d.byH[j]; qy = (unsigned)((kbeg * (2 * P) + (j < P ? j : j - (N2 - 2 * P))) * N1 + i); py = F4(d.psi[13], (unsigned)(qy) * 4u); }
        const int kg = d.k0 + kbeg;
        if (valid && (kg < P || kg >= d.N3 - P)) pz = F4(d.psi[17], (unsigned)((long)(kg < P ? kg : kg - (d.N3 - 2 * P)) * pl + cij) * 4u);
    }

    // own material id runs two planes ahead because the z face needs 1/rho of plane kl+1; neighbour ids
    // one plane ahead; table rows at the top of the iteration
    float sm1 = 0, s0 = 0, sp1 = 0, sp2 = 0, vx = 0, vy = 0, vz = 0, av = 0, r0 = ru;
    unsigned mraw = 0, mraw1 = 0, mx = 0, my = 0;
    if (valid) {
        const float *bS = d.Szz + kbeg * pl;
        sm1 = F4((bS - pl), cij * 4u); s0 = F4(bS, cij * 4u); sp1 = F4((bS + pl), cij * 4u); sp2 = F4((bS + 2 * pl), cij * 4u);
        vx = LD4((d.Vx + kbeg * pl), cij * 4u); vy = LD4((d.Vy + kbeg * pl), cij * 4u); vz = LD4((d.Vz + kbeg * pl), cij * 4u);
        if (accA) av = LD4((accP + kbeg * pl), cij * 4u);
        if (!UNI) {
            const uint16_t *bM = d.mat + kbeg * pl;
            mraw = U2(bM, cij * 2u); mraw1 = U2((bM + pl), cij * 2u); mx = U2(bM, cx * 2u); my = U2(bM, cy * 2u);
            r0 = d.invRho[mraw & BFD_MAT_MASK];
        }
    }
    float hv = t.ok ? F4(ph + kbeg * pl, hofs) : 0.0f;

    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        const long ko = (long)kl * pl;
        const int k = d.k0 + kl;
        sS[b][own] = s0;
        if (has) lh[b * (LH * LW)] = hv;
        float r1 = ru, rx = ru, ry = ru;
        if (!UNI && valid) {
            r1 = d.invRho[mraw1 & BFD_MAT_MASK];       // plane kl+1, becomes r0 of the next iteration
            rx = d.invRho[mx & BFD_MAT_MASK];
            ry = d.invRho[my & BFD_MAT_MASK];
        }
        __syncthreads();

        float ns = 0, nh = 0, nvx = 0, nvy = 0, nvz = 0, nav = 0, npx = 0, npy = 0, npz = 0;
        unsigned nm2 = 0, nmx = 0, nmy = 0;
        if (!UNI && valid) nm2 = U2((d.mat + ko + 2 * pl), cij * 2u);     // ghost planes make kl+2 addressable
        if (kl + 1 < kend) {
            if (valid) {
                ns = F4((d.Szz + ko + 3 * pl), cij * 4u);
                nvx = LD4((d.Vx + ko + pl), cij * 4u); nvy = LD4((d.Vy + ko + pl), cij * 4u); nvz = LD4((d.Vz + ko + pl), cij * 4u);
                if (!UNI) { nmx = U2((d.mat + ko + pl), cx * 2u); nmy = U2((d.mat + ko + pl), cy * 2u); }
                if (accA) nav = LD4((accP + ko + pl), cij * 4u);
            }
            if (t.ok) nh = F4(ph + ko + pl, hofs);
            if (PML) {
                if (zi) npx = F4(d.psi[9], (unsigned)(qx + dqx) * 4u);
                if (zj) npy = F4(d.psi[13], (unsigned)(qy + dqy) * 4u);
                const int kn = k + 1;
                if (valid && (kn < P || kn >= d.N3 - P)) npz = F4(d.psi[17], (unsigned)((long)(kn < P ? kn : kn - (d.N3 - 2 * P)) * pl + cij) * 4u);
            }
        }
        if (valid) {
            if (ACC) {
                if (inner && k >= d.ND && k < d.N3 - d.ND) {
                    const float s = (s0 + s0) + s0;
                    const float p = -s * (1.0f / 3.0f);
                    if (accA) ST4((accP + ko), cij * 4u, av + p * p);
                    // the running peak is read here, not a plane ahead: one live register fewer (the accumulating flavour spilled one at 8 waves)
                    if (accK) { const float ap = fabsf(p); if (ap > F4((pkP + ko), cij * 4u)) F4((pkP + ko), cij * 4u) = ap; }
                }
            }
            if (!UNI && (mraw & BFD_REFLECTOR_BIT)) {
                F4((d.VxW + ko), cij * 4u) = 0.f; F4((d.VyW + ko), cij * 4u) = 0.f; F4((d.VzW + ko), cij * 4u) = 0.f;
            } else {
                const float *p = &sS[b][own];
                float dx = dplus4(p[-1], s0, p[1], p[2]);
                float dy = dplus4(p[-LW], s0, p[LW], p[2 * LW]);
                float dz = dplus4(sm1, s0, sp1, sp2);
                if (PML) {
                    if (zi) { const float pn = bxc * px + ax * dx; F4(d.psi[9], (unsigned)(qx) * 4u) = pn; dx = dx + pn; }
                    if (zj) { const float pn = byc * py + ay * dy; F4(d.psi[13], (unsigned)(qy) * 4u) = pn; dy = dy + pn; }
                    if (k < P || k >= d.N3 - P) {
                        const float pn = d.bzH[k] * pz + d.azH[k] * dz;
                        F4(d.psi[17], (unsigned)((long)(k < P ? k : k - (d.N3 - 2 * P)) * pl + cij) * 4u) = pn;
                        dz = dz + pn;
                    }
                }
                if (!QUIET) {
                    ST4((d.VxW + ko), cij * 4u, vx + (0.5f * (r0 + rx)) * dx);
                    ST4((d.VyW + ko), cij * 4u, vy + (0.5f * (r0 + ry)) * dy);
                    ST4((d.VzW + ko), cij * 4u, vz + (0.5f * (r0 + r1)) * dz);
                } else {
                    { const float w = vx + (0.5f * (r0 + rx)) * dx; ST4((d.VxW + ko), cij * 4u, w); nzb |= fbits(w); }
                    { const float w = vy + (0.5f * (r0 + ry)) * dy; ST4((d.VyW + ko), cij * 4u, w); nzb |= fbits(w); }
                    { const float w = vz + (0.5f * (r0 + r1)) * dz; ST4((d.VzW + ko), cij * 4u, w); nzb |= fbits(w); }
                }
            }
        }
        sm1 = s0; s0 = sp1; sp1 = sp2; sp2 = ns;
        hv = nh; vx = nvx; vy = nvy; vz = nvz; av = nav;
        r0 = r1; mraw = mraw1; mraw1 = nm2; mx = nmx; my = nmy;
        px = npx; py = npy; pz = npz; qx += dqx; qy += dqy;
    }
    if (QUIET) run_mark_active(d, bx, by, kbeg, kend, nzb);
}

// ------------------------------------------------------------------------------------------------
// SOLID runs in the tiled path (variant 3): normal stresses by a pipelined marching kernel (stress_solid), shear
// stresses by a sparse per-cell kernel (stress_shear_sparse) over the list of cells with a solid centre. Each
// kernel is lean (no long dependent chains behind a workgroup barrier), which the monolithic stress_v2 is not.
// Values equal stress_v2's; CPML memory variables of the cross derivatives are advanced only at listed cells
// (they feed nothing else).
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// SOLID runs, class-predicated kernels (variants 0/3/4). A solid run is a 64 x 8 x (8..16) block that has a solid cell
// within 2 cells, but typically two thirds of its cells are fluid (a thin curved shell cuts through it). Every lane
// therefore works by the class byte of ITS cell (bfd_dev::cls, computed at setup):
//   * fluid cell: the three normal stresses are identical -> only Szz/Rzz is read and written (nobody reads Sxx/Syy of a
//     fluid cell: every reader substitutes Szz there, also across tile borders);
//   * a shear stress entry is loaded only where its edge bit says it is ever updated (elsewhere it is exactly 0).
// Loads are per-lane predicated, so only the lines of the cells that need an array are fetched; the arithmetic is the
// canonical sequence for every lane (zeros / substituted values make it equal to the dense kernels' bit for bit, up
// to the sign of an exact zero). Class bytes run two planes ahead of the state loads they steer.
// ------------------------------------------------------------------------------------------------
template <bool PML>
__device__ __forceinline__ void stress_solid_body(const bfd_dev &d, const int4 &run, int tilesX, float (*sV)[2][LH * LW])
{
    const int N1 = d.N1, N2 = d.N2;
    const int bx = run.x % tilesX, by = run.x / tilesX, kbeg = run.y & 0xFFFF, kend = run.y >> 16;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int i0 = bx * TX, j0 = by * TY;
    const int i = i0 + tx, j = j0 + ty;
    const bool valid = (i < N1) && (j < N2);
    const long pl = d.plane;
    const int P = d.P;
    const int own = (ty + 2) * LW + tx + 2;
    const unsigned cij = valid ? (unsigned)(j * N1 + i) : 0u;

    HaloTask t; t.lofs = -1; t.ok = false; t.arr = 0; t.gofs = 0;
    if (tid < YT) ytask(tid, 1, i0, j0, N1, N2, t);
    else if (tid < YT + XT) xtask(tid - YT, 0, i0, j0, N1, N2, t);
    const bool has = t.lofs >= 0;
    // the array of the task is uniform per wave (waves 0-3: Vy rows, wave 4: Vx columns): SGPR base + 32-bit byte offset
    const float *ph = __builtin_amdgcn_readfirstlane(t.arr) == 0 ? d.Vx : d.Vy;
    const unsigned hofs = t.ok ? (unsigned)t.gofs * 4u : 0u;
    float *lh = &sV[0][t.arr][has ? t.lofs : 0];
    const float c1 = d.c1;

    const bool zi = PML && valid && (i < P || i >= N1 - P);
    const bool zj = PML && valid && (j < P || j >= N2 - P);
    float ax = 0, bxc = 0, ay = 0, byc = 0, px = 0, py = 0, pz = 0;
    unsigned qx = 0, qy = 0;
    const unsigned dqx = (unsigned)(N2 * 2 * P), dqy = (unsigned)(2 * P * N1);
    if (zi) { ax = d.axI[i]; bxc = d.bxI[i]; qx = (unsigned)((kbeg * N2 + j) * (2 * P) + (i < P ? i : i - (N1 - 2 * P))); px = F4(d.psi[0], (unsigned)(qx) * 4u); }
    if (zj) { ay = d.ayI[j]; byc = d.byI[j]; qy = (unsigned)((kbeg * (2 * P) + (j < P ? j : j - (N2 - 2 * P))) * N1 + i); py = F4(d.psi[1], (unsigned)(qy) * 4u); }
    if (PML) {
        const int kg = d.k0 + kbeg;
        if (valid && (kg < P || kg >= d.N3 - P)) pz = F4(d.psi[2], (unsigned)((unsigned)((kg < P ? kg : kg - (d.N3 - 2 * P)) * d.plane) + cij) * 4u);
    }

    float vx0 = 0, vy0 = 0, vzm2 = 0, vzm1 = 0, vz0 = 0, vzp1 = 0;
    float sxx = 0, syy = 0, szz = 0, rxx = 0, ryy = 0, rzz = 0;
    unsigned mraw = 0, cl = BFD_CLS_FLUID | BFD_CLS_NOMEM, cl1 = BFD_CLS_FLUID | BFD_CLS_NOMEM;
    if (valid) {
        const float *bVz = d.Vz + kbeg * pl;
        cl = U1((d.cls + kbeg * pl), cij); cl1 = U1((d.cls + kbeg * pl + pl), cij);
        vx0 = F4((d.Vx + kbeg * pl), cij * 4u); vy0 = F4((d.Vy + kbeg * pl), cij * 4u);
        vzm2 = F4((bVz - 2 * pl), cij * 4u); vzm1 = F4((bVz - pl), cij * 4u); vz0 = F4(bVz, cij * 4u); vzp1 = F4((bVz + pl), cij * 4u);
        mraw = U2((d.mat + kbeg * pl), cij * 2u);
        szz = F4((d.Szz + kbeg * pl), cij * 4u);
        const bool fl = cl & BFD_CLS_FLUID, mem = !(cl & BFD_CLS_NOMEM) || !fl;
        if (mem) rzz = F4((d.Rzz + kbeg * pl), cij * 4u);
        if (!fl) {
            sxx = F4((d.Sxx + kbeg * pl), cij * 4u); syy = F4((d.Syy + kbeg * pl), cij * 4u);
            rxx = F4((d.Rxx + kbeg * pl), cij * 4u); ryy = F4((d.Ryy + kbeg * pl), cij * 4u);
        }
    }
    float hv = t.ok ? F4(ph + kbeg * pl, hofs) : 0.0f;

    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        // opaque to loop strength reduction: otherwise every array gets a 64-bit per-lane pointer that is bumped each plane
        // (a VGPR pair per array); this way the plane bases are recomputed on the scalar unit and stay in SGPRs
        const long ko = (long)__builtin_amdgcn_readfirstlane(kl) * pl;
        const int k = d.k0 + kl;
        float nvx = 0, nvy = 0, nvz = 0, nh = 0, nsxx = 0, nsyy = 0, nszz = 0, nrxx = 0, nryy = 0, nrzz = 0, npx = 0, npy = 0, npz = 0;
        unsigned nmraw = 0, ncl2 = BFD_CLS_FLUID | BFD_CLS_NOMEM;
        auto prefetch_next = [&]() {
        if (valid) ncl2 = U1((d.cls + ko + 2 * pl), cij);           // ghost planes make kl+2 addressable
            if (kl + 1 < kend) {
                if (valid) {
                    const bool nfl = cl1 & BFD_CLS_FLUID, nmem = !(cl1 & BFD_CLS_NOMEM) || !nfl;
                    nvx = F4((d.Vx + ko + pl), cij * 4u); nvy = F4((d.Vy + ko + pl), cij * 4u); nvz = F4((d.Vz + ko + 2 * pl), cij * 4u);
                    nmraw = U2((d.mat + ko + pl), cij * 2u);
                    nszz = F4((d.Szz + ko + pl), cij * 4u);
                    if (nmem) nrzz = F4((d.Rzz + ko + pl), cij * 4u);
                    if (!nfl) {
                        nsxx = F4((d.Sxx + ko + pl), cij * 4u); nsyy = F4((d.Syy + ko + pl), cij * 4u);
                        nrxx = F4((d.Rxx + ko + pl), cij * 4u); nryy = F4((d.Ryy + ko + pl), cij * 4u);
                    }
                }
                if (t.ok) nh = F4(ph + ko + pl, hofs);
                if (zi) npx = F4(d.psi[0], (unsigned)(qx + dqx) * 4u);
                if (zj) npy = F4(d.psi[1], (unsigned)(qy + dqy) * 4u);
                const int kn = k + 1;
                if (PML && valid && (kn < P || kn >= d.N3 - P)) npz = F4(d.psi[2], (unsigned)((unsigned)((kn < P ? kn : kn - (d.N3 - 2 * P)) * d.plane) + cij) * 4u);
            }
        };
        sV[b][0][own] = vx0; sV[b][1][own] = vy0;
        if (has) lh[b * (2 * LH * LW)] = hv;
        const int m = mraw & BFD_MAT_MASK;
        const bool fl = cl & BFD_CLS_FLUID, mem = !(cl & BFD_CLS_NOMEM) || !fl;
        float AP = 0, BP = 0, AS2 = 0, BS2 = 0;
        if (valid) { AP = d.AP[m]; if (mem) BP = d.BP[m]; if (!fl) { AS2 = d.AS2[m]; BS2 = d.BS2[m]; } }
        __syncthreads();

        prefetch_next();        // after the barrier (issuing these loads before it was measured slower: 2.37 -> 2.52 ms per step at C2-medium 512^3)
        if (valid) {
            const float *sx = &sV[b][0][own], *sy = &sV[b][1][own];
            float dxVx = dminus4(sx[-2], sx[-1], vx0, sx[1]);
            float dyVy = dminus4(sy[-2 * LW], sy[-LW], vy0, sy[LW]);
            float dzVz = dminus4(vzm2, vzm1, vz0, vzp1);
            if (cl & BFD_CLS_REFL) {
                F4((d.Sxx + ko), cij * 4u) = 0.f; F4((d.Syy + ko), cij * 4u) = 0.f; F4((d.SzzW + ko), cij * 4u) = 0.f;
                F4((d.Rxx + ko), cij * 4u) = 0.f; F4((d.Ryy + ko), cij * 4u) = 0.f; F4((d.RzzW + ko), cij * 4u) = 0.f;
                F4((d.Sxy + ko), cij * 4u) = 0.f; F4((d.Sxz + ko), cij * 4u) = 0.f; F4((d.Syz + ko), cij * 4u) = 0.f;
                F4((d.Rxy + ko), cij * 4u) = 0.f; F4((d.Rxz + ko), cij * 4u) = 0.f; F4((d.Ryz + ko), cij * 4u) = 0.f;
            } else {
                if (zi) { const float pn = bxc * px + ax * dxVx; F4(d.psi[0], (unsigned)(qx) * 4u) = pn; dxVx = dxVx + pn; }
                if (zj) { const float pn = byc * py + ay * dyVy; F4(d.psi[1], (unsigned)(qy) * 4u) = pn; dyVy = dyVy + pn; }
                if (PML && (k < P || k >= d.N3 - P)) {
                    const float pn = d.bzI[k] * pz + d.azI[k] * dzVz;
                    F4(d.psi[2], (unsigned)((unsigned)((k < P ? k : k - (d.N3 - 2 * P)) * d.plane) + cij) * 4u) = pn;
                    dzVz = dzVz + pn;
                }
                const float sXY = dxVx + dyVy;
                const float div = sXY + dzVz;
                if (fl) {               // fluid cell: one copy of the identical normal stresses
                    float val;
                    if (!mem) val = szz + AP * div;
                    else {
                        const float rn = c1 * rzz - BP * div;
                        val = szz + (AP * div + 0.5f * (rzz + rn));
                        ST4((d.RzzW + ko), cij * 4u, rn);
                    }
                    ST4((d.SzzW + ko), cij * 4u, val);
                } else {
                    const float sYZ = dyVy + dzVz, sXZ = dxVx + dzVz;
                    float rn;
                    rn = c1 * rxx - (BP * div - BS2 * sYZ);
                    ST4((d.Sxx + ko), cij * 4u, sxx + ((AP * div - AS2 * sYZ) + 0.5f * (rxx + rn))); ST4((d.Rxx + ko), cij * 4u, rn);
                    rn = c1 * ryy - (BP * div - BS2 * sXZ);
                    ST4((d.Syy + ko), cij * 4u, syy + ((AP * div - AS2 * sXZ) + 0.5f * (ryy + rn))); ST4((d.Ryy + ko), cij * 4u, rn);
                    rn = c1 * rzz - (BP * div - BS2 * sXY);
                    ST4((d.SzzW + ko), cij * 4u, szz + ((AP * div - AS2 * sXY) + 0.5f * (rzz + rn))); ST4((d.RzzW + ko), cij * 4u, rn);
                }
            }
        }
        vx0 = nvx; vy0 = nvy;
        vzm2 = vzm1; vzm1 = vz0; vz0 = vzp1; vzp1 = nvz;
        hv = nh; mraw = nmraw; cl = cl1; cl1 = ncl2;
        sxx = nsxx; syy = nsyy; szz = nszz; rxx = nrxx; ryy = nryy; rzz = nrzz;
        px = npx; py = npy; pz = npz; qx += dqx; qy += dqy;
    }
}

#ifndef SOLID_STRESS_WAVES_PER_SIMD
#define SOLID_STRESS_WAVES_PER_SIMD 4
#endif
// (-DBFD_STRESS_SOLID_GLOBAL builds the GLOBAL / branch-free-prefetch body below instead: measured 1 % slower, 0.308-0.311 against
// 0.305 ms at the shear medium 512^3 -- this kernel is at the rate of the bytes it moves; profiles/r4/experiment_solid_kernels_prefetch.txt)
#ifndef BFD_STRESS_SOLID_GLOBAL
__global__ __launch_bounds__(NTHREADS, SOLID_STRESS_WAVES_PER_SIMD) void stress_solid(bfd_dev d, int tilesX, int nblocks, const int *__restrict__ xmap, const int4 *__restrict__ runs)
{
    __shared__ float sV[2][2][LH * LW];
    const int ri = run_index(nblocks, xmap);
    if (ri < 0) return;
    const int4 run = runs[ri];
    if (run.z & 8) stress_solid_body<true>(d, run, tilesX, sV);
    else stress_solid_body<false>(d, run, tilesX, sV);
}
#endif

// ------------------------------------------------------------------------------------------------
// SOLID runs, normal AND shear stresses in one pass (BFD_SOLID_MERGED=1): stress_solid_body plus the shear update of the
// lanes whose class byte carries an edge bit, from the V planes the kernel stages anyway -- all three components with
// their full ring in LDS, Vx / Vy queues k-1 .. k+2 in registers -- instead of the sparse kernel's 21-value gather, its
// list and one launch. Edge coefficients come from the per-material table (the four cells of an active edge hold ONE
// material: class bit MIXED clear); cells with an edge between different solids (BFD_CLS_MIXED) keep the sparse kernel,
// whose list then holds only them. The memory variables Rxy, Rxz, Ryz live in the full-volume arrays in this mode.
// CPML memory variables of the cross derivatives advance where their edge is active (each belongs to one edge of one cell).
// Same arithmetic and operation order as stress_v2 / stress_shear_sparse: bit-identical.
// ------------------------------------------------------------------------------------------------
template <bool PML>
__device__ __forceinline__ void stress_solid_merged_body(const bfd_dev &d, const int4 &run, int tilesX, const float *__restrict__ shearTab,
                                                         float (*sV)[3][LH * LW])
{
    const int N1 = d.N1, N2 = d.N2;
    const int bx = run.x % tilesX, by = run.x / tilesX, kbeg = run.y & 0xFFFF, kend = run.y >> 16;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int wv = __builtin_amdgcn_readfirstlane(ty);
    const int i0 = bx * TX, j0 = by * TY;
    const int i = i0 + tx, j = j0 + ty;
    const bool valid = (i < N1) && (j < N2);
    const long pl = d.plane;
    const int P = d.P;
    const int own = (ty + 2) * LW + tx + 2;
    const unsigned cij = valid ? (unsigned)(j * N1 + i) : 0u;

    // halo tasks, the array of a task uniform per wave: A: waves 0-3 the 4 halo rows of Vx, waves 4-7 those of Vy;
    // B: waves 0-3 the 4 halo rows of Vz; lanes 0..31 of waves 4 / 5 / 6 the halo columns of Vx / Vy / Vz
    HaloTask ta, tb;
    const int arrA = wv < 4 ? 0 : 1;
    ytask(tid & 255, arrA, i0, j0, N1, N2, ta);
    const int arrB = wv < 4 ? 2 : (wv == 4 ? 0 : (wv == 5 ? 1 : 2));
    if (wv < 4) ytask(tid, 2, i0, j0, N1, N2, tb);
    else if (wv < 7 && tx < XT) xtask(tx, arrB, i0, j0, N1, N2, tb);
    else { tb.lofs = -1; tb.ok = false; tb.arr = arrB; tb.gofs = 0; }
    const float *baseA = arrA == 0 ? d.Vx : d.Vy;
    const float *baseB = arrB == 0 ? d.Vx : (arrB == 1 ? d.Vy : d.Vz);
    const unsigned offA = ta.ok ? (unsigned)ta.gofs * 4u : 0u, offB = tb.ok ? (unsigned)tb.gofs * 4u : 0u;
    float *la = &sV[0][ta.arr][ta.lofs];
    float *lb = &sV[0][tb.arr][tb.lofs < 0 ? 0 : tb.lofs];
    const bool hasB = tb.lofs >= 0;
    const int bufStride = 3 * LH * LW;
    const float c1 = d.c1;

    const bool zi = PML && valid && (i < P || i >= N1 - P);
    const bool zj = PML && valid && (j < P || j >= N2 - P);
    float ax = 0, bxc = 0, ay = 0, byc = 0, px = 0, py = 0, pz = 0;
    unsigned qx = 0, qy = 0;
    const unsigned dqx = (unsigned)(N2 * 2 * P), dqy = (unsigned)(2 * P * N1);
    if (zi) { ax = d.axI[i]; bxc = d.bxI[i]; qx = (unsigned)((kbeg * N2 + j) * (2 * P) + (i < P ? i : i - (N1 - 2 * P))); px = F4(d.psi[0], (unsigned)(qx) * 4u); }
    if (zj) { ay = d.ayI[j]; byc = d.byI[j]; qy = (unsigned)((kbeg * (2 * P) + (j < P ? j : j - (N2 - 2 * P))) * N1 + i); py = F4(d.psi[1], (unsigned)(qy) * 4u); }
    if (PML) {
        const int kg = d.k0 + kbeg;
        if (valid && (kg < P || kg >= d.N3 - P)) pz = F4(d.psi[2], (unsigned)((unsigned)((kg < P ? kg : kg - (d.N3 - 2 * P)) * d.plane) + cij) * 4u);
    }

    // z queues: Vx, Vy planes k-1 .. k+2, Vz planes k-2 .. k+1
    float vxm1 = 0, vx0 = 0, vxp1 = 0, vxp2 = 0, vym1 = 0, vy0 = 0, vyp1 = 0, vyp2 = 0, vzm2 = 0, vzm1 = 0, vz0 = 0, vzp1 = 0;
    float sxx = 0, syy = 0, szz = 0, rxx = 0, ryy = 0, rzz = 0;
    unsigned mraw = 0, cl = BFD_CLS_FLUID | BFD_CLS_NOMEM, cl1 = BFD_CLS_FLUID | BFD_CLS_NOMEM;
    if (valid) {
        const float *bVx = d.Vx + kbeg * pl, *bVy = d.Vy + kbeg * pl, *bVz = d.Vz + kbeg * pl;
        cl = U1((d.cls + kbeg * pl), cij); cl1 = U1((d.cls + kbeg * pl + pl), cij);
        vxm1 = F4((bVx - pl), cij * 4u); vx0 = F4(bVx, cij * 4u); vxp1 = F4((bVx + pl), cij * 4u); vxp2 = F4((bVx + 2 * pl), cij * 4u);
        vym1 = F4((bVy - pl), cij * 4u); vy0 = F4(bVy, cij * 4u); vyp1 = F4((bVy + pl), cij * 4u); vyp2 = F4((bVy + 2 * pl), cij * 4u);
        vzm2 = F4((bVz - 2 * pl), cij * 4u); vzm1 = F4((bVz - pl), cij * 4u); vz0 = F4(bVz, cij * 4u); vzp1 = F4((bVz + pl), cij * 4u);
        mraw = U2((d.mat + kbeg * pl), cij * 2u);
        szz = F4((d.Szz + kbeg * pl), cij * 4u);
        const bool fl = cl & BFD_CLS_FLUID, mem = !(cl & BFD_CLS_NOMEM) || !fl;
        if (mem) rzz = F4((d.Rzz + kbeg * pl), cij * 4u);
        if (!fl) {
            sxx = F4((d.Sxx + kbeg * pl), cij * 4u); syy = F4((d.Syy + kbeg * pl), cij * 4u);
            rxx = F4((d.Rxx + kbeg * pl), cij * 4u); ryy = F4((d.Ryy + kbeg * pl), cij * 4u);
        }
    }
    float ha = ta.ok ? F4(baseA + kbeg * pl, offA) : 0.0f;
    float hb = tb.ok ? F4(baseB + kbeg * pl, offB) : 0.0f;

    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        const long ko = (long)__builtin_amdgcn_readfirstlane(kl) * pl;
        const int k = d.k0 + kl;
        const int bo = b * bufStride;
        float nvx = 0, nvy = 0, nvz = 0, nha = 0, nhb = 0, nsxx = 0, nsyy = 0, nszz = 0, nrxx = 0, nryy = 0, nrzz = 0, npx = 0, npy = 0, npz = 0;
        unsigned nmraw = 0, ncl2 = BFD_CLS_FLUID | BFD_CLS_NOMEM;
        // the shear entries of THIS plane (own cell only, where the edge is updated here): issued with the prefetch, used at the end
        float sxy = 0, rxy = 0, sxz = 0, rxz = 0, syz = 0, ryz = 0;
        const bool shearHere = valid && !(cl & BFD_CLS_MIXED);
        const bool eXY = shearHere && (cl & BFD_CLS_EXY), eXZ = shearHere && (cl & BFD_CLS_EXZ), eYZ = shearHere && (cl & BFD_CLS_EYZ);
        auto prefetch_next = [&]() {
            if (valid) ncl2 = U1((d.cls + ko + 2 * pl), cij);           // ghost planes make kl+2 addressable
            if (eXY) { sxy = F4((d.Sxy + ko), cij * 4u); rxy = F4((d.Rxy + ko), cij * 4u); }
            if (eXZ) { sxz = F4((d.Sxz + ko), cij * 4u); rxz = F4((d.Rxz + ko), cij * 4u); }
            if (eYZ) { syz = F4((d.Syz + ko), cij * 4u); ryz = F4((d.Ryz + ko), cij * 4u); }
            if (kl + 1 < kend) {
                if (valid) {
                    const bool nfl = cl1 & BFD_CLS_FLUID, nmem = !(cl1 & BFD_CLS_NOMEM) || !nfl;
                    nvx = F4((d.Vx + ko + 3 * pl), cij * 4u); nvy = F4((d.Vy + ko + 3 * pl), cij * 4u); nvz = F4((d.Vz + ko + 2 * pl), cij * 4u);
                    nmraw = U2((d.mat + ko + pl), cij * 2u);
                    nszz = F4((d.Szz + ko + pl), cij * 4u);
                    if (nmem) nrzz = F4((d.Rzz + ko + pl), cij * 4u);
                    if (!nfl) {
                        nsxx = F4((d.Sxx + ko + pl), cij * 4u); nsyy = F4((d.Syy + ko + pl), cij * 4u);
                        nrxx = F4((d.Rxx + ko + pl), cij * 4u); nryy = F4((d.Ryy + ko + pl), cij * 4u);
                    }
                }
                if (ta.ok) nha = F4(baseA + ko + pl, offA);
                if (tb.ok) nhb = F4(baseB + ko + pl, offB);
                if (zi) npx = F4(d.psi[0], (unsigned)(qx + dqx) * 4u);
                if (zj) npy = F4(d.psi[1], (unsigned)(qy + dqy) * 4u);
                const int kn = k + 1;
                if (PML && valid && (kn < P || kn >= d.N3 - P)) npz = F4(d.psi[2], (unsigned)((unsigned)((kn < P ? kn : kn - (d.N3 - 2 * P)) * d.plane) + cij) * 4u);
            }
        };
        sV[0][0][bo + own] = vx0; sV[0][1][bo + own] = vy0; sV[0][2][bo + own] = vz0;
        la[bo] = ha;
        if (hasB) lb[bo] = hb;
        const int m = mraw & BFD_MAT_MASK;
        const bool fl = cl & BFD_CLS_FLUID, mem = !(cl & BFD_CLS_NOMEM) || !fl;
        float AP = 0, BP = 0, AS2 = 0, BS2 = 0, As = 0, Bs = 0;
        if (valid) { AP = d.AP[m]; if (mem) BP = d.BP[m]; if (!fl) { AS2 = d.AS2[m]; BS2 = d.BS2[m]; } if (eXY || eXZ || eYZ) { As = shearTab[8 * m]; Bs = shearTab[8 * m + 1]; } }
        __syncthreads();

        prefetch_next();
        if (valid) {
            const float *sx = &sV[0][0][bo + own], *sy = &sV[0][1][bo + own], *sz = &sV[0][2][bo + own];
            float dxVx = dminus4(sx[-2], sx[-1], vx0, sx[1]);
            float dyVy = dminus4(sy[-2 * LW], sy[-LW], vy0, sy[LW]);
            float dzVz = dminus4(vzm2, vzm1, vz0, vzp1);
            if (cl & BFD_CLS_REFL) {
                F4((d.Sxx + ko), cij * 4u) = 0.f; F4((d.Syy + ko), cij * 4u) = 0.f; F4((d.SzzW + ko), cij * 4u) = 0.f;
                F4((d.Rxx + ko), cij * 4u) = 0.f; F4((d.Ryy + ko), cij * 4u) = 0.f; F4((d.RzzW + ko), cij * 4u) = 0.f;
                F4((d.Sxy + ko), cij * 4u) = 0.f; F4((d.Sxz + ko), cij * 4u) = 0.f; F4((d.Syz + ko), cij * 4u) = 0.f;
                F4((d.Rxy + ko), cij * 4u) = 0.f; F4((d.Rxz + ko), cij * 4u) = 0.f; F4((d.Ryz + ko), cij * 4u) = 0.f;
            } else {
                if (zi) { const float pn = bxc * px + ax * dxVx; F4(d.psi[0], (unsigned)(qx) * 4u) = pn; dxVx = dxVx + pn; }
                if (zj) { const float pn = byc * py + ay * dyVy; F4(d.psi[1], (unsigned)(qy) * 4u) = pn; dyVy = dyVy + pn; }
                if (PML && (k < P || k >= d.N3 - P)) {
                    const float pn = d.bzI[k] * pz + d.azI[k] * dzVz;
                    F4(d.psi[2], (unsigned)((unsigned)((k < P ? k : k - (d.N3 - 2 * P)) * d.plane) + cij) * 4u) = pn;
                    dzVz = dzVz + pn;
                }
                const float sXY = dxVx + dyVy;
                const float div = sXY + dzVz;
                if (fl) {               // fluid cell: one copy of the identical normal stresses
                    float val;
                    if (!mem) val = szz + AP * div;
                    else {
                        const float rn = c1 * rzz - BP * div;
                        val = szz + (AP * div + 0.5f * (rzz + rn));
                        ST4((d.RzzW + ko), cij * 4u, rn);
                    }
                    ST4((d.SzzW + ko), cij * 4u, val);
                } else {
                    const float sYZ = dyVy + dzVz, sXZ = dxVx + dzVz;
                    float rn;
                    rn = c1 * rxx - (BP * div - BS2 * sYZ);
                    ST4((d.Sxx + ko), cij * 4u, sxx + ((AP * div - AS2 * sYZ) + 0.5f * (rxx + rn))); ST4((d.Rxx + ko), cij * 4u, rn);
                    rn = c1 * ryy - (BP * div - BS2 * sXZ);
                    ST4((d.Syy + ko), cij * 4u, syy + ((AP * div - AS2 * sXZ) + 0.5f * (ryy + rn))); ST4((d.Ryy + ko), cij * 4u, rn);
                    rn = c1 * rzz - (BP * div - BS2 * sXY);
                    ST4((d.SzzW + ko), cij * 4u, szz + ((AP * div - AS2 * sXY) + 0.5f * (rzz + rn))); ST4((d.RzzW + ko), cij * 4u, rn);
                }
                // shear entries of this cell (same expressions as stress_v2 / stress_shear_sparse)
                if (eXY) {
                    float dyVx = dplus4(sx[-LW], vx0, sx[LW], sx[2 * LW]);
                    float dxVy = dplus4(sy[-1], vy0, sy[1], sy[2]);
                    if (zi) dxVy = cpml(d.psi[4], (unsigned)((kl * N2 + j) * (2 * P) + (i < P ? i : i - (N1 - 2 * P))), d.axH[i], d.bxH[i], dxVy);
                    if (zj) dyVx = cpml(d.psi[3], (unsigned)((kl * (2 * P) + (j < P ? j : j - (N2 - 2 * P))) * N1 + i), d.ayH[j], d.byH[j], dyVx);
                    const float e = dyVx + dxVy;
                    const float rn = c1 * rxy - Bs * e;
                    F4((d.Sxy + ko), cij * 4u) = sxy + (As * e + 0.5f * (rxy + rn)); F4((d.Rxy + ko), cij * 4u) = rn;
                }
                if (eXZ) {
                    float dzVx = dplus4(vxm1, vx0, vxp1, vxp2);
                    float dxVz = dplus4(sz[-1], vz0, sz[1], sz[2]);
                    if (zi) dxVz = cpml(d.psi[6], (unsigned)((kl * N2 + j) * (2 * P) + (i < P ? i : i - (N1 - 2 * P))), d.axH[i], d.bxH[i], dxVz);
                    if (PML && (k < P || k >= d.N3 - P)) dzVx = cpml(d.psi[5], (unsigned)((k < P ? k : k - (d.N3 - 2 * P)) * d.plane) + cij, d.azH[k], d.bzH[k], dzVx);
                    const float e = dzVx + dxVz;
                    const float rn = c1 * rxz - Bs * e;
                    F4((d.Sxz + ko), cij * 4u) = sxz + (As * e + 0.5f * (rxz + rn)); F4((d.Rxz + ko), cij * 4u) = rn;
                }
                if (eYZ) {
                    float dzVy = dplus4(vym1, vy0, vyp1, vyp2);
                    float dyVz = dplus4(sz[-LW], vz0, sz[LW], sz[2 * LW]);
                    if (zj) dyVz = cpml(d.psi[8], (unsigned)((kl * (2 * P) + (j < P ? j : j - (N2 - 2 * P))) * N1 + i), d.ayH[j], d.byH[j], dyVz);
                    if (PML && (k < P || k >= d.N3 - P)) dzVy = cpml(d.psi[7], (unsigned)((k < P ? k : k - (d.N3 - 2 * P)) * d.plane) + cij, d.azH[k], d.bzH[k], dzVy);
                    const float e = dzVy + dyVz;
                    const float rn = c1 * ryz - Bs * e;
                    F4((d.Syz + ko), cij * 4u) = syz + (As * e + 0.5f * (ryz + rn)); F4((d.Ryz + ko), cij * 4u) = rn;
                }
            }
        }
        vxm1 = vx0; vx0 = vxp1; vxp1 = vxp2; vxp2 = nvx;
        vym1 = vy0; vy0 = vyp1; vyp1 = vyp2; vyp2 = nvy;
        vzm2 = vzm1; vzm1 = vz0; vz0 = vzp1; vzp1 = nvz;
        ha = nha; hb = nhb; mraw = nmraw; cl = cl1; cl1 = ncl2;
        sxx = nsxx; syy = nsyy; szz = nszz; rxx = nrxx; ryy = nryy; rzz = nrzz;
        px = npx; py = npy; pz = npz; qx += dqx; qy += dqy;
    }
}

#ifndef SOLID_MERGED_WAVES_PER_SIMD
#define SOLID_MERGED_WAVES_PER_SIMD 4
#endif
__global__ __launch_bounds__(NTHREADS, SOLID_MERGED_WAVES_PER_SIMD) void stress_solid_merged(bfd_dev d, int tilesX, int nblocks, const int *__restrict__ xmap, const int4 *__restrict__ runs,
                                                                                             const float *__restrict__ shearTab)
{
    __shared__ float sV[2][3][LH * LW];
    const int ri = run_index(nblocks, xmap);
    if (ri < 0) return;
    const int4 run = runs[ri];
    if (run.z & 8) stress_solid_merged_body<true>(d, run, tilesX, shearTab, sV);
    else stress_solid_merged_body<false>(d, run, tilesX, shearTab, sV);
}

// one halo value of the solid velocity kernel: SUBST (Sxx / Syy halos): Szz where the halo cell is fluid; otherwise a
// shear array, loaded only where its edge bit is set. base / alt are wave-uniform (SGPR) plane bases.
__device__ __forceinline__ float halo_value(const float *__restrict__ base, const float *__restrict__ alt, bool subst, unsigned bit,
                                            unsigned hc, unsigned off)
{
    if (subst) return (hc & BFD_CLS_FLUID) ? F4(alt, off * 4u) : F4(base, off * 4u);
    return (hc & bit) ? F4(base, off * 4u) : 0.0f;
}

// LDS set: 0 Sxx (x halo), 1 Syy (y halo), 2 Sxy (x and y halo), 3 Sxz (x halo), 4 Syz (y halo)
// PML: the run touches an absorbing-layer zone (otherwise no CPML code, pointers or registers)
template <bool ACC, bool PML>
__device__ __forceinline__ void velocity_solid_body(const bfd_dev &d, const int4 &run, int tilesX, float (*sS)[5][LH * LW],
                                                    float *__restrict__ accP, float *__restrict__ pkP)
{
    const int N1 = d.N1, N2 = d.N2;
    const int bx = run.x % tilesX, by = run.x / tilesX;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int wv = __builtin_amdgcn_readfirstlane(ty);      // wave index = tile row (uniform)
    const int i0 = bx * TX, j0 = by * TY;
    const int i = i0 + tx, j = j0 + ty;
    const bool valid = (i < N1) && (j < N2);
    const long pl = d.plane;
    const int kbeg = run.y & 0xFFFF, kend = run.y >> 16;
    const int P = d.P;
    const int own = (ty + 2) * LW + tx + 2;
    const unsigned cij = valid ? (unsigned)(j * N1 + i) : 0u;
    // ids of the cells (i+1, j) and (i, j+1), clamped at the domain edge: byte offsets derived from cij when needed (the two
    // comparisons live in scalar masks, not in registers)
    const bool hasX = valid && i + 1 < N1, hasY = valid && j + 1 < N2;
    const unsigned rowBytes = (unsigned)N1 * 2u;
    const bool accA = ACC && accP != nullptr, accK = ACC && pkP != nullptr;

    // halo tasks, the array of a task is uniform per wave (its plane base stays in SGPRs):
    //   A: waves 0-3 the 4 halo rows of Syy, waves 4-7 those of Sxy
    //   B: waves 0-3 the 4 halo rows of Syz; lanes 0..31 of wave 4 / 5 / 6 the halo columns of Sxx / Sxy / Sxz
    HaloTask ta, tb;
    const int arrA = wv < 4 ? 1 : 2;
    ytask(tid & 255, arrA, i0, j0, N1, N2, ta);
    const int arrB = wv < 4 ? 4 : (wv == 4 ? 0 : (wv == 5 ? 2 : 3));
    if (wv < 4) ytask(tid, 4, i0, j0, N1, N2, tb);
    else if (wv < 7 && tx < XT) xtask(tx, arrB, i0, j0, N1, N2, tb);
    else { tb.lofs = -1; tb.ok = false; tb.arr = arrB; tb.gofs = 0; }
    const float *baseA = arrA == 1 ? d.Syy : d.Sxy;
    const float *baseB = arrB == 4 ? d.Syz : (arrB == 0 ? d.Sxx : (arrB == 2 ? d.Sxy : d.Sxz));
    const bool substA = arrA == 1, substB = arrB == 0;
    const unsigned bitA = BFD_CLS_EXY, bitB = arrB == 4 ? BFD_CLS_EYZ : (arrB == 2 ? BFD_CLS_EXY : BFD_CLS_EXZ);
    const unsigned offA = ta.ok ? (unsigned)ta.gofs : 0u, offB = tb.ok ? (unsigned)tb.gofs : 0u;
    float *la = &sS[0][ta.arr][ta.lofs];
    float *lb = &sS[0][tb.arr][tb.lofs < 0 ? 0 : tb.lofs];
    const bool hasB = tb.lofs >= 0;
    const int bufStride = 5 * LH * LW;

    const bool zi = PML && valid && (i < P || i >= N1 - P);
    const bool zj = PML && valid && (j < P || j >= N2 - P);
    const bool inner = valid && i >= d.ND && i < N1 - d.ND && j >= d.ND && j < N2 - d.ND;

    // z queues: Szz k-1..k+2 ; Sxz, Syz k-2..k+1 ; in-plane arrays one plane ahead. Class bytes of the own column:
    // cB = plane kl+1, cC = plane kl+2 (they steer the loads of the coming iterations)
    float zzm1 = 0, zz0 = 0, zzp1 = 0, zzp2 = 0, xzm2 = 0, xzm1 = 0, xz0 = 0, xzp1 = 0, yzm2 = 0, yzm1 = 0, yz0 = 0, yzp1 = 0;
    unsigned cB = BFD_CLS_FLUID, cC = BFD_CLS_FLUID;
    {
    float sxx = 0, syy = 0, sxy = 0;
    if (valid) {
        const float *bzz = d.Szz + kbeg * pl, *bxz = d.Sxz + kbeg * pl, *byz = d.Syz + kbeg * pl;
        const uint8_t *bc = d.cls + kbeg * pl;
        const unsigned cm2 = U1((bc - 2 * pl), cij), cm1 = U1((bc - pl), cij), c0 = U1(bc, cij);
        cB = U1((bc + pl), cij); cC = U1((bc + 2 * pl), cij);
        zzm1 = F4((bzz - pl), cij * 4u); zz0 = F4(bzz, cij * 4u); zzp1 = F4((bzz + pl), cij * 4u); zzp2 = F4((bzz + 2 * pl), cij * 4u);
        if (cm2 & BFD_CLS_EXZ) xzm2 = F4((bxz - 2 * pl), cij * 4u);
        if (cm1 & BFD_CLS_EXZ) xzm1 = F4((bxz - pl), cij * 4u);
        if (c0 & BFD_CLS_EXZ) xz0 = F4(bxz, cij * 4u);
        if (cB & BFD_CLS_EXZ) xzp1 = F4((bxz + pl), cij * 4u);
        if (cm2 & BFD_CLS_EYZ) yzm2 = F4((byz - 2 * pl), cij * 4u);
        if (cm1 & BFD_CLS_EYZ) yzm1 = F4((byz - pl), cij * 4u);
        if (c0 & BFD_CLS_EYZ) yz0 = F4(byz, cij * 4u);
        if (cB & BFD_CLS_EYZ) yzp1 = F4((byz + pl), cij * 4u);
        if (c0 & BFD_CLS_FLUID) { sxx = zz0; syy = zz0; }
        else { sxx = F4((d.Sxx + kbeg * pl), cij * 4u); syy = F4((d.Syy + kbeg * pl), cij * 4u); }
        if (c0 & BFD_CLS_EXY) sxy = F4((d.Sxy + kbeg * pl), cij * 4u);
    }
    // the values of a plane are staged in LDS at the end of the iteration before it (here: plane kbeg into buffer kbeg & 1)
    const int bo0 = (kbeg & 1) * bufStride;
    sS[0][0][bo0 + own] = sxx; sS[0][1][bo0 + own] = syy; sS[0][2][bo0 + own] = sxy; sS[0][3][bo0 + own] = xz0; sS[0][4][bo0 + own] = yz0;
    }
    // V and the accumulators are loaded in the iteration that uses them (their latency hides behind the barrier and the
    // other waves; prefetching them one plane ahead costs 5 registers this kernel does not have)
    float r0 = 0;
    unsigned mraw = 0, mraw1 = 0, mx = 0, my = 0;
    if (valid) {
        const uint16_t *bM = d.mat + kbeg * pl;
        mraw = U2(bM, cij * 2u); mraw1 = U2((bM + pl), cij * 2u); mx = U2(bM, cij * 2u + (hasX ? 2u : 0u)); my = U2(bM, cij * 2u + (hasY ? rowBytes : 0u));
        r0 = d.invRho[mraw & BFD_MAT_MASK];
    }
    // halo values of plane kbeg and the class bytes of the halo cells one plane ahead
    unsigned hcA = 0, hcB = 0;
    {
        float ha = 0.f, hb = 0.f;
        if (ta.ok) { ha = halo_value(baseA + kbeg * pl, d.Szz + kbeg * pl, substA, bitA, U1((d.cls + kbeg * pl), offA), offA); hcA = U1((d.cls + kbeg * pl + pl), offA); }
        if (tb.ok) { hb = halo_value(baseB + kbeg * pl, d.Szz + kbeg * pl, substB, bitB, U1((d.cls + kbeg * pl), offB), offB); hcB = U1((d.cls + kbeg * pl + pl), offB); }
        la[(kbeg & 1) * bufStride] = ha;
        if (hasB) lb[(kbeg & 1) * bufStride] = hb;
    }

    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        // opaque to loop strength reduction: otherwise every array gets a 64-bit per-lane pointer that is bumped each plane
        // (a VGPR pair per array); this way the plane bases are recomputed on the scalar unit and stay in SGPRs
        const long ko = (long)__builtin_amdgcn_readfirstlane(kl) * pl;
        const int k = d.k0 + kl;
        const int bn = (b ^ 1) * bufStride;     // buffer of plane kl+1 (free: every thread is past the barrier of iteration kl-1's reads)
        float nzz = 0, nxz = 0, nyz = 0, nxx = 0, nyy = 0, nxy = 0, nha = 0, nhb = 0;
        unsigned nm2 = 0, nmx = 0, nmy = 0, nc3 = BFD_CLS_FLUID, nhcA = 0, nhcB = 0;
        auto prefetch_next = [&]() {
        if (valid) nm2 = U2((d.mat + ko + 2 * pl), cij * 2u);              // ghost planes make kl+2 addressable
        if (valid && kl + 2 < kend) nc3 = U1((d.cls + ko + 3 * pl), cij);       // steers the Sxz / Syz loads of iteration kl+1 (plane kl+3 <= nk+1)
            if (kl + 1 < kend) {
                if (valid) {
                    nzz = F4((d.Szz + ko + 3 * pl), cij * 4u);
                    if (cC & BFD_CLS_EXZ) nxz = F4((d.Sxz + ko + 2 * pl), cij * 4u);
                    if (cC & BFD_CLS_EYZ) nyz = F4((d.Syz + ko + 2 * pl), cij * 4u);
                    if (cB & BFD_CLS_FLUID) { nxx = zzp1; nyy = zzp1; }
                    else { nxx = F4((d.Sxx + ko + pl), cij * 4u); nyy = F4((d.Syy + ko + pl), cij * 4u); }
                    if (cB & BFD_CLS_EXY) nxy = F4((d.Sxy + ko + pl), cij * 4u);
                    nmx = U2((d.mat + ko + pl), cij * 2u + (hasX ? 2u : 0u)); nmy = U2((d.mat + ko + pl), cij * 2u + (hasY ? rowBytes : 0u));
                }
                if (ta.ok) { nha = halo_value(baseA + ko + pl, d.Szz + ko + pl, substA, bitA, hcA, offA); nhcA = U1((d.cls + ko + 2 * pl), offA); }
                if (tb.ok) { nhb = halo_value(baseB + ko + pl, d.Szz + ko + pl, substB, bitB, hcB, offB); nhcB = U1((d.cls + ko + 2 * pl), offB); }
            }
        };
        float r1 = 0, rx = 0, ry = 0, vx = 0, vy = 0, vz = 0, av = 0, pv = 0;
        if (valid) {
            r1 = d.invRho[mraw1 & BFD_MAT_MASK];       // plane kl+1, becomes r0 of the next iteration
            rx = d.invRho[mx & BFD_MAT_MASK];
            ry = d.invRho[my & BFD_MAT_MASK];
            vx = LD4((d.Vx + ko), cij * 4u); vy = LD4((d.Vy + ko), cij * 4u); vz = LD4((d.Vz + ko), cij * 4u);
            if (accA) av = LD4((accP + ko), cij * 4u);
            if (accK) pv = F4((pkP + ko), cij * 4u);
        }
        __syncthreads();

        float *wVx = d.VxW + ko, *wVy = d.VyW + ko, *wVz = d.VzW + ko;
        prefetch_next();        // after the barrier (issuing these loads before it was measured slower: 2.37 -> 2.52 ms per step at C2-medium 512^3)
        if (valid) {
            const float sxx = sS[b][0][own], syy = sS[b][1][own], sxy = sS[b][2][own];
            if (ACC) {
                if (inner && k >= d.ND && k < d.N3 - d.ND) {
                    const float s = (sxx + syy) + zz0;
                    const float p = -s * (1.0f / 3.0f);
                    if (accA) F4((accP + ko), cij * 4u) = av + p * p;
                    if (accK) { const float ap = fabsf(p); if (ap > pv) F4((pkP + ko), cij * 4u) = ap; }
                }
            }
            if (mraw & BFD_REFLECTOR_BIT) {
                F4(wVx, cij * 4u) = 0.f; F4(wVy, cij * 4u) = 0.f; F4(wVz, cij * 4u) = 0.f;
            } else {
                const float *pxx = &sS[b][0][own], *pyy = &sS[b][1][own], *pxy = &sS[b][2][own];
                const float *pxz = &sS[b][3][own], *pyz = &sS[b][4][own];
                float dxSxx = dplus4(pxx[-1], sxx, pxx[1], pxx[2]);
                float dySxy = dminus4(pxy[-2 * LW], pxy[-LW], sxy, pxy[LW]);
                float dzSxz = dminus4(xzm2, xzm1, xz0, xzp1);
                float dxSxy = dminus4(pxy[-2], pxy[-1], sxy, pxy[1]);
                float dySyy = dplus4(pyy[-LW], syy, pyy[LW], pyy[2 * LW]);
                float dzSyz = dminus4(yzm2, yzm1, yz0, yzp1);
                float dxSxz = dminus4(pxz[-2], pxz[-1], xz0, pxz[1]);
                float dySyz = dminus4(pyz[-2 * LW], pyz[-LW], yz0, pyz[LW]);
                float dzSzz = dplus4(zzm1, zz0, zzp1, zzp2);
                if (zi) {
                    const int xi = i < P ? i : i - (N1 - 2 * P);
                    const unsigned q = (unsigned)((kl * N2 + j) * (2 * P) + xi);
                    dxSxx = cpml(d.psi[9], q, d.axH[i], d.bxH[i], dxSxx);
                    dxSxy = cpml(d.psi[12], q, d.axI[i], d.bxI[i], dxSxy);
                    dxSxz = cpml(d.psi[15], q, d.axI[i], d.bxI[i], dxSxz);
                }
                if (zj) {
                    const int yj = j < P ? j : j - (N2 - 2 * P);
                    const unsigned q = (unsigned)((kl * (2 * P) + yj) * N1 + i);
                    dySxy = cpml(d.psi[10], q, d.ayI[j], d.byI[j], dySxy);
                    dySyy = cpml(d.psi[13], q, d.ayH[j], d.byH[j], dySyy);
                    dySyz = cpml(d.psi[16], q, d.ayI[j], d.byI[j], dySyz);
                }
                if (PML && (k < P || k >= d.N3 - P)) {
                    const int zk = k < P ? k : k - (d.N3 - 2 * P);
                    const unsigned q = (unsigned)(zk * d.plane) + cij;
                    dzSxz = cpml(d.psi[11], q, d.azI[k], d.bzI[k], dzSxz);
                    dzSyz = cpml(d.psi[14], q, d.azI[k], d.bzI[k], dzSyz);
                    dzSzz = cpml(d.psi[17], q, d.azH[k], d.bzH[k], dzSzz);
                }
                const float bxv = 0.5f * (r0 + rx), byv = 0.5f * (r0 + ry), bzv = 0.5f * (r0 + r1);
                ST4(wVx, cij * 4u, vx + bxv * ((dxSxx + dySxy) + dzSxz));
                ST4(wVy, cij * 4u, vy + byv * ((dxSxy + dySyy) + dzSyz));
                ST4(wVz, cij * 4u, vz + bzv * ((dxSxz + dySyz) + dzSzz));
            }
        }
        zzm1 = zz0; zz0 = zzp1; zzp1 = zzp2; zzp2 = nzz;
        xzm2 = xzm1; xzm1 = xz0; xz0 = xzp1; xzp1 = nxz;
        yzm2 = yzm1; yzm1 = yz0; yz0 = yzp1; yzp1 = nyz;
        // stage plane kl+1 (the loads were issued before this plane's arithmetic)
        sS[0][0][bn + own] = nxx; sS[0][1][bn + own] = nyy; sS[0][2][bn + own] = nxy; sS[0][3][bn + own] = xz0; sS[0][4][bn + own] = yz0;
        la[bn] = nha;
        if (hasB) lb[bn] = nhb;
        hcA = nhcA; hcB = nhcB;
        r0 = r1; mraw = mraw1; mraw1 = nm2; mx = nmx; my = nmy;
        cB = cC; cC = nc3;
    }
}

// ------------------------------------------------------------------------------------------------
// The solid-run kernels again with GLOBAL loads and a prefetch WITHOUT branches (round 4). In the bodies above every predicated
// load of the next plane sits in its own basic block and the loads are FLAT: the first LDS wait of an iteration (lgkmcnt, which
// flat loads count on) drains the whole prefetch, so an iteration is one memory round trip followed by the arithmetic, nothing
// overlaps. Here a load whose class predicate is off reads the first dword of its plane instead (one line every lane shares:
// an L1 hit, no traffic) and its result is replaced by 0, so every iteration issues the same loads in the same order, the
// compiler counts them (vmcnt(n) waits) and the plane k+1 loads stay in flight across the arithmetic of plane k.
// Same values, same arithmetic: bit-identical.
// ------------------------------------------------------------------------------------------------
#define BFD_GA __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ BFD_GA T *gbase(const T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (BFD_GA T *)(((unsigned long long)hi << 32) | lo);
}
// (round 6: pinning the BASE as an opaque scalar pair instead of the lane's offset -- to save the v_mov each pinned offset costs -- makes the compiler form
// 64-bit vector addresses again: v_lshl_add_u64 6 -> 30 per plane, 92 -> 100 registers; not kept)
__device__ __forceinline__ unsigned gpin(unsigned v) { asm("" : "+v"(v)); return v; }
__device__ __forceinline__ float gl4(const float *b, unsigned ofs) { return *(BFD_GA const float *)((BFD_GA const char *)gbase(b) + gpin(ofs)); }
__device__ __forceinline__ float gl4nt(const float *b, unsigned ofs) { return __builtin_nontemporal_load((BFD_GA const float *)((BFD_GA const char *)gbase(b) + gpin(ofs))); }
__device__ __forceinline__ unsigned gl2(const uint16_t *b, unsigned ofs) { return *(BFD_GA const uint16_t *)((BFD_GA const char *)gbase(b) + gpin(ofs)); }
__device__ __forceinline__ unsigned gl1(const uint8_t *b, unsigned ofs) { return *(BFD_GA const uint8_t *)((BFD_GA const char *)gbase(b) + gpin(ofs)); }
// predicated: lanes with c == false read dword 0 of the plane and get 0
__device__ __forceinline__ float glp(const float *b, unsigned ofs, bool c) { const float v = gl4(b, c ? ofs : 0u); return c ? v : 0.0f; }
__device__ __forceinline__ void gs4(float *b, unsigned ofs, float v) { *(BFD_GA float *)((BFD_GA char *)gbase(b) + gpin(ofs)) = v; }
__device__ __forceinline__ void gs4nt(float *b, unsigned ofs, float v)
{
#ifndef BFD_NT_STORES_OFF
    __builtin_nontemporal_store(v, (BFD_GA float *)((BFD_GA char *)gbase(b) + gpin(ofs)));
#else
    gs4(b, ofs, v);
#endif
}
// one halo value: SUBST (Sxx / Syy halos): Szz where the halo cell is fluid, the array itself otherwise; else a shear array where its
// edge bit is set, 0 elsewhere. One load on every path: the plane base is chosen per lane, the offset falls back to dword 0.
__device__ __forceinline__ float halo_value_g(const float *base, const float *alt, bool subst, unsigned bit, unsigned hc, unsigned off4, bool ok)
{
    const bool fluid = (hc & BFD_CLS_FLUID) != 0;
    const bool take = ok && (subst || (hc & bit));
    const unsigned long long pb = (unsigned long long)gbase(base), pa = (unsigned long long)gbase(alt);
    const unsigned long long pp = (subst && fluid) ? pa : pb;
    const float v = *(BFD_GA const float *)(pp + (take ? off4 : 0u));
    return take ? v : 0.0f;
}

// compact form: the halo cell's value comes from Szz (SUBST and fluid), from the compact array at byte offset e4 (SUBST: listed cell, else: edge bit
// set), or is 0
__device__ __forceinline__ float halo_value_c(const float *cbase, const float *alt, bool subst, unsigned bit, unsigned hc, unsigned off4, unsigned e4, bool ok)
{
    const bool fromAlt = subst && (hc & BFD_CLS_FLUID) != 0;
    const bool take = ok && (fromAlt || (subst ? css_listed(hc) : (hc & bit) != 0));
    const unsigned long long pp = fromAlt ? (unsigned long long)gbase(alt) : (unsigned long long)gbase(cbase);
    const float v = *(BFD_GA const float *)(pp + (take ? (fromAlt ? off4 : e4) : 0u));
    return take ? v : 0.0f;
}

template <bool PML>
__device__ __forceinline__ void stress_solid_body_g(const bfd_dev &d, const int4 &run, int tilesX, float (*sV)[2][LH * LW])
{
    const int N1 = d.N1, N2 = d.N2;
    const int bx = run.x % tilesX, by = run.x / tilesX, kbeg = run.y & 0xFFFF, kend = run.y >> 16;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int i0 = bx * TX, j0 = by * TY;
    const int i = i0 + tx, j = j0 + ty;
    const bool valid = (i < N1) && (j < N2);
    const long pl = d.plane;
    const int P = d.P;
    const int own = (ty + 2) * LW + tx + 2;
    const unsigned cij = valid ? (unsigned)(j * N1 + i) : 0u;
    const unsigned c4 = cij * 4u;

    HaloTask t; t.lofs = -1; t.ok = false; t.arr = 0; t.gofs = 0;
    if (tid < YT) ytask(tid, 1, i0, j0, N1, N2, t);
    else if (tid < YT + XT) xtask(tid - YT, 0, i0, j0, N1, N2, t);
    const bool has = t.lofs >= 0;
    const float *ph = __builtin_amdgcn_readfirstlane(t.arr) == 0 ? d.Vx : d.Vy;
    const unsigned hofs = t.ok ? (unsigned)t.gofs * 4u : 0u;
    float *lh = &sV[0][t.arr][has ? t.lofs : 0];
    const float c1 = d.c1;

    const bool zi = PML && valid && (i < P || i >= N1 - P);
    const bool zj = PML && valid && (j < P || j >= N2 - P);
    float ax = 0, bxc = 0, ay = 0, byc = 0, px = 0, py = 0, pz = 0;
    unsigned qx = 0, qy = 0;
    const unsigned dqx = (unsigned)(N2 * 2 * P), dqy = (unsigned)(2 * P * N1);
    if (zi) { ax = d.axI[i]; bxc = d.bxI[i]; qx = (unsigned)((kbeg * N2 + j) * (2 * P) + (i < P ? i : i - (N1 - 2 * P))); px = F4(d.psi[0], (unsigned)(qx) * 4u); }
    if (zj) { ay = d.ayI[j]; byc = d.byI[j]; qy = (unsigned)((kbeg * (2 * P) + (j < P ? j : j - (N2 - 2 * P))) * N1 + i); py = F4(d.psi[1], (unsigned)(qy) * 4u); }
    if (PML) {
        const int kg = d.k0 + kbeg;
        if (valid && (kg < P || kg >= d.N3 - P)) pz = F4(d.psi[2], (unsigned)((unsigned)((kg < P ? kg : kg - (d.N3 - 2 * P)) * d.plane) + cij) * 4u);
    }

    float vx0 = 0, vy0 = 0, vzm2 = 0, vzm1 = 0, vz0 = 0, vzp1 = 0;
    float sxx = 0, syy = 0, szz = 0, rxx = 0, ryy = 0, rzz = 0;
    unsigned mraw = 0, cl = BFD_CLS_FLUID | BFD_CLS_NOMEM, cl1 = BFD_CLS_FLUID | BFD_CLS_NOMEM;
    if (valid) {
        const float *bVz = d.Vz + kbeg * pl;
        cl = gl1(d.cls + kbeg * pl, cij); cl1 = gl1(d.cls + kbeg * pl + pl, cij);
        vx0 = gl4(d.Vx + kbeg * pl, c4); vy0 = gl4(d.Vy + kbeg * pl, c4);
        vzm2 = gl4(bVz - 2 * pl, c4); vzm1 = gl4(bVz - pl, c4); vz0 = gl4(bVz, c4); vzp1 = gl4(bVz + pl, c4);
        mraw = gl2(d.mat + kbeg * pl, cij * 2u);
        szz = gl4(d.Szz + kbeg * pl, c4);
        const bool fl = cl & BFD_CLS_FLUID, mem = !(cl & BFD_CLS_NOMEM) || !fl;
        rzz = glp(d.Rzz + kbeg * pl, c4, mem);
        sxx = glp(d.Sxx + kbeg * pl, c4, !fl); syy = glp(d.Syy + kbeg * pl, c4, !fl);
        rxx = glp(d.Rxx + kbeg * pl, c4, !fl); ryy = glp(d.Ryy + kbeg * pl, c4, !fl);
    }
    float hv = glp(ph + kbeg * pl, hofs, t.ok);

    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        const long ko = (long)__builtin_amdgcn_readfirstlane(kl) * pl;
        const int k = d.k0 + kl;
        sV[b][0][own] = vx0; sV[b][1][own] = vy0;
        if (has) lh[b * (2 * LH * LW)] = hv;
        const int m = mraw & BFD_MAT_MASK;
        const bool fl = cl & BFD_CLS_FLUID, mem = !(cl & BFD_CLS_NOMEM) || !fl;
        float AP = 0, BP = 0, AS2 = 0, BS2 = 0;
        if (valid) { AP = d.AP[m]; if (mem) BP = d.BP[m]; if (!fl) { AS2 = d.AS2[m]; BS2 = d.BS2[m]; } }
        __syncthreads();

        // everything plane kl+1 needs: the same loads in every iteration (a predicate that is off reads dword 0 of the plane); the raw
        // values are not touched before the end of the iteration
        const bool more = kl + 1 < kend;
        const long kn = more ? pl : 0;
        const bool nfl = cl1 & BFD_CLS_FLUID, nmem = !(cl1 & BFD_CLS_NOMEM) || !nfl;
        const bool pM = more && valid && nmem, pS = more && valid && !nfl, pH = more && t.ok;
        const unsigned ncl2R = gl1(d.cls + ko + 2 * pl, cij);           // ghost planes make kl+2 addressable
        const float nvxR = gl4(d.Vx + ko + kn, c4), nvyR = gl4(d.Vy + ko + kn, c4), nvzR = gl4(d.Vz + ko + pl + kn, c4);
        const unsigned nmrawR = gl2(d.mat + ko + kn, cij * 2u);
        const float nszzR = gl4(d.Szz + ko + kn, c4);
        const float nrzzR = gl4(d.Rzz + ko + kn, pM ? c4 : 0u);
        const float nsxxR = gl4(d.Sxx + ko + kn, pS ? c4 : 0u), nsyyR = gl4(d.Syy + ko + kn, pS ? c4 : 0u);
        const float nrxxR = gl4(d.Rxx + ko + kn, pS ? c4 : 0u), nryyR = gl4(d.Ryy + ko + kn, pS ? c4 : 0u);
        const float nhR = gl4(ph + ko + kn, pH ? hofs : 0u);
        float npx = 0, npy = 0, npz = 0;
        if (PML) {
            if (more && zi) npx = F4(d.psi[0], (unsigned)(qx + dqx) * 4u);
            if (more && zj) npy = F4(d.psi[1], (unsigned)(qy + dqy) * 4u);
            const int kq = k + 1;
            if (more && valid && (kq < P || kq >= d.N3 - P)) npz = F4(d.psi[2], (unsigned)((unsigned)((kq < P ? kq : kq - (d.N3 - 2 * P)) * d.plane) + cij) * 4u);
        }

        if (valid) {
            const float *sx = &sV[b][0][own], *sy = &sV[b][1][own];
            float dxVx = dminus4(sx[-2], sx[-1], vx0, sx[1]);
            float dyVy = dminus4(sy[-2 * LW], sy[-LW], vy0, sy[LW]);
            float dzVz = dminus4(vzm2, vzm1, vz0, vzp1);
            if (cl & BFD_CLS_REFL) {
                gs4(d.Sxx + ko, c4, 0.f); gs4(d.Syy + ko, c4, 0.f); gs4(d.SzzW + ko, c4, 0.f);
                gs4(d.Rxx + ko, c4, 0.f); gs4(d.Ryy + ko, c4, 0.f); gs4(d.RzzW + ko, c4, 0.f);
                gs4(d.Sxy + ko, c4, 0.f); gs4(d.Sxz + ko, c4, 0.f); gs4(d.Syz + ko, c4, 0.f);
                gs4(d.Rxy + ko, c4, 0.f); gs4(d.Rxz + ko, c4, 0.f); gs4(d.Ryz + ko, c4, 0.f);
            } else {
                if (zi) { const float pn = bxc * px + ax * dxVx; F4(d.psi[0], (unsigned)(qx) * 4u) = pn; dxVx = dxVx + pn; }
                if (zj) { const float pn = byc * py + ay * dyVy; F4(d.psi[1], (unsigned)(qy) * 4u) = pn; dyVy = dyVy + pn; }
                if (PML && (k < P || k >= d.N3 - P)) {
                    const float pn = d.bzI[k] * pz + d.azI[k] * dzVz;
                    F4(d.psi[2], (unsigned)((unsigned)((k < P ? k : k - (d.N3 - 2 * P)) * d.plane) + cij) * 4u) = pn;
                    dzVz = dzVz + pn;
                }
                const float sXY = dxVx + dyVy;
                const float div = sXY + dzVz;
                if (fl) {               // fluid cell: one copy of the identical normal stresses
                    float val;
                    if (!mem) val = szz + AP * div;
                    else {
                        const float rn = c1 * rzz - BP * div;
                        val = szz + (AP * div + 0.5f * (rzz + rn));
                        gs4nt(d.RzzW + ko, c4, rn);
                    }
                    gs4nt(d.SzzW + ko, c4, val);
                } else {
                    const float sYZ = dyVy + dzVz, sXZ = dxVx + dzVz;
                    float rn;
                    rn = c1 * rxx - (BP * div - BS2 * sYZ);
                    gs4nt(d.Sxx + ko, c4, sxx + ((AP * div - AS2 * sYZ) + 0.5f * (rxx + rn))); gs4nt(d.Rxx + ko, c4, rn);
                    rn = c1 * ryy - (BP * div - BS2 * sXZ);
                    gs4nt(d.Syy + ko, c4, syy + ((AP * div - AS2 * sXZ) + 0.5f * (ryy + rn))); gs4nt(d.Ryy + ko, c4, rn);
                    rn = c1 * rzz - (BP * div - BS2 * sXY);
                    gs4nt(d.SzzW + ko, c4, szz + ((AP * div - AS2 * sXY) + 0.5f * (rzz + rn))); gs4nt(d.RzzW + ko, c4, rn);
                }
            }
        }
        // masks of the prefetched values, queues
        const bool mv = more && valid;
        vx0 = mv ? nvxR : 0.0f; vy0 = mv ? nvyR : 0.0f;
        vzm2 = vzm1; vzm1 = vz0; vz0 = vzp1; vzp1 = mv ? nvzR : 0.0f;
        hv = pH ? nhR : 0.0f; mraw = mv ? nmrawR : 0u; cl = cl1; cl1 = valid ? ncl2R : (unsigned)(BFD_CLS_FLUID | BFD_CLS_NOMEM);
        sxx = pS ? nsxxR : 0.0f; syy = pS ? nsyyR : 0.0f; szz = mv ? nszzR : 0.0f; rxx = pS ? nrxxR : 0.0f; ryy = pS ? nryyR : 0.0f; rzz = pM ? nrzzR : 0.0f;
        px = npx; py = npy; pz = npz; qx += dqx; qy += dqy;
    }
}

// CSS: Sxx, Syy and the three shear stresses of listed cells come from the compact arrays (bfd_dev::cssRow). Entry of a cell = base of its
// row (per plane; own row and halo row: one scalar each, the halo columns: one table word per lane) + the listed cells before it in the row;
// the bases a plane needs are fetched one iteration ahead, so the prefetch still issues without waiting for anything.
// WHOLE (compact form only): the engine holds a whole domain and updates V in place -- the stores share the plane bases of the loads and a
// ghost plane's Sxz / Syz are the zeros nobody ever wrote, so the full-volume fallback and its two array bases go: ten scalar registers
// fewer in a kernel that spilled twelve (each spilled one comes back through a v_readlane in every plane, and scalar address arithmetic
// that no longer fits turns into vector instructions + readfirstlane). 0.400 -> 0.384 ms at the shear medium 512^3; a further flavour without
// the peak accumulator (six spilled registers less) changed nothing measurable and was not kept. profiles/r5/velocity_solid_scalar_registers.txt
template <bool ACC, bool PML, bool CSS, bool WHOLE = false, bool QUIET = false>
__device__ __forceinline__ void velocity_solid_body_g(const bfd_dev &d, const int4 &run, int tilesX, float (*sS)[5][LH * LW],
                                                      float *__restrict__ accP, float *__restrict__ pkP)
{
    unsigned nzb = 0u;          // QUIET: bits of everything stored
    const int N1 = d.N1, N2 = d.N2;
    const int bx = run.x % tilesX, by = run.x / tilesX;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int wv = __builtin_amdgcn_readfirstlane(ty);      // wave index = tile row (uniform)
    const int i0 = bx * TX, j0 = by * TY;
    const int i = i0 + tx, j = j0 + ty;
    const bool valid = (i < N1) && (j < N2);
    const long pl = d.plane;
    const int kbeg = run.y & 0xFFFF, kend = run.y >> 16;
    const int P = d.P;
    const int own = (ty + 2) * LW + tx + 2;
    const unsigned cij = valid ? (unsigned)(j * N1 + i) : 0u;
    const unsigned c4 = cij * 4u, c2 = cij * 2u;
    const bool hasX = valid && i + 1 < N1, hasY = valid && j + 1 < N2;
    const unsigned cx2 = c2 + (hasX ? 2u : 0u), cy2 = c2 + (hasY ? (unsigned)N1 * 2u : 0u);
    const bool accA = ACC && accP != nullptr, accK = ACC && pkP != nullptr;

    HaloTask ta, tb;
    const int arrA = wv < 4 ? 1 : 2;
    ytask(tid & 255, arrA, i0, j0, N1, N2, ta);
    const int arrB = wv < 4 ? 4 : (wv == 4 ? 0 : (wv == 5 ? 2 : 3));
    if (wv < 4) ytask(tid, 4, i0, j0, N1, N2, tb);
    else if (wv < 7 && tx < XT) xtask(tx, arrB, i0, j0, N1, N2, tb);
    else { tb.lofs = -1; tb.ok = false; tb.arr = arrB; tb.gofs = 0; }
    const float *baseA = arrA == 1 ? d.Syy : d.Sxy;
    const float *baseB = arrB == 4 ? d.Syz : (arrB == 0 ? d.Sxx : (arrB == 2 ? d.Sxy : d.Sxz));
    const bool substA = arrA == 1, substB = arrB == 0;
    const unsigned bitA = BFD_CLS_EXY, bitB = arrB == 4 ? BFD_CLS_EYZ : (arrB == 2 ? BFD_CLS_EXY : BFD_CLS_EXZ);
    const unsigned offA = ta.ok ? (unsigned)ta.gofs : 0u, offB = tb.ok ? (unsigned)tb.gofs : 0u;
    float *la = &sS[0][ta.arr][ta.lofs];
    float *lb = &sS[0][tb.arr][tb.lofs < 0 ? 0 : tb.lofs];
    const bool hasB = tb.lofs >= 0;
    const int bufStride = 5 * LH * LW;

    const bool zi = PML && valid && (i < P || i >= N1 - P);
    const bool zj = PML && valid && (j < P || j >= N2 - P);
    const bool inner = valid && i >= d.ND && i < N1 - d.ND && j >= d.ND && j < N2 - d.ND;

    // compact solid state: row table of the own row and of the halo row of task A (waves 0-3: also of task B), planes as a scalar stride;
    // the halo-column lanes (waves 4-6, lanes 0..31) read their table word themselves: cells i0-2, i0-1 count back from the base of this
    // tile's row, cells i0+64, i0+65 count on from the base of the next tile's
    const long rs = CSS ? (long)N2 * d.cssStride : 0;
    const unsigned *rowp = nullptr, *rowpA = nullptr;
    const float *cbaseA = nullptr, *cbaseB = nullptr;
    unsigned rtOfs = 0;
    const int cc = tx & 3;
    if (CSS) {
        rowp = d.cssRow + ((long)2 * N2 + min(j0 + wv, N2 - 1)) * d.cssStride + bx;
        const int rA = wv & 3, gjA = j0 - 2 + (rA < 2 ? rA : TY + rA);
        rowpA = d.cssRow + ((long)2 * N2 + min(max(gjA, 0), N2 - 1)) * d.cssStride + bx;
        cbaseA = arrA == 1 ? d.cSyy : d.cSxy;
        cbaseB = arrB == 4 ? d.cSyz : (arrB == 0 ? d.cSxx : (arrB == 2 ? d.cSxy : d.cSxz));
        if (wv >= 4 && wv < 7 && tx < XT) rtOfs = (unsigned)(((2 * N2 + min(j0 + (tx >> 2), N2 - 1)) * d.cssStride + bx + (cc >= 2 ? 1 : 0)) * 4);
    }
    // a scalar load (constant address space, wave-uniform address): it returns on lgkmcnt, so nothing in the prefetch below waits for it -- as a vector
    // load + readfirstlane the compiler put an s_waitcnt vmcnt in the middle of the prefetch section (a second memory round trip per plane)
    auto rowbase = [&](const unsigned *rp, int plane) { return *(const __attribute__((address_space(4))) unsigned *)(unsigned long long)(rp + plane * rs); };
    // entry of this lane's halo cell of task B: waves 0-3 a row like task A, waves 4-6 the columns
    auto entryB = [&](unsigned rbRow, unsigned rbx, unsigned hc) {
        const bool li = tb.ok && css_listed(hc);
        if (wv < 4) return rbRow + css_rank(li);
        const unsigned long long lb = __ballot(li);
        const unsigned pb = (unsigned)(lb >> ((cc == 0 ? tx + 1 : tx - 1) & 63)) & 1u;
        return cc < 2 ? rbx - 1u - (cc == 0 ? pb : 0u) : rbx + (cc == 3 ? pb : 0u);
    };
    unsigned rbB = BFD_CSS_NONE, rbC = BFD_CSS_NONE, rbA = 0, rbx = 0;      // row bases: own row at planes kl+1 / kl+2, halo row and column word at plane kl+1

    float zzm1 = 0, zz0 = 0, zzp1 = 0, zzp2 = 0, xzm2 = 0, xzm1 = 0, xz0 = 0, xzp1 = 0, yzm2 = 0, yzm1 = 0, yz0 = 0, yzp1 = 0;
    unsigned cB = BFD_CLS_FLUID, cC = BFD_CLS_FLUID;
    {
        const float *bzz = d.Szz + kbeg * pl, *bxz = d.Sxz + kbeg * pl, *byz = d.Syz + kbeg * pl;
        const uint8_t *bc = d.cls + kbeg * pl;
        unsigned cm2 = BFD_CLS_FLUID, cm1 = BFD_CLS_FLUID, c0 = BFD_CLS_FLUID;
        if (valid) { cm2 = gl1(bc - 2 * pl, cij); cm1 = gl1(bc - pl, cij); c0 = gl1(bc, cij); cB = gl1(bc + pl, cij); cC = gl1(bc + 2 * pl, cij); }
        zzm1 = glp(bzz - pl, c4, valid); zz0 = glp(bzz, c4, valid); zzp1 = glp(bzz + pl, c4, valid); zzp2 = glp(bzz + 2 * pl, c4, valid);
        const bool fl0 = (c0 & BFD_CLS_FLUID) != 0;
        float sxx, syy, sxy;
        if (CSS) {
            const unsigned rm2 = rowbase(rowp, kbeg - 2), rm1 = rowbase(rowp, kbeg - 1), rb0 = rowbase(rowp, kbeg);
            rbB = rowbase(rowp, kbeg + 1); rbC = rowbase(rowp, kbeg + 2);
            const unsigned em2 = (rm2 + css_rank(css_listed(cm2))) * 4u, em1 = (rm1 + css_rank(css_listed(cm1))) * 4u;
            const unsigned e0 = (rb0 + css_rank(css_listed(c0))) * 4u, e1 = (rbB + css_rank(css_listed(cB))) * 4u;
            // a ghost plane (row base BFD_CSS_NONE) has no compact values: its Sxz / Syz come out of the full-volume arrays, where a Z-neighbour's
            // planes arrive (the sparse kernel keeps full-volume copies of the planes a neighbour reads); zeros at the ends of the domain
            const bool gm2 = rm2 != BFD_CSS_NONE, gm1 = rm1 != BFD_CSS_NONE, g1 = rbB != BFD_CSS_NONE;
            if (WHOLE) {        // a ghost plane of a whole domain: zeros
                xzm2 = glp(d.cSxz, em2, gm2 && (cm2 & BFD_CLS_EXZ) != 0); xzm1 = glp(d.cSxz, em1, gm1 && (cm1 & BFD_CLS_EXZ) != 0);
                xz0 = glp(d.cSxz, e0, (c0 & BFD_CLS_EXZ) != 0); xzp1 = glp(d.cSxz, e1, g1 && (cB & BFD_CLS_EXZ) != 0);
                yzm2 = glp(d.cSyz, em2, gm2 && (cm2 & BFD_CLS_EYZ) != 0); yzm1 = glp(d.cSyz, em1, gm1 && (cm1 & BFD_CLS_EYZ) != 0);
                yz0 = glp(d.cSyz, e0, (c0 & BFD_CLS_EYZ) != 0); yzp1 = glp(d.cSyz, e1, g1 && (cB & BFD_CLS_EYZ) != 0);
            } else {
            xzm2 = glp(gm2 ? d.cSxz : bxz - 2 * pl, gm2 ? em2 : c4, (cm2 & BFD_CLS_EXZ) != 0); xzm1 = glp(gm1 ? d.cSxz : bxz - pl, gm1 ? em1 : c4, (cm1 & BFD_CLS_EXZ) != 0);
            xz0 = glp(d.cSxz, e0, (c0 & BFD_CLS_EXZ) != 0); xzp1 = glp(g1 ? d.cSxz : bxz + pl, g1 ? e1 : c4, (cB & BFD_CLS_EXZ) != 0);
            yzm2 = glp(gm2 ? d.cSyz : byz - 2 * pl, gm2 ? em2 : c4, (cm2 & BFD_CLS_EYZ) != 0); yzm1 = glp(gm1 ? d.cSyz : byz - pl, gm1 ? em1 : c4, (cm1 & BFD_CLS_EYZ) != 0);
            yz0 = glp(d.cSyz, e0, (c0 & BFD_CLS_EYZ) != 0); yzp1 = glp(g1 ? d.cSyz : byz + pl, g1 ? e1 : c4, (cB & BFD_CLS_EYZ) != 0);
            }
            sxx = glp(d.cSxx, e0, css_listed(c0)); syy = glp(d.cSyy, e0, css_listed(c0));
            sxy = glp(d.cSxy, e0, (c0 & BFD_CLS_EXY) != 0);
        } else {
            xzm2 = glp(bxz - 2 * pl, c4, valid && (cm2 & BFD_CLS_EXZ)); xzm1 = glp(bxz - pl, c4, valid && (cm1 & BFD_CLS_EXZ));
            xz0 = glp(bxz, c4, valid && (c0 & BFD_CLS_EXZ)); xzp1 = glp(bxz + pl, c4, valid && (cB & BFD_CLS_EXZ));
            yzm2 = glp(byz - 2 * pl, c4, valid && (cm2 & BFD_CLS_EYZ)); yzm1 = glp(byz - pl, c4, valid && (cm1 & BFD_CLS_EYZ));
            yz0 = glp(byz, c4, valid && (c0 & BFD_CLS_EYZ)); yzp1 = glp(byz + pl, c4, valid && (cB & BFD_CLS_EYZ));
            sxx = glp(d.Sxx + kbeg * pl, c4, valid && !fl0); syy = glp(d.Syy + kbeg * pl, c4, valid && !fl0);
            sxy = glp(d.Sxy + kbeg * pl, c4, valid && (c0 & BFD_CLS_EXY));
        }
        if (valid && fl0) { sxx = zz0; syy = zz0; }
        const int bo0 = (kbeg & 1) * bufStride;
        sS[0][0][bo0 + own] = sxx; sS[0][1][bo0 + own] = syy; sS[0][2][bo0 + own] = sxy; sS[0][3][bo0 + own] = xz0; sS[0][4][bo0 + own] = yz0;
    }
    float r0 = 0;
    unsigned mraw = 0, mraw1 = 0, mx = 0, my = 0;
    if (valid) {
        const uint16_t *bM = d.mat + kbeg * pl;
        mraw = gl2(bM, c2); mraw1 = gl2(bM + pl, c2); mx = gl2(bM, cx2); my = gl2(bM, cy2);
        r0 = d.invRho[mraw & BFD_MAT_MASK];
    }
    unsigned hcA = 0, hcB = 0;
    {
        unsigned h0A = 0, h0B = 0;
        if (ta.ok) { h0A = gl1(d.cls + kbeg * pl, offA); hcA = gl1(d.cls + kbeg * pl + pl, offA); }
        if (tb.ok) { h0B = gl1(d.cls + kbeg * pl, offB); hcB = gl1(d.cls + kbeg * pl + pl, offB); }
        float ha, hb;
        if (CSS) {
            const unsigned ra0 = rowbase(rowpA, kbeg);
            const unsigned rx0 = *(BFD_GA const unsigned *)((BFD_GA const char *)gbase(d.cssRow + kbeg * rs) + gpin(rtOfs));
            rbA = rowbase(rowpA, kbeg + 1);
            rbx = *(BFD_GA const unsigned *)((BFD_GA const char *)gbase(d.cssRow + (kbeg + 1) * rs) + gpin(rtOfs));
            const unsigned ea = ra0 + css_rank(ta.ok && css_listed(h0A)), eb = entryB(ra0, rx0, h0B);
            ha = halo_value_c(cbaseA, d.Szz + kbeg * pl, substA, bitA, h0A, offA * 4u, ea * 4u, ta.ok);
            hb = halo_value_c(cbaseB, d.Szz + kbeg * pl, substB, bitB, h0B, offB * 4u, eb * 4u, tb.ok);
        } else {
            ha = halo_value_g(baseA + kbeg * pl, d.Szz + kbeg * pl, substA, bitA, h0A, offA * 4u, ta.ok);
            hb = halo_value_g(baseB + kbeg * pl, d.Szz + kbeg * pl, substB, bitB, h0B, offB * 4u, tb.ok);
        }
        la[(kbeg & 1) * bufStride] = ha;
        if (hasB) lb[(kbeg & 1) * bufStride] = hb;
    }

    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        const long ko = (long)__builtin_amdgcn_readfirstlane(kl) * pl;
        const int k = d.k0 + kl;
        const int bn = (b ^ 1) * bufStride;     // buffer of plane kl+1
#ifndef BFD_VS_HOLD_BASES
        // Round 6: the array bases are re-read from the kernel argument segment (d is the kernel's first argument; scalar loads, in the scalar cache
        // after the first plane) at three points of every plane instead of being held in scalar registers across the loop: ~19 pairs + ~10 lane
        // masks did not fit (17 scalars spilled to vector lanes, scalar address arithmetic done in vector registers). Spills 17 -> 8 (accumulating
        // flavour) / 27 -> 0 (Z-slab flavour) / 96 -> 34 (absorbing layer), 92 -> 80-82 vector registers: with the bound of 6 waves per SIMD every
        // flavour outside the layer fits 80 registers WITHOUT scratch (the round-5 build spilled 6-7 registers to scratch there and lost 12 %), so
        // three workgroups share a CU instead of two: 0.386 -> 0.362 ms at the shear medium 512^3 (the reload alone, at 5 waves: 0.386).
        // -DBFD_VS_HOLD_BASES restores the round-5 form. profiles/r6/velocity_solid_bases_reloaded.txt
        const __attribute__((address_space(4))) bfd_dev *kd = (const __attribute__((address_space(4))) bfd_dev *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kd));
#define DD (*kd)
#else
#define DD d
#endif
        // own V and sums of this plane (before the barrier), then, after it, everything plane kl+1 needs: always the same loads
        const float vx = gl4nt(DD.Vx + ko, c4), vy = gl4nt(DD.Vy + ko, c4), vz = gl4nt(DD.Vz + ko, c4);
        float av = 0, pv = 0;
        if (accA) av = gl4nt(accP + ko, c4);
        if (accK) pv = gl4(pkP + ko, c4);
        float r1 = 0, rx = 0, ry = 0;
        if (valid) {
            r1 = DD.invRho[mraw1 & BFD_MAT_MASK];       // plane kl+1, becomes r0 of the next iteration
            rx = DD.invRho[mx & BFD_MAT_MASK];
            ry = DD.invRho[my & BFD_MAT_MASK];
        }
        __syncthreads();
#ifndef BFD_VS_HOLD_BASES
        asm volatile("" : "+s"(kd));
#endif

        // The raw values are not touched before the end of the iteration (their class masks are applied there): nothing between
        // here and the staging waits for them.
        const bool more = kl + 1 < kend;                // uniform
        const unsigned nm2 = gl2(DD.mat + ko + 2 * pl, c2);                       // ghost planes make kl+2 addressable
        const unsigned nc3raw = gl1(DD.cls + ko + (kl + 2 < kend ? 3 : 2) * pl, cij);
        const bool bfl = (cB & BFD_CLS_FLUID) != 0;
        const long kn = more ? pl : 0;                  // the last iteration reads its own plane again (addressable, unused)
        const float nzzR = gl4(DD.Szz + ko + 2 * pl + kn, c4);
        bool pXZ, pYZ, pNN, pXY, takeA, takeB;
        float nxzR, nyzR, nxxR, nyyR, nxyR, nhaR, nhbR;
        unsigned nrb = BFD_CSS_NONE, nrbA = 0, nrbx = 0;
        if (CSS) {
            // entries of the own cell in planes kl+1 (in-plane values) and kl+2 (Sxz, Syz); the bases of the next iteration
            const bool lB = css_listed(cB);
            const unsigned eB = (rbB + css_rank(lB)) * 4u, eC = (rbC + css_rank(css_listed(cC))) * 4u;
            nrb = rowbase(rowp, kl + (kl + 2 < kend ? 3 : 2));
            nrbA = rowbase(rowpA, kl + (more ? 2 : 1));
            nrbx = *(BFD_GA const unsigned *)((BFD_GA const char *)gbase(DD.cssRow + (kl + (more ? 2 : 1)) * rs) + gpin(rtOfs));
            const bool gC = rbC != BFD_CSS_NONE;        // plane kl+2 may be the ghost plane nk: full-volume arrays there (uniform choice)
            pXZ = more && (cC & BFD_CLS_EXZ) && (!WHOLE || gC); pYZ = more && (cC & BFD_CLS_EYZ) && (!WHOLE || gC);
            pNN = more && lB; pXY = more && (cB & BFD_CLS_EXY);
            const unsigned eZ = (WHOLE || gC) ? eC : c4;
            if (WHOLE) { nxzR = gl4(DD.cSxz, pXZ ? eZ : 0u); nyzR = gl4(DD.cSyz, pYZ ? eZ : 0u); }
            else { nxzR = gl4(gC ? DD.cSxz : DD.Sxz + ko + pl + kn, pXZ ? eZ : 0u); nyzR = gl4(gC ? DD.cSyz : DD.Syz + ko + pl + kn, pYZ ? eZ : 0u); }
            nxxR = gl4(DD.cSxx, pNN ? eB : 0u); nyyR = gl4(DD.cSyy, pNN ? eB : 0u); nxyR = gl4(DD.cSxy, pXY ? eB : 0u);
            const unsigned eA = (rbA + css_rank(ta.ok && css_listed(hcA))) * 4u, eH = entryB(rbA, rbx, hcB) * 4u;
            const bool flA = substA && (hcA & BFD_CLS_FLUID), flB = substB && (hcB & BFD_CLS_FLUID);
            takeA = more && ta.ok && (flA || (substA ? css_listed(hcA) : (hcA & bitA) != 0));
            takeB = more && tb.ok && (flB || (substB ? css_listed(hcB) : (hcB & bitB) != 0));
            const unsigned long long paA = (unsigned long long)gbase(DD.Szz + ko + kn);
            nhaR = *(BFD_GA const float *)((flA ? paA : (unsigned long long)gbase(cbaseA)) + (takeA ? (flA ? offA * 4u : eA) : 0u));
            nhbR = *(BFD_GA const float *)((flB ? paA : (unsigned long long)gbase(cbaseB)) + (takeB ? (flB ? offB * 4u : eH) : 0u));
        } else {
            pXZ = more && valid && (cC & BFD_CLS_EXZ); pYZ = more && valid && (cC & BFD_CLS_EYZ);
            pNN = more && valid && !bfl; pXY = more && valid && (cB & BFD_CLS_EXY);
            takeA = more && ta.ok && (substA || (hcA & bitA)); takeB = more && tb.ok && (substB || (hcB & bitB));
            nxzR = gl4(DD.Sxz + ko + pl + kn, pXZ ? c4 : 0u);
            nyzR = gl4(DD.Syz + ko + pl + kn, pYZ ? c4 : 0u);
            nxxR = gl4(DD.Sxx + ko + kn, pNN ? c4 : 0u);
            nyyR = gl4(DD.Syy + ko + kn, pNN ? c4 : 0u);
            nxyR = gl4(DD.Sxy + ko + kn, pXY ? c4 : 0u);
            const unsigned long long pbA = (unsigned long long)gbase(baseA + ko + kn), paA = (unsigned long long)gbase(DD.Szz + ko + kn);
            const unsigned long long pbB = (unsigned long long)gbase(baseB + ko + kn);
            nhaR = *(BFD_GA const float *)(((substA && (hcA & BFD_CLS_FLUID)) ? paA : pbA) + (takeA ? offA * 4u : 0u));
            nhbR = *(BFD_GA const float *)(((substB && (hcB & BFD_CLS_FLUID)) ? paA : pbB) + (takeB ? offB * 4u : 0u));
        }
        const unsigned nmx = gl2(DD.mat + ko + kn, cx2), nmy = gl2(DD.mat + ko + kn, cy2);
        const unsigned nhcA = gl1(DD.cls + ko + pl + kn, offA), nhcB = gl1(DD.cls + ko + pl + kn, offB);

#ifndef BFD_VS_HOLD_BASES
        asm volatile("" : "+s"(kd));
#endif
        float *wVx = (WHOLE ? DD.Vx : DD.VxW) + ko, *wVy = (WHOLE ? DD.Vy : DD.VyW) + ko, *wVz = (WHOLE ? DD.Vz : DD.VzW) + ko;
        if (valid) {
            const float sxx = sS[b][0][own], syy = sS[b][1][own], sxy = sS[b][2][own];
            if (ACC) {
                if (inner && k >= d.ND && k < d.N3 - d.ND) {
                    const float s = (sxx + syy) + zz0;
                    const float p = -s * (1.0f / 3.0f);
                    if (accA) gs4(accP + ko, c4, av + p * p);
                    if (accK) { const float ap = fabsf(p); if (ap > pv) gs4(pkP + ko, c4, ap); }
                }
            }
            if (mraw & BFD_REFLECTOR_BIT) {
                gs4(wVx, c4, 0.f); gs4(wVy, c4, 0.f); gs4(wVz, c4, 0.f);
            } else {
                const float *pxx = &sS[b][0][own], *pyy = &sS[b][1][own], *pxy = &sS[b][2][own];
                const float *pxz = &sS[b][3][own], *pyz = &sS[b][4][own];
                float dxSxx = dplus4(pxx[-1], sxx, pxx[1], pxx[2]);
                float dySxy = dminus4(pxy[-2 * LW], pxy[-LW], sxy, pxy[LW]);
                float dzSxz = dminus4(xzm2, xzm1, xz0, xzp1);
                float dxSxy = dminus4(pxy[-2], pxy[-1], sxy, pxy[1]);
                float dySyy = dplus4(pyy[-LW], syy, pyy[LW], pyy[2 * LW]);
                float dzSyz = dminus4(yzm2, yzm1, yz0, yzp1);
                float dxSxz = dminus4(pxz[-2], pxz[-1], xz0, pxz[1]);
                float dySyz = dminus4(pyz[-2 * LW], pyz[-LW], yz0, pyz[LW]);
                float dzSzz = dplus4(zzm1, zz0, zzp1, zzp2);
                if (zi) {
                    const int xi = i < P ? i : i - (N1 - 2 * P);
                    const unsigned q = (unsigned)((kl * N2 + j) * (2 * P) + xi);
                    dxSxx = cpml(d.psi[9], q, d.axH[i], d.bxH[i], dxSxx);
                    dxSxy = cpml(d.psi[12], q, d.axI[i], d.bxI[i], dxSxy);
                    dxSxz = cpml(d.psi[15], q, d.axI[i], d.bxI[i], dxSxz);
                }
                if (zj) {
                    const int yj = j < P ? j : j - (N2 - 2 * P);
                    const unsigned q = (unsigned)((kl * (2 * P) + yj) * N1 + i);
                    dySxy = cpml(d.psi[10], q, d.ayI[j], d.byI[j], dySxy);
                    dySyy = cpml(d.psi[13], q, d.ayH[j], d.byH[j], dySyy);
                    dySyz = cpml(d.psi[16], q, d.ayI[j], d.byI[j], dySyz);
                }
                if (PML && (k < P || k >= d.N3 - P)) {
                    const int zk = k < P ? k : k - (d.N3 - 2 * P);
                    const unsigned q = (unsigned)(zk * d.plane) + cij;
                    dzSxz = cpml(d.psi[11], q, d.azI[k], d.bzI[k], dzSxz);
                    dzSyz = cpml(d.psi[14], q, d.azI[k], d.bzI[k], dzSyz);
                    dzSzz = cpml(d.psi[17], q, d.azH[k], d.bzH[k], dzSzz);
                }
                const float bxv = 0.5f * (r0 + rx), byv = 0.5f * (r0 + ry), bzv = 0.5f * (r0 + r1);
                { const float w = vx + bxv * ((dxSxx + dySxy) + dzSxz); gs4nt(wVx, c4, w); if (QUIET) nzb |= fbits(w); }
                { const float w = vy + byv * ((dxSxy + dySyy) + dzSyz); gs4nt(wVy, c4, w); if (QUIET) nzb |= fbits(w); }
                { const float w = vz + bzv * ((dxSxz + dySyz) + dzSzz); gs4nt(wVz, c4, w); if (QUIET) nzb |= fbits(w); }
            }
        }
        // masks of the prefetched values, queues, staging of plane kl+1
        const float nzz = (more && valid) ? nzzR : 0.0f, nxz = pXZ ? nxzR : 0.0f, nyz = pYZ ? nyzR : 0.0f;
        const float nxx = pNN ? nxxR : ((more && valid && bfl) ? zzp1 : 0.0f), nyy = pNN ? nyyR : ((more && valid && bfl) ? zzp1 : 0.0f);
        const float nxy = pXY ? nxyR : 0.0f, nha = takeA ? nhaR : 0.0f, nhb = takeB ? nhbR : 0.0f;
        const unsigned nc3 = (kl + 2 < kend && valid) ? nc3raw : (unsigned)BFD_CLS_FLUID;
        zzm1 = zz0; zz0 = zzp1; zzp1 = zzp2; zzp2 = nzz;
        xzm2 = xzm1; xzm1 = xz0; xz0 = xzp1; xzp1 = nxz;
        yzm2 = yzm1; yzm1 = yz0; yz0 = yzp1; yzp1 = nyz;
        sS[0][0][bn + own] = nxx; sS[0][1][bn + own] = nyy; sS[0][2][bn + own] = nxy; sS[0][3][bn + own] = xz0; sS[0][4][bn + own] = yz0;
        la[bn] = nha;
        if (hasB) lb[bn] = nhb;
        hcA = nhcA; hcB = nhcB;
        r0 = r1; mraw = mraw1; mraw1 = nm2; mx = nmx; my = nmy;
        cB = cC; cC = nc3;
        if (CSS) { rbB = rbC; rbC = nrb; rbA = nrbA; rbx = nrbx; }
    }
    if (QUIET) run_mark_active(d, run.x % tilesX, run.x / tilesX, run.y & 0xFFFF, run.y >> 16, nzb);
}
#undef DD

#ifdef BFD_STRESS_SOLID_GLOBAL
__global__ __launch_bounds__(NTHREADS, SOLID_STRESS_WAVES_PER_SIMD) void stress_solid(bfd_dev d, int tilesX, int nblocks, const int *__restrict__ xmap, const int4 *__restrict__ runs)
{
    __shared__ float sV[2][2][LH * LW];
    const int ri = run_index(nblocks, xmap);
    if (ri < 0) return;
    const int4 run = runs[ri];
    if (run.z & 8) stress_solid_body_g<true>(d, run, tilesX, sV);
    else stress_solid_body_g<false>(d, run, tilesX, sV);
}
#endif

// two kernels (the absorbing-layer flavour needs ~18 registers more and would spill inside a common one); the solid run
// list keeps the runs that touch the layer at its two ends (bfd_tiles::nSolidBP / nSolidIP)
#ifndef SOLID_VELOCITY_WAVES_PER_SIMD
#define SOLID_VELOCITY_WAVES_PER_SIMD 6      // 80 registers = three workgroups per CU; no scratch since the bases are re-read per plane (round 6). The absorbing-layer flavour needs 106 registers and gets 4
#endif
template <bool ACC, bool PML, bool CSS, bool WHOLE = false, bool QUIET = false>
__global__ __launch_bounds__(NTHREADS, PML ? 4 : SOLID_VELOCITY_WAVES_PER_SIMD) void velocity_solid(bfd_dev d, int tilesX, int nblocks, const int *__restrict__ xmap,
                                                        float *__restrict__ accP, float *__restrict__ pkP,
                                                        const int4 *__restrict__ runs)
{
    __shared__ float sS[2][5][LH * LW];
#ifdef VELOCITY_SOLID_LDS_PAD     // experiment: caps the workgroups per CU through the LDS footprint
    __shared__ float sPad[VELOCITY_SOLID_LDS_PAD];
    if (nblocks < 0) { sPad[threadIdx.x] = 1.f; sS[0][0][0] = sPad[(threadIdx.x + 1) & 511]; }
#endif
    const int ri = run_index(nblocks, xmap);
    if (ri < 0) return;
    const int4 run = runs[ri];
    if (QUIET && run_all_quiet(d, run.x % tilesX, run.x / tilesX, run.y & 0xFFFF, run.y >> 16)) return;
#ifndef BFD_VELOCITY_SOLID_FLAT      // default since round 4: GLOBAL loads, prefetch without branches (0.405 -> 0.378 ms at the shear medium 512^3)
    velocity_solid_body_g<ACC, PML, CSS, WHOLE, QUIET>(d, run, tilesX, sS, accP, pkP);
#else
    static_assert(!CSS, "the FLAT body has no compact form");
    velocity_solid_body<ACC, PML>(d, run, tilesX, sS, accP, pkP);
#endif
}

// placement probe (bfd_api.hip, choose_placement): two float32 arrays updated in place at the same cell offset along the
// engine's own runs, planes below kmax only. a' = a + b, b' = b + a: the all-zero state of step 0 stays all zero.
__global__ __launch_bounds__(NTHREADS, 8) void probe_pair(float *__restrict__ a, float *__restrict__ b, long pl, int N1, int N2, int tilesX,
                                                          int nblocks, const int4 *__restrict__ runs, int kmax)
{
    const int4 run = runs[remap_block(blockIdx.x, nblocks)];
    const int bx = run.x % tilesX, by = run.x / tilesX;
    const int i = bx * TX + threadIdx.x, j = by * TY + threadIdx.y;
    const int kbeg = run.y & 0xFFFF, kend = min(run.y >> 16, kmax);
    if (i >= N1 || j >= N2 || kbeg >= kend) return;
    const unsigned o = (unsigned)(j * N1 + i) * 4u;
    float va = F4(a + kbeg * pl, o), vb = F4(b + kbeg * pl, o);
    for (int kl = kbeg; kl < kend; kl++) {
        const long ko = (long)kl * pl;
        float na = 0.f, nb = 0.f;
        if (kl + 1 < kend) { na = F4(a + ko + pl, o); nb = F4(b + ko + pl, o); }
        F4(a + ko, o) = va + vb; F4(b + ko, o) = vb + va;
        va = na; vb = nb;
    }
}

// setup: class byte of every allocated cell (local planes -2 .. nk+1). base pointers address allocation plane 0.
__global__ void cell_classes(bfd_dev d, const uint16_t *__restrict__ matBase, uint8_t *__restrict__ clsBase, long nalloc, int nplanes)
{
    const int N1 = d.N1, N2 = d.N2;
    const long pl = d.plane;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nalloc; v += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(v / pl);
        const int r = (int)(v - (long)kk * pl);
        const int j = r / N1, i = r - j * N1;
        const unsigned raw = matBase[v];
        const int m = raw & BFD_MAT_MASK;
        const bool refl = raw & BFD_REFLECTOR_BIT;
        const float iv0 = d.invMu[m];
        unsigned c = 0;
        if (refl) c |= BFD_CLS_REFL;
        else if (!(iv0 > 0.f)) c |= BFD_CLS_FLUID;
        if (d.BP[m] == 0.f && d.BS2[m] == 0.f) c |= BFD_CLS_NOMEM;
        if (!refl && iv0 > 0.f) {
            const int i1 = min(i + 1, N1 - 1), j1 = min(j + 1, N2 - 1), k1 = min(kk + 1, nplanes - 1);
            const long r0 = (long)kk * pl + (long)j * N1, r1 = (long)kk * pl + (long)j1 * N1;
            const long z0 = (long)k1 * pl + (long)j * N1, z1 = (long)k1 * pl + (long)j1 * N1;
            const bool sx = d.invMu[matBase[r0 + i1] & BFD_MAT_MASK] > 0.f, sy = d.invMu[matBase[r1 + i] & BFD_MAT_MASK] > 0.f;
            const bool sz = d.invMu[matBase[z0 + i] & BFD_MAT_MASK] > 0.f;
            const int mX = matBase[r0 + i1] & BFD_MAT_MASK, mY = matBase[r1 + i] & BFD_MAT_MASK, mZ = matBase[z0 + i] & BFD_MAT_MASK;
            const int mXY = matBase[r1 + i1] & BFD_MAT_MASK, mXZ = matBase[z0 + i1] & BFD_MAT_MASK, mYZ = matBase[z1 + i] & BFD_MAT_MASK;
            if (sx && sy && d.invMu[mXY] > 0.f) { c |= BFD_CLS_EXY; if (mX != m || mY != m || mXY != m) c |= BFD_CLS_MIXED; }
            if (sx && sz && d.invMu[mXZ] > 0.f) { c |= BFD_CLS_EXZ; if (mX != m || mZ != m || mXZ != m) c |= BFD_CLS_MIXED; }
            if (sy && sz && d.invMu[mYZ] > 0.f) { c |= BFD_CLS_EYZ; if (mY != m || mZ != m || mYZ != m) c |= BFD_CLS_MIXED; }
        }
        clsBase[v] = (uint8_t)c;
    }
}

// class counts over the cells of the solid runs (byte accounting): one workgroup per run
__global__ __launch_bounds__(NTHREADS) void count_solid_cells(bfd_dev d, int tilesX, const int4 *__restrict__ runs, unsigned long long *__restrict__ out)
{
    const int4 run = runs[blockIdx.x];
    const int bx = run.x % tilesX, by = run.x / tilesX, kbeg = run.y & 0xFFFF, kend = run.y >> 16;
    const int i = bx * TX + threadIdx.x, j = by * TY + threadIdx.y;
    unsigned n[6] = {0, 0, 0, 0, 0, 0};
    if (i < d.N1 && j < d.N2)
        for (int kl = kbeg; kl < kend; kl++) {
            const unsigned c = d.cls[(long)kl * d.plane + (long)j * d.N1 + i];
            if (c & BFD_CLS_REFL) n[5]++;
            else n[((c & BFD_CLS_FLUID) ? 0 : 2) + ((c & BFD_CLS_NOMEM) && (c & BFD_CLS_FLUID) ? 0 : 1)]++;
            n[4] += ((c & BFD_CLS_EXY) ? 1 : 0) + ((c & BFD_CLS_EXZ) ? 1 : 0) + ((c & BFD_CLS_EYZ) ? 1 : 0);
        }
    for (int q = 0; q < 6; q++) {
        unsigned v = n[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (threadIdx.x == 0 && v) atomicAdd(&out[q], (unsigned long long)v);
    }
}

// sparse shear update: one thread per listed cell (solid centre). coef[6*t..] = A,B of the xy, xz, yz edges
// (0,0 where the 4-cell condition fails), computed once at setup with the canonical arithmetic.
__device__ __forceinline__ float ldv(const float *__restrict__ a, int N1, int N2, int i, int j, long kofs)
{
    return (i >= 0 && i < N1 && j >= 0 && j < N2) ? a[kofs + (long)j * N1 + i] : 0.0f;
}

// x / d for x < 2^31 and d >= 2 as a multiply-high and a shift (M = ceil(2^(31+L) / d), L = ceil(log2 d)): the three run-time integer
// divisions of the cell index cost ~100 of this kernel's ~310 VALU instructions per thread
struct FastDiv { unsigned M, s; };
__device__ __forceinline__ unsigned fdiv(unsigned x, FastDiv f) { return __umulhi(x, f.M) >> f.s; }

#ifndef SPARSE_WAVES_PER_SIMD
#define SPARSE_WAVES_PER_SIMD 5
#endif
#ifndef SPARSE_HOIST
#define SPARSE_HOIST 1
#endif
// NORMAL (compact solid state): the kernel also updates Sxx, Syy and their memory variables of its cell (compact arrays, entry t of this launch's
// part of the list) -- Szz / Rzz of the cell were written by the fluid stress kernel, which ran before and has advanced the absorbing-layer
// memory variables of dxVx, dyVy, dzVz: they are read here, not advanced.
template <bool NORMAL>
__global__ __launch_bounds__(256, SPARSE_WAVES_PER_SIMD) void stress_shear_sparse(bfd_dev d, const unsigned *__restrict__ cells, const unsigned *__restrict__ codes,
                                                           const float *__restrict__ tab, const float *__restrict__ coef,
                                                           float *__restrict__ Rc, long nTotal, long n, FastDiv divN1, FastDiv divPlane,
                                                           float *__restrict__ cSxy, float *__restrict__ cSxz, float *__restrict__ cSyz,
                                                           float *__restrict__ cSxx, float *__restrict__ cSyy, float *__restrict__ cRxx, float *__restrict__ cRyy,
                                                           float *__restrict__ cRxy, float *__restrict__ cRxz, float *__restrict__ cRyz)
{
    // XCD e works through the e-th contiguous eighth of the list (order: shear_order_keys): the V values a cell gathers from its
    // row / plane neighbours were fetched by blocks just before it on the SAME XCD (its own L2)
    const long t = (long)remap_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (t >= n) return;
    const int N1 = d.N1, N2 = d.N2, P = d.P;
    const long pl = d.plane;
    const unsigned c = LDNT(cells + t);
    // compact solid state: the ten values of the cell are dense streams in list order and depend on nothing -- in flight before the gathers start
    float oSxx = 0.f, oSyy = 0.f, oRxx = 0.f, oRyy = 0.f, oSxy = 0.f, oSxz = 0.f, oSyz = 0.f, oRxy = 0.f, oRxz = 0.f, oRyz = 0.f;
    if (NORMAL && SPARSE_HOIST) {
        oSxx = LDNT(cSxx + t); oSyy = LDNT(cSyy + t); oRxx = LDNT(cRxx + t); oRyy = LDNT(cRyy + t);
        oSxy = LDNT(cSxy + t); oSxz = LDNT(cSxz + t); oSyz = LDNT(cSyz + t);
        oRxy = LDNT(cRxy + t); oRxz = LDNT(cRxz + t); oRyz = LDNT(cRyz + t);
    }
    const unsigned ukl = fdiv(c, divPlane), rem = c - ukl * (unsigned)d.plane, uj = fdiv(rem, divN1);
    const int i = (int)(rem - uj * (unsigned)N1), j = (int)uj, kl = (int)ukl;
    const long ko = (long)kl * pl;
    const int k = d.k0 + kl;
    // edge coefficients: from the per-material table where the four cells of the edge hold one material (most of a bone's
    // interior), explicit otherwise (24 B per cell less to stream)
    const unsigned cw = LDNT(codes + t);
    float AP = 0.f, BP = 0.f, AS2 = 0.f, BS2 = 0.f;
    if (NORMAL) {       // material of the cell: from the code word (its fourth byte), else from the id array
        const unsigned mb = cw >> 24;
        const int m = mb ? (int)mb - 1 : (int)(d.mat[c] & BFD_MAT_MASK);
        const float4 cm = *(const float4 *)(tab + 8 * m + 4);
        AP = cm.x; BP = cm.y; AS2 = cm.z; BS2 = cm.w;
    }
    float Axy = 0.f, Bxy = 0.f, Axz = 0.f, Bxz = 0.f, Ayz = 0.f, Byz = 0.f;
    {
        const unsigned q = cw & 255u;
        if (q == 255u) { Axy = LDNT(coef + 6 * t); Bxy = LDNT(coef + 6 * t + 1); } else if (q) { const float2 ab = *(const float2 *)(tab + 8 * (q - 1)); Axy = ab.x; Bxy = ab.y; }
    }
    {
        const unsigned q = (cw >> 8) & 255u;
        if (q == 255u) { Axz = LDNT(coef + 6 * t + 2); Bxz = LDNT(coef + 6 * t + 3); } else if (q) { const float2 ab = *(const float2 *)(tab + 8 * (q - 1)); Axz = ab.x; Bxz = ab.y; }
    }
    {
        const unsigned q = (cw >> 16) & 255u;
        if (q == 255u) { Ayz = LDNT(coef + 6 * t + 4); Byz = LDNT(coef + 6 * t + 5); } else if (q) { const float2 ab = *(const float2 *)(tab + 8 * (q - 1)); Ayz = ab.x; Byz = ab.y; }
    }
    // the 21 velocities: wave-uniform bases + one 32-bit byte offset (c < 2^30); a cell whose stencil stays inside the domain in x and y
    // (all but the cells on the outermost two rows / columns) takes them without a test per value
    const unsigned c4 = c * 4u, r4 = (unsigned)N1 * 4u;
    float vx0, vy0, vz0;
    float dyVx, dxVy, dxVz, dyVz, dxVx = 0.f, dyVy = 0.f, dzVz = 0.f;
    const bool inside = i >= 2 && i + 2 < N1 && j >= 2 && j + 2 < N2;
    if (inside) {
#ifndef SPARSE_NO_X4  // round 6: the four x-taps of a component as ONE unaligned 16-byte load: 12 of the 30 gathers become 3 (0.245 -> 0.232 ms at the shear medium 512^3)
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        const f4u qx = *(const f4u *)((const char *)uni(d.Vx) + (c4 - 8u));        // Vx at i-2 .. i+1
        const f4u qy = *(const f4u *)((const char *)uni(d.Vy) + (c4 - 4u));        // Vy at i-1 .. i+2
        const f4u qz = *(const f4u *)((const char *)uni(d.Vz) + (c4 - 4u));        // Vz at i-1 .. i+2
        vx0 = qx.z; vy0 = qy.y; vz0 = qz.y;
        dyVx = dplus4(F4(d.Vx, c4 - r4), vx0, F4(d.Vx, c4 + r4), F4(d.Vx, c4 + 2 * r4));
        dxVy = dplus4(qy.x, vy0, qy.z, qy.w);
        dxVz = dplus4(qz.x, vz0, qz.z, qz.w);
        dyVz = dplus4(F4(d.Vz, c4 - r4), vz0, F4(d.Vz, c4 + r4), F4(d.Vz, c4 + 2 * r4));
        if (NORMAL) { dxVx = dminus4(qx.x, qx.y, vx0, qx.w); dyVy = dminus4(F4(d.Vy, c4 - 2 * r4), F4(d.Vy, c4 - r4), vy0, F4(d.Vy, c4 + r4)); }
#else
        vx0 = F4(d.Vx, c4); vy0 = F4(d.Vy, c4); vz0 = F4(d.Vz, c4);
        dyVx = dplus4(F4(d.Vx, c4 - r4), vx0, F4(d.Vx, c4 + r4), F4(d.Vx, c4 + 2 * r4));
        dxVy = dplus4(F4(d.Vy, c4 - 4u), vy0, F4(d.Vy, c4 + 4u), F4(d.Vy, c4 + 8u));
        dxVz = dplus4(F4(d.Vz, c4 - 4u), vz0, F4(d.Vz, c4 + 4u), F4(d.Vz, c4 + 8u));
        dyVz = dplus4(F4(d.Vz, c4 - r4), vz0, F4(d.Vz, c4 + r4), F4(d.Vz, c4 + 2 * r4));
        if (NORMAL) { dxVx = dminus4(F4(d.Vx, c4 - 8u), F4(d.Vx, c4 - 4u), vx0, F4(d.Vx, c4 + 4u)); dyVy = dminus4(F4(d.Vy, c4 - 2 * r4), F4(d.Vy, c4 - r4), vy0, F4(d.Vy, c4 + r4)); }
#endif
    } else {
        vx0 = F4(d.Vx, c4); vy0 = F4(d.Vy, c4); vz0 = F4(d.Vz, c4);
        dyVx = dplus4(ldv(d.Vx, N1, N2, i, j - 1, ko), vx0, ldv(d.Vx, N1, N2, i, j + 1, ko), ldv(d.Vx, N1, N2, i, j + 2, ko));
        dxVy = dplus4(ldv(d.Vy, N1, N2, i - 1, j, ko), vy0, ldv(d.Vy, N1, N2, i + 1, j, ko), ldv(d.Vy, N1, N2, i + 2, j, ko));
        dxVz = dplus4(ldv(d.Vz, N1, N2, i - 1, j, ko), vz0, ldv(d.Vz, N1, N2, i + 1, j, ko), ldv(d.Vz, N1, N2, i + 2, j, ko));
        dyVz = dplus4(ldv(d.Vz, N1, N2, i, j - 1, ko), vz0, ldv(d.Vz, N1, N2, i, j + 1, ko), ldv(d.Vz, N1, N2, i, j + 2, ko));
        if (NORMAL) {
            dxVx = dminus4(ldv(d.Vx, N1, N2, i - 2, j, ko), ldv(d.Vx, N1, N2, i - 1, j, ko), vx0, ldv(d.Vx, N1, N2, i + 1, j, ko));
            dyVy = dminus4(ldv(d.Vy, N1, N2, i, j - 2, ko), ldv(d.Vy, N1, N2, i, j - 1, ko), vy0, ldv(d.Vy, N1, N2, i, j + 1, ko));
        }
    }
    float dzVx = dplus4(F4(d.Vx - pl, c4), vx0, F4(d.Vx + pl, c4), F4(d.Vx + 2 * pl, c4));
    float dzVy = dplus4(F4(d.Vy - pl, c4), vy0, F4(d.Vy + pl, c4), F4(d.Vy + 2 * pl, c4));
    if (NORMAL) dzVz = dminus4(F4(d.Vz - 2 * pl, c4), F4(d.Vz - pl, c4), vz0, F4(d.Vz + pl, c4));
    if (i < P || i >= N1 - P) {
        const int xi = i < P ? i : i - (N1 - 2 * P);
        const unsigned q = (unsigned)((kl * N2 + j) * (2 * P) + xi);
        dxVy = cpml(d.psi[4], q, d.axH[i], d.bxH[i], dxVy);
        dxVz = cpml(d.psi[6], q, d.axH[i], d.bxH[i], dxVz);
        if (NORMAL) dxVx = dxVx + d.psi[0][q];          // advanced by the fluid stress kernel in this half-step
    }
    if (j < P || j >= N2 - P) {
        const int yj = j < P ? j : j - (N2 - 2 * P);
        const unsigned q = (unsigned)((kl * (2 * P) + yj) * N1 + i);
        dyVx = cpml(d.psi[3], q, d.ayH[j], d.byH[j], dyVx);
        dyVz = cpml(d.psi[8], q, d.ayH[j], d.byH[j], dyVz);
        if (NORMAL) dyVy = dyVy + d.psi[1][q];
    }
    if (k < P || k >= d.N3 - P) {
        const int zk = k < P ? k : k - (d.N3 - 2 * P);
        const unsigned q = (unsigned)(zk * d.plane) + (unsigned)(j * N1 + i);
        dzVx = cpml(d.psi[5], q, d.azH[k], d.bzH[k], dzVx);
        dzVy = cpml(d.psi[7], q, d.azH[k], d.bzH[k], dzVy);
        if (NORMAL) dzVz = dzVz + d.psi[2][q];
    }
    const float c1 = d.c1;
    if (NORMAL) {       // Sxx, Syy of the cell: the canonical expressions of stress_v2 / stress_solid
        const float sXY = dxVx + dyVy;
        const float div = sXY + dzVz;
        const float sYZ = dyVy + dzVz, sXZ = dxVx + dzVz;
        if (!SPARSE_HOIST) { oSxx = LDNT(cSxx + t); oSyy = LDNT(cSyy + t); oRxx = LDNT(cRxx + t); oRyy = LDNT(cRyy + t); }
        float rn = c1 * oRxx - (BP * div - BS2 * sYZ);
        __builtin_nontemporal_store(oSxx + ((AP * div - AS2 * sYZ) + 0.5f * (oRxx + rn)), cSxx + t); __builtin_nontemporal_store(rn, cRxx + t);
        rn = c1 * oRyy - (BP * div - BS2 * sXZ);
        __builtin_nontemporal_store(oSyy + ((AP * div - AS2 * sXZ) + 0.5f * (oRyy + rn)), cSyy + t); __builtin_nontemporal_store(rn, cRyy + t);
    }
    // memory variables: beside the list (Rc, list order) or, when the list only holds the cells the merged solid kernel leaves
    // out (Rc == null), in the full-volume arrays
    float *pRxy = NORMAL ? cRxy + t : (Rc ? Rc + t : d.Rxy + c), *pRxz = NORMAL ? cRxz + t : (Rc ? Rc + nTotal + t : d.Rxz + c), *pRyz = NORMAL ? cRyz + t : (Rc ? Rc + 2 * nTotal + t : d.Ryz + c);
    // shear stresses: in list order too when the solid state is compact (cSxy .. = the entries of this launch's part of the list)
    float *pSxy = cSxy ? cSxy + t : d.Sxy + c, *pSxz = cSxz ? cSxz + t : d.Sxz + c, *pSyz = cSyz ? cSyz + t : d.Syz + c;
    if (Axy != 0.f) {
        const float e = dyVx + dxVy;
        const float r = (NORMAL && SPARSE_HOIST) ? oRxy : LDNT(pRxy), rn = c1 * r - Bxy * e;
        *pSxy = ((NORMAL && SPARSE_HOIST) ? oSxy : LDNT(pSxy)) + (Axy * e + 0.5f * (r + rn)); *pRxy = rn;
    }
    // compact solid state in a Z-slab: the planes a neighbour reads (its ghost planes: my plane 0 and my last two) also go to the full-volume
    // Sxz / Syz, which is where the halo exchange takes them from
    const bool shared = NORMAL && (kl == 0 || kl >= d.nk - 2);
    if (Axz != 0.f) {
        const float e = dzVx + dxVz;
        const float r = (NORMAL && SPARSE_HOIST) ? oRxz : LDNT(pRxz), rn = c1 * r - Bxz * e;
        const float v = ((NORMAL && SPARSE_HOIST) ? oSxz : LDNT(pSxz)) + (Axz * e + 0.5f * (r + rn));
        *pSxz = v; *pRxz = rn;
        if (shared) d.Sxz[c] = v;
    }
    if (Ayz != 0.f) {
        const float e = dzVy + dyVz;
        const float r = (NORMAL && SPARSE_HOIST) ? oRyz : LDNT(pRyz), rn = c1 * r - Byz * e;
        const float v = ((NORMAL && SPARSE_HOIST) ? oSyz : LDNT(pSyz)) + (Ayz * e + 0.5f * (r + rn));
        *pSyz = v; *pRyz = rn;
        if (shared) d.Syz[c] = v;
    }
}

// outputs: the list-ordered memory variables of the shear stresses into the full-volume arrays
__global__ void scatter_shear_memory(bfd_dev d, const unsigned *__restrict__ cells, const float *__restrict__ Rc, long n)
{
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const unsigned c = cells[t];
        d.Rxy[c] = Rc[t]; d.Rxz[c] = Rc[n + t]; d.Ryz[c] = Rc[2 * n + t];
    }
}

// the reverse, after a list has been rebuilt in the middle of a run
__global__ void gather_shear_memory(bfd_dev d, const unsigned *__restrict__ cells, float *__restrict__ Rc, long n)
{
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const unsigned c = cells[t];
        Rc[t] = d.Rxy[c]; Rc[n + t] = d.Rxz[c]; Rc[2 * n + t] = d.Ryz[c];
    }
}

// setup: flag cells with a solid, non-reflector centre; then the edge coefficients of the listed cells
__global__ void mark_solid_cells(bfd_dev d, unsigned char *__restrict__ flag, long n, bool mixedOnly)
{
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x) {
        const unsigned raw = d.mat[v];
        const bool solid = !(raw & BFD_REFLECTOR_BIT) && d.invMu[raw & BFD_MAT_MASK] > 0.f;
        flag[v] = (solid && (!mixedOnly || (d.cls[v] & BFD_CLS_MIXED))) ? 1 : 0;
    }
}
// codes (may be null): one word per listed cell, a byte per edge (xy, xz, yz): 0 = the edge is never updated, 1 + m = its four
// cells hold material m < 254 (coefficients from the per-material table shear_material_table builds with the very same
// expression), 255 = mixed materials: the explicit coefficients in coef are read; the fourth byte: 1 + material of the cell itself (0 when
// it does not fit: the kernel reads the id array then) -- the sparse kernel's Sxx / Syy coefficients start from this word, which arrives
// with the cell index, instead of from the id behind the index (one dependent memory round trip less)
__global__ void shear_coefficients(bfd_dev d, const unsigned *__restrict__ cells, float *__restrict__ coef, unsigned *__restrict__ codes, long n)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int N1 = d.N1, N2 = d.N2;
    const long pl = d.plane;
    const unsigned c = cells[t];
    const int i = (int)(c % (unsigned)N1), j = (int)((c / (unsigned)N1) % (unsigned)N2);
    const long ko = (long)(c / (unsigned)d.plane) * pl;
    const int i1 = min(i + 1, N1 - 1), j1 = min(j + 1, N2 - 1);
    const long r0 = ko + (long)j * N1, r1 = ko + (long)j1 * N1;
    const int m = d.mat[c] & BFD_MAT_MASK;
    const int mx = d.mat[r0 + i1] & BFD_MAT_MASK, my = d.mat[r1 + i] & BFD_MAT_MASK, mz = d.mat[r0 + pl + i] & BFD_MAT_MASK;
    const int mxy = d.mat[r1 + i1] & BFD_MAT_MASK, mxz = d.mat[r0 + pl + i1] & BFD_MAT_MASK, myz = d.mat[r1 + pl + i] & BFD_MAT_MASK;
    const float iv0 = d.invMu[m], t0 = d.tauS[m], k2 = d.k2;
    const float ivx = d.invMu[mx], ivy = d.invMu[my], ivz = d.invMu[mz];
    float o[6] = {0, 0, 0, 0, 0, 0};
    {
        const float e4 = d.invMu[mxy];
        if (ivx > 0.f && ivy > 0.f && e4 > 0.f) {
            const float muH = 4.0f / ((iv0 + ivx) + (ivy + e4));
            const float tau = 0.25f * ((t0 + d.tauS[mx]) + (d.tauS[my] + d.tauS[mxy]));
            o[0] = muH * (1.0f + tau); o[1] = (muH * tau) * k2;
        }
    }
    {
        const float e4 = d.invMu[mxz];
        if (ivx > 0.f && ivz > 0.f && e4 > 0.f) {
            const float muH = 4.0f / ((iv0 + ivx) + (ivz + e4));
            const float tau = 0.25f * ((t0 + d.tauS[mx]) + (d.tauS[mz] + d.tauS[mxz]));
            o[2] = muH * (1.0f + tau); o[3] = (muH * tau) * k2;
        }
    }
    {
        const float e4 = d.invMu[myz];
        if (ivy > 0.f && ivz > 0.f && e4 > 0.f) {
            const float muH = 4.0f / ((iv0 + ivy) + (ivz + e4));
            const float tau = 0.25f * ((t0 + d.tauS[my]) + (d.tauS[mz] + d.tauS[myz]));
            o[4] = muH * (1.0f + tau); o[5] = (muH * tau) * k2;
        }
    }
    for (int q = 0; q < 6; q++) coef[6 * t + q] = o[q];
    if (codes) {
        auto code = [&](float A, int ma, int mb, int mc) { return A == 0.f ? 0u : ((m == ma && m == mb && m == mc && m < 254) ? (unsigned)(1 + m) : 255u); };
        codes[t] = code(o[0], mx, my, mxy) | (code(o[2], mx, mz, mxz) << 8) | (code(o[4], my, mz, myz) << 16) | (m < 255 ? (unsigned)(1 + m) << 24 : 0u);
    }
}
// A, B of an edge whose four cells hold material m: tab[2 m], tab[2 m + 1]
__global__ void shear_material_table(bfd_dev d, float *__restrict__ tab, int nMat)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nMat) return;
    const float iv0 = d.invMu[m], t0 = d.tauS[m], k2 = d.k2;
    float A = 0.f, B = 0.f;
    if (iv0 > 0.f) {
        const float muH = 4.0f / ((iv0 + iv0) + (iv0 + iv0));
        const float tau = 0.25f * ((t0 + t0) + (t0 + t0));
        A = muH * (1.0f + tau); B = (muH * tau) * k2;
    }
    // 32 bytes per material: (A, B) of a one-material edge as one 8-byte load, the cell's own AP, BP, AS2, BS2 as one 16-byte load (round 6: the sparse
    // kernel takes its per-material coefficients with 1 + 3 loads instead of 4 + 6)
    tab[8 * m] = A; tab[8 * m + 1] = B; tab[8 * m + 2] = 0.f; tab[8 * m + 3] = 0.f;
    tab[8 * m + 4] = d.AP[m]; tab[8 * m + 5] = d.BP[m]; tab[8 * m + 6] = d.AS2[m]; tab[8 * m + 7] = d.BS2[m];
}

// ---- dispatchers: one launch for all fluid runs; block-uniform switch on the run's flags ----
// run = (x: bx + tilesX*by, y: kbeg | kend<<16, z: flags, w: material id of UNI runs)
// flags: bit0 solid, bit1 lossy, bit2 UNI, bit3 PML, bit4 LEAN
template <bool COLLAPSED, bool QUIET = false>
__device__ __forceinline__ void stress_fluid_switch(const bfd_dev &d, const int4 &run, int tilesX, float (*sV)[2][LH * LW])
{
    const int bx = run.x % tilesX, by = run.x / tilesX, kbeg = run.y & 0xFFFF, kend = run.y >> 16, tm = run.w;
    if (QUIET && run_all_quiet(d, bx, by, kbeg, kend)) return;
    if (run.z & 1) {          // a solid run in the fluid launch (compact solid state): memory variables, several materials
        if (run.z & 8) stress_fluid_body<true, true, false, true, true, QUIET>(d, bx, by, kbeg, kend, tm, sV);
        else stress_fluid_body<true, true, false, false, true, QUIET>(d, bx, by, kbeg, kend, tm, sV);
        return;
    }
    switch ((run.z >> 1) & 7) {
    case 0: stress_fluid_body<false, COLLAPSED, false, false, false, QUIET>(d, bx, by, kbeg, kend, tm, sV); break;
    case 1: stress_fluid_body<true, COLLAPSED, false, false, false, QUIET>(d, bx, by, kbeg, kend, tm, sV); break;
    case 2: stress_fluid_body<false, COLLAPSED, true, false, false, QUIET>(d, bx, by, kbeg, kend, tm, sV); break;
    case 3: stress_fluid_body<true, COLLAPSED, true, false, false, QUIET>(d, bx, by, kbeg, kend, tm, sV); break;
    case 4: stress_fluid_body<false, COLLAPSED, false, true, false, QUIET>(d, bx, by, kbeg, kend, tm, sV); break;
    case 5: stress_fluid_body<true, COLLAPSED, false, true, false, QUIET>(d, bx, by, kbeg, kend, tm, sV); break;
    case 6: stress_fluid_body<false, COLLAPSED, true, true, false, QUIET>(d, bx, by, kbeg, kend, tm, sV); break;
    default: stress_fluid_body<true, COLLAPSED, true, true, false, QUIET>(d, bx, by, kbeg, kend, tm, sV); break;
    }
}

#ifdef BFD_EXP_XCD_CLOCK
// Experiment build: when does each XCD finish its part of a fluid launch? Every block leaves the wall clock (100 MHz) of its start and of
// its end in its own slot (plain stores: atomics on a few shared addresses serialise in one L2 channel and triple the launch time), the
// end together with the XCC_ID hardware register, so that the assumption remap_block rests on (block b runs on XCD b & 7) can be checked.
// kind 0 / 1 = stress_fluid / velocity_fluid. bfd_debug_xcd_clock() copies the slots out.
constexpr int XCLK_MAX = 1 << 17;
__device__ unsigned long long g_blkStart[2][XCLK_MAX], g_blkEnd[2][XCLK_MAX];
__device__ __forceinline__ void xcd_clock_begin(int kind) { if (threadIdx.x == 0 && blockIdx.x < XCLK_MAX) g_blkStart[kind][blockIdx.x] = wall_clock64(); }
__device__ __forceinline__ void xcd_clock_end(int kind)
{
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < XCLK_MAX) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;          // HW_REG_XCC_ID, bits 3:0
        g_blkEnd[kind][blockIdx.x] = (wall_clock64() << 4) | xcc;
    }
}
#else
__device__ __forceinline__ void xcd_clock_begin(int) {}
__device__ __forceinline__ void xcd_clock_end(int) {}
#endif

// COLLAPSED = true: all-fluid slab, every run keeps only Szz/Rzz. false: slab with solid tiles; runs flagged LEAN
// (bit4) still take the collapsed bodies, the others write all three normal stresses.
template <bool COLLAPSED, bool QUIET = false>
__global__ __launch_bounds__(NTHREADS, FLUID_WAVES_PER_SIMD) void stress_fluid(bfd_dev d, int tilesX, int nblocks, const int *__restrict__ xmap,
                                                                               const int4 *__restrict__ runs)
{
    __shared__ float sV[2][2][LH * LW];
    const int ri = run_index(nblocks, xmap);
    if (ri < 0) return;
    const int4 run = runs[ri];
    xcd_clock_begin(0);
    if (COLLAPSED || (run.z & 16)) stress_fluid_switch<true, QUIET>(d, run, tilesX, sV);
    else stress_fluid_switch<false, QUIET>(d, run, tilesX, sV);
    xcd_clock_end(0);
}

template <bool ACC, bool QUIET = false>
__global__ __launch_bounds__(NTHREADS, VELOCITY_FLUID_WAVES_PER_SIMD) void velocity_fluid(bfd_dev d, int tilesX, int nblocks, const int *__restrict__ xmap,
                                                                                 const int4 *__restrict__ runs,
                                                                                 float *__restrict__ accP, float *__restrict__ pkP)
{
    __shared__ float sS[2][LH * LW];
    const int ri = run_index(nblocks, xmap);
    if (ri < 0) return;
    const int4 run = runs[ri];
    const int bx = run.x % tilesX, by = run.x / tilesX, kbeg = run.y & 0xFFFF, kend = run.y >> 16, tm = run.w;
    xcd_clock_begin(1);
    if (QUIET && run_all_quiet(d, bx, by, kbeg, kend)) return;
    switch ((run.z >> 2) & 3) {
    case 0: velocity_fluid_body<ACC, false, false, QUIET>(d, bx, by, kbeg, kend, tm, sS, accP, pkP); break;
    case 1: velocity_fluid_body<ACC, true, false, QUIET>(d, bx, by, kbeg, kend, tm, sS, accP, pkP); break;
    case 2: velocity_fluid_body<ACC, false, true, QUIET>(d, bx, by, kbeg, kend, tm, sS, accP, pkP); break;
    default: velocity_fluid_body<ACC, true, true, QUIET>(d, bx, by, kbeg, kend, tm, sS, accP, pkP); break;
    }
    xcd_clock_end(1);
}

// one workgroup per 64 x 8 x SUBZ sub-tile. flags: bit0 = a solid cell within the sub-tile grown by 2 cells;
// bit1 = a cell of the sub-tile relaxes (BP != 0); bit2 = UNI: one material and no reflector in the grown
// region; bit3 = PML: a cell of the sub-tile lies inside an absorbing-layer zone. mat = id at its first cell.
__global__ void classify_tiles(bfd_dev d, int tilesX, int tilesY, int *__restrict__ flags, int *__restrict__ tileMat)
{
    const int tile = blockIdx.x;
    const int bx = tile % tilesX, by = (tile / tilesX) % tilesY, bz = tile / (tilesX * tilesY);
    const int i0 = bx * TX - 2, j0 = by * TY - 2, k0 = bz * SUBZ - 2;
    const int nzOwn = min(SUBZ, d.nk - bz * SUBZ);
    const int nx = TX + 4, ny = TY + 4, nz = nzOwn + 4;
    const unsigned first = d.mat[(long)(bz * SUBZ) * d.plane + (long)min(by * TY, d.N2 - 1) * d.N1 + min(bx * TX, d.N1 - 1)];
    int solid = 0, lossy = 0, lossyG = 0, refl = 0, mixed = (first & BFD_REFLECTOR_BIT) ? 1 : 0;
    for (int v = threadIdx.x; v < nx * ny * nz; v += blockDim.x) {
        const int li = v % nx, lj = (v / nx) % ny, lk = v / (nx * ny);
        const int i = i0 + li, j = j0 + lj, kl = k0 + lk;       // kl in [-2, nk+2): ghost planes exist
        if (i < 0 || i >= d.N1 || j < 0 || j >= d.N2) continue;
        const unsigned raw = d.mat[(long)kl * d.plane + (long)j * d.N1 + i];
        const int m = raw & BFD_MAT_MASK;
        if (raw != first) mixed = 1;
        if (raw & BFD_REFLECTOR_BIT) refl = 1;
        if (d.invMu[m] > 0.f) solid = 1;
        if (d.BP[m] != 0.f) { lossyG = 1; if (li >= 2 && li < nx - 2 && lj >= 2 && lj < ny - 2 && lk >= 2 && lk < nz - 2) lossy = 1; }
    }
    solid = __syncthreads_or(solid);
    lossy = __syncthreads_or(lossy);
    lossyG = __syncthreads_or(lossyG);
    refl = __syncthreads_or(refl);
    mixed = __syncthreads_or(mixed);
    if (threadIdx.x == 0) {
        const int P = d.P;
        const int xa = bx * TX, xb = min(xa + TX, d.N1), ya = by * TY, yb = min(ya + TY, d.N2);
        const int za = d.k0 + bz * SUBZ, zb = za + nzOwn;
        const bool pml = xa < P || xb > d.N1 - P || ya < P || yb > d.N2 - P || za < P || zb > d.N3 - P;
        // bit6: an absorbing-layer cell (or the domain edge) within the sub-tile grown by 2 cells, or a ragged tile
        const bool pmlGrown = xa - 2 < P || xb + 2 > d.N1 - P || ya - 2 < P || yb + 2 > d.N2 - P || za - 2 < P || zb + 2 > d.N3 - P ||
                              xb - xa < TX || yb - ya < TY;
        // bit7: a cell of the GROWN region relaxes (the fused time step recomputes the stress there)
        flags[tile] = solid | (lossy << 1) | (mixed ? 0 : 4) | (pml ? 8 : 0) | (pmlGrown ? 64 : 0) | (lossyG ? 128 : 0) | (refl ? 256 : 0);      // bit8: a reflector voxel in the grown region
        tileMat[tile] = (int)(first & BFD_MAT_MASK);
    }
}

}  // namespace

// tiles in x and y; sub-tiles of SUBZ planes in z; ZCHUNK = longest run a workgroup marches
void bfd_tile_grid(const bfd_dev &d, int *tilesX, int *tilesY, int *subZ)
{
    *tilesX = (d.N1 + TX - 1) / TX; *tilesY = (d.N2 + TY - 1) / TY; *subZ = (d.nk + SUBZ - 1) / SUBZ;
}
int bfd_tile_zchunk(void) { return ZCHUNK; }
int bfd_tile_subz(void) { return SUBZ; }
bool bfd_css_supported(void)
{
#if defined(BFD_VELOCITY_SOLID_FLAT) || defined(BFD_STRESS_SOLID_GLOBAL)
    return false;
#else
    return true;
#endif
}

void bfd_launch_probe_pair(const bfd_dev &d, hipStream_t s, const bfd_tiles *t, float *a, float *b, int kmax)
{
    const int n = t->nFluid + t->nSolid;
    if (n > 0) hipLaunchKernelGGL(probe_pair, dim3(n), dim3(TX, TY, 1), 0, s, a, b, (long)d.plane, d.N1, d.N2, (d.N1 + TX - 1) / TX, n, t->runs, kmax);
}

void bfd_launch_cell_classes(const bfd_dev &d, hipStream_t s, uint8_t *clsBase, long nalloc)
{
    const uint16_t *matBase = d.mat - 2 * (long)d.plane;
    hipLaunchKernelGGL(cell_classes, dim3((unsigned)std::min<long>((nalloc + 255) / 256, 16384)), dim3(256), 0, s, d, matBase, clsBase, nalloc, d.nk + 4);
}
void bfd_launch_count_solid_cells(const bfd_dev &d, hipStream_t s, const int4 *solidRuns, int nSolid, unsigned long long *counts6)
{
    if (nSolid) hipLaunchKernelGGL(count_solid_cells, dim3(nSolid), dim3(TX, TY, 1), 0, s, d, (d.N1 + TX - 1) / TX, solidRuns, counts6);
}

// Order of the sparse shear list. Ascending in the cell index (k, j, i), a cell's z neighbours (planes k-2 .. k+2, 9 of its 21
// gathered V values) were touched a whole plane of listed cells earlier: 1 MB of traffic at 512^2 planes, 3.8 MB at 1024^2, five
// planes of that beyond the 4 MB L2 of an XCD -- at 1024^3 the kernel moved 1.68 x its algorithmic bytes (8.58 GB per launch).
// Mode 2 (default): (z-chunk of 16 planes, band of 8 rows, plane, row, i): rows keep their whole length in x, the z neighbours are
// one band-plane away: 6.21 GB = 1.21 x at 1024^3, unchanged at 512^3 (0.89 GB, where the planes fitted), kernel time unchanged:
// it follows neither the bytes nor the number of gathers (the six x neighbours taken from the neighbouring lanes by shuffles: 4-9 %
// slower). Mode 1 cuts the rows at the 64-wide tiles as well: same bytes, 3 % slower.
// The three plane ranges the split half-steps launch separately (first / middle / last planes of a slab) stay contiguous: their
// number leads the key. profiles/r3/shear_list_order.txt
// mode 2: rows keep their whole length in x: (part of the slab, z-chunk, band of 8 rows, plane, row, i)
__device__ __forceinline__ unsigned long long shear_key2(unsigned i, unsigned j, unsigned k, int lowPlanes, int hiStart)
{
    const unsigned long long seg = (int)k < lowPlanes ? 0ull : ((int)k >= hiStart ? 2ull : 1ull);
    return (seg << 44) | ((unsigned long long)(k >> 4) << 32) | ((unsigned long long)(j >> 3) << 19) |
           ((unsigned long long)(k & 15u) << 15) | ((unsigned long long)(j & 7u) << 12) | (unsigned long long)(i & 4095u);
}
__global__ void shear_order_keys(bfd_dev d, const unsigned *__restrict__ cells, unsigned long long *__restrict__ keys, long n, int lowPlanes, int hiStart, int mode)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const unsigned c = cells[t];
    const unsigned i = c % (unsigned)d.N1, j = (c / (unsigned)d.N1) % (unsigned)d.N2, k = c / (unsigned)d.plane;
    const unsigned long long seg = (int)k < lowPlanes ? 0ull : ((int)k >= hiStart ? 2ull : 1ull);
    if (mode == 2)
        keys[t] = shear_key2(i, j, k, lowPlanes, hiStart);
    else
        keys[t] = (seg << 44) | ((unsigned long long)(k >> 4) << 32) | ((unsigned long long)(j >> 3) << 19) | ((unsigned long long)(i >> 6) << 13) |
                  ((unsigned long long)(k & 15u) << 9) | ((unsigned long long)(j & 7u) << 6) | (unsigned long long)(i & 63u);
}
// Row table of the compact solid state (bfd_dev::cssRow) for a list in order mode 2: entry (plane kk = kl + 2, row j, tile bx) = number of listed
// cells that precede cell (64 bx, j, kl) in list order = lower bound of its key in the sorted list (the keys are recomputed from the cells). A
// row of the list is contiguous and ascending in i, so this is the entry of the row's first listed cell at or after x = 64 bx; bx = tilesX
// (the key's i field carries into the row bits: still the next key in order) gives one past the row. Ghost planes: BFD_CSS_NONE.
__global__ void css_row_table(bfd_dev d, const unsigned *__restrict__ cells, long n, unsigned *__restrict__ table, int stride, int lowPlanes, int hiStart)
{
    const long total = (long)(d.nk + 4) * d.N2 * stride;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int bx = (int)(e % stride);
        const long r = e / stride;
        const int j = (int)(r % d.N2), kl = (int)(r / d.N2) - 2;
        if (kl < 0 || kl >= d.nk) { table[e] = BFD_CSS_NONE; continue; }
        const unsigned long long probe = shear_key2(0u, (unsigned)j, (unsigned)kl, lowPlanes, hiStart) + (unsigned long long)(64 * bx);
        long lo = 0, hi = n;
        while (lo < hi) {
            const long mid = (lo + hi) >> 1;
            const unsigned c = cells[mid];
            const unsigned ci = c % (unsigned)d.N1, cj = (c / (unsigned)d.N1) % (unsigned)d.N2, ck = c / (unsigned)d.plane;
            if (shear_key2(ci, cj, ck, lowPlanes, hiStart) < probe) lo = mid + 1; else hi = mid;
        }
        table[e] = (unsigned)lo;
    }
}
void bfd_launch_css_row_table(const bfd_dev &d, hipStream_t s, const unsigned *cells, long n, unsigned *rowTable, int stride, int lowPlanes, int hiStart)
{
    const long total = (long)(d.nk + 4) * d.N2 * stride;
    hipLaunchKernelGGL(css_row_table, dim3((unsigned)std::min<long>((total + 255) / 256, 65536)), dim3(256), 0, s, d, cells, n, rowTable, stride, lowPlanes, hiStart);
}
// one compact array <-> a full-volume buffer at the listed cells (outputs; a list rebuilt in the middle of a run)
__global__ void css_scatter(const unsigned *__restrict__ cells, long n, const float *__restrict__ compact, float *__restrict__ dstFull)
{
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) dstFull[cells[t]] = compact[t];
}
__global__ void css_gather(const unsigned *__restrict__ cells, long n, const float *__restrict__ srcFull, float *__restrict__ dstCompact)
{
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) dstCompact[t] = srcFull[cells[t]];
}
void bfd_launch_css_scatter(hipStream_t s, const unsigned *cells, long n, const float *compact, float *dstFull)
{
    if (n > 0) hipLaunchKernelGGL(css_scatter, dim3((unsigned)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, s, cells, n, compact, dstFull);
}
void bfd_launch_css_gather(hipStream_t s, const unsigned *cells, long n, const float *srcFull, float *dstCompact)
{
    if (n > 0) hipLaunchKernelGGL(css_gather, dim3((unsigned)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, s, cells, n, srcFull, dstCompact);
}

void bfd_launch_shear_order_keys(const bfd_dev &d, hipStream_t s, const unsigned *cells, unsigned long long *keys, long n, int lowPlanes, int hiStart, int mode)
{
    if (n) hipLaunchKernelGGL(shear_order_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, cells, keys, n, lowPlanes, hiStart, mode);
}

void bfd_launch_mark_solid(const bfd_dev &d, hipStream_t s, unsigned char *flag, long n, bool mixedOnly)
{
    hipLaunchKernelGGL(mark_solid_cells, dim3((unsigned)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, s, d, flag, n, mixedOnly);
}
void bfd_launch_scatter_shear_memory(const bfd_dev &d, hipStream_t s, const bfd_tiles *t)
{
    if (t->shearCells && t->shearR && t->nShear)
        hipLaunchKernelGGL(scatter_shear_memory, dim3((unsigned)std::min<long>((t->nShear + 255) / 256, 8192)), dim3(256), 0, s, d, t->shearCells, t->shearR, t->nShear);
}
void bfd_launch_gather_shear_memory(const bfd_dev &d, hipStream_t s, const bfd_tiles *t)
{
    if (t->shearCells && t->shearR && t->nShear)
        hipLaunchKernelGGL(gather_shear_memory, dim3((unsigned)std::min<long>((t->nShear + 255) / 256, 8192)), dim3(256), 0, s, d, t->shearCells, t->shearR, t->nShear);
}
void bfd_launch_shear_coefficients(const bfd_dev &d, hipStream_t s, const unsigned *cells, float *coef, unsigned *codes, float *tab, int nMat, long n)
{
    if (n) hipLaunchKernelGGL(shear_coefficients, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, cells, coef, codes, n);
    if (tab) hipLaunchKernelGGL(shear_material_table, dim3((unsigned)((nMat + 255) / 256)), dim3(256), 0, s, d, tab, nMat);
}

// activity map (bfd_dev::act): the sub-tiles that hold source voxels are active from the start
__global__ void mark_source_subtiles(bfd_dev d, const uint32_t *__restrict__ lin, long n)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const unsigned c = lin[t];
    const int kl = (int)(c / (unsigned)d.plane), r = (int)(c - (unsigned)kl * (unsigned)d.plane), j = r / d.N1, i = r - j * d.N1;
    d.act[((kl / SUBZ + 1) * d.actY + j / TY + 1) * d.actX + i / TX + 1] = 1;
}
void bfd_launch_mark_source_subtiles(const bfd_dev &d, hipStream_t s, const uint32_t *lin, long n)
{
    if (n > 0) hipLaunchKernelGGL(mark_source_subtiles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, lin, n);
}

void bfd_launch_classify(const bfd_dev &d, hipStream_t s, int *flagsDev, int *tileMatDev)
{
    int tx, ty, tz; bfd_tile_grid(d, &tx, &ty, &tz);
    hipLaunchKernelGGL(classify_tiles, dim3(tx * ty * tz), dim3(256), 0, s, d, tx, ty, flagsDev, tileMatDev);
}

#define BFD_LAUNCH(K, n, ...) hipLaunchKernelGGL(K, dim3(n), dim3(TX, TY, 1), 0, s, d, tilesX, n, __VA_ARGS__)
// launch over a range that has a cost-balanced map (bfd_tiles::xmap, index m): 8 x maxcnt blocks, those without a run return at once
#define BFD_LAUNCH_X(K, n, m, ...) do { const int *xm_ = t->xmap ? t->xmap + 10 * (m) : nullptr; \
        hipLaunchKernelGGL(K, dim3(xm_ ? 8 * t->xmapH[m][9] : (n)), dim3(TX, TY, 1), 0, s, d, tilesX, n, xm_, __VA_ARGS__); } while (0)
#define BFD_KT(cls, end) do { if (t->ktimer) bfd_kmark(t->ktimer, cls, end, s); } while (0)

// run list layout: [fluid boundary | fluid interior | solid boundary | solid interior]; "boundary" = runs
// inside the first and the last ZCHUNK planes of the slab (the planes a Z-neighbour reads).
// part: 0 = every run, 1 = boundary runs, 2 = interior runs
static inline void part_range(int n, int nB, int part, int *off, int *cnt)
{
    if (part == 1) { *off = 0; *cnt = nB; }
    else if (part == 2) { *off = nB; *cnt = n - nB; }
    else { *off = 0; *cnt = n; }
}

void bfd_launch_stress_v2(const bfd_dev &d, hipStream_t s0, const bfd_tiles *t, int part)
{
    const int tilesX = (d.N1 + TX - 1) / TX;
    int off, n, offS, nS;
    part_range(t->nFluid, t->nFluidB, part, &off, &n);
    part_range(t->nSolid, t->nSolidB, part, &offS, &nS);
    hipStream_t s = s0;
    // Compact solid state: Szz / Rzz of EVERY run, solid ones included, in one launch of the fluid kernel over the combined list (natural order:
    // a solid run sits between its fluid neighbours, their ring lines are shared in L2), then the sparse kernel with everything that exists
    // only at solid cells. It must come second: it reads the absorbing-layer memory variables the fluid kernel advances.
    const bool all = d.cssRow && t->runsAll;
    if (all) {
        int offA, nA;
        part_range(t->nAll, t->nAllB, part, &offA, &nA);
        if (nA) {
            BFD_KT(BFD_K_STRESS_FLUID, 0);
            if (d.act) BFD_LAUNCH((stress_fluid<true, true>), nA, (const int *)nullptr, t->runsAll + offA);
            else BFD_LAUNCH((stress_fluid<true>), nA, (const int *)nullptr, t->runsAll + offA);
            BFD_KT(BFD_K_STRESS_FLUID, 1);
        }
        n = 0; nS = 0;
    }
    const bool conc = t->sideStream[0] && !t->ktimer && (nS || (t->shearCells && t->nShear)) && n;
    if (conc) { hipEventRecord(t->sideFork, s0); hipStreamWaitEvent(t->sideStream[0], t->sideFork, 0); hipStreamWaitEvent(t->sideStream[1], t->sideFork, 0); s = t->sideStream[0]; }
    if (nS) {
        BFD_KT(BFD_K_STRESS_SOLID, 0);
        if (t->shearCells && t->merged) BFD_LAUNCH_X(stress_solid_merged, nS, BFD_XM_SS + part, t->runs + t->nFluid + offS, (const float *)t->shearTab);
        else if (t->shearCells) BFD_LAUNCH_X(stress_solid, nS, BFD_XM_SS + part, t->runs + t->nFluid + offS);
        else BFD_LAUNCH(stress_v2, nS, t->runs + t->nFluid + offS, (const unsigned short *)nullptr);     // variant 2: monolithic, dense
        BFD_KT(BFD_K_STRESS_SOLID, 1);
    }
    if (conc) s = t->sideStream[1];
    if (t->shearCells && t->nShear) {     // sparse shear: cells sorted by index; [0,lowEnd) and [highBeg,n) are the boundary chunks
        long b0 = 0, e0 = t->nShear, b1 = 0, e1 = 0;
        if (part == 1) { e0 = t->shearLowEnd; b1 = t->shearHighBeg; e1 = t->nShear; }
        else if (part == 2) { b0 = t->shearLowEnd; e0 = t->shearHighBeg; }
        BFD_KT(BFD_K_STRESS_SHEAR, 0);
        float *R0 = t->shearR ? t->shearR + b0 : nullptr, *R1 = t->shearR ? t->shearR + b1 : nullptr;
        auto magic = [](unsigned dv) { FastDiv f; unsigned L = 0; while ((1ull << L) < dv) L++; if (L == 0) L = 1;
                                       f.M = (unsigned)(((1ull << (31 + L)) + dv - 1) / dv); f.s = L - 1; return f; };
        const FastDiv dN1 = magic((unsigned)d.N1), dPl = magic((unsigned)d.plane);
        auto sparse = [&](long b, long e, float *R) {
            if (e <= b) return;
            const dim3 g((unsigned)((e - b + 255) / 256));
            if (d.cssRow) hipLaunchKernelGGL(stress_shear_sparse<true>, g, dim3(256), 0, s, d, t->shearCells + b, t->shearCodes + b, t->shearTab, t->shearCoef + 6 * b, (float *)nullptr, t->nShear, e - b, dN1, dPl,
                                             d.cSxy + b, d.cSxz + b, d.cSyz + b, d.cSxx + b, d.cSyy + b, d.cRxx + b, d.cRyy + b, d.cRxy + b, d.cRxz + b, d.cRyz + b);
            else hipLaunchKernelGGL(stress_shear_sparse<false>, g, dim3(256), 0, s, d, t->shearCells + b, t->shearCodes + b, t->shearTab, t->shearCoef + 6 * b, R, t->nShear, e - b, dN1, dPl,
                                    (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr);
        };
        sparse(b0, e0, R0); sparse(b1, e1, R1);
        BFD_KT(BFD_K_STRESS_SHEAR, 1);
    }
    s = s0;
    if (n) {
        BFD_KT(BFD_K_STRESS_FLUID, 0);
        if (d.act) BFD_LAUNCH_X((stress_fluid<true, true>), n, BFD_XM_SF + part, t->runs + off);
        else BFD_LAUNCH_X((stress_fluid<true>), n, BFD_XM_SF + part, t->runs + off);      // fluid cells keep one copy of their normal stresses (bfd_dev::cls)
        BFD_KT(BFD_K_STRESS_FLUID, 1);
    }
    if (conc) for (int q = 0; q < 2; q++) { hipEventRecord(t->sideJoin[q], t->sideStream[q]); hipStreamWaitEvent(s0, t->sideJoin[q], 0); }
}

void bfd_launch_velocity_v2(const bfd_dev &d, hipStream_t s0, float *accP, float *pkP, const bfd_tiles *t, int part)
{
    const int tilesX = (d.N1 + TX - 1) / TX;
    const bool acc = accP || pkP;
    int offF, nF, off, n;
    part_range(t->nFluid, t->nFluidB, part, &offF, &nF);
    part_range(t->nSolid, t->nSolidB, part, &off, &n);
    hipStream_t s = s0;
    const bool conc = t->sideStream[0] && !t->ktimer && n && nF && t->shearCells;
    if (conc) { hipEventRecord(t->sideFork, s0); hipStreamWaitEvent(t->sideStream[0], t->sideFork, 0); s = t->sideStream[0]; }
    if (n) {
        BFD_KT(BFD_K_VELOCITY_SOLID, 0);
        if (t->shearCells) {
            // solid list = [boundary: PML | plain][interior: plain | PML]: the plain runs of the requested part are contiguous
            const int4 *base = t->runs + t->nFluid;
            int pb = 0, pe = 0, qb = 0, qe = 0;                 // PML pieces [pb,pe) and [qb,qe); plain piece between / around
            if (part != 2) { pb = 0; pe = t->nSolidBP; }
            if (part != 1) { qb = t->nSolid - t->nSolidIP; qe = t->nSolid; }
            const int nb = part == 2 ? t->nSolidB : t->nSolidBP, ne = part == 1 ? t->nSolidB : t->nSolid - t->nSolidIP;
            auto go = [&](bool pml, int a0, int a1, int m) {
                const int cnt = a1 - a0;
                if (cnt <= 0) return;
#ifndef BFD_VELOCITY_SOLID_FLAT
                // a whole domain, or the interior runs of a Z-slab (part 2 of a split half-step: they reach no ghost plane), V in place
                if (d.act && d.cssRow && d.VxW == d.Vx && d.VyW == d.Vy && d.VzW == d.Vz) {       // quiet runs return at entry (production calls)
                    if ((d.k0 == 0 && d.nk == d.N3) || part == 2) {
                        if (pml) { if (acc) BFD_LAUNCH_X((velocity_solid<true, true, true, true, true>), cnt, m, accP, pkP, base + a0); else BFD_LAUNCH_X((velocity_solid<false, true, true, true, true>), cnt, m, accP, pkP, base + a0); }
                        else { if (acc) BFD_LAUNCH_X((velocity_solid<true, false, true, true, true>), cnt, m, accP, pkP, base + a0); else BFD_LAUNCH_X((velocity_solid<false, false, true, true, true>), cnt, m, accP, pkP, base + a0); }
                    } else {
                        if (pml) { if (acc) BFD_LAUNCH_X((velocity_solid<true, true, true, false, true>), cnt, m, accP, pkP, base + a0); else BFD_LAUNCH_X((velocity_solid<false, true, true, false, true>), cnt, m, accP, pkP, base + a0); }
                        else { if (acc) BFD_LAUNCH_X((velocity_solid<true, false, true, false, true>), cnt, m, accP, pkP, base + a0); else BFD_LAUNCH_X((velocity_solid<false, false, true, false, true>), cnt, m, accP, pkP, base + a0); }
                    }
                    return;
                }
                if (d.cssRow && d.VxW == d.Vx && d.VyW == d.Vy && d.VzW == d.Vz && ((d.k0 == 0 && d.nk == d.N3) || part == 2)) {
                    if (pml) { if (acc) BFD_LAUNCH_X((velocity_solid<true, true, true, true>), cnt, m, accP, pkP, base + a0); else BFD_LAUNCH_X((velocity_solid<false, true, true, true>), cnt, m, accP, pkP, base + a0); }
                    else { if (acc) BFD_LAUNCH_X((velocity_solid<true, false, true, true>), cnt, m, accP, pkP, base + a0); else BFD_LAUNCH_X((velocity_solid<false, false, true, true>), cnt, m, accP, pkP, base + a0); }
                    return;
                }
                if (d.cssRow) {
                    if (pml) { if (acc) BFD_LAUNCH_X((velocity_solid<true, true, true>), cnt, m, accP, pkP, base + a0); else BFD_LAUNCH_X((velocity_solid<false, true, true>), cnt, m, accP, pkP, base + a0); }
                    else { if (acc) BFD_LAUNCH_X((velocity_solid<true, false, true>), cnt, m, accP, pkP, base + a0); else BFD_LAUNCH_X((velocity_solid<false, false, true>), cnt, m, accP, pkP, base + a0); }
                    return;
                }
#endif
                if (pml) { if (acc) BFD_LAUNCH_X((velocity_solid<true, true, false>), cnt, m, accP, pkP, base + a0); else BFD_LAUNCH_X((velocity_solid<false, true, false>), cnt, m, accP, pkP, base + a0); }
                else { if (acc) BFD_LAUNCH_X((velocity_solid<true, false, false>), cnt, m, accP, pkP, base + a0); else BFD_LAUNCH_X((velocity_solid<false, false, false>), cnt, m, accP, pkP, base + a0); }
            };
            go(false, nb, ne, BFD_XM_VS + part); go(true, pb, pe, BFD_XM_VSP_LO); go(true, qb, qe, BFD_XM_VSP_HI);
        } else {                                                     // variant 2: dense
            if (acc) BFD_LAUNCH((velocity_v2<true>), n, accP, pkP, t->runs + t->nFluid + off);
            else BFD_LAUNCH((velocity_v2<false>), n, accP, pkP, t->runs + t->nFluid + off);
        }
        BFD_KT(BFD_K_VELOCITY_SOLID, 1);
    }
    s = s0;
    if (nF) {
        BFD_KT(BFD_K_VELOCITY_FLUID, 0);
        if (d.act) {
            if (acc) BFD_LAUNCH_X((velocity_fluid<true, true>), nF, BFD_XM_VF + part, t->runs + offF, accP, pkP);
            else BFD_LAUNCH_X((velocity_fluid<false, true>), nF, BFD_XM_VF + part, t->runs + offF, accP, pkP);
        } else if (acc) BFD_LAUNCH_X((velocity_fluid<true>), nF, BFD_XM_VF + part, t->runs + offF, accP, pkP);
        else BFD_LAUNCH_X((velocity_fluid<false>), nF, BFD_XM_VF + part, t->runs + offF, accP, pkP);
        BFD_KT(BFD_K_VELOCITY_FLUID, 1);
    }
    if (conc) { hipEventRecord(t->sideJoin[0], t->sideStream[0]); hipStreamWaitEvent(s0, t->sideJoin[0], 0); }
}

#ifdef BFD_EXP_XCD_CLOCK
// experiment build only: copies the first n block slots of a kind out: start[n], end[n] (end << 4 | xcc id)
extern "C" int bfd_debug_xcd_clock(int kind, int n, unsigned long long *start, unsigned long long *end)
{
    if (kind < 0 || kind > 1 || n < 0 || n > XCLK_MAX) return -1;
    if (hipMemcpyFromSymbol(start, HIP_SYMBOL(g_blkStart), (size_t)n * 8, (size_t)kind * XCLK_MAX * 8) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(end, HIP_SYMBOL(g_blkEnd), (size_t)n * 8, (size_t)kind * XCLK_MAX * 8) == hipSuccess ? 0 : -1;
}
#endif
