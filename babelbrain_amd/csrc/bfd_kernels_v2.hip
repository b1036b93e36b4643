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

namespace {

constexpr int TX = 64;
constexpr int TY = 8;
constexpr int LW = TX + 4;          // LDS row length (floats)
constexpr int LH = TY + 4;          // LDS rows
constexpr int NTHREADS = TX * TY;   // 512
constexpr int YT = 4 * TX;          // y-halo tasks per array (4 rows x 64)
constexpr int XT = 4 * TY;          // x-halo tasks per array (4 cols x TY)

__device__ __forceinline__ float dminus4(float fm2, float fm1, float f0, float fp1)
{
    float t1 = f0 - fm1;
    float t2 = fp1 - fm2;
    return BFD_CA * t1 - BFD_CB * t2;
}
__device__ __forceinline__ float dplus4(float fm1, float f0, float fp1, float fp2)
{
    float t1 = fp1 - f0;
    float t2 = fp2 - fm1;
    return BFD_CA * t1 - BFD_CB * t2;
}
__device__ __forceinline__ float cpml(float *__restrict__ psi, long idx, float a, float b, float D)
{
    float pn = b * psi[idx] + a * D;
    psi[idx] = pn;
    return D + pn;
}

// XCD-aware tile order: consecutive block ids land on different XCDs (round robin over 8), so give
// XCD e the e-th contiguous run of tiles.
__device__ __forceinline__ int remap_block(int bid, int nblocks)
{
    const int per = nblocks >> 3;
    if (per == 0 || bid >= (per << 3)) return bid;     // tail blocks keep their id
    return (bid & 7) * per + (bid >> 3);
}

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
template <int ZC>
__global__ __launch_bounds__(NTHREADS) void stress_v2(bfd_dev d, int tilesX, int tilesY, int nblocks)
{
    __shared__ float sV[2][3][LH * LW];
    const int N1 = d.N1, N2 = d.N2;
    const int tile = remap_block(blockIdx.x, nblocks);
    const int bx = tile % tilesX, by = (tile / tilesX) % tilesY, bz = tile / (tilesX * tilesY);
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int i0 = bx * TX, j0 = by * TY;
    const int i = i0 + tx, j = j0 + ty;
    const bool valid = (i < N1) && (j < N2);
    const long pl = d.plane;
    const int kbeg = bz * ZC, kend = min(kbeg + ZC, d.nk);
    const int P = d.P;
    const int own = (ty + 2) * LW + tx + 2;
    const long cij = valid ? (long)j * N1 + i : 0;

    // halo tasks: [Vx-y, Vy-y, Vz-y] 3*256, then [Vx-x, Vy-x, Vz-x] 3*32
    HaloTask ta, tb;
    ytask(tid & 255, tid >> 8, i0, j0, N1, N2, ta);
    {
        const int t2 = tid + NTHREADS;
        if (t2 < 3 * YT) ytask(t2 - 2 * YT, 2, i0, j0, N1, N2, tb);
        else if (t2 < 3 * YT + 3 * XT) { const int u = t2 - 3 * YT; xtask(u % XT, u / XT, i0, j0, N1, N2, tb); }
        else { tb.lofs = -1; tb.ok = false; tb.arr = 0; tb.gofs = 0; }
    }
    const float *Varr[3] = {d.Vx, d.Vy, d.Vz};
    const float *pa = ta.arr == 0 ? d.Vx : (ta.arr == 1 ? d.Vy : d.Vz);
    const float *pb = tb.arr == 0 ? d.Vx : (tb.arr == 1 ? d.Vy : d.Vz);
    (void)Varr;

    // per-thread constants of the absorbing layer in x and y
    const bool zi = valid && (i < P || i >= N1 - P);
    const bool zj = valid && (j < P || j >= N2 - P);
    float axI = 0, bxI = 0, axH = 0, bxH = 0, ayI = 0, byI = 0, ayH = 0, byH = 0;
    int xi = 0, yj = 0;
    if (zi) { axI = d.axI[i]; bxI = d.bxI[i]; axH = d.axH[i]; bxH = d.bxH[i]; xi = i < P ? i : i - (N1 - 2 * P); }
    if (zj) { ayI = d.ayI[j]; byI = d.byI[j]; ayH = d.ayH[j]; byH = d.byH[j]; yj = j < P ? j : j - (N2 - 2 * P); }
    const float c1 = d.c1, k2 = d.k2;

    // z register queues, primed for plane kbeg (ghost planes make kbeg-2 .. always addressable)
    float vxm1 = 0, vx0 = 0, vxp1 = 0, vxp2 = 0, vym1 = 0, vy0 = 0, vyp1 = 0, vyp2 = 0, vzm2 = 0, vzm1 = 0, vz0 = 0, vzp1 = 0;
    if (valid) {
        const long c = (long)kbeg * pl + cij;
        vxm1 = d.Vx[c - pl]; vx0 = d.Vx[c]; vxp1 = d.Vx[c + pl]; vxp2 = d.Vx[c + 2 * pl];
        vym1 = d.Vy[c - pl]; vy0 = d.Vy[c]; vyp1 = d.Vy[c + pl]; vyp2 = d.Vy[c + 2 * pl];
        vzm2 = d.Vz[c - 2 * pl]; vzm1 = d.Vz[c - pl]; vz0 = d.Vz[c]; vzp1 = d.Vz[c + pl];
    }
    float ha = ta.ok ? pa[(long)kbeg * pl + ta.gofs] : 0.0f;
    float hb = tb.ok ? pb[(long)kbeg * pl + tb.gofs] : 0.0f;

    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        const long ko = (long)kl * pl;
        const long c = ko + cij;
        const int k = d.k0 + kl;
        // stage plane kl in LDS
        sV[b][0][own] = vx0; sV[b][1][own] = vy0; sV[b][2][own] = vz0;
        sV[b][ta.arr][ta.lofs] = ha;
        if (tb.lofs >= 0) sV[b][tb.arr][tb.lofs] = hb;
        __syncthreads();

        // this plane's state first (needed soonest), then the prefetches for plane kl+1
        uint16_t mraw = 0;
        float sxx = 0, syy = 0, szz = 0, rxx = 0, ryy = 0, rzz = 0;
        if (valid) {
            mraw = d.mat[c];
            sxx = d.Sxx[c]; syy = d.Syy[c]; szz = d.Szz[c];
            rxx = d.Rxx[c]; ryy = d.Ryy[c]; rzz = d.Rzz[c];
        }
        float nvx = 0, nvy = 0, nvz = 0, nha = 0, nhb = 0;
        if (kl + 1 < kend) {
            if (valid) { nvx = d.Vx[c + 3 * pl]; nvy = d.Vy[c + 3 * pl]; nvz = d.Vz[c + 2 * pl]; }
            if (ta.ok) nha = pa[ko + pl + ta.gofs];
            if (tb.ok) nhb = pb[ko + pl + tb.gofs];
        }

        if (valid) {
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
                d.Sxx[c] = 0.f; d.Syy[c] = 0.f; d.Szz[c] = 0.f; d.Sxy[c] = 0.f; d.Sxz[c] = 0.f; d.Syz[c] = 0.f;
                d.Rxx[c] = 0.f; d.Ryy[c] = 0.f; d.Rzz[c] = 0.f; d.Rxy[c] = 0.f; d.Rxz[c] = 0.f; d.Ryz[c] = 0.f;
            } else {
                const int m = mraw & BFD_MAT_MASK;
                if (zi) {
                    const long q = ((long)kl * N2 + j) * (2 * P) + xi;
                    dxVx = cpml(d.psi[0], q, axI, bxI, dxVx);
                    dxVy = cpml(d.psi[4], q, axH, bxH, dxVy);
                    dxVz = cpml(d.psi[6], q, axH, bxH, dxVz);
                }
                if (zj) {
                    const long q = ((long)kl * (2 * P) + yj) * N1 + i;
                    dyVy = cpml(d.psi[1], q, ayI, byI, dyVy);
                    dyVx = cpml(d.psi[3], q, ayH, byH, dyVx);
                    dyVz = cpml(d.psi[8], q, ayH, byH, dyVz);
                }
                if (k < P || k >= d.N3 - P) {
                    const int zk = k < P ? k : k - (d.N3 - 2 * P);
                    const long q = (long)zk * pl + cij;
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
                    d.Sxx[c] = sxx + ((AP * div - AS2 * sYZ) + 0.5f * (rxx + rn)); d.Rxx[c] = rn;
                    rn = c1 * ryy - (BP * div - BS2 * sXZ);
                    d.Syy[c] = syy + ((AP * div - AS2 * sXZ) + 0.5f * (ryy + rn)); d.Ryy[c] = rn;
                    rn = c1 * rzz - (BP * div - BS2 * sXY);
                    d.Szz[c] = szz + ((AP * div - AS2 * sXY) + 0.5f * (rzz + rn)); d.Rzz[c] = rn;
                }
                const float iv0 = d.invMu[m];
                if (iv0 > 0.f) {    // shear only where the centre cell is solid
                    const float t0 = d.tauS[m];
                    const int i1 = min(i + 1, N1 - 1), j1 = min(j + 1, N2 - 1);
                    const long r0 = ko + (long)j * N1, r1 = ko + (long)j1 * N1;
                    const int mx = d.mat[r0 + i1] & BFD_MAT_MASK, my = d.mat[r1 + i] & BFD_MAT_MASK;
                    const int mz = d.mat[r0 + pl + i] & BFD_MAT_MASK, mxy = d.mat[r1 + i1] & BFD_MAT_MASK;
                    const int mxz = d.mat[r0 + pl + i1] & BFD_MAT_MASK, myz = d.mat[r1 + pl + i] & BFD_MAT_MASK;
                    const float ivx = d.invMu[mx], ivy = d.invMu[my], ivz = d.invMu[mz];
                    {
                        const float e4 = d.invMu[mxy];
                        if (ivx > 0.f && ivy > 0.f && e4 > 0.f) {
                            const float muH = 4.0f / ((iv0 + ivx) + (ivy + e4));
                            const float tau = 0.25f * ((t0 + d.tauS[mx]) + (d.tauS[my] + d.tauS[mxy]));
                            const float A = muH * (1.0f + tau), B = (muH * tau) * k2;
                            const float e = dyVx + dxVy;
                            const float r = d.Rxy[c], rn = c1 * r - B * e;
                            d.Sxy[c] = d.Sxy[c] + (A * e + 0.5f * (r + rn)); d.Rxy[c] = rn;
                        }
                    }
                    {
                        const float e4 = d.invMu[mxz];
                        if (ivx > 0.f && ivz > 0.f && e4 > 0.f) {
                            const float muH = 4.0f / ((iv0 + ivx) + (ivz + e4));
                            const float tau = 0.25f * ((t0 + d.tauS[mx]) + (d.tauS[mz] + d.tauS[mxz]));
                            const float A = muH * (1.0f + tau), B = (muH * tau) * k2;
                            const float e = dzVx + dxVz;
                            const float r = d.Rxz[c], rn = c1 * r - B * e;
                            d.Sxz[c] = d.Sxz[c] + (A * e + 0.5f * (r + rn)); d.Rxz[c] = rn;
                        }
                    }
                    {
                        const float e4 = d.invMu[myz];
                        if (ivy > 0.f && ivz > 0.f && e4 > 0.f) {
                            const float muH = 4.0f / ((iv0 + ivy) + (ivz + e4));
                            const float tau = 0.25f * ((t0 + d.tauS[my]) + (d.tauS[mz] + d.tauS[myz]));
                            const float A = muH * (1.0f + tau), B = (muH * tau) * k2;
                            const float e = dzVy + dyVz;
                            const float r = d.Ryz[c], rn = c1 * r - B * e;
                            d.Syz[c] = d.Syz[c] + (A * e + 0.5f * (r + rn)); d.Ryz[c] = rn;
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
template <int ZC, bool ACC>
__global__ __launch_bounds__(NTHREADS) void velocity_v2(bfd_dev d, int tilesX, int tilesY, int nblocks,
                                                        float *__restrict__ accP, float *__restrict__ pkP)
{
    __shared__ float sS[2][5][LH * LW];
    const int N1 = d.N1, N2 = d.N2;
    const int tile = remap_block(blockIdx.x, nblocks);
    const int bx = tile % tilesX, by = (tile / tilesX) % tilesY, bz = tile / (tilesX * tilesY);
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    const int i0 = bx * TX, j0 = by * TY;
    const int i = i0 + tx, j = j0 + ty;
    const bool valid = (i < N1) && (j < N2);
    const long pl = d.plane;
    const int kbeg = bz * ZC, kend = min(kbeg + ZC, d.nk);
    const int P = d.P;
    const int own = (ty + 2) * LW + tx + 2;
    const long cij = valid ? (long)j * N1 + i : 0;

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
    const float *pa = ta.arr == 1 ? d.Syy : d.Sxy;
    const float *pb = tb.arr == 4 ? d.Syz : (tb.arr == 0 ? d.Sxx : (tb.arr == 2 ? d.Sxy : d.Sxz));

    const bool zi = valid && (i < P || i >= N1 - P);
    const bool zj = valid && (j < P || j >= N2 - P);
    float axI = 0, bxI = 0, axH = 0, bxH = 0, ayI = 0, byI = 0, ayH = 0, byH = 0;
    int xi = 0, yj = 0;
    if (zi) { axI = d.axI[i]; bxI = d.bxI[i]; axH = d.axH[i]; bxH = d.bxH[i]; xi = i < P ? i : i - (N1 - 2 * P); }
    if (zj) { ayI = d.ayI[j]; byI = d.byI[j]; ayH = d.ayH[j]; byH = d.byH[j]; yj = j < P ? j : j - (N2 - 2 * P); }
    const bool inner = valid && i >= d.ND && i < N1 - d.ND && j >= d.ND && j < N2 - d.ND;

    // z queues: Szz k-1..k+2 ; Sxz, Syz k-2..k+1 ; in-plane arrays one plane ahead
    float zzm1 = 0, zz0 = 0, zzp1 = 0, zzp2 = 0, xzm2 = 0, xzm1 = 0, xz0 = 0, xzp1 = 0, yzm2 = 0, yzm1 = 0, yz0 = 0, yzp1 = 0;
    float sxx = 0, syy = 0, sxy = 0;
    if (valid) {
        const long c = (long)kbeg * pl + cij;
        zzm1 = d.Szz[c - pl]; zz0 = d.Szz[c]; zzp1 = d.Szz[c + pl]; zzp2 = d.Szz[c + 2 * pl];
        xzm2 = d.Sxz[c - 2 * pl]; xzm1 = d.Sxz[c - pl]; xz0 = d.Sxz[c]; xzp1 = d.Sxz[c + pl];
        yzm2 = d.Syz[c - 2 * pl]; yzm1 = d.Syz[c - pl]; yz0 = d.Syz[c]; yzp1 = d.Syz[c + pl];
        sxx = d.Sxx[c]; syy = d.Syy[c]; sxy = d.Sxy[c];
    }
    float ha = ta.ok ? pa[(long)kbeg * pl + ta.gofs] : 0.0f;
    float hb = tb.ok ? pb[(long)kbeg * pl + tb.gofs] : 0.0f;

    for (int kl = kbeg; kl < kend; kl++) {
        const int b = kl & 1;
        const long ko = (long)kl * pl;
        const long c = ko + cij;
        const int k = d.k0 + kl;
        sS[b][0][own] = sxx; sS[b][1][own] = syy; sS[b][2][own] = sxy; sS[b][3][own] = xz0; sS[b][4][own] = yz0;
        sS[b][ta.arr][ta.lofs] = ha;
        if (tb.lofs >= 0) sS[b][tb.arr][tb.lofs] = hb;
        __syncthreads();

        uint16_t mraw = 0;
        float vx = 0, vy = 0, vz = 0;
        if (valid) { mraw = d.mat[c]; vx = d.Vx[c]; vy = d.Vy[c]; vz = d.Vz[c]; }
        float nzz = 0, nxz = 0, nyz = 0, nxx = 0, nyy = 0, nxy = 0, nha = 0, nhb = 0;
        if (kl + 1 < kend) {
            if (valid) {
                nzz = d.Szz[c + 3 * pl]; nxz = d.Sxz[c + 2 * pl]; nyz = d.Syz[c + 2 * pl];
                nxx = d.Sxx[c + pl]; nyy = d.Syy[c + pl]; nxy = d.Sxy[c + pl];
            }
            if (ta.ok) nha = pa[ko + pl + ta.gofs];
            if (tb.ok) nhb = pb[ko + pl + tb.gofs];
        }

        if (valid) {
            if (ACC) {
                // Pressure RMS / peak of this step's final stresses (stress sources were injected
                // before this kernel), outside the absorbing layer only
                if (inner && k >= d.ND && k < d.N3 - d.ND) {
                    const float s = (sxx + syy) + zz0;
                    const float p = -s * (1.0f / 3.0f);
                    if (accP) accP[c] = accP[c] + p * p;
                    if (pkP) { const float ap = fabsf(p); if (ap > pkP[c]) pkP[c] = ap; }
                }
            }
            if (mraw & BFD_REFLECTOR_BIT) {
                d.Vx[c] = 0.f; d.Vy[c] = 0.f; d.Vz[c] = 0.f;
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
                    const long q = ((long)kl * N2 + j) * (2 * P) + xi;
                    dxSxx = cpml(d.psi[9], q, axH, bxH, dxSxx);
                    dxSxy = cpml(d.psi[12], q, axI, bxI, dxSxy);
                    dxSxz = cpml(d.psi[15], q, axI, bxI, dxSxz);
                }
                if (zj) {
                    const long q = ((long)kl * (2 * P) + yj) * N1 + i;
                    dySxy = cpml(d.psi[10], q, ayI, byI, dySxy);
                    dySyy = cpml(d.psi[13], q, ayH, byH, dySyy);
                    dySyz = cpml(d.psi[16], q, ayI, byI, dySyz);
                }
                if (k < P || k >= d.N3 - P) {
                    const int zk = k < P ? k : k - (d.N3 - 2 * P);
                    const long q = (long)zk * pl + cij;
                    dzSxz = cpml(d.psi[11], q, d.azI[k], d.bzI[k], dzSxz);
                    dzSyz = cpml(d.psi[14], q, d.azI[k], d.bzI[k], dzSyz);
                    dzSzz = cpml(d.psi[17], q, d.azH[k], d.bzH[k], dzSzz);
                }
                const int m = mraw & BFD_MAT_MASK;
                const int i1 = min(i + 1, N1 - 1), j1 = min(j + 1, N2 - 1);
                const float r0 = d.invRho[m];
                const float bxv = 0.5f * (r0 + d.invRho[d.mat[ko + (long)j * N1 + i1] & BFD_MAT_MASK]);
                const float byv = 0.5f * (r0 + d.invRho[d.mat[ko + (long)j1 * N1 + i] & BFD_MAT_MASK]);
                const float bzv = 0.5f * (r0 + d.invRho[d.mat[c + pl] & BFD_MAT_MASK]);
                d.Vx[c] = vx + bxv * ((dxSxx + dySxy) + dzSxz);
                d.Vy[c] = vy + byv * ((dxSxy + dySyy) + dzSyz);
                d.Vz[c] = vz + bzv * ((dxSxz + dySyz) + dzSzz);
            }
        }
        zzm1 = zz0; zz0 = zzp1; zzp1 = zzp2; zzp2 = nzz;
        xzm2 = xzm1; xzm1 = xz0; xz0 = xzp1; xzp1 = nxz;
        yzm2 = yzm1; yzm1 = yz0; yz0 = yzp1; yzp1 = nyz;
        sxx = nxx; syy = nyy; sxy = nxy;
        ha = nha; hb = nhb;
    }
}

constexpr int ZCHUNK = 32;

}  // namespace

void bfd_launch_stress_v2(const bfd_dev &d, hipStream_t s)
{
    const int tilesX = (d.N1 + TX - 1) / TX, tilesY = (d.N2 + TY - 1) / TY, tilesZ = (d.nk + ZCHUNK - 1) / ZCHUNK;
    const int nblocks = tilesX * tilesY * tilesZ;
    hipLaunchKernelGGL((stress_v2<ZCHUNK>), dim3(nblocks), dim3(TX, TY, 1), 0, s, d, tilesX, tilesY, nblocks);
}

void bfd_launch_velocity_v2(const bfd_dev &d, hipStream_t s, float *accP, float *pkP)
{
    const int tilesX = (d.N1 + TX - 1) / TX, tilesY = (d.N2 + TY - 1) / TY, tilesZ = (d.nk + ZCHUNK - 1) / ZCHUNK;
    const int nblocks = tilesX * tilesY * tilesZ;
    if (accP || pkP)
        hipLaunchKernelGGL((velocity_v2<ZCHUNK, true>), dim3(nblocks), dim3(TX, TY, 1), 0, s, d, tilesX, tilesY, nblocks, accP, pkP);
    else
        hipLaunchKernelGGL((velocity_v2<ZCHUNK, false>), dim3(nblocks), dim3(TX, TY, 1), 0, s, d, tilesX, tilesY, nblocks, accP, pkP);
}
