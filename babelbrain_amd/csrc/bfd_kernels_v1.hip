// Variant 1: plain one-thread-per-voxel kernels (no LDS). They are the simple, obviously-correct
// device statement of the canonical arithmetic (DESIGN.md "Canonical arithmetic") and the
// on-device cross-check for the LDS-tiled variant 2. gfx950 only.
//
// Replaces the per-step device kernels of the reference's solver backends (package
// BabelViscoFDTD, absent from /root/reference; call site BabelIntegrationBASE.py:2338).
#include "bfd_internal.h"

namespace {

__device__ __forceinline__ float ldxy(const float *__restrict__ a, int N1, int N2, int i, int j, long kofs)
{
    return (i >= 0 && i < N1 && j >= 0 && j < N2) ? a[kofs + (long)j * N1 + i] : 0.0f;
}
// backward difference CA*(f[0]-f[-1]) - CB*(f[+1]-f[-2])
__device__ __forceinline__ float dminus4(float fm2, float fm1, float f0, float fp1)
{
    float t1 = f0 - fm1;
    float t2 = fp1 - fm2;
    return BFD_CA * t1 - BFD_CB * t2;
}
// forward difference CA*(f[+1]-f[0]) - CB*(f[+2]-f[-1])
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

#define DXM(A) dminus4(ldxy(A, N1, N2, i - 2, j, ko), ldxy(A, N1, N2, i - 1, j, ko), ldxy(A, N1, N2, i, j, ko), ldxy(A, N1, N2, i + 1, j, ko))
#define DXP(A) dplus4(ldxy(A, N1, N2, i - 1, j, ko), ldxy(A, N1, N2, i, j, ko), ldxy(A, N1, N2, i + 1, j, ko), ldxy(A, N1, N2, i + 2, j, ko))
#define DYM(A) dminus4(ldxy(A, N1, N2, i, j - 2, ko), ldxy(A, N1, N2, i, j - 1, ko), ldxy(A, N1, N2, i, j, ko), ldxy(A, N1, N2, i, j + 1, ko))
#define DYP(A) dplus4(ldxy(A, N1, N2, i, j - 1, ko), ldxy(A, N1, N2, i, j, ko), ldxy(A, N1, N2, i, j + 1, ko), ldxy(A, N1, N2, i, j + 2, ko))
#define DZM(A) dminus4(A[c - 2 * pl], A[c - pl], A[c], A[c + pl])
#define DZP(A) dplus4(A[c - pl], A[c], A[c + pl], A[c + 2 * pl])

__global__ __launch_bounds__(256) void stress_v1(bfd_dev d)
{
    const int N1 = d.N1, N2 = d.N2;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int kl = blockIdx.z;
    if (i >= N1 || j >= N2) return;
    const long pl = d.plane;
    const long ko = (long)kl * pl;
    const long c = ko + (long)j * N1 + i;
    const int k = d.k0 + kl;
    const int P = d.P;

    const uint16_t mraw = d.mat[c];
    if (mraw & BFD_REFLECTOR_BIT) {
        d.Sxx[c] = 0.f; d.Syy[c] = 0.f; d.Szz[c] = 0.f; d.Sxy[c] = 0.f; d.Sxz[c] = 0.f; d.Syz[c] = 0.f;
        d.Rxx[c] = 0.f; d.Ryy[c] = 0.f; d.Rzz[c] = 0.f; d.Rxy[c] = 0.f; d.Rxz[c] = 0.f; d.Ryz[c] = 0.f;
        return;
    }
    const int m = mraw & BFD_MAT_MASK;

    float dxVx = DXM(d.Vx), dyVy = DYM(d.Vy), dzVz = DZM(d.Vz);
    float dyVx = DYP(d.Vx), dxVy = DXP(d.Vy);
    float dzVx = DZP(d.Vx), dxVz = DXP(d.Vz);
    float dzVy = DZP(d.Vy), dyVz = DYP(d.Vz);

    if (i < P || i >= N1 - P) {
        const int xi = i < P ? i : i - (N1 - 2 * P);
        const long q = ((long)kl * N2 + j) * (2 * P) + xi;
        dxVx = cpml(d.psi[0], q, d.axI[i], d.bxI[i], dxVx);
        dxVy = cpml(d.psi[4], q, d.axH[i], d.bxH[i], dxVy);
        dxVz = cpml(d.psi[6], q, d.axH[i], d.bxH[i], dxVz);
    }
    if (j < P || j >= N2 - P) {
        const int yj = j < P ? j : j - (N2 - 2 * P);
        const long q = ((long)kl * (2 * P) + yj) * N1 + i;
        dyVy = cpml(d.psi[1], q, d.ayI[j], d.byI[j], dyVy);
        dyVx = cpml(d.psi[3], q, d.ayH[j], d.byH[j], dyVx);
        dyVz = cpml(d.psi[8], q, d.ayH[j], d.byH[j], dyVz);
    }
    if (k < P || k >= d.N3 - P) {
        const int zk = k < P ? k : k - (d.N3 - 2 * P);
        const long q = ((long)zk * N2 + j) * N1 + i;
        dzVz = cpml(d.psi[2], q, d.azI[k], d.bzI[k], dzVz);
        dzVx = cpml(d.psi[5], q, d.azH[k], d.bzH[k], dzVx);
        dzVy = cpml(d.psi[7], q, d.azH[k], d.bzH[k], dzVy);
    }

    const float c1 = d.c1, k2 = d.k2;
    {
        const float AP = d.AP[m], BP = d.BP[m], AS2 = d.AS2[m], BS2 = d.BS2[m];
        const float sXY = dxVx + dyVy;
        const float div = sXY + dzVz;
        const float sYZ = dyVy + dzVz, sXZ = dxVx + dzVz;
        float r, rn;
        r = d.Rxx[c]; rn = c1 * r - (BP * div - BS2 * sYZ);
        d.Sxx[c] = d.Sxx[c] + ((AP * div - AS2 * sYZ) + 0.5f * (r + rn)); d.Rxx[c] = rn;
        r = d.Ryy[c]; rn = c1 * r - (BP * div - BS2 * sXZ);
        d.Syy[c] = d.Syy[c] + ((AP * div - AS2 * sXZ) + 0.5f * (r + rn)); d.Ryy[c] = rn;
        r = d.Rzz[c]; rn = c1 * r - (BP * div - BS2 * sXY);
        d.Szz[c] = d.Szz[c] + ((AP * div - AS2 * sXY) + 0.5f * (r + rn)); d.Rzz[c] = rn;
    }
    {
        const int i1 = min(i + 1, N1 - 1), j1 = min(j + 1, N2 - 1);
        const long r0 = ko + (long)j * N1, r1 = ko + (long)j1 * N1;
        const int mx = d.mat[r0 + i1] & BFD_MAT_MASK, my = d.mat[r1 + i] & BFD_MAT_MASK;
        const int mz = d.mat[r0 + pl + i] & BFD_MAT_MASK, mxy = d.mat[r1 + i1] & BFD_MAT_MASK;
        const int mxz = d.mat[r0 + pl + i1] & BFD_MAT_MASK, myz = d.mat[r1 + pl + i] & BFD_MAT_MASK;
        const float i0 = d.invMu[m], t0 = d.tauS[m];
        {
            const float a = d.invMu[mx], b = d.invMu[my], e4 = d.invMu[mxy];
            if (i0 > 0.f && a > 0.f && b > 0.f && e4 > 0.f) {
                const float muH = 4.0f / ((i0 + a) + (b + e4));
                const float tau = 0.25f * ((t0 + d.tauS[mx]) + (d.tauS[my] + d.tauS[mxy]));
                const float A = muH * (1.0f + tau), B = (muH * tau) * k2;
                const float e = dyVx + dxVy;
                const float r = d.Rxy[c], rn = c1 * r - B * e;
                d.Sxy[c] = d.Sxy[c] + (A * e + 0.5f * (r + rn)); d.Rxy[c] = rn;
            }
        }
        {
            const float a = d.invMu[mx], b = d.invMu[mz], e4 = d.invMu[mxz];
            if (i0 > 0.f && a > 0.f && b > 0.f && e4 > 0.f) {
                const float muH = 4.0f / ((i0 + a) + (b + e4));
                const float tau = 0.25f * ((t0 + d.tauS[mx]) + (d.tauS[mz] + d.tauS[mxz]));
                const float A = muH * (1.0f + tau), B = (muH * tau) * k2;
                const float e = dzVx + dxVz;
                const float r = d.Rxz[c], rn = c1 * r - B * e;
                d.Sxz[c] = d.Sxz[c] + (A * e + 0.5f * (r + rn)); d.Rxz[c] = rn;
            }
        }
        {
            const float a = d.invMu[my], b = d.invMu[mz], e4 = d.invMu[myz];
            if (i0 > 0.f && a > 0.f && b > 0.f && e4 > 0.f) {
                const float muH = 4.0f / ((i0 + a) + (b + e4));
                const float tau = 0.25f * ((t0 + d.tauS[my]) + (d.tauS[mz] + d.tauS[myz]));
                const float A = muH * (1.0f + tau), B = (muH * tau) * k2;
                const float e = dzVy + dyVz;
                const float r = d.Ryz[c], rn = c1 * r - B * e;
                d.Syz[c] = d.Syz[c] + (A * e + 0.5f * (r + rn)); d.Ryz[c] = rn;
            }
        }
    }
}

__global__ __launch_bounds__(256) void velocity_v1(bfd_dev d)
{
    const int N1 = d.N1, N2 = d.N2;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int kl = blockIdx.z;
    if (i >= N1 || j >= N2) return;
    const long pl = d.plane;
    const long ko = (long)kl * pl;
    const long c = ko + (long)j * N1 + i;
    const int k = d.k0 + kl;
    const int P = d.P;

    const uint16_t mraw = d.mat[c];
    if (mraw & BFD_REFLECTOR_BIT) { d.Vx[c] = 0.f; d.Vy[c] = 0.f; d.Vz[c] = 0.f; return; }

    float dxSxx = DXP(d.Sxx), dySxy = DYM(d.Sxy), dzSxz = DZM(d.Sxz);
    float dxSxy = DXM(d.Sxy), dySyy = DYP(d.Syy), dzSyz = DZM(d.Syz);
    float dxSxz = DXM(d.Sxz), dySyz = DYM(d.Syz), dzSzz = DZP(d.Szz);

    if (i < P || i >= N1 - P) {
        const int xi = i < P ? i : i - (N1 - 2 * P);
        const long q = ((long)kl * N2 + j) * (2 * P) + xi;
        dxSxx = cpml(d.psi[9], q, d.axH[i], d.bxH[i], dxSxx);
        dxSxy = cpml(d.psi[12], q, d.axI[i], d.bxI[i], dxSxy);
        dxSxz = cpml(d.psi[15], q, d.axI[i], d.bxI[i], dxSxz);
    }
    if (j < P || j >= N2 - P) {
        const int yj = j < P ? j : j - (N2 - 2 * P);
        const long q = ((long)kl * (2 * P) + yj) * N1 + i;
        dySxy = cpml(d.psi[10], q, d.ayI[j], d.byI[j], dySxy);
        dySyy = cpml(d.psi[13], q, d.ayH[j], d.byH[j], dySyy);
        dySyz = cpml(d.psi[16], q, d.ayI[j], d.byI[j], dySyz);
    }
    if (k < P || k >= d.N3 - P) {
        const int zk = k < P ? k : k - (d.N3 - 2 * P);
        const long q = ((long)zk * N2 + j) * N1 + i;
        dzSxz = cpml(d.psi[11], q, d.azI[k], d.bzI[k], dzSxz);
        dzSyz = cpml(d.psi[14], q, d.azI[k], d.bzI[k], dzSyz);
        dzSzz = cpml(d.psi[17], q, d.azH[k], d.bzH[k], dzSzz);
    }
    const int m = mraw & BFD_MAT_MASK;
    const int i1 = min(i + 1, N1 - 1), j1 = min(j + 1, N2 - 1);
    const float r0 = d.invRho[m];
    const float bx = 0.5f * (r0 + d.invRho[d.mat[ko + (long)j * N1 + i1] & BFD_MAT_MASK]);
    const float by = 0.5f * (r0 + d.invRho[d.mat[ko + (long)j1 * N1 + i] & BFD_MAT_MASK]);
    const float bz = 0.5f * (r0 + d.invRho[d.mat[c + pl] & BFD_MAT_MASK]);
    d.Vx[c] = d.Vx[c] + bx * ((dxSxx + dySxy) + dzSxz);
    d.Vy[c] = d.Vy[c] + by * ((dxSxy + dySyy) + dzSyz);
    d.Vz[c] = d.Vz[c] + bz * ((dxSxz + dySyz) + dzSzz);
}

}  // namespace

void bfd_launch_stress_v1(const bfd_dev &d, hipStream_t s)
{
    dim3 block(64, 4, 1), grid((d.N1 + 63) / 64, (d.N2 + 3) / 4, d.nk);
    hipLaunchKernelGGL(stress_v1, grid, block, 0, s, d);
}
void bfd_launch_velocity_v1(const bfd_dev &d, hipStream_t s)
{
    dim3 block(64, 4, 1), grid((d.N1 + 63) / 64, (d.N2 + 3) / 4, d.nk);
    hipLaunchKernelGGL(velocity_v1, grid, block, 0, s, d);
}
