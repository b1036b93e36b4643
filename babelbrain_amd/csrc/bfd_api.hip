// C ABI of libbabelfdtd_hip.so (include/babelfdtd.h): host logic, coefficient preparation,
// layout conversion, sources, sensors, accumulation. gfx950 only.
//
// Mirrors what BabelIntegrationBASE.py:2338-2365 hands to the reference's solver
// (package BabelViscoFDTD==1.2.4, absent from /root/reference).
#include <functional>
#include <cstring>
#include <sys/mman.h>
#include "bfd_internal.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <mutex>
#include <thread>
#include <hipcub/hipcub.hpp>

extern "C" int64_t bfd_placement_cache_release(void);
static thread_local std::string g_err;
// pinned 16 MB pieces of copy_out_large, kept from one readback of a call to the next (allocating and releasing eight of them costs 22 ms, as much as
// moving a 512^3 map); given back to the system with the placement cache (bfd_placement_cache_release: the drop-in call does that at its end)
static std::mutex g_pinMutex;
static std::vector<char *> g_pinFree;
static const size_t kPinPiece = (size_t)16 << 20;
static char *pin_take(size_t bytes)
{
    if (bytes == kPinPiece) {
        std::lock_guard<std::mutex> lk(g_pinMutex);
        if (!g_pinFree.empty()) { char *p = g_pinFree.back(); g_pinFree.pop_back(); return p; }
    }
    char *p = nullptr;
    if (hipHostMalloc((void **)&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
static void pin_give(char *p, size_t bytes)
{
    if (!p) return;
    if (bytes == kPinPiece) {
        std::lock_guard<std::mutex> lk(g_pinMutex);
        if (g_pinFree.size() < 32) { g_pinFree.push_back(p); return; }
    }
    hipHostFree(p);
}
static void pin_release_all()
{
    std::lock_guard<std::mutex> lk(g_pinMutex);
    for (char *p : g_pinFree) hipHostFree(p);
    g_pinFree.clear();
}
void bfd_set_error(const std::string &s) { g_err = s; }
#define BFD_FAIL(code, msg) do { bfd_set_error(msg); return (code); } while (0)

// ------------------------------------------------------------------------------------------------
// coefficient preparation (float64, rounded once to float32). DESIGN.md "Material model".
// ------------------------------------------------------------------------------------------------
namespace {

struct HostTables {
    std::vector<float> t;   // 7*nMat: AP,BP,AS2,BS2,invMu,tauS,invRho
    float c1, k2;
    double cmax;
};

// one standard-linear-solid mechanism, tau_sigma = 1/omega:  M(omega) = MR[(1+tau/2) + i tau/2]
void fit_sls(double rho, double c, double alpha, double q, double omega, bool exact, double &MR, double &tau)
{
    if (c <= 0.0) { MR = 0.0; tau = 0.0; return; }
    if (alpha <= 0.0) { MR = rho * c * c; tau = 0.0; return; }
    const double a = alpha / q;
    if (exact) {
        double x = a * c / omega;
        if (x > 0.4) x = 0.4;
        const double theta = 2.0 * atan(x);
        const double tt = tan(theta);
        const double t = 2.0 * tt / (1.0 - tt);
        const double re = 1.0 + 0.5 * t, im = 0.5 * t;
        const double mag = sqrt(re * re + im * im);
        const double ch = cos(0.5 * theta);
        tau = t;
        MR = rho * c * c * ch * ch / mag;
    } else {
        double Q = omega / (2.0 * c * a);
        if (Q < 1.5) Q = 1.5;
        tau = 2.0 / Q;
        MR = rho * c * c;
    }
}

void make_tables(int nMat, const double *matlist, const double *qcorr, double freq, bool exact,
                 double h, double dt, HostTables &T)
{
    const double omega = 2.0 * M_PI * freq;
    const double tauSigma = 1.0 / omega;
    const double dtoh = dt / h;
    const double half = dt / (2.0 * tauSigma);
    const double k2 = (dt / tauSigma) / (1.0 + half);
    T.c1 = (float)((1.0 - half) / (1.0 + half));
    T.k2 = (float)k2;
    T.t.assign(7 * (size_t)nMat, 0.0f);
    float *AP = T.t.data(), *BP = AP + nMat, *AS2 = BP + nMat, *BS2 = AS2 + nMat;
    float *invMu = BS2 + nMat, *tauS = invMu + nMat, *invRho = tauS + nMat;
    T.cmax = 0.0;
    for (int m = 0; m < nMat; m++) {
        const double *r = matlist + 5 * m;
        const double q = qcorr ? qcorr[m] : 1.0;
        double MRp, tauP, MRs, tauSh;
        fit_sls(r[0], r[1], r[3], q, omega, exact, MRp, tauP);
        fit_sls(r[0], r[2], r[4], q, omega, exact, MRs, tauSh);
        AP[m] = (float)(MRp * (1.0 + tauP) * dtoh);
        BP[m] = (float)(MRp * tauP * dtoh * k2);
        AS2[m] = (float)(2.0 * MRs * (1.0 + tauSh) * dtoh);
        BS2[m] = (float)(2.0 * MRs * tauSh * dtoh * k2);
        invMu[m] = (MRs > 0.0) ? (float)(1.0 / (MRs * dtoh)) : 0.0f;
        tauS[m] = (float)tauSh;
        invRho[m] = (float)(dtoh / r[0]);
        const double cu = sqrt(MRp * (1.0 + tauP) / r[0]);
        T.cmax = std::max(T.cmax, cu);
    }
}

void cpml_one(double depth, double d0, double amax, double dt, float &a, float &b)
{
    if (depth <= 0.0) { a = 0.0f; b = 0.0f; return; }
    if (depth > 1.0) depth = 1.0;
    const double d = d0 * depth * depth;
    const double al = amax * (1.0 - depth);
    const double bb = exp(-(d + al) * dt);
    b = (float)bb;
    a = (float)(d / (d + al) * (bb - 1.0));
}

// aI,bI at integer positions, aH,bH at half positions; see DESIGN.md "Absorbing layer"
void cpml_axis(int N, int ND, double cmax, double h, double dt, double freq, double R, float *out4N)
{
    const double d0 = -3.0 * cmax * log(R) / (2.0 * ND * h);
    const double amax = M_PI * freq;
    float *aI = out4N, *bI = aI + N, *aH = bI + N, *bH = aH + N;
    for (int i = 0; i < N; i++) {
        const double li = (double)(ND - i) / ND, lh = ((double)(ND - i) - 0.5) / ND;
        const double ri = (double)(i - (N - 1 - ND)) / ND, rh = ((double)(i - (N - 1 - ND)) + 0.5) / ND;
        cpml_one(std::max(li, ri), d0, amax, dt, aI[i], bI[i]);
        cpml_one(std::max(lh, rh), d0, amax, dt, aH[i], bH[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// aux kernels
// ------------------------------------------------------------------------------------------------
// strided caller layout -> x-fastest device layout. One thread per destination voxel.
// MODE 0: material ids (value must be < limit, else *flag = 1); MODE 1: boolean (value != 0)
template <int MODE, typename TO>
__global__ void gather_to_xfast(const uint32_t *__restrict__ in, long s1, long s2, long s3, TO *__restrict__ out,
                                int N1, int N2, int nk, int kSrcOffset, int kSrcMin, int kSrcMax, uint32_t limit, int *flag)
{
    const long n = (long)N1 * N2 * nk;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x) {
        const int i = (int)(v % N1);
        const int j = (int)((v / N1) % N2);
        int k = (int)(v / ((long)N1 * N2)) + kSrcOffset;
        k = min(max(k, kSrcMin), kSrcMax);   // replicate missing ghost planes
        const uint32_t x = in[i * s1 + j * s2 + k * s3];
        if (MODE == 0) {
            if (x >= limit) *flag = 1;
            out[v] = (TO)x;
        } else {
            out[v] = (TO)(x != 0u);
        }
    }
}
__global__ void or_reflector(const uint32_t *__restrict__ in, long s1, long s2, long s3, uint16_t *__restrict__ mat,
                             int N1, int N2, int nk, int clear)
{
    const long n = (long)N1 * N2 * nk;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x) {
        uint16_t m = mat[v] & BFD_MAT_MASK;
        if (!clear) {
            const int i = (int)(v % N1), j = (int)((v / N1) % N2), k = (int)(v / ((long)N1 * N2));
            if (in[i * s1 + j * s2 + k * s3]) m |= BFD_REFLECTOR_BIT;
        }
        mat[v] = m;
    }
}
// x-fastest device layout -> strided caller layout (through a dense device staging buffer)
__global__ void scatter_from_xfast(const float *__restrict__ in, float *__restrict__ out, long s1, long s2, long s3,
                                   int N1, int N2, int nk)
{
    const long n = (long)N1 * N2 * nk;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x) {
        const int i = (int)(v % N1), j = (int)((v / N1) % N2), k = (int)(v / ((long)N1 * N2));
        out[i * s1 + j * s2 + k * s3] = in[v];
    }
}
__global__ void transpose_pulse(const double *__restrict__ in, float *__restrict__ out, int nSrc, int L)
{   // in [nSrc][L] f64 -> out [L][nSrc] f32
    const long n = (long)nSrc * L;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x) {
        const int s = (int)(v % nSrc);
        const long t = v / nSrc;
        out[v] = (float)in[(long)s * L + t];
    }
}

// does cell c (local linear index) hold only the Szz/Rzz copy of its normal stresses? (every fluid cell: bfd_dev::cls)
__device__ __forceinline__ bool normal_collapsed(const bfd_dev &d, long c)
{
    return (d.cls[c] & BFD_CLS_FLUID) != 0;
}

// a value that exists only at solid cells: from the compact arrays when the solid state is compact (0 where the cell is not listed:
// a reflector, or a fluid cell asked for a shear stress), else from the full-volume array. KNOWN: the caller has the cell's list entry
// already (e, -1 = not listed) -- the sensor kernels take it from a per-sensor table, accumulate_maps from the row table and a ballot --
// otherwise css_index walks the class bytes of the row
template <bool KNOWN>
__device__ __forceinline__ float solid_value(const bfd_dev &d, const float *full, const float *comp, long c, long e)
{
    if (!d.cssRow) return full[c];
    if (!KNOWN) e = css_index(d, c);
    return e >= 0 ? comp[e] : 0.0f;
}
template <bool KNOWN>
__device__ __forceinline__ float map_value_t(const bfd_dev &d, int sel, long c, long e)
{
    switch (sel) {
    case BFD_MAP_VX: return d.Vx[c];
    case BFD_MAP_VY: return d.Vy[c];
    case BFD_MAP_VZ: return d.Vz[c];
    case BFD_MAP_SIGMAXX: return normal_collapsed(d, c) ? d.Szz[c] : solid_value<KNOWN>(d, d.Sxx, d.cSxx, c, e);     // a fluid cell keeps one copy of its normal stresses
    case BFD_MAP_SIGMAYY: return normal_collapsed(d, c) ? d.Szz[c] : solid_value<KNOWN>(d, d.Syy, d.cSyy, c, e);
    case BFD_MAP_SIGMAZZ: return d.Szz[c];
    case BFD_MAP_SIGMAXY: return solid_value<KNOWN>(d, d.Sxy, d.cSxy, c, e);
    case BFD_MAP_SIGMAXZ: return solid_value<KNOWN>(d, d.Sxz, d.cSxz, c, e);
    case BFD_MAP_SIGMAYZ: return solid_value<KNOWN>(d, d.Syz, d.cSyz, c, e);
    case BFD_MAP_PRESSURE: {
        const float zz = d.Szz[c];
        if (normal_collapsed(d, c)) return -((zz + zz) + zz) * (1.0f / 3.0f);
        float xx, yy;
        if (d.cssRow) { if (!KNOWN) e = css_index(d, c); xx = e >= 0 ? d.cSxx[e] : 0.0f; yy = e >= 0 ? d.cSyy[e] : 0.0f; }
        else { xx = d.Sxx[c]; yy = d.Syy[c]; }
        const float s = (xx + yy) + zz;
        return -s * (1.0f / 3.0f);
    }
    default: return 0.0f;
    }
}
__device__ __forceinline__ float map_value(const bfd_dev &d, int sel, long c) { return map_value_t<false>(d, sel, c, -1); }
// value of map `sel` as the outputs define it (ALLV = |V|) with the cell's list entry known
__device__ __forceinline__ float output_value(const bfd_dev &d, int sel, long c, long e)
{
    if (sel == BFD_MAP_ALLV) { const float x = d.Vx[c], y = d.Vy[c], z = d.Vz[c]; return sqrtf((x * x + y * y) + z * z); }
    return map_value_t<true>(d, sel, c, e);
}
__device__ __forceinline__ bool map_needs_entry(int sel)
{
    return sel == BFD_MAP_SIGMAXX || sel == BFD_MAP_SIGMAYY || sel == BFD_MAP_SIGMAXY || sel == BFD_MAP_SIGMAXZ || sel == BFD_MAP_SIGMAYZ || sel == BFD_MAP_PRESSURE;
}
// list entries of the sensor voxels (compact solid state), resolved once per list: a capture is a plain gather again
__global__ void sensor_entries(bfd_dev d, const uint32_t *__restrict__ lin, long nSens, int *__restrict__ out)
{
    for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < nSens; s += (long)gridDim.x * blockDim.x)
        out[s] = (int)css_index(d, (long)lin[s]);
}
// fluid cells keep only Szz/Rzz of their identical normal stresses: restore the other copies
__global__ void expand_normal(bfd_dev d, long n)
{
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x) {
        if (!normal_collapsed(d, v)) continue;
        const float s = d.Szz[v], r = d.Rzz[v];
        d.Sxx[v] = s; d.Syy[v] = s; d.Rxx[v] = r; d.Ryy[v] = r;
    }
}
// output assembly (bfd_get_field, compact solid state): out = src at the cells that keep one copy of their normal stresses
__global__ void copy_at_fluid_cells(bfd_dev d, const float *__restrict__ src, float *__restrict__ out, long n)
{
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x)
        if (normal_collapsed(d, v)) out[v] = src[v];
}
__device__ __forceinline__ float map_sq(const bfd_dev &d, int sel, long c)
{
    if (sel == BFD_MAP_ALLV) {
        const float x = d.Vx[c], y = d.Vy[c], z = d.Vz[c];
        return (x * x + y * y) + z * z;
    }
    const float v = map_value(d, sel, c);
    return v * v;
}

struct SelList { int n; int sel[BFD_MAP_COUNT]; int skip[BFD_MAP_COUNT]; };

// RMS / peak accumulation outside the absorbing layer (generic path, any map selection). A wave = 64 consecutive x cells of one row: with a
// compact solid state the cell's list entry is the row-table base of this x tile + the listed lanes below (one ballot), once for all selections
__global__ __launch_bounds__(256) void accumulate_maps(bfd_dev d, SelList L, float *__restrict__ acc, float *__restrict__ pk, long nloc)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int kl = blockIdx.z;
    const int k = d.k0 + kl;
    const bool inDomain = i < d.N1 && j < d.N2;
    const long c = (long)kl * d.plane + (long)j * d.N1 + i;
    long e = -1;
    if (d.cssRow) {
        bool need = false;
        for (int q = 0; q < L.n; q++) need = need || (!L.skip[q] && map_needs_entry(L.sel[q]));
        if (need) {                                                      // block-uniform
            const bool listed = inDomain && css_listed(d.cls[c]);
            const unsigned rank = css_rank(listed);
            const unsigned rb = j < d.N2 ? d.cssRow[((long)(kl + 2) * d.N2 + j) * d.cssStride + blockIdx.x] : BFD_CSS_NONE;
            if (listed && rb != BFD_CSS_NONE) e = (long)rb + rank;
        }
    }
    if (i < d.ND || i >= d.N1 - d.ND || j < d.ND || j >= d.N2 - d.ND || k < d.ND || k >= d.N3 - d.ND) return;
    for (int q = 0; q < L.n; q++) {
        if (L.skip[q]) continue;    // accumulated inside the velocity kernel
        const float v = output_value(d, L.sel[q], c, e);
        if (acc) acc[q * nloc + c] = acc[q * nloc + c] + (L.sel[q] == BFD_MAP_ALLV ? map_sq(d, BFD_MAP_ALLV, c) : v * v);
        if (pk) {
            const float a = fabsf(v);
            if (a > pk[q * nloc + c]) pk[q * nloc + c] = a;
        }
    }
}
__global__ void finalize_rms(const float *__restrict__ acc, float *__restrict__ out, long n, float cnt)
{
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x)
        out[v] = sqrtf(acc[v] / cnt);
}
__global__ void last_map(bfd_dev d, int sel, float *__restrict__ out, long n)
{
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x)
        out[v] = (sel == BFD_MAP_ALLV) ? sqrtf(map_sq(d, BFD_MAP_ALLV, v)) : map_value(d, sel, v);
}

// Sensor sets that are a dense box of voxels (what the caller's CreateSensorMap makes: everything inside the absorbing layer past the source plane,
// BASE:2279-2290) need no index list: sensor s of the box is voxel (i0 + s % bx, j0 + (s / bx) % by, k0 + s / (bx by)) -- the order of the ascending
// x-fastest linear index the list is in. lin == null selects that form (4 of the 13 bytes a captured sample moved were the index).
struct SensorBox { unsigned bx, bxy, i0, j0, k0; };
__device__ __forceinline__ long sensor_cell(const bfd_dev &d, const uint32_t *__restrict__ lin, const SensorBox &B, long s)
{
    if (lin) return (long)lin[s];
    const unsigned u = (unsigned)s, kk = u / B.bxy, r = u - kk * B.bxy, jj = r / B.bx, ii = r - jj * B.bx;
    return (long)(B.k0 + kk) * d.plane + (long)(B.j0 + jj) * d.N1 + (B.i0 + ii);
}
// bounding box of a sensor list (min / max of i, j, k): six atomics per workgroup
__global__ __launch_bounds__(256) void sensor_bounds(bfd_dev d, const uint32_t *__restrict__ lin, long n, unsigned *__restrict__ mm)
{
    unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (long)gridDim.x * blockDim.x) {
        const unsigned c = lin[s], kl = c / (unsigned)d.plane, r = c - kl * (unsigned)d.plane, j = r / (unsigned)d.N1, i = r - j * (unsigned)d.N1;
        lo[0] = min(lo[0], i); hi[0] = max(hi[0], i); lo[1] = min(lo[1], j); hi[1] = max(hi[1], j); lo[2] = min(lo[2], kl); hi[2] = max(hi[2], kl);
    }
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) { lo[a] = min(lo[a], (unsigned)__shfl_down((int)lo[a], o)); hi[a] = max(hi[a], (unsigned)__shfl_down((int)hi[a], o)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(mm + a, lo[a]); atomicMax(mm + 3 + a, hi[a]); }
    }
}
// sensors: out[q][col][s]; ent = the sensors' list entries (compact solid state) or null
__global__ void record_sensors(bfd_dev d, SelList L, const uint32_t *__restrict__ lin, SensorBox B, const int *__restrict__ ent, long nSens,
                               float *__restrict__ out, int col, int nTs)
{
    for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < nSens; s += (long)gridDim.x * blockDim.x) {
        const long c = sensor_cell(d, lin, B, s);
        const long e = (ent && !normal_collapsed(d, c)) ? (long)ent[s] : -1;
        for (int q = 0; q < L.n; q++)
            out[((long)q * nTs + col) * nSens + s] = output_value(d, L.sel[q], c, e);
    }
}
// sensorMode 1: the sample of this step goes straight into the running single-bin DFT sums and peaks of its sensor
// (same arithmetic, sample by sample, as dft_series applies to a stored series)
__global__ void accumulate_sensor_dft(bfd_dev d, SelList L, const uint32_t *__restrict__ lin, SensorBox B, const int *__restrict__ ent, long nSens,
                                      double *__restrict__ acc, float *__restrict__ pk, int col, int nTs, int bin)
{
    const int r = (int)(((long)bin * col) % nTs);                 // exact phase index
    double sn, cs;
    sincospi(2.0 * (double)r / (double)nTs, &sn, &cs);
    for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < nSens; s += (long)gridDim.x * blockDim.x) {
        const long c = sensor_cell(d, lin, B, s);
        const long e = (ent && !normal_collapsed(d, c)) ? (long)ent[s] : -1;
        for (int q = 0; q < L.n; q++) {
            const float x = output_value(d, L.sel[q], c, e);
            double *a = acc + 2 * ((long)q * nSens + s);
            a[0] += (double)x * cs; a[1] -= (double)x * sn;
            float *p = pk + (long)q * nSens + s;
            *p = fmaxf(*p, x);
        }
    }
}
__global__ void fill_float(float *__restrict__ p, long n, float v)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void finalize_sensor_dft(const double *__restrict__ acc, float *__restrict__ out, long n2, double sc)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) out[i] = (float)(acc[i] * sc);
}

// [q][nTs][nSens] -> [q][nSens][nTs]
__global__ void transpose_sensors(const float *__restrict__ in, float *__restrict__ out, long nSens, int nTs, int nq)
{
    const long n = nSens * nTs * nq;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x) {
        const int t = (int)(v % nTs);
        const long s = (v / nTs) % nSens;
        const long q = v / ((long)nTs * nSens);
        out[v] = in[(q * nTs + t) * nSens + s];
    }
}

// the same for elements [v0, v0 + cnt) of the transposed block only (out = a piece buffer): the sensor series leave the device piece by piece
__global__ void transpose_sensors_range(const float *__restrict__ in, float *__restrict__ out, long nSens, int nTs, long v0, long cnt)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (long)gridDim.x * blockDim.x) {
        const long v = v0 + i;
        const int t = (int)(v % nTs);
        const long s = (v / nTs) % nSens;
        const long q = v / ((long)nTs * nSens);
        out[i] = in[(q * nTs + t) * nSens + s];
    }
}

// single-bin DFT + peak of sensor series. x(s,n) = in[s*sS + n*sN]; out[s] = (2/nTs) sum_n x exp(-2 pi i bin n/nTs)
__global__ void dft_series(const float *__restrict__ in, long sS, long sN, long nSens, int nTs, int bin,
                           float *__restrict__ outReIm, float *__restrict__ outPeak)
{
    for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < nSens; s += (long)gridDim.x * blockDim.x) {
        double re = 0.0, im = 0.0;
        float pk = -INFINITY;
        for (int n = 0; n < nTs; n++) {
            const float x = in[s * sS + n * sN];
            const int r = (int)(((long)bin * n) % nTs);                 // exact phase index
            double sn, cs;
            sincospi(2.0 * (double)r / (double)nTs, &sn, &cs);
            re += (double)x * cs; im -= (double)x * sn;
            pk = fmaxf(pk, x);
        }
        const double sc = 2.0 / (double)nTs;
        outReIm[2 * s] = (float)(re * sc); outReIm[2 * s + 1] = (float)(im * sc);
        if (outPeak) outPeak[s] = pk;
    }
}

// sources. typeSource 0/1: velocity (after the velocity half-step), 2/3: normal stresses (after the stress half-step)
__global__ void inject_sources(bfd_dev d, int typeSource, const uint32_t *__restrict__ lin, const uint32_t *__restrict__ row,
                               const float *__restrict__ wx, const float *__restrict__ wy, const float *__restrict__ wz,
                               const float *__restrict__ pulseAtStep, long nVox)
{
    for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < nVox; s += (long)gridDim.x * blockDim.x) {
        const long c = lin[s];
        const float val = pulseAtStep[row[s]];
        const float x = wx ? wx[s] : 1.0f;
        if (typeSource >= 2) {
            const float v = val * x;
            float *pxx = d.Sxx + c, *pyy = d.Syy + c;
            if (d.cssRow) {         // compact solid state: a listed cell's Sxx, Syy live in the list; elsewhere nobody reads them (fluid cells keep Szz only) and
                const long e = css_index(d, c);      // the full-volume arrays are not to be written: they may host the compact ones
                pxx = e >= 0 ? d.cSxx + e : nullptr; pyy = e >= 0 ? d.cSyy + e : nullptr;
            }
            if (typeSource == 2) { if (pxx) { *pxx = *pxx + v; *pyy = *pyy + v; } d.Szz[c] = d.Szz[c] + v; }
            else { if (pxx) { *pxx = v; *pyy = v; } d.Szz[c] = v; }
        } else {
            const float y = wy ? wy[s] : 1.0f, z = wz ? wz[s] : 1.0f;
            if (typeSource == 0) { d.Vx[c] = d.Vx[c] + val * x; d.Vy[c] = d.Vy[c] + val * y; d.Vz[c] = d.Vz[c] + val * z; }
            else { d.Vx[c] = val * x; d.Vy[c] = val * y; d.Vz[c] = val * z; }
        }
    }
}

// graph replay: the step counter lives on the device
__global__ void inject_sources_at(bfd_dev d, int typeSource, const uint32_t *__restrict__ lin, const uint32_t *__restrict__ row,
                                  const float *__restrict__ wx, const float *__restrict__ wy, const float *__restrict__ wz,
                                  const float *__restrict__ pulseT, const int *__restrict__ stepDev, int nSources, int lengthSource, long nVox)
{
    const int step = *stepDev;
    if (step >= lengthSource) return;
    const float *pulseAtStep = pulseT + (size_t)step * nSources;
    for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < nVox; s += (long)gridDim.x * blockDim.x) {
        const long c = lin[s];
        const float val = pulseAtStep[row[s]];
        const float x = wx ? wx[s] : 1.0f;
        if (typeSource >= 2) {
            const float v = val * x;
            float *pxx = d.Sxx + c, *pyy = d.Syy + c;
            if (d.cssRow) {         // compact solid state: a listed cell's Sxx, Syy live in the list; elsewhere nobody reads them (fluid cells keep Szz only) and
                const long e = css_index(d, c);      // the full-volume arrays are not to be written: they may host the compact ones
                pxx = e >= 0 ? d.cSxx + e : nullptr; pyy = e >= 0 ? d.cSyy + e : nullptr;
            }
            if (typeSource == 2) { if (pxx) { *pxx = *pxx + v; *pyy = *pyy + v; } d.Szz[c] = d.Szz[c] + v; }
            else { if (pxx) { *pxx = v; *pyy = v; } d.Szz[c] = v; }
        } else {
            const float y = wy ? wy[s] : 1.0f, z = wz ? wz[s] : 1.0f;
            if (typeSource == 0) { d.Vx[c] = d.Vx[c] + val * x; d.Vy[c] = d.Vy[c] + val * y; d.Vz[c] = d.Vz[c] + val * z; }
            else { d.Vx[c] = val * x; d.Vy[c] = val * y; d.Vz[c] = val * z; }
        }
    }
}
__global__ void set_step(int *stepDev, int v) { *stepDev = v; }
__global__ void advance_step(int *stepDev) { *stepDev = *stepDev + 1; }

inline int grid_for(long n, int block = 256) { return (int)std::min<long>((n + block - 1) / block, 256L * 32); }

// inputs changed: a recorded step graph holds stale pointers
static void drop_step_graph(bfd_sim *s)
{
    if (s->stepGraph) { hipGraphExecDestroy(s->stepGraph); s->stepGraph = nullptr; }
    if (s->graphState == 1) s->graphState = 0;
    s->stepDevValid = false;
}

// hipMalloc that gives idle buffers of the placement cache (placement_cache_*) back to the device before it reports that memory ran out
static hipError_t malloc_or_release_cache(void **q, size_t bytes)
{
    hipError_t e = hipMalloc(q, bytes);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
        (void)hipGetLastError();
        if (bfd_placement_cache_release() > 0) e = hipMalloc(q, bytes);
    }
    return e;
}

template <typename T>
int dev_alloc(bfd_sim *s, T **p, size_t count, bool zero = true)
{
    void *q = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    BFD_HIP(malloc_or_release_cache(&q, bytes));
    if (zero) BFD_HIP(hipMemsetAsync(q, 0, bytes, s->stream));
    s->allocs.push_back(q);
    s->devBytes += (int64_t)bytes;
    *p = (T *)q;
    return 0;
}

// frees an array obtained from dev_alloc before the sim is destroyed (inputs that are set again)
template <typename T>
void dev_release(bfd_sim *s, T **p)
{
    if (!*p) return;
    auto it = std::find(s->allocs.begin(), s->allocs.end(), (void *)*p);
    if (it != s->allocs.end()) { hipFree(*it); s->allocs.erase(it); }
    *p = nullptr;
}

inline size_t span_elems(int N1, int N2, int nk, int64_t s1, int64_t s2, int64_t s3)
{
    return (size_t)((N1 - 1) * s1 + (N2 - 1) * s2 + (nk - 1) * s3 + 1);
}

int sel_list(uint32_t mask, int *sel)
{
    int n = 0;
    for (int b = 0; b < BFD_MAP_COUNT; b++) if (mask & (1u << b)) sel[n++] = b;
    return n;
}

hipEvent_t get_event(bfd_sim *s)
{
    hipEvent_t e;
    if (!s->evPool.empty()) { e = s->evPool.back(); s->evPool.pop_back(); return e; }
    if (hipEventCreate(&e) != hipSuccess) return nullptr;      // callers skip the timing pair
    return e;
}

// number of non-zero edge coefficients A (active shear updates) in the sparse shear list
__global__ void count_active_edges(const float *__restrict__ coef, const unsigned *__restrict__ codes, long n, unsigned long long *__restrict__ out)
{
    unsigned c = 0, x = 0;          // out[0] active edges, out[1] edges with explicit coefficients
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        c += (coef[6 * t] != 0.f) + (coef[6 * t + 2] != 0.f) + (coef[6 * t + 4] != 0.f);
        const unsigned w = codes[t];
        x += ((w & 255u) == 255u) + (((w >> 8) & 255u) == 255u) + (((w >> 16) & 255u) == 255u);
    }
    if (c) atomicAdd(out, (unsigned long long)c);
    if (x) atomicAdd(out + 1, (unsigned long long)x);
}


// ---- streamed source table -------------------------------------------------------------------------------------------
// tile t = steps [t*TS, min((t+1)*TS, L)) of the caller's [nSources][L] float64 table -> pinned [steps][nSources] float32
// (the same float64 -> float32 rounding the resident path applies on the device). Row blocks keep reads and writes in cache.
void pack_tile_rows(const double *pulse, int nSources, int L, int t0, int len, float *out, int r0, int r1)
{
    const int RB = 64;
    for (int rb = r0; rb < r1; rb += RB) {
        const int re = std::min(rb + RB, r1);
        for (int q = 0; q < len; q++) {
            float *o = out + (size_t)q * nSources;
            for (int r = rb; r < re; r++) o[r] = (float)pulse[(size_t)r * L + t0 + q];
        }
    }
}
void pack_tile(const double *pulse, int nSources, int L, int TS, int t, float *out)
{
    const int t0 = t * TS, len = std::min(TS, L - t0);
    const int nth = (size_t)nSources * len > (1u << 20) ? 4 : 1;
    if (nth == 1) { pack_tile_rows(pulse, nSources, L, t0, len, out, 0, nSources); return; }
    std::vector<std::thread> th;
    for (int a = 0; a < nth; a++) {
        const int r0 = (int)((long)nSources * a / nth) / 64 * 64, r1 = a + 1 == nth ? nSources : (int)((long)nSources * (a + 1) / nth) / 64 * 64;
        th.emplace_back(pack_tile_rows, pulse, nSources, L, t0, len, out, r0, r1);
    }
    for (auto &x : th) x.join();
}

void drain_pack_jobs(bfd_sim *s)
{
    for (int b = 0; b < 2; b++) if (s->packJob[b].valid()) s->packJob[b].wait();
}

void release_streaming(bfd_sim *s)
{
    drain_pack_jobs(s);
    for (int b = 0; b < 2; b++) {
        if (s->packJob[b].valid()) s->packJob[b].get();
        dev_release(s, &s->tileDev[b]);
        if (s->tilePinned[b]) { hipHostFree(s->tilePinned[b]); s->tilePinned[b] = nullptr; }
        s->tileLoaded[b] = s->tilePacked[b] = -1; s->evTileUsed[b] = false; s->evReadUsed[b][0] = s->evReadUsed[b][1] = false;
    }
    s->pulseHost = nullptr; s->tileSteps = s->nTiles = 0;
}

// the float32 [nSources] row of source values for time step `step`, resident on the device when the kernels of stream st
// run: either a row of the resident table or of the time tile that holds the step (uploaded here when the run enters it)
int pulse_row(bfd_sim *s, int step, hipStream_t st, const float **row)
{
    if (!s->pulseHost) { *row = s->pulseT + (size_t)step * s->nSources; return 0; }
    const int TS = s->tileSteps, t = step / TS, b = t & 1;
    if (s->tileLoaded[b] != t) {
        if (s->tilePacked[b] == t && s->packJob[b].valid()) s->packJob[b].get();
        else {           // first tile, or the run jumped (bfd_reset): pack it here
            if (s->packJob[b].valid()) s->packJob[b].get();
            if (s->evTileUsed[b]) BFD_HIP(hipEventSynchronize(s->evTile[b]));
            pack_tile(s->pulseHost, s->nSources, s->lengthSource, TS, t, s->tilePinned[b]);
            s->tilePacked[b] = t;
        }
        const int len = std::min(TS, s->lengthSource - t * TS);
        // the kernels that read the tile this buffer held two tiles ago may have run on another stream (split half-steps):
        // the copy waits for the last of them
        for (int q = 0; q < 2; q++) if (s->evReadUsed[b][q]) BFD_HIP(hipStreamWaitEvent(st, s->evRead[b][q], 0));      // the readers on the engine's stream and on a caller's side stream
        BFD_HIP(hipMemcpyAsync(s->tileDev[b], s->tilePinned[b], (size_t)len * s->nSources * sizeof(float), hipMemcpyHostToDevice, st));
        BFD_HIP(hipEventRecord(s->evTile[b], st));
        s->evTileUsed[b] = true; s->tileLoaded[b] = t;
        // the next tile is packed beside the GPU's work on this one (the pinned buffer is free once its last upload is done)
        const int nb = b ^ 1, nt = t + 1;
        if (nt < s->nTiles && s->tilePacked[nb] != nt) {
            if (s->packJob[nb].valid()) s->packJob[nb].get();
            s->tilePacked[nb] = nt;
            const bool waitEv = s->evTileUsed[nb];
            hipEvent_t ev = s->evTile[nb];
            const int dev = s->cfg.device;
            const double *ph = s->pulseHost; const int nS = s->nSources, L = s->lengthSource; float *dst = s->tilePinned[nb];
            s->packJob[nb] = std::async(std::launch::async, [=]() {
                if (waitEv) { hipSetDevice(dev); hipEventSynchronize(ev); }
                pack_tile(ph, nS, L, TS, nt, dst);
            });
        }
    } else {
        BFD_HIP(hipStreamWaitEvent(st, s->evTile[b], 0));      // a part launched on another stream than the upload
    }
    *row = s->tileDev[b] + (size_t)(step - t * TS) * s->nSources;
    return 0;
}

// after the kernels of stream st that read the row pulse_row() returned: the tile buffer may be overwritten behind them
void pulse_row_read(bfd_sim *s, int step, hipStream_t st)
{
    if (!s->pulseHost) return;
    // one event per buffer and stream kind: the parts of a split half-step run on two streams at the same time, and a single
    // event recorded by both would keep only the later record
    const int b = (step / s->tileSteps) & 1, q = st == s->stream ? 0 : 1;
    if (s->evRead[b][q] && hipEventRecord(s->evRead[b][q], st) == hipSuccess) s->evReadUsed[b][q] = true;
}

}  // namespace

static int dft_bin(int n, double d, double freq);

void bfd_kmark(bfd_sim *s, int cls, int end, hipStream_t st)
{
    hipEvent_t e = get_event(s);
    if (!e) return;
    if (!end && (s->evK[cls].size() & 1)) { s->evPool.push_back(e); return; }    // unmatched begin: keep pairs intact
    if (end && !(s->evK[cls].size() & 1)) { s->evPool.push_back(e); return; }
    hipEventRecord(e, st);
    s->evK[cls].push_back(e);
}

// ------------------------------------------------------------------------------------------------
extern "C" {

int bfd_abi_version(void) { return BFD_ABI_VERSION; }
const char *bfd_last_error(void) { return g_err.c_str(); }

int bfd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int bfd_device_name(int device, char *buf, int buflen)
{
    hipDeviceProp_t p;
    BFD_HIP(hipGetDeviceProperties(&p, device));
    snprintf(buf, buflen, "%s (%s)", p.name, p.gcnArchName);
    return 0;
}

double bfd_stable_dt(int32_t nMat, const double *matlist, const double *qcorr, double freq,
                     int32_t qfactorCorrection, double h, double alphaCFL)
{
    if (nMat <= 0 || !matlist) { bfd_set_error("bfd_stable_dt: no materials"); return -1.0; }
    HostTables T;
    make_tables(nMat, matlist, qcorr, freq, qfactorCorrection != 0, h, 1.0, T);
    // O(2,4) staggered leapfrog: dt <= (6/7) h / (sqrt(3) cmax)
    return alphaCFL * BFD_STAB * h / (sqrt(3.0) * T.cmax);
}

int bfd_material_tables(int32_t nMat, const double *matlist, const double *qcorr, double freq,
                        int32_t qfactorCorrection, double h, double dt, float *tables7, float *c1k2, double *cmax)
{
    if (nMat <= 0 || !matlist) BFD_FAIL(-1, "bfd_material_tables: no materials");
    HostTables T;
    make_tables(nMat, matlist, qcorr, freq, qfactorCorrection != 0, h, dt, T);
    if (tables7) memcpy(tables7, T.t.data(), T.t.size() * sizeof(float));
    if (c1k2) { c1k2[0] = T.c1; c1k2[1] = T.k2; }
    if (cmax) *cmax = T.cmax;
    return 0;
}

static bool placement_cache_put(int device, size_t bytes, void *p);
static void placement_cache_evict_other_sizes(int device, size_t bytes);
int bfd_create(const bfd_config *cfg, bfd_sim **out)
{
    if (!cfg || !out) BFD_FAIL(-1, "bfd_create: null argument");
    const int P = cfg->NDelta + 1;
    if (cfg->N1 < 2 * P + 4 || cfg->N2 < 2 * P + 4 || cfg->N3 < 2 * P + 4)
        BFD_FAIL(-2, "bfd_create: every dimension must be at least 2*(NDelta+1)+4 voxels");
    if (cfg->k0 < 0 || cfg->nk < 2 || cfg->k0 + cfg->nk > cfg->N3)
        BFD_FAIL(-2, "bfd_create: slab [k0,k0+nk) outside the domain or thinner than 2 planes");
    if (cfg->nMat <= 0 || cfg->nMat > (int)BFD_MAT_MASK) BFD_FAIL(-2, "bfd_create: nMat must be in 1..32767");
    if (cfg->sensorSub <= 0 || cfg->sensorStart < 0 || cfg->nt < 0) BFD_FAIL(-2, "bfd_create: bad sensor sampling / nt");
    if (cfg->typeSource < 0 || cfg->typeSource > 3) BFD_FAIL(-2, "bfd_create: TypeSource must be 0..3");
    if (cfg->kernelVariant < 0 || cfg->kernelVariant > 4) BFD_FAIL(-2, "bfd_create: kernelVariant must be 0..4");
    if (cfg->rmsFirstStep < 0) BFD_FAIL(-2, "bfd_create: rmsFirstStep must be >= 0");
    if (cfg->selRMSorPeak < 0 || cfg->selRMSorPeak > 3) BFD_FAIL(-2, "bfd_create: SelRMSorPeak must be 0..3");
    if (cfg->sensorMode < 0 || cfg->sensorMode > 1) BFD_FAIL(-2, "bfd_create: sensorMode must be 0 or 1");
    if ((long)cfg->N1 * cfg->N2 * (cfg->nk + 4) >= (1L << 31)) BFD_FAIL(-2, "bfd_create: slab exceeds 2^31 voxels (split it into Z-slabs)");
    if (!(cfg->h > 0) || !(cfg->dt > 0) || !(cfg->freq > 0) || !(cfg->reflectionLimit > 0 && cfg->reflectionLimit < 1))
        BFD_FAIL(-2, "bfd_create: h, dt, freq must be > 0 and 0 < reflectionLimit < 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        BFD_FAIL(-3, "bfd_create: no HIP device available (this engine has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) BFD_FAIL(-3, "bfd_create: device ordinal out of range");
    BFD_HIP(hipSetDevice(cfg->device));

    bfd_sim *s = new bfd_sim();
    s->cfg = *cfg;
    s->step = 0; s->devBytes = 0; s->haveMaterials = s->haveMap = false; s->tilesReady = false;
    s->nSrcVox = 0; s->srcLin = s->srcRow = nullptr; s->srcW[0] = s->srcW[1] = s->srcW[2] = nullptr; s->pulseT = nullptr;
    s->nSources = s->lengthSource = 0;
    s->pulseHost = nullptr; s->tileSteps = s->nTiles = 0;
    for (int b = 0; b < 2; b++) { s->tileDev[b] = s->tilePinned[b] = nullptr; s->tileLoaded[b] = s->tilePacked[b] = -1; s->evTile[b] = nullptr; s->evTileUsed[b] = false; for (int q = 0; q < 2; q++) { s->evRead[b][q] = nullptr; s->evReadUsed[b][q] = false; } }
    s->sensEnt = nullptr; s->sensEntValid = false; s->sensIsBox = false; memset(s->sensBox, 0, sizeof s->sensBox);
    s->actBase = nullptr; s->actBytes = 0; s->actReady = false;
    s->nSensors = 0; s->sensLin = nullptr; s->sensOut = nullptr; s->dftAcc = nullptr; s->dftPk = nullptr; s->dftBin = 0;
    s->acc = s->pk = nullptr; s->timing = s->perKernel = false;
    s->tables = nullptr; s->profiles = nullptr; s->cmax = 0;
    memset(s->algBytes, 0, sizeof s->algBytes); s->tiles.ktimer = nullptr;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; BFD_FAIL(-10, "hipStreamCreate failed"); }
    s->ownStream = true;
    if (const char *ev = getenv("BFD_CONCURRENT")) if (atoi(ev) != 0) {     // experiment: solid-run kernels beside the fluid kernel (bfd_tiles::sideStream)
        bfd_tiles &T = s->tiles;
        bool ok = hipEventCreateWithFlags(&T.sideFork, hipEventDisableTiming) == hipSuccess;
        for (int q = 0; q < 2 && ok; q++) ok = hipStreamCreateWithFlags(&T.sideStream[q], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&T.sideJoin[q], hipEventDisableTiming) == hipSuccess;
        if (!ok) { T.sideStream[0] = T.sideStream[1] = nullptr; (void)hipGetLastError(); }
    }
    if (hipEventCreate(&s->evBegin) != hipSuccess) { hipStreamDestroy(s->stream); delete s; BFD_FAIL(-10, "hipEventCreate failed"); }
    if (hipEventCreate(&s->evEnd) != hipSuccess) { hipEventDestroy(s->evBegin); hipStreamDestroy(s->stream); delete s; BFD_FAIL(-10, "hipEventCreate failed"); }

    bfd_dev &d = s->d;
    memset(&d, 0, sizeof d);
    d.N1 = cfg->N1; d.N2 = cfg->N2; d.N3 = cfg->N3; d.k0 = cfg->k0; d.nk = cfg->nk;
    d.ND = cfg->NDelta; d.P = P; d.plane = cfg->N1 * cfg->N2;
    s->nloc = (size_t)d.plane * d.nk;
    s->nalloc = (size_t)d.plane * (d.nk + 4);
    placement_cache_evict_other_sizes(cfg->device, s->nalloc * sizeof(float));

    int rc = 0;
    float **fp[15] = {&d.Vx, &d.Vy, &d.Vz, &d.Sxx, &d.Syy, &d.Szz, &d.Sxy, &d.Sxz, &d.Syz,
                      &d.Rxx, &d.Ryy, &d.Rzz, &d.Rxy, &d.Rxz, &d.Ryz};
    for (int a = 0; a < 15 && !rc; a++) {
        rc = dev_alloc(s, &s->stateBase[a], s->nalloc);
        if (!rc) *fp[a] = s->stateBase[a] + 2 * (size_t)d.plane;
    }
    if (!rc) rc = dev_alloc(s, &s->matBase, s->nalloc);
    if (!rc) d.mat = s->matBase + 2 * (size_t)d.plane;
    if (!rc) rc = dev_alloc(s, &s->clsBase, s->nalloc);
    if (!rc) d.cls = s->clsBase + 2 * (size_t)d.plane;
    s->classesReady = false; s->placementDone = false; s->haloHandedOut = false;
    s->placementMode = 1; s->placementLimit = -1;
    d.VxW = d.Vx; d.VyW = d.Vy; d.VzW = d.Vz; d.SzzW = d.Szz; d.RzzW = d.Rzz;
    // variant 4 (fused fluid time step) needs the old fields to survive the step: second copies of V, Szz, Rzz. A
    // Z-slab keeps the in-place update (its neighbours alias the halo planes once), i.e. behaves like variant 3.
    s->pingpong = cfg->kernelVariant == 4 && cfg->k0 == 0 && cfg->nk == cfg->N3;
    if (s->pingpong) {
        float **wp[5] = {&d.VxW, &d.VyW, &d.VzW, &d.SzzW, &d.RzzW};
        for (int a = 0; a < 5 && !rc; a++) {
            rc = dev_alloc(s, &s->ppBase[a], s->nalloc);
            if (!rc) *wp[a] = s->ppBase[a] + 2 * (size_t)d.plane;
        }
    }
    // CPML memory variables
    const bool zTouch = (d.k0 < P) || (d.k0 + d.nk > d.N3 - P);
    static const int dirOf[18] = {0, 1, 2, 1, 0, 2, 0, 2, 1, 0, 1, 2, 0, 1, 2, 0, 1, 2};
    for (int a = 0; a < 18 && !rc; a++) {
        size_t n = 0;
        if (dirOf[a] == 0) n = (size_t)d.nk * d.N2 * 2 * P;
        else if (dirOf[a] == 1) n = (size_t)d.nk * 2 * P * d.N1;
        else n = zTouch ? (size_t)2 * P * d.plane : 0;
        rc = dev_alloc(s, &d.psi[a], n);
    }
    if (!rc) rc = dev_alloc(s, &s->tables, 7 * (size_t)cfg->nMat);
    if (!rc) rc = dev_alloc(s, &s->profiles, 4 * (size_t)(d.N1 + d.N2 + d.N3));
    s->nSelR = sel_list(cfg->selMapsRMS, s->selR);
    s->nSelS = sel_list(cfg->selMapsSensors, s->selS);
    if (!rc && (cfg->selRMSorPeak & 1) && s->nSelR) rc = dev_alloc(s, &s->acc, (size_t)s->nSelR * s->nloc);
    if (!rc && (cfg->selRMSorPeak & 2) && s->nSelR) rc = dev_alloc(s, &s->pk, (size_t)s->nSelR * s->nloc);
    s->accStart = cfg->rmsFirstStep > 0 ? cfg->rmsFirstStep - 1 : cfg->sensorStart * cfg->sensorSub;
    s->nTs = 0;
    for (int n = 0; n < cfg->nt; n++) if (n % cfg->sensorSub == 0 && n / cfg->sensorSub >= cfg->sensorStart) s->nTs++;
    if (rc) { bfd_destroy(s); return rc; }
    if (hipStreamSynchronize(s->stream) != hipSuccess) { bfd_destroy(s); BFD_FAIL(-10, "bfd_create: sync failed"); }
    *out = s;
    return 0;
}

void bfd_destroy(bfd_sim *s)
{
    if (!s) return;
    hipSetDevice(s->cfg.device);
    hipDeviceSynchronize();
    if (s->stepGraph) hipGraphExecDestroy(s->stepGraph);
    if (s->captureStream) hipStreamDestroy(s->captureStream);
    release_streaming(s);
    for (int b = 0; b < 2; b++) { if (s->evTile[b]) hipEventDestroy(s->evTile[b]); for (int q = 0; q < 2; q++) if (s->evRead[b][q]) hipEventDestroy(s->evRead[b][q]); }
    for (void *p : s->allocs) {
        const bool searched = std::find(s->searched.begin(), s->searched.end(), p) != s->searched.end();
        if (searched && placement_cache_put(s->cfg.device, s->nalloc * sizeof(float), p)) continue;
        hipFree(p);
    }
    for (hipEvent_t e : s->evPool) hipEventDestroy(e);
    for (hipEvent_t e : s->evStress) hipEventDestroy(e);
    for (hipEvent_t e : s->evVelocity) hipEventDestroy(e);
    for (auto &v : s->evK) for (hipEvent_t e : v) hipEventDestroy(e);
    hipEventDestroy(s->evBegin); hipEventDestroy(s->evEnd);
    for (int q = 0; q < 2; q++) { if (s->tiles.sideStream[q]) hipStreamDestroy(s->tiles.sideStream[q]); if (s->tiles.sideJoin[q]) hipEventDestroy(s->tiles.sideJoin[q]); }
    if (s->tiles.sideFork) hipEventDestroy(s->tiles.sideFork);
    if (s->ownStream) hipStreamDestroy(s->stream);
    delete s;
}

int bfd_set_stream(bfd_sim *s, void *hipStream)
{
    if (!s) BFD_FAIL(-1, "null sim");
    BFD_HIP(hipSetDevice(s->cfg.device));
    BFD_HIP(hipStreamSynchronize(s->stream));
    if (s->ownStream) { hipStreamDestroy(s->stream); s->ownStream = false; }
    s->stream = (hipStream_t)hipStream;      // NULL = the device's default (null) stream, which is torch's default too
    return 0;
}

int bfd_use_private_stream(bfd_sim *s)
{
    if (!s) BFD_FAIL(-1, "null sim");
    BFD_HIP(hipSetDevice(s->cfg.device));
    if (s->ownStream) return 0;
    BFD_HIP(hipStreamSynchronize(s->stream));
    BFD_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    s->ownStream = true;
    return 0;
}

int bfd_set_materials(bfd_sim *s, const double *matlist, const double *qcorr)
{
    if (!s || !matlist) BFD_FAIL(-1, "bfd_set_materials: null argument");
    BFD_HIP(hipSetDevice(s->cfg.device));
    const bfd_config &c = s->cfg;
    for (int m = 0; m < c.nMat; m++) {
        const double *r = matlist + 5 * m;
        if (!(r[0] > 0) || !(r[1] > 0) || r[2] < 0 || r[3] < 0 || r[4] < 0)
            BFD_FAIL(-2, "bfd_set_materials: need rho>0, cL>0, cS>=0, alpha>=0 in every row");
        if (qcorr && !(qcorr[m] > 0)) BFD_FAIL(-2, "bfd_set_materials: QCorrection must be > 0");
    }
    HostTables T;
    make_tables(c.nMat, matlist, qcorr, c.freq, c.qfactorCorrection != 0, c.h, c.dt, T);
    const double cfl = T.cmax * c.dt / c.h;
    if (cfl > BFD_STAB / sqrt(3.0) * 1.0000001)
        BFD_FAIL(-4, "bfd_set_materials: DT violates the stability limit dt <= (6/7) h / (sqrt(3) cmax)");
    s->cmax = T.cmax;
    BFD_HIP(hipMemcpyAsync(s->tables, T.t.data(), T.t.size() * sizeof(float), hipMemcpyHostToDevice, s->stream));
    bfd_dev &d = s->d;
    const int n = c.nMat;
    d.AP = s->tables; d.BP = d.AP + n; d.AS2 = d.BP + n; d.BS2 = d.AS2 + n;
    d.invMu = d.BS2 + n; d.tauS = d.invMu + n; d.invRho = d.tauS + n;
    d.c1 = T.c1; d.k2 = T.k2;
    std::vector<float> prof(4 * (size_t)(d.N1 + d.N2 + d.N3));
    float *px = prof.data(), *py = px + 4 * d.N1, *pz = py + 4 * d.N2;
    cpml_axis(d.N1, d.ND, T.cmax, c.h, c.dt, c.freq, c.reflectionLimit, px);
    cpml_axis(d.N2, d.ND, T.cmax, c.h, c.dt, c.freq, c.reflectionLimit, py);
    cpml_axis(d.N3, d.ND, T.cmax, c.h, c.dt, c.freq, c.reflectionLimit, pz);
    BFD_HIP(hipMemcpyAsync(s->profiles, prof.data(), prof.size() * sizeof(float), hipMemcpyHostToDevice, s->stream));
    BFD_HIP(hipStreamSynchronize(s->stream));
    const float *bx = s->profiles, *by = bx + 4 * d.N1, *bz = by + 4 * d.N2;
    d.axI = bx; d.bxI = bx + d.N1; d.axH = bx + 2 * d.N1; d.bxH = bx + 3 * d.N1;
    d.ayI = by; d.byI = by + d.N2; d.ayH = by + 2 * d.N2; d.byH = by + 3 * d.N2;
    d.azI = bz; d.bzI = bz + d.N3; d.azH = bz + 2 * d.N3; d.bzH = bz + 3 * d.N3;
    s->haveMaterials = true; s->tilesReady = false; s->classesReady = false; s->actReady = false; drop_step_graph(s);
    return 0;
}

int bfd_set_material_map(bfd_sim *s, const uint32_t *map, int64_t s1, int64_t s2, int64_t s3,
                         int32_t ghostLow, int32_t ghostHigh)
{
    if (!s || !map) BFD_FAIL(-1, "bfd_set_material_map: null argument");
    if (ghostLow < 0 || ghostLow > 2 || ghostHigh < 0 || ghostHigh > 2) BFD_FAIL(-2, "ghost plane counts must be 0..2");
    BFD_HIP(hipSetDevice(s->cfg.device));
    const bfd_dev &d = s->d;
    // upload the readable span [k=-ghostLow .. nk-1+ghostHigh]
    const uint32_t *base = map - (int64_t)ghostLow * s3;
    const int nkSpan = d.nk + ghostLow + ghostHigh;
    if (s1 < 0 || s2 < 0 || s3 < 0) BFD_FAIL(-2, "negative strides are not supported");
    const size_t span = span_elems(d.N1, d.N2, nkSpan, s1, s2, s3);
    uint32_t *tmp = nullptr;
    BFD_HIP(hipMalloc((void **)&tmp, span * sizeof(uint32_t)));
    hipError_t e = hipMemcpyAsync(tmp, base, span * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream);
    int *flag = nullptr; int hflag = 0;
    if (e == hipSuccess) e = hipMalloc((void **)&flag, sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(flag, 0, sizeof(int), s->stream);
    if (e == hipSuccess) {
        // destination covers local planes -2..nk+1; source plane index = local k + ghostLow, clamped to the span
        hipLaunchKernelGGL((gather_to_xfast<0, uint16_t>), dim3(grid_for((long)s->nalloc)), dim3(256), 0, s->stream,
                           tmp, (long)s1, (long)s2, (long)s3, s->matBase, d.N1, d.N2, d.nk + 4, ghostLow - 2, 0, nkSpan - 1,
                           (uint32_t)s->cfg.nMat, flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    hipFree(tmp); if (flag) hipFree(flag);
    if (e != hipSuccess) BFD_FAIL(-10, std::string("bfd_set_material_map: ") + hipGetErrorString(e));
    if (hflag) BFD_FAIL(-5, "bfd_set_material_map: MaterialMap holds an id >= number of MaterialList rows");
    s->haveMap = true; s->tilesReady = false; s->classesReady = false; s->actReady = false; drop_step_graph(s);
    return 0;
}

int bfd_set_reflector(bfd_sim *s, const uint32_t *mask, int64_t s1, int64_t s2, int64_t s3)
{
    if (!s) BFD_FAIL(-1, "null sim");
    if (!s->haveMap) BFD_FAIL(-6, "bfd_set_reflector: set the material map first");
    BFD_HIP(hipSetDevice(s->cfg.device));
    const bfd_dev &d = s->d;
    uint32_t *tmp = nullptr;
    if (mask) {
        const size_t span = span_elems(d.N1, d.N2, d.nk, s1, s2, s3);
        BFD_HIP(hipMalloc((void **)&tmp, span * sizeof(uint32_t)));
        BFD_HIP(hipMemcpyAsync(tmp, mask, span * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    }
    hipLaunchKernelGGL(or_reflector, dim3(grid_for((long)s->nloc)), dim3(256), 0, s->stream, tmp, (long)s1, (long)s2, (long)s3,
                       s->matBase + 2 * (size_t)d.plane, d.N1, d.N2, d.nk, mask ? 0 : 1);
    BFD_HIP(hipStreamSynchronize(s->stream));
    if (tmp) hipFree(tmp);
    s->tilesReady = false; s->classesReady = false; s->actReady = false; drop_step_graph(s);      // reflector cells end the UNI class of their tiles
    return 0;
}

int bfd_set_sources(bfd_sim *s, int64_t nVox, const uint32_t *localIndex, const uint32_t *row,
                    const float *wx, const float *wy, const float *wz,
                    const double *pulse, int32_t nSources, int32_t lengthSource)
{
    if (!s) BFD_FAIL(-1, "null sim");
    if (nVox < 0 || (nVox > 0 && (!localIndex || !row || !pulse))) BFD_FAIL(-1, "bfd_set_sources: null argument");
    if (nSources < 0 || lengthSource < 0) BFD_FAIL(-2, "bfd_set_sources: bad PulseSource shape");
    BFD_HIP(hipSetDevice(s->cfg.device));
    for (int64_t v = 0; v < nVox; v++) {
        if (localIndex[v] >= s->nloc) BFD_FAIL(-2, "bfd_set_sources: voxel index outside the slab");
        if ((int)row[v] >= nSources) BFD_FAIL(-2, "bfd_set_sources: SourceMap id exceeds PulseSource rows");
    }
    BFD_HIP(hipStreamSynchronize(s->stream));
    dev_release(s, &s->srcLin); dev_release(s, &s->srcRow); dev_release(s, &s->pulseT);
    release_streaming(s);
    for (int a = 0; a < 3; a++) dev_release(s, &s->srcW[a]);
    s->nSrcVox = nVox; s->nSources = nSources; s->lengthSource = lengthSource;
    s->srcLowEnd = 0; s->srcHighBeg = nVox; s->tilesReady = false; s->actReady = false; drop_step_graph(s);
    if (nVox == 0) return 0;
    // keep the source voxels sorted by voxel index: the boundary/interior split of a half-step injects
    // the sources of the first and last z-chunk separately (build_tile_lists)
    std::vector<int64_t> order((size_t)nVox);
    for (int64_t v = 0; v < nVox; v++) order[v] = v;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return localIndex[a] < localIndex[b]; });
    std::vector<uint32_t> hl((size_t)nVox), hr((size_t)nVox);
    for (int64_t v = 0; v < nVox; v++) { hl[v] = localIndex[order[v]]; hr[v] = row[order[v]]; }
    int rc = 0;
    if ((rc = dev_alloc(s, &s->srcLin, nVox, false))) return rc;
    if ((rc = dev_alloc(s, &s->srcRow, nVox, false))) return rc;
    BFD_HIP(hipMemcpy(s->srcLin, hl.data(), nVox * sizeof(uint32_t), hipMemcpyHostToDevice));
    BFD_HIP(hipMemcpy(s->srcRow, hr.data(), nVox * sizeof(uint32_t), hipMemcpyHostToDevice));
    const float *w[3] = {wx, wy, wz};
    std::vector<float> hw((size_t)nVox);
    for (int a = 0; a < 3; a++) {
        s->srcW[a] = nullptr;
        if (w[a]) {
            if ((rc = dev_alloc(s, &s->srcW[a], nVox, false))) return rc;
            for (int64_t v = 0; v < nVox; v++) hw[v] = w[a][order[v]];
            BFD_HIP(hipMemcpy(s->srcW[a], hw.data(), nVox * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    const size_t np = (size_t)nSources * lengthSource;
    // Large tables are streamed: the float64 table stays where the caller built it (it must stay valid until the run is
    // over, as it does inside the solver call of the drop-in) and the device holds two time tiles. BFD_SOURCE_TILE=<steps>
    // forces streaming with that tile length (tests).
    int tile = 0;
    if (const char *ev = getenv("BFD_SOURCE_TILE")) tile = atoi(ev);
    if (tile <= 0 && np * sizeof(float) > ((size_t)1 << 30)) tile = 64;
    if (tile > 0 && lengthSource > 0 && nSources > 0) {
        tile = std::min(tile, (int)lengthSource);
        s->pulseHost = pulse; s->tileSteps = tile; s->nTiles = (lengthSource + tile - 1) / tile;
        for (int b = 0; b < 2; b++) {
            hipError_t e = hipSuccess;
            rc = dev_alloc(s, &s->tileDev[b], (size_t)tile * nSources, false);
            if (!rc) e = hipHostMalloc((void **)&s->tilePinned[b], (size_t)tile * nSources * sizeof(float), hipHostMallocDefault);
            if (!rc && e == hipSuccess && !s->evTile[b]) e = hipEventCreateWithFlags(&s->evTile[b], hipEventDisableTiming);
            for (int q = 0; q < 2; q++) if (!rc && e == hipSuccess && !s->evRead[b][q]) e = hipEventCreateWithFlags(&s->evRead[b][q], hipEventDisableTiming);
            if (rc || e != hipSuccess) {        // nothing half-built stays behind: the sim is back to "no sources"
                const std::string why = rc ? std::string(bfd_last_error()) : std::string("bfd_set_sources: ") + hipGetErrorString(e);
                release_streaming(s);
                s->nSrcVox = 0; s->srcHighBeg = 0;
                bfd_set_error(why);
                return rc ? rc : -10;
            }
        }
        s->graphState = -1;        // the recorded step graph indexes a resident table
        return 0;
    }
    if ((rc = dev_alloc(s, &s->pulseT, np, false))) return rc;
    double *tmp = nullptr;
    BFD_HIP(hipMalloc((void **)&tmp, std::max<size_t>(np, 1) * sizeof(double)));
    BFD_HIP(hipMemcpyAsync(tmp, pulse, np * sizeof(double), hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(transpose_pulse, dim3(grid_for((long)np)), dim3(256), 0, s->stream, tmp, s->pulseT, nSources, lengthSource);
    BFD_HIP(hipStreamSynchronize(s->stream));
    hipFree(tmp);
    return 0;
}

int bfd_set_sensor_map(bfd_sim *s, const uint32_t *map, int64_t s1, int64_t s2, int64_t s3, int64_t *nSensors)
{
    if (!s || !map) BFD_FAIL(-1, "bfd_set_sensor_map: null argument");
    BFD_HIP(hipSetDevice(s->cfg.device));
    const bfd_dev &d = s->d;
    const size_t span = span_elems(d.N1, d.N2, d.nk, s1, s2, s3);
    uint32_t *tmp = nullptr; uint8_t *flags = nullptr; uint32_t *sel = nullptr; int *dcount = nullptr; void *work = nullptr;
    hipError_t e = hipMalloc((void **)&tmp, span * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&flags, s->nloc);
    if (e == hipSuccess) e = hipMalloc((void **)&sel, s->nloc * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&dcount, sizeof(int));
    if (e == hipSuccess) e = hipMemcpyAsync(tmp, map, span * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream);
    int count = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL((gather_to_xfast<1, uint8_t>), dim3(grid_for((long)s->nloc)), dim3(256), 0, s->stream,
                           tmp, (long)s1, (long)s2, (long)s3, flags, d.N1, d.N2, d.nk, 0, 0, d.nk - 1, 0u, (int *)nullptr);
        size_t wbytes = 0;
        hipcub::CountingInputIterator<uint32_t> ids(0);
        e = hipcub::DeviceSelect::Flagged(nullptr, wbytes, ids, flags, sel, dcount, (int)s->nloc, s->stream);
        if (e == hipSuccess) e = hipMalloc(&work, std::max<size_t>(wbytes, 1));
        if (e == hipSuccess) e = hipcub::DeviceSelect::Flagged(work, wbytes, ids, flags, sel, dcount, (int)s->nloc, s->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&count, dcount, sizeof(int), hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    }
    int rc = 0;
    if (e == hipSuccess) {
        s->nSensors = count;
        dev_release(s, &s->sensLin); dev_release(s, &s->sensOut); dev_release(s, &s->dftAcc); dev_release(s, &s->dftPk);
        dev_release(s, &s->sensEnt); s->sensEntValid = false;
        rc = dev_alloc(s, &s->sensLin, (size_t)count, false);
        if (!rc && count) e = hipMemcpyAsync(s->sensLin, sel, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream);
        if (!rc && s->nSelS && s->nTs > 0) {
            if (s->cfg.sensorMode == 0) rc = dev_alloc(s, &s->sensOut, (size_t)s->nSelS * s->nTs * (size_t)count);
            else {
                rc = dev_alloc(s, &s->dftAcc, 2 * (size_t)s->nSelS * (size_t)count);
                if (!rc) rc = dev_alloc(s, &s->dftPk, (size_t)s->nSelS * (size_t)std::max(count, 1), false);
                if (!rc && count > 0) hipLaunchKernelGGL(fill_float, dim3(grid_for((long)s->nSelS * count)), dim3(256), 0, s->stream, s->dftPk, (long)s->nSelS * count, -INFINITY);
                s->dftBin = dft_bin(s->nTs, s->cfg.dt * s->cfg.sensorSub, s->cfg.freq);
            }
        }
        // a dense box of voxels? (count == volume of the bounding box: every voxel of the box is a sensor); BFD_SENSOR_BOX=0 keeps the index list in use
        s->sensIsBox = false;
        bool tryBox = count > 0;
        if (const char *ev = getenv("BFD_SENSOR_BOX")) tryBox = tryBox && atoi(ev) != 0;
        if (!rc && e == hipSuccess && tryBox) {
            unsigned init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u}, mm[6];
            unsigned *dmm = (unsigned *)dcount;          // reuse: 4 bytes are not enough
            unsigned *dbox = nullptr;
            e = hipMalloc((void **)&dbox, sizeof init);
            if (e == hipSuccess) e = hipMemcpyAsync(dbox, init, sizeof init, hipMemcpyHostToDevice, s->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(sensor_bounds, dim3(std::min(grid_for((long)count), 1024)), dim3(256), 0, s->stream, d, s->sensLin, (long)count, dbox);
                e = hipMemcpyAsync(mm, dbox, sizeof mm, hipMemcpyDeviceToHost, s->stream);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
            if (dbox) hipFree(dbox);
            (void)dmm;
            if (e == hipSuccess) {
                const long bx = (long)mm[3] - mm[0] + 1, by = (long)mm[4] - mm[1] + 1, bz = (long)mm[5] - mm[2] + 1;
                if (bx * by * bz == (long)count) {
                    s->sensIsBox = true;
                    s->sensBox[0] = (int)bx; s->sensBox[1] = (int)by; s->sensBox[2] = (int)mm[0]; s->sensBox[3] = (int)mm[1]; s->sensBox[4] = (int)mm[2];
                }
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    }
    hipFree(tmp); hipFree(flags); hipFree(sel); hipFree(dcount); if (work) hipFree(work);
    if (e != hipSuccess) BFD_FAIL(-10, std::string("bfd_set_sensor_map: ") + hipGetErrorString(e));
    if (rc) return rc;
    if (nSensors) *nSensors = s->nSensors;
    return 0;
}

// Build the run lists of the tiled kernels (bfd_kernels_v2.hip). Sub-tiles of 64 x 8 x 8 cells are
// classified on the device; consecutive sub-tiles of one (bx,by) column with identical class merge into
// runs that never cross a 32-plane chunk boundary. Variant 2: every sub-tile counts as solid (dense kernels).
#ifndef BFD_SOLID_MERGED_DEFAULT
#define BFD_SOLID_MERGED_DEFAULT 0
#endif
// where the ten compact arrays live (bfd_tiles::cssHosted); called when the list is built and again whenever the state buffers change hands
// (choose_placement exchanges them at step 0, when everything is still zero)
static void bind_compact_views(bfd_sim *s)
{
    bfd_dev &d = s->d;
    if (!d.cssRow) return;
    float **cp[10] = {&d.cSxx, &d.cSyy, &d.cSxy, &d.cSxz, &d.cSyz, &d.cRxx, &d.cRyy, &d.cRxy, &d.cRxz, &d.cRyz};
    static const int host[10] = {3, 4, 6, 7, 8, 9, 10, 12, 13, 14};      // Sxx Syy Sxy Sxz Syz Rxx Ryy Rxy Rxz Ryz among the 15 state arrays
    for (int a = 0; a < 10; a++)
        *cp[a] = s->tiles.cssHosted ? s->stateBase[host[a]] + 4 * (size_t)d.plane : s->tiles.css + (size_t)a * s->tiles.cssCap;
}
static int build_tile_lists(bfd_sim *s)
{
    int tx, ty, nsub; bfd_tile_grid(s->d, &tx, &ty, &nsub);
    const int n = tx * ty * nsub;
    // a list rebuilt in the middle of a run (inputs set again at step > 0): the shear memory variables travel through the
    // full-volume arrays
    const bool carryShearMemory = s->step > 0 && s->tilesReady == false && s->tiles.shearR && s->tiles.nShear > 0;
    const bool hadList = s->tiles.shearCells != nullptr, hadListR = s->tiles.shearR != nullptr, wasMerged = s->tiles.merged;
    if (carryShearMemory) { bfd_launch_scatter_shear_memory(s->d, s->stream, &s->tiles); BFD_HIP(hipStreamSynchronize(s->stream)); }
    // the same for a compact solid state: its ten arrays are set aside with their list (the full-volume buffers may host the compact arrays
    // themselves) and re-entered into the new list through one full-volume temporary, array by array, once that list exists
    const bool carryCompact = s->step > 0 && s->d.cssRow && s->tiles.nShear > 0;
    unsigned *oldCells = nullptr; float *oldComp = nullptr; const long oldN = s->tiles.nShear;
    struct FreeOnExit { float **p; ~FreeOnExit() { if (*p) hipFree(*p); } } freeOldComp{&oldComp};       // also on the error returns below
    if (carryCompact) {
        float *src[10] = {s->d.cSxx, s->d.cSyy, s->d.cSxy, s->d.cSxz, s->d.cSyz, s->d.cRxx, s->d.cRyy, s->d.cRxy, s->d.cRxz, s->d.cRyz};
        BFD_HIP(hipMalloc((void **)&oldComp, 10 * (size_t)oldN * sizeof(float)));
        for (int a = 0; a < 10; a++) BFD_HIP(hipMemcpyAsync(oldComp + (size_t)a * oldN, src[a], (size_t)oldN * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
        BFD_HIP(hipStreamSynchronize(s->stream));
        oldCells = s->tiles.shearCells; s->tiles.shearCells = nullptr;       // released below, after the new list has taken the values over
    }
    s->sensEntValid = false;
    s->d.cssRow = nullptr; s->d.cSxx = s->d.cSyy = s->d.cSxy = s->d.cSxz = s->d.cSyz = s->d.cRxx = s->d.cRyy = s->d.cRxy = s->d.cRxz = s->d.cRyz = nullptr;
    dev_release(s, &s->tiles.cssRow); dev_release(s, &s->tiles.css); s->tiles.cssCap = 0; s->tiles.cssHosted = false;
    dev_release(s, &s->tiles.runsAll); s->tiles.nAll = s->tiles.nAllB = 0;
    dev_release(s, &s->tiles.runs); dev_release(s, &s->tiles.xmap); dev_release(s, &s->tiles.shearCells); dev_release(s, &s->tiles.shearCoef); dev_release(s, &s->tiles.shearR);    // lists of an earlier build
    dev_release(s, &s->tiles.shearCodes); dev_release(s, &s->tiles.shearTab);
    const int SUB = bfd_tile_subz();
    // longest run one workgroup marches: 16 planes; 8 on small grids so that the launch still has a few thousand
    // workgroups (measured: 256^3 49 -> 58, 128^3 29 -> 46 Gvoxel-steps/s). 32 was best at 512^3 while the z-chunks of
    // a column were consecutive in the list; under the banded order 16 is 2.5 % faster than 32 and 3 % faster than 8.
    const int t32 = tx * ty * ((s->d.nk + 31) / 32);
    // round 6: the switch point moved from 1500 to 3000 columns (320^3: 8 planes +4 % in water, +7 % with bone -- velocity_solid wants the workgroups; 352^3: +1.5 %;
    // 384^3: 16 planes +4 %; profiles/r6/pml_flavour_cost_and_run_length_mid_size.txt)
    s->zchunk = t32 >= 3000 ? 16 : 8;
    if (const char *ev = getenv("BFD_ZRUN")) { const int z = atoi(ev); if (z >= SUB && z % SUB == 0) s->zchunk = z; }   // tuning experiments
    if (s->zchunk > bfd_tile_zchunk()) s->zchunk = bfd_tile_zchunk();
    const int perChunk = s->zchunk / SUB;
    const int nChunks = (nsub + perChunk - 1) / perChunk;
    std::vector<int> flags(n, 1), mats(n, 0);
    if (s->cfg.kernelVariant != 2) {
        int *dflags = nullptr, *dmats = nullptr;
        BFD_HIP(hipMalloc((void **)&dflags, n * sizeof(int)));
        hipError_t e = hipMalloc((void **)&dmats, n * sizeof(int));
        if (e == hipSuccess) {
            bfd_launch_classify(s->d, s->stream, dflags, dmats);
            e = hipMemcpyAsync(flags.data(), dflags, n * sizeof(int), hipMemcpyDeviceToHost, s->stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(mats.data(), dmats, n * sizeof(int), hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        hipFree(dflags); if (dmats) hipFree(dmats);
        if (e != hipSuccess) BFD_FAIL(-10, std::string("classify tiles: ") + hipGetErrorString(e));
    }
    bfd_tiles &T = s->tiles;
    T.nMat = s->cfg.nMat;
    // solid runs: normal and shear stresses in one kernel (stress_solid_merged), the sparse list keeps the cells with an edge between
    // different solids. BFD_SOLID_MERGED=0 selects the two-kernel form (stress_solid + stress_shear_sparse over every solid cell).
    T.merged = BFD_SOLID_MERGED_DEFAULT != 0;
    if (const char *ev = getenv("BFD_SOLID_MERGED")) T.merged = atoi(ev) != 0;
    if (s->step > 0 && hadList) T.merged = hadListR ? false : wasMerged;       // a list rebuilt in the middle of a run keeps the form it started with (where its shear memory variables live)
    T.nFluid = T.nFluidB = T.nSolid = T.nSolidB = T.nSolidBP = T.nSolidIP = T.nFused = T.nLossless = T.nLossy = T.nSolidSub = T.nUni = T.nPml = T.nLean = T.nFusedSub = 0;
    s->d.tilesX = tx; s->d.tilesY = ty;
    // Every fluid sub-tile is LEAN (bit4): fluid cells keep a single copy of their identical normal stresses, whatever
    // tile they sit in and whatever reads them (bfd_dev::cls)
    if (s->cfg.kernelVariant != 2) for (int id = 0; id < n; id++) if (!(flags[id] & 1)) flags[id] |= 16;
    // "boundary" sub-tiles hold the 2 first / 2 last planes of the slab (what a Z-neighbour reads): they form the
    // small part 1 of a split half-step; everything else is part 2. lowPlanes / hiStart delimit them in planes.
    const int nkl = s->d.nk;
    const int lowPlanes = std::min(SUB, nkl);
    const int hiStart = std::max(((nkl - 2) / SUB) * SUB, lowPlanes);
    auto subBnd = [&](int q) { return q * SUB < lowPlanes || std::min((q + 1) * SUB, nkl) > hiStart; };
    std::vector<int4> lists[5];      // fluid boundary, fluid interior, solid boundary, solid interior, fused fluid
    std::vector<int4> listsAll[2];   // every run of the two-kernel path in list order: boundary, interior (bfd_tiles::runsAll)
    std::vector<char> taken((size_t)n, 0);
    // Runs of the fused time step (variant 4, bfd_kernels_fused.hip): 64 x 24 cells = three tiles of this grid in y, a z-run of
    // 2 .. fusedSub sub-tiles. A sub-tile qualifies (bit5) when it is fluid, has nothing of the absorbing layer or the domain
    // edge within 2 cells (bit6), is not a boundary sub-tile of the slab, and the sources are of velocity type. Per column and
    // z-chunk the rows are scanned upwards: where three consecutive tile rows qualify over a z-stretch they form a run and the
    // scan moves on by three rows. A run is UNI when every sub-tile is (one material in the grown regions) and lossy when a
    // cell of a grown region relaxes (bit7); stretches are cut where that class changes (a single sub-tile joins its neighbour).
    int fusedSub = 32 / SUB;
    if (const char *ev = getenv("BFD_FUSED_ZRUN")) { const int z = atoi(ev); if (z >= 2 * SUB && z % SUB == 0 && z <= 0x7000) fusedSub = z / SUB; }
    struct FusedRun { int bx, by, q0, q1, cls, mat; };
    std::vector<FusedRun> fruns;
    if (s->pingpong && s->cfg.typeSource < 2) {
        const int FRW = bfd_fused_rows() / 8;
        const size_t layer = (size_t)tx * ty;
        for (int q = 0; q < nsub; q++)
            for (size_t txy = 0; txy < layer; txy++) { int &f = flags[(size_t)q * layer + txy]; if (!(f & 1) && !(f & 64) && !(f & 256) && !subBnd(q)) f |= 32; }
        for (int bx = 0; bx < tx; bx++)
            for (int qc = 0; qc < nsub; qc += fusedSub) {
                const int qe = std::min(qc + fusedSub, nsub), L = qe - qc;
                int by = 0;
                while (by + FRW <= ty) {
                    // class of the three rows at every q of the chunk: -1 = not available, else bit0 UNI, bit1 lossy
                    std::vector<int> cls(L, -1);
                    for (int q = qc; q < qe; q++) {
                        bool ok = true; int uni = 1, lossy = 0;
                        const int m0 = mats[(size_t)q * layer + (size_t)by * tx + bx];
                        for (int r = 0; r < FRW; r++) {
                            const size_t id = (size_t)q * layer + (size_t)(by + r) * tx + bx;
                            const int f = flags[id];
                            if (!(f & 32) || taken[id]) ok = false;
                            if (!(f & 4) || mats[id] != m0) uni = 0;
                            if (f & 128) lossy = 1;
                        }
                        if (!uni && s->cfg.nMat > bfd_fused_max_materials()) ok = false;
                        if (ok) cls[q - qc] = uni | (lossy << 1);
                    }
                    bool any = false;
                    int a = 0;
                    while (a < L) {
                        if (cls[a] < 0) { a++; continue; }
                        int b = a; while (b < L && cls[b] >= 0) b++;        // stretch [a, b)
                        if (b - a >= 2) {
                            any = true;
                            // segments of equal class (start, end, class); a segment of one sub-tile joins a neighbour
                            std::vector<std::array<int, 3>> seg;
                            for (int q = a; q < b; q++) {
                                if (seg.empty() || seg.back()[2] != cls[q]) seg.push_back({q, q + 1, cls[q]});
                                else seg.back()[1] = q + 1;
                            }
                            for (size_t u = 0; u < seg.size() && seg.size() > 1;) {
                                if (seg[u][1] - seg[u][0] >= 2) { u++; continue; }
                                const size_t v = u > 0 ? u - 1 : u + 1;      // UNI only if both are, lossy if either is
                                seg[v][0] = std::min(seg[v][0], seg[u][0]); seg[v][1] = std::max(seg[v][1], seg[u][1]);
                                seg[v][2] = (seg[v][2] & seg[u][2] & 1) | ((seg[v][2] | seg[u][2]) & 2);
                                seg.erase(seg.begin() + u);
                                u = 0;
                            }
                            for (const auto &rq : seg) {
                                fruns.push_back({bx, by, qc + rq[0], qc + rq[1], rq[2], mats[(size_t)(qc + rq[0]) * layer + (size_t)by * tx + bx]});
                                for (int q = qc + rq[0]; q < qc + rq[1]; q++)
                                    for (int r = 0; r < FRW; r++) { taken[(size_t)q * layer + (size_t)(by + r) * tx + bx] = 1; T.nFusedSub++; }
                            }
                        }
                        a = b;
                    }
                    by += any ? FRW : 1;
                }
            }
        // list order like the other runs: eight y-bands, inside a band z-chunk slowest, then row, bx fastest
        auto key = [&](const FusedRun &r) { const long band = (long)r.by * 8 / ty; return ((band * 4096 + r.q0 / fusedSub) * 4096 + r.by) * 4096 + r.bx; };
        std::stable_sort(fruns.begin(), fruns.end(), [&](const FusedRun &x, const FusedRun &y) { return key(x) < key(y); });
        for (const auto &r : fruns) {
            int4 run; run.x = r.by * tx + r.bx; run.y = (r.q0 * SUB) | (std::min(r.q1 * SUB, s->d.nk) << 16);
            run.z = 32 | 16 | ((r.cls & 2) ? 2 : 0) | ((r.cls & 1) ? 4 : 0); run.w = r.mat;
            lists[4].push_back(run);
        }
    }
    // List order = what is in flight together. The launch gives XCD e the e-th contiguous eighth of the list
    // (remap_block) and an XCD keeps ~100 workgroups in flight, all marching in z at the same pace; a halo line
    // (128 B for 2 or 3 floats of a neighbour tile's row) is an L2 hit only if that neighbour is in flight on the
    // same XCD. Default (2): eight y-bands, inside a band z-chunk slowest, then by, bx fastest -> x neighbours are
    // 1 apart, y neighbours tilesX apart. Measured at 512^3 (rocprofv3 FETCH/WRITE_SIZE, water): stress traffic
    // 3.55 -> 3.10 GB and 79 -> 88 Gvoxel-steps/s against (0) column order (z-chunks of a column consecutive);
    // (1) z-chunk slowest over the whole plane moves as few bytes but piles the absorbing-layer and lossy
    // z-levels onto single XCDs (C3: 75 -> 72). BFD_RUN_ORDER=0/1 select the other orders for experiments.
    std::vector<std::pair<int, int>> seq;       // (column, z-chunk) in list order
    {
        const char *ev = getenv("BFD_RUN_ORDER");
        const int mode = ev ? atoi(ev) : 2;
        if (mode == 1) {            // z-chunk slowest
            for (int c = 0; c < nChunks; c++) for (int txy = 0; txy < tx * ty; txy++) seq.push_back({txy, c});
        } else if (mode == 0) {     // column order
            for (int txy = 0; txy < tx * ty; txy++) for (int c = 0; c < nChunks; c++) seq.push_back({txy, c});
        } else {                    // 8 y-bands (one per XCD part), inside a band z-chunk slowest
            // mode 3 (experiment): the two z-chunks of the absorbing layer first, the interior chunks after them -- the cheapest
            // workgroups at the end of every XCD's part of the list
            std::vector<int> corder;
            if (mode == 3 && nChunks > 2) { corder.push_back(0); corder.push_back(nChunks - 1); for (int c = 1; c + 1 < nChunks; c++) corder.push_back(c); }
            else for (int c = 0; c < nChunks; c++) corder.push_back(c);
            for (int e = 0; e < 8; e++) {
                const int y0 = (int)((long)ty * e / 8), y1 = (int)((long)ty * (e + 1) / 8);
                for (int c : corder) for (int by = y0; by < y1; by++) for (int bx = 0; bx < tx; bx++) seq.push_back({by * tx + bx, c});
            }
        }
    }
    for (const auto &pc : seq) {
        const int txy = pc.first, c = pc.second;
            const int sb = c * perChunk, se0 = std::min(sb + perChunk, nsub);
            int q = sb;
            while (q < se0) {
                if (taken[(size_t)q * tx * ty + txy]) { q++; continue; }
                const int f = flags[(size_t)q * tx * ty + txy], m = mats[(size_t)q * tx * ty + txy];
                const bool solid = f & 1;
                const bool bnd = subBnd(q);
                const int se = se0;
                int r = q + 1, pmlAny = f & 8;
                while (r < se) {
                    const int f2 = flags[(size_t)r * tx * ty + txy], m2 = mats[(size_t)r * tx * ty + txy];
                    if (subBnd(r) != bnd || taken[(size_t)r * tx * ty + txy]) break;
                    if (solid ? !(f2 & 1) : (((f2 ^ f) & ~(32 | 128 | 256)) != 0 || ((f & 4) && m2 != m))) break;
                    pmlAny |= f2 & 8;
                    r++;
                }
                const int kbeg = q * SUB, kend = std::min(r * SUB, s->d.nk);
                // solid runs: bit0 + bit3 (a sub-tile of the run touches the absorbing layer)
                int4 run; run.x = txy; run.y = kbeg | (kend << 16); run.z = solid ? (1 | pmlAny) : (f & ~(32 | 128 | 256)); run.w = m;
                lists[(solid ? 2 : 0) + (bnd ? 0 : 1)].push_back(run);
                listsAll[bnd ? 0 : 1].push_back(run);
                for (int u = q; u < r; u++) {
                    taken[(size_t)u * tx * ty + txy] = 1;
                    if (solid) T.nSolidSub++;
                    else { if (f & 2) T.nLossy++; else T.nLossless++; if (f & 4) T.nUni++; if (f & 8) T.nPml++; if (f & 16) T.nLean++; }
                }
                q = r;
            }
        }
    // solid runs that touch the absorbing layer go to the two ends of the solid list: [boundary: PML | plain][interior: plain | PML]
    {
        auto isPml = [](const int4 &r) { return (r.z & 8) != 0; };
        auto mid = std::stable_partition(lists[2].begin(), lists[2].end(), isPml);
        T.nSolidBP = (int)(mid - lists[2].begin());
        auto mid2 = std::stable_partition(lists[3].begin(), lists[3].end(), [&](const int4 &r) { return !isPml(r); });
        T.nSolidIP = (int)(lists[3].end() - mid2);
    }
    T.nFluidB = (int)lists[0].size(); T.nFluid = T.nFluidB + (int)lists[1].size();
    T.nSolidB = (int)lists[2].size(); T.nSolid = T.nSolidB + (int)lists[3].size();
    T.nFused = (int)lists[4].size();
    std::vector<int4> all;
    for (int a = 0; a < 5; a++) all.insert(all.end(), lists[a].begin(), lists[a].end());
    int rc = dev_alloc(s, &s->tiles.runs, all.size(), false);
    if (rc) return rc;
    BFD_HIP(hipMemcpy(s->tiles.runs, all.data(), all.size() * sizeof(int4), hipMemcpyHostToDevice));
    // Cost-balanced block -> run maps (experiment, BFD_XCD_BALANCE=1; default off). A launch's blocks go to the 8 XCDs round-robin and every
    // XCD works through its own blocks at its own pace (-DBFD_EXP_XCD_CLOCK build: block b always runs on XCD (x0 + b) mod 8). With equal
    // COUNTS per XCD (remap_block) the XCD that holds the short boundary runs is idle for the last fifth of every fluid launch and the bands
    // with more tissue finish last. Here the contiguous parts of the list are cut by estimated cost instead (planes + prologue, weighted by
    // the bytes per cell of the run's class), the launch gets 8 x (longest part) blocks and a block beyond its part returns at once.
    // Measured: the ends of the XCDs move together (spread 19 % -> 13 % of a launch) and the step time does not -- C3 +1.2 %, shear medium
    // -0.4 %, other weightings +-2 % either way: an XCD that runs dry leaves its share of the memory system to the others.
    // profiles/r4/xcd_balance.txt.
    {
        bool on = false;
        if (const char *ev = getenv("BFD_XCD_BALANCE")) on = atoi(ev) != 0 && s->cfg.kernelVariant != 2 && s->cfg.kernelVariant != 1;
        s->tiles.xmap = nullptr;
        memset(s->tiles.xmapH, 0, sizeof s->tiles.xmapH);
        if (on) {
            double wPml = 0.25, wLossy = 8, wMulti = 2, wRun = 2.0;
            if (const char *ev = getenv("BFD_XCD_WEIGHTS")) sscanf(ev, "%lf,%lf,%lf,%lf", &wPml, &wLossy, &wMulti, &wRun);
            // cost of run r for kernel class c: 0 fluid stress, 1 fluid velocity, 2 solid stress, 3 solid velocity
            auto cost = [&](const int4 &r, int c) {
                const double planes = (double)((r.y >> 16) - (r.y & 0xFFFF)) + wRun;
                const int f = r.z;
                double w;
                if (c == 0) w = 20 + ((f & 2) ? wLossy : 0) + ((f & 4) ? 0 : wMulti);
                else if (c == 1) w = 36 + ((f & 4) ? 0 : wMulti);
                else w = 40;
                if (f & 8) w *= 1.0 + wPml;
                return planes * w;
            };
            auto make = [&](int m, size_t a0, size_t a1, int c) {
                int *seg = s->tiles.xmapH[m];
                const size_t n = a1 > a0 ? a1 - a0 : 0;
                std::vector<double> cum(n + 1, 0.0);
                for (size_t i = 0; i < n; i++) cum[i + 1] = cum[i] + cost(all[a0 + i], c);
                int maxcnt = 0;
                seg[0] = 0;
                for (int x = 1; x <= 8; x++) {
                    const double target = cum[n] * x / 8.0;
                    size_t j = std::lower_bound(cum.begin(), cum.end(), target) - cum.begin();
                    if (j > n || x == 8) j = n;
                    if ((int)j < seg[x - 1]) j = seg[x - 1];
                    seg[x] = (int)j;
                    maxcnt = std::max(maxcnt, seg[x] - seg[x - 1]);
                }
                seg[9] = std::max(maxcnt, 1);
            };
            const size_t F = T.nFluid, FB = T.nFluidB, S0 = F, SB = T.nSolidB, SN = T.nSolid;
            for (int c = 0; c < 2; c++) {            // fluid stress (c = 0), fluid velocity (c = 1): parts 0, 1, 2
                const int m = c == 0 ? BFD_XM_SF : BFD_XM_VF;
                make(m + 0, 0, F, c); make(m + 1, 0, FB, c); make(m + 2, FB, F, c);
            }
            make(BFD_XM_SS + 0, S0, S0 + SN, 2); make(BFD_XM_SS + 1, S0, S0 + SB, 2); make(BFD_XM_SS + 2, S0 + SB, S0 + SN, 2);
            const size_t bp = T.nSolidBP, ip = T.nSolidIP;          // solid list = [boundary: PML | plain][interior: plain | PML]
            make(BFD_XM_VS + 0, S0 + bp, S0 + SN - ip, 3); make(BFD_XM_VS + 1, S0 + bp, S0 + SB, 3); make(BFD_XM_VS + 2, S0 + SB, S0 + SN - ip, 3);
            make(BFD_XM_VSP_LO, S0, S0 + bp, 3); make(BFD_XM_VSP_HI, S0 + SN - ip, S0 + SN, 3);
            make(BFD_XM_FUSED, S0 + SN, S0 + SN + T.nFused, 0);
            if (getenv("BFD_XCD_VERBOSE"))
                for (int m = 0; m < BFD_XMAP_COUNT; m++) {
                    const int *g = s->tiles.xmapH[m];
                    fprintf(stderr, "xcd map %2d: runs %d, parts", m, g[8]);
                    for (int x = 0; x < 8; x++) fprintf(stderr, " %d", g[x + 1] - g[x]);
                    fprintf(stderr, " (longest %d)\n", g[9]);
                }
            rc = dev_alloc(s, &s->tiles.xmap, (size_t)BFD_XMAP_COUNT * 10, false);
            if (rc) return rc;
            BFD_HIP(hipMemcpy(s->tiles.xmap, s->tiles.xmapH, sizeof s->tiles.xmapH, hipMemcpyHostToDevice));
        }
    }
    s->tiles.shearCells = nullptr; s->tiles.shearCoef = nullptr; s->tiles.shearR = nullptr; s->tiles.shearCodes = nullptr; s->tiles.shearTab = nullptr;
    s->tiles.nShearExplicit = 0;
    s->tiles.nShear = s->tiles.shearLowEnd = s->tiles.shearHighBeg = 0;
    if (T.nSolid && s->cfg.kernelVariant != 2) {     // variant 2 stays monolithic and fully dense
        // sparse shear list: cells with a solid centre, ascending index, + their edge coefficients
        unsigned char *flag = nullptr; unsigned *sel = nullptr; int *dcount = nullptr; void *work = nullptr;
        hipError_t e = hipMalloc((void **)&flag, s->nloc);
        if (e == hipSuccess) e = hipMalloc((void **)&sel, s->nloc * sizeof(unsigned));
        if (e == hipSuccess) e = hipMalloc((void **)&dcount, sizeof(int));
        int count = 0;
        if (e == hipSuccess) {
            bfd_launch_mark_solid(s->d, s->stream, flag, (long)s->nloc, T.merged);
            size_t wbytes = 0;
            hipcub::CountingInputIterator<unsigned> ids(0);
            e = hipcub::DeviceSelect::Flagged(nullptr, wbytes, ids, flag, sel, dcount, (int)s->nloc, s->stream);
            if (e == hipSuccess) e = hipMalloc(&work, std::max<size_t>(wbytes, 1));
            if (e == hipSuccess) e = hipcub::DeviceSelect::Flagged(work, wbytes, ids, flag, sel, dcount, (int)s->nloc, s->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&count, dcount, sizeof(int), hipMemcpyDeviceToHost, s->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        }
        std::vector<unsigned> hostCells((size_t)count);
        if (e == hipSuccess && count) e = hipMemcpy(hostCells.data(), sel, (size_t)count * sizeof(unsigned), hipMemcpyDeviceToHost);
        // list order (bfd_kernels_v2.hip, shear_order_keys): by z-chunk and band of 8 rows, so that the z neighbours a cell gathers were
        // touched one band-plane earlier instead of one whole plane of the shell; BFD_SHEAR_ORDER=0 keeps the ascending index, 1 = by tiles
        int orderMode = 2;
        if (const char *ev = getenv("BFD_SHEAR_ORDER")) orderMode = atoi(ev);
        const bool reorder = count > 0 && orderMode != 0;
        if (e == hipSuccess && reorder) {
            unsigned long long *k0 = nullptr, *k1 = nullptr; unsigned *v1 = nullptr; void *w2 = nullptr; size_t w2b = 0;
            e = hipMalloc((void **)&k0, (size_t)count * 8);
            if (e == hipSuccess) e = hipMalloc((void **)&k1, (size_t)count * 8);
            if (e == hipSuccess) e = hipMalloc((void **)&v1, (size_t)count * 4);
            if (e == hipSuccess) {
                bfd_launch_shear_order_keys(s->d, s->stream, sel, k0, count, lowPlanes, hiStart, orderMode);
                e = hipcub::DeviceRadixSort::SortPairs(nullptr, w2b, k0, k1, sel, v1, count, 0, 46, s->stream);
            }
            if (e == hipSuccess) e = hipMalloc(&w2, std::max<size_t>(w2b, 1));
            if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(w2, w2b, k0, k1, sel, v1, count, 0, 46, s->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(sel, v1, (size_t)count * 4, hipMemcpyDeviceToDevice, s->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
            if (k0) hipFree(k0); if (k1) hipFree(k1); if (v1) hipFree(v1); if (w2) hipFree(w2);
        }
        // Compact solid state (bfd_dev::cssRow): Sxx, Syy, the shear stresses and the five memory variables Rxx, Ryy, Rxy, Rxz, Ryz of the listed cells in
        // list order. Needs the row-contiguous list order (mode 2) and the two-kernel form. In a Z-slab the ghost planes of Sxz / Syz stay in the
        // full-volume arrays (the sparse kernel keeps full-volume copies of the planes a neighbour reads, the velocity kernel takes ghost planes from
        // there): the halo exchange is unchanged. BFD_COMPACT_SOLID=0 keeps the full-volume arrays.
        bool compact = count > 0 && orderMode == 2 && !T.merged && s->d.N1 <= 4095 && bfd_css_supported();
        if (const char *ev = getenv("BFD_COMPACT_SOLID")) compact = compact && atoi(ev) != 0;
        if (e == hipSuccess) {
            rc = dev_alloc(s, &s->tiles.shearCells, (size_t)std::max(count, 1), false);
            if (!rc) rc = dev_alloc(s, &s->tiles.shearCoef, 6 * (size_t)std::max(count, 1), false);
            if (!rc && !T.merged && !compact) rc = dev_alloc(s, &s->tiles.shearR, 3 * (size_t)std::max(count, 1), true);      // lists are built at step 0: the memory variables start at zero (merged form: they live in the full-volume arrays; compact form: with the other compact arrays)
            if (!rc && count) e = hipMemcpyAsync(s->tiles.shearCells, sel, (size_t)count * sizeof(unsigned), hipMemcpyDeviceToDevice, s->stream);
            if (!rc) rc = dev_alloc(s, &s->tiles.shearCodes, (size_t)std::max(count, 1), false);
            if (!rc) rc = dev_alloc(s, &s->tiles.shearTab, 8 * (size_t)s->cfg.nMat, false);
            if (!rc && e == hipSuccess) bfd_launch_shear_coefficients(s->d, s->stream, s->tiles.shearCells, s->tiles.shearCoef, s->tiles.shearCodes, s->tiles.shearTab, s->cfg.nMat, count);
            if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        }
        if (flag) hipFree(flag); if (sel) hipFree(sel); if (dcount) hipFree(dcount); if (work) hipFree(work);
        if (e != hipSuccess) BFD_FAIL(-10, std::string("shear list: ") + hipGetErrorString(e));
        if (rc) return rc;
        s->tiles.nShear = count;
        if (s->step > 0 && s->tiles.shearR) { bfd_launch_gather_shear_memory(s->d, s->stream, &s->tiles); BFD_HIP(hipStreamSynchronize(s->stream)); }
        if (compact) {
            const int stride = tx + 1;
            rc = dev_alloc(s, &s->tiles.cssRow, (size_t)(s->d.nk + 4) * s->d.N2 * stride, false);
            // the compact arrays live inside the full-volume buffers of their fields when the listed cells fit between the planes a Z-neighbour
            // exchanges (local planes 0, 1 and nk-2, nk-1 of Sxz / Syz travel: allocation planes 4 .. nk-1 are free); BFD_COMPACT_HOSTED=0 or
            // too many solid cells: one block of their own
            bool hosted = (size_t)count <= (size_t)std::max(s->d.nk - 4, 0) * s->d.plane;
            if (const char *ev = getenv("BFD_COMPACT_HOSTED")) hosted = hosted && atoi(ev) != 0;
            if (!rc && !hosted) rc = dev_alloc(s, &s->tiles.css, 10 * (size_t)count, true);
            if (rc) return rc;
            s->tiles.cssCap = count; s->tiles.cssHosted = hosted;
            bfd_launch_css_row_table(s->d, s->stream, s->tiles.shearCells, count, s->tiles.cssRow, stride, lowPlanes, hiStart);
            s->d.cssRow = s->tiles.cssRow; s->d.cssStride = stride;
            bind_compact_views(s);
            BFD_HIP(hipStreamSynchronize(s->stream));
            std::vector<int4> ra(listsAll[0]);
            ra.insert(ra.end(), listsAll[1].begin(), listsAll[1].end());
            rc = dev_alloc(s, &s->tiles.runsAll, ra.size(), false);
            if (rc) return rc;
            BFD_HIP(hipMemcpy(s->tiles.runsAll, ra.data(), ra.size() * sizeof(int4), hipMemcpyHostToDevice));
            s->tiles.nAllB = (int)listsAll[0].size(); s->tiles.nAll = (int)ra.size();
        }
        if (carryCompact) {       // the values of the old list into the new one (or, should the new state not be compact, into the full-volume arrays)
            float *tmp = nullptr;
            const size_t g = 2 * (size_t)s->d.plane;
            float *newComp[10] = {s->d.cSxx, s->d.cSyy, s->d.cSxy, s->d.cSxz, s->d.cSyz, s->d.cRxx, s->d.cRyy, s->d.cRxy, s->d.cRxz, s->d.cRyz};
            float *full[10] = {s->d.Sxx, s->d.Syy, s->d.Sxy, s->d.Sxz, s->d.Syz, s->d.Rxx, s->d.Ryy, s->d.Rxy, s->d.Rxz, s->d.Ryz};
            hipError_t e2 = s->d.cssRow ? malloc_or_release_cache((void **)&tmp, s->nalloc * sizeof(float)) : hipSuccess;
            // the new state is not compact (no solid cell left, or the form was switched off): the full-volume arrays take over, and the owned planes
            // of buffers that hosted compact arrays hold list-ordered values, not fields: cleared before the old values are scattered into them
            for (int a = 0; a < 10 && e2 == hipSuccess && !s->d.cssRow; a++) e2 = hipMemsetAsync(full[a], 0, s->nloc * sizeof(float), s->stream);
            for (int a = 0; a < 10 && e2 == hipSuccess; a++) {
                if (s->d.cssRow) {
                    e2 = hipMemsetAsync(tmp, 0, s->nalloc * sizeof(float), s->stream);
                    bfd_launch_css_scatter(s->stream, oldCells, oldN, oldComp + (size_t)a * oldN, tmp + g);
                    bfd_launch_css_gather(s->stream, s->tiles.shearCells, count, tmp + g, newComp[a]);
                } else bfd_launch_css_scatter(s->stream, oldCells, oldN, oldComp + (size_t)a * oldN, full[a]);
            }
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(s->stream);
            if (tmp) hipFree(tmp);
            if (e2 != hipSuccess) BFD_FAIL(-10, std::string("compact solid state, list rebuilt in the middle of a run: ") + hipGetErrorString(e2));
            if (!s->d.cssRow && s->tiles.shearR) { bfd_launch_gather_shear_memory(s->d, s->stream, &s->tiles); BFD_HIP(hipStreamSynchronize(s->stream)); }
        }
        s->tiles.shearLowEnd = std::lower_bound(hostCells.begin(), hostCells.end(), (unsigned)lowPlanes * (unsigned)s->d.plane) - hostCells.begin();
        s->tiles.shearHighBeg = std::lower_bound(hostCells.begin(), hostCells.end(), (unsigned)hiStart * (unsigned)s->d.plane) - hostCells.begin();
    }
    {   // sources of the first / last z-chunk (bfd_set_sources sorted them by voxel)
        std::vector<uint32_t> lin((size_t)s->nSrcVox);
        if (s->nSrcVox) BFD_HIP(hipMemcpy(lin.data(), s->srcLin, lin.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        s->srcLowEnd = std::lower_bound(lin.begin(), lin.end(), (uint32_t)lowPlanes * (uint32_t)s->d.plane) - lin.begin();
        s->srcHighBeg = std::lower_bound(lin.begin(), lin.end(), (uint32_t)hiStart * (uint32_t)s->d.plane) - lin.begin();
    }
    {   // algorithmic bytes per launch and kernel class (DESIGN.md "Kernels": per-cell byte tables of the tile classes):
        // what each kernel has to move once per half-step if every value were fetched exactly once -- float32 fields 4 B,
        // material ids 2 B; absorbing-layer memory variables, tables and halo re-reads excluded
        double (*B)[BFD_K_COUNT] = s->algBytes;
        memset(s->algBytes, 0, sizeof s->algBytes);
        const int N1 = s->d.N1, N2 = s->d.N2, N3 = s->d.N3, ND = s->d.ND, k0g = s->d.k0;
        // class counts over the cells of the solid runs (fluid / solid centre, with / without memory variables, active edges)
        unsigned long long cnt[6] = {0, 0, 0, 0, 0, 0};
        if (T.nSolid && s->cfg.kernelVariant != 2) {
            unsigned long long *dc = nullptr;
            BFD_HIP(hipMalloc((void **)&dc, sizeof cnt));
            hipMemsetAsync(dc, 0, sizeof cnt, s->stream);
            bfd_launch_count_solid_cells(s->d, s->stream, s->tiles.runs + T.nFluid, T.nSolid, dc);
            hipMemcpyAsync(cnt, dc, sizeof cnt, hipMemcpyDeviceToHost, s->stream);
            const hipError_t e = hipStreamSynchronize(s->stream);
            hipFree(dc);
            if (e != hipSuccess) BFD_FAIL(-10, std::string("solid cell counts: ") + hipGetErrorString(e));
        }
        auto overlap = [](int a, int b, int lo, int hi) { return (double)std::max(0, std::min(b, hi) - std::max(a, lo)); };
        for (size_t r = 0; r < all.size(); r++) {
            const int4 &run = all[r];
            const int bx = run.x % tx, by = run.x / tx, kb = run.y & 0xFFFF, ke = run.y >> 16, f = run.z;
            const int xa = bx * 64, xb = std::min(xa + 64, N1), ya = by * 8, yb = std::min(ya + 8, N2);
            const double cells = (double)(xb - xa) * (yb - ya) * (ke - kb);
            const double inner = overlap(xa, xb, ND, N1 - ND) * overlap(ya, yb, ND, N2 - ND) * overlap(k0g + kb, k0g + ke, ND, N3 - ND);
            const bool fusedRun = r >= (size_t)(T.nFluid + T.nSolid);
            if (fusedRun) {          // 64 x 24 cells per plane, all outside the absorbing layer: V, Szz (Rzz) read and written once per step (+ ids)
                const double fc = 64.0 * bfd_fused_rows() * (ke - kb);
                const double b = 32.0 + ((f & 2) ? 8.0 : 0.0) + ((f & 4) ? 0.0 : 2.0);
                B[0][BFD_K_FUSED] += b * fc; B[1][BFD_K_FUSED] += b * fc + 8.0 * fc;
            } else if (r < (size_t)T.nFluid) {
                const bool lossy = f & 2, uni = f & 4, single = (f & 16) != 0;
                double bs = 12.0 + 8.0 + (lossy ? 8.0 : 0.0) + (uni ? 0.0 : 2.0);
                if (!single) bs += 8.0 + (lossy ? 8.0 : 0.0);
                const double bv = 4.0 + 24.0 + (uni ? 0.0 : 2.0);
                for (int a = 0; a < 2; a++) { B[a][BFD_K_STRESS_FLUID] += bs * cells; B[a][BFD_K_VELOCITY_FLUID] += bv * cells + (a ? 8.0 * inner : 0.0); }
            } else if (s->cfg.kernelVariant == 2) {     // dense: V + 6 S + 6 R read, 6 S + 6 R written, id; 6 S + V read, V written, id
                for (int a = 0; a < 2; a++) { B[a][BFD_K_STRESS_SOLID] += 110.0 * cells; B[a][BFD_K_VELOCITY_SOLID] += 50.0 * cells + (a ? 8.0 * inner : 0.0); }
            } else {                                    // class-predicated solid kernels: per-cell terms come from the class counts below
                for (int a = 0; a < 2; a++) B[a][BFD_K_VELOCITY_SOLID] += (a ? 8.0 * inner : 0.0);
            }
        }
        if (T.nSolid && s->cfg.kernelVariant != 2) {
            // stress: V 12 + id 2 + class 1, + Szz r/w 8 (+ Rzz r/w 8) at a fluid cell, + 3 S r/w 24 + 3 R r/w 24 at a solid one
            // (a reflector cell: 48 B of zero stores); velocity: V r/w 24 + ids 2 + class 1 + Szz 4, + Sxx, Syy 8 at a solid cell,
            // + 4 per active shear edge
            const double nF0 = (double)cnt[0], nF1 = (double)cnt[1], nS = (double)cnt[2] + (double)cnt[3], nE = (double)cnt[4], nR = (double)cnt[5];
            const double all = nF0 + nF1 + nS + nR;
            const double bs = 15.0 * all + 8.0 * nF0 + 16.0 * nF1 + 48.0 * nS + 48.0 * nR;
            const double bv = 31.0 * all + 8.0 * (nS + nR) + 4.0 * nE;
            // compact solid state: the fluid kernel takes Szz / Rzz of the solid runs too (V 12 + id 2, + Szz r/w 8 (+ Rzz r/w 8); no class byte), the
            // sparse kernel Sxx, Syy, Rxx, Ryy of its cells (+ 32 + id 2 per listed cell, below)
            const double bsf = 14.0 * all + 8.0 * nF0 + 16.0 * nF1 + 16.0 * nS + 8.0 * nR;
            for (int a = 0; a < 2; a++) { if (s->d.cssRow) B[a][BFD_K_STRESS_FLUID] += bsf; else B[a][BFD_K_STRESS_SOLID] += bs; B[a][BFD_K_VELOCITY_SOLID] += bv; }
        }
        if (s->tiles.nShear) {       // sparse shear: cell index + 6 coefficients + V of the cell + read-modify-write of S and R per active edge
            unsigned long long *dc = nullptr, hc[2] = {0, 0};
            BFD_HIP(hipMalloc((void **)&dc, sizeof hc));
            hipMemsetAsync(dc, 0, sizeof hc, s->stream);
            hipLaunchKernelGGL(count_active_edges, dim3(grid_for(s->tiles.nShear)), dim3(256), 0, s->stream, s->tiles.shearCoef, s->tiles.shearCodes, s->tiles.nShear, dc);
            hipMemcpyAsync(hc, dc, sizeof hc, hipMemcpyDeviceToHost, s->stream);
            const hipError_t e = hipStreamSynchronize(s->stream);
            hipFree(dc);
            if (e != hipSuccess) BFD_FAIL(-10, std::string("shear edge count: ") + hipGetErrorString(e));
            s->tiles.nShearExplicit = (long)hc[1];
            // per listed cell: index 4 + edge codes 4 + V 12; per edge with explicit coefficients 8; per active edge S and R r/w 16; compact solid
            // state: + Sxx, Syy, Rxx, Ryy r/w 32 (+ the cell's id 2 when it does not fit the code word's fourth byte)
            const double perCell = s->d.cssRow ? (s->cfg.nMat <= 255 ? 52.0 : 54.0) : 20.0;
            for (int a = 0; a < 2; a++) B[a][BFD_K_STRESS_SHEAR] = perCell * (double)s->tiles.nShear + 8.0 * (double)hc[1] + 16.0 * (double)hc[0];
            if (T.merged) for (int a = 0; a < 2; a++) B[a][BFD_K_STRESS_SOLID] -= 16.0 * (double)hc[0];      // those edges are the sparse kernel's
        }
        if (T.merged && T.nSolid && s->cfg.kernelVariant != 2) for (int a = 0; a < 2; a++) B[a][BFD_K_STRESS_SOLID] += 16.0 * (double)cnt[4];   // S and R of every active edge, read and written
    }
    if (oldCells) dev_release(s, &oldCells);
    s->tilesReady = true;
    return 0;
}

// ---- placement of the per-voxel arrays ----------------------------------------------------------------------------------
// The tiled kernels stream 6 to 20 arrays at the same cell offset. Round 2 found that the same kernels on the same data run
// 1.50 or 1.70 ms per step at C3 depending on nothing but where hipMalloc put those arrays, and chose among whole sets of
// allocations by timing the kernels (a lottery). Round 3 found the cause (scripts/ubench_layout.hip, ubench_pairmap.hip;
// profiles/r3/placement_*): the 288 GB of HBM fall into three contiguous physical regions of about 90 GiB (the ranks of the
// 12-high stacks, as far as can be told from outside), and streams that advance together are slow when they all lie in ONE
// region and fast as soon as they are spread over two: every array in one region 0.98 + 0.76 ms for the two fluid proxies,
// arrays alternating between two regions 0.86 + 0.68 ms, on every draw. A fresh process gets all its allocations from one
// region, a fragmented device gives a mix -- the lottery's "fast sets".
// What counts are the arrays a kernel WRITES (mixing experiment of the same benchmark: the stress proxy turns fast when Szz and
// Rzz lie apart, whatever V does; the velocity proxy when Vx, Vy, Vz and the accumulator are split two and two).
// So the arrays are placed, not drawn: a pair probe (two arrays updated in place at the same cell offset along the engine's
// own runs; zeros stay zeros, so it runs on the initial state) tells whether an array lies in the region of the reference
// array Vx (about 7 % slower) or in another one. The arrays of the stream order Vx Vy Vz Szz Rzz [Sxx ... Ryz] then alternate
// between "region of Vx" and "another region": first by exchanging buffers among the 15 state arrays (all the same size, all
// zero), then, if one kind is short, with freshly allocated candidates (misses are held until the search ends, so that the
// allocator moves on; bounded by the free memory). The Pressure accumulators are re-allocated likewise when they fall on
// the wrong side. A few dozen probes of well under a millisecond; results do not depend on it.
// BFD_PLACEMENT=0 (or BFD_PLACEMENT_TRIALS=0, the round-2 name) switches it off; BFD_PLACEMENT_VERBOSE=1 prints what it does.
// Buffers that a search found in another memory region are kept when their engine is destroyed and offered to the next engine of this
// process that wants arrays of the same size on the same device (re-probed there: region classes are relative): the two or three solver calls
// of one RunCases (BASE:2338, 2374, 2401) pay the search once. Bounded (BABELFDTD_PLACEMENT_CACHE_GIB, default 48, 0 = off);
// bfd_placement_cache_release() frees it.
struct CachedBuf { int device; size_t bytes; void *p; };
static std::mutex g_cacheMutex;
static std::vector<CachedBuf> g_cache;
// at most BABELFDTD_PLACEMENT_CACHE_GIB (default 48) and never more than an eighth of the device's memory
static size_t placement_cache_cap()
{
    double gib = 48.0;
    if (const char *ev = getenv("BABELFDTD_PLACEMENT_CACHE_GIB")) gib = atof(ev);
    size_t cap = gib > 0 ? (size_t)(gib * 1073741824.0) : 0;
    size_t freeB = 0, totalB = 0;
    if (cap && hipMemGetInfo(&freeB, &totalB) == hipSuccess && totalB) cap = std::min(cap, totalB / 8);
    else (void)hipGetLastError();
    return cap;
}
// a new engine on `device` whose state arrays have `bytes` each: cached buffers of any other size are of no use to it and go back to the device
// before it allocates (they used to wait for the next bfd_destroy)
static void placement_cache_evict_other_sizes(int device, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_cacheMutex);
    for (size_t q = 0; q < g_cache.size();) {
        if (g_cache[q].device == device && g_cache[q].bytes != bytes) { hipFree(g_cache[q].p); g_cache.erase(g_cache.begin() + q); }
        else q++;
    }
    (void)hipGetLastError();
}
static size_t placement_cache_bytes(int device)
{
    std::lock_guard<std::mutex> lk(g_cacheMutex);
    size_t held = 0;
    for (const CachedBuf &c : g_cache) if (c.device == device) held += c.bytes;
    return held;
}
static std::vector<void *> placement_cache_take(int device, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_cacheMutex);
    std::vector<void *> out;
    for (size_t q = 0; q < g_cache.size();) {
        if (g_cache[q].device == device && g_cache[q].bytes == bytes) { out.push_back(g_cache[q].p); g_cache.erase(g_cache.begin() + q); }
        else q++;
    }
    return out;
}
// true if the cache took the buffer (the caller must not free it)
static bool placement_cache_put(int device, size_t bytes, void *p)
{
    std::lock_guard<std::mutex> lk(g_cacheMutex);
    hipSetDevice(device);
    const size_t cap = placement_cache_cap();
    // buffers of another size on this device are of no use to the caller that is coming: they make room first
    for (size_t q = 0; q < g_cache.size();) {
        if (g_cache[q].device == device && g_cache[q].bytes != bytes) { hipSetDevice(device); hipFree(g_cache[q].p); g_cache.erase(g_cache.begin() + q); }
        else q++;
    }
    size_t held = 0;
    for (const CachedBuf &c : g_cache) held += c.bytes;
    if (held + bytes > cap) return false;
    g_cache.push_back({device, bytes, p});
    return true;
}

static void bind_state_views(bfd_sim *s)
{
    bfd_dev &d = s->d;
    const size_t g = 2 * (size_t)d.plane;
    float **fp[15] = {&d.Vx, &d.Vy, &d.Vz, &d.Sxx, &d.Syy, &d.Szz, &d.Sxy, &d.Sxz, &d.Syz, &d.Rxx, &d.Ryy, &d.Rzz, &d.Rxy, &d.Rxz, &d.Ryz};
    for (int a = 0; a < 15; a++) *fp[a] = s->stateBase[a] + g;
    d.mat = s->matBase + g; d.cls = s->clsBase + g;
    d.VxW = d.Vx; d.VyW = d.Vy; d.VzW = d.Vz; d.SzzW = d.Szz; d.RzzW = d.Rzz;      // in-place variants: no second copies
    if (s->pingpong) { d.VxW = s->ppBase[0] + g; d.VyW = s->ppBase[1] + g; d.VzW = s->ppBase[2] + g; d.SzzW = s->ppBase[3] + g; d.RzzW = s->ppBase[4] + g; }
}

// device time of `reps` pair probes of the arrays at a and b (pointers to local plane 0), planes [0, kmax); < 0 on error
static float time_pair(bfd_sim *s, float *a, float *b, int kmax, int reps)
{
    (void)hipGetLastError();                    // a stale error of some earlier call is not this probe's
    hipEvent_t e0 = get_event(s), e1 = get_event(s);
    if (!e0 || !e1) { if (e0) s->evPool.push_back(e0); if (e1) s->evPool.push_back(e1); return -1.f; }
    for (int r = -1; r < reps; r++) {           // r = -1: untimed
        if (r == 0) hipEventRecord(e0, s->stream);
        bfd_launch_probe_pair(s->d, s->stream, &s->tiles, a, b, kmax);
    }
    hipEventRecord(e1, s->stream);
    float ms = -1.f;
    if (hipEventSynchronize(e1) == hipSuccess && hipGetLastError() == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ms /= reps;
    else ms = -1.f;
    s->evPool.push_back(e0); s->evPool.push_back(e1);
    return ms;
}

static int choose_placement(bfd_sim *s)
{
    s->placementNote = "off";
    bool on = s->placementMode != 0;
    if (const char *ev = getenv("BFD_PLACEMENT")) on = atoi(ev) != 0;
    if (const char *ev = getenv("BFD_PLACEMENT_TRIALS")) on = on && atoi(ev) != 0;
    // below ~32 M voxels the arrays a kernel streams (5 x 4 B per voxel and up) fit the 256 MB memory-side cache, where they lie
    // in DRAM stops mattering, and the probe (0.02 ms at 256^3) cannot tell the regions apart any more
    size_t minVoxels = (size_t)32 << 20;
    if (const char *ev = getenv("BFD_PLACEMENT_MIN_VOXELS")) minVoxels = (size_t)atol(ev);      // tests: exercise it on small grids too
    if (!on) return 0;
    if (s->step != 0 || s->haloHandedOut || s->cfg.kernelVariant == 1 || s->nloc < minVoxels || s->d.nk < 8 ||
        s->tiles.nFluid + s->tiles.nSolid == 0) { s->placementNote = "skipped (small grid or arrays already handed out)"; return 0; }
    // buffers 0-14: the state arrays; 15-19 (variant 4 on a whole domain): the second copies of Vx Vy Vz Szz Rzz, written in
    // the steps in which the first copies are read
    const int nBuf = s->pingpong ? 20 : 15;
    auto bufBase = [&](int a) -> float *& { return a < 15 ? s->stateBase[a] : s->ppBase[a - 15]; };
    BFD_HIP(hipSetDevice(s->cfg.device));
    const auto tStart = std::chrono::steady_clock::now();
    const bool verbose = getenv("BFD_PLACEMENT_VERBOSE") != nullptr;
    const size_t g = 2 * (size_t)s->d.plane;
    const int kmax = s->d.nk / 2;
    const size_t half = (size_t)(s->d.nk - kmax) * s->d.plane;
    const size_t bytes = s->nalloc * sizeof(float);
    const int reps = 3;
    int nProbes = 0;
    auto pair = [&](float *a, float *b) { nProbes++; return time_pair(s, a, b, kmax, reps); };
    // Two levels of pair times: "same region" (an array against itself half a slab further on is always one of these) and,
    // 7-15 % below, "different regions". Their absolute values move with the run lists of the medium, so the threshold is
    // read off the samples: the five self-pairs and Vx against every other array, sorted; the widest gap below the fastest
    // self-pair separates the levels if it is wider than 3.5 % (measured gaps: 7-10 %, scatter inside a level up to 5 %
    // top to bottom but dense); without such a gap every array lies in the region of Vx.
    float tSame = 0;
    std::vector<float> samples;
    for (int a : {0, 1, 2, 5, 11}) {
        const float t = pair(s->stateBase[a] + g, s->stateBase[a] + g + half);
        if (t <= 0) BFD_FAIL(-10, "placement: the pair probe failed on the zero state");
        if (tSame == 0 || t < tSame) tSame = t;
    }
    samples.push_back(tSame);
    std::vector<float> t0(nBuf, 0.f);
    std::string times;
    for (int a = 1; a < nBuf; a++) {
        t0[a] = pair(s->stateBase[0] + g, bufBase(a) + g);
        if (t0[a] <= 0) BFD_FAIL(-10, "placement: the pair probe failed");
        if (t0[a] <= tSame) samples.push_back(t0[a]);
        if (verbose) { char q[48]; snprintf(q, sizeof q, " %d:%.3f", a, t0[a]); times += q; }
    }
    std::sort(samples.begin(), samples.end());
    float thr = 0.95f * samples.front(), widest = 0.f;
    for (size_t q = 0; q + 1 < samples.size(); q++) {
        const float gap = samples[q + 1] / samples[q] - 1.0f;
        if (gap > widest) { widest = gap; if (gap >= 0.035f) thr = 0.5f * (samples[q] + samples[q + 1]); }
    }
    // region classes of the 15 state-sized buffers, by comparison with one representative per class
    struct Buf { float *base; int cls; bool fresh; };
    std::vector<Buf> pool;
    std::vector<float *> repOf;                                              // class -> representative (pointer to local plane 0)
    for (int a = 0; a < nBuf; a++) {
        Buf b = {bufBase(a), -1, false};
        for (size_t c = 0; c < repOf.size() && b.cls < 0; c++) {
            const float t = (c == 0 && a > 0) ? t0[a] : pair(repOf[c], b.base + g);
            if (t <= 0) BFD_FAIL(-10, "placement: the pair probe failed");
            if (verbose && c > 0) { char q[48]; snprintf(q, sizeof q, " %d/%zu:%.3f", a, c, t); times += q; }
            if (t >= thr) b.cls = (int)c;
        }
        if (b.cls < 0) { b.cls = (int)repOf.size(); repOf.push_back(b.base + g); }
        pool.push_back(b);
    }
    const bool solids = s->tiles.nSolid > 0 || s->cfg.kernelVariant == 2;
    // stream order: arrays that a kernel WRITES together are neighbours in this list, and the list alternates between the
    // most populated region M and "anywhere else":
    //   velocity kernels write Vx Vy Vz (+ accumulator); stress_fluid Szz Rzz; stress_solid Sxx Syy Szz Rxx Ryy Rzz;
    //   the sparse shear kernel Sxy Sxz Syz Rxy Rxz Ryz
    std::vector<int> order = {0, 1, 2, 5, 11};                               // Vx Vy Vz Szz Rzz
    std::vector<int> side = {0, 1, 0, 1, 0};                                 // 0: region M, 1: elsewhere
    if (s->pingpong) for (int a = 15; a < 20; a++) { order.push_back(a); side.push_back((a - 15) & 1); }   // the second copies like the first (the sums lie apart from Vz and from its copy)
    if (solids) { int q = 1; for (int a : {3, 9, 4, 10, 6, 12, 7, 13, 8, 14}) { order.push_back(a); side.push_back(q); q ^= 1; } }   // Sxx Rxx Syy Ryy Sxy Rxy Sxz Rxz Syz Ryz
    std::string before;
    for (int a : order) before += (char)('0' + std::min(pool[a].cls, 9));
    int M = 0;
    {
        std::vector<int> cnt(repOf.size(), 0);
        for (const Buf &b : pool) cnt[b.cls]++;
        for (size_t c = 0; c < cnt.size(); c++) if (cnt[c] > cnt[M]) M = (int)c;
    }
    auto sideOf = [&](const Buf &b) { return b.cls == M ? 0 : 1; };            // 0: region M, 1: elsewhere
    int need[2] = {0, 0};
    for (int sd : side) need[sd]++;
    for (const Buf &b : pool) need[sideOf(b)]--;                              // spare arrays count: their buffers can be exchanged in
    std::vector<void *> held;                                                // candidates on the side that is not short, spacers: freed at the end
    size_t heldBytes = 0;
    bool gaveUp = false;
    int nFresh = 0;
    bool probeTells = tSame >= 0.05f;                                          // ms; shorter probes are launch overhead, not memory time
    // How much throw-away memory the search for another region may hold at a time (candidates that missed + spacers, all freed
    // before this function returns). A region is up to ~96 GiB wide and a fresh process may start at the beginning of one (boxes
    // needed 92-160 GiB of candidates), but a solver call must not take the device away from whoever shares it: NOTHING is searched
    // when other allocations than this engine's are present on the device (another process, the other slabs of a group, a GUI's
    // bio-heat volumes): the buffers are then only exchanged among themselves. On a device the engine has to itself the search may
    // hold up to 192 GiB while always leaving 48 GiB of what was free on entry untouched (round 4 had capped it at 64 GiB, which
    // gives up on some boxes -- C3 85 instead of 91 Gvoxel-steps/s there; since round 5 a successful search is paid once per
    // process: its buffers are kept for the next engine, placement_cache_*). bfd_set_placement(sim, mode, limitBytes) /
    // BFD_PLACEMENT_SEARCH_MB set the limit explicitly (then the shared-device rule is off: the caller has decided);
    // BABELFDTD_PLACEMENT_SEARCH_GIB replaces the 192 GiB.
    size_t free0 = 0, total0 = 0;
    if (hipMemGetInfo(&free0, &total0) != hipSuccess) { free0 = total0 = 0; (void)hipGetLastError(); }
    // what this process keeps from an earlier engine's search (placement_cache_*) is the engine's to take, not somebody else's memory
    const size_t mine = (size_t)s->devBytes + placement_cache_bytes(s->cfg.device);
    const size_t others = total0 > free0 + mine ? total0 - free0 - mine : 0;
    // default bound: 192 GiB, never more than two thirds of what was free on entry (an empty 288 GB device keeps 90 GiB for whoever comes), always
    // leaving 48 GiB; while it walks the search watches the device's free memory: allocations that are neither this engine's nor the search's
    // own (another process that started at the same moment) end it at once and everything held goes back
    size_t heldCap = free0 > ((size_t)48 << 30) ? std::min(std::min((size_t)192 << 30, free0 / 3 * 2), free0 - ((size_t)48 << 30)) : 0;
    bool defaultRule = true;
    double searchSeconds = 2.0;                                                // BABELFDTD_PLACEMENT_SEARCH_SECONDS
    if (const char *ev = getenv("BABELFDTD_PLACEMENT_SEARCH_SECONDS")) searchSeconds = atof(ev);
    std::string capNote;
    if (others > ((size_t)6 << 30)) { heldCap = 0; char q[96]; snprintf(q, sizeof q, "; device shared (%.0f GiB of other allocations): no search beyond the own buffers", others / 1073741824.0); capNote = q; }
    if (const char *ev = getenv("BABELFDTD_PLACEMENT_SEARCH_GIB")) {           // the owner of the device raises (or lowers) the default bound without code
        const double gib = atof(ev);
        if (gib >= 0 && others <= ((size_t)6 << 30)) heldCap = std::min((size_t)(gib * 1073741824.0), free0 > ((size_t)48 << 30) ? free0 - ((size_t)48 << 30) : 0);
    }
    if (s->placementLimit >= 0) { heldCap = (size_t)s->placementLimit; capNote.clear(); defaultRule = false; }
    if (const char *ev = getenv("BFD_PLACEMENT_SEARCH_MB")) { probeTells = true; heldCap = (size_t)atol(ev) << 20; capNote.clear(); defaultRule = false; }   // tests: walk a little on any grid
    if (heldCap == 0) probeTells = false;
    int nCached = 0;
    // whatever leaves this function early gives back what the search holds: the misses, the spacers and the fresh buffers no array has taken
    struct GiveBack {
        std::vector<void *> *held; std::vector<Buf> *pool; bool armed;
        ~GiveBack() { if (!armed) return; for (void *h : *held) hipFree(h); for (const Buf &b : *pool) if (b.fresh) hipFree(b.base); (void)hipGetLastError(); }
    } giveBack{&held, &pool, true};
    size_t ownFreshBytes = 0;                                                  // fresh buffers drawn below (the cached ones are part of `mine`)
    if ((need[0] > 0 || need[1] > 0) && tSame >= 0.05f) {                      // first what an earlier engine of this process found
        for (void *cp : placement_cache_take(s->cfg.device, bytes)) {
            float *c = (float *)cp;
            int sd = -1;
            if (hipMemsetAsync(c, 0, bytes, s->stream) == hipSuccess) { const float t = pair(repOf[M], c + g); if (t > 0) sd = t >= thr ? 0 : 1; }
            if (sd >= 0 && need[sd] > 0) { pool.push_back({c, sd == 0 ? M : 100 + nFresh, true}); need[sd]--; nFresh++; nCached++; }
            else { hipStreamSynchronize(s->stream); hipFree(c); }
        }
    }
    int forcedWalk = 0;                                                        // BFD_PLACEMENT_FORCE_WALK=n (experiments): n more candidates are drawn, held and released, as an unlucky search does
    if (const char *ev = getenv("BFD_PLACEMENT_FORCE_WALK")) forcedWalk = atoi(ev);
    while ((need[0] > 0 || need[1] > 0 || forcedWalk-- > 0) && !gaveUp && probeTells) {            // draw candidates until both sides have enough
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess || freeB < 2 * bytes + totalB / 8 || heldBytes + bytes > heldCap) { gaveUp = true; break; }
        if (defaultRule) {
            // ... and a clock: what the placement is worth to ONE solver call is a few per cent of its run time, so a search that has walked for longer than
            // that (seen once: ~6 s in a process that had built and destroyed many engines before) stops and keeps what exchanging gives
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count() > searchSeconds) {
                gaveUp = true; capNote += "; search ended by its time bound"; break;
            }
            const size_t known = mine + heldBytes + ownFreshBytes + others;
            if (totalB > freeB + known && totalB - freeB - known > ((size_t)6 << 30)) { gaveUp = true; capNote += "; somebody else began to allocate on the device: search ended"; break; }
        }
        float *c = nullptr;
        if (hipMalloc((void **)&c, bytes) != hipSuccess) { (void)hipGetLastError(); gaveUp = true; break; }
        if (hipMemsetAsync(c, 0, bytes, s->stream) != hipSuccess) { hipFree(c); gaveUp = true; break; }
        const float t = pair(repOf[M], c + g);
        if (t <= 0) { hipFree(c); gaveUp = true; break; }
        const int side = t >= thr ? 0 : 1;
        if (need[side] > 0) { pool.push_back({c, side == 0 ? M : 100 + nFresh, true}); need[side]--; nFresh++; ownFreshBytes += bytes; continue; }
        held.push_back(c); heldBytes += bytes;
        // a region is ~90 GiB wide: walk on in growing strides (an unprobed throw-away block as large as everything held so
        // far, 4 GiB at most: hipMalloc of 4 GiB takes 0.3 ms, of 16 GiB 650 ms -- scripts/r3/malloc_cost.hip) instead of one
        // array at a time
        const size_t stride = std::min(heldBytes, (size_t)4 << 30);
        void *sp = nullptr;
        if (heldBytes + stride <= heldCap && hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB > stride + 2 * bytes + totalB / 8 && hipMalloc(&sp, stride) == hipSuccess) { held.push_back(sp); heldBytes += stride; }
        else (void)hipGetLastError();
    }
    std::vector<char> taken(pool.size(), 0);
    std::vector<int> slotBuf(nBuf, -1);
    auto pick = [&](int side, int prefer) -> int {
        if (!taken[prefer] && sideOf(pool[prefer]) == side) return prefer;
        for (int pass = 0; pass < 2; pass++)                                   // original buffers first, fresh ones after
            for (size_t q = 0; q < pool.size(); q++) if (!taken[q] && sideOf(pool[q]) == side && pool[q].fresh == (pass == 1)) return (int)q;
        return -1;
    };
    for (size_t q = 0; q < order.size(); q++) {
        const int a = order[q];
        int p = pick(side[q], a);
        if (p < 0) p = pick(1 - side[q], a);                                   // nothing on the wanted side
        taken[p] = 1; slotBuf[a] = p;
    }
    // the arrays outside the list take what is left of the original buffers; unused fresh ones and the held misses are released
    for (int a = 0; a < nBuf; a++) {
        if (slotBuf[a] >= 0) continue;
        int p = -1;
        for (size_t q = 0; q < pool.size() && p < 0; q++) if (!taken[q] && !pool[q].fresh) p = (int)q;
        for (size_t q = 0; q < pool.size() && p < 0; q++) if (!taken[q]) p = (int)q;
        taken[p] = 1; slotBuf[a] = p;
    }
    BFD_HIP(hipStreamSynchronize(s->stream));
    giveBack.armed = false;                                                   // from here on every buffer has an owner again
    struct FreeHeld { std::vector<void *> *held; ~FreeHeld() { for (void *h : *held) hipFree(h); held->clear(); } } freeHeld{&held};
    for (size_t q = 0; q < pool.size(); q++) {
        if (taken[q]) { if (pool[q].fresh) { s->allocs.push_back(pool[q].base); s->searched.push_back(pool[q].base); } continue; }
        if (!pool[q].fresh) {                                                 // an original buffer displaced by a fresh one
            auto it = std::find(s->allocs.begin(), s->allocs.end(), (void *)pool[q].base);
            if (it != s->allocs.end()) s->allocs.erase(it);
        }
        hipFree(pool[q].base);
    }
    for (int a = 0; a < nBuf; a++) bufBase(a) = pool[slotBuf[a]].base;
    bind_state_views(s);
    // Pressure accumulators: written beside Vx Vy Vz by the velocity kernels: the RMS sums go to another region than Vz, a
    // peak map beside them to another region than the sums
    std::string accNote;
    float *prevRep = s->stateBase[2] + g;
    for (int which = 0; which < 2; which++) {
        float **pp = which == 0 ? &s->acc : &s->pk;
        if (!*pp) continue;
        int qP = -1;
        for (int q = 0; q < s->nSelR; q++) if (s->selR[q] == BFD_MAP_PRESSURE) qP = q;
        if (qP < 0) continue;
        const size_t accBytes = (size_t)s->nSelR * s->nloc * sizeof(float);
        float *cur = *pp;
        float t = pair(prevRep, cur + (size_t)qP * s->nloc);
        std::vector<void *> miss;
        size_t missBytes = 0;
        while (t >= thr && miss.size() < 80 && probeTells) {                  // same region as its neighbour: look for another buffer
            size_t freeB = 0, totalB = 0;
            if (hipMemGetInfo(&freeB, &totalB) != hipSuccess || freeB < 2 * accBytes + totalB / 8 || missBytes + accBytes > heldCap) break;
            float *c = nullptr;
            if (hipMalloc((void **)&c, accBytes) != hipSuccess) { (void)hipGetLastError(); break; }
            if (hipMemsetAsync(c, 0, accBytes, s->stream) != hipSuccess) { hipFree(c); break; }
            const float tc = pair(prevRep, c + (size_t)qP * s->nloc);
            if (tc > 0 && tc < thr) {
                auto it = std::find(s->allocs.begin(), s->allocs.end(), (void *)cur);
                if (it != s->allocs.end()) s->allocs.erase(it);
                hipStreamSynchronize(s->stream);
                hipFree(cur);
                cur = c; s->allocs.push_back(c); t = tc;
            } else {
                miss.push_back(c); missBytes += accBytes;
                if (tc <= 0) break;
                void *sp = nullptr;                                            // walk on, as above
                const size_t stride = std::min(missBytes, (size_t)4 << 30);
                if (missBytes + stride <= heldCap && hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB > stride + 2 * accBytes + totalB / 8 && hipMalloc(&sp, stride) == hipSuccess) { miss.push_back(sp); missBytes += stride; }
                else (void)hipGetLastError();
            }
        }
        hipStreamSynchronize(s->stream);
        for (void *m : miss) hipFree(m);
        *pp = cur;
        accNote += std::string(which == 0 ? " sums " : " peaks ") + (t > 0 && t < thr ? "apart from" : "WITH") + (which == 0 ? " Vz," : " the sums,");
        prevRep = cur + (size_t)qP * s->nloc;
    }
    BFD_HIP(hipStreamSynchronize(s->stream));
    const size_t nHeld = held.size();
    for (void *h : held) hipFree(h);
    held.clear();
    std::string after;
    for (int a : order) { const Buf &b = pool[slotBuf[a]]; after += b.fresh ? (b.cls == M ? 'm' : 'n') : (char)('0' + std::min(b.cls, 9)); }
    char buf[1024];
    snprintf(buf, sizeof buf, "arrays placed by memory region (pair probe on the zero state: %d probes, within-region %.3f ms, threshold %.3f ms, gap between the levels %.0f %%; %zu regions seen): "
             "regions of %s %s -> %s (m / n = fresh allocation in / outside the most populated region),%s %d fresh, %zu candidates / spacers (%.1f GiB) released%s; %.2f s", nProbes, tSame, thr, 100.0 * widest, repOf.size(),
             (std::string("Vx Vy Vz Szz Rzz") + (s->pingpong ? " + their second copies" : "") + (solids ? " Sxx Rxx Syy Ryy Sxy Rxy Sxz Rxz Syz Ryz" : "")).c_str(), before.c_str(), after.c_str(), accNote.c_str(), nFresh, nHeld, heldBytes / 1073741824.0,
             (std::string(gaveUp ? "; search for another region given up (limit / memory)" : "") + capNote + (nCached ? "; " + std::to_string(nCached) + " of the fresh buffers came from an earlier search of this process" : "")).c_str(), std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count());
    s->placementNote = buf;
    if (verbose) fprintf(stderr, "placement: %s\nplacement: probe times, array:ms against Vx, array/class:ms against the other representatives:%s\n", buf, times.c_str());
    return 0;
}

// Quiet runs (bfd_dev::act; bfd_kernels_v2.hip): a production call lets the runs ahead of the wave front return at entry. On when the
// engine (a whole domain or a Z-slab of at least three sub-tiles: its boundary runs always work) accumulates its maps over the caller's own window (rmsFirstStep = 0: bench.py's timed windows, which accumulate from
// step 1, never see it), runs the class-specialised kernels (variants 0 / 3, in-place update) and keeps solid-only values compact; BFD_SKIP_ZERO=0
// switches it off. The map starts with the sub-tiles of the source voxels at step 0; when inputs are set again in the middle of a run every sub-tile
// counts as active from then on.
static int setup_activity_map(bfd_sim *s)
{
    bfd_dev &d = s->d;
    const bool whole = d.k0 == 0 && d.nk == d.N3;
    bool on = s->cfg.rmsFirstStep == 0 && (s->cfg.kernelVariant == 0 || s->cfg.kernelVariant == 3) && !s->pingpong &&
              s->tilesReady && (s->tiles.nSolid == 0 || d.cssRow != nullptr) && d.nk >= 3 * bfd_tile_subz();
    if (const char *ev = getenv("BFD_SKIP_ZERO_SLABS")) on = on && (whole || atoi(ev) != 0);      // 0: whole domains only (the first form of the round)
    if (const char *ev = getenv("BFD_SKIP_ZERO")) on = on && atoi(ev) != 0;
    s->actReady = true;
    BFD_HIP(hipSetDevice(s->cfg.device));
    if (!on) { d.act = nullptr; drop_step_graph(s); return 0; }
    int tx, ty, nsub; bfd_tile_grid(d, &tx, &ty, &nsub);
    const size_t bytes = (size_t)(tx + 2) * (ty + 2) * (nsub + 2);
    if (!s->actBase || s->actBytes != bytes) {
        dev_release(s, &s->actBase);
        const int rc = dev_alloc(s, &s->actBase, bytes, false);
        if (rc) return rc;
        s->actBytes = bytes;
    }
    d.act = s->actBase; d.actX = tx + 2; d.actY = ty + 2;
    {   // a Z-slab: the runs of the sub-tiles that hold the planes next to a neighbour (the "boundary" runs of a split half-step) always work
        const int SUB = bfd_tile_subz(), lowPlanes = std::min(SUB, d.nk), hiStart = std::max(((d.nk - 2) / SUB) * SUB, lowPlanes);
        d.actLo = whole ? 0 : lowPlanes; d.actHi = whole ? 0x7fffffff : hiStart;
    }
    if (s->step == 0) {
        BFD_HIP(hipMemsetAsync(s->actBase, 0, bytes, s->stream));
        bfd_launch_mark_source_subtiles(d, s->stream, s->srcLin, (long)s->nSrcVox);
    } else {
        // the border stays clear (it stands for what lies outside the domain); every sub-tile inside is taken as active
        std::vector<unsigned char> h(bytes, 0);
        for (int q = 1; q <= nsub; q++) for (int y = 1; y <= ty; y++) memset(&h[((size_t)q * (ty + 2) + y) * (tx + 2) + 1], 1, (size_t)tx);
        BFD_HIP(hipStreamSynchronize(s->stream));                            // the engine's stream does not wait for the legacy stream this copy runs on
        BFD_HIP(hipMemcpy(s->actBase, h.data(), bytes, hipMemcpyHostToDevice));
    }
    BFD_HIP(hipGetLastError());
    BFD_HIP(hipStreamSynchronize(s->stream));        // a slab's half-steps may be queued on other streams than the engine's
    drop_step_graph(s);
    return 0;
}

static int check_ready(bfd_sim *s)
{
    if (!s) BFD_FAIL(-1, "null sim");
    if (!s->haveMaterials || !s->haveMap) BFD_FAIL(-6, "materials and material map must be set before stepping");
    if (!s->classesReady) {      // per-cell class bytes (every variant: the output kernels consult them too)
        BFD_HIP(hipSetDevice(s->cfg.device));
        bfd_launch_cell_classes(s->d, s->stream, s->clsBase, (long)s->nalloc);
        BFD_HIP(hipGetLastError());
        BFD_HIP(hipStreamSynchronize(s->stream));
        s->classesReady = true;
    }
    if (!s->tilesReady && s->cfg.kernelVariant != 1) {
        BFD_HIP(hipSetDevice(s->cfg.device));
        const int rc = build_tile_lists(s);
        if (rc) return rc;
    }
    if (!s->placementDone) {
        s->placementDone = true;
        const int rc = choose_placement(s);
        if (rc) return rc;
        bind_compact_views(s);        // the state buffers may have changed hands
    }
    if (!s->actReady) { const int rc = setup_activity_map(s); if (rc) return rc; }
    return 0;
}

// sources of one part of a half-step: part 0 all, 1 = first+last z-chunk, 2 = the chunks between
// Views of the fields for the launches of one time step when V, Szz, Rzz are kept in two copies (variant 4): kernels
// read d.X and write d.XW. After the stress half-step the new Szz/Rzz are the W copies; after the velocity half-step
// so are the new V; swap_fields() then makes the W copies current. In-place variants: W == X, all views equal d.
static bfd_dev new_stress_view(const bfd_dev &d) { bfd_dev v = d; v.Szz = d.SzzW; v.Rzz = d.RzzW; return v; }
static bfd_dev new_velocity_view(const bfd_dev &d) { bfd_dev v = d; v.Vx = d.VxW; v.Vy = d.VyW; v.Vz = d.VzW; return v; }
static void swap_fields(bfd_dev &d)
{
    std::swap(d.Vx, d.VxW); std::swap(d.Vy, d.VyW); std::swap(d.Vz, d.VzW); std::swap(d.Szz, d.SzzW); std::swap(d.Rzz, d.RzzW);
}

static int inject_part(bfd_sim *s, int part, const bfd_dev &view, hipStream_t st)
{
    const int64_t n = s->nSrcVox;
    int64_t beg[2] = {0, 0}, end[2] = {0, 0};
    if (part == 0 || s->cfg.kernelVariant == 1) { if (part == 1) return 0; end[0] = n; }
    else if (part == 1) { end[0] = s->srcLowEnd; beg[1] = s->srcHighBeg; end[1] = n; }
    else { beg[0] = s->srcLowEnd; end[0] = s->srcHighBeg; }
    const float *pulse = nullptr;
    { const int rc = pulse_row(s, s->step, st, &pulse); if (rc) return rc; }
    for (int r = 0; r < 2; r++) {
        const int64_t c = end[r] - beg[r];
        if (c <= 0) continue;
        hipLaunchKernelGGL(inject_sources, dim3(grid_for(c)), dim3(256), 0, st, view, s->cfg.typeSource,
                           s->srcLin + beg[r], s->srcRow + beg[r], s->srcW[0] ? s->srcW[0] + beg[r] : nullptr,
                           s->srcW[1] ? s->srcW[1] + beg[r] : nullptr, s->srcW[2] ? s->srcW[2] + beg[r] : nullptr, pulse, (long)c);
    }
    pulse_row_read(s, s->step, st);
    return 0;
}

// part 0 = whole half-step; 1 = boundary tiles (first/last z-chunk: what a Z-neighbour reads) with their
// sources; 2 = interior tiles with theirs. Variant 1 has no tiles: part 1 is empty, part 2 is everything.
static int stress_part(bfd_sim *s, int part, hipStream_t st)
{
    int rc = check_ready(s); if (rc) return rc;
    if (part < 0 || part > 2) BFD_FAIL(-2, "half-step part must be 0, 1 or 2");
    BFD_HIP(hipSetDevice(s->cfg.device));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s->timing && s->perKernel) {
        e0 = get_event(s); e1 = get_event(s);
        if (!e0 || !e1) { if (e0) s->evPool.push_back(e0); if (e1) s->evPool.push_back(e1); e0 = e1 = nullptr; }
        else hipEventRecord(e0, st);
    }
    if (s->pingpong && part != 0) BFD_FAIL(-2, "split half-steps are not available with kernelVariant 4 on a whole domain");
    if (s->cfg.kernelVariant == 1) { if (part != 1) bfd_launch_stress_v1(s->d, st); }
    else bfd_launch_stress_v2(s->d, st, &s->tiles, part);
    if (e0) { hipEventRecord(e1, st); s->evStress.push_back(e0); s->evStress.push_back(e1); }
    if (s->nSrcVox && s->cfg.typeSource >= 2 && s->step < s->lengthSource) { rc = inject_part(s, part, new_stress_view(s->d), st); if (rc) return rc; }
    BFD_HIP(hipGetLastError());
    return 0;
}

static int velocity_part(bfd_sim *s, int part, hipStream_t st)
{
    int rc = check_ready(s); if (rc) return rc;
    if (part < 0 || part > 2) BFD_FAIL(-2, "half-step part must be 0, 1 or 2");
    BFD_HIP(hipSetDevice(s->cfg.device));
    const bfd_dev &d = s->d;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s->timing && s->perKernel) {
        e0 = get_event(s); e1 = get_event(s);
        if (!e0 || !e1) { if (e0) s->evPool.push_back(e0); if (e1) s->evPool.push_back(e1); e0 = e1 = nullptr; }
        else hipEventRecord(e0, st);
    }
    const int n = s->step;
    const bool accNow = (s->acc || s->pk) && n >= s->accStart;
    int qP = -1;   // Pressure is accumulated inside the tiled velocity kernels
    if (accNow && s->cfg.kernelVariant != 1)
        for (int q = 0; q < s->nSelR; q++) if (s->selR[q] == BFD_MAP_PRESSURE) qP = q;
    if (s->cfg.kernelVariant == 1) { if (part != 1) bfd_launch_velocity_v1(d, st); }
    else {
        float *accP = (qP >= 0 && s->acc) ? s->acc + (size_t)qP * s->nloc : nullptr;
        float *pkP = (qP >= 0 && s->pk) ? s->pk + (size_t)qP * s->nloc : nullptr;
        if (s->pingpong) {
            if (part != 0) BFD_FAIL(-2, "split half-steps are not available with kernelVariant 4 on a whole domain");
            bfd_launch_fused(d, st, accP, pkP, &s->tiles, 0, s->tiles.nFused);        // both half-steps of its runs: old fields -> W copies
        }
        bfd_launch_velocity_v2(new_stress_view(d), st, accP, pkP, &s->tiles, part);
    }
    if (e0) { hipEventRecord(e1, st); s->evVelocity.push_back(e0); s->evVelocity.push_back(e1); }
    if (s->nSrcVox && s->cfg.typeSource < 2 && s->step < s->lengthSource) { rc = inject_part(s, part, new_velocity_view(d), st); if (rc) return rc; }
    if (part == 1) { BFD_HIP(hipGetLastError()); return 0; }
    if (s->pingpong) swap_fields(s->d);
    // end of the time step: remaining accumulators, sensors
    if (accNow && !(qP >= 0 && s->nSelR == 1)) {
        SelList L; L.n = s->nSelR; memcpy(L.sel, s->selR, sizeof L.sel);
        for (int q = 0; q < BFD_MAP_COUNT; q++) L.skip[q] = (q == qP);
        dim3 block(64, 4, 1), grid((d.N1 + 63) / 64, (d.N2 + 3) / 4, d.nk);
        hipLaunchKernelGGL(accumulate_maps, grid, block, 0, st, d, L, s->acc, s->pk, (long)s->nloc);
    }
    if (s->nSensors && (s->sensOut || s->dftAcc) && n % s->cfg.sensorSub == 0 && n / s->cfg.sensorSub >= s->cfg.sensorStart) {
        const int col = n / s->cfg.sensorSub - s->cfg.sensorStart;
        if (col < s->nTs) {
            SelList L; L.n = s->nSelS; memcpy(L.sel, s->selS, sizeof L.sel); memset(L.skip, 0, sizeof L.skip);
            // compact solid state: the sensors' list entries, resolved when first needed after a list was built (round 6; every capture used to
            // walk the class bytes of the row for each solid sensor: 1.08 ms per capture on the shear medium at 512^3 against 0.42 ms at C3)
            const int *ent = nullptr;
            if (d.cssRow) {
                if (!s->sensEntValid) {
                    if (!s->sensEnt) { rc = dev_alloc(s, &s->sensEnt, (size_t)s->nSensors, false); if (rc) return rc; }
                    hipLaunchKernelGGL(sensor_entries, dim3(grid_for(s->nSensors)), dim3(256), 0, st, d, s->sensLin, (long)s->nSensors, s->sensEnt);
                    s->sensEntValid = true;
                }
                ent = s->sensEnt;
            }
            SensorBox B = {(unsigned)s->sensBox[0], (unsigned)s->sensBox[0] * (unsigned)s->sensBox[1], (unsigned)s->sensBox[2], (unsigned)s->sensBox[3], (unsigned)s->sensBox[4]};
            const uint32_t *lin = s->sensIsBox ? nullptr : s->sensLin;
            if (s->sensOut)
                hipLaunchKernelGGL(record_sensors, dim3(grid_for(s->nSensors)), dim3(256), 0, st, d, L, lin, B, ent,
                                   (long)s->nSensors, s->sensOut, col, s->nTs);
            else
                hipLaunchKernelGGL(accumulate_sensor_dft, dim3(grid_for(s->nSensors)), dim3(256), 0, st, d, L, lin, B, ent,
                                   (long)s->nSensors, s->dftAcc, s->dftPk, col, s->nTs, s->dftBin);
        }
    }
    BFD_HIP(hipGetLastError());
    s->step++;
    s->stepDevValid = false;
    return 0;
}

int bfd_half_step_stress(bfd_sim *s) { return s ? stress_part(s, 0, s->stream) : check_ready(s); }
int bfd_half_step_velocity(bfd_sim *s) { return s ? velocity_part(s, 0, s->stream) : check_ready(s); }
int bfd_half_step_stress_part(bfd_sim *s, int32_t part) { return s ? stress_part(s, part, s->stream) : check_ready(s); }
int bfd_half_step_velocity_part(bfd_sim *s, int32_t part) { return s ? velocity_part(s, part, s->stream) : check_ready(s); }
// the same on a caller-chosen stream (no synchronisation here: the caller orders the streams with events); the
// engine's own stream is left untouched, so parts may be queued on different streams back to back
int bfd_half_step_stress_part_on(bfd_sim *s, int32_t part, void *hipStream)
{
    if (!s) BFD_FAIL(-1, "null sim");
    return stress_part(s, part, (hipStream_t)hipStream);
}
int bfd_half_step_velocity_part_on(bfd_sim *s, int32_t part, void *hipStream)
{
    if (!s) BFD_FAIL(-1, "null sim");
    return velocity_part(s, part, (hipStream_t)hipStream);
}

// One plain time step recorded into the capture stream: same launches as stress_part / velocity_part (part 0),
// sources indexed by the device step counter, which the last node advances.
static void record_plain_step(bfd_sim *s, hipStream_t cs)
{
    const bfd_dev &d = s->d;
    auto inject = [&]() {
        if (!s->nSrcVox) return;
        hipLaunchKernelGGL(inject_sources_at, dim3(grid_for(s->nSrcVox)), dim3(256), 0, cs, d, s->cfg.typeSource, s->srcLin, s->srcRow,
                           s->srcW[0], s->srcW[1], s->srcW[2], s->pulseT, s->stepDev, s->nSources, s->lengthSource, (long)s->nSrcVox);
    };
    if (s->cfg.kernelVariant == 1) bfd_launch_stress_v1(d, cs); else bfd_launch_stress_v2(d, cs, &s->tiles, 0);
    if (s->cfg.typeSource >= 2) inject();
    if (s->cfg.kernelVariant == 1) bfd_launch_velocity_v1(d, cs); else bfd_launch_velocity_v2(d, cs, nullptr, nullptr, &s->tiles, 0);
    if (s->cfg.typeSource < 2) inject();
    hipLaunchKernelGGL(advance_step, dim3(1), dim3(1), 0, cs, s->stepDev);
}

static void build_step_graph(bfd_sim *s)
{
    s->graphState = -1;
    if (!s->stepDev && dev_alloc(s, &s->stepDev, 1, true)) return;
    if (!s->captureStream && hipStreamCreateWithFlags(&s->captureStream, hipStreamNonBlocking) != hipSuccess) { s->captureStream = nullptr; return; }
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(s->captureStream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return; }
    for (int q = 0; q < BFD_GRAPH_STEPS; q++) record_plain_step(s, s->captureStream);
    if (hipStreamEndCapture(s->captureStream, &g) != hipSuccess || !g) { (void)hipGetLastError(); return; }
    const hipError_t e = hipGraphInstantiate(&s->stepGraph, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) { s->stepGraph = nullptr; (void)hipGetLastError(); return; }
    s->graphState = 1;
}

// steps [n, n+G) neither accumulate nor sample sensors
static bool plain_steps(const bfd_sim *s, int n, int G)
{
    if ((s->timing && s->perKernel) || s->pingpong) return false;
    if ((s->acc || s->pk) && n + G > s->accStart) return false;
    if (s->nSensors && (s->sensOut || s->dftAcc)) {
        const int sub = s->cfg.sensorSub;
        const int last = n + G - 1;
        // a sample is taken at step m when m % sub == 0 and m / sub in [sensorStart, sensorStart + nTs)
        const int firstCol = (n + sub - 1) / sub, lastCol = last / sub;
        if (lastCol >= firstCol && lastCol >= s->cfg.sensorStart && firstCol < s->cfg.sensorStart + s->nTs) return false;
    }
    return true;
}

int bfd_run(bfd_sim *s, int32_t nSteps)
{
    int rc = check_ready(s); if (rc) return rc;
    // Graph replay is opt-in (BFD_USE_GRAPH=1): on ROCm 7.2 / MI355X replaying the 8-step graph is SLOWER than the same
    // launches issued directly (water, no accumulation: 128^3 38.8 vs 31.3 us/step, 256^3 210 vs 196 us/step), the
    // in-order stream already keeps the GPU fed from one host thread. Kept for runtimes where that changes.
    const char *ug = getenv("BFD_USE_GRAPH");
    const bool useGraph = ug && atoi(ug) != 0;
    int n = 0;
    while (n < nSteps) {
        if (useGraph && nSteps - n >= BFD_GRAPH_STEPS && s->graphState >= 0 && plain_steps(s, s->step, BFD_GRAPH_STEPS)) {
            BFD_HIP(hipSetDevice(s->cfg.device));
            if (s->graphState == 0) build_step_graph(s);
            if (s->graphState == 1) {
                if (!s->stepDevValid) { hipLaunchKernelGGL(set_step, dim3(1), dim3(1), 0, s->stream, s->stepDev, s->step); s->stepDevValid = true; }
                if (hipGraphLaunch(s->stepGraph, s->stream) == hipSuccess) { s->step += BFD_GRAPH_STEPS; n += BFD_GRAPH_STEPS; continue; }
                (void)hipGetLastError();
                s->graphState = -1;          // this runtime cannot replay into the engine's stream: direct launches from here on
            }
        }
        s->stepDevValid = false;
        rc = bfd_half_step_stress(s); if (rc) return rc;
        rc = bfd_half_step_velocity(s); if (rc) return rc;
        n++;
    }
    return 0;
}

int bfd_sync(bfd_sim *s)
{
    if (!s) BFD_FAIL(-1, "null sim");
    BFD_HIP(hipSetDevice(s->cfg.device));
    BFD_HIP(hipStreamSynchronize(s->stream));
    return 0;
}
int bfd_current_step(bfd_sim *s) { return s ? s->step : -1; }
int bfd_prepare(bfd_sim *s) { return check_ready(s); }

int bfd_halo_region(bfd_sim *s, int32_t group, int32_t f, int32_t side, int32_t send, void **devPtr, size_t *bytes)
{
    if (!s || !devPtr || !bytes) BFD_FAIL(-1, "bfd_halo_region: null argument");
    if (group < 0 || group > 1 || f < 0 || f > 2 || side < 0 || side > 1) BFD_FAIL(-2, "bfd_halo_region: bad selector");
    s->haloHandedOut = true;              // from here on the arrays stay where they are (bfd_prepare)
    const bfd_dev &d = s->d;
    float *arr[2][3] = {{d.Vx, d.Vy, d.Vz}, {d.Sxz, d.Syz, d.Szz}};
    float *a = arr[group][f];
    long kl;
    if (side == 0) kl = send ? 0 : -2; else kl = send ? d.nk - 2 : d.nk;
    *devPtr = a + kl * (long)d.plane;
    *bytes = 2 * (size_t)d.plane * sizeof(float);
    return 0;
}

int64_t bfd_placement_cache_release(void)
{
    std::lock_guard<std::mutex> lk(g_cacheMutex);
    int64_t freed = 0;
    int cur = 0;
    const bool haveCur = hipGetDevice(&cur) == hipSuccess;
    for (const CachedBuf &c : g_cache) { hipSetDevice(c.device); hipFree(c.p); freed += (int64_t)c.bytes; }
    g_cache.clear();
    pin_release_all();          // host memory: not part of the count
    if (haveCur) hipSetDevice(cur);
    (void)hipGetLastError();
    return freed;
}

int bfd_set_placement(bfd_sim *s, int32_t mode, int64_t searchLimitBytes)
{
    if (!s) BFD_FAIL(-1, "null sim");
    if (s->placementDone) BFD_FAIL(-6, "bfd_set_placement: the arrays are placed already (call it before the first step / bfd_prepare)");
    s->placementMode = mode != 0; s->placementLimit = searchLimitBytes;
    return 0;
}

const char *bfd_placement_note(bfd_sim *s)
{
    if (!s) return "";
    if (!s->placementDone) return "not prepared yet";
    return s->placementNote.c_str();
}

int bfd_halo_fields(bfd_sim *s, int32_t group, uint32_t *mask)
{
    if (!mask || group < 0 || group > 1) BFD_FAIL(-1, "bfd_halo_fields: bad argument");
    int rc = check_ready(s); if (rc) return rc;
    // an all-fluid slab of the tiled kernels keeps a single normal stress and no shear: its stencils reach across a Z face
    // only through Vz (stress half-step) and Szz (velocity half-step) -- field 2 of either group
    const bool tiled = s->cfg.kernelVariant != 1 && s->cfg.kernelVariant != 2;
    *mask = (tiled && s->tilesReady && s->tiles.nSolid == 0) ? 4u : 7u;
    return 0;
}

int bfd_timing_begin(bfd_sim *s, int32_t perKernel)
{
    if (!s) BFD_FAIL(-1, "null sim");
    BFD_HIP(hipSetDevice(s->cfg.device));
    for (hipEvent_t e : s->evStress) s->evPool.push_back(e);
    for (hipEvent_t e : s->evVelocity) s->evPool.push_back(e);
    s->evStress.clear(); s->evVelocity.clear();
    for (auto &v : s->evK) { for (hipEvent_t e : v) s->evPool.push_back(e); v.clear(); }
    s->timing = true; s->perKernel = perKernel != 0;
    s->tiles.ktimer = perKernel == 2 ? s : nullptr;      // 2: additionally one event pair around every kernel launch
    BFD_HIP(hipEventRecord(s->evBegin, s->stream));
    return 0;
}

int bfd_timing_end(bfd_sim *s, double *totalMs, double *stressMs, double *velocityMs, double *otherMs,
                   int64_t *nStress, int64_t *nVelocity)
{
    if (!s) BFD_FAIL(-1, "null sim");
    if (!s->timing) BFD_FAIL(-6, "bfd_timing_end without bfd_timing_begin");
    BFD_HIP(hipSetDevice(s->cfg.device));
    BFD_HIP(hipEventRecord(s->evEnd, s->stream));
    BFD_HIP(hipEventSynchronize(s->evEnd));
    float ms = 0;
    BFD_HIP(hipEventElapsedTime(&ms, s->evBegin, s->evEnd));
    double st = 0, ve = 0;
    for (size_t a = 0; a + 1 < s->evStress.size(); a += 2) { float t; BFD_HIP(hipEventElapsedTime(&t, s->evStress[a], s->evStress[a + 1])); st += t; }
    for (size_t a = 0; a + 1 < s->evVelocity.size(); a += 2) { float t; BFD_HIP(hipEventElapsedTime(&t, s->evVelocity[a], s->evVelocity[a + 1])); ve += t; }
    if (totalMs) *totalMs = ms;
    if (stressMs) *stressMs = st;
    if (velocityMs) *velocityMs = ve;
    if (otherMs) *otherMs = ms - st - ve;
    if (nStress) *nStress = (int64_t)s->evStress.size() / 2;
    if (nVelocity) *nVelocity = (int64_t)s->evVelocity.size() / 2;
    s->timing = false;
    s->tiles.ktimer = nullptr;
    return 0;
}

int bfd_timing_kernels(bfd_sim *s, double *msPerClass, int64_t *launchesPerClass)
{
    if (!s || !msPerClass) BFD_FAIL(-1, "bfd_timing_kernels: null argument");
    BFD_HIP(hipSetDevice(s->cfg.device));
    for (int c = 0; c < BFD_K_COUNT; c++) {
        double sum = 0;
        const std::vector<hipEvent_t> &v = s->evK[c];
        for (size_t a = 0; a + 1 < v.size(); a += 2) {
            BFD_HIP(hipEventSynchronize(v[a + 1]));
            float t; BFD_HIP(hipEventElapsedTime(&t, v[a], v[a + 1])); sum += t;
        }
        msPerClass[c] = sum;
        if (launchesPerClass) launchesPerClass[c] = (int64_t)v.size() / 2;
    }
    return 0;
}

int bfd_algorithmic_bytes(bfd_sim *s, int32_t accumulating, double *bytesPerClass)
{
    int rc = check_ready(s); if (rc) return rc;
    if (!bytesPerClass) BFD_FAIL(-1, "bfd_algorithmic_bytes: null argument");
    if (s->cfg.kernelVariant == 1) BFD_FAIL(-2, "bfd_algorithmic_bytes: kernelVariant 1 has no tile classes");
    for (int c = 0; c < BFD_K_COUNT; c++) bytesPerClass[c] = s->algBytes[accumulating ? 1 : 0][c];
    return 0;
}

int bfd_reset(bfd_sim *s)
{
    if (!s) BFD_FAIL(-1, "null sim");
    BFD_HIP(hipSetDevice(s->cfg.device));
    const bfd_dev &d = s->d;
    for (int a = 0; a < 15; a++) BFD_HIP(hipMemsetAsync(s->stateBase[a], 0, s->nalloc * sizeof(float), s->stream));
    if (s->pingpong) for (int a = 0; a < 5; a++) BFD_HIP(hipMemsetAsync(s->ppBase[a], 0, s->nalloc * sizeof(float), s->stream));
    const int P = d.P;
    const bool zTouch = (d.k0 < P) || (d.k0 + d.nk > d.N3 - P);
    static const int dirOf[18] = {0, 1, 2, 1, 0, 2, 0, 2, 1, 0, 1, 2, 0, 1, 2, 0, 1, 2};
    for (int a = 0; a < 18; a++) {
        size_t n = dirOf[a] == 0 ? (size_t)d.nk * d.N2 * 2 * P : (dirOf[a] == 1 ? (size_t)d.nk * 2 * P * d.N1 : (zTouch ? (size_t)2 * P * d.plane : 0));
        if (n) BFD_HIP(hipMemsetAsync(d.psi[a], 0, n * sizeof(float), s->stream));
    }
    if (s->tiles.shearR && s->tiles.nShear) BFD_HIP(hipMemsetAsync(s->tiles.shearR, 0, 3 * (size_t)s->tiles.nShear * sizeof(float), s->stream));
    if (s->tiles.css && s->tiles.cssCap) BFD_HIP(hipMemsetAsync(s->tiles.css, 0, 10 * (size_t)s->tiles.cssCap * sizeof(float), s->stream));      // hosted compact arrays were zeroed with their buffers
    if (s->acc) BFD_HIP(hipMemsetAsync(s->acc, 0, (size_t)s->nSelR * s->nloc * sizeof(float), s->stream));
    if (s->pk) BFD_HIP(hipMemsetAsync(s->pk, 0, (size_t)s->nSelR * s->nloc * sizeof(float), s->stream));
    if (s->sensOut) BFD_HIP(hipMemsetAsync(s->sensOut, 0, (size_t)s->nSelS * s->nTs * (size_t)s->nSensors * sizeof(float), s->stream));
    if (s->dftAcc) {
        BFD_HIP(hipMemsetAsync(s->dftAcc, 0, 2 * (size_t)s->nSelS * (size_t)s->nSensors * sizeof(double), s->stream));
        if (s->nSensors > 0) hipLaunchKernelGGL(fill_float, dim3(grid_for((long)s->nSelS * s->nSensors)), dim3(256), 0, s->stream, s->dftPk, (long)s->nSelS * s->nSensors, -INFINITY);
    }
    s->step = 0; s->stepDevValid = false; s->actReady = false;         // the activity map starts over with the state
    BFD_HIP(hipStreamSynchronize(s->stream));
    drain_pack_jobs(s);
    for (int b = 0; b < 2; b++) s->tileLoaded[b] = -1;        // the streamed source table starts over (tiles are re-packed on demand)
    return 0;
}

int64_t bfd_num_sensors(bfd_sim *s) { return s ? s->nSensors : -1; }
int32_t bfd_num_sensor_steps(bfd_sim *s) { return s ? s->nTs : -1; }

int bfd_get_sensor_index(bfd_sim *s, uint32_t *index)
{
    if (!s || (!index && s->nSensors)) BFD_FAIL(-1, "bfd_get_sensor_index: null argument");
    if ((long)s->d.N1 * s->d.N2 * s->d.N3 >= (1L << 32) - 1) BFD_FAIL(-2, "domain too large for 32-bit sensor indices");
    if (!s->nSensors) return 0;
    BFD_HIP(hipSetDevice(s->cfg.device));
    bfd_advise_result_buffer(index, (size_t)s->nSensors * sizeof(uint32_t));
    BFD_HIP(hipMemcpy(index, s->sensLin, (size_t)s->nSensors * sizeof(uint32_t), hipMemcpyDeviceToHost));
    const uint32_t off = (uint32_t)((size_t)s->d.k0 * s->d.plane + 1);
    for (int64_t v = 0; v < s->nSensors; v++) index[v] += off;
    return 0;
}

}  // extern "C"

// A large result block lands in host memory the caller has just allocated and never touched (a fresh numpy array): the copy then runs at the rate the
// kernel can fault 4 KB pages in. Transparent huge pages are in `madvise` mode on the ROCm images, so the range is advised first: one hipMemcpy of
// 4.3 GiB into untouched memory 0.46 -> 0.22 s on the MI355X box (profiles/r6/d2h_into_untouched_memory.txt). Advice only: a mapping that cannot
// take it (file-backed, already populated) is left as it is.
void bfd_advise_result_buffer(void *p, size_t bytes)
{
    if (!p || bytes < ((size_t)8 << 20)) return;
    const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, b = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
    if (b > a) (void)madvise((void *)a, (size_t)(b - a), MADV_HUGEPAGE);
}

// Device -> pageable host for the large result blocks (sensor series, maps): T host threads, each moving its slice in 16 MB pieces through two pinned
// buffers of its own (the next piece in flight while the last one is copied out). One hipMemcpy is bound by the single thread that copies out of the
// runtime's staging buffers: 4.3 GiB into untouched huge-page memory 0.22 s, this way 0.11 s (profiles/r6/d2h_into_untouched_memory.txt). `after`: the
// stream whose work produces src (waited for here). Anything that fails on the way falls back to the plain copy.
// produce (optional): fills a device piece buffer with bytes [o, o + len) of the block on the given stream -- the block then never exists as a whole
// on the device (the sensor series: no 4.6 GB scratch allocation, which in a short call waited 5 s when a placement search had just released its
// candidates -- the runtime gives them back in the background). Returns hipErrorNotSupported when it declines (small block, threads off) and a producer was given: the caller takes its old way.
static hipError_t copy_out_large(int device, void *dst, const void *src, size_t bytes, hipStream_t after,
                                 const std::function<void(float *, size_t, size_t, hipStream_t)> *produce = nullptr)
{
    int T = 4;
    if (const char *ev = getenv("BFD_D2H_THREADS")) T = atoi(ev);
    size_t PIECE = (size_t)16 << 20, least = (size_t)256 << 20;
    if (const char *ev = getenv("BFD_D2H_PIECE_KB")) PIECE = std::max((size_t)4096, ((size_t)atol(ev) << 10) & ~(size_t)4095);      // tests: small grids through the same code
    if (const char *ev = getenv("BFD_D2H_MIN_MB")) least = (size_t)atol(ev) << 20;
    if (T < 2 || bytes < least || bytes < (size_t)T * 4096) {
        if (produce) return hipErrorNotSupported;
        const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, after);
        return e == hipSuccess ? hipStreamSynchronize(after) : e;
    }
    T = std::min(T, 16);
    hipError_t e = hipStreamSynchronize(after);
    if (e != hipSuccess) return e;
    const bool traceCopy = getenv("BFD_TRACE_COPY") != nullptr;
    const auto tc0 = std::chrono::steady_clock::now();
    std::vector<char *> pin(2 * (size_t)T, nullptr);
    bool ok = true;
    for (auto &q : pin) if (ok && !(q = pin_take(PIECE))) ok = false;
    std::vector<float *> dpiece(produce ? 2 * (size_t)T : 0, nullptr);
    for (auto &q : dpiece) if (ok && hipMalloc((void **)&q, PIECE) != hipSuccess) { q = nullptr; ok = false; (void)hipGetLastError(); }
    std::vector<int> failed((size_t)T, 0);
    const auto tc1 = std::chrono::steady_clock::now();
    if (ok) {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                const size_t a = (bytes / T * t) & ~(size_t)4095, b = t + 1 == T ? bytes : (bytes / T * (t + 1)) & ~(size_t)4095;
                hipStream_t st = nullptr;
                hipEvent_t ev[2] = {nullptr, nullptr};
                if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) failed[t] = 1;
                const size_t np = (b - a + PIECE - 1) / PIECE;
                auto issue = [&](size_t i) {
                    const size_t o = a + i * PIECE, len = std::min(PIECE, b - o);
                    const void *from = (const char *)src + o;
                    if (produce) { (*produce)(dpiece[2 * t + (i & 1)], o, len, st); from = dpiece[2 * t + (i & 1)]; if (hipGetLastError() != hipSuccess) failed[t] = 1; }
                    if (hipMemcpyAsync(pin[2 * t + (i & 1)], from, len, hipMemcpyDeviceToHost, st) != hipSuccess || hipEventRecord(ev[i & 1], st) != hipSuccess) failed[t] = 1;
                };
                if (!failed[t] && np) issue(0);
                for (size_t i = 0; i < np && !failed[t]; i++) {
                    if (i + 1 < np) issue(i + 1);
                    if (failed[t] || hipEventSynchronize(ev[i & 1]) != hipSuccess) { failed[t] = 1; break; }
                    const size_t o = a + i * PIECE, len = std::min(PIECE, b - o);
                    memcpy((char *)dst + o, pin[2 * t + (i & 1)], len);
                }
                if (st) { hipStreamSynchronize(st); hipStreamDestroy(st); }
                for (int q = 0; q < 2; q++) if (ev[q]) hipEventDestroy(ev[q]);
            });
        for (auto &x : th) x.join();
        for (int t = 0; t < T; t++) if (failed[t]) ok = false;
    }
    const auto tc2 = std::chrono::steady_clock::now();
    for (auto &q : pin) pin_give(q, PIECE);
    for (auto &q : dpiece) if (q) hipFree(q);
    if (traceCopy) fprintf(stderr, "copy_out_large: %.2f GB, %d threads: buffers %.3f s, copy %.3f s, release %.3f s\n", bytes * 1e-9, T, std::chrono::duration<double>(tc1 - tc0).count(),
                           std::chrono::duration<double>(tc2 - tc1).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - tc2).count());
    if (ok) return hipSuccess;
    (void)hipGetLastError();
    if (produce) return hipErrorNotSupported;
    return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);      // the plain way
}

int bfd_sensors_into(bfd_sim *s, float *out, int64_t rowElems)
{
    // the series of selected map q go to out + q * rowElems (rowElems >= nSensors * nTs): a slab of a group
    // writes straight into its columns of the caller's array
    if (!s) BFD_FAIL(-1, "null sim");
    const size_t row = (size_t)s->nTs * (size_t)s->nSensors, n = (size_t)s->nSelS * row;
    if (!n) return 0;
    if (s->cfg.sensorMode != 0) BFD_FAIL(-6, "bfd_get_sensors: the series are not stored with sensorMode 1 (use bfd_get_sensor_dft)");
    if (!out || !s->sensOut) BFD_FAIL(-1, "bfd_get_sensors: null argument");
    if (rowElems < (int64_t)row) BFD_FAIL(-2, "bfd_get_sensors: row shorter than nSensors * nSteps");
    BFD_HIP(hipSetDevice(s->cfg.device));
    {   // piece by piece through the host threads, no scratch block on the device
        hipError_t e = hipSuccess;
        for (int q = 0; q < s->nSelS; q++) bfd_advise_result_buffer(out + (size_t)q * rowElems, row * sizeof(float));
        const bool oneBlock = (size_t)rowElems == row;
        for (int q = 0; q < (oneBlock ? 1 : s->nSelS) && e == hipSuccess; q++) {
            const long v00 = oneBlock ? 0 : (long)q * (long)row;
            const std::function<void(float *, size_t, size_t, hipStream_t)> produce = [&, v00](float *piece, size_t o, size_t len, hipStream_t st) {
                const long cnt = (long)(len / sizeof(float));
                hipLaunchKernelGGL(transpose_sensors_range, dim3(grid_for(cnt)), dim3(256), 0, st, s->sensOut, piece, (long)s->nSensors, s->nTs, v00 + (long)(o / sizeof(float)), cnt);
            };
            e = copy_out_large(s->cfg.device, out + (size_t)q * rowElems, nullptr, (oneBlock ? n : row) * sizeof(float), s->stream, &produce);
        }
        if (e == hipSuccess) return 0;
        (void)hipGetLastError();      // declined (small block, BFD_D2H_THREADS < 2) or failed on the way: the whole block through a scratch copy
    }
    float *tmp = nullptr;
    const bool trace = getenv("BFD_TRACE_SENSORS") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tt[5]; tt[0] = tnow();
    BFD_HIP(hipMalloc((void **)&tmp, n * sizeof(float)));
    tt[1] = tnow();
    hipLaunchKernelGGL(transpose_sensors, dim3(grid_for((long)n)), dim3(256), 0, s->stream, s->sensOut, tmp, (long)s->nSensors, s->nTs, s->nSelS);
    if (trace) hipStreamSynchronize(s->stream);
    tt[2] = tnow();
    hipError_t e = hipSuccess;
    for (int q = 0; q < s->nSelS; q++) bfd_advise_result_buffer(out + (size_t)q * rowElems, row * sizeof(float));
    if ((size_t)rowElems == row) e = copy_out_large(s->cfg.device, out, tmp, n * sizeof(float), s->stream);
    else
        for (int q = 0; q < s->nSelS && e == hipSuccess; q++)
            e = copy_out_large(s->cfg.device, out + (size_t)q * rowElems, tmp + (size_t)q * row, row * sizeof(float), s->stream);
    tt[3] = tnow();
    hipFree(tmp);
    tt[4] = tnow();
    if (trace) fprintf(stderr, "bfd_get_sensors: %.2f GB: hipMalloc %.3f s, transpose %.3f s, copy %.3f s, hipFree %.3f s\n", n * 4e-9, tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3]);
    if (e != hipSuccess) BFD_FAIL(-10, std::string("bfd_get_sensors: ") + hipGetErrorString(e));
    return 0;
}

extern "C" {

int bfd_get_sensors(bfd_sim *s, float *out)
{
    if (!s) BFD_FAIL(-1, "null sim");
    return bfd_sensors_into(s, out, (int64_t)s->nTs * (int64_t)s->nSensors);
}

static int download_volume(bfd_sim *s, const float *devXfast, float *out, int64_t s1, int64_t s2, int64_t s3)
{
    const bfd_dev &d = s->d;
    if (s1 < 0 || s2 < 0 || s3 < 0) BFD_FAIL(-2, "negative strides are not supported");
    const size_t span = span_elems(d.N1, d.N2, d.nk, s1, s2, s3);
    float *tmp = nullptr;
    BFD_HIP(hipMalloc((void **)&tmp, span * sizeof(float)));
    hipError_t e = hipSuccess;
    bfd_advise_result_buffer(out, span * sizeof(float));
    if (span != s->nloc) {   // non-dense view: keep what the caller has in the gaps
        e = hipMemcpyAsync(tmp, out, span * sizeof(float), hipMemcpyHostToDevice, s->stream);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(scatter_from_xfast, dim3(grid_for((long)s->nloc)), dim3(256), 0, s->stream, devXfast, tmp,
                           (long)s1, (long)s2, (long)s3, d.N1, d.N2, d.nk);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = copy_out_large(s->cfg.device, out, tmp, span * sizeof(float), s->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    hipFree(tmp);
    if (e != hipSuccess) BFD_FAIL(-10, std::string("download: ") + hipGetErrorString(e));
    return 0;
}

static void expand_if_collapsed(bfd_sim *s);

int bfd_get_map(bfd_sim *s, int32_t kind, int32_t map, float *out, int64_t s1, int64_t s2, int64_t s3)
{
    if (!s || !out) BFD_FAIL(-1, "bfd_get_map: null argument");
    BFD_HIP(hipSetDevice(s->cfg.device));
    int q = -1;
    for (int a = 0; a < s->nSelR; a++) if (s->selR[a] == map) q = a;
    if (kind != BFD_KIND_LAST && q < 0) BFD_FAIL(-2, "bfd_get_map: map was not selected in selMapsRMS");
    float *tmp = nullptr;
    BFD_HIP(hipMalloc((void **)&tmp, s->nloc * sizeof(float)));
    int rc = 0;
    if (kind == BFD_KIND_RMS) {
        if (!s->acc) { hipFree(tmp); BFD_FAIL(-2, "bfd_get_map: RMS was not selected (SelRMSorPeak)"); }
        const int nAcc = s->step - s->accStart;
        hipLaunchKernelGGL(finalize_rms, dim3(grid_for((long)s->nloc)), dim3(256), 0, s->stream, s->acc + (size_t)q * s->nloc, tmp,
                           (long)s->nloc, (float)(nAcc > 0 ? nAcc : 1));
        rc = download_volume(s, tmp, out, s1, s2, s3);
    } else if (kind == BFD_KIND_PEAK) {
        if (!s->pk) { hipFree(tmp); BFD_FAIL(-2, "bfd_get_map: peak was not selected (SelRMSorPeak)"); }
        rc = download_volume(s, s->pk + (size_t)q * s->nloc, out, s1, s2, s3);
    } else if (kind == BFD_KIND_LAST) {
        if (map < 0 || map >= BFD_MAP_COUNT) { hipFree(tmp); BFD_FAIL(-2, "bfd_get_map: bad map id"); }
        if (map >= BFD_MAP_SIGMAXX && map <= BFD_MAP_SIGMAZZ) expand_if_collapsed(s);
        hipLaunchKernelGGL(last_map, dim3(grid_for((long)s->nloc)), dim3(256), 0, s->stream, s->d, map, tmp, (long)s->nloc);
        rc = download_volume(s, tmp, out, s1, s2, s3);
    } else {
        rc = -2; bfd_set_error("bfd_get_map: bad kind");
    }
    hipFree(tmp);
    return rc;
}

static void expand_if_collapsed(bfd_sim *s)
{
    if (s->d.cssRow) return;        // compact solid state: Sxx, Syy, Rxx, Ryy full-volume are not output scratch (they may host the compact arrays); bfd_get_field builds its outputs in a temporary
    if (s->classesReady)
        hipLaunchKernelGGL(expand_normal, dim3(grid_for((long)s->nloc)), dim3(256), 0, s->stream, s->d, (long)s->nloc);
}

int bfd_get_field(bfd_sim *s, int32_t a, float *out, int64_t s1, int64_t s2, int64_t s3)
{
    if (!s || !out || a < 0 || a > 14) BFD_FAIL(-1, "bfd_get_field: bad argument");
    BFD_HIP(hipSetDevice(s->cfg.device));
    expand_if_collapsed(s);
    const bfd_dev &d = s->d;
    static const int cssOf[15] = {-1, -1, -1, 0, 1, -1, 2, 3, 4, 5, 6, -1, 7, 8, 9};
    if (d.cssRow && cssOf[a] >= 0) {       // not on tilesReady: a setter at step > 0 clears that flag, but list, row table and compact arrays stay as they are until the next step rebuilds them
        // compact solid state: the field is assembled in a temporary -- zeros, the Szz / Rzz copy at the fluid cells (Sxx, Syy, Rxx, Ryy), the compact
        // values at the listed cells
        const float *comp[10] = {d.cSxx, d.cSyy, d.cSxy, d.cSxz, d.cSyz, d.cRxx, d.cRyy, d.cRxy, d.cRxz, d.cRyz};
        float *tmp = nullptr;
        BFD_HIP(malloc_or_release_cache((void **)&tmp, s->nloc * sizeof(float)));
        hipError_t e = hipMemsetAsync(tmp, 0, s->nloc * sizeof(float), s->stream);
        if (e == hipSuccess && (a == 3 || a == 4 || a == 9 || a == 10))
            hipLaunchKernelGGL(copy_at_fluid_cells, dim3(grid_for((long)s->nloc)), dim3(256), 0, s->stream, d, a >= 9 ? (const float *)d.Rzz : (const float *)d.Szz, tmp, (long)s->nloc);
        bfd_launch_css_scatter(s->stream, s->tiles.shearCells, s->tiles.nShear, comp[cssOf[a]], tmp);
        const int rc = e == hipSuccess ? download_volume(s, tmp, out, s1, s2, s3) : -10;
        hipFree(tmp);
        if (e != hipSuccess) BFD_FAIL(-10, std::string("bfd_get_field: ") + hipGetErrorString(e));
        return rc;
    }
    if (a >= 12 && s->tilesReady && s->cfg.kernelVariant != 1) bfd_launch_scatter_shear_memory(s->d, s->stream, &s->tiles);   // Rxy, Rxz, Ryz live beside the sparse list
    float *cur[15] = {d.Vx, d.Vy, d.Vz, d.Sxx, d.Syy, d.Szz, d.Sxy, d.Sxz, d.Syz, d.Rxx, d.Ryy, d.Rzz, d.Rxy, d.Rxz, d.Ryz};
    return download_volume(s, cur[a], out, s1, s2, s3);
}

// numpy.fft.fftfreq(n, d) bin closest to freq (first minimum, like np.argmin; BASE:2498-2499)
static int dft_bin(int n, double d, double freq)
{
    int best = 0; double bd = INFINITY;
    for (int k = 0; k < n; k++) {
        const int kk = (k < (n + 1) / 2) ? k : k - n;
        const double f = (double)kk / ((double)n * d);
        const double e = fabs(f - freq);
        if (e < bd) { bd = e; best = k; }
    }
    return best;
}

int bfd_get_sensor_dft(bfd_sim *s, double freq, float *outReIm, float *outPeak)
{
    if (!s) BFD_FAIL(-1, "null sim");
    const size_t n = (size_t)s->nSelS * (size_t)s->nSensors;
    if (!n || s->nTs <= 0) return 0;
    if (!outReIm || !(s->sensOut || s->dftAcc)) BFD_FAIL(-1, "bfd_get_sensor_dft: null argument");
    BFD_HIP(hipSetDevice(s->cfg.device));
    const int bin = dft_bin(s->nTs, s->cfg.dt * s->cfg.sensorSub, freq);
    bfd_advise_result_buffer(outReIm, 2 * n * sizeof(float));
    bfd_advise_result_buffer(outPeak, n * sizeof(float));
    if (s->dftAcc) {         // accumulated in the loop
        if (bin != s->dftBin) BFD_FAIL(-2, "bfd_get_sensor_dft: with sensorMode 1 the bin is the one of the sim's own frequency");
        float *dre = nullptr;
        BFD_HIP(hipMalloc((void **)&dre, 2 * n * sizeof(float)));
        hipLaunchKernelGGL(finalize_sensor_dft, dim3(grid_for((long)(2 * n))), dim3(256), 0, s->stream, s->dftAcc, dre, (long)(2 * n), 2.0 / (double)s->nTs);
        hipError_t e = hipMemcpyAsync(outReIm, dre, 2 * n * sizeof(float), hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess && outPeak) e = hipMemcpyAsync(outPeak, s->dftPk, n * sizeof(float), hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        hipFree(dre);
        if (e != hipSuccess) BFD_FAIL(-10, std::string("bfd_get_sensor_dft: ") + hipGetErrorString(e));
        return 0;
    }
    float *dre = nullptr, *dpk = nullptr;
    BFD_HIP(hipMalloc((void **)&dre, 2 * n * sizeof(float)));
    hipError_t e = hipMalloc((void **)&dpk, n * sizeof(float));
    for (int q = 0; q < s->nSelS && e == hipSuccess; q++) {      // device block is [q][nTs][nSensors]
        hipLaunchKernelGGL(dft_series, dim3(grid_for(s->nSensors)), dim3(256), 0, s->stream,
                           s->sensOut + (size_t)q * s->nTs * s->nSensors, 1L, (long)s->nSensors, (long)s->nSensors, s->nTs, bin,
                           dre + 2 * (size_t)q * s->nSensors, dpk + (size_t)q * s->nSensors);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(outReIm, dre, 2 * n * sizeof(float), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && outPeak) e = hipMemcpyAsync(outPeak, dpk, n * sizeof(float), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    hipFree(dre); if (dpk) hipFree(dpk);
    if (e != hipSuccess) BFD_FAIL(-10, std::string("bfd_get_sensor_dft: ") + hipGetErrorString(e));
    return 0;
}

int bfd_dft_series(int32_t device, int64_t nSensors, int32_t nTs, const float *series, double dtSensor, double freq,
                   float *outReIm, float *outPeak)
{
    if (nSensors < 0 || nTs <= 0 || (nSensors && (!series || !outReIm))) BFD_FAIL(-1, "bfd_dft_series: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) BFD_FAIL(-3, "bfd_dft_series: no HIP device available (no CPU fallback)");
    if (device < 0 || device >= ndev) BFD_FAIL(-3, "bfd_dft_series: device ordinal out of range");
    if (!nSensors) return 0;
    BFD_HIP(hipSetDevice(device));
    const size_t n = (size_t)nSensors;
    float *din = nullptr, *dre = nullptr, *dpk = nullptr;
    hipError_t e = hipMalloc((void **)&din, n * nTs * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&dre, 2 * n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&dpk, n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(din, series, n * nTs * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(dft_series, dim3(grid_for((long)n)), dim3(256), 0, 0, din, (long)nTs, 1L, (long)n, nTs, dft_bin(nTs, dtSensor, freq), dre, dpk);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(outReIm, dre, 2 * n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && outPeak) e = hipMemcpy(outPeak, dpk, n * sizeof(float), hipMemcpyDeviceToHost);
    if (din) hipFree(din); if (dre) hipFree(dre); if (dpk) hipFree(dpk);
    if (e != hipSuccess) BFD_FAIL(-10, std::string("bfd_dft_series: ") + hipGetErrorString(e));
    return 0;
}

int bfd_tile_counts(bfd_sim *s, int32_t *nLossless, int32_t *nLossy, int32_t *nSolid, int32_t *nUni, int32_t *nPml)
{
    int rc = check_ready(s); if (rc) return rc;
    if (nLossless) *nLossless = s->tilesReady ? s->tiles.nLossless : 0;
    if (nLossy) *nLossy = s->tilesReady ? s->tiles.nLossy : 0;
    if (nSolid) *nSolid = s->tilesReady ? s->tiles.nSolidSub : 0;
    if (nUni) *nUni = s->tilesReady ? s->tiles.nUni : 0;
    if (nPml) *nPml = s->tilesReady ? s->tiles.nPml : 0;
    return 0;
}

int bfd_activity_counts(bfd_sim *s, int64_t *active, int64_t *total)
{
    if (!s || !active || !total) BFD_FAIL(-1, "bfd_activity_counts: null argument");
    *active = 0; *total = 0;
    if (!s->d.act || !s->actReady) return 0;
    BFD_HIP(hipSetDevice(s->cfg.device));
    std::vector<unsigned char> h(s->actBytes);
    BFD_HIP(hipStreamSynchronize(s->stream));
    BFD_HIP(hipMemcpy(h.data(), s->actBase, s->actBytes, hipMemcpyDeviceToHost));
    int tx, ty, nsub; bfd_tile_grid(s->d, &tx, &ty, &nsub);
    for (unsigned char v : h) *active += v ? 1 : 0;
    *total = (int64_t)tx * ty * nsub;
    return 0;
}

int bfd_tile_count_lean(bfd_sim *s, int32_t *nLean)
{
    int rc = check_ready(s); if (rc) return rc;
    if (nLean) *nLean = !s->tilesReady ? 0 : s->tiles.nLean;
    return 0;
}

int bfd_tile_count_fused(bfd_sim *s, int32_t *nFused)
{
    int rc = check_ready(s); if (rc) return rc;
    if (nFused) *nFused = s->tilesReady ? s->tiles.nFusedSub : 0;
    return 0;
}

int64_t bfd_device_bytes(bfd_sim *s) { return s ? s->devBytes : -1; }

}  // extern "C"
