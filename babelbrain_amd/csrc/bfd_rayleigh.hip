// Rayleigh-Sommerfeld integral on MI355X: complex field at N points from M baffled sub-sources.
//
// Replaces `BabelViscoFDTD.tools.RayleighAndBHTE.ForwardSimple` (package absent from /root/reference),
// which the Step-2 driver calls right before the FDTD run to build the source plane, and for the
// water-only field when bUseRayleighForWater is set:
//     u2 = ForwardSimple(cwvnb_extlay, TxRC['center'], TxRC['ds'], u0, rf, deviceMetal=deviceName)
//                                   TranscranialModeling/BabelIntegrationSingle.py:295
//     (also BabelIntegrationANNULAR_ARRAY.py:383,411; BabelIntegrationCONCAVE_PHASEDARRAY.py:307,328,425,446)
//
//     u2(r_n) = (i k / 2 pi) * sum_m u0_m dS_m exp(-i k R_nm) / R_nm ,   k = k_r + i k_i  (k_i < 0 attenuates)
//
// Bound: vector ALU (not HBM, not MFMA: one sqrt + one sincos per source/point pair, no matrix shape).
// One thread per field point; sources are staged through LDS in blocks and read as broadcasts.
// Geometry and phase reduction run in float64 (inputs are float32, so differences are exact), the
// trigonometry and amplitudes in float32, the sums in float64 -- per-term phase error ~1e-7 rad.
#include "bfd_internal.h"
#include <math.h>

namespace {

constexpr int RB = 256;      // threads per workgroup = field points per workgroup
constexpr int SB = 512;      // sources per LDS block

__global__ __launch_bounds__(RB) void rayleigh_forward(const float *__restrict__ cen, const float *__restrict__ ds,
                                                       const float *__restrict__ u0, long nSrc, double kr, double ki,
                                                       const float *__restrict__ rf, long nPts, float *__restrict__ out)
{
    __shared__ float sx[SB], sy[SB], sz[SB], sre[SB], sim[SB];
    const long n = (long)blockIdx.x * RB + threadIdx.x;
    const bool live = n < nPts;
    double px = 0, py = 0, pz = 0;
    if (live) { px = rf[3 * n]; py = rf[3 * n + 1]; pz = rf[3 * n + 2]; }
    const double krev = kr * (1.0 / (2.0 * M_PI));      // phase in revolutions
    double accr = 0.0, acci = 0.0;
    for (long base = 0; base < nSrc; base += SB) {
        const int cnt = (int)min((long)SB, nSrc - base);
        for (int q = threadIdx.x; q < cnt; q += RB) {
            const long m = base + q;
            const float a = ds[m];
            sx[q] = cen[3 * m]; sy[q] = cen[3 * m + 1]; sz[q] = cen[3 * m + 2];
            sre[q] = u0[2 * m] * a; sim[q] = u0[2 * m + 1] * a;
        }
        __syncthreads();
        if (live) {
#pragma unroll 4
            for (int q = 0; q < cnt; q++) {
                const double dx = px - (double)sx[q], dy = py - (double)sy[q], dz = pz - (double)sz[q];
                const double r2 = dx * dx + dy * dy + dz * dz;
                // 1/R: float estimate refined by one Newton step in float64
                double inv = (double)rsqrtf((float)r2);
                inv = inv * (1.5 - 0.5 * r2 * inv * inv);
                const double R = r2 * inv;
                const double rev = R * krev;
                const float fr = (float)(rev - floor(rev));           // phase / 2pi in [0,1)
                float sn, cs;
                sincosf(fr * 6.283185307179586f, &sn, &cs);
                float amp = (float)inv;
                if (ki != 0.0) amp *= expf((float)(ki * R));     // exp(-i k R) with complex k: Im k < 0 attenuates
                // exp(-i k R) = amp * (cos - i sin)
                const float er = amp * cs, ei = -amp * sn;
                accr += (double)(sre[q] * er - sim[q] * ei);
                acci += (double)(sre[q] * ei + sim[q] * er);
            }
        }
        __syncthreads();
    }
    if (live) {
        // multiply by i k / (2 pi):  (i kr - ki) (a + i b) / 2pi
        const double c = 1.0 / (2.0 * M_PI);
        out[2 * n] = (float)((-ki * accr - kr * acci) * c);
        out[2 * n + 1] = (float)((kr * accr - ki * acci) * c);
    }
}

}  // namespace

extern "C" int bfd_rayleigh_forward(int32_t device, int64_t nSrc, const float *center, const float *ds, const float *u0,
                                    double kReal, double kImag, int64_t nPts, const float *rf, float *out, double *kernelMs)
{
    if (nSrc < 0 || nPts < 0 || (nSrc && (!center || !ds || !u0)) || (nPts && (!rf || !out))) {
        bfd_set_error("bfd_rayleigh_forward: null argument"); return -1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { bfd_set_error("bfd_rayleigh_forward: no HIP device available (no CPU fallback)"); return -3; }
    if (device < 0 || device >= ndev) { bfd_set_error("bfd_rayleigh_forward: device ordinal out of range"); return -3; }
    BFD_HIP(hipSetDevice(device));
    if (nPts == 0) return 0;
    float *dc = nullptr, *dd = nullptr, *du = nullptr, *dr = nullptr, *dout = nullptr;
    hipError_t e = hipMalloc((void **)&dc, std::max<size_t>(3 * nSrc, 1) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&dd, std::max<size_t>(nSrc, 1) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&du, std::max<size_t>(2 * nSrc, 1) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&dr, 3 * (size_t)nPts * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&dout, 2 * (size_t)nPts * sizeof(float));
    if (e == hipSuccess && nSrc) e = hipMemcpy(dc, center, 3 * nSrc * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && nSrc) e = hipMemcpy(dd, ds, nSrc * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && nSrc) e = hipMemcpy(du, u0, 2 * nSrc * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dr, rf, 3 * (size_t)nPts * sizeof(float), hipMemcpyHostToDevice);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e == hipSuccess) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0); }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rayleigh_forward, dim3((unsigned)((nPts + RB - 1) / RB)), dim3(RB), 0, 0, dc, dd, du, (long)nSrc, kReal, kImag, dr, (long)nPts, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) { hipEventRecord(e1, 0); e = hipEventSynchronize(e1); }
    if (e == hipSuccess && kernelMs) { float ms = 0; hipEventElapsedTime(&ms, e0, e1); *kernelMs = ms; }
    if (e == hipSuccess) e = hipMemcpy(out, dout, 2 * (size_t)nPts * sizeof(float), hipMemcpyDeviceToHost);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(dc); hipFree(dd); hipFree(du); hipFree(dr); hipFree(dout);
    if (e != hipSuccess) { bfd_set_error(std::string("bfd_rayleigh_forward: ") + hipGetErrorString(e)); return -10; }
    return 0;
}
