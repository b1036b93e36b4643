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
// Two field points per thread; sources are staged through LDS in blocks and read as broadcasts.
// Geometry and phase reduction run in float64 (inputs are float32, so differences are exact), the
// trigonometry (hardware sine / cosine of the phase in revolutions) and amplitudes in float32, sums of 16 sources in
// float32, totals in float64.
#include "bfd_internal.h"
#include <math.h>

#ifndef BFD_RAYLEIGH_UNROLL
#define BFD_RAYLEIGH_UNROLL 2
#endif

namespace {

constexpr int RB = 256;      // threads per workgroup
constexpr int SB = 512;      // sources per LDS block
constexpr int SUB = 16;      // sources summed in float32 before the sum goes to the float64 totals

// Per pair: float64 difference vector and squared distance (the inputs are float32, so the differences are exact), 1/R from
// the float32 rsqrt refined by one Newton step in float64, phase k R / 2 pi reduced to [0,1) in float64, then the hardware
// sine / cosine (v_sin_f32 / v_cos_f32 take their argument in revolutions), amplitudes in float32. Sixteen sources at a time are
// summed in float32 and added to the float64 sums of the point. Round 2 spent ~70 vector instructions per pair (libm sincosf,
// float64 accumulation of every term, five scalar LDS reads per pair); this form about half of that.
// PPL = field points per lane: every source record read from LDS serves that many pairs. 4 for large point sets (899-905
// against 845-855 Gpairs/s with 2 at 6.5 M points; 128 VGPRs), 2 below a million points, 1 where even that leaves CUs without a
// workgroup (a 488 x 488 source plane is 465 workgroups at 2)
// ATT: complex wavenumber (Im k != 0). A template parameter, not a run-time test: as a uniform condition inside the pair loop the
// compiler turned it into exp + multiply + select for EVERY pair (v_exp_f32 is a quarter-rate instruction), 10 % of the loop in water.
template <int PPL, bool ATT>
__global__ __launch_bounds__(RB) void rayleigh_forward(const float *__restrict__ cen, const float *__restrict__ ds,
                                                       const float *__restrict__ u0, long nSrc, double kr, double ki,
                                                       const float *__restrict__ rf, long nPts, float *__restrict__ out)
{
    __shared__ float4 sA[SB];        // x, y, z, re(u0 dS)
    __shared__ float sB[SB];         // im(u0 dS)
    const long n0 = ((long)blockIdx.x * RB + threadIdx.x) * PPL;
    double px[PPL], py[PPL], pz[PPL], accr[PPL], acci[PPL];
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        const long n = min(n0 + p, nPts - 1);
        px[p] = rf[3 * n]; py[p] = rf[3 * n + 1]; pz[p] = rf[3 * n + 2];
        accr[p] = 0.0; acci[p] = 0.0;
    }
    const double krev = kr * (1.0 / (2.0 * M_PI));      // phase in revolutions
    const float kif = (float)ki;
    for (long base = 0; base < nSrc; base += SB) {
        const int cnt = (int)min((long)SB, nSrc - base);
        for (int q = threadIdx.x; q < cnt; q += RB) {
            const long m = base + q;
            const float a = ds[m];
            sA[q] = make_float4(cen[3 * m], cen[3 * m + 1], cen[3 * m + 2], u0[2 * m] * a);
            sB[q] = u0[2 * m + 1] * a;
        }
        __syncthreads();
        for (int q0 = 0; q0 < cnt; q0 += SUB) {
        const int q1 = min(q0 + SUB, cnt);
        float br[PPL], bi[PPL];
#pragma unroll
        for (int p = 0; p < PPL; p++) { br[p] = 0.f; bi[p] = 0.f; }
#pragma unroll BFD_RAYLEIGH_UNROLL
        for (int q = q0; q < q1; q++) {
            const float4 s = sA[q];
            const float sim = sB[q];
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                const double dx = px[p] - (double)s.x, dy = py[p] - (double)s.y, dz = pz[p] - (double)s.z;
#ifdef BFD_RAYLEIGH_NO_FMA
                const double r2 = dx * dx + dy * dy + dz * dz;
                double inv = (double)rsqrtf((float)r2);
                inv = inv * (1.5 - 0.5 * r2 * inv * inv);
#else           // explicit fused multiply-adds (the build contracts nothing by itself): 11 float64 operations per pair instead of 15
                const double r2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz * dz));
                double inv = (double)rsqrtf((float)r2);
                inv = __builtin_fma(inv, __builtin_fma(-0.5 * r2, inv * inv, 0.5), inv);          // y + y (1 - r2 y^2) / 2
#endif
                const double R = r2 * inv;
                const double rev = R * krev;
                // hardware sine / cosine: v_sin_f32 / v_cos_f32 take their argument in revolutions (a float32 Taylor polynomial on
                // the folded phase was measured too: 606 against 851 Gpairs/s, and less accurate -- profiles/README.md);
                // v_fract_f64 instead of floor + subtract: 910 -> 935 Gpairs/s. A Newton step on the square root itself (R0 = r2 y,
                // R = R0 + (r2 - R0 R0) y / 2, amplitude y: two float64 operations fewer) gives 940-953, but the amplitude then has
                // the float32 reciprocal square root's 1.2e-7 instead of 6e-8 and one near-field tie of the reference study's flat-
                // array rows moves (tests/test_rayleigh_study_gpu.py): not taken
                const float fr = (float)__builtin_amdgcn_fract(rev);   // phase / 2 pi in [0,1)
                const float sn = __builtin_amdgcn_sinf(fr), cs = __builtin_amdgcn_cosf(fr);
                float amp = (float)inv;
                if (ATT) amp *= __expf(kif * (float)R);              // exp(-i k R) with complex k: Im k < 0 attenuates
                // (re + i im) * amp * (cos - i sin)
                const float er = amp * cs, ei = amp * sn;
#ifdef BFD_RAYLEIGH_NO_FMA
                br[p] += s.w * er + sim * ei;
                bi[p] += sim * er - s.w * ei;
#else
                br[p] = __builtin_fmaf(s.w, er, __builtin_fmaf(sim, ei, br[p]));
                bi[p] = __builtin_fmaf(sim, er, __builtin_fmaf(-s.w, ei, bi[p]));
#endif
            }
        }
#pragma unroll
        for (int p = 0; p < PPL; p++) { accr[p] += (double)br[p]; acci[p] += (double)bi[p]; }
        }
        __syncthreads();
    }
    // multiply by i k / (2 pi):  (i kr - ki) (a + i b) / 2pi
    const double c = 1.0 / (2.0 * M_PI);
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        const long n = n0 + p;
        if (n < nPts) {
            out[2 * n] = (float)((-ki * accr[p] - kr * acci[p]) * c);
            out[2 * n + 1] = (float)((kr * accr[p] - ki * acci[p]) * c);
        }
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
        int ppl = nPts >= (1 << 20) ? 4 : nPts >= (1 << 18) ? 2 : 1;
        if (const char *ev = getenv("BFD_RAYLEIGH_PPL")) { const int v = atoi(ev); if (v == 1 || v == 2 || v == 4) ppl = v; }
        const dim3 grid((unsigned)((nPts + RB * ppl - 1) / (RB * ppl)));
#define RAYLEIGH_LAUNCH(P, A) hipLaunchKernelGGL((rayleigh_forward<P, A>), grid, dim3(RB), 0, 0, dc, dd, du, (long)nSrc, kReal, kImag, dr, (long)nPts, dout)
        const bool att = kImag != 0.0;
        if (ppl == 4) { if (att) RAYLEIGH_LAUNCH(4, true); else RAYLEIGH_LAUNCH(4, false); }
        else if (ppl == 2) { if (att) RAYLEIGH_LAUNCH(2, true); else RAYLEIGH_LAUNCH(2, false); }
        else { if (att) RAYLEIGH_LAUNCH(1, true); else RAYLEIGH_LAUNCH(1, false); }
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
