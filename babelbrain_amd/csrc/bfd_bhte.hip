// Pennes bio-heat equation on MI355X: explicit 7-point FDTD + CEM43 thermal dose.
//
// Replaces `BabelViscoFDTD.tools.RayleighAndBHTE.BHTE` (package absent from /root/reference), which
// BabelBrain's Step 3 calls with the Step-2 pressure amplitude map:
//     ResTemp,ResDose,MonitorSlice,Qarr,TemperaturePoints = BHTE(PMaps, MaterialMap, MaterialList, dx,
//            TotalDurationSteps, nStepsOn, cy, nFactorMonitoring=, dt=, DutyCycle=, Backend=, initT0=, initDose=,
//            MonitoringPointsMap=, stableTemp=)            ThermalModeling/CalculateTemperatureEffects.py:365-456, 960
// Scheme (documented restatement; parity with the package unpinned):
//   T' = T + cd[m] * (((((Txm+Txp)+Tym)+Typ)+Tzm)+Tzp - 6 T) + cp[m] * (Tcore - T) + (n < nStepsOn ? q : 0)
//   cd = dt k/(rho c dx^2),  cp = dt rho_b c_b w / (6e7 c)  (w in mL/min/kg),  q = dt * duty * a_abs p^2/(rho c_s) / (rho c)
//   dose += dt/60 * R^(43 - T'),  R = 0.5 for T' >= 43 else 0.25 (evaluated as exp2).   Faces of the volume keep their temperature.
// Bound: HBM (T read + write, q read, dose RMW, uint8 ids: ~21 B per voxel-step); x-fastest layout, one thread per voxel.
#include "bfd_internal.h"
#include <math.h>
#include <vector>

namespace {

__global__ __launch_bounds__(256) void bhte_step(const float *__restrict__ Tin, float *__restrict__ Tout, float *__restrict__ dose,
                                                 const float *__restrict__ q, const unsigned char *__restrict__ mat,
                                                 const float *__restrict__ cd, const float *__restrict__ cp,
                                                 int N1, int N2, int N3, float Tcore, int heating, float dtMin)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= N1 || j >= N2) return;
    const long pl = (long)N1 * N2, c = (long)k * pl + (long)j * N1 + i;
    const float T = Tin[c];
    float Tn = T;
    if (i > 0 && i < N1 - 1 && j > 0 && j < N2 - 1 && k > 0 && k < N3 - 1) {
        const int m = mat[c];
        const float s = ((((Tin[c - 1] + Tin[c + 1]) + Tin[c - N1]) + Tin[c + N1]) + Tin[c - pl]) + Tin[c + pl];
        Tn = T + cd[m] * (s - 6.0f * T);
        Tn = Tn + cp[m] * (Tcore - T);
        if (heating) Tn = Tn + q[c];
    }
    Tout[c] = Tn;
    // R^(43 - T') with R = 0.5 (T' >= 43) or 0.25: a power of two, so one exp2 instead of the generic powf (0.45 -> 0.47-0.48 of
    // the HBM peak on 21 B per voxel-step). Measured in round 3 and not kept (profiles/README.md): an XCD-contiguous block order
    // (163 against 181 Gvoxel-steps/s), a z-marching form with T(k-1), T(k), T(k+1) in registers (179-186 against 184: the
    // re-reads of T it saves, 0.98 GB per launch where 0.62 are needed, come out of the memory-side cache), the dose array in
    // another memory region than T (a linear pair probe cannot tell the regions apart at 226 MB per array)
    const float e = 43.0f - Tn;
    dose[c] = dose[c] + dtMin * exp2f(Tn >= 43.0f ? -e : -2.0f * e);
}

__global__ void gather_points(const float *__restrict__ T, const unsigned *__restrict__ idx, float *__restrict__ out, long n, long stride, long col)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t * stride + col] = T[idx[t]];
}
__global__ void gather_slice(const float *__restrict__ T, float *__restrict__ out, int N1, int N2, int N3, int jsel, long sample, long nSamples)
{   // out[(i*N3 + k)*nSamples + sample]
    const long n = (long)N1 * N3;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x) {
        const int i = (int)(v % N1), k = (int)(v / N1);
        out[((long)i * N3 + k) * nSamples + sample] = T[(long)k * N1 * N2 + (long)jsel * N1 + i];
    }
}

}  // namespace

// All volumes x-fastest (i + N1*(j + N2*k)), float32; mat uint8 ids into cd/cp (nMat <= 256).
// T, dose: in/out (initial -> final). q: heat increment per ON step (already multiplied by dt*duty/(rho c)).
// monitorSlice (may be NULL): [N1][N3][nSliceSamples], plane j = sliceJ sampled every nFactorMonitoring steps.
// points (may be NULL): [nPoints][nSteps], temperature after every step at the listed voxels.
// q: nFields volumes, one per pressure field (BHTEMultiplePressureFields: steered multi-point sonications,
// CalculateTemperatureEffects.py:381, 978). fieldOfStep[s] = which of them heats during step s, -1 = none.
extern "C" int bfd_bhte_run_fields(int32_t device, int32_t N1, int32_t N2, int32_t N3, int32_t nMat, const unsigned char *mat,
                                   const float *cd, const float *cp, int32_t nFields, const float *q, float *T, float *dose,
                                   float Tcore, double dt, int32_t nSteps, const int32_t *fieldOfStep, int32_t sliceJ,
                                   int32_t nFactorMonitoring, float *monitorSlice, int64_t nPoints, const uint32_t *pointIndex,
                                   float *points, double *kernelMs)
{
    if (N1 < 3 || N2 < 3 || N3 < 3 || nMat < 1 || nMat > 256 || nFields < 1 || !mat || !cd || !cp || !q || !T || !dose || nSteps < 0 ||
        (nSteps > 0 && !fieldOfStep)) {
        bfd_set_error("bfd_bhte_run: bad argument"); return -1;
    }
    for (int s = 0; s < nSteps; s++)
        if (fieldOfStep[s] < -1 || fieldOfStep[s] >= nFields) { bfd_set_error("bfd_bhte_run: fieldOfStep entry out of range"); return -1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { bfd_set_error("bfd_bhte_run: no HIP device available (no CPU fallback)"); return -3; }
    if (device < 0 || device >= ndev) { bfd_set_error("bfd_bhte_run: device ordinal out of range"); return -3; }
    BFD_HIP(hipSetDevice(device));
    const size_t n = (size_t)N1 * N2 * N3;
    const int fm = nFactorMonitoring > 0 ? nFactorMonitoring : 1;
    const long nSamples = (monitorSlice && sliceJ >= 0) ? (nSteps + fm - 1) / fm : 0;
    float *dT[2] = {nullptr, nullptr}, *dDose = nullptr, *dq = nullptr, *dcd = nullptr, *dcp = nullptr, *dSlice = nullptr, *dPts = nullptr;
    unsigned char *dmat = nullptr; unsigned *dIdx = nullptr;
    std::vector<void *> allocs;
    auto A = [&](void **p, size_t bytes) { hipError_t e = hipMalloc(p, bytes ? bytes : 1); if (e == hipSuccess) allocs.push_back(*p); return e; };
    hipError_t e = A((void **)&dT[0], n * 4);
    if (e == hipSuccess) e = A((void **)&dT[1], n * 4);
    if (e == hipSuccess) e = A((void **)&dDose, n * 4);
    if (e == hipSuccess) e = A((void **)&dq, n * 4 * (size_t)nFields);
    if (e == hipSuccess) e = A((void **)&dmat, n);
    if (e == hipSuccess) e = A((void **)&dcd, nMat * 4);
    if (e == hipSuccess) e = A((void **)&dcp, nMat * 4);
    if (e == hipSuccess && nSamples) e = A((void **)&dSlice, (size_t)N1 * N3 * nSamples * 4);
    if (e == hipSuccess && nPoints && points) { e = A((void **)&dIdx, nPoints * 4); if (e == hipSuccess) e = A((void **)&dPts, (size_t)nPoints * nSteps * 4); }
    if (e == hipSuccess) e = hipMemcpy(dT[0], T, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dDose, dose, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dq, q, n * 4 * (size_t)nFields, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dmat, mat, n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dcd, cd, nMat * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dcp, cp, nMat * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && dIdx) e = hipMemcpy(dIdx, pointIndex, nPoints * 4, hipMemcpyHostToDevice);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int cur = 0;
    if (e == hipSuccess) {
        hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0);
        const dim3 block(64, 4, 1), grid((N1 + 63) / 64, (N2 + 3) / 4, N3);
        const float dtMin = (float)(dt / 60.0);
        for (int s = 0; s < nSteps; s++) {
            const int f = fieldOfStep[s];
            hipLaunchKernelGGL(bhte_step, grid, block, 0, 0, dT[cur], dT[1 - cur], dDose, dq + (f < 0 ? 0 : (size_t)f * n), dmat, dcd, dcp, N1, N2, N3, Tcore, f >= 0 ? 1 : 0, dtMin);
            cur = 1 - cur;
            if (dPts) hipLaunchKernelGGL(gather_points, dim3((unsigned)((nPoints + 255) / 256)), dim3(256), 0, 0, dT[cur], dIdx, dPts, (long)nPoints, (long)nSteps, (long)s);
            if (dSlice && s % fm == 0) hipLaunchKernelGGL(gather_slice, dim3(256), dim3(256), 0, 0, dT[cur], dSlice, N1, N2, N3, sliceJ, (long)(s / fm), nSamples);
        }
        hipEventRecord(e1, 0);
        e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess && kernelMs) { float ms = 0; hipEventElapsedTime(&ms, e0, e1); *kernelMs = ms; }
    }
    if (e == hipSuccess) e = hipMemcpy(T, dT[cur], n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(dose, dDose, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && dSlice) e = hipMemcpy(monitorSlice, dSlice, (size_t)N1 * N3 * nSamples * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && dPts) e = hipMemcpy(points, dPts, (size_t)nPoints * nSteps * 4, hipMemcpyDeviceToHost);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    for (void *p : allocs) hipFree(p);
    if (e != hipSuccess) { bfd_set_error(std::string("bfd_bhte_run: ") + hipGetErrorString(e)); return -10; }
    return 0;
}

// One pressure field heating during the first nStepsOn steps (the reference's BHTE call).
extern "C" int bfd_bhte_run(int32_t device, int32_t N1, int32_t N2, int32_t N3, int32_t nMat, const unsigned char *mat,
                            const float *cd, const float *cp, const float *q, float *T, float *dose, float Tcore, double dt,
                            int32_t nSteps, int32_t nStepsOn, int32_t sliceJ, int32_t nFactorMonitoring, float *monitorSlice,
                            int64_t nPoints, const uint32_t *pointIndex, float *points, double *kernelMs)
{
    if (nSteps < 0) { bfd_set_error("bfd_bhte_run: bad argument"); return -1; }
    std::vector<int32_t> sched((size_t)nSteps + 1, -1);
    for (int s = 0; s < nSteps && s < nStepsOn; s++) sched[s] = 0;
    return bfd_bhte_run_fields(device, N1, N2, N3, nMat, mat, cd, cp, 1, q, T, dose, Tcore, dt, nSteps, sched.data(), sliceJ,
                               nFactorMonitoring, monitorSlice, nPoints, pointIndex, points, kernelMs);
}
