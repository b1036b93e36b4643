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
// Bound: HBM. One step moves T read + write, q read, dose RMW, uint8 ids: ~21 B per voxel; the default path takes FOUR steps per
// launch (round 6: bhte_stepNg; stretches it cannot take go to the two-step kernel bhte_step2g and the one-step kernel) and moves those bytes
// once for all of them. x-fastest layout.
#include "bfd_internal.h"
#include <math.h>
#include <vector>

// steps per pass of the default path (bhte_stepNg): four (bhte_run_core says how that was chosen)
#ifndef BFD_BHTE_STEPS_HEATING
#define BFD_BHTE_STEPS_HEATING 4
#endif
#ifndef BFD_BHTE_STEPS_COOLING
#define BFD_BHTE_STEPS_COOLING 4
#endif

namespace {

// (wave-uniform plane base) + (32-bit byte offset in a VGPR): see bfd_kernels_v2.hip, uni() / F4()
template <typename T>
__device__ __forceinline__ T *uni(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float &F4(float *base, unsigned byteOfs) { return *(float *)((char *)uni(base) + byteOfs); }
__device__ __forceinline__ const float &F4(const float *base, unsigned byteOfs) { return *(const float *)((const char *)uni(base) + byteOfs); }

// One cell, one step, in the oracle's operation order (oracle/bhte_oracle.py); the roundings are pinned so that the one-step
// kernel, the two-step kernel and the monitors of an intermediate step give the same bits.
// REV: the neighbours are summed slowest axis first (volumes handed over in numpy C order, whose LAST axis is the fastest one
// here: the oracle sums axis 0 first), otherwise fastest axis first (x-fastest volumes of the legacy entry points).
template <bool REV>
__device__ __forceinline__ float bhte_update(float T, float xm, float xp, float ym, float yp, float zm, float zp, float cd, float cp,
                                             float Tcore, bool heating, float q)
{
    const float s = REV ? __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(zm, zp), ym), yp), xm), xp)
                        : __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(xm, xp), ym), yp), zm), zp);
    float Tn = __fadd_rn(T, __fmul_rn(cd, __fsub_rn(s, __fmul_rn(6.0f, T))));
    Tn = __fadd_rn(Tn, __fmul_rn(cp, __fsub_rn(Tcore, T)));
    if (heating) Tn = __fadd_rn(Tn, q);
    return Tn;
}
// dt/60 * R^(43 - T') with R = 0.5 (T' >= 43) or 0.25: a power of two, so one exp2 instead of the generic powf
__device__ __forceinline__ float bhte_dose_rate(float Tn, float dtMin)
{
    const float e = 43.0f - Tn;
    // v_exp_f32 alone: exp2f() wraps it in a range check for results below 2^-126 (five more instructions per call, two calls per
    // cell); such a term (T' < -20 degC) is 1e-38 of a minute and flushes to zero here
    return __fmul_rn(dtMin, __builtin_amdgcn_exp2f(Tn >= 43.0f ? -e : -2.0f * e));
}
template <bool REV>
__device__ __forceinline__ float bhte_cell(const float *__restrict__ Tin, const float *__restrict__ q, const unsigned char *__restrict__ mat,
                                           const float *__restrict__ cd, const float *__restrict__ cp, int i, int j, int k, int N1, int N2, int N3,
                                           float Tcore)
{
    const long pl = (long)N1 * N2, c = (long)k * pl + (long)j * N1 + i;
    const float T = Tin[c];
    if (!(i > 0 && i < N1 - 1 && j > 0 && j < N2 - 1 && k > 0 && k < N3 - 1)) return T;
    const int m = mat[c];
    return bhte_update<REV>(T, Tin[c - 1], Tin[c + 1], Tin[c - N1], Tin[c + N1], Tin[c - pl], Tin[c + pl], cd[m], cp[m], Tcore, q != nullptr, q ? q[c] : 0.0f);
}

// One step, one thread per voxel (odd step counts, BFD_BHTE_FUSE=0). q == nullptr: no heating in this step.
template <bool REV>
__global__ __launch_bounds__(256) void bhte_step(const float *__restrict__ Tin, float *__restrict__ Tout, float *__restrict__ dose,
                                                 const float *__restrict__ q, const unsigned char *__restrict__ mat,
                                                 const float *__restrict__ cd, const float *__restrict__ cp,
                                                 int N1, int N2, int N3, float Tcore, float dtMin)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= N1 || j >= N2) return;
    const long c = (long)k * N1 * N2 + (long)j * N1 + i;
    const float Tn = bhte_cell<REV>(Tin, q, mat, cd, cp, i, j, k, N1, N2, N3, Tcore);
    Tout[c] = Tn;
    // Measured in round 3 and not kept (profiles/README.md): an XCD-contiguous block order (163 against 181 Gvoxel-steps/s), a
    // z-marching form with T(k-1), T(k), T(k+1) in registers (179-186 against 184: the re-reads of T it saves come out of the
    // memory-side cache), the dose array in another memory region than T
    dose[c] = __fadd_rn(dose[c], bhte_dose_rate(Tn, dtMin));
}

// Two steps per launch, round-3 form (BFD_BHTE_KERNEL=1; the default is bhte_step2g below). One step moves 21 B per voxel (T in, T out, dose in and out, heat source, material id) for ~12
// flops; two steps in one pass move the same 21 B: T(n) in, T(n+2) out, the dose read once and written once with both
// increments, heat source and id read once. A workgroup marches a z-run over a tile of 64 x 26 output cells: the region it
// keeps is 68 x 30 cells (two rings: T(n+1) is needed one cell around the outputs, T(n) one cell around that), 2040 cells =
// 4 per thread in a FIXED assignment (cell e = tid + 512 n, row-major in the region), so every thread holds the z-queues of its
// cells in registers (T(n) at p-1, p, p+1; T(n+1) at p-2, p-1, p) and only the in-plane neighbours go through LDS: plane p of
// T(n) and planes p-1, p of T(n+1), each double-buffered so that one barrier per plane is enough.
constexpr int B2_W = 68, B2_H = 30, B2_TY = B2_H - 4, B2_T = 512, B2_NC = 4, B2_CELLS = B2_W * B2_H;
// Loads: T(n) of plane p+1 and the dose at the top of iteration p, material id and heat source where they are used. Issuing
// all of them at the top, or one plane ahead (before or after the barrier, +6 VGPRs: 6 waves/SIMD instead of 7) was measured
// slower: 312-319 / 338-347 against 374-392 Gvoxel-steps/s at 384^3 (profiles/r3/bhte_two_steps_per_launch.txt).
// Addressing: plane bases are wave-uniform (scalar registers), every cell keeps one unsigned 32-bit offset inside the plane.
#define B2_ARGS const float *__restrict__ Tin, float *__restrict__ Tout, float *__restrict__ dose, const float *__restrict__ qa, const float *__restrict__ qb, \
                const unsigned char *__restrict__ mat, const float *__restrict__ cd, const float *__restrict__ cp, int nMat, int N1, int N2, int N3, float Tcore, \
                float dtMin, int zrun, int tilesX, int tilesY, int nBlocks, int xcdOrder
template <bool REV>
__device__ __forceinline__ void bhte_step2_body(int b, B2_ARGS)
{
    __shared__ float A[2][B2_NC * B2_T], B[2][B2_NC * B2_T];
    __shared__ float sCd[256], sCp[256];
    const int tid = threadIdx.x;
    for (int m = tid; m < nMat; m += B2_T) { sCd[m] = cd[m]; sCp[m] = cp[m]; }
    const int bx = b % tilesX, by = (b / tilesX) % tilesY, bz = b / (tilesX * tilesY);
    const int x0 = bx * 64 - 2, y0 = by * B2_TY - 2, z0 = bz * zrun, z1 = min(z0 + zrun, N3);
    const long pl = (long)N1 * N2;

    // flags: bit 0 inside the volume, 1 T(n+1) computed here (off the x / y faces), 2 output cell, 3 on an x / y face;
    // bits 8-15: material id of the cell one plane down
    unsigned off[B2_NC], flags[B2_NC];
    float t0m[B2_NC], t0c[B2_NC], t1m[B2_NC], t1c[B2_NC], qprev[B2_NC];
    #pragma unroll
    for (int n = 0; n < B2_NC; n++) {
        const int e = tid + n * B2_T, ry = e / B2_W, rx = e - ry * B2_W, gi = x0 + rx, gj = y0 + ry;
        const bool in = e < B2_CELLS && gi >= 0 && gi < N1 && gj >= 0 && gj < N2;
        const bool face = gi == 0 || gi == N1 - 1 || gj == 0 || gj == N2 - 1;
        unsigned f = in ? 1u : 0u;
        if (in && !face && rx >= 1 && rx <= B2_W - 2 && ry >= 1 && ry <= B2_H - 2) f |= 2u;
        if (in && rx >= 2 && rx <= B2_W - 3 && ry >= 2 && ry <= B2_H - 3) f |= 4u;
        if (face) f |= 8u;
        flags[n] = f; off[n] = in ? 4u * (unsigned)(gj * N1 + gi) : 0u;          // byte offset inside a plane
        t0m[n] = (in && z0 - 2 >= 0) ? F4(Tin + (long)(z0 - 2) * pl, off[n]) : 0.0f;
        t0c[n] = (in && z0 - 1 >= 0) ? F4(Tin + (long)(z0 - 1) * pl, off[n]) : 0.0f;
        t1m[n] = t1c[n] = qprev[n] = 0.0f;
    }
    for (int p = z0 - 1; p <= z1; p++) {                            // T(n+1) of plane p (none for p = -1, N3), then T(n+2) of plane p-1
        float *Ap = A[p & 1], *Bp = B[p & 1]; const float *Bq = B[(p & 1) ^ 1];
        float t0p[B2_NC], dz[B2_NC];
        const bool outs = p - 1 >= z0 && p - 1 < z1;
        const bool inner = p > 0 && p < N3 - 1, innerOut = p - 1 > 0 && p - 1 < N3 - 1;
        const long cp0 = (long)__builtin_amdgcn_readfirstlane(p) * pl;                              // wave-uniform plane bases
        const float *TinUp = Tin + cp0 + pl, *qaP = qa ? qa + cp0 : nullptr, *qbO = qb ? qb + cp0 - pl : nullptr;
        float *doseO = dose + cp0 - pl, *ToutO = Tout + cp0 - pl;
        const unsigned char *matP = mat + cp0;
        #pragma unroll
        for (int n = 0; n < B2_NC; n++) {
            t0p[n] = (p + 1 < N3 && (flags[n] & 1u)) ? F4(TinUp, off[n]) : 0.0f;
            dz[n] = (outs && (flags[n] & 4u)) ? F4(doseO, off[n]) : 0.0f;
            Ap[tid + n * B2_T] = t0c[n];
        }
        __syncthreads();
        #pragma unroll
        for (int n = 0; n < B2_NC; n++) {
            const int e = tid + n * B2_T;
            float T1 = t0c[n], q = 0.0f; int m = 0;
            if (inner && (flags[n] & 2u)) {
                m = uni(matP)[off[n] >> 2]; if (qa) q = F4(qaP, off[n]);
                T1 = bhte_update<REV>(t0c[n], Ap[e - 1], Ap[e + 1], Ap[e - B2_W], Ap[e + B2_W], t0m[n], t0p[n], sCd[m], sCp[m], Tcore, qa != nullptr, q);
            }
            Bp[e] = T1;
            if (outs && (flags[n] & 4u)) {
                float T2 = t1c[n];
                if (innerOut && !(flags[n] & 8u)) {
                    const int m2 = flags[n] >> 8;
                    const float q2 = qb ? (qb == qa ? qprev[n] : F4(qbO, off[n])) : 0.0f;
                    T2 = bhte_update<REV>(t1c[n], Bq[e - 1], Bq[e + 1], Bq[e - B2_W], Bq[e + B2_W], t1m[n], T1, sCd[m2], sCp[m2], Tcore, qb != nullptr, q2);
                }
                F4(ToutO, off[n]) = T2;
                F4(doseO, off[n]) = __fadd_rn(__fadd_rn(dz[n], bhte_dose_rate(t1c[n], dtMin)), bhte_dose_rate(T2, dtMin));
            }
            t0m[n] = t0c[n]; t0c[n] = t0p[n]; t1m[n] = t1c[n]; t1c[n] = T1; flags[n] = (flags[n] & 15u) | ((unsigned)m << 8); qprev[n] = q;
        }
    }
}

template <bool REV>
__global__ __launch_bounds__(B2_T) void bhte_step2(B2_ARGS)
{
    int b = blockIdx.x;
    if (xcdOrder) {                               // workgroups go to the 8 XCDs round-robin: give each XCD a contiguous piece of the tile order
        const int per = nBlocks >> 3, rem = nBlocks & 7, x = b & 7, slot = b >> 3;
        b = x * per + (x < rem ? x : rem) + slot;
    }
    bhte_step2_body<REV>(b, Tin, Tout, dose, qa, qb, mat, cd, cp, nMat, N1, N2, N3, Tcore, dtMin, zrun, tilesX, tilesY, nBlocks, xcdOrder);
}

// ---- round 4: the same two steps with loads that stay in flight (bhte_step2g, the default; BFD_BHTE_KERNEL=1 selects bhte_step2) ----
// What the ISA of bhte_step2 shows: its loads are FLAT instructions -- they count on lgkmcnt as well, so every wait for an LDS read
// drains them -- and the material id and heat source of every cell are loaded inside that cell's branch and waited for at once:
// five to eight exposed memory round trips per plane, 71 VALU instructions per cell of which a third is address arithmetic, flag tests
// and queue rotation. Here:
//   * GLOBAL (saddr) loads: wave-uniform plane base in SGPRs + ONE 32-bit byte offset per thread. The region is 68 x 28 (outputs
//     64 x 24) so that a thread's four cells are rows r, r+7, r+14, r+21 of one column: the same offset register serves all of them
//     (the row stride goes into the scalar base), one LDS index with constant displacements;
//   * no branch around a load: the arrays carry pads in front and behind (bhte_run_core), so every cell of the region is
//     addressable whether it lies in the volume or not; what must not be used is masked by the per-cell flags when it is USED;
//   * T(n) of plane p+2, id and heat source of plane p+1 are issued in the MIDDLE of iteration p, after the first update has
//     consumed the registers they replace (no extra registers), and fly through the second update, the stores and the barrier; the
//     dose of plane p-1 is issued at the top and flies through the first update;
//   * the plane loop is unrolled three times with the queue slots as compile-time constants (no register moves).
constexpr int G2_W = 68, G2_H = 28, G2_TY = G2_H - 4, G2_T = 512, G2_NC = 4, G2_ROWS = 7, G2_ACT = G2_ROWS * G2_W, G2_CELLS = G2_W * G2_H;
#define BFD_GA __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ BFD_GA const T *gbase(const T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (BFD_GA const T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned gpin(unsigned v) { asm("" : "+v"(v)); return v; }
__device__ __forceinline__ float gl4(const float *b, unsigned ofs) { return *(BFD_GA const float *)((BFD_GA const char *)gbase(b) + ofs); }
__device__ __forceinline__ unsigned gl1(const unsigned char *b, unsigned ofs) { return *(BFD_GA const unsigned char *)((BFD_GA const unsigned char *)gbase(b) + ofs); }
__device__ __forceinline__ void gs4(float *b, unsigned ofs, float v) { *(BFD_GA float *)((BFD_GA char *)gbase((const float *)b) + ofs) = v; }
template <int K> struct Ph3 { static constexpr int v = K; };

// QM: 0 no heating in either step, 1 the same heat source in both, 2 anything else (qa / qb independent, either may be null)
template <bool REV, int QM>
__device__ __forceinline__ void bhte_step2g_body(int b, B2_ARGS)
{
    // one row of margin before and behind each tile: the ring cells read "neighbours" there (never used)
    __shared__ float A[2][G2_CELLS + 2 * G2_W], B[2][G2_CELLS + 2 * G2_W];
    __shared__ float2 sC[256];
    for (int m = threadIdx.x; m < nMat; m += G2_T) sC[m] = make_float2(cd[m], cp[m]);
#ifdef G2_EXP_LDS_PAD      // experiment: fewer workgroups per CU
    __shared__ float sPad[G2_EXP_LDS_PAD];
    if (nMat < 0) { sPad[threadIdx.x] = 1.f; sC[0].x = sPad[(threadIdx.x + 1) % 512]; }
#endif
    // 476 threads own the 4 x 476 region cells; the last 36 repeat the work of threads 0..35 without storing (same LDS values)
    const bool mirror = threadIdx.x >= G2_ACT;
    const int tid = mirror ? threadIdx.x - G2_ACT : threadIdx.x;
    const int bx = b % tilesX, by = (b / tilesX) % tilesY, bz = b / (tilesX * tilesY);
    const int x0 = bx * 64 - 2, y0 = by * G2_TY - 2, z0 = bz * zrun, z1 = min(z0 + zrun, N3);
    const long pl = (long)N1 * N2;
    const int r = tid / G2_W, rx = tid - r * G2_W, e0 = tid + G2_W;
    // byte offsets of the thread's cells from the region's corner: one VGPR per cell, so that an array needs one scalar base per plane
    unsigned g[G2_NC];
    #pragma unroll
    for (int n = 0; n < G2_NC; n++) g[n] = 4u * (unsigned)((r + G2_ROWS * n) * N1 + rx);
    // scalar address parts (cells): the region's corner inside a plane (may lie in the pads) and the row stride between a thread's cells
    const int corner = __builtin_amdgcn_readfirstlane(y0 * N1 + x0);
    // per cell: inside the volume / T(n+1) computed here / output cell / on an x or y face (loop-invariant lane masks)
    bool fC1[G2_NC], fOut[G2_NC], fFace[G2_NC];
    #pragma unroll
    for (int n = 0; n < G2_NC; n++) {
        const int ry = r + G2_ROWS * n, gi = x0 + rx, gj = y0 + ry;
        const bool in = gi >= 0 && gi < N1 && gj >= 0 && gj < N2;
        const bool face = gi == 0 || gi == N1 - 1 || gj == 0 || gj == N2 - 1;
        fC1[n] = in && !face && rx >= 1 && rx <= G2_W - 2 && ry >= 1 && ry <= G2_H - 2;
        fOut[n] = in && !mirror && rx >= 2 && rx <= G2_W - 3 && ry >= 2 && ry <= G2_H - 3;
        fFace[n] = in && face;
    }
    auto clampz = [&](int k) { return (long)__builtin_amdgcn_readfirstlane(min(max(k, 0), N3 - 1)) * pl + corner; };
    const bool heatA = QM == 1 || (QM == 2 && qa != nullptr), heatB = QM == 1 || (QM == 2 && qb != nullptr);
    const float *qaE = heatA ? qa : Tin, *qbE = heatB ? qb : Tin;                 // a valid address whatever the mode

    // queues, slot = (plane + const) mod 3: t0 = T(n) of planes p-1, p, p+1; t1 = T(n+1) of planes p-2, p-1, p; mi / qv = id and heat
    // source of planes p-1, p, p+1
    float t0[G2_NC][3], t1[G2_NC][3], qv[G2_NC][3];
    unsigned mi[G2_NC][3];
    const int p0 = z0 - 1;
    #pragma unroll
    for (int n = 0; n < G2_NC; n++) {
        asm volatile("" : "+v"(g[n]));
        const unsigned gp = g[n];
        // phase 0 of the first iteration: planes p0-1, p0, p0+1 in slots 2, 0, 1
        t0[n][2] = gl4(Tin + clampz(p0 - 1), gp);
        t0[n][0] = gl4(Tin + clampz(p0), gp);
        t0[n][1] = gl4(Tin + clampz(p0 + 1), gp);
        mi[n][0] = gl1(mat + clampz(p0), gp >> 2);
        qv[n][0] = QM ? gl4(qaE + clampz(p0), gp) : 0.0f;
        t1[n][0] = t1[n][1] = t1[n][2] = 0.0f; mi[n][1] = mi[n][2] = 0u; qv[n][1] = qv[n][2] = 0.0f;
    }
    __syncthreads();                                  // the coefficient table

    auto plane = [&](auto PH, const int p) {
        constexpr int c = decltype(PH)::v, m1 = (c + 2) % 3, p1 = (c + 1) % 3;      // slots of planes p, p-1, p+1 (t1: p-2 lives in p1)
        const int par = p & 1;
        float *Ap = A[par], *Bp = B[par]; const float *Bq = B[par ^ 1];
        const bool outs = p - 1 >= z0 && p - 1 < z1;
        const bool inner = p > 0 && p < N3 - 1, innerOut = p - 1 > 0 && p - 1 < N3 - 1;
        const long kO = clampz(p - 1);                                              // the plane of this iteration's outputs
        float dz[G2_NC], q2[G2_NC];
        // the offsets become values of THIS block (asm on the loop-carried registers themselves, no copy): hoisted out of the loop
        // their zero-extensions would turn into 64-bit VGPR addresses
        #pragma unroll
        for (int n = 0; n < G2_NC; n++) asm volatile("" : "+v"(g[n]));
        #pragma unroll
        for (int n = 0; n < G2_NC; n++) {
            dz[n] = gl4(dose + kO, g[n]);
            q2[n] = QM == 2 ? gl4(qbE + kO, g[n]) : 0.0f;
            Ap[e0 + n * G2_ACT] = t0[n][c];
        }
        __syncthreads();
        // ---- first update: T(n+1) of plane p ----
        #pragma unroll
        for (int n = 0; n < G2_NC; n++) {
            const int e = e0 + n * G2_ACT;
            const float2 cc = sC[mi[n][c]];
            float T1 = bhte_update<REV>(t0[n][c], Ap[e - 1], Ap[e + 1], Ap[e - G2_W], Ap[e + G2_W], t0[n][m1], t0[n][p1], cc.x, cc.y, Tcore, heatA, qv[n][c]);
#ifndef G2_BRANCHY
            asm volatile("" : "+v"(T1));            // computed by every lane: no branch around the update (the select below masks it)
#endif
            t1[n][c] = (inner && fC1[n]) ? T1 : t0[n][c];
            Bp[e] = t1[n][c];
        }
        // ---- what the next iteration's first update needs: T(n) of plane p+2 replaces plane p-1, id and heat source of plane p+1 ----
        const long kN = clampz(p + 1), kNN = clampz(p + 2);
        #pragma unroll
        for (int n = 0; n < G2_NC; n++) {
            mi[n][p1] = gl1(mat + kN, g[n] >> 2);
            if (QM) qv[n][p1] = gl4(qaE + kN, g[n]);
            t0[n][m1] = gl4(Tin + kNN, g[n]);
        }
        // ---- second update: T(n+2) of plane p-1, dose ----
        if (outs) {
            #pragma unroll
            for (int n = 0; n < G2_NC; n++) {
                const int e = e0 + n * G2_ACT;
                const float2 cc = sC[mi[n][m1]];
                const float qq = QM == 1 ? qv[n][m1] : q2[n];
                float U = bhte_update<REV>(t1[n][m1], Bq[e - 1], Bq[e + 1], Bq[e - G2_W], Bq[e + G2_W], t1[n][p1], t1[n][c], cc.x, cc.y, Tcore, heatB, qq);
#ifndef G2_BRANCHY
                asm volatile("" : "+v"(U));
#endif
                const float T2 = (innerOut && !fFace[n]) ? U : t1[n][m1];
                if (fOut[n]) {
                    gs4(Tout + kO, g[n], T2);
                    gs4(dose + kO, g[n], __fadd_rn(__fadd_rn(dz[n], bhte_dose_rate(t1[n][m1], dtMin)), bhte_dose_rate(T2, dtMin)));
                }
            }
        }
    };
    int p = p0;
    for (; p + 2 <= z1; p += 3) { plane(Ph3<0>(), p); plane(Ph3<1>(), p + 1); plane(Ph3<2>(), p + 2); }
    if (p <= z1) plane(Ph3<0>(), p);
    if (p + 1 <= z1) plane(Ph3<1>(), p + 1);
}

#ifndef G2_WAVES
#define G2_WAVES 6           // waves per SIMD the register budget is held to (6: 80 VGPRs, three workgroups per CU)
#endif
template <bool REV, int QM>
__global__ __launch_bounds__(G2_T, G2_WAVES) void bhte_step2g(B2_ARGS)
{
    int b = blockIdx.x;
    if (xcdOrder) {
        const int per = nBlocks >> 3, rem = nBlocks & 7, x = b & 7, slot = b >> 3;
        b = x * per + (x < rem ? x : rem) + slot;
    }
    bhte_step2g_body<REV, QM>(b, Tin, Tout, dose, qa, qb, mat, cd, cp, nMat, N1, N2, N3, Tcore, dtMin, zrun, tilesX, tilesY, nBlocks, xcdOrder);
}
// ---- round 6: S = 3 or 4 steps per pass (bhte_stepNg) ----
// Two steps per pass move T in / out, the dose in / out, the heat source and the ids once for two steps (10.5 B per voxel-step while heating); S steps
// move them once for S. Same machinery as bhte_step2g, one more level of it per step: a workgroup marches a z-run over a region of (64 + 2 S) x 28
// cells (outputs 64 x (28 - 2 S)); thread = one column position, four cells 7 rows apart (or two, 14 apart: GNCells); level k = T(n + k) lives in per-thread z-queues of three
// planes (compile-time slots, plane loop unrolled by three) and, for the in-plane neighbours, in a double-buffered LDS plane per level (one barrier
// per plane). In iteration p level 1 is computed for plane p, level 2 for plane p - 1, ..., the output level S for plane p - S + 1; a level-k value
// is computed where the cell lies k cells inside the region (ring cells copy the level below: never used by a cell that counts), face cells keep
// their temperature. The dose takes its S increments in step order: (((d + r(T1)) + r(T2)) + ...) -- T1 of the output plane has left its queue
// when S = 4 and is taken from the slot the new plane is about to overwrite. Heat source of the last S planes and their ids ride in small
// rotating queues (the ids as bytes of one register). QM: 0 no heating in any of the S steps, 1 the same field in all of them; mixed stretches
// take the two-step kernel. Same expressions in the same order: bit-identical to S launches of bhte_step.
// Cells per thread: four (512 threads per workgroup, rows 7 apart), or two (1024 threads, rows 14 apart). Two workgroups of 8 waves and one of 16 are
// the same 4 waves per SIMD and the same 128-register budget, but a thread with two cells carries half the queues: the four-step flavour WITH a heat
// source fits 82 registers with two cells (with four it needs 128 and spills 18 to scratch: 382-397 against 564-642 Gvoxel-steps/s), while the
// flavours that fit anyway are faster with four (cooling, S = 4: 728 / 805 against 651 / 734) -- so the heating flavour of S = 4 takes two, the rest four.
template <int QM, int S> struct GNCells { static constexpr int v = (QM == 1 && S == 4) ? 2 : 4; };
template <int S, int NCELLS> struct GN { static constexpr int W = 64 + 2 * S, H = 28, TY = H - 2 * S, NC = NCELLS, T = 2048 / NC, ROWS = H / NC, ACT = ROWS * W, CELLS = W * H; };
template <bool REV, int QM, int S>
__device__ __forceinline__ void bhte_stepNg_body(int b, B2_ARGS)
{
    using G = GN<S, GNCells<QM, S>::v>;
    static_assert(G::ACT <= G::T && S >= 3 && S <= 4, "region / thread mapping");
    __shared__ float L[S][2][G::CELLS + 2 * G::W];
    __shared__ float2 sC[256];
    for (int m = threadIdx.x; m < nMat; m += G::T) sC[m] = make_float2(cd[m], cp[m]);
    const bool mirror = threadIdx.x >= G::ACT;
    const int tid = mirror ? threadIdx.x - G::ACT : threadIdx.x;
    const int bx = b % tilesX, by = (b / tilesX) % tilesY, bz = b / (tilesX * tilesY);
    const int x0 = bx * 64 - S, y0 = by * G::TY - S, z0 = bz * zrun, z1 = min(z0 + zrun, N3);
    const long pl = (long)N1 * N2;
    const int r = tid / G::W, rx = tid - r * G::W, e0 = tid + G::W;
    unsigned g[G::NC];
    int dist[G::NC];
    bool fIn[G::NC], fFace[G::NC], fOut[G::NC];
    #pragma unroll
    for (int n = 0; n < G::NC; n++) {
        const int ry = r + G::ROWS * n, gi = x0 + rx, gj = y0 + ry;
        g[n] = 4u * (unsigned)(ry * N1 + rx);
        const bool in = gi >= 0 && gi < N1 && gj >= 0 && gj < N2;
        const bool face = gi == 0 || gi == N1 - 1 || gj == 0 || gj == N2 - 1;
        dist[n] = min(min(rx, G::W - 1 - rx), min(ry, G::H - 1 - ry));
        fIn[n] = in && !face;
        fFace[n] = in && face;
        fOut[n] = in && !mirror && dist[n] >= S;
    }
    const int corner = __builtin_amdgcn_readfirstlane(y0 * N1 + x0);
    auto clampz = [&](int k) { return (long)__builtin_amdgcn_readfirstlane(min(max(k, 0), N3 - 1)) * pl + corner; };
    const bool heat = QM == 1;
    const float *qE = heat ? qa : Tin;

    float t[S][G::NC][3];                 // level k (T(n + k)), planes by slot (plane - p0) mod 3
    float q[G::NC][S], qn[G::NC];         // heat source of planes p, p - 1, ..., p - S + 1; of plane p + 1 on its way
    unsigned mi[G::NC], mn[G::NC];        // ids of planes p .. p - 3 as bytes; of plane p + 1 on its way
    const int p0 = z0 - (S - 1);
    #pragma unroll
    for (int n = 0; n < G::NC; n++) {
        asm volatile("" : "+v"(g[n]));
        const unsigned gp = g[n];
        t[0][n][2] = gl4(Tin + clampz(p0 - 1), gp);
        t[0][n][0] = gl4(Tin + clampz(p0), gp);
        t[0][n][1] = gl4(Tin + clampz(p0 + 1), gp);
        mi[n] = gl1(mat + clampz(p0), gp >> 2);
        q[n][0] = QM ? gl4(qE + clampz(p0), gp) : 0.0f;
        #pragma unroll
        for (int k = 1; k < S; k++) { t[k][n][0] = t[k][n][1] = t[k][n][2] = 0.0f; q[n][k] = 0.0f; }
        qn[n] = 0.0f; mn[n] = 0u;
    }
    __syncthreads();

    auto plane = [&](auto PH, const int p) {
        constexpr int c = decltype(PH)::v;
        const int par = p & 1;
        const int o = p - S + 1;                                                    // the plane of this iteration's outputs
        const bool outs = o >= z0 && o < z1;
        const long kO = clampz(o);
        float dz[G::NC], t1old[G::NC];
        #pragma unroll
        for (int n = 0; n < G::NC; n++) asm volatile("" : "+v"(g[n]));
        #pragma unroll
        for (int n = 0; n < G::NC; n++) {
            dz[n] = gl4(dose + kO, g[n]);
            L[0][par][e0 + n * G::ACT] = t[0][n][c];
            t1old[n] = t[1][n][c];                                                  // S = 4: T(n + 1) of plane p - 3, about to be overwritten
        }
        __syncthreads();
        #pragma unroll
        for (int k = 1; k <= S; k++) {
            // level k of plane pk = p - k + 1 from level k - 1: planes pk - 1, pk, pk + 1 in the queue, plane pk in LDS
            constexpr int dummy = 0; (void)dummy;
            const int pk = p - k + 1;
            const bool innerz = pk > 0 && pk < N3 - 1;
            const float *Lp = k == 1 ? L[0][par] : L[k - 1][par ^ 1];
            #pragma unroll
            for (int n = 0; n < G::NC; n++) {
                const int e = e0 + n * G::ACT;
                const int sc = ((c - k + 1) % 3 + 3) % 3, sm = ((c - k) % 3 + 3) % 3, sp = ((c - k + 2) % 3 + 3) % 3;
                const float tc = t[k - 1][n][sc];
                const float2 cc = sC[(mi[n] >> (8 * (k - 1))) & 255u];
                float U = bhte_update<REV>(tc, Lp[e - 1], Lp[e + 1], Lp[e - G::W], Lp[e + G::W], t[k - 1][n][sm], t[k - 1][n][sp], cc.x, cc.y, Tcore, heat, q[n][k - 1]);
                asm volatile("" : "+v"(U));                                         // computed by every lane, selected below
                const float val = (innerz && fIn[n] && dist[n] >= k) ? U : tc;
                if (k < S) {
                    t[k][n][sc] = val;
                    L[k][par][e] = val;
                } else if (outs && fOut[n]) {
                    // dose: the increments of the S steps in step order
                    float dsum = dz[n];
                    #pragma unroll
                    for (int j = 1; j < S; j++) {
                        const int so = ((c - S + 1) % 3 + 3) % 3;                    // slot of plane o
                        const float Tj = (S == 4 && j == 1) ? t1old[n] : t[j][n][so];
                        dsum = __fadd_rn(dsum, bhte_dose_rate(Tj, dtMin));
                    }
                    gs4(Tout + kO, g[n], val);
                    gs4(dose + kO, g[n], __fadd_rn(dsum, bhte_dose_rate(val, dtMin)));
                }
            }
            if (k == 1) {
                // what the next iteration's level 1 needs: T(n) of plane p + 2 replaces plane p - 1, id and heat source of plane p + 1
                const long kN = clampz(p + 1), kNN = clampz(p + 2);
                #pragma unroll
                for (int n = 0; n < G::NC; n++) {
                    mn[n] = gl1(mat + kN, g[n] >> 2);
                    if (QM) qn[n] = gl4(qE + kN, g[n]);
                    t[0][n][(c + 2) % 3] = gl4(Tin + kNN, g[n]);
                }
            }
        }
        #pragma unroll
        for (int n = 0; n < G::NC; n++) {
            mi[n] = (mi[n] << 8) | mn[n];
            #pragma unroll
            for (int k = S - 1; k >= 1; k--) q[n][k] = q[n][k - 1];
            q[n][0] = qn[n];
        }
    };
    const int pend = z1 + S - 2;
    int p = p0;
    for (; p + 2 <= pend; p += 3) { plane(Ph3<0>(), p); plane(Ph3<1>(), p + 1); plane(Ph3<2>(), p + 2); }
    if (p <= pend) plane(Ph3<0>(), p);
    if (p + 1 <= pend) plane(Ph3<1>(), p + 1);
}

#ifndef GN_WAVES
#define GN_WAVES 4           // 128 registers: two workgroups of 8 waves per CU, or one of 16
#endif
template <bool REV, int QM, int S>
__global__ __launch_bounds__((GN<S, GNCells<QM, S>::v>::T), GN_WAVES) void bhte_stepNg(B2_ARGS)
{
    int b = blockIdx.x;
    if (xcdOrder) {
        const int per = nBlocks >> 3, rem = nBlocks & 7, x = b & 7, slot = b >> 3;
        b = x * per + (x < rem ? x : rem) + slot;
    }
    bhte_stepNg_body<REV, QM, S>(b, Tin, Tout, dose, qa, qb, mat, cd, cp, nMat, N1, N2, N3, Tcore, dtMin, zrun, tilesX, tilesY, nBlocks, xcdOrder);
}

// Monitor points of the intermediate steps of an S-step pass: T(n + depth) at the listed voxels from T(n), depth 1 .. 3 -- the cube of side
// 2 depth + 1 around the voxel advanced level by level in LDS (one workgroup per point; the caller monitors 1 - 4 points,
// CalculateTemperatureEffects.py:1003-1023). The same cell update as everywhere: equal to the value a one-step run would hold.
template <bool REV>
__global__ __launch_bounds__(64) void cone_points(const float *__restrict__ Tin, const float *__restrict__ q, const unsigned char *__restrict__ mat,
                                                  const float *__restrict__ cd, const float *__restrict__ cp, int N1, int N2, int N3, float Tcore,
                                                  const unsigned *__restrict__ idx, float *__restrict__ out, long stride, long col, int depth)
{
    __shared__ float A[2][343];
    const int side = 2 * depth + 1, ncell = side * side * side;
    const unsigned c0 = idx[blockIdx.x];
    const int ci = (int)(c0 % (unsigned)N1), cj = (int)((c0 / (unsigned)N1) % (unsigned)N2), ck = (int)(c0 / ((unsigned)N1 * (unsigned)N2));
    const long pl = (long)N1 * N2;
    for (int v = threadIdx.x; v < ncell; v += 64) {
        const int li = v % side, lj = (v / side) % side, lk = v / (side * side);
        const int i = min(max(ci - depth + li, 0), N1 - 1), j = min(max(cj - depth + lj, 0), N2 - 1), k = min(max(ck - depth + lk, 0), N3 - 1);
        A[0][v] = Tin[(long)k * pl + (long)j * N1 + i];
    }
    __syncthreads();
    for (int lev = 1; lev <= depth; lev++) {
        const float *src = A[(lev - 1) & 1]; float *dst = A[lev & 1];
        for (int v = threadIdx.x; v < ncell; v += 64) {
            const int li = v % side, lj = (v / side) % side, lk = v / (side * side);
            float val = src[v];
            if (li >= lev && li < side - lev && lj >= lev && lj < side - lev && lk >= lev && lk < side - lev) {
                const int i = ci - depth + li, j = cj - depth + lj, k = ck - depth + lk;
                if (i > 0 && i < N1 - 1 && j > 0 && j < N2 - 1 && k > 0 && k < N3 - 1) {       // inside the volume and off its faces: all six neighbours exist
                    const long c = (long)k * pl + (long)j * N1 + i;
                    const int m = mat[c];
                    val = bhte_update<REV>(src[v], src[v - 1], src[v + 1], src[v - side], src[v + side], src[v - side * side], src[v + side * side], cd[m], cp[m], Tcore,
                                           q != nullptr, q ? q[c] : 0.0f);
                }
            }
            dst[v] = val;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(long)blockIdx.x * stride + col] = A[depth & 1][(depth * side + depth) * side + depth];
}

// The monitored plane at a step INSIDE a pass: T(n + depth) on the row j = jsel for a patch of 16 x 16 cells in (i, k), from T(n) -- the slab of
// (16 + 2 depth) x (2 depth + 1) x (16 + 2 depth) cells around the patch advanced level by level in LDS, like cone_points does for a cube. One sample of
// a 320 x 320 plane at depth 3 costs 400 workgroups x 3388 cells: a few per cent of one volume step, once per nFactorMonitoring steps.
template <bool REV>
__global__ __launch_bounds__(256) void cone_slice(const float *__restrict__ Tin, const float *__restrict__ q, const unsigned char *__restrict__ mat,
                                                  const float *__restrict__ cd, const float *__restrict__ cp, int N1, int N2, int N3, float Tcore,
                                                  float *__restrict__ out, int jsel, long sample, long nSamples, int depth)
{
    __shared__ float A[2][22 * 7 * 22];
    const int sx = 16 + 2 * depth, sy = 2 * depth + 1, sz = 16 + 2 * depth, ncell = sx * sy * sz;
    const int i0 = blockIdx.x * 16 - depth, j0 = jsel - depth, k0 = blockIdx.y * 16 - depth;
    const long pl = (long)N1 * N2;
    for (int v = threadIdx.x; v < ncell; v += 256) {
        const int li = v % sx, lj = (v / sx) % sy, lk = v / (sx * sy);
        const int i = min(max(i0 + li, 0), N1 - 1), j = min(max(j0 + lj, 0), N2 - 1), k = min(max(k0 + lk, 0), N3 - 1);
        A[0][v] = Tin[(long)k * pl + (long)j * N1 + i];
    }
    __syncthreads();
    for (int lev = 1; lev <= depth; lev++) {
        const float *src = A[(lev - 1) & 1]; float *dst = A[lev & 1];
        for (int v = threadIdx.x; v < ncell; v += 256) {
            const int li = v % sx, lj = (v / sx) % sy, lk = v / (sx * sy);
            float val = src[v];
            if (li >= lev && li < sx - lev && lj >= lev && lj < sy - lev && lk >= lev && lk < sz - lev) {
                const int i = i0 + li, j = j0 + lj, k = k0 + lk;
                if (i > 0 && i < N1 - 1 && j > 0 && j < N2 - 1 && k > 0 && k < N3 - 1) {
                    const long c = (long)k * pl + (long)j * N1 + i;
                    const int m = mat[c];
                    val = bhte_update<REV>(src[v], src[v - 1], src[v + 1], src[v - sx], src[v + sx], src[v - sx * sy], src[v + sx * sy], cd[m], cp[m], Tcore,
                                           q != nullptr, q ? q[c] : 0.0f);
                }
            }
            dst[v] = val;
        }
        __syncthreads();
    }
    const float *res = A[depth & 1];
    const int li = depth + (int)(threadIdx.x & 15), lk = depth + (int)(threadIdx.x >> 4);
    const int i = i0 + li, k = k0 + lk;
    if (i < N1 && k < N3) out[(REV ? (long)k * N1 + i : (long)i * N3 + k) * nSamples + sample] = res[(lk * sy + depth) * sx + li];
}

// Monitors of the first of two fused steps: T(n+1) at the listed voxels / on the monitored plane, computed from T(n)
template <bool REV>
__global__ void step_points(const float *__restrict__ Tin, const float *__restrict__ q, const unsigned char *__restrict__ mat,
                            const float *__restrict__ cd, const float *__restrict__ cp, int N1, int N2, int N3, float Tcore,
                            const unsigned *__restrict__ idx, float *__restrict__ out, long n, long stride, long col)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const unsigned c = idx[t];
    const int i = (int)(c % (unsigned)N1), j = (int)((c / (unsigned)N1) % (unsigned)N2), k = (int)(c / ((unsigned)N1 * (unsigned)N2));
    out[t * stride + col] = bhte_cell<REV>(Tin, q, mat, cd, cp, i, j, k, N1, N2, N3, Tcore);
}
template <bool REV>
__global__ void step_slice(const float *__restrict__ Tin, const float *__restrict__ q, const unsigned char *__restrict__ mat,
                           const float *__restrict__ cd, const float *__restrict__ cp, int N1, int N2, int N3, float Tcore,
                           float *__restrict__ out, int jsel, long sample, long nSamples)
{
    const long n = (long)N1 * N3;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x) {
        const int i = (int)(v % N1), k = (int)(v / N1);
        out[(REV ? (long)k * N1 + i : (long)i * N3 + k) * nSamples + sample] = bhte_cell<REV>(Tin, q, mat, cd, cp, i, jsel, k, N1, N2, N3, Tcore);
    }
}

__global__ void gather_points(const float *__restrict__ T, const unsigned *__restrict__ idx, float *__restrict__ out, long n, long stride, long col)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t * stride + col] = T[idx[t]];
}
template <bool REV>
__global__ void gather_slice(const float *__restrict__ T, float *__restrict__ out, int N1, int N2, int N3, int jsel, long sample, long nSamples)
{   // out[(i*N3 + k)*nSamples + sample]; REV: out[(k*N1 + i)*nSamples + sample] (slow axis first, the caller's C order)
    const long n = (long)N1 * N3;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x) {
        const int i = (int)(v % N1), k = (int)(v / N1);
        out[(REV ? (long)k * N1 + i : (long)i * N3 + k) * nSamples + sample] = T[(long)k * N1 * N2 + (long)jsel * N1 + i];
    }
}

__global__ void table_lookup(const unsigned char *__restrict__ mat, const float *__restrict__ tab, float *__restrict__ out, size_t n)
{
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (size_t)gridDim.x * blockDim.x) out[v] = tab[mat[v]];
}
// heat increment of one ON step from the pressure amplitude: q = (p p) qf[m], float32 with the oracle's roundings
__global__ void heat_source(const float *p, const unsigned char *__restrict__ mat, const float *__restrict__ qf, float *q, size_t n)      // q may be p
{
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (size_t)gridDim.x * blockDim.x) {
        const float a = p[v];
        q[v] = __fmul_rn(__fmul_rn(a, a), qf[mat[v]]);
    }
}

}  // namespace

// The run behind all entry points. F, M, S: extents of the fastest, middle and slowest axis of the volumes as they lie in memory.
// q (host, nFields volumes) or, if null, pressure + qf: the heat increments are then computed on the device (and copied to qOut).
// initT (per material) replaces the upload of T when flags bit 0 is clear; the dose starts from zero when bit 1 is clear.
template <bool REV>
static int bhte_run_core(int32_t device, int32_t F, int32_t M, int32_t S, int32_t nMat, const unsigned char *mat, const float *cd, const float *cp,
                         const float *qf, const float *initT, int32_t nFields, const float *q, const float *pressure, float *qOut, float *T, float *dose,
                         int32_t flags, float Tcore, double dt, int32_t nSteps, const int32_t *fieldOfStep, int32_t sliceJ, int32_t nFactorMonitoring,
                         float *monitorSlice, int64_t nPoints, const uint32_t *pointIndex, float *points, double *kernelMs)
{
    if (F < 3 || M < 3 || S < 3 || nMat < 1 || nMat > 256 || nFields < 1 || !mat || !cd || !cp || (!q && !(pressure && qf)) || !T || !dose || nSteps < 0 ||
        (nSteps > 0 && !fieldOfStep) || (!(flags & 1) && !initT)) {
        bfd_set_error("bfd_bhte_run: bad argument"); return -1;
    }
    for (int s = 0; s < nSteps; s++)
        if (fieldOfStep[s] < -1 || fieldOfStep[s] >= nFields) { bfd_set_error("bfd_bhte_run: fieldOfStep entry out of range"); return -1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { bfd_set_error("bfd_bhte_run: no HIP device available (no CPU fallback)"); return -3; }
    if (device < 0 || device >= ndev) { bfd_set_error("bfd_bhte_run: device ordinal out of range"); return -3; }
    BFD_HIP(hipSetDevice(device));
    const int N1 = F, N2 = M, N3 = S;                            // the kernels' names: x fastest
    const size_t n = (size_t)N1 * N2 * N3;
    if (sliceJ >= N2) { bfd_set_error("bfd_bhte_run: monitored plane outside the volume"); return -1; }
    for (int64_t t = 0; t < nPoints && pointIndex && points; t++)
        if (pointIndex[t] >= n) { bfd_set_error("bfd_bhte_run: monitored point outside the volume"); return -1; }
    const int fm = nFactorMonitoring > 0 ? nFactorMonitoring : 1;
    const long nSamples = (monitorSlice && sliceJ >= 0) ? (nSteps + fm - 1) / fm : 0;
    float *dT[2] = {nullptr, nullptr}, *dDose = nullptr, *dq = nullptr, *dcd = nullptr, *dcp = nullptr, *dqf = nullptr, *dSlice = nullptr, *dPts = nullptr;
    unsigned char *dmat = nullptr; unsigned *dIdx = nullptr;
    std::vector<void *> allocs;
    auto A = [&](void **p, size_t bytes) { hipError_t e = hipMalloc(p, bytes ? bytes : 1); if (e == hipSuccess) allocs.push_back(*p); return e; };
    // the volumes carry pads (elements): bhte_step2g addresses every cell of its 68 x 28 regions, also those that hang over the faces
    const size_t padF = (((size_t)4 * N1 + 64 + 255) / 256) * 256, padB = (size_t)49 * N1 + 256;       // bhte_stepNg: 72 x 28 regions, 4 rows / cells in front
    auto AP = [&](void **p, size_t elems, size_t elemBytes) {
        void *raw = nullptr; const hipError_t e = A(&raw, (padF + elems + padB) * elemBytes);
        if (e == hipSuccess) *p = (char *)raw + padF * elemBytes;
        return e;
    };
    hipError_t e = AP((void **)&dT[0], n, 4);
    if (e == hipSuccess) e = AP((void **)&dT[1], n, 4);
    if (e == hipSuccess) e = AP((void **)&dDose, n, 4);
    if (e == hipSuccess) e = AP((void **)&dq, n * (size_t)nFields, 4);
    if (e == hipSuccess) e = AP((void **)&dmat, n, 1);
    if (e == hipSuccess) e = A((void **)&dcd, nMat * 4);
    if (e == hipSuccess) e = A((void **)&dcp, nMat * 4);
    if (e == hipSuccess) e = A((void **)&dqf, nMat * 4);
    if (e == hipSuccess && nSamples) e = A((void **)&dSlice, (size_t)N1 * N3 * nSamples * 4);
    if (e == hipSuccess && nPoints && points) { e = A((void **)&dIdx, nPoints * 4); if (e == hipSuccess) e = A((void **)&dPts, (size_t)nPoints * nSteps * 4); }
    if (e == hipSuccess) e = hipMemcpy(dmat, mat, n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dcd, cd, nMat * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dcp, cp, nMat * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (flags & 1) e = hipMemcpy(dT[0], T, n * 4, hipMemcpyHostToDevice);
        else {                                                                   // T = initT[material], via the table slot of qf
            e = hipMemcpy(dqf, initT, nMat * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) { hipLaunchKernelGGL(table_lookup, dim3(2048), dim3(256), 0, 0, dmat, dqf, dT[0], n); e = hipDeviceSynchronize(); }
        }
    }
    if (e == hipSuccess) e = (flags & 2) ? hipMemcpy(dDose, dose, n * 4, hipMemcpyHostToDevice) : hipMemset(dDose, 0, n * 4);
    if (e == hipSuccess) {
        if (q) e = hipMemcpy(dq, q, n * 4 * (size_t)nFields, hipMemcpyHostToDevice);
        else {                                                                   // q = (p p) qf[material], in place
            e = hipMemcpy(dqf, qf, nMat * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(dq, pressure, n * 4 * (size_t)nFields, hipMemcpyHostToDevice);
            for (int f = 0; f < nFields && e == hipSuccess; f++) hipLaunchKernelGGL(heat_source, dim3(2048), dim3(256), 0, 0, dq + (size_t)f * n, dmat, dqf, dq + (size_t)f * n, n);
            if (e == hipSuccess && qOut) e = hipMemcpy(qOut, dq, n * 4 * (size_t)nFields, hipMemcpyDeviceToHost);
        }
    }
    if (e == hipSuccess && dIdx) e = hipMemcpy(dIdx, pointIndex, nPoints * 4, hipMemcpyHostToDevice);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int cur = 0;
    if (e == hipSuccess) {
        hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0);
        const dim3 block(64, 4, 1), grid((N1 + 63) / 64, (N2 + 3) / 4, N3);
        const float dtMin = (float)(dt / 60.0);
        // two steps per launch wherever two are left (BFD_BHTE_FUSE=0: every step on its own); the monitors of the first of the
        // two are computed from T(n) at the monitored cells only
        const char *ev = getenv("BFD_BHTE_FUSE");
        const bool fuse = !(ev && atoi(ev) == 0);
        // z-runs of 8..24 planes (every run re-reads 4 planes of T and recomputes 2 of the intermediate step), short enough for
        // ~2900 workgroups (768 fit the chip at once). Measured at 384^3 (profiles/r3/bhte_two_steps_per_launch.txt): runs of 8 / 12 /
        // 16 / 24 planes -> 381 / 400 / 372 / 365 Gvoxel-steps/s; 372-378 -> 386-389 with the XCD-contiguous order; not kept: loads
        // issued a plane ahead, 8 waves/SIMD, items handed out at run time to resident workgroups (81 VGPRs: 320)
        ev = getenv("BFD_BHTE_KERNEL");
        const bool gform = !(ev && atoi(ev) == 1);                               // 1: the round-3 kernel (bhte_step2, 64 x 26 tiles)
        const int tileY = gform ? G2_TY : B2_TY;
        const int tilesX = (N1 + 63) / 64, tilesY = (N2 + tileY - 1) / tileY;
        ev = getenv("BFD_BHTE_ZRUN");
        int zrun = (ev && atoi(ev) > 0) ? atoi(ev) : 0;
        if (!zrun && !gform) { const int runs = (2880 + tilesX * tilesY - 1) / (tilesX * tilesY); zrun = (N3 + runs - 1) / runs; zrun = zrun < 8 ? 8 : zrun > 24 ? 24 : zrun; }
        // bhte_step2g runs at the memory system's rate with one to three workgroups per CU alike (profiles/r4/bhte_two_step_kernel_rewrite.txt),
        // so a CU's time is the number of workgroups it gets times the planes each of them marches (its run + 2): pick the run length
        // that minimises ceil(workgroups / 256) x (zrun + 2). The measured order of run lengths follows this count at 256^3, 384^3 and 512^3.
        if (!zrun) {
            long best = -1;
            for (int z = 8; z <= 64; z++) {
                const long w = (long)tilesX * tilesY * ((N3 + z - 1) / z), cost = ((w + 255) / 256) * (z + 2);
                if (best < 0 || cost <= best) { best = cost; zrun = z; }
            }
        }
        ev = getenv("BFD_BHTE_XCD_ORDER");
        const int xcdOrder = (ev && atoi(ev) == 0) ? 0 : 1;
        const int runsZ = (N3 + zrun - 1) / zrun;
        const long nBlocks2 = (long)tilesX * tilesY * runsZ;
        auto Q = [&](int f) { return f < 0 ? (const float *)nullptr : dq + (size_t)f * n; };
        auto monitors = [&](int s) {
            if (dPts) hipLaunchKernelGGL(gather_points, dim3((unsigned)((nPoints + 255) / 256)), dim3(256), 0, 0, dT[cur], dIdx, dPts, (long)nPoints, (long)nSteps, (long)s);
            if (dSlice && s % fm == 0) hipLaunchKernelGGL(gather_slice<REV>, dim3(256), dim3(256), 0, 0, dT[cur], dSlice, N1, N2, N3, sliceJ, (long)(s / fm), nSamples);
        };
        // S steps per pass (round 6; BFD_BHTE_STEPS=2 keeps two): where the next S steps carry the same heat field (or none). Monitors of the steps
        // inside a pass are recomputed from the pass's input: the points by cone_points, a sample of the monitored plane by step_slice (first step) or
        // cone_slice (steps in between); the last step's are read off the result.
        // BFD_BHTE_STEPS=2 / 3 / 4 forces one pass length; default: FOUR steps per pass. Gvoxel-steps/s at 320^3 / 512^3 (scripts/r6/bhte_phases.sh, bhte_nc_ab.sh,
        // profiles/r6/bhte_steps_per_pass.txt): nothing heats -- two steps 470 / 515, three 568 / 679, four 710-728 / 795-810 (four cells per thread: 99-128 registers,
        // 2 spilled); a field heats -- two 374 / 395, three 487 / 568, four with four cells per thread 382-397 / 380-403 (its heat-source queue spills 18 registers
        // to scratch), four with TWO cells per thread 564 / 642 (82 registers, none spilled: GNCells)
        ev = getenv("BFD_BHTE_STEPS");
        const int forced = (ev && atoi(ev) >= 2 && atoi(ev) <= 4) ? atoi(ev) : 0;
        int stepsHeat = forced ? forced : BFD_BHTE_STEPS_HEATING, stepsCool = forced ? forced : BFD_BHTE_STEPS_COOLING;
        if (!gform || !fuse) stepsHeat = stepsCool = 2;
        struct PassGeom { int tileY, tilesY, zrun; long nBlocks; } geom[5] = {};
        for (int S = 3; S <= 4; S++) {
            PassGeom &G = geom[S];
            G.tileY = 28 - 2 * S; G.tilesY = (N2 + G.tileY - 1) / G.tileY;
            ev = getenv("BFD_BHTE_ZRUN");
            G.zrun = (ev && atoi(ev) > 0) ? atoi(ev) : 0;
            if (!G.zrun) {
                long best = -1;
                for (int z = 8; z <= 96; z++) {
                    const long w = (long)tilesX * G.tilesY * ((N3 + z - 1) / z), cost = ((w + 255) / 256) * (z + 2 * S - 2);
                    if (best < 0 || cost <= best) { best = cost; G.zrun = z; }
                }
            }
            G.nBlocks = (long)tilesX * G.tilesY * ((N3 + G.zrun - 1) / G.zrun);
        }
        for (int s = 0; s < nSteps;) {
            const int stepsN = fieldOfStep[s] >= 0 ? stepsHeat : stepsCool;
            const PassGeom &G = geom[stepsN >= 3 ? stepsN : 3];
            bool passN = stepsN >= 3 && s + stepsN <= nSteps && G.nBlocks < 0x7fffffffL;
            for (int j = 1; j < stepsN && passN; j++) passN = fieldOfStep[s + j] == fieldOfStep[s];
            if (passN) {
                const float *qa = Q(fieldOfStep[s]);
                if (dPts) {
                    hipLaunchKernelGGL(cone_points<REV>, dim3((unsigned)nPoints), dim3(64), 0, 0, dT[cur], qa, dmat, dcd, dcp, N1, N2, N3, Tcore, dIdx, dPts, (long)nSteps, (long)s, 1);
                    for (int j = 2; j < stepsN; j++)
                        hipLaunchKernelGGL(cone_points<REV>, dim3((unsigned)nPoints), dim3(64), 0, 0, dT[cur], qa, dmat, dcd, dcp, N1, N2, N3, Tcore, dIdx, dPts, (long)nSteps, (long)(s + j - 1), j);
                }
                if (dSlice && s % fm == 0) hipLaunchKernelGGL(step_slice<REV>, dim3(256), dim3(256), 0, 0, dT[cur], qa, dmat, dcd, dcp, N1, N2, N3, Tcore, dSlice, sliceJ, (long)(s / fm), nSamples);
                for (int j = 1; j + 1 < stepsN && dSlice; j++)           // samples of the monitored plane at the steps inside the pass
                    if ((s + j) % fm == 0)
                        hipLaunchKernelGGL(cone_slice<REV>, dim3((unsigned)((N1 + 15) / 16), (unsigned)((N3 + 15) / 16)), dim3(256), 0, 0, dT[cur], qa, dmat, dcd, dcp, N1, N2, N3, Tcore,
                                           dSlice, sliceJ, (long)((s + j) / fm), nSamples, j + 1);
#define BNG_LAUNCH(QM, SS) hipLaunchKernelGGL((bhte_stepNg<REV, QM, SS>), dim3((unsigned)G.nBlocks), dim3(GN<SS, GNCells<QM, SS>::v>::T), 0, 0, dT[cur], dT[1 - cur], dDose, qa, qa, dmat, dcd, dcp, nMat, N1, N2, N3, \
                                              Tcore, dtMin, G.zrun, tilesX, G.tilesY, (int)G.nBlocks, xcdOrder)
                if (stepsN == 4) { if (qa) BNG_LAUNCH(1, 4); else BNG_LAUNCH(0, 4); }
                else { if (qa) BNG_LAUNCH(1, 3); else BNG_LAUNCH(0, 3); }
                cur = 1 - cur; s += stepsN;
                monitors(s - 1);
                continue;
            }
            if (fuse && s + 1 < nSteps && nBlocks2 < 0x7fffffffL) {
                const float *qa = Q(fieldOfStep[s]), *qb = Q(fieldOfStep[s + 1]);
                if (dPts) hipLaunchKernelGGL(step_points<REV>, dim3((unsigned)((nPoints + 255) / 256)), dim3(256), 0, 0, dT[cur], qa, dmat, dcd, dcp, N1, N2, N3, Tcore, dIdx, dPts, (long)nPoints, (long)nSteps, (long)s);
                if (dSlice && s % fm == 0) hipLaunchKernelGGL(step_slice<REV>, dim3(256), dim3(256), 0, 0, dT[cur], qa, dmat, dcd, dcp, N1, N2, N3, Tcore, dSlice, sliceJ, (long)(s / fm), nSamples);
#define B2G_LAUNCH(QM) hipLaunchKernelGGL((bhte_step2g<REV, QM>), dim3((unsigned)nBlocks2), dim3(G2_T), 0, 0, dT[cur], dT[1 - cur], dDose, qa, qb, dmat, dcd, dcp, nMat, N1, N2, N3, \
                                          Tcore, dtMin, zrun, tilesX, tilesY, (int)nBlocks2, xcdOrder)
                if (gform) { if (!qa && !qb) B2G_LAUNCH(0); else if (qa == qb) B2G_LAUNCH(1); else B2G_LAUNCH(2); }
                else hipLaunchKernelGGL(bhte_step2<REV>, dim3((unsigned)nBlocks2), dim3(B2_T), 0, 0, dT[cur], dT[1 - cur], dDose, qa, qb, dmat, dcd, dcp, nMat, N1, N2, N3, Tcore, dtMin,
                                        zrun, tilesX, tilesY, (int)nBlocks2, xcdOrder);
                cur = 1 - cur; s += 2;
                monitors(s - 1);
            } else {
                hipLaunchKernelGGL(bhte_step<REV>, grid, block, 0, 0, dT[cur], dT[1 - cur], dDose, Q(fieldOfStep[s]), dmat, dcd, dcp, N1, N2, N3, Tcore, dtMin);
                cur = 1 - cur; s += 1;
                monitors(s - 1);
            }
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
    if (!q) { bfd_set_error("bfd_bhte_run: bad argument"); return -1; }
    return bhte_run_core<false>(device, N1, N2, N3, nMat, mat, cd, cp, nullptr, nullptr, nFields, q, nullptr, nullptr, T, dose, 3, Tcore, dt, nSteps, fieldOfStep,
                                sliceJ, nFactorMonitoring, monitorSlice, nPoints, pointIndex, points, kernelMs);
}

// The same run on volumes in the CALLER'S numpy C order -- [N1][N2][N3], the last axis fastest -- so that the host transposes
// nothing (at 384^3 the transposes and the float32 products of the heat source were 2 s of a 2.3 s call whose kernels take
// 0.03 s), and with the heat source computed on the device: q = (p p) qf[material] from the pressure amplitude(s) `pressure`
// (nFields volumes, float32 Pa) and the per-material factor qf; qOut (may be NULL) receives it. The neighbours are summed
// axis 0 first, like the oracle does on such arrays. flags bit 0: T holds the initial temperature (else initT[material]);
// bit 1: dose holds the initial dose (else zero). monitorSlice: [N1][N3][nSamples] = T[:, sliceJ, :]; pointIndex: C-order
// linear indices (i*N2 + j)*N3 + k.
extern "C" int bfd_bhte_run_volumes(int32_t device, int32_t N1, int32_t N2, int32_t N3, int32_t nMat, const unsigned char *mat,
                                    const float *cd, const float *cp, const float *qf, const float *initT, int32_t nFields,
                                    const float *pressure, float *qOut, float *T, float *dose, int32_t flags, float Tcore, double dt,
                                    int32_t nSteps, const int32_t *fieldOfStep, int32_t sliceJ, int32_t nFactorMonitoring,
                                    float *monitorSlice, int64_t nPoints, const uint32_t *pointIndex, float *points, double *kernelMs)
{
    if (!pressure || !qf) { bfd_set_error("bfd_bhte_run_volumes: bad argument"); return -1; }
    return bhte_run_core<true>(device, N3, N2, N1, nMat, mat, cd, cp, qf, initT, nFields, nullptr, pressure, qOut, T, dose, flags, Tcore, dt, nSteps, fieldOfStep,
                               sliceJ, nFactorMonitoring, monitorSlice, nPoints, pointIndex, points, kernelMs);
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
