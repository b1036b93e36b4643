/*
 * fdtd_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the algorithm that BabelBrain's Step-2 driver calls as
 *   PModel.StaggeredFDTD_3D_with_relaxation(...)     /root/reference/TranscranialModeling/BabelIntegrationBASE.py:2338,2374,2401
 *   PModel.CalculateMatricesForPropagation(...)      BabelIntegrationBASE.py:1799,1801
 * The implementation of those two calls lives in the third-party package
 * `BabelViscoFDTD` (pinned ==1.2.4, environment_linux.yml:44), which is NOT present
 * under /root/reference and cannot be installed here.
 *
 * PARITY UNPINNED: nothing in the reference snapshot holds a golden pressure field or a
 * runnable solver, so this file restates the *published* scheme of that package
 * (README.md:24 of the reference: "isotropic viscoelastic ... O(2) time / O(4) space
 * staggered grid FDTD with PML"; BabelIntegrationBASE.py:1612) from textbook pieces:
 *   - Virieux/Levander velocity-stress staggered grid, O(2,4) leapfrog
 *   - one standard-linear-solid relaxation mechanism per modulus (tau-method memory variables,
 *     Robertsson/Blanch/Bohlen), relaxation frequency = central frequency
 *   - 12-cell absorbing layer with design reflection R=1e-5 (BASE:1628-1629), realised here
 *     as an unsplit convolutional PML (recursive-convolution memory variables)
 *   - sources / sensors / RMS-peak accumulation with the caller-visible semantics of
 *     BASE:2325-2365, 2433-2456, 2503-2518.
 * It is checked by the physics known-answer tests in tests/test_oracle_physics.py and is
 * the arithmetic the HIP engine must reproduce (tests/test_parity_gpu.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Canonical arithmetic: float32, denormals flushed to zero, no FMA contraction (build with
 * -ffp-contract=off), the operation order written below. The HIP engine follows the same order so both agree
 * to rounding (bit-exact in practice).
 *
 * Memory layout: "x-fastest": linear index = i + N1*(j + N2*k)  -- the same order the
 * reference decodes IndexSensorMap with (BASE:2508-2511).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BFO_EXPORT __attribute__((visibility("default")))

/* selectable maps, bit positions (shared convention with include/babelfdtd.h) */
enum { M_VX = 0, M_VY, M_VZ, M_SXX, M_SYY, M_SZZ, M_SXY, M_SXZ, M_SYZ, M_PRESSURE, M_ALLV, M_COUNT };

typedef struct {
    int32_t N1, N2, N3;
    int32_t nMat;
    int32_t NDelta;
    int32_t nt;
    int32_t typeSource;      /* 0 add velocity, 1 set velocity, 2 add stress, 3 set stress */
    int32_t lengthSource;    /* columns of PulseSource */
    int32_t nSources;        /* rows of PulseSource */
    int32_t sensorSub;       /* SensorSubSampling */
    int32_t sensorStart;     /* SensorStart (in sub-sampled units) */
    int32_t selRMSorPeak;    /* 1 RMS, 2 peak, 3 both */
    uint32_t selMapsRMS;     /* bitmask of M_* */
    uint32_t selMapsSensors; /* bitmask of M_* */
    int32_t qfactorCorrection;
    int32_t nthreads;        /* 0 = OpenMP default */
    double h, dt, freq, reflectionLimit;
} bfo_params;

typedef struct {
    /* per material, float32 */
    float *AP, *BP, *AS2, *BS2, *invMu, *tauS, *invRho;
    float c1, k2;
    double cmax;    /* fastest (unrelaxed) velocity, for CFL and PML */
} bfo_tables;

/* ------------------------------------------------------------------------------------------
 * Material coefficients (float64 maths, rounded once to float32).
 * Row layout of MaterialList: [rho, cL, cS, alphaL, alphaS] (BASE:1712-1729), alpha in Np/m at
 * `freq`. qcorr[m] multiplies the quality factor of material m (QCorrection, BASE:1290,2362).
 *
 * Single SLS with tau_sigma = 1/omega:   M(omega) = MR * [ (1+tau/2) + i tau/2 ].
 * qfactorCorrection != 0: tau and MR are solved so that, at `freq`, the continuum medium has
 *   exactly the requested attenuation alpha/q and exactly the requested phase velocity c.
 * qfactorCorrection == 0: low-loss forms tau = 2/Q, MR = rho c^2.
 * ---------------------------------------------------------------------------------------- */
static void sls_fit(double rho, double c, double alpha, double q, double omega, int corr,
                    double *MR, double *tau)
{
    if (c <= 0.0) { *MR = 0.0; *tau = 0.0; return; }
    if (alpha <= 0.0) { *MR = rho * c * c; *tau = 0.0; return; }
    double a = alpha / q;
    if (corr) {
        double x = a * c / omega;           /* tan(theta/2) */
        if (x > 0.4) x = 0.4;               /* keep theta < 45 deg */
        double theta = 2.0 * atan(x);
        double tt = tan(theta);
        double t = 2.0 * tt / (1.0 - tt);
        double re = 1.0 + 0.5 * t, im = 0.5 * t;
        double mag = sqrt(re * re + im * im);
        double ch = cos(0.5 * theta);
        *tau = t;
        *MR = rho * c * c * ch * ch / mag;
    } else {
        double Q = omega / (2.0 * c * a);
        if (Q < 1.5) Q = 1.5;
        *tau = 2.0 / Q;
        *MR = rho * c * c;
    }
}

static int build_tables(const bfo_params *p, const double *matlist, const double *qcorr, bfo_tables *t)
{
    int n = p->nMat;
    double omega = 2.0 * M_PI * p->freq;
    double tauSigma = 1.0 / omega;
    double dtoh = p->dt / p->h;
    double half = p->dt / (2.0 * tauSigma);
    double k2 = (p->dt / tauSigma) / (1.0 + half);
    t->c1 = (float)((1.0 - half) / (1.0 + half));
    t->k2 = (float)k2;
    t->AP = (float *)calloc(7 * (size_t)n, sizeof(float));
    if (!t->AP) return -1;
    t->BP = t->AP + n; t->AS2 = t->BP + n; t->BS2 = t->AS2 + n;
    t->invMu = t->BS2 + n; t->tauS = t->invMu + n; t->invRho = t->tauS + n;
    t->cmax = 0.0;
    for (int m = 0; m < n; m++) {
        const double *r = matlist + 5 * m;
        double q = qcorr ? qcorr[m] : 1.0;
        double MRp, tauP, MRs, tauSh;
        sls_fit(r[0], r[1], r[3], q, omega, p->qfactorCorrection, &MRp, &tauP);
        sls_fit(r[0], r[2], r[4], q, omega, p->qfactorCorrection, &MRs, &tauSh);
        t->AP[m] = (float)(MRp * (1.0 + tauP) * dtoh);
        t->BP[m] = (float)(MRp * tauP * dtoh * k2);
        t->AS2[m] = (float)(2.0 * MRs * (1.0 + tauSh) * dtoh);
        t->BS2[m] = (float)(2.0 * MRs * tauSh * dtoh * k2);
        t->invMu[m] = (MRs > 0.0) ? (float)(1.0 / (MRs * dtoh)) : 0.0f;
        t->tauS[m] = (float)tauSh;
        t->invRho[m] = (float)(dtoh / r[0]);
        double cu = sqrt(MRp * (1.0 + tauP) / r[0]);
        if (cu > t->cmax) t->cmax = cu;
    }
    return 0;
}

/* stable time step of the O(2,4) staggered scheme: dt <= (6/7) h / (sqrt(3) cmax) */
BFO_EXPORT double bfo_stable_dt(int nMat, const double *matlist, const double *qcorr, double freq,
                                int qfactorCorrection, double h, double alphaCFL)
{
    bfo_params p; memset(&p, 0, sizeof p);
    p.nMat = nMat; p.freq = freq; p.h = h; p.dt = 1.0; p.qfactorCorrection = qfactorCorrection;
    bfo_tables t;
    if (build_tables(&p, matlist, qcorr, &t)) return -1.0;
    double dt = alphaCFL * (6.0 / 7.0) * h / (sqrt(3.0) * t.cmax);
    free(t.AP);
    return dt;
}

/* expose the float32 tables for tests: out is 7*nMat floats (AP,BP,AS2,BS2,invMu,tauS,invRho), c1k2 2 floats */
BFO_EXPORT int bfo_tables_f32(const bfo_params *p, const double *matlist, const double *qcorr,
                              float *out, float *c1k2, double *cmax)
{
    bfo_tables t;
    if (build_tables(p, matlist, qcorr, &t)) return -1;
    memcpy(out, t.AP, 7 * (size_t)p->nMat * sizeof(float));
    c1k2[0] = t.c1; c1k2[1] = t.k2; *cmax = t.cmax;
    free(t.AP);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * CPML profiles for one axis of length N. Zone = first P and last P indices, P = NDelta+1.
 * aI/bI: derivative located at integer position i;  aH/bH: at i+1/2.
 * depth (fraction of layer thickness): left  int (NDelta-i)/NDelta, half (NDelta-i-0.5)/NDelta
 *                                      right int (i-(N-1-NDelta))/NDelta, half (+0.5)
 * d = d0 depth^2, d0 = -3 cmax ln(R) / (2 NDelta h);  alpha = pi f (1-depth);
 * b = exp(-(d+alpha) dt),  a = d/(d+alpha) (b-1);  outside the layer a = b = 0.
 * ---------------------------------------------------------------------------------------- */
static void cpml_coef(double depth, double d0, double amax, double dt, float *a, float *b)
{
    if (depth <= 0.0) { *a = 0.0f; *b = 0.0f; return; }
    if (depth > 1.0) depth = 1.0;
    double d = d0 * depth * depth;
    double al = amax * (1.0 - depth);
    double bb = exp(-(d + al) * dt);
    *b = (float)bb;
    *a = (float)(d / (d + al) * (bb - 1.0));
}

static void cpml_axis(int N, int ND, double d0, double amax, double dt,
                      float *aI, float *bI, float *aH, float *bH)
{
    for (int i = 0; i < N; i++) {
        double dl_i = (double)(ND - i) / ND, dl_h = ((double)(ND - i) - 0.5) / ND;
        double dr_i = (double)(i - (N - 1 - ND)) / ND, dr_h = ((double)(i - (N - 1 - ND)) + 0.5) / ND;
        double di = dl_i > dr_i ? dl_i : dr_i;
        double dh = dl_h > dr_h ? dl_h : dr_h;
        cpml_coef(di, d0, amax, dt, &aI[i], &bI[i]);
        cpml_coef(dh, d0, amax, dt, &aH[i], &bH[i]);
    }
}

BFO_EXPORT void bfo_cpml_profiles(int N, int NDelta, double cmax, double h, double dt, double freq,
                                  double reflectionLimit, float *aI, float *bI, float *aH, float *bH)
{
    double d0 = -3.0 * cmax * log(reflectionLimit) / (2.0 * NDelta * h);
    cpml_axis(N, NDelta, d0, M_PI * freq, dt, aI, bI, aH, bH);
}

/* ------------------------------------------------------------------------------------------ */
#define CA 1.125f
#define CB (1.0f / 24.0f)

typedef struct {
    int N1, N2, N3;
    size_t P1, P12;         /* padded pitches */
    float *f[15];           /* Vx Vy Vz Sxx Syy Szz Sxy Sxz Syz Rxx Ryy Rzz Rxy Rxz Ryz, each padded by 2 */
} fields_t;

/* padded index: ghost ring of 2 zeros in every dimension */
#define PIDX(F, i, j, k) ((size_t)((i) + 2) + (F)->P1 * (size_t)((j) + 2) + (F)->P12 * (size_t)((k) + 2))

static inline float dminus(const float *a, size_t c, size_t s)
{   /* backward: CA*(f[0]-f[-1]) - CB*(f[+1]-f[-2]) */
    float t1 = a[c] - a[c - s];
    float t2 = a[c + s] - a[c - 2 * s];
    return CA * t1 - CB * t2;
}
static inline float dplus(const float *a, size_t c, size_t s)
{   /* forward: CA*(f[+1]-f[0]) - CB*(f[+2]-f[-1]) */
    float t1 = a[c + s] - a[c];
    float t2 = a[c + 2 * s] - a[c - s];
    return CA * t1 - CB * t2;
}
static inline float cpml(float *psi, size_t idx, float a, float b, float D)
{
    float pn = b * psi[idx] + a * D;
    psi[idx] = pn;
    return D + pn;
}
static inline int clampi(int v, int hi) { return v > hi ? hi : v; }

static inline float map_value(const fields_t *F, int sel, size_t c)
{
    switch (sel) {
    case M_VX: return F->f[0][c];
    case M_VY: return F->f[1][c];
    case M_VZ: return F->f[2][c];
    case M_SXX: return F->f[3][c];
    case M_SYY: return F->f[4][c];
    case M_SZZ: return F->f[5][c];
    case M_SXY: return F->f[6][c];
    case M_SXZ: return F->f[7][c];
    case M_SYZ: return F->f[8][c];
    case M_PRESSURE: {
        float s = (F->f[3][c] + F->f[4][c]) + F->f[5][c];
        return -s * (1.0f / 3.0f);
    }
    default: return 0.0f;
    }
}
/* squared value used for RMS; ALLV = Vx^2+Vy^2+Vz^2 */
static inline float map_sq(const fields_t *F, int sel, size_t c)
{
    if (sel == M_ALLV) {
        float x = F->f[0][c], y = F->f[1][c], z = F->f[2][c];
        return (x * x + y * y) + z * z;
    }
    float v = map_value(F, sel, c);
    return v * v;
}

/* ------------------------------------------------------------------------------------------
 * Stateful oracle for one Z-slab [k0, k0+nk) of the global N1 x N2 x N3 domain (k0=0, nk=N3 is
 * the whole domain). The slab keeps 2 ghost planes per side; for an interior slab they are
 * filled by the caller between half-steps (bfo_halo_get / bfo_halo_put), at the ends of the
 * domain they stay zero. This mirrors the C ABI of the HIP engine so that the Z-slab
 * decomposition itself can be checked on CPU (tests/test_slab_gloo.py).
 * ---------------------------------------------------------------------------------------- */
typedef struct bfo_sim {
    bfo_params p;
    int k0, nk;
    size_t nloc;
    bfo_tables T;
    fields_t F;
    float *psi[18];
    float *prof;
    float *axI, *bxI, *axH, *bxH, *ayI, *byI, *ayH, *byH, *azI, *bzI, *azH, *bzH;
    uint32_t *mat;          /* (nk+4) planes, unpadded in x,y; pointer to local plane 0 */
    uint32_t *matBase;
    uint8_t *refl;          /* nloc or NULL */
    int selR[M_COUNT], nSelR, selS[M_COUNT], nSelS;
    int doRMS, doPeak;
    float *acc, *pk;
    size_t nSensors; uint32_t *sensLin; int nTs; float *sens;   /* sens[q][s][col] */
    size_t nSrcVox; uint32_t *srcIdx, *srcId; float *srcW[3]; float *pulse; /* pulse f32 [nSources][length] */
    int accStart;
    int step;
} bfo_sim;

static void ftz_on(void)
{
    /* canonical arithmetic flushes float32 denormals (inputs and results) to zero, as the HIP
       engine does (-fgpu-flush-denormals-to-zero); wave-front precursors otherwise sit in the
       denormal range and slow x86 by ~7x. MXCSR is per thread: set on every worker. */
#pragma omp parallel
    { unsigned int csr; __asm__ volatile("stmxcsr %0" : "=m"(csr)); csr |= 0x8040u; __asm__ volatile("ldmxcsr %0" : : "m"(csr)); }
}
static void ftz_off(void)
{
#pragma omp parallel
    { unsigned int csr; __asm__ volatile("stmxcsr %0" : "=m"(csr)); csr &= ~0x8040u; __asm__ volatile("ldmxcsr %0" : : "m"(csr)); }
}

/* zero-filled allocation whose pages are first touched by the OpenMP team in the same plane order the
 * half-step loops use, so on a multi-socket host every thread works on memory of its own NUMA node */
static float *numa_zeros(size_t n, size_t planeElems)
{
    float *p = (float *)malloc(n * sizeof(float));
    if (!p) return NULL;
    const long nplanes = (long)((n + planeElems - 1) / planeElems);
#pragma omp parallel for schedule(static)
    for (long q = 0; q < nplanes; q++) {
        const size_t a = (size_t)q * planeElems, b = a + planeElems < n ? a + planeElems : n;
        memset(p + a, 0, (b - a) * sizeof(float));
    }
    return p;
}

BFO_EXPORT void bfo_destroy(bfo_sim *S)
{
    if (!S) return;
    for (int a = 0; a < 15; a++) free(S->F.f[a]);
    for (int a = 0; a < 18; a++) free(S->psi[a]);
    free(S->prof); free(S->matBase); free(S->refl); free(S->acc); free(S->pk); free(S->sensLin); free(S->sens);
    free(S->srcIdx); free(S->srcId); free(S->srcW[0]); free(S->srcW[1]); free(S->srcW[2]); free(S->pulse); free(S->T.AP);
    free(S);
}

/*
 * All volume inputs are x-fastest and LOCAL to the slab (N1*N2*nk), except matmap which carries
 * ghostLow planes below and ghostHigh planes above (0..2 each; missing ones replicate the edge).
 *   matmap     uint32 material ids
 *   srcmap     uint32, 0 = none, s>=1 -> row s-1 of pulse (Single:326-344)
 *   pulse      float64 [nSources][lengthSource]
 *   Ox,Oy,Oz   float64 per-voxel weights, or NULL (= 1)                         (BASE:2325-2335)
 *   sensormap  uint32, nonzero = record                                          (BASE:2283-2290)
 *   reflector  uint32 or NULL, nonzero = fields forced to 0                      (BASE:2365)
 */
BFO_EXPORT bfo_sim *bfo_create(const bfo_params *p, int k0, int nk, const uint32_t *matmap, int ghostLow, int ghostHigh,
                               const double *matlist, const double *qcorr, const uint32_t *srcmap, const double *pulse,
                               const double *Ox, const double *Oy, const double *Oz,
                               const uint32_t *sensormap, const uint32_t *reflector, int *rcOut)
{
    const int N1 = p->N1, N2 = p->N2, N3 = p->N3, ND = p->NDelta;
    int rc = -1;
    bfo_sim *S = (bfo_sim *)calloc(1, sizeof(bfo_sim));
    if (!S) { if (rcOut) *rcOut = -1; return NULL; }
    S->p = *p; S->k0 = k0; S->nk = nk;
    const size_t plane = (size_t)N1 * N2;
    const size_t N = plane * nk;
    S->nloc = N;
#ifdef _OPENMP
    if (p->nthreads > 0) omp_set_num_threads(p->nthreads);
#endif
    if (build_tables(p, matlist, qcorr, &S->T)) goto fail;
    fields_t *F = &S->F;
    F->N1 = N1; F->N2 = N2; F->N3 = nk;
    F->P1 = (size_t)N1 + 4; F->P12 = F->P1 * ((size_t)N2 + 4);
    const size_t NP = F->P12 * ((size_t)nk + 4);
    for (int a = 0; a < 15; a++) { F->f[a] = numa_zeros(NP, F->P12); if (!F->f[a]) goto fail; }
    /* CPML memory variables, slab-size unpadded (only layer voxels are ever touched):
       0..8  stress half-step: dxVx dyVy dzVz | dyVx dxVy | dzVx dxVz | dzVy dyVz
       9..17 velocity half-step: dxSxx dySxy dzSxz | dxSxy dySyy dzSyz | dxSxz dySyz dzSzz */
    for (int a = 0; a < 18; a++) { S->psi[a] = numa_zeros(N, plane); if (!S->psi[a]) goto fail; }
    const int Nmax = N1 > N2 ? (N1 > N3 ? N1 : N3) : (N2 > N3 ? N2 : N3);
    S->prof = (float *)calloc(12 * (size_t)Nmax, sizeof(float));
    if (!S->prof) goto fail;
    S->axI = S->prof; S->bxI = S->axI + Nmax; S->axH = S->bxI + Nmax; S->bxH = S->axH + Nmax;
    S->ayI = S->bxH + Nmax; S->byI = S->ayI + Nmax; S->ayH = S->byI + Nmax; S->byH = S->ayH + Nmax;
    S->azI = S->byH + Nmax; S->bzI = S->azI + Nmax; S->azH = S->bzI + Nmax; S->bzH = S->azH + Nmax;
    bfo_cpml_profiles(N1, ND, S->T.cmax, p->h, p->dt, p->freq, p->reflectionLimit, S->axI, S->bxI, S->axH, S->bxH);
    bfo_cpml_profiles(N2, ND, S->T.cmax, p->h, p->dt, p->freq, p->reflectionLimit, S->ayI, S->byI, S->ayH, S->byH);
    bfo_cpml_profiles(N3, ND, S->T.cmax, p->h, p->dt, p->freq, p->reflectionLimit, S->azI, S->bzI, S->azH, S->bzH);

    /* material ids with 2 ghost planes per side */
    S->matBase = (uint32_t *)malloc(plane * ((size_t)nk + 4) * sizeof(uint32_t));
    if (!S->matBase) goto fail;
    S->mat = S->matBase + 2 * plane;
    for (int kl = -2; kl < nk + 2; kl++) {
        int ks = kl + ghostLow;                        /* plane index in the caller's buffer */
        if (ks < 0) ks = 0;
        if (ks > nk + ghostLow + ghostHigh - 1) ks = nk + ghostLow + ghostHigh - 1;
        memcpy(S->mat + (ptrdiff_t)kl * (ptrdiff_t)plane, matmap + (size_t)ks * plane, plane * sizeof(uint32_t));
    }
    for (size_t v = 0; v < plane * ((size_t)nk + 4); v++) if ((int)S->matBase[v] >= p->nMat) { rc = -3; goto fail; }
    if (reflector) {
        S->refl = (uint8_t *)malloc(N);
        if (!S->refl) goto fail;
        for (size_t v = 0; v < N; v++) S->refl[v] = reflector[v] != 0;
    }
    for (int b = 0; b < M_COUNT; b++) {
        if (p->selMapsRMS & (1u << b)) S->selR[S->nSelR++] = b;
        if (p->selMapsSensors & (1u << b)) S->selS[S->nSelS++] = b;
    }
    S->doRMS = (p->selRMSorPeak & 1) && S->nSelR; S->doPeak = (p->selRMSorPeak & 2) && S->nSelR;
    if (S->doRMS) { S->acc = numa_zeros((size_t)S->nSelR * N, plane); if (!S->acc) goto fail; }
    if (S->doPeak) { S->pk = numa_zeros((size_t)S->nSelR * N, plane); if (!S->pk) goto fail; }
    /* sensors: ascending x-fastest order */
    if (sensormap) for (size_t v = 0; v < N; v++) if (sensormap[v]) S->nSensors++;
    S->sensLin = (uint32_t *)malloc((S->nSensors + 1) * sizeof(uint32_t));
    if (!S->sensLin) goto fail;
    { size_t c = 0; if (sensormap) for (size_t v = 0; v < N; v++) if (sensormap[v]) S->sensLin[c++] = (uint32_t)v; }
    for (int n = 0; n < p->nt; n++) if (n % p->sensorSub == 0 && n / p->sensorSub >= p->sensorStart) S->nTs++;
    S->sens = (float *)calloc((size_t)S->nSelS * S->nSensors * (size_t)(S->nTs > 0 ? S->nTs : 1) + 1, sizeof(float));
    if (!S->sens) goto fail;
    S->accStart = p->sensorStart * p->sensorSub;
    /* sources */
    if (srcmap) {
        for (size_t v = 0; v < N; v++) if (srcmap[v]) S->nSrcVox++;
        S->srcIdx = (uint32_t *)malloc((S->nSrcVox + 1) * sizeof(uint32_t));
        S->srcId = (uint32_t *)malloc((S->nSrcVox + 1) * sizeof(uint32_t));
        for (int a = 0; a < 3; a++) S->srcW[a] = (float *)malloc((S->nSrcVox + 1) * sizeof(float));
        if (!S->srcIdx || !S->srcId || !S->srcW[0] || !S->srcW[1] || !S->srcW[2]) goto fail;
        size_t c = 0;
        for (size_t v = 0; v < N; v++) if (srcmap[v]) {
            if ((int)srcmap[v] > p->nSources) { rc = -2; goto fail; }
            S->srcIdx[c] = (uint32_t)v; S->srcId[c] = srcmap[v] - 1;
            S->srcW[0][c] = Ox ? (float)Ox[v] : 1.0f;
            S->srcW[1][c] = Oy ? (float)Oy[v] : 1.0f;
            S->srcW[2][c] = Oz ? (float)Oz[v] : 1.0f;
            c++;
        }
        const size_t np = (size_t)p->nSources * p->lengthSource;
        S->pulse = (float *)malloc((np + 1) * sizeof(float));
        if (!S->pulse) goto fail;
        for (size_t v = 0; v < np; v++) S->pulse[v] = (float)pulse[v];
    }
    if (rcOut) *rcOut = 0;
    return S;
fail:
    bfo_destroy(S);
    if (rcOut) *rcOut = rc;
    return NULL;
}

#define MATL(S, ii, jj, kl) (S)->mat[(ptrdiff_t)(ii) + (ptrdiff_t)N1 * ((ptrdiff_t)(jj) + (ptrdiff_t)N2 * (ptrdiff_t)(kl))]

BFO_EXPORT void bfo_half_stress(bfo_sim *S)
{
    const bfo_params *p = &S->p;
    const int N1 = p->N1, N2 = p->N2, N3 = p->N3, ND = p->NDelta, nk = S->nk, k0 = S->k0;
    const int PZ = ND + 1; /* layer zone width */
    fields_t *F = &S->F;
    const bfo_tables T = S->T;
    float **psi = S->psi;
    float *Vx = F->f[0], *Vy = F->f[1], *Vz = F->f[2];
    float *Sxx = F->f[3], *Syy = F->f[4], *Szz = F->f[5], *Sxy = F->f[6], *Sxz = F->f[7], *Syz = F->f[8];
    float *Rxx = F->f[9], *Ryy = F->f[10], *Rzz = F->f[11], *Rxy = F->f[12], *Rxz = F->f[13], *Ryz = F->f[14];
    const size_t sx = 1, sy = F->P1, sz = F->P12;
    const float c1 = T.c1, k2 = T.k2;
    const float *axI = S->axI, *bxI = S->bxI, *axH = S->axH, *bxH = S->bxH;
    const float *ayI = S->ayI, *byI = S->byI, *ayH = S->ayH, *byH = S->byH;
    const float *azI = S->azI, *bzI = S->bzI, *azH = S->azH, *bzH = S->bzH;
    const int n = S->step;
    ftz_on();
#pragma omp parallel for collapse(2) schedule(static)
    for (int kl = 0; kl < nk; kl++)
        for (int j = 0; j < N2; j++) {
            const int k = k0 + kl;
            const int zk = (k < PZ || k >= N3 - PZ), zj = (j < PZ || j >= N2 - PZ);
            for (int i = 0; i < N1; i++) {
                const size_t c = PIDX(F, i, j, kl);
                const size_t u = (size_t)i + (size_t)N1 * ((size_t)j + (size_t)N2 * kl);
                const int zi = (i < PZ || i >= N1 - PZ);
                const uint32_t m = MATL(S, i, j, kl);
                if (S->refl && S->refl[u]) {
                    Sxx[c] = Syy[c] = Szz[c] = Sxy[c] = Sxz[c] = Syz[c] = 0.0f;
                    Rxx[c] = Ryy[c] = Rzz[c] = Rxy[c] = Rxz[c] = Ryz[c] = 0.0f;
                    continue;
                }
                float dxVx = dminus(Vx, c, sx), dyVy = dminus(Vy, c, sy), dzVz = dminus(Vz, c, sz);
                float dyVx = dplus(Vx, c, sy), dxVy = dplus(Vy, c, sx);
                float dzVx = dplus(Vx, c, sz), dxVz = dplus(Vz, c, sx);
                float dzVy = dplus(Vy, c, sz), dyVz = dplus(Vz, c, sy);
                if (zi) {
                    dxVx = cpml(psi[0], u, axI[i], bxI[i], dxVx);
                    dxVy = cpml(psi[4], u, axH[i], bxH[i], dxVy);
                    dxVz = cpml(psi[6], u, axH[i], bxH[i], dxVz);
                }
                if (zj) {
                    dyVy = cpml(psi[1], u, ayI[j], byI[j], dyVy);
                    dyVx = cpml(psi[3], u, ayH[j], byH[j], dyVx);
                    dyVz = cpml(psi[8], u, ayH[j], byH[j], dyVz);
                }
                if (zk) {
                    dzVz = cpml(psi[2], u, azI[k], bzI[k], dzVz);
                    dzVx = cpml(psi[5], u, azH[k], bzH[k], dzVx);
                    dzVy = cpml(psi[7], u, azH[k], bzH[k], dzVy);
                }
                /* normal stresses */
                {
                    const float AP = T.AP[m], BP = T.BP[m], AS2 = T.AS2[m], BS2 = T.BS2[m];
                    const float sXY = dxVx + dyVy;
                    const float div = sXY + dzVz;
                    const float sYZ = dyVy + dzVz, sXZ = dxVx + dzVz;
                    float r, rn;
                    r = Rxx[c]; rn = c1 * r - (BP * div - BS2 * sYZ);
                    Sxx[c] = Sxx[c] + ((AP * div - AS2 * sYZ) + 0.5f * (r + rn)); Rxx[c] = rn;
                    r = Ryy[c]; rn = c1 * r - (BP * div - BS2 * sXZ);
                    Syy[c] = Syy[c] + ((AP * div - AS2 * sXZ) + 0.5f * (r + rn)); Ryy[c] = rn;
                    r = Rzz[c]; rn = c1 * r - (BP * div - BS2 * sXY);
                    Szz[c] = Szz[c] + ((AP * div - AS2 * sXY) + 0.5f * (r + rn)); Rzz[c] = rn;
                }
                /* shear stresses: harmonic mean of mu over the 4 cells around the edge,
                   arithmetic mean of tauS; any fluid neighbour => no shear. The k+1 neighbour comes
                   from the ghost plane (neighbour slab, or the replicated edge at the domain end). */
                {
                    const int i1 = clampi(i + 1, N1 - 1), j1 = clampi(j + 1, N2 - 1), k1 = kl + 1;
                    const uint32_t mx = MATL(S, i1, j, kl), my = MATL(S, i, j1, kl), mz = MATL(S, i, j, k1);
                    const uint32_t mxy = MATL(S, i1, j1, kl), mxz = MATL(S, i1, j, k1), myz = MATL(S, i, j1, k1);
                    const float i0 = T.invMu[m], t0s = T.tauS[m];
                    {   /* xy */
                        const float a = T.invMu[mx], b = T.invMu[my], d = T.invMu[mxy];
                        if (i0 > 0.0f && a > 0.0f && b > 0.0f && d > 0.0f) {
                            const float muH = 4.0f / ((i0 + a) + (b + d));
                            const float tau = 0.25f * ((t0s + T.tauS[mx]) + (T.tauS[my] + T.tauS[mxy]));
                            const float A = muH * (1.0f + tau), B = (muH * tau) * k2;
                            const float e = dyVx + dxVy;
                            const float r = Rxy[c], rn = c1 * r - B * e;
                            Sxy[c] = Sxy[c] + (A * e + 0.5f * (r + rn)); Rxy[c] = rn;
                        }
                    }
                    {   /* xz */
                        const float a = T.invMu[mx], b = T.invMu[mz], d = T.invMu[mxz];
                        if (i0 > 0.0f && a > 0.0f && b > 0.0f && d > 0.0f) {
                            const float muH = 4.0f / ((i0 + a) + (b + d));
                            const float tau = 0.25f * ((t0s + T.tauS[mx]) + (T.tauS[mz] + T.tauS[mxz]));
                            const float A = muH * (1.0f + tau), B = (muH * tau) * k2;
                            const float e = dzVx + dxVz;
                            const float r = Rxz[c], rn = c1 * r - B * e;
                            Sxz[c] = Sxz[c] + (A * e + 0.5f * (r + rn)); Rxz[c] = rn;
                        }
                    }
                    {   /* yz */
                        const float a = T.invMu[my], b = T.invMu[mz], d = T.invMu[myz];
                        if (i0 > 0.0f && a > 0.0f && b > 0.0f && d > 0.0f) {
                            const float muH = 4.0f / ((i0 + a) + (b + d));
                            const float tau = 0.25f * ((t0s + T.tauS[my]) + (T.tauS[mz] + T.tauS[myz]));
                            const float A = muH * (1.0f + tau), B = (muH * tau) * k2;
                            const float e = dzVy + dyVz;
                            const float r = Ryz[c], rn = c1 * r - B * e;
                            Syz[c] = Syz[c] + (A * e + 0.5f * (r + rn)); Ryz[c] = rn;
                        }
                    }
                }
            }
        }
    /* stress sources (TypeSource 2 add, 3 set): value * Ox on the three normal stresses */
    if (S->nSrcVox && p->typeSource >= 2 && n < p->lengthSource) {
        for (size_t s = 0; s < S->nSrcVox; s++) {
            const size_t u = S->srcIdx[s];
            const int i = (int)(u % N1), j = (int)((u / N1) % N2), kl = (int)(u / ((size_t)N1 * N2));
            const size_t c = PIDX(F, i, j, kl);
            const float val = S->pulse[(size_t)S->srcId[s] * p->lengthSource + n] * S->srcW[0][s];
            if (p->typeSource == 2) { Sxx[c] = Sxx[c] + val; Syy[c] = Syy[c] + val; Szz[c] = Szz[c] + val; }
            else { Sxx[c] = val; Syy[c] = val; Szz[c] = val; }
        }
    }
    ftz_off();
}

BFO_EXPORT void bfo_half_velocity(bfo_sim *S)
{
    const bfo_params *p = &S->p;
    const int N1 = p->N1, N2 = p->N2, N3 = p->N3, ND = p->NDelta, nk = S->nk, k0 = S->k0;
    const int PZ = ND + 1;
    fields_t *F = &S->F;
    const bfo_tables T = S->T;
    float **psi = S->psi;
    float *Vx = F->f[0], *Vy = F->f[1], *Vz = F->f[2];
    float *Sxx = F->f[3], *Syy = F->f[4], *Szz = F->f[5], *Sxy = F->f[6], *Sxz = F->f[7], *Syz = F->f[8];
    const size_t sx = 1, sy = F->P1, sz = F->P12;
    const float *axI = S->axI, *bxI = S->bxI, *axH = S->axH, *bxH = S->bxH;
    const float *ayI = S->ayI, *byI = S->byI, *ayH = S->ayH, *byH = S->byH;
    const float *azI = S->azI, *bzI = S->bzI, *azH = S->azH, *bzH = S->bzH;
    const size_t N = S->nloc;
    const int n = S->step;
    ftz_on();
#pragma omp parallel for collapse(2) schedule(static)
    for (int kl = 0; kl < nk; kl++)
        for (int j = 0; j < N2; j++) {
            const int k = k0 + kl;
            const int zk = (k < PZ || k >= N3 - PZ), zj = (j < PZ || j >= N2 - PZ);
            for (int i = 0; i < N1; i++) {
                const size_t c = PIDX(F, i, j, kl);
                const size_t u = (size_t)i + (size_t)N1 * ((size_t)j + (size_t)N2 * kl);
                const int zi = (i < PZ || i >= N1 - PZ);
                if (S->refl && S->refl[u]) { Vx[c] = Vy[c] = Vz[c] = 0.0f; continue; }
                float dxSxx = dplus(Sxx, c, sx), dySxy = dminus(Sxy, c, sy), dzSxz = dminus(Sxz, c, sz);
                float dxSxy = dminus(Sxy, c, sx), dySyy = dplus(Syy, c, sy), dzSyz = dminus(Syz, c, sz);
                float dxSxz = dminus(Sxz, c, sx), dySyz = dminus(Syz, c, sy), dzSzz = dplus(Szz, c, sz);
                if (zi) {
                    dxSxx = cpml(psi[9], u, axH[i], bxH[i], dxSxx);
                    dxSxy = cpml(psi[12], u, axI[i], bxI[i], dxSxy);
                    dxSxz = cpml(psi[15], u, axI[i], bxI[i], dxSxz);
                }
                if (zj) {
                    dySxy = cpml(psi[10], u, ayI[j], byI[j], dySxy);
                    dySyy = cpml(psi[13], u, ayH[j], byH[j], dySyy);
                    dySyz = cpml(psi[16], u, ayI[j], byI[j], dySyz);
                }
                if (zk) {
                    dzSxz = cpml(psi[11], u, azI[k], bzI[k], dzSxz);
                    dzSyz = cpml(psi[14], u, azI[k], bzI[k], dzSyz);
                    dzSzz = cpml(psi[17], u, azH[k], bzH[k], dzSzz);
                }
                const int i1 = clampi(i + 1, N1 - 1), j1 = clampi(j + 1, N2 - 1);
                const float r0 = T.invRho[MATL(S, i, j, kl)];
                const float bx = 0.5f * (r0 + T.invRho[MATL(S, i1, j, kl)]);
                const float by = 0.5f * (r0 + T.invRho[MATL(S, i, j1, kl)]);
                const float bz = 0.5f * (r0 + T.invRho[MATL(S, i, j, kl + 1)]);
                Vx[c] = Vx[c] + bx * ((dxSxx + dySxy) + dzSxz);
                Vy[c] = Vy[c] + by * ((dxSxy + dySyy) + dzSyz);
                Vz[c] = Vz[c] + bz * ((dxSxz + dySyz) + dzSzz);
            }
        }
    /* velocity sources (TypeSource 0 add, 1 set) */
    if (S->nSrcVox && p->typeSource < 2 && n < p->lengthSource) {
        for (size_t s = 0; s < S->nSrcVox; s++) {
            const size_t u = S->srcIdx[s];
            const int i = (int)(u % N1), j = (int)((u / N1) % N2), kl = (int)(u / ((size_t)N1 * N2));
            const size_t c = PIDX(F, i, j, kl);
            const float val = S->pulse[(size_t)S->srcId[s] * p->lengthSource + n];
            const float wx = S->srcW[0][s], wy = S->srcW[1][s], wz = S->srcW[2][s];
            if (p->typeSource == 0) {
                Vx[c] = Vx[c] + val * wx; Vy[c] = Vy[c] + val * wy; Vz[c] = Vz[c] + val * wz;
            } else {
                Vx[c] = val * wx; Vy[c] = val * wy; Vz[c] = val * wz;
            }
        }
    }
    /* ---------------- RMS / peak accumulation (outside the absorbing layer only) -------- */
    if ((S->doRMS || S->doPeak) && n >= S->accStart) {
        float *acc = S->acc, *pk = S->pk;
        const int nSelR = S->nSelR; const int *selR = S->selR;
#pragma omp parallel for collapse(2) schedule(static)
        for (int kl = 0; kl < nk; kl++)
            for (int j = ND; j < N2 - ND; j++) {
                const int k = k0 + kl;
                if (k < ND || k >= N3 - ND) continue;
                for (int i = ND; i < N1 - ND; i++) {
                    const size_t c = PIDX(F, i, j, kl);
                    const size_t u = (size_t)i + (size_t)N1 * ((size_t)j + (size_t)N2 * kl);
                    for (int q = 0; q < nSelR; q++) {
                        if (S->doRMS) acc[(size_t)q * N + u] = acc[(size_t)q * N + u] + map_sq(F, selR[q], c);
                        if (S->doPeak) {
                            float v = (selR[q] == M_ALLV) ? sqrtf(map_sq(F, M_ALLV, c)) : fabsf(map_value(F, selR[q], c));
                            if (v > pk[(size_t)q * N + u]) pk[(size_t)q * N + u] = v;
                        }
                    }
                }
            }
    }
    /* ---------------- sensors ---------------- */
    if (S->nSensors && S->nSelS && n % p->sensorSub == 0 && n / p->sensorSub >= p->sensorStart) {
        const int col = n / p->sensorSub - p->sensorStart;
        if (col < S->nTs) {
#pragma omp parallel for schedule(static)
            for (size_t s = 0; s < S->nSensors; s++) {
                const size_t u = S->sensLin[s];
                const int i = (int)(u % N1), j = (int)((u / N1) % N2), kl = (int)(u / ((size_t)N1 * N2));
                const size_t c = PIDX(F, i, j, kl);
                for (int q = 0; q < S->nSelS; q++) {
                    float v = (S->selS[q] == M_ALLV) ? sqrtf(map_sq(F, M_ALLV, c)) : map_value(F, S->selS[q], c);
                    S->sens[((size_t)q * S->nSensors + s) * S->nTs + col] = v;
                }
            }
        }
    }
    ftz_off();
    S->step++;
}

/* halo planes: group 0 = Vx,Vy,Vz  group 1 = Sxz,Syz,Szz; side 0 = low-k face, 1 = high-k face.
 * get copies the 2 OWNED boundary planes into buf[2][N2][N1]; put fills the 2 GHOST planes. */
static float *halo_field(bfo_sim *S, int group, int f)
{
    static const int map[2][3] = {{0, 1, 2}, {7, 8, 5}};
    return S->F.f[map[group][f]];
}
BFO_EXPORT void bfo_halo_get(bfo_sim *S, int group, int f, int side, float *buf)
{
    const int N1 = S->p.N1, N2 = S->p.N2;
    float *a = halo_field(S, group, f);
    const int kl0 = side == 0 ? 0 : S->nk - 2;
    for (int q = 0; q < 2; q++) for (int j = 0; j < N2; j++)
        memcpy(buf + ((size_t)q * N2 + j) * N1, a + PIDX(&S->F, 0, j, kl0 + q), (size_t)N1 * sizeof(float));
}
BFO_EXPORT void bfo_halo_put(bfo_sim *S, int group, int f, int side, const float *buf)
{
    const int N1 = S->p.N1, N2 = S->p.N2;
    float *a = halo_field(S, group, f);
    const int kl0 = side == 0 ? -2 : S->nk;
    for (int q = 0; q < 2; q++) for (int j = 0; j < N2; j++)
        memcpy(a + PIDX(&S->F, 0, j, kl0 + q), buf + ((size_t)q * N2 + j) * N1, (size_t)N1 * sizeof(float));
}

BFO_EXPORT size_t bfo_num_sensors(bfo_sim *S) { return S->nSensors; }
BFO_EXPORT int bfo_num_sensor_steps(bfo_sim *S) { return S->nTs; }

/*
 * Outputs (caller allocated, any may be NULL):
 *   sensorsOut [nSelSensors][nSensors][nTs] float32
 *   indexSensor[nSensors] uint32, 1-based GLOBAL x-fastest linear index           (BASE:2503)
 *   rmsOut     [nSelRMS][nloc] float32 (if selRMSorPeak & 1), zero inside the absorbing layer
 *   peakOut    [nSelRMS][nloc] float32 (if selRMSorPeak & 2)
 *   lastOut    [nSelRMS][nloc] float32 current values of the selected maps
 */
BFO_EXPORT void bfo_results(bfo_sim *S, float *sensorsOut, uint32_t *indexSensor, float *rmsOut, float *peakOut, float *lastOut)
{
    const int N1 = S->p.N1, N2 = S->p.N2, nk = S->nk;
    const size_t N = S->nloc;
    if (sensorsOut) memcpy(sensorsOut, S->sens, (size_t)S->nSelS * S->nSensors * (size_t)S->nTs * sizeof(float));
    if (indexSensor) {
        const uint32_t off = (uint32_t)((size_t)S->k0 * N1 * N2 + 1);
        for (size_t s = 0; s < S->nSensors; s++) indexSensor[s] = S->sensLin[s] + off;
    }
    const int nAcc = S->step - S->accStart;
    const float cnt = (float)(nAcc > 0 ? nAcc : 1);
    ftz_on();       /* the maps are finalised in the canonical arithmetic too: acc / cnt may land in the denormal range */
    for (int q = 0; q < S->nSelR; q++) {
        if (S->doRMS && rmsOut) for (size_t u = 0; u < N; u++) rmsOut[(size_t)q * N + u] = sqrtf(S->acc[(size_t)q * N + u] / cnt);
        if (S->doPeak && peakOut) memcpy(peakOut + (size_t)q * N, S->pk + (size_t)q * N, N * sizeof(float));
        if (lastOut)
            for (int kl = 0; kl < nk; kl++) for (int j = 0; j < N2; j++) for (int i = 0; i < N1; i++) {
                const size_t c = PIDX(&S->F, i, j, kl);
                const size_t u = (size_t)i + (size_t)N1 * ((size_t)j + (size_t)N2 * kl);
                lastOut[(size_t)q * N + u] = (S->selR[q] == M_ALLV) ? sqrtf(map_sq(&S->F, M_ALLV, c)) : map_value(&S->F, S->selR[q], c);
            }
    }
    ftz_off();
}

/* whole-domain convenience: create, nt steps, results, destroy */
BFO_EXPORT int bfo_run(const bfo_params *p, const uint32_t *matmap, const double *matlist,
                       const double *qcorr, const uint32_t *srcmap, const double *pulse,
                       const double *Ox, const double *Oy, const double *Oz,
                       const uint32_t *sensormap, const uint32_t *reflector,
                       float *sensorsOut, uint32_t *indexSensor, float *rmsOut, float *peakOut,
                       float *lastOut, double *stepLoopSeconds)
{
    int rc = 0;
    bfo_sim *S = bfo_create(p, 0, p->N3, matmap, 0, 0, matlist, qcorr, srcmap, pulse, Ox, Oy, Oz, sensormap, reflector, &rc);
    if (!S) return rc;
    double t0 = 0.0;
#ifdef _OPENMP
    t0 = omp_get_wtime();
#endif
    for (int n = 0; n < p->nt; n++) { bfo_half_stress(S); bfo_half_velocity(S); }
#ifdef _OPENMP
    if (stepLoopSeconds) *stepLoopSeconds = omp_get_wtime() - t0;
#else
    if (stepLoopSeconds) *stepLoopSeconds = 0.0;
#endif
    bfo_results(S, sensorsOut, indexSensor, rmsOut, peakOut, lastOut);
    bfo_destroy(S);
    return 0;
}

BFO_EXPORT int bfo_count_sensor_steps(int nt, int sensorSub, int sensorStart)
{
    int c = 0;
    for (int n = 0; n < nt; n++) if (n % sensorSub == 0 && n / sensorSub >= sensorStart) c++;
    return c;
}
