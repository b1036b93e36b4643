/* CPU ORACLE (test infrastructure) of the Rayleigh-Sommerfeld sum that the reference obtains from
 * BabelViscoFDTD.tools.RayleighAndBHTE.ForwardSimple (package absent from /root/reference; call sites
 * TranscranialModeling/BabelIntegrationSingle.py:295, BabelIntegrationCONCAVE_PHASEDARRAY.py:307-328):
 *     u2(r) = (i k / 2 pi) sum_m u0_m dS_m exp(-i k R_m) / R_m ,   k = kRe + i kIm
 * The same formula as oracle/rayleigh_oracle.py (float64 numpy), restated in C so that the water fields of the reference's
 * own acceptance study (tests/test_oracle_study.py: 2e4 sub-sources x 1.5e6 points) are affordable on CPU: geometry, distance
 * and phase reduction in float64, sine / cosine of the reduced phase in float32 (error 6e-8 of a field of order one),
 * accumulation in float64. PARITY UNPINNED against the absent package, like the Python form; tests hold this file to the
 * numpy form. Only tests/ may load it. */
#include <math.h>
#include <omp.h>
#include <stdlib.h>

#define EXPORT __attribute__((visibility("default")))

EXPORT int bro_forward(long nSrc, const float *center, const float *ds, const float *u0, double kRe, double kIm,
                       long nPts, const float *rf, float *out, int nthreads)
{
    double *cx = malloc(sizeof(double) * 5 * (size_t)(nSrc > 0 ? nSrc : 1));
    if (!cx) return -1;
    double *cy = cx + nSrc, *cz = cy + nSrc, *wr = cz + nSrc, *wi = wr + nSrc;
    for (long m = 0; m < nSrc; m++) {
        cx[m] = center[3 * m]; cy[m] = center[3 * m + 1]; cz[m] = center[3 * m + 2];
        wr[m] = (double)u0[2 * m] * ds[m]; wi[m] = (double)u0[2 * m + 1] * ds[m];
    }
    if (nthreads > 0) omp_set_num_threads(nthreads);
    const double TWO_PI = 6.283185307179586476925286766559, INV_2PI = 1.0 / TWO_PI, PI = 0.5 * TWO_PI, HALF_PI = 0.25 * TWO_PI;
    const int lossy = kIm != 0.0;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < nPts; p++) {
        const double px = rf[3 * p], py = rf[3 * p + 1], pz = rf[3 * p + 2];
        double sr = 0.0, si = 0.0;
#pragma omp simd reduction(+ : sr, si)
        for (long m = 0; m < nSrc; m++) {
            const double dx = px - cx[m], dy = py - cy[m], dz = pz - cz[m];
            const double R = sqrt(dx * dx + dy * dy + dz * dz);
            const double ph = kRe * R;
            const double red = ph - TWO_PI * floor(ph * INV_2PI + 0.5);
            /* sine / cosine of the reduced phase: folded into [-pi/2, pi/2] in float64, then float32 polynomials (Taylor to
             * x^11 / x^12: truncation below 6e-8 there); plain arithmetic, so the loop vectorises */
            const double ared = fabs(red);
            const int fold = ared > HALF_PI;
            const float xr = (float)(fold ? copysign(PI, red) - red : red);
            const float x2 = xr * xr;
            const float s = xr * (1.0f + x2 * (-1.0f / 6 + x2 * (1.0f / 120 + x2 * (-1.0f / 5040 + x2 * (1.0f / 362880 + x2 * (-1.0f / 39916800))))));
            const float c0 = 1.0f + x2 * (-0.5f + x2 * (1.0f / 24 + x2 * (-1.0f / 720 + x2 * (1.0f / 40320 + x2 * (-1.0f / 3628800 + x2 * (1.0f / 479001600))))));
            const float c = fold ? -c0 : c0;
            const double a = (lossy ? exp(kIm * R) : 1.0) / R;
            /* w exp(-i ph) = (wr + i wi)(c - i s) */
            sr += (wr[m] * c + wi[m] * s) * a;
            si += (wi[m] * c - wr[m] * s) * a;
        }
        /* (i k / 2 pi)(sr + i si), i k = -kIm + i kRe */
        out[2 * p] = (float)((-kIm * sr - kRe * si) * INV_2PI);
        out[2 * p + 1] = (float)((-kIm * si + kRe * sr) * INV_2PI);
    }
    free(cx);
    return 0;
}
