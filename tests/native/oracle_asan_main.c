/* Drives the CPU oracle (oracle/rvll_oracle.c) under AddressSanitizer + UBSan: tests/test_oracle_sanitizers.py
 * compiles this file together with the oracle source with -fsanitize=address,undefined and expects exit 0.
 * A 2-planet, 2-instrument model with drift over 37 epochs; also the itmax mid-array abort and the curves. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rvll.h"

int rvo_trueanomaly(const double*, int, double, double*, int, double, int32_t*);
int rvo_loglike_batch(const rvll_layout*, const double*, const double*, const double*, const int32_t*, int,
                      const double*, const double*, long, double*, int32_t*, int);
int rvo_kep_rv_batch(const rvll_layout*, const double*, long, const double*, int, unsigned, double*);
int rvo_iteration_counts(const rvll_layout*, const double*, int, const double*, int32_t*);

static rvll_slot freep(int i) { rvll_slot s = {i, 0, 0.0}; return s; }
static rvll_slot fixedp(double v) { rvll_slot s = {-1, 0, v}; return s; }

int main(void)
{
    enum { NE = 37, NP = 2, NI = 2, D = 15, B = 9 };
    double t[NE], y[NE], sv[NE], lin[NE];
    int32_t inst[NE];
    for (int j = 0; j < NE; ++j) {
        t[j] = 50000.0 + 11.3 * j; y[j] = 5.0 * sin(0.37 * j); sv[j] = 1.0 + 0.05 * j; inst[j] = j < 20 ? 0 : 1;
        lin[j] = cos(0.11 * j);
    }
    rvll_planet planets[NP];
    for (int p = 0; p < NP; ++p) {
        memset(&planets[p], 0, sizeof planets[p]);
        planets[p].k_kind = p == 0 ? RVLL_K_K1 : RVLL_K_LOGK1;
        planets[p].p_kind = RVLL_P_PERIOD;
        planets[p].ecc_kind = p == 0 ? RVLL_ECC_DIRECT : RVLL_ECC_SECOS_SESIN;
        planets[p].anom_kind = p == 0 ? RVLL_ANOM_MA0 : RVLL_ANOM_ML0;
        planets[p].k = freep(5 * p + 0); planets[p].p = freep(5 * p + 1); planets[p].e1 = freep(5 * p + 2);
        planets[p].e2 = freep(5 * p + 3); planets[p].anom = freep(5 * p + 4); planets[p].epoch = fixedp(50010.0);
    }
    rvll_inst insts[NI] = {{freep(10), freep(11)}, {freep(12), fixedp(0.7)}};
    rvll_slot linslots[1] = {freep(13)};
    rvll_layout L;
    memset(&L, 0, sizeof L);
    L.struct_size = (int32_t)sizeof L; L.ndim = D; L.nplanets = NP; L.ninst = NI; L.has_jitter = 1; L.has_drift = 1;
    L.tref_from_data = 1; L.nlinpar = 1; L.drift[0] = freep(14); L.drift[1] = fixedp(0.01); L.drift[2] = fixedp(0.0);
    L.drift[3] = fixedp(0.0); L.tref = fixedp(0.0); L.planets = planets; L.insts = insts; L.linpar = linslots;
    L.tol = 1e-4; L.itmax = 10000; L.precision = RVLL_PREC_FP64;

    double theta[B * D], logl[B], curves[B * NE];
    int32_t flags[B], iters[NP * NE];
    for (int b = 0; b < B; ++b) {
        double* th = theta + b * D;
        th[0] = 3.0 + b; th[1] = 17.0 + 3 * b; th[2] = 0.1 * b; th[3] = 0.4 * b; th[4] = 0.3 + b;
        th[5] = log(2.0 + b); th[6] = 33.0 + b; th[7] = 0.1 * b; th[8] = 0.05 * b; th[9] = 1.0;
        th[10] = -1.0; th[11] = 0.5 * b; th[12] = 2.0; th[13] = 0.3; th[14] = 0.2 * b;
    }
    theta[8 * D + 7] = 0.9; theta[8 * D + 8] = 0.8;                 /* last point: secos^2+sesin^2 > 1 -> -1e30 */
    if (rvo_loglike_batch(&L, t, y, sv, inst, NE, lin, theta, B, logl, flags, 1) != 0) return 2;
    if (logl[B - 1] != -1e30 || !(flags[B - 1] & RVLL_FLAG_INVALID_ORBIT)) return 3;
    for (int b = 0; b < B - 1; ++b) if (!isfinite(logl[b])) return 4;
    if (rvo_kep_rv_batch(&L, theta, B, t, NE, 0x3u, curves) != 0) return 5;
    if (!isnan(curves[(B - 1) * NE])) return 6;
    if (rvo_iteration_counts(&L, t, NE, theta, iters) != 0) return 7;
    L.itmax = 2;                                                     /* forces the mid-array abort path */
    if (rvo_loglike_batch(&L, t, y, sv, inst, NE, lin, theta, B, logl, flags, 1) != 0) return 8;
    if (!(flags[1] & RVLL_FLAG_NONCONVERGED)) return 9;
    double M[8] = {0.1, 0.5, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0}, nu[8] = {0};
    if (rvo_trueanomaly(M, 8, 0.5, nu, 3, 0.0, NULL) != -1) return 10;
    printf("oracle sanitizer run ok: logL[0] = %.12g\n", logl[0]);
    return 0;
}
