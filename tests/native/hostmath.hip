// hostmath.hip — compiles the kernels' host+device math headers for the CPU, so that the
// `-m "not gpu"` tests can compare the very same source with scipy / long-double libm.
// Test infrastructure; built on demand by tests/test_hostmath.py (needs hipcc, no GPU).
#include "rvll_math.h"
#include "rvll_special.h"
#define HM extern "C" __attribute__((visibility("default")))
HM void hm_sincos(const double* x, long n, double* s, double* c) { for (long i = 0; i < n; ++i) rvll::sincos_f64(x[i], s[i], c[i]); }
HM void hm_sincos_any(const double* x, long n, double* s, double* c) { for (long i = 0; i < n; ++i) rvll::sincos_any(x[i], s[i], c[i]); }
HM void hm_sincos_cr(const double* x, long n, double* s, double* c) { for (long i = 0; i < n; ++i) rvll::sincos_cr(x[i], s[i], c[i]); }
// the short route of sincos_cr by itself: ok[i] = 0 where it declines (too close to a rounding boundary)
HM void hm_sincos_cr_table(const double* x, long n, double* s, double* c, int* ok) {
    for (long i = 0; i < n; ++i) {
        rvll::DD r; uint32_t q; double sr = 0., cr = 0.;
        rvll::reduce_dd(x[i], r, q);
        ok[i] = rvll::sincos_dd_table(r, sr, cr) ? 1 : 0;
        const double a = (q & 1u) ? cr : sr, b = (q & 1u) ? sr : cr;
        s[i] = ((q & 2u) ? -a : a) * (x[i] < 0 ? -1. : 1.);
        c[i] = ((q + 1u) & 2u) ? -b : b;
    } }
// the long reduction by itself (any |x| >= 1), so that it can be checked where the short one is valid too
HM void hm_sincos_long(const double* x, long n, double* s, double* c) {
    for (long i = 0; i < n; ++i) {
        double r, sr, cr; uint32_t q;
        rvll::reduce_huge(x[i], r, q);
        rvll::sincos_kernel(r, sr, cr, rvll::sincos_consts());
        const double a = (q & 1u) ? cr : sr, b = (q & 1u) ? sr : cr;
        s[i] = ((q & 2u) ? -a : a) * (x[i] < 0 ? -1. : 1.);
        c[i] = ((q + 1u) & 2u) ? -b : b;
    } }
HM void hm_ndtri(const double* p, long n, double* out) { for (long i = 0; i < n; ++i) out[i] = rvll::ndtri_f64(p[i]); }
HM void hm_ndtri_cephes(const double* p, long n, double* out) { for (long i = 0; i < n; ++i) out[i] = rvll::ndtri_cephes(p[i]); }
HM void hm_beta_ppf(const double* q, long n, double a, double b, double lbeta, double* out) { for (long i = 0; i < n; ++i) out[i] = rvll::beta_ppf(q[i], a, b, lbeta); }
HM void hm_gamma_ppf(const double* q, long n, double alpha, double beta, double lgam, double* out) { for (long i = 0; i < n; ++i) out[i] = rvll::gamma_ppf(q[i], alpha, beta, lgam); }
HM void hm_alpha_ppf(const double* q, long n, double a, double phi_a, double* out) { for (long i = 0; i < n; ++i) out[i] = rvll::alpha_ppf(q[i], a, phi_a); }
HM void hm_betainc(const double* x, long n, double a, double b, double lbeta, double* out) { for (long i = 0; i < n; ++i) out[i] = rvll::betainc_lower(a, b, x[i], lbeta); }
// tables as prior_table_kernel builds them: z[n]; dz[2n] = slopes, then second derivatives.  The return value
// is what prior_table_check_kernel measures: max |quintic interpolant - full solver| over the interval midpoints.
HM double hm_beta_table(double a, double b, double lbeta, double* z, double* dz) {
    const int n = rvll::kTableN;
    const double h = 2. * rvll::kTableU / (n - 1);
    for (int i = 0; i < n; ++i) {
        const double u = -rvll::kTableU + i * h;
        rvll::beta_table_node(a, b, lbeta, u, z[i], dz[i]);
        dz[n + i] = rvll::beta_table_d2(a, b, u, z[i], dz[i]);
    }
    double worst = 0.;
    for (int i = 0; i < n - 1; ++i) {
        const double u = -rvll::kTableU + (i + 0.5) * h;
        double zt, dzt;
        rvll::beta_table_node(a, b, lbeta, u, zt, dzt);
        const double err = fabs(rvll::quintic_table(z, dz, u) - zt);
        if (!(err <= worst)) worst = err;
    }
    return worst; }
HM void hm_beta_ppf_table(const double* q, long n, double a, double b, double lbeta, const double* z, const double* dz, int direct, double* out) {
    for (long i = 0; i < n; ++i) out[i] = rvll::beta_ppf_table(q[i], a, b, lbeta, z, dz, direct != 0); }
HM double hm_gamma_table(double alpha, double lgam, double* z, double* dz) {
    const int n = rvll::kTableN;
    const double h = 2. * rvll::kTableU / (n - 1);
    for (int i = 0; i < n; ++i) {
        const double u = -rvll::kTableU + i * h;
        rvll::gamma_table_node(alpha, lgam, u, z[i], dz[i]);
        dz[n + i] = rvll::gamma_table_d2(alpha, u, z[i], dz[i]);
    }
    double worst = 0.;
    for (int i = 0; i < n - 1; ++i) {
        const double u = -rvll::kTableU + (i + 0.5) * h;
        double zt, dzt;
        rvll::gamma_table_node(alpha, lgam, u, zt, dzt);
        const double err = fabs(rvll::quintic_table(z, dz, u) - zt);
        if (!(err <= worst)) worst = err;
    }
    return worst; }
HM void hm_gamma_ppf_table(const double* q, long n, double alpha, double beta, double lgam, const double* z, const double* dz, int direct, double* out) {
    for (long i = 0; i < n; ++i) out[i] = rvll::gamma_ppf_table(q[i], alpha, beta, lgam, z, dz, direct != 0); }
HM double hm_table_direct_tol(void) { return rvll::kTableDirectTol; }
HM int hm_table_n(void) { return rvll::kTableN; }
