// hostmath.hip — compiles the kernels' host+device math headers for the CPU, so that the
// `-m "not gpu"` tests can compare the very same source with scipy / long-double libm.
// Test infrastructure; built on demand by tests/test_hostmath.py (needs hipcc, no GPU).
#include "rvll_math.h"
#include "rvll_special.h"
#define HM extern "C" __attribute__((visibility("default")))
HM void hm_sincos(const double* x, long n, double* s, double* c) { for (long i = 0; i < n; ++i) rvll::sincos_f64(x[i], s[i], c[i]); }
HM void hm_ndtri(const double* p, long n, double* out) { for (long i = 0; i < n; ++i) out[i] = rvll::ndtri_f64(p[i]); }
HM void hm_beta_ppf(const double* q, long n, double a, double b, double lbeta, double* out) { for (long i = 0; i < n; ++i) out[i] = rvll::beta_ppf(q[i], a, b, lbeta); }
HM void hm_gamma_ppf(const double* q, long n, double alpha, double beta, double lgam, double* out) { for (long i = 0; i < n; ++i) out[i] = rvll::gamma_ppf(q[i], alpha, beta, lgam); }
HM void hm_alpha_ppf(const double* q, long n, double a, double phi_a, double* out) { for (long i = 0; i < n; ++i) out[i] = rvll::alpha_ppf(q[i], a, phi_a); }
HM void hm_betainc(const double* x, long n, double a, double b, double lbeta, double* out) { for (long i = 0; i < n; ++i) out[i] = rvll::betainc_lower(a, b, x[i], lbeta); }
HM void hm_beta_table(double a, double b, double lbeta, double* z, double* dz) {
    const double h = 2. * rvll::kTableU / (rvll::kTableN - 1);
    for (int i = 0; i < rvll::kTableN; ++i) rvll::beta_table_node(a, b, lbeta, -rvll::kTableU + i * h, z[i], dz[i]); }
HM void hm_beta_ppf_table(const double* q, long n, double a, double b, double lbeta, const double* z, const double* dz, double* out) {
    for (long i = 0; i < n; ++i) out[i] = rvll::beta_ppf_table(q[i], a, b, lbeta, z, dz); }
HM void hm_gamma_table(double alpha, double lgam, double* z, double* dz) {
    const double h = 2. * rvll::kTableU / (rvll::kTableN - 1);
    for (int i = 0; i < rvll::kTableN; ++i) rvll::gamma_table_node(alpha, lgam, -rvll::kTableU + i * h, z[i], dz[i]); }
HM void hm_gamma_ppf_table(const double* q, long n, double alpha, double beta, double lgam, const double* z, const double* dz, double* out) {
    for (long i = 0; i < n; ++i) out[i] = rvll::gamma_ppf_table(q[i], alpha, beta, lgam, z, dz); }
HM int hm_table_n(void) { return rvll::kTableN; }
