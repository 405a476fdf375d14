// rvll_special.h — quantile functions behind the scipy-backed priors of the reference
// (evidence/priors.py:357-437): Normal/LogNormal (ndtri), Beta (betaincinv), Gamma
// (gammaincinv), Alpha.  Written host+device: the prior kernel uses them on the GPU and
// tests/native/hostmath.hip compiles the very same source for the CPU so that the
// `-m "not gpu"` suite can compare it with scipy.
//
// scipy's own implementations (cephes ndtri, Boost ibeta_inv / scipy igami) are third-party
// dependencies that are not part of the reference checkout; what is restated here is the
// published mathematics: Wichura's AS241 for the normal quantile, and for the incomplete
// beta/gamma inverses a bracketed Newton iteration in log space on the regularised functions,
// which are evaluated by their classical series / continued fractions (modified Lentz).
// Parity is anchored on golden vectors generated from the reference's .ppf calls
// (tests/golden/priors.npz).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "rvll_math.h"

#ifndef RVLL_HDF
#define RVLL_HDF __host__ __device__ inline
#endif

namespace rvll {

// ---- inverse normal CDF: Wichura, Algorithm AS241 (PPND16), rel. accuracy ~1e-16 ----------
RVLL_HDF double ndtri_f64(double p)
{
    if (!(p > 0.)) return p == 0. ? -INFINITY : NAN;
    if (!(p < 1.)) return p == 1. ? INFINITY : NAN;
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        double num = 2509.0809287301226727;
        num = __builtin_fma(num, r, 33430.575583588128105);
        num = __builtin_fma(num, r, 67265.770927008700853);
        num = __builtin_fma(num, r, 45921.953931549871457);
        num = __builtin_fma(num, r, 13731.693765509461125);
        num = __builtin_fma(num, r, 1971.5909503065514427);
        num = __builtin_fma(num, r, 133.14166789178437745);
        num = __builtin_fma(num, r, 3.387132872796366608);
        double den = 5226.495278852545925;
        den = __builtin_fma(den, r, 28729.085735721942674);
        den = __builtin_fma(den, r, 39307.89580009271061);
        den = __builtin_fma(den, r, 21213.794301586595867);
        den = __builtin_fma(den, r, 5394.1960214247511077);
        den = __builtin_fma(den, r, 687.1870074920579083);
        den = __builtin_fma(den, r, 42.313330701600911252);
        den = __builtin_fma(den, r, 1.0);
        return q * num / den;
    }
    double r = q < 0. ? p : 1. - p;
    r = sqrt(-log(r));
    double val;
    if (r <= 5.) {
        r -= 1.6;
        double num = 7.7454501427834140764e-4;
        num = __builtin_fma(num, r, 0.0227238449892691845833);
        num = __builtin_fma(num, r, 0.24178072517745061177);
        num = __builtin_fma(num, r, 1.27045825245236838258);
        num = __builtin_fma(num, r, 3.64784832476320460504);
        num = __builtin_fma(num, r, 5.7694972214606914055);
        num = __builtin_fma(num, r, 4.6303378461565452959);
        num = __builtin_fma(num, r, 1.42343711074968357734);
        double den = 1.05075007164441684324e-9;
        den = __builtin_fma(den, r, 5.475938084995344946e-4);
        den = __builtin_fma(den, r, 0.0151986665636164571966);
        den = __builtin_fma(den, r, 0.14810397642748007459);
        den = __builtin_fma(den, r, 0.68976733498510000455);
        den = __builtin_fma(den, r, 1.6763848301838038494);
        den = __builtin_fma(den, r, 2.05319162663775882187);
        den = __builtin_fma(den, r, 1.0);
        val = num / den;
    } else {
        r -= 5.;
        double num = 2.01033439929228813265e-7;
        num = __builtin_fma(num, r, 2.71155556874348757815e-5);
        num = __builtin_fma(num, r, 0.0012426609473880784386);
        num = __builtin_fma(num, r, 0.026532189526576123093);
        num = __builtin_fma(num, r, 0.29656057182850489123);
        num = __builtin_fma(num, r, 1.7848265399172913358);
        num = __builtin_fma(num, r, 5.4637849111641143699);
        num = __builtin_fma(num, r, 6.6579046435011037772);
        double den = 2.04426310338993978564e-15;
        den = __builtin_fma(den, r, 1.4215117583164458887e-7);
        den = __builtin_fma(den, r, 1.8463183175100546818e-5);
        den = __builtin_fma(den, r, 7.868691311456132591e-4);
        den = __builtin_fma(den, r, 0.0148753612908506148525);
        den = __builtin_fma(den, r, 0.13692988092273580531);
        den = __builtin_fma(den, r, 0.59983220655588793769);
        den = __builtin_fma(den, r, 1.0);
        val = num / den;
    }
    return q < 0. ? -val : val;
}

// ---- inverse normal CDF as scipy computes it: the algorithm of Cephes ndtri (S. L. Moshier, Cephes Math Library
// 2.1; the routine behind scipy.special.ndtri and hence scipy.stats.norm.ppf / lognorm.ppf / alpha.ppf), restated
// operation by operation: central region y + y (y^2 P0(y^2)/Q0(y^2)) scaled by sqrt(2 pi); tails x0 - x1 with
// x = sqrt(-2 ln y), x0 = x - ln(x)/x, x1 = (1/x) P(1/x)/Q(1/x) on two sub-ranges.  Plain Horner steps with a
// rounding after every multiply and add (scipy's build has no FMA) — compiled with -ffp-contract=off here.
// Why a second routine next to AS241: the reference's Alpha quantile is 1/(a - ndtri(q Phi(a)))
// (evidence/priors.py:375-376), which cancels as q -> 1, so a 1e-16 difference between two inverse-normal routines
// becomes 1e-10 at q = 1 - 1e-6 and 1e-7 at 1 - 1e-9: to follow the reference there, the same routine has to be
// evaluated.  (Coefficients: published tables of the Cephes routine.)
RVLL_HDF double ndtri_cephes(double y0)
{
    constexpr double P0[5] = {-5.99633501014107895267E1, 9.80010754185999661536E1, -5.66762857469070293439E1,
                              1.39312609387279679503E1, -1.23916583867381258016E0};
    constexpr double Q0[8] = {1.95448858338141759834E0, 4.67627912898881538453E0, 8.63602421390890590575E1,
                              -2.25462687854119370527E2, 2.00260212380060660359E2, -8.20372256168333339912E1,
                              1.59056225126211695515E1, -1.18331621121330003142E0};
    constexpr double P1[9] = {4.05544892305962419923E0, 3.15251094599893866154E1, 5.71628192246421288162E1,
                              4.40805073893200834700E1, 1.46849561928858024014E1, 2.18663306850790267539E0,
                              -1.40256079171354495875E-1, -3.50424626827848203418E-2, -8.57456785154685413611E-4};
    constexpr double Q1[8] = {1.57799883256466749731E1, 4.53907635128879210584E1, 4.13172038254672030440E1,
                              1.50425385692907503408E1, 2.50464946208309415979E0, -1.42182922854787788574E-1,
                              -3.80806407691578277194E-2, -9.33259480895457427372E-4};
    constexpr double P2[9] = {3.23774891776946035970E0, 6.91522889068984211695E0, 3.93881025292474443415E0,
                              1.33303460815807542389E0, 2.01485389549179081538E-1, 1.23716634817820021358E-2,
                              3.01581553508235416007E-4, 2.65806974686737550832E-6, 6.23974539184983293730E-9};
    constexpr double Q2[8] = {6.02427039364742014255E0, 3.67983563856160859403E0, 1.37702099489081330271E0,
                              2.16236993594496635890E-1, 1.34204006088543189037E-2, 3.28014464682127739104E-4,
                              2.89247864745380683936E-6, 6.79019408009981274425E-9};
    constexpr double s2pi = 2.50662827463100050242E0, expm2 = 0.13533528323661269189;
    if (!(y0 > 0.)) return y0 == 0. ? -INFINITY : NAN;
    if (!(y0 < 1.)) return y0 == 1. ? INFINITY : NAN;
    bool negate = true;
    double y = y0;
    if (y > 1.0 - expm2) { y = 1.0 - y; negate = false; }
    if (y > expm2) {
        y = y - 0.5;
        const double y2 = y * y;
        double num = P0[0];
        for (int i = 1; i < 5; ++i) num = num * y2 + P0[i];
        double den = y2 + Q0[0];
        for (int i = 1; i < 8; ++i) den = den * y2 + Q0[i];
        double x = y + y * (y2 * num / den);
        return x * s2pi;
    }
    double x = sqrt(-2.0 * log(y));
    const double x0 = x - log(x) / x;
    const double z = 1.0 / x;
    double num, den;
    if (x < 8.0) {
        num = P1[0];
        for (int i = 1; i < 9; ++i) num = num * z + P1[i];
        den = z + Q1[0];
        for (int i = 1; i < 8; ++i) den = den * z + Q1[i];
    } else {
        num = P2[0];
        for (int i = 1; i < 9; ++i) num = num * z + P2[i];
        den = z + Q2[0];
        for (int i = 1; i < 8; ++i) den = den * z + Q2[i];
    }
    const double x1 = z * num / den;
    x = x0 - x1;
    return negate ? -x : x;
}

// ---- regularised incomplete beta ---------------------------------------------------------
// Continued fraction of I_x(a,b), evaluated by the forward recurrence of its convergents
// A_n/B_n, renormalised by B once per double step: one reciprocal per two terms instead of the
// four divisions of the Lentz form (this loop is latency-bound on the GPU).  Converges quickly
// for x < (a+1)/(a+b+2).
RVLL_HDF double betacf(double a, double b, double x)
{
    const double qab = a + b, qap = a + 1., qam = a - 1.;
    double am = 1., bm = 1., az = 1.;
    double bz = 1. - qab * x / qap;
    for (int m = 1; m <= 400; ++m) {
        const double em = (double)m, tem = em + em;
        const double d_even = div_fast(em * (b - em) * x, (qam + tem) * (a + tem));
        const double ap = az + d_even * am;
        const double bp = bz + d_even * bm;
        const double d_odd = div_fast(-(a + em) * (qab + em) * x, (a + tem) * (qap + tem));
        const double app = ap + d_odd * az;
        const double bpp = bp + d_odd * bz;
        const double inv = recip_refined(bpp);
        const double aold = az;
        am = ap * inv;
        bm = bp * inv;
        az = app * inv;
        bz = 1.;
        if (fabs(az - aold) < 1e-16 * fabs(az)) break;
    }
    return az;
}

// I_x and 1 - I_x are both needed with RELATIVE accuracy in the tails, so the evaluation returns
// whichever of the two the fraction computes directly plus a flag, and x*pdf(x) from the same logs.
struct BetaEval { double direct; bool direct_is_lower; double xpdf; };
RVLL_HDF BetaEval betainc_eval(double a, double b, double x, double lbeta)
{
    BetaEval r;
    const double pre = exp(a * log(x) + b * log1p(-x) - lbeta);      // x^a (1-x)^b / B(a,b)
    r.xpdf = pre / (1. - x);                                         // x * pdf(x)
    if (x < (a + 1.) / (a + b + 2.)) {
        r.direct = pre * betacf(a, b, x) / a;
        r.direct_is_lower = true;
    } else {
        r.direct = pre * betacf(b, a, 1. - x) / b;
        r.direct_is_lower = false;
    }
    return r;
}
RVLL_HDF double betainc_lower(double a, double b, double x, double lbeta)
{
    if (!(x > 0.)) return 0.;
    if (!(x < 1.)) return 1.;
    const BetaEval e = betainc_eval(a, b, x, lbeta);
    return e.direct_is_lower ? e.direct : 1. - e.direct;
}

// Solve I_x(a,b) = p for p in (0, 0.5]; returns x.  Newton on ln I in u = ln x, bracketed.
RVLL_HD double betaincinv_lowerhalf(double a, double b, double p, double lbeta)
{
    // start: leading term I ~ x^a / (a B)  ->  x0 = (p a B)^(1/a), clipped into (0,1)
    double lx = (log(p) + log(a) + lbeta) / a;
    if (lx < -708.) return 2.2250738585072014e-308;   // the solution underflows; Boost (scipy) returns DBL_MIN here
    if (lx > -1e-3) lx = -1e-3;
    // a second candidate from the mean keeps Newton out of the far tail when a is large
    const double mean = a / (a + b);
    double x = exp(lx);
    if (a > 1. && x < 0.01 * mean) x = 0.5 * mean;
    double lo = 0., hi = 1.;
    int polish = 0;       // the evaluation noise (~1e-15 relative) makes the last steps jitter by a few ulp:
                          // once a step is below 1e-12 relative allow two more, then stop
    for (int it = 0; it < 64; ++it) {
        const BetaEval e = betainc_eval(a, b, x, lbeta);
        const double I = e.direct_is_lower ? e.direct : 1. - e.direct;
        if (I > p) hi = x; else lo = x;
        // F(u) = ln I - ln p ;  dF/du = x pdf(x) / I
        const double xpdf = e.xpdf;
        double xn;
        if (I > 0. && xpdf > 0.) {
            const double F = (fabs(I - p) < 0.5 * p) ? log1p((I - p) / p) : log(I / p);
            const double step = F * I / xpdf;            // du
            xn = x * exp(-step);
        } else {
            xn = -1.;
        }
        if (!(xn >= lo && xn <= hi) || !(xn > 0.)) xn = lo > 0. ? 0.5 * (lo + hi) : 0.5 * x;   // leave Newton: bisect the bracket
        const double dx = fabs(xn - x);
        x = xn;
        if (dx <= 4.5e-16 * x) break;
        if (dx <= 1e-12 * x && ++polish >= 2) break;
    }
    return x;
}

// scipy.stats.beta.ppf(q, a, b) through rv_continuous.ppf's front end (support [0, 1]).
RVLL_HDF double beta_ppf(double q, double a, double b, double lbeta)
{
    if (q == 0.) return 0.;
    if (q == 1.) return 1.;
    if (!(q > 0. && q < 1.)) return NAN;
    if (q <= 0.5) return betaincinv_lowerhalf(a, b, q, lbeta);
    // upper half: I_x(a,b) = q  <=>  I_{1-x}(b,a) = 1 - q   (1 - q is exact for q >= 0.5)
    return 1. - betaincinv_lowerhalf(b, a, 1. - q, lbeta);
}

// ---- regularised incomplete gamma -----------------------------------------------------------
// P(a,x) by its power series (x < a+1), Q(a,x) by the continued fraction (x >= a+1).
struct GammaEval { double direct; bool direct_is_lower; double xpdf; };
RVLL_HDF GammaEval gammainc_eval(double a, double x, double lgam)
{
    GammaEval r;
    const double pre = exp(a * log(x) - x - lgam);       // x^a e^-x / Gamma(a)  ( = x * pdf(x) )
    r.xpdf = pre;
    if (x < a + 1.) {
        double ap = a, del = 1. / a, sum = del;
        for (int n = 0; n < 2000; ++n) {
            ap += 1.;
            del *= x / ap;
            sum += del;
            if (fabs(del) < fabs(sum) * 1e-17) break;
        }
        r.direct = sum * pre;
        r.direct_is_lower = true;
    } else {
        const double tiny = 1e-300;
        double b = x + 1. - a, c = 1. / tiny, d = 1. / b, h = d;
        for (int i = 1; i <= 2000; ++i) {
            const double an = -i * (i - a);
            b += 2.;
            d = an * d + b; if (fabs(d) < tiny) d = tiny;
            c = b + an / c; if (fabs(c) < tiny) c = tiny;
            d = 1. / d;
            const double del = d * c;
            h *= del;
            if (fabs(del - 1.) < 1e-16) break;
        }
        r.direct = pre * h;
        r.direct_is_lower = false;
    }
    return r;
}

// scipy.special.gammaincinv(a, q): x with P(a, x) = q.
RVLL_HD double gammaincinv(double a, double q, double lgam)
{
    if (q == 0.) return 0.;
    if (q == 1.) return INFINITY;
    if (!(q > 0. && q < 1.)) return NAN;
    const bool lower = q <= 0.5;
    const double p = lower ? q : 1. - q;                 // target for P (lower) or Q (upper)
    // start: Wilson-Hilferty, else the small-x leading term P ~ x^a / Gamma(a+1)
    double x;
    {
        const double z = ndtri_f64(q);
        const double t = 1. - 1. / (9. * a) + z / (3. * sqrt(a));
        x = a * t * t * t;
        const double xs = exp((log(q) + lgam + log(a)) / a);
        if (!(t > 0.) || a < 1. || x < 0.2 * a) x = (xs < 0.9 * (a + 1.)) ? xs : fmax(x, 0.5 * a);
        if (!(x > 0.)) x = 1e-300;
    }
    double lo = 0., hi = INFINITY;
    int polish = 0;
    for (int it = 0; it < 96; ++it) {
        const GammaEval e = gammainc_eval(a, x, lgam);
        const double P = e.direct_is_lower ? e.direct : 1. - e.direct;
        const double Q = e.direct_is_lower ? 1. - e.direct : e.direct;
        const double val = lower ? P : Q;                // the quantity matched against p
        const bool x_too_big = lower ? (val > p) : (val < p);
        if (x_too_big) hi = x; else lo = x;
        const double xpdf = e.xpdf;                      // x * pdf(x)
        double xn = -1.;
        if (val > 0. && xpdf > 0.) {
            const double F = (fabs(val - p) < 0.5 * p) ? log1p((val - p) / p) : log(val / p);
            // d ln P / d ln x = x pdf / P ;  d ln Q / d ln x = - x pdf / Q
            const double du = lower ? F * val / xpdf : -F * val / xpdf;
            xn = x * exp(-du);
        }
        if (!(xn >= lo && xn <= hi) || !(xn > 0.))
            xn = (hi < INFINITY) ? (lo > 0. ? 0.5 * (lo + hi) : 0.5 * x) : 2. * x;
        const double dx = fabs(xn - x);
        x = xn;
        if (dx <= 4.5e-16 * x) break;
        if (dx <= 1e-12 * x && ++polish >= 2) break;
    }
    return x;
}

// ---- tabulated starts ---------------------------------------------------------------------------
// For a fixed prior the quantile function is tabulated ONCE (on the device, by the solvers above) in
// coordinates in which it is smooth and asymptotically linear in both tails:
//     u = logit(q)  ->  z = logit(x)  (Beta)   or   z = ln x  (Gamma),
// with the analytic slope dz/du = q(1-q) / (x(1-x) pdf(x))  (Beta)  or  q(1-q) / (x pdf(x))  (Gamma).
// Cubic Hermite interpolation on kTableN nodes over |u| <= kTableU (q in [1e-13, 1-1e-13]) gives a start
// with relative error ~1e-9; ONE log-space Newton step on the same equation the full solver uses then
// lands on the solver's answer.  Outside the table, or if that step is not tiny, the full solver runs.
constexpr int    kTableN = 4096;
constexpr double kTableU = 30.0;

RVLL_HDF void beta_table_node(double a, double b, double lbeta, double u, double& z, double& dz)
{
    if (u <= 0.) {
        const double q = 1. / (1. + exp(-u));                       // <= 0.5
        const double x = betaincinv_lowerhalf(a, b, q, lbeta);
        const BetaEval e = betainc_eval(a, b, x, lbeta);
        z = log(x) - log1p(-x);
        dz = q * (1. - q) / (e.xpdf * (1. - x));
    } else {
        const double p = 1. / (1. + exp(u));                        // 1 - q, computed without cancellation
        const double y = betaincinv_lowerhalf(b, a, p, lbeta);      // y = 1 - x
        const BetaEval e = betainc_eval(b, a, y, lbeta);
        z = log1p(-y) - log(y);
        dz = p * (1. - p) / (e.xpdf * (1. - y));
    }
}

RVLL_HDF double hermite_table(const double* zt, const double* dzt, double u)
{
    const double h = 2. * kTableU / (kTableN - 1);
    double pos = (u + kTableU) / h;
    int i = (int)pos;
    if (i < 0) i = 0;
    if (i > kTableN - 2) i = kTableN - 2;
    const double s = pos - i, s2 = s * s, s3 = s2 * s;
    return (2. * s3 - 3. * s2 + 1.) * zt[i] + (s3 - 2. * s2 + s) * h * dzt[i]
         + (-2. * s3 + 3. * s2) * zt[i + 1] + (s3 - s2) * h * dzt[i + 1];
}

// ---- verified direct interpolation ------------------------------------------------------------------
// The quantile function obeys an ODE in these coordinates, so its second derivative at a node is closed-form:
//     ln z' = ln q + ln(1-q) - a ln x - b ln(1-x) + ln B      (Beta)   =>  z'' = z' [(1-2q) - z' (a(1-x) - b x)]
//     ln z' = ln q + ln(1-q) - a ln x + x + ln Gamma(a)       (Gamma)  =>  z'' = z' [(1-2q) - z' (a - x)]
// With (z, z', z'') per node a QUINTIC Hermite interpolant has error h^6 z^(6) / 46080 ~ 1e-14 on 4096 nodes.
// rvll_set_priors measures that error once per prior — interpolant against the full solver at every interval
// midpoint, where the quintic error term peaks — and only if it is below kTableDirectTol are elements
// evaluated by interpolation alone (no incomplete-beta/gamma evaluation at all); otherwise the cubic start +
// Newton polish below is used.
constexpr double kTableDirectTol = 3e-14;        // on |dz| = relative error of x (Gamma) or of x/(1-x) (Beta)

RVLL_HD double one_minus_2q(double u)            // 1 - 2 logistic(u), without cancellation
{
    const double e = exp(-fabs(u));
    const double t = (1. - e) / (1. + e);        // tanh(|u|/2)
    return u > 0. ? -t : t;
}

RVLL_HD double beta_table_d2(double a, double b, double u, double z, double dz)
{
    const double x = 1. / (1. + exp(-z)), omx = 1. / (1. + exp(z));
    return dz * (one_minus_2q(u) - dz * (a * omx - b * x));
}

RVLL_HD double gamma_table_d2(double a, double u, double z, double dz)
{
    return dz * (one_minus_2q(u) - dz * (a - exp(z)));
}

// quintic Hermite on the node triples (z, z', z''); d2zt follows dzt in the same array (dzt[kTableN + i])
RVLL_HD double quintic_table(const double* zt, const double* dzt, double u)
{
    const double h = 2. * kTableU / (kTableN - 1);
    const double* d2zt = dzt + kTableN;
    double pos = (u + kTableU) / h;
    int i = (int)pos;
    if (i < 0) i = 0;
    if (i > kTableN - 2) i = kTableN - 2;
    const double s = pos - i, s2 = s * s, s3 = s2 * s;
    const double p5 = s3 * (10. + s * (-15. + 6. * s));                    // value weight of node i+1
    const double p1 = s + s3 * (-6. + s * (8. - 3. * s));                  // slope weight of node i
    const double p4 = s3 * (-4. + s * (7. - 3. * s));                      // slope weight of node i+1
    const double p2 = 0.5 * s2 + s3 * (-1.5 + s * (1.5 - 0.5 * s));        // curvature weight of node i
    const double p3 = s3 * (0.5 + s * (-1. + 0.5 * s));                    // curvature weight of node i+1
    return zt[i] + p5 * (zt[i + 1] - zt[i]) + h * (p1 * dzt[i] + p4 * dzt[i + 1])
         + (h * h) * (p2 * d2zt[i] + p3 * d2zt[i + 1]);
}

// logit(q) from two short logarithms; the absolute error (~2e-16 |ln|) is far below the table spacing
RVLL_HD double logit_fast(double q) { return log_pos(q) - log_pos(1. - q); }

// beta_ppf with a tabulated start
RVLL_HDF double beta_ppf_table(double q, double a, double b, double lbeta, const double* zt, const double* dzt,
                               bool direct = false)
{
    if (q == 0.) return 0.;
    if (q == 1.) return 1.;
    if (!(q > 0. && q < 1.)) return NAN;
    if (direct && zt) {
        const double ud = logit_fast(q);
        if (fabs(ud) <= kTableU) return 1. / (1. + exp(-quintic_table(zt, dzt, ud)));
    }
    const double u = log(q) - log1p(-q);
    if (zt && fabs(u) <= kTableU) {
        const double z = hermite_table(zt, dzt, u);
        const bool lower = q <= 0.5;
        // the solver's variable: x (lower half, target q) or y = 1 - x (upper half, target 1 - q)
        const double v0 = lower ? 1. / (1. + exp(-z)) : 1. / (1. + exp(z));
        const double p = lower ? q : 1. - q;
        const BetaEval e = lower ? betainc_eval(a, b, v0, lbeta) : betainc_eval(b, a, v0, lbeta);
        const double I = e.direct_is_lower ? e.direct : 1. - e.direct;
        if (I > 0. && e.xpdf > 0. && fabs(I - p) < 0.5 * p) {
            const double du = log1p((I - p) / p) * I / e.xpdf;
            if (fabs(du) <= 1e-6) {
                const double v = v0 * exp(-du);
                return lower ? v : 1. - v;
            }
        }
    }
    return beta_ppf(q, a, b, lbeta);
}

RVLL_HDF void gamma_table_node(double a, double lgam, double u, double& z, double& dz)
{
    const double q = 1. / (1. + exp(-u));
    const double omq = 1. / (1. + exp(u));
    // gammaincinv works from q <= 0.5 (P) or 1 - q (Q); give it the exact tail through its own argument form
    double x;
    if (u <= 0.) x = gammaincinv(a, q, lgam);
    else {                                                           // solve Q(a, x) = 1 - q with the exact 1 - q
        // gammaincinv(a, q) computes 1 - q itself; for u > 0 that loses the tail, so invert through the table's
        // own equation: bracketed Newton on Q with the exact target
        double lo = 0., hi = INFINITY;
        x = gammaincinv(a, q < 1. ? q : 1. - 1e-16, lgam);
        for (int it = 0; it < 60; ++it) {
            const GammaEval e = gammainc_eval(a, x, lgam);
            const double Q = e.direct_is_lower ? 1. - e.direct : e.direct;
            if (Q < omq) hi = x; else lo = x;
            double xn = -1.;
            if (Q > 0. && e.xpdf > 0.) {
                const double F = (fabs(Q - omq) < 0.5 * omq) ? log1p((Q - omq) / omq) : log(Q / omq);
                xn = x * exp(F * Q / e.xpdf);
            }
            if (!(xn >= lo && xn <= hi) || !(xn > 0.)) xn = (hi < INFINITY) ? 0.5 * (lo + hi) : 2. * x;
            const double dx = fabs(xn - x);
            x = xn;
            if (dx <= 1e-13 * x) break;
        }
    }
    const GammaEval e = gammainc_eval(a, x, lgam);
    z = log(x);
    dz = q * omq / e.xpdf;
}

RVLL_HDF double gammaincinv_table(double a, double q, double lgam, const double* zt, const double* dzt,
                                  bool direct = false)
{
    if (q == 0.) return 0.;
    if (q == 1.) return INFINITY;
    if (!(q > 0. && q < 1.)) return NAN;
    if (direct && zt) {
        const double ud = logit_fast(q);
        if (fabs(ud) <= kTableU) return exp(quintic_table(zt, dzt, ud));
    }
    const double u = log(q) - log1p(-q);
    if (zt && fabs(u) <= kTableU) {
        const double x0 = exp(hermite_table(zt, dzt, u));
        const bool lower = q <= 0.5;
        const double p = lower ? q : 1. - q;
        const GammaEval e = gammainc_eval(a, x0, lgam);
        const double P = e.direct_is_lower ? e.direct : 1. - e.direct;
        const double Q = e.direct_is_lower ? 1. - e.direct : e.direct;
        const double val = lower ? P : Q;
        if (val > 0. && e.xpdf > 0. && fabs(val - p) < 0.5 * p) {
            const double F = log1p((val - p) / p);
            const double du = lower ? F * val / e.xpdf : -F * val / e.xpdf;
            if (fabs(du) <= 1e-6) return x0 * exp(-du);
        }
    }
    return gammaincinv(a, q, lgam);
}

// scipy.stats.gamma.ppf(q, alpha, scale = 1/beta) (evidence/priors.py:424-425)
RVLL_HDF double gamma_ppf(double q, double alpha, double beta, double lgam)
{
    return gammaincinv(alpha, q, lgam) * (1.0 / beta);
}
RVLL_HDF double gamma_ppf_table(double q, double alpha, double beta, double lgam, const double* zt, const double* dzt,
                                bool direct = false)
{
    return gammaincinv_table(alpha, q, lgam, zt, dzt, direct) * (1.0 / beta);
}

// scipy.stats.alpha.ppf(q, a) = 1 / (a - ndtri(q * Phi(a)))   (Phi(a) precomputed on the host)
RVLL_HDF double alpha_ppf(double q, double a, double phi_a)
{
    if (q == 0.) return 0.;
    if (q == 1.) return INFINITY;
    if (!(q > 0. && q < 1.)) return NAN;
    return 1.0 / (a - ndtri_cephes(q * phi_a));   // the reference's own routine: the difference cancels (see above)
}

}  // namespace rvll
