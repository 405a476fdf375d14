/*
 * rvll_oracle.c — CPU ORACLE for the RV log-likelihood hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (evidence_amd/ -> librvll.so -> HIP kernels) never calls into
 * this file and has no CPU fallback.
 *
 * It is a from-scratch C restatement of the reference algorithm, written to
 * follow the reference's floating-point operation order (same libm calls,
 * no FMA contraction, numpy-style pairwise sums) so that it agrees with the
 * reference to rounding level.  Parity is pinned: tests/test_oracle_golden.py
 * checks it against golden vectors produced by importing the reference in the
 * build container (tests/golden/gen_golden.py) and, when oracle/_ref is built,
 * against the reference's own trueanomaly.c compiled from /root/reference.
 *
 * Reference lines restated (paths relative to the reference checkout):
 *   rvo_trueanomaly   evidence/rvmodel/trueanomaly.c:8-41
 *   planet_decode     evidence/rvmodel/__init__.py:412-456   (modelk, parameters)
 *   planet_rv         evidence/rvmodel/__init__.py:458-463   (modelk, curve)
 *   kepler sum        evidence/rvmodel/__init__.py:369-383   (kep_rv)
 *   drift             evidence/rvmodel/__init__.py:242-271   (drift)
 *   rvo_loglike_one   evidence/rvmodel/__init__.py:173-217   (log_likelihood)
 *   gauss_logl        evidence/rvmodel/__init__.py:76-80     (BaseModel.logL)
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, optional -fopenmp).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "rvll.h"

#define RVO_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* Kepler solver: restates evidence/rvmodel/trueanomaly.c:8-41.              */
/* Newton from E = M (unreduced), stop when the last step is <= tol, at      */
/* least one step; ecc clamped to 0.99 inside the solver only; on reaching   */
/* itmax the call returns -1 immediately and leaves nu[i..n) untouched.      */
/* iters (optional) receives the per-element Newton step counts.             */
/* Conditioning probe (tests only): sin / cos inside the Newton loop are multiplied by 1 + rvo_trig_perturb.  At 0 (the
 * default) the arithmetic is the reference's; at +-2^-53 it is what a libm that rounds the other way would give — where
 * the result then moves by more than the parity bar, the reference's own value is an accident of its libm's last bit
 * (Newton from E = M at the 0.99 clamp wanders before it settles, and the stop |dE| <= 1e-4 leaves E anywhere within
 * ~1e-4 of the root where f' ~ 0.01). */
static double rvo_trig_perturb = 0.;
RVO_EXPORT void rvo_set_trig_perturb(double p) { rvo_trig_perturb = p; }

RVO_EXPORT int rvo_trueanomaly(const double* M, int n, double ecc, double* nu,
                               int itmax, double tol, int32_t* iters)
{
    if (ecc > 0.99) ecc = 0.99;                      /* trueanomaly.c:11-12 */
    for (int i = 0; i < n; ++i) {
        const double m = M[i];
        double cur = m, prev = m;                    /* :17-18 */
        int steps = 0;
        while (fabs(cur - prev) > tol || steps == 0) {   /* :21 */
            prev = cur;
            const double f  = cur - ecc * sin(cur) - m;  /* :25 */
            const double fp = 1 - ecc * cos(cur);        /* :26 */
            cur = prev - f / fp;                         /* :29 */
            steps += 1;
            if (steps >= itmax) {                        /* :32-33 */
                if (iters) iters[i] = steps;
                return -1;
            }
        }
        if (iters) iters[i] = steps;
        nu[i] = 2. * atan(sqrt((1. + ecc) / (1. - ecc)) * tan(cur / 2.));  /* :36 */
    }
    return 0;
}

/* The same solver with sin / cos nudged (rvo_set_trig_perturb): a separate function so that the one above compiles to
 * exactly what it did — its bits depend even on whether the compiler fuses the sin and cos calls into one sincos. */
static int trueanomaly_nudged(const double* M, int n, double ecc, double* nu, int itmax, double tol)
{
    const double pert = 1. + rvo_trig_perturb;
    if (ecc > 0.99) ecc = 0.99;
    for (int i = 0; i < n; ++i) {
        const double m = M[i];
        double cur = m, prev = m;
        int steps = 0;
        while (fabs(cur - prev) > tol || steps == 0) {
            prev = cur;
            const double f  = cur - ecc * (sin(cur) * pert) - m;
            const double fp = 1 - ecc * (cos(cur) * pert);
            cur = prev - f / fp;
            steps += 1;
            if (steps >= itmax) return -1;
        }
        nu[i] = 2. * atan(sqrt((1. + ecc) / (1. - ecc)) * tan(cur / 2.));
    }
    return 0;
}

/* numpy's pairwise summation (the algorithm np.sum uses on a contiguous
 * float64 vector), so the two Sigma terms of logL round like the reference's. */
static double pairwise_sum(const double* a, long n)
{
    if (n < 8) {
        double s = 0.;
        for (long i = 0; i < n; ++i) s += a[i];
        return s;
    }
    if (n <= 128) {
        double r[8];
        long i;
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) s += a[i];
        return s;
    }
    long half = n / 2;
    half -= half % 8;
    return pairwise_sum(a, half) + pairwise_sum(a + half, n - half);
}

static inline double slot_value(const rvll_slot* s, const double* theta)
{
    return s->idx >= 0 ? theta[s->idx] : s->val;
}

typedef struct {
    double K, P, ecc, omega, ma0, epoch;
    int valid;
} planet_pars;

/* evidence/rvmodel/__init__.py:412-456 */
static planet_pars planet_decode(const rvll_planet* pl, const double* theta)
{
    planet_pars q;
    q.valid = 1;
    const double kraw = slot_value(&pl->k, theta);
    const double praw = slot_value(&pl->p, theta);
    q.K = (pl->k_kind == RVLL_K_LOGK1) ? exp(kraw) : kraw;            /* :412-415 */
    q.P = (pl->p_kind == RVLL_P_LOGPERIOD) ? exp(praw) : praw;        /* :417-420 */
    const double a = slot_value(&pl->e1, theta);
    const double b = slot_value(&pl->e2, theta);
    if (pl->ecc_kind == RVLL_ECC_SECOS_SESIN) {                       /* :425-431 */
        q.ecc = a * a + b * b;
        q.omega = atan2(b, a);
        if (q.ecc > 1) q.valid = 0;
    } else if (pl->ecc_kind == RVLL_ECC_ECOS_ESIN) {                  /* :433-439 */
        q.ecc = sqrt(a * a + b * b);
        q.omega = atan2(b, a);
        if (q.ecc > 1) q.valid = 0;
    } else {                                                          /* :441-447 */
        q.ecc = a;
        q.omega = b;
    }
    const double anom = slot_value(&pl->anom, theta);
    q.ma0 = (pl->anom_kind == RVLL_ANOM_ML0) ? anom - q.omega : anom; /* :449-454 */
    q.epoch = slot_value(&pl->epoch, theta);                          /* :456 */
    return q;
}

typedef struct {
    const rvll_layout* L;
    const double *time, *vrad, *svrad;
    const int32_t* inst;
    const double* linpar;   /* [nlinpar][Ne] */
    int Ne;
} rvo_problem;

/* scratch: 7*Ne doubles (the last Ne hold the solver's int32 step counts) */
static double rvo_loglike_one(const rvo_problem* pb, const double* theta,
                              double* scratch, int32_t* flag_out)
{
    const rvll_layout* L = pb->L;
    const int Ne = pb->Ne;
    double* rvm   = scratch;
    double* noise = scratch + Ne;
    double* ma    = scratch + 2 * (size_t)Ne;
    double* nu    = scratch + 3 * (size_t)Ne;
    double* ksum  = scratch + 4 * (size_t)Ne;
    double* term  = scratch + 5 * (size_t)Ne;
    int32_t* steps = (int32_t*)(scratch + 6 * (size_t)Ne);
    int32_t flag = 0;

    /* offsets and noise, evidence/rvmodel/__init__.py:181-192 */
    for (int j = 0; j < Ne; ++j) {
        const rvll_inst* in = &L->insts[pb->inst[j]];
        rvm[j] = 0. + slot_value(&in->offset, theta);
        if (L->has_jitter) {
            const double jit = slot_value(&in->jitter, theta);
            noise[j] = pb->svrad[j] * pb->svrad[j] + jit * jit;
        } else {
            noise[j] = pb->svrad[j] * pb->svrad[j];
        }
    }

    /* Keplerians, evidence/rvmodel/__init__.py:195-203 and :369-383 */
    if (L->nplanets > 0) {
        for (int j = 0; j < Ne; ++j) ksum[j] = 0.;
        for (int ip = 0; ip < L->nplanets; ++ip) {
            planet_pars q = planet_decode(&L->planets[ip], theta);
            if (!q.valid) {                       /* None -> -1e30, :198-203; only this bit is reported (rvll.h) */
                if (flag_out) *flag_out = RVLL_FLAG_INVALID_ORBIT;
                return -1e30;
            }
            const double w = 2 * M_PI / q.P;                           /* :459 */
            for (int j = 0; j < Ne; ++j)
                ma[j] = w * (pb->time[j] - q.epoch) + q.ma0;
            /* nu pre-zeroed (:488); return code ignored (:490-492) */
            memset(nu, 0, sizeof(double) * (size_t)Ne);
            memset(steps, 0, sizeof(int32_t) * (size_t)Ne);
            if ((rvo_trig_perturb == 0. ? rvo_trueanomaly(ma, Ne, q.ecc, nu, L->itmax, L->tol, steps)
                                        : trueanomaly_nudged(ma, Ne, q.ecc, nu, L->itmax, L->tol)) != 0)
                flag |= RVLL_FLAG_NONCONVERGED;
            /* RVLL_FLAG_WANDERED: some solve of the point took more than 8 Newton steps (rvll.h) */
            for (int j = 0; j < Ne; ++j)
                if (steps[j] > 8) { flag |= RVLL_FLAG_WANDERED; break; }
            const double ecw = q.ecc * cos(q.omega);
            for (int j = 0; j < Ne; ++j)                               /* :463 */
                ksum[j] += q.K * (cos(nu[j] + q.omega) + ecw);
        }
        for (int j = 0; j < Ne; ++j) rvm[j] += ksum[j];
    }

    /* drift, evidence/rvmodel/__init__.py:206-207 and :242-271 */
    if (L->has_drift) {
        const double lin  = slot_value(&L->drift[0], theta);
        const double quad = slot_value(&L->drift[1], theta);
        const double cub  = slot_value(&L->drift[2], theta);
        const double quar = slot_value(&L->drift[3], theta);
        const double tref = L->tref_from_data ? pb->time[0] : slot_value(&L->tref, theta);
        for (int j = 0; j < Ne; ++j) {
            const double tt = (pb->time[j] - tref) / 365.25;
            rvm[j] += lin * tt + quad * (tt * tt) + cub * pow(tt, 3.0) + quar * pow(tt, 4.0);
        }
    }

    /* linear activity terms, evidence/rvmodel/__init__.py:210-212 */
    for (int k = 0; k < L->nlinpar; ++k) {
        const double c = slot_value(&L->linpar[k], theta);
        const double* series = pb->linpar + (size_t)k * Ne;
        for (int j = 0; j < Ne; ++j) rvm[j] += c * series[j];
    }

    /* residuals + Gaussian log-L, evidence/rvmodel/__init__.py:215-217, :76-80 */
    const double cte = -0.5 * Ne * log(2 * M_PI);
    for (int j = 0; j < Ne; ++j) term[j] = log(sqrt(noise[j]));
    const double s1 = pairwise_sum(term, Ne);
    for (int j = 0; j < Ne; ++j) {
        const double res = pb->vrad[j] - rvm[j];
        term[j] = res * res / (2 * noise[j]);
    }
    const double s2 = pairwise_sum(term, Ne);
    if (flag_out) *flag_out = flag;
    return cte - s1 - s2;
}

/* Batched driver.  nthreads <= 1: serial; otherwise OpenMP over live points
 * (when built with -fopenmp).  Returns 0, or -1 on allocation failure.       */
RVO_EXPORT int rvo_loglike_batch(const rvll_layout* L,
                                 const double* time, const double* vrad,
                                 const double* svrad, const int32_t* inst, int Ne,
                                 const double* linpar_series,
                                 const double* theta, long B,
                                 double* logL, int32_t* flags, int nthreads)
{
    rvo_problem pb = { L, time, vrad, svrad, inst, linpar_series, Ne };
    const int D = L->ndim;
    int failed = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
    #pragma omp parallel num_threads(nthreads)
#endif
    {
        double* scratch = (double*)malloc(sizeof(double) * 7 * (size_t)(Ne > 0 ? Ne : 1));
        if (!scratch) {
            failed = 1;
        } else {
#ifdef _OPENMP
            #pragma omp for schedule(dynamic, 16)
#endif
            for (long b = 0; b < B; ++b) {
                int32_t f = 0;
                logL[b] = rvo_loglike_one(&pb, theta + (size_t)b * D, scratch, &f);
                if (flags) flags[b] = f;
            }
            free(scratch);
        }
    }
    return failed ? -1 : 0;
}

/* Diagnostics for the parity tests: per (planet, epoch) Newton step counts of
 * one live point, so an iteration-count flip between two implementations can
 * be located.  iters: [nplanets][Ne].                                          */
RVO_EXPORT int rvo_iteration_counts(const rvll_layout* L, const double* time, int Ne,
                                    const double* theta, int32_t* iters)
{
    double* ma = (double*)malloc(sizeof(double) * 2 * (size_t)Ne);
    if (!ma) return -1;
    double* nu = ma + Ne;
    for (int ip = 0; ip < L->nplanets; ++ip) {
        planet_pars q = planet_decode(&L->planets[ip], theta);
        const double w = 2 * M_PI / q.P;
        for (int j = 0; j < Ne; ++j) ma[j] = w * (time[j] - q.epoch) + q.ma0;
        memset(iters + (size_t)ip * Ne, 0, sizeof(int32_t) * (size_t)Ne);
        rvo_trueanomaly(ma, Ne, q.ecc, nu, L->itmax, L->tol, iters + (size_t)ip * Ne);
    }
    free(ma);
    return 0;
}

/* Keplerian curves at arbitrary times, restating evidence/rvmodel/__init__.py:343-385 (kep_rv with
 * exclude_planet) and :388-463 (modelk): planet ip contributes iff bit ip of include_mask is set.
 * out: [B][Nt].  An invalid orbit (the reference returns None, :430-431,438-439,380) gives a NaN row.      */
RVO_EXPORT int rvo_kep_rv_batch(const rvll_layout* L, const double* theta, long B, const double* times, int Nt,
                                unsigned include_mask, double* out)
{
    double* ma = (double*)malloc(sizeof(double) * 2 * (size_t)(Nt > 0 ? Nt : 1));
    if (!ma) return -1;
    double* nu = ma + Nt;
    for (long b = 0; b < B; ++b) {
        const double* th = theta + (size_t)b * L->ndim;
        double* row = out + (size_t)b * Nt;
        int valid = 1;
        for (int j = 0; j < Nt; ++j) row[j] = 0.;
        for (int ip = 0; ip < L->nplanets && valid; ++ip) {
            if (!((include_mask >> ip) & 1u)) continue;
            planet_pars q = planet_decode(&L->planets[ip], th);
            if (!q.valid) { valid = 0; break; }
            const double w = 2 * M_PI / q.P;
            for (int j = 0; j < Nt; ++j) ma[j] = w * (times[j] - q.epoch) + q.ma0;
            memset(nu, 0, sizeof(double) * (size_t)Nt);
            rvo_trueanomaly(ma, Nt, q.ecc, nu, L->itmax, L->tol, NULL);
            const double ecw = q.ecc * cos(q.omega);
            for (int j = 0; j < Nt; ++j) row[j] += q.K * (cos(nu[j] + q.omega) + ecw);
        }
        if (!valid) for (int j = 0; j < Nt; ++j) row[j] = NAN;
    }
    free(ma);
    return 0;
}

/* FIP periodogram accumulation, restating evidence/fip_criterion.py:319-339 for ONE run: rows are the
 * posterior samples of all planet models in loop order, periods NaN-padded to np_max, contrib[row] =
 * pky[k]*w_i.  Per row: f = 2*pi/x (:321), beg = searchsorted(nub, f, 'right') (:334), end =
 * searchsorted(nua, f, 'left') (:335), and every bin of the UNION of the [beg, end) ranges gets
 * fapnu[bin] -= contrib[row] once (:339: a repeated fancy index is applied once), rows in order.      */
static int fip_count_le(const double* a, int n, double v)
{
    int lo = 0, hi = n;
    if (v != v) return n;                        /* numpy sorts NaN last */
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}
static int fip_count_lt(const double* a, int n, double v)
{
    int lo = 0, hi = n;
    if (v != v) return n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
RVO_EXPORT int rvo_fip_accumulate(const double* nua, const double* nub, int nfreq, const double* periods,
                                  const double* contrib, long n_rows, int np_max, double* fapnu)
{
    int beg[16], end[16];
    if (np_max < 1 || np_max > 16) return -1;
    for (long r = 0; r < n_rows; ++r) {
        for (int j = 0; j < np_max; ++j) {
            const double f = 2 * M_PI / periods[(size_t)r * np_max + j];
            beg[j] = fip_count_le(nub, nfreq, f);
            end[j] = fip_count_lt(nua, nfreq, f);
        }
        for (int j = 0; j < np_max; ++j)
            for (int b = beg[j]; b < end[j]; ++b) {
                int seen = 0;
                for (int q = 0; q < j && !seen; ++q) seen = b >= beg[q] && b < end[q];
                if (!seen) fapnu[b] -= contrib[r];
            }
    }
    return 0;
}

RVO_EXPORT int rvo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
