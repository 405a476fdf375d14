// rvll_tile.h — device code shared by the gfx950 kernels of the RV log-likelihood hot path: prior transform
// routines, the per-(point, epoch) item, and the tile (stage, decode, items, reduce, write) every kernel form
// runs.  Included by rvll_kernels.hip (batch kernels, scalar-call server, prior kernels) and by rvll_walk.hip (the
// proposal walk, built with different code-generation flags — see csrc/Makefile).  Internal linkage throughout.
//
//   the tile fuses, for a batch of live points,
//       evidence/rvmodel/__init__.py:173-217  log_likelihood
//       evidence/rvmodel/__init__.py:343-385  kep_rv
//       evidence/rvmodel/__init__.py:388-463  modelk
//       evidence/rvmodel/trueanomaly.c:8-41   trueanomaly (Newton, tol 1e-4)
//       evidence/rvmodel/__init__.py:222-273  drift
//       evidence/rvmodel/__init__.py:59-80    logL
//   one thread per (live point, epoch) pair, Kepler iteration and sin/cos in registers, per-point reduction
//   through LDS + a cross-lane tree.
#pragma once
#include "rvll_kernels.h"
#include "rvll_math.h"
#include "rvll_special.h"

namespace rvll {

namespace {

constexpr double kTwoPi = 6.283185307179586476925286766559;   // fl(2*pi), as 2*np.pi

__device__ __forceinline__ double slot_get(const rvll_slot s, const double* th)
{
    return s.idx >= 0 ? th[s.idx] : s.val;
}

// the same for a slot staged in LDS: both candidates are read, no dependent branch
__device__ __forceinline__ double slot_lds(const rvll_slot* s, const double* th)
{
    const int idx = s->idx;
    const double v = s->val;
    const double tv = th[idx >= 0 ? idx : 0];
    return idx >= 0 ? tv : v;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// The value `off` lanes further up, for the lanes a wave_sum tree still needs at that step (lane < off), without
// the LDS crossbar: v_permlane32_swap / v_permlane16_swap (gfx950) exchange the upper half of one register with the
// lower half of another (32- or 16-lane halves), row_shl DPP shifts within a 16-lane row.  ~10 cycles per step
// instead of ~120 for a ds_bpermute pair — the reduction of a tile is a serial tail nothing else overlaps.
__device__ __forceinline__ double lanes_up_32(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const auto a = __builtin_amdgcn_permlane32_swap((unsigned)u, (unsigned)u, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
    return __builtin_bit_cast(double, ((unsigned long long)b[1] << 32) | a[1]);
}
__device__ __forceinline__ double lanes_up_16(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const auto a = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
    return __builtin_bit_cast(double, ((unsigned long long)b[1] << 32) | a[1]);
}
template <int N>
__device__ __forceinline__ double lanes_up_row(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int l = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x100 + N, 0xf, 0xf, true);          // row_shl:N
    const int h = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x100 + N, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)h << 32) | (unsigned)l);
}
// wave_sum's tree, valid in LANE 0 only (the other lanes hold partial garbage): same additions, same order
__device__ __forceinline__ double wave_sum_lane0(double v)
{
    v += lanes_up_32(v);
    v += lanes_up_16(v);
    v += lanes_up_row<8>(v);
    v += lanes_up_row<4>(v);
    v += lanes_up_row<2>(v);
    v += lanes_up_row<1>(v);
    return v;
}

// The CU-wide form's ticket word: [0, 16) the ticket count, [16, 26) the wave round the tickets start from, [26, 31) how
// eccentric the planet that asked for it is (tile_decode: atomicMax before the first ticket is drawn keeps the most
// eccentric one's round)
constexpr int kTicketBits = 16, kTicketRoundBits = 10, kTicketRoundMask = (1 << kTicketRoundBits) - 1;
static_assert(kCuLdsBudget / 8 / kWave < (size_t)kTicketRoundMask, "every wave round of a CU-wide tile has a number");
// LDS carve-up, all in units of doubles except the int tail.
constexpr int kPlanetDoubles = (int)(sizeof(rvll_planet) / sizeof(double));
constexpr int kInstDoubles   = (int)(sizeof(rvll_inst) / sizeof(double));
constexpr int kSlotDoubles   = (int)(sizeof(rvll_slot) / sizeof(double));
static_assert(sizeof(rvll_planet) % 8 == 0 && sizeof(rvll_inst) % 8 == 0 && sizeof(rvll_slot) == 16, "layout structs are copied as doubles");
struct Carve {
    int theta, pp, ins, dr, lin, acc, lay, contrib, ints, total_doubles;
};
__host__ __device__ inline int layout_doubles(int Np, int Ni, int nlin)
{
    return Np * kPlanetDoubles + Ni * kInstDoubles + (nlin + 5) * kSlotDoubles;   // + the four drift slots and tref
}
__host__ __device__ inline Carve carve(int PB, int D, int Np, int Ni, int nlin, int CH)
{
    Carve c;
    int o = 0;
    c.theta = o;   o += PB * D;
    o = (o + 1) & ~1;                       // 16-byte align the double2-read regions
    c.pp = o;      o += PB * Np * kPlanetFields;
    c.ins = o;     o += PB * Ni * 2;
    c.dr = o;      o += PB * 6;
    c.lin = o;     o += PB * nlin;
    c.acc = o;     o += PB;
    c.lay = o;     o += layout_doubles(Np, Ni, nlin);   // the layout structs, staged once per workgroup
    o = (o + 1) & ~1;
    c.contrib = o; o += CH;
    c.ints = o;    // ints: nfail[1] ticket[1] wide[1] pad[1] pflags[PB] anyfail[PB] jfail[PB*Np]
    const int nints = 4 + 2 * PB + PB * Np;
    o += (nints + 1) / 2;
    c.total_doubles = o;
    return c;
}

// ---------------------------------------------------------------------------
// prior transform (device functions shared by the prior kernels and the fused cube -> log-L kernel)
// ---------------------------------------------------------------------------

// Piecewise-linear inverse CDF on a host-built grid; the semantics of
// scipy.interpolate.interp1d(cdf, x)(q) (which evaluates 1-D linear tables through
// numpy.interp) as used by priors.py:118-124 and friends.
__device__ double table_ppf(const rvll_prior& pr, double q)
{
    const int n = pr.table_n;
    const double* xp = pr.table_cdf;
    const double* fp = pr.table_x;
    const bool wrapped = pr.args[2] != 0.;
    if (wrapped) {                       // scipy rv_continuous.ppf front end
        if (q == 0.) return pr.args[0];
        if (q == 1.) return pr.args[1];
        if (!(q > 0. && q < 1.)) return NAN;
    }
    if (!(q >= xp[0] && q <= xp[n - 1])) return NAN;   // interp1d raises ValueError here
    // last j with xp[j] <= q
    int lo = 0, hi = n;                  // invariant: xp[lo] <= q, (hi == n or xp[hi] > q)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xp[mid] <= q) lo = mid; else hi = mid;
    }
    const int j = lo;
    double y;
    if (j == n - 1) y = fp[j];
    else if (xp[j] == q) y = fp[j];
    else {
        const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
        y = slope * (q - xp[j]) + fp[j];
        if (isnan(y)) {
            y = slope * (q - xp[j + 1]) + fp[j + 1];
            if (isnan(y) && fp[j] == fp[j + 1]) y = fp[j];
        }
    }
    return pr.table_post ? pow(10., y) : y;
}

// Forced-identifiability transform of pypolychord's SortedUniformPrior / LogSortedUniformPrior
// (evidence/priors.py:462-467, grouped call in evidence/polychord/__init__.py:145-160):
//   t[N-1] = x[N-1]^(1/N),  t[n] = x[n]^(1/(n+1)) t[n+1]   over the group's members in
// parameter order, then a + (b-a) t  (or a (b/a)^t).  Every member of kind `kind` belongs to
// the one group, and the bounds of the LAST member are used, as the wrapper does.
__device__ double sorted_prior(const rvll_prior* priors, int D, const double* cube_row, int d, int kind)
{
    int rank = 0, last = d;
    for (int k = 0; k < D; ++k)
        if (priors[k].kind == kind) { if (k < d) ++rank; last = k; }
    double t = 1.;
    int n = rank;
    for (int k = d; k < D; ++k)
        if (priors[k].kind == kind) { t *= pow(cube_row[k], 1.0 / (double)(n + 1)); ++n; }
    const double lo = priors[last].args[0], hi = priors[last].args[1];
    return kind == RVLL_PRIOR_SORTED_UNIFORM ? lo + (hi - lo) * t : lo * pow(hi / lo, t);
}

__device__ __forceinline__ bool prior_is_heavy(int kind) { return kind == RVLL_PRIOR_BETA || kind == RVLL_PRIOR_GAMMA; }

// Light kinds: a handful of instructions (or one table search) for parameter d of one cube row.
__device__ double prior_light(const rvll_prior* priors, int D, const double* cube_row, int d)
{
    const rvll_prior& pr = priors[d];
    const double q = cube_row[d];
    switch (pr.kind) {
    case RVLL_PRIOR_UNIFORM:             // priors.py:41-42
        return pr.args[0] + (pr.args[1] - pr.args[0]) * q;
    case RVLL_PRIOR_JEFFREYS:            // priors.py:62-63
        return pr.args[0] * pow(pr.args[1] / pr.args[0], q);
    case RVLL_PRIOR_MODJEFFREYS:         // priors.py:82-83
        return pr.args[0] * pow(1 + pr.args[1] / pr.args[0], q) - pr.args[0];
    case RVLL_PRIOR_UNIFORMFREQUENCY:    // priors.py:100-101
        return pr.args[0] / (1 - q * (pr.args[1] - pr.args[0]) / pr.args[1]);
    case RVLL_PRIOR_NORMAL:              // stats.norm.ppf: loc + scale*ndtri(q)
        return (q >= 0. && q <= 1.) ? ndtri_cephes(q) * pr.args[1] + pr.args[0] : NAN;
    case RVLL_PRIOR_LOGNORMAL:           // stats.lognorm.ppf: loc + scale*exp(s*ndtri(q))
        return (q >= 0. && q <= 1.) ? exp(pr.args[0] * ndtri_cephes(q)) * pr.args[2] + pr.args[1] : NAN;
    case RVLL_PRIOR_TRUNCRAYLEIGH: {     // priors.py:249-252
        const double sg = pr.args[0], xm = pr.args[1];
        const double A = 1 - exp(-(xm * xm) / (2 * (sg * sg)));
        return sqrt(-2 * (sg * sg) * log(1 - (q * A))); }
    case RVLL_PRIOR_TABLE:
        return table_ppf(pr, q);
    case RVLL_PRIOR_ALPHA:               // stats.alpha.ppf(q, a); args[1] = Phi(a)
        return alpha_ppf(q, pr.args[0], pr.args[1]);
    case RVLL_PRIOR_SORTED_UNIFORM:
    case RVLL_PRIOR_SORTED_LOGUNIFORM:
        return sorted_prior(priors, D, cube_row, d, pr.kind);
    default:
        return NAN;
    }
}

// Iterative quantiles (bracketed Newton on the regularised incomplete beta / gamma) from the device-built
// start table (table_cdf / table_x hold z and dz/du of this prior, or null).
__device__ double prior_heavy(const rvll_prior& pr, double q)
{
    const bool direct = pr.table_post != 0;   // rvll_set_priors verified the quintic interpolant of this prior
    if (pr.kind == RVLL_PRIOR_BETA)      // stats.beta.ppf(q, a, b); args[2] = ln B(a,b)
        return beta_ppf_table(q, pr.args[0], pr.args[1], pr.args[2], pr.table_cdf, pr.table_x, direct);
    // stats.gamma.ppf(q, alpha, scale=1/beta); args[2] = ln Gamma(alpha)
    return gamma_ppf_table(q, pr.args[0], pr.args[1], pr.args[2], pr.table_cdf, pr.table_x, direct);
}

// The same quantile for the kernels that carry a whole log-L tile next to the prior stage (the one-launch
// cube -> log-L form, the walk): only the verified-table evaluation — two short logarithms, one quintic, one exp —
// so that the solvers' registers are not carved out of the tile's budget (inlined they cost the one-launch form 166
// spilled VGPRs and held the walk at 2 waves per SIMD).  Same operations as prior_heavy's direct branch, hence the
// same bits.  What it cannot take — a prior whose table failed its check, |logit q| beyond umax, q on the boundary
// or outside (0, 1) — is reported through `deferred` and redone by the host with the full routines.
__device__ __forceinline__ double prior_heavy_slim(const rvll_prior& pr, double q, double umax, bool& deferred)
{
    if (pr.table_post != 0 && q > 0. && q < 1.) {
        const double ud = logit_fast(q);
        if (fabs(ud) <= umax) {
            const double z = quintic_table(pr.table_cdf, pr.table_x, ud);
            return pr.kind == RVLL_PRIOR_BETA ? 1. / (1. + exp(-z)) : exp(z) * (1.0 / pr.args[1]);
        }
    }
    deferred = true;
    return NAN;
}

// One standard normal from two counter-based uniforms (Box-Muller, the cosine branch): u1 in [0, 1) so 1 - u1 is in
// (0, 1] — log_pos (rvll_math.h: a third of the library log's instructions) takes it.  Every form of the proposal walk
// (rvll_walk.hip, rvll_rounds.hip) draws its directions through this one function, so every form sees the same numbers.
__device__ __forceinline__ double walk_normal(unsigned long long seed, unsigned long long ctr)
{
    const double u1 = uniform01(seed, ctr), u2 = uniform01(seed, ctr + 1);
    double sn, cs;
    sincos_f64(kTwoPi * u2, sn, cs);
    return sqrt(-2. * log_pos(1. - u1)) * cs;
}

// Per-block LDS views handed to eval_item (plain pointers; the kernel arguments
// themselves are passed by reference so they stay in the scalar kernarg segment).
struct ItemCtx {
    const double* pp;
    const double* ins;
    const double* dr;
    const double* lin;
    int* nfail;
    int* anyfail;
    int* jfail;
    const int* wide;      // != 0: some planet of the tile has |M| beyond 2^48 (absurd period): no solve takes the shortcut
    int* pflags;          // per-point flag bits (RVLL_FLAG_WANDERED is set from the solver's second loop)
};


// One (point, epoch) item: returns res^2 / (2 var) (the ln sqrt(var) of rvmodel:80 is summed per point by tile_logdet).
// FAILCHECK: honour the itmax marks of earlier solves (nu = 0 from the first failing epoch of that planet on).  The
// first pass over a tile runs without it: the marks can only come from that very pass, and every point that got one
// is re-evaluated with FAILCHECK afterwards (3b of loglike_tile) — so the normal path carries no reads of the marks.
// EXTRAS = false: a model without drift and without linear activity terms (the kernel never looks at those switches)
// CR: the redo pass of points with a solve that WANDERED (more than kSafeSteps Newton steps): such a solve is done again from
// its start with correctly rounded sin / cos (rvll_math.h, sincos_cr).  Where the reference's iteration wanders, where it stops
// hangs on the last bit of sin / cos, and glibc's are correctly rounded nearly always: 90 of 20000 points at e = 0.95 .. 0.9925
// were beyond 1e-10 of the reference arithmetic with the ~1.2-ulp kernels, 2 with this (profiles/r04_high_ecc_parity.txt).  It lives in the
// redo pass because inside the first pass its mere presence cost every launch 3.7 % (63.3 -> 65.6 us: scalar registers).
// NP: the planet count as a compile-time constant (0: a.Np at run time).  With it the planet loop unrolls and the per-planet LDS
// addresses fold; instantiated for the lean fp64 batch kernels at one and three planets (rvll_kernels.hip): 63.5 -> 62.6 us at
// cfg3 on the final tree of round 4 (it measured SLOWER on the tree before the redo pass got its cold-path hints:
// profiles/r04_isa_budget_addendum.txt).  Same operations in the same order: same bits.
template <int PREC, bool FAILCHECK, bool EXTRAS = true, bool CR = false, int NP = 0>
__device__ __forceinline__ double eval_item(const LoglikeArgs& a, const ItemCtx& cx, int pl, int j)
{
    const double t  = a.t[j];
    const double y  = a.y[j];
    const double s2 = a.s2[j];
    const int    in = a.inst[j];

    const double2 oj = *reinterpret_cast<const double2*>(cx.ins + (pl * a.Ni + in) * 2);
    double rvm = 0. + oj.x;                                   // rvmodel:187
    const double var = s2 + oj.y;                             // rvmodel:189-192 (oj.y = jitter^2 or 0)

    const int Np = NP > 0 ? NP : a.Np;
    if (Np > 0) {
        const bool wide = __builtin_amdgcn_readfirstlane(*cx.wide) != 0;
        bool point_failed = false;
        if constexpr (FAILCHECK) point_failed = cx.anyfail[pl] != 0;
        double ksum = 0.;
#pragma clang loop unroll_count(NP > 0 ? NP : 1)
        for (int ip = 0; ip < Np; ++ip) {
            const double* P = cx.pp + (pl * Np + ip) * kPlanetFields;
            const double2 p01 = *reinterpret_cast<const double2*>(P);       // w, epoch
            const double2 p23 = *reinterpret_cast<const double2*>(P + 2);   // ma0, ec
            const double2 p45 = *reinterpret_cast<const double2*>(P + 4);   // A=K cos w, Bq=K q sin w
            const double  C0  = P[6];                                       // K e cos w
            const double ec = p23.y;
            double rv;
            if (FAILCHECK && point_failed && j >= cx.jfail[pl * Np + ip]) {
                rv = p45.x + C0;            // nu left at 0 (rvmodel:488, trueanomaly.c:32-33)
            } else if constexpr (PREC == RVLL_PREC_FP64) {
                // mean anomaly, rvmodel:459 — two roundings in (t-epoch), then mul, then add
                const double M = p01.x * (t - p01.y) + p23.x;
                // Newton, trueanomaly.c:17-33 — op-by-op, no contraction
                double E = M, s, c, dE = __builtin_inf();
#ifdef RVLL_LOCAL_CONSTS
                const SincosConsts kc = sincos_consts_pinned();     // loaded here, live through the loop (rvll_math.h)
#else
                const SincosConsts kc = sincos_consts();
#endif
                // The first kSafeSteps iterations: an iterate cannot leave the range of the short sin / cos reduction
                // (2^50, rvll_math.h) in fewer than eight steps — f' >= 0.01 and |f| <= |E - M| + e bound |E_k - M| by
                // 99 (101^k - 1) / 100, and |M| < 2^48 by the bound the decode step takes per planet — so these run on it
                // unconditionally (the flag is wave-uniform: a scalar branch).  Nearly every
                // solve ends here (cfg3's priors: 2.9 steps on average, 0.01 % of the waves go past eight).
                const bool shortcut = a.itmax > kSafeSteps && !wide;  // wave-uniform
                if (shortcut) {
                    int steps = 0;
                    do {
                        sincos_f64(E, s, c, kc);
                        const double f  = E - ec * s - M;
                        const double fp = 1 - ec * c;
                        const double En = E - div_exact(f, fp);        // == f / fp, correctly rounded
                        dE = En - E;
                        E = En;
                        ++steps;
                    } while (fabs(dE) > a.tol && steps < kSafeSteps);
                }
                bool hit_itmax = false, wander = false;
                // (Both this test and the redo pass of the tile are marked UNLIKELY, and that is worth 4 % of every launch: the
                // register allocator weighs a value by the estimated frequency of the blocks that use it, and with the correctly
                // rounded sin / cos in the redo pass — ~1000 instructions with scalar needs of their own — it spilled scalars
                // of the item loop instead of theirs: 63.3 -> 65.9 us, 7 -> 31 v_readlane per planet trip.  With the hints:
                // 63.3 us, 2 v_readlane.  profiles/r04_high_ecc_parity.txt)
                if (__builtin_expect(!(fabs(dE) <= a.tol), 0)) {
                    // Still going after eight steps (or not started: a tiny itmax, an absurd |M|).  At the eccentricity
                    // clamp the reference's iteration is thrown out to |E| ~ 1e9 .. 1e22 and finds its way back in
                    // 30 - 350 steps (rvll_math.h, sincos_any): from here on every sin / cos is reduced the long way
                    // where its argument needs it — the same arithmetic, bit for bit, wherever it does not.
#ifdef RVLL_AB_PRIO               // measured (round 3): raising the wave's priority in here changes nothing (69.4 vs 69.1 us)
                    __builtin_amdgcn_s_setprio(2);
#endif
                    int steps = shortcut ? kSafeSteps : 0;
                    if constexpr (CR) { E = M; steps = 0; }       // (the first eight steps seed where it ends as much as the later ones)
                    do {
#ifdef RVLL_CR_FAKE                // (measurement builds only: the redo pass with the ordinary sin / cos — what the correctly rounded pair costs)
                        sincos_any(E, s, c, kc);
#else
                        if constexpr (CR) sincos_cr(E, s, c);
                        else              sincos_any(E, s, c, kc);
#endif
                        const double f  = E - ec * s - M;
                        const double fp = 1 - ec * c;
                        const double En = E - div_exact(f, fp);
                        dE = En - E;
                        E = En;
                        ++steps;
                    } while (fabs(dE) > a.tol && steps < a.itmax);
                    hit_itmax = steps >= a.itmax;
                    // more than kSafeSteps steps: the iteration wandered (include/rvll.h, RVLL_FLAG_WANDERED)
                    if (steps > kSafeSteps) {
                        atomicOr(&cx.pflags[pl], RVLL_FLAG_WANDERED);
                        if constexpr (!CR) { if (a.cr_redo) { atomicOr(cx.nfail, 2); wander = true; } }   // the tile redoes THIS item (3b)
                    }
#ifdef RVLL_AB_PRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
                }
                if (hit_itmax) {
                    atomicMin(&cx.jfail[pl * Np + ip], j);
                    atomicOr(&cx.anyfail[pl], 1);
                    atomicOr(cx.nfail, 1);
                    rv = p45.x + C0;
                } else {
                    // (s, c) are at the previous iterate; the accepted step |dE| <= tol:
                    // rotate instead of a third range reduction.
                    if (a.tol <= 1e-3) rotate_small(dE, s, c);
                    else               sincos_any(E, s, c, kc);
                    // K (cos(nu+w) + e cos w) with cos nu = (cos E - e)/(1 - e cos E),
                    // sin nu = sqrt(1-e^2) sin E/(1 - e cos E)  == trueanomaly.c:36 + rvmodel:463
                    const double den = __builtin_fma(-ec, c, 1.0);
                    const double num = __builtin_fma(p45.x, c - ec, -(p45.y * s));
                    rv = div_fast(num, den) + C0;
                    // the first pass leaves an item with a wandering solve as NaN: the redo pass (3b) finds the item by it and
                    // does it again with correctly rounded sin / cos — that item, not the 200 of its point: in a sampler's
                    // candidates, unlike in prior draws, points with a planet at e >= 0.97 are common, and redoing whole points
                    // cost the walk a third of its rate (profiles/r04_high_ecc_parity.txt)
                    if constexpr (!CR && !FAILCHECK) { if (__builtin_expect(wander, 0)) rv = __builtin_nan(""); }
                }
            } else {
                // reduced precision: phase in fp64 (|M| ~ 1e4 rad, rvmodel:459), reduced to [-pi, pi]
                // in fp64, then the same Newton rule (start E = M, stop |dE| <= tol, >= 1 step) in fp32
                const double M = p01.x * (t - p01.y) + p23.x;
                const float Mf = reduce_2pi_to_f32(M);
                const float ecf = (float)ec, tolf = (float)a.tol;
                float E = Mf, s, c, dE;
                int steps = 0;
                do {
                    sincos_f32(E, s, c);
                    const float f  = E - ecf * s - Mf;
                    const float fp = 1.0f - ecf * c;
                    const float En = E - div_f32(f, fp);
                    dE = En - E;
                    E = En;
                    ++steps;
                } while (fabsf(dE) > tolf && steps < a.itmax && steps < kF32Steps);
                if (fabsf(dE) > tolf && steps < a.itmax) {
                    // not settled after kF32Steps: at the eccentricity clamp the iteration wanders far outside what a
                    // float (or its one-step reduction) can follow — such a solve is done in double from the start,
                    // as the parity mode does it (rare: cfg3's priors, 0.01 % of the waves)
                    double Ed = M, sd, cd, dd;
                    const SincosConsts kc = sincos_consts();
                    steps = 0;
                    do {
                        sincos_any(Ed, sd, cd, kc);
                        const double f  = Ed - ec * sd - M;
                        const double fp = 1 - ec * cd;
                        const double En = Ed - div_exact(f, fp);
                        dd = En - Ed;
                        Ed = En;
                        ++steps;
                    } while (fabs(dd) > a.tol && steps < a.itmax);
                    if (steps > kSafeSteps) atomicOr(&cx.pflags[pl], RVLL_FLAG_WANDERED);
                    if (steps < a.itmax) {
                        sincos_any(Ed, sd, cd, kc);
                        rv = div_fast(__builtin_fma(p45.x, cd - ec, -(p45.y * sd)), __builtin_fma(-ec, cd, 1.0)) + C0;
                    }
                } else if (steps < a.itmax) {
                    if (steps > kSafeSteps) atomicOr(&cx.pflags[pl], RVLL_FLAG_WANDERED);
                    sincos_f32(E, s, c);
                    const float den = __builtin_fmaf(-ecf, c, 1.0f);
                    const float num = __builtin_fmaf((float)p45.x, c - ecf, -((float)p45.y * s));
                    rv = (double)(div_f32(num, den) + (float)C0);
                }
                if (steps >= a.itmax) {
                    atomicMin(&cx.jfail[pl * Np + ip], j);
                    atomicOr(&cx.anyfail[pl], 1);
                    atomicOr(cx.nfail, 1);
                    rv = p45.x + C0;
                }
            }
            ksum += rv;                                                     // rvmodel:383
        }
        rvm += ksum;                                                        // rvmodel:199
    }

    if constexpr (EXTRAS) {
        if (a.has_drift) {                                                  // rvmodel:242-271
            const double* d = cx.dr + pl * 6;
            const double tt = (t - d[4]) * (1.0 / 365.25);
            const double t2 = tt * tt;
            rvm += d[0] * tt + d[1] * t2 + d[2] * (t2 * tt) + d[3] * (t2 * t2);
        }
        for (int k = 0; k < a.nlin; ++k)                                    // rvmodel:210-212
            rvm += cx.lin[pl * a.nlin + k] * a.linpar[(size_t)k * a.Ne + j];
    }

    const double res = y - rvm;                                             // rvmodel:215
#ifdef RVLL_AB_NO_LOGDET              // (measurement builds only: the normalisation term per item, as up to round 2)
    if constexpr (PREC == RVLL_PREC_FP32) {
        const float rf = (float)res, vf = (float)var;
        return (double)(0.5f * __logf(vf) + div_f32(rf * rf, 2.0f * vf));
    } else {
        return 0.5 * log_pos(var) + div_fast(res * res, 2 * var);           // rvmodel:80
    }
#else
    // res^2 / (2 var): the point's ln sqrt(var) terms are summed by tile_logdet, once per point
    if constexpr (PREC == RVLL_PREC_FP32) {
        const float rf = (float)res, vf = (float)var;
        return (double)div_f32(rf * rf, 2.0f * vf);
    } else {
        return div_fast(res * res, 2 * var);                                // rvmodel:80
    }
#endif
}

// ---- the reduced-precision modes, two items per lane (RVLL_PREC_MIXED / RVLL_PREC_FP32; a.itmax > kF32Steps) -----------------
// The items (plA, jA) and (plB, jB) side by side: the phase of each in fp64 as eval_item has it, the Newton iteration of the two
// as one packed fp32 iteration (rvll_math.h, sincos_f32x2) that runs until every item of the WAVE has met the stop rule — a
// scalar branch instead of a per-lane one; an item that has met it and takes further steps only comes closer to its root —
// and at most kF32Steps steps: an item still moving then is solved in double from its start, as eval_item does it.  Not a
// parity mode (tests/test_gpu_precision.py holds its distance from the fp64 path).
template <int PREC, bool EXTRAS>
__device__ __forceinline__ double pair_tail(const LoglikeArgs& a, const ItemCtx& cx, int pl, int j, double t, double y, double var, double rvm)
{
    if constexpr (EXTRAS) {
        if (a.has_drift) {                                                  // rvmodel:242-271
            const double* d = cx.dr + pl * 6;
            const double tt = (t - d[4]) * (1.0 / 365.25);
            const double t2 = tt * tt;
            rvm += d[0] * tt + d[1] * t2 + d[2] * (t2 * tt) + d[3] * (t2 * t2);
        }
        for (int k = 0; k < a.nlin; ++k)                                    // rvmodel:210-212
            rvm += cx.lin[pl * a.nlin + k] * a.linpar[(size_t)k * a.Ne + j];
    }
    const double res = y - rvm;                                             // rvmodel:215
    if constexpr (PREC == RVLL_PREC_FP32) {
        const float rf = (float)res, vf = (float)var;
        return (double)div_f32(rf * rf, 2.0f * vf);
    } else {
        return div_fast(res * res, 2 * var);                                // rvmodel:80
    }
}

// one solve in double from its start (the reduced-precision modes' way out of a wandering iteration); false: itmax reached
__device__ __forceinline__ bool pair_solve_f64(const LoglikeArgs& a, const ItemCtx& cx, int pl, double M, double ec, double A, double Bq,
                                               double C0, double& rv)
{
    double Ed = M, sd, cd, dd;
    const SincosConsts kc = sincos_consts();
    int steps = 0;
    do {
        sincos_any(Ed, sd, cd, kc);
        const double f  = Ed - ec * sd - M;
        const double fp = 1 - ec * cd;
        const double En = Ed - div_exact(f, fp);
        dd = En - Ed;
        Ed = En;
        ++steps;
    } while (fabs(dd) > a.tol && steps < a.itmax);
    if (steps > kSafeSteps) atomicOr(&cx.pflags[pl], RVLL_FLAG_WANDERED);
    if (steps >= a.itmax) return false;
    sincos_any(Ed, sd, cd, kc);
    rv = div_fast(__builtin_fma(A, cd - ec, -(Bq * sd)), __builtin_fma(-ec, cd, 1.0)) + C0;
    return true;
}

template <int PREC, bool EXTRAS, int NP = 0>
__device__ __forceinline__ void eval_item_pair(const LoglikeArgs& a, const ItemCtx& cx, int plA, int jA, int plB, int jB,
                                               double& outA, double& outB)
{
    const double tA = a.t[jA], yA = a.y[jA], tB = a.t[jB], yB = a.y[jB];
    const double2 ojA = *reinterpret_cast<const double2*>(cx.ins + (plA * a.Ni + a.inst[jA]) * 2);
    const double2 ojB = *reinterpret_cast<const double2*>(cx.ins + (plB * a.Ni + a.inst[jB]) * 2);
    double rvmA = 0. + ojA.x, rvmB = 0. + ojB.x;                            // rvmodel:187
    const double varA = a.s2[jA] + ojA.y, varB = a.s2[jB] + ojB.y;         // rvmodel:189-192
    const int Np = NP > 0 ? NP : a.Np;
    const f32x2 tolf = splat2((float)a.tol);
    double ksumA = 0., ksumB = 0.;
#pragma clang loop unroll_count(NP > 0 ? NP : 1)
    for (int ip = 0; ip < Np; ++ip) {
        const double* PA = cx.pp + (plA * Np + ip) * kPlanetFields;
        const double* PB = cx.pp + (plB * Np + ip) * kPlanetFields;
        const double2 a01 = *reinterpret_cast<const double2*>(PA), a23 = *reinterpret_cast<const double2*>(PA + 2);
        const double2 b01 = *reinterpret_cast<const double2*>(PB), b23 = *reinterpret_cast<const double2*>(PB + 2);
        const double2 a45 = *reinterpret_cast<const double2*>(PA + 4), b45 = *reinterpret_cast<const double2*>(PB + 4);
        const double aC0 = PA[6], bC0 = PB[6];
        const double MA = a01.x * (tA - a01.y) + a23.x;                     // rvmodel:459, in fp64
        const double MB = b01.x * (tB - b01.y) + b23.x;
        const f32x2 Mf = {reduce_2pi_to_f32(MA), reduce_2pi_to_f32(MB)};
        const f32x2 ecf = {(float)a23.y, (float)b23.y};
        f32x2 E = Mf, s, c, dE;
        int steps = 0;
        bool more;
        auto newton = [&]() {
            sincos_f32x2(E, s, c);
            const f32x2 f  = E - ecf * s - Mf;
            const f32x2 fp = splat2(1.0f) - ecf * c;
            const f32x2 En = E - div_f32x2(f, fp);
            dE = En - E;
            E = En;
            ++steps;
            more = __builtin_amdgcn_ballot_w64(fabsf(dE.x) > tolf.x || fabsf(dE.y) > tolf.y) != 0;
        };
        do newton(); while (more && steps < kSafeSteps);
        const f32x2 dE8 = dE;                 // every item settled: within tol; else the step at kSafeSteps (0.01 % of the waves)
        if (__builtin_expect(more, 0)) { do newton(); while (more && steps < kF32Steps); }
        const bool lateA = fabsf(dE8.x) > tolf.x, lateB = fabsf(dE8.y) > tolf.y;      // more than kSafeSteps steps: WANDERED
        const bool movA = fabsf(dE.x) > tolf.x, movB = fabsf(dE.y) > tolf.y;
        // the model term of both from the packed iterate: (s, c) are at the iterate before the last step, and that step is within
        // tol for every item that goes on from here — rotated by it (h^3 / 6 <= 2e-10 dropped) instead of a third reduction
        if (a.tol <= 1e-3) {
            const f32x2 hh = splat2(0.5f) * dE * dE, s0 = s, c0 = c;
            s = fma2(c0, dE, fma2(-s0, hh, s0));
            c = fma2(-s0, dE, fma2(-c0, hh, c0));
        } else {
            sincos_f32x2(E, s, c);
        }
        const f32x2 den = fma2(-ecf, c, splat2(1.0f));
        const f32x2 Af = {(float)a45.x, (float)b45.x}, Bf = {(float)a45.y, (float)b45.y}, Cf = {(float)aC0, (float)bC0};
        const f32x2 num = fma2(Af, c - ecf, -(Bf * s));
        const f32x2 rvf = div_f32x2(num, den) + Cf;
        double rvA = (double)rvf.x, rvB = (double)rvf.y;
        if (__builtin_expect(lateA || lateB || movA || movB, 0)) {
            if (lateA && !movA) atomicOr(&cx.pflags[plA], RVLL_FLAG_WANDERED);
            if (lateB && !movB) atomicOr(&cx.pflags[plB], RVLL_FLAG_WANDERED);
            if (movA && !pair_solve_f64(a, cx, plA, MA, a23.y, a45.x, a45.y, aC0, rvA)) {
                atomicMin(&cx.jfail[plA * Np + ip], jA);
                atomicOr(&cx.anyfail[plA], 1);
                atomicOr(cx.nfail, 1);
                rvA = a45.x + aC0;
            }
            if (movB && !pair_solve_f64(a, cx, plB, MB, b23.y, b45.x, b45.y, bC0, rvB)) {
                atomicMin(&cx.jfail[plB * Np + ip], jB);
                atomicOr(&cx.anyfail[plB], 1);
                atomicOr(cx.nfail, 1);
                rvB = b45.x + bC0;
            }
        }
        ksumA += rvA;                                                       // rvmodel:383
        ksumB += rvB;
    }
    rvmA += ksumA;                                                          // rvmodel:199
    rvmB += ksumB;
    outA = pair_tail<PREC, EXTRAS>(a, cx, plA, jA, tA, yA, varA, rvmA);
    outB = pair_tail<PREC, EXTRAS>(a, cx, plB, jB, tB, yB, varB, rvmB);
}

#ifndef RVLL_DECODE_INLINE
#define RVLL_DECODE_INLINE __forceinline__
#endif
// (point, epoch) of lane `lane` in a wave round that starts at flattened item i0 (wave-uniform): the division by the
// run-time epoch count happens once per wave round on the scalar unit instead of ~20 vector instructions per item
__device__ __forceinline__ void item_of(int i0, int lane, int Ne, int& pl, int& j)
{
    const int u0 = __builtin_amdgcn_readfirstlane(i0);
    const int pl0 = u0 / Ne;
    pl = pl0;
    j = u0 - pl0 * Ne + lane;
    while (j >= Ne) { j -= Ne; ++pl; }            // at most once when Ne >= 64
}

// LDS views of one workgroup's tile (carve()).
struct TileLds {
    double *theta_s, *pp, *ins, *dr, *lin, *acc, *lay, *contrib;
    int *nfail, *ticket, *wide, *pflags, *anyfail, *jfail;
};
__device__ __forceinline__ TileLds tile_views(const LoglikeArgs& a, double* smem)
{
    const Carve cv = carve(a.PB, a.D, a.Np, a.Ni, a.nlin, a.CH);
    TileLds L;
    L.theta_s = smem + cv.theta;  L.pp = smem + cv.pp;    L.ins = smem + cv.ins;  L.dr = smem + cv.dr;
    L.lin = smem + cv.lin;        L.acc = smem + cv.acc;  L.lay = smem + cv.lay;  L.contrib = smem + cv.contrib;
    int* ints = reinterpret_cast<int*>(smem + cv.ints);
    L.nfail = ints;  L.ticket = ints + 1;  L.wide = ints + 2;  L.pflags = ints + 4;  L.anyfail = L.pflags + a.PB;  L.jfail = L.anyfail + a.PB;
    return L;
}

// 1. stage the tile's theta rows (one contiguous, coalesced span) + the layout structs + init.  Fused form: the rows
//    are unit-cube coordinates and go through the prior transform on the way in (light kinds element by element,
//    the iterative kinds compacted so that consecutive lanes all run a solve, in other waves); theta lands in LDS and
//    is written back here (256-thread tiles) or by the caller after its barrier (CU-wide form, loglike_tile).
template <int FUSED, int NT>
__device__ __forceinline__ void tile_stage(const LoglikeArgs& __restrict__ a, const TileLds& L, long long p0, int npts,
                                           unsigned long long* stamp = nullptr, const double* cube_rows = nullptr)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < npts; i += NT) { L.acc[i] = 0.; L.pflags[i] = 0; L.anyfail[i] = 0; }
    for (int i = tid; i < npts * a.Np; i += NT) L.jfail[i] = 0x7fffffff;
    if (tid == 0) { L.nfail[0] = 0; L.ticket[0] = 0; L.wide[0] = 0; }
    if constexpr (FUSED == kFusedSlim) __syncthreads();          // deferrals are OR-ed into pflags below
    if constexpr (FUSED != kFusedNone) {
        // cube_rows: the tile's unit-cube rows where the caller already holds them (the walk: in LDS), else a.cube
        const double* src = cube_rows ? cube_rows : a.cube + p0 * a.D;
        // theta_out: the CU-wide form writes it from LDS in whole rows after the caller's barrier (loglike_tile); the
        // 256-thread tiles write it element by element here (a second pass over a small tile costs them more than the
        // lone stores); a caller that holds the rows in LDS (the walk) reads theta from LDS too and gets no copy
        double* dst = (NT != kCuThreads && !cube_rows) ? a.theta_out + p0 * a.D : nullptr;
        // Lanes run over the tile's elements POINT-fastest: the 64 lanes of a wave then work on one parameter — one
        // prior kind, one path through the switch — wherever the tile has 64 points (the CU-wide form), on 64 / npts
        // parameters otherwise; parameter-fastest, every wave ran every kind of the model one after the other.  The
        // iterative kinds are dealt from the BACK of the workgroup, so that they run in other waves than the tail of
        // the light ones instead of behind it.  (Same function per element either way: same bits.)
        const int n_light = a.D - a.n_heavy;
        for (int i = tid; i < npts * n_light; i += NT) {
            const int k = i / npts, pl = i - k * npts;
            const int d = a.light_dims[k];              // parameters of one kind are neighbours in this order (rvll_set_priors)
            const double v = prior_light(a.priors, a.D, src + pl * a.D, d);
            L.theta_s[pl * a.D + d] = v;
            if (dst) dst[pl * a.D + d] = v;
        }
        for (int i = NT - 1 - tid; i < npts * a.n_heavy; i += NT) {
            const int k = i / npts, pl = i - k * npts;
            const int d = a.heavy_dims[k];
            double v;
            if constexpr (FUSED == kFusedSlim) {
                bool deferred = false;
                v = prior_heavy_slim(a.priors[d], src[pl * a.D + d], a.slim_umax, deferred);
                if (deferred) {
                    atomicOr(&L.pflags[pl], kFlagDeferred);
                    if (a.defer) atomicOr(a.defer, 1);
                }
            } else {
                v = prior_heavy(a.priors[d], src[pl * a.D + d]);
            }
            L.theta_s[pl * a.D + d] = v;
            if (dst) dst[pl * a.D + d] = v;
        }
    } else {
        const double* src = cube_rows ? cube_rows : a.theta + p0 * a.D;       // (the theta form's rows, likewise)
        for (int i = tid; i < npts * a.D; i += NT) L.theta_s[i] = src[i];
        if (stamp && tid == 0) {                    // diagnostic builds: theta has landed in this wave
            __builtin_amdgcn_s_waitcnt(0);
            *stamp = __builtin_amdgcn_s_memrealtime();
        }
    }
    // the layout structs (planets, instruments, linear-term slots: one device blob) ride along, so that the decode
    // step reads LDS only — as dependent global loads they were most of a workgroup's prologue
    for (int i = tid; i < layout_doubles(a.Np, a.Ni, a.nlin); i += NT) L.lay[i] = a.layblob[i];
}

// 2. decode per-point scalars once (rvmodel:412-456, :181-192, :242-260): one lane per (point, planet) from the
//    front of the workgroup, one lane per point for the instrument / drift / linear terms from its back, so the two
//    kinds of work sit in different waves whenever the tile leaves room
template <int NT>
__device__ RVLL_DECODE_INLINE void tile_decode(const LoglikeArgs& __restrict__ a, const TileLds& L, int npts)
{
    const int tid = threadIdx.x;
    const rvll_planet* lp = reinterpret_cast<const rvll_planet*>(L.lay);
    const rvll_inst*   li = reinterpret_cast<const rvll_inst*>(L.lay + a.Np * kPlanetDoubles);
    const rvll_slot*   ll = reinterpret_cast<const rvll_slot*>(L.lay + a.Np * kPlanetDoubles + a.Ni * kInstDoubles);
    for (int wk = tid; wk < npts * a.Np; wk += NT) {
        const int pl = wk / a.Np;
        const int k  = wk - pl * a.Np;
        const double* th = L.theta_s + pl * a.D;
        // All six slots first (one 16-byte LDS read each), then the six theta values they point at: two LDS round
        // trips instead of twelve dependent ones — measured, the slot reads were 1900 of the decode's 2800 cycles.
        const rvll_planet d = lp[k];
        const double tk = th[max(d.k.idx, 0)], tp = th[max(d.p.idx, 0)], t1 = th[max(d.e1.idx, 0)],
                     t2 = th[max(d.e2.idx, 0)], ta = th[max(d.anom.idx, 0)], te = th[max(d.epoch.idx, 0)];
        const double kraw = d.k.idx >= 0 ? tk : d.k.val;
        const double praw = d.p.idx >= 0 ? tp : d.p.val;
        const double K = d.k_kind == RVLL_K_LOGK1 ? exp(kraw) : kraw;
        const double Pd = d.p_kind == RVLL_P_LOGPERIOD ? exp(praw) : praw;
        const double e1 = d.e1.idx >= 0 ? t1 : d.e1.val;
        const double e2 = d.e2.idx >= 0 ? t2 : d.e2.val;
        double ecc, omega;
        if (d.ecc_kind == RVLL_ECC_SECOS_SESIN) {
            ecc = e1 * e1 + e2 * e2;
            omega = atan2(e2, e1);
            if (ecc > 1) atomicOr(&L.pflags[pl], RVLL_FLAG_INVALID_ORBIT);
        } else if (d.ecc_kind == RVLL_ECC_ECOS_ESIN) {
            ecc = sqrt(e1 * e1 + e2 * e2);
            omega = atan2(e2, e1);
            if (ecc > 1) atomicOr(&L.pflags[pl], RVLL_FLAG_INVALID_ORBIT);
        } else {
            ecc = e1;
            omega = e2;
        }
        const double anom = d.anom.idx >= 0 ? ta : d.anom.val;
        const double ma0 = d.anom_kind == RVLL_ANOM_ML0 ? anom - omega : anom;
        const double ec = ecc > 0.99 ? 0.99 : ecc;                  // trueanomaly.c:11-12
        // a planet this eccentric may hold a solve that wanders for hundreds of steps: the CU-wide form starts its tickets
        // at the wave round the tile's most eccentric such point begins in (loglike_tile).  Key: eccentricity above the
        // threshold (0 .. 30: ec <= 0.99) over that round, in the upper bits of the ticket word — atomicMax keeps the most
        // eccentric; no ticket has been drawn yet (the barrier after this step comes first)
#ifndef RVLL_AB_NO_ROT
        if (ec >= kLongSolveEcc)
            atomicMax(L.ticket, ((int)((ec - kLongSolveEcc) * 340.) << (kTicketBits + kTicketRoundBits)) |
                                (min((pl * a.Ne) >> 6, kTicketRoundMask) << kTicketBits));
#endif
        double so, co;
        sincos_any(omega, so, co);                                  // (omega is a free parameter: any finite value)
        const double q = sqrt((1. - ec) * (1. + ec));
        double* P = L.pp + (pl * a.Np + k) * kPlanetFields;
        P[0] = kTwoPi / Pd;
        P[1] = d.epoch.idx >= 0 ? te : d.epoch.val;
        P[2] = ma0;
        P[3] = ec;
        P[4] = K * co;
        P[5] = K * q * so;
        P[6] = K * (ecc * co);
        // a bound on |M| over the epoch table: the solver's first eight steps skip the range check of their sin / cos
        // arguments when it is below 2^48 for every planet of the tile (eval_item)
        const double ep = d.epoch.idx >= 0 ? te : d.epoch.val;
        const double mbound = fabs(kTwoPi / Pd) * fmax(fabs(a.tmax - ep), fabs(a.tmin - ep)) + fabs(ma0);
        if (!(mbound < kExcursionM)) atomicOr(L.wide, 1);
        P[7] = 0.;
    }
    for (int pl = NT - 1 - tid; pl < npts; pl += NT) {
        const double* th = L.theta_s + pl * a.D;
        for (int i = 0; i < a.Ni; ++i) {
            L.ins[(pl * a.Ni + i) * 2] = slot_lds(&li[i].offset, th);
            double j2 = 0.;
            if (a.has_jitter) { const double jit = slot_lds(&li[i].jitter, th); j2 = jit * jit; }
            L.ins[(pl * a.Ni + i) * 2 + 1] = j2;
        }
        if (a.has_drift) {
            const rvll_slot* ld = ll + a.nlin;                  // drift[0..3], tref
            double* d = L.dr + pl * 6;
            d[0] = slot_lds(&ld[0], th);
            d[1] = slot_lds(&ld[1], th);
            d[2] = slot_lds(&ld[2], th);
            d[3] = slot_lds(&ld[3], th);
            d[4] = a.tref_from_data ? a.t[0] : slot_lds(&ld[4], th);
            d[5] = 0.;
        }
        for (int k2 = 0; k2 < a.nlin; ++k2) L.lin[pl * a.nlin + k2] = slot_lds(&ll[k2], th);
    }
}

// 2b. The Gaussian normalisation of every point of the tile, sum_j ln sqrt(var_j) with var_j = sigma_j^2 + jitter^2 of
//     epoch j's instrument (rvmodel:76-80, :189-192), into acc[pl] — apart from the items, because a logarithm per item was
//     26 of an item's ~640 vector instructions and a sum of logarithms is the logarithm of a product: sixteen lanes (a
//     whole wave from 513 epochs on) share a point, lane s multiplies the variances of epochs s, s + 16, ... as
//     (mantissa product, exponent sum) — four instructions a factor, no range to worry about — takes ONE logarithm,
//     and a row (wave) tree adds the lanes' values.
//     The order is the point's own (which epochs a lane takes, the row tree), so the bits depend on nothing but the point
//     and the epoch table: the same in every kernel form, tile size and shard.  ~36 wave instructions a point at 200
//     epochs against 81 for the logarithms inside the items.
// lanes per point: a 16-lane row up to 512 epochs, a whole wave beyond (a function of the epoch count alone, so that the
// order of the sum — and with it the bits — is the same whatever the tile, the kernel form or the shard)
__device__ __forceinline__ int logdet_group(int Ne) { return Ne > 512 ? kWave : 16; }
// one lane's (mantissa product, exponent sum) -> its logarithm -> the group's sum in the group's first lane -> acc[pl]
__device__ __forceinline__ void logdet_finish(const TileLds& L, int pl, int sub, int gs, double m, int e)
{
    double v = __builtin_fma((double)e, 6.93147180559945286227e-01, log_pos(m));
    if (gs == kWave) {
        v = wave_sum_lane0(v);
    } else {
        v += lanes_up_row<8>(v);
        v += lanes_up_row<4>(v);
        v += lanes_up_row<2>(v);
        v += lanes_up_row<1>(v);
    }
    if (sub == 0) L.acc[pl] = 0.5 * v;
}
template <int NT>
__device__ __forceinline__ void tile_logdet(const LoglikeArgs& __restrict__ a, const TileLds& L, int npts)
{
    const int gs = logdet_group(a.Ne);
    const int tid = threadIdx.x, sub = tid & (gs - 1);
    for (int pl = tid / gs; pl < npts; pl += NT / gs) {
        const double* jit = L.ins + pl * a.Ni * 2 + 1;
        double m = 1.;
        int e = 0, since = 0;
#pragma unroll 4
        for (int j = sub; j < a.Ne; j += gs) {
            const double var = a.s2[j] + jit[a.inst[j] * 2];
            m *= __builtin_amdgcn_frexp_mant(var);                 // [0.5, 1): the product cannot overflow ...
            e += __builtin_amdgcn_frexp_exp(var);
            if (++since == 512) {                                  // ... and is renormalised long before it could underflow
                e += __builtin_amdgcn_frexp_exp(m);
                m = __builtin_amdgcn_frexp_mant(m);
                since = 0;
            }
        }
        logdet_finish(L, pl, sub, gs, m, e);
    }
}
// The same for a ONE-point tile whose lanes hold their epochs' sigma^2 and instrument numbers in registers already (the
// persistent scalar-call kernel: the epoch table never changes between requests, and taken from memory these loads are the
// longest chain of a scalar callback — 9.6 -> 11.3 us, profiles/r03_call_latency.txt).  The same factors in the same
// order: the same bits.
constexpr int kLogdetPre = 16;                                     // factors a lane can hold: Ne <= 16 * logdet_group(Ne)
struct LogdetPre {
    double s2[kLogdetPre];
    int inst[kLogdetPre];
    bool valid;
};
__device__ __forceinline__ void logdet_preload(const LoglikeArgs& a, LogdetPre& pre)
{
    const int gs = logdet_group(a.Ne), sub = threadIdx.x & (gs - 1);
    pre.valid = a.Ne <= kLogdetPre * gs;
#pragma unroll
    for (int u = 0; u < kLogdetPre; ++u) {
        const int j = sub + u * gs;
        const bool in = pre.valid && (int)threadIdx.x < gs && j < a.Ne;
        pre.s2[u] = in ? a.s2[j] : 1.;
        pre.inst[u] = in ? a.inst[j] : 0;
    }
}
__device__ __forceinline__ void tile_logdet_pre(const LoglikeArgs& __restrict__ a, const TileLds& L, const LogdetPre& pre)
{
    const int gs = logdet_group(a.Ne), tid = threadIdx.x, sub = tid & (gs - 1);
    if (tid >= gs) return;                                         // (a whole wave or a whole row: the tree's lanes are all here)
    const double* jit = L.ins + 1;
    double m = 1.;
    int e = 0;
#pragma unroll
    for (int u = 0; u < kLogdetPre; ++u)
        if (sub + u * gs < a.Ne) {
            const double var = pre.s2[u] + jit[pre.inst[u] * 2];
            m *= __builtin_amdgcn_frexp_mant(var);
            e += __builtin_amdgcn_frexp_exp(var);
        }
    logdet_finish(L, 0, sub, gs, m, e);
}

// Per-point partial sums of the contributions [lo[k], hi[k]) of up to four points held in contrib[.. - base], each
// in the fixed order every kernel form uses (lane-strided, then the shuffle tree), so the bits do not depend on the
// launch geometry.  Four points go through the tree together: one point's six dependent cross-lane steps are
// ~150 cycles of latency each and nothing else is runnable in a reduction phase.  Unused entries: lo == hi.
__device__ __forceinline__ void point_partials4(const double* contrib, int base, const int (&lo)[4], const int (&hi)[4],
                                                int lane, double (&v)[4])
{
    // lane-strided sums, four strides of every point in flight per pass.  Out-of-range slots add +0.0, which
    // changes no partial sum (they start at +0.0 and can never become -0.0).
    int nmax = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = 0.; nmax = max(nmax, hi[k] - lo[k]); }
    for (int t = lane; t < nmax; t += 4 * kWave) {
        double x[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = lo[k] + t + u * kWave;
                x[k][u] = i < hi[k] ? contrib[i - base] : 0.;
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (((v[k] + x[k][0]) + x[k][1]) + x[k][2]) + x[k][3];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = wave_sum_lane0(v[k]);
}

// 4. one log-L per live point
// (the walk reads a finished tile's results through this, straight from LDS)
__device__ __forceinline__ double tile_point_result(const LoglikeArgs& __restrict__ a, const TileLds& L, int pl, int& flags)
{
    int f = L.pflags[pl];
    if (L.anyfail[pl]) f |= RVLL_FLAG_NONCONVERGED;
    // an invalid orbit is -1e30 whatever the solves of the point's other planets did (the reference never runs them,
    // rvmodel:198-203): only that bit is reported
    if (f & RVLL_FLAG_INVALID_ORBIT) f &= RVLL_FLAG_INVALID_ORBIT | kFlagDeferred;
    flags = f;
    const bool invalid = (f & RVLL_FLAG_INVALID_ORBIT) != 0 && a.Np > 0;
    return invalid ? -1e30 : a.cte - L.acc[pl];                            // rvmodel:203, :78-80
}
__device__ __forceinline__ void tile_write_point(const LoglikeArgs& __restrict__ a, const TileLds& L, long long p0, int pl)
{
    int f;
    a.logL[p0 + pl] = tile_point_result(a, L, pl, f);
    if (a.flags) a.flags[p0 + pl] = f;
}

// One tile of live points [p0, p0 + npts) by one workgroup of NT threads: stage, decode, items, reduce, write.
// Shared by the batch kernels (tile = blockIdx), the walk and the scalar-call server (one point per request).
//   NT = 256, DYN = false  the tile form: four such workgroups per CU, items dealt statically, LDS windows of a.CH
//                          contributions (<= kTileWindow);
//   NT = 1024, DYN = true  the CU-wide form: one workgroup fills the CU (16 waves = the same 4 per SIMD), ALL the
//                          tile's items sit in LDS at once (a.CH >= a.PB * a.Ne, host-checked) and the waves draw
//                          64-item rounds from an LDS ticket counter, so every wave stays busy until the tile's
//                          items are gone.  Why: four independent 256-thread workgroups on a CU do not finish
//                          together — VALU issue is arbitrated by age, the oldest runs fastest — so a launch ended
//                          with 3, 2, then 1 workgroup per CU for a third of its duration at well under the 4-wave
//                          issue rate (profiles/r02_wg_trace_cfg3.txt; two launches in flight, which backfill the
//                          freed slots, ran 20 % faster per launch).
// Every point's contributions are summed in the same order in both forms — slices at point-local multiples of
// kTileWindow, each lane-strided then through the shuffle tree — so results are bit-identical across forms, tile
// sizes and shard sizes.
// TRACE: diagnostic build (launch_loglike_trace) — a few s_memrealtime stamps per workgroup go to a.trace, a
// buffer nothing else reads; no stamp executes in the product kernels.
// PRE: the caller holds its lanes' share of the epoch table for the per-point normalisation in registers (LogdetPre; the
// scalar-call server) — used when the tile has one point and the preload is valid, ignored otherwise
template <int PREC, int FUSED, bool TRACE = false, int NT = kThreads, bool DYN = false, bool EXTRAS = true, bool PRE = false, int NP = 0>
__device__ __forceinline__ __attribute__((flatten)) void loglike_tile(const LoglikeArgs& __restrict__ a, double* __restrict__ smem, long long p0, int npts,
                                                                      const double* cube_rows = nullptr, const LogdetPre& pre = LogdetPre{}
#ifdef RVLL_WALK_TRACE                 // diagnostic build of the walk only: thread 0 sums the tile's own phases into tph[0..3]
                                                                      , unsigned long long* tph = nullptr
#endif
                                                                      )
{
#ifdef RVLL_WALK_TRACE
    unsigned long long t_last = __builtin_amdgcn_s_memrealtime();
#define RVLL_TILE_STAMP(k) do { if (tph && threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); tph[k] += now_ - t_last; t_last = now_; } } while (0)
#else
#define RVLL_TILE_STAMP(k) do { } while (0)
#endif
    unsigned long long* tr = nullptr;
    if constexpr (TRACE) {
        tr = a.trace + (size_t)blockIdx.x * kTraceWords;
        if (threadIdx.x == 0) {
            tr[0] = __builtin_amdgcn_s_memrealtime();
            // s_getreg_b32 simm16 = (size-1) << 11 | offset << 6 | id; HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            tr[7] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        }
    }
    const TileLds L = tile_views(a, smem);
    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    constexpr int NW = NT / kWave;

    // The prologue is a short serial section (a few lanes, long dependent chains).  A young workgroup's waves get
    // only the issue slots older ones leave (arbitration is by priority, then age), which stretched it 2.5x next
    // to three workgroups in their item loops: run it at raised priority, the item loop at the default.
    if constexpr (!DYN) __builtin_amdgcn_s_setprio(3);
    tile_stage<FUSED, NT>(a, L, p0, npts, TRACE && DYN ? tr + 1 : nullptr, cube_rows);
    __syncthreads();
    RVLL_TILE_STAMP(0);
    if constexpr (FUSED != kFusedNone && NT == kCuThreads) {
        // theta goes out from LDS in whole rows (the stage's lanes run point-fastest: written there, every store would
        // have been a lone 8 bytes of its cache line); the stores drain under the decode step and the item loop
        if (!cube_rows) {
            double* dst = a.theta_out + p0 * a.D;
            for (int i = tid; i < npts * a.D; i += NT) dst[i] = L.theta_s[i];
        }
    }
    if constexpr (TRACE && DYN) { if (tid == 0) tr[2] = __builtin_amdgcn_s_memrealtime(); }
    tile_decode<NT>(a, L, npts);
    __syncthreads();
#ifndef RVLL_AB_NO_LOGDET
    bool done_pre = false;
    if constexpr (PRE) { if (pre.valid && npts == 1) { tile_logdet_pre(a, L, pre); done_pre = true; } }
    if (!done_pre) tile_logdet<NT>(a, L, npts);       // acc[pl] is next touched behind the barrier that ends the items (3c)
    RVLL_TILE_STAMP(1);
#endif
    if constexpr (!DYN) __builtin_amdgcn_s_setprio(0);
    if constexpr (TRACE) { if (tid == 0) tr[DYN ? 3 : 1] = __builtin_amdgcn_s_memrealtime(); }

    // 3. items: flattened (point, epoch) pairs of this block, CH at a time
    const ItemCtx cx{L.pp, L.ins, L.dr, L.lin, L.nfail, L.anyfail, L.jfail, L.wide, L.pflags};
    // CU-wide form: the point with the most eccentric planet goes FIRST.  One wandering solve keeps its wave busy for
    // tens of microseconds; drawn late from the ticket counter it is the launch's tail (profiles/r02_long_solve_tail.txt:
    // 65 -> 73 .. 88 us on the one prior draw in five to ten that holds such a point).  The decode step leaves the wave
    // round that point starts in IN THE TICKET WORD ITSELF (its upper bits: tile_decode, before any ticket is drawn), so
    // every ticket arrives with it and the rounds are walked from there, wrapping — every contribution still lands in
    // its own slot: same sums, same bits.  Nothing new lives in a scalar register across the item loop and nothing more
    // is fetched per round: the kernel sits at its 106 scalar registers, and every earlier form of this — a register
    // held across the loop, a second LDS word read next to the ticket — cost all launches 0.5 - 1 us (round 2's three
    // variants, round 3's first three; profiles/r03_long_solve_tail.txt).
    double* contrib = L.contrib;
    const int nitems = npts * a.Ne;
    // LDS windows are cut at point-local positions — whole points while a point fits the window, otherwise
    // every point by itself at multiples of CH — so each point's sum has one order whatever the tiling
    const int wpts = a.Ne <= a.CH ? a.CH / a.Ne : 0;
    // reduced precision: two items per lane (eval_item_pair); a stop after fewer steps than its loop takes is left to eval_item
    [[maybe_unused]] const bool pairs = PREC != RVLL_PREC_FP64 && a.itmax > kF32Steps;
    for (int base = 0, cend; base < nitems; base = cend) {
        cend = wpts ? min(base + wpts * a.Ne, nitems) : min(base + a.CH, (base / a.Ne + 1) * a.Ne);
        if constexpr (DYN) {
            // one window (host-checked): wave rounds of 64 consecutive items, drawn from the ticket counter
            for (;;) {
                int r = 0;
                // (reduced precision, item pairs: two tickets at once — which two rounds share their lanes is then fixed, and
                // with it every item's wave-wide step count: the same launch gives the same bits)
                [[maybe_unused]] int r2 = 0;
                if constexpr (PREC != RVLL_PREC_FP64) {
                    if (lane == 0) r = atomicAdd(L.ticket, pairs ? 2 : 1);
                } else {
                    if (lane == 0) r = atomicAdd(L.ticket, 1);
                }
                r = __builtin_amdgcn_readfirstlane(r);
#ifndef RVLL_AB_NO_ROT                      // (measurement builds only: scripts/build_variants.sh)
                const int r0 = (r >> kTicketBits) & kTicketRoundMask;      // where the decode step said to start
                r &= (1 << kTicketBits) - 1;
                if (r * kWave >= cend) break;
                r2 = r + 1;
                r += r0;
                if (r * kWave >= cend) r -= (cend + kWave - 1) >> 6;
#else
                if (r * kWave >= cend) break;
                r2 = r + 1;
#endif
                const int i = r * kWave + lane;
                if constexpr (PREC != RVLL_PREC_FP64) {
                    if (pairs) {
                        // the second round of 64 items for the same lanes (none left: the first one twice, its second result dropped)
                        const bool has2 = r2 * kWave < cend;
#ifndef RVLL_AB_NO_ROT
                        r2 += r0;
                        if (r2 * kWave >= cend) r2 -= (cend + kWave - 1) >> 6;
#endif
                        if (!has2) r2 = r;
                        const int i2 = r2 * kWave + lane;
                        int plA, jA, plB, jB;
                        item_of(r * kWave, lane, a.Ne, plA, jA);
                        item_of(r2 * kWave, lane, a.Ne, plB, jB);
                        const bool okA = i < cend, okB = has2 && i2 < cend;
                        if (!okA) { plA = 0; jA = 0; }
                        if (!okB) { plB = plA; jB = jA; }
                        double oA, oB;
                        eval_item_pair<PREC, EXTRAS, NP>(a, cx, plA, jA, plB, jB, oA, oB);
                        if (okA) contrib[i] = oA;
                        if (okB) contrib[i2] = oB;
                        continue;
                    }
                }
                if (i < cend) {
                    int pl, j;
                    item_of(r * kWave, lane, a.Ne, pl, j);
                    contrib[i] = eval_item<PREC, false, EXTRAS, false, NP>(a, cx, pl, j);
                }
            }
            if constexpr (TRACE) { if (tid == 0) tr[4] = __builtin_amdgcn_s_memrealtime(); }
        } else {
            bool done = false;
            if constexpr (PREC != RVLL_PREC_FP64) {
                if (pairs) {
                    for (int i = base + tid; i < cend; i += 2 * NT) {
                        const int i2 = i + NT;
                        const bool wave2 = i2 - lane < cend, okB = i2 < cend;       // (the wave's second round exists; this lane's item in it)
                        int plA, jA, plB, jB;
                        item_of(i - lane, lane, a.Ne, plA, jA);
                        item_of(wave2 ? i2 - lane : i - lane, lane, a.Ne, plB, jB);
                        if (!okB) { plB = plA; jB = jA; }
                        double oA, oB;
                        eval_item_pair<PREC, EXTRAS, NP>(a, cx, plA, jA, plB, jB, oA, oB);
                        contrib[i - base] = oA;
                        if (okB) contrib[i2 - base] = oB;
                    }
                    done = true;
                }
            }
            if (!done)
            for (int i = base + tid; i < cend; i += NT) {
                int pl, j;
                item_of(i - lane, lane, a.Ne, pl, j);            // i - lane = base + 64 wave + k NT: the same for the whole wave
                contrib[i - base] = eval_item<PREC, false, EXTRAS, false, NP>(a, cx, pl, j);
            }
            if constexpr (TRACE) { if (lane == 0) tr[2 + wave] = __builtin_amdgcn_s_memrealtime(); }
        }
        __syncthreads();
        RVLL_TILE_STAMP(2);
        if constexpr (TRACE && DYN) { if (tid == 0) tr[5] = __builtin_amdgcn_s_memrealtime(); }
        // 3b. rare: a solve hit itmax (bit 0 of nfail) — the reference aborts that planet's array there and leaves nu = 0 from
        // that epoch on: the affected points' items are redone now that the first failing epoch per (point, planet) is
        // known — or a solve wandered (bit 1): that point's items are redone with the wandering solves on correctly rounded
        // sin / cos (eval_item, CR).  Every other solve of those points takes the path it took before: the same bits.
        if (__builtin_expect(L.nfail[0] != 0, 0)) {                // (cold: the register allocator is to favour the item loop)
            for (int i = base + tid; i < cend; i += NT) {
                const int pl = i / a.Ne;
                if (L.anyfail[pl] || contrib[i - base] != contrib[i - base])       // a point with an itmax failure; an item left as NaN
                    contrib[i - base] = eval_item<PREC, true, EXTRAS, PREC == RVLL_PREC_FP64, NP>(a, cx, pl, i - pl * a.Ne);
            }
            __syncthreads();
        }
        // 3c. per-point partial sums of this window, fixed order (deterministic): the part of each point that
        // lies in the window, slice by slice
        const int pl_lo = base / a.Ne;
        const int pl_hi = (cend - 1) / a.Ne;
        const int nslices = (a.Ne + kTileWindow - 1) / kTileWindow;
        for (int pl0 = pl_lo + wave; pl0 <= pl_hi; pl0 += 4 * NW) {
            for (int sl = 0; sl < nslices; ++sl) {
                int lo[4], hi[4];
                double v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pl = pl0 + k * NW;
                    const int first = pl * a.Ne + sl * kTileWindow;                  // this slice of this point
                    const int l = max(base, first), h = min(min(cend, (pl + 1) * a.Ne), first + kTileWindow);
                    const bool any = pl <= pl_hi && l < h;
                    lo[k] = any ? l : 0;
                    hi[k] = any ? h : 0;
                }
                point_partials4(contrib, base, lo, hi, lane, v);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (lo[k] < hi[k]) L.acc[pl0 + k * NW] += v[k];
                }
            }
        }
        __syncthreads();
    }

    for (int pl = tid; pl < npts; pl += NT) tile_write_point(a, L, p0, pl);
    RVLL_TILE_STAMP(3);
#undef RVLL_TILE_STAMP
    if constexpr (TRACE) {
        __syncthreads();
        if (threadIdx.x == 0) tr[6] = __builtin_amdgcn_s_memrealtime();
    }
}


}  // namespace

}  // namespace rvll
