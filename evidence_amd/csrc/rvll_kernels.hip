// rvll_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the RV
// log-likelihood hot path.  Compiled with -ffp-contract=off (see rvll_math.h).
//
//   loglike_kernel  fuses, for a whole batch of live points,
//       evidence/rvmodel/__init__.py:173-217  log_likelihood
//       evidence/rvmodel/__init__.py:343-385  kep_rv
//       evidence/rvmodel/__init__.py:388-463  modelk
//       evidence/rvmodel/trueanomaly.c:8-41   trueanomaly (Newton, tol 1e-4)
//       evidence/rvmodel/__init__.py:222-273  drift
//       evidence/rvmodel/__init__.py:59-80    logL
//   into one launch: one thread per (live point, epoch) pair, Kepler iteration and
//   sin/cos in registers, per-point reduction through LDS + wave shuffles.
//
//   prior_kernel    evidence/priors.py .ppf of each distribution, one thread per
//       (live point, parameter).
//
// Roofline: this path is fp64-VALU bound (software sin/cos, IEEE division); HBM
// traffic is theta in + log-L out (+ the epoch table, L2-resident).  No MFMA: there
// is no contraction here.
#include "rvll_tile.h"
#include "rvll_rounds.h"

namespace rvll {

namespace {

#ifdef RVLL_AB_NO_NP                   // (measurement builds only: without the instantiations for a compile-time planet count)
constexpr bool kPlanetCountKernels = false;
#else
constexpr bool kPlanetCountKernels = true;
#endif
#ifdef RVLL_AB_NO_LEAN                 // (measurement builds only: scripts/build_variants.sh)
constexpr bool kLeanKernels = false;
#else
constexpr bool kLeanKernels = true;
#endif

// The CU-wide form: one 1024-thread workgroup per tile of a.PB points (loglike_tile, NT = 1024, DYN); FUSED as in
// loglike_kernel
// EXTRAS = false: the instantiation for models without drift and without linear activity terms — most of them, the
// headline configuration among them: the item loop then carries neither those branches nor the scalar registers they hold
// across it (the kernel sits at its 106; profiles/r03_isa_budget_loglike_cu.txt)
// NP > 0: the planet count at compile time (rvll_tile.h, eval_item): the lean fp64 kernels at one and three planets
template <int PREC, bool TRACE, int FUSED = kFusedNone, bool EXTRAS = true, int NP = 0>
__global__ __launch_bounds__(kCuThreads) __attribute__((flatten))
void loglike_cu_kernel(const LoglikeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long long p0 = (long long)blockIdx.x * a.PB;
    const int npts = (int)min((long long)a.PB, a.B - p0);
    if (npts <= 0) return;
    loglike_tile<PREC, FUSED, TRACE, kCuThreads, true, EXTRAS, false, NP>(a, smem, p0, npts);
}

// 4 workgroups of 256 per CU (4 waves/SIMD): caps the kernel at 128 VGPRs
// FUSED: kFusedNone (theta rows in) or kFusedSlim (unit-cube rows in, verified-table quantiles; rvll_tile.h)
template <int PREC, int FUSED, bool EXTRAS = true, int NP = 0>
__global__ __launch_bounds__(kThreads, 4) __attribute__((flatten))      // flatten: the prior routines of the fused
void loglike_kernel(const LoglikeArgs a)                                // form must live under the same VGPR cap
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long long p0 = (long long)blockIdx.x * a.PB;
    const int npts = (int)min((long long)a.PB, a.B - p0);
    if (npts <= 0) return;
    loglike_tile<PREC, FUSED, false, kThreads, false, EXTRAS, false, NP>(a, smem, p0, npts);
}


__global__ __launch_bounds__(kThreads)
void rounds_dirs_kernel(const RoundsDirs g)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    rounds_dirs(g, smem);
}

// The two parts as kernels of their own (a group's round on the group's stream: step, then tiles)
// (128 VGPRs at most: a step's wave is to fit a SIMD next to three waves of tiles)
__global__ __launch_bounds__(kThreads, 4) __attribute__((flatten))
void rounds_step_kernel(const RoundsArgs g, const int r)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __builtin_amdgcn_s_setprio(3);
    rounds_step(g, r, (int)blockIdx.x, smem);
}

template <int PREC, bool EXTRAS, int NP = 0>
__global__ __launch_bounds__(kThreads, 4) __attribute__((flatten))
void rounds_tiles_kernel(const LoglikeArgs a, const RoundsTiles o, const int r)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (o.progress && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(o.progress, ((unsigned long long)(unsigned)(r + 1) << 32) | (unsigned)o.ring_entry[1], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    const int B = __builtin_amdgcn_readfirstlane(o.ring_entry[0]);
    const int per = (B + (int)gridDim.x - 1) / (int)gridDim.x;
    const long long p0 = (long long)blockIdx.x * per;
    const int npts = (int)min((long long)per, (long long)B - p0);
    if (npts <= 0) return;
    loglike_tile<PREC, kFusedNone, false, kThreads, false, EXTRAS, false, NP>(a, smem, p0, npts);
    const TileLds L = tile_views(a, smem);
    for (int pl = threadIdx.x; pl < npts; pl += kThreads) {
        int f;
        const double v = tile_point_result(a, L, pl, f);
        const int ow = o.owner[p0 + pl];
        o.wres_logl[ow] = v;
        o.wres_flags[ow] = f;
    }
}

// The tiles of a round in the CU-wide form (loglike_cu_kernel): no step beside them in the launch — a 1024-thread workgroup
// leaves a compute unit no room for one.
template <int PREC, bool EXTRAS>
__global__ __launch_bounds__(kCuThreads) __attribute__((flatten))
void rounds_cu_kernel(const LoglikeArgs a, const RoundsTiles o, const int r)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (o.progress && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(o.progress, ((unsigned long long)(unsigned)(r + 1) << 32) | (unsigned)o.ring_entry[1], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    const int B = __builtin_amdgcn_readfirstlane(o.ring_entry[0]);
    const int per = (B + (int)gridDim.x - 1) / (int)gridDim.x;
    const long long p0 = (long long)blockIdx.x * per;
    const int npts = (int)min((long long)per, (long long)B - p0);
    if (npts <= 0) return;
    loglike_tile<PREC, kFusedNone, false, kCuThreads, true, EXTRAS>(a, smem, p0, npts);
    const TileLds L = tile_views(a, smem);
    for (int pl = threadIdx.x; pl < npts; pl += kCuThreads) {
        int f;
        const double v = tile_point_result(a, L, pl, f);
        const int ow = o.owner[p0 + pl];
        o.wres_logl[ow] = v;
        o.wres_flags[ow] = f;
    }
}

// Diagnostic twin of loglike_kernel<RVLL_PREC_FP64, kFusedNone>: same tile, same launch bounds, plus the stamps.
__global__ __launch_bounds__(kThreads, 4) __attribute__((flatten))
void loglike_trace_kernel(const LoglikeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long long p0 = (long long)blockIdx.x * a.PB;
    const int npts = (int)min((long long)a.PB, a.B - p0);
    if (npts <= 0) return;
    loglike_tile<RVLL_PREC_FP64, kFusedNone, true>(a, smem, p0, npts);
}

// Scalar-call server (rvll_kernels.h, ServerCtl).  Thread 0 polls the request word in host memory (system-scope
// acquire; one PCIe read per poll — op and number travel in that one word), the workgroup evaluates the one
// point exactly as a one-point launch would (same loglike_tile, same bits) into device-local scratch, and
// thread 0 sends log-L, flags and the request number back as ONE 16-byte store, so the host needs no second
// flag and the GPU no fence between result and flag.  Every wave leaves through the same uniform test: a quit
// request, or idle_ticks of the constant 100 MHz clock without a request — so the kernel ends by itself if the
// host stops asking, whatever the reason (the quit request is just another op in the same word).
template <int PREC>
__global__ __launch_bounds__(kThreads, 1) __attribute__((flatten))     // one workgroup on the chip: no VGPR cap
void scalar_server_kernel(const LoglikeArgs a, ServerCtl* ctl, unsigned long long last, unsigned long long idle_ticks)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const Carve cv = carve(a.PB, a.D, a.Np, a.Ni, a.nlin, a.CH);
    unsigned long long* word = reinterpret_cast<unsigned long long*>(smem + cv.total_doubles);   // [0] request, [1] stop
    LogdetPre pre;                                    // this thread's share of the epoch table for the per-point normalisation:
    logdet_preload(a, pre);                           // fetched once, it never changes between requests (rvll_tile.h)
    for (;;) {
        double* trow = reinterpret_cast<double*>(word + 2);
        const bool slots = a.D <= kServerSlotDims;                       // (uniform) request and row arrive together: ServerCtl::in
        if (threadIdx.x < kWave) {
            const int lane = threadIdx.x;
            const unsigned long long t0 = wall_clock64();
            unsigned long long r = last, stop = 0;
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            for (unsigned polls = 1;; ++polls) {      // one PCIe read per poll; the clock only every 64 polls
                if (slots) {
                    u64x2 sl = {0ull, 0ull};
                    if (lane < a.D) sl = *reinterpret_cast<const volatile u64x2*>(&ctl->in[lane]);     // volatile: system-scope cache bits
                    const unsigned long long mine = sl.y ^ ServerCtl::slot_key(sl.x);      // the request this slot's (value, word) decode to
                    r = __builtin_amdgcn_readfirstlane((unsigned)mine) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mine >> 32)) << 32);
                    const bool all_same = __builtin_amdgcn_ballot_w64(lane < a.D && mine != r) == 0;
                    if (all_same && r != last) {
                        if (lane < a.D) trow[lane] = __builtin_bit_cast(double, sl.x);
                        stop = (unsigned)(r >> 32) == kServerQuit;
                        break;
                    }
                    // (three such reads in flight a third of a round trip apart, to see a request sooner, were measured: 12.9 us a call
                    // against 9.9 — the waits on the oldest read wait for the youngest too; profiles/r04_call_latency.txt)
                } else {
                    r = __hip_atomic_load(&ctl->request, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (r != last) { stop = (unsigned)(r >> 32) == kServerQuit; break; }
                }
                if ((polls & 63u) == 0u && __builtin_amdgcn_readfirstlane((int)(wall_clock64() - t0 > idle_ticks))) { stop = 1; break; }   // (one lane's clock decides for the wave)
            }
            if (lane == 0) {
                word[0] = r;
                word[1] = stop;
            }
        }
        __syncthreads();
        const unsigned long long r = word[0];
        if (word[1] != 0) break;
        const unsigned op = (unsigned)(r >> 32);
        ServerAnswer ans{0., (unsigned)r, 0};
        if (op == kServerPrior || op == kServerPriorLogLike) {
            // prior(cube): the row is transformed in place (staged through LDS first: the sorted kinds read
            // the whole row).  a.priors is null until rvll_set_priors; the host does not send the op before.
            // kServerPriorLogLike: the sampler's next call is loglike(of exactly this theta) — evaluate it now, from
            // the LDS copy of theta rather than back over PCIe, and send both with one answer.
            double* row = smem;
            for (int d = threadIdx.x; d < a.D; d += kThreads) row[d] = slots ? trow[d] : ctl->theta[d];
            __syncthreads();
            for (int d = threadIdx.x; d < a.D; d += kThreads) {
                const double v = prior_is_heavy(a.priors[d].kind) ? prior_heavy(a.priors[d], row[d])
                                                                  : prior_light(a.priors, a.D, row, d);
                ctl->theta[d] = v;
                trow[d] = v;
            }
            __threadfence_system();                   // every thread's theta stores are out before the answer
            __syncthreads();
            if (op == kServerPriorLogLike) {
                loglike_tile<PREC, kFusedNone, false, kThreads, false, true, true>(a, smem, 0, 1, trow, pre);
                if (threadIdx.x == 0) {               // the point's sums are in LDS (this thread wrote them out itself: tile_write_point)
                    int f;
                    ans.logL = tile_point_result(a, tile_views(a, smem), 0, f);
                    ans.flags = f;
                }
            }
        } else if (op == kServerLogLike) {
            loglike_tile<PREC, kFusedNone, false, kThreads, false, true, true>(a, smem, 0, 1, slots ? trow : nullptr, pre);
            if (threadIdx.x == 0) {                   // straight from the tile's LDS accumulators (no trip through device memory)
                int f;
                ans.logL = tile_point_result(a, tile_views(a, smem), 0, f);
                ans.flags = f;
            }
        }                                             // any other op: acknowledged only (round-trip probe)
        if (threadIdx.x == 0) {
            static_assert(sizeof(ServerAnswer) == 16, "the answer must leave as one 16-byte store");
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            // volatile: emitted with system-scope cache bits (sc0 sc1), i.e. written through to host memory
            *reinterpret_cast<volatile u64x2*>(&ctl->answer) = __builtin_bit_cast(u64x2, ans);
        }
        last = r;
        __syncthreads();                              // the LDS words and stage are reused by the next request
    }
    if (threadIdx.x == 0) __hip_atomic_store(&ctl->state, kServerExited, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------
// prior transform kernels
// ---------------------------------------------------------------------------

// Light kinds: one thread per (live point, parameter), grid-stride.  Iterative kinds are left to
// prior_heavy_kernel.
__global__ __launch_bounds__(kThreads)
void prior_kernel(const PriorArgs a)
{
    const long long n = a.B * a.D;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (long long)gridDim.x * kThreads) {
        const int d = (int)(i % a.D);
        if (prior_is_heavy(a.priors[d].kind)) continue;
        a.theta[i] = prior_light(a.priors, a.D, a.cube + (i - d), d);
    }
}

// Only the parameters that need an iterative quantile, compacted so that every lane of a wave is doing a solve.
__global__ __launch_bounds__(kThreads)
void prior_heavy_kernel(const PriorArgs a)
{
    const long long n = a.B * a.n_heavy;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (long long)gridDim.x * kThreads) {
        const long long b = i / a.n_heavy;
        const int d = a.heavy_dims[(int)(i - b * a.n_heavy)];
        a.theta[b * a.D + d] = prior_heavy(a.priors[d], a.cube[b * a.D + d]);
    }
}

// Tabulate one heavy prior's quantile function in smooth coordinates (rvll_special.h, "tabulated
// starts"); runs once per rvll_set_priors, one node per thread.  dz holds 2 * kTableN values: slopes, then
// the closed-form second derivatives.
__global__ __launch_bounds__(kThreads)
void prior_table_kernel(int kind, double a0, double a1, double a2, double* z, double* dz)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= kTableN) return;
    const double u = -kTableU + i * (2. * kTableU / (kTableN - 1));
    double zi, dzi;
    if (kind == RVLL_PRIOR_BETA) beta_table_node(a0, a1, a2, u, zi, dzi);
    else                         gamma_table_node(a0, a2, u, zi, dzi);
    z[i] = zi;
    dz[i] = dzi;
    dz[kTableN + i] = kind == RVLL_PRIOR_BETA ? beta_table_d2(a0, a1, u, zi, dzi) : gamma_table_d2(a0, u, zi, dzi);
}

// Measure the quintic interpolant against the full solver at every interval midpoint (where its error term
// peaks): max |dz| over the table, as the bits of a non-negative double (integer max == float max).
__global__ __launch_bounds__(kThreads)
void prior_table_check_kernel(int kind, double a0, double a1, double a2, const double* z, const double* dz,
                              unsigned long long* max_err_bits)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= kTableN - 1) return;
    const double u = -kTableU + (i + 0.5) * (2. * kTableU / (kTableN - 1));
    double zt, dzt;
    if (kind == RVLL_PRIOR_BETA) beta_table_node(a0, a1, a2, u, zt, dzt);
    else                         gamma_table_node(a0, a2, u, zt, dzt);
    double err = fabs(quintic_table(z, dz, u) - zt);
    if (!(err >= 0.)) err = INFINITY;                     // NaN anywhere disqualifies the table
    atomicMax(max_err_bits, (unsigned long long)__double_as_longlong(err));
}

__global__ __launch_bounds__(kThreads)
void fill_cube_kernel(double* cube, long long n, uint64_t seed)
{
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (long long)gridDim.x * kThreads)
        cube[i] = uniform01(seed, (uint64_t)i);
}

// Keplerian curves at arbitrary times: kep_rv(pardict, time, exclude_planet) and modelk(pardict, time, planet)
// of evidence/rvmodel/__init__.py:343-463 for a batch of parameter vectors, as post_processing.py:413-428 uses
// them for phase folds.  One thread per (live point, time); planets selected by a bit mask.  Not a hot path:
// every thread decodes its planets itself.  An invalid orbit (the reference returns None) gives NaN.
__global__ __launch_bounds__(kThreads)
void keprv_kernel(const LoglikeArgs a, const double* times, int Nt, unsigned include_mask, double* out)
{
    const long long n = a.B * (long long)Nt;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const long long b = i / Nt;
        const double t = times[(int)(i - b * Nt)];
        const double* th = a.theta + b * a.D;
        double sum = 0.;
        bool valid = true;
        for (int ip = 0; ip < a.Np; ++ip) {
            if (!((include_mask >> ip) & 1u)) continue;
            const rvll_planet& d = a.planets[ip];
            const double kraw = slot_get(d.k, th), praw = slot_get(d.p, th);
            const double K = d.k_kind == RVLL_K_LOGK1 ? exp(kraw) : kraw;
            const double Pd = d.p_kind == RVLL_P_LOGPERIOD ? exp(praw) : praw;
            const double e1 = slot_get(d.e1, th), e2 = slot_get(d.e2, th);
            double ecc, omega;
            if (d.ecc_kind == RVLL_ECC_SECOS_SESIN) { ecc = e1 * e1 + e2 * e2; omega = atan2(e2, e1); if (ecc > 1) valid = false; }
            else if (d.ecc_kind == RVLL_ECC_ECOS_ESIN) { ecc = sqrt(e1 * e1 + e2 * e2); omega = atan2(e2, e1); if (ecc > 1) valid = false; }
            else { ecc = e1; omega = e2; }
            const double anom = slot_get(d.anom, th);
            const double ma0 = d.anom_kind == RVLL_ANOM_ML0 ? anom - omega : anom;
            const double ec = ecc > 0.99 ? 0.99 : ecc;
            const double M = (kTwoPi / Pd) * (t - slot_get(d.epoch, th)) + ma0;
            double E = M, s, c, dE;
            int steps = 0;
            do {
                // correctly rounded, as glibc's nearly always are (round 4): this is not a hot path, and where the iteration
                // wanders (e >= 0.97) where it stops hangs on the last bit; any finite argument is reduced exactly
                sincos_cr(E, s, c);
                const double f = E - ec * s - M;
                const double fp = 1 - ec * c;
                const double En = E - div_exact(f, fp);
                dE = En - E;
                E = En;
                ++steps;
            } while (fabs(dE) > a.tol && steps < a.itmax);
            double so, co;
            sincos_any(omega, so, co);
            double rv;
            if (steps >= a.itmax) {
                rv = K * (co + ecc * co);     // a single time has no "rest of the array": nu stays 0 for this element
            } else {
                // E is wherever M is: an absurd period puts both beyond the short reduction's 2^50 (the tile guards its
                // solves with the |M| < 2^48 bound of the decode step; a curve at arbitrary times has no such bound)
                sincos_any(E, s, c);
                const double q = sqrt((1. - ec) * (1. + ec));
                const double den = __builtin_fma(-ec, c, 1.0);
                rv = K * (div_exact((c - ec) * co - q * s * so, den) + ecc * co);
            }
            sum += rv;
        }
        out[i] = valid ? sum : NAN;
    }
}

// device-math self test (rvll_debug_eval): out[i] = op(x[i], y[i])
__global__ __launch_bounds__(kThreads)
void debug_eval_kernel(int op, const double* x, const double* y, long long n, double* out)
{
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const double a = x[i];
        const double b = y ? y[i] : 0.;
        double s, c, r;
        switch (op) {
        case 0: sincos_f64(a, s, c); r = s; break;
        case 1: sincos_f64(a, s, c); r = c; break;
        case 2: r = div_exact(a, b); break;
        case 3: r = a / b; break;
        case 4: r = div_fast(a, b); break;
        case 5: r = log_pos(a); break;
        case 6: r = log(a); break;
        case 7: r = ndtri_f64(a); break;
        case 8: sincos_f64(a, s, c); rotate_small(b, s, c); r = s; break;
        case 9: sincos_f64(a, s, c); rotate_small(b, s, c); r = c; break;
        case 10: r = div_1nr(a, b); break;
        case 11: r = __builtin_amdgcn_rcp(a); break;
        case 12: r = wave_sum(a); break;                 // lane 0 of every wave: the shuffle tree
        case 13: r = wave_sum_lane0(a); break;           //                        the same tree without the LDS crossbar
        case 14: r = ndtri_cephes(a); break;
        case 20: sincos_cr(a, s, c); r = s; break;       // the redo pass's correctly rounded pair (tests/test_gpu_math.py)
        case 21: sincos_cr(a, s, c); r = c; break;
        // latency chains, 1000 dependent evaluations a lane (scripts/cr_latency_probe.py): what one Newton step of a wandering
        // solve waits for — the whole routine, its reduction, its two kernels, the ordinary pair
        case 15: { double v = a; for (int k = 0; k < 1000; ++k) { sincos_cr(v, s, c); v = v * 1.0000001 + s * 1e-3; } r = v; break; }
        case 16: { double v = a; for (int k = 0; k < 1000; ++k) { DD rr; uint32_t q; reduce_dd(v, rr, q); v = v * 1.0000001 + rr.hi * 1e-3 + (double)q; } r = v; break; }
        case 17: { double v = a; for (int k = 0; k < 1000; ++k) { sincos_dd_table(DD{v, 1e-20}, s, c); v = 0.7 * s + 1e-3 * c; } r = v; break; }
        case 18: { double v = a; for (int k = 0; k < 1000; ++k) { sincos_dd_kernel(DD{v, 1e-20}, s, c); v = 0.7 * s + 1e-3 * c; } r = v; break; }
        case 19: { double v = a; for (int k = 0; k < 1000; ++k) { sincos_any(v, s, c); v = v * 1.0000001 + s * 1e-3; } r = v; break; }
        default: r = NAN; break;
        }
        out[i] = r;
    }
}

}  // namespace

hipError_t launch_debug_eval(int op, const double* x, const double* y, long long n, double* out, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    long long blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(debug_eval_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, op, x, y, n, out);
    return hipGetLastError();
}

hipError_t launch_keprv(const LoglikeArgs& a, const double* times, int Nt, unsigned include_mask, double* out,
                        hipStream_t stream)
{
    const long long n = a.B * (long long)Nt;
    if (n <= 0) return hipSuccess;
    long long blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(keprv_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, a, times, Nt, include_mask, out);
    return hipGetLastError();
}

size_t loglike_lds_bytes(const LoglikeArgs& a)
{
    return (size_t)carve(a.PB, a.D, a.Np, a.Ni, a.nlin, a.CH).total_doubles * sizeof(double);
}

int loglike_blocks_per_cu(size_t lds_bytes)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, loglike_kernel<RVLL_PREC_FP64, kFusedNone>, kThreads, lds_bytes) != hipSuccess || n < 1)
        n = 1;
    return n > 8 ? 8 : n;
}

hipError_t launch_loglike(const LoglikeArgs& a, hipStream_t stream)
{
    if (a.B <= 0) return hipSuccess;
    const long long blocks = (a.B + a.PB - 1) / a.PB;
    const size_t lds = loglike_lds_bytes(a);
    const dim3 grid((unsigned)blocks), block(kThreads);
    switch (a.precision) {
    case RVLL_PREC_MIXED:
        if (kPlanetCountKernels && a.Np == 5) hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_MIXED, kFusedNone, true, 5>), grid, block, lds, stream, a);
        else                                  hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_MIXED, kFusedNone>), grid, block, lds, stream, a);
        break;
    case RVLL_PREC_FP32:
        if (kPlanetCountKernels && a.Np == 5) hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_FP32, kFusedNone, true, 5>), grid, block, lds, stream, a);
        else                                  hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_FP32, kFusedNone>), grid, block, lds, stream, a);
        break;
    default:
        if (kLeanKernels && !a.has_drift && a.nlin == 0) {
            if (kPlanetCountKernels && a.Np == 3)      hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_FP64, kFusedNone, false, 3>), grid, block, lds, stream, a);
            else if (kPlanetCountKernels && a.Np == 1) hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_FP64, kFusedNone, false, 1>), grid, block, lds, stream, a);
            else                                       hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_FP64, kFusedNone, false>), grid, block, lds, stream, a);
        } else if (kPlanetCountKernels && a.Np == 5) {                   // (BASELINE configs[4]: five planets + drift)
            hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_FP64, kFusedNone, true, 5>), grid, block, lds, stream, a);
        } else {
            hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_FP64, kFusedNone>), grid, block, lds, stream, a);
        }
        break;
    }
    return hipGetLastError();
}

hipError_t launch_loglike_cu(const LoglikeArgs& a, int grid, hipStream_t stream)
{
    if (a.B <= 0) return hipSuccess;
    const size_t lds = loglike_lds_bytes(a);
    if (grid < 1 || (long long)grid * a.PB < a.B || a.PB < 1 || a.PB > kCuMaxPoints || (long long)a.CH < (long long)a.PB * a.Ne || lds > kCuLdsBudget ||
        (a.trace && a.precision != RVLL_PREC_FP64))
        return hipErrorInvalidValue;
    const bool fused = a.cube != nullptr;                   // the slim prior stage in front (launch_prior_loglike's arguments)
    if (fused && (!a.theta_out || !a.priors || (a.n_heavy > 0 && !a.heavy_dims) || a.trace)) return hipErrorInvalidValue;
    static bool attr_set_dev[64] = {};                      // raise the dynamic-LDS limit of every instance once per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& attr_set = attr_set_dev[dev & 63];
    if (!attr_set) {
        const int lim = (int)kCuLdsBudget;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(loglike_cu_kernel<RVLL_PREC_FP64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(loglike_cu_kernel<RVLL_PREC_MIXED, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(loglike_cu_kernel<RVLL_PREC_FP32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(loglike_cu_kernel<RVLL_PREC_FP64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(loglike_cu_kernel<RVLL_PREC_FP64, false, kFusedNone, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(loglike_cu_kernel<RVLL_PREC_FP64, false, kFusedNone, false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(loglike_cu_kernel<RVLL_PREC_FP64, false, kFusedNone, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(loglike_cu_kernel<RVLL_PREC_FP64, false, kFusedSlim>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(loglike_cu_kernel<RVLL_PREC_FP64, false, kFusedSlim, true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(loglike_cu_kernel<RVLL_PREC_MIXED, false, kFusedSlim>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(loglike_cu_kernel<RVLL_PREC_FP32, false, kFusedSlim>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const dim3 g((unsigned)grid), block(kCuThreads);
    if (a.trace) { hipLaunchKernelGGL((loglike_cu_kernel<RVLL_PREC_FP64, true>), g, block, lds, stream, a); return hipGetLastError(); }
    if (fused) {
        switch (a.precision) {
        case RVLL_PREC_MIXED: hipLaunchKernelGGL((loglike_cu_kernel<RVLL_PREC_MIXED, false, kFusedSlim>), g, block, lds, stream, a); break;
        case RVLL_PREC_FP32:  hipLaunchKernelGGL((loglike_cu_kernel<RVLL_PREC_FP32, false, kFusedSlim>), g, block, lds, stream, a); break;
        default:
            if (kPlanetCountKernels && a.Np == 3) hipLaunchKernelGGL((loglike_cu_kernel<RVLL_PREC_FP64, false, kFusedSlim, true, 3>), g, block, lds, stream, a);
            else                                  hipLaunchKernelGGL((loglike_cu_kernel<RVLL_PREC_FP64, false, kFusedSlim>), g, block, lds, stream, a);
            break;
        }
        return hipGetLastError();
    }
    switch (a.precision) {
    case RVLL_PREC_MIXED: hipLaunchKernelGGL((loglike_cu_kernel<RVLL_PREC_MIXED, false>), g, block, lds, stream, a); break;
    case RVLL_PREC_FP32:  hipLaunchKernelGGL((loglike_cu_kernel<RVLL_PREC_FP32, false>), g, block, lds, stream, a); break;
    default:
        if (kLeanKernels && !a.has_drift && a.nlin == 0) {
            if (kPlanetCountKernels && a.Np == 3)      hipLaunchKernelGGL((loglike_cu_kernel<RVLL_PREC_FP64, false, kFusedNone, false, 3>), g, block, lds, stream, a);
            else if (kPlanetCountKernels && a.Np == 1) hipLaunchKernelGGL((loglike_cu_kernel<RVLL_PREC_FP64, false, kFusedNone, false, 1>), g, block, lds, stream, a);
            else                                       hipLaunchKernelGGL((loglike_cu_kernel<RVLL_PREC_FP64, false, kFusedNone, false>), g, block, lds, stream, a);
        } else {
            hipLaunchKernelGGL((loglike_cu_kernel<RVLL_PREC_FP64, false>), g, block, lds, stream, a);
        }
        break;
    }
    return hipGetLastError();
}

// ---- the walk in its rounds form: host side
// walkers per workgroup of the step: as many as `lds_budget` bytes hold (the step shares its launch, and with it the size of
// the dynamic LDS, with the log-L tiles: four workgroups are to fit a compute unit), one lane of a wave each at most
int rounds_walkers_per_block(int D, int spec_max, size_t lds_budget)
{
    int W = kWave;
    while (W > 1 && step_lds_bytes(W, D, spec_max) > lds_budget) --W;
    return step_lds_bytes(W, D, spec_max) <= 64 * 1024 ? W : 0;
}

size_t rounds_step_lds_bytes(int W, int D, int spec_max) { return step_lds_bytes(W, D, spec_max); }

// workgroups of the tiles kernel a compute unit holds with `lds` bytes of dynamic LDS each (0: the query failed)
int rounds_blocks_per_cu(size_t lds)
{
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (rounds_tiles_kernel<RVLL_PREC_FP64, false>), kThreads, lds) != hipSuccess) return 0;
    return occ;
}


hipError_t launch_rounds_dirs(const RoundsDirs& g, int max_blocks, hipStream_t stream)
{
    if (g.K <= 0 || g.nsteps <= 0) return hipSuccess;
    const size_t lds = sizeof(double) * dirs_lds_doubles(g.D);
    if (g.D < 1 || !g.dirs || !g.chol || g.nsteps >= (1 << 18) || lds > 64 * 1024 || max_blocks < 1) return hipErrorInvalidValue;
    const long long want = (g.K * g.nsteps + kDirPairs - 1) / kDirPairs;
    hipLaunchKernelGGL(rounds_dirs_kernel, dim3((unsigned)(want < max_blocks ? want : max_blocks)), dim3(kThreads), lds, stream, g);
    return hipGetLastError();
}

static bool rounds_step_ok(const RoundsArgs& g, int r)
{
    return !(g.W < 1 || g.W > kWave || g.D < 1 || g.spec_max < 1 || g.C < g.K || g.c_free > g.C || g.c_free < 1 || r < 0 ||
             g.nsteps >= (1 << 18) || g.max_rounds < 1 || g.max_rounds > 4096 || (long long)g.K * g.spec_max >= (1LL << 31) ||
             !g.priors || !g.light_dims || (g.n_heavy > 0 && !g.heavy_dims) || g.n_heavy > g.D || !g.theta_c[0] || !g.theta_c[1] || !g.wdef || !g.wflag || !g.dirs || !g.dirnext);
}

hipError_t launch_rounds_step(const RoundsArgs& g, int round, hipStream_t stream)
{
    if (g.K <= 0) return hipSuccess;
    const size_t lds = step_lds_bytes(g.W, g.D, g.spec_max);
    if (!rounds_step_ok(g, round) || lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rounds_step_kernel, dim3((unsigned)((g.K + g.W - 1) / g.W)), dim3(kThreads), lds, stream, g, round);
    return hipGetLastError();
}

hipError_t launch_rounds_tiles(const LoglikeArgs& a, const RoundsTiles& o, int tiles, int round, hipStream_t stream)
{
    const size_t lds = loglike_lds_bytes(a);
    if (tiles < 1 || !a.theta || a.cube || !a.flags || !a.logL || a.PB < 1 || !o.ring_entry || !o.owner || !o.wres_logl || !o.wres_flags ||
        lds > 64 * 1024)
        return hipErrorInvalidValue;
    const bool lean = !a.has_drift && a.nlin == 0;
    const dim3 grid((unsigned)tiles), block(kThreads);
    switch (a.precision) {
    case RVLL_PREC_MIXED: hipLaunchKernelGGL((rounds_tiles_kernel<RVLL_PREC_MIXED, true>), grid, block, lds, stream, a, o, round); break;
    case RVLL_PREC_FP32:  hipLaunchKernelGGL((rounds_tiles_kernel<RVLL_PREC_FP32, true>), grid, block, lds, stream, a, o, round); break;
    default:
        if (lean && kPlanetCountKernels && a.Np == 3) hipLaunchKernelGGL((rounds_tiles_kernel<RVLL_PREC_FP64, false, 3>), grid, block, lds, stream, a, o, round);
        else if (lean) hipLaunchKernelGGL((rounds_tiles_kernel<RVLL_PREC_FP64, false>), grid, block, lds, stream, a, o, round);
        else           hipLaunchKernelGGL((rounds_tiles_kernel<RVLL_PREC_FP64, true>), grid, block, lds, stream, a, o, round);
        break;
    }
    return hipGetLastError();
}

hipError_t launch_rounds_cu(const LoglikeArgs& a, const RoundsTiles& o, int tiles, int round, hipStream_t stream)
{
    const size_t lds = loglike_lds_bytes(a);
    if (tiles < 1 || !a.theta || a.cube || !a.flags || !a.logL || a.PB < 1 ||
        !o.ring_entry || !o.owner || !o.wres_logl || !o.wres_flags || a.PB > kCuMaxPoints || (long long)a.CH < (long long)a.PB * a.Ne ||
        lds > kCuLdsBudget)
        return hipErrorInvalidValue;
    static bool attr_set_dev[64] = {};                      // raise the dynamic-LDS limit of every instance once per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set_dev[dev & 63]) {
        hipError_t e = hipSuccess;
#define RVLL_ROUNDS_ATTR(PREC, EX) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(rounds_cu_kernel<PREC, EX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCuLdsBudget)
        RVLL_ROUNDS_ATTR(RVLL_PREC_FP64, true); RVLL_ROUNDS_ATTR(RVLL_PREC_FP64, false);
        RVLL_ROUNDS_ATTR(RVLL_PREC_MIXED, true); RVLL_ROUNDS_ATTR(RVLL_PREC_FP32, true);
#undef RVLL_ROUNDS_ATTR
        if (e != hipSuccess) return e;
        attr_set_dev[dev & 63] = true;
    }
    const bool lean = !a.has_drift && a.nlin == 0;
    const dim3 grid((unsigned)tiles), block(kCuThreads);
    switch (a.precision) {
    case RVLL_PREC_MIXED: hipLaunchKernelGGL((rounds_cu_kernel<RVLL_PREC_MIXED, true>), grid, block, lds, stream, a, o, round); break;
    case RVLL_PREC_FP32:  hipLaunchKernelGGL((rounds_cu_kernel<RVLL_PREC_FP32, true>), grid, block, lds, stream, a, o, round); break;
    default:
        if (lean) hipLaunchKernelGGL((rounds_cu_kernel<RVLL_PREC_FP64, false>), grid, block, lds, stream, a, o, round);
        else      hipLaunchKernelGGL((rounds_cu_kernel<RVLL_PREC_FP64, true>), grid, block, lds, stream, a, o, round);
        break;
    }
    return hipGetLastError();
}

hipError_t launch_loglike_trace(const LoglikeArgs& a, hipStream_t stream)
{
    if (a.B <= 0) return hipSuccess;
    if (!a.trace || a.precision != RVLL_PREC_FP64) return hipErrorInvalidValue;
    const long long blocks = (a.B + a.PB - 1) / a.PB;
    hipLaunchKernelGGL(loglike_trace_kernel, dim3((unsigned)blocks), dim3(kThreads), loglike_lds_bytes(a), stream, a);
    return hipGetLastError();
}

hipError_t launch_scalar_server(const LoglikeArgs& a, ServerCtl* ctl, unsigned long long last,
                                unsigned long long idle_ticks, hipStream_t stream)
{
    if (a.B != 1 || a.PB != 1 || !ctl || a.D > kServerMaxDim) return hipErrorInvalidValue;
    const size_t lds = loglike_lds_bytes(a) + 2 * sizeof(unsigned long long) + sizeof(double) * (size_t)a.D;   // + request words, theta copy
    const dim3 grid(1), block(kThreads);
    switch (a.precision) {
    case RVLL_PREC_MIXED: hipLaunchKernelGGL((scalar_server_kernel<RVLL_PREC_MIXED>), grid, block, lds, stream, a, ctl, last, idle_ticks); break;
    case RVLL_PREC_FP32:  hipLaunchKernelGGL((scalar_server_kernel<RVLL_PREC_FP32>), grid, block, lds, stream, a, ctl, last, idle_ticks); break;
    default:              hipLaunchKernelGGL((scalar_server_kernel<RVLL_PREC_FP64>), grid, block, lds, stream, a, ctl, last, idle_ticks); break;
    }
    return hipGetLastError();
}

hipError_t launch_prior_loglike(const LoglikeArgs& a, hipStream_t stream)
{
    if (a.B <= 0) return hipSuccess;
    if (!a.cube || !a.theta_out || !a.priors || (a.n_heavy > 0 && !a.heavy_dims)) return hipErrorInvalidValue;
    const long long blocks = (a.B + a.PB - 1) / a.PB;
    const size_t lds = loglike_lds_bytes(a);
    const dim3 grid((unsigned)blocks), block(kThreads);
    switch (a.precision) {
    case RVLL_PREC_MIXED: hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_MIXED, kFusedSlim>), grid, block, lds, stream, a); break;
    case RVLL_PREC_FP32:  hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_FP32, kFusedSlim>), grid, block, lds, stream, a); break;
    default:
        if (kPlanetCountKernels && a.Np == 3) hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_FP64, kFusedSlim, true, 3>), grid, block, lds, stream, a);
        else                                  hipLaunchKernelGGL((loglike_kernel<RVLL_PREC_FP64, kFusedSlim>), grid, block, lds, stream, a);
        break;
    }
    return hipGetLastError();
}

hipError_t launch_prior(const PriorArgs& a, hipStream_t stream)
{
    const long long n = a.B * a.D;
    if (n <= 0) return hipSuccess;
    long long blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(prior_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.n_heavy <= 0) return e;
    long long hb = (a.B * a.n_heavy + kThreads - 1) / kThreads;
    if (hb > 8192) hb = 8192;
    hipLaunchKernelGGL(prior_heavy_kernel, dim3((unsigned)hb), dim3(kThreads), 0, stream, a);
    return hipGetLastError();
}

int prior_table_nodes() { return kTableN; }
double prior_table_umax() { return kTableU; }

hipError_t launch_prior_table(int kind, const double* args, double* z, double* dz, unsigned long long* max_err_bits,
                              hipStream_t stream)
{
    const dim3 grid((kTableN + kThreads - 1) / kThreads), block(kThreads);
    hipLaunchKernelGGL(prior_table_kernel, grid, block, 0, stream, kind, args[0], args[1], args[2], z, dz);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !max_err_bits) return e;
    hipLaunchKernelGGL(prior_table_check_kernel, grid, block, 0, stream, kind, args[0], args[1], args[2], z, dz,
                       max_err_bits);
    return hipGetLastError();
}

double prior_table_direct_tol() { return kTableDirectTol; }

hipError_t launch_fill_cube(double* cube, long long n, uint64_t seed, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    long long blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fill_cube_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, cube, n, seed);
    return hipGetLastError();
}

}  // namespace rvll
