// rvll_walk.hip — the device-resident proposal walk of nested sampling (rvll_slice_walk), a translation unit of
// its own because it is built with -mllvm -disable-machine-licm (csrc/Makefile): the walk wraps the prior
// transform and the whole log-L tile in an iteration loop, and with loop-invariant code motion hipcc hoists every
// constant of that nest (polynomial coefficients of pow / exp / log / ndtri and of the tile) out of the loop and
// keeps them live across it — 242 VGPRs, 2 waves per SIMD, or spills under any lower cap.  Materialising them where
// they are used instead takes 117 VGPRs with nothing spilled: 4 waves per SIMD.  (For the batch kernels the same
// flag would put ~30 extra scalar moves into every Newton iteration, enough to saturate the CU's one scalar unit;
// they keep the default.)
#define RVLL_LOCAL_CONSTS 1      // the Newton loop's constants are loaded in front of it, not inside it (rvll_math.h)
#include "rvll_tile.h"

namespace rvll {

namespace {

// Device-resident slice-sampling walk (rvll_kernels.h, WalkArgs; the scheme of evidence_amd/nested.py
// run_nested_slice, which follows the reference's UltraNest wrapper: region slice sampling, nsteps moves per new
// point, circular omega / ml0 — evidence/ultranest/__init__.py:159-175).  Everything a move needs stays on the
// chip: counter-based random numbers, directions, chords, candidates (written to the workgroup's scratch rows),
// prior transform + log-L of the candidates through the same loglike_tile as every other path, accept / shrink.
// The walkers of a workgroup are NOT in lock step: every iteration evaluates one candidate for every walker that
// still has moves left, and a walker whose candidate was accepted draws its next direction in the following
// iteration — so the tile stays full until the walkers run out of moves (their totals over nsteps moves are
// close), instead of idling behind the slowest walker of every move.  Trip counts are bounded by
// nsteps * max_rounds and shared through LDS, so all waves loop alike.
// FAT = false: the prior stage evaluates Beta / Gamma quantiles by their verified tables only (rvll_tile.h,
// prior_heavy_slim).  A walker whose candidate needs anything else stops at the START of that move and reports the
// number of completed moves in steps_done; the host finishes those walkers with the FAT instantiation (full solvers
// inline, 2 waves per SIMD), whose counter-based random numbers make it retrace the interrupted move exactly — so
// the pair returns what a FAT-only walk would.
template <int PREC, bool FAT>
__global__ __launch_bounds__(kThreads, FAT ? 2 : 4) __attribute__((flatten))
void slice_walk_kernel(const LoglikeArgs a, const WalkArgs w)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const Carve cv = carve(a.PB, a.D, a.Np, a.Ni, a.nlin, a.CH);
    const int D = a.D, PB = a.PB, tid = threadIdx.x;
    const long long w0 = (long long)blockIdx.x * PB;
    const int nw = (int)min((long long)PB, w.K - w0);
    if (nw <= 0) return;
    double* wu   = smem + ((cv.total_doubles + 1) & ~1);   // [PB][D] current positions
    double* dir  = wu + PB * D;                            // [PB][D] normals, then unit directions
    double* lo_s = dir + PB * D;                           // [PB][D] per-coordinate chord limits of a starting move
    double* hi_s = lo_s + PB * D;
    double* tmin = hi_s + PB * D;                          // [PB]
    double* tmax = tmin + PB;
    double* tcur = tmax + PB;
    double* wl   = tcur + PB;
    int* act     = reinterpret_cast<int*>(wl + PB);         // [PB] walkers with moves left (local index), compacted
    int* state   = act + PB;                                // [PB] 0: needs a new direction, 1: in a move, 2: accepted just now, 3: deferred
    int* step_of = state + PB;                              // [PB] moves completed
    int* round_of = step_of + PB;                           // [PB] candidates tried in the current move
    int* nact_s  = round_of + PB;                           // [1]
    const double one_below = 0.99999999999999988898;        // nextafter(1, 0)

    for (int i = tid; i < nw * D; i += kThreads) wu[i] = w.u[w0 * D + i];
    for (int i = tid; i < nw; i += kThreads) {
        wl[i] = w.logl[w0 + i]; state[i] = 0; round_of[i] = 0;
        step_of[i] = w.step_start ? w.step_start[w0 + i] : 0;
    }
    __syncthreads();
    if (tid == 0) {                                         // walkers that still have moves to make
        int n = 0;
        for (int i = 0; i < nw; ++i) if (step_of[i] < w.nsteps) act[n++] = i;
        nact_s[0] = n;
    }
    unsigned long long calls = 0;                           // thread 0 only
    __syncthreads();

    const long long max_iters = (long long)w.nsteps * w.max_rounds;
    for (long long iter = 0; iter < max_iters; ++iter) {
        const int nact = nact_s[0];
        if (nact == 0) break;
        // walkers starting a move: standard normals (Box-Muller on two counter-based uniforms) ...
        for (int i = tid; i < nact * D; i += kThreads) {
            const int pl = act[i / D], k = i % D;
            if (state[pl] != 0) continue;
            const unsigned long long wid = (unsigned long long)(w.walker_base + (w.walker_id ? (long long)w.walker_id[w0 + pl] : w0 + pl));
            const unsigned long long ctr = (wid << 32) | ((unsigned long long)step_of[pl] << 14) | (unsigned)(2 * k);
            const double u1 = uniform01(w.seed, ctr), u2 = uniform01(w.seed, ctr + 1);
            double sn, cs;
            sincos_f64(kTwoPi * u2, sn, cs);
            dir[pl * D + k] = sqrt(-2. * log(1. - u1)) * cs;
        }
        __syncthreads();
        // ... direction = chol * z (lower triangular; held in registers until every z has been read) ...
        double mine[4];                                     // PB * D <= 4 * kThreads
        int cnt = 0;
        for (int i = tid; i < nact * D; i += kThreads, ++cnt) {
            const int pl = act[i / D], k = i % D;
            if (state[pl] != 0) continue;
            double acc = 0.;
            for (int j = 0; j <= k; ++j) acc += w.chol[k * D + j] * dir[pl * D + j];
            mine[cnt & 3] = acc;
        }
        __syncthreads();
        cnt = 0;
        for (int i = tid; i < nact * D; i += kThreads, ++cnt) {
            const int pl = act[i / D], k = i % D;
            if (state[pl] == 0) dir[pl * D + k] = mine[cnt & 3];
        }
        __syncthreads();
        // ... normalise: 1 / |dir| per walker (parked in tmin until the chord is known) ...
        for (int ai = tid; ai < nact; ai += kThreads) {
            const int pl = act[ai];
            if (state[pl] != 0) continue;
            double n2 = 0.;
            for (int k = 0; k < D; ++k) n2 += dir[pl * D + k] * dir[pl * D + k];
            tmin[pl] = 1. / sqrt(n2);
        }
        __syncthreads();
        // ... unit direction and the chord limits of every coordinate, one lane per (walker, coordinate): the two
        // divisions per coordinate used to run serially in ONE lane per walker ...
        for (int i = tid; i < nact * D; i += kThreads) {
            const int pl = act[i / D], k = i % D;
            if (state[pl] != 0) continue;
            const double d = dir[pl * D + k] * tmin[pl], u = wu[pl * D + k];
            dir[pl * D + k] = d;
            double lo = -INFINITY, hi = INFINITY;
            if (d != 0.) {
                if (w.wrapped[k]) {
                    const double half = 0.5 / fabs(d);
                    lo = -half; hi = half;
                } else {
                    const double t0 = (0. - u) / d, t1 = (1. - u) / d;
                    lo = fmin(t0, t1); hi = fmax(t0, t1);
                }
            }
            lo_s[pl * D + k] = lo; hi_s[pl * D + k] = hi;
        }
        __syncthreads();
        // ... the chord (same max / min sequence over the coordinates as before); then the candidate position along it
        for (int ai = tid; ai < nact; ai += kThreads) {
            const int pl = act[ai];
            if (state[pl] == 0) {
                double lo = -INFINITY, hi = INFINITY;
                for (int k = 0; k < D; ++k) { lo = fmax(lo, lo_s[pl * D + k]); hi = fmin(hi, hi_s[pl * D + k]); }
                tmin[pl] = lo; tmax[pl] = hi;
                round_of[pl] = 0;
                state[pl] = 1;
            }
            const unsigned long long wid = (unsigned long long)(w.walker_base + (w.walker_id ? (long long)w.walker_id[w0 + pl] : w0 + pl));
            const unsigned long long ctr = (wid << 32) | ((unsigned long long)step_of[pl] << 14) | (unsigned)(8192 + round_of[pl]);
            tcur[pl] = tmin[pl] + (tmax[pl] - tmin[pl]) * uniform01(w.seed, ctr);
        }
        __syncthreads();
        double* crow = const_cast<double*>(a.cube) + w0 * D;           // this workgroup's scratch rows
        for (int i = tid; i < nact * D; i += kThreads) {
            const int ai = i / D, k = i - ai * D, pl = act[ai];
            double c = wu[pl * D + k] + tcur[pl] * dir[pl * D + k];
            if (w.wrapped[k]) c -= floor(c);
            crow[ai * D + k] = fmin(fmax(c, 0.), one_below);
        }
        __syncthreads();
        loglike_tile<PREC, FAT ? kFusedFull : kFusedSlim>(a, smem, w0, nact);   // prior transform + log-L of the candidates
        __syncthreads();
        for (int ai = tid; ai < nact; ai += kThreads) {
            const int pl = act[ai];
            const double cl = a.logL[w0 + ai];
            if (!FAT && (a.flags[w0 + ai] & kFlagDeferred)) state[pl] = 3;      // leave at the start of this move
            else if (cl > w.lstar) { state[pl] = 2; wl[pl] = cl; }
            else {
                if (tcur[pl] < 0.) tmin[pl] = tcur[pl]; else tmax[pl] = tcur[pl];
                if (++round_of[pl] >= w.max_rounds) { state[pl] = 0; step_of[pl] += 1; }   // give the move up, stay put
            }
        }
        __syncthreads();
        for (int i = tid; i < nact * D; i += kThreads) {
            const int ai = i / D, k = i - ai * D, pl = act[ai];
            if (state[pl] != 2) continue;
            wu[pl * D + k] = crow[ai * D + k];
            w.theta[(w0 + pl) * D + k] = a.theta_out[(w0 + ai) * D + k];
        }
        __syncthreads();
        if (tid == 0) {
            calls += (unsigned long long)nact;
            int n = 0;
            for (int ai = 0; ai < nact; ++ai) {
                const int pl = act[ai];
                if (state[pl] == 2) { state[pl] = 0; step_of[pl] += 1; }
                if (step_of[pl] < w.nsteps && state[pl] != 3) act[n++] = pl;
            }
            nact_s[0] = n;
        }
        __syncthreads();
    }
    for (int i = tid; i < nw * D; i += kThreads) w.u[w0 * D + i] = wu[i];
    for (int i = tid; i < nw; i += kThreads) { w.logl[w0 + i] = wl[i]; if (w.steps_done) w.steps_done[w0 + i] = step_of[i]; }
    if (tid == 0 && calls) atomicAdd(w.ncalls, calls);
}

}  // namespace

size_t walk_lds_bytes(const LoglikeArgs& a)
{
    const size_t base = (loglike_lds_bytes(a) + 15) & ~(size_t)15;
    return base + sizeof(double) * ((size_t)4 * a.PB * a.D + 4 * a.PB) + sizeof(int) * (4 * a.PB + 2) + 16;
}

hipError_t launch_slice_walk(const LoglikeArgs& a, const WalkArgs& w, bool fat, hipStream_t stream)
{
    if (w.K <= 0 || w.nsteps <= 0) return hipSuccess;
    if (!a.cube || !a.theta_out || !a.priors || !a.flags || a.PB * a.D > 4 * kThreads || w.nsteps >= (1 << 18) ||
        w.max_rounds < 1 || w.max_rounds > 4096 || a.D > 4096 || (!fat && !w.steps_done))
        return hipErrorInvalidValue;
    const size_t lds = walk_lds_bytes(a);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((w.K + a.PB - 1) / a.PB)), block(kThreads);
#define RVLL_WALK(PREC)                                                                                              \
    if (fat) hipLaunchKernelGGL((slice_walk_kernel<PREC, true>), grid, block, lds, stream, a, w);                    \
    else     hipLaunchKernelGGL((slice_walk_kernel<PREC, false>), grid, block, lds, stream, a, w)
    switch (a.precision) {
    case RVLL_PREC_MIXED: RVLL_WALK(RVLL_PREC_MIXED); break;
    case RVLL_PREC_FP32:  RVLL_WALK(RVLL_PREC_FP32); break;
    default:              RVLL_WALK(RVLL_PREC_FP64); break;
    }
#undef RVLL_WALK
    return hipGetLastError();
}

}  // namespace rvll
