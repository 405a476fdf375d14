// rvll_rounds.h — the STEP of the proposal walk in its rounds form (rvll_kernels.h, RoundsArgs; the scheme of
// evidence_amd/nested.py run_nested_slice, which follows the reference's UltraNest wrapper: region slice sampling, nsteps moves
// per new point, circular omega / ml0 — evidence/ultranest/__init__.py:159-175): accept / shrink on the previous round's
// results, directions for the walkers that start a move, this round's candidates (compacted: one array, one count on the
// device), prior transform included; and the directions of all moves, made ahead of time.  Device code, included by
// rvll_kernels.hip (kernels rounds_dirs_kernel, rounds_step_kernel; rounds_tiles_kernel evaluates the candidates).
// Same arithmetic per walker as slice_walk_kernel (rvll_walk.hip), operation for operation — the counters of the random
// numbers name the walker and the move, never where or when it is evaluated — so the results are the same bits.
#pragma once
#include "rvll_tile.h"

namespace rvll {

namespace {

// ---- directions, ahead of time ------------------------------------------------------------------------------------------
// The direction of a walker's move m is chol * z / |chol z| with z drawn from counters that name (seed, walker, m) alone: it
// does not depend on where the walker is, so all of a walk's directions are made by ONE launch in front of the rounds
// ([K, nsteps, D] doubles: 142 MB at 16384 walkers x 57 moves x 19 parameters), instead of inside every step — where the
// Box-Muller normals, the triangular product and the norm of the walkers that start a move were half of a step's 35 us
// (profiles/r04_rounds_step_stamps.txt).  Same routines, same order of operations as slice_walk_kernel: same bits.
// One workgroup walks (walker, move) pairs, kDirPairs at a time: a lane per (pair, coordinate).
constexpr int kDirPairs = 12;
__host__ __device__ inline size_t dirs_lds_doubles(int D) { return (size_t)2 * kDirPairs * D + (D <= kWalkCholLds ? D * D : 0) + kDirPairs; }
__device__ __forceinline__ void rounds_dirs(const RoundsDirs& g, double* sm)
{
    const int D = g.D, tid = threadIdx.x;
    double* z = sm;                         // [kDirPairs][D] normals
    double* cv = z + kDirPairs * D;         // [kDirPairs][D] chol z
    double* chol_s = cv + kDirPairs * D;
    const bool chol_in_lds = D <= kWalkCholLds;
    double* rn = chol_s + (chol_in_lds ? D * D : 0);     // [kDirPairs] 1 / |chol z|
    if (chol_in_lds) for (int k = tid; k < D * D; k += kThreads) chol_s[k] = g.chol[k];
    const double* chol = chol_in_lds ? chol_s : g.chol;
    const long long npairs = g.K * g.nsteps;
    for (long long p0 = (long long)blockIdx.x * kDirPairs; p0 < npairs; p0 += (long long)gridDim.x * kDirPairs) {
        const int np = (int)min((long long)kDirPairs, npairs - p0);
        for (int idx = tid; idx < np * D; idx += kThreads) {
            const int q = idx / D, k = idx - q * D;
            const long long pr = p0 + q, w = pr / g.nsteps, m = pr - w * g.nsteps;
            const unsigned long long wid = g.wid0 + (unsigned long long)w;
            const unsigned long long ctr = (wid << 32) | ((unsigned long long)m << 14) | (unsigned)(2 * k);
            z[idx] = walk_normal(g.seed, ctr);
        }
        __syncthreads();                    // (the first time round: chol_s as well)
        for (int idx = tid; idx < np * D; idx += kThreads) {
            const int q = idx / D, k = idx - q * D;
            const double* cr = chol + k * D;
            const double* zq = z + q * D;
            double a = 0.;
            int j = 0;
            for (; j + 3 <= k; j += 4) {    // the sum in index order, the loads of four terms issued together
                const double c0 = cr[j], c1 = cr[j + 1], c2 = cr[j + 2], c3 = cr[j + 3];
                const double z0 = zq[j], z1 = zq[j + 1], z2 = zq[j + 2], z3 = zq[j + 3];
                a += c0 * z0; a += c1 * z1; a += c2 * z2; a += c3 * z3;
            }
            for (; j <= k; ++j) a += cr[j] * zq[j];
            cv[idx] = a;
        }
        __syncthreads();
        if (tid < np) {
            const double* v = cv + tid * D;
            double n2 = 0.;
            for (int j = 0; j < D; ++j) n2 += v[j] * v[j];
            rn[tid] = 1. / sqrt(n2);
        }
        __syncthreads();
        for (int idx = tid; idx < np * D; idx += kThreads) g.dirs[p0 * D + idx] = cv[idx] * rn[idx / D];
        __syncthreads();
    }
}

// ---- the step ---------------------------------------------------------------------------------------------------------------
// LDS of a step workgroup: W walkers' positions, directions (of the move in progress / of the move after it) and a chunk of
// candidate rows; their result records
struct StepLds {
    double *us, *dr, *dn, *rows, *rec_l, *rec_t, *tmn, *tmx, *tacc;
    int *rec_f, *rec_d, *st, *lf, *nsp, *acc, *stp, *rnd, *first, *starts, *cnt, *wrapped;
};
__host__ __device__ inline size_t step_lds_doubles(int W, int D, int SM) { return (size_t)4 * W * D + (size_t)2 * W * SM + 3 * W; }
__host__ __device__ inline size_t step_lds_ints(int W, int D, int SM) { return (size_t)2 * W * SM + 8 * W + 4 + D; }
__device__ __forceinline__ StepLds step_views(double* sm, int W, int D, int SM)
{
    StepLds s;
    s.us = sm;                      s.dr = s.us + W * D;   s.dn = s.dr + W * D;   s.rows = s.dn + W * D;
    s.rec_l = s.rows + W * D;       s.rec_t = s.rec_l + W * SM;
    s.tmn = s.rec_t + W * SM;       s.tmx = s.tmn + W;     s.tacc = s.tmx + W;
    s.rec_f = reinterpret_cast<int*>(s.tacc + W);
    s.rec_d = s.rec_f + W * SM;
    s.st = s.rec_d + W * SM;        s.lf = s.st + W;       s.nsp = s.lf + W;      s.acc = s.nsp + W;     s.stp = s.acc + W;
    s.rnd = s.stp + W;              s.first = s.rnd + W;   s.starts = s.first + W;  s.cnt = s.starts + W;  s.wrapped = s.cnt + 4;
    return s;
}

// One workgroup = W walkers of the group (a lane of wave 0 each for what is serial per walker, all 256 threads over
// (walker, coordinate) for the rest).  `block`: the workgroup's index among the step's workgroups.
// What bounds a step is latency, not work (a few thousand instructions per walker against ~2 us per dependent access to HBM, a
// single wave per SIMD): everything it needs is fetched in ONE round trip at the top — the walkers' words, brackets,
// positions, the direction of the move in progress and of the one after it, the result records (indexed by walker: the tiles
// put them there) — then one atomic for the slots.
__device__ __forceinline__ void rounds_step(const RoundsArgs& g, const int r, const int block, double* sm)
{
    const int D = g.D, W = g.W, SM = g.spec_max, tid = threadIdx.x, lane = tid & (kWave - 1);
    const long long i0 = (long long)block * W;
    const int nw = (int)min((long long)W, g.K - i0);
    if (nw <= 0) return;
    const StepLds s = step_views(sm, W, D, SM);
    const double one_below = 0.99999999999999988898;        // nextafter(1, 0)
    int* const ring_now = g.ring + 2 * (r % kRoundsRing);
    unsigned long long* const stamp = (g.stamps && r < g.stamp_rounds && threadIdx.x == 0)
                                          ? g.stamps + 8 * ((size_t)r * ((g.K + W - 1) / W) + block) : nullptr;
    if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();

    // ---- the one round trip
    const bool wlane = tid < kWave, valid = wlane && lane < nw;
    int4 w4 = make_int4(4, 0, 0, 0);
    int stp = 0, nprev = (int)min(g.K, (long long)0x7fffffff);
    double tmn = 0., tmx = 0.;
    if (r > 0) {
        if (valid) {
            w4 = reinterpret_cast<const int4*>(g.ws)[i0 + lane];
            stp = g.step[i0 + lane]; tmn = g.tmin[i0 + lane]; tmx = g.tmax[i0 + lane];
        }
        if (wlane) nprev = g.ring[2 * ((r - 1) % kRoundsRing) + 1];
        for (int idx = tid; idx < nw * SM; idx += kThreads) {
            s.rec_l[idx] = g.wres_logl[i0 * SM + idx];
            s.rec_f[idx] = g.wres_flags[i0 * SM + idx];
            s.rec_d[idx] = g.wdef[i0 * SM + idx];
            s.rec_t[idx] = g.wt[i0 * SM + idx];
        }
        for (int idx = tid; idx < nw * D; idx += kThreads) { s.dr[idx] = g.dir[i0 * D + idx]; s.dn[idx] = g.dirnext[i0 * D + idx]; }
    } else {
        if (valid) w4.x = g.nsteps > 0 ? 0 : 4;
        if (g.nsteps > 0)                        // every walker starts move 0: its direction is the first row of its table
            for (int idx = tid; idx < nw * D; idx += kThreads) s.dn[idx] = g.dirs[((i0 + idx / D) * g.nsteps) * D + idx % D];
    }
    for (int idx = tid; idx < nw * D; idx += kThreads) s.us[idx] = g.u[i0 * D + idx];
    for (int k = tid; k < D; k += kThreads) s.wrapped[k] = g.wrapped[k];
    __syncthreads();
    if (stamp) stamp[1] = __builtin_amdgcn_s_memrealtime();

    if (wlane) {
        // ---- A. a lane per walker: consume the slots it was given in the previous round, in the order it would have met
        //         them — accept the first candidate above lstar, shrink past the others (slice_walk_kernel's accept step)
        int state = w4.x, round = w4.y, nsp = w4.w;
        const int first = w4.z;
        int acc = -1, accw = 0;
        long long used = 0;
        double newl = 0., tacc = 0.;
        if (r > 0 && state == 1) {
            const int round0 = round;
            for (int j = 0; j < nsp; ++j) {
                const double cl = s.rec_l[lane * SM + j];
                const double t = s.rec_t[lane * SM + j];
                used = j + 1;
                if (s.rec_d[lane * SM + j]) {
                    // leave at the START of this move: the full-solver walk retraces it from its first candidate, so none
                    // of this move's candidates count here (the ones of earlier rounds were counted then)
                    state = 3;
                    used = -(long long)round0;
                    break;
                }
                if (cl > g.lstar) { state = 0; stp += 1; acc = j; newl = cl; tacc = t; accw = (s.rec_f[lane * SM + j] & RVLL_FLAG_WANDERED) ? 1 : 0; break; }
                if (t < 0.) tmn = t; else tmx = t;
                if (++round >= g.max_rounds) { state = 0; stp += 1; break; }     // give the move up, stay put
            }
        }
        if (state <= 1 && stp >= g.nsteps) state = 4;
        const bool listed = valid && state <= 1;
        // this round's slots: every listed walker one; while the round is below the log-L kernel's latency floor
        // (c_free slots) the free ones go to candidates AHEAD, at most spec_max per walker and never past the move's last
        // round.  Judged by the number of walkers listed in the PREVIOUS round (this round's is only known when every
        // workgroup has been here): never fewer than now, so the slots handed out cannot exceed C.
        int S = 0;
        if (listed) {
            S = min(SM, g.c_free / max(1, nprev));
            S = max(1, min(S, g.max_rounds - (state == 0 ? 0 : round)));
        }
        int incl = S;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { const int v = __shfl_up(incl, off, kWave); if (lane >= off) incl += v; }
        const int tot = __shfl(incl, kWave - 1, kWave), excl = incl - S;
        const unsigned long long m_listed = __ballot(listed), lanes_below = (1ull << lane) - 1ull;
        int base = 0;
        if (lane == 0) {
            // slots (low word) and walkers listed (high word) of the round in ONE atomic: same-address atomics of a launch
            // are served one after the other
            if (tot) base = (int)(unsigned)atomicAdd(reinterpret_cast<unsigned long long*>(ring_now),
                                                     ((unsigned long long)__popcll(m_listed) << 32) | (unsigned)tot);
            if (block == 0) {                                        // the entry two rounds on is free again: zero it for its round
                int* z = g.ring + 2 * ((r + 2) % kRoundsRing);
                z[0] = 0; z[1] = 0;
            }
        }
        base = __shfl(base, 0, kWave);
        const bool begins = listed && state == 0;
        if (valid) {
            reinterpret_cast<int4*>(g.ws)[i0 + lane] = make_int4(listed ? 1 : state, begins ? 0 : round, base + excl, S);
            g.step[i0 + lane] = stp;
            if (acc >= 0) { g.logl[i0 + lane] = newl; g.wflag[i0 + lane] = accw; }
            if (listed && state == 1) { g.tmin[i0 + lane] = tmn; g.tmax[i0 + lane] = tmx; }
        }
        if (lane < W) {
            s.st[lane] = listed ? state : 4;  s.lf[lane] = excl;  s.nsp[lane] = S;  s.acc[lane] = valid ? acc : -1;
            s.stp[lane] = stp;  s.rnd[lane] = begins ? 0 : round;  s.tmn[lane] = tmn;  s.tmx[lane] = tmx;
            s.tacc[lane] = tacc;  s.first[lane] = first;
        }
        const unsigned long long m_begin = __ballot(begins);
        if (begins) s.starts[__popcll(m_begin & lanes_below)] = lane;
        long long usum = used;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) usum += __shfl_down(usum, off, kWave);
        if (lane == 0) {
            s.cnt[0] = __popcll(m_begin);  s.cnt[1] = tot;  s.cnt[2] = base;
            if (usum) g.calls_part[block] += usum;
            if (tot) g.slots_part[block] += (unsigned long long)tot;
        }
    }
    __syncthreads();
    if (stamp) stamp[2] = __builtin_amdgcn_s_memrealtime();

    // ---- accepted candidates move in: recomputed from position, direction and step — the arithmetic that made them, as
    //      slice_walk_kernel does at this point — and theta from the row the step of that round wrote for the slot
    for (int idx = tid; idx < nw * D; idx += kThreads) {
        const int wl = idx / D, k = idx - wl * D, aj = s.acc[wl];
        if (aj < 0) continue;
        double c = s.us[idx] + s.tacc[wl] * s.dr[idx];
        if (s.wrapped[k]) c -= floor(c);
        c = fmin(fmax(c, 0.), one_below);
        s.us[idx] = c;
        g.u[i0 * D + idx] = c;
        g.theta[i0 * D + idx] = g.theta_c[(r + 1) & 1][(long long)(s.first[wl] + aj) * D + k];
    }
    __syncthreads();                             // (the lanes below read positions, and replace directions, other lanes handled above)
    // ---- B. walkers starting a move: its direction was fetched above (made ahead of time, rounds_dirs); the chord limits
    //      of every coordinate, a lane per (walker, coordinate); and the direction of the move after it is put where the
    //      step that needs it will find it
    const int nstart = s.cnt[0];
    for (int idx = tid; idx < nstart * D; idx += kThreads) {
        const int q = idx / D, k = idx - q * D, wl = s.starts[q], e = wl * D + k;
        const double d = s.dn[e], u = s.us[e];
        s.dr[e] = d;
        g.dir[i0 * D + e] = d;
        if (s.stp[wl] + 1 < g.nsteps) g.dirnext[i0 * D + e] = g.dirs[((i0 + wl) * g.nsteps + s.stp[wl] + 1) * D + k];
        double lo = -INFINITY, hi = INFINITY;
        if (d != 0.) {
            if (s.wrapped[k]) {
                const double half = 0.5 / fabs(d);
                lo = -half; hi = half;
            } else {
                const double t0 = (0. - u) / d, t1 = (1. - u) / d;
                lo = fmin(t0, t1); hi = fmax(t0, t1);
            }
        }
        s.rows[e] = lo; s.dn[e] = hi;
    }
    __syncthreads();
    // ... the chord (max / min over the coordinates in order)
    if (tid < nstart) {
        const int wl = s.starts[tid];
        double lo = -INFINITY, hi = INFINITY;
        const double* pl_ = s.rows + wl * D;
        const double* ph_ = s.dn + wl * D;
        int k = 0;
        for (; k + 4 <= D; k += 4) {
            const double l0 = pl_[k], l1 = pl_[k + 1], l2 = pl_[k + 2], l3 = pl_[k + 3];
            const double h0 = ph_[k], h1 = ph_[k + 1], h2 = ph_[k + 2], h3 = ph_[k + 3];
            lo = fmax(fmax(fmax(fmax(lo, l0), l1), l2), l3);
            hi = fmin(fmin(fmin(fmin(hi, h0), h1), h2), h3);
        }
        for (; k < D; ++k) { lo = fmax(lo, pl_[k]); hi = fmin(hi, ph_[k]); }
        s.tmn[wl] = lo; s.tmx[wl] = hi;
        g.tmin[i0 + wl] = lo; g.tmax[i0 + wl] = hi;
    }
    __syncthreads();
    if (stamp) stamp[3] = __builtin_amdgcn_s_memrealtime();
    // this round's candidate along the chord and, in the walker's further slots, the ones the next rounds draw if it is
    // rejected (the bracket after a rejection ends at the rejected candidate); positions along the direction into the
    // walker's record, which slot it went to into the slot's owner word
    const int base = s.cnt[2];
    int* const slot_w = s.rec_f;                 // [W * SM] local slot -> walker (the records have been consumed)
    int* const slot_def = s.rec_d;               // [W * SM] local slot: a coordinate's quantile is beyond the verified tables
    double* const slot_t = s.rec_l;
    if (tid < nw && s.st[tid] <= 1) {
        const int wl = tid;
        double lo = s.tmn[wl], hi = s.tmx[wl];
        const unsigned long long wid = g.wid0 + (unsigned long long)(i0 + wl);
        const unsigned long long ctr = (wid << 32) | ((unsigned long long)s.stp[wl] << 14) | (unsigned)(8192 + s.rnd[wl]);
        const int f = s.lf[wl], S = s.nsp[wl];
        for (int j = 0; j < S; ++j) {
            const double t = lo + (hi - lo) * uniform01(g.seed, ctr + (unsigned)j);
            slot_t[f + j] = t; slot_w[f + j] = wl; slot_def[f + j] = 0;
            g.wt[(i0 + wl) * SM + j] = t;
            g.owner[base + f + j] = (int)((i0 + wl) * SM + j);
            if (t < 0.) lo = t; else hi = t;
        }
    }
    __syncthreads();
    if (stamp) stamp[4] = __builtin_amdgcn_s_memrealtime();
    // The candidates — and their PRIOR TRANSFORM, here rather than in front of the tiles: a tile of eight or ten points runs
    // every prior kind of the model in every wave of its staging step (what made a candidate of the single-kernel walk cost
    // 2600 vector instructions against the batch kernel's 1990), while this loop has a workgroup's candidates side by side
    // and walks them parameter by parameter, point-fastest — a wave works on one kind.  The tiles then are the plain
    // theta -> log-L kernel.  Rows are made in LDS, transformed in place (same routines as every other path: prior_light,
    // prior_heavy_slim — same bits) and leave as whole rows.
    const int nloc = s.cnt[1], n_light = D - g.n_heavy, CR = W;
    double* const rows = s.rows;                 // [CR][D]
    double* const thc = g.theta_c[r & 1] + (long long)base * D;
    for (int c0 = 0; c0 < nloc; c0 += CR) {
        const int nr = min(CR, nloc - c0);
        for (int idx = tid; idx < nr * D; idx += kThreads) {
            const int q = idx / D, k = idx - q * D, wl = slot_w[c0 + q];
            double c = s.us[wl * D + k] + slot_t[c0 + q] * s.dr[wl * D + k];
            if (s.wrapped[k]) c -= floor(c);
            rows[idx] = fmin(fmax(c, 0.), one_below);
        }
        __syncthreads();
        if (stamp && c0 == 0) stamp[5] = __builtin_amdgcn_s_memrealtime();
        for (int i = tid; i < nr * n_light; i += kThreads) {
            const int k = i / nr, q = i - k * nr, d = g.light_dims[k];
            const double v = prior_light(g.priors, D, rows + q * D, d);
            if (g.inplace) rows[q * D + d] = v; else thc[(long long)(c0 + q) * D + d] = v;
        }
        for (int i = kThreads - 1 - tid; i < nr * g.n_heavy; i += kThreads) {      // (dealt from the back: other waves than the light tail)
            const int k = i / nr, q = i - k * nr, d = g.heavy_dims[k];
            bool deferred = false;
            const double v = prior_heavy_slim(g.priors[d], rows[q * D + d], g.slim_umax, deferred);
            if (deferred) slot_def[c0 + q] = 1;
            if (g.inplace) rows[q * D + d] = v; else thc[(long long)(c0 + q) * D + d] = v;
        }
        __syncthreads();
        if (stamp && c0 == 0) stamp[6] = __builtin_amdgcn_s_memrealtime();
        if (g.inplace) for (int idx = tid; idx < nr * D; idx += kThreads) thc[(long long)c0 * D + idx] = rows[idx];
        if (c0 + CR < nloc) __syncthreads();
    }
    for (int ls = tid; ls < nloc; ls += kThreads) {
        const int wl = slot_w[ls];
        g.wdef[(i0 + wl) * SM + (ls - s.lf[wl])] = slot_def[ls];
    }
    if (stamp) stamp[7] = __builtin_amdgcn_s_memrealtime();
}

inline size_t step_lds_bytes(int W, int D, int SM)
{
    return sizeof(double) * step_lds_doubles(W, D, SM) + sizeof(int) * step_lds_ints(W, D, SM) + 16;
}

}  // namespace

}  // namespace rvll
