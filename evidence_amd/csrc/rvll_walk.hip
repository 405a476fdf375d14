// rvll_walk.hip — the device-resident proposal walk of nested sampling (rvll_slice_walk), a translation unit of
// its own because it is built with -mllvm -disable-machine-licm (csrc/Makefile): the walk wraps the prior
// transform and the whole log-L tile in an iteration loop, and with loop-invariant code motion hipcc hoists every
// constant of that nest (polynomial coefficients of pow / exp / log / ndtri and of the tile) out of the loop and
// keeps them live across it — 242 VGPRs, 2 waves per SIMD, or spills under any lower cap.  Materialising them where
// they are used instead takes 117 VGPRs with nothing spilled: 4 waves per SIMD.  (For the batch kernels the same
// flag would put ~30 extra scalar moves into every Newton iteration, enough to saturate the CU's one scalar unit;
// they keep the default.)
#define RVLL_LOCAL_CONSTS 1      // the Newton loop's constants are loaded in front of it, not inside it (rvll_math.h)
#ifndef RVLL_WALK_WAVES
#define RVLL_WALK_WAVES 4        // workgroups (= waves per SIMD) of the slim walk kernels a compute unit is to hold
#endif
#include <algorithm>
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
// SPECULATION.  The moves of a walker are a chain of dependent evaluations, and chains differ a lot in length (a
// walker in a narrow mode shrinks its bracket many more times per move): towards the end of a workgroup's life most
// of its tile is empty while the last walkers crawl on, one candidate per iteration.  Free tile slots are therefore
// given to the walkers that are left: a walker with S slots evaluates candidates r, r+1, .. r+S-1 of its move in ONE
// iteration, where candidate r+j is exactly the one it would draw in round r+j if r .. r+j-1 are all rejected (the
// bracket after a rejection is known before the rejection is: it ends at the rejected candidate).  The results are
// then consumed in order — accept the first one above lstar, shrink past the others — so positions, log-L, counters
// and ncalls are bit for bit those of the one-candidate-per-iteration walk, whatever the workgroup geometry; only
// the iterations a slow walker needs drop (nslots counts what was evaluated, ncalls what was used).
// WALKER QUEUE.  The launch has at most as many workgroups as the chip holds at once; a workgroup's PB walker slots
// start with walkers blockIdx * PB .. and every slot whose walker has finished (or was deferred) sends that walker's
// row home and takes the next one from a global ticket counter, so all slots stay busy until no walker is left —
// with a static split (one workgroup per PB walkers, two residency rounds at 16384 walkers) 28 % of the kernel's
// duration was its tail, workgroups waiting for their slowest walker while the rest of the chip had drained.  The
// random-number counters name the WALKER (row index + walker_base), never the slot, so which slot walks which row
// changes nothing in the results.  The order in which the rows are taken is the host's (w.order): it hands the second
// part of a walk out longest-expected first, from what every row cost in the first part (w.cost, rvll_api.hip).
// FAT = false: the prior stage evaluates Beta / Gamma quantiles by their verified tables only (rvll_tile.h,
// prior_heavy_slim).  A walker whose candidate needs anything else stops at the START of that move and reports the
// number of completed moves in steps_done; the host finishes those walkers with the FAT instantiation (full solvers
// inline, 2 waves per SIMD), whose counter-based random numbers make it retrace the interrupted move exactly — so
// the pair returns what a FAT-only walk would.
template <int PREC, bool FAT, int NP = 0>           // NP: the planet count at compile time (rvll_tile.h, eval_item), 0 = a.Np
__global__ __launch_bounds__(kThreads, FAT ? 2 : RVLL_WALK_WAVES) __attribute__((flatten))
void slice_walk_kernel(const LoglikeArgs a, const WalkArgs w)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const Carve cv = carve(a.PB, a.D, a.Np, a.Ni, a.nlin, a.CH);
    const int D = a.D, PB = a.PB, tid = threadIdx.x;
    const long long w0 = (long long)blockIdx.x * PB;
    const int nw = (int)min((long long)PB, w.K - w0);
    if (nw <= 0) return;
    double* wu   = smem + ((cv.total_doubles + 1) & ~1);   // [PB][D] current positions
    double* dir  = wu + PB * D;                            // [PB][D] unit directions
    double* tmin = dir + PB * D;                           // [PB]
    double* tmax = tmin + PB;
    double* slot_t = tmax + PB;                            // [PB] per tile slot: position along its walker's direction
    double* wl   = slot_t + PB;
    // The candidates' unit-cube rows [PB][D] and the per-coordinate chord limits of a starting move (2 x [PB][D]) live in the
    // tile's contribution window (round 3): the tile's staging step has read the candidates, and a barrier has passed,
    // before its items write the first contribution, and nothing outside the tile needs them afterwards — an accepted
    // candidate is recomputed from its walker's position, direction and step (the same arithmetic: the same bits).  That is
    // 3 PB D doubles less LDS per workgroup: twelve walker slots now fit four workgroups per compute unit, and the walk's own
    // phases — paid per workgroup iteration whatever the number of slots — are spread over half as many again.
    double* cand = smem + cv.contrib;
    double* lo_s = cand + PB * D;
    double* hi_s = lo_s + PB * D;
    const bool chol_in_lds = D <= kWalkCholLds;
    double* chol_s = wl + PB;                              // [D][D] the whitening factor, when it is small enough to stage
    int* act     = reinterpret_cast<int*>(chol_s + (chol_in_lds ? D * D : 0));   // [2][PB] walkers with moves left (local index),
                                                            // compacted; the list of the next iteration is written while this one's is read
    int* state   = act + 2 * PB;                              // [PB] 0: needs a new direction, 1: in a move, 2: accepted just now, 3: deferred
    int* step_of = state + PB;                              // [PB] moves completed
    int* round_of = step_of + PB;                           // [PB] candidates tried in the current move
    int* first_of = round_of + PB;                          // [PB] per walker: its first tile slot of this iteration ...
    int* nsp_of  = first_of + PB;                           // [PB] ... and how many it has (>= 1)
    int* acc_slot = nsp_of + PB;                            // [PB] per walker: the slot whose candidate was accepted
    int* used_of = acc_slot + PB;                           // [PB] per walker: candidates consumed this iteration
    int* slot_pl = used_of + PB;                            // [PB] per tile slot: its walker
    int* gid     = slot_pl + PB;                            // [PB] per walker slot: the row it is walking
    int* gold    = gid + PB;                                // [PB] ... the row that has just finished there (to be sent home)
    int* refill  = gold + PB;                               // [PB] 0 / 1: send gold home and load gid / 2: send gold home, slot stays empty
    int* acc_g   = refill + PB;                             // [PB] row of the walker whose candidate was accepted (= gid then)
    int* cost_of = acc_g + PB;                              // [PB] candidates the slot's current row has used in this launch
    int* starts  = cost_of + PB;                            // [PB] the listed walkers that start a move in this iteration
    int* nact_s  = starts + PB;                             // [4]  active walkers, tile slots, slots to refill, walkers starting a move
    int* wrapped_s = nact_s + 4;                            // [D]  circular parameters
    const double* chol = chol_in_lds ? chol_s : w.chol;
    const TileLds L = tile_views(a, smem);                  // the tile's results are read back from LDS (tile_point_result)
    const double one_below = 0.99999999999999988898;        // nextafter(1, 0)

    // the k-th row to be handed out (k: position in the host's order, or the row itself)
    auto row_at = [&](long long k) -> int { return w.order ? w.order[k] : (int)k; };
    for (int i = tid; i < nw * D; i += kThreads) wu[i] = w.u[(long long)row_at(w0 + i / D) * D + i % D];
    if (chol_in_lds) for (int i = tid; i < D * D; i += kThreads) chol_s[i] = w.chol[i];
    for (int i = tid; i < D; i += kThreads) wrapped_s[i] = w.wrapped[i];
    for (int i = tid; i < nw; i += kThreads) {
        const int g = row_at(w0 + i);
        wl[i] = w.logl[g]; state[i] = 0; round_of[i] = 0; refill[i] = 0; gid[i] = g; cost_of[i] = 0;
        step_of[i] = w.step_start ? w.step_start[g] : 0;
    }
    __syncthreads();
    // ---- bookkeeping by the LAST WAVE, one lane per walker slot (round 3; as one thread walking the slots it was a chain of
    //      dependent LDS reads — a tenth of a workgroup's life) ----
    // A lane accounts for its walker's candidates, moves it on after an accept, and — when the walker is done (all moves
    // made, or deferred) — marks its row to go home at the top of the next iteration and draws the next row nobody walks yet
    // from the global ticket counter (rows with nothing left to do are ticked off on the way); a slot that finds none stays
    // empty (gid = -1).  Then the next iteration's list, its tile slots (every listed walker one, the free ones dealt out
    // evenly, at most spec_max per walker and never past the move's last round; dealt by each row's own rejection rate
    // instead, 0.5 % more of the evaluated slots were used and the kernel was 3 % slower) and the walkers that start a move,
    // all by ballots.  Which slot draws which row is not deterministic any more — nor need it be: the random-number
    // counters name the row.
    const long long qbase = (long long)gridDim.x * PB;
    const int lane = tid & (kWave - 1);
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    bool queue_empty = false;                               // (per lane; a lane stops asking once it has seen the end)
    long long calls = 0;                                    // per lane of the bookkeeping wave
    unsigned long long slots = 0;
    auto bookkeep = [&](int* act_out, bool first) {
        const int pl = lane;
        const bool slot = pl < nw;
        const bool active = slot && gid[pl] >= 0;
        int st = active ? state[pl] : 1, stp = active ? step_of[pl] : 0;
        if (active && !first) { const int u = used_of[pl]; calls += u; cost_of[pl] += u; }
        if (active && st == 2) { st = 0; stp += 1; }
        const bool done = active && !(stp < w.nsteps && st != 3);
        int g = active ? gid[pl] : -1, rf = 0;
        if (done) {
            gold[pl] = g;
            used_of[pl] = cost_of[pl];                      // (carried to the top of the next iteration, where the row goes home)
            cost_of[pl] = 0;
            g = -1;
            while (!queue_empty) {
                const long long k = qbase + (long long)atomicAdd(w.queue, 1ull);
                if (k >= w.K) { queue_empty = true; break; }
                const int q = w.order ? w.order[k] : (int)k;
                const int ss = w.step_start ? w.step_start[q] : 0;
                if (ss < w.nsteps) { g = q; break; }
                if (w.steps_done) w.steps_done[q] = ss;
                if (w.cost) w.cost[q] = 0;
            }
            rf = g >= 0 ? 1 : 2;
            if (g >= 0) { st = 0; round_of[pl] = 0; }
        }
        if (slot) { refill[pl] = rf; gid[pl] = g; }
        if (active) { state[pl] = st == 3 ? 0 : st; step_of[pl] = stp; }
        const bool listed = slot && g >= 0;
        const unsigned long long m_act = __ballot(listed);
        const int n = __popcll(m_act), ai = __popcll(m_act & lanes_below);
        int S = 0;
        if (listed) {
            act_out[ai] = pl;
            S = min(w.spec_max, n ? nw / n + (ai < nw % n ? 1 : 0) : 0);
            S = max(1, min(S, w.max_rounds - (st == 0 ? 0 : round_of[pl])));
        }
        if (slot) nsp_of[pl] = S;
        // (the lanes of this wave read each other's word next: a wave-scope release + barrier, so that the order holds by the
        // memory model and not only by how the hardware happens to run a wave — ADVICE r3)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int f = 0;
        for (int k = 0; k < pl && k < nw; ++k) f += nsp_of[k];
        if (listed) first_of[pl] = f;
        if (lane == 63 - __builtin_clzll(m_act | 1ull) && listed) nact_s[1] = f + S;       // the last listed slot: the total
        const bool begins = listed && st == 0;
        const unsigned long long m_begin = __ballot(begins), m_done = __ballot(done);
        if (begins) starts[__popcll(m_begin & lanes_below)] = pl;
        if (lane == 0) {
            nact_s[0] = n;
            if (n == 0) nact_s[1] = 0;
            nact_s[2] = __popcll(m_done);
            nact_s[3] = __popcll(m_begin);
        }
    };
    int* const act0 = act;
    if (tid >= kThreads - kWave) bookkeep(act, true);       // walkers that still have moves to make; rows with none go home at once
    __syncthreads();

    // phase clock of a diagnostic build (make walktrace; scripts/walk_phase_probe.py): thread 0 sums the time between
    // barriers into four bins — directions and chord limits / candidates / prior transform + log-L tile / accept + copy
#ifdef RVLL_WALK_TRACE
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tph[4] = {0, 0, 0, 0}, last = __builtin_amdgcn_s_memrealtime();
#define WALK_STAMP(k) do { if (tid == 0) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); ph[k] += now - last; last = now; } } while (0)
#else
#define WALK_STAMP(k) do { } while (0)
#endif
    // every iteration consumes at least one candidate of every listed walker, so a slot's walkers end after at most
    // nsteps * max_rounds iterations each; the bound below is the formal exit for the case that all rows pass one slot
    const long long max_iters = ((long long)w.nsteps * w.max_rounds + 1) * (w.K + 1);
    for (long long iter = 0; iter < max_iters; ++iter) {
        const int nact = nact_s[0], nslots = nact_s[1], nref = nact_s[2];
        if (nref) {                                         // finished rows go home, the rows taking their slots come in
            for (int i = tid; i < nw * D; i += kThreads) {
                const int pl = i / D, k = i - pl * D, r = refill[pl];
                if (!r) continue;
                const long long go = gold[pl];
                w.u[go * D + k] = wu[i];
                if (k == 0) {
                    w.logl[go] = wl[pl];
                    if (w.steps_done) w.steps_done[go] = step_of[pl];
                    if (w.cost) w.cost[go] = used_of[pl];
                }
                if (r == 1) {
                    const long long gn = gid[pl];
                    wu[i] = w.u[gn * D + k];
                    if (k == 0) { wl[pl] = w.logl[gn]; step_of[pl] = w.step_start ? w.step_start[gn] : 0; }
                }
            }
            __syncthreads();
        }
        if (nact == 0) break;
        act = act0 + (iter & 1) * PB;
        int* const act_next = act0 + ((iter + 1) & 1) * PB;
        // walkers starting a move: standard normals (Box-Muller on two counter-based uniforms), parked in lo_s ...
        // (only the walkers that START a move — a third of them per iteration at cfg3 — draw directions; profiles/r03_walk_forms.txt)
        const int nstart = nact_s[3];
        for (int i = tid; i < nstart * D; i += kThreads) {
            const int pl = starts[i / D], k = i % D;
            const unsigned long long wid = (unsigned long long)(w.walker_base + (w.walker_id ? (long long)w.walker_id[gid[pl]] : (long long)gid[pl]));
            const unsigned long long ctr = (wid << 32) | ((unsigned long long)step_of[pl] << 14) | (unsigned)(2 * k);
            lo_s[pl * D + k] = walk_normal(w.seed, ctr);
        }
        __syncthreads();
        // ... direction = chol * z (lower triangular), parked in the candidate rows, which are free until the tile ...
        for (int i = tid; i < nstart * D; i += kThreads) {
            const int pl = starts[i / D], k = i % D;
            double acc = 0.;
            for (int j = 0; j <= k; ++j) acc += chol[k * D + j] * lo_s[pl * D + j];
            cand[pl * D + k] = acc;
        }
        __syncthreads();
        // ... unit direction and the chord limits of every coordinate, one lane per (walker, coordinate); every lane
        // sums the walker's norm itself (same order, same bits) rather than wait a barrier for one lane to do it ...
        for (int i = tid; i < nstart * D; i += kThreads) {
            const int pl = starts[i / D], k = i % D;
            double n2 = 0.;
            for (int j = 0; j < D; ++j) n2 += cand[pl * D + j] * cand[pl * D + j];
            const double d = cand[pl * D + k] * (1. / sqrt(n2)), u = wu[pl * D + k];
            dir[pl * D + k] = d;
            double lo = -INFINITY, hi = INFINITY;
            if (d != 0.) {
                if (wrapped_s[k]) {
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
        WALK_STAMP(0);
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
            // this round's candidate and, in the walker's further slots, the ones the next rounds draw if it is rejected
            const unsigned long long wid = (unsigned long long)(w.walker_base + (w.walker_id ? (long long)w.walker_id[gid[pl]] : (long long)gid[pl]));
            const unsigned long long ctr = (wid << 32) | ((unsigned long long)step_of[pl] << 14) | (unsigned)(8192 + round_of[pl]);
            double lo = tmin[pl], hi = tmax[pl];
            const int first = first_of[pl], S = nsp_of[pl];
            for (int j = 0; j < S; ++j) {
                const double t = lo + (hi - lo) * uniform01(w.seed, ctr + (unsigned)j);
                slot_t[first + j] = t; slot_pl[first + j] = pl;
                if (t < 0.) lo = t; else hi = t;
            }
        }
        __syncthreads();
        for (int i = tid; i < nslots * D; i += kThreads) {
            const int sl = i / D, k = i - sl * D, pl = slot_pl[sl];
            double c = wu[pl * D + k] + slot_t[sl] * dir[pl * D + k];
            if (wrapped_s[k]) c -= floor(c);
            cand[i] = fmin(fmax(c, 0.), one_below);
        }
        __syncthreads();
        WALK_STAMP(1);
        // prior transform + log-L of the candidates: rows read from LDS, results left in LDS (and in the scratch rows)
#ifdef RVLL_WALK_TRACE
        loglike_tile<PREC, FAT ? kFusedFull : kFusedSlim, false, kThreads, false, true, false, NP>(a, smem, w0, nslots, cand, LogdetPre{}, tph);
#else
        loglike_tile<PREC, FAT ? kFusedFull : kFusedSlim, false, kThreads, false, true, false, NP>(a, smem, w0, nslots, cand);
#endif
        // the walk's own phases are short and serial (a lane per walker, one thread for the bookkeeping): at the
        // default priority they get every fourth issue slot next to three workgroups in their item loops and a
        // barrier-to-barrier phase of ~50 instructions takes 1-2 us (phase clock: 33 % of a workgroup's life for a
        // few per cent of its instructions).  Raised here, lowered again by the tile in front of its item loop.
        __builtin_amdgcn_s_setprio(3);
        __syncthreads();
        WALK_STAMP(2);
        for (int ai = tid; ai < nact; ai += kThreads) {
            const int pl = act[ai];
            const int first = first_of[pl], S = nsp_of[pl];
            int used = 0;
            acc_slot[pl] = -1;
            for (int j = 0; j < S; ++j) {                               // in the order the walker would have met them
                int fl;
                const double cl = tile_point_result(a, L, first + j, fl);
                used = j + 1;
                if (!FAT && (fl & kFlagDeferred)) {
                    // leave at the start of this move; the full-solver pass retraces it from its first candidate, so none
                    // of this move's candidates count here (round_of: the ones of earlier iterations, counted then) —
                    // ncalls is what the full-solver walk alone reports
                    state[pl] = 3;
                    used = -(round_of[pl] - j);
                    break;
                }
                if (cl > w.lstar) {
                    state[pl] = 2; wl[pl] = cl; acc_slot[pl] = first + j; acc_g[pl] = gid[pl];
                    if (w.wflag) w.wflag[gid[pl]] = (fl & RVLL_FLAG_WANDERED) ? 1 : 0;
                    break;
                }
                const double t = slot_t[first + j];
                if (t < 0.) tmin[pl] = t; else tmax[pl] = t;
                if (++round_of[pl] >= w.max_rounds) { state[pl] = 0; step_of[pl] += 1; break; }     // give the move up, stay put
            }
            used_of[pl] = used;
        }
        __syncthreads();
        // accepted rows move in (every thread but the last) while the last thread does the bookkeeping: it writes the
        // NEXT iteration's list and touches nothing the copy reads
        for (int i = tid; i < nact * D; i += kThreads) {
            const int ai = i / D, k = i - ai * D, pl = act[ai];
            const int sl = acc_slot[pl];
            if (sl < 0) continue;
            double c = wu[pl * D + k] + slot_t[sl] * dir[pl * D + k];      // the accepted candidate, as it was made above
            if (wrapped_s[k]) c -= floor(c);
            wu[pl * D + k] = fmin(fmax(c, 0.), one_below);
            w.theta[(long long)acc_g[pl] * D + k] = L.theta_s[sl * D + k];
        }
        if (tid >= kThreads - kWave) {
            if (lane == 0) slots += (unsigned long long)nslots;
            bookkeep(act_next, false);
        }
        __syncthreads();
        WALK_STAMP(3);
    }
    // (only if the formal bound above ended the loop: rows still being walked go home as they are)
    for (int i = tid; i < nact_s[0] * D; i += kThreads) {
        const int pl = (act0 + (max_iters & 1) * PB)[i / D], k = i % D;
        const long long g = gid[pl];
        w.u[g * D + k] = wu[pl * D + k];
        if (k == 0) { w.logl[g] = wl[pl]; if (w.steps_done) w.steps_done[g] = step_of[pl]; if (w.cost) w.cost[g] = cost_of[pl]; }
    }
    if (tid >= kThreads - kWave && calls) atomicAdd(w.ncalls, (unsigned long long)calls);   // (two's complement: a negative share adds up right)
    if (tid >= kThreads - kWave && slots && w.nslots) atomicAdd(w.nslots, slots);
#ifdef RVLL_WALK_TRACE
    if (tid == 0 && w.nslots) {
        for (int k = 0; k < 4; ++k) atomicAdd(w.nslots + 1 + k, ph[k]);
        atomicAdd(w.nslots + 5, 1ull);
        atomicMax(w.nslots + 6, ph[0] + ph[1] + ph[2] + ph[3]);         // the longest workgroup life of the launch(es)
        for (int k = 0; k < 4; ++k) atomicAdd(w.nslots + 7 + k, tph[k]);  // the tile's own phases: stage, decode, items, reduce + write
    }
#endif
#undef WALK_STAMP
}


// ---- the same walk with the rows dealt to the workgroups in advance ("rows" form) -------------------------------
// The queue of slice_walk_kernel hands whole rows to walker slots, and a row is nsteps SEQUENTIAL moves: with about two
// rows per slot (16384 rows on the chip's 8192 slots) a slot that takes its last row a little before the others end
// runs a whole row alone — workgroup life mean 7.3 ms in a 9.4 ms kernel (profiles/r02_walk_phase_probe.txt: a quarter of
// the kernel is that tail).  Handing rows over between workgroups at a finer grain needs device-scope release / acquire,
// which on this part is an L2 write-back (round 2: time slices through a global ring ran 9x slower).  Here nothing is
// handed over: workgroup b OWNS rows_per_wg rows for the whole launch — the host sorts the rows by what they cost in the
// first part of the walk and the kernel deals them snake-wise (tier r of G rows forwards, tier r + 1 backwards), so every
// workgroup holds the same share of expensive and cheap rows — parks them in LDS (position, log-L, moves done: 22 doubles
// a row) and INTERLEAVES them over its PB walker slots at move boundaries: a slot whose walker ends a move parks it and
// takes the parked row that is furthest behind (kRowSlack moves or more).  All rows of a workgroup therefore advance
// together and end within kRowSlack moves of each other, all workgroups hold equal work, and the kernel ends when the work
// does.  The random-number counters name the row, so which slot walks which move of which row changes nothing: end points,
// log-L, theta and ncalls are those of slice_walk_kernel, bit for bit (tests/test_gpu_walk.py).
constexpr int kRowSlack = 2;
// NT = 256: four such workgroups per compute unit, a.PB (8) walker slots each, the tile in its 256-thread form.
// NT = 1024 (the CU-wide form, round 3): ONE workgroup fills the compute unit, up to 64 walker slots, the tile in its
// CU-wide form (every candidate's contributions resident in LDS, wave rounds from the ticket counter).  A candidate of
// the 8-slot form costs 2600 vector instructions against the batch tile's 1990 (profiles/r03_walk_forms.txt): the prior
// stage runs every prior kind in every wave at 8 points a tile, and the walk's own phases (directions, chords,
// candidates, accept, bookkeeping) are paid per workgroup iteration whatever the number of slots.  With 48 - 64 slots
// under one set of phases the stage's waves each run ONE kind, and the phases are amortised over six to eight times the
// candidates.
// NT = 512: two workgroups per compute unit, about 30 slots each, tickets as in the CU-wide form — one's phases under the
// other's tile.
template <int PREC, bool FAT, int NT>
__global__ __launch_bounds__(NT, NT == kCuThreads ? 1 : NT == 512 ? 4 : (FAT ? 2 : RVLL_WALK_WAVES)) __attribute__((flatten))
void slice_walk_rows_kernel(const LoglikeArgs a, const WalkArgs w)
{
    constexpr bool DYN = NT != kThreads;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const Carve cv = carve(a.PB, a.D, a.Np, a.Ni, a.nlin, a.CH);
    const int D = a.D, PB = a.PB, tid = threadIdx.x, R = w.rows_per_wg, G = gridDim.x;
    const long long w0 = (long long)blockIdx.x * PB;         // the workgroup's scratch rows of the tile
    const int nw = PB;                                        // tile slots
    double* wu   = smem + ((cv.total_doubles + 1) & ~1);   // [PB][D] current positions of the walkers in the slots
    double* dir  = wu + PB * D;
    double* tmin = dir + PB * D;
    double* tmax = tmin + PB;
    double* slot_t = tmax + PB;
    double* wl   = slot_t + PB;
    // The candidates' unit-cube rows and the per-coordinate chord limits live in the tile's contribution window: the tile's
    // staging step has read the candidates (and a barrier has passed) before its items write the first contribution, and
    // nothing outside the tile needs them afterwards — an accepted candidate is recomputed from its walker's position,
    // direction and step (the same arithmetic: the same bits).  3 PB D doubles of a window of a.CH (host-checked).
    double* cand = smem + cv.contrib;
    double* lo_s = cand + PB * D;
    double* hi_s = lo_s + PB * D;
    const bool chol_in_lds = D <= kWalkCholLds;
    double* chol_s = wl + PB;
    double* ru   = chol_s + (chol_in_lds ? D * D : 0);      // [R][D] parked rows: position ...
    double* rl   = ru + R * D;                              // [R]    ... log-L ...
    int* act     = reinterpret_cast<int*>(rl + R);
    int* state   = act + 2 * PB;
    int* step_of = state + PB;
    int* round_of = step_of + PB;
    int* first_of = round_of + PB;
    int* nsp_of  = first_of + PB;
    int* acc_slot = nsp_of + PB;
    int* used_of = acc_slot + PB;
    int* slot_pl = used_of + PB;
    int* srow    = slot_pl + PB;                            // [PB] the parked-row index of the walker in the slot, or -1
    int* park    = srow + PB;                               // [PB] row to park the slot's walker into at the top of the next iteration, or -1
    int* fetch   = park + PB;                               // [PB] row to load into the slot there, or -1
    int* acc_g   = fetch + PB;                              // [PB] global row of the walker whose candidate was accepted
    int* turn    = acc_g + PB;                              // [PB] moves the slot's walker has made since it took the slot
    int* starts  = turn + PB;                               // [PB] the listed walkers that start a move in this iteration
    int* ring_s  = starts + PB;                             // [2]  the ring of parked rows that wait: head, count
    int* nact_s  = ring_s + 2;                              // [4]  active walkers, tile slots, swaps pending, walkers starting a move
    int* wrapped_s = nact_s + 4;                            // [D]
    int* rrow    = wrapped_s + D;                           // [R] ... global row (or -1: none) ...
    int* rstep   = rrow + R;                                // [R] ... moves done ...
    int* rcost   = rstep + R;                               // [R] ... candidates used in this launch ...
    int* ring    = rcost + R;                               // [R] parked rows with moves left, first in first out
    const double* chol = chol_in_lds ? chol_s : w.chol;
    const TileLds L = tile_views(a, smem);
    const double one_below = 0.99999999999999988898;

    auto row_at = [&](long long k) -> int { return w.order ? w.order[k] : (int)k; };
    for (int r = tid; r < R; r += NT) {
        const long long pos = (long long)r * G + ((r & 1) ? G - 1 - (int)blockIdx.x : (int)blockIdx.x);    // snake deal
        const int g = pos < w.K ? row_at(pos) : -1;
        rrow[r] = g;
        rcost[r] = 0;
        rl[r] = g >= 0 ? w.logl[g] : 0.;
        const int ss = g >= 0 ? (w.step_start ? w.step_start[g] : 0) : w.nsteps;
        rstep[r] = ss;
    }
    __syncthreads();
    for (int i = tid; i < R * D; i += NT) { const int g = rrow[i / D]; ru[i] = g >= 0 ? w.u[(long long)g * D + i % D] : 0.; }
    if (chol_in_lds) for (int i = tid; i < D * D; i += NT) chol_s[i] = w.chol[i];
    for (int i = tid; i < D; i += NT) wrapped_s[i] = w.wrapped[i];
    for (int i = tid; i < PB; i += NT) { state[i] = 0; round_of[i] = 0; srow[i] = -1; park[i] = -1; fetch[i] = -1; step_of[i] = 0; turn[i] = 0; used_of[i] = 0; }
    const int lane = tid & (kWave - 1);
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    if (tid >= NT - kWave) {                          // the ring starts with every row that has moves left, in row order
        int count = 0;
        for (int r0 = 0; r0 < R; r0 += kWave) {
            const int r = r0 + lane;
            const bool has = r < R && rstep[r] < w.nsteps;
            const unsigned long long m = __ballot(has);
            if (has) ring[count + __popcll(m & lanes_below)] = r;
            count += __popcll(m);
        }
        if (lane == 0) { ring_s[0] = 0; ring_s[1] = count; }
    }
    __syncthreads();

    // Bookkeeping by the LAST WAVE, one lane per walker slot (as one thread walking the slots it was a chain of dependent LDS
    // reads: a third of a workgroup's life in this form, where most slots change rows at most move boundaries).  A walker
    // that has ended a move leaves its slot when it is done (all moves made, or deferred) or has made kRowSlack moves
    // there and a parked row waits; rows that leave with moves left go to the back of the ring, free slots take from its
    // front — first in, first out is furthest-behind first, because every turn in a slot is the same number of moves.
    // Positions in the ring come from ballots, so the lanes never wait for one another; a row pushed in this pass is
    // taken in a later one (its position is stored at the top of the next iteration, where the fetches happen too).
    long long calls = 0;                                    // per lane of the bookkeeping wave
    unsigned long long slots = 0;
    auto bookkeep = [&](int* act_out) {
        const int pl = lane;
        const bool slot = pl < PB;
        const int r = slot ? srow[pl] : -1;
        const bool active = r >= 0;
        int st = active ? state[pl] : 1, stp = active ? step_of[pl] : 0, tn = slot ? turn[pl] : 0;
        if (active) { const int u = used_of[pl]; calls += u; rcost[r] += u; }
        if (active && st == 2) { st = 0; stp += 1; }
        const bool boundary = active && st != 1;
        const bool done = boundary && (st == 3 || stp >= w.nsteps);
        if (boundary) { tn += 1; rstep[r] = stp; }
        const bool want = boundary && (done || tn >= kRowSlack);
        const int head = ring_s[0], navail = ring_s[1];
        const unsigned long long m_want = __ballot(want);
        const int ngets = min(__popcll(m_want), navail);
        const bool gets = want && __popcll(m_want & lanes_below) < navail;
        const bool leaves = want && (done || gets);
        const bool pushes = leaves && !done;
        const unsigned long long m_push = __ballot(pushes);
        const bool empty = slot && (!active || (leaves && !gets));
        const unsigned long long m_empty = __ballot(empty);
        const bool takes = empty && __popcll(m_empty & lanes_below) < navail - ngets;
        const int ntakes = min(__popcll(m_empty), navail - ngets);
        int q = -1;
        if (gets)  q = ring[(head + __popcll(m_want & lanes_below)) % R];
        if (takes) q = ring[(head + ngets + __popcll(m_empty & lanes_below)) % R];
        if (pushes) ring[(head + navail + __popcll(m_push & lanes_below)) % R] = r;
        int rnew = r;
        if (q >= 0) { rnew = q; st = 0; tn = 0; }
        else if (leaves) rnew = -1;
        if (slot) {
            park[pl] = leaves ? r : -1;
            fetch[pl] = q;
            srow[pl] = rnew;
            state[pl] = st == 3 ? 0 : st;
            step_of[pl] = stp;
            turn[pl] = tn;
            if (q >= 0) round_of[pl] = 0;
        }
        // the next iteration's list and its tile slots: every listed walker one, the free ones dealt out evenly, at most
        // spec_max per walker and never past the move's last round
        const bool listed = slot && rnew >= 0;
        const unsigned long long m_act = __ballot(listed);
        const int n = __popcll(m_act), ai = __popcll(m_act & lanes_below);
        int S = 0;
        if (listed) {
            act_out[ai] = pl;
            S = min(w.spec_max, n ? nw / n + (ai < nw % n ? 1 : 0) : 0);
            S = max(1, min(S, w.max_rounds - (st == 0 ? 0 : round_of[pl])));
        }
        if (slot) nsp_of[pl] = S;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (as in slice_walk_kernel: the lanes read each other's word next)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int f = 0;
        for (int k = 0; k < pl && k < PB; ++k) f += nsp_of[k];
        if (listed) first_of[pl] = f;
        if (lane == 63 - __builtin_clzll(m_act | 1ull) && listed) nact_s[1] = f + S;        // the last listed slot: the total
        const unsigned long long m_swap = __ballot(leaves || q >= 0);
        const bool begins = listed && st == 0;
        const unsigned long long m_begin = __ballot(begins);
        if (begins) starts[__popcll(m_begin & lanes_below)] = pl;
        if (lane == 0) {
            nact_s[0] = n;
            nact_s[3] = __popcll(m_begin);
            if (n == 0) nact_s[1] = 0;
            nact_s[2] = m_swap != 0ull ? 1 : 0;
            ring_s[0] = (head + ngets + ntakes) % R;
            ring_s[1] = navail - ngets - ntakes + __popcll(m_push);
        }
    };
    int* const act0 = act;
    if (tid >= NT - kWave) bookkeep(act);             // every slot is free: they take the first rows of the ring
#ifdef RVLL_AB_STAGGER
    // The workgroups of this form all start together and do the same work per iteration: the four that share a compute
    // unit would reach their tiles together and their serial phases together.  A start offset (0 .. 3 quarters of an
    // iteration, by a hash of the workgroup index) spreads them.
    {
        unsigned hsh = blockIdx.x * 2654435761u;
        const unsigned long long wait = ((hsh >> 13) & 3u) * (unsigned long long)RVLL_AB_STAGGER;   // ticks of 10 ns
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t_start < wait) __builtin_amdgcn_s_sleep(32);
    }
#endif
    __syncthreads();
#ifdef RVLL_WALK_TRACE
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tph[4] = {0, 0, 0, 0}, last = __builtin_amdgcn_s_memrealtime();
#define WALK_STAMP(k) do { if (tid == 0) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); ph[k] += now - last; last = now; } } while (0)
#else
#define WALK_STAMP(k) do { } while (0)
#endif
    // every iteration consumes at least one candidate of every listed walker: at most R * nsteps * max_rounds of them
    const long long max_iters = ((long long)w.nsteps * w.max_rounds + 1) * ((long long)R + 1);
    for (long long iter = 0; iter < max_iters; ++iter) {
        const int nact = nact_s[0], nslots = nact_s[1], nswap = nact_s[2];
        if (nswap) {                                        // walkers change places with parked rows (LDS to LDS)
            for (int i = tid; i < PB * D; i += NT) {
                const int pl = i / D, k = i - pl * D, st = park[pl], ld = fetch[pl];
                if (st >= 0) { ru[st * D + k] = wu[i]; if (k == 0) rl[st] = wl[pl]; }
                if (ld >= 0) { wu[i] = ru[ld * D + k]; if (k == 0) { wl[pl] = rl[ld]; step_of[pl] = rstep[ld]; } }
            }
            __syncthreads();
        }
        if (nact == 0) break;
        act = act0 + (iter & 1) * PB;
        int* const act_next = act0 + ((iter + 1) & 1) * PB;
        const int nstart = nact_s[3];                       // the walkers that start a move, packed (see slice_walk_kernel)
        for (int i = tid; i < nstart * D; i += NT) {
            const int pl = starts[i / D], k = i % D;
            const int g = rrow[srow[pl]];
            const unsigned long long wid = (unsigned long long)(w.walker_base + (w.walker_id ? (long long)w.walker_id[g] : (long long)g));
            const unsigned long long ctr = (wid << 32) | ((unsigned long long)step_of[pl] << 14) | (unsigned)(2 * k);
            lo_s[pl * D + k] = walk_normal(w.seed, ctr);
        }
        __syncthreads();
        for (int i = tid; i < nstart * D; i += NT) {
            const int pl = starts[i / D], k = i % D;
            double acc = 0.;
            for (int j = 0; j <= k; ++j) acc += chol[k * D + j] * lo_s[pl * D + j];
            cand[pl * D + k] = acc;
        }
        __syncthreads();
        for (int i = tid; i < nstart * D; i += NT) {
            const int pl = starts[i / D], k = i % D;
            double n2 = 0.;
            for (int j = 0; j < D; ++j) n2 += cand[pl * D + j] * cand[pl * D + j];
            const double d = cand[pl * D + k] * (1. / sqrt(n2)), u = wu[pl * D + k];
            dir[pl * D + k] = d;
            double lo = -INFINITY, hi = INFINITY;
            if (d != 0.) {
                if (wrapped_s[k]) {
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
        WALK_STAMP(0);
        for (int ai = tid; ai < nact; ai += NT) {
            const int pl = act[ai];
            if (state[pl] == 0) {
                double lo = -INFINITY, hi = INFINITY;
                for (int k = 0; k < D; ++k) { lo = fmax(lo, lo_s[pl * D + k]); hi = fmin(hi, hi_s[pl * D + k]); }
                tmin[pl] = lo; tmax[pl] = hi;
                round_of[pl] = 0;
                state[pl] = 1;
            }
            const int g = rrow[srow[pl]];
            const unsigned long long wid = (unsigned long long)(w.walker_base + (w.walker_id ? (long long)w.walker_id[g] : (long long)g));
            const unsigned long long ctr = (wid << 32) | ((unsigned long long)step_of[pl] << 14) | (unsigned)(8192 + round_of[pl]);
            double lo = tmin[pl], hi = tmax[pl];
            const int first = first_of[pl], S = nsp_of[pl];
            for (int j = 0; j < S; ++j) {
                const double t = lo + (hi - lo) * uniform01(w.seed, ctr + (unsigned)j);
                slot_t[first + j] = t; slot_pl[first + j] = pl;
                if (t < 0.) lo = t; else hi = t;
            }
        }
        __syncthreads();
        for (int i = tid; i < nslots * D; i += NT) {
            const int sl = i / D, k = i - sl * D, pl = slot_pl[sl];
            double c = wu[pl * D + k] + slot_t[sl] * dir[pl * D + k];
            if (wrapped_s[k]) c -= floor(c);
            cand[i] = fmin(fmax(c, 0.), one_below);
        }
        __syncthreads();
        WALK_STAMP(1);
#ifdef RVLL_WALK_TRACE
        loglike_tile<PREC, FAT ? kFusedFull : kFusedSlim, false, NT, DYN>(a, smem, w0, nslots, cand, LogdetPre{}, tph);
#else
        loglike_tile<PREC, FAT ? kFusedFull : kFusedSlim, false, NT, DYN>(a, smem, w0, nslots, cand);
#endif
        __builtin_amdgcn_s_setprio(3);
        __syncthreads();
        WALK_STAMP(2);
        for (int ai = tid; ai < nact; ai += NT) {
            const int pl = act[ai];
            const int first = first_of[pl], S = nsp_of[pl];
            int used = 0;
            acc_slot[pl] = -1;
            for (int j = 0; j < S; ++j) {
                int fl;
                const double cl = tile_point_result(a, L, first + j, fl);
                used = j + 1;
                if (!FAT && (fl & kFlagDeferred)) {
                    // leave at the start of this move; the full-solver pass retraces it from its first candidate, so none
                    // of this move's candidates count here (round_of: the ones of earlier iterations, counted then)
                    state[pl] = 3;
                    used = -(round_of[pl] - j);
                    break;
                }
                if (cl > w.lstar) {
                    state[pl] = 2; wl[pl] = cl; acc_slot[pl] = first + j; acc_g[pl] = rrow[srow[pl]];
                    if (w.wflag) w.wflag[rrow[srow[pl]]] = (fl & RVLL_FLAG_WANDERED) ? 1 : 0;
                    break;
                }
                const double t = slot_t[first + j];
                if (t < 0.) tmin[pl] = t; else tmax[pl] = t;
                if (++round_of[pl] >= w.max_rounds) { state[pl] = 0; step_of[pl] += 1; break; }
            }
            used_of[pl] = used;
        }
        __syncthreads();
        for (int i = tid; i < nact * D; i += NT) {
            const int ai = i / D, k = i - ai * D, pl = act[ai];
            const int sl = acc_slot[pl];
            if (sl < 0) continue;
            double c = wu[pl * D + k] + slot_t[sl] * dir[pl * D + k];      // the accepted candidate, as it was made above
            if (wrapped_s[k]) c -= floor(c);
            wu[pl * D + k] = fmin(fmax(c, 0.), one_below);
            w.theta[(long long)acc_g[pl] * D + k] = L.theta_s[sl * D + k];
        }
        if (tid >= NT - kWave) {
            if (lane == 0) slots += (unsigned long long)nslots;
            bookkeep(act_next);
        }
        __syncthreads();
        WALK_STAMP(3);
    }
    // rows go home
    for (int i = tid; i < R * D; i += NT) { const int g = rrow[i / D]; if (g >= 0) w.u[(long long)g * D + i % D] = ru[i]; }
    for (int r = tid; r < R; r += NT) {
        const int g = rrow[r];
        if (g < 0) continue;
        w.logl[g] = rl[r];
        if (w.steps_done) w.steps_done[g] = rstep[r];
        if (w.cost) w.cost[g] = rcost[r];
    }
    if (tid >= NT - kWave && calls) atomicAdd(w.ncalls, (unsigned long long)calls);   // (two's complement: a negative share adds up right)
    if (tid >= NT - kWave && slots && w.nslots) atomicAdd(w.nslots, slots);
#ifdef RVLL_WALK_TRACE
    if (tid == 0 && w.nslots) {
        for (int k = 0; k < 4; ++k) atomicAdd(w.nslots + 1 + k, ph[k]);
        atomicAdd(w.nslots + 5, 1ull);
        atomicMax(w.nslots + 6, ph[0] + ph[1] + ph[2] + ph[3]);
        for (int k = 0; k < 4; ++k) atomicAdd(w.nslots + 7 + k, tph[k]);
    }
#endif
#undef WALK_STAMP
}

}  // namespace

size_t walk_lds_bytes(const LoglikeArgs& a)
{
    const size_t base = (loglike_lds_bytes(a) + 15) & ~(size_t)15;
    return base + sizeof(double) * ((size_t)2 * a.PB * a.D + 4 * a.PB + (a.D <= kWalkCholLds ? a.D * a.D : 0)) +
           sizeof(int) * (16 * a.PB + 4 + a.D) + 16;
}

size_t walk_rows_lds_bytes(const LoglikeArgs& a, int R)
{
    const size_t base = (loglike_lds_bytes(a) + 15) & ~(size_t)15;
    return base + sizeof(double) * ((size_t)2 * a.PB * a.D + 4 * a.PB + (a.D <= kWalkCholLds ? a.D * a.D : 0) + (size_t)R * a.D + R) +
           sizeof(int) * (16 * a.PB + 6 + a.D + 4 * (size_t)R) + 16;
}

// workgroups of the walk kernel the given number of compute units holds at once (0: the query failed)
long long slice_walk_resident_blocks(const LoglikeArgs& a, bool fat, int cus)
{
    const size_t lds = walk_lds_bytes(a);
    int occ = 0;
    hipError_t e;
#define RVLL_WALK_OCC(PREC)                                                                                          \
    e = fat ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (slice_walk_kernel<PREC, true>), kThreads, lds)     \
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (slice_walk_kernel<PREC, false>), kThreads, lds)
    switch (a.precision) {
    case RVLL_PREC_MIXED: RVLL_WALK_OCC(RVLL_PREC_MIXED); break;
    case RVLL_PREC_FP32:  RVLL_WALK_OCC(RVLL_PREC_FP32); break;
    default:              RVLL_WALK_OCC(RVLL_PREC_FP64); break;
    }
#undef RVLL_WALK_OCC
    return e == hipSuccess ? (long long)std::max(1, occ) * cus : 0;
}

hipError_t launch_slice_walk(const LoglikeArgs& a, const WalkArgs& w, bool fat, int max_cus, hipStream_t stream)
{
    if (w.K <= 0 || w.nsteps <= 0) return hipSuccess;
    if (!a.cube || !a.theta_out || !a.priors || !a.flags || a.PB * a.D > 4 * kThreads || w.nsteps >= (1 << 18) ||
        w.max_rounds < 1 || w.max_rounds > 4096 || a.D > 4096 || (!fat && !w.steps_done) || w.spec_max < 1 ||
        3LL * a.PB * a.D > a.CH)                            // the candidates and chord limits borrow the tile's window
        return hipErrorInvalidValue;
    const size_t lds = walk_lds_bytes(a);
    if (lds > 64 * 1024 || !w.queue) return hipErrorInvalidValue;
    // max_cus > 0: no more workgroups than the chip holds at once; their slots draw the remaining rows from the queue
    long long nblocks = (w.K + a.PB - 1) / a.PB;
    const dim3 block(kThreads);
#define RVLL_WALK_ONE(KERNEL)                                                                                        \
    do {                                                                                                             \
        if (max_cus > 0) {                                                                                           \
            int occ = 0;                                                                                             \
            const hipError_t e_ = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, KERNEL, kThreads, lds);        \
            if (e_ != hipSuccess) return e_;                                                                         \
            nblocks = std::min(nblocks, (long long)std::max(1, occ) * max_cus);                                     \
        }                                                                                                            \
        hipLaunchKernelGGL(KERNEL, dim3((unsigned)nblocks), block, lds, stream, a, w);                               \
    } while (0)
#define RVLL_WALK(PREC)                                                                                              \
    if (fat) RVLL_WALK_ONE((slice_walk_kernel<PREC, true>));                                                        \
    else     RVLL_WALK_ONE((slice_walk_kernel<PREC, false>))
    switch (a.precision) {
    case RVLL_PREC_MIXED: RVLL_WALK(RVLL_PREC_MIXED); break;
    case RVLL_PREC_FP32:  RVLL_WALK(RVLL_PREC_FP32); break;
    default:
#ifndef RVLL_AB_NO_NP                  // (measurement builds only: without the three-planet instantiation)
        if (!fat && a.Np == 3) { RVLL_WALK_ONE((slice_walk_kernel<RVLL_PREC_FP64, false, 3>)); break; }
#endif
        RVLL_WALK(RVLL_PREC_FP64);
        break;
    }
#undef RVLL_WALK
#undef RVLL_WALK_ONE
    return hipGetLastError();
}

hipError_t launch_slice_walk_rows(const LoglikeArgs& a, const WalkArgs& w, bool fat, int nblocks, int nt, hipStream_t stream)
{
    if (w.K <= 0 || w.nsteps <= 0) return hipSuccess;
    const bool wide = nt != kThreads;
    if ((nt != kThreads && nt != 512 && nt != kCuThreads) || (wide && fat) ||
        !a.cube || !a.theta_out || !a.priors || !a.flags || a.PB * a.D > 4 * nt || a.PB > kWave || w.nsteps >= (1 << 18) ||
        w.max_rounds < 1 || w.max_rounds > 4096 || a.D > 4096 || (!fat && !w.steps_done) || w.spec_max < 1 ||
        nblocks < 1 || w.rows_per_wg < 1 || (long long)nblocks * w.rows_per_wg < w.K || 3LL * a.PB * a.D > a.CH ||
        (wide && (long long)a.CH < (long long)a.PB * a.Ne))
        return hipErrorInvalidValue;
    const size_t lds = walk_rows_lds_bytes(a, w.rows_per_wg);
    const size_t budget = nt == kCuThreads ? kCuLdsBudget : nt == 512 ? kCuLdsBudget / 2 : (size_t)64 * 1024;
    if (lds > budget) return hipErrorInvalidValue;
    const dim3 grid((unsigned)nblocks), block(nt);
    if (wide) {
        static bool attr_set_dev[64] = {};                      // raise the dynamic-LDS limit of every instance once per device
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!attr_set_dev[dev & 63]) {
            hipError_t e = hipSuccess;
#define RVLL_WALK_ATTR(PREC, NTV, LIM) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(slice_walk_rows_kernel<PREC, false, NTV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LIM))
            RVLL_WALK_ATTR(RVLL_PREC_FP64, kCuThreads, kCuLdsBudget); RVLL_WALK_ATTR(RVLL_PREC_MIXED, kCuThreads, kCuLdsBudget); RVLL_WALK_ATTR(RVLL_PREC_FP32, kCuThreads, kCuLdsBudget);
            RVLL_WALK_ATTR(RVLL_PREC_FP64, 512, kCuLdsBudget / 2); RVLL_WALK_ATTR(RVLL_PREC_MIXED, 512, kCuLdsBudget / 2); RVLL_WALK_ATTR(RVLL_PREC_FP32, 512, kCuLdsBudget / 2);
#undef RVLL_WALK_ATTR
            if (e != hipSuccess) return e;
            attr_set_dev[dev & 63] = true;
        }
    }
#define RVLL_WALK(PREC)                                                                                              \
    do {                                                                                                             \
        if (nt == kCuThreads)  hipLaunchKernelGGL((slice_walk_rows_kernel<PREC, false, kCuThreads>), grid, block, lds, stream, a, w); \
        else if (nt == 512)    hipLaunchKernelGGL((slice_walk_rows_kernel<PREC, false, 512>), grid, block, lds, stream, a, w);        \
        else if (fat)          hipLaunchKernelGGL((slice_walk_rows_kernel<PREC, true, kThreads>), grid, block, lds, stream, a, w);    \
        else                   hipLaunchKernelGGL((slice_walk_rows_kernel<PREC, false, kThreads>), grid, block, lds, stream, a, w);   \
    } while (0)
    switch (a.precision) {
    case RVLL_PREC_MIXED: RVLL_WALK(RVLL_PREC_MIXED); break;
    case RVLL_PREC_FP32:  RVLL_WALK(RVLL_PREC_FP32); break;
    default:              RVLL_WALK(RVLL_PREC_FP64); break;
    }
#undef RVLL_WALK
    return hipGetLastError();
}

}  // namespace rvll
