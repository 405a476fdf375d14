// rvll_walk_host.hip — host side of the sampler's proposal step: the device-resident slice-sampling walk
// in its forms (rvll_slice_walk) and the live set of nested sampling kept in HBM (rvll_live_*).  Entry points of include/rvll.h;
// the kernels are in rvll_walk.hip, rvll_rounds.hip and rvll_live.hip.
#include <chrono>
#include <thread>
#include "rvll_host.h"

using rvll::report_error;
using namespace rvll::host;

namespace {

// device buffers of the walk for K rows (grown on demand)
int walk_reserve(rvll_handle* h, int64_t K)
{
    const size_t D = (size_t)h->L.ndim;
    int rc = rvll_dev_reserve(h, K + rvll::kMaxPointsPerBlock);   // scratch rows (one tile per workgroup): d_theta, log-L / flags of lane 0
    if (rc) return rc;
    rc = sync_other_lanes(h);
    if (rc) return rc;
    if (K > h->walk_cap || !h->d_walk_chol) {
        HIP_TRY(hipStreamSynchronize(h->compute));
        dev_free(h->d_walk_u); dev_free(h->d_walk_theta); dev_free(h->d_walk_logl);
        dev_free(h->d_walk_steps); dev_free(h->d_walk_wid); dev_free(h->d_walk_start);
        dev_free(h->d_walk_cost); dev_free(h->d_walk_order); dev_free(h->d_walk_wflag);
        h->walk_cap = 0;
        const size_t cap = (size_t)std::max<long long>(K, 1024);
        HIP_TRY(hipMalloc(&h->d_walk_u, sizeof(double) * D * cap));
        HIP_TRY(hipMalloc(&h->d_walk_theta, sizeof(double) * D * cap));
        HIP_TRY(hipMalloc(&h->d_walk_logl, sizeof(double) * cap));
        HIP_TRY(hipMalloc(&h->d_walk_steps, sizeof(int32_t) * cap));
        HIP_TRY(hipMalloc(&h->d_walk_wid, sizeof(int32_t) * cap));
        HIP_TRY(hipMalloc(&h->d_walk_start, sizeof(int32_t) * cap));
        HIP_TRY(hipMalloc(&h->d_walk_cost, sizeof(int32_t) * cap));
        HIP_TRY(hipMalloc(&h->d_walk_order, sizeof(int32_t) * cap));
        HIP_TRY(hipMalloc(&h->d_walk_wflag, sizeof(int32_t) * cap));
        if (!h->d_walk_chol) {
            HIP_TRY(hipMalloc(&h->d_walk_chol, sizeof(double) * D * D));
            HIP_TRY(hipMalloc(&h->d_walk_wrapped, sizeof(int32_t) * D));
            HIP_TRY(hipMalloc(&h->d_walk_ncalls, kWalkWords * sizeof(unsigned long long)));   // calls used, tile slots evaluated, diagnostic bins
        }
        h->walk_cap = (long long)cap;
    }
    return RVLL_OK;
}

// ---- the walk as rounds of launches (rvll_rounds.h, rvll_kernels.hip) ----------------------------------------------------
// Which walks take it: RVLL_WALK_ROUNDS = 0 never / 1 whenever the slim prior stage applies.  By default the walks it was
// measured to win (profiles/r04_rounds_sizes.txt, cfg3, nested sampling end to end, calls/s inside the walk against the
// single-kernel form): 8192 walkers 1.23 against 1.05e8, 16384: 1.72 against 1.62e8 — and not the ones it loses: 4096 walkers
// and fewer (a round's fixed ~35 us against a workgroup iteration of the single kernel: 5.2 against 5.8e7), 32768 (1.86 against
// 1.90e8: the single kernel's queue has enough rows per slot there).  None of the switches that select a single-kernel form
// may be set (RVLL_WALK_QUEUE / _PARTS / _ROWS / _FAT: the tests and the measurements of those forms).
bool rounds_wanted(int64_t K)
{
    if (const char* e = getenv("RVLL_WALK_ROUNDS")) return atoi(e) != 0;
    if (getenv("RVLL_WALK_QUEUE") || getenv("RVLL_WALK_PARTS") || getenv("RVLL_WALK_ROWS") || getenv("RVLL_WALK_FAT")) return false;
    return K >= 6144 && K <= 24576;
}

// The K rows of d_walk_u / d_walk_theta / d_walk_logl (chol, wrapped uploaded) walked in rounds: G groups of rows; a group's
// round = its step, then the batch log-L tiles over its candidates, the groups on streams of their own (rvll_kernels.hip,
// rounds_step_kernel / rounds_tiles_kernel).  The host only keeps the queue a few rounds
// deep: the tiles publish (round, walkers listed) to a word per group in mapped pinned memory; a group is done when a round
// listed nobody (what is already queued behind it finds nothing to do).  Leaves the end points in the walk buffers, moves
// completed in d_walk_steps (a walker the slim prior stage deferred: < nsteps), and synchronises the stream.
// *calls: likelihood calls consumed; *slots: candidates evaluated.
// Returns RVLL_E_UNSUPPORTED without having launched anything if the step's state does not fit (huge D).
int walk_rounds(rvll_handle* h, int64_t K, double lstar, int32_t nsteps, int32_t max_rounds, uint64_t seed, int64_t walker_base,
                long long* calls, long long* slots)
{
    using namespace std::chrono;
    const int D = h->L.ndim;
    int spec = h->walk_spec_rounds;
    if (const char* e = getenv("RVLL_WALK_SPEC")) spec = atoi(e);       // measurement switch (1: no candidates ahead)
    spec = std::max(1, std::min(spec, rvll::kMaxPointsPerBlock));
    int G = 3;                                                          // (measured at cfg3: 2: 1.64, 3: 1.72, 4: 1.67e8 calls/s; profiles/r04_rounds_sweep.txt)
    if (const char* e = getenv("RVLL_ROUNDS_GROUPS")) G = atoi(e);
    G = (int)std::max<long long>(1, std::min<long long>(std::min(G, kMaxLanes), K));
    // Slots a round may hold before walkers stop getting candidates AHEAD: while a group lists fewer walkers than this, the free
    // slots go to its walkers' next candidates (consumed in order: the results do not change).  A round costs ~35 us whatever it
    // holds (a step, a launch, a tile's prologue and last wave round) plus ~3 ns a slot, and a walk is a chain of such rounds:
    // below ~6000 slots at 200 epochs a round is cheaper filled with candidates of which two in three go unused than run again
    // (measured, 16384 walkers: 1300: 1.50, 2600: 1.61, 4000: 1.68, 6000: 1.72, 8000: 1.71e8 calls/s).  Scaled by the epochs.
    long long c_free = std::max<long long>(256, (long long)h->n_cu * 4608 / std::max(1, h->Ne));
    if (const char* e = getenv("RVLL_ROUNDS_FREE")) c_free = std::max(1, atoi(e));
    // How the rounds are issued: a stream per group; a group's round = a step launch, then a tiles launch (256-thread form, or
    // CU-wide: RVLL_ROUNDS_FORM=cu, a measurement switch).  (A third structure — one stream, every launch one group's tiles next
    // to another group's step in one kernel — was measured slower, 1.2e8 against 1.7e8 calls/s, its kernel spilled, and it is
    // gone: profiles/r04_rounds_sweep.txt.)
    const char* fenv = getenv("RVLL_ROUNDS_FORM");
    const bool cu_form = fenv && !strcmp(fenv, "cu");
    rvll::LoglikeArgs a;
    long long per = (K + G - 1) / G;
    if (const char* e = getenv("RVLL_ROUNDS_PER")) per = std::max<long long>(1, std::min<long long>(K, atoll(e)));   // measurement switch: rows a group (the last group takes the rest: unequal groups)
    int rc = build_args(h, nullptr, nullptr, nullptr, std::max(per, c_free), &a);
    if (rc) return rc;
    // the step's walkers per workgroup: within what leaves four workgroups a compute unit when it shares its launches (and with
    // them the size of the dynamic LDS) with the tiles, up to a wave's lanes otherwise
    const size_t lds_budget = (size_t)60 * 1024;
    int W = rvll::rounds_walkers_per_block(D, spec, lds_budget);
    if (W < 1) return RVLL_E_UNSUPPORTED;
    if (const char* e = getenv("RVLL_ROUNDS_W")) W = std::max(1, std::min(W, atoi(e)));      // measurement switch
    per = (per + W - 1) / W * W;
    G = (int)((K + per - 1) / per);
    const long long C = std::max(per, c_free), nblk = per / W;
    c_free = std::min(c_free, C);
    if (C >= (1LL << 30) || per * spec >= (1LL << 31)) return RVLL_E_UNSUPPORTED;
    auto window = [&](int pb) { return (std::min(h->chunk_items, std::max(rvll::kThreads, pb * h->Ne)) + 1) & ~1; };
    if (cu_form) {
        // one CU-wide tile per compute unit: every contribution of the tile resident in LDS
        int pb = (int)std::min<long long>(rvll::kCuMaxPoints, (C + h->n_cu - 1) / h->n_cu);
        rvll::LoglikeArgs b = a;
        auto fits = [&](int q) { b.PB = q; b.CH = (q * h->Ne + 1) & ~1; return rvll::loglike_lds_bytes(b) <= rvll::kCuLdsBudget; };
        while (pb > 1 && !fits(pb)) --pb;
        if (!fits(pb)) return RVLL_E_UNSUPPORTED;
        a = b;
    } else {
        // 256-thread tiles sized to the wave slots the chip has left beside one group's step workgroups (but never smaller
        // than the cost model's choice): a tile that waits for a step's slot ends its launch a step's latency late
        const size_t lds0 = rvll::loglike_lds_bytes(a);
        const int occ = rvll::rounds_blocks_per_cu(lds0);
        const long long room = (long long)std::max(1, occ) * h->n_cu - (G > 1 ? nblk : 0);
        const long long want = room > 0 ? (C + room - 1) / room : a.PB;
        if (want > a.PB && want <= rvll::kMaxPointsPerBlock) {
            rvll::LoglikeArgs b = a;
            b.PB = (int)want; b.CH = window((int)want);
            if (rvll::loglike_lds_bytes(b) <= lds_budget) a = b;
        }
        if (const char* e = getenv("RVLL_ROUNDS_PB")) {          // measurement switch: points per tile
            const int pb = std::max(1, std::min(atoi(e), rvll::kMaxPointsPerBlock));
            rvll::LoglikeArgs b = a;
            b.PB = pb; b.CH = window(pb);
            if (rvll::loglike_lds_bytes(b) <= 60 * 1024) a = b;
        }
    }
    const size_t lds = rvll::loglike_lds_bytes(a);
    const int tiles = (int)((C + a.PB - 1) / a.PB);
    a.cr_redo = getenv("RVLL_WALK_CR") ? atoi(getenv("RVLL_WALK_CR")) : 0;                 // (the walk's tiles leave the exact redo of wandering solves to walk_core: see there)
    if (getenv("RVLL_WALK_GEOM_DUMP"))
        fprintf(stderr, "[rounds] K=%lld G=%d %s %s per=%lld C=%lld c_free=%lld W=%d step blocks=%lld PB=%d tiles=%d lds=%zu spec=%d\n", (long long)K, G,
                "streams", cu_form ? "cu" : "tile", per, C, c_free, W, nblk, a.PB, tiles, lds, spec);
    // arena: per group  doubles | 64-bit words | walker words (int4) | ints
    const size_t n_dbl = 2 * (size_t)per * D + 2 * (size_t)per + 2 * (size_t)C * D + 2 * (size_t)per * spec + (size_t)C;
    const size_t n_ll = 2 * (size_t)nblk + rvll::kRoundsRing;
    const size_t n_int = 4 * (size_t)per + 2 * (size_t)per * spec + 2 * (size_t)C;
    auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t g_bytes = up16(8 * n_dbl) + up16(8 * n_ll) + up16(4 * n_int);
    hipStream_t st = h->compute;
    if ((size_t)G * g_bytes > h->rounds_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        dev_free(h->d_rounds);
        h->rounds_bytes = 0;
        HIP_TRY(hipMalloc(&h->d_rounds, (size_t)G * g_bytes));
        h->rounds_bytes = (size_t)G * g_bytes;
    }
    if (!h->pin_rounds) {
        void *p = nullptr, *pd = nullptr;
        HIP_TRY(hipHostMalloc(&p, sizeof(unsigned long long) * kMaxLanes, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer(&pd, p, 0));
        h->pin_rounds = static_cast<unsigned long long*>(p);
        h->pin_rounds_dev = static_cast<unsigned long long*>(pd);
    }
    // the directions of every move of every walker, ahead of the rounds (rvll_rounds.h, rounds_dirs)
    {
        const size_t need = (size_t)K * (size_t)nsteps * (size_t)D;
        if (need > h->walk_dirs_cap) {
            HIP_TRY(hipStreamSynchronize(h->compute));
            dev_free(h->d_walk_dirs);
            h->walk_dirs_cap = 0;
            HIP_TRY(hipMalloc(&h->d_walk_dirs, sizeof(double) * need));
            h->walk_dirs_cap = need;
        }
        const rvll::RoundsDirs dg{h->d_walk_dirs, h->d_walk_chol, (long long)K, (unsigned long long)walker_base, (unsigned long long)seed, D, nsteps};
        HIP_TRY(rvll::launch_rounds_dirs(dg, 16 * h->n_cu, h->compute));
    }
    std::vector<rvll::RoundsArgs> ga((size_t)G);
    std::vector<rvll::LoglikeArgs> la((size_t)G, a);
    std::vector<rvll::RoundsTiles> ta((size_t)G);
    for (int g = 0; g < G; ++g) {
        char* base = static_cast<char*>(h->d_rounds) + (size_t)g * g_bytes;
        double* d = reinterpret_cast<double*>(base);
        long long* ll = reinterpret_cast<long long*>(base + up16(8 * n_dbl));
        int32_t* in = reinterpret_cast<int32_t*>(base + up16(8 * n_dbl) + up16(8 * n_ll));
        const long long row0 = (long long)g * per, Kg = std::min<long long>(per, K - row0);
        rvll::RoundsArgs& r = ga[(size_t)g];
        r.u = h->d_walk_u + (size_t)row0 * D;  r.theta = h->d_walk_theta + (size_t)row0 * D;  r.logl = h->d_walk_logl + row0;
        r.step = h->d_walk_steps + row0;  r.wflag = h->d_walk_wflag + row0;
        r.dir = d;                       d += (size_t)per * D;
        r.dirnext = d;                   d += (size_t)per * D;
        r.dirs = h->d_walk_dirs + (size_t)row0 * nsteps * D;
        r.tmin = d;                      d += per;
        r.tmax = d;                      d += per;
        r.theta_c[0] = d;                d += (size_t)C * D;
        r.theta_c[1] = d;                d += (size_t)C * D;
        r.wt = d;                        d += (size_t)per * spec;
        double* wres_logl = d;           d += (size_t)per * spec;
        double* res_logl = d;
        r.wres_logl = wres_logl;
        r.calls_part = ll;  r.slots_part = reinterpret_cast<unsigned long long*>(ll + nblk);
        r.ring = reinterpret_cast<int32_t*>(ll + 2 * nblk);            // (8-byte aligned pairs: one 64-bit atomic a workgroup)
        r.ws = in;                       in += 4 * per;
        int32_t* wres_flags = in;        in += per * spec;
        r.wdef = in;                     in += per * spec;
        int32_t* res_flags = in;         in += C;
        r.owner = in;
        r.wres_flags = wres_flags;
        r.priors = h->d_priors;  r.heavy_dims = h->d_heavy;  r.n_heavy = h->n_heavy;  r.light_dims = h->d_heavy + h->n_heavy;
        r.inplace = h->priors_rowwise ? 0 : 1;  r.slim_umax = h->slim_umax;
        r.K = Kg;  r.wid0 = (unsigned long long)(walker_base + row0);
        r.C = (int)C;  r.c_free = (int)c_free;
        r.wrapped = h->d_walk_wrapped;
        r.D = D;  r.W = W;  r.nsteps = nsteps;  r.max_rounds = max_rounds;  r.spec_max = spec;
        r.seed = seed;  r.lstar = lstar;
        rvll::LoglikeArgs& l = la[(size_t)g];
        l.theta = r.theta_c[0];  l.logL = res_logl;  l.flags = res_flags;  l.B = C;      // the tiles: theta -> log-L
        ta[(size_t)g] = rvll::RoundsTiles{nullptr, h->pin_rounds_dev + g, r.owner, wres_logl, wres_flags};
        r.stamps = nullptr;  r.stamp_rounds = 0;
        HIP_TRY(hipMemsetAsync(ll, 0, up16(8 * n_ll), st));          // the partial sums and the ring
        h->pin_rounds[g] = 0;
    }
    // diagnostic (RVLL_ROUNDS_STAMPS=n): time stamps of group 0's step workgroups over its first n rounds, dumped to stderr
    unsigned long long* d_stamps = nullptr;
    int stamp_rounds = 0;
    if (const char* e = getenv("RVLL_ROUNDS_STAMPS")) {
        stamp_rounds = std::max(0, std::min(atoi(e), 4096));
        if (stamp_rounds) {
            const size_t nb = sizeof(unsigned long long) * 8 * (size_t)stamp_rounds * (size_t)nblk;
            HIP_TRY(hipMalloc(&d_stamps, nb));
            HIP_TRY(hipMemsetAsync(d_stamps, 0, nb, h->compute));
            ga[0].stamps = d_stamps;  ga[0].stamp_rounds = stamp_rounds;
        }
    }
    int depth = 4;                                                      // rounds queued beyond the last one seen starting
    if (const char* e = getenv("RVLL_ROUNDS_DEPTH")) depth = std::max(1, atoi(e));
    // Every round consumes a candidate of every listed walker, so a group has listed nobody by round nsteps * max_rounds + 1 — and the
    // host may have queued `depth` rounds beyond the last one it has seen start (without that allowance a walk with max_rounds = 1
    // was refused while its last rounds were still in the queue: found by scripts/walk_soak.py).
    const long long r_max = (long long)nsteps * max_rounds + 2 + depth;
    // per group: rounds whose step / whose tiles have been launched (tiles <= step <= tiles + 1: the next thing a group needs
    // is its step when they are equal, its tiles otherwise)
    std::vector<long long> n_step((size_t)G, 0), n_tile((size_t)G, 0);
    std::vector<char> done((size_t)G, 0);
    int ndone = 0, status = RVLL_OK;
    auto last_progress = steady_clock::now();
    const auto t_walk0 = steady_clock::now();
    long long launch_ns = 0, n_launch = 0;
    unsigned long long seen_sum = 0;
    const bool listed_dump = getenv("RVLL_ROUNDS_LISTED_DUMP") != nullptr;      // diagnostics: group 0's walkers listed, round by round (as the host saw them)
    std::vector<unsigned long long> listed_seen;
    // The groups' streams: the handle's lanes (RVLL_ROUNDS_PRIO=1, a measurement switch: streams of DIFFERENT priorities — it did
    // not keep the groups out of lock step: 1.45 against 1.51e8 calls/s)
    std::vector<hipStream_t> gs((size_t)G, h->compute);
    const char* cenv = getenv("RVLL_ROUNDS_CHAIN");
    const bool chain = G > 1 && cenv && atoi(cenv) != 0;       // measured: 1.57 against 1.71e8 calls/s unchained
    if (chain)
        for (int g = 0; g < G; ++g)
            if (!h->ev_chain[g]) HIP_TRY(hipEventCreateWithFlags(&h->ev_chain[g], hipEventDisableTiming));
    if (G > 1) {
        const char* penv = getenv("RVLL_ROUNDS_PRIO");
        const bool prio = penv && atoi(penv) != 0;
        int lo_p = 0, hi_p = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);      // (numerically: hi_p <= lo_p)
        for (int g = 0; g < G; ++g) {
            if (!prio) { gs[(size_t)g] = h->lanes[g]; continue; }
            if (!h->rounds_streams[g]) {
                const int p = std::max(hi_p, std::min(lo_p, hi_p + g));
                HIP_TRY(hipStreamCreateWithPriority(&h->rounds_streams[g], hipStreamNonBlocking, p));
            }
            gs[(size_t)g] = h->rounds_streams[g];
        }
        // they start behind what lane 0 holds (uploads, the live step's gathers, the zeroing above)
        if (!h->ev_rounds) HIP_TRY(hipEventCreateWithFlags(&h->ev_rounds, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(h->ev_rounds, h->compute));
        for (int g = 0; g < G; ++g) if (gs[(size_t)g] != h->compute) HIP_TRY(hipStreamWaitEvent(gs[(size_t)g], h->ev_rounds, 0));
    }
    auto tiles_of = [&](int g, int r, hipStream_t s) -> hipError_t {
        ta[(size_t)g].ring_entry = ga[(size_t)g].ring + 2 * (r % rvll::kRoundsRing);
        la[(size_t)g].theta = ga[(size_t)g].theta_c[r & 1];
        return cu_form ? rvll::launch_rounds_cu(la[(size_t)g], ta[(size_t)g], tiles, r, s)
                       : rvll::launch_rounds_tiles(la[(size_t)g], ta[(size_t)g], tiles, r, s);
    };
    while (ndone < G && status == RVLL_OK) {
        unsigned long long sum = 0;
        bool any = false;
        if (chain) {
            // (RVLL_ROUNDS_CHAIN=1, a measurement switch.)  Left to themselves the groups fall into lock step: all step together
            // (the chip idle for a step's 20 us), then all run their tiles together.  Here the steps are CHAINED: group g's step
            // of a round waits (an event, on the device) for group g - 1's step of that round — it starts when g - 1's tiles do
            // and runs beside them — and the host issues a round for all groups at once.  Slower: the groups then all wait for
            // the slowest of them, and every link adds an event's latency.
            bool ready = true;
            for (int g = 0; g < G; ++g) {
                if (done[(size_t)g]) continue;
                const unsigned long long p = __atomic_load_n(&h->pin_rounds[g], __ATOMIC_ACQUIRE);
                const long long pub = (long long)(p >> 32);
                sum += p;
                if (pub > 0 && (unsigned)p == 0u) { done[(size_t)g] = 1; ++ndone; continue; }
                if (n_tile[(size_t)g] - pub >= depth) ready = false;
            }
            if (ready && ndone < G) {
                int prev = -1;
                const auto tl0 = steady_clock::now();
                for (int g = 0; g < G && status == RVLL_OK; ++g) {
                    if (done[(size_t)g]) continue;
                    if (n_tile[(size_t)g] >= r_max) { status = report_error(RVLL_E_HIP, "rounds walk: group %d did not finish in %lld rounds", g, r_max); break; }
                    const int r = (int)n_tile[(size_t)g];
                    hipError_t e = hipSuccess;
                    if (prev >= 0) e = hipStreamWaitEvent(gs[(size_t)g], h->ev_chain[prev], 0);
                    if (e == hipSuccess) e = rvll::launch_rounds_step(ga[(size_t)g], r, gs[(size_t)g]);
                    if (e == hipSuccess) e = hipEventRecord(h->ev_chain[g], gs[(size_t)g]);
                    if (e == hipSuccess) e = tiles_of(g, r, gs[(size_t)g]);
                    if (e != hipSuccess) { status = report_error(RVLL_E_HIP, "rounds walk launch failed: %s", hipGetErrorString(e)); break; }
                    n_step[(size_t)g] += 1;
                    n_tile[(size_t)g] += 1;
                    n_launch += 2;
                    prev = g;
                    any = true;
                }
                launch_ns += duration_cast<nanoseconds>(steady_clock::now() - tl0).count();
            }
        } else {
            for (int g = 0; g < G && status == RVLL_OK; ++g) {
                if (done[(size_t)g]) continue;
                const unsigned long long p = __atomic_load_n(&h->pin_rounds[g], __ATOMIC_ACQUIRE);
                const long long pub = (long long)(p >> 32);
                sum += p;
                if (listed_dump && g == 0 && (listed_seen.empty() || listed_seen.back() != p)) listed_seen.push_back(p);
                if (pub > 0 && (unsigned)p == 0u) { done[(size_t)g] = 1; ++ndone; continue; }
                if (n_tile[(size_t)g] - pub >= depth) continue;
                if (n_tile[(size_t)g] >= r_max) { status = report_error(RVLL_E_HIP, "rounds walk: group %d did not finish in %lld rounds", g, r_max); break; }
                const int r = (int)n_tile[(size_t)g];
                const auto tl0 = steady_clock::now();
                hipError_t e = rvll::launch_rounds_step(ga[(size_t)g], r, gs[(size_t)g]);
                if (e == hipSuccess) e = tiles_of(g, r, gs[(size_t)g]);
                launch_ns += duration_cast<nanoseconds>(steady_clock::now() - tl0).count();  n_launch += 2;
                if (e != hipSuccess) { status = report_error(RVLL_E_HIP, "rounds walk launch failed: %s", hipGetErrorString(e)); break; }
                n_step[(size_t)g] += 1;
                n_tile[(size_t)g] += 1;
                any = true;
            }
        }
        if (any || sum != seen_sum) { last_progress = steady_clock::now(); seen_sum = sum; continue; }
        if (ndone < G && steady_clock::now() - last_progress > seconds(20)) {
            status = report_error(RVLL_E_HIP, "rounds walk made no progress for 20 s");
            break;
        }
        std::this_thread::yield();
    }
    for (int g = 0; g < G; ++g) {
        const hipError_t e = hipStreamSynchronize(gs[(size_t)g]);
        if (e != hipSuccess && status == RVLL_OK) status = report_error(RVLL_E_HIP, "rounds walk: %s", hipGetErrorString(e));
    }
    if (listed_dump && !listed_seen.empty()) {
        long long below[4] = {0, 0, 0, 0};
        const long long full = ga[0].K;
        for (unsigned long long p : listed_seen) {
            const long long l = (long long)(unsigned)p;
            if (l * 2 < full) ++below[0];
            if (l * 4 < full) ++below[1];
            if (l * 10 < full) ++below[2];
            if (l * 50 < full) ++below[3];
        }
        fprintf(stderr, "[rounds listed] group 0: %lld walkers, %lld rounds to the end (%zu seen by the host); rounds with fewer than 1/2, 1/4, 1/10, 1/50 of the walkers listed: %lld %lld %lld %lld (of the seen);",
                full, (long long)(listed_seen.back() >> 32), listed_seen.size(), below[0], below[1], below[2], below[3]);
        for (size_t k = 0; k < listed_seen.size(); k += std::max<size_t>(1, listed_seen.size() / 24))
            fprintf(stderr, " r%lld:%u", (long long)(listed_seen[k] >> 32), (unsigned)listed_seen[k]);
        fprintf(stderr, "\n");
    }
    if (getenv("RVLL_WALK_GEOM_DUMP"))
        fprintf(stderr, "[rounds host] %lld launches, %.2f us each inside the launch calls, walk %.2f ms\n", n_launch, n_launch ? launch_ns / 1e3 / n_launch : 0.,
                duration_cast<nanoseconds>(steady_clock::now() - t_walk0).count() / 1e6);
    if (d_stamps) {
        std::vector<unsigned long long> sv((size_t)8 * stamp_rounds * nblk);
        (void)hipMemcpy(sv.data(), d_stamps, sizeof(unsigned long long) * sv.size(), hipMemcpyDeviceToHost);
        (void)hipFree(d_stamps);
        const long long nb0 = (ga[0].K + W - 1) / W;
        for (int r = 0; r < stamp_rounds && r < n_tile[0]; ++r) {
            unsigned long long t0 = ~0ull, t_last_start = 0, t_end = 0, n = 0;
            double seg[7] = {};
            for (long long b = 0; b < nb0; ++b) {
                const unsigned long long* q = &sv[8 * ((size_t)r * nb0 + b)];
                if (!q[7]) continue;
                t0 = std::min(t0, q[0]); t_last_start = std::max(t_last_start, q[0]); t_end = std::max(t_end, q[7]);
                for (int k = 0; k < 7; ++k) seg[k] += (q[k + 1] >= q[k] && q[k + 1]) ? (double)(q[k + 1] - q[k]) / 100. : 0.;
                ++n;
            }
            if (n) fprintf(stderr, "[step stamps] round %d: %llu blocks, last start +%.2f, end +%.2f us; per block: fetch %.2f, accept+atomic %.2f, move-in+directions %.2f, "
                           "t-phase %.2f, rows %.2f, prior %.2f, store %.2f us\n", r, n, (t_last_start - t0) / 100., (t_end - t0) / 100.,
                           seg[0] / n, seg[1] / n, seg[2] / n, seg[3] / n, seg[4] / n, seg[5] / n, seg[6] / n);
        }
    }
    if (status != RVLL_OK) return status;
    long long total = 0, evaluated = 0, rounds = 0;
    std::vector<long long> part(2 * (size_t)nblk);
    for (int g = 0; g < G; ++g) {
        HIP_TRY(hipMemcpy(part.data(), ga[(size_t)g].calls_part, sizeof(long long) * part.size(), hipMemcpyDeviceToHost));
        for (long long b = 0; b < nblk; ++b) { total += part[(size_t)b]; evaluated += part[(size_t)(nblk + b)]; }
        rounds = std::max(rounds, n_tile[(size_t)g]);
    }
    h->walk_rounds_used = (int)std::min<long long>(rounds, 0x7fffffff);
    if (calls) *calls = total;
    if (slots) *slots = evaluated;
    return RVLL_OK;
}

// The walk of the K rows resident in d_walk_u / d_walk_theta / d_walk_logl (chol and wrapped already uploaded): every
// launch it takes — the first part, the rest (rows dealt to the workgroups by what they cost so far), the full-solver
// finish of rows the slim kernel deferred — leaves the end points in those buffers.  Synchronises the compute stream.
int walk_core(rvll_handle* h, int64_t K, double lstar, int32_t nsteps, int32_t max_rounds, uint64_t seed,
              int64_t walker_base, int64_t* ncalls)
{
    const size_t D = (size_t)h->L.ndim;
    hipStream_t st = h->compute;
    int rc;
    HIP_TRY(hipMemsetAsync(h->d_walk_ncalls, 0, kWalkWords * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(h->d_walk_wflag, 0, sizeof(int32_t) * (size_t)K, st));
    // the walk keeps per-walker state in LDS next to the tile's carve: shrink the group until both fit
    auto walk_args = [&](long long n, rvll::LoglikeArgs* a) -> int {
        int r = build_args(h, h->d_theta, h->d_logL2[0], h->d_flags2[0], n, a);
        if (r) return r;
        make_fused(h, h->d_cube, h->d_theta, a);
        a->defer = nullptr;                            // deferrals are per walker here (steps_done), not per batch
        // A Kepler solve that wanders is 30 - 350 sequential Newton steps, and its exact redo (correctly rounded sin / cos in
        // double-double, ~2 us a step per wave) holds up whatever waits for its tile — a whole round of the rounds form: with
        // the redo inside the walk's tiles 1.65e8 calls/s became 1.04e8 (a candidate in ten thousand wanders; a round of 16384
        // candidates nearly always has one).  So the walk's tiles evaluate such candidates with the ordinary sin / cos — the
        // accept decision cannot tell the two values apart unless they straddle lstar, 1e-9 of |log-L| apart — and report which
        // walkers END on one (w.wflag); their log-L is put right below, by the batch kernel, which carries the redo.
        a->cr_redo = getenv("RVLL_WALK_CR") ? atoi(getenv("RVLL_WALK_CR")) : 0;      // (measurement switch)
        auto window = [&](int pb) {                    // the tile's contribution window also holds 3 PB D doubles of the walk
            int ch = std::min(h->chunk_items, std::max(rvll::kThreads, pb * h->Ne));
            ch = std::max(ch, 3 * pb * a->D);
            return (ch + 1) & ~1;
        };
        // Walker slots per workgroup: the walk's own phases cost a workgroup iteration the same whatever the number of
        // slots, so more slots spread them thinner — as long as four workgroups still fit a compute unit's LDS.  Measured at
        // cfg3 (profiles/r03_walk_forms.txt): 8: 1.54, 10: 1.58, 12: 1.58, 14: 1.53, 16: 1.47e8 calls/s inside the walk.
        if (h->pb_override <= 0 && n >= 4096) {
            a->PB = std::min(10, rvll::kMaxPointsPerBlock);
            a->CH = window(a->PB);
            while (a->PB > 1 && 4 * rvll::walk_lds_bytes(*a) > rvll::kCuLdsBudget) { a->PB -= 1; a->CH = window(a->PB); }
        }
        a->CH = window(a->PB);
        while (a->PB > 1 && (rvll::walk_lds_bytes(*a) > 60 * 1024 || (long long)a->PB * a->D > 4 * rvll::kThreads)) {
            a->PB -= 1;
            a->CH = window(a->PB);
        }
        if (rvll::walk_lds_bytes(*a) > 64 * 1024 || (long long)a->PB * a->D > 4 * rvll::kThreads)
            return report_error(RVLL_E_UNSUPPORTED, "%d parameters exceed the walk kernel's LDS budget", a->D);
        return RVLL_OK;
    };
    rvll::LoglikeArgs a;
    rc = walk_args(K, &a);
    if (rc) return rc;
    // Slim walk (verified-table quantiles only, 4 waves per SIMD) when every Beta / Gamma prior has such a table;
    // walkers it could not finish come back with steps_done < nsteps and are finished by the fat kernel below.
    const bool slim = h->all_direct && !getenv("RVLL_WALK_FAT");
    // the rounds form (walk_rounds above) takes every walk the slim stage applies to; the single-kernel forms below remain for
    // the full-solver walk, for the rows the rounds form hands back (deferred at a candidate the tables do not cover), and
    // behind their switches
    long long rounds_calls = 0, rounds_slots = 0;
    bool by_rounds = false;
    h->walk_rounds_used = 0;
    if (slim && rounds_wanted(K)) {
        rc = walk_rounds(h, K, lstar, nsteps, max_rounds, seed, walker_base, &rounds_calls, &rounds_slots);
        if (rc == RVLL_OK) by_rounds = true;
        else if (rc != RVLL_E_UNSUPPORTED) return rc;
    }
    int spec = h->walk_spec;
    if (const char* e = getenv("RVLL_WALK_SPEC")) spec = atoi(e);       // measurement switch (1: no speculation)
    spec = std::max(1, std::min(spec, rvll::kMaxPointsPerBlock));
    rvll::WalkArgs w{h->d_walk_u, h->d_walk_theta, h->d_walk_logl, h->d_walk_chol, h->d_walk_wrapped, (long long)K,
                     nsteps, max_rounds, (unsigned long long)seed, lstar, h->d_walk_ncalls,
                     h->d_walk_steps, nullptr, nullptr, (long long)walker_base, spec, h->d_walk_ncalls + 1,
                     h->d_walk_ncalls + kWalkWords - 1, nullptr, nullptr, 0, h->d_walk_wflag};
    // no more workgroups than the chip holds at once; freed walker slots draw the remaining rows from a queue
    // (RVLL_WALK_QUEUE, a measurement / test switch: 0 = one workgroup per PB rows, as many residency rounds as that
    // takes; n > 0 = as many workgroups as n compute units hold, so that a small walk goes through the queue too)
    const char* qenv = getenv("RVLL_WALK_QUEUE");
    const int max_cus = qenv ? std::max(0, std::min(atoi(qenv), h->n_cu)) : h->n_cu;
    // With more rows than walker slots a row handed out late still takes a whole walk — nsteps sequential moves — and the
    // kernel ends in a drain (phase clock: mean workgroup life 7.2 ms of a 9.5 ms kernel at 16384 rows).  What a row costs
    // per move is a property of where it walks, so the walk is launched in two parts: the first moves of every row through
    // the queue (short rows: a fine grain), counting the candidates each one needs; then the rest in the "rows" form —
    // every workgroup OWNS an equal share of the rows by that cost and interleaves them over its walker slots, so all rows
    // of the launch end together (rvll_walk.hip, slice_walk_rows_kernel).  Results are those of one launch (the moves of
    // a row do not care which launch makes them).  RVLL_WALK_PARTS=1: one launch (measurement / test switch).
    // RVLL_WALK_ROWS=1 selects the rows form; the DEFAULT is the second part through the queue as well, most expensive
    // rows first (round 2's form): measured on bench.py's nested run (profiles/r03_walk_forms.txt) the rows form balances
    // the workgroups as designed — and is 7 % slower (1.38 vs 1.48e8 calls/s inside the walk): with every slot always
    // holding a walker no tile slot is ever free for candidates ahead, and the kernel is bound by what a workgroup's
    // iteration costs (2600 vector instructions per candidate against the batch kernel's 1990, VALUs busy 75 %), not by
    // its tail.  Kept, tested bit-identical, for walks whose rows differ more than cfg3's.
    const long long resident = max_cus > 0 ? rvll::slice_walk_resident_blocks(a, !slim, max_cus) : 0;
    const char* penv = getenv("RVLL_WALK_PARTS");
    const bool two_parts = resident > 0 && K > resident * a.PB && nsteps >= 8 && !(penv && atoi(penv) == 1);
    const char* renv = getenv("RVLL_WALK_ROWS");
    const bool rows_form = renv && atoi(renv) >= 1 && 3LL * a.PB * a.D <= a.CH;      // (the rows kernels park their candidates in the tile's window)
    const int rows_wide = renv && atoi(renv) == 2 ? rvll::kCuThreads : renv && atoi(renv) == 3 ? 512 : 0;   // 2: one 1024-thread workgroup per CU, 3: two of 512
    if (two_parts && !by_rounds) {
        w.nsteps = std::max(1, rows_form ? nsteps / 8 : nsteps / 4);
        if (const char* e = getenv("RVLL_WALK_FIRST")) w.nsteps = std::max(1, std::min(nsteps - 1, atoi(e)));   // measurement switch
        w.cost = h->d_walk_cost;
    }
    if (!by_rounds) HIP_TRY(rvll::launch_slice_walk(a, w, !slim, max_cus, st));
    if (two_parts && !by_rounds) {
        const int first = w.nsteps;
        std::vector<int32_t> cost((size_t)K), done((size_t)K), order((size_t)K);
        HIP_TRY(hipMemcpyAsync(cost.data(), h->d_walk_cost, sizeof(int32_t) * (size_t)K, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(done.data(), h->d_walk_steps, sizeof(int32_t) * (size_t)K, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        // counting sort, most expensive first; rows that did not complete the first part (deferred) go last
        const int32_t cmax = std::min<int32_t>(first * max_rounds, 1 << 16);
        std::vector<int32_t> count((size_t)cmax + 2, 0);
        auto key = [&](int64_t i) { return done[(size_t)i] >= first ? std::min(std::max(cost[(size_t)i], 0), cmax) + 1 : 0; };
        for (int64_t i = 0; i < K; ++i) ++count[(size_t)key(i)];
        const int32_t nkey0 = count[0];                  // rows the first part deferred: left to the full-solver pass below
        if (getenv("RVLL_WALK_COST_DUMP")) {
            std::vector<int32_t> cs(cost);
            std::sort(cs.begin(), cs.end());
            double sum = 0; for (int32_t c : cs) sum += c;
            fprintf(stderr, "[walk cost, first %d moves] K=%lld mean %.1f  p50 %d  p90 %d  p99 %d  p99.9 %d  max %d\n", first, (long long)K,
                    sum / (double)K, cs[(size_t)(K / 2)], cs[(size_t)(K * 9 / 10)], cs[(size_t)(K * 99 / 100)], cs[(size_t)(K * 999 / 1000)], cs.back());
        }
        int32_t pos = 0;
        for (int32_t c = cmax + 1; c >= 0; --c) { const int32_t n_c = count[(size_t)c]; count[(size_t)c] = pos; pos += n_c; }
        for (int64_t i = 0; i < K; ++i) order[(size_t)count[(size_t)key(i)]++] = (int32_t)i;
        const int64_t K2 = K - nkey0;
        w.nsteps = nsteps;
        w.cost = nullptr;
        w.step_start = h->d_walk_steps;    // every row resumes where the first part left it (read before it is rewritten)
        if (K2 > 0 && rows_form) {
            // as many workgroups as the chip holds, every one an equal share of the rows (snake deal of the sorted order,
            // in the kernel); a share that does not fit the kernel's LDS goes in several launches, one after the other.
            // RVLL_WALK_ROWS=2: the CU-wide form — one 1024-thread workgroup per compute unit with as many walker slots
            // (<= 64) as its LDS holds next to the parked rows, the tile in its CU-wide form; 3: two 512-thread workgroups
            rvll::LoglikeArgs ar = a;
            int64_t G = std::min<int64_t>((K2 + a.PB - 1) / a.PB, resident);
            // (the wide forms exist for the slim stage only: the full-solver instantiation does not fit 128 VGPRs unspilled)
            const int nt = (rows_wide && slim) ? rows_wide : rvll::kThreads;
            const bool cu_wide = nt != rvll::kThreads;
            const size_t wide_budget = nt == rvll::kCuThreads ? rvll::kCuLdsBudget : rvll::kCuLdsBudget / 2;
            if (cu_wide) {
                G = std::min<int64_t>((int64_t)(max_cus > 0 ? max_cus : h->n_cu) * (rvll::kCuThreads / nt), K2);
                const int64_t rows = (K2 + G - 1) / G;
                int slots = (int)std::min<int64_t>(rvll::kWave, rows);
                auto fits = [&](int sl) {
                    ar.PB = sl;
                    ar.CH = (std::max(sl * h->Ne, 3 * sl * h->L.ndim) + 1) & ~1;
                    return rvll::walk_rows_lds_bytes(ar, (int)rows) <= wide_budget;
                };
                while (slots > 1 && !fits(slots)) --slots;
                if (!fits(slots)) return report_error(RVLL_E_UNSUPPORTED, "the wide walk does not fit %lld rows per workgroup", (long long)rows);
                rc = rvll_dev_reserve(h, std::max<int64_t>(K, G * slots) + rvll::kMaxPointsPerBlock);   // the tiles' scratch rows
                if (rc) return rc;
                ar.theta = h->d_theta; ar.logL = h->d_logL2[0]; ar.flags = h->d_flags2[0];
                make_fused(h, h->d_cube, h->d_theta, &ar);
                ar.defer = nullptr;
            }
            int64_t rmax = 1;
            const size_t budget = cu_wide ? wide_budget : (size_t)60 * 1024;
            while (rmax < 4096 && rvll::walk_rows_lds_bytes(ar, (int)rmax + 1) <= budget) ++rmax;
            const int64_t chunk = G * rmax;
            HIP_TRY(hipMemcpyAsync(h->d_walk_order, order.data(), sizeof(int32_t) * (size_t)K2, hipMemcpyHostToDevice, st));
            for (int64_t lo = 0; lo < K2; lo += chunk) {
                const int64_t n = std::min<int64_t>(chunk, K2 - lo);
                rvll::WalkArgs wr = w;
                wr.K = n;
                wr.order = h->d_walk_order + lo;
                const int64_t g = std::min<int64_t>(G, (n + ar.PB - 1) / ar.PB);
                wr.rows_per_wg = (int)((n + g - 1) / g);
                HIP_TRY(rvll::launch_slice_walk_rows(ar, wr, !slim, (int)g, nt, st));
            }
            HIP_TRY(hipStreamSynchronize(st)); // `order` goes out of scope
        } else if (K2 > 0) {
            {
                // the workgroups' first rows: deal the G * PB most expensive ones round the workgroups like cards, so that
                // every workgroup starts with one of the G longest, one of the next G, ... — eight long rows in one
                // workgroup would leave it no free tile slot to evaluate candidates ahead with, and they are the critical path
                const int64_t G = std::min<int64_t>((K2 + a.PB - 1) / a.PB, resident), first_rows = std::min<int64_t>(G * a.PB, K2);
                std::vector<int32_t> dealt((size_t)first_rows);
                int64_t k = 0;
                for (int64_t pl = 0; pl < a.PB; ++pl)
                    for (int64_t b = 0; b < G; ++b) {
                        const int64_t slot = b * a.PB + pl;
                        if (slot < first_rows && k < first_rows) dealt[(size_t)slot] = order[(size_t)k++];
                    }
                std::copy(dealt.begin(), dealt.end(), order.begin());
            }
            HIP_TRY(hipMemcpyAsync(h->d_walk_order, order.data(), sizeof(int32_t) * (size_t)K2, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemsetAsync(h->d_walk_ncalls + kWalkWords - 1, 0, sizeof(unsigned long long), st));   // the queue; the counts go on
            w.K = K2;
            w.order = h->d_walk_order;
            HIP_TRY(rvll::launch_slice_walk(a, w, !slim, max_cus, st));
            HIP_TRY(hipStreamSynchronize(st)); // `order` goes out of scope
        }
        w.K = K;
        w.order = nullptr;
        w.step_start = nullptr;
    }
    unsigned long long evaluated[kWalkWords] = {};
    h->walk_evaluated = 0;
    std::vector<int32_t> steps(slim ? (size_t)K : 0), wf((size_t)K);
    HIP_TRY(hipMemcpyAsync(wf.data(), h->d_walk_wflag, sizeof(int32_t) * (size_t)K, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(evaluated, h->d_walk_ncalls, sizeof evaluated, hipMemcpyDeviceToHost, st));
    if (slim) HIP_TRY(hipMemcpyAsync(steps.data(), h->d_walk_steps, sizeof(int32_t) * (size_t)K, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    long long total = (long long)evaluated[0] + rounds_calls;
    h->walk_evaluated = (long long)evaluated[1] + rounds_slots;
    for (int k = 0; k < 6; ++k) h->walk_phase[k] = evaluated[2 + k];
    if (getenv("RVLL_WALK_TILE_DUMP") && evaluated[6])       // diagnostic build: the tile's own phases inside the walk (100 MHz ticks)
        fprintf(stderr, "[walk tile phases, summed over %llu workgroups] stage %llu  decode %llu  items %llu  reduce+write %llu ticks\n",
                evaluated[6], evaluated[8], evaluated[9], evaluated[10], evaluated[11]);
    if (slim) {
        std::vector<int32_t> ids, start;
        for (int64_t i = 0; i < K; ++i)
            if (steps[(size_t)i] < nsteps) { ids.push_back((int32_t)i); start.push_back(steps[(size_t)i]); }
        if (!ids.empty()) {
            // finish the interrupted walkers with the full solvers inline: same seed, same walker index in the
            // random-number counters, resumed at the start of the move that was interrupted.  Rare: the rows travel
            // through the host (the whole buffers down, the interrupted rows compacted to their front, walked, and
            // everything put back)
            const size_t M = ids.size();
            std::vector<double> hu(D * (size_t)K), hth(D * (size_t)K), hl((size_t)K), su(M * D), sth(M * D), sl(M);
            HIP_TRY(hipMemcpyAsync(hu.data(), h->d_walk_u, sizeof(double) * D * (size_t)K, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(hth.data(), h->d_walk_theta, sizeof(double) * D * (size_t)K, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(hl.data(), h->d_walk_logl, sizeof(double) * (size_t)K, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (size_t j = 0; j < M; ++j) {
                memcpy(&su[j * D], &hu[(size_t)ids[j] * D], sizeof(double) * D);
                memcpy(&sth[j * D], &hth[(size_t)ids[j] * D], sizeof(double) * D);
                sl[j] = hl[(size_t)ids[j]];
            }
            HIP_TRY(hipMemcpyAsync(h->d_walk_u, su.data(), sizeof(double) * D * M, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(h->d_walk_theta, sth.data(), sizeof(double) * D * M, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(h->d_walk_logl, sl.data(), sizeof(double) * M, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(h->d_walk_wid, ids.data(), sizeof(int32_t) * M, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(h->d_walk_start, start.data(), sizeof(int32_t) * M, hipMemcpyHostToDevice, st));
            std::vector<int32_t> wf_sub(M);
            for (size_t j = 0; j < M; ++j) wf_sub[j] = wf[(size_t)ids[j]];
            HIP_TRY(hipMemcpyAsync(h->d_walk_wflag, wf_sub.data(), sizeof(int32_t) * M, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemsetAsync(h->d_walk_ncalls, 0, kWalkWords * sizeof(unsigned long long), st));
            rvll::LoglikeArgs a2;
            rc = walk_args((long long)M, &a2);
            if (rc) return rc;
            rvll::WalkArgs w2 = w;
            w2.K = (long long)M;
            w2.walker_id = h->d_walk_wid;
            w2.step_start = h->d_walk_start;
            HIP_TRY(rvll::launch_slice_walk(a2, w2, true, max_cus, st));
            HIP_TRY(hipMemcpyAsync(su.data(), h->d_walk_u, sizeof(double) * D * M, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(sth.data(), h->d_walk_theta, sizeof(double) * D * M, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(sl.data(), h->d_walk_logl, sizeof(double) * M, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(wf_sub.data(), h->d_walk_wflag, sizeof(int32_t) * M, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(evaluated, h->d_walk_ncalls, sizeof evaluated, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            total += (long long)evaluated[0];
            h->walk_evaluated += (long long)evaluated[1];
            for (int k = 0; k < 6; ++k) h->walk_phase[k] += evaluated[2 + k];
            for (size_t j = 0; j < M; ++j) {
                memcpy(&hu[(size_t)ids[j] * D], &su[j * D], sizeof(double) * D);
                memcpy(&hth[(size_t)ids[j] * D], &sth[j * D], sizeof(double) * D);
                hl[(size_t)ids[j]] = sl[j];
                wf[(size_t)ids[j]] = wf_sub[j];
            }
            HIP_TRY(hipMemcpyAsync(h->d_walk_u, hu.data(), sizeof(double) * D * (size_t)K, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(h->d_walk_theta, hth.data(), sizeof(double) * D * (size_t)K, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(h->d_walk_logl, hl.data(), sizeof(double) * (size_t)K, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
    }
    // the walkers that END on a candidate with a wandering solve: their log-L as the batch kernel gives it (the exact redo the
    // walk's tiles leave out) — a handful of rows in a million candidates, gathered, evaluated, scattered back
    if (h->wander_exact) {
        std::vector<int32_t> rows;
        for (int64_t i = 0; i < K; ++i) if (wf[(size_t)i]) rows.push_back((int32_t)i);
        if (!rows.empty()) {
            const int64_t n = (int64_t)rows.size();
            HIP_TRY(hipMemcpyAsync(h->d_walk_order, rows.data(), sizeof(int32_t) * rows.size(), hipMemcpyHostToDevice, st));
            HIP_TRY(rvll::launch_gather_rows(h->d_walk_theta, h->d_walk_order, n, (int)D, h->d_theta, st));
            rvll::LoglikeArgs ax;
            int cu = 0;
            rc = build_args(h, h->d_theta, h->d_logL2[0], h->d_flags2[0], n, &ax, &cu);
            if (rc) return rc;
            HIP_TRY(cu > 0 ? rvll::launch_loglike_cu(ax, cu, st) : rvll::launch_loglike(ax, st));
            HIP_TRY(rvll::launch_scatter_rows(h->d_logL2[0], h->d_walk_order, n, 1, h->d_walk_logl, st));
            HIP_TRY(hipStreamSynchronize(st));         // `rows` goes out of scope
        }
    }
    if (ncalls) *ncalls = (int64_t)total;
    h->theta_async = false;
    return RVLL_OK;
}

int walk_check_args(rvll_handle* h, int64_t K, int32_t nsteps, int32_t max_rounds, int64_t walker_base)
{
    if (!h->have_priors) return report_error(RVLL_E_NOPRIORS, "rvll_set_priors has not been called");
    if (K < 0 || nsteps < 0) return report_error(RVLL_E_INVALID, "negative size");
    if (max_rounds < 1 || max_rounds > 4096 || nsteps >= (1 << 18) || K >= (1LL << 31) || walker_base < 0 ||
        walker_base + K >= (1LL << 32))
        return report_error(RVLL_E_INVALID, "nsteps / max_rounds / K / walker_base out of range");
    if (h->L.ndim < 1) return report_error(RVLL_E_INVALID, "no free parameter to walk in");
    return RVLL_OK;
}

int walk_upload_frame(rvll_handle* h, const double* chol, const int32_t* wrapped)
{
    const size_t D = (size_t)h->L.ndim;
    std::vector<int32_t> wr(D, 0);
    if (wrapped) for (size_t k = 0; k < D; ++k) wr[k] = wrapped[k] != 0;
    HIP_TRY(hipMemcpyAsync(h->d_walk_chol, chol, sizeof(double) * D * D, hipMemcpyHostToDevice, h->compute));
    HIP_TRY(hipMemcpyAsync(h->d_walk_wrapped, wr.data(), sizeof(int32_t) * D, hipMemcpyHostToDevice, h->compute));
    HIP_TRY(hipStreamSynchronize(h->compute));         // wr (and pageable sources) may go out of scope
    return RVLL_OK;
}

}  // namespace

extern "C" {

int rvll_slice_walk(rvll_handle* h, double* cube, double* theta, double* logl, int64_t K, double lstar,
                    const double* chol, const int32_t* wrapped, int32_t nsteps, int32_t max_rounds,
                    uint64_t seed, int64_t walker_base, int64_t* ncalls)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (ncalls) *ncalls = 0;
    rc = walk_check_args(h, K, nsteps, max_rounds, walker_base);
    if (rc) return rc;
    if (K == 0 || nsteps == 0) return RVLL_OK;
    if (!cube || !theta || !logl || !chol) return report_error(RVLL_E_INVALID, "null buffer");
    const size_t D = (size_t)h->L.ndim;
    rc = walk_reserve(h, K);
    if (rc) return rc;
    hipStream_t st = h->compute;
    HIP_TRY(hipMemcpyAsync(h->d_walk_u, cube, sizeof(double) * D * (size_t)K, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->d_walk_theta, theta, sizeof(double) * D * (size_t)K, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->d_walk_logl, logl, sizeof(double) * (size_t)K, hipMemcpyHostToDevice, st));
    rc = walk_upload_frame(h, chol, wrapped);
    if (rc) return rc;
    rc = walk_core(h, K, lstar, nsteps, max_rounds, seed, walker_base, ncalls);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(cube, h->d_walk_u, sizeof(double) * D * (size_t)K, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(theta, h->d_walk_theta, sizeof(double) * D * (size_t)K, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(logl, h->d_walk_logl, sizeof(double) * (size_t)K, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RVLL_OK;
}

// ---- nested sampling with the live points resident on the device -------------------------------------------
int rvll_live_init(rvll_handle* h, const double* cube, int64_t N, double* logl_out)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->have_priors) return report_error(RVLL_E_NOPRIORS, "rvll_set_priors has not been called");
    if (N < 1 || N >= (1LL << 31) || !cube) return report_error(RVLL_E_INVALID, "rvll_live_init: bad arguments");
    // a new run starts here: whatever fails below, no earlier run's live set is left looking valid (rvll_live_step and
    // rvll_live_get refuse live_n = 0) — live_n is set again as the last thing, on success
    h->live_n = 0;
    h->dead_n = 0;
    h->sorted_kdead = -1;
    const size_t D = (size_t)std::max(1, h->L.ndim);
    rc = rvll_dev_upload_cube(h, cube, N);
    if (rc) return rc;
    rc = rvll_dev_prior_loglike(h, N);
    if (rc) return rc;
    rc = rvll_dev_sync(h);
    if (rc) return rc;
    rc = use_device(h);                                  // (elements the table-only prior stage handed over are redone here)
    if (rc) return rc;
    if (N > h->live_cap) {
        dev_free(h->d_live_u); dev_free(h->d_live_theta); dev_free(h->d_live_logl); dev_free(h->d_live_idx);
        dev_free(h->d_sort_keys); dev_free(h->d_sort_rows);
        h->live_cap = 0;
        HIP_TRY(hipMalloc(&h->d_live_u, sizeof(double) * D * (size_t)N));
        HIP_TRY(hipMalloc(&h->d_live_theta, sizeof(double) * D * (size_t)N));
        HIP_TRY(hipMalloc(&h->d_live_logl, sizeof(double) * (size_t)N));
        HIP_TRY(hipMalloc(&h->d_live_idx, sizeof(int32_t) * 2 * (size_t)N));
        HIP_TRY(hipMalloc(&h->d_sort_keys, sizeof(unsigned long long) * 2 * (size_t)N));
        HIP_TRY(hipMalloc(&h->d_sort_rows, sizeof(int32_t) * (size_t)N));
        h->live_cap = N;
    }
    if (!h->d_live_mom) HIP_TRY(hipMalloc(&h->d_live_mom, sizeof(double) * (rvll::moments_scratch_doubles((int)D) + D + D * D)));
    hipStream_t st = h->compute;
    HIP_TRY(hipMemcpyAsync(h->d_live_u, h->d_cube, sizeof(double) * D * (size_t)N, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->d_live_theta, h->d_theta, sizeof(double) * D * (size_t)N, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->d_live_logl, h->d_logL2[h->logl_last], sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, st));
    if (logl_out) HIP_TRY(hipMemcpyAsync(logl_out, h->d_live_logl, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->live_n = N;
    h->dead_n = 0;
    return RVLL_OK;
}

int rvll_live_step(rvll_handle* h, const int32_t* order, int64_t kdead, const int32_t* start, double lstar,
                   const double* chol, const int32_t* wrapped, int32_t nsteps, int32_t max_rounds, uint64_t seed,
                   int64_t walker_base, int64_t* ncalls, double* logl_new, double* chol_out)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (ncalls) *ncalls = 0;
    const int64_t N = h->live_n;
    if (N < 1) return report_error(RVLL_E_INVALID, "rvll_live_init has not been called");
    if (!start || !logl_new || kdead < 1 || kdead >= N) return report_error(RVLL_E_INVALID, "rvll_live_step: bad arguments");
    rc = walk_check_args(h, kdead, nsteps, max_rounds, walker_base);
    if (rc) return rc;
    const bool dev_order = order == nullptr;             // the order rvll_live_sort left on the device; start[] are ranks among the survivors
    if (dev_order) {
        if (h->sorted_kdead != kdead) return report_error(RVLL_E_INVALID, "rvll_live_step: order is NULL but no rvll_live_sort(kdead = %lld) precedes", (long long)kdead);
        if (!(lstar == h->sorted_lstar)) return report_error(RVLL_E_INVALID, "rvll_live_step: lstar is not the one rvll_live_sort returned");
        for (int64_t i = 0; i < kdead; ++i)
            if (start[i] < 0 || start[i] >= N - kdead) return report_error(RVLL_E_INVALID, "rvll_live_step: start[%lld] is not a rank among the survivors", (long long)i);
        h->sorted_kdead = -1;                            // (used up, whatever happens below: the step changes the rows)
    }
    for (int64_t i = 0; !dev_order && i < N; ++i)
        if (order[i] < 0 || order[i] >= N) return report_error(RVLL_E_INVALID, "rvll_live_step: order[%lld] out of range", (long long)i);
    if (!dev_order) {
        // the dying rows are scattered into in parallel and appended to the dead store: a row listed twice would race and be counted twice
        std::vector<uint64_t> seen(((size_t)N + 63) / 64, 0);
        for (int64_t i = 0; i < kdead; ++i) {
            uint64_t& word = seen[(size_t)order[i] >> 6];
            const uint64_t bit = 1ull << (order[i] & 63);
            if (word & bit) return report_error(RVLL_E_INVALID, "rvll_live_step: row %d is listed twice among the dying rows", (int)order[i]);
            word |= bit;
        }
    }
    for (int64_t i = 0; !dev_order && i < kdead; ++i)
        if (start[i] < 0 || start[i] >= N) return report_error(RVLL_E_INVALID, "rvll_live_step: start[%lld] out of range", (long long)i);
    const size_t D = (size_t)h->L.ndim;
    const int Di = h->L.ndim;
    rc = walk_reserve(h, kdead);
    if (rc) return rc;
    hipStream_t st = h->compute;
    int32_t* d_order = h->d_live_idx;
    int32_t* d_start = h->d_live_idx + h->live_cap;
    if (dev_order) {
        // d_order holds the device's own order; the ranks go up through the sort's row scratch and become rows on the device
        HIP_TRY(hipMemcpyAsync(h->d_sort_rows, start, sizeof(int32_t) * (size_t)kdead, hipMemcpyHostToDevice, st));
        HIP_TRY(rvll::launch_compose_index(d_order, kdead, h->d_sort_rows, kdead, d_start, st));
    } else {
        h->sorted_kdead = -1;
        HIP_TRY(hipMemcpyAsync(d_order, order, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_start, start, sizeof(int32_t) * (size_t)kdead, hipMemcpyHostToDevice, st));
    }
    // the points that die (rows order[0 .. kdead)) go to the dead store before their rows are overwritten
    if (h->dead_n + kdead > h->dead_cap) {
        const long long cap = std::max<long long>(2 * h->dead_cap, h->dead_n + 4 * kdead);
        double *nt = nullptr, *nl = nullptr;
        HIP_TRY(hipMalloc(&nt, sizeof(double) * D * (size_t)cap));
        {
            const hipError_t e = hipMalloc(&nl, sizeof(double) * (size_t)cap);
            if (e != hipSuccess) {
                (void)hipFree(nt);
                return report_error(e == hipErrorOutOfMemory ? RVLL_E_NOMEM : RVLL_E_HIP, "rvll_live_step: dead store: %s", hipGetErrorString(e));
            }
        }
        if (h->dead_n) {
            hipError_t e = hipMemcpyAsync(nt, h->d_dead_theta, sizeof(double) * D * (size_t)h->dead_n, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(nl, h->d_dead_logl, sizeof(double) * (size_t)h->dead_n, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) {
                (void)hipFree(nt); (void)hipFree(nl);
                return report_error(RVLL_E_HIP, "rvll_live_step: dead store: %s", hipGetErrorString(e));
            }
        }
        dev_free(h->d_dead_theta); dev_free(h->d_dead_logl);
        h->d_dead_theta = nt; h->d_dead_logl = nl; h->dead_cap = cap;
    }
    HIP_TRY(rvll::launch_gather_rows(h->d_live_theta, d_order, kdead, Di, h->d_dead_theta + (size_t)h->dead_n * D, st));
    HIP_TRY(rvll::launch_gather_rows(h->d_live_logl, d_order, kdead, 1, h->d_dead_logl + h->dead_n, st));
    // (dead_n moves on when the step has succeeded, at the bottom: a step that fails below — a covariance that is not positive
    // definite, a walk that fails — leaves the dead store as it was, so a retry does not append the same rows twice)
    // whitening: the caller's factor, or the covariance of the surviving rows order[kdead .. N) summed on the device (in a
    // fixed order) and factored here (19 x 19: host arithmetic; + 1e-14 on the diagonal as evidence_amd/nested.py adds)
    std::vector<double> factor(D * D, 0.);
    if (chol) {
        memcpy(factor.data(), chol, sizeof(double) * D * D);
    } else {
        double* scratch = h->d_live_mom;
        double* d_mean = scratch + rvll::moments_scratch_doubles(Di);
        double* d_cov = d_mean + D;
        HIP_TRY(rvll::launch_moments(h->d_live_u, d_order + kdead, N - kdead, Di, scratch, d_mean, d_cov, st));
        std::vector<double> cov(D * D);
        HIP_TRY(hipMemcpyAsync(cov.data(), d_cov, sizeof(double) * D * D, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (size_t j = 0; j < D; ++j) {                 // Cholesky - Banachiewicz, lower triangle
            for (size_t l = 0; l <= j; ++l) {
                double sum = cov[j * D + l] + (j == l ? 1e-14 : 0.);
                for (size_t m = 0; m < l; ++m) sum -= factor[j * D + m] * factor[l * D + m];
                if (j == l) {
                    if (!(sum > 0.)) return report_error(RVLL_E_INVALID, "rvll_live_step: the live points' covariance is not positive definite");
                    factor[j * D + j] = std::sqrt(sum);
                } else {
                    factor[j * D + l] = sum / factor[l * D + l];
                }
            }
        }
    }
    if (chol_out) memcpy(chol_out, factor.data(), sizeof(double) * D * D);
    // the walkers start from rows start[0 .. kdead)
    HIP_TRY(rvll::launch_gather_rows(h->d_live_u, d_start, kdead, Di, h->d_walk_u, st));
    HIP_TRY(rvll::launch_gather_rows(h->d_live_theta, d_start, kdead, Di, h->d_walk_theta, st));
    HIP_TRY(rvll::launch_gather_rows(h->d_live_logl, d_start, kdead, 1, h->d_walk_logl, st));
    rc = walk_upload_frame(h, factor.data(), wrapped);
    if (rc) return rc;
    rc = walk_core(h, kdead, lstar, nsteps, max_rounds, seed, walker_base, ncalls);
    if (rc) return rc;
    // ... and their end points replace the dead rows
    HIP_TRY(rvll::launch_scatter_rows(h->d_walk_u, d_order, kdead, Di, h->d_live_u, st));
    HIP_TRY(rvll::launch_scatter_rows(h->d_walk_theta, d_order, kdead, Di, h->d_live_theta, st));
    HIP_TRY(rvll::launch_scatter_rows(h->d_walk_logl, d_order, kdead, 1, h->d_live_logl, st));
    HIP_TRY(hipMemcpyAsync(logl_new, h->d_walk_logl, sizeof(double) * (size_t)kdead, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->dead_n += kdead;
    return RVLL_OK;
}

int rvll_live_sort(rvll_handle* h, int64_t kdead, double* dead_logl, double* lstar, double* max_logl)
{
    int rc = use_device(h);
    if (rc) return rc;
    const int64_t N = h->live_n;
    if (N < 1) return report_error(RVLL_E_INVALID, "rvll_live_init has not been called");
    if (kdead < 1 || kdead >= N || !dead_logl || !lstar || !max_logl) return report_error(RVLL_E_INVALID, "rvll_live_sort: bad arguments");
    h->sorted_kdead = -1;
    hipStream_t st = h->compute;
    const size_t need = rvll::sort_temp_bytes(N);
    if (need > h->sort_temp_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        dev_free(h->d_sort_temp);
        h->sort_temp_bytes = 0;
        HIP_TRY(hipMalloc(&h->d_sort_temp, need));
        h->sort_temp_bytes = need;
    }
    int32_t* d_order = h->d_live_idx;
    HIP_TRY(rvll::launch_sort_logl(h->d_live_logl, N, h->d_sort_keys, h->d_sort_keys + h->live_cap, h->d_sort_rows, d_order,
                                   h->d_sort_temp, h->sort_temp_bytes, st));
    // the log-L of the kdead lowest, in order, and of the highest: gathered into the walk's log-L scratch, one download
    rc = walk_reserve(h, kdead + 1);
    if (rc) return rc;
    HIP_TRY(rvll::launch_gather_rows(h->d_live_logl, d_order, kdead, 1, h->d_walk_logl, st));
    HIP_TRY(rvll::launch_gather_rows(h->d_live_logl, d_order + (N - 1), 1, 1, h->d_walk_logl + kdead, st));
    std::vector<double> got((size_t)kdead + 1);
    HIP_TRY(hipMemcpyAsync(got.data(), h->d_walk_logl, sizeof(double) * ((size_t)kdead + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(dead_logl, got.data(), sizeof(double) * (size_t)kdead);
    *lstar = got[(size_t)kdead - 1];
    *max_logl = got[(size_t)kdead];
    h->sorted_kdead = kdead;
    h->sorted_lstar = *lstar;
    return RVLL_OK;
}

int rvll_live_get(rvll_handle* h, double* cube, double* theta, double* logl)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (h->live_n < 1) return report_error(RVLL_E_INVALID, "rvll_live_init has not been called");
    const size_t D = (size_t)h->L.ndim, N = (size_t)h->live_n;
    hipStream_t st = h->compute;
    const bool staged = sizeof(double) * D * N >= kDownloadStagedMin;
    if (cube && staged) { rc = download_rows(h, cube, h->d_live_u, sizeof(double) * D * N); if (rc) return rc; }
    else if (cube) HIP_TRY(hipMemcpyAsync(cube, h->d_live_u, sizeof(double) * D * N, hipMemcpyDeviceToHost, st));
    if (theta && staged) { rc = download_rows(h, theta, h->d_live_theta, sizeof(double) * D * N); if (rc) return rc; }
    else if (theta) HIP_TRY(hipMemcpyAsync(theta, h->d_live_theta, sizeof(double) * D * N, hipMemcpyDeviceToHost, st));
    if (logl) HIP_TRY(hipMemcpyAsync(logl, h->d_live_logl, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RVLL_OK;
}

int rvll_live_dead(rvll_handle* h, int64_t* n_dead, double* theta, double* logl)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!n_dead) return report_error(RVLL_E_INVALID, "n_dead is null");
    const int64_t have = h->dead_n, want = (theta || logl) ? std::min<int64_t>(*n_dead, have) : 0;
    *n_dead = have;
    const size_t D = (size_t)h->L.ndim;
    hipStream_t st = h->compute;
    if (want > 0 && theta && sizeof(double) * D * (size_t)want >= kDeadStagedMin) {
        rc = download_rows(h, theta, h->d_dead_theta, sizeof(double) * D * (size_t)want);
        if (rc) return rc;
    } else if (want > 0 && theta) {
        HIP_TRY(hipMemcpyAsync(theta, h->d_dead_theta, sizeof(double) * D * (size_t)want, hipMemcpyDeviceToHost, st));
    }
    if (want > 0 && logl) HIP_TRY(hipMemcpyAsync(logl, h->d_dead_logl, sizeof(double) * (size_t)want, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RVLL_OK;
}

int rvll_set_walk_speculation(rvll_handle* h, int32_t max_ahead)
{
    if (!h) return report_error(RVLL_E_INVALID, "null handle");
    if (max_ahead < 1 || max_ahead > rvll::kMaxPointsPerBlock)
        return report_error(RVLL_E_INVALID, "max_ahead must be in [1, %d]", rvll::kMaxPointsPerBlock);
    h->walk_spec = max_ahead;
    h->walk_spec_rounds = max_ahead;
    return RVLL_OK;
}

int rvll_slice_walk_evaluated(rvll_handle* h, int64_t* evaluated)
{
    if (!h || !evaluated) return report_error(RVLL_E_INVALID, "null argument");
    *evaluated = h->walk_evaluated;
    return RVLL_OK;
}

int rvll_slice_walk_rounds(rvll_handle* h, int32_t* rounds)
{
    if (!h || !rounds) return report_error(RVLL_E_INVALID, "null argument");
    *rounds = h->walk_rounds_used;
    return RVLL_OK;
}

int rvll_slice_walk_phases(rvll_handle* h, uint64_t out[6])
{
    if (!h || !out) return report_error(RVLL_E_INVALID, "null argument");
    for (int k = 0; k < 6; ++k) out[k] = h->walk_phase[k];
    return RVLL_OK;
}

}  // extern "C"
