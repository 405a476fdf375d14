// rvll_walk_host.hip — host side of the sampler's proposal step: the device-resident slice-sampling walk
// in its forms (rvll_slice_walk) and the live set of nested sampling kept in HBM (rvll_live_*).  Entry points of include/rvll.h;
// the kernels are in rvll_walk.hip, rvll_rounds.hip and rvll_live.hip.
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
        dev_free(h->d_walk_cost); dev_free(h->d_walk_order);
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
        if (!h->d_walk_chol) {
            HIP_TRY(hipMalloc(&h->d_walk_chol, sizeof(double) * D * D));
            HIP_TRY(hipMalloc(&h->d_walk_wrapped, sizeof(int32_t) * D));
            HIP_TRY(hipMalloc(&h->d_walk_ncalls, kWalkWords * sizeof(unsigned long long)));   // calls used, tile slots evaluated, diagnostic bins
        }
        h->walk_cap = (long long)cap;
    }
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
    // the walk keeps per-walker state in LDS next to the tile's carve: shrink the group until both fit
    auto walk_args = [&](long long n, rvll::LoglikeArgs* a) -> int {
        int r = build_args(h, h->d_theta, h->d_logL2[0], h->d_flags2[0], n, a);
        if (r) return r;
        make_fused(h, h->d_cube, h->d_theta, a);
        a->defer = nullptr;                            // deferrals are per walker here (steps_done), not per batch
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
    int spec = h->walk_spec;
    if (const char* e = getenv("RVLL_WALK_SPEC")) spec = atoi(e);       // measurement switch (1: no speculation)
    spec = std::max(1, std::min(spec, rvll::kMaxPointsPerBlock));
    rvll::WalkArgs w{h->d_walk_u, h->d_walk_theta, h->d_walk_logl, h->d_walk_chol, h->d_walk_wrapped, (long long)K,
                     nsteps, max_rounds, (unsigned long long)seed, lstar, h->d_walk_ncalls,
                     h->d_walk_steps, nullptr, nullptr, (long long)walker_base, spec, h->d_walk_ncalls + 1,
                     h->d_walk_ncalls + kWalkWords - 1, nullptr, nullptr, 0};
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
    if (two_parts) {
        w.nsteps = std::max(1, rows_form ? nsteps / 8 : nsteps / 4);
        if (const char* e = getenv("RVLL_WALK_FIRST")) w.nsteps = std::max(1, std::min(nsteps - 1, atoi(e)));   // measurement switch
        w.cost = h->d_walk_cost;
    }
    HIP_TRY(rvll::launch_slice_walk(a, w, !slim, max_cus, st));
    if (two_parts) {
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
    std::vector<int32_t> steps(slim ? (size_t)K : 0);
    HIP_TRY(hipMemcpyAsync(evaluated, h->d_walk_ncalls, sizeof evaluated, hipMemcpyDeviceToHost, st));
    if (slim) HIP_TRY(hipMemcpyAsync(steps.data(), h->d_walk_steps, sizeof(int32_t) * (size_t)K, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    long long total = (long long)evaluated[0];
    h->walk_evaluated = (long long)evaluated[1];
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
            HIP_TRY(hipMemcpyAsync(evaluated, h->d_walk_ncalls, sizeof evaluated, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            total += (long long)evaluated[0];
            h->walk_evaluated += (long long)evaluated[1];
            for (int k = 0; k < 6; ++k) h->walk_phase[k] += evaluated[2 + k];
            for (size_t j = 0; j < M; ++j) {
                memcpy(&hu[(size_t)ids[j] * D], &su[j * D], sizeof(double) * D);
                memcpy(&hth[(size_t)ids[j] * D], &sth[j * D], sizeof(double) * D);
                hl[(size_t)ids[j]] = sl[j];
            }
            HIP_TRY(hipMemcpyAsync(h->d_walk_u, hu.data(), sizeof(double) * D * (size_t)K, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(h->d_walk_theta, hth.data(), sizeof(double) * D * (size_t)K, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(h->d_walk_logl, hl.data(), sizeof(double) * (size_t)K, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
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
        h->live_cap = 0;
        HIP_TRY(hipMalloc(&h->d_live_u, sizeof(double) * D * (size_t)N));
        HIP_TRY(hipMalloc(&h->d_live_theta, sizeof(double) * D * (size_t)N));
        HIP_TRY(hipMalloc(&h->d_live_logl, sizeof(double) * (size_t)N));
        HIP_TRY(hipMalloc(&h->d_live_idx, sizeof(int32_t) * 2 * (size_t)N));
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
    if (!order || !start || !logl_new || kdead < 1 || kdead >= N) return report_error(RVLL_E_INVALID, "rvll_live_step: bad arguments");
    rc = walk_check_args(h, kdead, nsteps, max_rounds, walker_base);
    if (rc) return rc;
    for (int64_t i = 0; i < N; ++i)
        if (order[i] < 0 || order[i] >= N) return report_error(RVLL_E_INVALID, "rvll_live_step: order[%lld] out of range", (long long)i);
    for (int64_t i = 0; i < kdead; ++i)
        if (start[i] < 0 || start[i] >= N) return report_error(RVLL_E_INVALID, "rvll_live_step: start[%lld] out of range", (long long)i);
    const size_t D = (size_t)h->L.ndim;
    const int Di = h->L.ndim;
    rc = walk_reserve(h, kdead);
    if (rc) return rc;
    hipStream_t st = h->compute;
    int32_t* d_order = h->d_live_idx;
    int32_t* d_start = h->d_live_idx + h->live_cap;
    HIP_TRY(hipMemcpyAsync(d_order, order, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_start, start, sizeof(int32_t) * (size_t)kdead, hipMemcpyHostToDevice, st));
    // the points that die (rows order[0 .. kdead)) go to the dead store before their rows are overwritten
    if (h->dead_n + kdead > h->dead_cap) {
        const long long cap = std::max<long long>(2 * h->dead_cap, h->dead_n + 4 * kdead);
        double *nt = nullptr, *nl = nullptr;
        HIP_TRY(hipMalloc(&nt, sizeof(double) * D * (size_t)cap));
        HIP_TRY(hipMalloc(&nl, sizeof(double) * (size_t)cap));
        if (h->dead_n) {
            HIP_TRY(hipMemcpyAsync(nt, h->d_dead_theta, sizeof(double) * D * (size_t)h->dead_n, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(nl, h->d_dead_logl, sizeof(double) * (size_t)h->dead_n, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        dev_free(h->d_dead_theta); dev_free(h->d_dead_logl);
        h->d_dead_theta = nt; h->d_dead_logl = nl; h->dead_cap = cap;
    }
    HIP_TRY(rvll::launch_gather_rows(h->d_live_theta, d_order, kdead, Di, h->d_dead_theta + (size_t)h->dead_n * D, st));
    HIP_TRY(rvll::launch_gather_rows(h->d_live_logl, d_order, kdead, 1, h->d_dead_logl + h->dead_n, st));
    h->dead_n += kdead;
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
    return RVLL_OK;
}

int rvll_slice_walk_evaluated(rvll_handle* h, int64_t* evaluated)
{
    if (!h || !evaluated) return report_error(RVLL_E_INVALID, "null argument");
    *evaluated = h->walk_evaluated;
    return RVLL_OK;
}

int rvll_slice_walk_phases(rvll_handle* h, uint64_t out[6])
{
    if (!h || !out) return report_error(RVLL_E_INVALID, "null argument");
    for (int k = 0; k < 6; ++k) out[k] = h->walk_phase[k];
    return RVLL_OK;
}

}  // extern "C"
