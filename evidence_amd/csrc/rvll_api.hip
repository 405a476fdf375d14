// rvll_api.hip — host side of the C-ABI declared in include/rvll.h.
//
// One handle = one device + its streams (compute = pipeline lane 0, further lanes for the multi-GPU step, one
// for the scalar-call server) + resident epoch table, layout, prior tables and batch buffers.  No PyTorch, no
// other runtime: plain HIP, and RCCL (loaded lazily with dlopen) for the multi-GPU all-gathers.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include <unordered_map>

#include "rvll_host.h"      // the handle, HIP_TRY, what the other host units share

namespace {

thread_local std::string g_last_error;

int vfail(int code, const char* fmt, va_list ap)
{
    char buf[1024];
    vsnprintf(buf, sizeof buf, fmt, ap);
    g_last_error = buf;
    return code;
}

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    code = vfail(code, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace

// error reporting for the other translation units of the library (rvll_fip.hip)
namespace rvll {
int report_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    code = vfail(code, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace rvll

using rvll::host::dev_free;
using rvll::host::kDownloadStagedMin;
using rvll::host::kDeadStagedMin;

namespace {

hipStream_t lane_stream(const rvll_handle* h, int lane) { return h->lanes[lane]; }

int sync_other_lanes(rvll_handle* h)
{
    for (int l = 1; l < kMaxLanes; ++l)
        if (h->lanes[l]) HIP_TRY(hipStreamSynchronize(h->lanes[l]));
    return RVLL_OK;
}

// A request goes out as the legacy word (models of more than kServerSlotDims parameters: the kernel polls it, then fetches theta) AND
// through the slots (rvll_kernels.h, ServerCtl::in): every value, then beside every value the word keyed with that value — a slot
// read torn or early does not decode to the request, so nothing rests on store order or on how the link splits a read.
void server_post(rvll_handle* h, unsigned long long request, const double* row)
{
    rvll::ServerCtl* c = h->srv;
    const int D = h->L.ndim;
    if (row) memcpy(c->theta, row, sizeof(double) * (size_t)D);
    if (D <= rvll::kServerSlotDims) {
        if (row) for (int i = 0; i < D; ++i) c->in[i].v = row[i];
        __atomic_thread_fence(__ATOMIC_RELEASE);
        for (int i = 0; i < D; ++i) {                                   // the word of a slot is keyed with the slot's value (ServerCtl)
            unsigned long long vb;
            memcpy(&vb, &c->in[i].v, sizeof vb);
            __atomic_store_n(&c->in[i].word, request ^ rvll::ServerCtl::slot_key(vb), __ATOMIC_RELAXED);
        }
    }
    __atomic_store_n(&c->request, request, __ATOMIC_RELEASE);
}

// Ask a running scalar-call server to leave and wait for it.  Every entry point other than the scalar call
// itself goes through use_device(), so the persistent kernel never coexists with allocations, frees or
// collectives of its own handle (hipFree and friends synchronise the whole device).
int server_stop(rvll_handle* h)
{
    if (!h->srv_running || h->srv_dead) return RVLL_OK;
    const unsigned long long request = ((unsigned long long)rvll::kServerQuit << 32) | (unsigned)++h->srv_seq;
    server_post(h, request, nullptr);
    hipError_t e = hipStreamSynchronize(h->srv_stream);
    h->srv_last = request;
    h->srv_running = false;
    if (e != hipSuccess) return fail(RVLL_E_HIP, "scalar server did not stop: %s", hipGetErrorString(e));
    return RVLL_OK;
}

int resolve_fused(rvll_handle* h);
void stream_free(rvll_handle* h);
int download_rows(rvll_handle* h, void* dst, const void* src_dev, size_t bytes);
// device -> pageable host copies from here on go through download_rows: arrays the caller makes per call (above glibc's mmap threshold
// every array is a fresh mapping), and the dead points' one download per run into an array made for it

int use_device(rvll_handle* h)
{
    if (!h) return fail(RVLL_E_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    int rc = server_stop(h);
    if (rc) return rc;
    return h->fused_pending ? resolve_fused(h) : RVLL_OK;
}

bool slot_ok(const rvll_slot& s, int D) { return s.idx < D; }

int validate_layout(const rvll_layout* L)
{
    if (!L) return fail(RVLL_E_INVALID, "layout is null");
    if (L->struct_size != (int32_t)sizeof(rvll_layout))
        return fail(RVLL_E_INVALID, "rvll_layout.struct_size %d != %zu (ABI mismatch)",
                    L->struct_size, sizeof(rvll_layout));
    if (L->ndim < 0 || L->nplanets < 0 || L->ninst < 1 || L->nlinpar < 0)
        return fail(RVLL_E_INVALID, "layout counts out of range (ndim %d, nplanets %d, ninst %d, nlinpar %d)",
                    L->ndim, L->nplanets, L->ninst, L->nlinpar);
    if (L->nplanets > 0 && !L->planets) return fail(RVLL_E_INVALID, "planets pointer is null");
    if (!L->insts) return fail(RVLL_E_INVALID, "insts pointer is null");
    if (L->nlinpar > 0 && !L->linpar) return fail(RVLL_E_INVALID, "linpar pointer is null");
    if (L->precision < RVLL_PREC_FP64 || L->precision > RVLL_PREC_FP32)
        return fail(RVLL_E_INVALID, "unknown precision %d", L->precision);
    if (!(L->tol > 0.) || L->itmax < 1) return fail(RVLL_E_INVALID, "tol/itmax out of range");
    const int D = L->ndim;
    for (int i = 0; i < L->nplanets; ++i) {
        const rvll_planet& p = L->planets[i];
        if (!slot_ok(p.k, D) || !slot_ok(p.p, D) || !slot_ok(p.e1, D) || !slot_ok(p.e2, D) ||
            !slot_ok(p.anom, D) || !slot_ok(p.epoch, D))
            return fail(RVLL_E_INVALID, "planet %d: slot index >= ndim", i + 1);
        if (p.k_kind < 0 || p.k_kind > 1 || p.p_kind < 0 || p.p_kind > 1 || p.ecc_kind < 0 ||
            p.ecc_kind > 2 || p.anom_kind < 0 || p.anom_kind > 1)
            return fail(RVLL_E_INVALID, "planet %d: bad parametrisation enum", i + 1);
    }
    for (int i = 0; i < L->ninst; ++i)
        if (!slot_ok(L->insts[i].offset, D) || (L->has_jitter && !slot_ok(L->insts[i].jitter, D)))
            return fail(RVLL_E_INVALID, "instrument %d: slot index >= ndim", i);
    for (int i = 0; i < 4; ++i)
        if (!slot_ok(L->drift[i], D)) return fail(RVLL_E_INVALID, "drift slot index >= ndim");
    if (!slot_ok(L->tref, D)) return fail(RVLL_E_INVALID, "tref slot index >= ndim");
    for (int i = 0; i < L->nlinpar; ++i)
        if (!slot_ok(L->linpar[i], D)) return fail(RVLL_E_INVALID, "linpar slot index >= ndim");
    return RVLL_OK;
}

// Launch geometry: how many live points one 256-thread workgroup takes.
//
// A workgroup puts one wave on each SIMD of its CU and walks ceil(PB*Ne/256) rounds of
// flattened (point, epoch) items; workgroups are dealt round-robin over the CUs.  The
// fp64 pipes are issue-bound, so a CU's time ~ (workgroups it receives) x (rounds each),
// mildly worse when fewer than ~3 waves per SIMD are resident to hide latency.  Pick the
// PB that minimises that, subject to the LDS carve fitting.
size_t lds_bytes_for(const rvll_handle* h, int pb)
{
    rvll::LoglikeArgs a{};
    a.PB = pb; a.D = h->L.ndim; a.Np = h->L.nplanets; a.Ni = h->L.ninst; a.nlin = h->L.nlinpar;
    a.CH = std::min(h->chunk_items, std::max(rvll::kThreads, pb * h->Ne));
    a.CH = (a.CH + 1) & ~1;
    return rvll::loglike_lds_bytes(a);
}

int choose_points_per_block(rvll_handle* h, long long B)
{
    if (h->pb_override > 0) return std::min(h->pb_override, rvll::kMaxPointsPerBlock);
    { auto it = h->geo.find(B); if (it != h->geo.end()) return it->second; }
    // Cost model fitted to the MI355X sweeps in profiles/r01_sweep_configs.txt.  A workgroup costs its CU
    // W = ceil(PB*Ne/64)/4 wave-rounds per SIMD (a partly filled last round only occupies the waves that
    // have items), plus ~0.6 for staging, decode, barriers and the reduction, plus ~0.5 per extra LDS
    // window when PB*Ne exceeds it.  Workgroups are dealt round-robin and run in residency rounds of
    // up to `occ` per CU; a round with fewer than 4 workgroups per CU hides latency worse; and the
    // kernel ends with a tail of about half of one workgroup's duration (step-count imbalance), which
    // favours more, shorter workgroups.
    const int Ne = h->Ne;
    int best = 1;
    double best_cost = 1e300;
    for (int pb = 1; pb <= rvll::kMaxPointsPerBlock; ++pb) {
        if ((long long)pb > std::max(1LL, B)) break;
        const size_t lds = lds_bytes_for(h, pb);
        if (pb > 1 && lds > 60 * 1024) break;
        const long long items = (long long)pb * Ne;
        // LDS windows hold whole points (or, beyond the window size, slices of one point)
        const double chunks = Ne <= h->chunk_items ? std::ceil((double)pb / (h->chunk_items / Ne))
                                                   : (double)pb * std::ceil((double)Ne / h->chunk_items);
        const double W = (double)((items + rvll::kWave - 1) / rvll::kWave) / 4.0 + 0.6 + 0.5 * (chunks - 1.0);
        const double blocks = std::ceil((double)B / pb);
        int occ;
        { auto it = h->occ_by_lds.find(lds);
          if (it != h->occ_by_lds.end()) occ = it->second;
          else { occ = rvll::loglike_blocks_per_cu(lds); h->occ_by_lds[lds] = occ; } }
        // with two launches in flight (pipeline lanes) the next launch fills the tail, so the tail term drops out
        double remaining = std::ceil(blocks / h->n_cu), cost = h->pipelined ? 0.0 : 0.5 * W;
        while (remaining > 0) {
            const double k = std::min((double)occ, remaining);
            const double pen = k >= 4 ? 1.0 : k >= 3 ? 1.08 : k >= 2 ? 1.25 : 1.6;
            cost += k * W * pen;
            remaining -= k;
        }
        if (cost < best_cost * (1.0 - 1e-9)) { best_cost = cost; best = pb; }
    }
    if (h->geo.size() > 64) h->geo.clear();
    h->geo[B] = best;
    return best;
}

// The CU-wide form (rvll_kernels.hip, loglike_cu_kernel): 1024-thread workgroups, one per CU at a time, each taking
// a tile of PB points whose items all sit in LDS.  Measured against the 256-thread tiles over batch sizes
// (profiles/r02_form_sweep.txt): 7-14 % faster whenever the batch is one to four rounds of tiles over the CUs —
// every CU stays full until its last wave round — down to a few wave rounds per CU; with many rounds per CU the
// tile form wins by 2-7 %, because a CU-filling workgroup's prologue and reduction (~5 us per tile) overlap
// nothing, while four independent workgroups per CU hide each other's.  The tile size is the largest that fits
// the LDS budget, lowered so that the number of tiles is a whole number of rounds over the CUs.
// Returns the grid (0: use the tile form) and sets a->PB / a->CH.
int choose_cu_form(rvll_handle* h, long long B, rvll::LoglikeArgs* a)
{
    if (h->form_override == 1 || h->pb_override > 0) return 0;
    const bool forced = h->form_override == 2;
    // two launches in flight (pipeline lanes): the next launch's tiles backfill every freed slot, which is all the
    // CU-wide form buys, and the 256-thread tiles hide their prologues behind each other (2.71 vs 2.59e8 evals/s)
    if (h->pipelined && !forced) return 0;
    const long long wave_rounds_per_cu = B * h->Ne / rvll::kWave / std::max(1, h->n_cu);
    if (!forced && wave_rounds_per_cu < 8) return 0;
    const long long ncu = std::min<long long>(h->n_cu, B);
    const long long ppc = (B + ncu - 1) / ncu;                      // points per CU
    rvll::LoglikeArgs t = *a;
    auto fits = [&](int pb) {
        t.PB = pb; t.CH = (pb * h->Ne + 1) & ~1;
        return rvll::loglike_lds_bytes(t) <= rvll::kCuLdsBudget;
    };
    int pbmax = (int)std::min<long long>(rvll::kCuMaxPoints, ppc);
    while (pbmax >= 1 && !fits(pbmax)) --pbmax;
    if (pbmax < 1) return 0;
    const long long rounds = (ppc + pbmax - 1) / pbmax;             // tiles per CU
    if (!forced && rounds > 4) return 0;
    const int pb = (int)((B + ncu * rounds - 1) / (ncu * rounds));
    if (!forced && rounds > 1) {
        // several tiles per CU: each one's ~5 us of prologue + reduction is exposed, so the tile has to be long —
        // wave rounds per wave x (0.9 + 1.33 Np) us per round (fitted to the sweep) — for the form to pay
        const double t_tile_us = (double)pb * h->Ne / rvll::kWave / (rvll::kCuThreads / rvll::kWave) * (0.9 + 1.33 * h->L.nplanets);
        if (t_tile_us < 40.0) return 0;
    }
    a->PB = pb;
    a->CH = (pb * h->Ne + 1) & ~1;
    return (int)((B + pb - 1) / pb);
}

int build_args(rvll_handle* h, const double* d_theta, double* d_logL, int32_t* d_flags,
               long long B, rvll::LoglikeArgs* out, int* cu_grid = nullptr)
{
    rvll::LoglikeArgs a{};
    a.layblob = h->d_layblob;
    a.theta = d_theta; a.logL = d_logL; a.flags = d_flags; a.B = B;
    a.t = h->d_t; a.y = h->d_y; a.s2 = h->d_s2; a.inst = h->d_inst; a.linpar = h->d_linpar;
    a.Ne = h->Ne;
    a.planets = h->d_planets; a.insts = h->d_insts; a.linslots = h->d_linslots;
    a.D = h->L.ndim; a.Np = h->L.nplanets; a.Ni = h->L.ninst; a.nlin = h->L.nlinpar;
    a.has_jitter = h->L.has_jitter; a.has_drift = h->L.has_drift;
    a.tref_from_data = h->L.tref_from_data;
    a.tol = h->L.tol; a.itmax = h->L.itmax; a.precision = h->L.precision;
    a.cr_redo = h->wander_exact;
    a.PB = choose_points_per_block(h, B);
    a.CH = std::min(h->chunk_items, std::max(rvll::kThreads, a.PB * h->Ne));
    a.CH = (a.CH + 1) & ~1;
    a.cte = h->cte;
    a.tmin = h->tmin; a.tmax = h->tmax;
    if (cu_grid) {
        *cu_grid = choose_cu_form(h, B, &a);
        if (*cu_grid > 0) { *out = a; return RVLL_OK; }
    }
    // shrink PB until the LDS carve fits the 64 KiB default dynamic limit
    while (a.PB > 1 && rvll::loglike_lds_bytes(a) > 60 * 1024) {
        a.PB -= 1;
        a.CH = std::min(h->chunk_items, std::max(rvll::kThreads, a.PB * h->Ne));
        a.CH = (a.CH + 1) & ~1;
    }
    if (rvll::loglike_lds_bytes(a) > 64 * 1024)
        return fail(RVLL_E_UNSUPPORTED, "per-point state (%d parameters, %d planets) exceeds the LDS budget",
                    a.D, a.Np);
    *out = a;
    return RVLL_OK;
}

hipError_t launch_form(const rvll::LoglikeArgs& a, int cu_grid, hipStream_t stream)
{
    return cu_grid > 0 ? rvll::launch_loglike_cu(a, cu_grid, stream) : rvll::launch_loglike(a, stream);
}

// fused cube -> theta -> log-L launch: the same arguments plus the prior table and the two row buffers
void make_fused(const rvll_handle* h, const double* d_cube, double* d_theta_out, rvll::LoglikeArgs* a)
{
    a->cube = d_cube;
    a->theta_out = d_theta_out;
    a->priors = h->d_priors;
    a->heavy_dims = h->d_heavy;
    a->n_heavy = h->n_heavy;
    a->light_dims = h->d_heavy + h->n_heavy;
    a->defer = h->pin_defer_dev;
    a->slim_umax = h->slim_umax;
}

int ensure_capacity(rvll_handle* h, long long B)
{
    if (B <= h->cap) return RVLL_OK;
    long long cap = std::max<long long>(B, 1024);
    { int rc_ = sync_other_lanes(h); if (rc_) return rc_; }
    if (h->compute) HIP_TRY(hipStreamSynchronize(h->compute));
    dev_free(h->d_theta); dev_free(h->d_cube);
    for (int l = 0; l < kMaxLanes; ++l) { dev_free(h->d_logL2[l]); dev_free(h->d_flags2[l]); }
    h->cap = 0;
    const size_t D = (size_t)std::max(1, h->L.ndim);
    HIP_TRY(hipMalloc(&h->d_theta, sizeof(double) * D * (size_t)cap));
    HIP_TRY(hipMalloc(&h->d_cube,  sizeof(double) * D * (size_t)cap));
    for (int l = 0; l < kMaxLanes; ++l) {
        HIP_TRY(hipMalloc(&h->d_logL2[l], sizeof(double) * (size_t)cap));
        HIP_TRY(hipMalloc(&h->d_flags2[l], sizeof(int32_t) * (size_t)cap));
    }
    h->cap = cap;
    return RVLL_OK;
}

void free_priors(rvll_handle* h)
{
    for (double*& p : h->d_tables) dev_free(p);
    h->d_tables.clear();
    h->table_err.clear();
    h->table_direct.clear();
    dev_free(h->d_priors);
    dev_free(h->d_heavy);
    h->n_heavy = 0;
    h->all_direct = true;
    h->have_priors = false;
}

}  // namespace

extern "C" {

const char* rvll_last_error(void) { return g_last_error.c_str(); }

int rvll_version(int32_t* major, int32_t* minor)
{
    if (major) *major = RVLL_VERSION_MAJOR;
    if (minor) *minor = RVLL_VERSION_MINOR;
    return RVLL_OK;
}

int rvll_device_count(int32_t* count)
{
    if (!count) return fail(RVLL_E_INVALID, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(RVLL_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return RVLL_OK;
}

int rvll_device_name(int32_t device, char* buf, int32_t buflen)
{
    if (!buf || buflen < 1) return fail(RVLL_E_INVALID, "bad buffer");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return RVLL_OK;
}

int rvll_create(const rvll_layout* layout, const double* time, const double* vrad,
                const double* svrad, const int32_t* inst, int32_t n_epochs,
                const double* linpar_series, int32_t device, rvll_handle** out)
{
    if (!out) return fail(RVLL_E_INVALID, "out is null");
    *out = nullptr;
    int rc = validate_layout(layout);
    if (rc) return rc;
    if (!time || !vrad || !svrad || !inst || n_epochs < 1)
        return fail(RVLL_E_INVALID, "epoch table is empty or null");
    if (layout->nlinpar > 0 && !linpar_series)
        return fail(RVLL_E_INVALID, "nlinpar > 0 but linpar_series is null");
    for (int j = 0; j < n_epochs; ++j)
        if (inst[j] < 0 || inst[j] >= layout->ninst)
            return fail(RVLL_E_INVALID, "inst[%d] = %d outside [0, %d)", j, inst[j], layout->ninst);

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(RVLL_E_NODEVICE, "no HIP device available (%s); rvll has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return fail(RVLL_E_NODEVICE, "device %d >= device count %d", device, ndev);

    rvll_handle* h = new (std::nothrow) rvll_handle;
    if (!h) return fail(RVLL_E_NOMEM, "out of host memory");
    h->device = device;
    h->L = *layout;
    h->planets.assign(layout->planets, layout->planets + layout->nplanets);
    h->insts.assign(layout->insts, layout->insts + layout->ninst);
    if (layout->nlinpar) h->linslots.assign(layout->linpar, layout->linpar + layout->nlinpar);
    h->L.planets = h->planets.data();
    h->L.insts = h->insts.data();
    h->L.linpar = h->linslots.data();
    h->Ne = n_epochs;
    h->cte = -0.5 * (double)n_epochs * std::log(2 * M_PI);          // rvmodel:77-78
    h->tmin = h->tmax = time[0];
    for (int j = 1; j < n_epochs; ++j) { h->tmin = std::min(h->tmin, time[j]); h->tmax = std::max(h->tmax, time[j]); }

#define CREATE_TRY(expr)                                                             \
    do {                                                                             \
        hipError_t e2_ = (expr);                                                     \
        if (e2_ != hipSuccess) {                                                     \
            int c_ = fail(e2_ == hipErrorOutOfMemory ? RVLL_E_NOMEM : RVLL_E_HIP,    \
                          "%s failed: %s", #expr, hipGetErrorString(e2_));           \
            rvll_destroy(h);                                                         \
            return c_;                                                               \
        }                                                                            \
    } while (0)

    CREATE_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    CREATE_TRY(hipGetDeviceProperties(&prop, device));
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    CREATE_TRY(hipStreamCreateWithFlags(&h->compute, hipStreamNonBlocking));
    h->lanes[0] = h->compute;
    for (int l = 1; l < kMaxLanes; ++l) CREATE_TRY(hipStreamCreateWithFlags(&h->lanes[l], hipStreamNonBlocking));
    if (const char* e = getenv("RVLL_LANES")) h->nlanes_dev = std::max(1, std::min(kMaxLanes, atoi(e)));
    if (const char* e = getenv("RVLL_WANDER_EXACT")) h->wander_exact = atoi(e) != 0;
    CREATE_TRY(hipHostMalloc(&h->pin_in, rvll_handle::kPinBytes, hipHostMallocMapped));
    CREATE_TRY(hipHostMalloc(&h->pin_out, rvll_handle::kPinBytes, hipHostMallocMapped));
    CREATE_TRY(hipHostGetDevicePointer(&h->pin_in_dev, h->pin_in, 0));
    CREATE_TRY(hipHostGetDevicePointer(&h->pin_out_dev, h->pin_out, 0));
    {
        void *p = nullptr, *pd = nullptr;
        CREATE_TRY(hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent));
        CREATE_TRY(hipHostGetDevicePointer(&pd, p, 0));
        h->pin_defer = static_cast<int*>(p);
        h->pin_defer_dev = static_cast<int*>(pd);
        *h->pin_defer = 0;
        h->slim_umax = rvll::prior_table_umax();
    }
    {
        void* p = nullptr;
        CREATE_TRY(hipHostMalloc(&p, sizeof(rvll::ServerCtl), hipHostMallocMapped | hipHostMallocCoherent));
        h->srv = new (p) rvll::ServerCtl();
        void* pd = nullptr;
        CREATE_TRY(hipHostGetDevicePointer(&pd, p, 0));
        h->srv_dev = static_cast<rvll::ServerCtl*>(pd);
        CREATE_TRY(hipStreamCreateWithFlags(&h->srv_stream, hipStreamNonBlocking));
        CREATE_TRY(hipMalloc(&h->d_srv_out, 2 * sizeof(double)));
        // the env switch obeys the same limit as rvll_scalar_server(): the control block holds kServerMaxDim parameters
        if (const char* e = getenv("RVLL_SCALAR_SERVER")) h->srv_enabled = atoi(e) != 0 && layout->ndim <= rvll::kServerMaxDim;
    }

    const size_t nb = sizeof(double) * (size_t)n_epochs;
    std::vector<double> s2((size_t)n_epochs);
    for (int j = 0; j < n_epochs; ++j) s2[j] = svrad[j] * svrad[j];  // rvmodel:190,192
    CREATE_TRY(hipMalloc(&h->d_t, nb));
    CREATE_TRY(hipMalloc(&h->d_y, nb));
    CREATE_TRY(hipMalloc(&h->d_s2, nb));
    CREATE_TRY(hipMalloc(&h->d_inst, sizeof(int32_t) * (size_t)n_epochs));
    CREATE_TRY(hipMemcpy(h->d_t, time, nb, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->d_y, vrad, nb, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->d_s2, s2.data(), nb, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->d_inst, inst, sizeof(int32_t) * (size_t)n_epochs, hipMemcpyHostToDevice));
    if (layout->nlinpar) {
        CREATE_TRY(hipMalloc(&h->d_linpar, nb * (size_t)layout->nlinpar));
        CREATE_TRY(hipMemcpy(h->d_linpar, linpar_series, nb * (size_t)layout->nlinpar, hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&h->d_linslots, sizeof(rvll_slot) * (size_t)layout->nlinpar));
        CREATE_TRY(hipMemcpy(h->d_linslots, h->linslots.data(), sizeof(rvll_slot) * (size_t)layout->nlinpar, hipMemcpyHostToDevice));
    }
    if (layout->nplanets) {
        CREATE_TRY(hipMalloc(&h->d_planets, sizeof(rvll_planet) * (size_t)layout->nplanets));
        CREATE_TRY(hipMemcpy(h->d_planets, h->planets.data(), sizeof(rvll_planet) * (size_t)layout->nplanets, hipMemcpyHostToDevice));
    }
    CREATE_TRY(hipMalloc(&h->d_insts, sizeof(rvll_inst) * (size_t)layout->ninst));
    CREATE_TRY(hipMemcpy(h->d_insts, h->insts.data(), sizeof(rvll_inst) * (size_t)layout->ninst, hipMemcpyHostToDevice));
    {
        std::vector<char> blob;
        auto append = [&](const void* p, size_t n) { const char* c = static_cast<const char*>(p); blob.insert(blob.end(), c, c + n); };
        append(h->planets.data(), sizeof(rvll_planet) * h->planets.size());
        append(h->insts.data(), sizeof(rvll_inst) * h->insts.size());
        append(h->linslots.data(), sizeof(rvll_slot) * h->linslots.size());
        append(h->L.drift, sizeof(rvll_slot) * 4);
        append(&h->L.tref, sizeof(rvll_slot));
        CREATE_TRY(hipMalloc(&h->d_layblob, std::max<size_t>(blob.size(), 8)));
        if (!blob.empty()) CREATE_TRY(hipMemcpy(h->d_layblob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    }
    if (const char* e = getenv("RVLL_FORM")) h->form_override = !strcmp(e, "tile") ? 1 : !strcmp(e, "cu") ? 2 : 0;
#undef CREATE_TRY
    *out = h;
    return RVLL_OK;
}

int rvll_destroy(rvll_handle* h)
{
    if (!h) return RVLL_OK;
    (void)hipSetDevice(h->device);
    if (h->srv) (void)server_stop(h);
    if (h->compute) (void)hipStreamSynchronize(h->compute);
    (void)sync_other_lanes(h);
    rvll::host::comm_release(h);
    free_priors(h);
    dev_free(h->d_theta); dev_free(h->d_cube);
    for (int l = 0; l < kMaxLanes; ++l) { dev_free(h->d_logL2[l]); dev_free(h->d_flags2[l]); dev_free(h->d_gather2[l]); }
    dev_free(h->d_gather_theta); dev_free(h->d_gather_host_in); dev_free(h->d_gather_host_out);
    dev_free(h->d_live_u); dev_free(h->d_live_theta); dev_free(h->d_live_logl); dev_free(h->d_live_idx); dev_free(h->d_live_mom);
    dev_free(h->d_sort_keys); dev_free(h->d_sort_rows); dev_free(h->d_sort_temp);
    dev_free(h->d_dead_theta); dev_free(h->d_dead_logl);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->pin_defer) (void)hipHostFree(h->pin_defer);
    stream_free(h);
    dev_free(h->d_walk_steps); dev_free(h->d_walk_wid); dev_free(h->d_walk_start); dev_free(h->d_walk_cost); dev_free(h->d_walk_order); dev_free(h->d_walk_wflag);
    dev_free(h->d_rounds); dev_free(h->d_walk_dirs);
    if (h->pin_rounds) (void)hipHostFree(h->pin_rounds);
    if (h->ev_rounds) (void)hipEventDestroy(h->ev_rounds);
    for (auto& s : h->rounds_streams) if (s) (void)hipStreamDestroy(s);
    for (auto& e : h->ev_chain) if (e) (void)hipEventDestroy(e);
    for (auto& e : h->marks) if (e) (void)hipEventDestroy(e);
    if (h->srv_stream) (void)hipStreamDestroy(h->srv_stream);
    if (h->srv) (void)hipHostFree(h->srv);
    dev_free(h->d_srv_out);
    dev_free(h->d_walk_u); dev_free(h->d_walk_theta); dev_free(h->d_walk_logl); dev_free(h->d_walk_chol);
    dev_free(h->d_walk_wrapped); dev_free(h->d_walk_ncalls);
    dev_free(h->d_t); dev_free(h->d_y); dev_free(h->d_s2); dev_free(h->d_inst); dev_free(h->d_linpar);
    dev_free(h->d_planets); dev_free(h->d_insts); dev_free(h->d_linslots); dev_free(h->d_layblob);
    for (int l = 1; l < kMaxLanes; ++l) if (h->lanes[l]) (void)hipStreamDestroy(h->lanes[l]);
    if (h->compute) (void)hipStreamDestroy(h->compute);
    delete h;
    return RVLL_OK;
}

int rvll_set_points_per_block(rvll_handle* h, int32_t points_per_block)
{
    if (!h) return fail(RVLL_E_INVALID, "null handle");
    if (points_per_block > rvll::kMaxPointsPerBlock)
        return fail(RVLL_E_INVALID, "points_per_block > %d", rvll::kMaxPointsPerBlock);
    h->pb_override = points_per_block > 0 ? points_per_block : 0;
    return RVLL_OK;
}

int rvll_set_kernel_form(rvll_handle* h, int32_t form)
{
    if (!h) return fail(RVLL_E_INVALID, "null handle");
    if (form < 0 || form > 2) return fail(RVLL_E_INVALID, "form must be 0 (auto), 1 (tile) or 2 (CU-wide)");
    h->form_override = form;
    return RVLL_OK;
}

// ---- priors -------------------------------------------------------------------
int rvll_set_priors(rvll_handle* h, const rvll_prior* priors, int32_t ndim)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!priors || ndim != h->L.ndim)
        return fail(RVLL_E_INVALID, "set_priors: ndim %d != layout ndim %d (or priors null)", ndim, h->L.ndim);
    for (int d = 0; d < ndim; ++d) {
        const rvll_prior& p = priors[d];
        switch (p.kind) {
        case RVLL_PRIOR_UNIFORM: case RVLL_PRIOR_JEFFREYS: case RVLL_PRIOR_MODJEFFREYS:
        case RVLL_PRIOR_UNIFORMFREQUENCY: case RVLL_PRIOR_NORMAL: case RVLL_PRIOR_LOGNORMAL:
        case RVLL_PRIOR_TRUNCRAYLEIGH: case RVLL_PRIOR_BETA: case RVLL_PRIOR_GAMMA: case RVLL_PRIOR_ALPHA:
        case RVLL_PRIOR_SORTED_UNIFORM: case RVLL_PRIOR_SORTED_LOGUNIFORM:
            break;
        case RVLL_PRIOR_TABLE:
            if (p.table_n < 2 || !p.table_cdf || !p.table_x)
                return fail(RVLL_E_INVALID, "prior %d: table needs >= 2 knots", d);
            for (int i = 1; i < p.table_n; ++i)
                if (!(p.table_cdf[i] >= p.table_cdf[i - 1]))
                    return fail(RVLL_E_INVALID, "prior %d: table knots not sorted at %d", d, i);
            break;
        default:
            return fail(RVLL_E_UNSUPPORTED, "prior %d: kind %d has no device implementation in this build", d, p.kind);
        }
    }
    free_priors(h);
    std::vector<rvll_prior> dev(priors, priors + ndim);
    h->table_err.assign((size_t)ndim, NAN);
    h->table_direct.assign((size_t)ndim, 0);
    for (int d = 0; d < ndim; ++d) {
        if (dev[d].kind == RVLL_PRIOR_BETA || dev[d].kind == RVLL_PRIOR_GAMMA) {
            // the device tabulates this prior's quantile function once; the kernel starts from it
            // (nodes: value, slope, second derivative) and measures its own quintic interpolant against the
            // full solver; a verified table is evaluated by interpolation alone (table_post = 1)
            const size_t nb = sizeof(double) * (size_t)rvll::prior_table_nodes();
            double *dz0 = nullptr, *dz1 = nullptr, *derr = nullptr;
            HIP_TRY(hipMalloc(&dz0, nb)); h->d_tables.push_back(dz0);
            HIP_TRY(hipMalloc(&dz1, 2 * nb)); h->d_tables.push_back(dz1);
            HIP_TRY(hipMalloc(&derr, sizeof(double))); h->d_tables.push_back(derr);
            HIP_TRY(hipMemsetAsync(derr, 0, sizeof(double), h->compute));
            HIP_TRY(rvll::launch_prior_table(dev[d].kind, priors[d].args, dz0, dz1,
                                             reinterpret_cast<unsigned long long*>(derr), h->compute));
            double err = INFINITY;
            HIP_TRY(hipMemcpyAsync(&err, derr, sizeof(double), hipMemcpyDeviceToHost, h->compute));
            HIP_TRY(hipStreamSynchronize(h->compute));
            dev[d].table_cdf = dz0;
            dev[d].table_x = dz1;
            dev[d].table_n = rvll::prior_table_nodes();
            dev[d].table_post = (err <= rvll::prior_table_direct_tol() && !getenv("RVLL_NO_DIRECT_TABLES")) ? 1 : 0;
            h->table_err[(size_t)d] = err;
            h->table_direct[(size_t)d] = dev[d].table_post;
            continue;
        }
        if (dev[d].kind != RVLL_PRIOR_TABLE) { dev[d].table_cdf = dev[d].table_x = nullptr; continue; }
        const size_t nb = sizeof(double) * (size_t)dev[d].table_n;
        double *dc = nullptr, *dx = nullptr;
        HIP_TRY(hipMalloc(&dc, nb)); h->d_tables.push_back(dc);
        HIP_TRY(hipMalloc(&dx, nb)); h->d_tables.push_back(dx);
        HIP_TRY(hipMemcpy(dc, priors[d].table_cdf, nb, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dx, priors[d].table_x, nb, hipMemcpyHostToDevice));
        dev[d].table_cdf = dc;
        dev[d].table_x = dx;
    }
    HIP_TRY(hipMalloc(&h->d_priors, sizeof(rvll_prior) * (size_t)std::max(1, ndim)));
    if (ndim) HIP_TRY(hipMemcpy(h->d_priors, dev.data(), sizeof(rvll_prior) * (size_t)ndim, hipMemcpyHostToDevice));
    std::vector<int32_t> heavy, light;
    h->priors_rowwise = false;
    for (int d = 0; d < ndim; ++d)
        if (priors[d].kind == RVLL_PRIOR_SORTED_UNIFORM || priors[d].kind == RVLL_PRIOR_SORTED_LOGUNIFORM) h->priors_rowwise = true;
    for (int d = 0; d < ndim; ++d)
        (priors[d].kind == RVLL_PRIOR_BETA || priors[d].kind == RVLL_PRIOR_GAMMA ? heavy : light).push_back(d);
    // the light kinds grouped by what their quantile costs (pow / ndtri / exp + log / table search / a division / nothing),
    // costliest first: in a tile of a few points every wave of the staging step otherwise runs every kind of the model
    auto cost_class = [&](int32_t d) {
        switch (priors[d].kind) {
        case RVLL_PRIOR_JEFFREYS: case RVLL_PRIOR_MODJEFFREYS: case RVLL_PRIOR_SORTED_UNIFORM: case RVLL_PRIOR_SORTED_LOGUNIFORM: return 0;
        case RVLL_PRIOR_NORMAL: case RVLL_PRIOR_LOGNORMAL: case RVLL_PRIOR_ALPHA: return 1;
        case RVLL_PRIOR_TRUNCRAYLEIGH: return 2;
        case RVLL_PRIOR_TABLE: return 3;
        case RVLL_PRIOR_UNIFORMFREQUENCY: return 4;
        default: return 5;
        }
    };
    std::stable_sort(light.begin(), light.end(), [&](int32_t x, int32_t y) {
        return cost_class(x) != cost_class(y) ? cost_class(x) < cost_class(y) : priors[x].kind < priors[y].kind; });
    {
        std::vector<int32_t> both(heavy);
        both.insert(both.end(), light.begin(), light.end());
        HIP_TRY(hipMalloc(&h->d_heavy, sizeof(int32_t) * std::max<size_t>(1, both.size())));
        if (!both.empty()) HIP_TRY(hipMemcpy(h->d_heavy, both.data(), sizeof(int32_t) * both.size(), hipMemcpyHostToDevice));
    }
    h->n_heavy = (int)heavy.size();
    h->all_direct = true;
    for (int32_t d : heavy) if (!h->table_direct[(size_t)d]) h->all_direct = false;
    HIP_TRY(hipStreamSynchronize(h->compute));          // start tables are built
    h->have_priors = true;
    return RVLL_OK;
}

int rvll_prior_table_info(rvll_handle* h, int32_t dim, double* max_err, int32_t* direct)
{
    if (!h) return fail(RVLL_E_INVALID, "null handle");
    if (!h->have_priors) return fail(RVLL_E_NOPRIORS, "rvll_set_priors has not been called");
    if (dim < 0 || dim >= (int32_t)h->table_err.size()) return fail(RVLL_E_INVALID, "dim %d out of range", dim);
    if (max_err) *max_err = h->table_err[(size_t)dim];
    if (direct) *direct = h->table_direct[(size_t)dim];
    return RVLL_OK;
}

// ---- device-resident forms ----------------------------------------------------
int rvll_dev_reserve(rvll_handle* h, int64_t B)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (B < 0) return fail(RVLL_E_INVALID, "B < 0");
    return ensure_capacity(h, B);
}

int rvll_dev_upload_theta(rvll_handle* h, const double* theta, int64_t B)
{
    int rc = rvll_dev_reserve(h, B);
    if (rc) return rc;
    if (B == 0) return RVLL_OK;
    if (!theta) return fail(RVLL_E_INVALID, "theta is null");
    const size_t nbytes = sizeof(double) * (size_t)B * (size_t)h->L.ndim;
    rc = sync_other_lanes(h);                                       // other lanes may still be reading the old theta
    if (rc) return rc;
    if (nbytes <= rvll_handle::kPinBytes) {
        // small: stage through pinned memory (a true asynchronous DMA; the caller's buffer is free at once,
        // and every host-buffer call ends in a stream sync before the staging buffer is written again)
        HIP_TRY(hipStreamSynchronize(h->compute));
        memcpy(h->pin_in, theta, nbytes);
        HIP_TRY(hipMemcpyAsync(h->d_theta, h->pin_in, nbytes, hipMemcpyHostToDevice, h->compute));
        h->theta_async = true;
        return RVLL_OK;
    }
    HIP_TRY(hipMemcpyAsync(h->d_theta, theta, nbytes, hipMemcpyHostToDevice, h->compute));
    HIP_TRY(hipStreamSynchronize(h->compute));
    return RVLL_OK;
}

int rvll_dev_upload_cube(rvll_handle* h, const double* cube, int64_t B)
{
    int rc = rvll_dev_reserve(h, B);
    if (rc) return rc;
    if (B == 0) return RVLL_OK;
    if (!cube) return fail(RVLL_E_INVALID, "cube is null");
    const size_t nbytes = sizeof(double) * (size_t)B * (size_t)h->L.ndim;
    if (nbytes <= rvll_handle::kPinBytes) {
        // small: stage through pinned memory (a true asynchronous DMA; the caller's buffer is free at once,
        // and every host-buffer call ends in a stream sync before the staging buffer is written again)
        HIP_TRY(hipStreamSynchronize(h->compute));
        memcpy(h->pin_in, cube, nbytes);
        HIP_TRY(hipMemcpyAsync(h->d_cube, h->pin_in, nbytes, hipMemcpyHostToDevice, h->compute));
        return RVLL_OK;
    }
    HIP_TRY(hipMemcpyAsync(h->d_cube, cube, nbytes, hipMemcpyHostToDevice, h->compute));
    HIP_TRY(hipStreamSynchronize(h->compute));
    return RVLL_OK;
}

int rvll_dev_fill_cube(rvll_handle* h, int64_t B, uint64_t seed)
{
    int rc = rvll_dev_reserve(h, B);
    if (rc) return rc;
    HIP_TRY(rvll::launch_fill_cube(h->d_cube, (long long)B * h->L.ndim, seed, h->compute));
    return RVLL_OK;
}

int rvll_dev_prior(rvll_handle* h, int64_t B)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->have_priors) return fail(RVLL_E_NOPRIORS, "rvll_set_priors has not been called");
    if (B < 0 || B > h->cap) return fail(RVLL_E_INVALID, "B %lld outside reserved capacity %lld", (long long)B, h->cap);
    rc = sync_other_lanes(h);                                       // other lanes may still be reading the old theta
    if (rc) return rc;
    rvll::PriorArgs a{h->d_cube, h->d_theta, (long long)B, h->L.ndim, h->d_priors, h->d_heavy, h->n_heavy};
    HIP_TRY(rvll::launch_prior(a, h->compute));
    h->theta_async = true;
    return RVLL_OK;
}

int rvll_dev_loglike(rvll_handle* h, int64_t B)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (B < 0 || B > h->cap) return fail(RVLL_E_INVALID, "B %lld outside reserved capacity %lld", (long long)B, h->cap);
    if (B == 0) return RVLL_OK;
    const int lane = h->logl_cur;
    if (lane != 0 && h->theta_async) {        // theta was produced on lane 0's stream: order this lane behind it
        HIP_TRY(hipStreamSynchronize(h->compute));
        h->theta_async = false;
    }
    // launches that alternate pipeline lanes keep two kernels in flight: that decides the launch form and geometry
    // (choose_cu_form, choose_points_per_block) and is re-derived per launch, so a handle that went back to one lane
    // goes back to the single-stream choices
    const bool alternating = lane != h->logl_last;
    if (alternating != h->pipelined) { h->pipelined = alternating; h->geo.clear(); }
    rvll::LoglikeArgs a;
    int cu = 0;
    rc = build_args(h, h->d_theta, h->d_logL2[lane], h->d_flags2[lane], B, &a, &cu);
    if (rc) return rc;
    HIP_TRY(launch_form(a, cu, lane_stream(h, lane)));
    h->logl_last = lane;
    return RVLL_OK;
}

int rvll_dev_prior_loglike(rvll_handle* h, int64_t B)
{
    // Back-to-back one-launch batches over the same buffers stay asynchronous: the deferral word is sticky (cleared
    // only by resolve_fused), only the LATEST batch's results can still be looked at, and whoever looks at them goes
    // through use_device -> resolve_fused first — so the pending mark is carried over instead of resolved here.
    if (h) h->fused_pending = 0;
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->have_priors) return fail(RVLL_E_NOPRIORS, "rvll_set_priors has not been called");
    if (B < 0 || B > h->cap) return fail(RVLL_E_INVALID, "B %lld outside reserved capacity %lld", (long long)B, h->cap);
    if (B == 0) return RVLL_OK;
    // One launch (the prior transform in the log-L tile's staging step) needs every Beta / Gamma prior to have a
    // verified table (the slim stage); otherwise: the prior kernels, then the log-L kernel.
    if (!h->all_direct) {
        rc = rvll_dev_prior(h, B);
        if (rc) return rc;
        if (h->logl_cur != 0) h->logl_cur = 0;     // results of this call live on lane 0, as the one-launch form's do
        return rvll_dev_loglike(h, B);
    }
    rc = sync_other_lanes(h);                  // theta is rewritten: no other lane may still be reading it
    if (rc) return rc;
    rvll::LoglikeArgs a;
    int cu = 0;                                // small batches: 256-thread tiles; larger ones: the CU-wide form, as for plain log-L
    h->pipelined = false;
    rc = build_args(h, h->d_theta, h->d_logL2[0], h->d_flags2[0], B, &a, B > kFusedMaxPoints ? &cu : nullptr);
    if (rc) return rc;
    make_fused(h, h->d_cube, h->d_theta, &a);
    if (cu > 0) HIP_TRY(rvll::launch_loglike_cu(a, cu, h->compute));
    else        HIP_TRY(rvll::launch_prior_loglike(a, h->compute));
    h->theta_async = true;
    h->logl_last = 0;
    h->fused_pending = B;                      // whoever touches the results next looks at the defer word first
    return RVLL_OK;
}

int rvll_dev_download(rvll_handle* h, int64_t B, double* theta, double* logL, int32_t* flags)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (B < 0 || B > h->cap) return fail(RVLL_E_INVALID, "B %lld outside reserved capacity %lld", (long long)B, h->cap);
    if (h->logl_last != 0) HIP_TRY(hipStreamSynchronize(h->lanes[h->logl_last]));   // read through lane 0's stream
    if (B > 0) {
        const size_t nt = theta ? sizeof(double) * (size_t)B * (size_t)h->L.ndim : 0;
        const size_t nl = logL ? sizeof(double) * (size_t)B : 0;
        const size_t nf = flags ? sizeof(int32_t) * (size_t)B : 0;
        if (nt + nl + nf <= rvll_handle::kPinBytes) {          // small: one pinned landing zone, one sync
            char* p = static_cast<char*>(h->pin_out);
            if (nt) HIP_TRY(hipMemcpyAsync(p, h->d_theta, nt, hipMemcpyDeviceToHost, h->compute));
            if (nl) HIP_TRY(hipMemcpyAsync(p + nt, h->d_logL2[h->logl_last], nl, hipMemcpyDeviceToHost, h->compute));
            if (nf) HIP_TRY(hipMemcpyAsync(p + nt + nl, h->d_flags2[h->logl_last], nf, hipMemcpyDeviceToHost, h->compute));
            HIP_TRY(hipStreamSynchronize(h->compute));
            if (nt) memcpy(theta, p, nt);
            if (nl) memcpy(logL, p + nt, nl);
            if (nf) memcpy(flags, p + nt + nl, nf);
            return RVLL_OK;
        }
        if (theta) {
            // A device-to-host copy into pageable memory runs at 50 GB/s once the runtime has pinned the destination
            // range and cached that — which it cannot for a range it has never seen: 38 MB of theta into a fresh numpy
            // array take 20-28 ms (1.5 GB/s), and numpy arrays above 32 MB are fresh mappings every time (glibc's mmap
            // threshold stops growing there; below it the heap hands the same block out again and the copy is fast:
            // profiles/r02_d2h_probe.txt).  Large downloads therefore go through pinned staging blocks, the workers
            // copying chunk c to its place while chunks c + 1 .. c + 3 land (download_rows; round 2: two 1 MiB blocks on
            // the calling thread, 6 ms for those 38 MB).
            if (nt < kDownloadStagedMin) {
                HIP_TRY(hipMemcpyAsync(theta, h->d_theta, nt, hipMemcpyDeviceToHost, h->compute));
            } else {
                rc = download_rows(h, theta, h->d_theta, nt);
                if (rc) return rc;
            }
        }
        if (logL)  HIP_TRY(hipMemcpyAsync(logL, h->d_logL2[h->logl_last], nl, hipMemcpyDeviceToHost, h->compute));
        if (flags) HIP_TRY(hipMemcpyAsync(flags, h->d_flags2[h->logl_last], nf, hipMemcpyDeviceToHost, h->compute));
    }
    HIP_TRY(hipStreamSynchronize(h->compute));
    return RVLL_OK;
}

int rvll_dev_mark(rvll_handle* h, int32_t which)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (which < 0 || which > 1) return fail(RVLL_E_INVALID, "which must be 0 (start) or 1 (stop)");
    if (!h->marks[which]) HIP_TRY(hipEventCreate(&h->marks[which]));
    HIP_TRY(hipEventRecord(h->marks[which], h->compute));
    return RVLL_OK;
}

int rvll_dev_mark_elapsed(rvll_handle* h, double* ms)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!ms || !h->marks[0] || !h->marks[1]) return fail(RVLL_E_INVALID, "both marks must have been recorded");
    HIP_TRY(hipEventSynchronize(h->marks[1]));
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, h->marks[0], h->marks[1]));
    *ms = (double)f;
    return RVLL_OK;
}

int rvll_dev_sync(rvll_handle* h)
{
    int rc = use_device(h);
    if (rc) return rc;
    // A blocking hipStreamSynchronize wakes the caller some tens of microseconds after the stream has drained; a caller
    // that is about to read a result (a sampler's proposal round, bench.py's 20-step timed region: 1.3 ms) waits on the
    // stream's status first — a poll costs well under a microsecond — and only falls back to the blocking call when the
    // work is long (the poll gives up after ~2 ms).
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(h->compute);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIP_TRY(q);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    HIP_TRY(hipStreamSynchronize(h->compute));
    rc = sync_other_lanes(h);
    if (rc) return rc;
    h->theta_async = false;
    return RVLL_OK;
}

int rvll_dev_flip_lane(rvll_handle* h)
{
    if (!h) return fail(RVLL_E_INVALID, "null handle");
    h->logl_cur = (h->logl_cur + 1) % (h->nccl_comm[0] ? h->nlanes : h->nlanes_dev);
    return h->logl_cur;
}

int rvll_dev_time_loglike(rvll_handle* h, int64_t B, int32_t warmup, int32_t iters, rvll_timing* out)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!out || iters < 1 || warmup < 0) return fail(RVLL_E_INVALID, "bad timing arguments");
    if (B < 1 || B > h->cap) return fail(RVLL_E_INVALID, "B %lld outside reserved capacity %lld", (long long)B, h->cap);
    rvll::LoglikeArgs a;
    rc = sync_other_lanes(h);
    if (rc) return rc;
    int cu = 0;
    rc = build_args(h, h->d_theta, h->d_logL2[0], h->d_flags2[0], B, &a, &cu);
    if (rc) return rc;
    h->logl_last = 0;
    for (int i = 0; i < warmup; ++i) HIP_TRY(launch_form(a, cu, h->compute));
    std::vector<hipEvent_t> ev((size_t)iters + 1, nullptr);
    int status = RVLL_OK;
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) { status = fail(RVLL_E_HIP, "hipEventCreate failed"); break; }
    if (status == RVLL_OK) {
        hipError_t e = hipEventRecord(ev[0], h->compute);
        for (int i = 0; i < iters && e == hipSuccess; ++i) {
            e = launch_form(a, cu, h->compute);
            if (e == hipSuccess) e = hipEventRecord(ev[(size_t)i + 1], h->compute);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->compute);
        if (e != hipSuccess) status = fail(RVLL_E_HIP, "timed launches failed: %s", hipGetErrorString(e));
    }
    if (status == RVLL_OK) {
        std::vector<double> ms((size_t)iters);
        double sum = 0.;
        for (int i = 0; i < iters; ++i) {
            float f = 0.f;
            (void)hipEventElapsedTime(&f, ev[(size_t)i], ev[(size_t)i + 1]);
            ms[(size_t)i] = f;
            sum += f;
        }
        float total = 0.f;
        (void)hipEventElapsedTime(&total, ev.front(), ev.back());
        std::sort(ms.begin(), ms.end());
        out->kernel_ms_mean = sum / iters;
        out->kernel_ms_min = ms.front();
        out->kernel_ms_median = ms[(size_t)iters / 2];
        out->total_ms = total;
        out->evals = (int64_t)B * iters;
        out->launches = iters;
        out->points_per_block = a.PB;
        out->blocks = (int32_t)((B + a.PB - 1) / a.PB);
        out->threads = cu > 0 ? rvll::kCuThreads : rvll::kThreads;
    }
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    return status;
}

// Diagnostic: ONE launch of the stamped twin of the fp64 log-L kernel over the resident theta (after `warmup`
// ordinary launches, so the clocks are where a bench run has them); returns kTraceWords stamps per workgroup.
int rvll_dev_trace_loglike(rvll_handle* h, int64_t B, int32_t warmup, uint64_t* out, int64_t out_words,
                           int32_t* blocks, int32_t* points_per_block)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (B < 1 || B > h->cap) return fail(RVLL_E_INVALID, "B %lld outside reserved capacity %lld", (long long)B, h->cap);
    if (h->L.precision != RVLL_PREC_FP64) return fail(RVLL_E_UNSUPPORTED, "the trace kernel exists for fp64 only");
    rc = sync_other_lanes(h);
    if (rc) return rc;
    rvll::LoglikeArgs a;
    int cu = 0;
    rc = build_args(h, h->d_theta, h->d_logL2[0], h->d_flags2[0], B, &a, &cu);
    if (rc) return rc;
    const long long nb = (B + a.PB - 1) / a.PB;
    if (blocks) *blocks = (int32_t)nb;
    if (points_per_block) *points_per_block = a.PB;
    if (!out) return RVLL_OK;                               // size query
    const long long words = nb * rvll::kTraceWords;
    if (out_words < words) return fail(RVLL_E_INVALID, "trace buffer too small: %lld < %lld words", (long long)out_words, words);
    unsigned long long* d_trace = nullptr;
    HIP_TRY(hipMalloc(&d_trace, sizeof(unsigned long long) * (size_t)words));
    int status = RVLL_OK;
    hipError_t e = hipMemsetAsync(d_trace, 0, sizeof(unsigned long long) * (size_t)words, h->compute);
    for (int i = 0; i < warmup && e == hipSuccess; ++i) e = launch_form(a, cu, h->compute);
    a.trace = d_trace;
    if (e == hipSuccess) e = cu > 0 ? rvll::launch_loglike_cu(a, cu, h->compute) : rvll::launch_loglike_trace(a, h->compute);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_trace, sizeof(unsigned long long) * (size_t)words, hipMemcpyDeviceToHost, h->compute);
    if (e == hipSuccess) e = hipStreamSynchronize(h->compute);
    if (e != hipSuccess) status = fail(RVLL_E_HIP, "trace launch failed: %s", hipGetErrorString(e));
    (void)hipStreamSynchronize(h->compute);
    dev_free(d_trace);
    h->logl_last = 0;
    return status;
}

// ---- host-buffer hot calls -------------------------------------------------------
}  // extern "C"

namespace {

// A one-launch cube -> log-L batch is in flight or finished: wait for it and, if its slim prior stage deferred any
// element (a quantile outside its table: |logit q| > 30, q = 0 or 1, ...), redo the batch with the prior kernels
// that carry the full solvers — the same rows, the same buffers, bit-identical results for every other point.
int resolve_fused(rvll_handle* h)
{
    const long long B = h->fused_pending;
    h->fused_pending = 0;
    HIP_TRY(hipStreamSynchronize(h->compute));
    if (__atomic_load_n(h->pin_defer, __ATOMIC_ACQUIRE) == 0) return RVLL_OK;
    *h->pin_defer = 0;
    int rc = rvll_dev_prior(h, B);
    if (rc) return rc;
    h->logl_cur = 0;
    return rvll_dev_loglike(h, B);
}

int server_start(rvll_handle* h)
{
    HIP_TRY(hipSetDevice(h->device));
    rvll::LoglikeArgs a;
    int rc = build_args(h, h->srv_dev->theta, h->d_srv_out, reinterpret_cast<int32_t*>(h->d_srv_out + 1), 1, &a);
    if (rc) return rc;
    if (a.PB != 1) { a.PB = 1; a.CH = std::max(rvll::kThreads, std::min(h->chunk_items, h->Ne)); a.CH = (a.CH + 1) & ~1; }
    if (h->have_priors) make_fused(h, h->srv_dev->theta, h->srv_dev->theta, &a);   // the prior op needs the tables
    __atomic_store_n(&h->srv->state, rvll::kServerRunning, __ATOMIC_RELEASE);
    HIP_TRY(rvll::launch_scalar_server(a, h->srv_dev, h->srv_last, h->srv_idle_ticks, h->srv_stream));
    h->srv_running = true;
    return RVLL_OK;
}

// One log-L through the persistent kernel: write theta and a new request number into the control block, spin on
// the answer.  If the kernel left in the meantime (idle timeout) it is started again with the request pending.
int scalar_call(rvll_handle* h, unsigned op, const double* theta, double* logL, int32_t* flags, double* theta_out)
{
    rvll::ServerCtl* c = h->srv;
    if (h->L.ndim > rvll::kServerMaxDim)
        return fail(RVLL_E_UNSUPPORTED, "scalar server supports up to %d parameters", rvll::kServerMaxDim);
    if (h->srv_dead)
        return fail(RVLL_E_HIP, "the scalar server of this handle stopped answering earlier; destroy the handle");
    if (h->srv_running && __atomic_load_n(&c->state, __ATOMIC_ACQUIRE) == rvll::kServerExited) {
        HIP_TRY(hipStreamSynchronize(h->srv_stream));
        h->srv_running = false;
    }
    if (!h->srv_running) {
        HIP_TRY(hipSetDevice(h->device));
        // the server reads the resident epoch table: anything still queued on the other streams goes first
        HIP_TRY(hipStreamSynchronize(h->compute));
        int rc = sync_other_lanes(h);
        if (rc) return rc;
    }
    const unsigned number = (unsigned)++h->srv_seq;
    const unsigned long long request = ((unsigned long long)op << 32) | number;
    server_post(h, request, theta);
    if (!h->srv_running) { int rc = server_start(h); if (rc) return rc; }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(&c->answer.number, __ATOMIC_ACQUIRE) == number) break;
        __builtin_ia32_pause();
        if ((spins & 0xfff) != 0xfff) continue;
        if (__atomic_load_n(&c->state, __ATOMIC_ACQUIRE) == rvll::kServerExited &&
            __atomic_load_n(&c->answer.number, __ATOMIC_ACQUIRE) != number) {
            HIP_TRY(hipStreamSynchronize(h->srv_stream));           // left before it saw this request
            h->srv_running = false;
            int rc = server_start(h);
            if (rc) return rc;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
            // Post the quit word and mark the server dead WITHOUT waiting for the stream: if the persistent kernel is
            // really stuck, a stream synchronisation would never return.  The kernel also leaves by itself after its
            // idle timeout; the handle refuses further scalar calls and the caller should exit, not retry.
            const unsigned long long quit = ((unsigned long long)rvll::kServerQuit << 32) | (unsigned)++h->srv_seq;
            server_post(h, quit, nullptr);
            h->srv_last = quit;
            h->srv_dead = true;
            h->srv_enabled = false;
            return fail(RVLL_E_HIP, "scalar server did not answer within 5 s (handle marked dead)");
        }
    }
    h->srv_last = request;
    if (op == rvll::kServerPrior) {
        memcpy(theta_out, c->theta, sizeof(double) * (size_t)h->L.ndim);
        return RVLL_OK;
    }
    if (op == rvll::kServerPriorLogLike && theta_out) memcpy(theta_out, c->theta, sizeof(double) * (size_t)h->L.ndim);
    if (logL) *logL = c->answer.logL;          // same 16-byte store as the number just seen
    if (flags) *flags = c->answer.flags;
    return RVLL_OK;
}


// ---- large host batches: a pipeline of chunks, the host's copies on worker threads (round 3) --------------------------
// A 262144-point cube -> theta -> log-L call moves 80 MB between the caller's pageable arrays and the device.  Through copy
// commands on pageable memory that is ONE host thread at ~12 GB/s (the runtime's staging copy on the way up, ours — into a
// freshly mapped result array, page faults included — on the way down): 6.5 ms around 1.2 ms of kernels (the
// "cliff").  Here the batch goes in chunks of 16384 rows through pinned staging blocks: worker threads copy chunk c + 2 in
// and chunk c - 1 out while the DMA engines and the kernels work on chunk c (uploads, kernels and downloads each on a stream of
// their own, chained by events: the link carries both directions at once).
// The same kernels on the same rows: the same bits (tests/test_gpu_boundary.py runs every size class).
constexpr long long kStreamChunkRows = 16384;
// Streamed from 24 MB of rows on (165565 points at 19 parameters).  Below, the two-chunk route copies straight from and to the
// caller's arrays and is level or ahead (131072 rows: 1.04 - 1.08 ms against 1.0 - 1.2 ms); above, the caller's arrays are
// fresh mappings every call (glibc's 32 MB mmap threshold), the runtime's cache of pinned ranges misses, and that route
// collapses (262144 rows: 3.5 - 4.1 ms against 1.9 - 2.2 ms; profiles/r03_stream_probe.txt).  RVLL_STREAM_MIN (points): measurement switch.
long long stream_min_points(const rvll_handle* h)
{
    if (const char* e = getenv("RVLL_STREAM_MIN")) return std::max(1ll, atoll(e));
    return (long long)((kStreamMinBytes + sizeof(double) * (size_t)h->L.ndim - 1) / (sizeof(double) * (size_t)h->L.ndim));
}

void stream_free(rvll_handle* h)
{
    delete h->pool;
    h->pool = nullptr;
    for (int s = 0; s < kStageSlots; ++s) {
        if (h->stage_in[s]) (void)hipHostFree(h->stage_in[s]);
        if (h->stage_out[s]) (void)hipHostFree(h->stage_out[s]);
        for (hipEvent_t* e : {&h->stage_ev[s], &h->stage_up[s], &h->stage_done[s]}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
        h->stage_in[s] = h->stage_out[s] = nullptr;
    }
    if (h->stream_up) (void)hipStreamDestroy(h->stream_up);
    if (h->stream_down) (void)hipStreamDestroy(h->stream_down);
    h->stream_up = h->stream_down = nullptr;
    h->stage_in_bytes = h->stage_out_bytes = 0;
}

// workers, events and staging blocks for chunks of `rows` rows (grown on demand, kept with the handle)
int stream_reserve(rvll_handle* h, long long rows)
{
    const size_t D = (size_t)h->L.ndim;
    const size_t in_bytes = sizeof(double) * D * (size_t)rows;
    const size_t out_bytes = (sizeof(double) * (D + 1) + sizeof(int32_t)) * (size_t)rows;
    if (!h->pool) {
        // (2, 4 and 8 workers measure the same within the run-to-run spread — the GPU side of the pipeline is the longer one)
        int n = (int)std::min(4u, std::max(2u, std::thread::hardware_concurrency() / 2));
        if (const char* e = getenv("RVLL_COPY_THREADS")) n = std::max(1, std::min(32, atoi(e)));
        try {
            h->pool = new CopyPool(n);
        } catch (const std::exception& e) {             // (no memory, or the system refuses more threads)
            h->pool = nullptr;
            return fail(RVLL_E_NOMEM, "cannot start %d copy workers: %s", n, e.what());
        }
    }
    for (int s = 0; s < kStageSlots; ++s)
        for (hipEvent_t* e : {&h->stage_ev[s], &h->stage_up[s], &h->stage_done[s]})
            if (!*e) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    if (!h->stream_up) {
        // The runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4; this handle alone has seven
        // streams), and a copy's completion marker holds up whatever else shares its queue.  When one of the two copy streams
        // lands on the kernels' queue, every chunk's kernels wait for the previous chunk's download: 3.3 against 2.0 ms at
        // 262144 rows — and which stream lands where depends on what the process has done before (inside bench.py streams made
        // here collided and the handle's spare lanes did not; in the standalone probe it was the other way round; stream
        // priorities changed nothing).  The Python host therefore asks for 8 queues before the runtime starts
        // (evidence_amd/_abi.py); a C caller sets GPU_MAX_HW_QUEUES=8 in its environment.  Results are the
        // same either way.
        HIP_TRY(hipStreamCreateWithFlags(&h->stream_up, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&h->stream_down, hipStreamNonBlocking));
    }
    if (in_bytes > h->stage_in_bytes || out_bytes > h->stage_out_bytes) {
        for (int s = 0; s < kStageSlots; ++s) {
            if (h->stage_in[s]) (void)hipHostFree(h->stage_in[s]);
            if (h->stage_out[s]) (void)hipHostFree(h->stage_out[s]);
            h->stage_in[s] = h->stage_out[s] = nullptr;
        }
        h->stage_in_bytes = h->stage_out_bytes = 0;
        for (int s = 0; s < kStageSlots; ++s) {
            HIP_TRY(hipHostMalloc(&h->stage_in[s], in_bytes, hipHostMallocDefault));
            HIP_TRY(hipHostMalloc(&h->stage_out[s], out_bytes, hipHostMallocDefault));
        }
        h->stage_in_bytes = in_bytes;
        h->stage_out_bytes = out_bytes;
    }
    return RVLL_OK;
}

// A large device array into the caller's pageable memory: chunks through the pinned staging blocks on the download stream, up to
// three in flight, each copied on to its place by the workers as it lands (a copy command straight into pageable memory runs
// at 50 GB/s into a range the runtime has pinned and cached, at 1.5 - 7 GB/s into one it has never seen — and a result array
// usually is one; round 2 staged through two 1 MiB blocks on the calling thread: 6 ms for 38 MB).  Work queued on lane 0's
// stream before the call is waited for first.
int download_rows(rvll_handle* h, void* dst, const void* src_dev, size_t bytes)
{
    if (!bytes) return RVLL_OK;
    int rc = stream_reserve(h, kStreamChunkRows);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->compute));
    const size_t chunk = h->stage_out_bytes & ~(size_t)4095;
    const int n = (int)((bytes + chunk - 1) / chunk);
    CopyPool& pool = *h->pool;
    CopyPool::Ticket tout[kStageSlots];
    char* to = static_cast<char*>(dst);
    const char* from = static_cast<const char*>(src_dev);
    auto settle = [&](int code) {
        for (int s = 0; s < kStageSlots; ++s) CopyPool::wait(&tout[s]);
        (void)hipStreamSynchronize(h->stream_down);
        pool.busy(false);
        return code;
    };
#define DOWN_TRY(expr) do { const hipError_t err_ = (expr); if (err_ != hipSuccess) { settle(0); HIP_TRY(err_); } } while (0)
    auto fetch = [&](int c) -> hipError_t {
        const int s = c % kStageSlots;
        const size_t off = (size_t)c * chunk;
        hipError_t e = hipMemcpyAsync(h->stage_out[s], from + off, std::min(chunk, bytes - off), hipMemcpyDeviceToHost, h->stream_down);
        return e != hipSuccess ? e : hipEventRecord(h->stage_ev[s], h->stream_down);
    };
    pool.busy(true);
    for (int c = 0; c < std::min(n, kStageSlots - 1); ++c) DOWN_TRY(fetch(c));
    for (int c = 0; c < n; ++c) {
        const int s = c % kStageSlots;
        const size_t off = (size_t)c * chunk;
        DOWN_TRY(hipEventSynchronize(h->stage_ev[s]));
        pool.copy(to + off, h->stage_out[s], std::min(chunk, bytes - off), &tout[s]);
        const int next = c + kStageSlots - 1;               // its block held chunk c - 1: wait until the workers have emptied it
        if (next < n) {
            CopyPool::wait(&tout[next % kStageSlots]);
            DOWN_TRY(fetch(next));
        }
    }
#undef DOWN_TRY
    return settle(RVLL_OK);
}

// rows of `in` (unit-cube rows if in_is_cube, else theta rows) -> [theta_out], logL, [flags], all host arrays of B rows
int stream_host_batch(rvll_handle* h, const double* in, bool in_is_cube, int64_t B, double* theta_out, double* logL, int32_t* flags)
{
    long long rows = kStreamChunkRows;
    if (const char* e = getenv("RVLL_STREAM_CHUNK")) rows = std::max(256, atoi(e));      // measurement switch
    int rc = rvll_dev_reserve(h, B);
    if (rc) return rc;
    rc = sync_other_lanes(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->compute));
    rc = stream_reserve(h, rows);
    if (rc) return rc;
    const long long D = h->L.ndim;
    const int n = (int)((B + rows - 1) / rows);
    CopyPool& pool = *h->pool;
    CopyPool::Ticket tin[kStageSlots], tout[kStageSlots];
    auto lo_of = [&](int c) { return (long long)c * rows; };
    auto hi_of = [&](int c) { return std::min<long long>(B, (long long)(c + 1) * rows); };
    auto copy_in = [&](int c) {
        if (c < n) pool.copy(h->stage_in[c % kStageSlots], in + lo_of(c) * D, sizeof(double) * (size_t)((hi_of(c) - lo_of(c)) * D), &tin[c % kStageSlots]);
    };
    auto copy_out = [&](int c) {
        const long long lo = lo_of(c), m = hi_of(c) - lo;
        const char* so = static_cast<const char*>(h->stage_out[c % kStageSlots]);
        CopyPool::Ticket* t = &tout[c % kStageSlots];
        if (theta_out) pool.copy(theta_out + lo * D, so, sizeof(double) * (size_t)(m * D), t);
        pool.copy(logL + lo, so + sizeof(double) * (size_t)(rows * D), sizeof(double) * (size_t)m, t);
        if (flags) pool.copy(flags + lo, so + sizeof(double) * (size_t)(rows * (D + 1)), sizeof(int32_t) * (size_t)m, t);
    };
    hipStream_t s_up = h->stream_up, s_down = h->stream_down;
    int lag = 2;    // the calling thread runs this many chunks ahead of the results it waits for (1: the next chunk's commands were
                    // issued only when the last-but-one's results had landed, and the kernels waited for that: 2.3 -> 1.8 ms at 262144 rows)
    if (const char* e = getenv("RVLL_STREAM_LAG")) lag = std::max(1, std::min(2, atoi(e)));   // measurement switch
    // whatever happens, leave with no copy in flight into or out of the caller's arrays and the workers asleep
    auto settle = [&](int code) {
        for (int s = 0; s < kStageSlots; ++s) { CopyPool::wait(&tin[s]); CopyPool::wait(&tout[s]); }
        (void)hipStreamSynchronize(s_up);
        (void)hipStreamSynchronize(h->compute);
        (void)hipStreamSynchronize(s_down);
        pool.busy(false);
        return code;
    };
#define STREAM_TRY(expr) do { const hipError_t err_ = (expr); if (err_ != hipSuccess) { settle(0); HIP_TRY(err_); } } while (0)
    const bool timing = getenv("RVLL_STREAM_TIMING") != nullptr;                          // measurement switch: where the calling thread waits
    double waited[4] = {0., 0., 0., 0.};
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = timing ? now() : 0.;
    pool.busy(true);
    copy_in(0);
    copy_in(1);
    double* dev_in = in_is_cube ? h->d_cube : h->d_theta;
    for (int c = 0; c < n; ++c) {
        const int s = c % kStageSlots;
        const long long lo = lo_of(c), m = hi_of(c) - lo;
        double t0 = timing ? now() : 0.;
        CopyPool::wait(&tin[s]);                            // the chunk's rows are in their pinned block ...
        if (timing) { const double t1 = now(); waited[0] += t1 - t0; t0 = t1; }
        CopyPool::wait(&tout[s]);                           // ... and the results of the chunk that used the block before are out of theirs
        if (timing) { const double t1 = now(); waited[1] += t1 - t0; t0 = t1; }
        // up | kernels | down on three streams, so that chunk c + 1 comes up and chunk c - 1 goes down (the link is full duplex)
        // under the kernels of chunk c
        STREAM_TRY(hipMemcpyAsync(dev_in + lo * D, h->stage_in[s], sizeof(double) * (size_t)(m * D), hipMemcpyHostToDevice, s_up));
        STREAM_TRY(hipEventRecord(h->stage_up[s], s_up));
        STREAM_TRY(hipStreamWaitEvent(h->compute, h->stage_up[s], 0));
        if (in_is_cube) {
            rvll::PriorArgs pa{h->d_cube + lo * D, h->d_theta + lo * D, m, h->L.ndim, h->d_priors, h->d_heavy, h->n_heavy};
            STREAM_TRY(rvll::launch_prior(pa, h->compute));
        }
        rvll::LoglikeArgs a;
        int cu = 0;
        rc = build_args(h, h->d_theta + lo * D, h->d_logL2[0] + lo, h->d_flags2[0] + lo, m, &a, &cu);
        if (rc) return settle(rc);
        STREAM_TRY(launch_form(a, cu, h->compute));
        STREAM_TRY(hipEventRecord(h->stage_done[s], h->compute));
        STREAM_TRY(hipStreamWaitEvent(s_down, h->stage_done[s], 0));
        char* so = static_cast<char*>(h->stage_out[s]);
        hipStream_t st = s_down;
        if (theta_out) STREAM_TRY(hipMemcpyAsync(so, h->d_theta + lo * D, sizeof(double) * (size_t)(m * D), hipMemcpyDeviceToHost, st));
        STREAM_TRY(hipMemcpyAsync(so + sizeof(double) * (size_t)(rows * D), h->d_logL2[0] + lo, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, st));
        if (flags) STREAM_TRY(hipMemcpyAsync(so + sizeof(double) * (size_t)(rows * (D + 1)), h->d_flags2[0] + lo, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost, st));
        STREAM_TRY(hipEventRecord(h->stage_ev[s], st));
        if (timing) { const double t1 = now(); waited[2] += t1 - t0; t0 = t1; }
        if (c >= lag) {
            STREAM_TRY(hipEventSynchronize(h->stage_ev[(c - lag) % kStageSlots]));
            if (timing) { const double t1 = now(); waited[3] += t1 - t0; t0 = t1; }
            copy_out(c - lag);
        }
        copy_in(c + 2);     // (its block was last read by the upload of chunk c - 2, whose event has been waited for)
    }
    for (int c = std::max(0, n - lag); c < n; ++c) {
        STREAM_TRY(hipEventSynchronize(h->stage_ev[c % kStageSlots]));
        copy_out(c);
    }
#undef STREAM_TRY
    h->theta_async = false;
    h->logl_last = 0;
    if (timing) {
        const double t1 = now();
        settle(0);
        fprintf(stderr, "stream %lld rows, %d chunks, %d workers: %.0f us to the last event, %.0f us in all; waited for rows in %.0f, results out %.0f, "
                "issue %.0f, events %.0f us\n", (long long)B, n, pool.size(), t1 - t_start, now() - t_start, waited[0], waited[1], waited[2], waited[3]);
        return RVLL_OK;
    }
    return settle(RVLL_OK);
}

}  // namespace

extern "C" {

int rvll_scalar_server(rvll_handle* h, int32_t enable)
{
    int rc = use_device(h);                   // also stops a running server
    if (rc) return rc;
    if (enable && h->L.ndim > rvll::kServerMaxDim)
        return fail(RVLL_E_UNSUPPORTED, "scalar server supports up to %d parameters", rvll::kServerMaxDim);
    h->srv_enabled = enable != 0;
    return RVLL_OK;
}

int rvll_loglike_batch(rvll_handle* h, const double* theta, int64_t B, double* logL, int32_t* flags)
{
    if (!h) return fail(RVLL_E_INVALID, "null handle");
    if (B == 1 && h->srv_enabled && theta && logL)
        return scalar_call(h, getenv("RVLL_SERVER_NOOP") ? rvll::kServerNoop : rvll::kServerLogLike, theta, logL, flags, nullptr);
    if (B < 0) return fail(RVLL_E_INVALID, "B < 0");
    if (B == 0) return use_device(h);
    if (!theta || !logL) return fail(RVLL_E_INVALID, "theta/logL is null");
    const size_t nin = sizeof(double) * (size_t)B * (size_t)h->L.ndim;
    const size_t nout = (sizeof(double) + sizeof(int32_t)) * (size_t)B;
    // theta rows that fit the pinned block (6898 points at 19 parameters) go there by memcpy and are read by the kernel
    // over PCIe: 10 us less than an upload command at 512 .. 4096 points (RVLL_ZERO_COPY_IN_KB: measurement switch)
    size_t zc_in_max = rvll_handle::kPinBytes;
    if (const char* e = getenv("RVLL_ZERO_COPY_IN_KB")) zc_in_max = std::min((size_t)std::max(0, atoi(e)) * 1024, rvll_handle::kPinBytes);
    if (nin <= zc_in_max && nout <= rvll_handle::kPinBytes) {
        // scalar / small-batch callback: zero-copy.  The kernel reads theta straight from mapped pinned host
        // memory and writes log-L and flags back into it — no copy commands, one launch, one sync.
        int rc = use_device(h);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->compute));
        memcpy(h->pin_in, theta, nin);
        double* out_l = static_cast<double*>(h->pin_out_dev);
        int32_t* out_f = reinterpret_cast<int32_t*>(out_l + B);
        rvll::LoglikeArgs a;
        int cu = 0;
        rc = build_args(h, static_cast<const double*>(h->pin_in_dev), out_l, out_f, B, &a, &cu);
        if (rc) return rc;
        HIP_TRY(launch_form(a, cu, h->compute));
        HIP_TRY(hipStreamSynchronize(h->compute));
        memcpy(logL, h->pin_out, sizeof(double) * (size_t)B);
        if (flags) memcpy(flags, static_cast<char*>(h->pin_out) + sizeof(double) * (size_t)B, sizeof(int32_t) * (size_t)B);
        return RVLL_OK;
    }
    // (theta -> log-L has no large download: the runtime's own staged upload from pageable memory is as fast as ours and the
    //  chunked route below wins at every size, fresh arrays or kept ones — profiles/r03_stream_probe.txt; streamed only by switch)
    if (getenv("RVLL_STREAM_LOGLIKE") && B >= stream_min_points(h)) return stream_host_batch(h, theta, false, B, nullptr, logL, flags);
    int nsplit = B >= kSplitMinPoints ? (int)std::min<long long>(kSplitMaxChunks, std::max<long long>(2, B / kSplitChunkPoints)) : 1;
    if (const char* e = getenv("RVLL_SPLIT")) nsplit = std::max(1, std::min(64, atoi(e)));   // measurement switch
    if (nsplit > 1 && B >= 2 * nsplit) {
        // Large host batch: chunks alternate between two streams.  A copy from pageable memory occupies the
        // calling thread while the runtime stages it, so the upload of chunk i+1 overlaps the kernel of chunk i;
        // all chunks write into the same log-L / flags buffers and come back with one download.
        int rc = rvll_dev_reserve(h, B);
        if (rc) return rc;
        rc = sync_other_lanes(h);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->compute));
        const long long D = h->L.ndim;
        // on an error in the middle of the loop, copies from the caller's buffers may still be in flight on either
        // stream: wait for both before handing the buffers back
        auto settle = [&]() { (void)hipStreamSynchronize(h->lanes[0]); (void)hipStreamSynchronize(h->lanes[1]); };
        // results that fit the pinned block leave the kernels as stores into mapped host memory (as in the small-batch
        // path): no download commands behind the last kernel, one memcpy on the host
        const bool out_pinned = nout <= rvll_handle::kPinBytes && !getenv("RVLL_NO_PINNED_OUT");
        double* out_l = out_pinned ? static_cast<double*>(h->pin_out_dev) : h->d_logL2[0];
        int32_t* out_f = out_pinned ? reinterpret_cast<int32_t*>(static_cast<double*>(h->pin_out_dev) + B) : h->d_flags2[0];
        for (int c = 0; c < nsplit; ++c) {
            const long long lo = B * c / nsplit, hi = B * (c + 1) / nsplit;
            hipStream_t st = h->lanes[c & 1];
            { const hipError_t err_ = hipMemcpyAsync(h->d_theta + lo * D, theta + lo * D, sizeof(double) * (size_t)((hi - lo) * D),
                                                     hipMemcpyHostToDevice, st);
              if (err_ != hipSuccess) { settle(); HIP_TRY(err_); } }
            rvll::LoglikeArgs a;
            int cu = 0;
            rc = build_args(h, h->d_theta + lo * D, out_l + lo, out_f + lo, hi - lo, &a, &cu);
            if (rc) { settle(); return rc; }
            { const hipError_t err_ = launch_form(a, cu, st); if (err_ != hipSuccess) { settle(); HIP_TRY(err_); } }
        }
        HIP_TRY(hipStreamSynchronize(h->lanes[1]));
        h->theta_async = false;
        h->logl_last = 0;
        if (!out_pinned) return rvll_dev_download(h, B, nullptr, logL, flags);
        HIP_TRY(hipStreamSynchronize(h->lanes[0]));
        memcpy(logL, h->pin_out, sizeof(double) * (size_t)B);
        if (flags) memcpy(flags, static_cast<char*>(h->pin_out) + sizeof(double) * (size_t)B, sizeof(int32_t) * (size_t)B);
        return RVLL_OK;
    }
    if (nout <= rvll_handle::kPinBytes && !getenv("RVLL_NO_PINNED_OUT")) {
        // one upload, one launch whose results are stores into mapped pinned host memory, one synchronisation
        int rc = rvll_dev_reserve(h, B);
        if (rc) return rc;
        rc = sync_other_lanes(h);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->compute));     // the pinned block may still be read by an earlier call's copy
        HIP_TRY(hipMemcpyAsync(h->d_theta, theta, nin, hipMemcpyHostToDevice, h->compute));
        double* out_l = static_cast<double*>(h->pin_out_dev);
        rvll::LoglikeArgs a;
        int cu = 0;
        rc = build_args(h, h->d_theta, out_l, reinterpret_cast<int32_t*>(out_l + B), B, &a, &cu);
        if (rc) { (void)hipStreamSynchronize(h->compute); return rc; }
        { const hipError_t err_ = launch_form(a, cu, h->compute); if (err_ != hipSuccess) { (void)hipStreamSynchronize(h->compute); HIP_TRY(err_); } }
        HIP_TRY(hipStreamSynchronize(h->compute));
        h->theta_async = false;
        memcpy(logL, h->pin_out, sizeof(double) * (size_t)B);
        if (flags) memcpy(flags, static_cast<char*>(h->pin_out) + sizeof(double) * (size_t)B, sizeof(int32_t) * (size_t)B);
        return RVLL_OK;
    }
    int rc = rvll_dev_upload_theta(h, theta, B);
    if (rc) return rc;
    rc = rvll_dev_loglike(h, B);
    if (rc) return rc;
    return rvll_dev_download(h, B, nullptr, logL, flags);
}

int rvll_prior_batch(rvll_handle* h, const double* cube, int64_t B, double* theta)
{
    if (!h) return fail(RVLL_E_INVALID, "null handle");
    if (B < 0) return fail(RVLL_E_INVALID, "B < 0");
    if (h && !h->have_priors) return fail(RVLL_E_NOPRIORS, "rvll_set_priors has not been called");
    if (B == 1 && h->srv_enabled && cube && theta) return scalar_call(h, rvll::kServerPrior, cube, nullptr, nullptr, theta);
    if (B == 0) return use_device(h);
    if (!cube || !theta) return fail(RVLL_E_INVALID, "cube/theta is null");
    const size_t nrow = sizeof(double) * (size_t)B * (size_t)h->L.ndim;
    if (nrow <= 384 * 1024 && !getenv("RVLL_NO_PINNED_OUT")) {
        // a vectorized prior callback of up to 2586 points (19 parameters; beyond, the two host memcpys cost more than they save): the prior kernels read the cube from and write
        // theta to mapped pinned host memory — no copy command in either direction
        int rc = use_device(h);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->compute));
        memcpy(h->pin_in, cube, nrow);
        rvll::PriorArgs a{static_cast<const double*>(h->pin_in_dev), static_cast<double*>(h->pin_out_dev), (long long)B,
                          h->L.ndim, h->d_priors, h->d_heavy, h->n_heavy};
        HIP_TRY(rvll::launch_prior(a, h->compute));
        HIP_TRY(hipStreamSynchronize(h->compute));
        memcpy(theta, h->pin_out, nrow);
        return RVLL_OK;
    }
    int rc = rvll_dev_upload_cube(h, cube, B);
    if (rc) return rc;
    rc = rvll_dev_prior(h, B);
    if (rc) return rc;
    return rvll_dev_download(h, B, theta, nullptr, nullptr);
}

int rvll_prior_loglike_batch(rvll_handle* h, const double* cube, int64_t B,
                             double* theta_out, double* logL, int32_t* flags)
{
    if (!h) return fail(RVLL_E_INVALID, "null handle");
    if (B < 0) return fail(RVLL_E_INVALID, "B < 0");
    if (h && !h->have_priors) return fail(RVLL_E_NOPRIORS, "rvll_set_priors has not been called");
    if (B == 0) return use_device(h);
    if (!cube || !logL) return fail(RVLL_E_INVALID, "cube/logL is null");
    if (B == 1 && h->srv_enabled)             // the scalar pair prior(cube) + loglike(theta) as ONE request of the server
        return scalar_call(h, rvll::kServerPriorLogLike, cube, logL, flags, theta_out);
    const size_t nin = sizeof(double) * (size_t)B * (size_t)h->L.ndim;
    const size_t nout = (sizeof(double) + sizeof(int32_t)) * (size_t)B;
    if (nin + nout + 16 <= rvll_handle::kPinBytes && h->all_direct && !getenv("RVLL_NO_PINNED_OUT")) {
        // a sampler's proposal round (up to 6393 points at 19 parameters): the fused kernel reads the cube from mapped
        // pinned host memory and stores theta, log-L and flags into it — no copy command in either direction
        int rc = rvll_dev_reserve(h, B);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->compute));
        memcpy(h->pin_in, cube, nin);
        rc = sync_other_lanes(h);
        if (rc) return rc;
        double* out_l = static_cast<double*>(h->pin_out_dev);
        int32_t* out_f = reinterpret_cast<int32_t*>(out_l + B);
        rvll::LoglikeArgs a;
        rc = build_args(h, h->d_theta, out_l, out_f, B, &a);
        if (rc) return rc;
        // (flags end on a multiple of 4 bytes; theta starts on the next multiple of 16)
        const size_t theta_off = (nout + 15) & ~(size_t)15;
        if (theta_off + nin > rvll_handle::kPinBytes) return fail(RVLL_E_INVALID, "pinned block too small");
        make_fused(h, static_cast<const double*>(h->pin_in_dev),
                   reinterpret_cast<double*>(static_cast<char*>(h->pin_out_dev) + theta_off), &a);
        *h->pin_defer = 0;
        HIP_TRY(rvll::launch_prior_loglike(a, h->compute));
        char* host_out = static_cast<char*>(h->pin_out);
        HIP_TRY(hipStreamSynchronize(h->compute));
        if (__atomic_load_n(h->pin_defer, __ATOMIC_ACQUIRE) != 0) {
            // an element fell outside the slim stage's tables (rvll_tile.h): the full prior kernels take the batch
            *h->pin_defer = 0;
            rc = rvll_dev_upload_cube(h, cube, B);
            if (rc) return rc;
            rc = rvll_dev_prior(h, B);
            if (rc) return rc;
            h->logl_cur = 0;
            rc = rvll_dev_loglike(h, B);
            if (rc) return rc;
            return rvll_dev_download(h, B, theta_out, logL, flags);
        }
        memcpy(logL, host_out, sizeof(double) * (size_t)B);
        if (flags) memcpy(flags, host_out + sizeof(double) * (size_t)B, sizeof(int32_t) * (size_t)B);
        if (theta_out) memcpy(theta_out, host_out + theta_off, nin);
        return RVLL_OK;
    }
    if (B >= stream_min_points(h)) return stream_host_batch(h, cube, true, B, theta_out, logL, flags);
    // measured (profiles/r01_split_probe.txt): two halves help from 16384 points (+15 %) to 65536 (+35 %); more chunks
    // lose to the per-copy fixed costs, and at 262144 points the large pageable downloads on two streams collapse
    int nsplit = (B >= kSplitMinPoints && B <= 131072) ? 2 : 1;
    if (const char* e = getenv("RVLL_SPLIT")) nsplit = std::max(1, std::min(64, atoi(e)));   // measurement switch
    if (nsplit > 1 && B >= 2 * nsplit) {
        // Large host batch: chunks alternate between two streams, software-pipelined by one chunk.  Copies from /
        // to pageable memory occupy the calling thread, so while it stages the upload of chunk c+1 and the
        // download of chunk c-1, the GPU runs the kernels of chunk c.
        int rc = rvll_dev_reserve(h, B);
        if (rc) return rc;
        rc = sync_other_lanes(h);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->compute));
        const long long D = h->L.ndim;
        auto bounds = [&](int c, long long* lo, long long* hi) { *lo = B * c / nsplit; *hi = B * (c + 1) / nsplit; };
        // an error in the middle leaves copies to / from the caller's buffers in flight: settle both streams first
        auto settle = [&](int code) { (void)hipStreamSynchronize(h->lanes[0]); (void)hipStreamSynchronize(h->lanes[1]); return code; };
#define SPLIT_TRY(expr) do { const hipError_t err_ = (expr); if (err_ != hipSuccess) { settle(0); HIP_TRY(err_); } } while (0)
        auto fetch = [&](int c) -> int {
            long long lo, hi;
            bounds(c, &lo, &hi);
            hipStream_t st = h->lanes[c & 1];
            if (theta_out) SPLIT_TRY(hipMemcpyAsync(theta_out + lo * D, h->d_theta + lo * D, sizeof(double) * (size_t)((hi - lo) * D), hipMemcpyDeviceToHost, st));
            SPLIT_TRY(hipMemcpyAsync(logL + lo, h->d_logL2[0] + lo, sizeof(double) * (size_t)(hi - lo), hipMemcpyDeviceToHost, st));
            if (flags) SPLIT_TRY(hipMemcpyAsync(flags + lo, h->d_flags2[0] + lo, sizeof(int32_t) * (size_t)(hi - lo), hipMemcpyDeviceToHost, st));
            return RVLL_OK;
        };
        for (int c = 0; c < nsplit; ++c) {
            long long lo, hi;
            bounds(c, &lo, &hi);
            hipStream_t st = h->lanes[c & 1];
            SPLIT_TRY(hipMemcpyAsync(h->d_cube + lo * D, cube + lo * D, sizeof(double) * (size_t)((hi - lo) * D), hipMemcpyHostToDevice, st));
            rvll::PriorArgs pa{h->d_cube + lo * D, h->d_theta + lo * D, hi - lo, h->L.ndim, h->d_priors, h->d_heavy, h->n_heavy};
            SPLIT_TRY(rvll::launch_prior(pa, st));
            rvll::LoglikeArgs a;
            int cu = 0;
            rc = build_args(h, h->d_theta + lo * D, h->d_logL2[0] + lo, h->d_flags2[0] + lo, hi - lo, &a, &cu);
            if (rc) return settle(rc);
            SPLIT_TRY(launch_form(a, cu, st));
            if (c > 0) { rc = fetch(c - 1); if (rc) return settle(rc); }
        }
        rc = fetch(nsplit - 1);
        if (rc) return settle(rc);
#undef SPLIT_TRY
        HIP_TRY(hipStreamSynchronize(h->compute));
        HIP_TRY(hipStreamSynchronize(h->lanes[1]));
        h->theta_async = false;
        h->logl_last = 0;
        return RVLL_OK;
    }
    int rc = rvll_dev_upload_cube(h, cube, B);
    if (rc) return rc;
    // measured (profiles/r01_fused_probe.txt): one launch saves ~1 us up to a few thousand points; beyond that
    // the separate prior kernels win by ~5 % because their work spreads over the whole chip instead of
    // running as a short serial prologue of every log-L workgroup
    rc = rvll_dev_prior_loglike(h, B);
    if (rc) return rc;
    return rvll_dev_download(h, B, theta_out, logL, flags);
}


int rvll_set_wander_exact(rvll_handle* h, int32_t on)
{
    if (!h) return fail(RVLL_E_INVALID, "null handle");
    h->wander_exact = on != 0;
    return RVLL_OK;
}

int rvll_set_slim_table_range(rvll_handle* h, double umax)
{
    if (!h) return fail(RVLL_E_INVALID, "null handle");
    if (!(umax >= 0.)) return fail(RVLL_E_INVALID, "umax must be >= 0");
    h->slim_umax = std::min(umax, rvll::prior_table_umax());
    return RVLL_OK;
}

// ---- Keplerian curves (post-processing helper) ------------------------------------------------------
int rvll_kep_rv_batch(rvll_handle* h, const double* theta, int64_t B, const double* times, int32_t n_times,
                      uint32_t include_mask, double* out)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (B < 0 || n_times < 0) return fail(RVLL_E_INVALID, "negative size");
    if (B == 0 || n_times == 0) return RVLL_OK;
    if (!theta || !times || !out) return fail(RVLL_E_INVALID, "null buffer");
    rc = rvll_dev_upload_theta(h, theta, B);
    if (rc) return rc;
    double *d_times = nullptr, *d_out = nullptr;
    int status = RVLL_OK;
    hipError_t e = hipMalloc(&d_times, sizeof(double) * (size_t)n_times);
    if (e == hipSuccess) e = hipMalloc(&d_out, sizeof(double) * (size_t)B * (size_t)n_times);
    if (e == hipSuccess) e = hipMemcpyAsync(d_times, times, sizeof(double) * (size_t)n_times, hipMemcpyHostToDevice, h->compute);
    if (e == hipSuccess) {
        rvll::LoglikeArgs a;
        status = build_args(h, h->d_theta, h->d_logL2[0], h->d_flags2[0], B, &a);
        if (status == RVLL_OK) e = rvll::launch_keprv(a, d_times, n_times, include_mask, d_out, h->compute);
    }
    if (status == RVLL_OK && e == hipSuccess)
        e = hipMemcpyAsync(out, d_out, sizeof(double) * (size_t)B * (size_t)n_times, hipMemcpyDeviceToHost, h->compute);
    if (status == RVLL_OK && e == hipSuccess) e = hipStreamSynchronize(h->compute);
    if (status == RVLL_OK && e != hipSuccess) status = fail(RVLL_E_HIP, "kep_rv_batch: %s", hipGetErrorString(e));
    (void)hipStreamSynchronize(h->compute);
    dev_free(d_times); dev_free(d_out);
    return status;
}

// ---- diagnostics ---------------------------------------------------------------------
int rvll_debug_eval(rvll_handle* h, int32_t op, const double* x, const double* y, int64_t n, double* out)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!x || !out || n < 1) return fail(RVLL_E_INVALID, "bad debug_eval arguments");
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    const size_t nb = sizeof(double) * (size_t)n;
    int status = RVLL_OK;
    hipError_t e = hipMalloc(&dx, nb);
    if (e == hipSuccess) e = hipMalloc(&dout, nb);
    if (e == hipSuccess && y) e = hipMalloc(&dy, nb);
    if (e == hipSuccess) e = hipMemcpy(dx, x, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess && y) e = hipMemcpy(dy, y, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rvll::launch_debug_eval(op, dx, dy, (long long)n, dout, h->compute);
    if (e == hipSuccess) e = hipStreamSynchronize(h->compute);
    if (e == hipSuccess) e = hipMemcpy(out, dout, nb, hipMemcpyDeviceToHost);
    if (e != hipSuccess) status = fail(RVLL_E_HIP, "debug_eval: %s", hipGetErrorString(e));
    dev_free(dx); dev_free(dy); dev_free(dout);
    return status;
}

}  // extern "C"

// ---- the helpers the other host units call (rvll_host.h) ----
namespace rvll {
namespace host {
int use_device(rvll_handle* h) { return ::use_device(h); }
int sync_other_lanes(rvll_handle* h) { return ::sync_other_lanes(h); }
int build_args(rvll_handle* h, const double* d_theta, double* d_logL, int32_t* d_flags, long long B, rvll::LoglikeArgs* out, int* cu_grid)
{
    return ::build_args(h, d_theta, d_logL, d_flags, B, out, cu_grid);
}
void make_fused(const rvll_handle* h, const double* d_cube, double* d_theta_out, rvll::LoglikeArgs* a) { ::make_fused(h, d_cube, d_theta_out, a); }
int download_rows(rvll_handle* h, void* dst, const void* src_dev, size_t bytes) { return ::download_rows(h, dst, src_dev, bytes); }
}  // namespace host
}  // namespace rvll
