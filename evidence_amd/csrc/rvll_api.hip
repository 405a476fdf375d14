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

// the C-ABI entry points are the only exported symbols (built with -fvisibility=hidden)
#pragma GCC visibility push(default)
#include "rvll.h"
#pragma GCC visibility pop
#include "rvll_kernels.h"
#include "rvll_copypool.h"

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

namespace {

#define HIP_TRY(expr)                                                               \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess)                                                       \
            return fail(e_ == hipErrorOutOfMemory ? RVLL_E_NOMEM : RVLL_E_HIP,      \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                        __FILE__, __LINE__);                                        \
    } while (0)

// ---- RCCL, resolved at run time so the library loads on boxes without it ----
struct Id128 { char bytes[RVLL_COMM_ID_BYTES]; };   // ncclUniqueId, passed by value
static_assert(sizeof(Id128) == 128, "ncclUniqueId is 128 bytes");
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommSplit)(void*, int, int, void**, void*) = nullptr;     // optional (second pipeline lane)
    const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclFloat64 = 8;   // ncclDouble / ncclFloat64 in rccl.h

Rccl g_rccl;

std::string g_rccl_path;      // where the loaded librccl lives (dladdr)
int g_rccl_version = 0;       // ncclGetVersion

std::string lib_path_of(const void* symbol)
{
    Dl_info info;
    return (symbol && dladdr(symbol, &info) && info.dli_fname) ? std::string(info.dli_fname) : std::string("?");
}

// Which librccl: the one next to the HIP runtime this library is linked against — the ROCm it was built and tested
// with — not whatever a soname lookup finds first (a process that imported torch first would get torch's bundled
// RCCL and HIP runtime).  Order: RVLL_RCCL_PATH, the directory of the loaded libamdhip64, /opt/rocm/lib, sonames.
int rccl_load()
{
    if (g_rccl.lib) return RVLL_OK;
    std::vector<std::string> names;
    if (const char* e = getenv("RVLL_RCCL_PATH")) names.push_back(e);
    const std::string hip = lib_path_of(reinterpret_cast<const void*>(&hipGetDeviceCount));
    const size_t slash = hip.rfind('/');
    if (slash != std::string::npos) {
        names.push_back(hip.substr(0, slash) + "/librccl.so.1");
        names.push_back(hip.substr(0, slash) + "/librccl.so");
    }
    names.push_back("/opt/rocm/lib/librccl.so.1");
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    void* lib = nullptr;
    std::string tried;
    for (const std::string& n : names) {
        lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
        tried += n + " ";
    }
    if (!lib) return fail(RVLL_E_RCCL, "cannot dlopen librccl (tried %s): %s", tried.c_str(), dlerror());
    Rccl r;
    r.lib = lib;
    r.GetUniqueId    = (int (*)(void*))dlsym(lib, "ncclGetUniqueId");
    r.CommInitRank   = (int (*)(void**, int, Id128, int))dlsym(lib, "ncclCommInitRank");
    r.AllGather      = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(lib, "ncclAllGather");
    r.CommDestroy    = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
    r.CommSplit      = (int (*)(void*, int, int, void**, void*))dlsym(lib, "ncclCommSplit");
    r.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString) {
        dlclose(lib);
        return fail(RVLL_E_RCCL, "librccl lacks an expected nccl* symbol");
    }
    g_rccl = r;
    g_rccl_path = lib_path_of(reinterpret_cast<const void*>(r.AllGather));
    if (auto ver = (int (*)(int*))dlsym(lib, "ncclGetVersion")) (void)ver(&g_rccl_version);
    return RVLL_OK;
}

#define RCCL_TRY(expr)                                                              \
    do {                                                                            \
        int r_ = (expr);                                                            \
        if (r_ != 0)                                                                \
            return fail(RVLL_E_RCCL, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

template <typename T>
void dev_free(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

}  // namespace

constexpr int kMaxLanes = 4;
// rvll_loglike_batch: host batches from kSplitMinPoints on go up in overlapped chunks of about kSplitChunkPoints
constexpr long long kSplitMinPoints = 16384, kSplitChunkPoints = 16384, kSplitMaxChunks = 8;   // profiles/r02_split_probe.txt
constexpr int kWalkWords = 14;                // counters of the walk kernel: calls, tile slots, 4 phase bins + workgroups + longest life + 4 tile phases (diagnostic build), the queue
constexpr size_t kStreamMinBytes = 24u << 20;   // cube -> theta -> log-L host batches whose rows take this much are streamed (stream_host_batch)
constexpr long long kFusedMaxPoints = 4096;   // rvll_prior_loglike_batch: one launch up to here, two beyond

using rvll::CopyPool;
constexpr int kStageSlots = 4;              // pinned staging blocks each way of a streamed host batch (stream_host_batch)

struct rvll_handle {
    int device = 0;
    hipStream_t compute = nullptr;          // lane 0: every single-GPU call runs here
    hipStream_t lanes[kMaxLanes] = {};      // lanes[0] == compute; further pipeline lanes of the multi-GPU step (rvll_allgather_logl)
    int      nlanes_dev = 2;                // lanes rvll_dev_flip_lane cycles through

    // layout (host mirror, then device copies)
    rvll_layout L{};
    std::vector<rvll_planet> planets;
    std::vector<rvll_inst>   insts;
    std::vector<rvll_slot>   linslots;
    rvll_planet* d_planets = nullptr;
    rvll_inst*   d_insts = nullptr;
    rvll_slot*   d_linslots = nullptr;
    double*      d_layblob = nullptr;       // planets, insts, linslots, drift[4], tref back to back: staged into LDS by the kernels
    int          form_override = 0;         // RVLL_FORM: 1 = tile kernel only, 2 = CU-wide kernel wherever it fits

    // resident epoch table
    int Ne = 0;
    double*  d_t = nullptr;
    double*  d_y = nullptr;
    double*  d_s2 = nullptr;
    int32_t* d_inst = nullptr;
    double*  d_linpar = nullptr;
    double   cte = 0.;
    double   tmin = 0., tmax = 0.;              // range of the epoch times

    // priors
    bool have_priors = false;
    rvll_prior* d_priors = nullptr;
    int32_t* d_heavy = nullptr;
    int n_heavy = 0;
    std::vector<double*> d_tables;
    std::vector<double> table_err;              // measured quintic-interpolant error per parameter (NaN: no table)
    std::vector<int> table_direct;              // per parameter: evaluated by verified interpolation alone
    bool all_direct = true;                     // every Beta / Gamma prior has a verified table: the slim prior stage applies
    double slim_umax = 0.;                      // |logit q| range the slim stage takes (rvll_set_slim_table_range; default: the table's)
    int* pin_defer = nullptr;                   // mapped pinned word the slim stage sets when it defers an element
    int* pin_defer_dev = nullptr;
    long long fused_pending = 0;                // rows of a one-launch cube -> log-L batch whose defer word has not been looked at yet

    // batch buffers
    long long cap = 0;
    double*  d_theta = nullptr;
    double*  d_cube = nullptr;
    double*  d_logL2[kMaxLanes] = {};           // one log-L buffer per pipeline lane
    int      logl_cur = 0;                      // lane the next device-resident launch uses
    int      logl_last = 0;                     // lane the last launch used (download source)
    bool     theta_async = false;               // theta was (re)written asynchronously on lane 0's stream
    bool     pipelined = false;                 // launches alternate lanes: two are in flight, no kernel-end tail
    int32_t* d_flags2[kMaxLanes] = {};          // per lane, like log-L

    // pinned host staging for small transfers (scalar / small-batch callbacks)
    static constexpr size_t kPinBytes = 1u << 20;
    void* pin_in = nullptr;
    void* pin_out = nullptr;
    void* pin_in_dev = nullptr;      // device-visible aliases of the two pinned buffers (zero-copy path)
    void* pin_out_dev = nullptr;

    // large host batches (stream_host_batch): worker threads for the host's copies, pinned staging blocks, one event per block
    CopyPool* pool = nullptr;
    void* stage_in[kStageSlots] = {};
    void* stage_out[kStageSlots] = {};
    size_t stage_in_bytes = 0, stage_out_bytes = 0;
    hipEvent_t stage_ev[kStageSlots] = {};      // a chunk's results are in its pinned block
    hipEvent_t stage_up[kStageSlots] = {}, stage_done[kStageSlots] = {};   // ... its rows are on the device / its kernels have run
    hipStream_t stream_up = nullptr, stream_down = nullptr;                // the two copy directions, beside lane 0's kernels (stream_reserve)

    // scalar-call server (rvll_scalar_server): persistent one-workgroup kernel + host-coherent control block
    rvll::ServerCtl* srv = nullptr;             // pinned, mapped, coherent
    rvll::ServerCtl* srv_dev = nullptr;         // its device address
    hipStream_t srv_stream = nullptr;
    bool srv_enabled = false, srv_running = false, srv_dead = false;
    unsigned long long srv_seq = 0;             // request numbers (low 32 bits travel)
    unsigned long long srv_last = 0;            // the last request word that was answered
    double*  d_srv_out = nullptr;               // device-local {logL, flags} the server's tile writes
    unsigned long long srv_idle_ticks = 500000; // 5 ms of the 100 MHz constant clock

    // device-resident slice-sampling walk (rvll_slice_walk)
    long long walk_cap = 0;                     // rows
    double *d_walk_u = nullptr, *d_walk_theta = nullptr, *d_walk_logl = nullptr, *d_walk_chol = nullptr;
    int32_t* d_walk_wrapped = nullptr;
    unsigned long long* d_walk_ncalls = nullptr;
    int32_t *d_walk_steps = nullptr, *d_walk_wid = nullptr, *d_walk_start = nullptr;   // [walk_cap] each
    int32_t *d_walk_cost = nullptr, *d_walk_order = nullptr;                            // [walk_cap] each (two-part walks)
    int walk_spec = 4;                          // candidates a walker may evaluate ahead per iteration (rvll_set_walk_speculation)
    long long walk_evaluated = 0;               // tile slots the last rvll_slice_walk evaluated (>= its ncalls)
    unsigned long long walk_phase[6] = {};      // diagnostic build (make walktrace): 100 MHz ticks per phase, summed over workgroups; workgroups

    // device-resident live set (rvll_live_*): nested sampling's live points, and the points that died, stay in HBM
    long long live_n = 0, live_cap = 0;
    double *d_live_u = nullptr, *d_live_theta = nullptr, *d_live_logl = nullptr;
    int32_t* d_live_idx = nullptr;              // [2 * live_cap] order, then start rows, of the current step
    double *d_live_mom = nullptr;               // scratch, mean, covariance of the whitening
    long long dead_n = 0, dead_cap = 0;
    double *d_dead_theta = nullptr, *d_dead_logl = nullptr;

    hipEvent_t marks[2] = {nullptr, nullptr};   // rvll_dev_mark: HIP events on lane 0's stream

    // geometry
    int pb_override = 0;
    std::unordered_map<long long, int> geo;     // batch size -> points per workgroup chosen for it
    std::unordered_map<size_t, int> occ_by_lds; // dynamic LDS bytes -> resident workgroups per CU
    int chunk_items = rvll::kTileWindow;
    int n_cu = 256;

    // multi-GPU
    void* nccl_comm[kMaxLanes] = {};            // one communicator per lane (all but the first by ncclCommSplit)
    int nranks = 1, rank = 0;
    int nlanes = 1;
    long long gather_cap = 0;
    double* d_gather2[kMaxLanes] = {};
    double *d_gather_host_in = nullptr, *d_gather_host_out = nullptr;   // rvll_allgather_host: grow-only staging
    long long gather_host_cap = 0;              // in doubles
    double* d_gather_theta = nullptr;           // [nranks * B_local, D] (rvll_allgather_theta)
    long long gather_theta_cap = 0;             // in rows
    int gather_last = 0;
};

namespace {

hipStream_t lane_stream(const rvll_handle* h, int lane) { return h->lanes[lane]; }

int sync_other_lanes(rvll_handle* h)
{
    for (int l = 1; l < kMaxLanes; ++l)
        if (h->lanes[l]) HIP_TRY(hipStreamSynchronize(h->lanes[l]));
    return RVLL_OK;
}

// Ask a running scalar-call server to leave and wait for it.  Every entry point other than the scalar call
// itself goes through use_device(), so the persistent kernel never coexists with allocations, frees or
// collectives of its own handle (hipFree and friends synchronise the whole device).
int server_stop(rvll_handle* h)
{
    if (!h->srv_running || h->srv_dead) return RVLL_OK;
    const unsigned long long request = ((unsigned long long)rvll::kServerQuit << 32) | (unsigned)++h->srv_seq;
    __atomic_store_n(&h->srv->request, request, __ATOMIC_RELEASE);
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
constexpr size_t kDownloadStagedMin = 32u << 20, kDeadStagedMin = 8u << 20;

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
    for (auto& c : h->nccl_comm) if (c && g_rccl.lib) { (void)g_rccl.CommDestroy(c); c = nullptr; }
    free_priors(h);
    dev_free(h->d_theta); dev_free(h->d_cube);
    for (int l = 0; l < kMaxLanes; ++l) { dev_free(h->d_logL2[l]); dev_free(h->d_flags2[l]); dev_free(h->d_gather2[l]); }
    dev_free(h->d_gather_theta); dev_free(h->d_gather_host_in); dev_free(h->d_gather_host_out);
    dev_free(h->d_live_u); dev_free(h->d_live_theta); dev_free(h->d_live_logl); dev_free(h->d_live_idx); dev_free(h->d_live_mom);
    dev_free(h->d_dead_theta); dev_free(h->d_dead_logl);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->pin_defer) (void)hipHostFree(h->pin_defer);
    stream_free(h);
    dev_free(h->d_walk_steps); dev_free(h->d_walk_wid); dev_free(h->d_walk_start);
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
    memcpy(c->theta, theta, sizeof(double) * (size_t)h->L.ndim);
    const unsigned number = (unsigned)++h->srv_seq;
    const unsigned long long request = ((unsigned long long)op << 32) | number;
    __atomic_store_n(&c->request, request, __ATOMIC_RELEASE);
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
            __atomic_store_n(&c->request, quit, __ATOMIC_RELEASE);
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

// ---- device-resident slice-sampling walk ---------------------------------------------------------------
}   // extern "C"

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
            return fail(RVLL_E_UNSUPPORTED, "%d parameters exceed the walk kernel's LDS budget", a->D);
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
                if (!fits(slots)) return fail(RVLL_E_UNSUPPORTED, "the wide walk does not fit %lld rows per workgroup", (long long)rows);
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
    if (!h->have_priors) return fail(RVLL_E_NOPRIORS, "rvll_set_priors has not been called");
    if (K < 0 || nsteps < 0) return fail(RVLL_E_INVALID, "negative size");
    if (max_rounds < 1 || max_rounds > 4096 || nsteps >= (1 << 18) || K >= (1LL << 31) || walker_base < 0 ||
        walker_base + K >= (1LL << 32))
        return fail(RVLL_E_INVALID, "nsteps / max_rounds / K / walker_base out of range");
    if (h->L.ndim < 1) return fail(RVLL_E_INVALID, "no free parameter to walk in");
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
    if (!cube || !theta || !logl || !chol) return fail(RVLL_E_INVALID, "null buffer");
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
    if (!h->have_priors) return fail(RVLL_E_NOPRIORS, "rvll_set_priors has not been called");
    if (N < 1 || N >= (1LL << 31) || !cube) return fail(RVLL_E_INVALID, "rvll_live_init: bad arguments");
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
    if (N < 1) return fail(RVLL_E_INVALID, "rvll_live_init has not been called");
    if (!order || !start || !logl_new || kdead < 1 || kdead >= N) return fail(RVLL_E_INVALID, "rvll_live_step: bad arguments");
    rc = walk_check_args(h, kdead, nsteps, max_rounds, walker_base);
    if (rc) return rc;
    for (int64_t i = 0; i < N; ++i)
        if (order[i] < 0 || order[i] >= N) return fail(RVLL_E_INVALID, "rvll_live_step: order[%lld] out of range", (long long)i);
    for (int64_t i = 0; i < kdead; ++i)
        if (start[i] < 0 || start[i] >= N) return fail(RVLL_E_INVALID, "rvll_live_step: start[%lld] out of range", (long long)i);
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
                    if (!(sum > 0.)) return fail(RVLL_E_INVALID, "rvll_live_step: the live points' covariance is not positive definite");
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
    if (h->live_n < 1) return fail(RVLL_E_INVALID, "rvll_live_init has not been called");
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
    if (!n_dead) return fail(RVLL_E_INVALID, "n_dead is null");
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
    if (!h) return fail(RVLL_E_INVALID, "null handle");
    if (max_ahead < 1 || max_ahead > rvll::kMaxPointsPerBlock)
        return fail(RVLL_E_INVALID, "max_ahead must be in [1, %d]", rvll::kMaxPointsPerBlock);
    h->walk_spec = max_ahead;
    return RVLL_OK;
}

int rvll_slice_walk_evaluated(rvll_handle* h, int64_t* evaluated)
{
    if (!h || !evaluated) return fail(RVLL_E_INVALID, "null argument");
    *evaluated = h->walk_evaluated;
    return RVLL_OK;
}

int rvll_slice_walk_phases(rvll_handle* h, uint64_t out[6])
{
    if (!h || !out) return fail(RVLL_E_INVALID, "null argument");
    for (int k = 0; k < 6; ++k) out[k] = h->walk_phase[k];
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

// ---- multi-GPU ---------------------------------------------------------------------
int rvll_comm_unique_id(unsigned char id[RVLL_COMM_ID_BYTES])
{
    if (!id) return fail(RVLL_E_INVALID, "id is null");
    int rc = rccl_load();
    if (rc) return rc;
    Id128 u;
    memset(&u, 0, sizeof u);
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, u.bytes, RVLL_COMM_ID_BYTES);
    return RVLL_OK;
}

int rvll_runtime_info(char* buf, int32_t buflen)
{
    if (!buf || buflen < 1) return fail(RVLL_E_INVALID, "bad buffer");
    int hip_rt = 0, hip_drv = 0;
    (void)hipRuntimeGetVersion(&hip_rt);
    (void)hipDriverGetVersion(&hip_drv);
    snprintf(buf, (size_t)buflen,
             "{\"hip_runtime_version\": %d, \"hip_driver_version\": %d, \"libamdhip64\": \"%s\", "
             "\"librccl\": \"%s\", \"rccl_version\": %d, \"librvll\": \"%s\"}",
             hip_rt, hip_drv, lib_path_of(reinterpret_cast<const void*>(&hipGetDeviceCount)).c_str(),
             g_rccl.lib ? g_rccl_path.c_str() : "not loaded", g_rccl_version,
             lib_path_of(reinterpret_cast<const void*>(&rvll_runtime_info)).c_str());
    return RVLL_OK;
}

// One communicator, one pipeline lane: the gather of a step runs in-stream behind its kernel.  Further lanes are
// added by rvll_comm_add_lanes once the caller has seen a gathered step complete on this one.
int rvll_comm_init(rvll_handle* h, const unsigned char id[RVLL_COMM_ID_BYTES], int32_t nranks, int32_t rank)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return fail(RVLL_E_INVALID, "bad comm arguments");
    rc = rccl_load();
    if (rc) return rc;
    rc = rvll_comm_destroy(h);
    if (rc) return rc;
    Id128 u;
    memcpy(u.bytes, id, RVLL_COMM_ID_BYTES);
    RCCL_TRY(g_rccl.CommInitRank(&h->nccl_comm[0], nranks, u, rank));
    h->nranks = nranks;
    h->rank = rank;
    h->nlanes = 1;
    h->logl_cur = 0;
    return RVLL_OK;
}

// Further pipeline lanes: one communicator each (collectives of ONE communicator must not run concurrently on two
// streams), derived collectively from the first by ncclCommSplit — every rank must make this call.  Returns in
// *have how many lanes THIS rank now holds (<= want); the ranks must then agree on the minimum over all of them
// (out of band) and call rvll_comm_set_lanes with it, so that every rank cycles through the same communicators:
// a rank with fewer lanes than its peers would issue its gathers on mismatched communicators and hang them all.
int rvll_comm_add_lanes(rvll_handle* h, int32_t want, int32_t* have)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->nccl_comm[0]) return fail(RVLL_E_RCCL, "rvll_comm_init has not been called");
    want = std::max(1, std::min(kMaxLanes, (int)want));
    int got = 1;
    for (int l = 1; l < kMaxLanes; ++l) if (h->nccl_comm[l]) got = l + 1; else break;
    for (int l = got; l < want; ++l) {
        if (!g_rccl.CommSplit || g_rccl.CommSplit(h->nccl_comm[0], 0, h->rank, &h->nccl_comm[l], nullptr) != 0 || !h->nccl_comm[l]) {
            h->nccl_comm[l] = nullptr;
            break;
        }
        got = l + 1;
    }
    if (have) *have = got;
    return RVLL_OK;
}

int rvll_comm_set_lanes(rvll_handle* h, int32_t nlanes)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->nccl_comm[0]) return fail(RVLL_E_RCCL, "rvll_comm_init has not been called");
    if (nlanes < 1 || nlanes > kMaxLanes) return fail(RVLL_E_INVALID, "nlanes out of range");
    for (int l = 0; l < nlanes; ++l)
        if (!h->nccl_comm[l]) return fail(RVLL_E_INVALID, "lane %d has no communicator on this rank", l);
    HIP_TRY(hipStreamSynchronize(h->compute));
    rc = sync_other_lanes(h);
    if (rc) return rc;
    if (nlanes > h->nlanes && h->gather_cap > 0) {          // gather buffers of the new lanes
        for (int l = h->nlanes; l < nlanes; ++l)
            if (!h->d_gather2[l]) HIP_TRY(hipMalloc(&h->d_gather2[l], sizeof(double) * (size_t)h->gather_cap));
    }
    h->nlanes = nlanes;
    h->logl_cur = 0;
    return RVLL_OK;
}

// One multi-GPU step is  rvll_dev_loglike(B_local) ; rvll_allgather_logl(B_local).  Steps alternate between two
// pipeline LANES, each with its own stream, communicator, log-L and gather buffer: the gather of step k runs
// in-stream right behind kernel k on lane (k mod 2) while kernel k+1 runs on the other lane, so collective
// latency hides behind compute without any cross-stream event (measured on MI355X: event record + stream wait
// pairs cost ~10 us per step, an in-stream gather ~2 us; scripts/comm_overhead_probe.py).
int rvll_allgather_logl(rvll_handle* h, int64_t B_local)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->nccl_comm[0]) return fail(RVLL_E_RCCL, "rvll_comm_init has not been called");
    if (B_local < 1 || B_local > h->cap) return fail(RVLL_E_INVALID, "B_local %lld outside reserved capacity %lld", (long long)B_local, h->cap);
    const long long total = (long long)B_local * h->nranks;
    if (total > h->gather_cap) {
        HIP_TRY(hipStreamSynchronize(h->compute));
        rc = sync_other_lanes(h);
        if (rc) return rc;
        for (int l = 0; l < kMaxLanes; ++l) dev_free(h->d_gather2[l]);
        h->gather_cap = 0;
        for (int l = 0; l < h->nlanes; ++l) HIP_TRY(hipMalloc(&h->d_gather2[l], sizeof(double) * (size_t)total));
        h->gather_cap = total;
    }
    const int lane = h->logl_last;            // the lane whose kernel just wrote its log-L
    RCCL_TRY(g_rccl.AllGather(h->d_logL2[lane], h->d_gather2[lane], (size_t)B_local, kNcclFloat64,
                              h->nccl_comm[lane], lane_stream(h, lane)));
    h->gather_last = lane;
    if (h->nlanes > 1) h->logl_cur = (lane + 1) % h->nlanes;   // the next step runs on the next lane
    return RVLL_OK;
}

int rvll_allgather_theta(rvll_handle* h, int64_t B_local)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->nccl_comm[0]) return fail(RVLL_E_RCCL, "rvll_comm_init has not been called");
    if (B_local < 1 || B_local > h->cap) return fail(RVLL_E_INVALID, "B_local %lld outside reserved capacity %lld", (long long)B_local, h->cap);
    const long long total = (long long)B_local * h->nranks;
    const size_t D = (size_t)std::max(1, h->L.ndim);
    if (total > h->gather_theta_cap) {
        HIP_TRY(hipStreamSynchronize(h->compute));
        dev_free(h->d_gather_theta);
        h->gather_theta_cap = 0;
        HIP_TRY(hipMalloc(&h->d_gather_theta, sizeof(double) * D * (size_t)total));
        h->gather_theta_cap = total;
    }
    // theta is written on lane 0's stream (upload or prior kernel); the gather queues behind it there, on
    // lane 0's communicator — the same stream and communicator lane 0's log-L gathers use, so the two never
    // run concurrently on one communicator
    RCCL_TRY(g_rccl.AllGather(h->d_theta, h->d_gather_theta, (size_t)B_local * D, kNcclFloat64,
                              h->nccl_comm[0], h->compute));
    return RVLL_OK;
}

// All-gather of a small host buffer (the sampler's sharded host state: walk end points, call counts): n_local
// doubles per rank go up, are gathered on the device by RCCL on lane 0's communicator and stream, and nranks *
// n_local come back, rank-major.  Synchronous.
int rvll_allgather_host(rvll_handle* h, const double* mine, int64_t n_local, double* all)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->nccl_comm[0]) return fail(RVLL_E_RCCL, "rvll_comm_init has not been called");
    if (!mine || !all || n_local < 1) return fail(RVLL_E_INVALID, "bad allgather_host arguments");
    const size_t total = (size_t)n_local * (size_t)h->nranks;
    // a grow-only pair of device buffers kept in the handle (freed by rvll_destroy): a sharded sampler calls this once
    // per iteration, and hipMalloc / hipFree per call synchronise the whole device — every lane of every handle
    // 
    if ((long long)total > h->gather_host_cap) {
        HIP_TRY(hipStreamSynchronize(h->compute));
        dev_free(h->d_gather_host_in); dev_free(h->d_gather_host_out);
        h->gather_host_cap = 0;
        const size_t cap = std::max<size_t>(total, 4096);
        HIP_TRY(hipMalloc(&h->d_gather_host_in, sizeof(double) * cap));      // (n_local <= total)
        HIP_TRY(hipMalloc(&h->d_gather_host_out, sizeof(double) * cap));
        h->gather_host_cap = (long long)cap;
    }
    HIP_TRY(hipMemcpyAsync(h->d_gather_host_in, mine, sizeof(double) * (size_t)n_local, hipMemcpyHostToDevice, h->compute));
    const int r = g_rccl.AllGather(h->d_gather_host_in, h->d_gather_host_out, (size_t)n_local, kNcclFloat64, h->nccl_comm[0], h->compute);
    if (r != 0) {
        (void)hipStreamSynchronize(h->compute);
        return fail(RVLL_E_RCCL, "ncclAllGather failed: %s", g_rccl.GetErrorString(r));
    }
    HIP_TRY(hipMemcpyAsync(all, h->d_gather_host_out, sizeof(double) * total, hipMemcpyDeviceToHost, h->compute));
    HIP_TRY(hipStreamSynchronize(h->compute));
    return RVLL_OK;
}

int rvll_download_gathered_theta(rvll_handle* h, int64_t B_total, double* theta_all)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!theta_all || B_total < 1 || B_total > h->gather_theta_cap) return fail(RVLL_E_INVALID, "bad gathered theta download");
    HIP_TRY(hipMemcpyAsync(theta_all, h->d_gather_theta, sizeof(double) * (size_t)B_total * (size_t)h->L.ndim,
                           hipMemcpyDeviceToHost, h->compute));
    HIP_TRY(hipStreamSynchronize(h->compute));
    return RVLL_OK;
}

int rvll_download_gathered(rvll_handle* h, int64_t B_total, double* logL_all)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!logL_all || B_total < 1 || B_total > h->gather_cap) return fail(RVLL_E_INVALID, "bad gathered download");
    hipStream_t st = lane_stream(h, h->gather_last);
    HIP_TRY(hipMemcpyAsync(logL_all, h->d_gather2[h->gather_last], sizeof(double) * (size_t)B_total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RVLL_OK;
}

int rvll_comm_destroy(rvll_handle* h)
{
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->compute));
    rc = sync_other_lanes(h);
    if (rc) return rc;
    for (int lane = kMaxLanes - 1; lane >= 0; --lane) {
        if (h->nccl_comm[lane] && g_rccl.lib) RCCL_TRY(g_rccl.CommDestroy(h->nccl_comm[lane]));
        h->nccl_comm[lane] = nullptr;
    }
    h->nranks = 1;
    h->rank = 0;
    h->nlanes = 1;
    h->logl_cur = 0;
    return RVLL_OK;
}

}  // extern "C"
