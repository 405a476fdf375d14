// rvll_host.h — what the host-side translation units of librvll.so share: the handle behind the C-ABI's opaque
// pointer, the error macro, and the helpers of rvll_api.hip the other units call.  Internal; not part of the ABI.
//   rvll_api.hip        the ABI core: create / destroy, priors, resident and host-buffer batch calls, scalar-call server,
//                       streamed host batches, curves, self tests
//   rvll_walk_host.hip  the sampler's proposal step: the device walk in its forms, the resident live set (rvll_live_*)
//   rvll_comm.hip       multi-GPU: RCCL communicators, lanes, all-gathers
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

// the C-ABI entry points are the only exported symbols (built with -fvisibility=hidden)
#pragma GCC visibility push(default)
#include "rvll.h"
#pragma GCC visibility pop
#include "rvll_kernels.h"
#include "rvll_copypool.h"

namespace rvll {
// sets the thread's last-error string (rvll_last_error) and returns `code`
int report_error(int code, const char* fmt, ...);
}  // namespace rvll

#define HIP_TRY(expr)                                                               \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess)                                                       \
            return ::rvll::report_error(e_ == hipErrorOutOfMemory ? RVLL_E_NOMEM : RVLL_E_HIP, \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                        __FILE__, __LINE__);                                        \
    } while (0)

namespace rvll {
namespace host {
template <typename T>
void dev_free(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }
}  // namespace host
}  // namespace rvll

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
    bool priors_rowwise = false;                // some prior reads other coordinates of its row (the sorted kinds)
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
    int32_t* d_walk_wflag = nullptr;            // [walk_cap] 1: the walker's last accepted candidate had a wandering solve (walk_core puts its log-L right)
    int wander_exact = 1;                       // wandering solves are redone with correctly rounded sin / cos (rvll_set_wander_exact; RVLL_WANDER_EXACT)
    int walk_spec = 4;                          // candidates a walker may evaluate ahead per iteration (rvll_set_walk_speculation)
    // the walk as rounds of launches (rvll_rounds.hip; walk_rounds in rvll_walk_host.hip): one arena with every group's walker
    // state, candidate slots and counters; a progress word per group in mapped pinned memory
    int walk_spec_rounds = 8;                   // ... and per round of the rounds form, while a round is below the latency floor
    void* d_rounds = nullptr;
    double* d_walk_dirs = nullptr;              // [K, nsteps, D] the directions of all moves of the walk in progress
    size_t walk_dirs_cap = 0;                   // in doubles
    size_t rounds_bytes = 0;
    unsigned long long* pin_rounds = nullptr;   // [kMaxLanes]
    unsigned long long* pin_rounds_dev = nullptr;
    hipEvent_t ev_rounds = nullptr;             // orders the groups' streams behind lane 0
    hipEvent_t ev_chain[kMaxLanes] = {};        // group g's step of a round has been issued to the device (the next group's step waits for it)
    hipStream_t rounds_streams[kMaxLanes] = {}; // the groups' streams, at DIFFERENT priorities (walk_rounds)
    int walk_rounds_used = 0;                   // rounds the last walk took (0: it ran in one of the single-kernel forms)
    long long walk_evaluated = 0;               // tile slots the last rvll_slice_walk evaluated (>= its ncalls)
    unsigned long long walk_phase[6] = {};      // diagnostic build (make walktrace): 100 MHz ticks per phase, summed over workgroups; workgroups

    // device-resident live set (rvll_live_*): nested sampling's live points, and the points that died, stay in HBM
    long long live_n = 0, live_cap = 0;
    double *d_live_u = nullptr, *d_live_theta = nullptr, *d_live_logl = nullptr;
    int32_t* d_live_idx = nullptr;              // [2 * live_cap] order, then start rows, of the current step
    double *d_live_mom = nullptr;               // scratch, mean, covariance of the whitening
    // the order on the device (rvll_live_sort): keys in / out, rows in (the order itself lands in d_live_idx), rocPRIM's scratch
    unsigned long long* d_sort_keys = nullptr;  // [2 * live_cap]
    int32_t* d_sort_rows = nullptr;             // [live_cap]
    void* d_sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
    long long sorted_kdead = -1;                // kdead of the rvll_live_sort whose order d_live_idx holds (-1: none, or used up)
    double sorted_lstar = 0.;
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

namespace rvll {
namespace host {

// helpers of rvll_api.hip the other host units use
int use_device(rvll_handle* h);                     // select the device, stop a running scalar-call server, settle a pending one-launch batch
int sync_other_lanes(rvll_handle* h);
int build_args(rvll_handle* h, const double* d_theta, double* d_logL, int32_t* d_flags, long long B, rvll::LoglikeArgs* out,
               int* cu_grid = nullptr);
void make_fused(const rvll_handle* h, const double* d_cube, double* d_theta_out, rvll::LoglikeArgs* a);
int download_rows(rvll_handle* h, void* dst, const void* src_dev, size_t bytes);
void comm_release(rvll_handle* h);                  // rvll_comm.hip: destroy the handle's communicators
constexpr size_t kDownloadStagedMin = 32u << 20, kDeadStagedMin = 8u << 20;

}  // namespace host
}  // namespace rvll
