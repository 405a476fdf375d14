// rvll_comm.hip — multi-GPU: one process per GPU, live points sharded by rows, ONE all-gather of the per-shard log-L (and, for
// cube shards, of theta) per step over RCCL (xGMI), pipeline lanes with a communicator each.  This replaces the MPI fan-out the
// reference leaves to its samplers (evidence/polychord/__init__.py:21-29,176-199).  RCCL is resolved at run time with dlopen, so
// the library loads on boxes without it.
#include <dlfcn.h>
#include "rvll_host.h"

using rvll::report_error;
using namespace rvll::host;

namespace {

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
    if (!lib) return report_error(RVLL_E_RCCL, "cannot dlopen librccl (tried %s): %s", tried.c_str(), dlerror());
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
        return report_error(RVLL_E_RCCL, "librccl lacks an expected nccl* symbol");
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
            return report_error(RVLL_E_RCCL, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

}  // namespace

namespace rvll {
namespace host {
// rvll_destroy: the handle's communicators go before its streams do
void comm_release(rvll_handle* h)
{
    for (auto& c : h->nccl_comm) if (c && g_rccl.lib) { (void)g_rccl.CommDestroy(c); c = nullptr; }
}
}  // namespace host
}  // namespace rvll

extern "C" {

// ---- multi-GPU ---------------------------------------------------------------------
int rvll_comm_unique_id(unsigned char id[RVLL_COMM_ID_BYTES])
{
    if (!id) return report_error(RVLL_E_INVALID, "id is null");
    int rc = rccl_load();
    if (rc) return rc;
    Id128 u;
    memset(&u, 0, sizeof u);
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, u.bytes, RVLL_COMM_ID_BYTES);
    return RVLL_OK;
}

// The streamed host batches want eight hardware queues (include/rvll.h, rvll_runtime_info): a library should not leave its fast
// path to an environment variable its caller has to know about, so librvll asks for them itself when it is
// loaded — unless the caller has chosen — which is before the process's first HIP call whenever librvll is what brings the HIP
// runtime in.  rvll_runtime_info says who set it.
namespace {
bool g_hwq_by_library = false;
__attribute__((constructor)) void rvll_default_hw_queues()
{
    if (!getenv("GPU_MAX_HW_QUEUES") && !getenv("RVLL_KEEP_HW_QUEUES")) {
        setenv("GPU_MAX_HW_QUEUES", "8", 0);
        g_hwq_by_library = true;
    }
}
}  // namespace

int rvll_runtime_info(char* buf, int32_t buflen)
{
    if (!buf || buflen < 1) return report_error(RVLL_E_INVALID, "bad buffer");
    int hip_rt = 0, hip_drv = 0;
    (void)hipRuntimeGetVersion(&hip_rt);
    (void)hipDriverGetVersion(&hip_drv);
    // GPU_MAX_HW_QUEUES as this process's environment has it now (the runtime read it when it started: a caller that set it
    // later, or loaded another HIP user first, runs on the default 4 whatever this says — evidence_amd/_abi.py records which)
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    snprintf(buf, (size_t)buflen,
             "{\"hip_runtime_version\": %d, \"hip_driver_version\": %d, \"libamdhip64\": \"%s\", "
             "\"librccl\": \"%s\", \"rccl_version\": %d, \"librvll\": \"%s\", \"gpu_max_hw_queues_env\": \"%s\", \"gpu_max_hw_queues_set_by\": \"%s\"}",
             hip_rt, hip_drv, lib_path_of(reinterpret_cast<const void*>(&hipGetDeviceCount)).c_str(),
             g_rccl.lib ? g_rccl_path.c_str() : "not loaded", g_rccl_version,
             lib_path_of(reinterpret_cast<const void*>(&rvll_runtime_info)).c_str(), q ? q : "unset (runtime default: 4)",
             g_hwq_by_library ? "librvll when it was loaded" : (q ? "the caller's environment" : "nobody"));
    return RVLL_OK;
}

// One communicator, one pipeline lane: the gather of a step runs in-stream behind its kernel.  Further lanes are
// added by rvll_comm_add_lanes once the caller has seen a gathered step complete on this one.
int rvll_comm_init(rvll_handle* h, const unsigned char id[RVLL_COMM_ID_BYTES], int32_t nranks, int32_t rank)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return report_error(RVLL_E_INVALID, "bad comm arguments");
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
    if (!h->nccl_comm[0]) return report_error(RVLL_E_RCCL, "rvll_comm_init has not been called");
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
    if (!h->nccl_comm[0]) return report_error(RVLL_E_RCCL, "rvll_comm_init has not been called");
    if (nlanes < 1 || nlanes > kMaxLanes) return report_error(RVLL_E_INVALID, "nlanes out of range");
    for (int l = 0; l < nlanes; ++l)
        if (!h->nccl_comm[l]) return report_error(RVLL_E_INVALID, "lane %d has no communicator on this rank", l);
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
    if (!h->nccl_comm[0]) return report_error(RVLL_E_RCCL, "rvll_comm_init has not been called");
    if (B_local < 1 || B_local > h->cap) return report_error(RVLL_E_INVALID, "B_local %lld outside reserved capacity %lld", (long long)B_local, h->cap);
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
                              h->nccl_comm[lane], h->lanes[lane]));
    h->gather_last = lane;
    if (h->nlanes > 1) h->logl_cur = (lane + 1) % h->nlanes;   // the next step runs on the next lane
    return RVLL_OK;
}

int rvll_allgather_theta(rvll_handle* h, int64_t B_local)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->nccl_comm[0]) return report_error(RVLL_E_RCCL, "rvll_comm_init has not been called");
    if (B_local < 1 || B_local > h->cap) return report_error(RVLL_E_INVALID, "B_local %lld outside reserved capacity %lld", (long long)B_local, h->cap);
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
    if (!h->nccl_comm[0]) return report_error(RVLL_E_RCCL, "rvll_comm_init has not been called");
    if (!mine || !all || n_local < 1) return report_error(RVLL_E_INVALID, "bad allgather_host arguments");
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
        return report_error(RVLL_E_RCCL, "ncclAllGather failed: %s", g_rccl.GetErrorString(r));
    }
    HIP_TRY(hipMemcpyAsync(all, h->d_gather_host_out, sizeof(double) * total, hipMemcpyDeviceToHost, h->compute));
    HIP_TRY(hipStreamSynchronize(h->compute));
    return RVLL_OK;
}

int rvll_download_gathered_theta(rvll_handle* h, int64_t B_total, double* theta_all)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!theta_all || B_total < 1 || B_total > h->gather_theta_cap) return report_error(RVLL_E_INVALID, "bad gathered theta download");
    HIP_TRY(hipMemcpyAsync(theta_all, h->d_gather_theta, sizeof(double) * (size_t)B_total * (size_t)h->L.ndim,
                           hipMemcpyDeviceToHost, h->compute));
    HIP_TRY(hipStreamSynchronize(h->compute));
    return RVLL_OK;
}

int rvll_download_gathered(rvll_handle* h, int64_t B_total, double* logL_all)
{
    int rc = use_device(h);
    if (rc) return rc;
    if (!logL_all || B_total < 1 || B_total > h->gather_cap) return report_error(RVLL_E_INVALID, "bad gathered download");
    hipStream_t st = h->lanes[h->gather_last];
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
