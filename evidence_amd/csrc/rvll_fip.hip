// FIP periodogram accumulation on gfx950 — the loop of evidence/fip_criterion.py:305-339 as two kernels.
//
// Reference semantics (per independent run r, planet models k = 1.., posterior samples i in file order):
//     x_freqs = 2*pi / samples[i]                               (:321)
//     beg = searchsorted(nub, x_freqs, 'right')                 (:334)
//     end = searchsorted(nua, x_freqs, 'left')                  (:335)
//     fapnu[r, union of range(beg_j, end_j)] -= pky[k]*weights[i]   (:336-339; numpy applies a repeated
//                                                                index once, so it is a set union)
// Every bin is a sequential fold  v = ((1 - c_1) - c_2) - ...  over the samples that cover it, in sample
// order; fp64 subtraction does not commute in the last bit, so the fold order is part of the result.
//
// Mapping: the host flattens (k, i) into rows in the reference's loop order with c = pky[k]*weights[i].
//   fip_index_kernel       one thread per (row, planet): IEEE division, two binary searches on the caller's
//                          own nua/nub arrays (index work — exactly numpy's answer), intervals stored SoA.
//   fip_accumulate_kernel  one thread per (run, bin); a workgroup owns 256 consecutive bins of one run and
//                          streams that run's rows 256 at a time: each thread tests one row against the
//                          tile, the hits are compacted IN ROW ORDER into LDS (ballot + prefix), then every
//                          thread folds the hits that cover its bin.  No atomics, no sort: the fold order is
//                          the reference's, the result is bit-identical and run-to-run deterministic.
// Integer/HBM-side work: per tile the rows are re-read from L2 (n_rows * (8 + 8*np) bytes); the intervals
// of one posterior are a few MB and stay cache-resident.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#pragma GCC visibility push(default)
#include "rvll.h"
#pragma GCC visibility pop

namespace rvll {
int report_error(int code, const char* fmt, ...);
}

namespace {

constexpr int kThreads = 256;
constexpr int kWave = 64;
constexpr int kMaxPlanets = RVLL_FIP_MAX_PLANETS;

#define FIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            status = rvll::report_error(e_ == hipErrorOutOfMemory ? RVLL_E_NOMEM : RVLL_E_HIP, \
                                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                        __FILE__, __LINE__);                                   \
            goto done;                                                                         \
        }                                                                                      \
    } while (0)

// number of a[i] <= v (numpy.searchsorted(a, v, 'right')); a NaN v compares false everywhere -> 0, and the
// matching lower bound is 0 too, i.e. the same empty interval numpy's (n, n) is
__device__ int count_le(const double* __restrict__ a, int n, double v)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// number of a[i] < v (numpy.searchsorted(a, v, 'left'))
__device__ int count_lt(const double* __restrict__ a, int n, double v)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kThreads)
void fip_index_kernel(const double* __restrict__ periods, long long n_rows, int np,
                      const double* __restrict__ nua, const double* __restrict__ nub, int nfreq,
                      int2* __restrict__ spans /*[np][n_rows]*/)
{
    const long long total = n_rows * np;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total;
         i += (long long)gridDim.x * kThreads) {
        const long long row = i / np;
        const int j = (int)(i - row * np);
        const double f = 6.283185307179586 / periods[i];          // 2*np.pi / x, correctly rounded
        int beg = count_le(nub, nfreq, f);
        int end = count_lt(nua, nfreq, f);
        if (!(beg < end)) beg = end = 0;                          // canonical empty span
        spans[(long long)j * n_rows + row] = make_int2(beg, end);
    }
}

template <int NP>
__global__ __launch_bounds__(kThreads)
void fip_accumulate_kernel(const int2* __restrict__ spans, const double* __restrict__ contrib,
                           const long long* __restrict__ run_start, long long n_rows, int nfreq,
                           double* __restrict__ fapnu)
{
    __shared__ int2   hit_span[NP][kThreads];
    __shared__ double hit_c[kThreads];
    __shared__ int    wave_hits[kThreads / kWave];

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int run = blockIdx.y;
    const int tile_lo = blockIdx.x * kThreads;
    const int tile_hi = min(tile_lo + kThreads, nfreq);
    const int bin = tile_lo + tid;
    const long long r0 = run_start[run], r1 = run_start[run + 1];
    double v = bin < nfreq ? fapnu[(long long)run * nfreq + bin] : 0.;

    for (long long base = r0; base < r1; base += kThreads) {
        const long long row = base + tid;
        int2 sp[NP];
        bool hit = false;
        double c = 0.;
        if (row < r1) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                sp[j] = spans[(long long)j * n_rows + row];
                hit |= sp[j].x < sp[j].y && sp[j].x < tile_hi && sp[j].y > tile_lo;
            }
            if (hit) c = contrib[row];
        }
        // ordered compaction of the hits: position = hits in earlier waves + hits in lower lanes
        const unsigned long long mask = __ballot(hit);
        if (lane == 0) wave_hits[wave] = __popcll(mask);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kThreads / kWave; ++w) {
            const int h = wave_hits[w];
            if (w < wave) before += h;
            total += h;
        }
        if (hit) {
            const int pos = before + __popcll(mask & ((1ull << lane) - 1ull));
#pragma unroll
            for (int j = 0; j < NP; ++j) hit_span[j][pos] = sp[j];
            hit_c[pos] = c;
        }
        __syncthreads();
        for (int h = 0; h < total; ++h) {
            bool covered = false;
#pragma unroll
            for (int j = 0; j < NP; ++j) covered |= bin >= hit_span[j][h].x && bin < hit_span[j][h].y;
            if (covered) v -= hit_c[h];
        }
        __syncthreads();
    }
    if (bin < nfreq) fapnu[(long long)run * nfreq + bin] = v;
}

template <int NP>
hipError_t launch_accumulate(const int2* spans, const double* contrib, const long long* run_start, long long n_rows,
                             int nfreq, int n_runs, double* fapnu, hipStream_t s)
{
    const dim3 grid((unsigned)((nfreq + kThreads - 1) / kThreads), (unsigned)n_runs);
    hipLaunchKernelGGL(fip_accumulate_kernel<NP>, grid, dim3(kThreads), 0, s, spans, contrib, run_start, n_rows,
                       nfreq, fapnu);
    return hipGetLastError();
}

}  // namespace

extern "C" __attribute__((visibility("default")))
int rvll_fip_accumulate(int32_t device, const double* nua, const double* nub, int32_t nfreq,
                        const double* periods, const double* contrib, const int64_t* run_start,
                        int32_t n_runs, int32_t np_max, double* fapnu, int32_t repeats, rvll_fip_timing* timing)
{
    if (!nua || !nub || !run_start || !fapnu) return rvll::report_error(RVLL_E_INVALID, "null argument");
    if (nfreq < 1 || n_runs < 1) return rvll::report_error(RVLL_E_INVALID, "nfreq and n_runs must be >= 1");
    if (np_max < 1 || np_max > kMaxPlanets)
        return rvll::report_error(RVLL_E_INVALID, "np_max %d outside 1..%d", np_max, kMaxPlanets);
    if (n_runs > 65535) return rvll::report_error(RVLL_E_INVALID, "n_runs > 65535");
    if (run_start[0] != 0) return rvll::report_error(RVLL_E_INVALID, "run_start[0] must be 0");
    for (int r = 0; r < n_runs; ++r)
        if (run_start[r + 1] < run_start[r]) return rvll::report_error(RVLL_E_INVALID, "run_start must be non-decreasing");
    const long long n_rows = run_start[n_runs];
    if (n_rows > 0 && (!periods || !contrib)) return rvll::report_error(RVLL_E_INVALID, "null periods/contrib");
    if (repeats < 1) repeats = 1;
    for (int i = 1; i < nfreq; ++i)
        if (!(nua[i] >= nua[i - 1]) || !(nub[i] >= nub[i - 1]))
            return rvll::report_error(RVLL_E_INVALID, "nua/nub must be sorted ascending (searchsorted contract)");

    int status = RVLL_OK;
    int prev_device = -1;
    double *d_nua = nullptr, *d_nub = nullptr, *d_periods = nullptr, *d_contrib = nullptr, *d_fapnu = nullptr,
           *d_fapnu0 = nullptr;
    long long* d_run_start = nullptr;
    int2* d_spans = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    const size_t fbytes = sizeof(double) * (size_t)nfreq;
    const size_t out_bytes = fbytes * (size_t)n_runs;
    const size_t rows_alloc = (size_t)std::max<long long>(n_rows, 1);
    double index_ms = 0., acc_ms = 0.;

    FIP_TRY(hipGetDevice(&prev_device));
    if (device >= 0) FIP_TRY(hipSetDevice(device));
    FIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    for (auto& e : ev) FIP_TRY(hipEventCreate(&e));
    FIP_TRY(hipMalloc(&d_nua, fbytes));
    FIP_TRY(hipMalloc(&d_nub, fbytes));
    FIP_TRY(hipMalloc(&d_fapnu, out_bytes));
    FIP_TRY(hipMalloc(&d_fapnu0, out_bytes));
    FIP_TRY(hipMalloc(&d_run_start, sizeof(long long) * (size_t)(n_runs + 1)));
    FIP_TRY(hipMalloc(&d_periods, sizeof(double) * rows_alloc * (size_t)np_max));
    FIP_TRY(hipMalloc(&d_contrib, sizeof(double) * rows_alloc));
    FIP_TRY(hipMalloc(&d_spans, sizeof(int2) * rows_alloc * (size_t)np_max));
    FIP_TRY(hipMemcpyAsync(d_nua, nua, fbytes, hipMemcpyHostToDevice, stream));
    FIP_TRY(hipMemcpyAsync(d_nub, nub, fbytes, hipMemcpyHostToDevice, stream));
    FIP_TRY(hipMemcpyAsync(d_fapnu0, fapnu, out_bytes, hipMemcpyHostToDevice, stream));
    {
        std::vector<long long> rs((size_t)n_runs + 1);
        for (int r = 0; r <= n_runs; ++r) rs[(size_t)r] = run_start[r];
        FIP_TRY(hipMemcpyAsync(d_run_start, rs.data(), sizeof(long long) * rs.size(), hipMemcpyHostToDevice, stream));
        FIP_TRY(hipStreamSynchronize(stream));                    // rs goes out of scope
    }
    if (n_rows > 0) {
        FIP_TRY(hipMemcpyAsync(d_periods, periods, sizeof(double) * (size_t)n_rows * (size_t)np_max,
                               hipMemcpyHostToDevice, stream));
        FIP_TRY(hipMemcpyAsync(d_contrib, contrib, sizeof(double) * (size_t)n_rows, hipMemcpyHostToDevice, stream));
    }
    for (int rep = 0; rep < repeats; ++rep) {
        FIP_TRY(hipMemcpyAsync(d_fapnu, d_fapnu0, out_bytes, hipMemcpyDeviceToDevice, stream));
        FIP_TRY(hipEventRecord(ev[0], stream));
        if (n_rows > 0) {
            const long long total = n_rows * np_max;
            const unsigned blocks = (unsigned)std::min<long long>((total + kThreads - 1) / kThreads, 65535);
            hipLaunchKernelGGL(fip_index_kernel, dim3(blocks), dim3(kThreads), 0, stream, d_periods, n_rows,
                               (int)np_max, d_nua, d_nub, (int)nfreq, d_spans);
            FIP_TRY(hipGetLastError());
        }
        FIP_TRY(hipEventRecord(ev[1], stream));
        if (n_rows > 0) {
            hipError_t e = hipSuccess;
            switch (np_max) {
            case 1: e = launch_accumulate<1>(d_spans, d_contrib, d_run_start, n_rows, nfreq, n_runs, d_fapnu, stream); break;
            case 2: e = launch_accumulate<2>(d_spans, d_contrib, d_run_start, n_rows, nfreq, n_runs, d_fapnu, stream); break;
            case 3: e = launch_accumulate<3>(d_spans, d_contrib, d_run_start, n_rows, nfreq, n_runs, d_fapnu, stream); break;
            case 4: e = launch_accumulate<4>(d_spans, d_contrib, d_run_start, n_rows, nfreq, n_runs, d_fapnu, stream); break;
            case 5: e = launch_accumulate<5>(d_spans, d_contrib, d_run_start, n_rows, nfreq, n_runs, d_fapnu, stream); break;
            case 6: e = launch_accumulate<6>(d_spans, d_contrib, d_run_start, n_rows, nfreq, n_runs, d_fapnu, stream); break;
            case 7: e = launch_accumulate<7>(d_spans, d_contrib, d_run_start, n_rows, nfreq, n_runs, d_fapnu, stream); break;
            default: e = launch_accumulate<8>(d_spans, d_contrib, d_run_start, n_rows, nfreq, n_runs, d_fapnu, stream); break;
            }
            FIP_TRY(e);
        }
        FIP_TRY(hipEventRecord(ev[2], stream));
        FIP_TRY(hipStreamSynchronize(stream));
        float a = 0.f, b = 0.f;
        FIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
        FIP_TRY(hipEventElapsedTime(&b, ev[1], ev[2]));
        index_ms += a;
        acc_ms += b;
    }
    FIP_TRY(hipMemcpyAsync(fapnu, d_fapnu, out_bytes, hipMemcpyDeviceToHost, stream));
    FIP_TRY(hipStreamSynchronize(stream));
    if (timing) {
        timing->index_ms = index_ms / repeats;
        timing->accumulate_ms = acc_ms / repeats;
        timing->rows = n_rows;
        timing->repeats = repeats;
    }

done:
    for (void* p : {(void*)d_nua, (void*)d_nub, (void*)d_periods, (void*)d_contrib, (void*)d_fapnu, (void*)d_fapnu0,
                    (void*)d_run_start, (void*)d_spans})
        if (p) (void)hipFree(p);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
    if (prev_device >= 0 && device >= 0) (void)hipSetDevice(prev_device);
    return status;
}
