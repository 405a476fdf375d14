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
//   fip_accumulate_kernel  one thread per (run, bin); see the comment at the kernel.  No atomics, no sort: the
//                          fold order is the reference's, the result is bit-identical and deterministic.
// Integer / cache-side work around a strictly sequential fp64 chain per bin: every tile re-reads its run's
// spans from L2 (rows * 8 * np bytes, a few MB per posterior, cache-resident); the time is set by the hottest
// bin — a posterior peak receives most of the samples — at ~7 vector instructions per covering sample.
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
constexpr int kRowsPerThread = 2;
constexpr int kDenseHits = 16;          // rows of a 64-row group that must touch a wave for the straight-line path

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

// coverage of the 64 bins [lo, lo + 64) by the span [beg, end), one bit per bin
__device__ __forceinline__ unsigned long long span_bits(int2 sp, int lo)
{
    const int a = max(sp.x - lo, 0), b = min(sp.y - lo, kWave);
    if (a >= b) return 0ull;
    const unsigned long long upto_b = b >= kWave ? ~0ull : ((1ull << b) - 1ull);
    return upto_b & ~((1ull << a) - 1ull);
}

__device__ __forceinline__ unsigned long long lane_u64(unsigned long long x, int h)
{
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)x, h);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(x >> 32), h);
    return ((unsigned long long)hi << 32) | lo;
}

// v -= c in the lanes whose bit is set in the wave-uniform mask m (a scalar register pair; c is the same in
// every lane): one v_add_f64 under a narrowed EXEC instead of an add plus two v_cndmask
__device__ __forceinline__ void masked_sub(double& v, unsigned long long m, double c)
{
    unsigned long long saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_and_b64 exec, %[sv], %[m]\n\t"
                 "v_add_f64 %[v], %[v], -%[c]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [v] "+v"(v), [sv] "=&s"(saved)
                 : [m] "s"(m), [c] "v"(c)
                 : "scc");
}

// four of them back to back (rows h .. h+3 in order): the sixteen v_readlane of the batch issue ahead of the
// dependent add chain, and EXEC is saved and restored once
__device__ __forceinline__ void masked_sub4(double& v, unsigned long long m0, double c0, unsigned long long m1,
                                            double c1, unsigned long long m2, double c2, unsigned long long m3,
                                            double c3)
{
    unsigned long long saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_and_b64 exec, %[sv], %[m0]\n\t"
                 "v_add_f64 %[v], %[v], -%[c0]\n\t"
                 "s_and_b64 exec, %[sv], %[m1]\n\t"
                 "v_add_f64 %[v], %[v], -%[c1]\n\t"
                 "s_and_b64 exec, %[sv], %[m2]\n\t"
                 "v_add_f64 %[v], %[v], -%[c2]\n\t"
                 "s_and_b64 exec, %[sv], %[m3]\n\t"
                 "v_add_f64 %[v], %[v], -%[c3]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [v] "+v"(v), [sv] "=&s"(saved)
                 : [m0] "s"(m0), [c0] "v"(c0), [m1] "s"(m1), [c1] "v"(c1), [m2] "s"(m2), [c2] "v"(c2),
                   [m3] "s"(m3), [c3] "v"(c3)
                 : "scc");
}

// Workgroup barrier that only waits for this wave's LDS traffic: the global loads prefetching the next chunk
// stay in flight across it (a full __syncthreads would drain them first).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A workgroup owns 256 consecutive bins of one run (consumer wave w: bins tile_lo + 64 w ..) and streams the run's
// rows kRowsPerThread * 256 at a time.  It has 512 threads: waves 4-7 are PRODUCERS — one thread per row, the
// union of the row's spans becomes one 64-bit coverage mask per consumer wave (index work, vectorised over rows)
// and goes to LDS with the row's contribution; the spans are requested one chunk ahead — and waves 0-3 are the
// CONSUMERS that fold: the rows with a non-zero mask are visited in row order; the mask of the visited row is
// moved to a scalar register pair (v_readlane), its contribution is an LDS broadcast, and the subtraction runs
// under EXEC = mask, which is bit-identical to skipping it in the uncovered lanes.  Groups of 64 rows that mostly
// touch the wave take a straight-line path over all 64 (an empty mask changes nothing); sparse groups walk the
// set bits.  The LDS stage is double-buffered: the producers fill stage i+1 while the consumers fold stage i, one
// barrier per chunk.  The per-bin fold stays strictly sequential; its cost is ~3 vector instructions per
// (row, wave) the row touches.
template <int NP>
__global__ __launch_bounds__(2 * kThreads)
void fip_accumulate_kernel(const int2* __restrict__ spans, const double* __restrict__ contrib,
                           const long long* __restrict__ run_start, long long n_rows, int nfreq,
                           double* __restrict__ fapnu)
{
    constexpr int kWaves = kThreads / kWave;
    constexpr int kChunk = kThreads * kRowsPerThread;
    __shared__ unsigned long long s_mask[2][kWaves][kChunk];
    __shared__ double s_c[2][kChunk];
    __shared__ int s_any[2][kWaves];

    // 512 threads: waves 0-3 own the 256 bins (the fold), waves 4-7 prepare the NEXT chunk's masks meanwhile
    const bool producer = threadIdx.x >= kThreads;
    const int tid = threadIdx.x & (kThreads - 1), lane = tid & (kWave - 1), wave = tid >> 6;
    const int run = blockIdx.y;
    const int tile_lo = blockIdx.x * kThreads;
    const int tile_hi = min(tile_lo + kThreads, nfreq);
    const int bin = tile_lo + tid;
    const long long r0 = run_start[run], r1 = run_start[run + 1];
    double v = (!producer && bin < nfreq) ? fapnu[(long long)run * nfreq + bin] : 0.;

    int2 sp_next[kRowsPerThread][NP];
    double c_next[kRowsPerThread];
    auto fetch = [&](long long base) {
#pragma unroll
        for (int q = 0; q < kRowsPerThread; ++q) {
            const long long row = base + tid + q * kThreads;
            const bool ok = row < r1;
#pragma unroll
            for (int j = 0; j < NP; ++j) sp_next[q][j] = ok ? spans[(long long)j * n_rows + row] : make_int2(0, 0);
            c_next[q] = ok ? contrib[row] : 0.;
        }
    };
    // phase 1 of one chunk into stage `b`: registers -> per-wave coverage masks + contributions in LDS
    auto produce = [&](int b) {
        int2 sp[kRowsPerThread][NP];
        double c[kRowsPerThread];
#pragma unroll
        for (int q = 0; q < kRowsPerThread; ++q) {
#pragma unroll
            for (int j = 0; j < NP; ++j) sp[q][j] = sp_next[q][j];
            c[q] = c_next[q];
        }
        bool any = false;
#pragma unroll
        for (int q = 0; q < kRowsPerThread; ++q) {
            unsigned long long m[kWaves];
#pragma unroll
            for (int w = 0; w < kWaves; ++w) m[w] = 0ull;
            bool hit = false;
#pragma unroll
            for (int j = 0; j < NP; ++j)
                hit |= sp[q][j].x < sp[q][j].y && sp[q][j].x < tile_hi && sp[q][j].y > tile_lo;
            if (hit) {
#pragma unroll
                for (int j = 0; j < NP; ++j)
#pragma unroll
                    for (int w = 0; w < kWaves; ++w) m[w] |= span_bits(sp[q][j], tile_lo + w * kWave);
                any = true;
            }
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s_mask[b][w][tid + q * kThreads] = m[w];
            s_c[b][tid + q * kThreads] = c[q];
        }
        const unsigned long long wave_any = __ballot(any);
        if (lane == 0) s_any[b][wave] = wave_any != 0ull;
    };

    if (producer) {
        fetch(r0);
        if (r0 < r1) {
                    produce(0);                                   // chunk 0 into stage 0
            if (r0 + kChunk < r1) fetch(r0 + kChunk);
        }
    }
    lds_barrier();

    int buf = 0;
    for (long long base = r0; base < r1; base += kChunk, buf ^= 1) {
        if (producer) {
            // chunk i+1 goes into the other stage while the consumers fold chunk i; its spans were requested one
            // iteration ago, the ones of chunk i+2 are requested now
            if (base + kChunk < r1) {
                produce(buf ^ 1);
                if (base + 2 * kChunk < r1) fetch(base + 2 * kChunk);
            }
        } else if (s_any[buf][0] | s_any[buf][1] | s_any[buf][2] | s_any[buf][3]) {
            // phase 2
            for (int g = 0; g < kChunk / kWave; ++g) {
                const unsigned long long mv = s_mask[buf][wave][g * kWave + lane];
                const double* cg = &s_c[buf][g * kWave];
                unsigned long long todo = __ballot(mv != 0ull);
                if (__popcll(todo) >= kDenseHits) {
                    double cn[8];                       // contributions are fetched two batches ahead of their use
#pragma unroll
                    for (int i = 0; i < 8; ++i) cn[i] = cg[i];
#pragma unroll
                    for (int h = 0; h < kWave; h += 4) {
                        const double c0 = cn[0], c1 = cn[1], c2 = cn[2], c3 = cn[3];
#pragma unroll
                        for (int i = 0; i < 4; ++i) cn[i] = cn[i + 4];
                        if (h + 8 < kWave) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) cn[4 + i] = cg[h + 8 + i];
                        }
                        masked_sub4(v, lane_u64(mv, h), c0, lane_u64(mv, h + 1), c1, lane_u64(mv, h + 2), c2,
                                    lane_u64(mv, h + 3), c3);
                    }
                } else {
                    while (todo) {
                        const int h = __builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        masked_sub(v, lane_u64(mv, h), cg[h]);
                    }
                }
            }
        }
        lds_barrier();                                    // stage buf is free again, stage buf^1 is complete
    }
    if (!producer && bin < nfreq) fapnu[(long long)run * nfreq + bin] = v;
}

template <int NP>
hipError_t launch_accumulate(const int2* spans, const double* contrib, const long long* run_start, long long n_rows,
                             int nfreq, int n_runs, double* fapnu, hipStream_t s)
{
    const dim3 grid((unsigned)((nfreq + kThreads - 1) / kThreads), (unsigned)n_runs);
    hipLaunchKernelGGL(fip_accumulate_kernel<NP>, grid, dim3(2 * kThreads), 0, s, spans, contrib, run_start, n_rows,
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
