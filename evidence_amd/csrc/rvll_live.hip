// rvll_live.hip — small data-movement kernels of the device-resident live set (rvll_live_*; include/rvll.h): row gather /
// scatter by index, and the mean and covariance of a subset of rows in a fixed summation order.
//
// The reference leaves nested sampling to UltraNest (evidence/ultranest/__init__.py:159-185), whose region slice sampler
// whitens its directions with the live points' covariance; evidence_amd/nested.py does the same.  With the walk on the
// device (rvll_walk.hip) what was left on the host per iteration was index copies of the live points and of the walker rows
// through PCIe (a fifth of the end-to-end time): these kernels keep all of it in HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "rvll_kernels.h"

namespace rvll {

namespace {

constexpr int kMomBlocks = 128;       // partial sums: a fixed number of blocks, so the summation order is fixed

// dst[i][:] = src[idx[i]][:]   (rows of `width` doubles)
__global__ __launch_bounds__(kThreads)
void gather_rows_kernel(const double* src, const int32_t* idx, long long n, int width, double* dst)
{
    const long long total = n * width;
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long long)gridDim.x * kThreads) {
        const long long i = e / width;
        dst[e] = src[(long long)idx[i] * width + (e - i * width)];
    }
}

// dst[idx[i]][:] = src[i][:]   (idx holds distinct rows)
__global__ __launch_bounds__(kThreads)
void scatter_rows_kernel(const double* src, const int32_t* idx, long long n, int width, double* dst)
{
    const long long total = n * width;
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long long)gridDim.x * kThreads) {
        const long long i = e / width;
        dst[(long long)idx[i] * width + (e - i * width)] = src[e];
    }
}

// partial column sums of the rows idx[0..n): block b takes rows b, b + kMomBlocks, ...; its threads are kThreads / D groups
// of D, group g summing every (kThreads / D)-th of the block's rows, and the groups' sums are added in group order — a
// fixed order for a given (n, D).  (One thread per column, as first written, left 237 of 256 threads idle: 63 us per call.)
__global__ __launch_bounds__(kThreads)
void moments_sum_kernel(const double* u, const int32_t* idx, long long n, int D, double* part /*[kMomBlocks][D]*/)
{
    __shared__ double acc[kThreads];
    const int groups = D <= kThreads ? kThreads / D : 1;
    const int g = threadIdx.x / D, d = threadIdx.x - g * D;
    if (D <= kThreads) {
        double s = 0.;
        if (g < groups)
            for (long long i = blockIdx.x + (long long)kMomBlocks * g; i < n; i += (long long)kMomBlocks * groups)
                s += u[(long long)idx[i] * D + d];
        acc[threadIdx.x] = s;
        __syncthreads();
        if (g == 0) {
            double t = 0.;
            for (int k = 0; k < groups; ++k) t += acc[k * D + d];
            part[(long long)blockIdx.x * D + d] = t;
        }
    } else {
        for (int dd = threadIdx.x; dd < D; dd += kThreads) {
            double s = 0.;
            for (long long i = blockIdx.x; i < n; i += kMomBlocks) s += u[(long long)idx[i] * D + dd];
            part[(long long)blockIdx.x * D + dd] = s;
        }
    }
}

// out[d] = (sum over blocks) * scale: four threads a column, each a quarter of the blocks in block order, the quarters added in
// order — a fixed order for a given width.  (One thread a column walked 128 dependent loads: 9.5 us a call, two calls a step.)
constexpr int kFoldCols = kThreads / 4;
__global__ __launch_bounds__(kThreads)
void moments_fold_kernel(const double* part, int width, double scale, double* out)
{
    __shared__ double q[4][kFoldCols];
    const int c = threadIdx.x % kFoldCols, k = threadIdx.x / kFoldCols;
    const int d = blockIdx.x * kFoldCols + c;
    double s = 0.;
    if (d < width)
        for (int b = k * (kMomBlocks / 4); b < (k + 1) * (kMomBlocks / 4); ++b) s += part[(long long)b * width + d];
    q[k][c] = s;
    __syncthreads();
    if (k == 0 && d < width) out[d] = ((q[0][c] + q[1][c]) + (q[2][c] + q[3][c])) * scale;
}

// partial sums of the centred products (u_j - m_j)(u_l - m_l): thread p one (j, l) pair, rows as above — staged through LDS
// kCovRows at a time (read from HBM row by row, every (j, l) pair waited ~2 us for each of its block's rows: 51 us a call at
// 16384 rows; the sums are the same terms in the same order)
constexpr int kCovRows = 32;
__global__ __launch_bounds__(kThreads)
void moments_cov_kernel(const double* u, const int32_t* idx, long long n, int D, const double* mean,
                        double* part /*[kMomBlocks][D*D]*/)
{
    extern __shared__ double rows[];                         // [kCovRows][D], centred
    double* ms = rows + (size_t)kCovRows * D;                // [D]
    for (int d = threadIdx.x; d < D; d += kThreads) ms[d] = mean[d];
    constexpr int kPairs = 8;                                // (j, l) pairs a thread may hold: D * D <= kPairs * kThreads
    double s[kPairs];
#pragma unroll
    for (int q = 0; q < kPairs; ++q) s[q] = 0.;
    const bool in_regs = (long long)D * D <= (long long)kPairs * kThreads;
    if (!in_regs) for (int p = threadIdx.x; p < D * D; p += kThreads) part[(long long)blockIdx.x * D * D + p] = 0.;
    __syncthreads();
    for (long long i0 = blockIdx.x; i0 < n; i0 += (long long)kMomBlocks * kCovRows) {
        // this block's next kCovRows rows: i0, i0 + kMomBlocks, ...
        int nr = 0;
        for (long long i = i0; i < n && nr < kCovRows; i += kMomBlocks) ++nr;
        for (int e = threadIdx.x; e < nr * D; e += kThreads) {
            const int r = e / D, d = e - r * D;
            rows[e] = u[(long long)idx[i0 + (long long)r * kMomBlocks] * D + d] - ms[d];
        }
        __syncthreads();
        if (in_regs) {
#pragma unroll
            for (int q = 0; q < kPairs; ++q) {
                const int p = threadIdx.x + q * kThreads;
                if (p < D * D) {
                    const int j = p / D, l = p - j * D;
                    double acc = s[q];
                    for (int r = 0; r < nr; ++r) acc += rows[r * D + j] * rows[r * D + l];
                    s[q] = acc;
                }
            }
        } else {
            for (int p = threadIdx.x; p < D * D; p += kThreads) {
                const int j = p / D, l = p - j * D;
                double acc = part[(long long)blockIdx.x * D * D + p];
                for (int r = 0; r < nr; ++r) acc += rows[r * D + j] * rows[r * D + l];
                part[(long long)blockIdx.x * D * D + p] = acc;
            }
        }
        __syncthreads();
    }
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < kPairs; ++q) {
            const int p = threadIdx.x + q * kThreads;
            if (p < D * D) part[(long long)blockIdx.x * D * D + p] = s[q];
        }
    }
}

// ---- the live points' order, on the device (rvll_live_sort) ----------------------------------------------------------------
// log-L -> a key whose unsigned order is the doubles' order (the sign bit flipped for positives, every bit for negatives)
__global__ __launch_bounds__(kThreads)
void sort_keys_kernel(const double* logl, long long n, unsigned long long* keys, int32_t* rows)
{
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(logl[i]);
        keys[i] = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
        rows[i] = (int32_t)i;
    }
}

// out[i] = order[offset + rank[i]]
__global__ __launch_bounds__(kThreads)
void compose_index_kernel(const int32_t* order, long long offset, const int32_t* rank, long long n, int32_t* out)
{
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads)
        out[i] = order[offset + rank[i]];
}

int blocks_for(long long total)
{
    long long b = (total + kThreads - 1) / kThreads;
    return (int)(b < 1 ? 1 : b > 8192 ? 8192 : b);
}

}  // namespace

hipError_t launch_gather_rows(const double* src, const int32_t* idx, long long n, int width, double* dst, hipStream_t st)
{
    if (n <= 0 || width <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_for(n * width)), dim3(kThreads), 0, st, src, idx, n, width, dst);
    return hipGetLastError();
}

hipError_t launch_scatter_rows(const double* src, const int32_t* idx, long long n, int width, double* dst, hipStream_t st)
{
    if (n <= 0 || width <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(blocks_for(n * width)), dim3(kThreads), 0, st, src, idx, n, width, dst);
    return hipGetLastError();
}

size_t moments_scratch_doubles(int D) { return (size_t)kMomBlocks * D * D + (size_t)D; }

// mean[D] and cov[D*D] (divided by n - 1, as numpy's d.T @ d / (n - 1)) of the rows idx[0..n) of u; scratch:
// moments_scratch_doubles(D) doubles.  Two passes, every sum in a fixed order: the same bits for the same rows.
hipError_t launch_moments(const double* u, const int32_t* idx, long long n, int D, double* scratch, double* mean, double* cov,
                          hipStream_t st)
{
    if (n <= 0 || D <= 0) return hipErrorInvalidValue;
    double* part = scratch;
    hipLaunchKernelGGL(moments_sum_kernel, dim3(kMomBlocks), dim3(kThreads), 0, st, u, idx, n, D, part);
    hipLaunchKernelGGL(moments_fold_kernel, dim3((D + kFoldCols - 1) / kFoldCols), dim3(kThreads), 0, st, part, D, 1.0 / (double)n, mean);
    hipLaunchKernelGGL(moments_cov_kernel, dim3(kMomBlocks), dim3(kThreads), sizeof(double) * ((size_t)kCovRows * D + D), st, u, idx, n, D, mean, part);
    hipLaunchKernelGGL(moments_fold_kernel, dim3((D * D + kFoldCols - 1) / kFoldCols), dim3(kThreads), 0, st, part, D * D, 1.0 / (double)(n > 1 ? n - 1 : 1), cov);
    return hipGetLastError();
}

// Stable ascending order of logl[0..n) (ties by row, as numpy's stable argsort): keys / rows are scratch of n each (in and out),
// temp of sort_temp_bytes(n) bytes; order_out[n] receives the rows, sorted_out[n] (or null) the sorted keys' log-L is not
// returned — the caller gathers what it needs by index.
size_t sort_temp_bytes(long long n)
{
    size_t bytes = 0;
    unsigned long long* k = nullptr;
    int32_t* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0, 64, (hipStream_t) nullptr);
    return bytes;
}

hipError_t launch_sort_logl(const double* logl, long long n, unsigned long long* keys_in, unsigned long long* keys_out, int32_t* rows_in,
                            int32_t* order_out, void* temp, size_t temp_bytes, hipStream_t st)
{
    if (n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sort_keys_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, st, logl, n, keys_in, rows_in);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, rows_in, order_out, (size_t)n, 0, 64, st);
}

hipError_t launch_compose_index(const int32_t* order, long long offset, const int32_t* rank, long long n, int32_t* out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(compose_index_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, st, order, offset, rank, n, out);
    return hipGetLastError();
}

}  // namespace rvll
