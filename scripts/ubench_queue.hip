// What does handing a row from one workgroup to another cost on gfx950?  (time slices.)
// 1024 workgroups x 256 threads (the walk's launch), each iterating: 152 doubles to a private global row (+ scratch
// stores like the walk's tile), then one of
//   0  nothing                                     (baseline: stores + barrier)
//   1  __threadfence() by every thread            (what the time-slice build did)
//   2  s_waitcnt + barrier, __threadfence() by ONE thread
//   3  as 2 + atomicAdd on ONE global counter by that thread
//   4  as 2 + compare-and-swap loop on ONE global counter (a ring's head)
//   5  atomicAdd only (no fence)
//   6  atomicAdd on a per-XCD counter (8 counters, chosen by XCC_ID), no fence
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_queue.hip -o /tmp/ubq && /tmp/ubq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k(int variant, int iters, double* rows, double* scratch, unsigned long long* ctr)
{
    const int tid = threadIdx.x;
    double* row = rows + (size_t)blockIdx.x * 160;
    double* scr = scratch + (size_t)blockIdx.x * 2048;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;   // HW_REG_XCC_ID, low bits
    for (int it = 0; it < iters; ++it) {
        if (tid < 152) row[tid] = it + tid;
        for (int i = tid; i < 1600; i += 256) scr[i] = it * 0.5 + i;       // the tile's theta_out / logL style traffic
        if (variant == 1) __threadfence();
        if (variant >= 2 && variant <= 4) { __builtin_amdgcn_s_waitcnt(0); }
        __syncthreads();
        if (tid == 255) {
            if (variant >= 2 && variant <= 4) __threadfence();
            if (variant == 3 || variant == 5) atomicAdd(ctr, 1ull);
            if (variant == 6) atomicAdd(ctr + 8 * (1 + xcc), 1ull);
            if (variant == 4) {
                for (;;) {
                    const unsigned long long h = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (atomicCAS(ctr, h, h + 1) == h) break;
                }
            }
        }
        __syncthreads();
    }
}

int main()
{
    double *rows, *scratch; unsigned long long* ctr;
    hipMalloc(&rows, 1024 * 160 * 8); hipMalloc(&scratch, 1024 * 2048 * 8); hipMalloc(&ctr, 128 * 8);
    hipMemset(ctr, 0, 128 * 8);
    const char* names[] = {"stores + barrier only", "fence by every thread", "fence by one thread", "fence by one thread + atomicAdd (one counter)",
                           "fence by one thread + CAS loop (one counter)", "atomicAdd only (one counter)", "atomicAdd only (per-XCD counter)"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 7; ++v) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, v, iters, rows, scratch, ctr);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-48s %8.2f us per iteration (1024 workgroups each doing one per iteration)\n", names[v], ms * 1e3 / iters);
        }
    return 0;
}
