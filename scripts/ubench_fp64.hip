// ubench_fp64.hip — fp64 VALU issue-rate microbenchmarks on gfx950 (build + run on the GPU box):
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I../evidence_amd/csrc ubench_fp64.hip -o /tmp/ubench && /tmp/ubench
// Each variant runs ITER iterations of a small body in every lane, grid = 256 CUs x 4 blocks x 256 threads
// (4 waves/SIMD), and reports wave-instructions/s/SIMD and the implied cycles per wave-instruction at the
// clock measured in-kernel (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "rvll_math.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096;

template <int V> __global__ __launch_bounds__(256, 4) void k(double* out, unsigned long long* clk, double seed)
{
    double a0 = seed + threadIdx.x * 1e-9, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3;
    double a4 = a0 + 0.4, a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7;
    const double m = 0.999999, c = 1e-7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITER; ++i) {
        if (V == 0) {       // 8 independent fma chains
            a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
            a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
        } else if (V == 1) { // 8 independent mul
            a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m;
        } else if (V == 2) { // 8 independent add
            a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c;
        } else if (V == 3) { // 1 dependent fma chain (latency)
            a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
            a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
        } else if (V == 4) { // 8 rcp
            a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3);
            a4 = __builtin_amdgcn_rcp(a4); a5 = __builtin_amdgcn_rcp(a5); a6 = __builtin_amdgcn_rcp(a6); a7 = __builtin_amdgcn_rcp(a7);
        } else if (V == 5) { // 8 IEEE divisions
            a0 = c / a0 + 1.5; a1 = c / a1 + 1.5; a2 = c / a2 + 1.5; a3 = c / a3 + 1.5; a4 = c / a4 + 1.5; a5 = c / a5 + 1.5; a6 = c / a6 + 1.5; a7 = c / a7 + 1.5;
        } else if (V == 6) { // 4 sincos
            double s, cc;
            rvll::sincos_f64(a0, s, cc); a0 += s * c + cc; rvll::sincos_f64(a1, s, cc); a1 += s * c + cc;
            rvll::sincos_f64(a2, s, cc); a2 += s * c + cc; rvll::sincos_f64(a3, s, cc); a3 += s * c + cc;
        } else if (V == 7) { // 8 x (cndmask pair): select between values
            a0 = a0 > a1 ? a2 : a3; a1 = a1 > a2 ? a3 : a4; a2 = a2 > a3 ? a4 : a5; a3 = a3 > a4 ? a5 : a6;
            a4 = a4 > a5 ? a6 : a7; a5 = a5 > a6 ? a7 : a0; a6 = a6 > a7 ? a0 : a1; a7 = a7 > a0 ? a1 : a2;
        } else if (V == 8) { // 4 x log
            a0 = log(a0 + 2.0); a1 = log(a1 + 2.0); a2 = log(a2 + 2.0); a3 = log(a3 + 2.0);
        } else if (V == 9) { // 4 x ocml sincos
            double s, cc;
            sincos(a0, &s, &cc); a0 += s * c + cc; sincos(a1, &s, &cc); a1 += s * c + cc;
            sincos(a2, &s, &cc); a2 += s * c + cc; sincos(a3, &s, &cc); a3 += s * c + cc;
        }
    }
    float f0 = (float)a0, f1 = f0 + 0.1f, f2 = f0 + 0.2f, f3 = f0 + 0.3f, f4 = f0 + 0.4f, f5 = f0 + 0.5f, f6 = f0 + 0.6f, f7 = f0 + 0.7f;
    const float mf = 0.999999f, cf = 1e-7f;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    if (V >= 10) for (int i = 0; i < ITER; ++i) {
        if (V == 10) {      // 8 independent f32 fma
            f0 = __builtin_fmaf(f0, mf, cf); f1 = __builtin_fmaf(f1, mf, cf); f2 = __builtin_fmaf(f2, mf, cf); f3 = __builtin_fmaf(f3, mf, cf);
            f4 = __builtin_fmaf(f4, mf, cf); f5 = __builtin_fmaf(f5, mf, cf); f6 = __builtin_fmaf(f6, mf, cf); f7 = __builtin_fmaf(f7, mf, cf);
        } else if (V == 11) { // 4 sincos_f32
            float s, cc;
            rvll::sincos_f32(f0, s, cc); f0 += s * cf + cc; rvll::sincos_f32(f1, s, cc); f1 += s * cf + cc;
            rvll::sincos_f32(f2, s, cc); f2 += s * cf + cc; rvll::sincos_f32(f3, s, cc); f3 += s * cf + cc;
        } else if (V == 12) { // 8 div_f32 (+1 add)
            f0 = rvll::div_f32(cf, f0) + 1.5f; f1 = rvll::div_f32(cf, f1) + 1.5f; f2 = rvll::div_f32(cf, f2) + 1.5f; f3 = rvll::div_f32(cf, f3) + 1.5f;
            f4 = rvll::div_f32(cf, f4) + 1.5f; f5 = rvll::div_f32(cf, f5) + 1.5f; f6 = rvll::div_f32(cf, f6) + 1.5f; f7 = rvll::div_f32(cf, f7) + 1.5f;
        } else if (V == 13) { // 8 integer mul-add style ops (v_mad_u32 / v_add / v_xor)
            i0 = (i0 ^ i1) + 12345; i1 = (i1 ^ i2) + 12345; i2 = (i2 ^ i3) + 12345; i3 = (i3 ^ i4) + 12345;
            i4 = (i4 ^ i5) + 12345; i5 = (i5 ^ i6) + 12345; i6 = (i6 ^ i7) + 12345; i7 = (i7 ^ i0) + 12345;
        } else if (V == 14) { // 8 x (cmp f32 + select)
            f0 = f0 > f1 ? f2 : f3; f1 = f1 > f2 ? f3 : f4; f2 = f2 > f3 ? f4 : f5; f3 = f3 > f4 ? f5 : f6;
            f4 = f4 > f5 ? f6 : f7; f5 = f5 > f6 ? f7 : f0; f6 = f6 > f7 ? f0 : f1; f7 = f7 > f0 ? f1 : f2;
        } else if (V == 15) { // 8 cvt f64->f32->f64 round trips
            a0 = (double)(float)a0 + c; a1 = (double)(float)a1 + c; a2 = (double)(float)a2 + c; a3 = (double)(float)a3 + c;
            a4 = (double)(float)a4 + c; a5 = (double)(float)a5 + c; a6 = (double)(float)a6 + c; a7 = (double)(float)a7 + c;
        } else if (V == 16) { // 4 x logf fast
            f0 = __logf(f0 + 2.f); f1 = __logf(f1 + 2.f); f2 = __logf(f2 + 2.f); f3 = __logf(f3 + 2.f);
        } else if (V == 17) { // 4 x v_pk_fma_f32 (two fp32 fma each): does packing two Horner chains pay?
            typedef float pk2 __attribute__((ext_vector_type(2)));
            pk2 p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7};
            const pk2 mm = {0.999999f, 0.999998f}, c2 = {1e-7f, 2e-7f};
            p0 = __builtin_elementwise_fma(p0, mm, c2); p1 = __builtin_elementwise_fma(p1, mm, c2);
            p2 = __builtin_elementwise_fma(p2, mm, c2); p3 = __builtin_elementwise_fma(p3, mm, c2);
            f0 = p0.x; f1 = p0.y; f2 = p1.x; f3 = p1.y; f4 = p2.x; f5 = p2.y; f6 = p3.x; f7 = p3.y;
        }
    }
    a0 += (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int V> int run(const char* name, int ops_per_iter, double* out, unsigned long long* clk, int blocks)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, clk, 1.25);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, clk, 1.25);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<unsigned long long> h(2 * blocks); CHECK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0; for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; real += h[2 * b + 1]; }
    const double ghz = cyc / real * 0.1;                  // s_memrealtime ticks at 100 MHz
    const double wave_ops = (double)blocks * 4 * ITER * ops_per_iter;      // wave-level source ops
    const double per_simd_per_s = wave_ops / (ms * 1e-3) / 1024.0;
    printf("%-28s %8.3f ms  clock %.2f GHz  %.1f cycles per wave-op  (%.2f Tops/s lane-ops)\n", name, ms, ghz,
           ghz * 1e9 / per_simd_per_s, wave_ops * 64 / (ms * 1e-3) / 1e12);
    return 0;
}

int main()
{
    const int blocks = 256 * 4;
    double* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * 256)); CHECK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks));
    run<0>("fma x8 independent", 8, out, clk, blocks);
    run<1>("mul x8 independent", 8, out, clk, blocks);
    run<2>("add x8 independent", 8, out, clk, blocks);
    run<3>("fma x8 dependent chain", 8, out, clk, blocks);
    run<4>("rcp x8", 8, out, clk, blocks);
    run<5>("IEEE div (+1 add) x8", 8, out, clk, blocks);
    run<6>("rvll sincos (+3 ops) x4", 4, out, clk, blocks);
    run<7>("cmp+select f64 x8", 8, out, clk, blocks);
    run<8>("ocml log (+1 add) x4", 4, out, clk, blocks);
    run<9>("ocml sincos (+3 ops) x4", 4, out, clk, blocks);
    run<10>("f32 fma x8 independent", 8, out, clk, blocks);
    run<11>("rvll sincos_f32 (+3 ops) x4", 4, out, clk, blocks);
    run<12>("div_f32 (+1 add) x8", 8, out, clk, blocks);
    run<13>("int xor+add x8 (2 ops each)", 8, out, clk, blocks);
    run<14>("cmp+select f32 x8", 8, out, clk, blocks);
    run<15>("cvt f64->f32->f64 (+1 add) x8", 8, out, clk, blocks);
    run<16>("__logf (+1 add) x4", 4, out, clk, blocks);
    run<17>("v_pk_fma_f32 x4 (8 fp32 fma)", 4, out, clk, blocks);
    return 0;
}
