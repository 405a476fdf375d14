// rvll_math.h — fp64 device math for the Kepler kernels (gfx950).
//
// The whole translation unit is compiled with -ffp-contract=off: the reference
// solver (evidence/rvmodel/trueanomaly.c:25-29, built without FMA) rounds after
// every multiply and add, and the Newton stop rule |E-E0| > 1e-4 is sensitive to
// that, so the solver arithmetic is written op by op and every fused multiply-add
// in this file is an explicit __builtin_fma() inside the transcendental approximations.
//
// gfx950 has no fp64 transcendental hardware; sin/cos here are a two-constant
// Cody-Waite reduction (exact first step through FMA) + the classic degree-13/14
// minimax kernels on [-pi/4, pi/4] (coefficients: Sun fdlibm k_sin.c/k_cos.c,
// public domain).  Absolute error <= ~1 ulp(1) for |x| < 2^40 — the mean
// anomalies of this path reach |M| ~ 1e4 rad (never range-reduced by the
// reference, rvmodel/__init__.py:459).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RVLL_HD __host__ __device__ __forceinline__

namespace rvll {

RVLL_HD double as_double(uint64_t u) { return __builtin_bit_cast(double, u); }
RVLL_HD uint64_t as_u64(double d) { return __builtin_bit_cast(uint64_t, d); }

// sin and cos of x, one shared range reduction.
RVLL_HD void sincos_f64(double x, double& s_out, double& c_out)
{
    constexpr double TWO_OVER_PI = 6.36619772367581382433e-01;
    constexpr double PIO2_HI     = 1.57079632679489655800e+00;  // 0x3FF921FB54442D18
    constexpr double PIO2_LO     = 6.12323399573676603587e-17;  // pi/2 - PIO2_HI
    constexpr double MAGIC       = 6755399441055744.0;          // 1.5 * 2^52

    // k = nearest integer to x*2/pi, via the round-to-nearest-even of the add;
    // its low bits sit in the low mantissa word of t.
    const double t  = x * TWO_OVER_PI + MAGIC;
    const double fk = t - MAGIC;
    const uint32_t q = (uint32_t)as_u64(t);

    // r1 = x - k*PIO2_HI is exact (|r1| < 1, multiple of 2^-53); second word rounds once.
    const double r1 = __builtin_fma(-fk, PIO2_HI, x);
    const double r  = __builtin_fma(-fk, PIO2_LO, r1);
    const double z  = r * r;

    // sin(r) on [-pi/4, pi/4]
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(r * z, ps, r);

    // cos(r) on [-pi/4, pi/4]
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    pc = __builtin_fma(z, pc, -0.5);
    const double cr = __builtin_fma(z, pc, 1.0);

    // quadrant: q&1 swaps, bit 1 of q flips sin, bit 1 of (q+1) flips cos
    const bool swap = (q & 1u) != 0u;
    double s = swap ? cr : sr;
    double c = swap ? sr : cr;
    const uint64_t ssign = (uint64_t)(q & 2u) << 62;
    const uint64_t csign = (uint64_t)((q + 1u) & 2u) << 62;
    s_out = as_double(as_u64(s) ^ ssign);
    c_out = as_double(as_u64(c) ^ csign);
}

// Rotate (s, c) = (sin E0, cos E0) to E0 + h for a small step |h| <= ~1e-3:
// sin h and cos h by short Taylor sums (h^7/5040 < 2e-25 at 1e-3).
RVLL_HD void rotate_small(double h, double& s, double& c)
{
    const double h2 = h * h;
    // sin h = h (1 - h2/6 + h2^2/120),  cos h = 1 - h2/2 + h2^2/24 - h2^3/720
    const double sh = h * __builtin_fma(h2, __builtin_fma(h2, 8.33333333333333333333e-03, -1.66666666666666666667e-01), 1.0);
    const double ch1 = h2 * __builtin_fma(h2, __builtin_fma(h2, -1.38888888888888888889e-03, 4.16666666666666666667e-02), -0.5);
    // s' = s cos h + c sin h = s + (s*ch1 + c*sh);  c' = c + (c*ch1 - s*sh)
    const double s0 = s, c0 = c;
    s = s0 + __builtin_fma(s0, ch1, c0 * sh);
    c = c0 + __builtin_fma(c0, ch1, -(s0 * sh));
}

// counter-based uniform in [0,1): splitmix64 finaliser over (seed, index)
RVLL_HD double uniform01(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (index + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * 1.1102230246251565404e-16;   // 2^-53
}

}  // namespace rvll
