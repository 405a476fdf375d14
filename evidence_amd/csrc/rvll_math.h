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

// Both minimax kernels on r in [-pi/4, pi/4] (sin: degree 13 odd, cos: degree 14 even).
// On the device the two Horner chains are written as ONE block of interleaved
// v_fma_f64: hipcc otherwise lowers each step to v_mov_b64 (coefficient copy) + v_fmac_f64,
// which costs a third of the sin/cos time on an issue-bound fp64 pipe, and it pads
// separate asm statements with s_nop.  Pure VALU: no memory operations, no hazards.
// The constants of sincos_f64 as VALUES: by default compile-time constants the compiler places as it likes (the batch
// kernels: hoisted out of the Newton loop by loop-invariant code motion).  A translation unit built without machine LICM
// (rvll_walk.hip) defines RVLL_LOCAL_CONSTS and gets them as opaque register values it creates once in front of the loop
// (sincos_consts_pinned) — same numbers, same instructions, only where the registers are loaded differs.
struct SincosConsts {
    double S1, S2, S3, S4, S5, S6, C1, C2, C3, C4, C5, C6, TWO_OVER_PI, PIO2_HI, PIO2_LO, MAGIC;
};
RVLL_HD SincosConsts sincos_consts()
{
    return SincosConsts{-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
                        2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10,
                        4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
                        -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11,
                        6.36619772367581382433e-01, 1.57079632679489655800e+00, 6.12323399573676603587e-17,
                        6755399441055744.0};
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ SincosConsts sincos_consts_pinned()
{
    SincosConsts k = sincos_consts();
    asm volatile("" : "+s"(k.S1), "+s"(k.S2), "+s"(k.S3), "+s"(k.S4), "+s"(k.S5), "+v"(k.S6));
    asm volatile("" : "+s"(k.C1), "+s"(k.C2), "+s"(k.C3), "+s"(k.C4), "+s"(k.C5), "+v"(k.C6));
    asm volatile("" : "+s"(k.TWO_OVER_PI), "+s"(k.PIO2_HI), "+s"(k.PIO2_LO), "+v"(k.MAGIC));
    return k;
}
#else
RVLL_HD SincosConsts sincos_consts_pinned() { return sincos_consts(); }
#endif

RVLL_HD void sincos_kernel(double r, double& sr, double& cr, const SincosConsts& k)
{
    const double S1 = k.S1, S2 = k.S2, S3 = k.S3, S4 = k.S4, S5 = k.S5, S6 = k.S6;
    const double C1 = k.C1, C2 = k.C2, C3 = k.C3, C4 = k.C4, C5 = k.C5, C6 = k.C6;
#if defined(__HIP_DEVICE_COMPILE__)
    double z, t;
    asm("v_mul_f64 %2, %4, %4\n\t"
        "v_fma_f64 %0, %2, %10, %9\n\t"
        "v_fma_f64 %1, %2, %16, %15\n\t"
        "v_fma_f64 %0, %2, %0, %8\n\t"
        "v_fma_f64 %1, %2, %1, %14\n\t"
        "v_fma_f64 %0, %2, %0, %7\n\t"
        "v_fma_f64 %1, %2, %1, %13\n\t"
        "v_fma_f64 %0, %2, %0, %6\n\t"
        "v_fma_f64 %1, %2, %1, %12\n\t"
        "v_fma_f64 %0, %2, %0, %5\n\t"
        "v_fma_f64 %1, %2, %1, %11\n\t"
        "v_mul_f64 %3, %4, %2\n\t"
        "v_fma_f64 %1, %2, %1, -0.5\n\t"
        "v_fma_f64 %0, %3, %0, %4\n\t"
        "v_fma_f64 %1, %2, %1, 1.0"
        : "=&v"(sr), "=&v"(cr), "=&v"(z), "=&v"(t)
        : "v"(r), "s"(S1), "s"(S2), "s"(S3), "s"(S4), "s"(S5), "v"(S6),
          "s"(C1), "s"(C2), "s"(C3), "s"(C4), "s"(C5), "v"(C6));      // one scalar operand per VALU instruction
#else
    const double z = r * r;
    double ps = __builtin_fma(z, S6, S5);
    double pc = __builtin_fma(z, C6, C5);
    ps = __builtin_fma(z, ps, S4);
    pc = __builtin_fma(z, pc, C4);
    ps = __builtin_fma(z, ps, S3);
    pc = __builtin_fma(z, pc, C3);
    ps = __builtin_fma(z, ps, S2);
    pc = __builtin_fma(z, pc, C2);
    ps = __builtin_fma(z, ps, S1);
    pc = __builtin_fma(z, pc, C1);
    const double t = r * z;
    pc = __builtin_fma(z, pc, -0.5);
    sr = __builtin_fma(t, ps, r);
    cr = __builtin_fma(z, pc, 1.0);
#endif
}

// sin and cos of x, one shared range reduction.
RVLL_HD void sincos_f64(double x, double& s_out, double& c_out, const SincosConsts& k)
{
    const double TWO_OVER_PI = k.TWO_OVER_PI;
    const double PIO2_HI     = k.PIO2_HI;                       // 0x3FF921FB54442D18
    const double PIO2_LO     = k.PIO2_LO;                       // pi/2 - PIO2_HI
    const double MAGIC       = k.MAGIC;                         // 1.5 * 2^52

    // k = nearest integer to x*2/pi, via the round-to-nearest-even of the add;
    // its low bits sit in the low mantissa word of t.
    const double t  = __builtin_fma(x, TWO_OVER_PI, MAGIC);
    const double fk = t - MAGIC;
    const uint32_t q = (uint32_t)as_u64(t);

    // r1 = x - k*PIO2_HI is exact (|r1| < 1, multiple of 2^-53); second word rounds once.
    const double r1 = __builtin_fma(-fk, PIO2_HI, x);
    const double r  = __builtin_fma(-fk, PIO2_LO, r1);
    double sr, cr;
    sincos_kernel(r, sr, cr, k);

    // quadrant: q&1 swaps, bit 1 of q flips sin, bit 1 of (q+1) flips cos.  Done with bit selects
    // (v_bfi_b32 / v_xor_b32, ~2 cycles each) rather than compare + v_cndmask (~4 cycles each, measured).
    const uint32_t m = 0u - (q & 1u);                              // all ones when the quadrant swaps
    const uint64_t sb = as_u64(sr), cb = as_u64(cr);
    const uint32_t s_lo = (uint32_t)sb, s_hi = (uint32_t)(sb >> 32);
    const uint32_t c_lo = (uint32_t)cb, c_hi = (uint32_t)(cb >> 32);
    uint32_t rs_lo, rs_hi, rc_lo, rc_hi;
#if defined(__HIP_DEVICE_COMPILE__)
    // v_bfi_b32 D, M, A, B = (M & A) | (~M & B): one instruction per 32-bit half (hipcc expands the C form
    // below into twice as many and/or operations)
    asm("v_bfi_b32 %0, %4, %7, %5\n\t"
        "v_bfi_b32 %1, %4, %8, %6\n\t"
        "v_bfi_b32 %2, %4, %5, %7\n\t"
        "v_bfi_b32 %3, %4, %6, %8"
        : "=&v"(rs_lo), "=&v"(rs_hi), "=&v"(rc_lo), "=&v"(rc_hi)
        : "v"(m), "v"(s_lo), "v"(s_hi), "v"(c_lo), "v"(c_hi));
#else
    rs_lo = (c_lo & m) | (s_lo & ~m);
    rs_hi = (c_hi & m) | (s_hi & ~m);
    rc_lo = (s_lo & m) | (c_lo & ~m);
    rc_hi = (s_hi & m) | (c_hi & ~m);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    // sign flips: hi ^= (quadrant bit moved to bit 31) — "and" and "xor" as ONE three-input bit operation each
    // (v_bitop3_b32, truth table 0x78 = A ^ (B & C)); hipcc picks this form itself in some surroundings and two
    // instructions each in others, which is two of the Newton loop's 43
    const uint32_t q30 = q << 30;
    rs_hi = __builtin_amdgcn_bitop3_b32(rs_hi, q30, 0x80000000u, 0x78);
    rc_hi = __builtin_amdgcn_bitop3_b32(rc_hi, q30 + 0x40000000u, 0x80000000u, 0x78);
#else
    rs_hi ^= (q & 2u) << 30;
    rc_hi ^= ((q + 1u) & 2u) << 30;
#endif
    s_out = as_double(((uint64_t)rs_hi << 32) | rs_lo);
    c_out = as_double(((uint64_t)rc_hi << 32) | rc_lo);
}

RVLL_HD void sincos_f64(double x, double& s_out, double& c_out) { sincos_f64(x, s_out, c_out, sincos_consts()); }

// ---- sin / cos of ANY finite double ---------------------------------------------------------------------------------
// The reduction above is exact while x * 2/pi fits the 2^51 the magic-number rounding holds (and the second constant
// keeps the error below 2^-60 far beyond the |M| ~ 1e4 of this path).  The reference's Newton iteration, however,
// takes E far outside that at the 0.99 eccentricity clamp: started at E = M next to a zero of f' = 1 - e cos E it is
// thrown out to |E| ~ 1e9 ... 1e22 (measured with glibc on the golden sweep, tests/golden/loglike_high_ecc.npz) and
// finds its way back in 30 - 350 steps, and glibc reduces those arguments exactly.  An iteration fed anything else
// there takes another way home and stops a step-width (~1e-8 .. 1e-4 in E) from where the reference stops — or, with
// sin / cos that are not even a point of the unit circle, not at all.  So beyond 2^50 the argument is reduced the
// long way (Payne - Hanek): x = M 2^e with M the 53-bit integer significand; of 2/pi only the 192 bits from position
// e - 1 on can matter modulo 4 (everything before multiplies M 2^e to a multiple of 4); M times those 192 bits, modulo
// 2^192, is x 2/pi modulo 4 to 2^-137: two bits of quadrant, 190 bits of fraction.  The fraction is renormalised
// (double arguments come within 2^-61 of a multiple of pi/2, never closer), its leading 106 bits times pi/2 in
// double-double give the reduced argument to better than an ulp.
RVLL_HD uint64_t mulhi_u64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
// 2/pi: 64 zero bits (positions <= 0), then its first 1280 bits, most significant word first
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#endif
static const uint64_t kTwoOverPiBits[21] = {
    0x0000000000000000ull,
    0xa2f9836e4e441529ull, 0xfc2757d1f534ddc0ull, 0xdb6295993c439041ull, 0xfe5163abdebbc561ull,
    0xb7246e3a424dd2e0ull, 0x06492eea09d1921cull, 0xfe1deb1cb129a73eull, 0xe88235f52ebb4484ull,
    0xe99c7026b45f7e41ull, 0x3991d639835339f4ull, 0x9c845f8bbdf9283bull, 0x1ff897ffde05980full,
    0xef2f118b5a0a6d1full, 0x6d367ecf27cb09b7ull, 0x4f463f669e5fea2dull, 0x7527bac7ebe5f17bull,
    0x3d0739f78a5292eaull, 0x6bfb5fb11f8d5d08ull, 0x56033046fc7b6babull, 0xf0cfbc209af4361dull};

constexpr double kSincosFastMax = 1125899906842624.0;   // 2^50: below, sincos_f64's own reduction (exact first step, second
                                                        // constant good to 2^-60 up to there); from here on, reduce_huge
constexpr int    kSafeSteps = 8;                        // a Newton iterate cannot pass 2^50 before its 8th step (rvll_tile.h)
constexpr double kExcursionM = 281474976710656.0;       // 2^48
constexpr double kLongSolveEcc = 0.9;                   // a planet at or above it may hold a wandering solve: its point goes first
constexpr int    kF32Steps = 16;                        // reduced-precision modes: a solve not settled by then is done in double

// |x| in [2^50, inf), finite (any |x| >= 2^-10 works): r in [-pi/4, pi/4] and the quadrant q (mod 4) with |x| = q pi/2 + r (mod 2 pi)
RVLL_HD void reduce_huge(double x, double& r, uint32_t& q)
{
    const uint64_t bits = as_u64(x) & 0x7fffffffffffffffull;
    const int ex = (int)(bits >> 52) - 1075;                           // |x| = M * 2^ex
    const uint64_t M = (bits & 0x000fffffffffffffull) | 0x0010000000000000ull;
    const int o = ex + 62;                                             // first table bit that matters (0-based, pad included)
    const int wi = o >> 6, sh = o & 63;
    const uint64_t a0 = kTwoOverPiBits[wi], a1 = kTwoOverPiBits[wi + 1], a2 = kTwoOverPiBits[wi + 2];
    const uint64_t a3 = wi + 3 < 21 ? kTwoOverPiBits[wi + 3] : 0;
    const uint64_t w0 = sh ? (a0 << sh) | (a1 >> (64 - sh)) : a0;
    const uint64_t w1 = sh ? (a1 << sh) | (a2 >> (64 - sh)) : a1;
    const uint64_t w2 = sh ? (a2 << sh) | (a3 >> (64 - sh)) : a2;
    // (M * w0:w1:w2) mod 2^192
    uint64_t p2 = M * w2;
    uint64_t c2 = mulhi_u64(M, w2);
    uint64_t p1 = M * w1;
    uint64_t c1 = mulhi_u64(M, w1);
    uint64_t p0 = M * w0;
    p1 += c2;
    c1 += p1 < c2 ? 1 : 0;
    p0 += c1;
    q = (uint32_t)(p0 >> 62);
    // the 190-bit fraction, left-aligned in 192 bits; from 1/2 on, take it as a negative fraction of the next quadrant
    uint64_t g0 = (p0 << 2) | (p1 >> 62), g1 = (p1 << 2) | (p2 >> 62), g2 = p2 << 2;
    const bool neg = (g0 >> 63) != 0;
    if (neg) {
        q += 1;
        g2 = ~g2 + 1;
        g1 = ~g1 + (g2 == 0 ? 1 : 0);
        g0 = ~g0 + ((g2 == 0 && g1 == 0) ? 1 : 0);
    }
    // renormalise: at most ~62 leading zeros for a double argument, a few more are harmless
    int lz = 0;
    if (g0 == 0) { g0 = g1; g1 = g2; g2 = 0; lz = 64; }
    const int z = g0 ? __builtin_clzll(g0) : 63;
    if (z) { g0 = (g0 << z) | (g1 >> (64 - z)); g1 = (g1 << z) | (g2 >> (64 - z)); }
    lz += z;
    const double fh = (double)(g0 >> 11);                              // leading 53 bits, exact
    const double fl = (double)(((g0 & 0x7ffull) << 42) | (g1 >> 22)); // the next 53, exact
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
    const double hi = fh * PIO2_HI;
    const double lo = __builtin_fma(fh, PIO2_HI, -hi) + (fh * PIO2_LO + fl * (PIO2_HI * 1.1102230246251565e-16));
    double rr = ldexp(hi + lo, -53 - lz);                              // fh carries weight 2^(-53 - lz), fl 2^-53 of that
    r = neg ? -rr : rr;
}

// sin and cos of any finite x (inf / nan: nan, like libm): the fast reduction below 2^50, the long one beyond
RVLL_HD void sincos_any(double x, double& s_out, double& c_out, const SincosConsts& k)
{
    if (__builtin_fabs(x) < kSincosFastMax) { sincos_f64(x, s_out, c_out, k); return; }
    if (!(__builtin_fabs(x) < __builtin_inf())) { s_out = c_out = __builtin_nan(""); return; }
    double r, sr, cr;
    uint32_t q;
    reduce_huge(x, r, q);
    sincos_kernel(r, sr, cr, k);
    const double s = (q & 1u) ? cr : sr, c = (q & 1u) ? sr : cr;
    const double ss = (q & 2u) ? -s : s, cc = ((q + 1u) & 2u) ? -c : c;
    s_out = x < 0. ? -ss : ss;
    c_out = cc;
}
RVLL_HD void sincos_any(double x, double& s_out, double& c_out) { sincos_any(x, s_out, c_out, sincos_consts()); }

// ---- sin / cos, correctly rounded (an experiment of round 4, used only where a Newton solve wanders) -------------------
// at e >= 0.97, where the reference's iteration wanders for tens to hundreds of steps, where it stops hangs on
// the last bit of sin / cos.  glibc's are correctly rounded nearly always; the kernels above are good to ~1.2 ulp.
// This pair evaluates in double-double — the Payne - Hanek reduction's leading 106 bits as (hi, lo), the Taylor series of
// sin and cos in double-double arithmetic to 2^-100 — and rounds once: the correctly rounded value except within ~2^-47 ulp
// of a rounding boundary.  ~700 flops: for the solver's second loop only (0.01 % of the waves at cfg3's priors).
struct DD { double hi, lo; };
RVLL_HD DD dd_two_sum(double a, double b) { const double s = a + b, bb = s - a; return {s, (a - (s - bb)) + (b - bb)}; }
RVLL_HD DD dd_quick_sum(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }
RVLL_HD DD dd_two_prod(double a, double b) { const double p = a * b; return {p, __builtin_fma(a, b, -p)}; }
RVLL_HD DD dd_add(DD a, DD b)
{
    DD s = dd_two_sum(a.hi, b.hi);
    const DD t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_quick_sum(s.hi, s.lo);
    s.lo += t.lo;
    return dd_quick_sum(s.hi, s.lo);
}
RVLL_HD DD dd_mul(DD a, DD b)
{
    DD p = dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_quick_sum(p.hi, p.lo);
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#endif
static const double kSinTaylorDD[14][2] = {        // (-1)^k / (2k+1)!, k = 1 .. 14, as (hi, lo)
    {-0.16666666666666666, -9.25185853854297e-18},   {0.008333333333333333, 1.1564823173178714e-19},
    {-0.0001984126984126984, -1.7209558293420705e-22}, {2.7557319223985893e-06, -1.858393274046472e-22},
    {-2.505210838544172e-08, 1.448814070935912e-24},  {1.6059043836821613e-10, 1.2585294588752098e-26},
    {-7.647163731819816e-13, -7.03872877733453e-30},  {2.8114572543455206e-15, 1.6508842730861433e-31},
    {-8.22063524662433e-18, -2.2141894119604265e-34}, {1.9572941063391263e-20, -1.3643503830087908e-36},
    {-3.868170170630684e-23, 8.843177655482344e-40},  {6.446950284384474e-26, -1.9330404233703465e-42},
    {-9.183689863795546e-29, -1.4303150396787322e-45}, {1.1309962886447716e-31, 1.0498015412959506e-47}};
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#endif
static const double kCosTaylorDD[14][2] = {        // (-1)^k / (2k)!, k = 1 .. 14
    {-0.5, 0.0},                                      {0.041666666666666664, 2.3129646346357427e-18},
    {-0.001388888888888889, 5.300543954373577e-20},   {2.48015873015873e-05, 2.1511947866775882e-23},
    {-2.755731922398589e-07, -2.3767714622250297e-23}, {2.08767569878681e-09, -1.20734505911326e-25},
    {-1.1470745597729725e-11, -2.0655512752830745e-28}, {4.779477332387385e-14, 4.399205485834081e-31},
    {-1.5619206968586225e-16, -1.1910679660273754e-32}, {4.110317623312165e-19, 1.4412973378659527e-36},
    {-8.896791392450574e-22, 7.911402614872376e-38},  {1.6117375710961184e-24, -3.6846573564509766e-41},
    {-2.4795962632247976e-27, 1.2953730964765229e-43}, {3.279889237069838e-30, 1.5117542744029879e-46}};

// |x| finite: the reduced argument as a double-double in [-pi/4, pi/4] and the quadrant (reduce_huge's arithmetic, kept in two
// words; below pi/4 the argument is its own reduction)
RVLL_HD void reduce_dd(double x, DD& r, uint32_t& q)
{
    const double ax = __builtin_fabs(x);
    if (ax <= 0.78539816339744828) { r = {ax, 0.}; q = 0; return; }
#ifndef RVLL_CR_LONG_REDUCTION_ONLY       // (measurement / test builds: every argument through the integer reduction)
    if (ax < 70368744177664.0) {
        // below 2^46 (k is then the right multiple to 2^-7: |r| <= 0.794, inside the table's range): k pi/2 taken off in double-double — pi/2 in four words, every product with its error term (FMA); the
        // absolute error is ~2^-108, so a result below 2^-20 (an argument that close to a multiple of pi/2: one in 2^19) is
        // left to the integer reduction below, which keeps 106 bits of whatever is left.  A third of its latency, and a
        // wandering Newton iteration spends most of its steps here (|E| grows by at most 1 / (1 - e) ~ 130 a step).
        const double kf = __builtin_rint(ax * 6.36619772367581382433e-01);
        const double P1 = 1.5707963267948966, P2 = 6.123233995736766e-17, P3 = -1.4973849048591698e-33, P4 = 5.562271104316826e-50;
        const DD p1 = dd_two_prod(kf, P1), p2 = dd_two_prod(kf, P2);
        const double t = ax - p1.hi;                                    // exact (p1.hi is within a factor two of ax)
        const DD a = dd_two_sum(t, -p1.lo);
        const DD b = dd_two_sum(a.hi, -p2.hi);
        const double tail = ((a.lo + b.lo) - p2.lo) - __builtin_fma(kf, P3, kf * P4);
        const DD s = dd_quick_sum(b.hi, tail);
        if (__builtin_fabs(s.hi) >= 9.5367431640625e-07) {
            r = s;
            q = (uint32_t)(int)(kf - 4.0 * __builtin_floor(kf * 0.25));
            return;
        }
    }
#endif
    const uint64_t bits = as_u64(x) & 0x7fffffffffffffffull;
    const int ex = (int)(bits >> 52) - 1075;
    const uint64_t M = (bits & 0x000fffffffffffffull) | 0x0010000000000000ull;
    const int o = ex + 62;
    const int wi = o >> 6, sh = o & 63;
    const uint64_t a0 = kTwoOverPiBits[wi], a1 = kTwoOverPiBits[wi + 1], a2 = kTwoOverPiBits[wi + 2];
    const uint64_t a3 = wi + 3 < 21 ? kTwoOverPiBits[wi + 3] : 0;
    const uint64_t w0 = sh ? (a0 << sh) | (a1 >> (64 - sh)) : a0;
    const uint64_t w1 = sh ? (a1 << sh) | (a2 >> (64 - sh)) : a1;
    const uint64_t w2 = sh ? (a2 << sh) | (a3 >> (64 - sh)) : a2;
    uint64_t p2 = M * w2;
    uint64_t c2 = mulhi_u64(M, w2);
    uint64_t p1 = M * w1;
    uint64_t c1 = mulhi_u64(M, w1);
    uint64_t p0 = M * w0;
    p1 += c2;
    c1 += p1 < c2 ? 1 : 0;
    p0 += c1;
    q = (uint32_t)(p0 >> 62);
    uint64_t g0 = (p0 << 2) | (p1 >> 62), g1 = (p1 << 2) | (p2 >> 62), g2 = p2 << 2;
    const bool neg = (g0 >> 63) != 0;
    if (neg) {
        q += 1;
        g2 = ~g2 + 1;
        g1 = ~g1 + (g2 == 0 ? 1 : 0);
        g0 = ~g0 + ((g2 == 0 && g1 == 0) ? 1 : 0);
    }
    int lz = 0;
    if (g0 == 0) { g0 = g1; g1 = g2; g2 = 0; lz = 64; }
    const int z = g0 ? __builtin_clzll(g0) : 63;
    if (z) { g0 = (g0 << z) | (g1 >> (64 - z)); g1 = (g1 << z) | (g2 >> (64 - z)); }
    lz += z;
    const double fh = (double)(g0 >> 11);                              // leading 53 bits, exact
    const double fl = (double)(((g0 & 0x7ffull) << 42) | (g1 >> 22)); // the next 53, exact (weight 2^-53 of fh's)
    // (fh + fl 2^-53) * (pi/2 as three words), in double-double
    const double P1 = 1.5707963267948966, P2 = 6.123233995736766e-17, P3 = -1.4973849048591698e-33;
    const DD a = dd_two_prod(fh, P1);
    const double flw = fl * 1.1102230246251565e-16;
    const DD b = dd_two_prod(fh, P2), c = dd_two_prod(flw, P1);
    DD s = dd_add(b, c);
    s.lo += fh * P3 + flw * P2;
    s = dd_add(a, dd_quick_sum(s.hi, s.lo));
    const double sc = ldexp(1.0, -53 - lz);                            // exact scaling
    r = {neg ? -s.hi * sc : s.hi * sc, neg ? -s.lo * sc : s.lo * sc};
}

// sin r, cos r for a double-double |r| <= pi/4, each rounded once
RVLL_HD void sincos_dd_kernel(DD r, double& s_out, double& c_out)
{
    const DD z = dd_mul(r, r);
    DD ps = {kSinTaylorDD[13][0], kSinTaylorDD[13][1]}, pc = {kCosTaylorDD[13][0], kCosTaylorDD[13][1]};
    for (int k = 12; k >= 0; --k) {
        ps = dd_add(dd_mul(ps, z), DD{kSinTaylorDD[k][0], kSinTaylorDD[k][1]});
        pc = dd_add(dd_mul(pc, z), DD{kCosTaylorDD[k][0], kCosTaylorDD[k][1]});
    }
    const DD s = dd_add(r, dd_mul(dd_mul(r, z), ps));                 // r + r z (c1 + c2 z + ...)
    const DD c = dd_add(DD{1.0, 0.0}, dd_mul(z, pc));
    s_out = s.hi;
    c_out = c.hi;
}

// ---- the same values by a short route (the redo pass's sin / cos: one wave in a CU-wide launch held the launch for 2 us per
// Newton step on the series above — a 300-step solve stretched a 1 ms launch to 1.4 ms) ----------------------------------------
// r = k / 64 + h, |h| <= 2^-7: sin(k/64), cos(k/64) from a table in double-double, sin h and cos h - 1 by their series with the
// leading terms in double-double, combined in double-double: 2^-68 of the value.  A result that close to a rounding boundary is
// not decided by it (Ziv's test, about one value in 2^12): those go through the full series (sincos_dd_kernel).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#endif
static const double kSinCosTabDD[52][4] = {        // sin(k/64) (hi, lo), cos(k/64) (hi, lo), k = 0 .. 51
    {0.0, 0.0, 1.0, 0.0},
    {0.015624364224883372, -1.2650937552759816e-19, 0.9998779321710066, 3.216122229972341e-17},
    {0.03124491398532608, -1.562781562225433e-18, 0.9995117584851364, -3.418806487972947e-17},
    {0.04685783574813424, -2.3419368365610254e-18, 0.9989015683384429, -2.1425557800399754e-17},
    {0.0624593178423802, -2.040259504585711e-18, 0.9980475107000991, 3.3232291674141346e-17},
    {0.07804555138996731, -5.449443782005793e-18, 0.9969497940760287, -1.2467075728553626e-17},
    {0.09361273123551289, 1.4628632005878733e-18, 0.9956086864580017, 3.312922430932991e-17},
    {0.10915705687532236, 6.6284699502736666e-18, 0.9940245152582091, 1.3287985046260087e-17},
    {0.12467473338522769, -2.925947496057858e-18, 0.992197667229329, 4.754870575189364e-17},
    {0.1401619723470637, -9.946847113883478e-18, 0.9901285883701071, -4.589906353553811e-18},
    {0.15561499277355603, 8.886053372342288e-18, 0.9878177838164719, 4.91917302237681e-17},
    {0.17103002203139503, -9.954774726452923e-18, 0.9852658177182139, -4.925721262944555e-17},
    {0.18640329676226988, 2.3493796901281573e-18, 0.9824733131012553, -3.919920375420088e-17},
    {0.2017310638016388, 5.587232815460113e-18, 0.9794409517155483, 1.3108769521526758e-17},
    {0.21700958109501015, 1.1170071073364376e-17, 0.9761694738686353, -7.850690609285027e-18},
    {0.23223511861151147, -8.318080852687206e-18, 0.9726596782449127, 2.3920264546490165e-17},
    {0.24740395925452294, -7.53102495590706e-18, 0.9689124217106447, 5.071436662403936e-17},
    {0.2625123997691533, -2.2534597527902125e-17, 0.964928619104771, -3.0345542681018625e-18},
    {0.2775567516463363, 1.7674070262791822e-17, 0.9607092430155619, -2.807827063516729e-17},
    {0.29253334202332754, 7.516944930327352e-18, 0.9562553235431753, -3.148450868841629e-17},
    {0.30743851458038085, 1.1004366442765296e-19, 0.9515679480481722, -3.8614834675674123e-17},
    {0.3222686304333866, 2.093773358126606e-17, 0.9466482608860534, -3.911683334934152e-17},
    {0.33702006902225307, 1.0312279860787216e-17, 0.9414974631278811, -4.8523830236797095e-18},
    {0.3516892289948141, -2.5616208736069942e-17, 0.9361168122670553, -5.2350302039683216e-17},
    {0.36627252908604757, -9.938814562106524e-18, 0.9305076219123143, 4.488760003328074e-18},
    {0.38076640899239017, 2.1372528646211374e-17, 0.924671261467036, 5.5444125388034563e-17},
    {0.39516733024093426, -1.9613487871414228e-17, 0.9186091557949183, -4.0564150104514996e-17},
    {0.40947177705329507, -5.679403000091266e-18, 0.9123227848721178, 2.6349040211413332e-17},
    {0.42367625720393803, -2.331800700068871e-17, 0.9058136834259364, 4.2864666490805214e-17},
    {0.4377773028727551, 7.64345629962023e-18, 0.8990834405601384, 9.076951775075616e-18},
    {0.4517714714916838, -8.234073942098903e-18, 0.8921336993669944, 2.3160655211380166e-17},
    {0.46565534658516017, 1.459870391051426e-17, 0.8849661565261433, -7.690557775987357e-18},
    {0.479425538604203, -5.103969860556013e-18, 0.8775825618903728, -4.2623149864279997e-17},
    {0.49307868575392305, 5.605083973871755e-18, 0.8699847180584174, 1.657385110740923e-17},
    {0.5066114548142574, -3.269413423618168e-17, 0.8621744799348805, 4.4132427578105805e-18},
    {0.520020541953727, -3.983266745698455e-17, 0.8541537542773854, 5.420565102675286e-18},
    {0.5333026735360201, 5.129318115032044e-17, 0.8459244992310679, 1.549506647350329e-17},
    {0.5464546069192036, 8.399754840929507e-18, 0.8374887238505236, 4.3337026043948396e-17},
    {0.5594731312473669, 1.575565514488728e-17, 0.8288484876093257, 1.1163935406617444e-17},
    {0.5723550682345072, 2.6575872357215316e-17, 0.820005899897234, -3.912431748209128e-17},
    {0.5850972729404622, -5.4883972461161805e-17, 0.8109631195052179, -3.091333486122179e-17},
    {0.5976966345387015, 5.450323593054385e-17, 0.8017223540984184, 4.0134533311087014e-17},
    {0.6101500770757914, -1.479826990758988e-17, 0.7922858596771786, -2.9049779312834576e-17},
    {0.6224545602223437, -6.049035765709707e-18, 0.7826559400262728, -1.474071641211487e-17},
    {0.6346070800152693, -3.4568582392624965e-17, 0.7728349461524715, 4.231014921891023e-17},
    {0.6466046695911524, 4.567647714393289e-19, 0.7628252757105762, 1.6672995021546628e-17},
    {0.6584443999105676, -3.7736386700306717e-17, 0.7526293724180665, -1.2970993013150526e-17},
    {0.6701233804731629, 6.183536725574959e-18, 0.7422497254585013, -1.2339303604869521e-17},
    {0.6816387600233341, 4.410467313197903e-17, 0.7316888688738209, -1.0475824306512768e-17},
    {0.692987727246318, -5.3543290798909455e-17, 0.7209493809456964, 3.494986701478816e-17},
    {0.7041675114545337, -3.94095700584825e-17, 0.7100338835660797, 1.505272211891291e-17},
    {0.7151753832640076, -1.466099578328228e-17, 0.6989450415971057, -5.5261332036460915e-18},
};

// hi is the correctly rounded value of hi + lo + d for every |d| <= 2^-66 |hi|  (hi = RN(hi + lo))
RVLL_HD bool dd_rounds_safely(double hi, double lo)
{
    const double err = __builtin_fabs(hi) * 1.3563e-20;             // 2^-66 and a little
    return hi + (lo + err) == hi && hi + (lo - err) == hi;
}

// sin r, cos r for a double-double |r| <= pi/4 (+ a little); false: one of them is too close to a rounding boundary to call
RVLL_HD bool sincos_dd_table(DD r, double& s_out, double& c_out)
{
    const double kf = __builtin_rint(r.hi * 64.0);
    const int k = (int)kf;
    int ka = k < 0 ? -k : k;
    if (ka > 51) ka = 51;                                           // (|r| <= pi/4 + 2^-12 by the reductions: 50 at most)
    const DD h = dd_two_sum(r.hi - kf * 0.015625, r.lo);            // (the subtraction is exact)
    const double Sh = k < 0 ? -kSinCosTabDD[ka][0] : kSinCosTabDD[ka][0], Sl = k < 0 ? -kSinCosTabDD[ka][1] : kSinCosTabDD[ka][1];
    const double Ch = kSinCosTabDD[ka][2], Cl = kSinCosTabDD[ka][3];
    const DD q = dd_two_prod(h.hi, h.hi);
    const double z = q.hi;
    // sin h = h + p,  cos h - 1 = -q / 2 + w  (q = h^2 in double-double)
    const double p = (h.hi * z) * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, 2.7557319223985893e-06, -1.984126984126984e-04),
                                                                  8.333333333333333e-03), -1.6666666666666666e-01);
    const double w = (z * z) * __builtin_fma(z, __builtin_fma(z, 2.48015873015873e-05, -1.3888888888888889e-03), 4.1666666666666664e-02);
    const double sl = h.lo + p;                                     // sin h = (h.hi, sl)
    const double cth = -0.5 * q.hi, ctl = __builtin_fma(-0.5, q.lo + 2.0 * h.hi * h.lo, w);      // cos h - 1 = (cth, ctl)
    // sin r = S + (S (cos h - 1) + C sin h)
    {
        DD a = dd_two_prod(Ch, h.hi);
        a.lo += Ch * sl + Cl * h.hi;
        DD b = dd_two_prod(Sh, cth);
        b.lo += Sh * ctl + Sl * cth;
        DD u = dd_two_sum(a.hi, b.hi);
        u.lo += a.lo + b.lo;
        DD x = dd_two_sum(Sh, u.hi);
        x.lo += u.lo + Sl;
        x = dd_quick_sum(x.hi, x.lo);
        s_out = x.hi;
        if (!dd_rounds_safely(x.hi, x.lo)) return false;
    }
    // cos r = C + (C (cos h - 1) - S sin h)
    {
        DD a = dd_two_prod(-Sh, h.hi);
        a.lo -= Sh * sl + Sl * h.hi;
        DD b = dd_two_prod(Ch, cth);
        b.lo += Ch * ctl + Cl * cth;
        DD u = dd_two_sum(a.hi, b.hi);
        u.lo += a.lo + b.lo;
        DD x = dd_two_sum(Ch, u.hi);
        x.lo += u.lo + Cl;
        x = dd_quick_sum(x.hi, x.lo);
        c_out = x.hi;
        if (!dd_rounds_safely(x.hi, x.lo)) return false;
    }
    return true;
}

RVLL_HD void sincos_cr(double x, double& s_out, double& c_out)
{
    if (!(__builtin_fabs(x) < __builtin_inf())) { s_out = c_out = __builtin_nan(""); return; }
    DD r;
    uint32_t q;
    reduce_dd(x, r, q);
    double sr, cr;
#ifdef RVLL_CR_SERIES_ONLY              // (measurement / test builds: every value through the full series)
    sincos_dd_kernel(r, sr, cr);
#else
    if (__builtin_expect(!sincos_dd_table(r, sr, cr), 0)) sincos_dd_kernel(r, sr, cr);
#endif
    const double s = (q & 1u) ? cr : sr, c = (q & 1u) ? sr : cr;
    const double ss = (q & 2u) ? -s : s, cc = ((q + 1u) & 2u) ? -c : c;
    s_out = x < 0. ? -ss : ss;
    c_out = cc;
}

// n / d by reciprocal refinement.  v_rcp_f64 is accurate to 2^-24.4 (measured, scripts/rcp_probe.py): one
// Newton step gives 2^-48, and the quotient with one residual correction q + (n - d q) y is then the
// correctly rounded quotient for finite, normal operands with a normal quotient — bit for bit what `n / d`
// (hipcc: v_div_scale, v_rcp, two Newton steps, v_div_fmas, v_div_fixup; 58 cycles) returns, on 8e6 + 4e6
// operand pairs of the solver's range (tests/test_gpu_math.py), in 38 cycles.
RVLL_HD double recip_refined(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(y, e, y);
#else
    return 1.0 / d;
#endif
}
RVLL_HD double div_1nr(double n, double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
    const double q = n * y;
    return __builtin_fma(__builtin_fma(-d, q, n), y, q);
#else
    return n / d;
#endif
}
RVLL_HD double div_exact(double n, double d) { return div_1nr(n, d); }
RVLL_HD double div_fast(double n, double d) { return div_1nr(n, d); }

// Natural log of a finite positive double: argument split by v_frexp, f = m - 1 with
// m in [sqrt(1/2), sqrt(2)), s = f/(2+f), and the degree-14 even minimax in s (coefficients:
// Sun fdlibm e_log.c, public domain).  ~1 ulp; a third of the cost of the library log.
RVLL_HD double log_pos(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (!(v > 0.0 && v < __builtin_inf())) return log(v);          // zero, negative, inf, nan: library path
    double m = __builtin_amdgcn_frexp_mant(v);                    // [0.5, 1)
    int    k = __builtin_amdgcn_frexp_exp(v);
    const bool lowhalf = m < 7.07106781186547524401e-01;
    m = lowhalf ? m + m : m;
    k = lowhalf ? k - 1 : k;
    const double f = m - 1.0;
    const double s = div_fast(f, 2.0 + f);
    const double z = s * s;
    // one block of v_fma_f64 with the coefficients as third VGPR operands: hipcc lowers each Horner step to a
    // v_mov_b64 (coefficient copy) + v_fmac_f64 (same arithmetic, 6 more instructions per log)
    constexpr double L7 = 1.479819860511658591e-01, L6 = 1.531383769920937332e-01, L5 = 1.818357216161805012e-01,
                     L4 = 2.222219843214978396e-01, L3 = 2.857142874366239149e-01, L2 = 3.999999999940941908e-01,
                     L1 = 6.666666666666735130e-01;
    double R;
    asm("v_fma_f64 %0, %1, %2, %3\n\t"
        "v_fma_f64 %0, %1, %0, %4\n\t"
        "v_fma_f64 %0, %1, %0, %5\n\t"
        "v_fma_f64 %0, %1, %0, %6\n\t"
        "v_fma_f64 %0, %1, %0, %7\n\t"
        "v_fma_f64 %0, %1, %0, %8\n\t"
        "v_mul_f64 %0, %0, %1"
        : "=&v"(R)
        : "v"(z), "v"(L7), "v"(L6), "v"(L5), "v"(L4), "v"(L3), "v"(L2), "v"(L1));
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    // k ln2_hi - ((hfsq - (s (hfsq + R) + k ln2_lo)) - f)
    const double inner = __builtin_fma(s, hfsq + R, dk * 1.90821492927058770002e-10);
    return __builtin_fma(dk, 6.93147180369123816490e-01, -((hfsq - inner) - f));
#else
    return __builtin_log(v);
#endif
}

// Rotate (s, c) = (sin E0, cos E0) to E0 + h for a small step |h| <= 1e-3:
// sin h = h (1 - h^2/6), cos h - 1 = h^2 (-1/2 + h^2/24); the neglected terms are h^5/120 <= 8.4e-18 and
// h^6/720 <= 1.4e-21 at the bound (the Newton stop rule gives |h| <= 1e-4: 8e-23).
RVLL_HD void rotate_small(double h, double& s, double& c)
{
    const double h2 = h * h;
    const double sh = h * __builtin_fma(h2, -1.66666666666666666667e-01, 1.0);
    const double ch1 = h2 * __builtin_fma(h2, 4.16666666666666666667e-02, -0.5);
    // s' = s cos h + c sin h = s + (s*ch1 + c*sh);  c' = c + (c*ch1 - s*sh)
    const double s0 = s, c0 = c;
    s = s0 + __builtin_fma(s0, ch1, c0 * sh);
    c = c0 + __builtin_fma(c0, ch1, -(s0 * sh));
}

// (sin r, cos r) of the reduced argument -> (sin x, cos x) by the quadrant: swap and signs with bit operations (compare +
// v_cndmask cost ~4 cycles each, measured)
RVLL_HD void quadrant_f32(float sr, float cr, uint32_t uq, float& s_out, float& c_out)
{
    const uint32_t m = 0u - (uq & 1u);
    const uint32_t sb = __builtin_bit_cast(uint32_t, sr), cb = __builtin_bit_cast(uint32_t, cr);
    uint32_t rs, rc;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_bfi_b32 %0, %2, %4, %3\n\t"
        "v_bfi_b32 %1, %2, %3, %4"
        : "=&v"(rs), "=&v"(rc) : "v"(m), "v"(sb), "v"(cb));
#else
    rs = (cb & m) | (sb & ~m);
    rc = (sb & m) | (cb & ~m);
#endif
    rs ^= (uq & 2u) << 30;
    rc ^= ((uq + 1u) & 2u) << 30;
    s_out = __builtin_bit_cast(float, rs);
    c_out = __builtin_bit_cast(float, rc);
}

// ---- fp32 pieces of the reduced-precision modes (RVLL_PREC_MIXED / RVLL_PREC_FP32) ----------
// sin and cos of a float in roughly [-8, 8] (a mean anomaly already reduced to [-pi, pi] in fp64,
// plus Newton steps): one Cody-Waite step to [-pi/4, pi/4], degree-7/8 minimax kernels
// (coefficients: Sun/FreeBSD k_sinf.c, k_cosf.c), ~1 ulp(float).
RVLL_HD void sincos_f32(float x, float& s_out, float& c_out)
{
    const float fk = __builtin_rintf(x * 6.36619772367581382433e-01f);
    const int q = (int)fk;
    float r = __builtin_fmaf(-fk, 1.5707963109016418e+00f, x);       // pi/2 high part (24 bits)
    r = __builtin_fmaf(-fk, 1.5893254773528196e-08f, r);              // pi/2 - high
    const float z = r * r;
    float ps = __builtin_fmaf(z, 2.7183114939898219064e-06f, -1.9839334836096632576e-04f);
    ps = __builtin_fmaf(z, ps, 8.3333293858894631756e-03f);
    ps = __builtin_fmaf(z, ps, -1.6666666641626524e-01f);
    const float sr = __builtin_fmaf(r * z, ps, r);
    float pc = __builtin_fmaf(z, 2.4390448796277409065e-05f, -1.3886763774609929416e-03f);
    pc = __builtin_fmaf(z, pc, 4.1666623323739063189e-02f);
    pc = __builtin_fmaf(z, pc, -0.5f);
    const float cr = __builtin_fmaf(z, pc, 1.0f);
    quadrant_f32(sr, cr, (uint32_t)q, s_out, c_out);
}

// x reduced to [-pi, pi] in fp64 (two-constant Cody-Waite on 2*pi), returned as float.
RVLL_HD float reduce_2pi_to_f32(double x)
{
    constexpr double INV_TWOPI = 1.59154943091895335769e-01;
    constexpr double TWOPI_HI  = 6.28318530717958623200e+00;
    constexpr double TWOPI_LO  = 2.44929359829470641435e-16;
    constexpr double MAGIC     = 6755399441055744.0;
    const double fk = __builtin_fma(x, INV_TWOPI, MAGIC) - MAGIC;
    double r = __builtin_fma(-fk, TWOPI_HI, x);
    r = __builtin_fma(-fk, TWOPI_LO, r);
    return (float)r;
}

RVLL_HD float div_f32(float n, float d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rcpf(d);                          // 1 ulp
    const float q = n * y;
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), y, q);             // one residual step
#else
    return n / d;
#endif
}

// ---- the same, two at a time (the reduced-precision modes' item pairs, rvll_tile.h eval_item_pair) ---------------------------
// Two items per lane, their arithmetic in v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: a wave of single items ran one dependent
// chain per lane and was bound by the latency of that chain, not by the VALUs' rate (PMC: the fp32 kernel issued as many
// vector instructions as the fp64 one and each took as long; profiles/r04_precision_modes.txt).  Component k of every result
// is exactly what the scalar routine above returns for component k.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

RVLL_HD f32x2 splat2(float v) { return f32x2{v, v}; }
RVLL_HD f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

RVLL_HD void sincos_f32x2(f32x2 x, f32x2& s_out, f32x2& c_out)
{
    const f32x2 t = x * splat2(6.36619772367581382433e-01f);
    const f32x2 fk = {__builtin_rintf(t.x), __builtin_rintf(t.y)};
    const int q0 = (int)fk.x, q1 = (int)fk.y;
    f32x2 r = fma2(-fk, splat2(1.5707963109016418e+00f), x);
    r = fma2(-fk, splat2(1.5893254773528196e-08f), r);
    const f32x2 z = r * r;
    f32x2 ps = fma2(z, splat2(2.7183114939898219064e-06f), splat2(-1.9839334836096632576e-04f));
    ps = fma2(z, ps, splat2(8.3333293858894631756e-03f));
    ps = fma2(z, ps, splat2(-1.6666666641626524e-01f));
    const f32x2 sr = fma2(r * z, ps, r);
    f32x2 pc = fma2(z, splat2(2.4390448796277409065e-05f), splat2(-1.3886763774609929416e-03f));
    pc = fma2(z, pc, splat2(4.1666623323739063189e-02f));
    pc = fma2(z, pc, splat2(-0.5f));
    const f32x2 cr = fma2(z, pc, splat2(1.0f));
    float s0, c0, s1, c1;
    quadrant_f32(sr.x, cr.x, (uint32_t)q0, s0, c0);
    quadrant_f32(sr.y, cr.y, (uint32_t)q1, s1, c1);
    s_out = f32x2{s0, s1};
    c_out = f32x2{c0, c1};
}

RVLL_HD f32x2 div_f32x2(f32x2 n, f32x2 d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const f32x2 y = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 q = n * y;
    return fma2(fma2(-d, q, n), y, q);
#else
    return f32x2{n.x / d.x, n.y / d.y};
#endif
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
