// f64 tanh(x/2) and log1p(x) for the BP kernel.
//
// The reference computes tanh with the platform libm and atanh as
// 0.5*ln_1p(2x/(1-x)) (belief_propagation.rs:154,158-159).  Neither is correctly
// rounded, so GPU and CPU differ by ulps whatever we call; what matters is
// ulp-level accuracy and IEEE behaviour at the edges (+-0, +-1, +-inf, NaN,
// arguments < -1), because the kernel has no clamps.  OCML's tanh and log1p cost
// 166 and 135 VALU instructions per call (double-double internals) and made the
// kernel issue-bound; these cost about 60 and 45 and round correctly about as often as the libms do
// (tests/test_gpu_bp_math.py, scripts/math_ulps.py).
#pragma once
#include <hip/hip_runtime.h>

namespace ldpc {

// A f64 constant whose low word is not zero needs a register pair (the hardware takes 32-bit literals only, which give
// a double's high word).  Left to the compiler, the two dozen constants of the polynomials below are built once at the
// top of the kernel and held in vector registers for its whole life, and where registers are short (bp_lds_kernel: 72)
// they are spilled and come back from scratch memory inside the check and bit updates.  With SK (scalar constants)
// BP_K(x) builds x where it is used instead: two scalar moves that cannot be hoisted (volatile), on the scalar unit's
// own issue port.  bp_lds_kernel asks for it (CSS example 150 -> 128 ms); bp_kernel, with 128 registers, measured 6 %
// slower with it (5 % with it in the bit update alone, the streamed form likewise) and keeps the compiler's choice.  The values are the same either way.  (Tried on the CSS example and
// C1, per 2^22 shots on one box, against 129 / 105.7 ms: fixed register pairs (one, or two or four taken in turn) so that a pair always
// goes into v_fma_f64 as it is, 141 / 101.8; a pair built by 64-bit scalar operations, 129.4 / 106.1 -- the scalar unit issues
// one instruction per SIMD turn like the vector unit and becomes the busier one; log1p of a check row's batch with the
// constants built once per batch, 149 / 105.9, more registers spilled; the compiler's own 64-bit scalar constants made
// opaque by an empty asm ("+s"), which go into v_fma_f64 without vector moves -- 13 vector instructions fewer per edge of a
// check row -- 130 / 109: the loops wait on their dependent chains, not on instruction issue.)
template <bool SK, unsigned long long BITS> __device__ inline double bp_k64()
{
    if constexpr (SK) {
        unsigned int hi, lo;
        asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "n"((unsigned int)BITS), "n"((unsigned int)(BITS >> 32)));
        return __hiloint2double((int)hi, (int)lo);
    } else
        return __builtin_bit_cast(double, BITS);
}
#define BP_K(x) (::ldpc::bp_k64<SK, __builtin_bit_cast(unsigned long long, (double)(x))>())

// a / b without the scaling and fix-up steps of the IEEE expansion (13-14 instructions): hardware
// reciprocal (about 23 bits), one Newton step, quotient, one residual correction.  The result is the
// correctly rounded quotient except in a vanishing fraction of near-tie cases (tests/test_gpu_bp_math.py)
// PROVIDED b is finite, non-zero and neither it, 1/b, a/b nor the residual leave the normal range;
// every call site below states why that holds or guards what does not.  A zero dividend gives +0 whatever
// its sign (the residual step adds +0): in BP the sign of a zero message reaches no other value or decision.
__device__ inline double bp_div_core(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

// true when |v| < 2^-511 (zero and denormals included).  NaN is NOT flagged: bp_div_core propagates a NaN operand
// as the IEEE division does, and the non-converging shots of a degenerate quantum code carry NaN messages through
// most of their 100 iterations (inf - inf in update_bits once tanh has saturated) -- flagged, they sent nearly
// every wave of the CSS workload through the slow division.
__device__ inline bool bp_exponent_below_band(double v) { return (uint32_t)(__double2hiint(v) & 0x7ff00000) < 0x20000000u; }

// prod / t of update_checks (belief_propagation.rs:158): |prod| <= |t| <= 1 (prod has t as a factor), or NaN.
// Fast path when t is in the normal band or NaN; a lane below it (t == 0 gives the reference's 0/0 = NaN,
// denormal t) sends the wave through the IEEE division.
__device__ inline double bp_div_prod(double prod, double t)
{
    double q = bp_div_core(prod, t);
    if (__builtin_expect(__ballot(bp_exponent_below_band(t)) != 0, 0)) q = prod / t;
    return q;
}

// tanh(x/2) = (1 - E) / (1 + E), E = e^-|x|.  A plain double E is not enough (its half-ulp error is
// amplified up to 4x near |x| = 0.5, and 1 - E, 1 + E and the quotient each round once more: about half
// of the results came out mis-rounded, which makes non-converging BP runs drift away from the CPU's
// faster than libm-to-libm differences do).  So E is carried as an unevaluated sum Eh + El (Cody-Waite
// reduction |x| = k ln2 - r, e^r = 1 + r + r^2 Q(r) with the rounding errors of the last additions kept),
// numerator and denominator likewise, and the quotient gets one correction step.  For k = 0 the same
// formulas give 1 - E = -(expm1 part) without cancellation, so one path serves every x.
// Measured: <= 0.55 ulp for |x| >= 0.5, <= 0.64 ulp below (the uncompensated rounding of r^2 Q(r) shows
// when r is near ln2/2), 96-100 % correctly rounded; glibc's tanh: <= 1.15 ulp, about 80 %.
template <bool SK = false> __device__ inline double bp_tanh_half(double x)
{
    const double ax = fabs(x);
    const double a = ax > 64.0 ? 64.0 : ax;       // tanh(32) is 1.0 in f64 already; keeps 2^-k normal; NaN stays NaN
    const double kn = rint(a * BP_K(-1.4426950408889634074));   // -k
    const double rh = -(kn * BP_K(6.93147180369123816490e-01)) - a;   // k ln2_hi - a, exact (Cody-Waite, Sterbenz)
    const double rl = -(kn * BP_K(1.90821492927058770002e-10));      // k ln2_lo
    const double r = rh + rl;
    const double rc = (rh - r) + rl;              // r + rc = rh + rl
    double q = BP_K(1.0 / 6227020800.0);          // 1/13!
    q = fma(q, r, BP_K(1.0 / 479001600.0));
    q = fma(q, r, BP_K(1.0 / 39916800.0));
    q = fma(q, r, BP_K(1.0 / 3628800.0));
    q = fma(q, r, BP_K(1.0 / 362880.0));
    q = fma(q, r, BP_K(1.0 / 40320.0));
    q = fma(q, r, BP_K(1.0 / 5040.0));
    q = fma(q, r, BP_K(1.0 / 720.0));
    q = fma(q, r, BP_K(1.0 / 120.0));
    q = fma(q, r, BP_K(1.0 / 24.0));
    q = fma(q, r, BP_K(1.0 / 6.0));
    q = fma(q, r, 0.5);
    const double r2q = (r * r) * q;               // e^r - 1 - r
    const double ph = r + r2q;
    const double pl = ((r - ph) + r2q) + fma(rc, ph, rc);   // low part of e^(r+rc) - 1
    const double sh = 1.0 + ph;
    const double sl = ((1.0 - sh) + ph) + pl;     // e^(r+rc) = sh + sl
    const int k = (int)kn;
    const double eh = ldexp(sh, k), el = ldexp(sl, k);      // E = e^-a
    const double nh = 1.0 - eh, nl = ((1.0 - nh) - eh) - el;
    const double dh = 1.0 + eh, dl = ((1.0 - dh) + eh) + el;
    // (nh + nl) / (dh + dl): dh in [1, 2], 0 <= nh <= 1, no special operands.  For tiny |x| nh holds
    // few bits (or none) and the residual step carries the result: x = 1e-300 gives q0 = 0, rem = x
    double rd = __builtin_amdgcn_rcp(dh);
    rd = fma(fma(-dh, rd, 1.0), rd, rd);
    const double q0 = nh * rd;
    const double rem = fma(-q0, dh, nh) + (nl - q0 * dl);
    const double big = fma(rem, rd, q0);
    // tanh is odd: put x's sign back (keeps -0; a NaN stays a NaN)
    return copysign(big, x);
}

// bp_tanh_half of the first cnt of N values, statement by statement across the batch: every value goes through the
// operations of bp_tanh_half in their order (the same results, bit for bit), and a constant of the polynomial is
// built once per batch instead of once per value.
// Round 2 tried this batch for bits of ANY degree up to eight in bp_lds_kernel's general bit update and dropped it after
// a wrong result in one irregular code (a build with 73 spilled registers).  Round 3 rebuilt that variant
// (-DBPL_TANH_BATCH_ANY=1: count = the lane's own degree; =2: the wave's largest) and could not make it fail: on the
// seven irregular codes of the suite (bit degrees 0 to 14, 35 cases each, scripts/tanh_batch_repro.py) it is
// bit-identical to the one-by-one form and to bp_kernel, and the routine itself equals bp_tanh_half bit for bit for every
// count, uniform or per lane, with either kind of constant (tests/test_gpu_bp_math.py::test_tanh_batch_equals_the_single_form;
// elements from the count on are never read or written).  So the routine has no predicate or indexing fault, and the
// variant as rebuilt here (values beyond a bit's degree zeroed before the batch, results stored under `k < dv`) is
// correct; the round-2 build that failed is not in the tree and its fault cannot be examined further.  The variant
// stays out for its cost alone: seven arrays of eight doubles per lane spill.
template <int N, bool SK> __device__ inline void bp_tanh_half_n(const double (&x)[N], double (&out)[N], int cnt)
{
    double a[N], kn[N], rh[N], rl[N], r[N], rc[N], q[N];
#pragma unroll
    for (int j = 0; j < N; j++)
        if (j < cnt) {
            const double ax = fabs(x[j]);
            a[j] = ax > 64.0 ? 64.0 : ax;
        }
    {
        const double c = BP_K(-1.4426950408889634074);
#pragma unroll
        for (int j = 0; j < N; j++)
            if (j < cnt) kn[j] = rint(a[j] * c);
    }
    {
        const double c = BP_K(6.93147180369123816490e-01);
#pragma unroll
        for (int j = 0; j < N; j++)
            if (j < cnt) rh[j] = -(kn[j] * c) - a[j];
    }
    {
        const double c = BP_K(1.90821492927058770002e-10);
#pragma unroll
        for (int j = 0; j < N; j++)
            if (j < cnt) {
                rl[j] = -(kn[j] * c);
                r[j] = rh[j] + rl[j];
                rc[j] = (rh[j] - r[j]) + rl[j];
            }
    }
    {
        const double c = BP_K(1.0 / 6227020800.0);
#pragma unroll
        for (int j = 0; j < N; j++)
            if (j < cnt) q[j] = c;
    }
#define BP_TANH_STEP(K)                                   \
    {                                                     \
        const double c = BP_K(K);                         \
        _Pragma("unroll") for (int j = 0; j < N; j++)     \
            if (j < cnt) q[j] = fma(q[j], r[j], c);       \
    }
    BP_TANH_STEP(1.0 / 479001600.0)
    BP_TANH_STEP(1.0 / 39916800.0)
    BP_TANH_STEP(1.0 / 3628800.0)
    BP_TANH_STEP(1.0 / 362880.0)
    BP_TANH_STEP(1.0 / 40320.0)
    BP_TANH_STEP(1.0 / 5040.0)
    BP_TANH_STEP(1.0 / 720.0)
    BP_TANH_STEP(1.0 / 120.0)
    BP_TANH_STEP(1.0 / 24.0)
    BP_TANH_STEP(1.0 / 6.0)
#undef BP_TANH_STEP
#pragma unroll
    for (int j = 0; j < N; j++)
        if (j < cnt) {
            const double qq = fma(q[j], r[j], 0.5);
            const double r2q = (r[j] * r[j]) * qq;
            const double ph = r[j] + r2q;
            const double pl = ((r[j] - ph) + r2q) + fma(rc[j], ph, rc[j]);
            const double sh = 1.0 + ph;
            const double sl = ((1.0 - sh) + ph) + pl;
            const int k = (int)kn[j];
            const double eh = ldexp(sh, k), el = ldexp(sl, k);
            const double nh = 1.0 - eh, nl = ((1.0 - nh) - eh) - el;
            const double dh = 1.0 + eh, dl = ((1.0 - dh) + eh) + el;
            double rd = __builtin_amdgcn_rcp(dh);
            rd = fma(fma(-dh, rd, 1.0), rd, rd);
            const double q0 = nh * rd;
            const double rem = fma(-q0, dh, nh) + (nl - q0 * dl);
            const double big = fma(rem, rd, q0);
            out[j] = copysign(big, x[j]);
        }
}

// log1p(x), argument reduction and kernel polynomial of the classic fdlibm scheme
// (1+x = 2^k (1+f), sqrt(1/2) < 1+f < sqrt(2); log(1+f) = f - f^2/2 + s (f^2/2 + R(s^2)), s = f/(2+f);
// the rounding error of 1+x enters as the correction c), written branch-free for a 64-lane wave;
// zeros, infinities, NaN and x <= -1 are patched in a block a wave enters only if one of its lanes
// holds such an argument.
template <bool SK = false> __device__ inline double bp_log1p_main(double x)
{
    const double u = 1.0 + x;
    int hu = __double2hiint(u);
    int k = (hu >> 20) - 1023;
    // rounding error of 1 + x, relative to u (a few digits suffice when |c| <= ulp(u) is small against the result)
    double c = (fmin(x, 1.0) - (u - fmax(x, 1.0))) * __builtin_amdgcn_rcp(u);   // Fast2Sum, larger operand first
    hu &= 0x000fffff;
    const bool low = hu < 0x6a09e;                       // mantissa below sqrt(2)
    k += low ? 0 : 1;
    const double um = __hiloint2double(hu | (low ? 0x3ff00000 : 0x3fe00000), __double2loint(u));
    double f = um - 1.0;
    // |x| < 2^-20: 1 + x keeps too few bits of x for c (known to 23 bits only) to make up; f = x exactly
    if (fabs(x) < 0x1p-20) { f = x; c = 0.0; }
    const double hfsq = 0.5 * f * f;
    const double s = bp_div_core(f, 2.0 + f);            // 2 + f in [1.70, 2.42], |f| < 0.42 (or 0)
    const double z = s * s;
    double r = BP_K(1.479819860511658591e-01);      // Lp7
    r = fma(r, z, BP_K(1.531383769920937332e-01));
    r = fma(r, z, BP_K(1.818357216161805012e-01));
    r = fma(r, z, BP_K(2.222219843214978396e-01));
    r = fma(r, z, BP_K(2.857142874366239149e-01));
    r = fma(r, z, BP_K(3.999999999940941908e-01));
    r = fma(r, z, BP_K(6.666666666666735130e-01));  // Lp1
    r = r * z;
    const double dk = (double)k;
    return dk * BP_K(6.93147180369123816490e-01) - ((hfsq - (s * (hfsq + r) + (dk * BP_K(1.90821492927058770002e-10) + c))) - f);   // ln2_hi, ln2_lo
}

// arguments bp_log1p_main does not serve: -inf, +-0, +inf (class mask 0x264) and x <= -1.  (A NaN comes out of
// bp_log1p_main as a NaN -- through c -- and is not flagged, for the reason given at bp_exponent_below_band.)
__device__ inline bool bp_log1p_is_special(double x) { return __builtin_amdgcn_class(x, 0x264) || x <= -1.0; }

__device__ inline double bp_log1p_fix(double x, double res)
{
    if (x == 0.0) res = x;   // keeps -0
    if (x == -1.0) res = -__builtin_huge_val();
    if (x < -1.0) res = __builtin_nan("");
    if (x == __builtin_huge_val()) res = x;
    if (x != x) res = x;
    return res;
}

template <bool SK = false> __device__ inline double bp_log1p(double x)
{
    double res = bp_log1p_main<SK>(x);
    if (__builtin_expect(__ballot(bp_log1p_is_special(x)) != 0, 0)) res = bp_log1p_fix(x, res);
    return res;
}

// f64::atanh of the reference's toolchain, 0.5 * ln_1p(2x / (1 - x)) (belief_propagation.rs:159), for
// |x| <= 1 or NaN (x is a product of tanh values divided by one of its factors): 1 - x is 0 (x == 1:
// the quotient is 2/0 = +inf), or lies in [2^-53, 2], where the fast division applies.
// bp_atanh_arg: argument of ln_1p, 2x / (1 - x).  x == 1 (every other tanh of the check saturated, the usual state
// of a shot that does not converge) is patched by a select rather than a branch to the IEEE division.
__device__ inline double bp_atanh_arg(double x)
{
    const double q = bp_div_core(2.0 * x, 1.0 - x);
    return x == 1.0 ? __builtin_huge_val() : q;
}

template <bool SK = false> __device__ inline double bp_ref_atanh(double x) { return 0.5 * bp_log1p<SK>(bp_atanh_arg(x)); }

}  // namespace ldpc
