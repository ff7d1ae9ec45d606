// f64 tanh(x/2) and log1p(x) for the BP kernel.
//
// The reference computes tanh with the platform libm and atanh as
// 0.5*ln_1p(2x/(1-x)) (belief_propagation.rs:154,158-159).  Neither is correctly
// rounded, so GPU and CPU differ by ulps whatever we call; what matters is
// ulp-level accuracy and IEEE behaviour at the edges (+-0, +-1, +-inf, NaN,
// arguments < -1), because the kernel has no clamps.  OCML's tanh and log1p cost
// 166 and 135 VALU instructions per call (double-double internals) and made the
// kernel issue-bound; these cost about 85 and 65 and round correctly about as often as the libms do
// (tests/test_gpu_bp_math.py, scripts/math_ulps.py).
#pragma once
#include <hip/hip_runtime.h>

namespace ldpc {

// tanh(x/2).
// |x| < 0.5: odd Taylor series of tanh in h = x/2 (|h| < 0.25, 12 terms, truncation < 1e-19).
// |x| >= 0.5: (1 - E) / (1 + E), E = e^-|x|.  A plain double E is not enough (its half-ulp error is
// amplified up to 4x near |x| = 0.5, and 1 - E, 1 + E and the quotient each round once more: about half
// of the results came out mis-rounded, which makes non-converging BP runs drift away from the CPU's
// faster than libm-to-libm differences do).  So E is carried as an unevaluated sum Eh + El (Cody-Waite
// reduction |x| = k ln2 - r, e^r = 1 + r + r^2 Q(r) with the rounding errors of the last additions kept),
// numerator and denominator likewise, and the quotient gets one correction step.
__device__ inline double bp_tanh_half(double x)
{
    const double a = fmin(fabs(x), 64.0);   // tanh(32) is 1.0 in f64 already; keeps 2^-k normal
    // --- big path ---
    const double kd = rint(a * 1.4426950408889634074);
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double rh = kd * ln2_hi - a;            // exact (Cody-Waite, Sterbenz)
    const double rl = kd * ln2_lo;
    const double r = rh + rl;
    const double rc = (rh - r) + rl;              // r + rc = rh + rl
    double q = 1.0 / 6227020800.0;                // 1/13!
    q = fma(q, r, 1.0 / 479001600.0);
    q = fma(q, r, 1.0 / 39916800.0);
    q = fma(q, r, 1.0 / 3628800.0);
    q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0);
    q = fma(q, r, 1.0 / 5040.0);
    q = fma(q, r, 1.0 / 720.0);
    q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    const double r2q = (r * r) * q;               // e^r - 1 - r
    const double ph = r + r2q;
    const double pl = ((r - ph) + r2q) + fma(rc, ph, rc);   // low part of e^(r+rc) - 1
    const double sh = 1.0 + ph;
    const double sl = ((1.0 - sh) + ph) + pl;     // e^(r+rc) = sh + sl
    const int k = -(int)kd;
    const double eh = ldexp(sh, k), el = ldexp(sl, k);      // E = e^-a
    const double nh = 1.0 - eh, nl = ((1.0 - nh) - eh) - el;
    const double dh = 1.0 + eh, dl = ((1.0 - dh) + eh) + el;
    const double q0 = nh / dh;
    const double rem = fma(-q0, dh, nh) + (nl - q0 * dl);
    const double big = q0 + rem * __builtin_amdgcn_rcp(dh);
    // --- small path ---
    const double h = 0.5 * x, z = h * h;
    double p = -0x1.497d8eea25259p-15;
    p = fma(p, z, 0x1.967e18afcafadp-14);
    p = fma(p, z, -0x1.f57d7734d1664p-13);
    p = fma(p, z, 0x1.3558248036744p-11);
    p = fma(p, z, -0x1.7da36452b75e3p-10);
    p = fma(p, z, 0x1.d6d3d0e157de0p-9);
    p = fma(p, z, -0x1.226e355e6c23dp-7);
    p = fma(p, z, 0x1.664f4882c10fap-6);
    p = fma(p, z, -0x1.ba1ba1ba1ba1cp-5);
    p = fma(p, z, 0x1.1111111111111p-3);
    p = fma(p, z, -0x1.5555555555555p-2);
    const double small = fma(h * z, p, h);   // h + h^3 * P(h^2)
    // tanh is odd: take the magnitude and put x's sign back (keeps -0)
    const double t = copysign(a < 0.5 ? small : big, x);
    return x != x ? x : t;
}

// log1p(x), argument reduction and kernel polynomial of the classic fdlibm scheme
// (1+x = 2^k (1+f), sqrt(1/2) < 1+f < sqrt(2); log(1+f) = f - f^2/2 + s (f^2/2 + R(s^2)), s = f/(2+f)),
// written branch-free for a 64-lane wave.
__device__ inline double bp_log1p(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    // -0.2929 < x < 0.4142: k = 0 and f = x exactly
    const bool direct = x > -0.2928932188134524 && x < 0.41421356237309515;
    const double u = 1.0 + x;
    int hu = __double2hiint(u);
    int k = (hu >> 20) - 1023;
    // rounding error of 1 + x, relative to u (a few digits suffice: |c| <= ulp(u))
    double c = (k > 0 ? 1.0 - (u - x) : x - (u - 1.0)) * __builtin_amdgcn_rcp(u);
    hu &= 0x000fffff;
    const bool low = hu < 0x6a09e;                       // mantissa below sqrt(2)
    k += low ? 0 : 1;
    const double um = __hiloint2double(hu | (low ? 0x3ff00000 : 0x3fe00000), __double2loint(u));
    double f = um - 1.0;
    if (direct) { k = 0; f = x; c = 0.0; }
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double r = Lp7;
    r = fma(r, z, Lp6);
    r = fma(r, z, Lp5);
    r = fma(r, z, Lp4);
    r = fma(r, z, Lp3);
    r = fma(r, z, Lp2);
    r = fma(r, z, Lp1);
    r = r * z;
    const double dk = (double)k;
    double res = dk * ln2_hi - ((hfsq - (s * (hfsq + r) + (dk * ln2_lo + c))) - f);
    // edges: log1p(-1) = -inf, log1p(x < -1) = NaN, log1p(+inf) = +inf, NaN stays NaN
    if (x == 0.0) res = x;   // keeps -0
    if (x == -1.0) res = -__builtin_huge_val();
    if (x < -1.0) res = __builtin_nan("");
    if (x == __builtin_huge_val()) res = x;
    if (x != x) res = x;
    return res;
}

}  // namespace ldpc
