// detmath.h -- deterministic double-precision log/exp for the engine's device code.
//
// hmmsearch (the reference's search stage, itsxpress/SeqSample.py:191-209) calls libm
// log()/exp() inside its per-target pipeline: one log per residue in the bias filter, one per
// rescaling event in Forward/Backward, exp() in the P-value tests.  Device libm is not
// bit-reproducible against a host libm, so these two functions are written with IEEE-754
// double add/mul/div and integer bit moves only (argument reduction + minimax polynomial,
// < 1 ulp), and the translation unit is built with -ffp-contract=off.  The same scheme,
// written independently, is what the CPU oracle uses; tests assert both agree bit-for-bit
// and agree with glibc after rounding to float.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ITSX_HD __host__ __device__ __forceinline__
#else
#define ITSX_HD inline
#endif

namespace itsx {

ITSX_HD uint64_t d2u(double x) { return __builtin_bit_cast(uint64_t, x); }
ITSX_HD double   u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

ITSX_HD double det_log(double x)
{
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  uint64_t u = d2u(x);
  int k = 0;
  if ((u << 1) == 0) return -__builtin_inf();
  if (u >> 63) return __builtin_nan("");
  if ((u >> 52) == 0x7ff) return x;
  if ((u >> 52) == 0) { x *= 18014398509481984.0; u = d2u(x); k -= 54; }
  uint32_t hx = (uint32_t)(u >> 32);
  hx += 0x3ff00000u - 0x3fe6a09eu;
  k += (int)(hx >> 20) - 0x3ff;
  hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
  u = ((uint64_t)hx << 32) | (u & 0xffffffffull);
  const double f = u2d(u) - 1.0;
  const double hfsq = 0.5 * f * f;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const double dk = (double)k;
  return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

ITSX_HD double det_exp(double x)
{
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
               P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
               P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 709.782712893383973096) return __builtin_inf();
  if (x < -745.13321910194110842) return 0.0;
  const double ax = x < 0 ? -x : x;
  double hi, lo, c, t, y;
  int k;
  if (ax > 0.34657359027997264) {
    k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
    t = (double)k;
    hi = x - t * ln2HI;
    lo = t * ln2LO;
    x = hi - lo;
  } else if (ax < 3.725290298461914e-09) {
    return 1.0 + x;
  } else {
    k = 0; hi = x; lo = 0.0;
  }
  t = x * x;
  c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
  y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021 && k <= 1023) return y * u2d((uint64_t)(0x3ff + k) << 52);
  if (k > 1023) return y * u2d((uint64_t)(0x3ff + (k - 1023)) << 52) * u2d((uint64_t)2046 << 52);
  return y * u2d((uint64_t)(0x3ff + (k + 1000)) << 52) * u2d((uint64_t)23 << 52);
}

ITSX_HD float det_logf(float x) { return (float)det_log((double)x); }

// (float)det_log((double)x), bit for bit, for the one place that takes a logarithm per residue (the bias filter).  det_log costs ~55
// double-precision operations, a division among them; but its result is only kept as a float.  So: a table-driven logarithm with a
// PROVEN error bound (128 intervals of the reduced argument m in [sqrt(1/2), sqrt(2)): r = m * inv_c - 1 with |r| <= 0.004, inv_c a
// double and logc = -log(inv_c) for exactly that double, log1p(r) to degree 5: 6.9e-16 truncation; k ln2 + logc + p: under
// 5e-16 |y| rounding in all, det_log's own error, < 1 ulp of y, included), and the float is taken from it only when every double
// within E = 1e-14 + 1e-15 |y| of the approximation rounds to the SAME float -- det_log's value is one of them.  Otherwise (about
// 4 in 10^7 arguments around |y| = 1, more often next to x = 1 where y itself is tiny) det_log decides.  ~27 operations, no division.
struct LogTab { double inv_c, logc; };
constexpr int LOGTAB_N = 128;
#if defined(__HIPCC__)
__host__
#endif
inline void build_logtab(LogTab *t)
{
  for (int i = 0; i < LOGTAB_N; i++) {
    const uint32_t h0 = 0x3fe6a09eu + (uint32_t)i * 0x2000u;
    const double m0 = u2d((uint64_t)h0 << 32), m1 = u2d((uint64_t)(h0 + 0x2000u) << 32);
    const double c = 0.5 * m0 + 0.5 * m1;
    t[i].inv_c = 1.0 / c;
    t[i].logc = (double)(-__builtin_logl((long double)t[i].inv_c));
  }
}
ITSX_HD float det_logf_fast(float xf, const LogTab *tab)
{
  const double x = (double)xf;
  const uint64_t u = d2u(x);
  uint32_t hx = (uint32_t)(u >> 32);
#ifdef ITSX_NO_FASTLOG                                   /* A/B builds (scripts/build_variant.sh): det_log for every argument */
  if (false) {
#else
  if ((hx - 0x00100000u) < 0x7fe00000u) {              // positive, normal, finite
#endif
    hx += 0x3ff00000u - 0x3fe6a09eu;
    const int k = (int)(hx >> 20) - 0x3ff;
    const uint32_t lo20 = hx & 0x000fffffu;
    const LogTab t = tab[lo20 >> 13];
    const double m = u2d(((uint64_t)(lo20 + 0x3fe6a09eu) << 32) | (u & 0xffffffffull));
    const double r = __builtin_fma(m, t.inv_c, -1.0);
    double p = __builtin_fma(r, 0.2, -0.25);
    p = __builtin_fma(r, p, 1.0 / 3.0);
    p = __builtin_fma(r, p, -0.5);
    p = __builtin_fma(r, p, 1.0);
    p = r * p;
    const double y = __builtin_fma((double)k, 6.93147180559945286227e-01, t.logc) + p;
    const double E = __builtin_fma(__builtin_fabs(y), 1e-15, 1e-14);
    const float lo = (float)(y - E), hi = (float)(y + E);
    if (lo == hi) return lo;
  }
  return (float)det_log(x);
}

// P-value tails, as Easel defines them
ITSX_HD double gumbel_surv(double x, double mu, double lambda)
{
  const double y = lambda * (x - mu);
  const double ey = -det_exp(-y);
  const double aey = ey < 0 ? -ey : ey;
  if (aey < 5e-9) return -ey;
  return 1 - det_exp(ey);
}
ITSX_HD double exp_surv(double x, double mu, double lambda)
{
  if (x < mu) return 1.0;
  return det_exp(-lambda * (x - mu));
}
ITSX_HD double exp_logsurv(double x, double mu, double lambda)
{
  if (x < mu) return 0.0;
  return -lambda * (x - mu);
}

}  // namespace itsx
