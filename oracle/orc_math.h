/*
 * orc_math.h -- deterministic log/exp used by the ORACLE (test infrastructure only).
 *
 * The reference path (hmmsearch, HMMER 3.1b2..3.4; called from
 * itsxpress/SeqSample.py:191-209) calls libm log()/exp() inside its per-target
 * pipeline.  libm results are not reproducible bit-for-bit on a GPU, so both the
 * oracle and the HIP engine use the same argument-reduction + polynomial scheme
 * (the classic Sun/fdlibm construction), written with plain IEEE-754 double
 * add/mul/div only (compile with -ffp-contract=off).  tests/test_detmath.py
 * checks it against glibc: <= 1 ulp in double, identical after rounding to
 * float on all sampled inputs.
 *
 * Nothing in the shipped product includes this file; the engine has its own copy
 * of the same scheme in itsxpress_amd/csrc/detmath.h and a test asserts the two
 * agree bit-for-bit.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H
#include <stdint.h>
#include <string.h>

static inline uint64_t orc_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double   orc_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* natural log of a double; x>0 finite normal or subnormal; 0 -> -inf; <0 -> nan */
static inline double orc_log(double x)
{
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  uint64_t u = orc_d2u(x);
  int k = 0;
  if ((u << 1) == 0) return -1.0 / 0.0;            /* +-0 */
  if (u >> 63) return 0.0 / 0.0;                   /* negative */
  if ((u >> 52) == 0x7ff) return x;                /* inf / nan */
  if ((u >> 52) == 0) {                            /* subnormal: scale up by 2^54 */
    x *= 18014398509481984.0; u = orc_d2u(x); k -= 54;
  }
  /* normalise mantissa into [sqrt(2)/2, sqrt(2)) */
  uint32_t hx = (uint32_t)(u >> 32);
  hx += 0x3ff00000 - 0x3fe6a09e;
  k += (int)(hx >> 20) - 0x3ff;
  hx = (hx & 0x000fffff) + 0x3fe6a09e;
  u = ((uint64_t)hx << 32) | (u & 0xffffffffu);
  double m = orc_u2d(u);
  double f = m - 1.0;
  double hfsq = 0.5 * f * f;
  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  double R = t2 + t1;
  double dk = (double)k;
  return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

/* exp of a double; handles overflow/underflow to inf/0 */
static inline double orc_exp(double x)
{
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
               P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
               P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 709.782712893383973096) return 1.0 / 0.0;
  if (x < -745.13321910194110842) return 0.0;
  double ax = x < 0 ? -x : x;
  double hi, lo, c, t, y;
  int k;
  if (ax > 0.34657359027997264 /* 0.5 ln2 */) {
    k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
    t = (double)k;
    hi = x - t * ln2HI;
    lo = t * ln2LO;
    x = hi - lo;
  } else if (ax < 3.725290298461914e-09 /* 2^-28 */) {
    return 1.0 + x;
  } else {
    k = 0; hi = x; lo = 0.0;
  }
  t = x * x;
  c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
  y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  /* scale by 2^k, in two steps so that subnormal results round once */
  if (k >= -1021 && k <= 1023) {
    return y * orc_u2d((uint64_t)(0x3ff + k) << 52);
  } else if (k > 1023) {
    return y * orc_u2d((uint64_t)(0x3ff + (k - 1023)) << 52) * orc_u2d((uint64_t)2046 << 52) /* 2^1023 */;
  } else {
    return y * orc_u2d((uint64_t)(0x3ff + (k + 1000)) << 52) * orc_u2d((uint64_t)23 << 52) /* 2^-1000 */;
  }
}

static inline float orc_logf(float x) { return (float)orc_log((double)x); }

#endif
