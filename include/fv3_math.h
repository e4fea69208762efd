/* fv3_math.h -- ONE deterministic exp / log for the column kernels AND their checker.
 *
 * The reference forms every pressure power as exp(kappa*log(p)) with the compiler's intrinsics
 * (model/nh_utils.F90:1299,1382-1392, model/nh_core.F90:140-159, model/fv_mapz.F90:212-219,
 * model/dyn_core.F90:2288-2320); two correct libm's differ from each other in the last bit, and a
 * last-bit difference in pk / pkz / peln flips limiter branches of the remap on vertically uniform
 * states.  So that the HIP kernels (csrc/nh_kernels.h, csrc/remap_kernels.h) and the C oracle
 * (oracle/nh_core.c, oracle/mapz.c) are comparable bit for bit, both evaluate the SAME sequence of
 * IEEE-754 operations: +, -, *, / and fused multiply-add (each correctly rounded on gfx950 and on
 * x86-64; no contraction is left to a compiler -- every fma is explicit), integer arithmetic on
 * the bit patterns.  No tables, no calls into a vendor math library.
 *
 * Accuracy: < 1 ulp on the whole finite range (tests/test_fv3_math.py checks 2e5 arguments against
 * 60-digit decimal arithmetic), i.e. as good an "exp" / "log" as any intrinsic the reference is
 * built with.
 *
 * Plain C99 / C++17.  Define FV3M_FN before including to add qualifiers (the HIP side uses
 * `__host__ __device__ static inline`).
 */
#ifndef FV3_MATH_H
#define FV3_MATH_H

#ifndef FV3M_FN
#define FV3M_FN static inline
#endif

FV3M_FN long long fv3m_bits(double x) {
  long long i;
  __builtin_memcpy(&i, &x, 8);
  return i;
}
FV3M_FN double fv3m_from_bits(long long i) {
  double x;
  __builtin_memcpy(&x, &i, 8);
  return x;
}

/* exp(x): x = k ln2 + r, |r| <= ln2/2; e^r = 1 + (r + r^2 q(r)), q = Taylor sum_{n=2..13} r^(n-2)/n!
 * (truncation r^14/14! < 5e-18); result scaled by 2^k in two exact steps. */
FV3M_FN double fv3_exp(double x) {
  const double INV_LN2 = 0x1.71547652b82fep+0;
  const double LN2_HI = 0x1.62e42fefa39efp-1, LN2_LO = 0x1.abc9e3b39803fp-56;
  const double SHIFT = 0x1.8p52; /* adding and subtracting it rounds to the nearest integer (ties to even) */
  if (!(x == x)) return x;
  if (x > 0x1.62e42fefa39efp+9) return fv3m_from_bits(0x7ff0000000000000LL); /* > 1024 ln2: +inf */
  if (x < -0x1.74910d52d3052p+9) return 0.0;                                  /* < -1075 ln2 */
  const double t = x * INV_LN2 + SHIFT;
  const double kd = t - SHIFT;
  const int k = (int)kd;
  double r = __builtin_fma(-kd, LN2_HI, x);
  r = __builtin_fma(-kd, LN2_LO, r);
  double q = 0x1.6124613a86d09p-33;           /* 1/13! */
  q = __builtin_fma(q, r, 0x1.1eed8eff8d898p-29); /* 1/12! */
  q = __builtin_fma(q, r, 0x1.ae64567f544e4p-26);
  q = __builtin_fma(q, r, 0x1.27e4fb7789f5cp-22);
  q = __builtin_fma(q, r, 0x1.71de3a556c734p-19);
  q = __builtin_fma(q, r, 0x1.a01a01a01a01ap-16);
  q = __builtin_fma(q, r, 0x1.a01a01a01a01ap-13);
  q = __builtin_fma(q, r, 0x1.6c16c16c16c17p-10);
  q = __builtin_fma(q, r, 0x1.1111111111111p-7);
  q = __builtin_fma(q, r, 0x1.5555555555555p-5);
  q = __builtin_fma(q, r, 0x1.5555555555555p-3);
  q = __builtin_fma(q, r, 0.5);
  const double y = 1.0 + __builtin_fma(r * r, q, r);
  const int k1 = k >> 1, k2 = k - k1; /* both in the normal exponent range for every k in [-1075, 1024] */
  return y * fv3m_from_bits((long long)(1023 + k1) << 52) * fv3m_from_bits((long long)(1023 + k2) << 52);
}

/* log(x): x = 2^k m, m in [sqrt(1/2), sqrt(2)), f = m - 1 (exact), s = f/(2+f), z = s^2:
 *   log(m) = 2 atanh(s) = 2s + s R(z),  R(z) = sum_{n>=1} 2/(2n+1) z^n  (11 terms: z <= 0.0295, tail < 1e-18)
 * and, because 2s = f - s f and f^2/2 (1 - s) = s f:   log(m) = f - (f^2/2 - s (f^2/2 + R)).
 * LN2_HI carries 32 significant bits, so k*LN2_HI is exact. */
FV3M_FN double fv3_log(double x) {
  const double LN2_HI = 0x1.62e42feep-1, LN2_LO = 0x1.a39ef35793c76p-33;
  long long ix = fv3m_bits(x);
  int k = 0;
  if (ix < 0x0010000000000000LL) { /* zero, subnormal or negative */
    if ((ix & 0x7fffffffffffffffLL) == 0) return fv3m_from_bits((long long)0xfff0000000000000ULL); /* -inf */
    if (ix < 0) return fv3m_from_bits(0x7ff8000000000000LL);                                       /* NaN */
    x *= 0x1p54;
    ix = fv3m_bits(x);
    k = -54;
  } else if (ix >= 0x7ff0000000000000LL) {
    return x; /* +inf, NaN */
  }
  const long long tmp = ix - 0x3fe6a09e667f3bcdLL;
  k += (int)(tmp >> 52);
  const double m = fv3m_from_bits(ix - (long long)((unsigned long long)tmp & 0xfff0000000000000ULL));
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s, w = z * z;
  double t1 = 0x1.8618618618618p-4;                /* 2/21 */
  t1 = __builtin_fma(t1, w, 0x1.e1e1e1e1e1e1ep-4); /* 2/17 */
  t1 = __builtin_fma(t1, w, 0x1.3b13b13b13b14p-3); /* 2/13 */
  t1 = __builtin_fma(t1, w, 0x1.c71c71c71c71cp-3); /* 2/9 */
  t1 = __builtin_fma(t1, w, 0x1.999999999999ap-2); /* 2/5 */
  double t2 = 0x1.642c8590b2164p-4;                /* 2/23 */
  t2 = __builtin_fma(t2, w, 0x1.af286bca1af28p-4); /* 2/19 */
  t2 = __builtin_fma(t2, w, 0x1.1111111111111p-3); /* 2/15 */
  t2 = __builtin_fma(t2, w, 0x1.745d1745d1746p-3); /* 2/11 */
  t2 = __builtin_fma(t2, w, 0x1.2492492492492p-2); /* 2/7 */
  t2 = __builtin_fma(t2, w, 0x1.5555555555555p-1); /* 2/3 */
  const double R = __builtin_fma(t1, w, t2 * z);
  const double hfsq = 0.5 * f * f;
  const double kd = (double)k;
  return kd * LN2_HI - ((hfsq - __builtin_fma(s, hfsq + R, kd * LN2_LO)) - f);
}

#endif /* FV3_MATH_H */
