/* Portable transcendentals of the PARITY arithmetic ("PM"): sin, cos, log, exp, pow, acos, atan2 of a float, computed
 * in double precision from +, -, *, /, sqrt, fusedMultiplyAdd, floor and rint only -- operations IEEE 754 defines bit for bit -- and rounded
 * to float once.  The same text is compiled into the CPU oracle (oracle/pm_math.h) and into the HIP kernels
 * (raymarching-engine_amd/csrc/rm_pm_math.hpp; tests/test_host_cpu.py checks that the two files agree), both with
 * -ffp-contract=off, so the strict build and the oracle agree on every transcendental bit for bit, on any libm / ocml.
 * (GLSL leaves the precision of these functions to the implementation -- the reference's own results differ between
 * drivers -- so there is no reference bit pattern to match; the goldens pin the oracle to SwiftShader's within the
 * tolerances of tests/test_oracle_golden.py.)  Accuracy: the double result is within ~1e-13 relative of the true value
 * (series truncated below 1e-13; arguments of sin/cos up to ~1e6 in magnitude), so the float is the correctly rounded
 * one except for about one argument in 10^5.  Series coefficients are the exact Taylor / Gregory rationals, not fitted.
 *
 * (Round 4: the Horner steps of the series are fused multiply-adds -- half the double-precision instructions on the GPU.)
 * The including file defines PM_FN (function qualifiers), PM_FMA / PM_FMAK (IEEE fusedMultiplyAdd of doubles; K: the addend is a constant), PM_D2U / PM_U2D (bit casts double <-> 64-bit unsigned) and
 * PM_F2U (float -> 32-bit unsigned). */

#define PM_PIO2_HI 1.5707963267341256      /* the first 31 bits of pi/2: k * PM_PIO2_HI is exact for |k| < 2^22 */
#define PM_PIO2_LO 6.077100506506192e-11   /* pi/2 - PM_PIO2_HI */
#define PM_LN2_HI 0.6931471803691238       /* the first 32 bits of ln 2 */
#define PM_LN2_LO 1.9082149292705877e-10   /* ln 2 - PM_LN2_HI */
#define PM_PI 3.141592653589793
#define PM_PI_2 1.5707963267948966
#define PM_PI_4 0.7853981633974483

/* sin and cos of a double: exact reduction for |x| < 3e6, abs. error < 1e-7 up to 1e9; NaN once the reduction has no
 * correct digit left (|x| beyond ~1e15), for +-Inf and for NaN.  The path's arguments are a few turns at most. */
PM_FN void pm_sincos_d(double x, double* s, double* c) {
  const double k = rint(x * 0.6366197723675814);               /* nearest multiple of pi/2 */
  const double r = PM_FMA(-k, PM_PIO2_LO, PM_FMA(-k, PM_PIO2_HI, x));  /* |r| <= pi/4 (k * PM_PIO2_HI is exact) */
  if (!(r >= -1.0 && r <= 1.0)) { *s = *c = (double)__builtin_nanf(""); return; }  /* |x| beyond ~1e15, Inf, NaN */
  const double z = r * r;
  const double sr = PM_FMA(r * z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, -7.647163731819816e-13,
      1.6059043836821613e-10), -2.505210838544172e-08), 2.7557319223985893e-06), -0.0001984126984126984),
      0.008333333333333333), -0.16666666666666666), r);
  const double cr = PM_FMA(z * z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, 4.779477332387385e-14,
      -1.1470745597729725e-11), 2.08767569878681e-09), -2.755731922398589e-07), 2.48015873015873e-05),
      -0.001388888888888889), 0.041666666666666664),
      PM_FMAK(-0.5, z, 1.0));
  const double q = k - 4.0 * floor(k * 0.25);                  /* quadrant 0..3 (NaN for a NaN argument) */
  if (q == 1.0) { *s = cr; *c = -sr; }
  else if (q == 2.0) { *s = -sr; *c = -cr; }
  else if (q == 3.0) { *s = -cr; *c = sr; }
  else { *s = sr; *c = cr; }
}

/* natural logarithm of a positive, finite, normal double */
PM_FN double pm_log_d(double x) {
  const unsigned long long b = PM_D2U(x);
  double e = (double)((int)((b >> 52) & 0x7ffull) - 1023);
  double m = PM_U2D((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);  /* [1, 2) */
  if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1.0; }                /* [sqrt 1/2, sqrt 2] */
  const double t = (m - 1.0) / (m + 1.0);                                    /* log m = 2 atanh t, |t| <= 0.1716 */
  const double z = t * t;
  const double p = z * PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z,
      0.047619047619047616, 0.05263157894736842), 0.058823529411764705), 0.06666666666666667), 0.07692307692307693),
      0.09090909090909091), 0.1111111111111111), 0.14285714285714285), 0.2), 0.3333333333333333);
  return PM_FMA(e, PM_LN2_HI, 2.0 * t) + PM_FMA(2.0 * t, p, e * PM_LN2_LO);
}

/* e^t for -150 <= t <= 150 */
PM_FN double pm_exp_d(double t) {
  const double k = rint(t * 1.4426950408889634);
  const double r = PM_FMA(-k, PM_LN2_LO, PM_FMA(-k, PM_LN2_HI, t));  /* |r| <= ln 2 / 2 (k * PM_LN2_HI is exact) */
  const double p = PM_FMA(r * r, PM_FMAK(r, PM_FMAK(r, PM_FMAK(r, PM_FMAK(r, PM_FMAK(r, PM_FMAK(r, PM_FMAK(r, PM_FMAK(r, PM_FMAK(r, PM_FMAK(r,
      PM_FMAK(r, 1.6059043836821613e-10, 2.08767569878681e-09), 2.505210838544172e-08), 2.755731922398589e-07),
      2.7557319223985893e-06), 2.48015873015873e-05), 0.0001984126984126984), 0.001388888888888889),
      0.008333333333333333), 0.041666666666666664), 0.16666666666666666), 0.5), 1.0 + r);
  const double scale = PM_U2D((unsigned long long)((int)k + 1023) << 52);  /* 2^k, normal: |k| <= 217 */
  return p * scale;
}

/* arc tangent of a finite double t >= 0 */
PM_FN double pm_atan_pos_d(double t) {
  double base = 0.0;
  if (t > 2.414213562373095) { t = -1.0 / t; base = PM_PI_2; }                      /* atan t = pi/2 - atan(1/t) */
  else if (t > 0.41421356237309503) { t = (t - 1.0) / (t + 1.0); base = PM_PI_4; }  /* atan t = pi/4 + atan((t-1)/(t+1)) */
  const double z = t * t;                                                              /* |t| <= tan(pi/8) */
  return base + PM_FMA(t * z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z,
      PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, PM_FMAK(z, -0.03225806451612903, 0.034482758620689655), -0.037037037037037035),
      0.04), -0.043478260869565216), 0.047619047619047616), -0.05263157894736842), 0.058823529411764705),
      -0.06666666666666667), 0.07692307692307693), -0.09090909090909091), 0.1111111111111111), -0.14285714285714285),
      0.2), -0.3333333333333333), t);
}

PM_FN float pm_sin(float x) { double s, c; pm_sincos_d((double)x, &s, &c); return (float)s; }
PM_FN float pm_cos(float x) { double s, c; pm_sincos_d((double)x, &s, &c); return (float)c; }
PM_FN void pm_sincos(float x, float* s, float* c) { double sd, cd; pm_sincos_d((double)x, &sd, &cd); *s = (float)sd; *c = (float)cd; }

/* log(x): NaN for x < 0 or NaN, -Inf for +-0, +Inf for +Inf */
PM_FN float pm_log(float x) {
  if (x != x || x < 0.0f) return x != x ? x : __builtin_nanf("");
  if (x == 0.0f) return -__builtin_inff();
  if (x == __builtin_inff()) return x;
  return (float)pm_log_d((double)x);
}

/* exp(x): 0 for very negative, +Inf for very positive arguments, NaN for NaN */
PM_FN float pm_exp(float x) {
  if (x != x) return x;
  if (x > 100.0f) return __builtin_inff();
  if (x < -120.0f) return 0.0f;
  return (float)pm_exp_d((double)x);
}

/* pow(x, y) for x >= 0 (GLSL leaves x < 0 undefined; callers pass |x|): the cases of C's powf for a non-negative base */
PM_FN float pm_pow(float x, float y) {
  if (y == 2.0f) return x * x;  /* exactly what correct rounding gives, without the 1-in-10^5 */
  if (y == 0.0f || x == 1.0f) return 1.0f;
  if (x != x || y != y) return x + y;
  if (x < 0.0f) return __builtin_nanf("");
  const float inf = __builtin_inff();
  if (x == 0.0f) return y > 0.0f ? 0.0f : inf;
  if (y == inf || y == -inf) return ((x > 1.0f) == (y > 0.0f)) ? inf : 0.0f;
  if (x == inf) return y > 0.0f ? inf : 0.0f;
  double t = (double)y * pm_log_d((double)x);
  if (t > 100.0) return inf;
  if (t < -120.0) return 0.0f;
  return (float)pm_exp_d(t);
}

/* acos(x): NaN outside [-1, 1] */
PM_FN float pm_acos(float x) {
  if (x != x || x > 1.0f || x < -1.0f) return x != x ? x : __builtin_nanf("");
  if (x == -1.0f) return (float)PM_PI;
  const double d = (double)x;
  return (float)(2.0 * pm_atan_pos_d(sqrt((1.0 - d) / (1.0 + d))));
}

/* atan2(y, x) with C's conventions for zeros and infinities */
PM_FN float pm_atan2(float y, float x) {
  if (x != x || y != y) return x + y;
  const float inf = __builtin_inff();
  const int xneg = (int)(PM_F2U(x) >> 31), yneg = (int)(PM_F2U(y) >> 31);
  const float ax = xneg ? -x : x, ay = yneg ? -y : y;
  double a;                                   /* the angle of (|x|, |y|) in [0, pi/2] */
  if (ay == 0.0f) a = 0.0;
  else if (ax == 0.0f) a = PM_PI_2;
  else if (ay == inf) a = ax == inf ? PM_PI_4 : PM_PI_2;
  else if (ax == inf) a = 0.0;
  else a = pm_atan_pos_d((double)ay / (double)ax);
  if (xneg) a = PM_PI - a;
  return yneg ? -(float)a : (float)a;
}
