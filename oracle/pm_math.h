/* Portable transcendentals of the PARITY arithmetic ("PM"): sin, cos, log, exp, pow, acos, atan2 of a float as fixed SEQUENCES OF
 * IEEE-754 BINARY32 OPERATIONS -- +, -, *, /, sqrt, fusedMultiplyAdd, floor, rint, comparisons and bit casts, each of which the
 * standard defines bit for bit -- in the order written here.  The same text is compiled into the CPU oracle (oracle/pm_math.h) and
 * into the HIP kernels (raymarching-engine_amd/csrc/rm_pm_math.hpp; tests/test_host_cpu.py checks that the two files agree), both
 * with -ffp-contract=off, so the strict build and the oracle agree on every transcendental bit for bit, on any libm / ocml.
 * (GLSL leaves the precision of these functions to the implementation -- the reference's own results differ between drivers -- so
 * there is no reference bit pattern to match; the goldens pin the oracle to the reference in its GL stack's OWN arithmetic,
 * oracle/ss_math.h, and to this one within the tolerances of tests/test_oracle_golden.py.)
 *
 * Round 5: fp32 throughout.  Rounds 2-4 evaluated double-precision series and rounded once -- correctly rounded results, at 1 500
 * fp64 instructions in the shading of every kernel and a strict Mandelbulb 26x slower than the fast one.  The reference's shader is
 * `precision highp float` on every GPU it runs on: what a GL stack computes for sin() is an fp32 sequence of this kind.  Accuracy
 * (tests/test_host_cpu.py measures it against double precision): log, exp, sin, cos (|x| <= 1e5), acos within 2 ulp, atan2 within 3, pow
 * within 2 ulp for exponents up to 16 (3 at 32, 5 at 64, 16 at 250) -- its logarithm is carried as a normalised pair (hi, lo).  Denormal-free by definition: an argument or a
 * result below 2^-126 counts as zero, so the sequences give the same bits whether or not fp32 denormals are flushed (the fast
 * kernels of the power-8 Mandelbulb flush them).  Polynomial coefficients: the classic single-precision minimax sets of
 * Cephes (sinf, cosf, expf, asinf, atanf; Moshier) and FreeBSD msun (e_logf.c).
 *
 * The including file defines PM_FN (function qualifiers), PM_FMAF (IEEE fusedMultiplyAdd of floats), PM_SQRTF (IEEE squareRoot), PM_DIV_ORDINARY
 * (IEEE division, used where the operands and the quotient are known to be far from the ends of the exponent range), PM_F2U / PM_U2F (bit casts
 * float <-> 32-bit unsigned). */

#define PM_INF __builtin_inff()
#define PM_NAN __builtin_nanf("")
#define PM_TINY 1.17549435e-38f /* 2^-126, the smallest normal float */

/* ---- sin, cos ------------------------------------------------------------------------------------------------------------
 * k = the nearest multiple of pi/2, r = x - k pi/2 by three fused steps (pi/2 = P1 + P2 + P3 to 2^-76), polynomials on |r| <= pi/4
 * (the sine's is Taylor's, the cosine's Cephes').  For |x| <= pi/4 -- k = 0, r = x: the angles that are scene parameters, whose
 * sine and cosine are folded into a fractal fourteen levels deep -- the results are the correctly rounded ones but for 1-2 % of
 * the arguments (0.55 / 0.66 ulp): a rotation by (c, s) with c^2 + s^2 one ulp above 1 instead of below makes the distance
 * estimate of examples/rotation-fractal.glsl overshoot from afar, and its sky renders differently (tests/test_oracle_golden.py).  Accurate while k is exact (|x| up to ~1e5; the path's arguments are a few turns); beyond that r drifts out of the
 * interval and the values degrade; NaN once it has left it altogether (|x| beyond ~1e7), for +-Inf and for NaN. */
PM_FN void pm_sincos(float x, float* s, float* c) {
  const float k = rintf(x * 0.636619747f);
  float r = PM_FMAF(-k, 1.57079637f, x);
  r = PM_FMAF(-k, -4.37113883e-8f, r);
  r = PM_FMAF(-k, -1.71512421e-15f, r);
  const float z = r * r;
  const float sp = PM_FMAF(PM_FMAF(PM_FMAF(2.7557319e-6f, z, -1.9841270e-4f), z, 8.3333333e-3f), z, -1.6666667e-1f);  /* Taylor: r^11 / 11! < 0.05 ulp */
  float sr = PM_FMAF(r * z, sp, r);
  const float cp = PM_FMAF(PM_FMAF(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  const float hz = 0.5f * z, w = 1.0f - hz;   /* cos r = 1 - z/2 + z^2 cp with the head's rounding error put back: (1 - w) - hz is exact, */
  float cr = w + (((1.0f - w) - hz) + PM_FMAF(z * z, cp, -0.5f * PM_FMAF(r, r, -z)));  /* and so is r^2 - z */
  if (!((r < 0.0f ? -r : r) <= 1.25f)) sr = cr = PM_NAN; /* |x| beyond ~1e7 (k is no longer the nearest multiple), Inf, NaN */
  const float q = k - 4.0f * floorf(k * 0.25f); /* quadrant 0..3 (NaN for a non-finite argument: the last branch, of NaNs) */
  if (q == 1.0f) { *s = cr; *c = -sr; }
  else if (q == 2.0f) { *s = -sr; *c = -cr; }
  else if (q == 3.0f) { *s = -cr; *c = sr; }
  else { *s = sr; *c = cr; }
}
PM_FN float pm_sin(float x) { float s, c; pm_sincos(x, &s, &c); return s; }
PM_FN float pm_cos(float x) { float s, c; pm_sincos(x, &s, &c); return c; }

/* ---- log -----------------------------------------------------------------------------------------------------------------
 * log x = hi + lo for a positive, finite, normal x: x = 2^k m, m in [sqrt 1/2, sqrt 2), f = m - 1, s = f / (2 + f),
 * log m = f - f^2/2 + s (f^2/2 + R(s^2)) (msun e_logf.c), kept as the rounded head f - f^2/2 and everything it leaves behind;
 * k ln 2 = k LN2_HI (exact: 17 bits x 8 bits) + k LN2_LO.  hi + lo carries about 27 bits, |lo| <= ulp(hi) / 2. */
PM_FN void pm_log_hl(float x, float* hi, float* lo) {
  unsigned int ix = PM_F2U(x) + (0x3f800000u - 0x3f3504f3u);
  const float k = (float)((int)(ix >> 23) - 127);
  ix = (ix & 0x007fffffu) + 0x3f3504f3u;
  const float f = PM_U2F(ix) - 1.0f;
  const float s = PM_DIV_ORDINARY(f, 2.0f + f);  /* |f| < 0.42 or 0, the divisor in [1.7, 2.42] */
  const float z = s * s, w = z * z;
  const float t1 = w * PM_FMAF(w, 0.24279078841f, 0.40000972152f);
  const float t2 = z * PM_FMAF(w, 0.28498786688f, 0.66666662693f);
  const float R = t2 + t1;
  const float hf = 0.5f * f, hfsq = hf * f;
  const float herr = PM_FMAF(hf, f, -hfsq);       /* f^2/2 = hfsq + herr exactly */
  const float mh = f - hfsq;                      /* |f| > |hfsq|: the two lines below recover its rounding error exactly */
  const float ml = ((f - mh) - hfsq) - herr + s * (hfsq + R);
  const float ah = k * 6.9313812256e-01f;         /* exact */
  const float h = ah + mh;                        /* |ah| > |mh| unless k = 0, and then h = mh exactly */
  const float l = ((ah - h) + mh) + PM_FMAF(k, 9.0580006145e-06f, ml);
  const float sum = h + l;                        /* the pair normalised (|h| > |l|: exact), so that |lo| <= ulp(hi) / 2 and a */
  *lo = l - (sum - h);                            /* large exponent multiplies a small tail: pow(1.4, 60) was 1 400 ulp off without */
  *hi = sum;
}
/* log(x): NaN for x < 0 or NaN, -Inf for zero (and for a denormal: see the header), +Inf for +Inf */
PM_FN float pm_log(float x) {
  float hi, lo;
  pm_log_hl(x, &hi, &lo);
  float r = hi + lo;
  if (x < PM_TINY) r = -PM_INF;
  if (x == PM_INF) r = PM_INF;
  if (x != x || x < 0.0f) r = PM_NAN;
  return r;
}

/* ---- exp -----------------------------------------------------------------------------------------------------------------
 * e^(t + tl), |tl| << 1: n = rint(t / ln 2), r = t - n ln 2 in two fused steps (n * 0.693359375 is exact), Cephes' polynomial on
 * |r| <= ln 2 / 2, e^tl = 1 + tl, the scale 2^n in two factors (n = 128 has no float of its own).  +Inf above the overflow
 * threshold, 0 where the result would be below 2^-126, NaN for NaN. */
PM_FN float pm_exp_core(float tc, float tl) { /* -104 <= tc <= 89: the evaluation itself, no special case */
  const float n = rintf(tc * 1.44269504f);
  float r = PM_FMAF(-n, 0.693359375f, tc);
  r = PM_FMAF(-n, -2.12194440e-4f, r) + tl;
  const float z = r * r;
  float p = PM_FMAF(PM_FMAF(PM_FMAF(PM_FMAF(PM_FMAF(1.9875691500e-4f, r, 1.3981999507e-3f), r, 8.3334519073e-3f), r, 4.1665795894e-2f), r,
                            1.6666665459e-1f), r, 5.0000001201e-1f);
  p = PM_FMAF(p, z, r) + 1.0f;
  const int ni = (int)n, n1 = ni >> 1, n2 = ni - n1;  /* (an arithmetic shift on every compiler this text meets; any split of ni gives the same bits) */
  return (p * PM_U2F((unsigned int)(n1 + 127) << 23)) * PM_U2F((unsigned int)(n2 + 127) << 23);
}
PM_FN float pm_exp_hl(float t, float tl) {
  const float tc = t > 89.0f ? 89.0f : (t < -104.0f ? -104.0f : (t == t ? t : 0.0f)); /* (keeps n in range; a NaN's result is replaced below) */
  float v = pm_exp_core(tc, tl);
  if (v < PM_TINY) v = 0.0f;
  if (t > 88.7228394f) v = PM_INF;
  if (t != t) v = t;
  return v;
}
PM_FN float pm_exp(float x) { return pm_exp_hl(x, 0.0f); }

/* ---- pow -----------------------------------------------------------------------------------------------------------------
 * pow(x, y) for x >= 0 (GLSL leaves x < 0 undefined; callers pass |x|) from the logarithm of x as a pair: the cases of C's powf
 * for a non-negative base, then exp(y (hi + lo)) with the product carried as a pair too.  pm_pow_from_log lets two powers of one
 * base share its logarithm (the Mandelbulb's r^(n-1) and r^n). */
PM_FN float pm_pow_from_log(float x, float y, float hi, float lo) {
  if (y == 2.0f) return x * x;  /* a product in every implementation met so far */
  if (y == 0.0f || x == 1.0f) return 1.0f;
  if (x != x || y != y) return x + y;
  if (x < 0.0f) return PM_NAN;
  if (x < PM_TINY) return y > 0.0f ? 0.0f : PM_INF;
  if (y == PM_INF || y == -PM_INF) return ((x > 1.0f) == (y > 0.0f)) ? PM_INF : 0.0f;
  if (x == PM_INF) return y > 0.0f ? PM_INF : 0.0f;
  const float th = y * hi;
  const float tl = PM_FMAF(y, hi, -th) + y * lo;
  if (!(th < 200.0f)) return PM_INF;  /* (also an overflowed product) */
  if (th < -200.0f) return 0.0f;
  return pm_exp_hl(th, tl);
}
PM_FN float pm_pow(float x, float y) {
  float hi, lo;
  pm_log_hl(x, &hi, &lo);
  return pm_pow_from_log(x, y, hi, lo);
}

/* ---- acos ----------------------------------------------------------------------------------------------------------------
 * Cephes asinf: asin t = t + t^3 P(t^2) on |t| <= 1/2; acos x = pi/2 - asin x there, 2 asin sqrt((1 - x) / 2) above,
 * pi - 2 asin sqrt((1 + x) / 2) below.  NaN outside [-1, 1]. */
PM_FN float pm_asin_poly(float t, float z) {
  const float p = PM_FMAF(PM_FMAF(PM_FMAF(PM_FMAF(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z,
                          1.6666752422e-1f);
  return PM_FMAF(t * z, p, t);
}
PM_FN float pm_acos(float x) {
  const float ax = x < 0.0f ? -x : x;
  const int inner = ax <= 0.5f;
  float t = x, z = x * x;                       /* the polynomial's arguments: (x, x^2) inside, (sqrt z, z = (1 - |x|) / 2) outside */
  if (!inner) { z = 0.5f * (1.0f - ax); t = PM_SQRTF(z); }
  const float a = pm_asin_poly(t, z);
  float r = inner ? 1.57079637f - (a + 4.37113883e-8f) : (x < 0.0f ? 3.14159274f - (2.0f * a + 8.74227766e-8f) : 2.0f * a);
  if (!(ax <= 1.0f)) r = x != x ? x : PM_NAN;
  return r;
}

/* ---- atan2 ---------------------------------------------------------------------------------------------------------------
 * atan2(y, x) with C's conventions for zeros and infinities.  The angle of (|x|, |y|) by Cephes' atanf reduction with ONE
 * division: beyond tan 3pi/8 it is pi/2 + atan(-|x| / |y|), beyond tan pi/8 it is pi/4 + atan((|y| - |x|) / (|y| + |x|)), and the
 * polynomial serves |t| <= tan pi/8.  (|x| + |y| has to stay finite: magnitudes up to 1e38.) */
PM_FN float pm_atan_quadrant(float ax, float ay) { /* the angle of (ax, ay), both positive and finite, in (0, pi/2) */
  float num = ay, den = ax, base = 0.0f;
  if (ay > 2.41421356f * ax) { num = -ax; den = ay; base = 1.57079637f; }
  else if (ay > 0.414213562f * ax) { num = ay - ax; den = ay + ax; base = 0.785398185f; }
  const float t = num / den, z = t * t;
  const float p = PM_FMAF(PM_FMAF(PM_FMAF(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
  return base + PM_FMAF(t * z, p, t);
}
PM_FN float pm_atan2(float y, float x) {
  if (x != x || y != y) return x + y;
  const int xneg = (int)(PM_F2U(x) >> 31), yneg = (int)(PM_F2U(y) >> 31);
  const float ax = xneg ? -x : x, ay = yneg ? -y : y;
  float a;                                    /* the angle of (|x|, |y|) in [0, pi/2] */
  if (ay == 0.0f) a = 0.0f;
  else if (ax == 0.0f) a = 1.57079637f;
  else if (ay == PM_INF) a = ax == PM_INF ? 0.785398185f : 1.57079637f;
  else if (ax == PM_INF) a = 0.0f;
  else a = pm_atan_quadrant(ax, ay);
  if (xneg) a = 3.14159274f - a;
  return yneg ? -a : a;
}
