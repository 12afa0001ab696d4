/* Test infrastructure (part of the oracle, see rm_oracle.c): the transcendental functions of the GL implementation the
 * goldens of tests/golden/ were rendered with -- SwiftShader's shader core as shipped in the HeadlessChrome 88 of the
 * kaleido wheel (Google, Apache-2.0; not part of /root/reference and not in this repository).  GLSL leaves the precision
 * of sin / cos / log / exp / pow / asin / acos / atan to the implementation, so "what the reference computes" on a
 * transcendental scene is defined only together with its GL stack; this file restates that stack's published
 * approximations as sequences of IEEE fp32 operations:
 *   log2   exponent extraction + a (2,3) rational in the mantissa        (the sign bit is ignored: log2(-x) = log2(x),
 *          log2(+-0) = -127, log2(NaN) ~ 128.6; +Inf stays +Inf)
 *   exp2   2^i by exponent construction (i = round(x - 0.5)) times a degree-5 polynomial in x - i, x clamped to [-127, 129]
 *   log x = log2 x * ln 2,  exp x = exp2(x * log2 e),  pow(x, y) = exp2(y * log2 x)
 *   sin    x / 2pi reduced to [-0.5, 0.5] by round-to-nearest; sine-cosine pair of a quarter of the angle (degree 7 / 6),
 *          two angle doublings, the result normalised by s^2 + c^2      (no range reduction beyond fp32: 8e-5 off at |x| ~ 900)
 *   cos x = sin(x + pi/2) clamped to [-1, 1],  tan x = sin x / cos x
 *   asin   Abramowitz & Stegun 4.4.45 (4 coefficients, 7e-5),  acos x = pi/2 - asin x
 *   atan   A&S 4.4.49 on [0, 1] (1 / |x| above 1); atan(y, x) by octant reduction
 * Pinned: tests/golden/swiftshader_math.npz holds that GL stack's own outputs on ~10^5 arguments per function (random over
 * the ranges the shaders use, edge values, both signs; oracle/gl/gen_random_golden.py math), and every function here
 * reproduces every one of them bit for bit (tests/test_reference_bits.py).  Used by the oracle's OR_MATH_SWIFTSHADER mode
 * only, i.e. when the oracle is compared with the GL goldens; the HIP kernels and their checker (OR_MATH_PORTABLE) never
 * see it -- except in the library's GL-stack arithmetic (rm_ctx_set_gl_stack), a parity mode of the strict build that
 * compiles the same text (csrc/rm_ss_math.hpp, identical from the marker line on) so that the GPU reproduces the goldens
 * themselves.  The includer defines SS_FN (function qualifiers), SS_F2U / SS_U2F (bit casts). */
/* ---- shared text: identical in oracle/ss_math.h and raymarching-engine_amd/csrc/rm_ss_math.hpp from here on ---- */
SS_FN float ss_log2(float x) {
  const unsigned int xi = SS_F2U(x);
  float x1 = SS_U2F(((xi & 0x7F800000u) >> 8) | 0x3F800000u);
  x1 = (x1 - 1.4960938f) * 256.0f;
  const float x0 = SS_U2F((xi & 0x007FFFFFu) | 0x3F800000u);
  float x2 = (9.5428179e-2f * x0 + 4.7779095e-1f) * x0 + 1.9782813e-1f;
  const float x3 = ((1.6618466e-2f * x0 + 2.0350508e-1f) * x0 + 2.7382900e-1f) * x0 + 4.0496687e-2f;
  x2 /= x3;
  x1 += (x0 - 1.0f) * x2;
  return xi == 0x7F800000u ? x : x1;
}

SS_FN float ss_exp2(float x) {
  float x0 = x;
  x0 = x0 < 129.0f ? x0 : 129.0f;                          /* min / max as x86 computes them: a NaN gives the second operand */
  x0 = x0 > SS_U2F(0xC2FDFFFFu) ? x0 : SS_U2F(0xC2FDFFFFu); /* -126.99999 */
  const int i = (int)rintf(x0 - 0.5f);
  const float ii = SS_U2F((unsigned int)(i + 127) << 23);
  const float f = x0 - (float)i;
  float ff = SS_U2F(0x3AF61905u);
  ff = ff * f + SS_U2F(0x3C134806u);
  ff = ff * f + SS_U2F(0x3D64AA23u);
  ff = ff * f + SS_U2F(0x3E75EAD4u);
  ff = ff * f + SS_U2F(0x3F31727Bu);
  ff = ff * f + 1.0f;
  return ii * ff;
}

SS_FN float ss_log(float x) { return ss_log2(x) * 6.93147181e-1f; }
SS_FN float ss_exp(float x) { return ss_exp2(x * 1.44269504f); }
SS_FN float ss_pow(float x, float y) { return ss_exp2(y * ss_log2(x)); }

SS_FN float ss_sin(float x) {
  float y = x * 1.59154943e-1f;
  y = y - rintf(y);
  const float y2 = y * y;
  const float c1 = y2 * (y2 * (y2 * -0.0204391631f + 0.2536086171f) + -1.2336977925f) + 1.0f;
  const float s1 = y * (y2 * (y2 * (y2 * -0.0046075748f + 0.0796819754f) + -0.645963615f) + 1.5707963235f);
  const float c2 = (c1 * c1) - (s1 * s1);
  const float s2 = 2.0f * s1 * c1;
  return 2.0f * s2 * c2 * (1.0f / (s2 * s2 + c2 * c2));
}
/* the cosine is clamped to [-1, 1] by an x86 max then min (the sine is not: it reaches 1.0000001) -- probed: cos(8.24e-5) = 1
 * where the sine of x + pi/2 gives 1.0000001, and a NaN or infinite argument gives -1, the max's second operand */
SS_FN float ss_cos(float x) {
  float v = ss_sin(x + 1.57079632e+0f);
  v = v > -1.0f ? v : -1.0f;
  return v < 1.0f ? v : 1.0f;
}
SS_FN float ss_tan(float x) { return ss_sin(x) / ss_cos(x); }

SS_FN float ss_asin(float x) {
  const float absx = fabsf(x);
  const float p = 1.5707288f + absx * (-0.2121144f + absx * (0.0742610f + absx * -0.0187293f));
  const float r = 1.57079632f - sqrtf(1.0f - absx) * p;
  return SS_U2F(SS_F2U(r) ^ (SS_F2U(x) & 0x80000000u));
}
SS_FN float ss_acos(float x) { return 1.57079632e+0f - ss_asin(x); }

SS_FN float ss_atan_01(float x) {
  const float x2 = x * x;
  return x + x * (x2 * (-0.3333314528f + x2 * (0.1999355085f + x2 * (-0.1420889944f + x2 * (0.1065626393f + x2 * (-0.0752896400f +
         x2 * (0.0429096138f + x2 * (-0.0161657367f + x2 * 0.0028662257f))))))));
}
SS_FN float ss_atan(float x) {
  const float absx = fabsf(x);
  const int o = !(absx < 1.0f);
  const float t = ss_atan_01(o ? 1.0f / absx : absx);
  const float r = o ? 1.57079632f - t : t;
  return SS_U2F(SS_F2U(r) ^ (SS_F2U(x) & 0x80000000u));
}
SS_FN float ss_atan2(float y, float x) {
  const float pi = 3.14159265f, half_pi = 1.57079632f, quarter_pi = 7.85398163e-1f;
  const int s = y < 0.0f;                                     /* lower half plane: rotate to the upper one */
  float theta = s ? -pi : 0.0f;
  const float x0 = SS_U2F((SS_F2U(y) & 0x80000000u) ^ SS_F2U(x));
  const float y0 = fabsf(y);
  const int q = x0 < 0.0f;                                    /* left quadrant: rotate to the right one */
  theta += q ? half_pi : 0.0f;
  const float x1 = q ? y0 : x0, y1 = q ? -x0 : y0;
  const int o = !(y1 < x1);                                   /* second octant: mirror to the first */
  const float x2 = o ? y1 : x1, y2 = o ? x1 : y1;
  const int zero_x = x2 == 0.0f, inf_y = isinf(y2);
  const float t = ss_atan_01(y2 / x2);
  if (inf_y) theta += quarter_pi;
  else if (!zero_x) theta += o ? half_pi - t : t;
  return (s && q && o && !inf_y) ? -t : theta;               /* -pi + pi/2 + pi/2 - t without the cancellation */
}
