// rm_device.hpp -- device-side building blocks of the gfx950 sphere tracer.
//
// Both kernel translation units are compiled with -ffp-contract=off, so an
// FMA exists only where the code asks for one.  Arithmetic is written against
// a math policy:
//
//   PM  "precise": exactly the operations the oracle performs -- IEEE
//       add/mul/div/sqrt, no FMA but the explicit ones of rm_pm_math.hpp, whose
//       fp32 sequences are pow/log/exp/sin/cos/acos/atan (the oracle compiles the
//       same text).  The parity build uses it everywhere.
//   FM  "fast": v_fma_f32, v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 / v_log_f32 at
//       hardware rate, trig-free power-8 Mandelbulb.  The fast build uses it for
//       every DISTANCE EVALUATION: castRay's (>= 98 % of the work) and, since
//       round 3, the four of sceneNormal and the subsurface test (rm_kernels.inc
//       RM_NORMAL_POLICY has the measurement); the RNG, the camera, the materials
//       and all shading arithmetic stay on PM, so a pixel whose rays never come
//       near the surface -- the sky -- has the parity build's bits.
//
// file:line = client/public/shader/raymarcher.frag of the reference unless stated.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/hip_raymarch.h"
#include "rm_params.hpp"

namespace rm {
#ifdef RM_LANE_STATS
static __device__ unsigned long long g_lane_stats[4];
#endif


#define RM_DEV __device__ __forceinline__

struct v3 {
  float x, y, z;
};

RM_DEV v3 V(float x, float y, float z) { return v3{x, y, z}; }
RM_DEV v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
RM_DEV v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
RM_DEV v3 operator*(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
RM_DEV v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
RM_DEV v3 adds(v3 a, float s) { return V(a.x + s, a.y + s, a.z + s); }
RM_DEV v3 vabs(v3 a) { return V(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }

#include "rm_pm_math.hpp"

// RM_GL_STACK (rm_glstack.hip only): the parity build in the arithmetic of the GL stack the reference's golden images were
// rendered under -- its twelve transcendental functions (rm_ss_math.hpp) and its conventions: min / max as x86 computes
// them, fract clamped below 1 -- and WITHOUT the exact shortcuts of this file and rm_kernels.inc that were derived for
// IEEE semantics (log 0 = -Inf, min / max dropping a NaN).  With it the kernels reproduce tests/golden/ bit for bit.
#ifndef RM_GL_STACK
#define RM_GL_STACK 0
#endif
#if RM_GL_STACK
#define SS_FN RM_DEV
#define SS_F2U(x) __float_as_uint(x)
#define SS_U2F(u) __uint_as_float(u)
#include "rm_ss_math.hpp"
#endif

// ---- math policies --------------------------------------------------------------

#if RM_GL_STACK
struct PM {
  static constexpr bool fast = false;
  static RM_DEV float fma(float a, float b, float c) { return a * b + c; }
  static RM_DEV float rcp(float x) { return 1.0f / x; }
  static RM_DEV float div(float a, float b) { return a / b; }
  static RM_DEV float sqrt(float x) { return sqrtf(x); }
  static RM_DEV float sqrt_of_ordinary(float x) { return sqrtf(x); }
  static RM_DEV float pow(float x, float y) { return ss_pow(x, y); }  // exp2(y log2 |x|), also for y = 2
  static RM_DEV float log(float x) { return ss_log(x); }
  static RM_DEV float exp(float x) { return ss_exp(x); }
  static RM_DEV float sin(float x) { return ss_sin(x); }
  static RM_DEV float cos(float x) { return ss_cos(x); }
  static RM_DEV float acos(float x) { return ss_acos(x); }
  static RM_DEV float atan2(float y, float x) { return ss_atan2(y, x); }
  static RM_DEV void sincos(float x, float& s, float& c) { s = ss_sin(x); c = ss_cos(x); }
  static RM_DEV void pow_pair(float r, float n, float& r_nm1, float& r_n) { r_nm1 = pow(r, n - 1.0f); r_n = pow(r, n); }
};
#else
#ifdef RM_COLD_OUTLINE  // (measurement builds: the shared text's special cases as functions of their own)
__device__ __attribute__((noinline)) static float pm_pow_from_log_cold(float x, float y, float hi, float lo) { return pm_pow_from_log(x, y, hi, lo); }
__device__ __attribute__((noinline)) static float pm_atan2_cold(float y, float x) { return pm_atan2(y, x); }
#else
#define pm_pow_from_log_cold pm_pow_from_log
#define pm_atan2_cold pm_atan2
#endif
struct PM {
  static constexpr bool fast = false;
  static RM_DEV float fma(float a, float b, float c) { return a * b + c; }  // two roundings, like GLSL/C
  static RM_DEV float rcp(float x) { return 1.0f / x; }
  static RM_DEV float div(float a, float b) { return a / b; }
  static RM_DEV float sqrt(float x) { return sqrtf(x); }
  // the same value by rm_pm_math.hpp's shorter way for an argument of ordinary size: for the Mandelbulb's rounds, whose other
  // functions are branches already (in a table's row loop the test costs more than it saves: strict C5 stripes 43.7 -> 45.9 ms)
  static RM_DEV float sqrt_of_ordinary(float x) { return rm_sqrt_rn(x); }
  // Transcendentals: rm_pm_math.hpp, the same text as the oracle's.  pow: on |x| (oracle/rm_oracle.c gl_pow; GLSL leaves
  // a negative base undefined, SwiftShader takes |x|); pm_pow returns x * x for the exponent 2 -- every use of the path
  // with that exponent has it as a literal (the GGX term :371, schlick's r0 :173), so the test folds away.
  static RM_DEV float pow(float x, float y) {
    float hi, lo;
    pm_log_hl(fabsf(x), &hi, &lo);
    return pow_from_log(fabsf(x), y, hi, lo);
  }
  static RM_DEV float log(float x) { return pm_log(x); }
  static RM_DEV float exp(float x) { return pm_exp(x); }
  static RM_DEV float sin(float x) { return pm_sin(x); }
  static RM_DEV float cos(float x) { return pm_cos(x); }
  static RM_DEV float acos(float x) { return pm_acos(x); }
  static RM_DEV void sincos(float x, float& s, float& c) { pm_sincos(x, &s, &c); }  // one reduction, the bits of sin and cos
  // pow(r, n - 1) and pow(r, n) from ONE logarithm of r: the bits of the two calls (pm_pow is pm_log_hl + pm_pow_from_log)
  static RM_DEV void pow_pair(float r, float n, float& r_nm1, float& r_n) {
    float hi, lo;
    pm_log_hl(fabsf(r), &hi, &lo);
    r_nm1 = pow_from_log(fabsf(r), n - 1.0f, hi, lo);
    r_n = pow_from_log(fabsf(r), n, hi, lo);
  }
  // The special cases of pm_pow_from_log / pm_exp_hl / pm_atan2 are a dozen comparisons each, every one a VALU instruction, on
  // arguments that are ordinary numbers nearly always.  One test for "ordinary" first (v_cmp_class: a positive normal base, a
  // finite non-zero exponent, neither 1 / 2, |y log x| < 87 so that the result is a normal number) and then the evaluation
  // alone -- pm_exp_core, what the shared text reaches for such arguments through all of its tests; the text itself for the
  // wave that holds anything else.  Same bits: rm_probe_math asks both ways on millions of arguments (tests/test_gpu_parity.py).
  static RM_DEV float pow_from_log(float x, float y, float hi, float lo) {
    const float th = y * hi, tl = PM_FMAF(y, hi, -th) + y * lo;  // (the product as pm_pow_from_log carries it)
    const bool plain = __builtin_amdgcn_classf(x, 0x100) & __builtin_amdgcn_classf(y, 0x198) & (x != 1.0f) & (y != 2.0f) & (fabsf(th) < 87.0f);
    if (__builtin_expect(plain, 1)) return pm_exp_core(th, tl);
    return pm_pow_from_log_cold(x, y, hi, lo);
  }
  static RM_DEV float atan2(float y, float x) {
    if (__builtin_expect(__builtin_amdgcn_classf(x, 0x198) & __builtin_amdgcn_classf(y, 0x198), 1)) {  // both finite and not zero
      const float a = pm_atan_quadrant(fabsf(x), fabsf(y));
      const float b = x < 0.0f ? 3.14159274f - a : a;
      return y < 0.0f ? -b : b;
    }
    return pm_atan2_cold(y, x);
  }
};
#endif

struct FM {
  static constexpr bool fast = true;
  static RM_DEV float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
  static RM_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static RM_DEV float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
  static RM_DEV float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  static RM_DEV float sqrt_of_ordinary(float x) { return __builtin_amdgcn_sqrtf(x); }
  static RM_DEV float pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(fabsf(x))); }
  static RM_DEV float log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718056f; }
  static RM_DEV float sin(float x) { return __builtin_amdgcn_sinf(x * 0.15915494309189535f); }
  static RM_DEV float cos(float x) { return __builtin_amdgcn_cosf(x * 0.15915494309189535f); }
  static RM_DEV float acos(float x) { return acosf(x); }
  static RM_DEV float atan2(float y, float x) { return atan2f(y, x); }
  static RM_DEV void sincos(float x, float& s, float& c) { s = sin(x); c = cos(x); }
  static RM_DEV void pow_pair(float r, float n, float& r_nm1, float& r_n) { r_nm1 = pow(r, n - 1.0f); r_n = pow(r, n); }
  // VOP3 output modifiers: the result doubled (mul:2) or quadrupled (mul:4) in the same instruction.  The hardware
  // ignores them while MODE.IEEE is set or fp32 denormals are kept -- the state a HIP kernel starts in -- so every
  // kernel that evaluates the power-8 Mandelbulb on this policy clears both first (omod_mode; enter_math_mode below).
  // tools/ubench/omod_probe.hip: the modifier takes effect only with BOTH cleared, as the ISA manual says.  With the
  // IEEE bit off v_min/v_max no longer quiet a signalling NaN (arithmetic produces none); fp32 denormals read and
  // write as zero in those kernels.
  static RM_DEV void omod_mode() {
    __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);  // hwreg(HW_REG_MODE, 4, 2): fp32 denormals flushed
    __builtin_amdgcn_s_setreg(1 | (9 << 6) | (0 << 11), 0);  // hwreg(HW_REG_MODE, 9, 1): IEEE off
  }
  static RM_DEV float mul_x2(float a, float b) { float r; asm("v_mul_f32_e64 %0, %1, %2 mul:2" : "=v"(r) : "v"(a), "v"(b)); return r; }
  static RM_DEV float mul_x4(float a, float b) { float r; asm("v_mul_f32_e64 %0, %1, %2 mul:4" : "=v"(r) : "v"(a), "v"(b)); return r; }
  static RM_DEV float fma_x2(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3 mul:2" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
  static RM_DEV float sq_minus_half_x2(float a) { float r; asm("v_fma_f32 %0, %1, %1, -0.5 mul:2" : "=v"(r) : "v"(a)); return r; }  // 2 a^2 - 1
};

// ---- vector helpers / GLSL built-ins (semantics pinned in oracle/rm_oracle.c)

template <class M> RM_DEV float dot(v3 a, v3 b) { return M::fma(a.z, b.z, M::fma(a.y, b.y, a.x * b.x)); }
template <class M> RM_DEV float length(v3 a) { return M::sqrt(dot<M>(a, a)); }
template <class M> RM_DEV float distance(v3 a, v3 b) { return length<M>(a - b); }
// v * (1/length(v)): the form pinned against the reference GLSL
template <class M> RM_DEV v3 normalize(v3 a) { return a * M::rcp(length<M>(a)); }
template <class M> RM_DEV v3 madd(v3 a, float s, v3 b) { return V(M::fma(a.x, s, b.x), M::fma(a.y, s, b.y), M::fma(a.z, s, b.z)); }  // a*s + b
RM_DEV v3 cross(v3 a, v3 b) { return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
template <class M> RM_DEV v3 reflect(v3 i, v3 n) { return i - n * (2.0f * dot<M>(n, i)); }

// min/max are IEEE minNum/maxNum = v_min_f32/v_max_f32 (NaN convention OR_NAN_IEEE)
#if RM_GL_STACK  // x > y ? x : y and x < y ? x : y: the second operand when either is a NaN (OR_NAN_X86)
RM_DEV float gmax(float x, float y) { return x > y ? x : y; }
RM_DEV float gmin(float x, float y) { return x < y ? x : y; }
RM_DEV float gfract(float x) { const float r = x - floorf(x); return r < 0.99999994f ? r : 0.99999994f; }
#else
RM_DEV float gmax(float x, float y) { return fmaxf(x, y); }
RM_DEV float gmin(float x, float y) { return fminf(x, y); }
RM_DEV float gfract(float x) { return x - floorf(x); }
#endif
RM_DEV float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
// UNORM8 conversion of the canvas (display.frag's output): NaN and negatives to 0, round to nearest; the GL stack of the
// goldens goes through 16 bits (oracle/rm_oracle.c or_present)
RM_DEV unsigned char unorm8(float g) {
  g = g != g ? 0.0f : (g < 0.0f ? 0.0f : (g > 1.0f ? 1.0f : g));
#if RM_GL_STACK
  const int c16 = (int)(g * 65535.0f);
  return (unsigned char)((c16 - (c16 >> 8) + 128) >> 8);
#else
  return (unsigned char)floorf(g * 255.0f + 0.5f);
#endif
}
template <class M> RM_DEV float gmod(float x, float y) { return x - y * floorf(M::div(x, y)); }
RM_DEV float gsign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
template <class M> RM_DEV float gmix(float x, float y, float a) { return M::fma(a, y - x, x); }  // x + a*(y-x)
RM_DEV v3 vmaxs(v3 a, float s) { return V(gmax(a.x, s), gmax(a.y, s), gmax(a.z, s)); }
template <class M> RM_DEV v3 vmods(v3 a, float s) { return V(gmod<M>(a.x, s), gmod<M>(a.y, s), gmod<M>(a.z, s)); }
RM_DEV bool is_inf(float x) { return fabsf(x) == __builtin_inff(); }
RM_DEV bool is_nan(float x) { return x != x; }
RM_DEV bool finite(float x) { return fabsf(x) < __builtin_inff(); }
RM_DEV bool any_nonfinite(v3 a) { return !finite(a.x) || !finite(a.y) || !finite(a.z); }
// the wave's mask of a condition, straight from the compare (HIP's __ballot goes through an integer 0/1 and a second compare)
RM_DEV unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
RM_DEV bool same_bits(v3 a, v3 b) {
  return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
         __float_as_uint(a.z) == __float_as_uint(b.z);
}

// mat4 * vec4(v, 0).xyz, column-major (:190,:195,:196,:199)
RM_DEV v3 mat_rotate(const float* m, v3 v) {
  return V(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
           m[2] * v.x + m[6] * v.y + m[10] * v.z);
}

// ---- the portable tangent (oracle/rm_oracle.c or_tan, oracle/gl/glref.py):
// one fixed sequence of IEEE operations (no FMA: the TU is contract-off; plain
// '/' and sqrtf are correctly rounded), the same bits in every build.
#if RM_GL_STACK
// rm_ctx_set_gl_stack(ctx, 2): the GL stack's OWN tan (sin / cos of rm_ss_math.hpp) instead of the portable tangent -- what the
// reference's unmodified text computes under that stack (random stream and camera); process-wide, set by rm_gl_set_native_tan
static __device__ int g_native_tan = 0;
#endif
RM_DEV float rm_tan(float x) {
#if RM_GL_STACK
  if (g_native_tan) return ss_tan(x);
#endif
  float k = floorf(x * 0.636619772f + 0.5f);
  float r = x - k * 1.5703125f;
  r = r - k * 4.83751296997e-4f;
  r = r - k * 7.54978995489e-8f;
  float r2 = r * r;
  float s = r2 * -1.9515295891e-4f + 8.3321608736e-3f;
  s = s * r2 + -1.6666654611e-1f;
  s = s * r2 * r + r;
  float c = r2 * 2.443315711809948e-5f + -1.388731625493765e-3f;
  c = c * r2 + 4.166664568298827e-2f;
  c = c * r2 * r2 + (1.0f - 0.5f * r2);
  float odd = k - 2.0f * floorf(k * 0.5f);
  // (odd > 0.5f) ? (-c / s) : (s / c) with ONE division: picking the operands first gives the same bits
  // (an IEEE division is ~12 instructions, and every random number is one tangent)
  const bool flip = odd > 0.5f;
  return (flip ? -c : s) / (flip ? s : c);
}

// ---- RNG: :44-49, :78-105.  distance(xy*PHI, xy) depends on the pixel only,
// so it is computed once; every sample is then the same operations as the GLSL.
struct Rng {
  float seed;    // :78
  float x1000;   // texcoord.x * 1000
  float dist;    // distance(xy * PHI, xy)
  float n0, n1;  // randNoise
};

RM_DEV Rng rng_init(float tcx, float tcy, float n0, float n1) {
  const float PHI = 1.61803398874989484820459f;
  Rng r;
  r.seed = 0.0f;
  float x = tcx * 1000.0f, y = tcy * 1000.0f;
  float dx = x * PHI - x, dy = y * PHI - y;
  r.x1000 = x;
  r.dist = sqrtf(dx * dx + dy * dy);
  r.n0 = n0;
  r.n1 = n1;
  return r;
}

// :46-49
RM_DEV float gold_noise(const Rng& r, float seed) {
  return gfract(rm_tan(r.dist * seed) * r.x1000);
}

// :91-94
RM_DEV float uniform_sample(Rng& r) {
  r.seed += 0.131223f;
  return gold_noise(r, gfract(r.n0 + r.seed));
}

// :80-89 (PI is 3.141592 there)
RM_DEV void box_muller(Rng& r, float& ox, float& oy) {
  r.seed += 0.123123213f;
  const float u1 = gold_noise(r, gfract(r.n0 + r.seed));
  r.seed += 0.123123213f;
  const float u2 = gold_noise(r, gfract(r.n1 + r.seed));
  const float two_pi_u2 = 2.0f * 3.141592f * u2;
#if RM_BUILD_FAST && defined(RM_SHADE_FAST)  // experiment builds (tools/): the hardware's transcendentals in the random directions
  const float rad = __builtin_amdgcn_sqrtf(-2.0f * (__builtin_amdgcn_logf(u1) * 0.69314718056f));
  float sn = __builtin_amdgcn_sinf(two_pi_u2 * 0.15915494309189535f), cs = __builtin_amdgcn_cosf(two_pi_u2 * 0.15915494309189535f);
#else
  const float rad = sqrtf(-2.0f * PM::log(u1));
  float sn, cs;
  PM::sincos(two_pi_u2, sn, cs);  // one range reduction for both
#endif
  ox = rad * cs;
  oy = rad * sn;
}

// :96-101
RM_DEV v3 sphere_sample(Rng& r) {
  float ax, ay, bx, by;
  box_muller(r, ax, ay);
  box_muller(r, bx, by);
  return normalize<PM>(V(ax, ay, bx));
}

// What the stream looks like after uniformSample() / sphereSample() without their values (a value that no
// live computation reads need not be computed; the seed has to advance by exactly the same additions).
RM_DEV void rng_skip_uniform(Rng& r) { r.seed += 0.131223f; }
RM_DEV void rng_skip_sphere(Rng& r) {
  r.seed += 0.123123213f; r.seed += 0.123123213f; r.seed += 0.123123213f; r.seed += 0.123123213f;
}

// sphereSample() * scale, exactly, for a `scale` that is usually 0 (dof.amount = 0: :186-193; a point light's size:
// :359).  With scale == 0 the product is a vector of zeros unless the sample is non-finite, and then it is NaN, which
// reaches the image (the pixel's ray origin is NaN).  The sample is non-finite exactly when the first uniform of one
// of its two Box-Muller pairs is 0 (log 0 = -Inf; a gold_noise value is 0 whenever |tan| x 1000 x texcoord.x >= 2^23,
// about 1e-5 of the draws), so two of the four random numbers decide, and the logarithms, square roots, sines,
// cosines and the normalisation are only computed for the lanes that need them.  The signs of the zeros are not
// reproduced (+0 is returned): the callers add the product to, or subtract it from, a value, where the sign of a zero
// addend shows only if that value is itself -0 -- `zero_sign_matters` says so, and then the full sample is taken.
RM_DEV v3 sphere_sample_times(Rng& r, float scale, bool zero_sign_matters) {
  if (RM_GL_STACK || scale != 0.0f || scale != scale) return sphere_sample(r) * scale;  // (the shortcut below is IEEE's: log 0 = -Inf)
  const float seed0 = r.seed;
  const float s1 = seed0 + 0.123123213f;   // first pair: log argument
  const float s2 = s1 + 0.123123213f;      //             angle (value not needed)
  const float s3 = s2 + 0.123123213f;      // second pair: log argument
  const float s4 = s3 + 0.123123213f;
  const float a = r.n0 + s1, b = r.n0 + s3;
  const float u1a = gold_noise(r, gfract(a)), u1b = gold_noise(r, gfract(b));
  r.seed = s4;
  // (a value of exactly 1 -- fract() of a tiny negative number -- makes a radius 0, two of them a 0/0: also the slow way)
  const bool ordinary = u1a > 0.0f && u1a < 1.0f && u1b > 0.0f && u1b < 1.0f;
  if (ordinary && !zero_sign_matters) return V(0.0f, 0.0f, 0.0f);
  r.seed = seed0;  // rare: a logarithm is -Inf or 0 (or the caller needs the zeros' signs)
  return sphere_sample(r) * scale;
}
RM_DEV bool is_neg_zero(float x) { return __float_as_uint(x) == 0x80000000u; }

// ---- scene in LDS ---------------------------------------------------------------

#define RM_TAB_POW 24  // per-level constants staged in LDS for the iterated kinds

// LDS image of the scene: the primitive table rows (2 x float4 each) or, for
// the iterated fractal kinds, the per-level scale factors pow(scale, i).
struct SceneLds {
  float4 rows[RM_MAX_PRIMS * 2];
  float4 surf[(RM_MAX_SURFACES + 1) * 3];  // RM_TABLE_HAS_SURFACES: the surfaces' values, 3 x float4 each ([0] = the scene's material block)
  unsigned int cull_here;  // smooth sphere tables: 0 while the workgroup's rays are too far apart to share a cell (eval_spheres_one_k)
};

// :74-76
template <class M> RM_DEV float sdf_sphere(v3 p, v3 c, float r) { return distance<M>(p, c) - r; }

// :108-112
template <class M> RM_DEV float sd_box(v3 p, v3 b) {
  v3 q = vabs(p) - b;
  return length<M>(vmaxs(q, 0.0f)) + gmin(gmax(q.x, gmax(q.y, q.z)), 0.0f);
}

// the shapes and operators of ABI 8 (include/hip_raymarch.h; oracle/rm_oracle.c sd_torus ...; the composer's rmTorus ...)
template <class M> RM_DEV float length2(float x, float y) { return M::sqrt(M::fma(y, y, x * x)); }
template <class M> RM_DEV float sd_torus(v3 p, float R, float r) { return length2<M>(length2<M>(p.x, p.z) - R, p.y) - r; }
template <class M> RM_DEV float sd_cylinder(v3 p, float r, float h) {
  const float dx = length2<M>(p.x, p.z) - r, dy = fabsf(p.y) - h;
  return gmin(gmax(dx, dy), 0.0f) + length2<M>(gmax(dx, 0.0f), gmax(dy, 0.0f));
}
template <class M> RM_DEV float sd_plane(v3 p, v3 n) { return dot<M>(p, n); }
template <class M> RM_DEV float op_smooth_subtract(float d, float di, float k) {
  float h = gclamp(0.5f - M::div(0.5f * (d + di), k), 0.0f, 1.0f);
  return gmix<M>(d, -di, h) + k * h * (1.0f - h);
}
template <class M> RM_DEV float op_smooth_intersect(float d, float di, float k) {
  float h = gclamp(0.5f - M::div(0.5f * (d - di), k), 0.0f, 1.0f);
  return gmix<M>(d, di, h) + k * h * (1.0f - h);
}

// examples/smooth-tree.glsl:20-22
template <class M> RM_DEV float op_smooth_union(float d1, float d2, float k) {
  float h = gclamp(0.5f + M::div(0.5f * (d2 - d1), k), 0.0f, 1.0f);
  return gmix<M>(d2, d1, h) - k * h * (1.0f - h);
}

template <int KIND>
struct Sdf;
// The exact far-field exits of the march -- the jump, the miss, the clear miss, the shadow ray's certain comparison -- are arguments
// about the scene's distance bound with a 1e-3 rounding allowance per product: they do not depend on the arithmetic policy, and
// since round 4 the parity build takes them too (round 3 compiled them into the fast build only).  Not the GL stack's arithmetic:
// its min / max keep a NaN where IEEE's drop it, which the end states lean on.
template <class M> struct ExactExits { static constexpr bool value = !RM_GL_STACK; };

// Steps an escaping ray needs at most to reach its end state when its distance estimate is bounded below by |p| - R' (the
// bounded scenes: tables, the sponge, the rotation fractal, the sphere-grid fractal; far_r2 = (2 R' + 1)^2 >= 1): the worst case of
// the recurrence (r^2, s) -> (r^2 + 2 d s + d^2 |dir|^2, s + d |dir|^2), d = r - R', started at a right angle with |dir|^2 = 0.98
// and every step shortened by 1e-3, overflows r^2 within 64 - log2 r + 3.5 steps from any r >= 2 R' + 1 (iterated over r = 1 .. 2^50
// and every R' the start allows); 3 more settle the end state.  The overflow is ABSOLUTE -- r has to reach 1.8e19 -- so for a
// given r / R' the SMALL scenes need the most steps: from the jump's radius 66 + 3 (R' -> 0), from 50 times it 60 + 3, from 5e5
// times it 47 + 3.  Three tiers: 71, 63, 50.  (They were 72, 60 and 48 until the end of round 3, from a derivation on CSG-64's
// scale, R' ~ 3: for scenes of R' <= 1 the two far tiers were up to 3 steps short of this worst case.  No test ray realised it --
// real directions are unit and rarely tangential -- but the jump is only exact if the bound is.  The count taken exactly per step,
// 71 - floor(log2 r), would let a few more budgets jump and costs the sphere-grid kernel, whose march is short, 12 %.)
RM_DEV int far_need(float r2, float far_r2) { return r2 >= far_r2 * 2.5e11f ? 50 : (r2 >= far_r2 * 2500.0f ? 63 : 71); }
// (tests/test_far_bounds_cpu.py iterates this recurrence in double precision for R' = 1e-6 .. 5e8 and asserts the three tiers, the
// miss's 90 and the shadow exit's growth bound; tests/test_gpu_far_field.py constructs these worst cases on the GPU.)
// The conditions all those jumps share: a unit direction without a zero component (0 x Inf = NaN), and a ray that is certain to
// escape with steps to spare --
//  * outside (r^2 > far_r2 = (2 Rp)^2, Rp = R' + 1/2) and not moving inward, far_need() steps left (at most 71); or
//  * about to MISS the scene: the ray's line ahead stays m >= 1.25 Rp from the origin.  All along it d >= |x| - Rp >= Rp / 4, so
//    the march never settles: one step brings a ray from far inside to within Rp of its closest point (d >= the distance still to
//    go, minus Rp), 12 steps of >= Rp / 4 (13 with |dir|^2 = 0.98 and the rounding) take it from there to 2 Rp beyond, where
//    |x| >= sqrt(1.25^2 + 4) Rp > 2 Rp and it moves outward: the first case, from r ~ 2 Rp -- far_need's 71 steps.  Asked for: 90 left (tests/test_far_bounds_cpu.py test_miss_budget iterates it).
//    (m^2 = r^2 - (p.dir)^2 / |dir|^2 cancels: only taken where r^2 < 1000 far_r2, against the 2.4 % between 0.4 and 1.25^2 / 4; a ray
//    cast from 10^6 away -- the shadow ray of a sky pixel, raymarcher.frag:279,354-362 -- is looked at again after its first step.)
//    Three quarters of the sky pixels' shadow rays of BASELINE's C4 end here after one evaluation instead of ten.
RM_DEV bool far_escape(v3 p, v3 dir, int left, float far_r2, bool need_nonzero_dir, v3& end) {
  const float r2 = FM::fma(p.z, p.z, FM::fma(p.y, p.y, p.x * p.x));
  const float miss_m2 = FM::fma(0.4f, far_r2, 0.05f);
  if (!(r2 >= miss_m2 && r2 < 1e30f) || left < 50) return false;  // (most steps of most rays: inside -- one compare and out)
  const float s = FM::fma(p.z, dir.z, FM::fma(p.y, dir.y, p.x * dir.x)), dd = FM::fma(dir.z, dir.z, FM::fma(dir.y, dir.y, dir.x * dir.x));
  if (!(dd > 0.98f && dd < 1.02f)) return false;
  const bool outside = r2 > far_r2 && s >= 0.0f && left >= far_need(r2, far_r2);
#ifndef RM_FAR_MISS
#define RM_FAR_MISS 1
#endif
  const float m2 = s >= 0.0f ? r2 : FM::fma(-1.03f * s, s, r2);  // the closest approach ahead, squared (from below: |dir|^2 >= 0.98)
  const bool miss = RM_FAR_MISS && left >= 90 && r2 < 1000.0f * far_r2 && m2 >= miss_m2;
  if (!(outside || miss)) return false;
  if (need_nonzero_dir && (dir.x == 0.0f || dir.y == 0.0f || dir.z == 0.0f)) return false;
  end = dir * __builtin_inff();
  return true;
}

// A SHADOW ray of a table that escapes with too few steps left for the overflow (far_need) -- the 64-step bounces of BASELINE's C5: a
// sky pixel's shadow ray, cast from 10^6 away, is past the scene after five steps and has 59 to go -- marches on to the end of its
// budget, doubling its distance every step, and ends finite and very far away.  Its end point is only ever compared:
// `distance(result, adj) >= distance(pos, adj)` (raymarcher.frag:362-363).  Once that comparison is certain the march may stop:
// outside (r^2 > far_r2) and not moving inward, as for the jump, and with `left` steps to go
//  * the end point stays finite: d <= |p| + rho for a table (rho holds every shape: each term is, and min / max / the smooth minimum
//    keep it) and |dir| <= 1.01, so r + rho at most grows by 2.01 per step: asked for log2(2.01) left + log2(r + rho) <= 63.9, i.e.
//    r_end < 1.73e19 and r_end^2 < 3.0e38 (rho <= r / 2 out there: log2(r + rho) <= log2 r + 0.585);
//  * and it is far enough: the worst case of the jump's own recurrence (started at a right angle, d = r - R', |dir|^2 = 0.98, every
//    step rounded down) has r_n >= 1.25 r_0 2^(n - 3) from n = 6 on: asked for left >= 8 and log2 r + left - 3 >= need_e, where
//    2^need_e >= distance(pos, adj) + |adj| -- then |result - adj| >= r_end - |adj| >= distance(pos, adj).
// The march then ends at dir * 2^(need_e + 2), for which the comparison holds as well.  The first condition is (nearly) invariant
// along an escape -- a step takes one from `left` and adds one to log2 r -- so it has to be sharp: with the bound 2^(left + 3) r of
// a first attempt the rays that escape early never met it, and the shortcut did nothing (C5 -1.5 %).
RM_DEV bool far_shadow_escape(v3 p, v3 dir, int left, float far_r2, int need_e, v3& end) {
  const float r2 = FM::fma(p.z, p.z, FM::fma(p.y, p.y, p.x * p.x));
  if (!(r2 > far_r2 && r2 < 1e30f) || left < 8 || need_e > 100) return false;
  const float s = FM::fma(p.z, dir.z, FM::fma(p.y, dir.y, p.x * dir.x)), dd = FM::fma(dir.z, dir.z, FM::fma(dir.y, dir.y, dir.x * dir.x));
  if (!(s >= 0.0f && dd > 0.98f && dd < 1.02f)) return false;
  const float log2_r = 0.5f * __builtin_amdgcn_logf(r2);  // v_log_f32: log2, to an ulp
  if (FM::fma(1.00716f, (float)left, log2_r) > 63.9f - 0.585f - 0.02f) return false;
  if (log2_r - 0.01f + (float)(left - 3) < (float)need_e) return false;
  end = dir * __uint_as_float((unsigned int)(127 + need_e + 2) << 23);
  return true;
}
// need_e for a shadow ray from pos to adj (1000: not a shadow ray, or nothing that can be promised)
RM_DEV int shadow_need_exponent(float dist_pos_adj, v3 adj) {
  const float n = dist_pos_adj + (fabsf(adj.x) + fabsf(adj.y) + fabsf(adj.z));  // >= distance + |adj|
  if (!(n < 1e30f)) return 1000;
  const int e = (int)((__float_as_uint(n) >> 23) & 0xffu) - 126;  // n < 2^e
  return e < -20 ? -20 : e;
}

// RM_SCENE_TABLE: left fold over the LDS-resident primitive table.  Rows are
// read with one address for the whole wave (LDS broadcast); type/operator go
// through readfirstlane so the per-row switch is a scalar branch.
template <>
struct Sdf<RM_SCENE_TABLE> {
  static constexpr bool has_cost_classes = false;  // every evaluation costs the same
  // A sphere's distance at a point with a non-finite coordinate is +-Inf or NaN (|p - c| is), and every operator and
  // domain row hands that on, so the forward-difference normal there is (NaN, NaN, NaN): sceneNormal may skip its four
  // evaluations -- unless the table has a box: sdBox takes max(q, 0), which DROPS a NaN (maxNum), so a point like
  // (NaN, 1, 1) has a finite distance to a box and a finite normal (found by the random probes of the test suite; an
  // escaping ray never gets there, its finite coordinates turn NaN with the first 0 * Inf)
  static RM_DEV bool nonfinite_normal_is_nan(const DevScene& sc) { return (sc.table_flags & RM_TABLE_NO_BOXES) != 0; }
  static RM_DEV void stage(const DevScene& sc, SceneLds& lds) {
    const float4* src = reinterpret_cast<const float4*>(sc.prims);
    if (threadIdx.x == 0) lds.cull_here = 1u;
    if (sc.table_flags & RM_TABLE_UNIFORM_K) {  // spheres, one k: also a compact image, (centre, radius) per row, behind the rows:
      for (int i = threadIdx.x; i < sc.nprims; i += blockDim.x) {  // ONE ds_read_b128 per row in the fast fold
        const float4 a = src[2 * i], b = src[2 * i + 1];
        lds.rows[2 * sc.nprims + i] = make_float4(a.z, a.w, b.x, b.y);
      }
    }
    if (sc.table_flags & RM_TABLE_HAS_SURFACES) {
      const float4* ssrc = reinterpret_cast<const float4*>(sc.surfaces);
      for (int i = threadIdx.x; i < (sc.nsurfaces + 1) * 3; i += blockDim.x) lds.surf[i] = ssrc[i];
    }
    for (int i = threadIdx.x; i < sc.nprims * 2; i += blockDim.x) {
      float4 v = src[i];
      if ((sc.table_flags & RM_TABLE_SPHERES_SMOOTH) && (i & 1)) {  // second half-row of a sphere: size[1] := 0.5/k
        const float k = reinterpret_cast<const float*>(sc.prims)[(i >> 1) * 8 + 1];
        v.z = 0.5f * (1.0f / k);  // the halving is exact: the bits of 0.5f * inv_k computed per evaluation
      }
      lds.rows[i] = v;
    }
  }
  // Fast-policy path for the common table "spheres folded with smooth unions" (BASELINE configs[3]/[4]):
  // no per-row dispatch, 0.5/k read from the row (a sphere does not use size[1]; stage() puts it there),
  // two rows per trip so that the LDS reads of the next rows are in flight during the arithmetic.
  static RM_DEV float sphere_row(const float4 a, const float4 b, v3 p) {
    const v3 q = p - V(a.z, a.w, b.x);
    return FM::sqrt(FM::fma(q.z, q.z, FM::fma(q.y, q.y, q.x * q.x))) - b.y;
  }
  // (Tried: a per-wave branch around the fold for spheres further than k behind the running distance -- h = 1 there
  // and the fold is the single subtraction di - (di - d), same bits.  The branch per row breaks the two-row software
  // pipeline: CSG-64 4096^2 went from 56 to 65 ms.  Not kept.)
  // (Round 3 tried the same polynomial written from the running value's side, d + g (t - k + k g) with g = 1 - h, which returns
  // d itself once a shape is further than k away and so lets an evaluation skip such rows exactly: C4 14.2 -> 7.0 ms -- and 14 %
  // fewer lit pixels, because the rounding of d to the grid of (di - d) that THIS form, like the reference's mix(d2, d1, 1), performs
  // for every far row is the noise the creeping shadow rays of a smooth-union scene live on.  Not kept:
  // profiles/r03_row_culling_smooth_union_experiment.txt.  Round 4 drops the rows for which that rounding is provably the identity:
  // rm_params.hpp rm_cull_cell.)
  static RM_DEV float smooth_row(float d, float di, float k, float half_inv_k) {
    const float t = di - d;  // d - di is -t exactly (up to the sign of a zero): one subtraction instead of two
    const float h = gclamp(FM::fma(half_inv_k, t, 0.5f), 0.0f, 1.0f);
    // mix(di, d, h) - k h (1 - h) = di - h (t + k (1 - h)): three instructions for five (the same polynomial, rounded differently)
    return FM::fma(-h, FM::fma(k, 1.0f - h, t), di);
  }
  // (Measured in round 2: the same fold with the rows read through the scalar data cache instead of LDS -- constant
  // address space, s_load_dwordx8 per row, SGPR operands -- is 1.9x SLOWER: 14.5 against 7.7 ms on a C4 shard.  LDS it is.)
  static RM_DEV float sphere_row1(const float4 r, v3 p) {
    const v3 q = p - V(r.x, r.y, r.z);
    return FM::sqrt(FM::fma(q.z, q.z, FM::fma(q.y, q.y, q.x * q.x))) - r.w;
  }
  // one smooth-union radius for the whole table (RM_TABLE_UNIFORM_K): the compact image, one ds_read_b128 per row instead of two
#ifndef RM_CULL_SMOOTH_MAX
#define RM_CULL_SMOOTH_MAX 52  // of 64 rows: above this the unrolled fold of the whole word is the cheaper one
#endif
  static RM_DEV float eval_spheres_one_k(const DevScene& sc, const SceneLds& lds, v3 p) {
    const int n = sc.nprims;
    const float k = sc.p[0], half_inv_k = sc.p[1];
    const float4* rows = &lds.rows[2 * n];  // the compact image (stage)
    float d = sphere_row1(rows[0], p);
    // four rows per trip (round 4): one address register with immediate offsets and four reads in flight; C4 12.19 -> 11.99 ms, C5 154.8 ->
    // 148.7, their 1/8 stripes -5 %.  (Two rows per trip with the next two read ahead: C4 11.79, C5 152.1; four with a rotating read-ahead
    // of two: no gain over two, 24 registers of rows.  With HALF the LDS reads and the same arithmetic -- a diagnostic build -- the fold
    // is not faster at all, 12.9 against 12.1 ms: it is not LDS-bound; and at 6 / 5 waves per SIMD, where nothing spills, C5 takes
    // 177 / 235 ms: it lives on occupancy.)
    // The square roots of a trip back to back, in ONE asm statement (round 6).  A quarter-rate instruction among ordinary ones costs ~6.8 issue
    // slots on this chip, not the 3.2 it costs in a stream of its own kind (tools/ubench/fold_rate.hip: the fold alone, nothing around it, took
    // 0.74 of the issue slots with its square roots where the compiler puts them -- after their own row's sum, apart -- and the same fold with an
    // ordinary multiply in their place 0.89): going into and out of the transcendental pipe is paid per GROUP.  Four in a row: the fold alone
    // 21.9 -> 18.8 ms (-14 %); in the kernels C5 140.8 -> 137.5 ms.  (Eight rows and eight roots per trip: 165 spilled VGPRs, C5 140.2.)  Same operations on the
    // same values: same bits.
#ifndef RM_SQRT_GROUPS
#define RM_SQRT_GROUPS 1
#endif
    auto len2 = [&](const float4 r) { const v3 q = p - V(r.x, r.y, r.z); return FM::fma(q.z, q.z, FM::fma(q.y, q.y, q.x * q.x)); };
    auto fold_range = [&](int i, int end) {
      for (; i + 3 < end; i += 4) {
        const float4 r0 = rows[i], r1 = rows[i + 1], r2 = rows[i + 2], r3 = rows[i + 3];
#if RM_SQRT_GROUPS
        float s0 = len2(r0), s1 = len2(r1), s2 = len2(r2), s3 = len2(r3);
        asm("v_sqrt_f32 %0, %0\n\tv_sqrt_f32 %1, %1\n\tv_sqrt_f32 %2, %2\n\tv_sqrt_f32 %3, %3" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
        d = smooth_row(d, s0 - r0.w, k, half_inv_k);
        d = smooth_row(d, s1 - r1.w, k, half_inv_k);
        d = smooth_row(d, s2 - r2.w, k, half_inv_k);
        d = smooth_row(d, s3 - r3.w, k, half_inv_k);
#else
        const float d0 = sphere_row1(r0, p), d1 = sphere_row1(r1, p);
        d = smooth_row(d, d0, k, half_inv_k);
        d = smooth_row(d, d1, k, half_inv_k);
        const float d2 = sphere_row1(r2, p), d3 = sphere_row1(r3, p);
        d = smooth_row(d, d2, k, half_inv_k);
        d = smooth_row(d, d3, k, half_inv_k);
#endif
      }
      for (; i < end; i++) d = smooth_row(d, sphere_row1(rows[i], p), k, half_inv_k);
    };
    // the listed rows of one word, in table order: the list walked with s_ff1 / s_bitset0 (round 6: 3 scalar instructions per row where
    // `j = ctz(u); u &= u - 1` compiles to 7 -- the scalar unit is shared by the CU's four SIMDs, and on C4 its instructions were 44 % of the
    // vector ones: 9.01 -> 8.76 ms), two rows per trip
    auto fold_listed = [&](unsigned long long u, int w) {
      const float4* wrows = rows + 64 * w;
      auto next_row = [&]() -> const float4* {  // the address of the lowest listed row, which leaves the list
        int j;
        asm volatile("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0" : "=&s"(j), "+s"(u));
        return wrows + j;
      };
#ifndef RM_LISTED4
#define RM_LISTED4 1  // four listed rows per trip while the list holds four, their square roots in one group (C4 8.62 -> 8.48 ms, its stripes 1.70 -> 1.63)
#endif
#if RM_LISTED4
      while (__builtin_popcountll(u) >= 4) {
        const float4* a0 = next_row();
        const float4* a1 = next_row();
        const float4* a2 = next_row();
        const float4* a3 = next_row();
        const float4 r0 = *a0, r1 = *a1, r2 = *a2, r3 = *a3;
        float s0 = len2(r0), s1 = len2(r1), s2 = len2(r2), s3 = len2(r3);
        asm("v_sqrt_f32 %0, %0\n\tv_sqrt_f32 %1, %1\n\tv_sqrt_f32 %2, %2\n\tv_sqrt_f32 %3, %3" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
        d = smooth_row(d, s0 - r0.w, k, half_inv_k);
        d = smooth_row(d, s1 - r1.w, k, half_inv_k);
        d = smooth_row(d, s2 - r2.w, k, half_inv_k);
        d = smooth_row(d, s3 - r3.w, k, half_inv_k);
      }
#endif
      while (u != 0ull) {
        const float4* a0 = next_row();
        if (u != 0ull) {
          const float4* a1 = next_row();
          const float4 r0 = *a0, r1 = *a1;
#if RM_SQRT_GROUPS
          float s0 = len2(r0), s1 = len2(r1);
          asm("v_sqrt_f32 %0, %0\n\tv_sqrt_f32 %1, %1" : "+v"(s0), "+v"(s1));
          d = smooth_row(d, s0 - r0.w, k, half_inv_k);
          d = smooth_row(d, s1 - r1.w, k, half_inv_k);
#else
          const float d0 = sphere_row1(r0, p), d1 = sphere_row1(r1, p);
          d = smooth_row(d, d0, k, half_inv_k);
          d = smooth_row(d, d1, k, half_inv_k);
#endif
        } else {
          d = smooth_row(d, sphere_row1(*a0, p), k, half_inv_k);
        }
      }
    };
    if (sc.cull.cells == nullptr) {  // kernel-uniform
      fold_range(1, n);
      return d;
    }
    // Round 4: the rows of this point's cell that are not exact no-ops (rm_params.hpp rm_cull_cell: a far row of a smooth
    // union ROUNDS the running value, and where that rounding is provably the identity the row is skipped) -- about half of CSG-64's.
    // Folded: the union of the wave's cells (any superset of a lane's list gives the same bits: the extra rows are identities for it),
    // and the whole word where the wave is spread over too many cells for a common list (the rays of a diffuse bounce) or the list is
    // not much shorter than the word (RM_CULL_SMOOTH_MAX: the scalar row loop costs more per row than the unrolled one).  The
    // pixel kernel switches the lookup off for the bounces after the first (lds.cull_here): their waves seldom share a list, and the
    // cell's read and the union would be paid for nothing (C5: 156 ms with it on throughout, 151 without any culling; asking there
    // only whether the whole wave sits in ONE cell, and reading its list by a scalar address if so: 153 against 144 -- the cell index
    // alone costs more than the few coherent steps at the start of a bounce return).  Whatever is
    // folded, the bits are those of the fold of every row (RM_RENDER_NO_CULL).
    if (lds.cull_here == 0u) {  // workgroup-uniform
      fold_range(1, n);
      return d;
    }
    const unsigned long long* cell = cull_cell(sc.cull, p);
    for (int w = 0; w < sc.cull.words; w++) {
      const unsigned long long u = listed_rows(n, cell[w], w);
      if (u == 0ull) fold_range(w == 0 ? 1 : 64 * w, min(n, 64 * w + 64));
      else fold_listed(u, w);
    }
    return d;
  }
  // which rows of a word to fold: the wave's union with row 0 (the fold's start) taken out, or 0 = all of them
  static RM_DEV unsigned long long listed_rows(int n, unsigned long long mine, int w) {
    const unsigned long long u = wave_union(mine);
    const int end = min(n, 64 * w + 64);
    if (u == 0ull || __builtin_popcountll(u) > RM_CULL_SMOOTH_MAX * (end - 64 * w) / 64) return 0ull;
    return w == 0 ? (u & ~1ull) : u;
  }
  static RM_DEV float eval_spheres_smooth(const DevScene& sc, const SceneLds& lds, v3 p) {
    const int n = sc.nprims;
    if (sc.table_flags & RM_TABLE_UNIFORM_K) return eval_spheres_one_k(sc, lds, p);  // kernel-uniform
    float d = sphere_row(lds.rows[0], lds.rows[1], p);
    if (sc.cull.cells != nullptr) {  // kernel-uniform: the rows of the wave's cells, in this fold's own arithmetic (culled_rows)
      culled_rows(sc, p,
                  [&](int j, bool) {
                    const float4 a0 = lds.rows[2 * j], b0 = lds.rows[2 * j + 1];
                    d = smooth_row(d, sphere_row(a0, b0, p), a0.y, b0.z);
                  },
                  [&](int j0, int j1) {
                    const float4 a0 = lds.rows[2 * j0], b0 = lds.rows[2 * j0 + 1], a1 = lds.rows[2 * j1], b1 = lds.rows[2 * j1 + 1];
                    const float d0 = sphere_row(a0, b0, p), d1 = sphere_row(a1, b1, p);
                    d = smooth_row(d, d0, a0.y, b0.z);
                    d = smooth_row(d, d1, a1.y, b1.z);
                  });
      return d;
    }
    int i = 1;
    for (; i + 1 < n; i += 2) {
      const float4 a0 = lds.rows[2 * i], b0 = lds.rows[2 * i + 1], a1 = lds.rows[2 * i + 2], b1 = lds.rows[2 * i + 3];
      const float d0 = sphere_row(a0, b0, p), d1 = sphere_row(a1, b1, p);
      d = smooth_row(d, d0, a0.y, b0.z);
      d = smooth_row(d, d1, a1.y, b1.z);
    }
    if (i < n) {
      const float4 a0 = lds.rows[2 * i], b0 = lds.rows[2 * i + 1];
      d = smooth_row(d, sphere_row(a0, b0, p), a0.y, b0.z);
    }
    return d;
  }
  // ---- row culling (round 3; tables without domain rows; rm_params.hpp rm_cull_cell has the rule; the parity build too since round 4) --
  // A row whose operator cannot change the running value of the fold is an exact no-op: min(d, di) with di >= d, max(d, -di) and
  // max(d, di) with the term below d.  (Not the smooth union: see smooth_row.)  The scene's grid lists, per cell, the rows for which
  // that cannot be said of every point of the cell; an evaluation folds the rows listed for ITS WAVE's points -- the union of the
  // lanes' cells, so that the loop stays wave-uniform and the rows come from LDS by one address; a row listed for a neighbour is
  // a no-op for this lane by construction -- in table order; lanes too far apart for a common set fold their own rows.  Same bits
  // as the whole table (RM_RENDER_NO_CULL).
  static RM_DEV const unsigned long long* cull_cell(const CullGrid& g, v3 p) {
    const float x = p.x - g.centre[0], y = p.y - g.centre[1], z = p.z - g.centre[2];
    const float t = gmax(fabsf(x), gmax(fabsf(y), fabsf(z))) * g.inv_half0;  // the max-norm in units of level 0's half-width
    // level: 0 inside level 0's cube, else ceil(log2 t) read off the exponent (t in [2^(l-1), 2^l) -> l); a point with a NaN or an
    // Inf and anything beyond the last level get the cell that lists every row
    const int e = (int)((__float_as_uint(t) >> 23) & 0xffu) - 126;
    const int level = t < 1.0f ? 0 : e;
    // (v_max drops a NaN, so t does not see one: the 1-norm does -- NaN or Inf for a point with a non-finite coordinate)
    const bool inside = fabsf(x) + fabsf(y) + fabsf(z) < 3.0e38f && level < g.levels;
    const bool outer = inside && level > 0;
    const int nl = outer ? g.n_outer : g.n;  // cells across this point's level
    // scale0 * 2^-level, and half of that where the outer levels' cells are twice as wide (n_outer is n or n / 2)
    const float scale = g.scale0 * __uint_as_float((unsigned int)(127 - (inside ? level : 0) - (outer && g.n_outer != g.n ? 1 : 0)) << 23);
    const float h = 0.5f * (float)nl;
    const int top = nl - 1;
    const int ix = min(max(__float2int_rd(FM::fma(x, scale, h)), 0), top), iy = min(max(__float2int_rd(FM::fma(y, scale, h)), 0), top),
              iz = min(max(__float2int_rd(FM::fma(z, scale, h)), 0), top);
    const int level0 = g.n * g.n * g.n, per_outer = g.n_outer * g.n_outer * g.n_outer;
    const int idx = !inside ? level0 + (g.levels - 1) * per_outer : (outer ? level0 + (level - 1) * per_outer : 0) + (iz * nl + iy) * nl + ix;
    return g.cells + (size_t)idx * (size_t)g.words;
  }
  // the union of the active lanes' row sets, as a scalar: the first lane's set, then -- a few times -- the set of the first lane that
  // still has a row outside it (lanes of a wave sit in the same or in neighbouring cells); 0 when that does not cover the wave (the
  // word that holds row 0 is never 0 otherwise)
#ifndef RM_CULL_UNION_ROUNDS
#define RM_CULL_UNION_ROUNDS 8  // (round 5: C4 9.35 -> 9.08 ms, C5 141.1 -> 140.0; 16 and an unbounded loop gain no more; rounds 3-4: 3)
#endif
  static RM_DEV unsigned long long wave_union(unsigned long long mine) {
    const unsigned int lo = (unsigned int)mine, hi = (unsigned int)(mine >> 32);
    unsigned long long u = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)lo);
#ifndef RM_CULL_UNION_LOOP
#define RM_CULL_UNION_LOOP 0  // 1 (measurement builds): the rounds as a loop, not unrolled
#endif
#if RM_CULL_UNION_LOOP
#pragma unroll 1
#else
#pragma unroll
#endif
    for (int it = 0; it < RM_CULL_UNION_ROUNDS; it++) {
      const unsigned long long more = ballot((mine & ~u) != 0ull);
      if (more == 0ull) return u;
      const int src = __builtin_ctzll(more);
      u |= ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)hi, src) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)lo, src);
    }
    return ballot((mine & ~u) != 0ull) != 0ull ? 0ull : u;
  }
  // calls row(i, uniform) for the rows 1 .. n-1 this lane has to fold, in table order -- uniform: every lane of the wave is at row
  // i (two at a time where there are two, rows2(i, j): the LDS reads of both are in flight during the arithmetic); else the lanes
  // are at rows of their own; row 0 starts the fold and is the caller's.  (Measured on a 64-row table, tools/r03_table.py: folding
  // the union when it exists and per lane otherwise beats the union alone by 4 %, per lane alone loses 10 %, and falling back to the
  // whole table when the union fails loses 14 %.)
  template <class F1, class F2>
  static RM_DEV void culled_rows(const DevScene& sc, v3 p, F1&& row, F2&& rows2) {
    const unsigned long long* cell = cull_cell(sc.cull, p);
    for (int w = 0; w < sc.cull.words; w++) {
      const unsigned long long mine = cell[w];
      unsigned long long u = wave_union(mine);
      if (u == 0ull) {  // no common set (or no row of this word at all): every lane folds its own rows, the two halves of the word in turn
        unsigned int half[2] = {(unsigned int)mine, (unsigned int)(mine >> 32)};
        if (w == 0) half[0] &= ~1u;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          unsigned int m = half[h];
          while (ballot(m != 0u) != 0ull) {
            if (m != 0u) {
              const int j = 64 * w + 32 * h + (int)__builtin_ctz(m);
              m &= m - 1u;
              row(j, false);
            }
          }
        }
        continue;
      }
      if (w == 0) u &= ~1ull;
      auto next_listed = [&]() -> int {  // the lowest listed row, which leaves the list: s_ff1 + s_bitset0 (see eval_spheres_one_k)
        int j;
        asm volatile("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0" : "=&s"(j), "+s"(u));
        return 64 * w + j;
      };
      while (u != 0ull) {
        const int j0 = next_listed();
        if (u != 0ull) {
          const int j1 = next_listed();
          rows2(j0, j1);
        } else {
          row(j0, true);
        }
      }
    }
  }
  // one shape row of the general fold (no domain rows): its distance, operator and k; `uniform`: the wave is at one row (its type is
  // then a scalar and the dispatch below a scalar branch)
  template <class M>
  static RM_DEV float shape_row(const SceneLds& lds, int i, v3 q, bool uniform, int& op, float& k) {
    const float4 a = lds.rows[2 * i];      // type, k, cx, cy
    const float4 b = lds.rows[2 * i + 1];  // cz, sx, sy, sz
    const int type = uniform ? __builtin_amdgcn_readfirstlane(__float_as_int(a.x)) : __float_as_int(a.x);
    op = (type >> 8) & 0xff;
    k = a.y;
    const v3 c = V(a.z, a.w, b.x);
    return (type & 0xff) == RM_PRIM_SPHERE ? sdf_sphere<M>(q, c, b.y) : sd_box<M>(q - c, V(b.y, b.z, b.w));
  }
  // MORE: the table uses ABI 8's shapes or smooth operators (RM_TABLE_MORE, kernel-uniform) -- compiled as a second copy of the fold, so that
  // the tables of the older vocabulary keep the code they had (with the two operators and three shapes in ONE fold, a table of 192 rows of
  // hard operators lost 12 %: 15.6 -> 17.6 ms at 1080p)
  template <class M, bool MORE = false>
  static RM_DEV float apply_op(float d, float di, int op, float k) {
    if (op == RM_OP_UNION) return gmin(d, di);
    if (op == RM_OP_SMOOTH_UNION) return op_smooth_union<M>(d, di, k);
    if (op == RM_OP_SUBTRACT) return gmax(d, -di);
    if (MORE && op == RM_OP_SMOOTH_SUBTRACT) return op_smooth_subtract<M>(d, di, k);
    if (MORE && op == RM_OP_SMOOTH_INTERSECT) return op_smooth_intersect<M>(d, di, k);
    return gmax(d, di);
  }
  // the distance term of a shape row other than a sphere or a kind: box, torus, capped cylinder, plane (about the row's centre)
  template <class M>
  static RM_DEV float other_shape(int prim, v3 at, float4 b) {
    if (prim == RM_PRIM_TORUS) return sd_torus<M>(at, b.y, b.z);
    if (prim == RM_PRIM_CYLINDER) return sd_cylinder<M>(at, b.y, b.z);
    if (prim == RM_PRIM_PLANE) return sd_plane<M>(at, V(b.y, b.z, b.w));
    return sd_box<M>(at, V(b.y, b.z, b.w));
  }
  template <class M>
  static RM_DEV float eval_general_culled(const DevScene& sc, const SceneLds& lds, v3 p) {
    int op;
    float k;
    float d = shape_row<M>(lds, 0, p, true, op, k);
    auto one = [&](int i, bool uniform) {
      const float di = shape_row<M>(lds, i, p, uniform, op, k);
      d = apply_op<M>(d, di, op, k);
    };
    culled_rows(sc, p, one, [&](int i, int j) { one(i, true); one(j, true); });
    return d;
  }
  // one level of a kaleidoscopic fold (RM_PRIM_FOLD): the operations of tree.glsl:24-32 on the running point
  template <class M>
  static RM_DEV v3 fold_row(v3 q, float scale, v3 off, v3 ang) {
    q = V(M::div(q.x, scale), M::div(q.y, scale), M::div(q.z, scale));
    q = vabs(q) - off;
    float c, s, nx, ny;
    PM::sincos(ang.x, s, c);  // angles of a table row: precise in both builds
    nx = M::fma(q.y, -s, q.x * c); ny = M::fma(q.y, c, q.x * s); q.x = nx; q.y = ny;
    PM::sincos(ang.y, s, c);
    nx = M::fma(q.z, -s, q.y * c); ny = M::fma(q.z, c, q.y * s); q.y = nx; q.z = ny;
    PM::sincos(ang.z, s, c);
    nx = M::fma(q.z, -s, q.x * c); ny = M::fma(q.z, c, q.x * s); q.x = nx; q.z = ny;
    return q;
  }
  // ---- the far field of a table, jumped (round 3; fast policy; rm_api.hip table_far_field has the argument) -----------------
  // A ray outside 2 R' (far_r2) that is not moving inward reaches the overflow of |p|^2 within far_need() steps; its end state is
  // the scene's far_end.  The +-Inf pattern is only taken for directions without a zero component (0 x Inf = NaN, and what a
  // NaN coordinate does next depends on the shapes: a box drops it); such rays march on.  Exact, like the Mandelbulb's jump:
  // tested against RM_RENDER_NO_FAR_JUMP on BASELINE's CSG frames and on random tables through the probe.
  static RM_DEV bool far_jump_applies(const DevScene& sc) { return sc.far_end != 0; }
  static RM_DEV bool far_jump(const DevScene& sc, v3 p, v3 dir, int left, v3& end) {
    if (!far_escape(p, dir, left, sc.far_r2, sc.far_end != 2, end)) return false;
    if (sc.far_end == 2) {
      const float nan = __builtin_nanf("");
      end = V(nan, nan, nan);
    }
    return true;
  }

  // A ray that passes every shape of the table at a distance (round 3; fast policy; rm_api.hip table_far_field has the argument): the
  // scene's bounding sphere is a poor judge of that -- a camera looking AT the scene sends every ray through it -- but the shapes'
  // own bounding spheres are not.  If the half-line p + t dir, t >= 0, stays k_max + b0 clear of every one of them, the march never
  // settles, leaves the scene within left - 71 steps (b0 is chosen for the steps there are; 84 are set aside) and ends where the jump ends: most of
  // C4's sky takes this exit at the first step of its camera ray instead of marching past the cluster for seven.  One pass over
  // the rows, ~15 instructions each (about one evaluation); asked of a ray once, at the start of its march, and only from outside
  // the scene's own sphere (a shadow ray leaving a surface has no clearance to show).  dist^2 = |v|^2 - (v.dir)^2 / |dir|^2 from
  // below (|dir|^2 >= 0.98), compared with 2 % and 0.01 to spare.  Exact like the jump: RM_RENDER_NO_FAR_JUMP switches it off.
  static RM_DEV bool clear_miss_applies(const DevScene& sc, v3 p, v3 dir, int left) {
    if (left < 100 || sc.clear_rho == 0.0f) return false;
    const float r2 = FM::fma(p.z, p.z, FM::fma(p.y, p.y, p.x * p.x)), dd = FM::fma(dir.z, dir.z, FM::fma(dir.y, dir.y, dir.x * dir.x));
    if (!(r2 > 0.3f * sc.far_r2 && r2 < 1000.0f * sc.far_r2 && dd > 0.98f && dd < 1.02f)) return false;  // outside 1.1 Rp, not too far for the cancellation
    return sc.far_end == 2 || !(dir.x == 0.0f || dir.y == 0.0f || dir.z == 0.0f);
  }
  static RM_DEV bool clear_miss(const DevScene& sc, const SceneLds& lds, v3 p, v3 dir, int left) {
    const int n = sc.nprims;
    const float kb = sc.clear_k + 2.05f * sc.clear_rho / (float)(gmin(left, 4096) - 84);  // k_max + b0 (left >= 100)
    bool clear = true;
    auto row = [&](v3 c, float e) {
      const v3 v = c - p;
      const float vv = FM::fma(v.z, v.z, FM::fma(v.y, v.y, v.x * v.x));
      const float t = gmax(FM::fma(v.z, dir.z, FM::fma(v.y, dir.y, v.x * dir.x)), 0.0f);
      const float m2 = FM::fma(-1.03f * t, t, vv), reach = e + kb;
      clear = clear && m2 >= FM::fma(1.02f * reach, reach, 0.01f);
    };
    if (sc.table_flags & RM_TABLE_UNIFORM_K) {  // kernel-uniform: the compact rows (centre, radius)
      const float4* rows = &lds.rows[2 * n];
      for (int i = 0; i < n; i++) {
        const float4 r = rows[i];
        row(V(r.x, r.y, r.z), fabsf(r.w));
      }
    } else {
      for (int i = 0; i < n; i++) {
        const float4 a = lds.rows[2 * i], b = lds.rows[2 * i + 1];
        const bool sphere = (__builtin_amdgcn_readfirstlane(__float_as_int(a.x)) & 0xff) == RM_PRIM_SPHERE;
        row(V(a.z, a.w, b.x), sphere ? fabsf(b.y) : PM::sqrt(b.y * b.y + b.z * b.z + b.w * b.w) * 1.0001f);
      }
    }
    return clear;
  }
  static RM_DEV v3 far_end_state(const DevScene& sc, v3 dir) {
    if (sc.far_end == 2) {
      const float nan = __builtin_nanf("");
      return V(nan, nan, nan);
    }
    return dir * __builtin_inff();
  }

  // Which surface the material functions use at p (RM_TABLE_HAS_SURFACES; include/hip_raymarch.h RmSurface): the one the
  // shape row with the smallest distance term names -- the terms of eval()'s fold, row by row, before their operators --
  // the earliest row on a tie; a NaN term never wins (`<` is false), so a point whose terms are all NaN has the first
  // shape row's surface.  The composer emits the same loop as GLSL (scene.py rmSurfaceIndex).
  // a RM_PRIM_KIND row's term: the scene kind's own estimator at q - centre, its parameters in the scene block (include/hip_raymarch.h)
  template <class M>
  static RM_DEV float kind_row(const DevScene& sc, const SceneLds& lds, int kind, v3 at);
  template <class M, bool KINDS = false>
  static RM_DEV int surface_index(const DevScene& sc, const SceneLds& lds, v3 p) {
    if (sc.table_flags & RM_TABLE_MORE) return surface_index_of<M, KINDS, true>(sc, lds, p);  // kernel-uniform
    return surface_index_of<M, KINDS, false>(sc, lds, p);
  }
  template <class M, bool KINDS, bool MORE>
  static RM_DEV int surface_index_of(const DevScene& sc, const SceneLds& lds, v3 p) {
    const bool domain = (sc.table_flags & RM_TABLE_HAS_DOMAIN) != 0;  // kernel-uniform
    float best = 0.0f, factor = 1.0f;
    int surface = 0;
    bool first = true;
    v3 q = p;
    const int n = sc.nprims;
    for (int i = 0; i < n; i++) {
      const float4 a = lds.rows[2 * i];
      float4 b = lds.rows[2 * i + 1];
      const int type = __builtin_amdgcn_readfirstlane(__float_as_int(a.x));
      const v3 c = V(a.z, a.w, b.x);
      const int prim = type & 0xff;
      if (domain && prim == RM_PRIM_REPEAT) {
        q = V(gmod<M>(q.x + 0.5f * b.y, b.y) - 0.5f * b.y, gmod<M>(q.y + 0.5f * b.z, b.z) - 0.5f * b.z, gmod<M>(q.z + 0.5f * b.w, b.w) - 0.5f * b.w);
        continue;
      }
      if (domain && prim == RM_PRIM_FOLD) {
        q = fold_row<M>(q, a.y, c, V(b.y, b.z, b.w));
        factor = factor * a.y;
        continue;
      }
      float di;
      if (prim == RM_PRIM_SPHERE) di = sdf_sphere<M>(q, c, b.y);
      else if (KINDS && prim == RM_PRIM_KIND) di = kind_row<M>(sc, lds, __builtin_amdgcn_readfirstlane((int)b.y), q - c);
      else di = MORE ? other_shape<M>(prim, q - c, b) : sd_box<M>(q - c, V(b.y, b.z, b.w));
      if (domain) di = di * factor;
      if (first || di < best) { best = di; surface = (type >> 16) & 0xff; }
      first = false;
    }
    return surface;
  }
  template <class M>
  static RM_DEV float eval(const DevScene& sc, const SceneLds& lds, v3 p) {
    if (M::fast && (sc.table_flags & RM_TABLE_SPHERES_SMOOTH)) return eval_spheres_smooth(sc, lds, p);
    // kernel-uniform.  Both builds (round 4; round 3: the fast one only): which rows are identities at a point is an argument about the
    // shapes' distances with an fp32 allowance and, for a far smooth-union row, about  d' = fl(di - fl(di - d))  -- which mix(di, d, 1)
    // is in either arithmetic.  Not the GL stack's (ExactExits).
    if (ExactExits<M>::value && sc.cull.cells != nullptr) return eval_general_culled<M>(sc, lds, p);
    return eval_general<M>(sc, lds, p);
  }
  template <class M, bool KINDS = false>
  static RM_DEV float eval_general(const DevScene& sc, const SceneLds& lds, v3 p) {
    if (sc.table_flags & RM_TABLE_MORE) return eval_general_of<M, KINDS, true>(sc, lds, p);  // kernel-uniform
    return eval_general_of<M, KINDS, false>(sc, lds, p);
  }
  template <class M, bool KINDS, bool MORE>
  static RM_DEV float eval_general_of(const DevScene& sc, const SceneLds& lds, v3 p) {
    const bool domain = (sc.table_flags & RM_TABLE_HAS_DOMAIN) != 0;  // kernel-uniform
    float d = 0.0f, factor = 1.0f;
    bool first = true;
    v3 q = p;
    const int n = sc.nprims;
    for (int i = 0; i < n; i++) {
      const float4 a = lds.rows[2 * i];      // type, k, cx, cy
      const float4 b = lds.rows[2 * i + 1];  // cz, sx, sy, sz
      const int type = __builtin_amdgcn_readfirstlane(__float_as_int(a.x));
      const v3 c = V(a.z, a.w, b.x);
      const int prim = type & 0xff;
      if (domain && prim == RM_PRIM_REPEAT) {  // q = mod(q + 0.5 * period, period) - 0.5 * period
        q = V(gmod<M>(q.x + 0.5f * b.y, b.y) - 0.5f * b.y, gmod<M>(q.y + 0.5f * b.z, b.z) - 0.5f * b.z, gmod<M>(q.z + 0.5f * b.w, b.w) - 0.5f * b.w);
        continue;
      }
      if (domain && prim == RM_PRIM_FOLD) {
        q = fold_row<M>(q, a.y, c, V(b.y, b.z, b.w));
        factor = factor * a.y;
        continue;
      }
      float di;
      if (prim == RM_PRIM_SPHERE) di = sdf_sphere<M>(q, c, b.y);
      else if (KINDS && prim == RM_PRIM_KIND) di = kind_row<M>(sc, lds, __builtin_amdgcn_readfirstlane((int)b.y), q - c);
      else di = MORE ? other_shape<M>(prim, q - c, b) : sd_box<M>(q - c, V(b.y, b.z, b.w));
      if (domain) di = di * factor;
      if (first) { d = di; first = false; continue; }
      const int op = (type >> 8) & 0xff;
      d = apply_op<M, MORE>(d, di, op, a.y);
    }
    return d;
  }
};

// RM_SCENE_MANDELBULB: spherical-coordinate power-n distance estimator, the
// text scene.Mandelbulb emits (the reference has no Mandelbulb scene).
template <>
struct Sdf<RM_SCENE_MANDELBULB> {
  static RM_DEV void stage(const DevScene&, SceneLds&) {}
  template <class M>
  static RM_DEV void generic_round(v3& z, float& dr, v3 pos, float r, float power) {
    float theta = M::acos(M::div(z.z, r));
    float phi = M::atan2(z.y, z.x);
    float r_nm1, zr;
    M::pow_pair(r, power, r_nm1, zr);
    dr = r_nm1 * power * dr + 1.0f;
    theta = theta * power;
    phi = phi * power;
    float st, ct, sp, cp;
    M::sincos(theta, st, ct);
    M::sincos(phi, sp, cp);
    z = V(st * cp, sp * st, ct) * zr;
    z = z + pos;
  }
  template <class M>
  static RM_DEV float eval_generic(const DevScene& sc, v3 pos) {
    const float power = sc.p[RM_P_BULB_POWER];
    const int iterations = (int)sc.p[RM_P_BULB_ITERATIONS];
    const float bailout = sc.p[RM_P_BULB_BAILOUT];
    v3 z = pos;
    float dr = 1.0f, r = 0.0f;
    for (int i = 0; i < iterations; i++) {
      r = M::sqrt_of_ordinary(dot<M>(z, z));  // length(z)
      if (r > bailout) break;
      generic_round<M>(z, dr, pos, r, power);
    }
    return M::div(0.5f * M::log(r) * r, dr);
  }
  // Power 8 without trigonometry: with rho = |z.xy|, the angles' 8-fold multiples come from 8th powers of complex
  // numbers, each by three complex squarings.  The same function as eval_generic up to
  // rounding; 1 sqrt + 1 rsq per iteration instead of 12 transcendentals.
  // x^2 + y^2 of a round, plus 1e-30 (the same float unless it is below 1e-23): see pow8_round
  static RM_DEV float pow8_rho2(v3 z) { return FM::fma(z.y, z.y, FM::fma(z.x, z.x, 1e-30f)); }
  // One round z -> z^8 + pos, dr -> 8 r^7 dr + 1.
  // 2 transcendentals and 29 other instructions per round (it was 40 before the output modifiers).
  // r = sqrt(r2) and q = 1/rho = rsq(rho2); rho = rho2 * q.
  // (z.z + i rho)^8 = r^8 (cos 8theta + i sin 8theta) =: A + iB is taken as it stands; the azimuth comes from the
  // UNIT vector (z.x + i z.y) / rho, whose 8th power is cos 8phi + i sin 8phi =: C + iD with no rho^8 to divide
  // out again.  On the axis x = y = 0 and rsq(0) = Inf would turn 0 * q into NaN: rho2 carries 1e-30 (pow8_rho2,
  // folded into its first fma), so there q is finite, C + iD = 0 and B = 0 anyway (sin 8theta = 0), and the new z
  // is (pos.x, pos.y, A + pos.z) as it should be.
  // (Round 4, tried: the whole round as ONE hand-scheduled asm statement.  hipcc's hazard recogniser assumes that the result of an asm
  // statement may be forwarded like an SDWA / op_sel write and puts an `s_nop 0` before an instruction that reads one straight away:
  // four per round with the one-instruction statements below, 530 in the headline kernel; a VOP3 output modifier has no such hazard.
  // One statement per round leaves one s_nop and 2 % fewer VALU instructions -- and the headline frame 6 % SLOWER, 1.93 against
  // 1.82 ms, with the two transcendentals next to each other or four instructions apart: the no-ops are not what the round waits
  // for, and the compiler's interleaving of the round with the code around it is worth more than they cost.  Not kept.)
#ifndef RM_TRANS_GROUPS
#define RM_TRANS_GROUPS 1  // the two transcendentals of a round, and of the distance, back to back in one asm statement (see eval_spheres_one_k: going into and out of the quarter-rate pipe is paid per group)
#endif
  static RM_DEV void pow8_round(v3& z, float& dr, v3 pos, float rho2, float r2) {
#if RM_TRANS_GROUPS
    float r, q;
    asm("v_sqrt_f32 %0, %2\n\tv_rsq_f32 %1, %3" : "=&v"(r), "=&v"(q) : "v"(r2), "v"(rho2));
#else
    const float r = FM::sqrt(r2);
    const float q = __builtin_amdgcn_rsqf(rho2);
#endif
    const float rho = rho2 * q;
    const float r4 = r2 * r2;
    dr = FM::fma_x2(FM::mul_x4(r4 * r2, r), dr, 0.5f);  // 8 r^7 dr + 1 = 2 ((4 r^7) dr + 0.5)
    // three complex squarings each.  A + iB: re' = re^2 - im^2 (mul + fma), im' = 2 re im (one mul, doubled by its
    // output modifier).  C + iD has modulus 1: re' = 2 re^2 - 1 (one fma, doubled), im' = 2 re im.
    float A = z.z, B = rho, C = z.x * q, D = z.y * q, t;
#pragma unroll
    for (int s = 0; s < 3; s++) {
      t = FM::fma(A, A, -(B * B)); B = FM::mul_x2(A, B); A = t;
      t = FM::sq_minus_half_x2(C); D = FM::mul_x2(C, D); C = t;
    }
    // (The same squarings written with |A + iB|^2 = r2 too -- 2 (A^2 - r2/2), one instruction less per round -- time
    // the same and let the modulus drift faster; packed fp32 -- v_pk_mul / v_pk_fma on the pairs (A, C), (B, D) --
    // was 2.80 ms against 2.69 on the headline frame: a packed instruction takes two issue slots of a busy SIMD.)
    z = V(FM::fma(B, C, pos.x), FM::fma(B, D, pos.y), A + pos.z);
  }
  // ... and the NEXT round's rho2 and r2
  static RM_DEV void pow8_round_next(v3& z, float& dr, v3 pos, float& rho2, float& r2) {
    pow8_round(z, dr, pos, rho2, r2);
    rho2 = pow8_rho2(z);
    r2 = FM::fma(z.z, z.z, rho2);
  }
  // the last of a fixed number of rounds: only its dr is read
  static RM_DEV float pow8_round_dr(float dr, float r2) { return FM::fma_x2(FM::mul_x4((r2 * r2) * r2, FM::sqrt(r2)), dr, 0.5f); }
  // 0.5 log(r) r / dr with r = sqrt(r2): log2(r2) and sqrt(r2) both start from r2 (no chain through r), and the
  // constants fold: 0.5 * ln 2 * 0.5 = 0.17328680
  // (Tried: skipping the reciprocal in waves whose lanes all bailed out before the first round -- dr = 1 everywhere,
  // rcp(1) = 1 and x * 1 = x exactly; such far-field steps are ~16 % of the headline frame's issue slots.  The ballot
  // and branch per evaluation cost more than the skipped instruction saves: 2.07 against 2.06 ms.)
  // Round 3: two transcendentals per evaluation instead of three.  sqrt(r2) / dr = r2 * rsq(r2 dr^2): one v_rsq for the
  // v_sqrt + v_rcp pair (a transcendental costs 3.2 issue slots, the two extra multiplications one each); when dr^2
  // overflows (dr > 1.8e19: the point is on the set, the true step is < 1e-18) rsq(Inf) = 0 makes the step 0, which moves
  // the ray exactly as far as the true value did.  A point that bailed out before its first round has dr = 1: no
  // reciprocal at all (pow8_distance_far: the bits of the three-transcendental form, rcp(1) = 1 and x * 1 = x).
  static RM_DEV float pow8_distance(float r2, float dr) {
#ifdef RM_DIST_3T  // experiment builds: round 2's form
    return __builtin_amdgcn_logf(r2) * 0.17328680f * FM::sqrt(r2) * FM::rcp(dr);
#else
#if RM_TRANS_GROUPS
    float l, w;
    const float a = (r2 * dr) * dr;
    asm("v_log_f32 %0, %2\n\tv_rsq_f32 %1, %3" : "=&v"(l), "=&v"(w) : "v"(r2), "v"(a));
    return (l * 0.17328680f) * (r2 * w);
#else
    return (__builtin_amdgcn_logf(r2) * 0.17328680f) * (r2 * __builtin_amdgcn_rsqf((r2 * dr) * dr));
#endif
#endif
  }
  static RM_DEV float pow8_distance_far(float r2) {
#if RM_TRANS_GROUPS
    float l, r;
    asm("v_log_f32 %0, %2\n\tv_sqrt_f32 %1, %2" : "=&v"(l), "=&v"(r) : "v"(r2));
    return l * 0.17328680f * r;
#else
    return __builtin_amdgcn_logf(r2) * 0.17328680f * FM::sqrt(r2);
#endif
  }
  // ITERS = 8: the usual round count, unrolled -- no loop bookkeeping between the exec-mask regions (11 % on the
  // headline frame); ITERS = 0: the count is a scene parameter
  template <int ITERS>
  static RM_DEV float eval_pow8_n(const DevScene& sc, v3 pos) {
    const float bail2 = sc.p[RM_P_BULB_BAILOUT] * sc.p[RM_P_BULB_BAILOUT];
    const float rho2 = pow8_rho2(pos);
    const float r2 = FM::fma(pos.z, pos.z, rho2);
    if (ITERS != 8 && (int)sc.p[RM_P_BULB_ITERATIONS] < 1) return pow8_distance(0.0f, 1.0f);  // no round at all: r stays 0 (as in eval_generic)
    if (r2 > bail2) {  // the far field: no round, dr = 1
#ifdef RM_LANE_STATS
      lane_stats(0);
#endif
      return pow8_distance_far(r2);
    }
    return pow8_near<ITERS>(sc, pos, rho2, r2, bail2);
  }
  // the rounds of a point inside the bailout sphere (at least one), and its distance
  template <int ITERS>
  static RM_DEV float pow8_near(const DevScene& sc, v3 pos, float rho2, float r2, float bail2) {
    v3 z = pos;
    float dr = 1.0f;
    const int iterations = ITERS == 8 ? 8 : (int)sc.p[RM_P_BULB_ITERATIONS];
#ifdef RM_LANE_STATS
    int rounds = 0;
#endif
    if (ITERS != 8 && iterations == 1) return pow8_distance(r2, pow8_round_dr(dr, r2));  // (the distance reads the r2 its last round started from)
    pow8_round_next(z, dr, pos, rho2, r2);
#ifdef RM_LANE_STATS
    rounds++;
#endif
    if (ITERS == 8) {
#pragma unroll
      for (int i = 1; i < 8; i++) {
        if (r2 > bail2) break;
        if (i == 7) dr = pow8_round_dr(dr, r2);  // (what the compiler's dead-code elimination left of the C++ round)
        else pow8_round_next(z, dr, pos, rho2, r2);
#ifdef RM_LANE_STATS
        rounds++;
#endif
      }
    } else {
      for (int i = 1; i < iterations; i++) {
        if (r2 > bail2) break;
        if (i == iterations - 1) dr = pow8_round_dr(dr, r2);
        else pow8_round_next(z, dr, pos, rho2, r2);
      }
    }
#ifdef RM_LANE_STATS
    lane_stats(rounds);
#endif
    return pow8_distance(r2, dr);
  }
#ifdef RM_LANE_STATS
  static RM_DEV void lane_stats(int rounds) {  // diagnostic build (tools/lane_stats.py): lanes x rounds used against lanes x rounds issued
    const unsigned long long act = ballot(true);
    unsigned long long lane_rounds = 0, wave_rounds = 0;
    for (int r = 1; r <= 8; r++) {
      const unsigned long long m = ballot(rounds >= r);
      lane_rounds += __popcll(m);
      wave_rounds += m != 0ull ? 1 : 0;
    }
    if ((int)__lane_id() == __ffsll((long long)act) - 1) {
      atomicAdd(&g_lane_stats[0], lane_rounds);
      atomicAdd(&g_lane_stats[1], wave_rounds * 64ull);
      atomicAdd(&g_lane_stats[2], (unsigned long long)__popcll(act));
      atomicAdd(&g_lane_stats[3], 64ull);
    }
  }
#endif
  static RM_DEV float eval_pow8(const DevScene& sc, v3 pos) {
    return (int)sc.p[RM_P_BULB_ITERATIONS] == 8 ? eval_pow8_n<8>(sc, pos) : eval_pow8_n<0>(sc, pos);
  }
  template <class M>
  static RM_DEV float eval(const DevScene& sc, const SceneLds&, v3 p) {
    if (M::fast && sc.p[RM_P_BULB_POWER] == 8.0f) return eval_pow8(sc, p);
    return eval_generic<M>(sc, p);
  }
  // ---- the far field, jumped (round 3; fast policy, power 8) ------------------------------------------------
  // Outside the bailout sphere the estimate is d = 0.25 ln(r^2) r (no round, dr = 1), and a ray that is out there and
  // not moving inward (p . dir >= 0) never comes back: with s = p . dir, one step maps (r^2, s) to (r^2 + 2 d s + d^2,
  // s + d), both growing, d >= 0.3466 r for r >= 2.  So the rest of its march is known: r^2 reaches the overflow of
  // fp32 in at most 26 steps (worst case: r = 2, s = 0, |dir|^2 = 0.98, every product rounded down by 1e-3; from r = 100
  // it takes 18, from 1e6 12), the step after that is d = Inf, which makes every coordinate +-Inf by the sign of its
  // direction component (NaN where that is 0: 0 x Inf), and that pattern is the march's fixed point -- unless it holds a
  // NaN: then the next evaluation is NaN (NaN > bailout is false: the rounds run on it) and one more step leaves every
  // coordinate NaN.  A ray with at least far_jump_steps steps left therefore ends exactly there, and 89 % of the
  // headline frame's pixels end their camera ray this way and all of those their shadow ray (the skipped steps were ~5 % of
  // the frame's issue slots).  Exact: the end point has the bits the stepwise march produces (tested on the whole
  // frame against RM_RENDER_NO_FAR_JUMP); both implementations jump (cast_ray / cast_ray_block, wf_march) and stay bit-identical.
  // (from r = 100 the overflow takes 18 steps, from 1e6 12: the rays a bounce starts a million units out -- :279 -- qualify with 16 left)
  static constexpr int far_jump_steps = 30;
  // (Round 4: any power.  Outside the bailout sphere no round runs, so the estimate is 0.5 ln(r) r whatever the power, in eval_generic
  // on either policy as in the power-8 form.  The generic evaluation compares r = sqrt(x^2 + y^2 + z^2) with the bailout in its own
  // rounding, so there the jump asks for 0.1 % more than the bailout sphere: the evaluations it replaces are certain to be far ones.)
  static RM_DEV bool far_jump_applies(const DevScene& sc) { return sc.p[RM_P_BULB_ITERATIONS] >= 1.0f && sc.p[RM_P_BULB_BAILOUT] >= 0.0f; }
  static RM_DEV bool far_jump(const DevScene& sc, v3 p, v3 dir, int left, v3& end) {
    const float bail2 = sc.p[RM_P_BULB_BAILOUT] * sc.p[RM_P_BULB_BAILOUT];
    const float r2 = FM::fma(p.z, p.z, pow8_rho2(p));  // what the power-8 evaluation starts with
    return far_jump_at(p, dir, left, r2, RM_BUILD_FAST && sc.p[RM_P_BULB_POWER] == 8.0f ? bail2 : 1.001f * bail2, end);
  }
  static RM_DEV bool far_jump_at(v3 p, v3 dir, int left, float r2, float bail2, v3& end) {
    if (!(r2 > gmax(bail2, 4.0f) && r2 < 1e30f) || left < (r2 >= 1e12f ? 16 : (r2 >= 1e4f ? 22 : far_jump_steps))) return false;
    const float s = dot<FM>(p, dir), dd = dot<FM>(dir, dir);
    if (!(s >= 0.0f && dd > 0.98f && dd < 1.02f)) return false;
    v3 e = dir * __builtin_inff();
    if (e.x != e.x || e.y != e.y || e.z != e.z) e = V(e.x + e.y + e.z, e.x + e.y + e.z, e.x + e.y + e.z);
    end = e;
    return true;
  }
  // One step's evaluation with the jump's test inside it (round 4; the block march of the pixel kernel): the jump can only apply
  // where the evaluation takes its far branch (r2 > bail2), so a ray inside the bailout sphere -- most steps of the rays that show
  // the fractal -- does not compute r2 twice and does not see the test at all.  The same decisions on the same values as
  // far_jump() followed by eval(): true and `end`, or false and `d`.
  template <int ITERS>
  static RM_DEV bool pow8_eval_or_jump(const DevScene& sc, v3 p, v3 dir, int left, bool jump, float& d, v3& end) {
    const float bail2 = sc.p[RM_P_BULB_BAILOUT] * sc.p[RM_P_BULB_BAILOUT];
    const float rho2 = pow8_rho2(p);
    const float r2 = FM::fma(p.z, p.z, rho2);
    if (ITERS != 8 && (int)sc.p[RM_P_BULB_ITERATIONS] < 1) { d = pow8_distance(0.0f, 1.0f); return false; }  // (far_jump_applies: never with jump)
    if (r2 > bail2) {
      if (jump && far_jump_at(p, dir, left, r2, bail2, end)) return true;
#ifdef RM_LANE_STATS
      lane_stats(0);
#endif
      d = pow8_distance_far(r2);
      return false;
    }
    d = pow8_near<ITERS>(sc, p, rho2, r2, bail2);
    return false;
  }

  // Cost classes.  The evaluation runs 0..`iterations` rounds of z -> z^n + c
  // depending on how close p is to the set: far points bail out at once, points
  // on the surface run them all (about 10x the cost).  eval_cheap is eval() with
  // the round count capped: it returns false -- and no distance -- when p needs
  // more than `cap` rounds; otherwise the same bits as eval().  The wavefront
  // march uses it to keep cheap and expensive rays in separate waves.
  static constexpr bool has_cost_classes = true;
  static RM_DEV bool nonfinite_normal_is_nan(const DevScene&) { return true; }  // |z| is Inf or NaN there: the estimate is Inf or NaN
  static constexpr int cheap_cap = 2;
  template <class M>
  static RM_DEV bool eval_cheap(const DevScene& sc, const SceneLds& lds, v3 p, float& d) {
    const int iterations = (int)sc.p[RM_P_BULB_ITERATIONS];
    if (iterations <= cheap_cap) { d = eval<M>(sc, lds, p); return true; }
    // a capped run that used all its rounds without bailing out is not a finished evaluation
    bool bailed = false;
    if (M::fast && sc.p[RM_P_BULB_POWER] == 8.0f) {
      const float bail2 = sc.p[RM_P_BULB_BAILOUT] * sc.p[RM_P_BULB_BAILOUT];
      v3 z = p;
      float dr = 1.0f, r2 = 0.0f;
      int rounds = 0;
      for (int i = 0; i <= cheap_cap; i++) {
        const float rho2 = pow8_rho2(z);
        r2 = FM::fma(z.z, z.z, rho2);
        if (r2 > bail2) { bailed = true; break; }
        if (i == cheap_cap) break;
        pow8_round(z, dr, p, rho2, r2);
        rounds++;
      }
      if (!bailed) return false;
      d = rounds == 0 ? pow8_distance_far(r2) : pow8_distance(r2, dr);  // as eval_pow8_n ends
      return true;
    }
    const float power = sc.p[RM_P_BULB_POWER], bailout = sc.p[RM_P_BULB_BAILOUT];
    v3 z = p;
    float dr = 1.0f, r = 0.0f;
    for (int i = 0; i <= cheap_cap; i++) {
      r = M::sqrt_of_ordinary(dot<M>(z, z));  // length(z)
      if (r > bailout) { bailed = true; break; }
      if (i == cheap_cap) break;
      generic_round<M>(z, dr, p, r, power);
    }
    if (!bailed) return false;
    d = M::div(0.5f * M::log(r) * r, dr);
    return true;
  }
};

// The Mandelbulb as the headline configuration has it (fast build, power 8, 8 rounds): a kernel of its own
// (rm_kernels.inc launch_pixels), so that the march loop carries neither the generic-power path (acos, atan, pow)
// nor the runtime round count.  Internal to the pixel kernel's dispatch; not a scene kind of the ABI.
#define RM_KIND_BULB8 RM_SCENE_KIND_COUNT
template <>
struct Sdf<RM_KIND_BULB8> : Sdf<RM_SCENE_MANDELBULB> {
  template <class M>
  static RM_DEV float eval(const DevScene& sc, const SceneLds&, v3 p) { return eval_pow8_n<8>(sc, p); }
  static RM_DEV bool far_jump_applies(const DevScene&) { return true; }
  template <class M>
  static RM_DEV bool eval_or_jump(const DevScene& sc, const SceneLds&, v3 p, v3 dir, int left, bool jump, float& d, v3& end) {
    return pow8_eval_or_jump<8>(sc, p, dir, left, jump, d, end);
  }
};

// A primitive table of many rows in full mode (rm_params.hpp rm_table_big): the same program as RM_SCENE_TABLE in a pixel kernel
// that compacts its rays like the Mandelbulb's (8-wave workgroups).  Since the far-field jump an escaping ray leaves its lane
// at once while its neighbours march on -- lanes active fell to 57 % on BASELINE's CSG-64 frames -- and an evaluation of 64
// rows is long enough to pay for the barriers: C4's 1/8 stripes 3.19 -> 2.38 ms, C5's 34.7 -> 31.4, C4's frame 19.2 -> 15.3;
// short tables lose (a single sphere +20 %, a 5-row mixed table +8 %) and keep the plain kernel.  Internal, like RM_KIND_BULB8.
#define RM_KIND_TABLE_BIG (RM_SCENE_KIND_COUNT + 1)
template <>
struct Sdf<RM_KIND_TABLE_BIG> : Sdf<RM_SCENE_TABLE> {};
// ... and the long table of spheres under ONE smooth-union radius, without surfaces (BASELINE's CSG-64), has that kernel to itself
// in the fast build: without the general fold, its per-row dispatch, the row culling and the surface lookups in the same function
// the compiler spills less around the march (C4's frame 13.9 -> 12.9 ms, its stripes 2.39 -> 2.19, C5's 30.8 -> 29.2)
#define RM_KIND_TABLE_SMOOTH (RM_SCENE_KIND_COUNT + 2)
template <>
struct Sdf<RM_KIND_TABLE_SMOOTH> : Sdf<RM_SCENE_TABLE> {
  template <class M>
  static RM_DEV float eval(const DevScene& sc, const SceneLds& lds, v3 p) {
    if constexpr (M::fast) return eval_spheres_one_k(sc, lds, p);
    else return Sdf<RM_SCENE_TABLE>::template eval<M>(sc, lds, p);
  }
};
// ... and a table with RM_PRIM_KIND rows (round 4): the general fold with the kind rows' branch compiled in -- a Mandelbulb's or a
// lattice's evaluator inlined into the row loop -- so that the other tables' kernels do not carry it.  No far-field exits (their
// arguments are about spheres and boxes), no row culling, no ray compaction; the pixel kernel only (rm_api.hip uses_wavefront).
#define RM_KIND_TABLE_KINDS (RM_SCENE_KIND_COUNT + 3)
template <>
struct Sdf<RM_KIND_TABLE_KINDS> : Sdf<RM_SCENE_TABLE> {
  template <class M>
  static RM_DEV float eval(const DevScene& sc, const SceneLds& lds, v3 p) { return Sdf<RM_SCENE_TABLE>::template eval_general<M, true>(sc, lds, p); }
  template <class M>
  static RM_DEV int surface_index(const DevScene& sc, const SceneLds& lds, v3 p) { return Sdf<RM_SCENE_TABLE>::template surface_index<M, true>(sc, lds, p); }
  static RM_DEV bool far_jump_applies(const DevScene&) { return false; }
};
template <int KIND> struct IsTable { static constexpr bool value = KIND == RM_SCENE_TABLE || KIND == RM_KIND_TABLE_BIG || KIND == RM_KIND_TABLE_SMOOTH || KIND == RM_KIND_TABLE_KINDS; };
// the long tables' kernels: where an evaluation is dear enough for the per-march and per-step tests of the far field's finer exits
// (clear_miss, far_shadow_escape) -- on a 5-row table they cost 8 % and save nothing
template <int KIND> struct IsBigTable { static constexpr bool value = KIND == RM_KIND_TABLE_BIG || KIND == RM_KIND_TABLE_SMOOTH; };

// kinds whose fast march may jump an escaping ray to its end state (Sdf<RM_SCENE_MANDELBULB>::far_jump)
template <int KIND> struct FarJump { static constexpr bool value = false; };
template <> struct FarJump<RM_SCENE_MANDELBULB> { static constexpr bool value = true; };
template <> struct FarJump<RM_KIND_BULB8> { static constexpr bool value = true; };
template <> struct FarJump<RM_SCENE_TABLE> { static constexpr bool value = true; };
template <> struct FarJump<RM_KIND_TABLE_BIG> { static constexpr bool value = true; };
template <> struct FarJump<RM_KIND_TABLE_SMOOTH> { static constexpr bool value = true; };
template <> struct FarJump<RM_SCENE_MENGER> { static constexpr bool value = true; };
template <> struct FarJump<RM_SCENE_KIFS_BOX> { static constexpr bool value = true; };
template <> struct FarJump<RM_SCENE_SPHERE_GRID> { static constexpr bool value = true; };

// kinds whose evaluation carries the jump's test itself (Sdf<KIND>::eval_or_jump; the generic Mandelbulb kernel only at power 8:
// far_jump_applies says so, and then eval() is the power-8 evaluation)
template <int KIND> struct FusedJump { static constexpr bool value = false; };
template <> struct FusedJump<RM_KIND_BULB8> { static constexpr bool value = true; };

// kernels that may evaluate the power-8 Mandelbulb on the fast policy start with this (FM::omod_mode)
template <int KIND, bool FAST> RM_DEV void enter_math_mode() {
  if (FAST && (KIND == RM_SCENE_MANDELBULB || KIND == RM_KIND_BULB8 || KIND == RM_KIND_TABLE_KINDS)) FM::omod_mode();
}

// per-level scale factors pow(base, i), i = first .. first + RM_TAB_POW - 1,
// computed once per workgroup with the same pow the per-evaluation GLSL uses
RM_DEV void stage_pow_table(SceneLds& lds, float base, float first) {
  float* t = reinterpret_cast<float*>(lds.rows);
  if (threadIdx.x < RM_TAB_POW) t[threadIdx.x] = PM::pow(base, first + (float)threadIdx.x);
}

// RM_SCENE_SPHERE_GRID: examples/guide.glsl:91-102 == examples/fractal1.glsl:23-34
template <>
struct Sdf<RM_SCENE_SPHERE_GRID> {
  static constexpr bool has_cost_classes = false;  // every evaluation costs the same
  static RM_DEV bool nonfinite_normal_is_nan(const DevScene&) { return false; }  // not shown for this kind: always evaluate
  static RM_DEV void stage(const DevScene& sc, SceneLds& lds) { stage_pow_table(lds, sc.p[RM_P_GRID_SCALE], -1.0f); }
  // the far field, jumped (round 3; fast policy): the estimate is max(|p - centre| - bigSphereSize, carving), i.e. >= |p| - R' with
  // R' = |centre| + bigSphereSize whatever the carving terms do; at the overflow of |p|^2 the sphere's distance is +Inf and
  // max(Inf, x) is +Inf for every x (a NaN is dropped), so the position becomes +-Inf by the signs of the direction, where the
  // grid's mod is NaN, min(NaN, 9999.9) keeps 9999.9 and max(Inf, -9999.9) is +Inf again: a fixed point (rm_api.hip sets far_r2).
  static RM_DEV bool far_jump_applies(const DevScene& sc) { return sc.far_end == 1; }
  static RM_DEV bool far_jump(const DevScene& sc, v3 p, v3 dir, int left, v3& end) { return far_escape(p, dir, left, sc.far_r2, true, end); }
  template <class M>
  static RM_DEV float eval(const DevScene& sc, const SceneLds& lds, v3 p) {
    const float* tab = reinterpret_cast<const float*>(lds.rows);
    const float iters = sc.p[RM_P_GRID_ITERATIONS];
    float min_dist = 9999.9f;
    int k = 0;
    for (float i = -1.0f; i < iters; i += 1.0f, k++) {
      const float sf = k < RM_TAB_POW ? tab[k] : PM::pow(sc.p[RM_P_GRID_SCALE], i);
      const float half = sf / 2.0f, third = sf / 3.0f;
      v3 d = vabs(adds(vmods<M>(adds(p, 0.5f * sf), sf), -half));
      d = adds(d, -third);
      min_dist = gmin(length<M>(d) - 0.21f * sf, min_dist);
    }
    const v3 c = V(sc.p[RM_P_GRID_CENTER], sc.p[RM_P_GRID_CENTER + 1], sc.p[RM_P_GRID_CENTER + 2]);
    return gmax(length<M>(p - c) - sc.p[RM_P_GRID_BIG_SIZE], -min_dist);
  }
};

// RM_SCENE_SPHERE_LATTICE: dist/examples/sphere-grid.glsl:42-49
template <>
struct Sdf<RM_SCENE_SPHERE_LATTICE> {
  static constexpr bool has_cost_classes = false;  // every evaluation costs the same
  static RM_DEV bool nonfinite_normal_is_nan(const DevScene&) { return false; }  // not shown for this kind: always evaluate
  static RM_DEV void stage(const DevScene&, SceneLds&) {}
  template <class M>
  static RM_DEV float eval(const DevScene& sc, const SceneLds&, v3 p) {
    const float period = sc.p[RM_P_LATTICE_PERIOD], half = period * 0.5f;
    v3 rep = adds(vmods<M>(adds(p, half), period), -half);
    return length<M>(rep - V(0.0f, 0.0f, 0.0f)) - sc.p[RM_P_LATTICE_RADIUS];
  }
};

// RM_SCENE_MENGER: examples/menger-sponge.glsl:6-23
template <>
struct Sdf<RM_SCENE_MENGER> {
  static constexpr bool has_cost_classes = false;  // every evaluation costs the same
  static RM_DEV bool nonfinite_normal_is_nan(const DevScene&) { return false; }  // not shown for this kind: always evaluate
  static RM_DEV void stage(const DevScene&, SceneLds& lds) { stage_pow_table(lds, 0.33333333333333f, 1.0f); }
  // ---- the far field of the sponge, jumped (round 3; fast policy) ----------------------------------------------------------
  // Beyond |p| = 4 the distance IS the unit box's: the carving terms are distances to crosses at mod(p, 3 sf) - 1.5 sf, i.e.
  // bounded by 2 plus the rounding of the mod (<= 3e-7 |p|), and max(box, -min(a, b, c)) picks the box, whose distance is
  // >= |p| - 1.8.  So a ray out there that is not moving inward doubles its distance every step: |p|^2 overflows within 64
  // steps (the bound of the tables, R' = 2), the box's length is then +Inf while the carving terms are finite: d = +Inf and
  // the position becomes +-Inf by the signs of the direction.  That is a fixed point: at an infinite coordinate the mod is
  // Inf - Inf = NaN, sdBox's max(q, 0) drops it and the crosses' distances are 0, max(Inf, -0) = Inf again.  Taken for
  // directions without a zero component only (0 x Inf = NaN; such rays march on).  Exact: tested against the stepwise march.
  static RM_DEV bool far_jump_applies(const DevScene&) { return true; }
  static RM_DEV bool far_jump(const DevScene&, v3 p, v3 dir, int left, v3& end) { return far_escape(p, dir, left, 25.0f, true, end); }  // R' = 2.5 >= 1.8
  template <class M>
  static RM_DEV float eval(const DevScene& sc, const SceneLds& lds, v3 p) {
    const float* tab = reinterpret_cast<const float*>(lds.rows);
    const float iters = sc.p[RM_P_MENGER_ITERATIONS];
    float min_dist = sd_box<M>(adds(p, 0.5f), V(0.5f, 0.5f, 0.5f));
    int k = 0;
    for (float i = 1.0f; i < iters; i += 1.0f, k++) {
      const float sf = k < RM_TAB_POW ? tab[k] : PM::pow(0.33333333333333f, i);
      v3 g = adds(vmods<M>(p, sf * 3.0f), -(sf * 1.5f));
      float a = sd_box<M>(g, V(sf * 1.51f, sf * 0.5f, sf * 0.5f));
      float b = sd_box<M>(g, V(sf * 0.5f, sf * 1.51f, sf * 0.5f));
      float c = sd_box<M>(g, V(sf * 0.5f, sf * 0.5f, sf * 1.51f));
      min_dist = gmax(min_dist, -gmin(gmin(a, b), c));
    }
    return min_dist;
  }
};

// the three plane rotations of tree.glsl:24-32, smooth-tree.glsl:45-53,
// rotation-fractal.glsl:21-29 (v.xy *= mat2(c,-s,s,c) -> (x*c + y*-s, x*s + y*c))
struct KifsTrig {
  float c0, s0, c1, s1, c2, s2;
};
RM_DEV KifsTrig kifs_trig(const DevScene& sc) {
  const float* a = &sc.p[RM_P_KIFS_ANGLES];
  return KifsTrig{PM::cos(a[0]), PM::sin(a[0]), PM::cos(a[1]), PM::sin(a[1]), PM::cos(a[2]), PM::sin(a[2])};
}
template <class M>
RM_DEV v3 kifs_rotate(v3 t, const KifsTrig& g) {
  float nx, ny;
  nx = M::fma(t.y, -g.s0, t.x * g.c0); ny = M::fma(t.y, g.c0, t.x * g.s0); t.x = nx; t.y = ny;
  nx = M::fma(t.z, -g.s1, t.y * g.c1); ny = M::fma(t.z, g.c1, t.y * g.s1); t.y = nx; t.z = ny;
  nx = M::fma(t.z, -g.s2, t.x * g.c2); ny = M::fma(t.z, g.c2, t.x * g.s2); t.x = nx; t.z = ny;
  return t;
}

// RM_SCENE_KIFS_TREE: examples/tree.glsl:16-36, examples/smooth-tree.glsl:30-56
template <>
struct Sdf<RM_SCENE_KIFS_TREE> {
  static constexpr bool has_cost_classes = false;  // every evaluation costs the same
  static RM_DEV bool nonfinite_normal_is_nan(const DevScene&) { return false; }  // not shown for this kind: always evaluate
  static RM_DEV void stage(const DevScene& sc, SceneLds& lds) { stage_pow_table(lds, sc.p[RM_P_KIFS_SCALE], 0.0f); }
  template <class M>
  static RM_DEV float eval(const DevScene& sc, const SceneLds& lds, v3 p) {
    const float* tab = reinterpret_cast<const float*>(lds.rows);
    const float iters = sc.p[RM_P_KIFS_ITERATIONS], scale = sc.p[RM_P_KIFS_SCALE], offset = sc.p[RM_P_KIFS_OFFSET];
    const bool smoothen = sc.p[RM_P_KIFS_SMOOTH] == 1.0f;
    if (ExactExits<M>::value && sc.far_end == 3) {  // beyond |p| = 9999 + R' the estimate is its starting value (rm_api.hip kifs_far_field): the same bits
      const float r2 = FM::fma(p.z, p.z, FM::fma(p.y, p.y, p.x * p.x));
      if (r2 > sc.far_r2 && r2 < 1e14f) return 9999.0f;
    }
    const KifsTrig g = kifs_trig(sc);
    v3 t = p;
    float min_dist = 9999.0f;
    int k = 0;
    for (float i = 0.0f; i < iters; i += 1.0f, k++) {
      const float csf = k < RM_TAB_POW ? tab[k] : PM::pow(scale, i);
      const float box = sd_box<M>(t * csf, V(1.0f * csf, 0.1f * csf, 0.1f * csf));
      min_dist = smoothen ? op_smooth_union<M>(min_dist, box, csf * 0.25f) : gmin(min_dist, box);
      t = V(M::div(t.x, scale), M::div(t.y, scale), M::div(t.z, scale));
      t = vabs(t) - V(1.0f * offset, 0.1f * offset, 0.1f * offset);
      t = kifs_rotate<M>(t, g);
    }
    return min_dist;
  }
};

// RM_SCENE_KIFS_BOX: examples/rotation-fractal.glsl:16-35
template <>
struct Sdf<RM_SCENE_KIFS_BOX> {
  static constexpr bool has_cost_classes = false;  // every evaluation costs the same
  // the far field, jumped (round 3; fast policy; rm_api.hip kifs_far_field has the argument and sets far_end / far_r2)
  static RM_DEV bool far_jump_applies(const DevScene& sc) { return sc.far_end == 1; }
  static RM_DEV bool far_jump(const DevScene& sc, v3 p, v3 dir, int left, v3& end) { return far_escape(p, dir, left, sc.far_r2, true, end); }
  static RM_DEV bool nonfinite_normal_is_nan(const DevScene&) { return false; }  // not shown for this kind: always evaluate
  static RM_DEV void stage(const DevScene&, SceneLds&) {}
  template <class M>
  static RM_DEV float eval(const DevScene& sc, const SceneLds&, v3 p) {
    const float iters = sc.p[RM_P_KIFS_ITERATIONS], scale = sc.p[RM_P_KIFS_SCALE], offset = sc.p[RM_P_KIFS_OFFSET];
    const KifsTrig g = kifs_trig(sc);
    v3 t = p;
    for (float i = 0.0f; i < iters; i += 1.0f) {
      t = V(M::div(t.x, scale), M::div(t.y, scale), M::div(t.z, scale));
      t = vabs(t) - V(offset, offset, offset);
      t = kifs_rotate<M>(t, g);
    }
    const float csf = PM::pow(scale, roundf(iters));
    return sd_box<M>(t * csf, V(csf, csf, csf));
  }
};

// the kinds a table row can evaluate (RM_PRIM_KIND), now that they are defined
template <class M>
RM_DEV float Sdf<RM_SCENE_TABLE>::kind_row(const DevScene& sc, const SceneLds& lds, int kind, v3 at) {
  if (kind == RM_SCENE_SPHERE_LATTICE) return Sdf<RM_SCENE_SPHERE_LATTICE>::template eval<M>(sc, lds, at);
  return Sdf<RM_SCENE_MANDELBULB>::template eval<M>(sc, lds, at);
}

// ---- material functions (Validate.tsx:18-51 with the constants of RmMaterial)

RM_DEV v3 cut_color(const float* col, float cutoff, v3 p) {
  return length<PM>(p) > cutoff ? V(0.0f, 0.0f, 0.0f) : V(col[0], col[1], col[2]);
}
RM_DEV v3 cut_color(v3 col, float cutoff, v3 p) { return length<PM>(p) > cutoff ? V(0.0f, 0.0f, 0.0f) : col; }
// the values of the material functions that can depend on the shape (RmSurface); the cut-offs and the sky are the scene's
struct Surface {
  v3 diffuse, specular, subsurface_color;
  float roughness, subsurface, ior;
};
RM_DEV Surface scene_surface(const DevScene& sc) {
  const RmMaterial& m = sc.mat;
  return Surface{V(m.diffuse[0], m.diffuse[1], m.diffuse[2]), V(m.specular[0], m.specular[1], m.specular[2]),
                 V(m.subsurface_color[0], m.subsurface_color[1], m.subsurface_color[2]), m.roughness, m.subsurface, m.ior};
}
RM_DEV v3 scene_diffuse(const DevScene& sc, v3 p) { return cut_color(sc.mat.diffuse, sc.mat.diffuse_cutoff, p); }
RM_DEV v3 scene_specular(const DevScene& sc, v3 p) { return cut_color(sc.mat.specular, sc.mat.specular_cutoff, p); }
// Validate.tsx:47-51
RM_DEV v3 scene_emission(const DevScene& sc, v3 p) {
  const RmMaterial& m = sc.mat;
  const v3 n = normalize<PM>(p);
  const float comp = m.sky_axis == 0 ? n.x : m.sky_axis == 1 ? n.y : n.z;
  const float d = gmax(comp, m.sky_floor);
  const v3 bright = V(m.sky_color[0] * d * 1.0f, m.sky_color[1] * d * 1.0f, m.sky_color[2] * d * 1.0f);
  return length<PM>(p) > m.sky_radius ? bright * m.sky_scale : V(0.0f, 0.0f, 0.0f);
}

// :148-150.  Without fog (lambda = +0, the default) -log(1 - x) / 0 is +Inf, or NaN when 1 - x rounds to 1
// (-0 / 0): the same values without the logarithm and the division.
RM_DEV float inv_exp_dist(float x, float lambda) {
  if (!RM_GL_STACK && __float_as_uint(lambda) == 0u) return (1.0f - x == 1.0f) ? __builtin_nanf("") : __builtin_inff();
  return -PM::log(1.0f - x) / lambda;
}

// :172-175
RM_DEV float schlick(float cos_theta, float n1, float n2) {
  const float r0 = PM::pow((n1 - n2) / (n1 + n2), 2.0f);
  return r0 + (1.0f - r0) * PM::pow(1.0f - cos_theta, 5.0f);
}

// :61-65
RM_DEV v3 rodrigues(v3 v, v3 k, float theta) {
  const float c = PM::cos(theta);
  const float s = sqrtf(1.0f - c * c);
  return v * c + cross(k, v) * s + k * (dot<PM>(k, v) * (1.0f - c));
}

}  // namespace rm
