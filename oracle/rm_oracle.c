/*
 * rm_oracle.c -- CPU restatement of the reference's per-pixel path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py load this library, and only as the checker
 * or the timed CPU baseline.  libhip_raymarch.so never links or calls it and
 * has no CPU fallback.
 *
 * What it restates: client/public/shader/raymarcher.frag of
 * radian628/raymarching-engine, one call of main() per pixel-sample, in plain
 * scalar fp32 C (compile with -ffp-contract=off so that no FMA is formed).
 * Every function cites the lines it follows.  GLSL built-ins are restated
 * from the GLSL ES 3.00 specification.
 *
 * Parity pin (bit for bit on every golden: tests/test_reference_bits.py, with the GL stack's own transcendentals restated
 * in oracle/ss_math.h): the reference has no tests or golden vectors of its own
 * (SURVEY.md section 4).  This restatement is pinned against outputs of the
 * reference's own GLSL run in this container under software GL (SwiftShader
 * in Kaleido's HeadlessChrome 88; oracle/gl/), committed as tests/golden/
 * with the generating script oracle/gl/gen_golden.py.
 *
 * NaN convention.  GLSL leaves min/max/clamp of a NaN undefined.  Rays that
 * leave the scene overflow to +-Inf/NaN by construction in the reference
 * (castRay has no distance bound, raymarcher.frag:163-170), so the convention
 * decides real pixels.  Two are implemented:
 *   OR_NAN_X86  : max(x,y) = x > y ? x : y, min(x,y) = x < y ? x : y -- what
 *                 SwiftShader does (probed); used against the GL goldens.
 *                 (With it goes SwiftShader's finite log(0): see o_log.)
 *   OR_NAN_IEEE : maxNum/minNum (the non-NaN operand wins) -- what gfx950's
 *                 v_max_f32/v_min_f32 and desktop GPUs do; used as the checker
 *                 for the HIP path.
 * The two differ only where an operand is NaN.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/hip_raymarch.h"

#ifdef OR_COUNT_FLOPS
/* Algorithmic flop count under the convention of SURVEY.md 8(d):
 * add/sub/mul/min/max/abs/compare/select = 1, every transcendental or
 * division = 1, pow = 2.  Counted where the work is done, so data-dependent
 * branches (Mandelbulb bailout) are included. */
static _Thread_local uint64_t or_flops;
/* Pruned count (or_set_count_pruned; tools/count_flops.py --pruned; bench.py's frac_useful): the steps of a march that cannot
 * change its result -- every step after the position has become a bitwise fixed point of p <- p + dir * sdf(p) (settled on a
 * surface, or overflowed to a stable Inf / NaN pattern) -- are not counted: what is left is the arithmetic a march NEEDS, the
 * numerator of a roofline fraction that cannot exceed 1.  The image is the same: only the counter pauses. */
static int or_count_pruned = 0;
static _Thread_local int or_count_paused;
void or_set_count_pruned(int on) { or_count_pruned = on; }
/* ... and the steps of a ray that is certain never to come back: outside a sphere that holds the scene twice over and not moving
 * inward (what the kernels' far-field exits test, without their step budgets: a perfect march stops there whatever it has left).
 * The radius: the Mandelbulb's bailout (no round runs beyond it), a table's shapes + its largest smooth-union radius; other
 * kinds have no such sphere here and are counted to their fixed point only. */
static float count_far_r2(const RmSceneDesc* sc);
#define FL(n) (or_flops += or_count_paused ? 0u : (uint64_t)(n))
#define FL_SETTLED(a, b) do { if (or_count_pruned && memcmp(&(a), &(b), sizeof(a)) == 0) or_count_paused = 1; } while (0)
#define FL_ESCAPING(p, dir, far_r2) do { if (or_count_pruned && (far_r2) > 0.0f && (p).x * (p).x + (p).y * (p).y + (p).z * (p).z > (far_r2) && \
                                             (p).x * (dir).x + (p).y * (dir).y + (p).z * (dir).z >= 0.0f) or_count_paused = 1; } while (0)
#define FL_RESUME() (or_count_paused = 0)
#else
#define FL_SETTLED(a, b) ((void)(a), (void)(b))
#define FL_ESCAPING(p, dir, far_r2) ((void)0)
#define FL_RESUME() ((void)0)
#define FL(n) ((void)0)
#endif

enum { OR_NAN_X86 = 0, OR_NAN_IEEE = 1 };
static int or_nan_mode = OR_NAN_IEEE;

void or_set_nan_mode(int mode) { or_nan_mode = mode; }

/* tan convention.  The reference's RNG is fract(tan(big)*x) (raymarcher.frag:46-49) and
 * GL implementations disagree on tan of hundreds of radians, so no two
 * platforms share a random stream.  OR_TAN_LIBM is "what the text says";
 * OR_TAN_PORTABLE is one fixed sequence of IEEE operations (3-term Cody-Waite
 * reduction by pi/2, minimax sin/cos polynomials, relative error <= 1.5e-7 on
 * [0, 870]) that SwiftShader, this file and the HIP kernel evaluate to the
 * same bits; the whole-image goldens are rendered from the reference's text
 * with its tan() routed to the same sequence (oracle/gl/glref.py). */
enum { OR_TAN_LIBM = 0, OR_TAN_PORTABLE = 1, OR_TAN_SWIFTSHADER = 2 }; /* 2: sin / cos of oracle/ss_math.h -- the GL stack's OWN tan, for goldens rendered from the unmodified text */
static int or_tan_mode = OR_TAN_PORTABLE;
void or_set_tan_mode(int mode) { or_tan_mode = mode; }

/* sin, cos, log, exp, pow, acos, atan2.  GLSL leaves their precision to the implementation, so there are no reference
 * bits to match; what matters is that the checker and the HIP parity build compute the SAME function.
 * OR_MATH_PORTABLE (default): oracle/pm_math.h -- fixed sequences of IEEE binary32 operations (round 5; double-precision series
 * until round 4), within 2-3 ulp, the same text the HIP kernels compile: bit-identical on both sides.
 * OR_MATH_LIBM: the C library's float functions, as this file used them until round 2 (kept to show that nothing hangs
 * on the choice: tests/test_oracle_golden.py renders the goldens' cases both ways).
 * OR_MATH_SWIFTSHADER: oracle/ss_math.h -- the approximations of the GL stack the goldens were rendered with, bit for bit
 * (pinned by tests/golden/swiftshader_math.npz).  With it "the reference under software GL" has ONE value on every scene,
 * transcendental or not, and the goldens are compared without a tolerance for the functions' last bits. */
enum { OR_MATH_LIBM = 0, OR_MATH_PORTABLE = 1, OR_MATH_SWIFTSHADER = 2 };
static int or_math_mode = OR_MATH_PORTABLE;
void or_set_math_mode(int mode) { or_math_mode = mode; }
/* Sensitivity probe for the tests (0 = off, the default): round every sin / cos / log / exp / pow / acos result to `bits`
 * significant bits.  tests/test_oracle_golden.py uses it to show that what separates this file from the reference's GLSL
 * under SwiftShader on glossy random materials is what ~20-bit transcendentals alone produce, case by case. */
static int or_math_round_bits = 0;
void or_set_math_round_bits(int bits) { or_math_round_bits = bits; }
static inline float o_rounded(float v) {
  if (or_math_round_bits <= 0 || or_math_round_bits >= 24 || !(v == v) || isinf(v)) return v;
  int e;
  float m = frexpf(v, &e), sc = ldexpf(1.0f, or_math_round_bits);
  return ldexpf(rintf(m * sc) / sc, e);
}

#define PM_FN static inline
#define PM_SQRTF(x) sqrtf(x)
#define PM_DIV_ORDINARY(a, b) ((a) / (b))
#define PM_FMAF(a, b, c) fmaf((a), (b), (c)) /* IEEE fusedMultiplyAdd: one rounding, in hardware (-mfma) or in libm, the same bits */
static inline unsigned int PM_F2U(float x) { unsigned int u; memcpy(&u, &x, 4); return u; }
static inline float PM_U2F(unsigned int u) { float x; memcpy(&x, &u, 4); return x; }
#include "pm_math.h"
#define SS_FN static inline
#define SS_F2U(x) PM_F2U(x)
#define SS_U2F(u) PM_U2F(u)
#include "ss_math.h"

static inline float o_sin(float x) { return or_math_mode == OR_MATH_SWIFTSHADER ? ss_sin(x) : o_rounded(or_math_mode == OR_MATH_LIBM ? sinf(x) : pm_sin(x)); }
static inline float o_cos(float x) { return or_math_mode == OR_MATH_SWIFTSHADER ? ss_cos(x) : o_rounded(or_math_mode == OR_MATH_LIBM ? cosf(x) : pm_cos(x)); }
/* log(0): GLSL leaves it undefined.  IEEE (and gfx950's v_log_f32, and desktop GPUs) give -Inf; SwiftShader gives
 * -127 ln 2 (probed: log(+-0) = log(1e-45) = 0xc2b00f34), a FINITE number -- and the reference's Box-Muller draw takes
 * log(u1) of a gold_noise value that is exactly 0 about once in a thousand draws (the noise is fract() of a huge product:
 * coarse dyadic values), so under SwiftShader such a sample continues in a finite direction where IEEE arithmetic makes
 * the direction NaN.  Like the min / max forms this belongs to the convention the GL goldens were rendered under. */
static inline float o_log(float x) {
  if (or_math_mode == OR_MATH_SWIFTSHADER) return ss_log(x);
  if (or_nan_mode == OR_NAN_X86 && x == 0.0f) return -88.02969360351562f;
  return o_rounded(or_math_mode == OR_MATH_LIBM ? logf(x) : pm_log(x));
}
static inline float o_exp(float x) { return or_math_mode == OR_MATH_SWIFTSHADER ? ss_exp(x) : o_rounded(or_math_mode == OR_MATH_LIBM ? expf(x) : pm_exp(x)); }
static inline float o_pow(float x, float y) {
  if (or_math_mode == OR_MATH_SWIFTSHADER) return ss_pow(x, y); /* also for y = 2: exp2(2 log2 x), not a product (misc_schlick golden) */
  float v = or_math_mode == OR_MATH_LIBM ? powf(x, y) : pm_pow(x, y);
  return y == 2.0f ? v : o_rounded(v); /* pow(x, 2.0) is a product in every implementation met so far */
}
static inline float o_acos(float x) { return or_math_mode == OR_MATH_SWIFTSHADER ? ss_acos(x) : o_rounded(or_math_mode == OR_MATH_LIBM ? acosf(x) : pm_acos(x)); }
static inline float o_atan2(float y, float x) {
  return or_math_mode == OR_MATH_SWIFTSHADER ? ss_atan2(y, x) : or_math_mode == OR_MATH_LIBM ? atan2f(y, x) : pm_atan2(y, x);
}
/* the functions of ss_math.h on arrays, for the test that pins them: which = 0 log2, 1 log, 2 exp2, 3 exp, 4 sin, 5 cos,
 * 6 pow(a, b), 7 acos, 8 atan(a, b), 9 atan, 10 asin, 11 tan */
void or_ss_math(int which, const float* a, const float* b, int n, float* out) {
  for (int i = 0; i < n; i++) {
    const float x = a[i], y = b ? b[i] : 0.0f;
    out[i] = which == 0 ? ss_log2(x) : which == 1 ? ss_log(x) : which == 2 ? ss_exp2(x) : which == 3 ? ss_exp(x) : which == 4 ? ss_sin(x)
           : which == 5 ? ss_cos(x) : which == 6 ? ss_pow(x, y) : which == 7 ? ss_acos(x) : which == 8 ? ss_atan2(x, y) : which == 9 ? ss_atan(x)
           : which == 10 ? ss_asin(x) : ss_tan(x);
  }
}

static float or_tan(float x) {
  if (or_tan_mode == OR_TAN_LIBM) return tanf(x);
  if (or_tan_mode == OR_TAN_SWIFTSHADER) return ss_tan(x);
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
  return (odd > 0.5f) ? (-c / s) : (s / c);
}

/* The transcendentals of the parity arithmetic on arrays, numbered like include/hip_raymarch.h RM_MATH_* (rm_probe_math is the
 * HIP side): in the current or_set_math mode -- the fp32 sequences of pm_math.h, or the GL stack's ss_math.h.  The shared
 * forms (8-11) are written here as the two separate calls whose bits they must have. */
void or_math(int which, const float* a, const float* b, int n, float* out) {
  for (int i = 0; i < n; i++) {
    const float x = a[i], y = b ? b[i] : 0.0f;
    float r = 0.0f;
    switch (which) {
      case 0: case 10: r = o_sin(x); break;
      case 1: case 11: r = o_cos(x); break;
      case 2: r = o_log(x); break;
      case 3: r = o_exp(x); break;
      case 4: case 9: r = o_pow(or_math_mode == OR_MATH_SWIFTSHADER ? x : fabsf(x), y); break; /* gl_pow */
      case 5: r = o_acos(x); break;
      case 6: r = o_atan2(x, y); break;
      case 7: r = or_tan(x); break;
      case 8: r = o_pow(or_math_mode == OR_MATH_SWIFTSHADER ? x : fabsf(x), y - 1.0f); break;
      case 12: r = sqrtf(x); break;
      case 13: r = x / y; break;
      default: break;
    }
    out[i] = r;
  }
}

/* ---- GLSL built-ins (GLSL ES 3.00 section 8) ---------------------------- */

typedef struct { float x, y, z; } v3;

/* OR_NAN_IEEE: IEEE 754-2019 maximumNumber / minimumNumber as v_max_f32 / v_min_f32 compute them -- the operand that
 * is not NaN wins, and +0 is greater than -0.  (C's fmaxf / fminf leave the zeros' order open and glibc returns its second
 * operand on a tie: the random probes of the GPU tests found sdf(p) = -0 here against +0 there at non-finite points.) */
static inline float gl_max(float x, float y) {
  FL(1);
  if (or_nan_mode == OR_NAN_IEEE) {
    if (x != x) return y;
    if (y != y) return x;
    if (x == y) { uint32_t a, b; memcpy(&a, &x, 4); memcpy(&b, &y, 4); a &= b; memcpy(&x, &a, 4); return x; }  /* +0 unless both are -0 */
    return x > y ? x : y;
  }
  return x > y ? x : y;
}
static inline float gl_min(float x, float y) {
  FL(1);
  if (or_nan_mode == OR_NAN_IEEE) {
    if (x != x) return y;
    if (y != y) return x;
    if (x == y) { uint32_t a, b; memcpy(&a, &x, 4); memcpy(&b, &y, 4); a |= b; memcpy(&x, &a, 4); return x; }  /* -0 if either is */
    return x < y ? x : y;
  }
  return x < y ? x : y;
}
static inline float gl_clamp(float x, float lo, float hi) { return gl_min(gl_max(x, lo), hi); }
/* fract(x) = x - floor(x) (GLSL); SwiftShader clamps the difference below 1 with an x86 min -- min(x - floor(x), 0x3F7FFFFF),
 * second operand for a NaN -- so fract(+-Inf) = fract(NaN) = 0.99999994 there (probed) where IEEE arithmetic gives NaN, and a
 * tiny negative x gives 0.99999994 instead of 1.  Part of the GL comparison convention like the min / max forms; it shows
 * once the random stream runs on the GL stack's own tan, whose cosine can be exactly 0. */
static inline float gl_fract(float x) {
  FL(2);
  const float r = x - floorf(x);
  if (or_nan_mode == OR_NAN_X86) return r < 0.99999994f ? r : 0.99999994f;
  return r;
}
static inline float gl_mod(float x, float y) { FL(4); return x - y * floorf(x / y); }
static inline float gl_sign(float x) { FL(1); return (float)((x > 0.0f) - (x < 0.0f)); }
/* the specification writes x*(1-a)+y*a; implementations evaluate the lerp form
 * below (SwiftShader: bit-exact on 16k smooth-union folds, oracle/gl/gen_golden.py) */
static inline float gl_mix(float x, float y, float a) { FL(3); return x + a * (y - x); }
/* pow(x,y) for x < 0 is undefined in GLSL; SwiftShader evaluates it on |x| (probed) */
/* pow on |x| (GLSL leaves a negative base undefined); the GL stack's own pow takes x as it comes -- its logarithm drops the
 * sign bit itself, so only pow(-Inf, y) differs (rm_probe_math asks) -- and the kernels' GL-stack build does the same */
static inline float gl_pow(float x, float y) { FL(2); return o_pow(or_math_mode == OR_MATH_SWIFTSHADER ? x : fabsf(x), y); }

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { FL(3); return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { FL(3); return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { FL(3); return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { FL(3); return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vadds(v3 a, float s) { FL(3); return V(a.x + s, a.y + s, a.z + s); }
static inline float vdot(v3 a, v3 b) { FL(5); return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float vlength(v3 a) { FL(1); return sqrtf(vdot(a, a)); }
static inline float vdistance(v3 a, v3 b) { return vlength(vsub(a, b)); }
/* v * (1/length(v)): the form SwiftShader evaluates (bit-exact on 16k vectors) */
static inline v3 vnormalize(v3 a) { float inv = 1.0f / vlength(a); FL(4); return V(a.x * inv, a.y * inv, a.z * inv); }
static inline v3 vcross(v3 a, v3 b) {
  FL(9);
  return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline v3 vreflect(v3 i, v3 n) { float d = vdot(n, i); FL(1); return vsub(i, vscale(n, 2.0f * d)); }
static inline v3 vabs(v3 a) { FL(3); return V(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline v3 vmaxs(v3 a, float s) { return V(gl_max(a.x, s), gl_max(a.y, s), gl_max(a.z, s)); }
static inline v3 vmods(v3 a, float s) { return V(gl_mod(a.x, s), gl_mod(a.y, s), gl_mod(a.z, s)); }
static inline int v_any_inf(v3 a) { return isinf(a.x) || isinf(a.y) || isinf(a.z); }
static inline int v_any_nan(v3 a) { return isnan(a.x) || isnan(a.y) || isnan(a.z); }
/* mat4 * vec4(v, 0).xyz, column-major (raymarcher.frag:190,195,196,199) */
static inline v3 mat_rotate(const float* m, v3 v) {
  FL(15);
  return V(m[0] * v.x + m[4] * v.y + m[8] * v.z,
           m[1] * v.x + m[5] * v.y + m[9] * v.z,
           m[2] * v.x + m[6] * v.y + m[10] * v.z);
}

/* ---- per-invocation state ---------------------------------------------- */

typedef struct {
  const RmSceneDesc* scene;
  const RmUniforms* u;
  float tcx, tcy; /* texcoord (raymarcher.vert:8-11) */
  float seed;     /* raymarcher.frag:78 */
} Inv;

/* ---- RNG: raymarcher.frag:44-49, 78-105 --------------------------------- */

static const float OR_PHI = 1.61803398874989484820459f; /* :44 */
static const float OR_PI = 3.141592f;                    /* :79 (truncated in the reference) */

/* :46-49 */
static float gold_noise(float x, float y, float seed) {
  float dx = x * OR_PHI - x, dy = y * OR_PHI - y;
  float dist = sqrtf(dx * dx + dy * dy);
  FL(4 + 4 + 2);
  return gl_fract(or_tan(dist * seed) * x);
}

/* :91-94 */
static float uniform_sample(Inv* s) {
  s->seed += 0.131223f;
  FL(4);
  return gold_noise(s->tcx * 1000.0f, s->tcy * 1000.0f, gl_fract(s->u->randNoise[0] + s->seed));
}

/* :80-89 */
static void box_muller(Inv* s, float* ox, float* oy) {
  s->seed += 0.123123213f;
  float u1 = gold_noise(s->tcx * 1000.0f, s->tcy * 1000.0f, gl_fract(s->u->randNoise[0] + s->seed));
  s->seed += 0.123123213f;
  float u2 = gold_noise(s->tcx * 1000.0f, s->tcy * 1000.0f, gl_fract(s->u->randNoise[1] + s->seed));
  float two_pi_u2 = 2.0f * OR_PI * u2;
  float r = sqrtf(-2.0f * o_log(u1));
  FL(8 + 2 + 3 + 4);
  *ox = r * o_cos(two_pi_u2);
  *oy = r * o_sin(two_pi_u2);
}

/* :96-101 */
static v3 sphere_sample(Inv* s) {
  float ax, ay, bx, by;
  box_muller(s, &ax, &ay);
  box_muller(s, &bx, &by);
  return vnormalize(V(ax, ay, bx));
}

/* ---- scene distance functions ------------------------------------------ */

/* raymarcher.frag:74-76 */
static float sdf_sphere(v3 p, v3 c, float r) { FL(1); return vdistance(p, c) - r; }

/* raymarcher.frag:108-112 */
static float sd_box(v3 p, v3 b) {
  v3 q = vsub(vabs(p), b);
  FL(1);
  return vlength(vmaxs(q, 0.0f)) + gl_min(gl_max(q.x, gl_max(q.y, q.z)), 0.0f);
}

/* examples/smooth-tree.glsl:20-22 */
/* the shapes and operators of ABI 8 (include/hip_raymarch.h), as the composer emits them (scene.py rmTorus, rmCylinder, rmPlane,
 * rmSmoothSubtract, rmSmoothIntersect): the text the goldens were rendered from, statement for statement */
static float length2(float x, float y) { FL(4); return sqrtf(x * x + y * y); }
static float sd_torus(v3 p, float R, float r) { FL(2); return length2(length2(p.x, p.z) - R, p.y) - r; }
static float sd_cylinder(v3 p, float r, float h) {
  FL(2);
  const float dx = length2(p.x, p.z) - r, dy = fabsf(p.y) - h;
  FL(1);
  return gl_min(gl_max(dx, dy), 0.0f) + length2(gl_max(dx, 0.0f), gl_max(dy, 0.0f));
}
static float sd_plane(v3 p, v3 n) { return vdot(p, n); }
static float op_smooth_subtract(float d, float di, float k) {
  FL(4 + 4);
  float h = gl_clamp(0.5f - 0.5f * (d + di) / k, 0.0f, 1.0f);
  return gl_mix(d, -di, h) + k * h * (1.0f - h);
}
static float op_smooth_intersect(float d, float di, float k) {
  FL(4 + 4);
  float h = gl_clamp(0.5f - 0.5f * (d - di) / k, 0.0f, 1.0f);
  return gl_mix(d, di, h) + k * h * (1.0f - h);
}
static float table_shape(const RmSceneDesc* sc, const RmPrim* pr, int prim, v3 q, v3 c);
static float op_smooth_union(float d1, float d2, float k) {
  FL(4 + 4);
  float h = gl_clamp(0.5f + 0.5f * (d2 - d1) / k, 0.0f, 1.0f);
  return gl_mix(d2, d1, h) - k * h * (1.0f - h);
}

/* RM_SCENE_TABLE: the GLSL the composer emits is the same left fold over the shape rows; domain rows
 * (RM_PRIM_REPEAT, RM_PRIM_FOLD: include/hip_raymarch.h) transform the point the following rows see */
static v3 kifs_rotate(v3 t, const float* ang);
static float sdf_mandelbulb(const RmSceneDesc* sc, v3 pos);
static float sdf_sphere_lattice(const RmSceneDesc* sc, v3 p);
static int is_domain_row(int prim) { return prim == RM_PRIM_REPEAT || prim == RM_PRIM_FOLD; }
/* RM_PRIM_KIND (include/hip_raymarch.h): a shape row whose distance term is a scene kind's own estimator at q - center, with the
 * kind's parameters in the scene block -- the composer emits that kind's text as rmKindSdf() and the call (scene.py CsgScene.shape) */
static float sdf_kind_row(const RmSceneDesc* sc, const RmPrim* pr, v3 q) {
  FL(3);
  const v3 at = vsub(q, V(pr->center[0], pr->center[1], pr->center[2]));
  return (int)pr->size[0] == RM_SCENE_SPHERE_LATTICE ? sdf_sphere_lattice(sc, at) : sdf_mandelbulb(sc, at);
}
static float table_shape(const RmSceneDesc* sc, const RmPrim* pr, int prim, v3 q, v3 c) {
  if (prim == RM_PRIM_SPHERE) return sdf_sphere(q, c, pr->size[0]);
  if (prim == RM_PRIM_KIND) return sdf_kind_row(sc, pr, q);
  if (prim == RM_PRIM_TORUS) return sd_torus(vsub(q, c), pr->size[0], pr->size[1]);
  if (prim == RM_PRIM_CYLINDER) return sd_cylinder(vsub(q, c), pr->size[0], pr->size[1]);
  if (prim == RM_PRIM_PLANE) return sd_plane(vsub(q, c), V(pr->size[0], pr->size[1], pr->size[2]));
  return sd_box(vsub(q, c), V(pr->size[0], pr->size[1], pr->size[2]));
}
static float sdf_table(const RmSceneDesc* sc, v3 p) {
  float d = 0.0f, factor = 1.0f;
  int first = 1, domain = 0;
  v3 q = p;
  for (int i = 0; i < sc->nprims; i++) domain |= is_domain_row(sc->prims[i].type & 0xff);
  for (int i = 0; i < sc->nprims; i++) {
    const RmPrim* pr = &sc->prims[i];
    v3 c = V(pr->center[0], pr->center[1], pr->center[2]);
    const int prim = pr->type & 0xff;
    if (prim == RM_PRIM_REPEAT) {  /* dist/examples/sphere-grid.glsl:42-49 */
      FL(3 * 6);
      q = V(gl_mod(q.x + 0.5f * pr->size[0], pr->size[0]) - 0.5f * pr->size[0], gl_mod(q.y + 0.5f * pr->size[1], pr->size[1]) - 0.5f * pr->size[1],
            gl_mod(q.z + 0.5f * pr->size[2], pr->size[2]) - 0.5f * pr->size[2]);
      continue;
    }
    if (prim == RM_PRIM_FOLD) {  /* examples/tree.glsl:24-32 */
      FL(3 + 3 + 1);
      q = V(q.x / pr->k, q.y / pr->k, q.z / pr->k);
      q = vsub(vabs(q), c);
      q = kifs_rotate(q, pr->size);
      factor = factor * pr->k;
      continue;
    }
    float di = table_shape(sc, pr, prim, q, c);
    if (domain) { FL(1); di = di * factor; }
    if (first) { d = di; first = 0; continue; }
    switch ((pr->type >> 8) & 0xff) {
      case RM_OP_UNION: d = gl_min(d, di); break;
      case RM_OP_SMOOTH_UNION: d = op_smooth_union(d, di, pr->k); break;
      case RM_OP_SUBTRACT: FL(1); d = gl_max(d, -di); break;
      case RM_OP_SMOOTH_SUBTRACT: d = op_smooth_subtract(d, di, pr->k); break;
      case RM_OP_SMOOTH_INTERSECT: d = op_smooth_intersect(d, di, pr->k); break;
      default: d = gl_max(d, di); break;
    }
  }
  return d;
}

/* RM_SCENE_MANDELBULB: spherical-coordinate power-n distance estimator; the
 * GLSL text of this scene is authored by this project (composer), the
 * reference has no Mandelbulb.  Line-for-line the same as that text. */
static float sdf_mandelbulb(const RmSceneDesc* sc, v3 pos) {
  const float power = sc->params[RM_P_BULB_POWER];
  const int iterations = (int)sc->params[RM_P_BULB_ITERATIONS];
  const float bailout = sc->params[RM_P_BULB_BAILOUT];
  v3 z = pos;
  float dr = 1.0f, r = 0.0f;
  for (int i = 0; i < iterations; i++) {
    r = vlength(z);
    FL(1);
    if (r > bailout) break;
    float theta = o_acos(z.z / r);
    float phi = o_atan2(z.y, z.x);
    FL(3);
    dr = gl_pow(r, power - 1.0f) * power * dr + 1.0f;
    FL(4);
    float zr = gl_pow(r, power);
    theta = theta * power;
    phi = phi * power;
    FL(2 + 4 + 3);
    z = vscale(V(o_sin(theta) * o_cos(phi), o_sin(phi) * o_sin(theta), o_cos(theta)), zr);
    z = vadd(z, pos);
  }
  FL(4);
  return 0.5f * o_log(r) * r / dr;
}

/* RM_SCENE_SPHERE_GRID: examples/guide.glsl:91-102 == examples/fractal1.glsl:23-34 */
static float sdf_sphere_grid(const RmSceneDesc* sc, v3 p) {
  const float big = sc->params[RM_P_GRID_BIG_SIZE];
  const float iters = sc->params[RM_P_GRID_ITERATIONS];
  const float gs = sc->params[RM_P_GRID_SCALE];
  const v3 centre = V(sc->params[RM_P_GRID_CENTER], sc->params[RM_P_GRID_CENTER + 1], sc->params[RM_P_GRID_CENTER + 2]);
  float min_dist = 9999.9f;
  for (float i = -1.0f; i < iters; i += 1.0f) {
    float sf = gl_pow(gs, i);
    FL(1 + 1 + 1 + 2 + 1);
    v3 d = vsub(vabs(vsub(vmods(vadds(p, 0.5f * sf), sf), V(sf / 2.0f, sf / 2.0f, sf / 2.0f))),
                V(sf / 3.0f, sf / 3.0f, sf / 3.0f));
    float dist = vlength(d) - 0.21f * sf;
    min_dist = gl_min(dist, min_dist);
  }
  FL(2);
  return gl_max(vlength(vsub(p, centre)) - big, -min_dist);
}

/* RM_SCENE_SPHERE_LATTICE: dist/examples/sphere-grid.glsl:42-49 (period 2, radius 0.4 there) */
static float sdf_sphere_lattice(const RmSceneDesc* sc, v3 p) {
  const float period = sc->params[RM_P_LATTICE_PERIOD];
  const float radius = sc->params[RM_P_LATTICE_RADIUS];
  const float half = period * 0.5f;
  v3 rep = vadds(vmods(vadds(p, half), period), -half);
  FL(1);
  return vlength(vsub(rep, V(0, 0, 0))) - radius;
}

/* RM_SCENE_MENGER: examples/menger-sponge.glsl:6-23 */
static float sdf_menger(const RmSceneDesc* sc, v3 p) {
  const float iters = sc->params[RM_P_MENGER_ITERATIONS];
  float min_dist = sd_box(vadds(p, 0.5f), V(0.5f, 0.5f, 0.5f));
  for (float i = 1.0f; i < iters; i += 1.0f) {
    float sf = gl_pow(0.33333333333333f, i);
    FL(2 + 6);
    v3 g = vadds(vmods(p, sf * 3.0f), -(sf * 1.5f));
    float a = sd_box(g, V(sf * 1.51f, sf * 0.5f, sf * 0.5f));
    float b = sd_box(g, V(sf * 0.5f, sf * 1.51f, sf * 0.5f));
    float c = sd_box(g, V(sf * 0.5f, sf * 0.5f, sf * 1.51f));
    FL(1);
    min_dist = gl_max(min_dist, -gl_min(gl_min(a, b), c));
  }
  return min_dist;
}

/* the three plane rotations shared by tree.glsl:24-32, smooth-tree.glsl:45-53,
 * rotation-fractal.glsl:21-29:  v.xy *= mat2(c,-s,s,c) etc.  GLSL's
 * `row_vector *= mat2(a,b,c,d)` (columns (a,b),(c,d)) gives
 * (x*a + y*b, x*c + y*d). */
static v3 kifs_rotate(v3 t, const float* ang) {
  float c, s, nx, ny;
  FL(6 + 18);
  c = o_cos(ang[0]); s = o_sin(ang[0]);
  nx = t.x * c + t.y * -s; ny = t.x * s + t.y * c; t.x = nx; t.y = ny;
  c = o_cos(ang[1]); s = o_sin(ang[1]);
  nx = t.y * c + t.z * -s; ny = t.y * s + t.z * c; t.y = nx; t.z = ny;
  c = o_cos(ang[2]); s = o_sin(ang[2]);
  nx = t.x * c + t.z * -s; ny = t.x * s + t.z * c; t.x = nx; t.z = ny;
  return t;
}

/* RM_SCENE_KIFS_TREE: examples/tree.glsl:16-36 (smoothen == 0),
 * examples/smooth-tree.glsl:30-56 (smoothen == 1) */
static float sdf_kifs_tree(const RmSceneDesc* sc, v3 p) {
  const float iters = sc->params[RM_P_KIFS_ITERATIONS];
  const float scale = sc->params[RM_P_KIFS_SCALE];
  const float* ang = &sc->params[RM_P_KIFS_ANGLES];
  const float offset = sc->params[RM_P_KIFS_OFFSET];
  const int smoothen = sc->params[RM_P_KIFS_SMOOTH] == 1.0f;
  v3 t = p;
  float min_dist = 9999.0f;
  for (float i = 0.0f; i < iters; i += 1.0f) {
    float csf = gl_pow(scale, i);
    v3 t2 = vscale(t, csf);
    FL(3);
    float box = sd_box(t2, V(1.0f * csf, 0.1f * csf, 0.1f * csf));
    if (smoothen) { FL(1); min_dist = op_smooth_union(min_dist, box, csf * 0.25f); }
    else min_dist = gl_min(min_dist, box);
    FL(3 + 3);
    t = V(t.x / scale, t.y / scale, t.z / scale);
    t = vsub(vabs(t), V(1.0f * offset, 0.1f * offset, 0.1f * offset));
    t = kifs_rotate(t, ang);
  }
  return min_dist;
}

/* RM_SCENE_KIFS_BOX: examples/rotation-fractal.glsl:16-35 */
static float sdf_kifs_box(const RmSceneDesc* sc, v3 p) {
  const float iters = sc->params[RM_P_KIFS_ITERATIONS];
  const float scale = sc->params[RM_P_KIFS_SCALE];
  const float* ang = &sc->params[RM_P_KIFS_ANGLES];
  const float offset = sc->params[RM_P_KIFS_OFFSET];
  v3 t = p;
  for (float i = 0.0f; i < iters; i += 1.0f) {
    FL(3);
    t = V(t.x / scale, t.y / scale, t.z / scale);
    t = vsub(vabs(t), V(offset, offset, offset));
    t = kifs_rotate(t, ang);
  }
  float csf = gl_pow(scale, roundf(iters));
  t = vscale(t, csf);
  return sd_box(t, V(csf, csf, csf));
}

static float scene_sdf(const RmSceneDesc* sc, v3 p) {
  switch (sc->kind) {
    case RM_SCENE_TABLE: return sdf_table(sc, p);
    case RM_SCENE_MANDELBULB: return sdf_mandelbulb(sc, p);
    case RM_SCENE_SPHERE_GRID: return sdf_sphere_grid(sc, p);
    case RM_SCENE_SPHERE_LATTICE: return sdf_sphere_lattice(sc, p);
    case RM_SCENE_MENGER: return sdf_menger(sc, p);
    case RM_SCENE_KIFS_TREE: return sdf_kifs_tree(sc, p);
    case RM_SCENE_KIFS_BOX: return sdf_kifs_box(sc, p);
    default: return 0.0f;
  }
}

#ifdef OR_COUNT_FLOPS
static float count_far_r2(const RmSceneDesc* sc) {
  if (sc->kind == RM_SCENE_MANDELBULB) {
    const float b = sc->params[RM_P_BULB_BAILOUT];
    return b * b > 4.0f ? b * b : 4.0f;
  }
  if (sc->kind != RM_SCENE_TABLE) return 0.0f;
  double reach = 0.0, kmax = 0.0;
  for (int i = 0; i < sc->nprims; i++) {
    const RmPrim* q = &sc->prims[i];
    const int type = q->type & 0xff;
    if (type != RM_PRIM_SPHERE && type != RM_PRIM_BOX) return 0.0f; /* domain rows, kind rows: no bound claimed */
    const double e = type == RM_PRIM_SPHERE ? fabs((double)q->size[0]) : sqrt((double)q->size[0] * q->size[0] + (double)q->size[1] * q->size[1] + (double)q->size[2] * q->size[2]);
    const double c = sqrt((double)q->center[0] * q->center[0] + (double)q->center[1] * q->center[1] + (double)q->center[2] * q->center[2]);
    if (c + e > reach) reach = c + e;
    if (((q->type >> 8) & 0xff) == RM_OP_SMOOTH_UNION && q->k > kmax) kmax = q->k;
  }
  const double r = 2.0 * (reach + 1.01 * kmax) + 1.0; /* the kernels' (2 R' + 1) */
  return (float)(r * r);
}
#endif

/* ---- material functions: Validate.tsx:18-51 with the constants in RmMaterial */

/* Position-dependent materials of a composed scene (include/hip_raymarch.h RmSurface; the reference's contract is seven
 * functions of position, Validate.tsx:18-51, examples/guide.glsl:51-88): the values of the shape row whose distance term at
 * p is the smallest -- the loop the composer emits as rmSurfaceIndex(), statement for statement: the terms of sdf()'s fold
 * before their operators, `if (di < best)`, so the earliest row wins a tie and a NaN term never wins.  surface 0 = the
 * scene's material block. */
static int table_has_surfaces(const RmSceneDesc* sc) {
  if (sc->kind != RM_SCENE_TABLE) return 0;
  for (int i = 0; i < sc->nprims; i++)
    if ((sc->prims[i].type >> 16) & 0xff) return 1;
  return 0;
}
static int surface_index(const RmSceneDesc* sc, v3 p) {
  float best = 0.0f, factor = 1.0f;
  int first = 1, domain = 0, surface = 0;
  v3 q = p;
  for (int i = 0; i < sc->nprims; i++) domain |= is_domain_row(sc->prims[i].type & 0xff);
  for (int i = 0; i < sc->nprims; i++) {
    const RmPrim* pr = &sc->prims[i];
    v3 c = V(pr->center[0], pr->center[1], pr->center[2]);
    const int prim = pr->type & 0xff;
    if (prim == RM_PRIM_REPEAT) {
      q = V(gl_mod(q.x + 0.5f * pr->size[0], pr->size[0]) - 0.5f * pr->size[0], gl_mod(q.y + 0.5f * pr->size[1], pr->size[1]) - 0.5f * pr->size[1],
            gl_mod(q.z + 0.5f * pr->size[2], pr->size[2]) - 0.5f * pr->size[2]);
      continue;
    }
    if (prim == RM_PRIM_FOLD) {
      q = V(q.x / pr->k, q.y / pr->k, q.z / pr->k);
      q = vsub(vabs(q), c);
      q = kifs_rotate(q, pr->size);
      factor = factor * pr->k;
      continue;
    }
    float di = table_shape(sc, pr, prim, q, c);
    if (domain) di = di * factor;
    if (first || di < best) { best = di; surface = (pr->type >> 16) & 0xff; }
    first = 0;
  }
  return surface;
}
typedef struct OSurface {
  float diffuse[3], specular[3], subsurface_color[3], roughness, subsurface, ior;
} OSurface;
static OSurface surface_at(const RmSceneDesc* sc, v3 p) {
  OSurface o;
  const int k = table_has_surfaces(sc) ? surface_index(sc, p) : 0;
  if (k > 0 && k <= sc->nsurfaces && sc->surfaces) {
    const RmSurface* f = &sc->surfaces[k - 1];
    memcpy(o.diffuse, f->diffuse, sizeof o.diffuse); memcpy(o.specular, f->specular, sizeof o.specular);
    memcpy(o.subsurface_color, f->subsurface_color, sizeof o.subsurface_color);
    o.roughness = f->roughness; o.subsurface = f->subsurface; o.ior = f->ior;
  } else {
    const RmMaterial* m = &sc->material;
    memcpy(o.diffuse, m->diffuse, sizeof o.diffuse); memcpy(o.specular, m->specular, sizeof o.specular);
    memcpy(o.subsurface_color, m->subsurface_color, sizeof o.subsurface_color);
    o.roughness = m->roughness; o.subsurface = m->subsurface; o.ior = m->ior;
  }
  return o;
}

static v3 cut_color(const float* col, float cutoff, v3 p) {
  FL(1);
  if (vlength(p) > cutoff) return V(0, 0, 0);
  return V(col[0], col[1], col[2]);
}
static v3 scene_diffuse(const RmSceneDesc* sc, v3 p) { const OSurface f = surface_at(sc, p); return cut_color(f.diffuse, sc->material.diffuse_cutoff, p); }
static v3 scene_specular(const RmSceneDesc* sc, v3 p) { const OSurface f = surface_at(sc, p); return cut_color(f.specular, sc->material.specular_cutoff, p); }
/* Validate.tsx:47-51 */
static v3 scene_emission(const RmSceneDesc* sc, v3 p) {
  const RmMaterial* m = &sc->material;
  v3 n = vnormalize(p);
  float comp = m->sky_axis == 0 ? n.x : m->sky_axis == 1 ? n.y : n.z;
  float d = gl_max(comp, m->sky_floor);
  FL(3 + 3 + 3 + 1);
  v3 bright = V(m->sky_color[0] * d * 1.0f, m->sky_color[1] * d * 1.0f, m->sky_color[2] * d * 1.0f);
  if (vlength(p) > m->sky_radius) return vscale(bright, m->sky_scale);
  return V(0, 0, 0);
}

/* ---- raymarcher.frag:148-175 -------------------------------------------- */

/* :148-150 */
static float inv_exp_dist(float x, float lambda) { FL(4); return -o_log(1.0f - x) / lambda; }

/* :153-160 -- forward differences */
static v3 scene_normal(const RmSceneDesc* sc, v3 p, float delta) {
  float at = scene_sdf(sc, p);
  FL(3 + 3);
  return vnormalize(V(scene_sdf(sc, V(p.x + delta, p.y, p.z)) - at,
                      scene_sdf(sc, V(p.x, p.y + delta, p.z)) - at,
                      scene_sdf(sc, V(p.x, p.y, p.z + delta)) - at));
}

/* :163-170 -- fixed step count, no early exit */
static v3 cast_ray(const RmSceneDesc* sc, v3 p, v3 dir, float steps) {
#ifdef OR_COUNT_FLOPS
  const float far_r2 = or_count_pruned ? count_far_r2(sc) : 0.0f;
#endif
  for (float i = 0.0f; i < steps; i += 1.0f) {
    FL_ESCAPING(p, dir, far_r2);
    float d = scene_sdf(sc, p);
    const v3 before = p;
    p = vadd(p, vscale(dir, d));
    FL_SETTLED(before, p); /* (pruned count: a fixed point -- the remaining steps repeat this one) */
  }
  FL_RESUME();
  return p;
}

/* :172-175 */
static float schlick(float cos_theta, float n1, float n2) {
  FL(3 + 4);
  float r0 = gl_pow((n1 - n2) / (n1 + n2), 2.0f);
  return r0 + (1.0f - r0) * gl_pow(1.0f - cos_theta, 5.0f);
}

/* :61-65 */
static v3 rodrigues(v3 v, v3 k, float theta) {
  float c = o_cos(theta);
  float s = sqrtf(1.0f - c * c);
  FL(1 + 3 + 2);
  return vadd(vadd(vscale(v, c), vscale(vcross(k, v), s)), vscale(k, vdot(k, v) * (1.0f - c)));
}

/* ---- main(): raymarcher.frag:178-388 ------------------------------------ */

typedef struct { float v[4]; } px4;

/* One pixel-sample.  prev_* are the pixel's previous values and are replaced
 * by the new ones (bind prev -> draw -> blit, RenderJobExecutor.tsx:195-326).
 * `jitter` == 0 evaluates the same code with randomDirectionOffset = 0 (used
 * for RNG-free goldens of the preview image). */
static void pixel_main(const RmSceneDesc* sc, const RmUniforms* u, int W, int H, int px, int py,
                       float* color, float* normal_dof, float* albedo_depth) {
  Inv s;
  s.scene = sc;
  s.u = u;
  s.tcx = ((float)px + 0.5f) / (float)W;
  s.tcy = ((float)py + 0.5f) / (float)H;
  s.seed = 0.0f;

  v3 dir = V(0, 0, 0), pos = V(0, 0, 0);
  const v3 cam = V(u->position[0], u->position[1], u->position[2]);
  /* :182-184 */
  float jx = uniform_sample(&s);
  float jy = uniform_sample(&s);
  jx = jx / (float)W * 1.0f;
  jy = jy / (float)H * 1.0f;
  float t2x = s.tcx + jx, t2y = s.tcy + jy;
  float delta_z = 1.0f;
  FL(6);
  if (u->cameraMode == 0) { /* :186-193 */
    v3 dof = vscale(sphere_sample(&s), u->dofAmount);
    pos = vadd(cam, dof);
    float th = or_tan(u->fov / 2.0f);
    float ppx = (t2x * 2.0f - 1.0f) * u->aspect * th;
    float ppy = (t2y * 2.0f - 1.0f) * 1.0f * th;
    FL(2 + 8 + 2);
    v3 not_norm = mat_rotate(u->rotation, V(ppx + jx, ppy + jy, 1.0f));
    v3 goal = vscale(not_norm, u->dofFocalPlaneDistance);
    delta_z = 1.0f / vlength(V(ppx, ppy, 1.0f));
    FL(1);
    dir = vnormalize(vsub(goal, dof));
  } else if (u->cameraMode == 1) { /* :194-196 */
    dir = vnormalize(mat_rotate(u->rotation, V(0, 0, 1.0f)));
    FL(6);
    pos = vadd(cam, mat_rotate(u->rotation, V((t2x - 0.5f) * u->aspect * u->fov, (t2y - 0.5f) * 1.0f * u->fov, 0.0f)));
  } else if (u->cameraMode == 2) { /* :197-205 */
    float ax = (t2x - 0.5f) * (2.0f * OR_PI);
    float ay = (t2y - 0.5f) * OR_PI;
    FL(5 + 6 + 2);
    dir = mat_rotate(u->rotation, V(o_cos(ax) * o_cos(ay), o_sin(ay), o_sin(ax) * o_cos(ay)));
    pos = cam;
  }

  const float* steps_arr = u->raymarchingStepCountsArray;

  if (u->renderMode == 1) { /* preview, :207-244 */
    float steps_taken = 0.0f, depth = 0.0f;
#ifdef OR_COUNT_FLOPS
    const float far_r2 = or_count_pruned ? count_far_r2(sc) : 0.0f;
#endif
    for (float i = 0.0f; i < steps_arr[0]; i += 1.0f) {
      FL_ESCAPING(pos, dir, far_r2);
      float d = scene_sdf(sc, pos);
      const v3 before = pos;
      FL(2);
      if (d < 100000000000.0f) {
        pos = vadd(pos, vscale(dir, d));
        FL(2);
        depth += delta_z * d;
      }
      if (d > 0.0001f) steps_taken = i;
      FL_SETTLED(before, pos);
    }
    FL_RESUME();
    FL(3);
    float shade = 1.0f - steps_taken / steps_arr[0];
    v3 out = vadd(vscale(vadd(scene_diffuse(sc, pos), scene_specular(sc, pos)), shade), scene_emission(sc, pos));
    float col[4];
    if (u->blendMode == 0) { /* :224-228 */
      const float f = u->blendWithPreviousFactor;
      col[0] = gl_mix(out.x, color[0], f);
      col[1] = gl_mix(out.y, color[1], f);
      col[2] = gl_mix(out.z, color[2], f);
      col[3] = gl_mix(1.0f, color[3], f);
    } else { /* :230 */
      FL(8);
      col[0] = color[0] + out.x * u->exposure;
      col[1] = color[1] + out.y * u->exposure;
      col[2] = color[2] + out.z * u->exposure;
      col[3] = color[3] + 0.0f * u->exposure;
    }
    if (u->showDofFocalPlane != 0) { /* :233-239 */
      float focus = fabsf(depth - u->dofFocalPlaneDistance) / depth;
      if (focus < u->dofFocalPlaneDistance * 0.005f) {
        col[0] = 1.0f;
        col[1] = gl_mod(col[1] + 0.5f, 1.0f);
        col[2] = gl_mod(col[2] + 0.5f, 1.0f);
        col[3] = 1.0f;
      }
    }
    memcpy(color, col, sizeof col);
    return;
  }

  /* full path trace, :246-387 */
  v3 albedo = V(1, 1, 1), light = V(0, 0, 0);
  for (float i = 0.0f; i < u->reflections; i += 1.0f) {
    const float steps = steps_arr[(int)i];
    v3 old = pos;
    pos = cast_ray(sc, pos, dir, steps); /* :255 */
    float path_length = inv_exp_dist(uniform_sample(&s), u->fogDensity); /* :257 */
    light = vadd(light, vmul(albedo, scene_emission(sc, pos))); /* :261 */
    v3 normal = scene_normal(sc, pos, 0.00001f); /* :264 */

    /* :266-271 */
    FL(4);
    const OSurface sf = surface_at(sc, pos); /* sceneSubsurfaceScattering / ...Color / IOR / SpecularRoughness(rayPosition), :266, :286, :325, :329 */
    float subsurf = -1.0f / sf.subsurface * o_log(1.0f - uniform_sample(&s));
    v3 sdir = vnormalize(sphere_sample(&s));
    sdir = V(gl_mix(dir.x, sdir.x, 1.0f), gl_mix(dir.y, sdir.y, 1.0f), gl_mix(dir.z, sdir.z, 1.0f));
    sdir = vnormalize(sdir);
    sdir = vscale(sdir, -gl_sign(vdot(sdir, normal)));
    v3 subsurf_pos = vadd(pos, vscale(sdir, subsurf));

    v3 prev_albedo = albedo;
    v3 diffuse_col = scene_diffuse(sc, pos);
    v3 specular_col = scene_specular(sc, pos);
    v3 prev_dir = dir;

    FL(1);
    if (vdistance(old, pos) > path_length || v_any_inf(pos) || v_any_nan(pos)) { /* :278-283 */
      pos = vadd(old, vscale(dir, gl_min(path_length, 1000000.0f)));
      dir = sphere_sample(&s);
      diffuse_col = V(1, 1, 1);
      specular_col = V(1, 1, 1);
      prev_dir = dir;
    } else if (scene_sdf(sc, subsurf_pos) > 0.001f) { /* :284-288 */
      albedo = vmul(albedo, V(sf.subsurface_color[0], sf.subsurface_color[1], sf.subsurface_color[2]));
      pos = subsurf_pos;
      v3 sp = sphere_sample(&s);
      dir = vnormalize(V(gl_mix(dir.x, sp.x, 1.0f), gl_mix(dir.y, sp.y, 1.0f), gl_mix(dir.z, sp.z, 1.0f)));
    } else { /* :290-331 */
      float db = vlength(diffuse_col), sb = vlength(specular_col);
      FL(4);
      float prob = (db > sb) ? (1.0f - sb / db / 2.0f) : (db / sb / 2.0f);
      FL(1);
      if (uniform_sample(&s) < prob) { /* diffuse, :300-319 */
        albedo = vmul(albedo, diffuse_col);
        v3 nd = sphere_sample(&s);
        dir = vscale(nd, gl_sign(vdot(normal, nd)));
      } else { /* specular, :322-330 */
        FL(1);
        float f = gl_clamp(schlick(-vdot(dir, normal), 1.0f, sf.ior), 0.0f, 1.0f);
        albedo = vmul(albedo, vscale(specular_col, f));
        v3 rv = sphere_sample(&s);
        dir = vreflect(dir, normal);
        v3 axis = vnormalize(vcross(rv, dir));
        FL(1);
        dir = rodrigues(dir, axis, sf.roughness * uniform_sample(&s));
      }
    }
    pos = vadd(pos, vscale(dir, 0.001f)); /* :334 */

    if (i == 0.0f) { /* :336-352 */
      float depth = gl_clamp(vdistance(pos, cam), 0.00001f, 100000000.0f);
      if (isinf(normal.x) || isnan(normal.x)) normal.x = 0.0f;
      if (isinf(normal.y) || isnan(normal.y)) normal.y = 0.0f;
      if (isinf(normal.z) || isnan(normal.z)) normal.z = 0.0f;
      FL(4);
      float dof_radius = gl_clamp(u->dofAmount * fabsf(depth - u->dofFocalPlaneDistance) / depth, 0.0f, 1.0f);
      if (isinf(dof_radius) || isnan(dof_radius)) dof_radius = 0.0f;
      if (normal_dof) {
        FL(8);
        normal_dof[0] += normal.x; normal_dof[1] += normal.y; normal_dof[2] += normal.z; normal_dof[3] += dof_radius;
        albedo_depth[0] += albedo.x; albedo_depth[1] += albedo.y; albedo_depth[2] += albedo.z; albedo_depth[3] += depth;
      }
    }

    for (int j = 0; j < u->lightCount; j++) { /* :354-373 */
      v3 lp = V(u->lightPositions[j][0], u->lightPositions[j][1], u->lightPositions[j][2]);
      v3 lc = V(u->lightColors[j][0], u->lightColors[j][1], u->lightColors[j][2]);
      v3 adj = vadd(lp, vscale(sphere_sample(&s), u->lightSizes[j]));
      v3 to_light = vnormalize(vsub(adj, pos));
      v3 result = cast_ray(sc, pos, to_light, steps);
      FL(1);
      if (vdistance(result, adj) >= vdistance(pos, adj)) {
        float r = gl_max(0.0f, vdot(to_light, vreflect(prev_dir, normal)));
        float rough = surface_at(sc, pos).roughness; /* sceneSpecularRoughness(rayPosition) at the MOVED position, :366 */
        float ndl = gl_max(0.0f, vdot(to_light, normal));
        FL(9);
        float denom = gl_pow(r * r * (rough * rough - 1.0f) + 1.0f, 2.0f);
        float pd = 3.14159265f * denom;
        v3 a = vscale(vmul(vmul(prev_albedo, diffuse_col), lc), ndl);
        /* left to right: vec * vec * vec * roughness * roughness / (pi * pow(..)) */
        v3 b = vscale(vscale(vmul(vmul(prev_albedo, specular_col), lc), rough), rough);
        b = V(b.x / pd, b.y / pd, b.z / pd);
        light = vadd(light, vadd(a, b));
      }
    }
  }

  /* :379-387 */
  if (u->blendMode == 0) {
    const float f = u->blendWithPreviousFactor;
    FL(3);
    color[0] = gl_mix(light.x * u->exposure, color[0], f);
    color[1] = gl_mix(light.y * u->exposure, color[1], f);
    color[2] = gl_mix(light.z * u->exposure, color[2], f);
    color[3] = gl_mix(1.0f, color[3], f);
  } else {
    FL(7);
    color[0] = light.x * u->exposure + color[0];
    color[1] = light.y * u->exposure + color[1];
    color[2] = light.z * u->exposure + color[2];
    color[3] = 1.0f + color[3];
  }
}

/* ---- exported entry points (ctypes) ------------------------------------- */

/* Renders one sample of the rows [row_begin, row_begin+row_count) x columns
 * [x0, x0+w) of a W x H image into planes that hold row_count rows (row 0 of
 * the plane = image row row_begin; the bottom row of the image is row 0).
 * normal_dof / albedo_depth may be NULL.  OpenMP over rows when threads > 1.
 * Returns the algorithmic flop count of the call when built with
 * -DOR_COUNT_FLOPS, else 0. */
uint64_t or_render(const RmSceneDesc* sc, const RmUniforms* u, int W, int H, int row_begin, int row_count,
                   int x0, int y0, int w, int h, float* color, float* normal_dof, float* albedo_depth, int threads) {
  uint64_t total = 0;
  int ya = y0 < row_begin ? row_begin : y0;
  int yb = y0 + h > row_begin + row_count ? row_begin + row_count : y0 + h;
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1) reduction(+ : total)
#endif
  for (int y = ya; y < yb; y++) {
#ifdef OR_COUNT_FLOPS
    or_flops = 0;
#endif
    for (int x = x0; x < x0 + w && x < W; x++) {
      size_t o = ((size_t)(y - row_begin) * (size_t)W + (size_t)x) * 4;
      pixel_main(sc, u, W, H, x, y, color + o, normal_dof ? normal_dof + o : NULL, albedo_depth ? albedo_depth + o : NULL);
    }
#ifdef OR_COUNT_FLOPS
    total += or_flops;
#endif
  }
  return total;
}

/* One sample of a LIST of image rows (an unbiased row sample of a large frame: bench.py's cpu_baseline and
 * tools/count_flops.py): row rows[i] of the W x H image goes to row i of the n x W colour plane (zeroed by the
 * caller).  OpenMP over the list.  Returns the flop count like or_render. */
uint64_t or_render_rows(const RmSceneDesc* sc, const RmUniforms* u, int W, int H, const int* rows, int n, float* color, int threads) {
  uint64_t total = 0;
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1) reduction(+ : total)
#endif
  for (int i = 0; i < n; i++) {
#ifdef OR_COUNT_FLOPS
    or_flops = 0;
#endif
    for (int x = 0; x < W; x++) pixel_main(sc, u, W, H, x, rows[i], color + ((size_t)i * (size_t)W + (size_t)x) * 4, NULL, NULL);
#ifdef OR_COUNT_FLOPS
    total += or_flops;
#endif
  }
  return total;
}

void or_eval_sdf(const RmSceneDesc* sc, const float* p, int n, float* out) {
  for (int i = 0; i < n; i++) out[i] = scene_sdf(sc, V(p[3 * i], p[3 * i + 1], p[3 * i + 2]));
}

void or_cast_ray(const RmSceneDesc* sc, const float* in, int n, float steps, float* out) {
  for (int i = 0; i < n; i++) {
    v3 r = cast_ray(sc, V(in[6 * i], in[6 * i + 1], in[6 * i + 2]), V(in[6 * i + 3], in[6 * i + 4], in[6 * i + 5]), steps);
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
  }
}

void or_normal(const RmSceneDesc* sc, const float* p, int n, float delta, float* out) {
  for (int i = 0; i < n; i++) {
    v3 r = scene_normal(sc, V(p[3 * i], p[3 * i + 1], p[3 * i + 2]), delta);
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
  }
}

/* diffuse, specular, emission, (roughness, subsurface, ior) */
void or_material(const RmSceneDesc* sc, const float* p, int n, float* out) {
  for (int i = 0; i < n; i++) {
    v3 q = V(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    v3 d = scene_diffuse(sc, q), s = scene_specular(sc, q), e = scene_emission(sc, q);
    float* o = out + 12 * i;
    o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = s.x; o[4] = s.y; o[5] = s.z;
    o[6] = e.x; o[7] = e.y; o[8] = e.z;
    { const OSurface f = surface_at(sc, q); o[9] = f.roughness; o[10] = f.subsurface; o[11] = f.ior; }
  }
}

/* camera block with randomDirectionOffset = dofOffset = 0: out = H*W*8 */
void or_camera(const RmUniforms* u, int W, int H, float* out) {
  const v3 cam = V(u->position[0], u->position[1], u->position[2]);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      float tcx = ((float)x + 0.5f) / (float)W, tcy = ((float)y + 0.5f) / (float)H;
      v3 dir = V(0, 0, 0), pos = V(0, 0, 0);
      float delta_z = 1.0f;
      if (u->cameraMode == 0) {
        float th = or_tan(u->fov / 2.0f);
        float ppx = (tcx * 2.0f - 1.0f) * u->aspect * th;
        float ppy = (tcy * 2.0f - 1.0f) * 1.0f * th;
        v3 goal = vscale(mat_rotate(u->rotation, V(ppx, ppy, 1.0f)), u->dofFocalPlaneDistance);
        delta_z = 1.0f / vlength(V(ppx, ppy, 1.0f));
        dir = vnormalize(goal);
        pos = cam;
      } else if (u->cameraMode == 1) {
        dir = vnormalize(mat_rotate(u->rotation, V(0, 0, 1.0f)));
        pos = vadd(cam, mat_rotate(u->rotation, V((tcx - 0.5f) * u->aspect * u->fov, (tcy - 0.5f) * 1.0f * u->fov, 0.0f)));
      } else if (u->cameraMode == 2) {
        float ax = (tcx - 0.5f) * (2.0f * OR_PI), ay = (tcy - 0.5f) * OR_PI;
        dir = mat_rotate(u->rotation, V(o_cos(ax) * o_cos(ay), o_sin(ay), o_sin(ax) * o_cos(ay)));
        pos = cam;
      }
      float* o = out + ((size_t)y * W + x) * 8;
      o[0] = pos.x; o[1] = pos.y; o[2] = pos.z; o[3] = delta_z;
      o[4] = dir.x; o[5] = dir.y; o[6] = dir.z; o[7] = 0.0f;
    }
}

/* first `count` uniformSample() values of every pixel: out = H*W*count */
void or_rng(const RmUniforms* u, int W, int H, int count, float* out) {
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      Inv s;
      s.scene = NULL; s.u = u; s.seed = 0.0f;
      s.tcx = ((float)x + 0.5f) / (float)W;
      s.tcy = ((float)y + 0.5f) / (float)H;
      for (int k = 0; k < count; k++) out[((size_t)y * W + x) * count + k] = uniform_sample(&s);
    }
}

/* ---- present pass: client/public/shader/display.frag:16-64 (+ index.tsx:25-59) ----
 * color / normal_dof: H x W x 4 floats, row 0 = bottom row; out: H x W x 4 bytes.
 * Textures are NEAREST + REPEAT (LoadRenderJobContext.tsx:28-37). */
static float gaussian_blur_factor(float ox, float oy, float sigma) {
  const float PI = 3.1415926535f; /* display.frag:14 */
  return 1.0f / (2.0f * PI * sigma * sigma) * o_exp(-((ox * ox + oy * oy) / (2.0f * sigma * sigma)));
}

static int wrap_texel(float coord, int size) { /* NEAREST + REPEAT */
  float t = coord - floorf(coord);
  int i = (int)floorf(t * (float)size);
  return i >= size ? size - 1 : i;
}

void or_present(const float* color, const float* normal_dof, int W, int H, int samples, uint8_t* out) {
  const float brightness = 1.0f / (float)samples; /* index.tsx:39 */
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const float tcx = ((float)x + 0.5f) / (float)W, tcy = ((float)y + 0.5f) / (float)H;
      const float dof = normal_dof ? normal_dof[((size_t)y * W + x) * 4 + 3] * brightness : 0.0f;
      const float kernel = gl_clamp(dof * 200.0f, 0.0f, 16.0f);
      float acc[4] = {0, 0, 0, 0}, count = 0.0f;
      for (float oy = -kernel; oy <= kernel; oy += 1.0f)
        for (float ox = -kernel; ox <= kernel; ox += 1.0f) {
          const float f = gaussian_blur_factor(ox, oy, gl_max(kernel, 1.0f) * 0.3f);
          count += f;
          const int sx = wrap_texel(tcx + ox / (float)W, W), sy = wrap_texel(tcy + oy / (float)H, H);
          const float* c = color + ((size_t)sy * W + sx) * 4;
          for (int k = 0; k < 4; k++) acc[k] += c[k] * f;
        }
      uint8_t* o = out + ((size_t)y * W + x) * 4;
      for (int k = 0; k < 3; k++) {
        float v = gl_pow(acc[k] / count * brightness, 1.0f / 2.2f);
        v = v != v ? 0.0f : (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v)); /* UNORM8 conversion of the canvas */
        if (or_math_mode == OR_MATH_SWIFTSHADER) {
          /* the GL stack the goldens come from goes through 16 bits: c16 = trunc(v * 65535), c8 = (c16 - (c16 >> 8) + 128) >> 8
           * (fitted to its canvases: e.g. v * 255 = 201.5017 is presented as 201) */
          const int c16 = (int)(v * 65535.0f);
          o[k] = (uint8_t)((c16 - (c16 >> 8) + 128) >> 8);
          continue;
        }
        o[k] = (uint8_t)floorf(v * 255.0f + 0.5f);
      }
      o[3] = 255; /* pow(1.0, 1/2.2) */
    }
}

/* Validate.tsx:18-51 */
void or_material_default(RmMaterial* m) {
  memset(m, 0, sizeof *m);
  m->diffuse[0] = m->diffuse[1] = m->diffuse[2] = 0.6f; m->diffuse_cutoff = 35.0f;
  m->specular[0] = m->specular[1] = m->specular[2] = 0.6f; m->specular_cutoff = 35.0f;
  m->roughness = 0.2f; m->subsurface = 11111115.0f;
  m->subsurface_color[0] = m->subsurface_color[1] = m->subsurface_color[2] = 1.0f;
  m->ior = 100.0f;
  m->sky_color[0] = 0.7f; m->sky_color[1] = 0.8f; m->sky_color[2] = 1.0f;
  m->sky_floor = 0.2f; m->sky_scale = 2.0f; m->sky_radius = 36.0f; m->sky_axis = 1;
}

int or_counts_flops(void) {
#ifdef OR_COUNT_FLOPS
  return 1;
#else
  return 0;
#endif
}
