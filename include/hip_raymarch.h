/*
 * hip_raymarch.h -- C ABI of libhip_raymarch.so
 *
 * MI355X (gfx950) implementation of the one hot path of
 * radian628/raymarching-engine: the per-pixel sphere-tracing fragment shader
 *   client/public/shader/raymarcher.frag            (kernel, 387 lines of GLSL)
 * behind the boundary the reference's host crosses for every sample,
 *   client/src/renderer/RenderJobExecutor.tsx:195-326 (bind prev, set uniforms, draw, blit).
 *
 * Every entry point cites the reference interface it replaces.  The ABI is
 * plain C: pointers, sizes and POD structs of fp32/int32 only, no C++ or
 * torch types.  Error convention (the reference returns errors as values,
 * RenderJobExecutor.tsx:56-68,112-136; ShaderCache.tsx:4-11): every call
 * returns an int status, RM_OK == 0, and the text for the last failure is
 * kept per context (rm_last_error).  Nothing throws across the ABI.
 *
 * Threading: one caller per rm_ctx (the reference is single threaded,
 * index.tsx:120); launches are asynchronous on the context's HIP stream and
 * rm_sync() is the completion point.  One rm_ctx per GPU.
 */
#ifndef HIP_RAYMARCH_H
#define HIP_RAYMARCH_H

#include <stddef.h>
#include <stdint.h>

/* The library is built -fvisibility=hidden: exactly the entry points below are exported (tests/test_host_cpu.py compares
 * this header with `nm -D`). */
#ifndef RM_API
#define RM_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define RM_ABI_VERSION 8 /* 8: RM_PRIM_TORUS / _CYLINDER / _PLANE, RM_OP_SMOOTH_SUBTRACT / _INTERSECT, rm_probe_math (additions only); 7: rm_present_sharded_finish / rm_present_sharded take the size of the host buffer (a changed signature), rm_ctx_set_cull_min_pixels, rm_ctx_set_cull_budget, rm_ctx_cull_stats; 6: rm_present_striped_rows, rm_present_sharded_start / _finish, rm_ctx_last_warning, RM_PROBE_CAST_SHADOW, RM_PRIM_KIND (additions only); 3: rm_ctx_set_sample_batch, rm_buffer_*; 4: rm_ctx_set_gl_stack; 5: RmSurface / RmSceneDesc.surfaces (the struct grew), rm_pack_present_rows, rm_present_sharded, rm_ctx_last_pipeline, RM_RENDER_NO_FAR_JUMP, RM_RENDER_NO_CULL (additions only) */

#define RM_MAX_BOUNCES 10 /* raymarchingStepCountsArray[10], raymarcher.frag:31 */
#define RM_MAX_LIGHTS 10  /* lightPositions[10],             raymarcher.frag:37-39 */
#define RM_MAX_PRIMS 256  /* primitive table rows staged in LDS (32 B each) */

/* status codes */
enum {
  RM_OK = 0,
  RM_ERR_INVALID = 1,  /* bad argument / failed validation (the "compile error" of a scene) */
  RM_ERR_DEVICE = 2,   /* HIP runtime error */
  RM_ERR_NO_DEVICE = 3 /* no usable GPU: the library has NO CPU fallback */
};

/* ---- uniforms ---------------------------------------------------------- */

/*
 * Exactly the uniform block of raymarcher.frag:6-42, as assembled per sample
 * at RenderJobExecutor.tsx:212-297.  `rotation` is column-major like
 * gl.uniformMatrix4fv(..., false, m) (RenderJobExecutor.tsx:293-297).
 * The three sampler uniforms are the framebuffer handle; textureSize() is the
 * framebuffer's full image size.  `reflections` is a float in the reference
 * (loop bound, raymarcher.frag:21,252).  `raymarchingSteps` and
 * `indirectLightingRaymarchingSteps` are set by the reference but never read
 * by the shader (raymarcher.frag:22-23): accepted, unused.
 */
typedef struct RmUniforms {
  float blendWithPreviousFactor;
  float randNoise[2];
  float position[3];
  float rotation[16];
  float dofAmount;
  float dofFocalPlaneDistance;
  int32_t cameraMode; /* 0 perspective, 1 orthographic, 2 panoramic (RenderJobExecutor.tsx:228-232) */
  float fov;          /* fov | orthographic size | 1 (RenderJobExecutor.tsx:233-239) */
  float reflections;
  float raymarchingSteps;
  float indirectLightingRaymarchingSteps;
  float aspect;
  float fogDensity;
  float exposure; /* render.exposure / samplesPerPixel (RenderJobExecutor.tsx:254-256) */
  float raymarchingStepCountsArray[RM_MAX_BOUNCES];
  int32_t blendMode;  /* 1 additive, 0 mix (RenderJobExecutor.tsx:258) */
  int32_t renderMode; /* 1 preview, 0 full (RenderJobExecutor.tsx:259) */
  float lightPositions[RM_MAX_LIGHTS][3];
  float lightColors[RM_MAX_LIGHTS][3];
  float lightSizes[RM_MAX_LIGHTS];
  int32_t lightCount;
  int32_t showDofFocalPlane;
} RmUniforms;

/* ---- scene ------------------------------------------------------------- */

/*
 * The reference's scene is GLSL text defining sdf() and up to seven material
 * functions, spliced into the shader and compiled by the GL driver
 * (RenderJobExecutor.tsx:121-127, Validate.tsx:8-57).  A HIP kernel cannot
 * take GLSL, so the scene crosses this ABI as data: a scene kind (the
 * kernel is specialised per kind), its parameters (the scene's "custom
 * uniforms", RenderJobExecutor.tsx:266), an optional primitive table, and a
 * material block holding the constants of the seven material functions.
 */
enum {
  RM_SCENE_TABLE = 0,          /* fold of primitives with CSG operators; sdfSphere raymarcher.frag:74, sdBox :108, opSmoothUnion examples/smooth-tree.glsl:20-22 */
  RM_SCENE_MANDELBULB = 1,     /* power-n Mandelbulb distance estimator (authored by this project; BASELINE.json configs[2]) */
  RM_SCENE_SPHERE_GRID = 2,    /* examples/guide.glsl:91-102, examples/fractal1.glsl:23-34 */
  RM_SCENE_SPHERE_LATTICE = 3, /* dist/examples/sphere-grid.glsl:42-49 */
  RM_SCENE_MENGER = 4,         /* examples/menger-sponge.glsl:6-23 */
  RM_SCENE_KIFS_TREE = 5,      /* examples/tree.glsl:16-36, examples/smooth-tree.glsl:30-56 */
  RM_SCENE_KIFS_BOX = 6,       /* examples/rotation-fractal.glsl:16-35 */
  RM_SCENE_KIND_COUNT = 7
};

/* Shapes, and the domain operators of the composition API: a domain row produces no distance, it
 * transforms the point at which the FOLLOWING rows are evaluated (and the factor their distances are scaled by):
 *   RM_PRIM_REPEAT  q = mod(q + 0.5 * period, period) - 0.5 * period         period = size[0..2] (> 0); the `repeat`
 *                   of the reference's sphere-grid example (dist/examples/sphere-grid.glsl:42-49)
 *   RM_PRIM_FOLD    q = abs(q / scale) - offset;  q = the three plane rotations by angles;  factor *= scale
 *                   scale = k (> 0), offset = center[0..2], angles = size[0..2] (radians): one level of the
 *                   reference's kaleidoscopic folds (examples/tree.glsl:24-32, rotation-fractal.glsl:21-29)
 * A shape row contributes  shape(q) * factor.
 *   RM_PRIM_KIND    (ABI 6) a SHAPE row whose distance term is a scene kind's own estimator, so that a composition like "a
 *                   Mandelbulb cut by a box" or "a lattice of spheres inside a ball" is table data and not a new kind
 *                   (the reference takes any GLSL sdf(): RenderJobExecutor.tsx:121-127):  kind(q - center)  with
 *                   kind = (int)size[0] one of RM_SCENE_MANDELBULB, RM_SCENE_SPHERE_LATTICE and the kind's parameters in
 *                   RmSceneDesc.params (its own slots: one kind per table, any number of rows of it).  Operator, k and
 *                   surface as for a sphere or a box.  A table with such a row renders with the pixel kernel, without the
 *                   far-field exits and the row culling (both are arguments about spheres and boxes).
 *   (ABI 8) three more shapes and the smooth forms of the other two operators -- the vocabulary a scene author composes with next
 *   to sphere / box / smooth union (the reference's helper set is those three: raymarcher.frag:74,:108, smooth-tree.glsl:20-22; any
 *   other shape is GLSL of the author's own, which cannot cross this ABI).  About the row's centre c, axis y:
 *   RM_PRIM_TORUS     length(vec2(length((q - c).xz) - R, (q - c).y)) - r                R = size[0], r = size[1]
 *   RM_PRIM_CYLINDER  capped: dx = length((q - c).xz) - r, dy = |(q - c).y| - h;          r = size[0], h = size[1]
 *                     min(max(dx, dy), 0) + length(max(vec2(dx, dy), 0))
 *   RM_PRIM_PLANE     dot(q - c, n)                                                      n = size[0..2] (the caller's unit normal)
 *   RM_OP_SMOOTH_SUBTRACT   h = clamp(0.5 - 0.5 (d + di) / k, 0, 1);  mix(d, -di, h) + k h (1 - h)      (k > 0)
 *   RM_OP_SMOOTH_INTERSECT  h = clamp(0.5 - 0.5 (d - di) / k, 0, 1);  mix(d,  di, h) + k h (1 - h)
 *   A table with one of these renders with the general fold, step by step: the far-field exits and the row culling are
 *   arguments about spheres, boxes and the four older operators. */
enum { RM_PRIM_SPHERE = 0, RM_PRIM_BOX = 1, RM_PRIM_REPEAT = 2, RM_PRIM_FOLD = 3, RM_PRIM_KIND = 4, RM_PRIM_TORUS = 5, RM_PRIM_CYLINDER = 6, RM_PRIM_PLANE = 7 };
enum { RM_OP_UNION = 0, RM_OP_SMOOTH_UNION = 1, RM_OP_SUBTRACT = 2, RM_OP_INTERSECT = 3, RM_OP_SMOOTH_SUBTRACT = 4, RM_OP_SMOOTH_INTERSECT = 5 };

/* One row of the primitive table, 32 bytes.  The scene distance is the left
 * fold  d = shape[0];  d = op_i(d, shape[i])  over the shape rows, in table order
 * (the operator of the first shape row is ignored); at least one row must be a shape. */
typedef struct RmPrim {
  int32_t type; /* RM_PRIM_* in bits 0..7, RM_OP_* in bits 8..15, surface index of a shape row in bits 16..23 (RmSurface) */
  float k;      /* smooth-union radius */
  float center[3];
  float size[3]; /* sphere: size[0] = radius; box: half extents; repeat: period; fold: angles; kind: size[0] = the RM_SCENE_* kind;
                    torus: R, r; cylinder: r, half height; plane: unit normal */
} RmPrim;

/* parameter slots of RmSceneDesc.params per kind */
enum {
  /* RM_SCENE_MANDELBULB */
  RM_P_BULB_POWER = 0, RM_P_BULB_ITERATIONS = 1, RM_P_BULB_BAILOUT = 2,
  /* RM_SCENE_SPHERE_GRID: bigSphereSize, fractalIterations, gridScaleFactor, bigSphereCenter.xyz */
  RM_P_GRID_BIG_SIZE = 0, RM_P_GRID_ITERATIONS = 1, RM_P_GRID_SCALE = 2, RM_P_GRID_CENTER = 3,
  /* RM_SCENE_SPHERE_LATTICE: period, radius */
  RM_P_LATTICE_PERIOD = 0, RM_P_LATTICE_RADIUS = 1,
  /* RM_SCENE_MENGER: fractalIterations */
  RM_P_MENGER_ITERATIONS = 0,
  /* RM_SCENE_KIFS_TREE / RM_SCENE_KIFS_BOX: fractalIterations, scaleFactor, angles.xyz, offset, smoothen */
  RM_P_KIFS_ITERATIONS = 0, RM_P_KIFS_SCALE = 1, RM_P_KIFS_ANGLES = 2, RM_P_KIFS_OFFSET = 5, RM_P_KIFS_SMOOTH = 6
};

/* Constants of the seven material functions (Validate.tsx:18-51 are the
 * defaults; examples override some).  colour = value inside `*_cutoff`,
 * 0 outside (length(position) > cutoff). */
typedef struct RmMaterial {
  float diffuse[3];
  float diffuse_cutoff;
  float specular[3];
  float specular_cutoff;
  float roughness;
  float subsurface;
  float subsurface_color[3];
  float ior;
  /* emission = sky_color * max(normalize(p)[sky_axis], sky_floor) * sky_scale  where length(p) > sky_radius, else 0 */
  float sky_color[3];
  float sky_floor;
  float sky_scale;
  float sky_radius;
  int32_t sky_axis;
  int32_t reserved;
} RmMaterial;

/* Position-dependent materials of a composed scene.  The reference's contract is seven FUNCTIONS of position
 * (Validate.tsx:18-51; examples/guide.glsl:51-88 writes its own); a primitive table gets them per shape: a shape row
 * names a surface (RmPrim.type bits 16..23: 0 = the scene's RmMaterial block, k = surfaces[k - 1]) and the material
 * functions at `position` return the values of the shape row whose distance there -- the row's own term of the fold,
 * domain rows applied, before its operator -- is the smallest (the earliest such row on a tie, a NaN distance never
 * wins); the cut-off radii and the sky (emission) stay the scene's.  The composer emits the same rule as GLSL
 * (scene.py / js/index.js: rmSurfaceIndex + the seven functions), which is what pins it against the reference's shader. */
#define RM_MAX_SURFACES 15
typedef struct RmSurface {
  float diffuse[3];
  float roughness;
  float specular[3];
  float subsurface;
  float subsurface_color[3];
  float ior;
} RmSurface;

typedef struct RmSceneDesc {
  int32_t kind;
  int32_t nprims;
  const RmPrim* prims; /* host pointer, nprims rows (RM_SCENE_TABLE only) */
  float params[16];
  RmMaterial material;
  int32_t nsurfaces;         /* 0..RM_MAX_SURFACES (RM_SCENE_TABLE only) */
  int32_t reserved;
  const RmSurface* surfaces; /* host pointer, nsurfaces entries, or NULL */
} RmSceneDesc;

/* Fills *m with the reference defaults of Validate.tsx:18-51. */
RM_API void rm_material_default(RmMaterial* m);

/* ---- handles ----------------------------------------------------------- */

typedef struct rm_ctx rm_ctx;
typedef struct rm_scene rm_scene;
typedef struct rm_fb rm_fb;

typedef struct RmRect {
  int32_t x, y, w, h; /* pixels, origin bottom-left like gl.scissor */
} RmRect;

/* render flags */
enum {
  RM_RENDER_STRICT = 0,      /* fixed step counts, IEEE division/sqrt, no contraction: the parity build */
  RM_RENDER_FAST = 1,        /* hardware-rate math in the distance evaluations; same image statistics, not the same bits */
  RM_RENDER_COLOR_ONLY = 2,  /* do not read/write the two G-buffer planes (benchmark "single colour frame" mode) */
  RM_RENDER_MEGAKERNEL = 4,  /* force the one-thread-one-pixel kernel (whole main() per thread) */
  RM_RENDER_WAVEFRONT = 16,  /* NOT in this library since round 5 (RM_ERR_INVALID): the wavefront pipeline -- the same per-pixel program cut
                                at its marches into queue-driven stages -- is the tests' second implementation and is compiled into
                                their cross-check build of these sources only (tests/_xcheck/libhip_raymarch_xcheck.so, -DRM_WITH_WAVEFRONT=1;
                                rm_api.hip says why).  There the flag is a request: tables with surfaces or kind rows take the pixel
                                kernel whatever is asked (rm_ctx_last_pipeline tells).  The product's own cross-check of its exact
                                shortcuts is the stepwise march: RM_RENDER_NO_FAR_JUMP | RM_RENDER_NO_CULL (same bits). */
  RM_RENDER_NO_COST_CLASSES = 8, /* (cross-check build) wavefront march in one pass even for scene kinds whose sdf cost depends on the
                                   point (Mandelbulb); a measurement switch, same results */
  RM_RENDER_NO_OVERLAP = 32, /* this sample runs alone on the context's stream and blends in its own kernel (see
                                rm_ctx_set_samples_in_flight); for timing one launch.  Same results. */
  RM_RENDER_NO_FAR_JUMP = 64, /* RM_RENDER_FAST, bounded scenes (Mandelbulb, tables without domain rows, the sponge, the
                                rotation and sphere-grid fractals): march an escaping ray step by step instead of setting
                                it to the end state its remaining steps are known to reach (castRay, raymarcher.frag:163-170,
                                has no distance bound: such a ray overflows to a fixed +-Inf / NaN pattern) -- also a ray that
                                is certain to miss the scene, or every shape of a long table -- and a long table's shadow ray
                                that escapes with too few steps left for the overflow is marched to the end instead of being
                                stopped once its comparison (raymarcher.frag:362-363) is certain.  A measurement and test
                                switch: the same bits in every plane either way. */
  RM_RENDER_NO_CULL = 128     /* long primitive tables (no domain rows): evaluate every row of the table at every point instead of
                                the rows the point's grid cell lists.  A row that is an exact no-op everywhere in the cell is
                                skipped: min(d, di) with the shape further away than the running value, max(d, +-di) with the term
                                below it; and a far smooth-union row whose rounding of the running value -- d' = fl(di -
                                fl(di - d)), what mix(di, d, 1) does -- is provably the identity because d already lies on a
                                grid at least as coarse.  Other smooth-union rows are never skipped.
                                Both builds since round 4 (the parity build's GL-stack arithmetic folds every row).  A measurement
                                and test switch: the same bits either way. */
};

enum { RM_PLANE_COLOR = 0, RM_PLANE_NORMAL_DOF = 1, RM_PLANE_ALBEDO_DEPTH = 2 };

/* ---- context ----------------------------------------------------------- */

/* Replaces loadRenderJobContext(gl) (LoadRenderJobContext.tsx:268-287): binds
 * a device, creates the stream the launches go to.  Fails with
 * RM_ERR_NO_DEVICE when there is no GPU -- there is no CPU path. */
RM_API int rm_ctx_create(int device, rm_ctx** out);
RM_API void rm_ctx_destroy(rm_ctx* ctx);
/* Text of the last error on this context ("" if none); ctx may be NULL for
 * errors of rm_ctx_create itself.  Replaces ShaderError.infoLog. */
RM_API const char* rm_last_error(const rm_ctx* ctx);
/* Use an externally owned hipStream_t (e.g. torch's current stream) for all
 * later launches; NULL restores the context's own stream.  The switch is
 * ordered on the device (an event, no host wait): work this library queued on
 * the old stream -- the zeroing of rm_fb_create / rm_fb_clear, uploads,
 * renders -- completes before anything enqueued on the new stream after the
 * call.  The old stream must still exist when this is called. */
RM_API int rm_ctx_set_stream(rm_ctx* ctx, void* hip_stream);
/* Samples in flight (default 3, 1 = off; environment RM_SAMPLES_IN_FLIGHT).  The reference submits its draw calls
 * back to back (RenderJobExecutor.tsx:240-260) and the GPU overlaps them; here a full-mode sample of the pixel
 * kernel renders on an internal side stream into a staging buffer (it reads no plane) and a small kernel on the
 * context's stream blends it into the planes, in call order, so up to n consecutive rm_render_sample(s) calls
 * overlap.  Every call is still ordered on the context's stream as far as the planes are concerned: work enqueued
 * there afterwards sees the sample blended.  The planes receive the same bits as without overlap.  Costs 3 planes
 * of staging per sample in flight. */
RM_API int rm_ctx_set_samples_in_flight(rm_ctx* ctx, int n);
/* Advice that came with the last rm_ctx_set_samples_in_flight that SUCCEEDED ("" = none): the process environment
 * (GPU_MAX_HW_QUEUES < 8, read by the HIP runtime when it starts) will keep the samples from overlapping.  Not an
 * error: rm_last_error is for failures only.  The pointer stays valid until the next such call on the context. */
RM_API const char* rm_ctx_last_warning(const rm_ctx* ctx);
/* Samples per launch of rm_render_samples (default 0 = automatic, 1 = one launch per sample, 2..8 = fixed).  A small
 * window -- one GPU's rows of a sharded frame -- has too few workgroups to keep the chip busy to the end of a launch
 * (a ray is a serial chain of ~1 ms); rm_render_samples therefore renders up to 8 consecutive samples of the job in
 * ONE launch, a workgroup per (tile, sample), each sample staged separately and blended into the planes in sample
 * order by one small kernel -- the same bits as one launch per sample.  Automatic = as many as bring the launch to
 * about 16 384 workgroups (a whole 3840x2160 frame has 16 320, so whole frames are not batched), within a staging
 * budget of a quarter of the device's free memory.  Staging costs 48 bytes per pixel of the TILE a launch renders, per
 * sample of a batch and per launch in flight; a launch whose staging cannot be allocated renders unstaged (same bits). */
RM_API int rm_ctx_set_sample_batch(rm_ctx* ctx, int n);
/* Cost-ordered dispatch (default on; environment RM_COST_ORDER=0 turns it off).  From the second sample of a job on
 * (same framebuffer window, tile and scene kind, >= 512 workgroups), the pixel kernel starts its tiles in the order of
 * their cost in the previous sample, most expensive first, so that a launch does not end on a few late, long
 * workgroups.  Scheduling only: the planes receive the same bits. */
RM_API int rm_ctx_set_cost_order(rm_ctx* ctx, int on);
/* RM_RENDER_FAST only, opt-in: a marching lane counts as settled once its step
 * |d| <= eps * max(1, |p|_inf).  The default is 0: only the exact test
 * (position bitwise unchanged), which is what RM_RENDER_STRICT always uses.
 * A tolerance is NOT neutral for the image: rays stop a few ulps above the
 * surface, where the forward-difference normals (delta = 1e-5) are less noisy
 * than at the reference's fixed point, and lit pixels come out brighter --
 * Mandelbulb, 32 spp, mean of the lit pixels against the parity build:
 * eps 2^-25 +0.7 %, 2^-24 +2.9 %, 2^-23 +4.7 %, 2^-21 +6.8 %, 1e-5 +14 %, for
 * 1 %, 3 %, 8 %, 12 % and 28 % less time (tools/eps_study.py). */
RM_API int rm_ctx_set_retire_eps(rm_ctx* ctx, float eps);
/* Parity mode of the strict build: on != 0 makes everything this context does WITHOUT RM_RENDER_FAST -- rm_render_sample(s),
 * rm_probe*, rm_present* -- compute in the arithmetic of the GL stack the reference's golden images were rendered under
 * (SwiftShader as shipped in HeadlessChrome 88): its sin / cos / log / exp / pow / acos / atan as IEEE operation sequences,
 * min / max as x86 computes them, fract clamped below 1, the canvas's unorm conversion through 16 bits.  GLSL leaves all of
 * these to the implementation (raymarcher.frag has one value only together with a GL stack); with this switch the GPU
 * reproduces tests/golden/ -- distances, marches, whole main() images, the present pass -- bit for bit.  The pixel kernel
 * only (the wavefront pipeline is not built in this arithmetic); slower than the default strict arithmetic's IEEE-derived
 * shortcuts allow.  Off by default.  on == 2 additionally replaces the portable tangent (the fixed operation sequence every
 * other arithmetic of this library uses for tan(), see RmUniforms / DESIGN.md) by that stack's own tan = sin / cos: what the
 * reference's UNMODIFIED shader text computes under it, random stream and camera included (that switch is one per DEVICE:
 * the GL-stack contexts of a device share it, and changing it waits for the device). */
RM_API int rm_ctx_set_gl_stack(rm_ctx* ctx, int on);
/* Culling grids (long primitive tables without domain rows; RM_RENDER_NO_CULL): a scene's grid is built -- on a stream of
 * the context's own, with no wait on the host; the render that asks for it waits for the build on the device -- once the
 * scene has been asked for `pixels` pixel-samples since it was created (default 4 Mi: a 4K frame's first sample builds it,
 * a host that shows a NEW scene in every 1080p frame, like the reference's sliders do (index.tsx:121-182), never pays for a
 * grid it would not earn back); 0 = with the first render.  Counted per JOB: every sample of an rm_render_samples call, and
 * a striped framebuffer's part of a frame as the whole frame (every rank of a sharded job reaches the threshold when a
 * single GPU would).  The same bits with and without a grid.  rm_ctx_cull_stats: out4 = {grids built so far, bytes the
 * scenes' grids hold now, grids held now, the budget in bytes (a sixteenth of the device's memory, at most 1 GiB: beyond it
 * the least recently rendered scene gives its grid up and renders on without one; a grid larger than the whole budget is
 * not built until the budget is raised)}.  Buffers of grids that were given up are kept for the next build (at most 4, at
 * most half the budget) and count against the budget: grids + recycled buffers never exceed it.  Lowering the budget
 * (rm_ctx_set_cull_budget) takes effect at once. */
RM_API int rm_ctx_set_cull_min_pixels(rm_ctx* ctx, long long pixels);
RM_API int rm_ctx_set_cull_budget(rm_ctx* ctx, size_t bytes); /* the bytes a context's grids may hold (a host that knows its memory better; the tests: small, to see grids go) */
RM_API int rm_ctx_cull_stats(const rm_ctx* ctx, unsigned long long* out4);
/* Diagnostics of the wavefront march, filled only by builds compiled with
 * -DRM_WF_STATS (zeros otherwise): out16[8*shadow + 4*pass2 + {0,1,2}] =
 * rays marched, lane-steps, wave-steps since the last reset. */
RM_API int rm_debug_counters(rm_ctx* ctx, unsigned long long* out16, int reset);

/* Debug (tests): the rows of a primitive table (no domain rows) that an evaluation anywhere in the ball (centre, radius) has
 * to fold -- what the culling grid stores per cell (RM_RENDER_NO_CULL) -- as (nprims + 63) / 64 64-bit words,
 * bit i = row i stays (row 0 always does); `margin` = the allowance for fp32 rounding (0 tests the rule in
 * exact arithmetic).  Host arithmetic:
 * needs no GPU and no context.  Returns RM_ERR_INVALID for a table with domain rows. */
RM_API int rm_debug_cull_cell(const RmSceneDesc* desc, const double* centre, double radius, double margin, unsigned long long* out_words);
/* Which implementation of the per-pixel program the LAST rm_render_sample(s) / rm_render_timed call on this context
 * dispatched (the library picks per job unless a flag forces one; same results either way): what a host reports next to a
 * timing instead of re-deriving the library's rule. */
enum { RM_PIPELINE_NONE = 0, RM_PIPELINE_PIXEL_KERNEL = 1, RM_PIPELINE_WAVEFRONT = 2 };
RM_API int rm_ctx_last_pipeline(const rm_ctx* ctx);
/* Free and total memory of the context's GPU (hipMemGetInfo): what a host sizes its frames against -- the planes of
 * a W x H frame take 48 W H bytes, staging 48 W H per sample in flight or in a batch, the wavefront pipeline 240 bytes
 * per pixel of a launch (the reference asks MAX_TEXTURE_SIZE instead). */
RM_API int rm_device_memory(rm_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
/* Completion point: the reference's generator yield / present cadence
 * (RenderJobExecutor.tsx:163-166) maps to "enqueue samples, rm_sync, present". */
RM_API int rm_sync(rm_ctx* ctx);
RM_API int rm_abi_version(void);

/* ---- scene ------------------------------------------------------------- */

/* Replaces programCache.getProgram(vert, frag-with-scene)
 * (RenderJobExecutor.tsx:121-127, ShaderCache.tsx:91-119): validates the
 * description (the analogue of a GLSL compile) and uploads the primitive
 * table.  On failure returns RM_ERR_INVALID and rm_last_error() is the
 * "info log". */
RM_API int rm_scene_create(rm_ctx* ctx, const RmSceneDesc* desc, rm_scene** out);
RM_API void rm_scene_destroy(rm_scene* scene);

/* ---- framebuffers ------------------------------------------------------ */

/* Replaces context.fbo.create(w, h, frameid) (LoadRenderJobContext.tsx:186-223):
 * three accumulation planes of float4 per pixel (color RGBA, normal+dofRadius,
 * albedo+depth; raymarcher.frag:70-72).  The reference ping-pongs prev/curr
 * and copies curr->prev after every draw (RenderJobExecutor.tsx:301-326);
 * here each pixel is read and written by the one thread that owns it, so a
 * single in-place set is the same result.  The frame may be a window of rows
 * [row_begin, row_begin+row_count) of a width x height image (row sharding
 * across GPUs): pixel coordinates, texcoord and aspect stay global.
 * Planes are zero-initialised.  Row 0 is the BOTTOM row (GL convention). */
RM_API int rm_fb_create(rm_ctx* ctx, int width, int height, int row_begin, int row_count, rm_fb** out);
/* Row-striped window for sharding one image over several GPUs with balanced
 * cost: the frame is cut into stripes of `stripe_rows` rows and this
 * framebuffer holds, packed in ascending order, the stripes k with
 * k % parts == part (rows r with (r / stripe_rows) % parts == part).
 * rm_fb_rows() tells how many rows that is.  Planes may be caller-owned
 * (non-NULL color; normal_dof/albedo_depth both or neither) or NULL to let the
 * library allocate them.  Pixel coordinates stay global as in rm_fb_create. */
RM_API int rm_fb_create_striped(rm_ctx* ctx, int width, int height, int stripe_rows, int parts, int part,
                         void* color, void* normal_dof, void* albedo_depth, rm_fb** out);
/* Number of image rows a framebuffer holds; width / height of the image it is a window of. */
RM_API int rm_fb_rows(const rm_fb* fb);
RM_API int rm_fb_width(const rm_fb* fb);
RM_API int rm_fb_height(const rm_fb* fb);
/* Same, but over caller-owned device memory (e.g. torch tensors): each plane
 * pointer addresses row_count*width float4; normal_dof/albedo_depth may be
 * NULL (then only RM_RENDER_COLOR_ONLY renders are accepted). */
RM_API int rm_fb_wrap(rm_ctx* ctx, int width, int height, int row_begin, int row_count,
               void* color, void* normal_dof, void* albedo_depth, rm_fb** out);
/* The clear-on-new-frameid of LoadRenderJobContext.tsx:196-208. */
RM_API int rm_fb_clear(rm_fb* fb);
RM_API void rm_fb_destroy(rm_fb* fb);
/* Copies one plane (row_count*width*4 floats) to / from host memory; synchronous. */
RM_API int rm_fb_download(rm_fb* fb, int plane, float* host);
RM_API int rm_fb_upload(rm_fb* fb, int plane, const float* host);
/* Device address of a plane (for collectives / zero-copy wrapping). */
RM_API void* rm_fb_device_ptr(rm_fb* fb, int plane);

/* Raw device memory for hosts that have no allocator of their own: rm_present_rows, rm_present_device and
 * rm_assemble_striped_bytes take DEVICE pointers (in the reference these are textures the GL context owns,
 * LoadRenderJobContext.tsx:43-124; a torch host passes tensor addresses instead).  Created zero-filled; the copies
 * take the buffer's base address and at most its size, are synchronous and ordered after the work on the context's
 * stream; rm_ctx_destroy frees what is left. */
RM_API int rm_buffer_create(rm_ctx* ctx, size_t bytes, void** device_ptr);
RM_API int rm_buffer_destroy(rm_ctx* ctx, void* device_ptr);
RM_API int rm_buffer_download(rm_ctx* ctx, const void* device_ptr, void* host, size_t bytes);
RM_API int rm_buffer_upload(rm_ctx* ctx, void* device_ptr, const void* host, size_t bytes);

/* ---- the hot path ------------------------------------------------------ */

/* One sample of every pixel of `tile` (clipped to the framebuffer's rows):
 * replaces  bind prev -> setUniforms -> gl.drawArrays -> blit
 * (RenderJobExecutor.tsx:195-326) = one run of raymarcher.frag main()
 * (raymarcher.frag:178-388) per pixel.  tile == NULL means the whole window.
 * Asynchronous on the context's stream. */
RM_API int rm_render_sample(rm_ctx* ctx, rm_scene* scene, rm_fb* fb, const RmUniforms* uniforms,
                     const RmRect* tile, int flags);

/* `count` samples back to back, sample i using randNoise[i] (pairs); every
 * other uniform as given: the sample loop of a render job
 * (RenderJobExecutor.tsx:160-222 -- only randNoise changes from sample to
 * sample).  Same planes, bit for bit, as `count` calls of rm_render_sample;
 * full-mode samples of the pixel kernel go out several to a launch (see
 * rm_ctx_set_sample_batch). */
RM_API int rm_render_samples(rm_ctx* ctx, rm_scene* scene, rm_fb* fb, const RmUniforms* uniforms,
                      const float* rand_noise_pairs, int count, const RmRect* tile, int flags);

/* Like rm_render_samples with one fixed randNoise, but brackets the `count`
 * launches with HIP events on the context's stream and returns the average
 * kernel time per launch in *ms_per_launch (synchronous). */
RM_API int rm_render_timed(rm_ctx* ctx, rm_scene* scene, rm_fb* fb, const RmUniforms* uniforms,
                    int count, const RmRect* tile, int flags, float* ms_per_launch);

/* ---- probes (RNG-free building blocks of the path) --------------------- */

/* The reference's own functions evaluated on caller-supplied inputs, the
 * device analogue of the harness main() used to generate the goldens.
 * All buffers are HOST pointers; synchronous.
 *   RM_PROBE_SDF      in: n x float3 p                  out: n x float   sdf(p)            (scene text / examples)
 *   RM_PROBE_CAST_RAY in: n x (float3 p, float3 dir)    out: n x float3  castRay(p,dir,steps)   raymarcher.frag:163-170
 *   RM_PROBE_NORMAL   in: n x float3 p                  out: n x float3  sceneNormal(p, delta)  raymarcher.frag:153-160
 *   RM_PROBE_MATERIAL in: n x float3 p                  out: n x 12 floats diffuse,specular,emission,(roughness,subsurface,ior)
 *   RM_PROBE_CAST_STEPS in: n x (float3 p, float3 dir)  out: n x float   number of castRay steps after which the ray's
 *                                                       position no longer changes bitwise (= steps if it never settles);
 *                                                       a measurement aid for the wave-retire, not a reference function
 *   RM_PROBE_CAST_SHADOW in: n x (float3 p, float3 dir, float3 light)  out: n x float  1 if the light is seen, else 0: the shadow
 *                                                       test of one light, distance(castRay(p, dir, steps), light) >= distance(p, light)
 *                                                       (raymarcher.frag:362-363), marched as the pixel kernel marches a shadow ray
 * A table of 16 rows or more is marched as its pixel kernels march it (the long tables' far-field exits included).
 */
enum { RM_PROBE_SDF = 0, RM_PROBE_CAST_RAY = 1, RM_PROBE_NORMAL = 2, RM_PROBE_MATERIAL = 3, RM_PROBE_CAST_STEPS = 4, RM_PROBE_CAST_SHADOW = 5 };
RM_API int rm_probe(rm_ctx* ctx, rm_scene* scene, int what, const float* in, int n, float param,
             int flags, float* out);

/* Camera block of main() without jitter or depth of field
 * (raymarcher.frag:186-205 with randomDirectionOffset = dofOffset = 0):
 * out = height*width x 8 floats (origin.xyz, deltaZ, dir.xyz, 0), HOST pointer. */
RM_API int rm_probe_camera(rm_ctx* ctx, const RmUniforms* uniforms, int width, int height, float* out);

/* The per-pixel random stream (raymarcher.frag:46-49,78-101): for each pixel
 * of a width x height image the first `count` values of uniformSample().
 * out = height*width*count floats, HOST pointer. */
RM_API int rm_probe_rng(rm_ctx* ctx, const RmUniforms* uniforms, int width, int height, int count, float* out);

/* The transcendental functions of the parity arithmetic on arrays, as the kernels call them: the fp32 sequences of
 * csrc/rm_pm_math.hpp (the text of oracle/pm_math.h), or the GL stack's (rm_ss_math.hpp = oracle/ss_math.h) on a context
 * with rm_ctx_set_gl_stack -- so that "the same bits as the oracle" is tested function by function on millions of
 * arguments, not only through rendered frames.  POW is pow(|a|, b) (the oracle's gl_pow), ATAN2 atan(a, b) with a = y,
 * TAN the portable tangent of the random stream and camera (oracle/rm_oracle.c or_tan); POW_PAIR_NM1 / _N are
 * pow(a, b - 1) / pow(a, b) from one logarithm and SINCOS_S / _C sin / cos from one reduction, the shared forms the
 * Mandelbulb uses (they must have the bits of the separate calls).  a, b, out: HOST pointers to n floats; b may be NULL
 * for the functions of one argument.  Test infrastructure like rm_probe: no part of a render job calls it. */
enum { RM_MATH_SIN = 0, RM_MATH_COS = 1, RM_MATH_LOG = 2, RM_MATH_EXP = 3, RM_MATH_POW = 4, RM_MATH_ACOS = 5, RM_MATH_ATAN2 = 6, RM_MATH_TAN = 7,
       RM_MATH_POW_PAIR_NM1 = 8, RM_MATH_POW_PAIR_N = 9, RM_MATH_SINCOS_S = 10, RM_MATH_SINCOS_C = 11, RM_MATH_SQRT = 12, RM_MATH_DIV = 13, RM_MATH_COUNT = 14 };
RM_API int rm_probe_math(rm_ctx* ctx, int fn, const float* a, const float* b, int n, float* out);

/* ---- assembling a sharded frame ------------------------------------------ */

/* Puts the planes of a frame that was rendered as `parts` striped windows (rm_fb_create_striped: stripes of
 * `stripe_rows` rows dealt round-robin) back into image order, in one launch on the context's stream.
 * src: DEVICE pointer to parts x max_rows x width float4 (the gathered planes, part p at p * max_rows * width;
 * a part's rows beyond its own count are padding), dst: DEVICE pointer to height x width float4.  What a
 * multi-GPU host runs on the rank that shows the frame, after the gather (raymarching_engine_amd/dist.py).
 * hip_stream: the hipStream_t to launch on, NULL = the context's stream (a host that keeps rendering while the
 * frame is put together gives it a stream of its own, ordered after the gather). */
RM_API int rm_assemble_striped(rm_ctx* ctx, const void* src, int parts, int max_rows, int width, int height, int stripe_rows, void* dst,
                        void* hip_stream);
/* The same for rows of `row_bytes` opaque bytes (a multiple of 4; buffers 16-byte aligned): RGBA8 rows after the
 * per-rank present (rm_present_rows, row_bytes = 4 * width), or any other per-pixel payload. */
RM_API int rm_assemble_striped_bytes(rm_ctx* ctx, const void* src, int parts, int max_rows, long long row_bytes, int height, int stripe_rows,
                              void* dst, void* hip_stream);

/* ---- present ----------------------------------------------------------- */

/* The present pass of the reference (display.frag:16-64, driven from
 * index.tsx:25-59): colour x 1/samples, Gaussian blur whose radius is the
 * accumulated DoF radius (normal_dof.w / samples x 200, clamped to 16 pixels;
 * NEAREST + REPEAT taps), gamma 1/2.2, RGBA8.  out_rgba8 = height*width*4
 * bytes, HOST pointer, row 0 = bottom row.  The blur reads neighbouring rows,
 * so rm_present needs a framebuffer holding the whole frame; for a sharded
 * frame gather the planes first and use rm_present_planes (device pointers;
 * normal_dof may be NULL = no blur). */
RM_API int rm_present(rm_ctx* ctx, rm_fb* fb, int samples, uint8_t* out_rgba8);
RM_API int rm_present_planes(rm_ctx* ctx, const void* color, const void* normal_dof, int width, int height, int samples, uint8_t* out_rgba8);
/* The same pass left on the device and asynchronous: out_rgba8_device = height*width*4 bytes of DEVICE memory,
 * launched on hip_stream (NULL = the context's stream); no allocation, no host wait.  For a host that shows the
 * frame from device memory or reads it back itself (index.tsx:25-59 draws straight to the canvas). */
RM_API int rm_present_device(rm_ctx* ctx, const void* color, const void* normal_dof, int width, int height, int samples,
                      void* out_rgba8_device, void* hip_stream);
/* The present pass of the rows ONE framebuffer window holds, for jobs without depth of field (dof.amount = 0: the
 * blur radius of display.frag:24-27 is 0 and its only tap is the pixel itself, so no neighbour row is needed and
 * the bytes equal rm_present's).  out_rgba8_device = rm_fb_rows(fb)*width*4 bytes of DEVICE memory, in the
 * window's own row order.  This is what a sharded run gathers instead of the fp32 colour plane: a quarter of the
 * bytes (raymarching_engine_amd/dist.py); rm_assemble_striped_bytes puts the stripes in image order. */
RM_API int rm_present_rows(rm_ctx* ctx, rm_fb* fb, int samples, void* out_rgba8_device, void* hip_stream);
/* What a sharded job WITH depth of field gathers instead: the blur of display.frag:44-55 reads up to 16 rows either
 * side of a pixel, rows that other GPUs hold, so the present pass has to run where the whole frame is.  It reads two
 * things per pixel -- the accumulated colour (display.frag:19,53) and the accumulated DoF radius, normalAndDofRadius.w
 * (:21-23) -- and this call packs exactly those for the rows `fb` holds: out_float4_device = rm_fb_rows(fb)*width
 * float4 (colour.r, colour.g, colour.b, normal_dof.w) of DEVICE memory, in the window's own row order, asynchronous on
 * hip_stream (NULL = the context's stream).  Gathered and put in image order (rm_assemble_striped) the buffer is
 * passed to rm_present_device / rm_present_planes as BOTH `color` and `normal_dof` (they read .rgb of the one and .w of
 * the other): the bytes are those rm_present gives for the unsharded frame (raymarching_engine_amd/dist.py,
 * index.tsx:25-59 is the caller this serves). */
RM_API int rm_pack_present_rows(rm_ctx* ctx, rm_fb* fb, void* out_float4_device, void* hip_stream);
/* The present pass (display.frag:16-64) of ONE PART of a striped frame: `color` / `normal_dof` are DEVICE pointers to the WHOLE
 * frame in image order (height x width float4 each; with depth of field the gathered and assembled rows of
 * rm_pack_present_rows serve as both), out_rgba8_device receives the rows part `part` of `parts` holds (stripes of
 * stripe_rows rows dealt round-robin, packed as in rm_fb_create_striped: rows(part) x width x 4 bytes).  The taps,
 * their order and the arithmetic are rm_present_device's: assembled (rm_assemble_striped_bytes) the parts' bytes are
 * rm_present's.  This is how a sharded frame WITH depth of field is shown without one GPU blurring all of it: every
 * GPU gets the packed frame (an all-gather), blurs the stripes it holds -- 1 / parts of the pass -- and only RGBA8
 * travels to the GPU that shows the frame.  Asynchronous on hip_stream (NULL = the context's stream). */
RM_API int rm_present_striped_rows(rm_ctx* ctx, const void* color, const void* normal_dof, int width, int height, int samples, int stripe_rows, int parts, int part,
                            void* out_rgba8_device, void* hip_stream);

/* The present of a frame that ONE process renders on several GPUs -- the shape of the reference's own host: one
 * thread, one render loop (index.tsx:120), here with a context per GPU, each holding one part of the frame's stripes
 * (rm_fb_create_striped with parts = `parts`, part = p on ctxs[p]; the host hands every sample to each context in turn
 * and the GPUs render concurrently, launches being asynchronous).
 *   rm_present_sharded_start: every context snapshots the rows it holds on its own stream -- tone-mapped (dof == 0,
 *     rm_present_rows) or packed (dof != 0, rm_pack_present_rows) -- and everything after that is enqueued on streams of
 *     the present's own, so the host can hand out the next samples at once (RenderJobExecutor.tsx:163-166 yields after a
 *     present; here the frame travels WHILE the next samples render).  dof == 0: the RGBA8 rows go to ctxs[0]'s GPU by
 *     peer copies over xGMI (hipMemcpyPeerAsync; no collective library, no second process) and are put in image order
 *     there.  dof != 0: the packed rows go to EVERY GPU, each puts the frame together, blurs and tone-maps the stripes
 *     it holds (rm_present_striped_rows: 1 / parts of the pass per GPU) and sends those bytes to ctxs[0]'s GPU.  The
 *     canvas is copied to pinned host memory.  Nothing is waited for.  One present at a time: finish before the next start.
 *   rm_present_sharded_finish: waits for THAT present only (an event; renders enqueued since keep running) and copies
 *     the canvas to out_rgba8 = height*width*4 bytes of HOST memory, row 0 = bottom: the bytes rm_present gives for the
 *     same samples on one framebuffer.  out_bytes = the size of that buffer: RM_ERR_INVALID, with the present still pending,
 *     when it is smaller than the canvas of the present that was STARTED (the library knows that size, the caller of a
 *     finish alone may not: round 4's signature trusted it).
 *   rm_present_sharded: both, one after the other (synchronous).
 * (Hosts with a process per GPU -- bench.py, job.RenderJobContext(group=...) -- move the same rows over RCCL instead:
 * raymarching_engine_amd/dist.py.) */
RM_API int rm_present_sharded_start(rm_ctx* const* ctxs, rm_fb* const* fbs, int parts, int samples, int dof);
RM_API int rm_present_sharded_finish(rm_ctx* const* ctxs, int parts, uint8_t* out_rgba8, size_t out_bytes);
RM_API int rm_present_sharded(rm_ctx* const* ctxs, rm_fb* const* fbs, int parts, int samples, int dof, uint8_t* out_rgba8, size_t out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* HIP_RAYMARCH_H */
