// rm_params.hpp -- kernel argument blocks shared by the API and both kernel builds.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/hip_raymarch.h"

// table_flags of DevScene (set by rm_scene_create)
#define RM_TABLE_SPHERES_SMOOTH 1  /* every row is a sphere and every fold after the first a smooth union */
#define RM_TABLE_HAS_DOMAIN 2      /* the table has domain rows (RM_PRIM_REPEAT / RM_PRIM_FOLD) */
#define RM_TABLE_NO_BOXES 8         /* no RM_PRIM_BOX row: the sdf of a point with a non-finite coordinate is itself non-finite */
#define RM_TABLE_UNIFORM_K 4       /* RM_TABLE_SPHERES_SMOOTH with one k for every fold: k in p[0], 0.5 / k in p[1] */
#define RM_TABLE_HAS_KIND 32      /* some shape row evaluates a scene kind's estimator (RM_PRIM_KIND): its own pixel kernel, no far-field exits, no row culling */
#define RM_TABLE_MORE 64          /* some row uses ABI 8's vocabulary (torus / cylinder / plane, smooth subtraction / intersection): the second copy of the general fold */
#define RM_TABLE_HAS_SURFACES 16   /* some shape row names a surface (RmPrim.type bits 16..23): the material functions depend on the position */

#define RM_BATCH_MAX 8  /* samples one pixel-kernel launch can render (KParams::batch) */

// Row culling for primitive tables without domain rows (round 3; both builds since round 4; rm_kernels.inc rm_cull_build_kernel fills
// it before the first launch that can use it, rm_device.hpp Sdf<RM_SCENE_TABLE>::culled_rows reads it).  Nested uniform grids about
// the shapes' bounding box: level l is a cube with half-width half0 * 2^l about `centre`, of n^3 cells for level 0 and n_outer^3 for
// the levels around it (round 5: a ray spends its steps near the shapes, where the cells have to be small; the levels it merely
// crosses on its way in take cells twice as wide -- 31.5 MB instead of 134 for CSG-64), and a point belongs to
// the lowest level that holds it.  Per cell one bit per row -- clear = the row's operator cannot change the running value of the fold
// anywhere in the cell (an exact no-op there), so an evaluation may skip it.  cells = [level 0: n^3][levels 1 ..: n_outer^3 each][words]
// 64-bit words, then one cell that lists every row (points beyond the last level, non-finite points).
struct CullGrid {
  const unsigned long long* cells;  // device pointer; nullptr = no culling
  float centre[3];
  float inv_half0;  // 1 / half0
  float scale0;     // cells per unit length at level 0: n / (2 half0)
  int n, n_outer, levels, words;  // n_outer is n or n / 2
};
struct CullBuild {  // argument block of rm_cull_build_kernel
  const RmPrim* prims;
  unsigned long long* cells;
  int nprims, n, n_outer, levels, words;
  double centre[3], half0;
  double reach;  // the largest |coordinate| of a shape: scale of the fp32 error allowance
};
__host__ __device__ inline long long rm_cull_cells(int n, int n_outer, int levels) {  // without the cell that lists every row
  return (long long)n * n * n + (long long)(levels - 1) * n_outer * n_outer * n_outer;
}

// the fp32 allowance of the culling tests: n + 8 roundings at the magnitude of the coordinates involved
__host__ __device__ inline double rm_cull_margin(int nprims, double magnitude) { return 1e-4 + 1.2e-7 * (nprims + 8) * magnitude; }
// a shape row as the rule reads it: the table's floats widened once, and what the rule derives from them per row and not per cell
// (round 5: the build kernel stages these in LDS once per workgroup; rounds 3-4 read the 32-byte RmPrim from memory and divided
// by k twice, per row AND cell)
struct CullRow {
  double c[3], size[3], k, half_inv_k;
  double reach;  // a box's half-diagonal (0 for a sphere): how far its nearest point can be from its centre
  int shape, op;
};
__host__ __device__ inline CullRow rm_cull_row(const RmPrim& p) {
  CullRow r;
  for (int a = 0; a < 3; a++) { r.c[a] = p.center[a]; r.size[a] = p.size[a]; }
  r.k = (double)p.k;
  r.half_inv_k = 0.5 / r.k;  // (Inf for k = 0: only read for smooth unions, whose k > 0)
  r.shape = p.type & 0xff;
  r.op = (p.type >> 8) & 0xff;
  r.reach = r.shape == RM_PRIM_SPHERE ? 0.0 : sqrt(r.size[0] * r.size[0] + r.size[1] * r.size[1] + r.size[2] * r.size[2]);
  return r;
}
__host__ __device__ inline double rm_cull_shape_distance(const CullRow& p, const double* c) {
  const double x = c[0] - p.c[0], y = c[1] - p.c[1], z = c[2] - p.c[2];
  if (p.shape == RM_PRIM_SPHERE) return sqrt(x * x + y * y + z * z) - p.size[0];
  const double qx = fabs(x) - p.size[0], qy = fabs(y) - p.size[1], qz = fabs(z) - p.size[2];  // sdBox, raymarcher.frag:108-112
  const double ox = fmax(qx, 0.0), oy = fmax(qy, 0.0), oz = fmax(qz, 0.0);
  return sqrt(ox * ox + oy * oy + oz * oz) + fmin(fmax(qx, fmax(qy, qz)), 0.0);
}
// The rows of a table (no domain rows) that an evaluation anywhere in the ball (c, rad) has to fold: bit i of out[] set = row i
// stays.  The argument is with rm_cull_build_kernel (rm_kernels.inc), which calls this once per cell; host-callable so that the
// CPU tests can check it against the fold itself (rm_debug_cull_cell).
//
// Hard operators: min(d, di) with the shape further away than the running value can be anywhere in the cell, max(d, +-di) with
// the term below it -- exact no-ops whatever the arithmetic.
//
// Smooth unions (round 4).  Round 3 built row culling for them, measured it (C4 14.2 -> 7.0 ms) and took it out: a row further than k
// from the running value does not leave it alone -- mix(di, d, 1), in both builds' arithmetic, is  d' = fl(di - fl(di - d)): it ROUNDS
// d to the grid of (di - d), and that noise is what the creeping shadow rays of a smooth-union scene live on (14 % fewer lit pixels
// without it).  The rounding itself can be reasoned about exactly.  Write e(x) = floor(log2 |x|); a float x is a multiple of
// 2^(e(x) - 23) (the argument is the same in any binary precision).
//  * EXECUTING a far row m (t = fl(dm - d), d' = fl(dm - t)): dm and t are multiples of 2^(g - 23), g = min(e(dm), e(t)), so is their
//    difference, and it is representable when |d'| < 2^(g + 1): then d' = dm - t exactly and d' lies ON THE GRID 2^(g - 23).
//  * a later far row i finds d on a grid 2^(B - 23) with B >= e(di - d) and e(di - d) <= e(di): then di and d are both multiples of
//    the unit of t = di - d, t is exact, and d'' = di - t = d: the row is an EXACT NO-OP and may be skipped.
//  * any row that is executed and is not certainly far -- a near smooth union, a hard operator that may take the other value --
//    moves d off every grid: the chain starts again after it; a far row that is executed without meeting the no-op conditions
//    leaves d on ITS grid, which may be finer than the one before; a row that is skipped leaves d, and the grid it is on, alone.
// Per cell the rule runs this bookkeeping on INTERVALS: the running value in [L, U] (the smooth minimum is monotone in both
// arguments, so its interval is the smooth minimum of the ends), row i's distance in [lo_i, hi_i], binades taken where the whole
// interval lies in one; a distance that is negative or changes sign in the cell claims nothing.  What the rule cannot decide it
// keeps: a kept row is the reference's own arithmetic, and any superset of a point's list gives the same bits (the extra rows are
// identities there) -- which is what lets a wave fold the union of its lanes' lists.  About half of CSG-64's 64 rows stay per cell.
// tests/test_cull_rule_cpu.py checks the rule against fp32 restatements of both builds' folds; the GPU suite holds the culled fold
// to RM_RENDER_NO_CULL bit for bit.
// the exponent of a positive, finite, normal double, read off its exponent field (round 5; rounds 3-4 called the library's
// logarithm -- three times per row and cell, most of the build kernel's 8.9 ms on CSG-64's grid)
__host__ __device__ inline int rm_exponent(double v) { return ilogb(v); }
__host__ __device__ inline double rm_smooth_min(double a, double b, double k, double half_inv_k) {  // examples/smooth-tree.glsl:20-22, in double (0.5 / k given)
  double h = 0.5 + (b - a) * half_inv_k;
  h = h < 0.0 ? 0.0 : (h > 1.0 ? 1.0 : h);
  return b + h * (a - b) - k * h * (1.0 - h);
}
// tables whose rows are mostly smooth unions get the finer grid: the binade tests want small cells (rm_api.hip table_cull_params)
__host__ __device__ inline bool rm_cull_mostly_smooth(const RmPrim* prims, int nprims) {
  int smooth = 0;
  for (int i = 1; i < nprims; i++) smooth += ((prims[i].type >> 8) & 0xff) == RM_OP_SMOOTH_UNION;
  return 2 * smooth > nprims;
}

__host__ __device__ inline void rm_cull_cell_rows(const CullRow* prims, int nprims, int words, const double* c, double rad, double margin, unsigned long long* out) {
  const int NONE = -100000;
  for (int w = 0; w < words; w++) out[w] = 0ull;
  const double d0 = rm_cull_shape_distance(prims[0], c);
  double L = d0 - rad, U = d0 + rad;
  int best = 0;  // the nearest earlier row at c that bounds the running value from above (-1: none)
  double best_d = d0;
  int B = NONE;  // the running value is certainly a multiple of 2^(B - 23)
  out[0] |= 1ull;
  for (int i = 1; i < nprims; i++) {
    const CullRow& p = prims[i];
    const int op = p.op;
    const double di = rm_cull_shape_distance(p, c), lo_i = di - rad, hi_i = di + rad;
    bool keep;
    if (op == RM_OP_SMOOTH_UNION) {
      const double k = p.k;  // > 0 (rm_scene_create)
      keep = true;
      if (!(lo_i - U >= 1.002 * k + 2.0 * margin)) {
        B = NONE;  // possibly near: whatever grid d was on, it leaves it
      } else {
        const double emin = lo_i - U - 2.0 * margin, emax = hi_i - L + 2.0 * margin;  // t = di - d over the cell, both > 0
        const int bmin = rm_exponent(emin * (1.0 - 1e-5)), bmax = rm_exponent(emax * (1.0 + 1e-5));
        // the lowest binade of |di| over the cell; a far row's distance may be negative (the running value is then further inside
        // still) or change sign in the cell: then |di| has no lowest binade and nothing is claimed
        const double lo_m = lo_i - margin, hi_m = hi_i + margin;
        const int bdi = lo_m > 0.0 ? rm_exponent(lo_m * (1.0 - 1e-5)) : (hi_m < 0.0 ? rm_exponent(-hi_m * (1.0 - 1e-5)) : NONE);
        if (B != NONE && B >= bmax && bmax <= bdi) {
          keep = false;  // an exact no-op everywhere in the cell
        } else {
          const int g = bmin < bdi ? bmin : bdi;
          const double big = (fabs(L) > fabs(U) ? fabs(L) : fabs(U)) + margin;
          B = g != NONE && big * (1.0 + 1e-5) < ldexp(1.0, g + 1) ? g : NONE;  // the grid this row leaves d on, when d' = dm - t is exact
        }
      }
      const double u_smooth = rm_smooth_min(U, hi_i, k, p.half_inv_k) + margin;
      U = fmin(fmin(U, hi_i), u_smooth);
      L = rm_smooth_min(L, lo_i, k, p.half_inv_k) - margin;  // (at most k / 4 below the minimum)
      if (best < 0 || di < best_d) { best = i; best_d = di; }
    } else if (op == RM_OP_UNION) {
      const double k = 0.0;
      keep = !(lo_i >= U + margin);
      if (keep && best >= 0) {
        const CullRow& q = prims[best];
        const double sx = p.c[0] - q.c[0], sy = p.c[1] - q.c[1], sz = p.c[2] - q.c[2];
        // the point the gradient of a shape's distance points away from: a sphere's centre; the nearest point of a box (within its
        // half-diagonal of the centre, at the distance itself -- which has to be positive over the ball)
        const bool ps = p.shape == RM_PRIM_SPHERE, qs = q.shape == RM_PRIM_SPHERE;
        const double pe = p.reach, qe = q.reach;
        const double s = sqrt(sx * sx + sy * sy + sz * sz) + pe + qe;
        const double a = (ps ? di + p.size[0] : di) - rad, b = (qs ? best_d + q.size[0] : best_d) - rad;
        if (a > 0.0 && b > 0.0) {
          const double lip = fmin(2.0, s / sqrt(a * b));
          keep = !(di - best_d - rad * lip >= k + margin);
        }
      }
      U = fmin(U, hi_i);
      L = fmin(L, lo_i);
      if (best < 0 || di < best_d) { best = i; best_d = di; }
      if (keep) B = NONE;  // min(d, di) may hand on di
    } else if (op == RM_OP_SUBTRACT) {
      keep = !(-lo_i <= L - margin);
      U = fmax(U, -lo_i);
      L = fmax(L, -hi_i);
      if (keep) best = -1, B = NONE;  // the value may now exceed every earlier term
    } else {
      keep = !(hi_i <= L - margin);
      U = fmax(U, hi_i);
      L = fmax(L, lo_i);
      if (keep) best = -1, B = NONE;  // max(d, di) where di may win: no earlier term bounds the value from above any more
    }
    if (keep) out[i >> 6] |= 1ull << (i & 63);
  }
}

// ... from the table's own rows (the host's rm_debug_cull_cell; at most RM_MAX_PRIMS of them)
__host__ __device__ inline void rm_cull_cell(const RmPrim* prims, int nprims, int words, const double* c, double rad, double margin, unsigned long long* out) {
  CullRow rows[RM_MAX_PRIMS];
  for (int i = 0; i < nprims; i++) rows[i] = rm_cull_row(prims[i]);
  rm_cull_cell_rows(rows, nprims, words, c, rad, margin, out);
}

struct DevScene {
  int kind;
  int nprims;
  int table_flags;
  int reserved;
  const RmPrim* prims;  // device pointer (RM_SCENE_TABLE)
  float p[16];
  RmMaterial mat;
  const RmSurface* surfaces;  // device pointer: nsurfaces + 1 entries, [0] = the values of `mat` (RM_TABLE_HAS_SURFACES)
  int nsurfaces;
  // The far field of a primitive table without domain rows (rm_device.hpp Sdf<RM_SCENE_TABLE>::far_jump; set by rm_scene_create):
  // far_end = 0: no jump; 1: an escaping ray ends at +-Inf by the sign of its direction components; 2: at (NaN, NaN, NaN).
  // far_r2 = (2 R' + 1)^2 with R' = the radius of a sphere about the origin that holds every shape, plus 1.01 x the largest smooth-union radius k (a chain of
  // smooth unions stays within k of the minimum): rm_api.hip table_far_field; far_escape's Rp is R' + 1/2.
  int far_end;
  float far_r2;
  // a ray that passes every shape of a table at a distance (rm_device.hpp Sdf<RM_SCENE_TABLE>::clear_miss): the clearance beyond a
  // shape's bounding sphere is clear_k (the largest smooth-union radius) + the step b0 the march is then sure of, which the steps it
  // has left decide: b0 = 2.05 clear_rho / (left - 84), clear_rho = the radius of a sphere about the origin that holds every
  // shape's sphere with 15 % to spare.  clear_rho = 0: no such test
  float clear_k;
  float clear_rho;
  CullGrid cull;
};

struct KParams {
  RmUniforms u;
  DevScene scene;
  float4* color;
  float4* normal_dof;    // nullptr = colour only
  float4* albedo_depth;
  int W, H;            // full image (textureSize(previousColor), raymarcher.frag:183)
  int row_begin;       // contiguous window: first image row held by the planes
  int stripe_rows, parts, part;  // striped window (stripe_rows > 0): rows r with (r / stripe_rows) % parts == part
  int tx, ty, tw, th;  // tile clipped to the window; ty/th count LOCAL rows (rows of the planes)
  float retire_eps;    // 0 = exact (fixed-point) retire only
  // Staged output (full mode, samples in flight): the pixel kernel does not touch the planes; it leaves what
  // the sample adds -- (light * exposure, -), (normal, dofRadius), (albedo, depth) -- in stage[0..2][pixel] and
  // rm_combine_kernel applies the blend afterwards, in sample order.  nullptr = blend in the kernel.
  float4* stage;
  long long stage_stride;  // elements between the three staged planes
  // Sample batch (rm_render_samples, staged output only): ONE launch renders `batch` consecutive samples of the job --
  // workgroup b renders tile b / batch for sample b % batch, with randNoise batch_noise[sample] instead of u.randNoise
  // and its output staged at stage + sample * 3 * stage_stride -- so that a small window (one GPU's share of a frame)
  // still gives the chip a full frame's worth of workgroups.  0 or 1 = one sample.
  int batch;
  int tile_wide;  // set by the launcher: 0 = the kind's own tile; n > 0 = 8-wave workgroups with 2^(n-1) waves across (3: 32 x 16 pixels instead of 16 x 32; rm_kernels.inc BlockShape)
  float batch_noise[RM_BATCH_MAX][2];
  // Cost-ordered dispatch (pixel kernel): workgroup b of the launch renders tile block_order[b], and every workgroup
  // leaves its duration in block_cost[tile] for the next sample's order (rm_order_kernel).  nullptr = in launch order.
  const unsigned int* block_order;
  unsigned int* block_cost;
  int no_far_jump;  // RM_RENDER_NO_FAR_JUMP: march escaping rays step by step (a measurement / test switch, same bits)
};

// a table long enough for the compacting pixel kernel (rm_device.hpp RM_KIND_TABLE_BIG); launcher and grid computation agree through this
#define RM_TABLE_BIG_ROWS 16
__host__ __device__ inline bool rm_table_big(const KParams& P) {
  return P.scene.kind == RM_SCENE_TABLE && P.scene.nprims >= RM_TABLE_BIG_ROWS && P.u.renderMode == 0 && !(P.scene.table_flags & RM_TABLE_HAS_KIND);
}

// image row of a local (plane) row
__host__ __device__ inline int rm_global_row(const KParams& P, int local) {
  if (P.stripe_rows <= 0) return P.row_begin + local;
  return ((local / P.stripe_rows) * P.parts + P.part) * P.stripe_rows + local % P.stripe_rows;
}

struct ProbeParams {
  DevScene scene;
  const float* in;
  float* out;
  int n;
  int what;
  float param;
  float retire_eps;
  int no_far_jump;
};
// (the probe follows RM_RENDER_NO_CULL through scene.cull.cells, which rm_probe clears)

namespace rm {
enum {
  WF_POS = 0,   // xyz ray position (march in/out), w = step budget
  WF_DIR,       // xyz ray direction, w = deltaZ (preview)
  WF_OLD,       // xyz position at the start of the bounce, w = rng.seed
  WF_ALB,       // xyz albedo accumulated over bounces
  WF_LIG,       // xyz light accumulated over bounces
  WF_NRM,       // xyz normal of the current bounce
  WF_PDIR,      // xyz prevRayDirection
  WF_DIF,       // xyz diffuseCol
  WF_SPC,       // xyz specularCol
  WF_PALB,      // xyz prevAlbedo
  WF_ADJ,       // xyz adjustedLightPosition, w = distance(rayPosition, adjustedLightPosition)
  WF_CONTRIB,   // xyz what the current light adds if its shadow test passes (:368-371)
  WF_SPOS,      // shadow ray position (march in/out), w = step budget
  WF_SDIR,      // shadow ray direction
  WF_AUX,       // preview: x = stepsTaken, y = depth
  WF_ARRAYS
};

struct WfParams {
  KParams k;
  float4* a[WF_ARRAYS];
  unsigned int* head;  // queue head of THIS march launch (zeroed by the host)
  const unsigned int* list_in;        // pass 2: the queue is list_in[0 .. *list_in_count)
  const unsigned int* list_in_count;
  unsigned int* list_out;             // where this launch parks rays (pass 1: rays that need the deep evaluation;
  unsigned int* list_out_count;       //   pass 2 with repark > 0: the survivors of thinned-out waves); zeroed by the host
  int repark;                         // pass 2: once the queue is drained, a wave with <= repark active lanes parks them and exits
  unsigned long long* stats;  // RM_WF_STATS builds: [shadow?][pass2?][rays, lane-steps, wave-steps, -]
  int n_rays;          // tiles_x * tiles_y * 64
  int tiles_x;
  int bounce;          // current bounce index
  int light;           // current light index
  int last_bounce;     // 1: this is the last bounce of the sample
  int pos_array, dir_array;  // which arrays the march kernel works on
  int claims_per_wave;       // queue chunks a wave claims over a launch, on average
};

// pass 0: single pass; 1: cheap pass (parks rays that need the deep evaluation); 2: full pass over the parked list
hipError_t wf_launch_march_strict(const WfParams& W, bool preview, int pass, int blocks, hipStream_t stream);
hipError_t wf_launch_march_fast(const WfParams& W, bool preview, int pass, int blocks, hipStream_t stream);
hipError_t launch_combine(const KParams& P, hipStream_t stream);
// scratch: 64 counters per RM_ORDER_TILES_PER_GROUP tiles (rm_order_scratch_elems)
hipError_t launch_order(unsigned int* cost, unsigned int* order, unsigned int* scratch, int n, hipStream_t stream);
inline size_t rm_order_scratch_elems(long long tiles) { return (size_t)(64 * ((tiles + 4095) / 4096)); }
hipError_t launch_cull_build(const CullBuild& B, hipStream_t stream);
void pixel_grid(const KParams& P, int* gx, int* gy);  // workgroup grid the pixel kernel uses for this job
hipError_t wf_launch_shade_strict(const WfParams& W, hipStream_t stream);
hipError_t wf_launch_shade_fast(const WfParams& W, hipStream_t stream);
bool wf_kind_has_cost_classes(int kind);
hipError_t launch_assemble(const void* src, int parts, int max_rows, long long row_bytes, int H, int stripe_rows, void* dst, hipStream_t stream);
hipError_t launch_pack_rows(const float4* color, const float4* normal_dof, long long pixels, float4* out, hipStream_t stream);
hipError_t launch_present_rows(const float4* color, long long pixels, float brightness, uchar4* out, hipStream_t stream);
hipError_t launch_present(const float4* color, const float4* normal_dof, int W, int H, float brightness, uchar4* out, hipStream_t stream);
hipError_t launch_present_striped(const float4* color, const float4* normal_dof, int W, int H, float brightness, uchar4* out, int stripe_rows, int parts, int part,
                                  int local_rows, hipStream_t stream);
hipError_t wf_launch_stage(const WfParams& W, int stage, hipStream_t stream);
hipError_t launch_pixels_strict(const KParams& P, hipStream_t stream);
hipError_t launch_pixels_fast(const KParams& P, hipStream_t stream);
hipError_t launch_probe_strict(const ProbeParams& P, hipStream_t stream);
hipError_t launch_probe_fast(const ProbeParams& P, hipStream_t stream);
hipError_t launch_camera_rng(const RmUniforms& u, int W, int H, int what, int count, float* out, hipStream_t stream);
hipError_t launch_math_probe(int fn, const float* a, const float* b, int n, float* out, hipStream_t stream);
}  // namespace rm
