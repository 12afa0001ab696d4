// rm_api.hip -- the C ABI of libhip_raymarch.so (include/hip_raymarch.h).
//
// Host side only: contexts, scene upload/validation, framebuffers and the
// launches.  There is NO CPU path: without a GPU rm_ctx_create fails with
// RM_ERR_NO_DEVICE and nothing else can be called.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hip_raymarch.h"
#include "rm_params.hpp"

#ifndef RM_WITH_WAVEFRONT
#define RM_WITH_WAVEFRONT 0  // 1: the tests' cross-check build (see "the wavefront pipeline" below)
#endif

#define RM_SP_MAX 8

// rm_glstack.hip: the parity build in the GL stack's arithmetic, behind C entry points (its types live in another namespace)
extern "C" {
hipError_t rm_gl_launch_pixels(const void* kparams, hipStream_t stream);
hipError_t rm_gl_launch_probe(const void* probe_params, hipStream_t stream);
hipError_t rm_gl_launch_math_probe(int fn, const float* a, const float* b, int n, float* out, hipStream_t stream);
hipError_t rm_gl_launch_camera_rng(const RmUniforms* u, int W, int H, int what, int count, float* out, hipStream_t stream);
hipError_t rm_gl_launch_present(const float4* color, const float4* normal_dof, int W, int H, float brightness, uchar4* out, hipStream_t stream);
hipError_t rm_gl_launch_present_striped(const float4* color, const float4* normal_dof, int W, int H, float brightness, uchar4* out, int stripe_rows, int parts,
                                        int part, int local_rows, hipStream_t stream);
hipError_t rm_gl_launch_present_rows(const float4* color, long long pixels, float brightness, uchar4* out, hipStream_t stream);
hipError_t rm_gl_set_native_tan(int on, hipStream_t stream);
}

struct rm_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float retire_eps = 0.0f;  // opt-in (rm_ctx_set_retire_eps): any tolerance brightens lit pixels, see the header
  bool gl_stack = false;    // rm_ctx_set_gl_stack: strict-flag work runs in the GL stack's arithmetic (rm_glstack.hip)
  int cu_count = 256;
  // persistent-grid sizes in workgroups per CU, from a sweep on the headline frame (tools/sweep.sh):
  // the Mandelbulb passes want FEW waves (every wave ends in a tail of a few long rays), the table march wants all slots
  int pass2_blocks_per_cu = 2;
  int pass1_blocks_per_cu = 2;
  int pass2_rounds = 1;  // launches over the parked rays (the last one runs every ray to its end); >1 measured slower
  int repark = 24;       // a drained pass-2 wave with this many active lanes or fewer hands them to the next round
  // wavefront pipeline workspace (per-ray state + queue heads), grown on demand
  float4* ws = nullptr;
  size_t ws_rays = 0;
  unsigned int* heads = nullptr;  // 3 counters per march launch: head(pass 0/1), head(pass 2), parked count
  unsigned int* ws_list = nullptr;  // parked ray ids (two lists, ping-pong between pass-2 rounds)
  unsigned int* ws_list2 = nullptr;
  unsigned long long* stats = nullptr;  // 16 counters, filled by RM_WF_STATS builds only
  hipStream_t wf_stream[4] = {nullptr, nullptr, nullptr, nullptr};  // side streams of the banded wavefront pipeline
  hipEvent_t wf_join[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t wf_fork = nullptr;
  int wf_bands = 0;  // bands of rows in flight on side streams; 0 = automatic (2 for large tiles: measured best)
  int wf_blocks_per_cu = 8;
  int claims_per_wave = 8;
  // Samples in flight (pixel-kernel path, full mode): sample n renders on side stream n % depth into a staging
  // buffer and is blended into the planes, in order, by a small kernel on the context's stream; the next samples
  // render meanwhile.  One sample alone leaves the chip partly idle: a ray is a serial chain of ~2e5 instructions
  // (~1 ms), so every launch ends in a tail and a small shard never fills the SIMDs.
  int samples_in_flight = 3;
  hipStream_t sp_stream[RM_SP_MAX] = {};
  hipEvent_t sp_done[RM_SP_MAX] = {}, sp_free[RM_SP_MAX] = {};
  float4* sp_stage[RM_SP_MAX] = {};
  size_t sp_elems = 0;     // float4 elements of every slot's staging buffer (3 planes x tile pixels x samples of a batch)
  int last_pipeline = 0;   // rm_ctx_last_pipeline: what the last render call dispatched
  // Sample batch of rm_render_samples (KParams::batch): 0 = as many samples per launch as bring it to about
  // RM_BATCH_TARGET_TILES workgroups (a whole 4K frame has 16 320), 1 = one launch per sample, 2..RM_BATCH_MAX = fixed.
  int sample_batch = 0;
  unsigned int sp_next = 0;
  bool sp_ready = false;
  // Cost-ordered dispatch of the pixel kernel when samples run one at a time: the tiles of a job in the order of their
  // cost in the previous sample of the same job (same framebuffer window, tile and scene kind), most expensive first.
  // One set per stream the kernel runs on (the side streams of the samples in flight, and the context's stream): a
  // launch orders by the costs the previous launch ON ITS STREAM left, so no ordering between streams is needed.
  struct Lpt {
    unsigned int* cost = nullptr;
    unsigned int* order = nullptr;
    int capacity = 0;   // tiles the buffers hold
    long long key[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // the job the costs belong to
    bool have_cost = false;
    // Off the critical path (the context's own slot): two cost / order buffers used in turn.  The costs of launch n
    // are sorted on a side stream WHILE launch n + 1 runs, and launch n + 2 starts in that order -- tile costs hardly
    // change from one sample to the next, so an order that is one sample old is as good, and the sort's two small launches
    // (one workgroup) no longer stand between two launches.
    unsigned int* cost2 = nullptr;
    unsigned int* order2 = nullptr;
    hipEvent_t rendered[2] = {nullptr, nullptr}, sorted[2] = {nullptr, nullptr};
    unsigned long long launches = 0;  // of this job
  } lpt[RM_SP_MAX + 1];
  hipStream_t lpt_stream = nullptr;
  bool lpt_enabled = true;
  hipEvent_t switch_ev = nullptr;  // orders the old stream before the new one in rm_ctx_set_stream
  std::unordered_map<void*, size_t> buffers;  // rm_buffer_create: base address -> bytes
  uchar4* present_buf = nullptr;   // device staging of rm_present / rm_present_planes, grown on demand
  size_t present_cap = 0;          // pixels
  // rm_present_sharded (one process driving several GPUs): this context's rows of the payload, and on the context that
  // shows the frame the gathered parts and the frame in image order; grown on demand, freed with the context
  void* shard_rows = nullptr;  size_t shard_rows_cap = 0;    // this context's rows of the payload (packed float4 or RGBA8)
  void* shard_rows8 = nullptr; size_t shard_rows8_cap = 0;   // depth of field: this context's rows after ITS blur (RGBA8)
  void* shard_all = nullptr;   size_t shard_all_cap = 0;     // depth of field: every part's packed rows (the all-gather's receive side)
  void* shard_recv = nullptr;  size_t shard_recv_cap = 0;    // root: every part's RGBA8 rows
  void* shard_frame = nullptr; size_t shard_frame_cap = 0;   // depth of field: the packed frame in image order (every context); root: the canvas
  void* shard_canvas = nullptr; size_t shard_canvas_cap = 0; // root: the RGBA8 canvas in image order
  void* shard_host = nullptr;  size_t shard_host_cap = 0;    // root: pinned host copy of the canvas (rm_present_sharded_finish reads it)
  hipStream_t shard_stream = nullptr;                        // copies, assembly and blur of a present travel here, next to the renders
  hipEvent_t shard_snap = nullptr, shard_ev = nullptr, shard_done = nullptr;  // snapshot written; this context's part delivered; root: canvas on the host
  bool shard_pending = false;                                // root: a start without its finish
  int shard_w = 0, shard_h = 0;                              // root: the canvas of the pending present
  unsigned long long peer_enabled = 0;  // devices this context's GPU has been given peer access to
  // Culling grids of this context's scenes (scene_cull_grid): built once a scene has been asked for cull_min_pixels pixel-samples
  // (rm_ctx_set_cull_min_pixels), held within cull_budget bytes -- the least recently rendered scene gives its grid up first, and
  // renders on without one (same bits) -- and their buffers recycled: no hipMalloc / hipFree, which wait for the device, per scene.
  long long cull_min_pixels = 4ll << 20;
  size_t cull_budget = 0, cull_bytes = 0;
  unsigned long long cull_built = 0, use_clock = 0;
  std::vector<rm_scene*> cull_scenes;
  struct CullBuffer { unsigned long long* p; size_t bytes; bool idle; };  // idle: nothing enqueued anywhere still reads it
  std::vector<CullBuffer> cull_pool;
  // builds run on a stream of their own, NOT behind the context's: a new scene's grid is then built while the previous frame still
  // renders, and the new frame's render -- which waits for cull_event, on whichever stream it runs -- overlaps that frame's tail as
  // it would without a build (on the context's stream the build sat behind the previous frame's blend, i.e. behind its render)
  hipStream_t cull_stream = nullptr;
  hipEvent_t cull_event = nullptr, cull_order = nullptr;
  std::string error;
  std::string warning;  // rm_ctx_last_warning: advice that came with a call that SUCCEEDED (never an error)
};

struct rm_scene {
  rm_ctx* ctx = nullptr;
  DevScene dev{};
  RmPrim* d_prims = nullptr;
  RmSurface* d_surfaces = nullptr;
  unsigned long long* d_cull = nullptr;
  // the culling grid of a long CSG table is built once the scene has earned it (scene_cull_grid: rm_ctx_set_cull_min_pixels): a host that
  // creates many scenes it never renders, or shows a new one in every small frame, does not pay ((n^3 + (levels - 1) n_outer^3 + 1) x words
  // x 8 B -- 4.7 MB at 12 rows, 14 MB at 192 of hard operators; 31.5 MB per 64 rows of smooth unions, whose rule wants 128^3 cells near the
  // shapes -- and a build kernel per scene)
  bool cull_wanted = false;
  CullGrid cull_grid{};
  CullBuild cull_build{};
  size_t cull_bytes = 0;           // of d_cull
  long long px_seen = 0;           // pixel-samples asked of this scene while it had no grid
  unsigned long long last_use = 0; // rm_ctx::use_clock at the last render / probe
};

struct rm_fb {
  rm_ctx* ctx = nullptr;
  int width = 0, height = 0, row_begin = 0, row_count = 0;  // row_count = rows held by the planes
  int stripe_rows = 0, parts = 1, part = 0;                 // striped window when stripe_rows > 0
  float4* plane[3] = {nullptr, nullptr, nullptr};
  bool owned = false;
};

static thread_local std::string g_create_error;

static int fail(rm_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->error = msg;
  else g_create_error = msg;
  return code;
}

#define RM_HIP(ctx, expr)                                                                              \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return fail(ctx, RM_ERR_DEVICE, std::string(#expr ": ") + hipGetErrorString(e_)); \
  } while (0)

extern "C" {

int rm_abi_version(void) { return RM_ABI_VERSION; }

void rm_material_default(RmMaterial* m) {  // Validate.tsx:18-51
  std::memset(m, 0, sizeof *m);
  m->diffuse[0] = m->diffuse[1] = m->diffuse[2] = 0.6f;
  m->diffuse_cutoff = 35.0f;
  m->specular[0] = m->specular[1] = m->specular[2] = 0.6f;
  m->specular_cutoff = 35.0f;
  m->roughness = 0.2f;
  m->subsurface = 11111115.0f;
  m->subsurface_color[0] = m->subsurface_color[1] = m->subsurface_color[2] = 1.0f;
  m->ior = 100.0f;
  m->sky_color[0] = 0.7f;
  m->sky_color[1] = 0.8f;
  m->sky_color[2] = 1.0f;
  m->sky_floor = 0.2f;
  m->sky_scale = 2.0f;
  m->sky_radius = 36.0f;
  m->sky_axis = 1;
}

int rm_ctx_create(int device, rm_ctx** out) {
  if (!out) return fail(nullptr, RM_ERR_INVALID, "rm_ctx_create: out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, RM_ERR_NO_DEVICE, "no HIP device: libhip_raymarch has no CPU fallback (" +
                                               std::string(e == hipSuccess ? "device count 0" : hipGetErrorString(e)) + ")");
  if (device < 0 || device >= count) return fail(nullptr, RM_ERR_INVALID, "rm_ctx_create: device index out of range");
  rm_ctx* ctx = new (std::nothrow) rm_ctx();
  if (!ctx) return fail(nullptr, RM_ERR_DEVICE, "out of host memory");
  ctx->device = device;
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&ctx->switch_ev, hipEventDisableTiming)) != hipSuccess) {
    std::string msg = std::string("rm_ctx_create: ") + hipGetErrorString(e);
    delete ctx;
    return fail(nullptr, RM_ERR_DEVICE, msg);
  }
  ctx->stream = ctx->own_stream;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->cu_count = cus;
  if (const char* v = std::getenv("RM_PASS1_BLOCKS_PER_CU")) { int n = std::atoi(v); if (n >= 1 && n <= 8) ctx->pass1_blocks_per_cu = n; }
  if (const char* v = std::getenv("RM_PASS2_ROUNDS")) { int n = std::atoi(v); if (n >= 1 && n <= 3) ctx->pass2_rounds = n; }
  if (const char* v = std::getenv("RM_REPARK")) { int n = std::atoi(v); if (n >= 0 && n <= 63) ctx->repark = n; }
  if (const char* v = std::getenv("RM_COST_ORDER")) ctx->lpt_enabled = std::atoi(v) != 0;
  if (const char* v = std::getenv("RM_SAMPLES_IN_FLIGHT")) { int n = std::atoi(v); if (n >= 1 && n <= RM_SP_MAX) ctx->samples_in_flight = n; }
  if (const char* v = std::getenv("RM_WF_CLAIMS")) { int n = std::atoi(v); if (n >= 1 && n <= 64) ctx->claims_per_wave = n; }
  if (const char* v = std::getenv("RM_WF_BLOCKS_PER_CU")) { int n = std::atoi(v); if (n >= 1 && n <= 8) ctx->wf_blocks_per_cu = n; }
  if (const char* v = std::getenv("RM_WF_BANDS")) { int n = std::atoi(v); if (n >= 1 && n <= 8) ctx->wf_bands = n; }
  if (const char* v = std::getenv("RM_PASS2_BLOCKS_PER_CU")) { int n = std::atoi(v); if (n >= 1 && n <= 8) ctx->pass2_blocks_per_cu = n; }
  if (const char* v = std::getenv("RM_CULL_MIN_PIXELS")) { long long n = std::atoll(v); if (n >= 0) ctx->cull_min_pixels = n; }  // (the tests: 0, so that small renders go through the grid)
  {  // the culling grids of this context's scenes: a sixteenth of the device's memory, at most 1 GiB (a 256-row table's grid is 126 MB)
    size_t free_b = 0, total_b = 0;
    ctx->cull_budget = (size_t)1 << 30;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b / 16 < ctx->cull_budget) ctx->cull_budget = total_b / 16;
    else (void)hipGetLastError();
  }
  *out = ctx;
  return RM_OK;
}

void rm_ctx_destroy(rm_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->heads) (void)hipFree(ctx->heads);
  if (ctx->ws_list) (void)hipFree(ctx->ws_list);
  if (ctx->ws_list2) (void)hipFree(ctx->ws_list2);
  if (ctx->stats) (void)hipFree(ctx->stats);
  for (int s = 0; s < 4; s++) {
    if (ctx->wf_stream[s]) { (void)hipStreamSynchronize(ctx->wf_stream[s]); (void)hipStreamDestroy(ctx->wf_stream[s]); }
    if (ctx->wf_join[s]) (void)hipEventDestroy(ctx->wf_join[s]);
  }
  if (ctx->wf_fork) (void)hipEventDestroy(ctx->wf_fork);
  if (ctx->lpt_stream) { (void)hipStreamSynchronize(ctx->lpt_stream); (void)hipStreamDestroy(ctx->lpt_stream); }
  for (auto& l : ctx->lpt) {
    if (l.cost) (void)hipFree(l.cost);
    if (l.order) (void)hipFree(l.order);
    if (l.cost2) (void)hipFree(l.cost2);
    if (l.order2) (void)hipFree(l.order2);
    for (int k = 0; k < 2; k++) {
      if (l.rendered[k]) (void)hipEventDestroy(l.rendered[k]);
      if (l.sorted[k]) (void)hipEventDestroy(l.sorted[k]);
    }
  }
  for (int s = 0; s < RM_SP_MAX; s++) {
    if (ctx->sp_stream[s]) { (void)hipStreamSynchronize(ctx->sp_stream[s]); (void)hipStreamDestroy(ctx->sp_stream[s]); }
    if (ctx->sp_done[s]) (void)hipEventDestroy(ctx->sp_done[s]);
    if (ctx->sp_free[s]) (void)hipEventDestroy(ctx->sp_free[s]);
    if (ctx->sp_stage[s]) (void)hipFree(ctx->sp_stage[s]);
  }
  if (ctx->present_buf) (void)hipFree(ctx->present_buf);
  if (ctx->shard_rows) (void)hipFree(ctx->shard_rows);
  if (ctx->shard_recv) (void)hipFree(ctx->shard_recv);
  if (ctx->shard_frame) (void)hipFree(ctx->shard_frame);
  if (ctx->shard_rows8) (void)hipFree(ctx->shard_rows8);
  if (ctx->shard_all) (void)hipFree(ctx->shard_all);
  if (ctx->shard_canvas) (void)hipFree(ctx->shard_canvas);
  if (ctx->shard_host) (void)hipHostFree(ctx->shard_host);
  if (ctx->shard_stream) { (void)hipStreamSynchronize(ctx->shard_stream); (void)hipStreamDestroy(ctx->shard_stream); }
  if (ctx->shard_snap) (void)hipEventDestroy(ctx->shard_snap);
  if (ctx->shard_done) (void)hipEventDestroy(ctx->shard_done);
  if (ctx->shard_ev) (void)hipEventDestroy(ctx->shard_ev);
  if (ctx->cull_stream) { (void)hipStreamSynchronize(ctx->cull_stream); (void)hipStreamDestroy(ctx->cull_stream); }
  for (auto& b : ctx->cull_pool) (void)hipFree(b.p);
  if (ctx->cull_event) (void)hipEventDestroy(ctx->cull_event);
  if (ctx->cull_order) (void)hipEventDestroy(ctx->cull_order);
  for (auto& b : ctx->buffers) (void)hipFree(b.first);  // rm_buffer_create'd memory the host did not destroy
  if (ctx->switch_ev) (void)hipEventDestroy(ctx->switch_ev);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

const char* rm_last_error(const rm_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int rm_ctx_set_stream(rm_ctx* ctx, void* hip_stream) {
  if (!ctx) return RM_ERR_INVALID;
  hipStream_t next = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  if (next == ctx->stream) return RM_OK;
  RM_HIP(ctx, hipSetDevice(ctx->device));
  // Everything already queued on the old stream (the zeroing of rm_fb_create / rm_fb_clear, uploads, renders, the
  // blends of samples in flight) is ordered before whatever the caller enqueues on the new one: an event, no host wait.
  RM_HIP(ctx, hipEventRecord(ctx->switch_ev, ctx->stream));
  RM_HIP(ctx, hipStreamWaitEvent(next, ctx->switch_ev, 0));
  ctx->stream = next;
  return RM_OK;
}

int rm_ctx_set_samples_in_flight(rm_ctx* ctx, int n) {
  if (!ctx) return RM_ERR_INVALID;
  if (n < 1 || n > RM_SP_MAX) return fail(ctx, RM_ERR_INVALID, "rm_ctx_set_samples_in_flight: n must be in 1..8");
  if (ctx->sp_ready) RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->samples_in_flight = n;
  ctx->warning.clear();
  if (n > 1) {
    // every sample in flight renders on a side stream; the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES
    // hardware queues (default 4, read when the runtime starts) and streams that share a queue serialise
    const char* q = std::getenv("GPU_MAX_HW_QUEUES");
    const int queues = q ? std::atoi(q) : 4;
    if (queues < 8) {
      char msg[320];
      std::snprintf(msg, sizeof msg, "warning: %d samples in flight but GPU_MAX_HW_QUEUES is %s%d: the side streams will share hardware queues and "
                    "serialise (measured: 0.59 instead of 0.42 ms per sample on a 1/8 shard); export GPU_MAX_HW_QUEUES=8 before the process "
                    "touches the GPU (raymarching_engine_amd.native and js/index.js do)", n, q ? "" : "unset = ", queues);
      ctx->warning = msg;  // accepted all the same: RM_OK, and rm_last_error stays what it was -- the note goes to rm_ctx_last_warning
    }
  }
  return RM_OK;
}

const char* rm_ctx_last_warning(const rm_ctx* ctx) { return ctx ? ctx->warning.c_str() : ""; }

int rm_ctx_set_cost_order(rm_ctx* ctx, int on) {
  if (!ctx) return RM_ERR_INVALID;
  ctx->lpt_enabled = on != 0;
  for (auto& l : ctx->lpt) l.have_cost = false;
  return RM_OK;
}

int rm_ctx_set_retire_eps(rm_ctx* ctx, float eps) {
  if (!ctx) return RM_ERR_INVALID;
  if (!(eps >= 0.0f && eps <= 1e-3f)) return fail(ctx, RM_ERR_INVALID, "rm_ctx_set_retire_eps: eps must be in [0, 1e-3]");
  ctx->retire_eps = eps;
  return RM_OK;
}

int rm_ctx_set_gl_stack(rm_ctx* ctx, int on) {
  if (!ctx) return RM_ERR_INVALID;
  if (on != 0 && on != 1 && on != 2) return fail(ctx, RM_ERR_INVALID, "rm_ctx_set_gl_stack: 0 (off), 1 (on) or 2 (on, with the stack's own tan)");
  RM_HIP(ctx, hipSetDevice(ctx->device));
  RM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // samples in flight finish in the arithmetic they started in
  // The kernels read the stack's-own-tan switch from a __device__ symbol: one per DEVICE (the contexts of a device share it).
  // What each device holds is remembered per device, under a lock; the copy is synchronous (its source is this frame's).
  static std::mutex guard;
  static int native_tan[64];  // 0 after the module is loaded on a device, like the symbol
  if (on != 0) {
    std::lock_guard<std::mutex> lock(guard);
    const int slot = ctx->device & 63, v = on == 2 ? 1 : 0;
    if (native_tan[slot] != v) {
      RM_HIP(ctx, hipDeviceSynchronize());  // launches of the device's other contexts finish with the value they started with
      RM_HIP(ctx, rm_gl_set_native_tan(v, nullptr));
      native_tan[slot] = v;
    }
  }
  ctx->gl_stack = on != 0;
  return RM_OK;
}

int rm_debug_counters(rm_ctx* ctx, unsigned long long* out16, int reset) {
  if (!ctx || !out16) return RM_ERR_INVALID;
  for (int i = 0; i < 16; i++) out16[i] = 0;
  if (!ctx->stats) return RM_OK;
  RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RM_HIP(ctx, hipMemcpy(out16, ctx->stats, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost));
  if (reset) RM_HIP(ctx, hipMemset(ctx->stats, 0, sizeof(unsigned long long) * 16));
  return RM_OK;
}

int rm_ctx_last_pipeline(const rm_ctx* ctx) { return ctx ? ctx->last_pipeline : RM_PIPELINE_NONE; }

int rm_device_memory(rm_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
  if (!ctx || !free_bytes || !total_bytes) return fail(ctx, RM_ERR_INVALID, "rm_device_memory: NULL argument");
  RM_HIP(ctx, hipSetDevice(ctx->device));
  RM_HIP(ctx, hipMemGetInfo(free_bytes, total_bytes));
  return RM_OK;
}

int rm_sync(rm_ctx* ctx) {
  if (!ctx) return RM_ERR_INVALID;
  RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RM_OK;
}

// ---- scene -------------------------------------------------------------------

static bool finite_all(const float* p, int n) {
  for (int i = 0; i < n; i++)
    if (!std::isfinite(p[i])) return false;
  return true;
}

// Where an escaping ray of a primitive table ends (Sdf<RM_SCENE_TABLE>::far_jump).  Without domain rows every shape lies
// inside a sphere about the origin, and outside it the fold is bounded below: a shape's distance is >= |p| - (|c| + extent),
// min / max keep that bound (max(d, -di) and max(d, di) only raise d), and a CHAIN of smooth unions stays within k of it however
// long it is: one smooth union is at most k / 4 (1 - |t| / k)^2 below min(d, di), t = di - d, so once the running value is D below the
// bound a term above the bound (|t| > D) lowers it by at most k / 4 (1 - D / k)^2 more -- D + k / 4 (1 - D / k)^2 <= k on [0, k], and
// nothing at all from D = k on -- and a term below the bound resets D to at most k as well (checked numerically: 0.995 k under a
// greedy adversary, 0.94 k for 200 equal terms).  Round 3 first allowed k / 4 per smooth union: R' = 5.9 for CSG-64 instead of 2.95,
// and every escaping ray marched to twice the radius before its jump.  A ray out
// there that is not moving inward never returns and doubles its distance from step to step until |p|^2 overflows; at that
// step p - c = p for every shape (the centres are below the spacing of floats of that size), so every shape's distance is
// +Inf at once and the march's step is the fold of all-+Inf terms: +Inf -- the position becomes +-Inf by the sign of the
// direction components, a fixed point -- or NaN, when a smooth union meets Inf - Inf -- then every coordinate is NaN, for
// good.  That fold is evaluated here, with the operators' own semantics (IEEE minNum / maxNum drop a NaN, as v_min / v_max do).
static void table_far_field(const RmSceneDesc* desc, DevScene* dev) {
  dev->far_end = 0;
  dev->far_r2 = 0.0f;
  double reach = 0.0, kmax = 0.0;
  int smooth = 0;
  float d = 0.0f;
  bool first = true;
  const float inf = std::numeric_limits<float>::infinity();
  for (int i = 0; i < desc->nprims; i++) {
    const RmPrim& p = desc->prims[i];
    const int type = p.type & 0xff, op = (p.type >> 8) & 0xff;
    if (type == RM_PRIM_REPEAT || type == RM_PRIM_FOLD || type == RM_PRIM_KIND) return;  // a tiled or folded space has no far field; a kind's estimator is not a sphere's or a box's
    if (type > RM_PRIM_KIND || op > RM_OP_INTERSECT) return;  // ABI 8's shapes and smooth operators: no end state has been derived for them (a plane has no far field at all)
    const double c = std::sqrt((double)p.center[0] * p.center[0] + (double)p.center[1] * p.center[1] + (double)p.center[2] * p.center[2]);
    const double extent = type == RM_PRIM_SPHERE ? std::fabs((double)p.size[0])
                                                 : std::sqrt((double)p.size[0] * p.size[0] + (double)p.size[1] * p.size[1] + (double)p.size[2] * p.size[2]);
    reach = std::fmax(reach, c + extent);
    if (first) { d = inf; first = false; continue; }
    if (op == RM_OP_UNION) d = std::fmin(d, inf);
    else if (op == RM_OP_SMOOTH_UNION) { d = std::numeric_limits<float>::quiet_NaN(); smooth++; kmax = std::fmax(kmax, std::fabs((double)p.k)); }  // Inf - Inf, or a NaN handed on
    else if (op == RM_OP_SUBTRACT) d = std::fmax(d, -inf);
    else d = std::fmax(d, inf);
  }
  const double r = 2.0 * (reach + (smooth ? 1.01 * kmax : 0.0)) + 1.0;
  if (first || !(r < 1e9)) return;
  dev->far_r2 = (float)(r * r);
  dev->far_end = d == inf ? 1 : (d != d ? 2 : 0);
  // A ray that passes every shape at a distance (Sdf<RM_SCENE_TABLE>::clear_miss): if it stays K = k_max + b0 clear of every shape's
  // bounding sphere, the fold is >= b0 all along it (a chain of smooth unions is within k_max of the nearest term): the march never
  // settles.  How long can it take to leave?  Let rho hold every shape's sphere and K (rho = 1.15 (reach + k_max), b0 <= rho / 8).
  // Inside that sphere a step is >= b0, outside >= |x| - rho + b0; two steps bring a ray from up to 60 Rp away to it, and the worst
  // chord (through the centre) is crossed and the ray out at 2.6 rho -- beyond the jump's 2 Rp, moving outward -- within
  // 2 rho / (0.99 b0) + 6 steps (tools: the recurrence iterated over every closest approach).  Then the jump's first case: far_need's 71 steps.
  // So a march with `left` steps may take b0 = 2.05 rho / (left - 84): 128 steps -> b0 = rho / 21 (C4: 0.16, clearance 0.37).
  if (dev->far_end != 0 && reach + kmax >= 1.05) {
    dev->clear_k = (float)((smooth ? 1.01 * kmax : 0.0) + 1e-3 * (1.0 + reach));
    dev->clear_rho = (float)(1.15 * (reach + (smooth ? kmax : 0.0)));
  }
}

// The culling grid of a primitive table without domain rows (rm_params.hpp CullGrid; rm_kernels.inc rm_cull_build_kernel has the
// argument and fills it on the device; fast policy only): nested cubes of RM_CULL_N^3 cells about the shapes' bounding box, each
// twice as wide as the one before, out to where fp32 no longer tells the rows apart.
// For tables of at least RM_CULL_MIN_ROWS rows (round 3: those of mostly hard operators; round 4: every one, rm_params.hpp rm_cull_cell).
#ifndef RM_CULL_MIN_ROWS
#define RM_CULL_MIN_ROWS 12
#endif
#ifndef RM_CULL_N
#define RM_CULL_N 32
#endif
#ifndef RM_CULL_N_SMOOTH
#define RM_CULL_N_SMOOTH 128  // C4 12.4 ms without the grid, 9.8 with 64^3 cells, 9.1 with 128^3 (profiles/r04_smooth_union_culling.txt)
#endif
#ifndef RM_CULL_SMOOTH_LEVELS
#define RM_CULL_SMOOTH_LEVELS 8  // out to 128 scene widths (a camera further away folds every row until its rays get there)
#endif
#ifndef RM_CULL_OUTER_DIV
#define RM_CULL_OUTER_DIV 2  // the levels around the first: cells twice as wide (1: as fine as level 0 -- rounds 3-4, 134 MB per 64 rows instead of 31.5)
#endif
#ifndef RM_CULL_MAX_LEVELS
#define RM_CULL_MAX_LEVELS 18
#endif
static bool table_cull_params(const RmSceneDesc* desc, CullGrid* g, CullBuild* build) {
  *g = CullGrid{};
  *build = CullBuild{};
  const int n = desc->nprims;
  if (n < RM_CULL_MIN_ROWS) return false;
  if (const char* v = std::getenv("RM_NO_CULL"))
    if (v[0] == '1') return false;
  double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30}, kmax = 0.0;
  for (int i = 0; i < n; i++) {
    const RmPrim& p = desc->prims[i];
    const int type = p.type & 0xff;
    if (type != RM_PRIM_SPHERE && type != RM_PRIM_BOX) return false;  // domain rows: no grid in the table's own space (ABI 8's shapes: no rule)
    if (((p.type >> 8) & 0xff) > RM_OP_INTERSECT) return false;         // ... nor for its smooth subtraction / intersection
    for (int a = 0; a < 3; a++) {
      const double e = std::fabs((double)(type == RM_PRIM_SPHERE ? p.size[0] : p.size[a]));
      lo[a] = std::fmin(lo[a], p.center[a] - e);
      hi[a] = std::fmax(hi[a], p.center[a] + e);
    }
    if (((p.type >> 8) & 0xff) == RM_OP_SMOOTH_UNION) kmax = std::fmax(kmax, (double)p.k);
  }
  // a table of mostly smooth unions (CSG-64; round 4): their far rows are dropped where the rounding they perform is provably the
  // identity (rm_params.hpp rm_cull_cell) -- on a finer grid, the binade tests want small cells
  const bool smooth_spheres = rm_cull_mostly_smooth(desc->prims, n);
  if (smooth_spheres && n < RM_TABLE_BIG_ROWS) return false;  // about half of such a table's rows stay: too few to pay for the lookup
  double half = 0.0, reach = 0.0;
  for (int a = 0; a < 3; a++) {
    half = std::fmax(half, 0.5 * (hi[a] - lo[a]));
    reach = std::fmax(reach, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
    build->centre[a] = 0.5 * (lo[a] + hi[a]);
  }
  half = 1.1 * half + kmax + 1e-3;
  if (!(half < 1e6) || !(reach < 1e6)) return false;
  // levels: out to where the fp32 allowance of the build kernel (1.2e-7 (n + 8) |coordinates|) exceeds the scene's own width
  int levels = 1;
  while (levels < RM_CULL_MAX_LEVELS && 1.2e-7 * (n + 8) * 1.74 * std::ldexp(half, levels) < 2.0 * half) levels++;
  if (smooth_spheres && levels > RM_CULL_SMOOTH_LEVELS) levels = RM_CULL_SMOOTH_LEVELS;
  build->half0 = half;
  build->reach = reach;
  build->nprims = n;
  const int cells = smooth_spheres ? (n >= 32 ? RM_CULL_N_SMOOTH : RM_CULL_N_SMOOTH / 2) : RM_CULL_N;  // (17 MB instead of 134 for the shorter tables)
  build->n = cells;
  build->n_outer = smooth_spheres ? cells / RM_CULL_OUTER_DIV : cells;  // (the hard operators' sharper test works on large cells as it is)
  build->levels = levels;
  build->words = (n + 63) / 64;
  for (int a = 0; a < 3; a++) g->centre[a] = (float)build->centre[a];
  g->inv_half0 = (float)(1.0 / half);
  g->scale0 = (float)(cells / (2.0 * half));
  g->n = cells;
  g->n_outer = build->n_outer;
  g->levels = levels;
  g->words = build->words;
  return true;
}

int rm_debug_cull_cell(const RmSceneDesc* desc, const double* centre, double radius, double margin, unsigned long long* out_words) {
  if (!desc || !centre || !out_words || desc->kind != RM_SCENE_TABLE || desc->nprims < 1 || desc->nprims > RM_MAX_PRIMS || !desc->prims) return RM_ERR_INVALID;
  for (int i = 0; i < desc->nprims; i++) {
    const int type = desc->prims[i].type & 0xff;
    if ((type != RM_PRIM_SPHERE && type != RM_PRIM_BOX) || ((desc->prims[i].type >> 8) & 0xff) > RM_OP_INTERSECT) return RM_ERR_INVALID;
  }
  rm_cull_cell(desc->prims, desc->nprims, (desc->nprims + 63) / 64, centre, radius, margin, out_words);
  return RM_OK;
}

// The far field of the kaleidoscopic kinds (rm_device.hpp Sdf<RM_SCENE_KIFS_BOX>::far_jump, Sdf<RM_SCENE_KIFS_TREE>::eval).
// A level maps t to rotate(|t / s| - off): norms obey |t'| >= |t| / s - |off|, so after n levels |t_n| s^n >= |p| - |off| s / (1 - s),
// and sdBox(q, b) >= |q| - |b|: every level's box is >= |p| - R' with R' = |off| s / (1 - s) + |b| (b <= the unit-level box).
//  * rotation-fractal (one box at the last level): the estimate grows with |p| and a ray that leaves overflows like a table's;
//    far_end = 1 (the +-Inf pattern is a fixed point whatever the folds make of an infinite point: sdBox drops their NaNs and
//    returns +Inf or 0), provided the deepest point t_n = p / s^n stays finite up to the overflow of |p|^2 (s^n >= 1e-15).
//  * tree / smooth-tree: the estimate starts from min_dist = 9999, so beyond |p| = 9999 + R' it IS 9999 -- min(9999, box), and
//    the smooth union's mix(box, 9999, 1) = box + (9999 - box) is exact while box < 2^24 -- and a ray out there walks 9999 per
//    step to the end of its budget: far_end = 3 lets the fast evaluation return the constant without its levels.
static void kifs_far_field(const RmSceneDesc* desc, DevScene* dev) {
  dev->far_end = 0;
  dev->far_r2 = 0.0f;
  const double iters = desc->params[RM_P_KIFS_ITERATIONS], s = desc->params[RM_P_KIFS_SCALE], off = std::fabs((double)desc->params[RM_P_KIFS_OFFSET]);
  if (!(s > 0.05 && s <= 0.9) || !(iters >= 0.0 && iters <= 64.0) || !(off < 1e6)) return;
  if (desc->kind == RM_SCENE_KIFS_BOX) {
    if (std::pow(s, std::ceil(iters)) < 1e-15) return;
    const double r = 2.0 * std::sqrt(3.0) * (off * s / (1.0 - s) + 1.0) + 1.0;
    dev->far_r2 = (float)(r * r);
    dev->far_end = 1;
  } else {
    const double r = 9999.0 + 1.01 * (off * s / (1.0 - s) + 1.0) + 2.0;
    dev->far_r2 = (float)(r * r);
    dev->far_end = 3;
  }
}

int rm_scene_create(rm_ctx* ctx, const RmSceneDesc* desc, rm_scene** out) {
  if (!ctx || !desc || !out) return fail(ctx, RM_ERR_INVALID, "rm_scene_create: NULL argument");
  *out = nullptr;
  char buf[256];
  // the analogue of the GLSL compile: reject what the kernels cannot run, with an "info log"
  if (desc->kind < 0 || desc->kind >= RM_SCENE_KIND_COUNT) {
    std::snprintf(buf, sizeof buf, "scene: unknown kind %d", desc->kind);
    return fail(ctx, RM_ERR_INVALID, buf);
  }
  if (!finite_all(desc->params, 16)) return fail(ctx, RM_ERR_INVALID, "scene: non-finite parameter");
  if (desc->material.sky_axis < 0 || desc->material.sky_axis > 2) return fail(ctx, RM_ERR_INVALID, "scene: material.sky_axis must be 0, 1 or 2");
  if (desc->nsurfaces < 0 || desc->nsurfaces > RM_MAX_SURFACES || (desc->nsurfaces > 0 && (!desc->surfaces || desc->kind != RM_SCENE_TABLE))) {
    std::snprintf(buf, sizeof buf, "scene: 0..%d surfaces, for a primitive table (got %d)", RM_MAX_SURFACES, desc->nsurfaces);
    return fail(ctx, RM_ERR_INVALID, buf);
  }
  for (int i = 0; i < desc->nsurfaces; i++)
    if (!finite_all(reinterpret_cast<const float*>(&desc->surfaces[i]), 12)) return fail(ctx, RM_ERR_INVALID, "scene: non-finite surface value");
  if (desc->kind == RM_SCENE_TABLE) {
    if (desc->nprims < 1 || desc->nprims > RM_MAX_PRIMS || !desc->prims) {
      std::snprintf(buf, sizeof buf, "scene: primitive table needs 1..%d rows (got %d)", RM_MAX_PRIMS, desc->nprims);
      return fail(ctx, RM_ERR_INVALID, buf);
    }
    int shapes = 0, kind_of_rows = -1;
    for (int i = 0; i < desc->nprims; i++) {
      const RmPrim& p = desc->prims[i];
      const int type = p.type & 0xff, op = (p.type >> 8) & 0xff;
      if (type > RM_PRIM_PLANE || op > RM_OP_SMOOTH_INTERSECT || (p.type >> 24) != 0) {
        std::snprintf(buf, sizeof buf, "scene: row %d: unknown primitive/operator 0x%x", i, p.type);
        return fail(ctx, RM_ERR_INVALID, buf);
      }
      const int surface = (p.type >> 16) & 0xff;
      if (type == RM_PRIM_KIND) {
        const int kind = (int)p.size[0];
        if ((float)kind != p.size[0] || (kind != RM_SCENE_MANDELBULB && kind != RM_SCENE_SPHERE_LATTICE)) {
          std::snprintf(buf, sizeof buf, "scene: row %d: a kind row evaluates RM_SCENE_MANDELBULB or RM_SCENE_SPHERE_LATTICE (size[0] = %g)", i, (double)p.size[0]);
          return fail(ctx, RM_ERR_INVALID, buf);
        }
        if (kind_of_rows >= 0 && kind_of_rows != kind) return fail(ctx, RM_ERR_INVALID, "scene: the kind rows of a table evaluate ONE kind (its parameters are the scene's parameter block)");
        kind_of_rows = kind;
        if (kind == RM_SCENE_MANDELBULB && !(desc->params[RM_P_BULB_ITERATIONS] >= 0.0f && desc->params[RM_P_BULB_ITERATIONS] <= 64.0f))
          return fail(ctx, RM_ERR_INVALID, "scene: mandelbulb iterations must be in 0..64");
        if (kind == RM_SCENE_SPHERE_LATTICE && !(desc->params[RM_P_LATTICE_PERIOD] > 0.0f)) return fail(ctx, RM_ERR_INVALID, "scene: the lattice's period must be > 0");
      }
      const bool shape = type != RM_PRIM_REPEAT && type != RM_PRIM_FOLD;
      if (surface > desc->nsurfaces || (surface != 0 && !shape)) {
        std::snprintf(buf, sizeof buf, "scene: row %d: surface %d of %d (only shape rows name a surface)", i, surface, desc->nsurfaces);
        return fail(ctx, RM_ERR_INVALID, buf);
      }
      if (!finite_all(p.center, 3) || !finite_all(p.size, 3) || !std::isfinite(p.k)) {
        std::snprintf(buf, sizeof buf, "scene: row %d: non-finite value", i);
        return fail(ctx, RM_ERR_INVALID, buf);
      }
      if (type == RM_PRIM_REPEAT && !(p.size[0] > 0.0f && p.size[1] > 0.0f && p.size[2] > 0.0f)) {
        std::snprintf(buf, sizeof buf, "scene: row %d: repeat needs a period > 0 on every axis", i);
        return fail(ctx, RM_ERR_INVALID, buf);
      }
      if (type == RM_PRIM_FOLD && !(p.k > 0.0f)) {
        std::snprintf(buf, sizeof buf, "scene: row %d: fold needs scale > 0", i);
        return fail(ctx, RM_ERR_INVALID, buf);
      }
      if (shape && (op == RM_OP_SMOOTH_UNION || op == RM_OP_SMOOTH_SUBTRACT || op == RM_OP_SMOOTH_INTERSECT) && !(p.k > 0.0f) && shapes > 0) {
        std::snprintf(buf, sizeof buf, "scene: row %d: smooth %s needs k > 0", i, op == RM_OP_SMOOTH_UNION ? "union" : op == RM_OP_SMOOTH_SUBTRACT ? "subtraction" : "intersection");
        return fail(ctx, RM_ERR_INVALID, buf);
      }
      if ((type == RM_PRIM_TORUS || type == RM_PRIM_CYLINDER) && !(p.size[0] >= 0.0f && p.size[1] >= 0.0f)) {
        std::snprintf(buf, sizeof buf, "scene: row %d: a torus / cylinder needs radii (and a half height) >= 0", i);
        return fail(ctx, RM_ERR_INVALID, buf);
      }
      if (type == RM_PRIM_PLANE && !(p.size[0] != 0.0f || p.size[1] != 0.0f || p.size[2] != 0.0f)) {
        std::snprintf(buf, sizeof buf, "scene: row %d: a plane needs a normal", i);
        return fail(ctx, RM_ERR_INVALID, buf);
      }
      shapes += shape ? 1 : 0;
    }
    if (shapes == 0) return fail(ctx, RM_ERR_INVALID, "scene: the table has no shape row (sphere / box), only domain operators");
  } else if (desc->kind == RM_SCENE_MANDELBULB) {
    if (!(desc->params[RM_P_BULB_ITERATIONS] >= 0.0f && desc->params[RM_P_BULB_ITERATIONS] <= 64.0f))
      return fail(ctx, RM_ERR_INVALID, "scene: mandelbulb iterations must be in 0..64");
  } else if (desc->kind == RM_SCENE_SPHERE_GRID || desc->kind == RM_SCENE_MENGER || desc->kind == RM_SCENE_KIFS_TREE ||
             desc->kind == RM_SCENE_KIFS_BOX) {
    // one evaluation loops `iterations` times, a sample evaluates ~10^3 times per pixel: an unbounded count is an endless kernel
    const int slot = desc->kind == RM_SCENE_SPHERE_GRID ? RM_P_GRID_ITERATIONS : desc->kind == RM_SCENE_MENGER ? RM_P_MENGER_ITERATIONS : RM_P_KIFS_ITERATIONS;
    if (!(desc->params[slot] <= 64.0f)) return fail(ctx, RM_ERR_INVALID, "scene: fractal iterations must be <= 64");
  }
  rm_scene* s = new (std::nothrow) rm_scene();
  if (!s) return fail(ctx, RM_ERR_DEVICE, "out of host memory");
  s->ctx = ctx;
  s->dev.kind = desc->kind;
  s->dev.nprims = desc->kind == RM_SCENE_TABLE ? desc->nprims : 0;
  if (desc->kind == RM_SCENE_TABLE) {
    bool spheres_smooth = true, domain = false, boxes = false, surfaces = false, kinds = false, more = false;
    for (int i = 0; i < desc->nprims; i++) {
      const int type = desc->prims[i].type & 0xff, op = (desc->prims[i].type >> 8) & 0xff;
      if (((desc->prims[i].type >> 16) & 0xff) != 0) surfaces = true;
      if (type != RM_PRIM_SPHERE || (i > 0 && op != RM_OP_SMOOTH_UNION)) spheres_smooth = false;
      if (type == RM_PRIM_REPEAT || type == RM_PRIM_FOLD) domain = true;
      if (type == RM_PRIM_BOX || type == RM_PRIM_TORUS || type == RM_PRIM_CYLINDER || type == RM_PRIM_PLANE) boxes = true;  // (RM_TABLE_NO_BOXES is a promise about spheres)
      if (type == RM_PRIM_KIND) kinds = true;
      if (type > RM_PRIM_KIND || op > RM_OP_INTERSECT) more = true;
    }
    // (RM_TABLE_NO_BOXES promises a non-finite distance at a non-finite point: not said of a kind's estimator, so a kind row withdraws it)
    s->dev.table_flags = (spheres_smooth ? RM_TABLE_SPHERES_SMOOTH : 0) | (domain ? RM_TABLE_HAS_DOMAIN : 0) | (boxes || kinds ? 0 : RM_TABLE_NO_BOXES) |
                         (surfaces ? RM_TABLE_HAS_SURFACES : 0) | (kinds ? RM_TABLE_HAS_KIND : 0) | (more ? RM_TABLE_MORE : 0);
    if (spheres_smooth && desc->nprims >= 2 && desc->nprims * 3 <= RM_MAX_PRIMS * 2) {  // one smooth-union radius for the whole table (the usual case): it travels as a kernel argument, and a compact image of the rows fits behind them in LDS
      bool one_k = true;
      for (int i = 2; i < desc->nprims; i++) one_k = one_k && desc->prims[i].k == desc->prims[1].k;
      if (one_k) s->dev.table_flags |= RM_TABLE_UNIFORM_K;
    }
  }
  if (desc->kind == RM_SCENE_TABLE) table_far_field(desc, &s->dev);
  if (desc->kind == RM_SCENE_KIFS_BOX || desc->kind == RM_SCENE_KIFS_TREE) kifs_far_field(desc, &s->dev);
  if (desc->kind == RM_SCENE_SPHERE_GRID) {  // the sphere-grid fractal lives inside its big sphere (Sdf<RM_SCENE_SPHERE_GRID>::far_jump)
    const float* c = &desc->params[RM_P_GRID_CENTER];
    const double r = 2.0 * (std::sqrt((double)c[0] * c[0] + (double)c[1] * c[1] + (double)c[2] * c[2]) + std::fabs((double)desc->params[RM_P_GRID_BIG_SIZE])) + 1.0;
    if (r < 1e9) { s->dev.far_r2 = (float)(r * r); s->dev.far_end = 1; }
  }
  std::memcpy(s->dev.p, desc->params, sizeof s->dev.p);
  if (s->dev.table_flags & RM_TABLE_UNIFORM_K) {
    s->dev.p[0] = desc->prims[1].k;
    s->dev.p[1] = 0.5f * (1.0f / desc->prims[1].k);  // the bits the kernel's staging computes per row otherwise
  }
  s->dev.mat = desc->material;
  (void)hipSetDevice(ctx->device);
  if (s->dev.nprims > 0) {
    const size_t bytes = sizeof(RmPrim) * (size_t)s->dev.nprims;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_prims), bytes);
    if (e == hipSuccess) e = hipMemcpy(s->d_prims, desc->prims, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      if (s->d_prims) (void)hipFree(s->d_prims);
      delete s;
      return fail(ctx, RM_ERR_DEVICE, std::string("rm_scene_create: ") + hipGetErrorString(e));
    }
    s->dev.prims = s->d_prims;
  }
  if (desc->kind == RM_SCENE_TABLE) {
    CullGrid g;
    CullBuild build;
    if (table_cull_params(desc, &g, &build)) {  // built on first use: scene_cull_grid
      s->cull_wanted = true;
      s->cull_grid = g;
      s->cull_build = build;
    }
  }
  if (s->dev.table_flags & RM_TABLE_HAS_SURFACES) {  // entry 0 = the scene's own material block, then the surfaces as given
    RmSurface all[RM_MAX_SURFACES + 1];
    const RmMaterial& m = desc->material;
    all[0] = RmSurface{{m.diffuse[0], m.diffuse[1], m.diffuse[2]}, m.roughness, {m.specular[0], m.specular[1], m.specular[2]}, m.subsurface,
                       {m.subsurface_color[0], m.subsurface_color[1], m.subsurface_color[2]}, m.ior};
    for (int i = 0; i < desc->nsurfaces; i++) all[i + 1] = desc->surfaces[i];
    const size_t bytes = sizeof(RmSurface) * (size_t)(desc->nsurfaces + 1);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_surfaces), bytes);
    if (e == hipSuccess) e = hipMemcpy(s->d_surfaces, all, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      if (s->d_surfaces) (void)hipFree(s->d_surfaces);
      if (s->d_prims) (void)hipFree(s->d_prims);
      delete s;
      return fail(ctx, RM_ERR_DEVICE, std::string("rm_scene_create: ") + hipGetErrorString(e));
    }
    s->dev.surfaces = s->d_surfaces;
    s->dev.nsurfaces = desc->nsurfaces;
  }
  *out = s;
  return RM_OK;
}

static void cull_release(rm_ctx* ctx, rm_scene* s, bool idle);
void rm_scene_destroy(rm_scene* scene) {
  if (!scene) return;
  (void)hipSetDevice(scene->ctx->device);
  (void)hipStreamSynchronize(scene->ctx->stream);
  if (scene->d_prims) (void)hipFree(scene->d_prims);
  if (scene->d_surfaces) (void)hipFree(scene->d_surfaces);
  cull_release(scene->ctx, scene, true);  // (to the context's pool: the next scene's grid is usually the same size; the stream has been waited for above)
  delete scene;
}

// ---- framebuffers --------------------------------------------------------------

static int fb_check(rm_ctx* ctx, int width, int height, int row_begin, int row_count) {
  if (width < 1 || height < 1 || width > 65536 || height > 65536) return fail(ctx, RM_ERR_INVALID, "framebuffer: size must be 1..65536");
  // ray and tile counts are 32-bit in the kernels' index arithmetic (8x8 tiles round a frame up by < 2 %)
  if ((long long)width * (long long)height > (1ll << 28)) return fail(ctx, RM_ERR_INVALID, "framebuffer: width x height must be <= 2^28 pixels");
  if (row_begin < 0 || row_count < 1 || row_begin + row_count > height) return fail(ctx, RM_ERR_INVALID, "framebuffer: row window outside the image");
  return RM_OK;
}

int rm_fb_create(rm_ctx* ctx, int width, int height, int row_begin, int row_count, rm_fb** out) {
  if (!ctx || !out) return fail(ctx, RM_ERR_INVALID, "rm_fb_create: NULL argument");
  *out = nullptr;
  if (int rc = fb_check(ctx, width, height, row_begin, row_count)) return rc;
  rm_fb* fb = new (std::nothrow) rm_fb();
  if (!fb) return fail(ctx, RM_ERR_DEVICE, "out of host memory");
  fb->ctx = ctx;
  fb->width = width; fb->height = height; fb->row_begin = row_begin; fb->row_count = row_count;
  fb->owned = true;
  (void)hipSetDevice(ctx->device);
  const size_t bytes = sizeof(float4) * (size_t)width * (size_t)row_count;
  for (int i = 0; i < 3; i++) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&fb->plane[i]), bytes);
    if (e == hipSuccess) e = hipMemsetAsync(fb->plane[i], 0, bytes, ctx->stream);
    if (e != hipSuccess) {
      for (int j = 0; j <= i; j++)
        if (fb->plane[j]) (void)hipFree(fb->plane[j]);
      delete fb;
      return fail(ctx, RM_ERR_DEVICE, std::string("rm_fb_create: ") + hipGetErrorString(e));
    }
  }
  *out = fb;
  return RM_OK;
}

// rows r < y owned by a striped framebuffer
static int striped_rows_below(int y, int stripe, int parts, int part) {
  const int period = stripe * parts;
  const int q = y / period, rem = y % period;
  int in = rem - part * stripe;
  in = in < 0 ? 0 : (in > stripe ? stripe : in);
  return q * stripe + in;
}

int rm_fb_create_striped(rm_ctx* ctx, int width, int height, int stripe_rows, int parts, int part, void* color,
                         void* normal_dof, void* albedo_depth, rm_fb** out) {
  if (!ctx || !out) return fail(ctx, RM_ERR_INVALID, "rm_fb_create_striped: NULL argument");
  *out = nullptr;
  if (int rc = fb_check(ctx, width, height, 0, 1)) return rc;
  if (stripe_rows < 1 || parts < 1 || part < 0 || part >= parts) return fail(ctx, RM_ERR_INVALID, "rm_fb_create_striped: need stripe_rows >= 1 and 0 <= part < parts");
  if ((normal_dof == nullptr) != (albedo_depth == nullptr) || (!color && normal_dof)) return fail(ctx, RM_ERR_INVALID, "rm_fb_create_striped: give all planes, colour only, or none");
  if ((reinterpret_cast<uintptr_t>(color) | reinterpret_cast<uintptr_t>(normal_dof) | reinterpret_cast<uintptr_t>(albedo_depth)) & 15u)
    return fail(ctx, RM_ERR_INVALID, "rm_fb_create_striped: planes must be 16-byte aligned");
  const int rows = striped_rows_below(height, stripe_rows, parts, part);
  if (rows < 1) return fail(ctx, RM_ERR_INVALID, "rm_fb_create_striped: this part holds no rows");
  rm_fb* fb = new (std::nothrow) rm_fb();
  if (!fb) return fail(ctx, RM_ERR_DEVICE, "out of host memory");
  fb->ctx = ctx;
  fb->width = width; fb->height = height; fb->row_begin = 0; fb->row_count = rows;
  fb->stripe_rows = stripe_rows; fb->parts = parts; fb->part = part;
  if (color) {
    fb->plane[0] = static_cast<float4*>(color);
    fb->plane[1] = static_cast<float4*>(normal_dof);
    fb->plane[2] = static_cast<float4*>(albedo_depth);
    fb->owned = false;
  } else {
    fb->owned = true;
    (void)hipSetDevice(ctx->device);
    const size_t bytes = sizeof(float4) * (size_t)width * (size_t)rows;
    for (int i = 0; i < 3; i++) {
      hipError_t e = hipMalloc(reinterpret_cast<void**>(&fb->plane[i]), bytes);
      if (e == hipSuccess) e = hipMemsetAsync(fb->plane[i], 0, bytes, ctx->stream);
      if (e != hipSuccess) {
        for (int j = 0; j <= i; j++)
          if (fb->plane[j]) (void)hipFree(fb->plane[j]);
        delete fb;
        return fail(ctx, RM_ERR_DEVICE, std::string("rm_fb_create_striped: ") + hipGetErrorString(e));
      }
    }
  }
  *out = fb;
  return RM_OK;
}

int rm_fb_rows(const rm_fb* fb) { return fb ? fb->row_count : 0; }
int rm_fb_width(const rm_fb* fb) { return fb ? fb->width : 0; }
int rm_fb_height(const rm_fb* fb) { return fb ? fb->height : 0; }

int rm_fb_wrap(rm_ctx* ctx, int width, int height, int row_begin, int row_count, void* color, void* normal_dof,
               void* albedo_depth, rm_fb** out) {
  if (!ctx || !out || !color) return fail(ctx, RM_ERR_INVALID, "rm_fb_wrap: NULL argument");
  *out = nullptr;
  if (int rc = fb_check(ctx, width, height, row_begin, row_count)) return rc;
  if ((normal_dof == nullptr) != (albedo_depth == nullptr)) return fail(ctx, RM_ERR_INVALID, "rm_fb_wrap: give both G-buffer planes or neither");
  if ((reinterpret_cast<uintptr_t>(color) | reinterpret_cast<uintptr_t>(normal_dof) | reinterpret_cast<uintptr_t>(albedo_depth)) & 15u)
    return fail(ctx, RM_ERR_INVALID, "rm_fb_wrap: planes must be 16-byte aligned");
  rm_fb* fb = new (std::nothrow) rm_fb();
  if (!fb) return fail(ctx, RM_ERR_DEVICE, "out of host memory");
  fb->ctx = ctx;
  fb->width = width; fb->height = height; fb->row_begin = row_begin; fb->row_count = row_count;
  fb->plane[0] = static_cast<float4*>(color);
  fb->plane[1] = static_cast<float4*>(normal_dof);
  fb->plane[2] = static_cast<float4*>(albedo_depth);
  fb->owned = false;
  *out = fb;
  return RM_OK;
}

int rm_fb_clear(rm_fb* fb) {
  if (!fb) return RM_ERR_INVALID;
  rm_ctx* ctx = fb->ctx;
  const size_t bytes = sizeof(float4) * (size_t)fb->width * (size_t)fb->row_count;
  for (int i = 0; i < 3; i++)
    if (fb->plane[i]) RM_HIP(ctx, hipMemsetAsync(fb->plane[i], 0, bytes, ctx->stream));
  return RM_OK;
}

void rm_fb_destroy(rm_fb* fb) {
  if (!fb) return;
  (void)hipSetDevice(fb->ctx->device);
  (void)hipStreamSynchronize(fb->ctx->stream);
  if (fb->owned)
    for (int i = 0; i < 3; i++)
      if (fb->plane[i]) (void)hipFree(fb->plane[i]);
  delete fb;
}

int rm_fb_download(rm_fb* fb, int plane, float* host) {
  if (!fb || !host || plane < 0 || plane > 2) return fb ? fail(fb->ctx, RM_ERR_INVALID, "rm_fb_download: bad argument") : RM_ERR_INVALID;
  rm_ctx* ctx = fb->ctx;
  if (!fb->plane[plane]) return fail(ctx, RM_ERR_INVALID, "rm_fb_download: this framebuffer has no such plane");
  const size_t bytes = sizeof(float4) * (size_t)fb->width * (size_t)fb->row_count;
  RM_HIP(ctx, hipMemcpyAsync(host, fb->plane[plane], bytes, hipMemcpyDeviceToHost, ctx->stream));
  RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RM_OK;
}

int rm_fb_upload(rm_fb* fb, int plane, const float* host) {
  if (!fb || !host || plane < 0 || plane > 2) return fb ? fail(fb->ctx, RM_ERR_INVALID, "rm_fb_upload: bad argument") : RM_ERR_INVALID;
  rm_ctx* ctx = fb->ctx;
  if (!fb->plane[plane]) return fail(ctx, RM_ERR_INVALID, "rm_fb_upload: this framebuffer has no such plane");
  const size_t bytes = sizeof(float4) * (size_t)fb->width * (size_t)fb->row_count;
  RM_HIP(ctx, hipMemcpyAsync(fb->plane[plane], host, bytes, hipMemcpyHostToDevice, ctx->stream));
  RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RM_OK;
}

void* rm_fb_device_ptr(rm_fb* fb, int plane) { return (fb && plane >= 0 && plane <= 2) ? fb->plane[plane] : nullptr; }

// ---- raw device memory for hosts without an allocator of their own ---------------------

int rm_buffer_create(rm_ctx* ctx, size_t bytes, void** device_ptr) {
  if (!ctx || !device_ptr || bytes == 0 || bytes > ((size_t)1 << 36)) return fail(ctx, RM_ERR_INVALID, "rm_buffer_create: NULL argument or a size outside 1 .. 2^36 bytes");
  RM_HIP(ctx, hipSetDevice(ctx->device));
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return fail(ctx, RM_ERR_DEVICE, "rm_buffer_create: out of device memory");
  hipError_t e = hipMemsetAsync(p, 0, bytes, ctx->stream);
  if (e != hipSuccess) { (void)hipFree(p); return fail(ctx, RM_ERR_DEVICE, std::string("rm_buffer_create: ") + hipGetErrorString(e)); }
  ctx->buffers[p] = bytes;
  *device_ptr = p;
  return RM_OK;
}

int rm_buffer_destroy(rm_ctx* ctx, void* device_ptr) {
  if (!ctx) return RM_ERR_INVALID;
  if (!device_ptr) return RM_OK;
  auto it = ctx->buffers.find(device_ptr);
  if (it == ctx->buffers.end()) return fail(ctx, RM_ERR_INVALID, "rm_buffer_destroy: not a buffer of this context");
  RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->buffers.erase(it);
  RM_HIP(ctx, hipFree(device_ptr));
  return RM_OK;
}

// the copies take the buffer's base address and at most its size
static int buffer_check(rm_ctx* ctx, const void* device_ptr, const void* host, size_t bytes, const char* what) {
  if (!ctx || !device_ptr || !host) return fail(ctx, RM_ERR_INVALID, std::string(what) + ": NULL argument");
  auto it = ctx->buffers.find(const_cast<void*>(device_ptr));
  if (it == ctx->buffers.end()) return fail(ctx, RM_ERR_INVALID, std::string(what) + ": not a buffer of this context");
  if (bytes > it->second) return fail(ctx, RM_ERR_INVALID, std::string(what) + ": more bytes than the buffer holds");
  return RM_OK;
}

int rm_buffer_download(rm_ctx* ctx, const void* device_ptr, void* host, size_t bytes) {
  if (int rc = buffer_check(ctx, device_ptr, host, bytes, "rm_buffer_download")) return rc;
  RM_HIP(ctx, hipMemcpyAsync(host, device_ptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
  RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RM_OK;
}

int rm_buffer_upload(rm_ctx* ctx, void* device_ptr, const void* host, size_t bytes) {
  if (int rc = buffer_check(ctx, device_ptr, host, bytes, "rm_buffer_upload")) return rc;
  RM_HIP(ctx, hipMemcpyAsync(device_ptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RM_OK;
}

// The culling grid of a scene that has one coming (rm_scene_create), before a call that reads it (either build; not the GL stack's
// arithmetic).  Round 5: built once the scene has been asked for ctx->cull_min_pixels pixel-samples (until then it renders without
// one: the same bits, and a host that shows a new scene every frame at a small size never pays for a grid it would not earn back);
// the build kernel is enqueued on a stream of its own, the render that triggered it waits for it on the device -- no wait on the host -- into a
// buffer recycled from the context's pool; the grids of a context stay within its budget, the least recently used scene giving its
// grid up first.  Running out of memory is not an error (the fold of every row gives the same bits); any other failure is.
// The pool: buffers of grids that were given up, kept for the next build of the same size (a host that shows a new scene in every frame
// builds a grid per frame: no hipMalloc / hipFree each time).  At most 4 buffers and half the budget, and -- since round 6 -- the
// pooled bytes COUNT against the budget: grids held + buffers pooled <= rm_ctx_set_cull_budget at every moment (cull_trim_pool).
static size_t cull_pooled_bytes(const rm_ctx* ctx) {
  size_t pooled = 0;
  for (auto& b : ctx->cull_pool) pooled += b.bytes;
  return pooled;
}
static void cull_trim_pool(rm_ctx* ctx, size_t room_for) {  // frees pooled buffers until grids + pool + room_for fit the budget
  while (!ctx->cull_pool.empty() && ctx->cull_bytes + cull_pooled_bytes(ctx) + room_for > ctx->cull_budget) {
    (void)hipFree(ctx->cull_pool.back().p);  // (waits for the device)
    ctx->cull_pool.pop_back();
  }
}
static void cull_release(rm_ctx* ctx, rm_scene* s, bool idle) {  // the scene's grid back to the pool; idle: the caller has waited for the renders that read it
  if (!s->d_cull) return;
  const size_t pooled = cull_pooled_bytes(ctx);
  if (ctx->cull_pool.size() < 4 && pooled + s->cull_bytes <= ctx->cull_budget / 2) ctx->cull_pool.push_back({s->d_cull, s->cull_bytes, idle});
  else (void)hipFree(s->d_cull);  // (waits for the device)
  ctx->cull_bytes -= s->cull_bytes;
  for (size_t i = 0; i < ctx->cull_scenes.size(); i++)
    if (ctx->cull_scenes[i] == s) { ctx->cull_scenes.erase(ctx->cull_scenes.begin() + (long)i); break; }
  s->d_cull = nullptr;
  s->cull_bytes = 0;
  s->dev.cull.cells = nullptr;
}

static int scene_cull_grid(rm_ctx* ctx, rm_scene* s, int flags, long long pixels) {
  s->last_use = ++ctx->use_clock;
  if (!s->cull_wanted || (flags & RM_RENDER_NO_CULL) || (ctx->gl_stack && !(flags & RM_RENDER_FAST))) return RM_OK;
  s->px_seen += pixels;
  if (s->px_seen < ctx->cull_min_pixels) return RM_OK;
  CullGrid g = s->cull_grid;
  CullBuild build = s->cull_build;
  const size_t bytes = (size_t)(rm_cull_cells(g.n, g.n_outer, g.levels) + 1) * (size_t)g.words * sizeof(unsigned long long);
  if (bytes > ctx->cull_budget) return RM_OK;  // (still wanted: a host may raise the budget -- rm_ctx_set_cull_budget -- and the next render builds it)
  s->cull_wanted = false;
  RM_HIP(ctx, hipSetDevice(ctx->device));
  while (ctx->cull_bytes + bytes > ctx->cull_budget && !ctx->cull_scenes.empty()) {  // the least recently used grid goes (a kernel still
    rm_scene* victim = ctx->cull_scenes[0];                                            // reading it is ahead of the next build on this stream)
    for (rm_scene* c : ctx->cull_scenes)
      if (c->last_use < victim->last_use) victim = c;
    cull_release(ctx, victim, false);
    victim->cull_wanted = true;  // ... and may earn it back
    victim->px_seen = 0;
  }
  unsigned long long* cells = nullptr;
  bool idle = true;
  for (size_t i = 0; i < ctx->cull_pool.size(); i++)
    if (ctx->cull_pool[i].bytes == bytes) { cells = ctx->cull_pool[i].p; idle = ctx->cull_pool[i].idle; ctx->cull_pool.erase(ctx->cull_pool.begin() + (long)i); break; }
  if (!cells) {
    cull_trim_pool(ctx, bytes);  // the pooled buffers count against the budget too
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&cells), bytes);
    if (e == hipErrorOutOfMemory) {  // give the pool back and try once more
      (void)hipGetLastError();
      for (auto& b : ctx->cull_pool) (void)hipFree(b.p);
      ctx->cull_pool.clear();
      e = hipMalloc(reinterpret_cast<void**>(&cells), bytes);
    }
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return RM_OK; }
    if (e != hipSuccess) return fail(ctx, RM_ERR_DEVICE, std::string("culling grid: ") + hipGetErrorString(e));
  }
  build.prims = s->d_prims;
  build.cells = cells;
  hipError_t e = hipSuccess;
  if (!ctx->cull_stream) e = hipStreamCreateWithFlags(&ctx->cull_stream, hipStreamNonBlocking);
  if (e == hipSuccess && !ctx->cull_event) e = hipEventCreateWithFlags(&ctx->cull_event, hipEventDisableTiming);
  if (e == hipSuccess && !ctx->cull_order) e = hipEventCreateWithFlags(&ctx->cull_order, hipEventDisableTiming);
  if (e == hipSuccess && !idle) {  // a grid given up for the budget's sake: renders enqueued before now may still read it -- every one of them is
    e = hipEventRecord(ctx->cull_order, ctx->stream);  // followed by its blend on the context's stream, so behind this point they are done
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->cull_stream, ctx->cull_order, 0);
  }
  if (e == hipSuccess) e = rm::launch_cull_build(build, ctx->cull_stream);
  if (e == hipSuccess) e = hipEventRecord(ctx->cull_event, ctx->cull_stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->cull_event, 0);  // what runs on the context's own stream; the side streams wait where they launch
  if (e != hipSuccess) {
    (void)hipFree(cells);
    return fail(ctx, RM_ERR_DEVICE, std::string("culling grid: ") + hipGetErrorString(e));
  }
  s->d_cull = cells;
  s->cull_bytes = bytes;
  g.cells = cells;
  s->dev.cull = g;
  ctx->cull_bytes += bytes;
  ctx->cull_built++;
  ctx->cull_scenes.push_back(s);
  return RM_OK;
}

int rm_ctx_set_cull_min_pixels(rm_ctx* ctx, long long pixels) {
  if (!ctx) return RM_ERR_INVALID;
  if (pixels < 0) return fail(ctx, RM_ERR_INVALID, "rm_ctx_set_cull_min_pixels: pixels must be >= 0");
  ctx->cull_min_pixels = pixels;
  return RM_OK;
}

int rm_ctx_set_cull_budget(rm_ctx* ctx, size_t bytes) {
  if (!ctx) return RM_ERR_INVALID;
  ctx->cull_budget = bytes;
  // a lowered budget takes effect now: the least recently rendered scenes give their grids up (and may earn them back), then the pool shrinks
  RM_HIP(ctx, hipSetDevice(ctx->device));
  while (ctx->cull_bytes > ctx->cull_budget && !ctx->cull_scenes.empty()) {
    rm_scene* victim = ctx->cull_scenes[0];
    for (rm_scene* c : ctx->cull_scenes)
      if (c->last_use < victim->last_use) victim = c;
    RM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (renders that read the grid are followed by their blend on this stream)
    for (int i = 0; i < RM_SP_MAX; i++)
      if (ctx->sp_stream[i]) RM_HIP(ctx, hipStreamSynchronize(ctx->sp_stream[i]));
    cull_release(ctx, victim, true);
    victim->cull_wanted = true;
    victim->px_seen = 0;
  }
  cull_trim_pool(ctx, 0);
  return RM_OK;
}

int rm_ctx_cull_stats(const rm_ctx* ctx, unsigned long long* out4) {
  if (!ctx || !out4) return RM_ERR_INVALID;
  out4[0] = ctx->cull_built;
  out4[1] = (unsigned long long)ctx->cull_bytes;  // (the grids scenes hold; recycled buffers waiting in the pool are extra, and within the budget with them)
  out4[2] = (unsigned long long)ctx->cull_scenes.size();
  out4[3] = (unsigned long long)ctx->cull_budget;
  return RM_OK;
}

// ---- the hot path ---------------------------------------------------------------

static int build_params(rm_ctx* ctx, rm_scene* scene, rm_fb* fb, const RmUniforms* u, const RmRect* tile, int flags,
                        KParams* P, bool* empty) {
  if (!ctx || !scene || !fb || !u) return fail(ctx, RM_ERR_INVALID, "render: NULL argument");
  if (scene->ctx != ctx || fb->ctx != ctx) return fail(ctx, RM_ERR_INVALID, "render: scene/framebuffer belong to another context");
  if (u->renderMode != 0 && u->renderMode != 1) return fail(ctx, RM_ERR_INVALID, "render: renderMode must be 0 (full) or 1 (preview)");
  if (!RM_WITH_WAVEFRONT && (flags & RM_RENDER_WAVEFRONT))
    return fail(ctx, RM_ERR_INVALID, "render: RM_RENDER_WAVEFRONT -- this library is built without the wavefront pipeline (it is the tests' second implementation: "
                                     "tests/_xcheck/libhip_raymarch_xcheck.so); the product's own cross-check is RM_RENDER_NO_FAR_JUMP | RM_RENDER_NO_CULL");
  if (!(u->reflections >= 0.0f && u->reflections <= (float)RM_MAX_BOUNCES)) return fail(ctx, RM_ERR_INVALID, "render: reflections must be in 0..10 (raymarchingStepCountsArray[10])");
  if (u->lightCount < 0 || u->lightCount > RM_MAX_LIGHTS) return fail(ctx, RM_ERR_INVALID, "render: lightCount must be in 0..10");
  // `for (float i = 0.; i < steps; i++)` (raymarcher.frag:165, :210) never ends once i stops growing at 2^24
  for (int b = 0; b < RM_MAX_BOUNCES; b++)
    if (u->raymarchingStepCountsArray[b] > 1048576.0f) return fail(ctx, RM_ERR_INVALID, "render: a step count above 2^20");
  const bool color_only = (flags & RM_RENDER_COLOR_ONLY) != 0;
  if (!color_only && u->renderMode == 0 && (!fb->plane[1] || !fb->plane[2]))
    return fail(ctx, RM_ERR_INVALID, "render: framebuffer has no G-buffer planes; pass RM_RENDER_COLOR_ONLY");
  RmRect t = tile ? *tile : RmRect{0, 0, fb->width, fb->height};
  // clip to the image, then to the rows this framebuffer holds (as LOCAL row indices)
  const int x0 = t.x < 0 ? 0 : t.x, x1 = t.x + t.w > fb->width ? fb->width : t.x + t.w;
  int y0 = t.y < 0 ? 0 : t.y, y1 = t.y + t.h > fb->height ? fb->height : t.y + t.h;
  if (y1 < y0) y1 = y0;
  int l0, l1;
  if (fb->stripe_rows > 0) {
    l0 = striped_rows_below(y0, fb->stripe_rows, fb->parts, fb->part);
    l1 = striped_rows_below(y1, fb->stripe_rows, fb->parts, fb->part);
  } else {
    l0 = (y0 < fb->row_begin ? fb->row_begin : y0) - fb->row_begin;
    l1 = (y1 > fb->row_begin + fb->row_count ? fb->row_begin + fb->row_count : y1) - fb->row_begin;
  }
  *empty = x1 <= x0 || l1 <= l0;
  // what the scene has been asked for, in pixel-samples of the JOB: a striped framebuffer holds one part of `parts` of the frame, and the
  // other ranks render the rest of it (each with a grid of its own to earn: the threshold is the same for a sharded job as for a whole one)
  if (int rc = scene_cull_grid(ctx, scene, flags, *empty ? 0ll : (long long)(x1 - x0) * (long long)(l1 - l0) * (fb->stripe_rows > 0 ? (long long)fb->parts : 1ll))) return rc;
  P->u = *u;
  P->scene = scene->dev;
  P->color = fb->plane[0];
  P->normal_dof = color_only ? nullptr : fb->plane[1];
  P->albedo_depth = color_only ? nullptr : fb->plane[2];
  P->W = fb->width; P->H = fb->height; P->row_begin = fb->row_begin;
  P->stripe_rows = fb->stripe_rows; P->parts = fb->parts; P->part = fb->part;
  P->tx = x0; P->ty = l0; P->tw = x1 - x0; P->th = l1 - l0;
  P->retire_eps = (flags & RM_RENDER_FAST) ? ctx->retire_eps : 0.0f;
  P->stage = nullptr;
  P->stage_stride = 0;
  P->batch = 0;
  P->tile_wide = 0;
  std::memset(P->batch_noise, 0, sizeof P->batch_noise);
  P->block_order = nullptr;
  P->block_cost = nullptr;
  P->no_far_jump = (flags & RM_RENDER_NO_FAR_JUMP) ? 1 : 0;
  if (P->no_far_jump) P->scene.far_end = 0, P->scene.clear_rho = 0.0f;  // the far-field shortcuts inside an evaluation (KIFS tree) read the scene block
  if (flags & RM_RENDER_NO_CULL) P->scene.cull.cells = nullptr;
  return RM_OK;
}

#define RM_BATCH_TARGET_TILES 16384ll  // workgroups a batch launch of rm_render_samples aims for

// ---- the wavefront pipeline (rm_wavefront.inc): compiled into the tests' CROSS-CHECK build only (round 5) ---------------------------
// The same per-pixel program cut at its marches into queue-driven stages.  Rounds 1-3 the library picked it for some jobs; since round
// 4 the pixel kernel is the faster one for every measured job in both builds, and it alone grew per-shape surfaces, kind rows and
// the GL stack's arithmetic.  It stays what it is good for -- a second implementation of the marches, bounces and lights that the
// tests hold the pixel kernel to, bit for bit -- in tests/_xcheck/libhip_raymarch_xcheck.so (build.py build_crosscheck,
// -DRM_WITH_WAVEFRONT=1); libhip_raymarch.so, the product, has neither its kernels nor this orchestration, and refuses
// RM_RENDER_WAVEFRONT.  The product's own cross-check is the stepwise march: RM_RENDER_NO_FAR_JUMP | RM_RENDER_NO_CULL.
#if RM_WITH_WAVEFRONT
#define RM_MAX_MARCHES (RM_MAX_BOUNCES * (1 + RM_MAX_LIGHTS))
#define RM_COUNTERS_PER_MARCH 8  // queue heads and parked counts of the launches of one march

// One sample through the wavefront pipeline (rm_wavefront.inc).
#define RM_WF_STREAMS 4
#define RM_WF_MAX_BANDS 8

// One band of rows through the wavefront pipeline (rm_wavefront.inc) on `stream`.
static hipError_t launch_wavefront_band(rm_ctx* ctx, const KParams& P, int flags, hipStream_t stream, float4* ws,
                                        unsigned int* list, unsigned int* list2, unsigned int* heads) {
  const bool fast = (flags & RM_RENDER_FAST) != 0;
  rm::WfParams W{};
  W.k = P;
  W.tiles_x = (P.tw + 7) / 8;
  const int tiles_y = (P.th + 7) / 8;
  W.n_rays = W.tiles_x * tiles_y * 64;
  for (int i = 0; i < rm::WF_ARRAYS; i++) W.a[i] = ws + (size_t)i * (size_t)W.n_rays;
  W.stats = ctx->stats;
  hipError_t e;
  const bool classes = rm::wf_kind_has_cost_classes(P.scene.kind) && !(flags & RM_RENDER_NO_COST_CLASSES);
  // persistent march grid: every SIMD slot of the chip, or fewer when there are few rays
  int blocks = ctx->cu_count * ctx->wf_blocks_per_cu;
  const int needed = (W.n_rays + 255) / 256;
  if (blocks > needed) blocks = needed;
  int march = 0;
  auto do_march = [&](int pos_array, int dir_array, bool preview) -> hipError_t {
    W.pos_array = pos_array;
    W.dir_array = dir_array;
    unsigned int* c = heads + RM_COUNTERS_PER_MARCH * march++;  // [0] head of pass 0/1, [1] parked by pass 1, [2..] heads/counts of the pass-2 rounds
    auto go = [&](int pass) { return fast ? rm::wf_launch_march_fast(W, preview, pass, blocks, stream) : rm::wf_launch_march_strict(W, preview, pass, blocks, stream); };
    W.head = c;
    W.claims_per_wave = ctx->claims_per_wave;
    W.repark = 0;
    W.list_in = nullptr;
    W.list_in_count = nullptr;
    W.list_out = list;
    W.list_out_count = c + 1;
    if (!classes) return go(0);
    const int saved = blocks;
    const int pass1 = ctx->cu_count * ctx->pass1_blocks_per_cu;
    if (blocks > pass1) blocks = pass1;
    hipError_t e1 = go(1);  // cheap evaluations; parks the rays that need the deep one
    blocks = saved;
    if (e1 != hipSuccess) return e1;
    // The parked rays, compacted, in up to three rounds.  Fewer waves than SIMD
    // slots on purpose (2 per SIMD keep the VALU of this dependent-chain code
    // busy), and a round whose queue has drained does not let its waves thin
    // out to a few never-settling rays each: a wave with <= repark active lanes
    // parks them again and the next, smaller round re-compacts the survivors.
    unsigned int* in = list;
    unsigned int* outl = list2;
    unsigned int* in_count = c + 1;
    int round_blocks = ctx->cu_count * ctx->pass2_blocks_per_cu;
    for (int round = 0; round < ctx->pass2_rounds; round++) {
      const bool last = round == ctx->pass2_rounds - 1;
      W.head = c + 2 + 2 * round;
      W.list_in = in;
      W.list_in_count = in_count;
      W.list_out = outl;
      W.list_out_count = c + 3 + 2 * round;
      W.repark = last ? 0 : ctx->repark;
      blocks = round_blocks < saved ? round_blocks : saved;
      hipError_t e2 = go(2);
      if (e2 != hipSuccess) { blocks = saved; return e2; }
      in_count = W.list_out_count;
      unsigned int* t = in; in = outl; outl = t;
      round_blocks = round_blocks / 4 > ctx->cu_count / 4 ? round_blocks / 4 : ctx->cu_count / 4;
    }
    blocks = saved;
    return hipSuccess;
  };
  if ((e = rm::wf_launch_stage(W, 0, stream)) != hipSuccess) return e;  // setup
  if (P.u.renderMode == 1) {
    if ((e = do_march(rm::WF_POS, rm::WF_DIR, true)) != hipSuccess) return e;
    return rm::wf_launch_stage(W, 1, stream);
  }
  int bounces = 0;
  for (float i = 0.0f; i < P.u.reflections; i += 1.0f) bounces++;
  if (bounces == 0) return rm::wf_launch_stage(W, 2, stream);
  for (int b = 0; b < bounces; b++) {
    W.bounce = b;
    W.last_bounce = b == bounces - 1;
    if ((e = do_march(rm::WF_POS, rm::WF_DIR, false)) != hipSuccess) return e;
    if ((e = fast ? rm::wf_launch_shade_fast(W, stream) : rm::wf_launch_shade_strict(W, stream)) != hipSuccess) return e;
    for (int j = 0; j < P.u.lightCount; j++) {
      W.light = j;
      if ((e = do_march(rm::WF_SPOS, rm::WF_SDIR, false)) != hipSuccess) return e;
      if ((e = rm::wf_launch_stage(W, 3, stream)) != hipSuccess) return e;  // light
    }
  }
  return hipSuccess;
}

// One sample through the wavefront pipeline.  A large tile is cut into bands of
// rows that go down the pipeline on RM_WF_STREAMS side streams: every kernel of
// the pipeline ends with a tail in which the chip drains (a few long rays, the
// last workgroups), and the next band's kernels fill those holes.  Bands are
// independent (every pixel is), so this changes nothing in the results.
static hipError_t launch_wavefront(rm_ctx* ctx, const KParams& P, int flags) {
  hipStream_t stream = ctx->stream;
  const int tiles_x = (P.tw + 7) / 8, tiles_y = (P.th + 7) / 8;
  int bands = ctx->wf_bands > 0 ? ctx->wf_bands : (tiles_y >= 32 ? 2 : 1);
  if (bands > RM_WF_MAX_BANDS) bands = RM_WF_MAX_BANDS;
  if (bands > tiles_y) bands = tiles_y;
  const size_t total_rays = (size_t)tiles_x * (size_t)tiles_y * 64;
  hipError_t e;
  if (!ctx->heads) {
    if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->heads), sizeof(unsigned int) * RM_COUNTERS_PER_MARCH * RM_MAX_MARCHES * RM_WF_MAX_BANDS)) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->stats), sizeof(unsigned long long) * 16)) != hipSuccess) return e;
    if ((e = hipMemset(ctx->stats, 0, sizeof(unsigned long long) * 16)) != hipSuccess) return e;
    for (int s = 0; s < RM_WF_STREAMS; s++) {
      if ((e = hipStreamCreateWithFlags(&ctx->wf_stream[s], hipStreamNonBlocking)) != hipSuccess) return e;
      if ((e = hipEventCreateWithFlags(&ctx->wf_join[s], hipEventDisableTiming)) != hipSuccess) return e;
    }
    if ((e = hipEventCreateWithFlags(&ctx->wf_fork, hipEventDisableTiming)) != hipSuccess) return e;
  }
  if (ctx->ws_rays < total_rays) {
    if (ctx->ws) {
      if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
      (void)hipFree(ctx->ws);
      (void)hipFree(ctx->ws_list);
      (void)hipFree(ctx->ws_list2);
      ctx->ws = nullptr;
      ctx->ws_list = nullptr;
      ctx->ws_list2 = nullptr;
      ctx->ws_rays = 0;
    }
    if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->ws), sizeof(float4) * (size_t)rm::WF_ARRAYS * total_rays)) != hipSuccess) {
      char msg[160];
      std::snprintf(msg, sizeof msg, "wavefront pipeline: cannot allocate its %.1f GB ray workspace (%zu rays x %d B); render in tiles or use RM_RENDER_MEGAKERNEL",
                    (double)(sizeof(float4) * (size_t)rm::WF_ARRAYS * total_rays) / 1e9, total_rays, (int)(sizeof(float4) * rm::WF_ARRAYS));
      ctx->error = msg;
      return e;
    }
    if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->ws_list), sizeof(unsigned int) * total_rays)) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->ws_list2), sizeof(unsigned int) * total_rays)) != hipSuccess) return e;
    ctx->ws_rays = total_rays;
  }
  if ((e = hipMemsetAsync(ctx->heads, 0, sizeof(unsigned int) * RM_COUNTERS_PER_MARCH * RM_MAX_MARCHES * RM_WF_MAX_BANDS, stream)) != hipSuccess) return e;
  if (bands == 1) return launch_wavefront_band(ctx, P, flags, stream, ctx->ws, ctx->ws_list, ctx->ws_list2, ctx->heads);
  if ((e = hipEventRecord(ctx->wf_fork, stream)) != hipSuccess) return e;
  for (int s = 0; s < RM_WF_STREAMS; s++)
    if ((e = hipStreamWaitEvent(ctx->wf_stream[s], ctx->wf_fork, 0)) != hipSuccess) return e;
  size_t rays_before = 0;
  for (int b = 0; b < bands; b++) {
    const int t0 = (int)((long long)tiles_y * b / bands), t1 = (int)((long long)tiles_y * (b + 1) / bands);
    KParams B = P;
    B.ty = P.ty + t0 * 8;
    B.th = (t1 * 8 < P.th ? t1 * 8 : P.th) - t0 * 8;
    const size_t band_rays = (size_t)tiles_x * (size_t)(t1 - t0) * 64;
    if ((e = launch_wavefront_band(ctx, B, flags, ctx->wf_stream[b % RM_WF_STREAMS], ctx->ws + (size_t)rm::WF_ARRAYS * rays_before,
                                   ctx->ws_list + rays_before, ctx->ws_list2 + rays_before, ctx->heads + (size_t)RM_COUNTERS_PER_MARCH * RM_MAX_MARCHES * b)) != hipSuccess)
      return e;
    rays_before += band_rays;
  }
  for (int s = 0; s < RM_WF_STREAMS; s++) {
    if ((e = hipEventRecord(ctx->wf_join[s], ctx->wf_stream[s])) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(stream, ctx->wf_join[s], 0)) != hipSuccess) return e;
  }
  return hipSuccess;
}

#endif  // RM_WITH_WAVEFRONT

// Which implementation of the per-pixel program runs a job (same results either way).
// Measured on MI355X (fast build, ms per sample, pixel kernel / wavefront pipeline).  Round 2:
//   Mandelbulb 3840x2160 full      2.49 / 4.5         sphere 1080p preview 0.11 / 1.4
//   CSG-64 4096x512  (2 Mpx)       6.79 / 7.67        CSG-64 4096x4096 (16.8 Mpx)  47.0 / 40.6
//   CSG-64 8192x1024 (8.4 Mpx)     60.5 / 54.6        CSG-64 8192x8192 (67 Mpx)    467  / 415
// Round 3, with the far-field jump in both and the compacting pixel kernel for long tables (RM_KIND_TABLE_BIG; tools/r03_table.py):
//   Mandelbulb 3840x2160 full      1.98 / 4.0
//   CSG-64 4096^2, rank 0's 1/8    2.33 / 6.33        CSG-64 4096x4096             15.3 / 14.4
//   CSG-64 8192^2, rank 0's 1/8    31.5 / 33.1        CSG-64 8192x8192             244  / 229
// and at the end of round 3, with the tighter far field and CSG-64's own pixel kernel (RM_KIND_TABLE_SMOOTH):
//   CSG-64 4096^2, rank 0's 1/8    2.19 / 5.79        CSG-64 4096x4096             12.9 / 13.2
//   CSG-64 8192^2, rank 0's 1/8    29.2 / 32.7        CSG-64 8192x8192             225.7 / 225.6
// The fast build's one-kernel form now wins or ties everywhere -- the pipeline's global ray compaction bought 10-16 % on the two
// full CSG frames in round 2, 6 % after the pixel kernel compacted its table rays, nothing now -- and it needs no 240 bytes of
// workspace per pixel: the fast build never picks the pipeline by itself any more (RM_RENDER_WAVEFRONT still forces it; it remains
// the second implementation the tests hold the pixel kernel to).  Round 4, with the exits, the job-shape variants and the row culling
// in the parity build too, the same holds there (strict, pixel kernel / pipeline):
//   CSG-64 4096x4096               26.5 / 29.8        CSG-64 8192^2, rank 0's 1/8    45.2 / 85.7
// so the library never picks the pipeline by itself.
static bool prefer_wavefront(const KParams&, int) { return false; }

// The implementation a render call uses.  The GL-stack arithmetic exists as the pixel kernel only, and so do
// position-dependent materials (RM_TABLE_HAS_SURFACES): the pipeline's stages carry one material block per scene -- its light
// stage has no scene table staged -- so such scenes render with the pixel kernel whatever the flags ask for (same results).
static bool uses_wavefront(const rm_ctx* ctx, const KParams& P, int flags) {
  if (!RM_WITH_WAVEFRONT) return false;  // the product library (build_params refuses the flag)
  if (ctx->gl_stack && !(flags & RM_RENDER_FAST)) return false;
  if (P.scene.table_flags & (RM_TABLE_HAS_SURFACES | RM_TABLE_HAS_KIND)) return false;
  return (flags & RM_RENDER_WAVEFRONT) ? true : (flags & RM_RENDER_MEGAKERNEL) ? false : prefer_wavefront(P, flags);
}

static hipError_t launch_pixels_ordered(rm_ctx* ctx, const KParams& P, int flags, hipStream_t stream, int slot);

// frees the staging buffers of the samples in flight (the caller has synchronised the device, or is giving the buffers up)
static void release_staging(rm_ctx* ctx) {
  for (int s = 0; s < RM_SP_MAX; s++) {
    if (ctx->sp_stage[s]) (void)hipFree(ctx->sp_stage[s]);
    ctx->sp_stage[s] = nullptr;
  }
  ctx->sp_elems = 0;
}

// Staging the library may hold for a job: a quarter of what the device has free (counting what staging already holds).
// rm_render_samples sizes its automatic batch by it, and a launch whose staging cannot be allocated at all renders
// unstaged, one sample at a time on the context's stream, instead of failing (launch / rm_render_samples).
static size_t staging_budget(rm_ctx* ctx) {
  if (const char* v = std::getenv("RM_STAGING_BUDGET_MB")) {  // a fixed budget (tests; hosts that share the device with other allocators)
    const long long mb = std::atoll(v);
    if (mb >= 0) return (size_t)mb << 20;
  }
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t)1 << 30;
  size_t held = 0;
  for (int s = 0; s < RM_SP_MAX; s++) held += ctx->sp_stage[s] ? sizeof(float4) * ctx->sp_elems : 0;
  return (free_b + held) / 4;
}

// The pixel kernel of one sample -- or of a batch of `batch` samples, randNoise pairs in `noise` -- on a side stream,
// staged, and its blend on the context's stream (see rm_ctx).
static hipError_t launch_pixels_in_flight(rm_ctx* ctx, const KParams& P, int flags, int batch = 1, const float* noise = nullptr) {
  hipError_t e;
  const int depth = ctx->samples_in_flight;
  // side streams are made as the depth asks for them, not all RM_SP_MAX at once: the HIP runtime deals a process's streams
  // over a few hardware queues, and streams that share a queue serialise (measured: see bench.py, GPU_MAX_HW_QUEUES)
  for (int s = 0; s < depth; s++) {
    if (ctx->sp_stream[s]) continue;
    if ((e = hipStreamCreateWithFlags(&ctx->sp_stream[s], hipStreamNonBlocking)) != hipSuccess) return e;
    if ((e = hipEventCreateWithFlags(&ctx->sp_done[s], hipEventDisableTiming)) != hipSuccess) return e;
    if ((e = hipEventCreateWithFlags(&ctx->sp_free[s], hipEventDisableTiming)) != hipSuccess) return e;
  }
  ctx->sp_ready = true;
  // A slot's staging holds the launch's TILE, compact (tile pixel i of sample k at k * 3 * stride + i), so it is sized by the
  // largest batch x tile a job has used -- not by the frame: a subdivided 8192^2 render stages a tile, not 3.2 GB x batch.
  // One buffer of `sp_elems` float4 per slot; a launch needs 3 * tw * th * batch of them.
  const size_t tile_px = (size_t)P.tw * (size_t)P.th;
  const size_t need = 3 * tile_px * (size_t)batch;
  if (ctx->sp_elems < need) {
    if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
    release_staging(ctx);
    for (int s = 0; s < depth; s++)
      if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->sp_stage[s]), sizeof(float4) * need)) != hipSuccess) { release_staging(ctx); return e; }
    ctx->sp_elems = need;
  }
  const int slot = (int)(ctx->sp_next++ % (unsigned int)depth);
  if (!ctx->sp_stage[slot]) {  // the depth was raised after the buffers were made
    if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->sp_stage[slot]), sizeof(float4) * ctx->sp_elems)) != hipSuccess) return e;
  }
  KParams Q = P;
  Q.stage = ctx->sp_stage[slot];
  Q.stage_stride = (long long)tile_px;
  if (batch > 1) {
    Q.batch = batch;
    std::memcpy(Q.batch_noise, noise, sizeof(float) * 2 * (size_t)batch);
  }
  hipStream_t side = ctx->sp_stream[slot];
  // the render reads no plane: it only has to wait until the blend that last used this staging buffer is done
  if ((e = hipStreamWaitEvent(side, ctx->sp_free[slot], 0)) != hipSuccess) return e;
  if (ctx->cull_event && (e = hipStreamWaitEvent(side, ctx->cull_event, 0)) != hipSuccess) return e;  // ... and for the scene's culling grid, built on the context's cull_stream (scene_cull_grid records cull_event behind the build)
  ctx->lpt[slot].launches = 0;
  if ((e = launch_pixels_ordered(ctx, Q, flags, side, slot)) != hipSuccess) return e;
  if ((e = hipEventRecord(ctx->sp_done[slot], side)) != hipSuccess) return e;
  if (ctx->lpt[slot].launches > 0) {  // the launch recorded tile costs: sort them now, behind the completion event
    rm_ctx::Lpt& Ls = ctx->lpt[slot];
    if ((e = rm::launch_order(Ls.cost, Ls.order, Ls.order + Ls.capacity, (int)Ls.launches, side)) != hipSuccess) return e;
  }
  if ((e = hipStreamWaitEvent(ctx->stream, ctx->sp_done[slot], 0)) != hipSuccess) return e;
  if ((e = rm::launch_combine(Q, ctx->stream)) != hipSuccess) return e;
  return hipEventRecord(ctx->sp_free[slot], ctx->stream);
}

// The pixel kernel on `stream`, its tiles in the order of their cost in the previous launch of the same job on that stream
// (slot = which of the context's streams: a sample-in-flight slot, or RM_SP_MAX for the context's own stream).
static hipError_t launch_pixels_ordered(rm_ctx* ctx, const KParams& P, int flags, hipStream_t stream, int slot) {
  const bool fast = (flags & RM_RENDER_FAST) != 0;
  int gx = 0, gy = 0;
  rm::pixel_grid(P, &gx, &gy);
  const long long tiles = (long long)gx * gy;
  if (!ctx->lpt_enabled || tiles < 512 || tiles > (1ll << 22))  // small jobs end on launch latency, not on a tail
    return fast ? rm::launch_pixels_fast(P, stream) : ctx->gl_stack ? rm_gl_launch_pixels(&P, stream) : rm::launch_pixels_strict(P, stream);
  rm_ctx::Lpt& L = ctx->lpt[slot];
  hipError_t e;
  const long long key[8] = {P.W, P.H, ((long long)P.tx << 32) | (unsigned int)P.ty, ((long long)P.tw << 32) | (unsigned int)P.th,
                            ((long long)P.stripe_rows << 40) | ((long long)P.parts << 20) | P.part, P.row_begin,
                            ((long long)P.scene.kind << 8) | P.u.renderMode, tiles};
  const bool async_sort = slot == RM_SP_MAX;  // the context's own stream: sort on a side stream, one sample behind
  if (L.capacity < tiles) {
    if (L.cost) {
      (void)hipStreamSynchronize(stream);
      if (ctx->lpt_stream) (void)hipStreamSynchronize(ctx->lpt_stream);
      (void)hipFree(L.cost); (void)hipFree(L.order);
      if (L.cost2) { (void)hipFree(L.cost2); (void)hipFree(L.order2); }
    }
    L.cost = L.order = L.cost2 = L.order2 = nullptr;
    L.capacity = 0;
    if ((e = hipMalloc(reinterpret_cast<void**>(&L.cost), sizeof(unsigned int) * (size_t)tiles)) != hipSuccess) return e;
    const size_t order_elems = (size_t)tiles + rm::rm_order_scratch_elems(tiles);  // the sort's per-workgroup histograms behind the order
    if ((e = hipMalloc(reinterpret_cast<void**>(&L.order), sizeof(unsigned int) * order_elems)) != hipSuccess) return e;
    if (async_sort) {
      if ((e = hipMalloc(reinterpret_cast<void**>(&L.cost2), sizeof(unsigned int) * (size_t)tiles)) != hipSuccess) return e;
      if ((e = hipMalloc(reinterpret_cast<void**>(&L.order2), sizeof(unsigned int) * order_elems)) != hipSuccess) return e;
    }
    L.capacity = (int)tiles;
    L.have_cost = false;
  }
  if (std::memcmp(key, L.key, sizeof key) != 0) {
    std::memcpy(L.key, key, sizeof key);
    L.have_cost = false;
  }
  KParams Q = P;
  if (!async_sort) {
    // a sample-in-flight slot: the costs are sorted on the slot's own stream right AFTER the render (sort_slot_costs,
    // called once the render's completion event is recorded), so the sort's two small launches sit in the
    // shadow of the other slots' renders instead of in front of this slot's next one
    Q.block_cost = L.cost;
    if (L.have_cost) Q.block_order = L.order;
    else if ((e = hipMemsetAsync(L.cost, 0, sizeof(unsigned int) * (size_t)tiles, stream)) != hipSuccess) return e;
    L.have_cost = true;
    L.launches = (unsigned long long)tiles;  // what sort_slot_costs has to sort
    return fast ? rm::launch_pixels_fast(Q, stream) : ctx->gl_stack ? rm_gl_launch_pixels(&Q, stream) : rm::launch_pixels_strict(Q, stream);
  }
  if (!ctx->lpt_stream) {
    if ((e = hipStreamCreateWithFlags(&ctx->lpt_stream, hipStreamNonBlocking)) != hipSuccess) return e;
  }
  for (int k = 0; k < 2; k++) {
    if (!L.rendered[k] && (e = hipEventCreateWithFlags(&L.rendered[k], hipEventDisableTiming)) != hipSuccess) return e;
    if (!L.sorted[k] && (e = hipEventCreateWithFlags(&L.sorted[k], hipEventDisableTiming)) != hipSuccess) return e;
  }
  if (!L.have_cost) {  // a new job: both cost buffers start from zero, no order yet
    if ((e = hipStreamSynchronize(ctx->lpt_stream)) != hipSuccess) return e;  // sorts of the previous job still use the buffers
    if ((e = hipMemsetAsync(L.cost, 0, sizeof(unsigned int) * (size_t)tiles, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(L.cost2, 0, sizeof(unsigned int) * (size_t)tiles, stream)) != hipSuccess) return e;
    L.launches = 0;
    L.have_cost = true;
  }
  const int cur = (int)(L.launches & 1ull), prev = cur ^ 1;
  unsigned int* cost_cur = cur ? L.cost2 : L.cost;
  unsigned int* order_cur = cur ? L.order2 : L.order;
  // launch n writes its costs into buffer n % 2 (zeroed by the sort of launch n - 2, which launch n therefore waits
  // for -- it ran during launch n - 1) and starts in the order that sort left in the same buffer's order array
  if (L.launches >= 2) {
    if ((e = hipStreamWaitEvent(stream, L.sorted[cur], 0)) != hipSuccess) return e;
    Q.block_order = order_cur;
  }
  Q.block_cost = cost_cur;
  (void)prev;
  if ((e = fast ? rm::launch_pixels_fast(Q, stream) : ctx->gl_stack ? rm_gl_launch_pixels(&Q, stream) : rm::launch_pixels_strict(Q, stream)) != hipSuccess) return e;
  if ((e = hipEventRecord(L.rendered[cur], stream)) != hipSuccess) return e;
  if ((e = hipStreamWaitEvent(ctx->lpt_stream, L.rendered[cur], 0)) != hipSuccess) return e;
  if ((e = rm::launch_order(cost_cur, order_cur, order_cur + L.capacity, (int)tiles, ctx->lpt_stream)) != hipSuccess) return e;  // sorts and zeroes the costs
  if ((e = hipEventRecord(L.sorted[cur], ctx->lpt_stream)) != hipSuccess) return e;
  L.launches++;
  return hipSuccess;
}

static hipError_t launch(rm_ctx* ctx, const KParams& P, int flags) {
  const bool wavefront = uses_wavefront(ctx, P, flags);
  ctx->last_pipeline = wavefront ? RM_PIPELINE_WAVEFRONT : RM_PIPELINE_PIXEL_KERNEL;
#if RM_WITH_WAVEFRONT
  if (wavefront) return launch_wavefront(ctx, P, flags);
#endif
  // full mode with at least one bounce: the kernel's only use of the planes is the final blend, which can be split off
  if (ctx->samples_in_flight > 1 && !(flags & RM_RENDER_NO_OVERLAP) && P.u.renderMode == 0 && P.u.reflections > 0.0f) {
    const hipError_t e = launch_pixels_in_flight(ctx, P, flags);
    if (e != hipErrorOutOfMemory) return e;
    (void)hipGetLastError();  // no room for the staging of this tile: render it unstaged (same bits, no overlap)
  }
  return launch_pixels_ordered(ctx, P, flags, ctx->stream, RM_SP_MAX);
}

int rm_render_sample(rm_ctx* ctx, rm_scene* scene, rm_fb* fb, const RmUniforms* uniforms, const RmRect* tile, int flags) {
  KParams P;
  bool empty = false;
  if (int rc = build_params(ctx, scene, fb, uniforms, tile, flags, &P, &empty)) return rc;
  if (empty) return RM_OK;
  RM_HIP(ctx, hipSetDevice(ctx->device));
  RM_HIP(ctx, launch(ctx, P, flags));
  return RM_OK;
}

int rm_render_samples(rm_ctx* ctx, rm_scene* scene, rm_fb* fb, const RmUniforms* uniforms, const float* rand_noise_pairs,
                      int count, const RmRect* tile, int flags) {
  KParams P;
  bool empty = false;
  if (int rc = build_params(ctx, scene, fb, uniforms, tile, flags, &P, &empty)) return rc;
  if (count < 0 || (count > 0 && !rand_noise_pairs)) return fail(ctx, RM_ERR_INVALID, "rm_render_samples: bad count / NULL randNoise");
  if (empty) return RM_OK;
  if (count > 1 && scene->cull_wanted)  // build_params counted one sample of the tile: the call asks for `count` (the grid, if this earns it, serves the next call)
    scene->px_seen += (long long)(count - 1) * (long long)P.tw * (long long)P.th * (fb->stripe_rows > 0 ? (long long)fb->parts : 1ll);
  RM_HIP(ctx, hipSetDevice(ctx->device));
  // Samples the pixel kernel can stage (the conditions of launch()) go out in batches: one launch per batch.
  const bool wavefront = uses_wavefront(ctx, P, flags);
  const bool stageable = !wavefront && !(flags & RM_RENDER_NO_OVERLAP) && P.u.renderMode == 0 && P.u.reflections > 0.0f;
  int per_launch = 1;
  if (stageable && ctx->sample_batch != 1) {
    per_launch = ctx->sample_batch;
    if (per_launch == 0) {
      int gx = 0, gy = 0;
      rm::pixel_grid(P, &gx, &gy);
      const long long tiles = (long long)gx * gy;
      per_launch = tiles > 0 ? (int)((RM_BATCH_TARGET_TILES + tiles / 2) / tiles) : 1;  // to the nearest: a 1/8 shard of the headline frame has 2 160 tiles -> 8
    }
    per_launch = per_launch < 1 ? 1 : per_launch > RM_BATCH_MAX ? RM_BATCH_MAX : per_launch;
    // the staging of a batch is 48 bytes per tile pixel, sample of the batch and launch in flight: within the budget
    const size_t per_sample = sizeof(float4) * 3 * (size_t)P.tw * (size_t)P.th * (size_t)ctx->samples_in_flight;
    const size_t fit = staging_budget(ctx) / (per_sample ? per_sample : 1);
    if ((size_t)per_launch > fit) per_launch = fit < 1 ? 1 : (int)fit;
  }
  ctx->last_pipeline = wavefront ? RM_PIPELINE_WAVEFRONT : RM_PIPELINE_PIXEL_KERNEL;
  for (int i = 0; i < count;) {
    int n = count - i < per_launch ? count - i : per_launch;
    if (n > 1) {
      const hipError_t e = launch_pixels_in_flight(ctx, P, flags, n, rand_noise_pairs + 2 * i);
      if (e == hipErrorOutOfMemory) {  // the batch's staging does not fit after all: one sample per launch from here on
        (void)hipGetLastError();
        per_launch = 1;
        continue;
      }
      RM_HIP(ctx, e);
    } else {
      P.u.randNoise[0] = rand_noise_pairs[2 * i];
      P.u.randNoise[1] = rand_noise_pairs[2 * i + 1];
      RM_HIP(ctx, launch(ctx, P, flags));
    }
    i += n;
  }
  return RM_OK;
}

int rm_ctx_set_sample_batch(rm_ctx* ctx, int n) {
  if (!ctx) return RM_ERR_INVALID;
  if (n < 0 || n > RM_BATCH_MAX) return fail(ctx, RM_ERR_INVALID, "rm_ctx_set_sample_batch: n must be 0 (automatic) or 1..8");
  ctx->sample_batch = n;
  return RM_OK;
}

int rm_render_timed(rm_ctx* ctx, rm_scene* scene, rm_fb* fb, const RmUniforms* uniforms, int count, const RmRect* tile,
                    int flags, float* ms_per_launch) {
  KParams P;
  bool empty = false;
  if (int rc = build_params(ctx, scene, fb, uniforms, tile, flags, &P, &empty)) return rc;
  if (count < 1 || !ms_per_launch) return fail(ctx, RM_ERR_INVALID, "rm_render_timed: bad count / NULL result");
  if (empty) return fail(ctx, RM_ERR_INVALID, "rm_render_timed: empty tile");
  RM_HIP(ctx, hipSetDevice(ctx->device));
  RM_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  for (int i = 0; i < count; i++) RM_HIP(ctx, launch(ctx, P, flags));
  RM_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  RM_HIP(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.0f;
  RM_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *ms_per_launch = ms / (float)count;
  return RM_OK;
}

// ---- assembling a sharded frame -------------------------------------------------------

int rm_assemble_striped_bytes(rm_ctx* ctx, const void* src, int parts, int max_rows, long long row_bytes, int height, int stripe_rows,
                              void* dst, void* hip_stream) {
  if (!ctx || !src || !dst) return fail(ctx, RM_ERR_INVALID, "rm_assemble_striped: NULL argument");
  if (parts < 1 || row_bytes < 4 || (row_bytes & 3) || row_bytes > (1ll << 24) || height < 1 || stripe_rows < 1)
    return fail(ctx, RM_ERR_INVALID, "rm_assemble_striped: parts, height and stripe_rows must be >= 1, row_bytes a multiple of 4");
  if (max_rows < striped_rows_below(height, stripe_rows, parts, 0))  // part 0 holds the most rows
    return fail(ctx, RM_ERR_INVALID, "rm_assemble_striped: max_rows is smaller than the largest part's window");
  if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) return fail(ctx, RM_ERR_INVALID, "rm_assemble_striped: buffers must be 16-byte aligned");
  RM_HIP(ctx, hipSetDevice(ctx->device));
  RM_HIP(ctx, rm::launch_assemble(src, parts, max_rows, row_bytes, height, stripe_rows, dst, hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
  return RM_OK;
}

int rm_assemble_striped(rm_ctx* ctx, const void* src, int parts, int max_rows, int width, int height, int stripe_rows, void* dst,
                        void* hip_stream) {
  if (width < 1) return fail(ctx, RM_ERR_INVALID, "rm_assemble_striped: width must be >= 1");
  return rm_assemble_striped_bytes(ctx, src, parts, max_rows, (long long)width * 16, height, stripe_rows, dst, hip_stream);
}

// ---- present ---------------------------------------------------------------------

int rm_present_device(rm_ctx* ctx, const void* color, const void* normal_dof, int width, int height, int samples, void* out_rgba8_device,
                      void* hip_stream) {
  if (!ctx || !color || !out_rgba8_device) return fail(ctx, RM_ERR_INVALID, "rm_present_device: NULL argument");
  if (width < 1 || height < 1 || samples < 1) return fail(ctx, RM_ERR_INVALID, "rm_present_device: width, height and samples must be >= 1");
  RM_HIP(ctx, hipSetDevice(ctx->device));
  RM_HIP(ctx, (ctx->gl_stack ? rm_gl_launch_present : rm::launch_present)(static_cast<const float4*>(color), static_cast<const float4*>(normal_dof), width, height,
                                 1.0f / (float)samples, static_cast<uchar4*>(out_rgba8_device), hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
  return RM_OK;
}

int rm_present_rows(rm_ctx* ctx, rm_fb* fb, int samples, void* out_rgba8_device, void* hip_stream) {
  if (!ctx || !fb || !out_rgba8_device) return fail(ctx, RM_ERR_INVALID, "rm_present_rows: NULL argument");
  if (fb->ctx != ctx) return fail(ctx, RM_ERR_INVALID, "rm_present_rows: framebuffer belongs to another context");
  if (samples < 1) return fail(ctx, RM_ERR_INVALID, "rm_present_rows: samples must be >= 1");
  RM_HIP(ctx, hipSetDevice(ctx->device));
  RM_HIP(ctx, (ctx->gl_stack ? rm_gl_launch_present_rows : rm::launch_present_rows)(fb->plane[0], (long long)fb->width * (long long)fb->row_count, 1.0f / (float)samples,
                                      static_cast<uchar4*>(out_rgba8_device), hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
  return RM_OK;
}

int rm_pack_present_rows(rm_ctx* ctx, rm_fb* fb, void* out_float4_device, void* hip_stream) {
  if (!ctx || !fb || !out_float4_device) return fail(ctx, RM_ERR_INVALID, "rm_pack_present_rows: NULL argument");
  if (fb->ctx != ctx) return fail(ctx, RM_ERR_INVALID, "rm_pack_present_rows: framebuffer belongs to another context");
  if (reinterpret_cast<uintptr_t>(out_float4_device) & 15u) return fail(ctx, RM_ERR_INVALID, "rm_pack_present_rows: the buffer must be 16-byte aligned");
  RM_HIP(ctx, hipSetDevice(ctx->device));
  RM_HIP(ctx, rm::launch_pack_rows(fb->plane[0], fb->plane[1], (long long)fb->width * (long long)fb->row_count, static_cast<float4*>(out_float4_device),
                                   hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
  return RM_OK;
}

int rm_present_planes(rm_ctx* ctx, const void* color, const void* normal_dof, int width, int height, int samples, uint8_t* out_rgba8) {
  if (!ctx || !color || !out_rgba8) return fail(ctx, RM_ERR_INVALID, "rm_present_planes: NULL argument");
  if (width < 1 || height < 1 || samples < 1) return fail(ctx, RM_ERR_INVALID, "rm_present_planes: width, height and samples must be >= 1");
  RM_HIP(ctx, hipSetDevice(ctx->device));
  const size_t pixels = (size_t)width * (size_t)height;
  if (ctx->present_cap < pixels) {  // the staging buffer lives with the context: a live loop presents every sample
    if (ctx->present_buf) {
      RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipFree(ctx->present_buf);
      ctx->present_buf = nullptr;
      ctx->present_cap = 0;
    }
    RM_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->present_buf), pixels * 4));
    ctx->present_cap = pixels;
  }
  if (int rc = rm_present_device(ctx, color, normal_dof, width, height, samples, ctx->present_buf, nullptr)) return rc;
  RM_HIP(ctx, hipMemcpyAsync(out_rgba8, ctx->present_buf, pixels * 4, hipMemcpyDeviceToHost, ctx->stream));
  RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RM_OK;
}

int rm_present(rm_ctx* ctx, rm_fb* fb, int samples, uint8_t* out_rgba8) {
  if (!ctx || !fb) return fail(ctx, RM_ERR_INVALID, "rm_present: NULL argument");
  if (fb->stripe_rows > 0 || fb->row_begin != 0 || fb->row_count != fb->height)
    return fail(ctx, RM_ERR_INVALID, "rm_present: the blur reads neighbouring rows, so it needs the whole frame: gather the planes and use rm_present_planes (or rm_present_rows when depth of field is off)");
  return rm_present_planes(ctx, fb->plane[0], fb->plane[1], fb->width, fb->height, samples, out_rgba8);
}

// ---- present of a frame sharded over the GPUs of ONE process ----------------------------------------

static int grow(rm_ctx* ctx, void** p, size_t* cap, size_t bytes, bool pinned_host = false) {
  if (*cap >= bytes) return RM_OK;
  RM_HIP(ctx, hipSetDevice(ctx->device));
  if (*p) {
    RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->shard_stream) RM_HIP(ctx, hipStreamSynchronize(ctx->shard_stream));
    (void)(pinned_host ? hipHostFree(*p) : hipFree(*p));
    *p = nullptr;
    *cap = 0;
  }
  if ((pinned_host ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes)) != hipSuccess) {
    (void)hipGetLastError();
    return fail(ctx, RM_ERR_DEVICE, "rm_present_sharded: out of memory");
  }
  *cap = bytes;
  return RM_OK;
}

int rm_present_striped_rows(rm_ctx* ctx, const void* color, const void* normal_dof, int width, int height, int samples, int stripe_rows, int parts, int part,
                            void* out_rgba8_device, void* hip_stream) {
  if (!ctx || !color || !out_rgba8_device) return fail(ctx, RM_ERR_INVALID, "rm_present_striped_rows: NULL argument");
  if (width < 1 || height < 1 || samples < 1 || stripe_rows < 1 || parts < 1 || part < 0 || part >= parts)
    return fail(ctx, RM_ERR_INVALID, "rm_present_striped_rows: width, height, samples, stripe_rows and parts must be >= 1, 0 <= part < parts");
  RM_HIP(ctx, hipSetDevice(ctx->device));
  RM_HIP(ctx, (ctx->gl_stack ? rm_gl_launch_present_striped : rm::launch_present_striped)(
                  static_cast<const float4*>(color), static_cast<const float4*>(normal_dof), width, height, 1.0f / (float)samples, static_cast<uchar4*>(out_rgba8_device),
                  stripe_rows, parts, part, striped_rows_below(height, stripe_rows, parts, part), hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
  return RM_OK;
}

// device-to-device, over xGMI between two GPUs (direct access where the topology has it; the copy works either way)
static int shard_copy(rm_ctx* root, rm_ctx* from, rm_ctx* to, void* dst, const void* src, size_t bytes, hipStream_t stream) {
  if (from->device == to->device) {
    RM_HIP(root, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
    return RM_OK;
  }
  if (!(from->peer_enabled & (1ull << (to->device & 63)))) {
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, from->device, to->device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(to->device, 0);
    (void)hipGetLastError();  // "already enabled" is fine
    from->peer_enabled |= 1ull << (to->device & 63);
  }
  RM_HIP(root, hipMemcpyPeerAsync(dst, to->device, src, from->device, bytes, stream));
  return RM_OK;
}

int rm_present_sharded_start(rm_ctx* const* ctxs, rm_fb* const* fbs, int parts, int samples, int dof) {
  rm_ctx* root = (ctxs && parts >= 1) ? ctxs[0] : nullptr;
  if (!root || !fbs) return fail(root, RM_ERR_INVALID, "rm_present_sharded: NULL argument");
  if (samples < 1) return fail(root, RM_ERR_INVALID, "rm_present_sharded: samples must be >= 1");
  if (root->shard_pending) return fail(root, RM_ERR_INVALID, "rm_present_sharded_start: the previous present has not been finished (rm_present_sharded_finish)");
  const rm_fb* f0 = fbs[0];
  for (int p = 0; p < parts; p++) {
    const rm_fb* f = fbs[p];
    if (!ctxs[p] || !f || f->ctx != ctxs[p]) return fail(root, RM_ERR_INVALID, "rm_present_sharded: framebuffer p must belong to context p");
    if (f->stripe_rows < 1 || f->parts != parts || f->part != p || f->width != f0->width || f->height != f0->height || f->stripe_rows != f0->stripe_rows)
      return fail(root, RM_ERR_INVALID, "rm_present_sharded: framebuffer p must be part p of `parts` striped windows of one image (rm_fb_create_striped)");
    if (dof && !f->plane[1]) return fail(root, RM_ERR_INVALID, "rm_present_sharded: depth of field needs the normal_dof plane");
  }
  const int W = f0->width, H = f0->height, stripe = f0->stripe_rows;
  const int max_rows = striped_rows_below(H, stripe, parts, 0);  // part 0 holds the most rows
  const size_t part8 = (size_t)max_rows * (size_t)W * sizeof(uchar4), part32 = (size_t)max_rows * (size_t)W * sizeof(float4);
  const size_t canvas_bytes = (size_t)H * (size_t)W * sizeof(uchar4);
  if (int rc = grow(root, &root->shard_recv, &root->shard_recv_cap, part8 * (size_t)parts)) return rc;
  if (int rc = grow(root, &root->shard_canvas, &root->shard_canvas_cap, canvas_bytes)) return rc;
  if (int rc = grow(root, &root->shard_host, &root->shard_host_cap, canvas_bytes, true)) return rc;
  for (int p = 0; p < parts; p++) {
    rm_ctx* c = ctxs[p];
    int rc = grow(c, &c->shard_rows, &c->shard_rows_cap, dof ? part32 : part8);
    if (!rc && dof) rc = grow(c, &c->shard_rows8, &c->shard_rows8_cap, part8);
    if (!rc && dof) rc = grow(c, &c->shard_all, &c->shard_all_cap, part32 * (size_t)parts);
    if (!rc && dof) rc = grow(c, &c->shard_frame, &c->shard_frame_cap, (size_t)H * (size_t)W * sizeof(float4));
    if (rc) return c == root ? rc : fail(root, rc, rm_last_error(c));
    RM_HIP(root, hipSetDevice(c->device));
    if (!c->shard_stream) RM_HIP(root, hipStreamCreateWithFlags(&c->shard_stream, hipStreamNonBlocking));
    if (!c->shard_snap) RM_HIP(root, hipEventCreateWithFlags(&c->shard_snap, hipEventDisableTiming));
    if (!c->shard_ev) RM_HIP(root, hipEventCreateWithFlags(&c->shard_ev, hipEventDisableTiming));
  }
  if (!root->shard_done) { RM_HIP(root, hipSetDevice(root->device)); RM_HIP(root, hipEventCreateWithFlags(&root->shard_done, hipEventDisableTiming)); }
  // 1. every context snapshots what it holds, on the stream its renders are ordered on: the next samples may start at once (the planes
  //    are accumulated in place).  (The previous present was finished -- checked above -- so its buffers are free.)
  for (int p = 0; p < parts; p++) {
    rm_ctx* c = ctxs[p];
    rm_fb* f = fbs[p];
    RM_HIP(root, hipSetDevice(c->device));
    const long long pixels = (long long)W * (long long)f->row_count;
    if (dof) RM_HIP(root, rm::launch_pack_rows(f->plane[0], f->plane[1], pixels, static_cast<float4*>(c->shard_rows), c->stream));
    else RM_HIP(root, (c->gl_stack ? rm_gl_launch_present_rows : rm::launch_present_rows)(f->plane[0], pixels, 1.0f / (float)samples, static_cast<uchar4*>(c->shard_rows), c->stream));
    RM_HIP(root, hipEventRecord(c->shard_snap, c->stream));
    RM_HIP(root, hipStreamWaitEvent(c->shard_stream, c->shard_snap, 0));
  }
  // 2. everything else travels on the contexts' present streams, beside the renders
  if (dof) {
    // all-gather: every context's packed rows to every context (display.frag:44-55 reads up to 16 rows either side of a pixel, and with
    // 8-row stripes those are other GPUs' rows) ...
    for (int p = 0; p < parts; p++) {
      rm_ctx* c = ctxs[p];
      RM_HIP(root, hipSetDevice(c->device));
      for (int q = 0; q < parts; q++)
        if (int rc = shard_copy(root, c, ctxs[q], static_cast<char*>(ctxs[q]->shard_all) + part32 * (size_t)p, c->shard_rows, (size_t)fbs[p]->row_count * (size_t)W * sizeof(float4), c->shard_stream)) return rc;
      RM_HIP(root, hipEventRecord(c->shard_ev, c->shard_stream));
    }
    // ... then every context puts the frame in image order, blurs and tone-maps ITS stripes (1 / parts of the pass each) and sends
    // the bytes to the root
    for (int q = 0; q < parts; q++) {
      rm_ctx* c = ctxs[q];
      RM_HIP(root, hipSetDevice(c->device));
      for (int p = 0; p < parts; p++) RM_HIP(root, hipStreamWaitEvent(c->shard_stream, ctxs[p]->shard_ev, 0));
      RM_HIP(root, rm::launch_assemble(c->shard_all, parts, max_rows, (long long)W * (long long)sizeof(float4), H, stripe, c->shard_frame, c->shard_stream));
      RM_HIP(root, (c->gl_stack ? rm_gl_launch_present_striped : rm::launch_present_striped)(static_cast<const float4*>(c->shard_frame), static_cast<const float4*>(c->shard_frame), W, H,
                                                                                                1.0f / (float)samples, static_cast<uchar4*>(c->shard_rows8), stripe, parts, q,
                                                                                                fbs[q]->row_count, c->shard_stream));
    }
    for (int q = 0; q < parts; q++) {
      rm_ctx* c = ctxs[q];
      RM_HIP(root, hipSetDevice(c->device));
      if (int rc = shard_copy(root, c, root, static_cast<char*>(root->shard_recv) + part8 * (size_t)q, c->shard_rows8, (size_t)fbs[q]->row_count * (size_t)W * sizeof(uchar4), c->shard_stream)) return rc;
      RM_HIP(root, hipEventRecord(c->shard_ev, c->shard_stream));  // (the root waited for the first record above; this one is the next in stream order)
    }
  } else {
    for (int p = 0; p < parts; p++) {
      rm_ctx* c = ctxs[p];
      RM_HIP(root, hipSetDevice(c->device));
      if (int rc = shard_copy(root, c, root, static_cast<char*>(root->shard_recv) + part8 * (size_t)p, c->shard_rows, (size_t)fbs[p]->row_count * (size_t)W * sizeof(uchar4), c->shard_stream)) return rc;
      RM_HIP(root, hipEventRecord(c->shard_ev, c->shard_stream));
    }
  }
  // 3. the root puts the bytes in image order and brings the canvas to the host (pinned: the copy is asynchronous)
  RM_HIP(root, hipSetDevice(root->device));
  for (int p = 0; p < parts; p++) RM_HIP(root, hipStreamWaitEvent(root->shard_stream, ctxs[p]->shard_ev, 0));
  RM_HIP(root, rm::launch_assemble(root->shard_recv, parts, max_rows, (long long)W * (long long)sizeof(uchar4), H, stripe, root->shard_canvas, root->shard_stream));
  RM_HIP(root, hipMemcpyAsync(root->shard_host, root->shard_canvas, canvas_bytes, hipMemcpyDeviceToHost, root->shard_stream));
  RM_HIP(root, hipEventRecord(root->shard_done, root->shard_stream));
  root->shard_pending = true;
  root->shard_w = W;
  root->shard_h = H;
  return RM_OK;
}

int rm_present_sharded_finish(rm_ctx* const* ctxs, int parts, uint8_t* out_rgba8, size_t out_bytes) {
  rm_ctx* root = (ctxs && parts >= 1) ? ctxs[0] : nullptr;
  if (!root || !out_rgba8) return fail(root, RM_ERR_INVALID, "rm_present_sharded_finish: NULL argument");
  if (!root->shard_pending) return fail(root, RM_ERR_INVALID, "rm_present_sharded_finish: no present was started");
  const size_t bytes = (size_t)root->shard_h * (size_t)root->shard_w * sizeof(uchar4);
  if (out_bytes < bytes) {  // (the present stays pending: the caller can come back with the right buffer)
    char buf[160];
    std::snprintf(buf, sizeof buf, "rm_present_sharded_finish: the pending present is %d x %d (%zu bytes), the buffer holds %zu", root->shard_w, root->shard_h, bytes, out_bytes);
    return fail(root, RM_ERR_INVALID, buf);
  }
  root->shard_pending = false;
  RM_HIP(root, hipSetDevice(root->device));
  RM_HIP(root, hipEventSynchronize(root->shard_done));  // the present's own work only: the renders enqueued since go on
  std::memcpy(out_rgba8, root->shard_host, bytes);
  return RM_OK;
}

int rm_present_sharded(rm_ctx* const* ctxs, rm_fb* const* fbs, int parts, int samples, int dof, uint8_t* out_rgba8, size_t out_bytes) {
  rm_ctx* root = (ctxs && parts >= 1) ? ctxs[0] : nullptr;
  if (!out_rgba8) return fail(root, RM_ERR_INVALID, "rm_present_sharded: NULL argument");
  if (fbs && parts >= 1 && fbs[0] && out_bytes < (size_t)fbs[0]->width * (size_t)fbs[0]->height * sizeof(uchar4))
    return fail(root, RM_ERR_INVALID, "rm_present_sharded: the buffer is smaller than width * height * 4 bytes");  // (before anything is started)
  if (int rc = rm_present_sharded_start(ctxs, fbs, parts, samples, dof)) return rc;
  return rm_present_sharded_finish(ctxs, parts, out_rgba8, out_bytes);
}

// ---- probes ----------------------------------------------------------------------

int rm_probe(rm_ctx* ctx, rm_scene* scene, int what, const float* in, int n, float param, int flags, float* out) {
  if (!ctx || !scene || !in || !out) return fail(ctx, RM_ERR_INVALID, "rm_probe: NULL argument");
  if (scene->ctx != ctx) return fail(ctx, RM_ERR_INVALID, "rm_probe: the scene belongs to another context");
  if (what < RM_PROBE_SDF || what > RM_PROBE_CAST_SHADOW) return fail(ctx, RM_ERR_INVALID, "rm_probe: unknown probe");
  if (n <= 0) return RM_OK;
  static const int in_w[6] = {3, 6, 3, 3, 6, 9}, out_w[6] = {1, 3, 3, 12, 1, 1};
  RM_HIP(ctx, hipSetDevice(ctx->device));
  float *d_in = nullptr, *d_out = nullptr;
  const size_t in_bytes = sizeof(float) * (size_t)in_w[what] * (size_t)n, out_bytes = sizeof(float) * (size_t)out_w[what] * (size_t)n;
  RM_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&d_in), in_bytes));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_out), out_bytes);
  if (e == hipSuccess) e = hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && scene_cull_grid(ctx, scene, flags, 1ll << 40) != RM_OK) e = hipErrorUnknown;  // (a probe is a test's call: with the grid)
  if (e == hipSuccess) {
    ProbeParams P{scene->dev, d_in, d_out, n, what, param, (flags & RM_RENDER_FAST) ? ctx->retire_eps : 0.0f, (flags & RM_RENDER_NO_FAR_JUMP) ? 1 : 0};
    if (P.no_far_jump) P.scene.far_end = 0, P.scene.clear_rho = 0.0f;
    if (flags & RM_RENDER_NO_CULL) P.scene.cull.cells = nullptr;
    e = (flags & RM_RENDER_FAST) ? rm::launch_probe_fast(P, ctx->stream) : ctx->gl_stack ? rm_gl_launch_probe(&P, ctx->stream) : rm::launch_probe_strict(P, ctx->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (e != hipSuccess) return fail(ctx, RM_ERR_DEVICE, std::string("rm_probe: ") + hipGetErrorString(e));
  return RM_OK;
}

static int camera_rng(rm_ctx* ctx, const RmUniforms* u, int width, int height, int what, int count, float* out) {
  if (!ctx || !u || !out || width < 1 || height < 1 || count < 1) return fail(ctx, RM_ERR_INVALID, "probe: bad argument");
  RM_HIP(ctx, hipSetDevice(ctx->device));
  float* d_out = nullptr;
  const size_t bytes = sizeof(float) * (size_t)width * (size_t)height * (size_t)(what == 1 ? count : 8);
  RM_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&d_out), bytes));
  hipError_t e = ctx->gl_stack ? rm_gl_launch_camera_rng(u, width, height, what, count, d_out, ctx->stream) : rm::launch_camera_rng(*u, width, height, what, count, d_out, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_out);
  if (e != hipSuccess) return fail(ctx, RM_ERR_DEVICE, std::string("probe: ") + hipGetErrorString(e));
  return RM_OK;
}

int rm_probe_math(rm_ctx* ctx, int fn, const float* a, const float* b, int n, float* out) {
  if (!ctx || !a || !out || n < 1 || fn < 0 || fn >= RM_MATH_COUNT) return fail(ctx, RM_ERR_INVALID, "rm_probe_math: bad argument");
  const bool two = fn == RM_MATH_POW || fn == RM_MATH_ATAN2 || fn == RM_MATH_POW_PAIR_NM1 || fn == RM_MATH_POW_PAIR_N || fn == RM_MATH_DIV;
  if (two && !b) return fail(ctx, RM_ERR_INVALID, "rm_probe_math: this function takes two arguments");
  RM_HIP(ctx, hipSetDevice(ctx->device));
  float* d = nullptr;  // a | b | out
  const size_t bytes = sizeof(float) * (size_t)n;
  RM_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&d), 3 * bytes));
  hipError_t e = hipMemcpyAsync(d, a, bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && two) e = hipMemcpyAsync(d + n, b, bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess)
    e = ctx->gl_stack ? rm_gl_launch_math_probe(fn, d, two ? d + n : nullptr, n, d + 2 * (size_t)n, ctx->stream)
                      : rm::launch_math_probe(fn, d, two ? d + n : nullptr, n, d + 2 * (size_t)n, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d + 2 * (size_t)n, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(ctx, RM_ERR_DEVICE, std::string("rm_probe_math: ") + hipGetErrorString(e));
  return RM_OK;
}

int rm_probe_camera(rm_ctx* ctx, const RmUniforms* uniforms, int width, int height, float* out) {
  return camera_rng(ctx, uniforms, width, height, 0, 1, out);
}

int rm_probe_rng(rm_ctx* ctx, const RmUniforms* uniforms, int width, int height, int count, float* out) {
  return camera_rng(ctx, uniforms, width, height, 1, count, out);
}

}  // extern "C"
