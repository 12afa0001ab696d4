// rm_napi.cc -- Node N-API binding of libhip_raymarch.so (include/hip_raymarch.h).
//
// Deliberately thin: structs are laid out in JS ArrayBuffers (index.js knows
// the byte offsets of RmUniforms / RmSceneDesc) and handed over as pointers;
// handles travel as N-API externals.  Every function returns a value or
// throws a JS Error carrying rm_last_error(); the render-job layer above
// (index.js doRenderJob) turns those into the reference's error VALUES
// ({success:false, why}), RenderJobExecutor.tsx:112-136.
#include <node_api.h>

#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/hip_raymarch.h"

#define NAPI_OK(call)                                             \
  do {                                                            \
    if ((call) != napi_ok) {                                      \
      napi_throw_error(env, nullptr, "N-API call failed: " #call); \
      return nullptr;                                             \
    }                                                             \
  } while (0)

static napi_value throw_rm(napi_env env, rm_ctx* ctx, const char* what) {
  std::string msg = std::string(what) + ": " + rm_last_error(ctx);
  napi_throw_error(env, "RM_ERROR", msg.c_str());
  return nullptr;
}

// Handles travel as externals that carry their kind, so a scene passed where a framebuffer is expected (or a
// destroyed handle used again) is a TypeError in JS, not a wild pointer in the library.
enum HandleKind : uint32_t { KIND_CTX = 0x726d6301u, KIND_SCENE = 0x726d7302u, KIND_FB = 0x726d6603u };
struct Handle {
  uint32_t kind;
  void* ptr;
};
template <class T> struct KindOf;
template <> struct KindOf<rm_ctx> { static constexpr uint32_t value = KIND_CTX; };
template <> struct KindOf<rm_scene> { static constexpr uint32_t value = KIND_SCENE; };
template <> struct KindOf<rm_fb> { static constexpr uint32_t value = KIND_FB; };

static void finalize_handle(napi_env, void* data, void*) { delete static_cast<Handle*>(data); }

template <class T>
static napi_value make_external(napi_env env, T* p) {
  Handle* h = new Handle{KindOf<T>::value, p};
  napi_value out;
  if (napi_create_external(env, h, finalize_handle, nullptr, &out) != napi_ok) {
    delete h;
    napi_throw_error(env, nullptr, "N-API call failed: napi_create_external");
    return nullptr;
  }
  return out;
}

// the handle of kind T, or nullptr (wrong kind, not an external, or already destroyed); take = the caller destroys it
template <class T>
static T* get_external(napi_env env, napi_value v, bool take = false) {
  void* p = nullptr;
  napi_valuetype t;
  if (napi_typeof(env, v, &t) != napi_ok || t != napi_external) return nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) return nullptr;
  Handle* h = static_cast<Handle*>(p);
  if (h->kind != KindOf<T>::value) return nullptr;
  T* out = static_cast<T*>(h->ptr);
  if (take) h->ptr = nullptr;
  return out;
}

static bool get_buffer(napi_env env, napi_value v, void** data, size_t* len) {
  bool is_ab = false;
  if (napi_is_arraybuffer(env, v, &is_ab) == napi_ok && is_ab) return napi_get_arraybuffer_info(env, v, data, len) == napi_ok;
  bool is_ta = false;
  if (napi_is_typedarray(env, v, &is_ta) == napi_ok && is_ta) {
    napi_typedarray_type t;
    size_t n, off;
    napi_value ab;
    if (napi_get_typedarray_info(env, v, &t, &n, data, &ab, &off) != napi_ok) return false;
    static const size_t width[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
    *len = n * width[t];
    return true;
  }
  return false;
}

static napi_value CtxCreate(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  int32_t device = 0;
  if (argc > 0) napi_get_value_int32(env, argv[0], &device);
  rm_ctx* ctx = nullptr;
  if (rm_ctx_create(device, &ctx) != RM_OK) return throw_rm(env, nullptr, "rm_ctx_create");
  return make_external(env, ctx);
}

static napi_value CtxDestroy(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_ctx_destroy(get_external<rm_ctx>(env, argv[0], true));
  return nullptr;
}

static napi_value Sync(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_ctx* ctx = get_external<rm_ctx>(env, argv[0]);
  if (rm_sync(ctx) != RM_OK) return throw_rm(env, ctx, "rm_sync");
  return nullptr;
}

// sceneCreate(ctx, descBuffer /* RmSceneDesc bytes, prims pointer ignored */, primsBuffer | null)
static napi_value SceneCreate(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_ctx* ctx = get_external<rm_ctx>(env, argv[0]);
  void* d = nullptr;
  size_t n = 0;
  if (!ctx || !get_buffer(env, argv[1], &d, &n) || n != sizeof(RmSceneDesc)) {
    napi_throw_type_error(env, nullptr, "sceneCreate(ctx, desc: ArrayBuffer(sizeof RmSceneDesc), prims)");
    return nullptr;
  }
  RmSceneDesc desc;
  std::memcpy(&desc, d, sizeof desc);
  desc.prims = nullptr;
  void* p = nullptr;
  size_t pn = 0;
  if (argc > 2 && get_buffer(env, argv[2], &p, &pn)) {
    if (pn != sizeof(RmPrim) * (size_t)desc.nprims) {
      napi_throw_type_error(env, nullptr, "sceneCreate: prims buffer must hold nprims rows of 32 bytes");
      return nullptr;
    }
    desc.prims = static_cast<const RmPrim*>(p);
  }
  // sceneCreate(ctx, desc, prims | null, surfaces | null): RmSurface rows (48 bytes each), desc.nsurfaces of them
  desc.surfaces = nullptr;
  void* sf = nullptr;
  size_t sn = 0;
  if (argc > 3 && get_buffer(env, argv[3], &sf, &sn)) {
    if (sn != sizeof(RmSurface) * (size_t)desc.nsurfaces) {
      napi_throw_type_error(env, nullptr, "sceneCreate: surfaces buffer must hold nsurfaces rows of 48 bytes");
      return nullptr;
    }
    desc.surfaces = static_cast<const RmSurface*>(sf);
  } else if (desc.nsurfaces != 0) {
    napi_throw_type_error(env, nullptr, "sceneCreate: desc.nsurfaces is set but no surfaces buffer was given");
    return nullptr;
  }
  rm_scene* scene = nullptr;
  if (rm_scene_create(ctx, &desc, &scene) != RM_OK) return throw_rm(env, ctx, "rm_scene_create");
  return make_external(env, scene);
}

static napi_value SceneDestroy(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_scene_destroy(get_external<rm_scene>(env, argv[0], true));
  return nullptr;
}

// fbCreate(ctx, width, height, rowBegin, rowCount)
static napi_value FbCreate(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value argv[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_ctx* ctx = get_external<rm_ctx>(env, argv[0]);
  int32_t v[4] = {0, 0, 0, 0};
  for (int i = 0; i < 4; i++) napi_get_value_int32(env, argv[1 + i], &v[i]);
  rm_fb* fb = nullptr;
  if (rm_fb_create(ctx, v[0], v[1], v[2], v[3], &fb) != RM_OK) return throw_rm(env, ctx, "rm_fb_create");
  return make_external(env, fb);
}

static napi_value FbClear(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_fb* fb = get_external<rm_fb>(env, argv[0]);
  if (!fb) {
    napi_throw_type_error(env, nullptr, "fbClear(fb): not a framebuffer handle");
    return nullptr;
  }
  if (rm_fb_clear(fb) != RM_OK) {
    napi_throw_error(env, "RM_ERROR", "rm_fb_clear failed");
    return nullptr;
  }
  return nullptr;
}

static napi_value FbDestroy(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_fb_destroy(get_external<rm_fb>(env, argv[0], true));
  return nullptr;
}

// fbDownload(ctx, fb, plane, out: Float32Array(rows*width*4))
static napi_value FbDownload(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_ctx* ctx = get_external<rm_ctx>(env, argv[0]);
  rm_fb* fb = get_external<rm_fb>(env, argv[1]);
  int32_t plane = 0;
  napi_get_value_int32(env, argv[2], &plane);
  void* d = nullptr;
  size_t n = 0;
  if (!fb || !get_buffer(env, argv[3], &d, &n)) {
    napi_throw_type_error(env, nullptr, "fbDownload(ctx, fb, plane, out: Float32Array)");
    return nullptr;
  }
  const size_t need = (size_t)rm_fb_rows(fb) * (size_t)rm_fb_width(fb) * 16;
  if (n < need) {
    napi_throw_range_error(env, nullptr, "fbDownload: out is smaller than rows * width * 4 floats");
    return nullptr;
  }
  if (rm_fb_download(fb, plane, static_cast<float*>(d)) != RM_OK) return throw_rm(env, ctx, "rm_fb_download");
  return nullptr;
}

// present(ctx, fb, samples, out: Uint8Array(width * height * 4)): display.frag on the GPU, RGBA8, row 0 = bottom
static napi_value Present(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_ctx* ctx = get_external<rm_ctx>(env, argv[0]);
  rm_fb* fb = get_external<rm_fb>(env, argv[1]);
  int32_t samples = 1;
  napi_get_value_int32(env, argv[2], &samples);
  void* d = nullptr;
  size_t n = 0;
  if (!ctx || !fb || !get_buffer(env, argv[3], &d, &n)) {
    napi_throw_type_error(env, nullptr, "present(ctx, fb, samples, out: Uint8Array)");
    return nullptr;
  }
  if (n < (size_t)rm_fb_width(fb) * (size_t)rm_fb_height(fb) * 4) {
    napi_throw_range_error(env, nullptr, "present: out is smaller than width * height * 4 bytes");
    return nullptr;
  }
  if (rm_present(ctx, fb, samples, static_cast<uint8_t*>(d)) != RM_OK) return throw_rm(env, ctx, "rm_present");
  return nullptr;
}

// fbCreateStriped(ctx, width, height, stripeRows, parts, part): the stripes k with k % parts == part of a width x height image,
// planes owned by the library (rm_fb_create_striped) -- what one GPU of a sharded frame holds
static napi_value FbCreateStriped(napi_env env, napi_callback_info info) {
  size_t argc = 6;
  napi_value argv[6];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_ctx* ctx = get_external<rm_ctx>(env, argv[0]);
  if (!ctx || argc < 6) {
    napi_throw_type_error(env, nullptr, "fbCreateStriped(ctx, width, height, stripeRows, parts, part)");
    return nullptr;
  }
  int32_t v[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 5; i++) napi_get_value_int32(env, argv[1 + i], &v[i]);
  rm_fb* fb = nullptr;
  if (rm_fb_create_striped(ctx, v[0], v[1], v[2], v[3], v[4], nullptr, nullptr, nullptr, &fb) != RM_OK) return throw_rm(env, ctx, "rm_fb_create_striped");
  return make_external(env, fb);
}

// fbRows(fb): image rows the framebuffer holds
static napi_value FbRows(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1], out;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_fb* fb = get_external<rm_fb>(env, argv[0]);
  if (!fb) {
    napi_throw_type_error(env, nullptr, "fbRows(fb): not a framebuffer handle");
    return nullptr;
  }
  NAPI_OK(napi_create_int32(env, rm_fb_rows(fb), &out));
  return out;
}

// setSamplesInFlight(ctx, n): rm_ctx_set_samples_in_flight
static napi_value SetSamplesInFlight(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_ctx* ctx = get_external<rm_ctx>(env, argv[0]);
  int32_t n = 1;
  if (argc > 1) napi_get_value_int32(env, argv[1], &n);
  if (!ctx) {
    napi_throw_type_error(env, nullptr, "setSamplesInFlight(ctx, n): not a context handle");
    return nullptr;
  }
  if (rm_ctx_set_samples_in_flight(ctx, n) != RM_OK) return throw_rm(env, ctx, "rm_ctx_set_samples_in_flight");
  return nullptr;
}

// The present of a frame whose stripes this process renders on several GPUs (part p on ctxs[p]); RGBA8, row 0 = bottom.
//   presentSharded(ctxs: ctx[], fbs: fb[], samples, dof: boolean, out: Uint8Array(width * height * 4)) = rm_present_sharded
//   presentShardedStart(ctxs, fbs, samples, dof)   = rm_present_sharded_start: snapshot and send, returns at once
//   presentShardedFinish(ctxs, out: Uint8Array)     = rm_present_sharded_finish: the canvas of the present that was started
static bool handle_arrays(napi_env env, napi_value a_ctxs, napi_value a_fbs, rm_ctx** ctxs, rm_fb** fbs, uint32_t* count, const char* who) {
  uint32_t n = 0, m = 0;
  bool is_a = false, is_b = true;
  if (napi_is_array(env, a_ctxs, &is_a) != napi_ok || !is_a || napi_get_array_length(env, a_ctxs, &n) != napi_ok || n == 0 || n > 64 ||
      (fbs && (napi_is_array(env, a_fbs, &is_b) != napi_ok || !is_b || napi_get_array_length(env, a_fbs, &m) != napi_ok || n != m))) {
    napi_throw_type_error(env, nullptr, (std::string(who) + ": ctxs (and fbs) must be arrays of the same length (1..64)").c_str());
    return false;
  }
  for (uint32_t i = 0; i < n; i++) {
    napi_value a, b;
    if (napi_get_element(env, a_ctxs, i, &a) != napi_ok) return false;
    ctxs[i] = get_external<rm_ctx>(env, a);
    if (fbs) {
      if (napi_get_element(env, a_fbs, i, &b) != napi_ok) return false;
      fbs[i] = get_external<rm_fb>(env, b);
    }
    if (!ctxs[i] || (fbs && !fbs[i])) {
      napi_throw_type_error(env, nullptr, (std::string(who) + ": wrong or destroyed handle in ctxs / fbs").c_str());
      return false;
    }
  }
  *count = n;
  return true;
}

static napi_value PresentShardedImpl(napi_env env, napi_callback_info info, bool start, bool finish) {
  size_t argc = 5;
  napi_value argv[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  const char* who = start && finish ? "presentSharded" : start ? "presentShardedStart" : "presentShardedFinish";
  const size_t need = start && finish ? 5 : start ? 4 : 2;
  if (argc < need) {
    napi_throw_type_error(env, nullptr, (std::string(who) + ": too few arguments").c_str());
    return nullptr;
  }
  rm_ctx* ctxs[64];
  rm_fb* fbs[64];
  uint32_t n = 0;
  if (!handle_arrays(env, argv[0], start ? argv[1] : nullptr, ctxs, start ? fbs : nullptr, &n, who)) return nullptr;
  void* d = nullptr;
  size_t len = 0;
  if (finish && !get_buffer(env, argv[start ? 4 : 1], &d, &len)) {
    napi_throw_type_error(env, nullptr, (std::string(who) + ": out must be a Uint8Array").c_str());
    return nullptr;
  }
  if (start && finish && len < (size_t)rm_fb_width(fbs[0]) * (size_t)rm_fb_height(fbs[0]) * 4) {
    napi_throw_range_error(env, nullptr, "presentSharded: out is smaller than width * height * 4 bytes");
    return nullptr;
  }
  if (start) {
    int32_t samples = 1;
    napi_get_value_int32(env, argv[2], &samples);
    bool dof = false;
    napi_get_value_bool(env, argv[3], &dof);
    if (rm_present_sharded_start(ctxs, fbs, (int)n, samples, dof ? 1 : 0) != RM_OK) return throw_rm(env, ctxs[0], "rm_present_sharded_start");
  }
  // (presentShardedFinish: the library knows the canvas of the pending present and refuses a smaller `out`, leaving the present pending)
  if (finish && rm_present_sharded_finish(ctxs, (int)n, static_cast<uint8_t*>(d), len) != RM_OK) return throw_rm(env, ctxs[0], "rm_present_sharded_finish");
  return nullptr;
}
static napi_value PresentSharded(napi_env env, napi_callback_info info) { return PresentShardedImpl(env, info, true, true); }
static napi_value PresentShardedStart(napi_env env, napi_callback_info info) { return PresentShardedImpl(env, info, true, false); }
static napi_value PresentShardedFinish(napi_env env, napi_callback_info info) { return PresentShardedImpl(env, info, false, true); }

// renderSample(ctx, scene, fb, uniforms: ArrayBuffer(sizeof RmUniforms), tile: Int32Array(4) | null, flags)
static napi_value RenderSample(napi_env env, napi_callback_info info) {
  size_t argc = 6;
  napi_value argv[6];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_ctx* ctx = get_external<rm_ctx>(env, argv[0]);
  rm_scene* scene = get_external<rm_scene>(env, argv[1]);
  rm_fb* fb = get_external<rm_fb>(env, argv[2]);
  void* u = nullptr;
  size_t un = 0;
  if (!ctx || !scene || !fb) {
    napi_throw_type_error(env, nullptr, "renderSample(ctx, scene, fb, ...): wrong or destroyed handle");
    return nullptr;
  }
  if (!get_buffer(env, argv[3], &u, &un) || un != sizeof(RmUniforms)) {
    napi_throw_type_error(env, nullptr, "renderSample: uniforms must be an ArrayBuffer of sizeof(RmUniforms) bytes");
    return nullptr;
  }
  RmRect tile;
  const RmRect* tp = nullptr;
  void* t = nullptr;
  size_t tn = 0;
  if (argc > 4 && get_buffer(env, argv[4], &t, &tn) && tn == sizeof(RmRect)) {
    std::memcpy(&tile, t, sizeof tile);
    tp = &tile;
  }
  int32_t flags = 0;
  if (argc > 5) napi_get_value_int32(env, argv[5], &flags);
  if (rm_render_sample(ctx, scene, fb, static_cast<const RmUniforms*>(u), tp, flags) != RM_OK) return throw_rm(env, ctx, "rm_render_sample");
  return nullptr;
}

// renderSamples(ctx, scene, fb, uniforms: ArrayBuffer(sizeof RmUniforms), randNoise: Float32Array(2 * count), tile: Int32Array(4) | null, flags)
// = rm_render_samples: the samples between two yields of a job in one call (RenderJobExecutor.tsx:160-222)
static napi_value RenderSamples(napi_env env, napi_callback_info info) {
  size_t argc = 7;
  napi_value argv[7];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  rm_ctx* ctx = get_external<rm_ctx>(env, argv[0]);
  rm_scene* scene = get_external<rm_scene>(env, argv[1]);
  rm_fb* fb = get_external<rm_fb>(env, argv[2]);
  void* u = nullptr;
  size_t un = 0;
  if (!ctx || !scene || !fb) {
    napi_throw_type_error(env, nullptr, "renderSamples(ctx, scene, fb, ...): wrong or destroyed handle");
    return nullptr;
  }
  if (!get_buffer(env, argv[3], &u, &un) || un != sizeof(RmUniforms)) {
    napi_throw_type_error(env, nullptr, "renderSamples: uniforms must be an ArrayBuffer of sizeof(RmUniforms) bytes");
    return nullptr;
  }
  void* rn = nullptr;
  size_t rnn = 0;
  if (!get_buffer(env, argv[4], &rn, &rnn) || rnn % (2 * sizeof(float)) != 0 || rnn > (size_t)1 << 24) {
    napi_throw_type_error(env, nullptr, "renderSamples: randNoise must be a Float32Array of 2 * count values");
    return nullptr;
  }
  RmRect tile;
  const RmRect* tp = nullptr;
  void* t = nullptr;
  size_t tn = 0;
  if (argc > 5 && get_buffer(env, argv[5], &t, &tn) && tn == sizeof(RmRect)) {
    std::memcpy(&tile, t, sizeof tile);
    tp = &tile;
  }
  int32_t flags = 0;
  if (argc > 6) napi_get_value_int32(env, argv[6], &flags);
  if (rm_render_samples(ctx, scene, fb, static_cast<const RmUniforms*>(u), static_cast<const float*>(rn), (int)(rnn / (2 * sizeof(float))), tp, flags) != RM_OK)
    return throw_rm(env, ctx, "rm_render_samples");
  return nullptr;
}

static napi_value Sizes(napi_env env, napi_callback_info) {
  napi_value o, v;
  NAPI_OK(napi_create_object(env, &o));
  const struct { const char* k; uint32_t v; } items[] = {
      {"RmUniforms", (uint32_t)sizeof(RmUniforms)}, {"RmSceneDesc", (uint32_t)sizeof(RmSceneDesc)}, {"RmPrim", (uint32_t)sizeof(RmPrim)},
      {"RmMaterial", (uint32_t)sizeof(RmMaterial)}, {"RmRect", (uint32_t)sizeof(RmRect)}, {"RmSurface", (uint32_t)sizeof(RmSurface)}, {"abi", (uint32_t)rm_abi_version()}};
  for (const auto& it : items) {
    NAPI_OK(napi_create_uint32(env, it.v, &v));
    NAPI_OK(napi_set_named_property(env, o, it.k, v));
  }
  return o;
}

static napi_value Init(napi_env env, napi_value exports) {
  const struct { const char* name; napi_callback fn; } fns[] = {
      {"ctxCreate", CtxCreate}, {"ctxDestroy", CtxDestroy}, {"sync", Sync}, {"sceneCreate", SceneCreate}, {"sceneDestroy", SceneDestroy},
      {"fbCreate", FbCreate}, {"fbClear", FbClear}, {"fbDestroy", FbDestroy}, {"fbDownload", FbDownload}, {"present", Present}, {"renderSample", RenderSample}, {"renderSamples", RenderSamples},
      {"fbCreateStriped", FbCreateStriped}, {"fbRows", FbRows}, {"setSamplesInFlight", SetSamplesInFlight}, {"presentSharded", PresentSharded}, {"presentShardedStart", PresentShardedStart}, {"presentShardedFinish", PresentShardedFinish},
      {"sizes", Sizes}};
  for (const auto& f : fns) {
    napi_value fn;
    if (napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &fn) != napi_ok) return nullptr;
    if (napi_set_named_property(env, exports, f.name, fn) != napi_ok) return nullptr;
  }
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
