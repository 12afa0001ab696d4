// index.js -- JavaScript host of the render-job API on top of the N-API addon.
//
// Mirrors client/src/renderer/RenderJobExecutor.tsx:77-341 of the reference:
// `doRenderJob(schema, context)` resolves to a generator factory; the generator
// yields every `sampleYieldInterval` samples after calling `present`, and
// returns {success:true} or {success:false, why} -- errors are values.  The
// gl.* block of the reference (:181-326) is one native call per sample.
// The RenderJobSchema is the reference's (RenderJobSchema.tsx:17-86) plus
// `sdfScene`, a scene built with the composition API below (a HIP kernel
// cannot consume GLSL text; the composer emits both back ends).
// Types: index.d.ts.  (The container has Node 12 / N-API 8 and no tsc, so the
// host is JavaScript with hand-written typings.)
"use strict";
// Process environment the device path depends on (read by the HIP / HSA runtime when it starts, i.e. when the addon
// first touches the GPU): more hardware queues than the runtime's default 4, so that the side streams of the samples in
// flight do not share one and serialise, and dmabuf IPC.  Set unless the user has set them (native.py does the same).
if (process.env.GPU_MAX_HW_QUEUES === undefined) process.env.GPU_MAX_HW_QUEUES = "8";
if (process.env.HSA_ENABLE_IPC_MODE_LEGACY === undefined) process.env.HSA_ENABLE_IPC_MODE_LEGACY = "0";
const addon = require("./rm_napi.node");

const RM = {
  MAX_BOUNCES: 10, MAX_LIGHTS: 10,
  SCENE_TABLE: 0, SCENE_MANDELBULB: 1, SCENE_SPHERE_GRID: 2, SCENE_SPHERE_LATTICE: 3, SCENE_MENGER: 4, SCENE_KIFS_TREE: 5, SCENE_KIFS_BOX: 6,
  PRIM_SPHERE: 0, PRIM_BOX: 1, PRIM_REPEAT: 2, PRIM_FOLD: 3, PRIM_KIND: 4, PRIM_TORUS: 5, PRIM_CYLINDER: 6, PRIM_PLANE: 7,
  OP_UNION: 0, OP_SMOOTH_UNION: 1, OP_SUBTRACT: 2, OP_INTERSECT: 3, OP_SMOOTH_SUBTRACT: 4, OP_SMOOTH_INTERSECT: 5,
  RENDER_STRICT: 0, RENDER_FAST: 1, RENDER_COLOR_ONLY: 2, RENDER_MEGAKERNEL: 4, RENDER_WAVEFRONT: 16, RENDER_NO_OVERLAP: 32, RENDER_NO_FAR_JUMP: 64, RENDER_NO_CULL: 128,
};

// ---- struct layouts (include/hip_raymarch.h; all fields are 4 bytes) ----------------
const U_FIELDS = [
  ["blendWithPreviousFactor", 1, "f"], ["randNoise", 2, "f"], ["position", 3, "f"], ["rotation", 16, "f"], ["dofAmount", 1, "f"],
  ["dofFocalPlaneDistance", 1, "f"], ["cameraMode", 1, "i"], ["fov", 1, "f"], ["reflections", 1, "f"], ["raymarchingSteps", 1, "f"],
  ["indirectLightingRaymarchingSteps", 1, "f"], ["aspect", 1, "f"], ["fogDensity", 1, "f"], ["exposure", 1, "f"],
  ["raymarchingStepCountsArray", 10, "f"], ["blendMode", 1, "i"], ["renderMode", 1, "i"], ["lightPositions", 30, "f"],
  ["lightColors", 30, "f"], ["lightSizes", 10, "f"], ["lightCount", 1, "i"], ["showDofFocalPlane", 1, "i"],
];
const U_OFFSET = {};
let uSize = 0;
for (const [name, n] of U_FIELDS) { U_OFFSET[name] = uSize; uSize += 4 * n; }
if (uSize !== addon.sizes().RmUniforms) throw new Error("RmUniforms layout mismatch: " + uSize + " != " + addon.sizes().RmUniforms);

function packUniforms(values) {
  const buf = new ArrayBuffer(uSize);
  const f = new Float32Array(buf), i = new Int32Array(buf);
  for (const [name, n, t] of U_FIELDS) {
    const v = values[name];
    if (v === undefined) continue;
    const at = U_OFFSET[name] / 4, arr = n === 1 && !Array.isArray(v) ? [v] : v;
    for (let k = 0; k < arr.length && k < n; k++) (t === "f" ? f : i)[at + k] = arr[k];
  }
  return buf;
}

// ---- scene composition --------------------------------------------------------------
const DEFAULT_MATERIAL = {  // Validate.tsx:18-51
  diffuse: [0.6, 0.6, 0.6], diffuse_cutoff: 35, specular: [0.6, 0.6, 0.6], specular_cutoff: 35, roughness: 0.2, subsurface: 11111115,
  subsurface_color: [1, 1, 1], ior: 100, sky_color: [0.7, 0.8, 1.0], sky_floor: 0.2, sky_scale: 2, sky_radius: 36, sky_axis: 1,
};
const glf = (x) => { const s = String(Math.fround(x)); return /[.e]/.test(s) ? s : s + ".0"; };
const glv = (v) => `vec3(${glf(v[0])}, ${glf(v[1])}, ${glf(v[2])})`;

class Scene {
  constructor(kind, params, material) { this.kind = kind; this.params = params || []; this.material = Object.assign({}, DEFAULT_MATERIAL, material || {}); this.prims = []; this.surfaces = []; }
  // RmSceneDesc bytes: kind, nprims, prims pointer (8, filled natively), params[16], material (22 x 4), nsurfaces, reserved, surfaces pointer (8, filled natively)
  desc() {
    const buf = new ArrayBuffer(addon.sizes().RmSceneDesc);
    const f = new Float32Array(buf), i = new Int32Array(buf);
    i[0] = this.kind; i[1] = this.prims.length;
    for (let k = 0; k < this.params.length; k++) f[4 + k] = this.params[k];
    const m = this.material, at = 20;
    f.set([...m.diffuse, m.diffuse_cutoff, ...m.specular, m.specular_cutoff, m.roughness, m.subsurface, ...m.subsurface_color, m.ior,
           ...m.sky_color, m.sky_floor, m.sky_scale, m.sky_radius], at);
    i[at + 20] = m.sky_axis;
    i[at + 22] = this.surfaces.length;
    let prims = null;
    if (this.prims.length) {
      prims = new ArrayBuffer(32 * this.prims.length);
      const pf = new Float32Array(prims), pi = new Int32Array(prims);
      this.prims.forEach((p, n) => { pi[8 * n] = p.prim | (p.op << 8) | ((p.surface || 0) << 16); pf[8 * n + 1] = p.k; pf.set(p.center, 8 * n + 2); pf.set(p.size, 8 * n + 5); });
    }
    let surfaces = null;
    if (this.surfaces.length) {  // RmSurface rows: diffuse[3], roughness, specular[3], subsurface, subsurface_color[3], ior
      surfaces = new ArrayBuffer(48 * this.surfaces.length);
      const sf = new Float32Array(surfaces);
      this.surfaces.forEach((u, n) => sf.set([...u.diffuse, u.roughness, ...u.specular, u.subsurface, ...u.subsurface_color, u.ior], 12 * n));
    }
    return { desc: buf, prims, surfaces };
  }
  key() { const d = this.desc(); return Buffer.from(d.desc).toString("hex") + (d.prims ? Buffer.from(d.prims).toString("hex") : "") + (d.surfaces ? Buffer.from(d.surfaces).toString("hex") : ""); }
}

class CsgScene extends Scene {
  constructor(material) { super(RM.SCENE_TABLE, [], material); this._op = RM.OP_UNION; this._k = 0; }
  union() { this._op = RM.OP_UNION; this._k = 0; return this; }
  smoothUnion(k) { this._op = RM.OP_SMOOTH_UNION; this._k = k; return this; }
  subtract() { this._op = RM.OP_SUBTRACT; this._k = 0; return this; }
  intersect() { this._op = RM.OP_INTERSECT; this._k = 0; return this; }
  // ABI 8: the smooth forms of the other two operators (h = clamp(0.5 - 0.5 (d +- di) / k, 0, 1); mix(d, -+di, h) + k h (1 - h)) ...
  smoothSubtract(k) { this._op = RM.OP_SMOOTH_SUBTRACT; this._k = k; return this; }
  smoothIntersect(k) { this._op = RM.OP_SMOOTH_INTERSECT; this._k = k; return this; }
  // `surface` (optional): material values of this shape's own -- {diffuse, specular, roughness, subsurface, subsurface_color, ior}, missing
  // ones from the reference defaults (Validate.tsx:18-51); the material functions then depend on the position (materialGlsl)
  _surface(surface) {
    if (!surface) return 0;
    const u = { diffuse: [0.6, 0.6, 0.6], specular: [0.6, 0.6, 0.6], roughness: 0.2, subsurface: 11111115, subsurface_color: [1, 1, 1], ior: 100, ...surface };
    const same = (a, b) => JSON.stringify(a) === JSON.stringify(b);
    let k = this.surfaces.findIndex((v) => same(v, u));
    if (k < 0) { if (this.surfaces.length >= 15) throw new RangeError("a scene has at most 15 surfaces"); this.surfaces.push(u); k = this.surfaces.length - 1; }
    return k + 1;
  }
  sphere(center, radius, surface) { this.prims.push({ prim: RM.PRIM_SPHERE, op: this._op, k: this._k, center, size: [radius, 0, 0], surface: this._surface(surface) }); return this; }
  box(center, half, surface) { this.prims.push({ prim: RM.PRIM_BOX, op: this._op, k: this._k, center, size: half, surface: this._surface(surface) }); return this; }
  // ... and three more shapes about `center`, axis y: a ring, a capped cylinder, a half space (unit normal); their GLSL text is the
  // Python composer's (scene.py CsgScene: rmTorus, rmCylinder, rmPlane): glsl() here throws
  torus(center, majorRadius, minorRadius, surface) { this.prims.push({ prim: RM.PRIM_TORUS, op: this._op, k: this._k, center, size: [majorRadius, minorRadius, 0], surface: this._surface(surface) }); return this; }
  cylinder(center, radius, halfHeight, surface) { this.prims.push({ prim: RM.PRIM_CYLINDER, op: this._op, k: this._k, center, size: [radius, halfHeight, 0], surface: this._surface(surface) }); return this; }
  plane(point, normal, surface) { this.prims.push({ prim: RM.PRIM_PLANE, op: this._op, k: this._k, center: point, size: normal, surface: this._surface(surface) }); return this; }
  // a shape whose distance term is another scene kind's own estimator at p - center (RM_PRIM_KIND, include/hip_raymarch.h): a Mandelbulb
  // (or a kind 3 lattice), folded like a sphere or a box -- new CsgScene().shape(new Mandelbulb()).intersect().box(...) is a Mandelbulb cut
  // by a box.  One kind with one parameter set per table (they travel in the scene's parameter block).  The GLSL text of such a table
  // is the Python composer's (scene.py CsgScene.shape); glsl() here throws.
  shape(scene, center = [0, 0, 0], surface) {
    if (scene.kind !== RM.SCENE_MANDELBULB && scene.kind !== 3) throw new TypeError("shape(): the kinds a table row can evaluate are the Mandelbulb and the sphere lattice");
    if (this._kind !== undefined && (this._kind !== scene.kind || JSON.stringify(this.params) !== JSON.stringify(scene.params))) throw new TypeError("shape(): a table evaluates ONE kind with one set of parameters");
    this._kind = scene.kind; this.params = scene.params.slice();
    this.prims.push({ prim: RM.PRIM_KIND, op: this._op, k: this._k, center, size: [scene.kind, 0, 0], surface: this._surface(surface) });
    return this;
  }
  // domain operators (include/hip_raymarch.h): they transform the point the FOLLOWING primitives are evaluated at
  repeat(period) { this.prims.push({ prim: RM.PRIM_REPEAT, op: 0, k: 0, center: [0, 0, 0], size: period }); return this; }
  fold(scale, offset, angles = [0, 0, 0]) { this.prims.push({ prim: RM.PRIM_FOLD, op: 0, k: scale, center: offset, size: angles }); return this; }
  glsl() {  // the reference's scene contract: float sdf(vec3); helpers sdfSphere/sdBox come from raymarcher.frag:74,108
    const lines = [];
    if (this.prims.some((n) => n.prim === RM.PRIM_KIND)) throw new Error("glsl(): a table with kind rows gets its text from the Python composer (scene.py CsgScene.shape)");
    if (this.prims.some((n) => n.prim > RM.PRIM_KIND || n.op > RM.OP_INTERSECT)) throw new Error("glsl(): a table with ABI 8's shapes or smooth operators gets its text from the Python composer (scene.py CsgScene)");
    const isShape = (n) => n.prim === RM.PRIM_SPHERE || n.prim === RM.PRIM_BOX;
    const shapes = this.prims.filter(isShape), domain = shapes.length !== this.prims.length;
    if (shapes.slice(1).some((p) => p.op === RM.OP_SMOOTH_UNION))
      lines.push("float rmSmoothUnion(float d1, float d2, float k) { float h = clamp(0.5 + 0.5 * (d2 - d1) / k, 0.0, 1.0); return mix(d2, d1, h) - k * h * (1.0 - h); }");
    if (this.prims.some((n) => n.prim === RM.PRIM_FOLD))
      lines.push("vec3 rmFold(vec3 q, float scale, vec3 off, vec3 ang) { q = q / scale; q = abs(q) - off; float c; float s; float nx; float ny;" +
        " c = cos(ang.x); s = sin(ang.x); nx = q.x * c + q.y * -s; ny = q.x * s + q.y * c; q.x = nx; q.y = ny;" +
        " c = cos(ang.y); s = sin(ang.y); nx = q.y * c + q.z * -s; ny = q.y * s + q.z * c; q.y = nx; q.z = ny;" +
        " c = cos(ang.z); s = sin(ang.z); nx = q.x * c + q.z * -s; ny = q.x * s + q.z * c; q.x = nx; q.z = ny; return q; }");
    lines.push("float sdf(vec3 p) {");
    const q = domain ? "q" : "p";
    if (domain) lines.push("  vec3 q = p; float factor = 1.0; float d;");
    let first = true;
    this.prims.forEach((n) => {
      if (n.prim === RM.PRIM_REPEAT) { lines.push(`  q = mod(q + 0.5 * ${glv(n.size)}, ${glv(n.size)}) - 0.5 * ${glv(n.size)};`); return; }
      if (n.prim === RM.PRIM_FOLD) { lines.push(`  q = rmFold(q, ${glf(n.k)}, ${glv(n.center)}, ${glv(n.size)}); factor = factor * ${glf(n.k)};`); return; }
      let e = n.prim === RM.PRIM_SPHERE ? `sdfSphere(${q}, ${glv(n.center)}, ${glf(n.size[0])})` : `sdBox(${q} - ${glv(n.center)}, ${glv(n.size)})`;
      if (domain) e = `(${e} * factor)`;
      if (first) { lines.push(domain ? `  d = ${e};` : `  float d = ${e};`); first = false; }
      else if (n.op === RM.OP_UNION) lines.push(`  d = min(d, ${e});`);
      else if (n.op === RM.OP_SMOOTH_UNION) lines.push(`  d = rmSmoothUnion(d, ${e}, ${glf(n.k)});`);
      else if (n.op === RM.OP_SUBTRACT) lines.push(`  d = max(d, -${e});`);
      else lines.push(`  d = max(d, ${e});`);
    });
    lines.push("  return d;", "}");
    if (this.prims.some((n) => n.surface)) lines.push(this.materialGlsl());
    return lines.join("\n");
  }
  // The seven material functions of a scene whose shapes name surfaces (scene.py material_glsl, statement for statement): at
  // `position` the values of the shape row whose distance term there is the smallest (the earliest on a tie; a NaN never wins).
  materialGlsl() {
    const isShape = (n) => n.prim === RM.PRIM_SPHERE || n.prim === RM.PRIM_BOX;
    const domain = this.prims.filter(isShape).length !== this.prims.length, q = domain ? "q" : "p", m = this.material;
    const all = [{ diffuse: m.diffuse, specular: m.specular, roughness: m.roughness, subsurface: m.subsurface, subsurface_color: m.subsurface_color, ior: m.ior }, ...this.surfaces];
    const n = all.length, lines = ["int rmSurfaceIndex(vec3 p) {"];
    if (domain) lines.push("  vec3 q = p; float factor = 1.0;");
    lines.push("  float best = 0.0; float di; int surface = 0;");
    let first = true;
    this.prims.forEach((node) => {
      if (node.prim === RM.PRIM_REPEAT) { lines.push(`  q = mod(q + 0.5 * ${glv(node.size)}, ${glv(node.size)}) - 0.5 * ${glv(node.size)};`); return; }
      if (node.prim === RM.PRIM_FOLD) { lines.push(`  q = rmFold(q, ${glf(node.k)}, ${glv(node.center)}, ${glv(node.size)}); factor = factor * ${glf(node.k)};`); return; }
      let e = node.prim === RM.PRIM_SPHERE ? `sdfSphere(${q}, ${glv(node.center)}, ${glf(node.size[0])})` : `sdBox(${q} - ${glv(node.center)}, ${glv(node.size)})`;
      if (domain) e = `(${e} * factor)`;
      if (first) { lines.push(`  best = ${e}; surface = ${node.surface || 0};`); first = false; }
      else lines.push(`  di = ${e}; if (di < best) { best = di; surface = ${node.surface || 0}; }`);
    });
    lines.push("  return surface;", "}");
    const table = (name, typ, vals) => `const ${typ} ${name}[${n}] = ${typ}[${n}](${vals.join(", ")});`;
    lines.push(table("rmDiffuse", "vec3", all.map((u) => glv(u.diffuse))), table("rmSpecular", "vec3", all.map((u) => glv(u.specular))),
      table("rmSubsurfaceColor", "vec3", all.map((u) => glv(u.subsurface_color))), table("rmRoughness", "float", all.map((u) => glf(u.roughness))),
      table("rmSubsurface", "float", all.map((u) => glf(u.subsurface))), table("rmIor", "float", all.map((u) => glf(u.ior))),
      `vec3 sceneDiffuseColor(vec3 position) { if (length(position) > ${glf(m.diffuse_cutoff)}) return vec3(0.0); return rmDiffuse[rmSurfaceIndex(position)]; }`,
      `vec3 sceneSpecularColor(vec3 position) { if (length(position) > ${glf(m.specular_cutoff)}) return vec3(0.0); return rmSpecular[rmSurfaceIndex(position)]; }`,
      "float sceneSpecularRoughness(vec3 position) { return rmRoughness[rmSurfaceIndex(position)]; }",
      "float sceneSubsurfaceScattering(vec3 position) { return rmSubsurface[rmSurfaceIndex(position)]; }",
      "vec3 sceneSubsurfaceScatteringColor(vec3 position) { return rmSubsurfaceColor[rmSurfaceIndex(position)]; }",
      "float sceneIOR(vec3 position) { return rmIor[rmSurfaceIndex(position)]; }",
      `vec3 sceneEmission(vec3 position) { float d = max(normalize(position).${"xyz"[m.sky_axis]}, ${glf(m.sky_floor)}); vec3 brightColor = ${glv(m.sky_color)} * d * 1.0;` +
      ` return (length(position) > ${glf(m.sky_radius)}) ? (brightColor * ${glf(m.sky_scale)}) : vec3(0.0); }`);
    return lines.join("\n");
  }
}
const singleSphere = (center = [0, 0, 0], radius = 1, material) => new CsgScene(material).sphere(center, radius);
class Mandelbulb extends Scene {
  constructor(power = 8, iterations = 8, bailout = 2, material) { super(RM.SCENE_MANDELBULB, [power, iterations, bailout], material); }
}

// ---- host side of the job -------------------------------------------------------------
function* halton(b) {  // util/Halton.tsx:1-19, value for value
  for (let i = 1; ; i++) { let num = 0, den = 1; for (let k = i; k > 0; k = Math.floor(k / b)) { num = num * b + (k % b); den *= b; } yield num / den; }
}
let halton2 = halton(2), halton3 = halton(3);  // module-level like RenderJobExecutor.tsx:70-71: continues across jobs
const resetHalton = () => { halton2 = halton(2); halton3 = halton(3); };

function uniformsFromSchema(schema, randNoise) {  // RenderJobExecutor.tsx:212-297
  const cam = schema.camera, r = schema.render, counts = schema.reflectionIterationCounts;
  if (counts.length > RM.MAX_BOUNCES || schema.lights.length > RM.MAX_LIGHTS) throw new Error("at most 10 bounces / 10 lights");
  const mode = cam.mode.type;
  const pos = [], col = [], size = [];
  for (const l of schema.lights) { pos.push(...(l.type === "point" ? l.position : l.direction)); col.push(...l.color); size.push(l.type === "point" ? l.size : 0); }
  return packUniforms({
    blendWithPreviousFactor: r.blendWithPreviousFrameFactor, randNoise, position: cam.position, rotation: Array.from(cam.rotation),
    dofAmount: schema.dof.amount, dofFocalPlaneDistance: schema.dof.distance,
    cameraMode: ["perspective", "orthographic", "panoramic"].indexOf(mode),
    fov: mode === "perspective" ? cam.mode.fov : mode === "orthographic" ? cam.mode.size : 1,
    reflections: counts.length, raymarchingSteps: counts[0], indirectLightingRaymarchingSteps: counts.length > 1 ? counts[1] : counts[0],
    aspect: r.width / r.height, fogDensity: schema.fogDensity, exposure: r.exposure / r.samplesPerPixel,
    raymarchingStepCountsArray: counts, blendMode: r.blendMode === "additive" ? 1 : 0, renderMode: r.renderMode === "preview" ? 1 : 0,
    lightPositions: pos, lightColors: col, lightSizes: size, lightCount: schema.lights.length, showDofFocalPlane: schema.dof.showFocusedArea ? 1 : 0,
  });
}

// :167-180.  The reference passes (x1, y1, x2, y2) to gl.scissor(x, y, width, height): identical to the intended tile
// for subdivisions <= 2, over-covering from 3 on.  Default: the intent; render.referenceScissor = true: the reference's rectangle.
function tileRect(schema, xp, yp) {
  const r = schema.render, n = r.subdivisions;
  const x1 = Math.floor((r.width / n) * xp), y1 = Math.floor((r.height / n) * yp);
  const x2 = Math.ceil((r.width / n) * (xp + 1)), y2 = Math.ceil((r.height / n) * (yp + 1));
  if (r.referenceScissor) return new Int32Array([x1, y1, Math.min(x2, r.width - x1), Math.min(y2, r.height - y1)]);
  return new Int32Array([x1, y1, x2 - x1, y2 - y1]);
}

// A scene cache keeps this many entries (a Map iterates in insertion order: the first key is the least recently used).  The
// reference's programCache grows with every edit of the shader text, for a page's lifetime; a long-running host that animates scene
// parameters would otherwise keep a device table per distinct scene (the culling grids of long CSG tables -- 31.5 MB per 64 rows --
// are bounded by the library itself, by bytes: rm_ctx_cull_stats).  A scene a job still renders is pinned (doRenderJob: jobs yield
// between samples) and is never the one that goes.
const SCENE_CACHE_ENTRIES = 64;
function evictScenes(map, destroy, pins, keep) {  // `keep`: the key being handed out (with 64 pinned scenes it is the only unpinned one)
  let spare = map.size - SCENE_CACHE_ENTRIES;
  for (const k of Array.from(map.keys())) {
    if (spare <= 0) break;
    if ((pins && pins.get(k) > 0) || k === keep) continue;
    const v = map.get(k); map.delete(k); destroy(v); spare--;
  }
}
const scenePins = {  // mixed into both contexts, for hosts that hold handles themselves (doRenderJob does not pin: see there): pinScene(scene) -> [key, handle or error value]; unpinScene(key)
  pinScene(scene) { const hit = this.getScene(scene), key = scene.key(); this.pins.set(key, (this.pins.get(key) || 0) + 1); return [key, hit]; },
  unpinScene(key) { const n = (this.pins.get(key) || 0) - 1; if (n > 0) this.pins.set(key, n); else { this.pins.delete(key); this.evict(); } },
};

class RenderJobContext {  // RenderJobContext + loadRenderJobContext (LoadRenderJobContext.tsx:162-287)
  constructor(device = 0, flags = RM.RENDER_STRICT) {
    this.ctx = addon.ctxCreate(device); this.flags = flags; this.scenes = new Map(); this.pins = new Map(); this.live = new Map(); this.purgatory = [];
  }
  evict(keep) { evictScenes(this.scenes, (s) => { if (!(s && s.infoLog)) addon.sceneDestroy(s); }, this.pins, keep); }
  getScene(scene) {  // programCache.getProgram: results AND errors are cached (ShaderCache.tsx:91-119); bounded, least recently used out first
    const key = scene.key();
    if (!this.scenes.has(key)) {
      try { const d = scene.desc(); this.scenes.set(key, addon.sceneCreate(this.ctx, d.desc, d.prims, d.surfaces)); }
      catch (e) { this.scenes.set(key, { type: "fragment", infoLog: String(e.message) }); }
      this.evict(key);
    } else { const hit = this.scenes.get(key); this.scenes.delete(key); this.scenes.set(key, hit); }
    return this.scenes.get(key);
  }
  fboCreate(w, h, frameid) {  // :186-223
    const key = `${w}x${h}#${frameid}`;
    if (this.live.has(key)) return this.live.get(key);
    const i = this.purgatory.findIndex((e) => e.w === w && e.h === h);
    let fb;
    if (i >= 0) { const e = this.purgatory.splice(i, 1)[0]; if (e.frameid !== frameid) addon.fbClear(e.fb); fb = e.fb; }
    else fb = addon.fbCreate(this.ctx, w, h, 0, h);
    const info = { fb, width: w, height: h, frameid, download: (plane = 0) => { const out = new Float32Array(w * h * 4); addon.fbDownload(this.ctx, fb, plane, out); return out; },
                   // the present pass (display.frag) on the GPU: RGBA8, row 0 = bottom
                   present: (samples) => { const out = new Uint8Array(w * h * 4); addon.present(this.ctx, fb, samples, out); return out; },
                   // canvas.toDataURL("image/png"), index.tsx:470-476
                   toDataURL: (samples) => "data:image/png;base64," + encodePng(info.present(samples), w, h).toString("base64") };
    this.live.set(key, info);
    return info;
  }
  fboDelete(w, h, frameid) {  // :227-248: parked in a <= 3 entry purgatory
    const key = `${w}x${h}#${frameid}`, info = this.live.get(key);
    if (!info) return;
    this.live.delete(key);
    this.purgatory.push({ w, h, frameid, fb: info.fb });
    while (this.purgatory.length > 3) addon.fbDestroy(this.purgatory.shift().fb);
  }
  close() { for (const e of this.purgatory) addon.fbDestroy(e.fb); for (const v of this.live.values()) addon.fbDestroy(v.fb);
            for (const s of this.scenes.values()) if (!(s && s.infoLog)) addon.sceneDestroy(s); addon.ctxDestroy(this.ctx); }
}

// A frame sharded over several GPUs by THIS process -- the reference's host is one thread with one render loop
// (index.tsx:120), and so is a Node host: one native context per GPU, each holding one part of the frame's 8-row stripes
// (rm_fb_create_striped: pixel coordinates stay global, so the assembled frame has the single-GPU bits).  doRenderJob
// hands every batch of samples to each context in turn -- the launches are asynchronous, so the GPUs render
// concurrently -- and `present(samples)` of the framebuffer set is rm_present_sharded: every GPU tone-maps (or, with
// depth of field, packs) its rows, peer copies over xGMI bring them to the first GPU, which puts them in image order and
// runs the blur.  Same interface as RenderJobContext, so the reference's `present` callback does not change.
// (The tile loop of the reference, RenderJobExecutor.tsx:148-182, is the precedent for cutting a job's frame.)
const STRIPE_ROWS = 8;
class ShardedRenderJobContext {
  constructor(devices = [0], flags = RM.RENDER_STRICT, samplesInFlight = 3) {
    if (!Array.isArray(devices) || devices.length < 1) throw new TypeError("ShardedRenderJobContext(devices: number[], flags?)");
    this.devices = devices.slice(); this.flags = flags;
    this.ctxs = devices.map((d) => addon.ctxCreate(d));
    for (const c of this.ctxs) addon.setSamplesInFlight(c, samplesInFlight);
    this.ctx = this.ctxs[0];
    this.scenes = new Map(); this.pins = new Map(); this.live = new Map(); this.purgatory = [];
  }
  evict(keep) { evictScenes(this.scenes, (s) => { if (s.handles) for (const h of s.handles) addon.sceneDestroy(h); }, this.pins, keep); }
  getScene(scene) {  // one handle per context; a failure (on any of them) is cached like a failed compile
    const key = scene.key();
    if (!this.scenes.has(key)) {
      const made = [];
      try { const d = scene.desc(); for (const c of this.ctxs) made.push(addon.sceneCreate(c, d.desc, d.prims, d.surfaces)); this.scenes.set(key, { handles: made }); }
      catch (e) { for (const h of made) addon.sceneDestroy(h); this.scenes.set(key, { type: "fragment", infoLog: String(e.message) }); }
      this.evict(key);
    } else { const hit = this.scenes.get(key); this.scenes.delete(key); this.scenes.set(key, hit); }
    return this.scenes.get(key);
  }
  fboCreate(w, h, frameid) {
    const key = `${w}x${h}#${frameid}`;
    if (this.live.has(key)) return this.live.get(key);
    const i = this.purgatory.findIndex((e) => e.w === w && e.h === h);
    let fbs;
    if (i >= 0) { const e = this.purgatory.splice(i, 1)[0]; if (e.frameid !== frameid) for (const fb of e.fbs) addon.fbClear(fb); fbs = e.fbs; }
    else {
      fbs = [];
      try { this.ctxs.forEach((c, p) => fbs.push(addon.fbCreateStriped(c, w, h, STRIPE_ROWS, this.ctxs.length, p))); }
      catch (e) { for (const fb of fbs) addon.fbDestroy(fb); throw e; }
    }
    const info = { fbs, width: w, height: h, frameid, sharded: true, dof: false,
                   rows: () => fbs.map((fb) => addon.fbRows(fb)),
                   // the canvas of the assembled frame (display.frag, with its blur when the job has depth of field): RGBA8, row 0 = bottom
                   present: (samples, dof = info.dof) => { const out = new Uint8Array(w * h * 4); addon.presentSharded(this.ctxs, fbs, samples, !!dof, out); return out; },
                   // the same in two halves (rm_present_sharded_start / _finish): startPresent snapshots and sends and returns at once, so a
                   // `present` callback that calls it lets doRenderJob hand out the next samples while the frame travels; finishPresent
                   // (at the next callback, or whenever the canvas is wanted) returns the canvas of THAT present.  One at a time.
                   startPresent: (samples, dof = info.dof) => { addon.presentShardedStart(this.ctxs, fbs, samples, !!dof); info.pendingPresent = true; },
                   finishPresent: () => {
                     // the pending present lives on the contexts, not on this framebuffer set: only the set that started it may finish it
                     // (another set's canvas may be smaller; the library checks the buffer's size as well)
                     if (!info.pendingPresent) throw new Error("finishPresent: this framebuffer set has no present pending (startPresent was called on another set, or not at all)");
                     const out = new Uint8Array(w * h * 4); addon.presentShardedFinish(this.ctxs, out); info.pendingPresent = false; return out; },
                   pendingPresent: false,
                   toDataURL: (samples) => "data:image/png;base64," + encodePng(info.present(samples), w, h).toString("base64") };
    this.live.set(key, info);
    return info;
  }
  fboDelete(w, h, frameid) {
    const key = `${w}x${h}#${frameid}`, info = this.live.get(key);
    if (!info) return;
    this.live.delete(key);
    this.purgatory.push({ w, h, frameid, fbs: info.fbs });
    while (this.purgatory.length > 3) for (const fb of this.purgatory.shift().fbs) addon.fbDestroy(fb);
  }
  close() {
    for (const e of this.purgatory) for (const fb of e.fbs) addon.fbDestroy(fb);
    for (const v of this.live.values()) for (const fb of v.fbs) addon.fbDestroy(fb);
    for (const s of this.scenes.values()) if (s.handles) for (const h of s.handles) addon.sceneDestroy(h);
    for (const c of this.ctxs) addon.ctxDestroy(c);
  }
}

Object.assign(RenderJobContext.prototype, scenePins);
Object.assign(ShardedRenderJobContext.prototype, scenePins);

async function doRenderJob(schema, context) {  // RenderJobExecutor.tsx:77
  const fail = (why) => function* () { return { success: false, why }; };
  const r = schema.render;
  let framebuffers;
  try { framebuffers = context.fboCreate(r.width, r.height, r.frameid); }
  catch (e) { return fail({ type: "general", infoLog: "Failed to load framebuffers. " + e.message }); }
  if (!schema.sdfScene) return fail({ type: "fragment", infoLog: "no sdfScene: the HIP back end takes a composed scene, not GLSL text" });
  const compiled = context.getScene(schema.sdfScene);
  if (compiled && compiled.infoLog !== undefined) return fail(compiled);
  let samples = 0;
  // a sharded context: the same calls on every GPU's context, each with its part of the stripes (the tile is clipped to them)
  const ctxs = framebuffers.sharded ? context.ctxs : [context.ctx];
  const fbs = framebuffers.sharded ? framebuffers.fbs : [framebuffers.fb];
  if (framebuffers.sharded) framebuffers.dof = schema.dof.amount !== 0;  // what a present gathers (rm_present_sharded)
  // (a sharded set's present is ordered on the contexts' streams behind the samples: no host-side wait before it, so that a
  // callback using startPresent overlaps the travelling frame with the next samples; the single context waits as it always did)
  const syncAll = () => { if (!framebuffers.sharded) for (const c of ctxs) addon.sync(c); };
  const syncEnd = () => { for (const c of ctxs) addon.sync(c); };
  return function* (present) {
    // The job yields between samples, and other jobs may bring in more scenes than the cache holds meanwhile.  No pin is held across a
    // yield (round 6: a JS generator that is started and then dropped without .return() never runs its `finally`, so a pin taken for
    // its lifetime would keep the scene un-evictable for the life of the context): the scene is looked up again -- and, had it been
    // evicted, made again -- in front of every batch of native calls, with nothing in between that could suspend the job.
    for (let yp = 0; yp < r.subdivisions; yp++) for (let xp = 0; xp < r.subdivisions; xp++) {
      const tile = tileRect(schema, xp, yp);
      for (let left = r.samplesPerPixel; left > 0;) {
        if (samples % r.sampleYieldInterval === 0) { syncAll(); present(schema, context, framebuffers, samples); yield; }
        // the samples up to the next yield differ in randNoise only (:219-222): one native call for all of them
        const k = Math.min(left, r.sampleYieldInterval - samples % r.sampleYieldInterval);
        const noise = new Float32Array(2 * k);
        for (let i = 0; i < k; i++) { noise[2 * i] = halton2.next().value; noise[2 * i + 1] = halton3.next().value; }
        const u = uniformsFromSchema(schema, [noise[0], noise[1]]);
        const scene = context.getScene(schema.sdfScene);
        if (scene && scene.infoLog !== undefined) { context.fboDelete(r.width, r.height, r.frameid); return { success: false, why: scene }; }
        const scenes = framebuffers.sharded ? scene.handles : [scene];
        for (let g = 0; g < ctxs.length; g++) {
          if (k === 1) addon.renderSample(ctxs[g], scenes[g], fbs[g], u, tile, context.flags);
          else addon.renderSamples(ctxs[g], scenes[g], fbs[g], u, noise, tile, context.flags);
        }
        samples += k; left -= k;
      }
    }
    context.fboDelete(r.width, r.height, r.frameid);
    syncEnd();  // the job's last word: an asynchronous failure becomes this call's exception (the caller's {success: false})
    present(schema, context, framebuffers, samples);
    return { success: true };
  };
}

// ---- PNG capture (index.tsx:470-476 canvas.toDataURL): RGBA8, filter 0, rows flipped to top-down ----
const zlib = require("zlib");
const CRC_TABLE = (() => { const t = new Uint32Array(256); for (let n = 0; n < 256; n++) { let c = n; for (let k = 0; k < 8; k++) c = c & 1 ? 0xedb88320 ^ (c >>> 1) : c >>> 1; t[n] = c >>> 0; } return t; })();
function crc32(buf) { let c = 0xffffffff; for (let i = 0; i < buf.length; i++) c = CRC_TABLE[(c ^ buf[i]) & 0xff] ^ (c >>> 8); return (c ^ 0xffffffff) >>> 0; }
function pngChunk(tag, data) {
  const out = Buffer.alloc(12 + data.length);
  out.writeUInt32BE(data.length, 0); out.write(tag, 4, "ascii"); data.copy(out, 8);
  out.writeUInt32BE(crc32(out.slice(4, 8 + data.length)), 8 + data.length);
  return out;
}
function encodePng(rgba, width, height, bottomUp = true) {
  if (rgba.length !== width * height * 4) throw new RangeError("encodePng: rgba must hold width * height * 4 bytes");
  const raw = Buffer.alloc(height * (1 + width * 4));
  for (let y = 0; y < height; y++) {
    const src = (bottomUp ? height - 1 - y : y) * width * 4;
    raw[y * (1 + width * 4)] = 0;
    Buffer.from(rgba.buffer, rgba.byteOffset + src, width * 4).copy(raw, y * (1 + width * 4) + 1);
  }
  const ihdr = Buffer.alloc(13);
  ihdr.writeUInt32BE(width, 0); ihdr.writeUInt32BE(height, 4); ihdr[8] = 8; ihdr[9] = 6;
  return Buffer.concat([Buffer.from([0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a]), pngChunk("IHDR", ihdr), pngChunk("IDAT", zlib.deflateSync(raw)), pngChunk("IEND", Buffer.alloc(0))]);
}

module.exports = { RM, addon, encodePng, Scene, CsgScene, Mandelbulb, singleSphere, DEFAULT_MATERIAL, halton, resetHalton, uniformsFromSchema, packUniforms,
                   tileRect, RenderJobContext, ShardedRenderJobContext, doRenderJob, U_OFFSET };
