// node render_cli.js <out.f32> -- renders a small job through doRenderJob and writes the colour plane (tests/test_js_host.py)
"use strict";
const fs = require("fs");
const rm = require("./index.js");
(async () => {
  const out = process.argv[2];
  const mode = process.argv[3] || "render";
  const schema = {
    reflectionIterationCounts: [48, 24], normalDelta: 1e-5, sdfShaderSource: "", customShaderParameters: {}, fogDensity: 0, time: 0, timeDelta: 0,
    sdfScene: new rm.CsgScene().box([0, 0, 0], [1.0, 0.6, 0.8]).subtract().sphere([0.4, 0.3, -0.6], 0.7).smoothUnion(0.3).sphere([-1.2, 0.2, 0.0], 0.5),
    dof: { amount: 0, distance: 1.5, showFocusedArea: false },
    camera: { position: [0.3, 0.2, -4.0], motion: [0, 0, 0], rotation: [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1], mode: { type: "perspective", fov: 1.5 } },
    render: { samplesPerPixel: 3, exposure: 0.5, subdivisions: 2, width: 64, height: 32, frameid: 1, blendWithPreviousFrameFactor: 0.9,
              sampleYieldInterval: 2, blendMode: "additive", renderMode: "full" },
    lights: [{ type: "point", position: [2, 3, -4], color: [255 * 3 / 256, 255 * 3 / 256, 255 * 3 / 256], size: 0 }],
  };
  if (mode === "png") {  // no GPU needed: the PNG writer on a synthetic image
    const w = 5, h = 3, px = new Uint8Array(w * h * 4);
    for (let i = 0; i < px.length; i++) px[i] = (i * 37 + 11) & 255;
    fs.writeFileSync(out, rm.encodePng(px, w, h));
    return;
  }
  if (mode === "domain") {  // no GPU needed: a scene with the domain operators, as the table and as GLSL
    const sc = new rm.CsgScene().repeat([3, 3, 3]).fold(0.8, [0.5, 0.2, 0.3], [0.3, -0.2, 0.1]).box([0, 0, 0], [0.4, 0.3, 0.2]).smoothUnion(0.15).sphere([0.3, 0.1, 0], 0.25);
    const d = sc.desc();
    process.stdout.write(JSON.stringify({ prims: Buffer.from(d.prims).toString("hex"), glsl: sc.glsl() }));
    return;
  }
  if (mode === "shapes") {  // no GPU needed: ABI 8's shapes and smooth operators (tests/golden_cases.py csg_shapes), as the table and the surface rows
    const sc = new rm.CsgScene().box([0.0, 0.0, 0.0], [1.25, 0.375, 1.0]).smoothSubtract(0.125).torus([0.0, 0.375, 0.0], 0.625, 0.1875)
      .union().cylinder([0.875, 0.75, -0.25], 0.25, 0.5)
      .smoothUnion(0.1875).torus([-0.75, 0.75, 0.25], 0.375, 0.125, { diffuse: [0.875, 0.5, 0.125], specular: [0.5, 0.5, 0.5], roughness: 0.375 })
      .smoothIntersect(0.25).plane([0.0, 1.0, 0.0], [0.0, 1.0, 0.0])
      .subtract().cylinder([-0.25, 0.0, 0.625], 0.1875, 1.0);
    const d = sc.desc();
    let refused = "";
    try { sc.glsl(); } catch (e) { refused = e.message; }
    process.stdout.write(JSON.stringify({ prims: Buffer.from(d.prims).toString("hex"), surfaces: Buffer.from(d.surfaces).toString("hex"), refused }));
    return;
  }
  if (mode === "kinds") {  // no GPU needed: a table whose rows evaluate a scene kind (RM_PRIM_KIND), as the description and the table
    const sc = new rm.CsgScene().shape(new rm.Mandelbulb(8, 5, 2)).intersect().box([0.0, 0.0, 0.25], [1.25, 1.25, 0.75])
      .subtract().sphere([0.5, 0.375, -0.5], 0.375, { diffuse: [0.875, 0.25, 0.125], specular: [0.5, 0.5, 0.5], roughness: 0.25 });
    const d = sc.desc();
    let refused = "";
    try { sc.glsl(); } catch (e) { refused = e.message; }
    let mixed = "";
    try { new rm.CsgScene().shape(new rm.Mandelbulb(8, 5, 2)).shape(new rm.Mandelbulb(8, 6, 2)); } catch (e) { mixed = e.name; }
    process.stdout.write(JSON.stringify({ desc: Buffer.from(d.desc).toString("hex"), prims: Buffer.from(d.prims).toString("hex"), surfaces: Buffer.from(d.surfaces).toString("hex"), refused, mixed }));
    return;
  }
  if (mode === "surfaces") {  // no GPU needed: a scene whose shapes name surfaces, as the table, the surface rows and GLSL
    const sc = new rm.CsgScene().box([0, 0, 0], [1.0, 0.5, 0.75]).smoothUnion(0.25)
      .sphere([-1.25, 0.25, 0.0], 0.5, { diffuse: [0.875, 0.125, 0.125], specular: [0.25, 0.25, 0.25], roughness: 0.5 })
      .union().sphere([1.25, 0.125, -0.25], 0.625, { diffuse: [0.125, 0.25, 0.875], specular: [0.75, 0.75, 0.75], roughness: 0.0625, ior: 1.5 })
      .sphere([0.0, 1.0, 0.0], 0.5, { diffuse: [0.5, 0.75, 0.5], specular: [0.5, 0.5, 0.5], subsurface: 4.0, subsurface_color: [0.875, 0.5, 0.25] })
      .subtract().box([0.0, 1.0, -0.5], [0.25, 0.25, 0.25], { diffuse: [0.75, 0.75, 0.125], specular: [0.125, 0.125, 0.125], roughness: 0.75 });
    const d = sc.desc();
    process.stdout.write(JSON.stringify({ desc: Buffer.from(d.desc).toString("hex"), prims: Buffer.from(d.prims).toString("hex"), surfaces: Buffer.from(d.surfaces).toString("hex"), glsl: sc.glsl() }));
    return;
  }
  if (mode === "fbo") {  // no GPU needed: RenderJobContext.fboCreate / fboDelete over the operations of tests/golden/fbo_reference.json
    const fx = JSON.parse(fs.readFileSync(process.argv[4]).toString());
    const a = rm.addon;
    let serial = 0, log = [];
    a.ctxCreate = () => ({}); a.fbCreate = () => { const uid = ++serial; log.push(["created", uid]); return { uid }; };
    a.fbClear = (fb) => log.push(["cleared", fb.uid]); a.fbDestroy = (fb) => log.push(["destroyed", fb.uid]);
    const ctx = new rm.RenderJobContext(0, rm.RM.RENDER_STRICT);
    const got = [];
    for (const [op, w, h, f] of fx.ops) {
      log = [];
      let uid = null;
      if (op === "create") uid = ctx.fboCreate(w, h, f).fb.uid; else ctx.fboDelete(w, h, f);
      got.push([uid, log]);
    }
    fs.writeFileSync(out, JSON.stringify(got));
    return;
  }
  if (mode === "replay") {  // no GPU needed: doRenderJob over the schemas of tests/golden/host_reference.json.gz with the addon's calls recorded
    const fx = JSON.parse(require("zlib").gunzipSync(fs.readFileSync(process.argv[4])).toString());
    const a = rm.addon;
    let events = null;
    a.ctxCreate = () => ({}); a.ctxDestroy = () => {}; a.sync = () => {}; a.sceneCreate = () => ({}); a.sceneDestroy = () => {};
    a.fbCreate = () => ({}); a.fbClear = () => {}; a.fbDestroy = () => {};
    a.renderSample = (c, sc, fb, u, tile, flags) => events.push({ draw: { tile: Array.from(tile || []), uniforms: Buffer.from(u).toString("hex") } });
    a.renderSamples = (c, sc, fb, u, noise, tile, flags) => {
      for (let i = 0; i < noise.length / 2; i++) {
        const v = Buffer.from(Buffer.from(u));  // a copy of the block, randNoise of sample i written over it
        v.writeFloatLE(noise[2 * i], rm.U_OFFSET.randNoise); v.writeFloatLE(noise[2 * i + 1], rm.U_OFFSET.randNoise + 4);
        events.push({ draw: { tile: Array.from(tile || []), uniforms: v.toString("hex") } });
      }
    };
    const ctx = new rm.RenderJobContext(0, rm.RM.RENDER_STRICT);
    const del = ctx.fboDelete.bind(ctx);
    ctx.fboDelete = (w, h, id) => { events.push({ fboDelete: [w, h, id] }); del(w, h, id); };
    rm.resetHalton();
    const all = [];
    const scene = rm.singleSphere();
    for (const schema0 of fx.schemas) {
      const schema = Object.assign({}, schema0, { sdfScene: scene, render: Object.assign({}, schema0.render, { referenceScissor: true }) });
      events = [];
      const gen = (await rm.doRenderJob(schema, ctx))((s, c, fb, n) => events.push({ present: n }));
      let r = gen.next();
      while (!r.done) { events.push({ yield: 1 }); r = gen.next(); }
      events.push({ done: r.value });
      all.push(events);
    }
    fs.writeFileSync(out, JSON.stringify(all));
    return;
  }
  if (mode === "sharded") {  // GPU: the same job on ONE context and on a ShardedRenderJobContext with N contexts (all on device 0 here), with and without depth of field
    const n = parseInt(process.argv[4] || "3", 10);
    const jobs = { plain: {}, dof: { dof: { amount: 0.03, distance: 3.5, showFocusedArea: false } } };
    const result = {};
    for (const name of Object.keys(jobs)) {
      const sch = Object.assign({}, schema, jobs[name], { render: Object.assign({}, schema.render, { width: 96, height: 52, frameid: name === "plain" ? 1 : 2 }) });
      const run = async (ctx) => {
        rm.resetHalton();
        const shown = [];
        const gen = (await rm.doRenderJob(sch, ctx))((s, c, fb, k) => { if (k > 0) shown.push([k, Buffer.from(fb.present(k)).toString("base64")]); });
        let it = gen.next();
        while (!it.done) it = gen.next();
        return { res: it.value, shown };
      };
      const one = new rm.RenderJobContext(0, rm.RM.RENDER_STRICT);
      const a = await run(one);
      one.close();
      const many = new rm.ShardedRenderJobContext(new Array(n).fill(0), rm.RM.RENDER_STRICT);
      const b = await run(many);
      const rows = many.fboCreate(96, 52, sch.render.frameid).rows();
      many.close();
      // the present in two halves: the callback starts frame k and collects frame k - 1 (which travelled while the samples between
      // the two callbacks rendered); the last one is collected after the job
      const lapped = new rm.ShardedRenderJobContext(new Array(n).fill(0), rm.RM.RENDER_STRICT);
      rm.resetHalton();
      const shown = [];
      let last = 0, set = null;
      const gen = (await rm.doRenderJob(sch, lapped))((s, c, fb, k) => {
        if (k === 0) return;
        if (fb.pendingPresent) shown.push([last, Buffer.from(fb.finishPresent()).toString("base64")]);
        fb.startPresent(k); last = k; set = fb;
      });
      let it = gen.next();
      while (!it.done) it = gen.next();
      shown.push([last, Buffer.from(set.finishPresent()).toString("base64")]);
      lapped.close();
      result[name] = { one: a, many: b, lapped: { res: it.value, shown }, rows };
    }
    fs.writeFileSync(out, JSON.stringify(result));
    return;
  }
  if (mode === "sharded-replay") {  // no GPU needed: the calls a sharded job makes, recorded
    const a = rm.addon;
    const events = [];
    let serial = 0;
    a.ctxCreate = (d) => ({ ctx: ++serial, device: d }); a.ctxDestroy = () => {}; a.sync = (c) => events.push(["sync", c.ctx]); a.setSamplesInFlight = () => {};
    a.sceneCreate = (c) => ({ scene: c.ctx }); a.sceneDestroy = () => {};
    a.fbCreateStriped = (c, w, h, stripe, parts, part) => { events.push(["fbCreateStriped", c.ctx, w, h, stripe, parts, part]); return { fb: c.ctx }; };
    a.fbClear = () => {}; a.fbDestroy = () => {}; a.fbRows = () => 0;
    a.renderSample = (c, sc, fb, u, tile) => events.push(["render", c.ctx, sc.scene, fb.fb, 1, Array.from(tile)]);
    a.renderSamples = (c, sc, fb, u, noise, tile) => events.push(["render", c.ctx, sc.scene, fb.fb, noise.length / 2, Array.from(tile)]);
    a.presentSharded = (ctxs, fbs, samples, dof, o) => events.push(["presentSharded", ctxs.map((c) => c.ctx), fbs.map((f) => f.fb), samples, dof]);
    const ctx = new rm.ShardedRenderJobContext([0, 1, 2], rm.RM.RENDER_FAST);
    rm.resetHalton();
    const sch = Object.assign({}, schema, { dof: { amount: 0.01, distance: 1.5, showFocusedArea: false } });
    const gen = (await rm.doRenderJob(sch, ctx))((s, c, fb, k) => { if (k > 0) fb.present(k); });
    let it = gen.next();
    while (!it.done) { events.push(["yield"]); it = gen.next(); }
    events.push(["done", it.value]);
    fs.writeFileSync(out, JSON.stringify(events));
    return;
  }
  if (mode === "layout") {  // no GPU needed: the uniform block bytes and the scene description
    const u = rm.uniformsFromSchema(schema, [0.5, 1 / 3]);
    const d = schema.sdfScene.desc();
    process.stdout.write(JSON.stringify({ uniforms: Buffer.from(u).toString("hex"), desc: Buffer.from(d.desc).toString("hex"),
      prims: Buffer.from(d.prims).toString("hex"), glsl: schema.sdfScene.glsl(), halton3: (() => { const g = rm.halton(3); return [g.next().value, g.next().value, g.next().value]; })() }));
    return;
  }
  const ctx = new rm.RenderJobContext(0, rm.RM.RENDER_STRICT);
  rm.resetHalton();
  const seen = [];
  const gen = (await rm.doRenderJob(schema, ctx))((s, c, fb, n) => seen.push(n));
  let res;
  for (;;) { const it = gen.next(); if (it.done) { res = it.value; break; } }
  const fb = ctx.fboCreate(64, 32, 1);
  fs.writeFileSync(out, Buffer.from(fb.download(0).buffer));
  fs.writeFileSync(out + ".png", rm.encodePng(fb.present(3), 64, 32));  // what canvas.toDataURL would hold
  const bad = await rm.doRenderJob(Object.assign({}, schema, { sdfScene: new rm.CsgScene().smoothUnion(-1).sphere([0, 0, 0], 1).sphere([1, 0, 0], 1) }), ctx);
  const badRes = bad(() => {}).next().value;
  // the addon's own argument checks: a short output buffer and a handle of the wrong kind are JS errors, not memory errors
  const addon = require("./rm_napi.node");
  const guards = {};
  try { addon.fbDownload(ctx.ctx, fb.fb, 0, new Float32Array(16)); guards.shortDownload = "no error"; } catch (e) { guards.shortDownload = e.name; }
  try { addon.present(ctx.ctx, fb.fb, 3, new Uint8Array(16)); guards.shortPresent = "no error"; } catch (e) { guards.shortPresent = e.name; }
  try { addon.fbDownload(ctx.ctx, ctx.ctx, 0, new Float32Array(64 * 32 * 4)); guards.wrongKind = "no error"; } catch (e) { guards.wrongKind = e.name; }
  process.stdout.write(JSON.stringify({ res, seen, badRes, guards }));
  ctx.close();
})().catch((e) => { console.error(e); process.exit(1); });
