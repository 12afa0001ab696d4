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
