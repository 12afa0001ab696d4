#!/usr/bin/env python3
"""tests/golden/host_reference.json.gz: what the REFERENCE's own render-job loop does, call by call.

Build-container only (needs /root/reference and node).  Reads client/src/renderer/RenderJobExecutor.tsx at run time,
cuts out the body of the generator doRenderJob returns (:147-339: the subdivision / sample loops, the scissor call, every
uniform assignment, the draw and blit calls, the yield cadence, fbo.delete, the final present), replaces its one
`a ?? b` (node 12 has no ??) and runs it under node against a recording WebGL mock -- with the reference's own
Uniforms.tsx helpers and Halton generator, types stripped -- for 40 random RenderJobSchemas in a row (the module-level
Halton pair continues across jobs, as in the page).  The recorded events (present(n) / draw{scissor, uniforms} /
fbo.delete) are the fixture; tests/test_host_cpu.py replays the same schemas through raymarching_engine_amd.job and
compares event by event.  Numbers only: no reference text is stored.    python oracle/ts/gen_host_golden.py"""
import json
import random
import re
import subprocess
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
REF = Path("/root/reference/client/src")


def random_schemas(n=40, seed=5):
    rnd = random.Random(seed)
    f = lambda a, b: round(rnd.uniform(a, b), 6)
    out = []
    for i in range(n):
        mode = rnd.choice(["perspective", "perspective", "orthographic", "panoramic"])
        cam_mode = {"type": "perspective", "fov": f(0.5, 2.0)} if mode == "perspective" else {"type": "orthographic", "size": f(1, 6)} if mode == "orthographic" else {"type": "panoramic", "angleX": 0, "angleY": 0}
        lights = []
        for _ in range(rnd.choice([0, 0, 1, 1, 2, 3, 10])):
            if rnd.random() < 0.3:
                lights.append({"type": "sun", "direction": [f(-1, 1), f(-1, 1), f(-1, 1)], "color": [f(0, 3), f(0, 3), f(0, 3)]})
            else:
                lights.append({"type": "point", "position": [f(-5, 5), f(-5, 5), f(-5, 5)], "color": [f(0, 3), f(0, 3), f(0, 3)], "size": rnd.choice([0, 0, f(0, 1)])})
        out.append({
            "reflectionIterationCounts": [rnd.choice([8, 32, 64, 128, 256, 12.5]) for _ in range(rnd.choice([1, 1, 2, 3, 5, 10]))],
            "sdfShaderSource": "", "customShaderParameters": {},
            "fogDensity": rnd.choice([0, 0, f(0, 0.5)]),
            "dof": {"amount": rnd.choice([0, 0, f(0, 0.2)]), "distance": f(0.5, 5), "showFocusedArea": rnd.random() < 0.2},
            "camera": {"position": [f(-3, 3), f(-3, 3), f(-5, 0)], "rotation": [f(-1, 1) for _ in range(16)], "mode": cam_mode},
            "render": {"samplesPerPixel": rnd.choice([1, 1, 2, 3, 5, 8]), "exposure": f(0.1, 2), "subdivisions": rnd.choice([1, 1, 2, 3, 4, 5]),
                       "width": rnd.choice([64, 100, 240, 257, 1920, 333]), "height": rnd.choice([64, 75, 135, 99, 1080, 217]), "frameid": rnd.randint(0, 3),
                       "blendWithPreviousFrameFactor": f(0, 1), "sampleYieldInterval": rnd.choice([1, 1, 2, 3, 4, 7]),
                       "blendMode": rnd.choice(["additive", "additive", "mix"]), "renderMode": rnd.choice(["full", "full", "preview"])},
            "lights": lights,
        })
    return out


def main():
    ex = (REF / "renderer/RenderJobExecutor.tsx").read_text()
    start = ex.index("    for (\n      let yPartitions = 0;")
    end = ex.index("    return { success: true };", start)
    body = ex[start:end]
    body, n_sub = re.subn(r"schema\.reflectionIterationCounts\[1\] \?\?\s*schema\.reflectionIterationCounts\[0\]",
                          "(schema.reflectionIterationCounts[1] !== undefined && schema.reflectionIterationCounts[1] !== null ? schema.reflectionIterationCounts[1] : schema.reflectionIterationCounts[0])", body)
    assert n_sub == 1 and "??" not in body
    uni = (REF / "renderer/Uniforms.tsx").read_text()
    uni = re.sub(r"export type[\s\S]*?;\n\n", "", uni)                        # the two type aliases
    uni = uni.replace("export namespace u {", "const u = (function () { const u = {};").replace("export const ", "u.")
    uni = re.sub(r"u\.(\w+) = \(([^)]*)\)\s*:\s*UniformData\s*=>", lambda m: "u.%s = (%s) =>" % (m.group(1), re.sub(r":\s*number", "", m.group(2))), uni)
    uni = re.sub(r"\n\}\n\nexport function setUniforms\([\s\S]*?\) \{", "\nreturn u; })();\n\nfunction setUniforms(gl, program, uniforms) {", uni)
    hal = re.sub(r"export function\* halton\(b: number\)[^{]*\{", "function* halton(b) {", (REF / "util/Halton.tsx").read_text())
    schemas = random_schemas()
    js = "\n".join([
        uni, hal,
        "const renderJobHalton2 = halton(2); const renderJobHalton3 = halton(3);",
        "function makeGl(events) { const state = {uniforms: {}, scissor: null, program: null};",
        "  const handler = { get(t, name) { if (name in t) return t[name]; if (typeof name !== 'string') return undefined;",
        "    if (/^[A-Z0-9_]+$/.test(name)) return name;",
        "    if (name === 'getUniformLocation') return (p, n) => n; if (name === 'getAttribLocation') return () => 0;",
        "    if (name === 'useProgram') return (p) => { state.program = p; };",
        "    if (name === 'scissor') return (a, b, c, d) => { state.scissor = [a, b, c, d]; };",
        "    if (/^uniform/.test(name)) return (...args) => { if (state.program === 'raymarcher') { const data = args[args.length - 1]; state.uniforms[args[0]] = Array.from(typeof data === 'number' ? [data] : data); } };",
        "    if (name === 'drawArrays') return () => { if (state.program === 'raymarcher') events.push({draw: {scissor: state.scissor, uniforms: JSON.parse(JSON.stringify(state.uniforms, (k, v) => (typeof v === 'number' && !isFinite(v)) ? String(v) : v))}}); else events.push({blit: 1}); };",
        "    return () => undefined; } };",
        "  return new Proxy({}, handler); }",
        "function* job(schema, context, gl, raymarcherProgram, framebuffers, present) { let samplesRenderedSoFar = 0;",
        body,
        "  return { success: true }; }",
        "const schemas = " + json.dumps(schemas) + ";",
        "const all = [];",
        "for (const schema of schemas) { const events = []; const gl = makeGl(events);",
        "  const context = { program: { blit: 'blit' }, fullscreenQuadBuffer: 'quad', fbo: { delete: (w, h, id) => events.push({fboDelete: [w, h, id]}) } };",
        "  const fbs = { prev: 'p', curr: 'c', prevTex: {}, currTex: {} };",
        "  const g = job(schema, context, gl, 'raymarcher', fbs, (gl_, s, c, f, n) => events.push({present: n}));",
        "  let r = g.next(); while (!r.done) { events.push({yield: 1}); r = g.next(); } events.push({done: r.value}); all.push(events); }",
        "console.log(JSON.stringify(all));",
    ])
    with tempfile.TemporaryDirectory() as td:
        p = Path(td) / "run.js"
        p.write_text(js)
        r = subprocess.run(["node", str(p)], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("node failed:\n" + r.stderr[-3000:])
    events = json.loads(r.stdout)
    out = {"_about": "events of the reference's render-job generator (RenderJobExecutor.tsx:147-339) under node against a recording WebGL mock, "
                     "for the random schemas below, in order (the Halton pair continues across jobs); oracle/ts/gen_host_golden.py",
           "schemas": schemas, "events": events}
    import gzip

    dest = ROOT / "tests" / "golden" / "host_reference.json.gz"
    with gzip.GzipFile(dest, "wb", mtime=0) as fh:
        fh.write(json.dumps(out).encode())
    print(len(schemas), "jobs,", sum(len(e) for e in events), "events,", dest.stat().st_size, "B")
    print(json.dumps(events[0][:3])[:600])


if __name__ == "__main__":
    main()
