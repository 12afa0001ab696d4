#!/usr/bin/env python3
"""tests/golden/fbo_reference.json: the reference's framebuffer cache, operation by operation.

Build-container only.  Reads client/src/renderer/LoadRenderJobContext.tsx at run time, takes the function that builds
context.fbo (loadRenderJobFramebufferGetter, :160-250: the (width, height, frameid) map, the <= 3 entry purgatory, the
clear-on-new-frameid, the eviction), strips its TypeScript annotations mechanically and runs it under node with a
recording WebGL mock and a stub for the texture allocation, over 600 random create / delete operations on a small key
space.  The fixture holds the operations and what each did: which framebuffer set came back (by serial number), whether
`prev` was cleared, which sets were destroyed.  tests/test_host_cpu.py replays it through job.RenderJobContext, and
tests/test_js_host.py through js/index.js.    python oracle/ts/gen_fbo_golden.py"""
import json
import random
import re
import subprocess
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
src = Path("/root/reference/client/src/renderer/LoadRenderJobContext.tsx").read_text()
start = src.index("export function loadRenderJobFramebufferGetter(")
end = src.index("export type RenderJobParameters")
fn = src[start:end]
fn = fn.replace("export function loadRenderJobFramebufferGetter(gl: WebGL2RenderingContext) {", "function loadRenderJobFramebufferGetter(gl) {")
fn, n1 = re.subn(r"createGenericMap<[\s\S]*?RenderJobFramebufferInfo\s*>\(", "createGenericMap(", fn)
fn, n2 = re.subn(r"const framebufferPurgatory: RenderJobFramebufferInfo\[\] = \[\];", "const framebufferPurgatory = [];", fn)
fn = re.sub(r":\s*number", "", fn)
assert n1 == 1 and n2 == 1 and not re.search(r":\s*(number|RenderJob\w*|WebGL\w*)", fn), "type annotations left"

rnd = random.Random(3)
keys = [(w, h, f) for w in (64, 100) for h in (64, 48) for f in (0, 1, 2)]
ops = []
for _ in range(600):
    k = rnd.choice(keys)
    ops.append(["create" if rnd.random() < 0.55 else "delete", *k])

js = "\n".join([
    "const log = console.log; console.log = () => {};",
    # util/Memoize.tsx createGenericMap: a map keyed by hash + equality (restated: the original uses ?. which node 12 lacks)
    "function createGenericMap(hash, eq) { const m = new Map(); return {",
    "  get(k) { const e = (m.get(hash(k)) || []).find((x) => eq(x[0], k)); return e ? e[1] : undefined; },",
    "  set(k, v) { let a = m.get(hash(k)); if (!a) { a = []; m.set(hash(k), a); } const i = a.findIndex((x) => eq(x[0], k)); if (i != -1) a.splice(i, 1); a.push([k, v]); },",
    "  delete(k) { const a = m.get(hash(k)); if (!a) return; const i = a.findIndex((x) => eq(x[0], k)); if (i != -1) a.splice(i, 1); } }; }",
    "let serial = 0; let events = [];",
    "function loadRenderJobFramebuffers(gl, width, height, frameid) { const uid = ++serial; events.push({created: uid});",
    "  return { prev: {uid, which: 'prev'}, curr: {uid, which: 'curr'}, prevTex: {color: {uid}}, currTex: {color: {uid}}, width, height, frameid, uid }; }",
    "let bound = null;",
    "const gl = new Proxy({}, { get(t, name) { if (typeof name !== 'string') return undefined; if (/^[A-Z0-9_]+$/.test(name)) return name;",
    "  if (name === 'bindFramebuffer') return (target, fb) => { bound = fb; };",
    "  if (name === 'clear') return () => events.push({cleared: bound.uid, which: bound.which});",
    "  if (name === 'deleteFramebuffer') return (fb) => { if (fb.which === 'prev') events.push({destroyed: fb.uid}); };",
    "  return () => undefined; } });",
    fn,
    "const fbo = loadRenderJobFramebufferGetter(gl);",
    "const ops = " + json.dumps(ops) + "; const out = [];",
    "for (const [op, w, h, f] of ops) { events = []; let r = null; if (op === 'create') { const info = fbo.create(w, h, f); r = info ? info.uid : null; } else fbo.delete(w, h, f); out.push({result: r, events}); }",
    "log(JSON.stringify(out));",
])
with tempfile.TemporaryDirectory() as td:
    p = Path(td) / "run.js"
    p.write_text(js)
    r = subprocess.run(["node", str(p)], capture_output=True, text=True)
if r.returncode != 0:
    raise SystemExit("node failed:\n" + r.stderr[-3000:])
res = json.loads(r.stdout)
dest = ROOT / "tests" / "golden" / "fbo_reference.json"
dest.write_text(json.dumps({"_about": "what the reference's framebuffer cache (LoadRenderJobContext.tsx:160-250, run under node by oracle/ts/gen_fbo_golden.py) "
                                      "does for each of these operations: the serial number of the set returned, clears of `prev`, sets destroyed",
                            "ops": ops, "results": res}))
from collections import Counter
print(len(ops), "operations;", Counter(k for x in res for e in x["events"] for k in e if k != "which"), dest.stat().st_size, "B")
