#!/usr/bin/env python3
"""tests/golden/halton_reference.json: the first 2048 values of the reference's own halton(b) generator
(client/src/util/Halton.tsx:1-19, read at run time, TypeScript annotations stripped, run under node) for the bases the
path uses (2 and 3: randNoise, RenderJobExecutor.tsx:70-71,219-222) and two more, as hexadecimal doubles.
Build-container only.    python oracle/ts/gen_halton_golden.py"""
import json
import re
import subprocess
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
src = Path("/root/reference/client/src/util/Halton.tsx").read_text()
js = re.sub(r"export function\* halton\(b: number\)[^{]*\{", "function* halton(b) {", src)
js += """
const out = {};
for (const b of [2, 3, 5, 7]) { const g = halton(b); const v = []; for (let i = 0; i < 2048; i++) { const buf = Buffer.alloc(8); buf.writeDoubleBE(g.next().value); v.push(buf.toString('hex')); } out[b] = v; }
console.log(JSON.stringify(out));
"""
with tempfile.TemporaryDirectory() as td:
    p = Path(td) / "h.js"
    p.write_text(js)
    r = subprocess.run(["node", str(p)], capture_output=True, text=True, check=True)
data = json.loads(r.stdout)
out = {"_about": "first 2048 values of the reference's halton(b) (Halton.tsx:1-19) under node, big-endian IEEE doubles in hex; oracle/ts/gen_halton_golden.py", "values": data}
(ROOT / "tests" / "golden" / "halton_reference.json").write_text(json.dumps(out))
print({k: len(v) for k, v in data.items()}, (ROOT / "tests" / "golden" / "halton_reference.json").stat().st_size, "B")
