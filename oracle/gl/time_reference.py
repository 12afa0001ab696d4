"""CPU baseline B1: the reference's own GLSL under software GL
(SwiftShader inside Kaleido's HeadlessChrome 88) on this container's host cores.
Protocol: compile, 2 warm-up draws, >= 10 timed draws each closed by a 1-pixel
readPixels; median ms -> Mpix/s.  Build-container only."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import glref  # noqa: E402
from raymarching_engine_amd import job as J, scene as S  # noqa: E402

LIGHT = [J.point_light((2.0, 3.0, -4.0))]
CASES = [
    ("C1 sphere 256x256 preview [128]", S.single_sphere(), dict(width=256, height=256, counts=(128,), render_mode="preview")),
    ("C2 sphere 960x540 preview [128] (quarter of 1920x1080)", S.single_sphere(), dict(width=960, height=540, counts=(128,), render_mode="preview")),
    ("C3a mandelbulb 480x270 preview [256]", S.Mandelbulb(), dict(width=480, height=270, counts=(256,), render_mode="preview", position=(0, 0, -2.5))),
    ("C3b mandelbulb 480x270 full [256] 1 light", S.Mandelbulb(), dict(width=480, height=270, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=LIGHT)),
    ("C3b mandelbulb 960x540 full [256] 1 light", S.Mandelbulb(), dict(width=960, height=540, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=LIGHT)),
    ("C4 csg64 256x256 full [128] 1 light", S.csg64(), dict(width=256, height=256, counts=(128,), render_mode="full", position=(0, 0, -5.0), lights=LIGHT)),
]
out = []
for name, sc, kw in CASES:
    schema = J.make_schema(sc, **kw)
    h2, h3 = glref.halton(2), glref.halton(3)
    n = 12
    draws = [{"randNoise": glref.u_float(next(h2), next(h3))} for _ in range(n)]
    base = glref.uniforms_from_schema(schema, (0.5, 1 / 3))
    r = glref.run_gl(glref.splice(sc.glsl()), kw["width"], kw["height"], base, draws=draws, time=True)
    t = np.array(r["timings_ms"][2:])
    mp = kw["width"] * kw["height"] / 1e6
    row = {"case": name, "median_ms": float(np.median(t)), "min_ms": float(t.min()), "Mpix_per_s": mp / (float(np.median(t)) / 1e3),
           "cores": r["info"]["cores"], "gl": r["info"]["version"], "ua": r["info"]["ua"]}
    out.append(row)
    print(json.dumps(row), flush=True)
(ROOT / "profiles" / "r01_b1_swiftshader_reference.json").write_text(json.dumps(out, indent=1))
