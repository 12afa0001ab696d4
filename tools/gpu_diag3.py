import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_cases as GC
from oracle import oracle as O
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0); O.set_tan_mode(O.TAN_PORTABLE)
sc = S.single_sphere(); h = ctx.create_scene(sc)
for counts in ((64,), (66,), (68,), (70,), (72,)):
    schema = J.make_schema(sc, 64, 32, render_mode="full", counts=counts, lights=GC.LIGHT)
    u = J.uniforms_from_schema(schema, (0.5, 1/3))
    fr = O.Frame(64, 32); O.render(sc, u, fr)
    out = {}
    for nm, fl in (("strict", 0), ("fast", 1)):
        fb = ctx.create_framebuffer(64, 32); ctx.render_sample(h, fb, u, None, fl)
        out[nm] = (fb.download(0), fb.download(2)); fb.destroy()
    print(counts, "oracle", fr.color[0,0], fr.albedo_depth[0,0,3], "strict", out["strict"][0][0,0], out["strict"][1][0,0,3], "fast", out["fast"][0][0,0], out["fast"][1][0,0,3])
