import os, sys
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
def render(sc, schema, noises, flags):
    h = ctx.create_scene(sc); fb = ctx.create_framebuffer(schema["render"]["width"], schema["render"]["height"])
    for x in noises: ctx.render_sample(h, fb, J.uniforms_from_schema(schema, tuple(x)), None, flags)
    ctx.sync(); out=[fb.download(k) for k in range(3)]; fb.destroy(); h.destroy(); return out
def rel_diff(a,b): return np.abs(a-b)/np.maximum(1.0, np.maximum(np.abs(a),np.abs(b)))
for name, size, counts, lights in (("c4", 2048, (128,), GC.LIGHT), ("c5", 1024, (128,64,64), GC.SOFT_LIGHT)):
    sc=S.csg64(); schema=J.make_schema(sc, size, size, counts=counts, render_mode="full", position=(0,0,-5.0), lights=lights)
    noises=GC.halton_pairs(4)
    strict=render(sc,schema,noises,abi.RM_RENDER_STRICT|abi.RM_RENDER_MEGAKERNEL)
    fast=render(sc,schema,noises,abi.RM_RENDER_FAST)
    ctx.set_gl_stack(1); legal=render(sc,schema,noises,abi.RM_RENDER_STRICT|abi.RM_RENDER_MEGAKERNEL); ctx.set_gl_stack(0)
    geo = strict[2][...,3] < 3.5e6
    a=strict[0][...,:3]
    for rn,m in (("whole",np.ones_like(geo)),("geometry",geo)):
        for other,img in (("fast",fast),("gl-stack",legal)):
            b=img[0][...,:3]; d=rel_diff(a,b).max(-1)[m]; fin=np.isfinite(a[m]).all(-1)&np.isfinite(b[m]).all(-1)
            ratio=float(b[m][fin].mean()/a[m][fin].mean()); diff=(b[m][fin]-a[m][fin]).mean(-1); se=float(diff.std()/np.sqrt(diff.size)/a[m][fin].mean())
            print(name, rn, other, f"> 1e-3 {float((d>1e-3).mean()):.4f} > 1e-5 {float((d>1e-5).mean()):.4f} means {ratio:.4f} +- {se:.4f}")
