"""Fast against strict on C4's job at 256 x 256 for the library in RM_LIB, 8 samples: pixels within 1e-3 / 1e-5, the ratio of the frames' means, and -- sky and
geometry pixels apart -- the mean and the share of pixels that received any light.  The measurement behind
profiles/r03_row_culling_smooth_union_experiment.txt (the lit pixels of a smooth-union scene are creeping shadow rays)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
def planes(h, schema, flags, noise):
    fb = ctx.create_framebuffer(schema["render"]["width"], schema["render"]["height"])
    ctx.render_sample(h, fb, J.uniforms_from_schema(schema, noise), None, flags)
    ctx.sync(); out = fb.download(0); fb.destroy(); return out
def planes2(h, schema, flags, noise):
    fb = ctx.create_framebuffer(schema["render"]["width"], schema["render"]["height"])
    ctx.render_sample(h, fb, J.uniforms_from_schema(schema, noise), None, flags)
    ctx.sync(); out = fb.download(2); fb.destroy(); return out
for name, size in (("csg64 256x256", 256),):
    sc = S.csg64(); h = ctx.create_scene(sc)
    schema = J.make_schema(sc, size, size, counts=(128,), render_mode="full", position=(0, 0, -5.0), lights=GC.LIGHT)
    w3, w5, ra = [], [], []
    acc = {}
    for noise in GC.halton_pairs(8):
        a = planes(h, schema, abi.RM_RENDER_STRICT | abi.RM_RENDER_MEGAKERNEL, noise)
        b = planes(h, schema, abi.RM_RENDER_FAST | abi.RM_RENDER_MEGAKERNEL, noise)
        hit = np.isfinite(a).all(-1) & (np.abs(a[..., :3]).sum(-1) > 0)
        d = np.abs(a - b)[..., :3].max(-1) / np.maximum(1e-6, np.abs(a[..., :3]).max(-1))
        sky = planes2(h, schema, abi.RM_RENDER_STRICT | abi.RM_RENDER_MEGAKERNEL, noise)[..., 3] > 1.5e5  # depth plane: the camera ray escaped
        for nm, sel in (("sky", sky), ("geometry", ~sky)):
            acc.setdefault(nm, []).append((a[sel][:, :3].mean(), b[sel][:, :3].mean(), (a[sel][:, :3].sum(-1) > 0).mean(), (b[sel][:, :3].sum(-1) > 0).mean()))
        w3.append((d <= 1e-3).mean()); w5.append((d <= 1e-5).mean()); ra.append(b[..., :3].mean() / a[..., :3].mean())
    for nm, v in acc.items():
        v = np.array(v)
        print(f"   {nm}: mean strict {v[:, 0].mean():.5f} fast {v[:, 1].mean():.5f}; pixels with light: strict {v[:, 2].mean():.4f} fast {v[:, 3].mean():.4f}")
    print(f"{os.environ.get('RM_LIB', 'default')[-20:]}: within 1e-3 {np.mean(w3):.4f} (min {min(w3):.4f}), within 1e-5 {np.mean(w5):.4f}, mean ratio {np.mean(ra):.5f} +- {np.std(ra):.5f}")
