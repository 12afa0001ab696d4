#!/usr/bin/env python3
"""The headline frame as a picture: Mandelbulb 3840x2160, full mode, [256], 1 light, SPP samples of the fast build,
tone-mapped by the present pass, box-filtered to 960x540, PNG.  python tools/render_headline.py out.png [spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_cases as GC
from raymarching_engine_amd import abi, capture, job as J, native, scene as S
out = sys.argv[1]; spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = native.Context(0)
sc = S.Mandelbulb(); h = ctx.create_scene(sc)
W, H = 3840, 2160
schema = J.make_schema(sc, W, H, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
fb = ctx.create_framebuffer(W, H)
t0 = time.perf_counter()
ctx.render_samples(h, fb, J.uniforms_from_schema(schema, (0.0, 0.0)), GC.halton_pairs(spp), None, abi.RM_RENDER_FAST)
ctx.sync()
print(f"{spp} samples of {W}x{H}: {time.perf_counter() - t0:.3f} s")
rgba = fb.present(spp).astype(np.float32)                     # rows bottom-up
small = rgba.reshape(H // 4, 4, W // 4, 4, 4).mean((1, 3))    # 4x4 box filter
png = capture.encode_png(np.clip(small + 0.5, 0, 255).astype(np.uint8), bottom_up=True, level=9)
open(out, "wb").write(png)
print("wrote", out, len(png), "bytes")
