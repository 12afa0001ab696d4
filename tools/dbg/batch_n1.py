#!/usr/bin/env python3
"""Whole headline frame on one GPU: ms per sample of rm_render_samples by batch size and launches in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
sc = S.Mandelbulb(); h = ctx.create_scene(sc)
schema = J.make_schema(sc, 3840, 2160, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
u = J.uniforms_from_schema(schema, (0.0, 0.0))
fb = ctx.create_framebuffer(3840, 2160)
noise = GC.halton_pairs(96)
for batch, depth in ((1, 1), (1, 2), (2, 1), (2, 2), (4, 1), (4, 2), (8, 1)):
    ctx.set_sample_batch(batch); ctx.set_samples_in_flight(depth)
    ctx.render_samples(h, fb, u, noise[:16], None, abi.RM_RENDER_FAST); ctx.sync()
    t0 = time.perf_counter(); ctx.render_samples(h, fb, u, noise, None, abi.RM_RENDER_FAST); ctx.sync()
    print(f"batch {batch} in flight {depth}: {(time.perf_counter() - t0) / len(noise) * 1e3:.4f} ms per sample")
