#!/usr/bin/env python3
"""Where does a context's device memory go, and does it come back?  Free memory after every phase of a context's life."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S, shard
base = native.Context(0)
sc = S.Mandelbulb()
schema = J.make_schema(sc, 640, 512, counts=(16,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
noises = GC.halton_pairs(9)
u = lambda n: J.uniforms_from_schema(schema, tuple(n))
def free(): return base.device_memory()[0] / 2**20
f0 = free(); last = [f0]
def mark(what):
    f = free(); print(f"{what:50s} free {f:10.1f} MiB  (delta {f - last[0]:+8.1f}, since start {f - f0:+8.1f})"); last[0] = f
which = sys.argv[1:] or ["all"]
for rep in range(3):
    print("--- context", rep)
    c = native.Context(0); mark("context created")
    h = c.create_scene(sc); fb = c.create_framebuffer(640, 512); mark("scene + framebuffer")
    c.set_samples_in_flight(1)
    c.render_sample(h, fb, u(noises[0]), None, abi.RM_RENDER_FAST); c.sync(); mark("one fast sample, no overlap")
    c.render_sample(h, fb, u(noises[0]), None, abi.RM_RENDER_STRICT); c.sync(); mark("one strict sample")
    c.set_samples_in_flight(3)
    for n in noises[:3]: c.render_sample(h, fb, u(n), None, abi.RM_RENDER_FAST)
    c.sync(); mark("3 samples in flight")
    c.render_samples(h, fb, u(noises[0]), [tuple(n) for n in noises], None, abi.RM_RENDER_FAST); c.sync(); mark("a batch of 9")
    c.render_sample(h, fb, u(noises[0]), None, abi.RM_RENDER_STRICT | abi.RM_RENDER_WAVEFRONT); c.sync(); mark("wavefront sample")
    fb.present(4); mark("present")
    b = c.buffer(640 * 512 * 4); mark("buffer")
    b.destroy(); fb.destroy(); h.destroy(); mark("objects destroyed")
    c.close(); mark("context closed")
