#!/usr/bin/env python3
"""Timeline of the workgroups of one headline-frame launch (diagnostic build -DRM_DIAG_TIMELINE, tools/_exp_timeline.so):
how many workgroups run at each moment, when the expensive ones start, how long the tail is."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RM_LIB", os.path.join(ROOT, "tools", "_exp_timeline.so"))
import numpy as np
import bench
from raymarching_engine_amd import abi, job as J, native

wl, sc, schema = bench.make_workload("c3b")
W, H = wl["width"], wl["height"]
ctx = native.Context(0)
ctx.set_samples_in_flight(1)
scene = ctx.create_scene(sc)
fb = ctx.create_framebuffer(W, H)
u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
flags = abi.RM_RENDER_FAST | abi.RM_RENDER_NO_OVERLAP
for rep in range(3):  # the third launch runs in cost order
    fb.clear()
    ctx.render_sample(scene, fb, u, None, flags)
    ctx.sync()
c = fb.download(0).view(np.uint32)
# thread 0 of a 16x32-pixel workgroup tile sits at its lower left pixel
t0 = c[0::32, 0::16, 0].astype(np.int64).ravel(); t1 = c[0::32, 0::16, 1].astype(np.int64).ravel()
hw = c[0::32, 0::16, 2].ravel(); xcc = c[0::32, 0::16, 3].ravel()
dur = ((t1 - t0) & 0xffffffff) * 0.01  # us (100 MHz)
start = ((t0 - t0.min()) & 0xffffffff) * 0.01
end = start + dur
total = end.max()
print(f"workgroups {len(dur)}, launch {total:.0f} us; duration us: median {np.median(dur):.0f}, p90 {np.quantile(dur, .9):.0f}, p99 {np.quantile(dur, .99):.0f}, max {dur.max():.0f}")
print(f"sum of durations {dur.sum() / 1e3:.1f} ms = {dur.sum() / total / 256:.2f} workgroups per CU on average (4 fit)")
grid = np.linspace(0, total, 28)
for a, b in zip(grid[:-1], grid[1:]):
    m = 0.5 * (a + b)
    running = int(((start <= m) & (end > m)).sum())
    long_running = int(((start <= m) & (end > m) & (dur > 200)).sum())
    started = int(((start >= a) & (start < b)).sum())
    print(f"t {a:7.0f}-{b:7.0f} us: running {running:5d} (of them > 200 us: {long_running:5d}); started {started:5d}")
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
print("xcc ids seen", np.unique(xcc & 0xf), " heavy (>200us) workgroups per xcc:", [int(((xcc & 0xf) == x)[dur > 200].sum()) for x in range(8)])
print("end time of the last workgroup per xcc (us):", [float(end[(xcc & 0xf) == x].max()) for x in range(8)])
