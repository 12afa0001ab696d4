#!/usr/bin/env python3
"""Time of the present pass (display.frag) on a 3840x2160 frame: no depth of field, a moderate blur, the 16-pixel cap."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from raymarching_engine_amd import native
W, H = 3840, 2160
ctx = native.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
color = torch.rand((H, W, 4), device="cuda") * 2
out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
for name, radius in (("no depth of field (radius 0)", None), ("radius 4 px", 4.0), ("radius 16 px (cap)", 16.0), ("radius 0..16 ramp", "ramp")):
    ndof = None
    if radius is not None:
        ndof = torch.zeros((H, W, 4), device="cuda")
        ndof[..., 3] = (torch.linspace(0, 16.0 / 200.0, W, device="cuda")[None, :] if radius == "ramp" else radius / 200.0)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            ctx.present_device(color.data_ptr(), ndof.data_ptr() if ndof is not None else None, W, H, 1, out.data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"present 3840x2160, {name}: {dt * 1e3:.3f} ms")
fb = ctx.wrap_framebuffer(W, H, 0, H, color.data_ptr())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    ctx.present_rows(fb, 1, out.data_ptr())
torch.cuda.synchronize(); print(f"present_rows 3840x2160: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
# Round 4: one part's share of the pass (rm_present_striped_rows: what every rank of a sharded job with depth of field runs, on the
# stripes it holds, after the all-gather of the packed rows) by the number of parts, at the blur's 16-pixel cap -- 1 / N of the whole pass
for WW, HH in ((3840, 2160), (8192, 8192)):
    color = torch.rand((HH, WW, 4), device="cuda") * 2
    color[..., 3] = 16.0 / 200.0  # the packed buffer: (colour.rgb, dofRadius) serves as both planes
    whole = None
    for parts in (1, 2, 4, 8):
        rows = (HH // 8 + parts - 1) // parts * 8
        out = torch.zeros((rows, WW, 4), dtype=torch.uint8, device="cuda")
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                ctx.present_striped_rows(color.data_ptr(), color.data_ptr(), WW, HH, 1, 8, parts, 0, out.data_ptr())
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        whole = whole or dt
        print(f"present_striped_rows {WW}x{HH}, radius 16 px (cap), part 0 of {parts}: {dt * 1e3:.3f} ms ({dt / whole:.3f} of the whole pass)")
    del color
