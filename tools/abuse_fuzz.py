#!/usr/bin/env python3
"""Nasty inputs through the C ABI: random scene descriptions and uniforms drawn from {0, -0, +-1, tiny, huge, +-Inf, NaN, ...},
out-of-range enums and counts.  Every call must return -- RM_OK or an error code with a message -- and a render that was
accepted must complete.  python tools/abuse_fuzz.py [n]  (progress is flushed: a call that never returns is named)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raymarching_engine_amd import abi, native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = native.Context(0)
lib = ctx.lib
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
NASTY = np.array([0.0, -0.0, 1.0, -1.0, 1e-30, 1e30, np.inf, -np.inf, np.nan, 0.5, 3.0, 100.0, 0.2, 2.0, 8.0], np.float32)
def nasty(p_nasty=0.3, lo=-3.0, hi=3.0):
    return float(rng.choice(NASTY)) if rng.random() < p_nasty else float(rng.uniform(lo, hi))
fb = ctx.create_framebuffer(32, 24)
ok_scenes = ok_renders = refused_scenes = refused_renders = 0
t0 = time.time()
for it in range(n):
    d = abi.RmSceneDesc()
    lib.rm_material_default(C.byref(d.material))
    d.kind = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 6, 5, 0, 0, 7, -1, 99]))
    p_nasty = 0.25 if rng.random() < 0.4 else 0.0  # a non-finite scene parameter is refused, so most scenes have none
    for k in range(16):
        d.params[k] = nasty(p_nasty, 0.0, 9.0)
    if rng.random() < 0.7:  # keep the loop counts of the iterated kinds small when they are valid at all
        for k in range(16):
            if d.params[k] == d.params[k] and abs(d.params[k]) > 12.0: d.params[k] = float(rng.integers(0, 9))
    nprims = int(rng.choice([1, 2, 5, 17, 64, 256, 1, 3, 0, 257, -3]))
    prims = (abi.RmPrim * max(nprims, 1))()
    for q in range(max(nprims, 0)):
        wild = rng.random() < 0.02
        prims[q].type = int(rng.choice([0, 1, 2, 3, 0, 1, 4, -1] if wild else [0, 1, 0, 1, 0, 1, 2, 3])) | (int(rng.choice([0, 1, 2, 3, 1, 5] if wild else [0, 1, 2, 3, 1])) << 8)
        prims[q].k = nasty(0.05, 0.01, 0.6)
        for a in range(3):
            prims[q].center[a] = nasty(0.01); prims[q].size[a] = nasty(0.01, 0.05, 1.5)
    d.nprims = nprims
    d.prims = C.cast(prims, C.POINTER(abi.RmPrim)) if rng.random() < 0.95 else None
    # round 3: surfaces (RmSurface) -- counts in and out of range, a missing array, indices beyond the count, on domain rows, nasty values
    if rng.random() < 0.4:
        nsurf = int(rng.choice([1, 2, 4, 15, 15, 3, 0, 16, -2, 200]))
        surf = (abi.RmSurface * max(min(nsurf, 256), 1))()
        for q in range(max(min(nsurf, 256), 0)):
            for name, ty in abi.RmSurface._fields_:
                if ty is C.c_float: setattr(surf[q], name, nasty(0.03, 0.0, 2.0))
                else:
                    for a in range(3): getattr(surf[q], name)[a] = nasty(0.03, 0.0, 1.0)
        d.nsurfaces = nsurf
        d.surfaces = C.cast(surf, C.POINTER(abi.RmSurface)) if rng.random() < 0.9 else None
        for q in range(max(nprims, 0)):
            if rng.random() < 0.5:
                prims[q].type |= int(rng.choice([1, 2, 3, 15, 1, 2, 16, 255] if rng.random() < 0.1 else [1, 2, 3, 1, 2, max(1, min(nsurf, 15))])) << 16
    if rng.random() < 0.3:
        for name, ty in abi.RmMaterial._fields_:
            if name in ("sky_axis", "reserved"): setattr(d.material, name, int(rng.choice([0, 1, 2, 3, -1])))
            elif ty is C.c_float: setattr(d.material, name, nasty(0.4, 0.0, 2.0))
    print(f"{it}: scene kind {d.kind} nprims {nprims}", end=" ", flush=True)
    h = C.c_void_p()
    rc = lib.rm_scene_create(ctx.h, C.byref(d), C.byref(h))
    if rc != abi.RM_OK:
        refused_scenes += 1
        print("refused:", lib.rm_last_error(ctx.h).decode()[:70], flush=True)
        continue
    ok_scenes += 1
    u = abi.RmUniforms()
    u.blendWithPreviousFactor = nasty(0.2, 0.0, 1.0)
    u.randNoise[0], u.randNoise[1] = nasty(0.2, 0.0, 1.0), nasty(0.2, 0.0, 1.0)
    for a in range(3): u.position[a] = nasty(0.15)
    for a in range(16): u.rotation[a] = nasty(0.1, -1.0, 1.0) if rng.random() < 0.5 else float(a % 5 == 0)
    u.dofAmount, u.dofFocalPlaneDistance = nasty(0.3, 0.0, 0.1), nasty(0.3, 0.5, 4.0)
    tame = rng.random() < 0.7  # enums and counts in range: only the floats are nasty
    u.cameraMode = int(rng.choice([0, 0, 1, 2] if tame else [0, 0, 1, 2, 3, -1]))
    u.fov, u.aspect, u.fogDensity, u.exposure = nasty(0.2, 0.5, 2.0), nasty(0.2, 0.5, 2.0), nasty(0.3, 0.0, 0.5), nasty(0.2, 0.1, 1.0)
    u.reflections = float(rng.choice([0, 1, 2, 3, 10, 2.5] if tame else [0, 1, 2, 3, 10, 11, -1, np.nan, 2.5, np.inf]))
    for b in range(10): u.raymarchingStepCountsArray[b] = float(rng.choice([0, 1, 7, 16, 32, 32, np.nan, -5, 0.5] if tame else [0, 1, 7, 16, 32, 32, np.nan, -5, np.inf, 2e6, 0.5]))
    u.blendMode, u.renderMode = int(rng.choice([0, 1, 1] if tame else [0, 1, 1, 2, -1])), int(rng.choice([0, 0, 1] if tame else [0, 0, 1, 2, -1]))
    u.lightCount = int(rng.choice([0, 1, 2, 3, 10] if tame else [0, 1, 2, 3, 10, 11, -1]))
    for j in range(10):
        for a in range(3): u.lightPositions[j][a] = nasty(0.2, -4.0, 4.0); u.lightColors[j][a] = nasty(0.2, 0.0, 3.0)
        u.lightSizes[j] = nasty(0.3, 0.0, 1.0)
    u.showDofFocalPlane = int(rng.choice([0, 1, 5]))
    flags = int(rng.choice([0, 1])) | int(rng.choice([0, 4, 16])) | int(rng.choice([0, 0, 2])) | int(rng.choice([0, 0, 64])) | int(rng.choice([0, 0, 128]))
    tile = abi.RmRect(int(rng.integers(-5, 30)), int(rng.integers(-5, 20)), int(rng.integers(-3, 40)), int(rng.integers(-3, 30)))
    print(f"render mode {u.renderMode} refl {u.reflections} lights {u.lightCount} flags {flags}", end=" ", flush=True)
    rc = lib.rm_render_sample(ctx.h, h, fb.h, C.byref(u), C.byref(tile) if rng.random() < 0.5 else None, flags)
    if rc != abi.RM_OK:
        refused_renders += 1
        print("refused:", lib.rm_last_error(ctx.h).decode()[:70], end=" ", flush=True)
    else:
        rc2 = lib.rm_sync(ctx.h)
        assert rc2 == abi.RM_OK, lib.rm_last_error(ctx.h).decode()
        ok_renders += 1
        print("done", end=" ", flush=True)
    lib.rm_scene_destroy(h)
    print(flush=True)
out = fb.download(0)
print(f"{n} cases in {time.time() - t0:.1f} s: scenes accepted {ok_scenes} / refused {refused_scenes}; renders completed {ok_renders} / refused {refused_renders}")
