"""tests/golden/random_tables.npz: the reference's own sdf() and castRay() (raymarcher.frag:163-170) run under software GL
on RANDOM scenes of the composition API -- primitive tables of 1..10 spheres / boxes under every operator, a third of
them behind a repeat and / or an unrotated fold row -- at random points and along random rays.  These scenes are
+ - * / sqrt floor abs min max only, so the reference's bits are the bar: the oracle (CPU tests) and the HIP strict build
(GPU tests) must reproduce them exactly.  Build-container only (needs /root/reference and the kaleido wheel):
    python oracle/gl/gen_random_golden.py [scenes | images | kinds | jobs | math | configs | fullsize]
The file holds numbers only: per scene the table rows, the points / rays, and the reference's outputs."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(Path(__file__).resolve().parent))
import glref  # noqa: E402
from gen_golden import FETCH, pack  # noqa: E402
from raymarching_engine_amd import abi, scene as S  # noqa: E402

N_SCENES = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
W = H = 16  # 256 points / rays per scene
STEPS = 24.0


def survives_translation(v: np.float32) -> bool:
    """Chrome 88's ANGLE re-emits a shader's float constants with 8 significant digits before the backend compiles
    it (measured: 2 of 241 random constants come back one ulp off, whether written with 17 digits, with the shortest
    round-trip form or as uintBitsToFloat -- exactly those that '%.8g' does not round-trip).  A scene constant that does
    not survive that is a different scene under this GL stack, so the random scenes only use constants that do."""
    return np.float32(float("%.8g" % float(v))) == np.float32(v)


def random_table(rng) -> S.CsgScene:
    def f32(v):
        v = np.float32(v)
        while not survives_translation(v):
            v = np.nextafter(v, np.float32(0.0))
        return float(v)

    sc = S.CsgScene()
    if rng.random() < 0.3:
        sc.repeat(tuple(f32(v) for v in rng.uniform(2.5, 4.0, 3)))
    if rng.random() < 0.3:
        sc.fold(f32(rng.uniform(0.6, 0.9)), tuple(f32(v) for v in rng.uniform(0.1, 0.5, 3)))
    for i in range(int(rng.integers(1, 11))):
        if i:
            op = rng.integers(0, 4)
            if op == 0: sc.union()
            elif op == 1: sc.smooth_union(f32(rng.uniform(0.05, 0.5)))
            elif op == 2: sc.subtract()
            else: sc.intersect()
        c = tuple(f32(v) for v in rng.uniform(-1.2, 1.2, 3))
        if rng.random() < 0.6: sc.sphere(c, f32(rng.uniform(0.2, 0.9)))
        else: sc.box(c, tuple(f32(v) for v in rng.uniform(0.15, 0.8, 3)))
    return sc


def rows_of(sc) -> np.ndarray:
    return np.array([[p.type & 0xff, (p.type >> 8) & 0xff, p.k, *p.center, *p.size] for p in sc.prims()], np.float32)


def main():
    rng = np.random.default_rng(20261003)
    out = {}
    for i in range(N_SCENES):
        sc = random_table(rng)
        text, uni = sc.glsl(), dict(sc.custom_shader_parameters())
        pts = rng.normal(scale=float(rng.choice([0.7, 2.0, 6.0])), size=(W * H, 3)).astype(np.float32)
        frag = glref.splice(text, "void main(void){ " + FETCH + " fragColor = vec4(sdf(t.xyz), 0.0, 0.0, 0.0); }")
        g = glref.run_gl(frag, W, H, uni, init_prev0=pack(pts, W, H))["planes"][0]
        org = np.array([0.2, 0.1, -4.0], np.float32)
        dirs = rng.normal(size=(W * H, 3)).astype(np.float32); dirs[:, 2] = np.abs(dirs[:, 2]) + np.float32(1.5)
        dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
        u2 = dict(uni); u2["position"] = glref.u_float(*[float(v) for v in org]); u2["hsteps"] = glref.u_float(STEPS)
        harness = "uniform float hsteps;\nvoid main(void){ " + FETCH + " vec3 e = castRay(position, t.xyz, hsteps); fragColor = vec4(e, sdf(e)); }"
        e = glref.run_gl(glref.splice(text, harness), W, H, u2, init_prev0=pack(dirs, W, H))["planes"][0].reshape(-1, 4)
        out[f"rows_{i}"] = rows_of(sc)
        out[f"points_{i}"] = pts
        out[f"sdf_{i}"] = g[..., 0].reshape(-1).astype(np.float32)
        out[f"rays_{i}"] = np.concatenate([np.tile(org, (W * H, 1)), dirs], 1).astype(np.float32)
        out[f"end_{i}"] = e[:, :3].astype(np.float32)
        print(f"scene {i}: {len(out[f'rows_{i}'])} rows, finite ends {np.isfinite(e[:, :3]).all(1).mean():.2f}")
    out["count"] = np.int32(N_SCENES); out["steps"] = np.float32(STEPS)
    dest = ROOT / "tests" / "golden" / "random_tables.npz"
    np.savez_compressed(dest, **out)
    print("wrote", dest, dest.stat().st_size, "B")


MAT_FIELDS = ("diffuse", "diffuse_cutoff", "specular", "specular_cutoff", "roughness", "subsurface", "subsurface_color", "ior", "sky_color", "sky_floor", "sky_scale",
              "sky_radius", "sky_axis")


def images(n_cases: int = 12):
    """tests/golden/random_images.npz: the reference's unmodified main() (tan routed to the portable tangent, as for every
    whole-image golden) on random tables with RANDOM MATERIALS -- colours, roughness, ior, subsurface on and off, sky
    colours -- under 1-3 random lights, two bounces of 128 and 64 steps, 64 x 32 pixels, 2 samples: three planes each."""
    import dataclasses

    import golden_cases as GC
    from raymarching_engine_amd import job as J

    rng = np.random.default_rng(777)

    def f32(v):
        v = np.float32(v)
        while not survives_translation(v):
            v = np.nextafter(v, np.float32(0.0))
        return float(v)

    out = {"count": np.int32(n_cases)}
    for i in range(n_cases):
        sc = random_table(rng)
        sc.material = S.Material(diffuse=tuple(f32(v) for v in rng.uniform(0.1, 0.9, 3)), specular=tuple(f32(v) for v in rng.uniform(0.1, 0.9, 3)),
                                 roughness=f32(rng.uniform(0.05, 0.8)), ior=f32(rng.choice([1.3, 1.5, 2.4, 100.0])), subsurface=f32(rng.choice([11111115.0, 4.0, 0.75])),
                                 subsurface_color=tuple(f32(v) for v in rng.uniform(0.3, 1.0, 3)), sky_color=tuple(f32(v) for v in rng.uniform(0.3, 1.0, 3)),
                                 sky_floor=f32(rng.uniform(0.05, 0.4)), sky_scale=f32(rng.uniform(0.5, 2.5)))
        lights = [J.point_light(tuple(f32(v) for v in rng.uniform(-4, 4, 3)), color=tuple(f32(v) for v in rng.uniform(0.3, 1, 3)), strength=f32(rng.uniform(1, 4)),
                                size=f32(rng.choice([0.0, 0.0, 0.3]))) for _ in range(int(rng.integers(1, 4)))]
        pos = tuple(f32(v) for v in (0.2 + rng.uniform(-0.3, 0.3), 0.1 + rng.uniform(-0.3, 0.3), -4.0))
        schema = J.make_schema(sc, 64, 32, counts=(128, 64), render_mode="full", position=pos, lights=lights, fov=f32(rng.uniform(0.9, 1.6)))
        schema["sdfShaderSource"] = sc.glsl()
        noise = GC.halton_pairs(2)
        base = glref.uniforms_from_schema(schema, noise[0])
        draws = [{"randNoise": glref.u_float(*x)} for x in noise]
        r = glref.run_gl(glref.with_portable_tan(glref.splice(sc.glsl())), 64, 32, base, draws=draws, read=(0, 1, 2))
        pl = r["planes"]
        m = dataclasses.asdict(sc.material)
        out[f"rows_{i}"] = rows_of(sc)
        out[f"material_{i}"] = np.array([x for k in MAT_FIELDS for x in (m[k] if isinstance(m[k], (tuple, list)) else [m[k]])], np.float64)
        out[f"lights_{i}"] = np.array([[*l["position"], *l["color"], l["size"]] for l in lights], np.float64)
        out[f"camera_{i}"] = np.array([*pos, schema["camera"]["mode"]["fov"]], np.float64)
        out[f"color_{i}"], out[f"normal_dof_{i}"], out[f"albedo_depth_{i}"] = pl[0], pl[1], pl[2]
        print(f"image {i}: {len(out[f'rows_{i}'])} rows, {len(lights)} lights, subsurface {sc.material.subsurface}, finite {np.isfinite(pl[0]).all(-1).mean():.2f}")
    out["rand_noise"] = np.array(noise, np.float64)
    dest = ROOT / "tests" / "golden" / "random_images.npz"
    np.savez_compressed(dest, **out)
    print("wrote", dest, dest.stat().st_size, "B")


def random_kind(rng, k: int):
    """One of the reference's example scenes with RANDOM values of its annotated uniforms (within their //@min ... //@max),
    and a camera position for its rays."""
    f = lambda lo, hi: float(np.float32(rng.uniform(lo, hi)))
    if k == 0:
        return S.SphereGridFractal(big_sphere_size=f(2.0, 6.0), iterations=float(rng.integers(1, 9)), grid_scale=f(0.2, 0.6),
                                   big_sphere_center=(f(-1, 1), f(-1, 1), f(6.0, 12.0))), (0.0, 0.0, 0.0)
    if k == 1:
        return S.MengerSponge(iterations=float(rng.integers(1, 9))), (0.5, 0.5, -2.0)
    if k == 2:
        return S.KifsTree(iterations=float(rng.integers(1, 11)), scale=f(0.5, 0.85), angles=(f(-3, 3), f(-3, 3), f(-3, 3)), offset=f(0.7, 1.6), smoothen=False), (0.0, 0.3, -3.5)
    if k == 3:
        return S.KifsTree(iterations=float(rng.integers(1, 11)), scale=f(0.5, 0.85), angles=(f(-3, 3), f(-3, 3), f(-3, 3)), offset=f(0.7, 1.6), smoothen=True), (0.0, 0.3, -3.5)
    return S.KifsBox(iterations=float(rng.integers(1, 17)), scale=f(0.35, 0.7), angles=(f(-1.5, 1.5), f(-1.5, 1.5), f(-1.5, 1.5)), offset=f(0.7, 1.6)), (0.2, 0.1, -3.0)


def kinds(per_kind: int = 5):
    """tests/golden/random_kinds.npz: the reference's own example scenes (fractal1, menger-sponge, tree, smooth-tree,
    rotation-fractal; their text read from /root/reference at run time) with random values of their annotated uniforms:
    sdf() at 256 random points and castRay() along 256 random rays each.  The file holds the parameter values, the
    points / rays and the reference's outputs."""
    rng = np.random.default_rng(4242)
    out, n = {}, 0
    for k in range(5):
        for _ in range(per_kind):
            sc, pos = random_kind(rng, k)
            text, uni = glref.example_scene_text(sc.example), dict(sc.custom_shader_parameters())
            pts = rng.normal(scale=float(rng.choice([0.7, 2.0, 6.0])), size=(W * H, 3)).astype(np.float32)
            frag = glref.splice(text, "void main(void){ " + FETCH + " fragColor = vec4(sdf(t.xyz), 0.0, 0.0, 0.0); }")
            g = glref.run_gl(frag, W, H, uni, init_prev0=pack(pts, W, H))["planes"][0]
            org = np.array(pos, np.float32)
            dirs = rng.normal(size=(W * H, 3)).astype(np.float32); dirs[:, 2] = np.abs(dirs[:, 2]) + np.float32(1.5)
            dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
            u2 = dict(uni); u2["position"] = glref.u_float(*[float(v) for v in org]); u2["hsteps"] = glref.u_float(STEPS)
            harness = "uniform float hsteps;\nvoid main(void){ " + FETCH + " vec3 e = castRay(position, t.xyz, hsteps); fragColor = vec4(e, sdf(e)); }"
            e = glref.run_gl(glref.splice(text, harness), W, H, u2, init_prev0=pack(dirs, W, H))["planes"][0].reshape(-1, 4)
            out[f"kind_{n}"] = np.int32(k)
            out[f"params_{n}"] = np.array(sc.params(), np.float64)
            out[f"points_{n}"] = pts
            out[f"sdf_{n}"] = g[..., 0].reshape(-1).astype(np.float32)
            out[f"rays_{n}"] = np.concatenate([np.tile(org, (W * H, 1)), dirs], 1).astype(np.float32)
            out[f"end_{n}"] = e[:, :3].astype(np.float32)
            print(f"scene {n}: {type(sc).__name__} {sc.params()}, finite ends {np.isfinite(e[:, :3]).all(1).mean():.2f}")
            n += 1
    out["count"] = np.int32(n); out["steps"] = np.float32(STEPS)
    dest = ROOT / "tests" / "golden" / "random_kinds.npz"
    np.savez_compressed(dest, **out)
    print("wrote", dest, dest.stat().st_size, "B")


def jobs(n_cases: int = 24):
    """tests/golden/random_jobs.npz: the reference's unmodified main() (tan routed to the portable tangent) on RANDOM render
    jobs -- a random table with a random material or one of the reference's example scenes with random parameter values;
    the three cameras, with and without rotation; depth of field, fog, 0-3 lights (points, soft points, a sun), 1-3
    bounces, both blend modes, preview (with the focal-plane overlay now and then) and full, 1-3 samples; 64 x 32.
    random_jobs.json holds, per case, the scene (rows + material, or kind + parameters) and the job's settings (numbers and
    option names); the .npz the reference's planes and the RGBA8 canvas its present pass (display.frag) makes of them."""
    import dataclasses
    import json

    import golden_cases as GC
    from raymarching_engine_amd import job as J

    rng = np.random.default_rng(90210)

    def f32(v):
        v = np.float32(v)
        while not survives_translation(v):
            v = np.nextafter(v, np.float32(0.0))
        return float(v)

    out, specs = {"count": np.int32(n_cases)}, []
    for i in range(n_cases):
        spec = {}
        if rng.random() < 0.5:
            sc = random_table(rng)
            sc.material = S.Material(diffuse=tuple(f32(v) for v in rng.uniform(0.1, 0.9, 3)), specular=tuple(f32(v) for v in rng.uniform(0.1, 0.9, 3)),
                                     roughness=f32(rng.uniform(0.05, 0.8)), ior=f32(rng.choice([1.3, 1.5, 2.4, 100.0])), subsurface=f32(rng.choice([11111115.0, 4.0, 0.75])),
                                     subsurface_color=tuple(f32(v) for v in rng.uniform(0.3, 1.0, 3)), sky_color=tuple(f32(v) for v in rng.uniform(0.3, 1.0, 3)),
                                     sky_floor=f32(rng.uniform(0.05, 0.4)), sky_scale=f32(rng.uniform(0.5, 2.5)))
            m = dataclasses.asdict(sc.material)
            spec["scene"] = {"rows": rows_of(sc).astype(np.float64).tolist(),
                             "material": [float(x) for k in MAT_FIELDS for x in (m[k] if isinstance(m[k], (tuple, list)) else [m[k]])]}
            text, pos = sc.glsl(), (0.2, 0.1, -4.0)
        else:
            k = int(rng.integers(0, 5))
            sc, pos = random_kind(rng, k)
            spec["scene"] = {"kind": k, "params": [float(v) for v in sc.params()]}
            text = glref.example_scene_text(sc.example)
        mode = "preview" if rng.random() < 0.25 else "full"
        lights = []
        for _ in range(int(rng.integers(0, 4))):
            if rng.random() < 0.25:
                lights.append(["sun", [f32(v) for v in rng.uniform(-4, 4, 3)], [f32(v) for v in rng.uniform(0.3, 1, 3)], f32(rng.uniform(1, 4)), 0.0])
            else:
                lights.append(["point", [f32(v) for v in rng.uniform(-4, 4, 3)], [f32(v) for v in rng.uniform(0.3, 1, 3)], f32(rng.uniform(1, 4)), f32(rng.choice([0.0, 0.0, 0.3, 1.0]))])
        cam = ("perspective", "perspective", "orthographic", "panoramic")[rng.integers(0, 4)]
        kw = dict(counts=[int(c) for c in rng.integers(16, 96, size=rng.integers(1, 4))], render_mode=mode,
                  position=[f32(pos[j] + rng.uniform(-0.2, 0.2)) for j in range(3)], rotate=bool(rng.random() < 0.5), camera=cam,
                  fov=f32(rng.uniform(0.8, 1.8)) if cam != "orthographic" else f32(rng.uniform(2.0, 5.0)), lights=lights,
                  blend_mode="mix" if rng.random() < 0.25 else "additive", fog_density=f32(rng.choice([0.0, 0.0, 0.05, 0.3])),
                  dof_amount=f32(rng.choice([0.0, 0.0, 0.05])), dof_distance=f32(rng.uniform(1.0, 4.0)),
                  show_focused_area=bool(mode == "preview" and rng.random() < 0.4), samples=int(rng.integers(1, 4)))
        spec["job"] = kw
        specs.append(json.loads(json.dumps(spec)))
        sc2, schema, noise = GC.random_job_from_spec(specs[-1])
        schema = dict(schema); schema["sdfShaderSource"] = text
        base = glref.uniforms_from_schema(schema, noise[0])
        draws = [{"randNoise": glref.u_float(*x)} for x in noise]
        r = glref.run_gl(glref.with_portable_tan(glref.splice(text)), 64, 32, base, draws=draws, read=(0, 1, 2), display_brightness=1.0 / kw["samples"])
        pl = r["planes"]
        out[f"color_{i}"] = pl[0]
        if mode == "full":
            out[f"normal_dof_{i}"], out[f"albedo_depth_{i}"] = pl[1], pl[2]
        out[f"rgba8_{i}"] = r["display"]  # the reference's present pass (display.frag) over these planes, brightness 1 / samples
        if i < 8:  # the same job from the UNMODIFIED text: the GL stack's own tan in the random stream and the camera
            rn = glref.run_gl(glref.splice(text), 64, 32, base, draws=draws, read=(0, 1, 2))["planes"]
            out[f"color_native_{i}"] = rn[0]
            if mode == "full":
                out[f"normal_dof_native_{i}"], out[f"albedo_depth_native_{i}"] = rn[1], rn[2]
        print(f"job {i}: {type(sc).__name__} {mode} {cam} counts {kw['counts']} lights {len(lights)} samples {kw['samples']} finite {np.isfinite(pl[0]).all(-1).mean():.2f}")
    dest = ROOT / "tests" / "golden" / "random_jobs.npz"
    np.savez_compressed(dest, **out)
    (ROOT / "tests" / "golden" / "random_jobs.json").write_text(json.dumps(specs, separators=(",", ":")))
    print("wrote", dest, dest.stat().st_size, "B")


def math(n_side: int = 128):
    """tests/golden/swiftshader_math.npz: the GL stack's own log2 / log / exp2 / exp / pow / sin / cos / tan / asin / acos /
    atan / atan(y, x) on 16 384 arguments each -- random over the ranges the shaders use and far beyond, both signs, zeros,
    denormals, infinities, NaN -- for oracle/ss_math.h, which has to reproduce every one of them bit for bit."""
    w = h = n_side
    n = w * h
    rng = np.random.default_rng(5)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 2.0, 0.5, 1e-45, 1e-40, 1.1754944e-38, 3e38, -3e38, np.inf, -np.inf, np.nan, 1e-20, 1e20], np.float32)
    x = (np.exp(rng.uniform(-20, 20, n)) * rng.choice([1.0, 1.0, 1.0, -1.0], n)).astype(np.float32)
    x[:2048] = rng.uniform(0, 2, 2048); x[2048:4096] = rng.uniform(0.9, 1.1, 2048); x[4096:4096 + 16] = edge
    y = rng.uniform(-8, 8, n).astype(np.float32)
    y[:512] = rng.uniform(-160, 160, 512); y[4096:4096 + 16] = edge[::-1]; y[5000:5016] = edge; x[5000:5016] = edge[::-1]
    ang = rng.uniform(-10, 10, n).astype(np.float32); ang[:4096] = rng.uniform(-900, 900, 4096); ang[4096:4096 + 16] = edge
    a = rng.uniform(-1, 1, n).astype(np.float32); a[:256] = rng.uniform(-1.3, 1.3, 256); a[4096:4096 + 16] = edge
    # arguments at which the restated sine of x or of x + pi/2 leaves [-1, 1] (the stack clamps its cosine, not its sine), tiny
    # angles (rodrigues' cos(theta) near 1), and arguments where x / 2pi is exactly k + 1/2 (the reduction's rounding)
    from oracle import oracle as O
    dense = np.linspace(-10, 10, 4000001).astype(np.float32)
    over = dense[(np.abs(O.ss_math("sin", dense)) > 1) | (np.abs(O.ss_math("sin", dense + np.float32(1.57079632))) > 1)][:64]
    ang[8192:8192 + len(over)] = over
    ang[8300:8556] = (10.0 ** rng.uniform(-7, -2, 256) * rng.choice([1.0, -1.0], 256)).astype(np.float32)
    c = np.float32(1.59154943e-1)
    ties = [v for k in range(-140, 140) for v in [np.float32(np.float32(k + 0.5) / c)] if np.float32(v * c) == np.float32(k + 0.5)]
    ang[8600:8600 + len(ties)] = ties
    pts = np.stack([x, y, ang, a], 1).astype(np.float32)
    text = S.single_sphere().glsl()

    def run(expr):
        frag = glref.splice(text, "void main(void){ " + FETCH + " fragColor = " + expr + "; }")
        return glref.run_gl(frag, w, h, {}, init_prev0=pts.reshape(h, w, 4))["planes"][0].reshape(-1, 4)

    r1 = run("vec4(log2(t.x), log(t.x), exp2(t.y), exp(t.y))")
    r2 = run("vec4(sin(t.z), cos(t.z), pow(t.x, t.y), acos(t.w))")
    r3 = run("vec4(atan(t.y, t.x), atan(t.y), asin(t.w), tan(t.z))")
    dest = ROOT / "tests" / "golden" / "swiftshader_math.npz"
    np.savez_compressed(dest, x=x, y=y, angle=ang, a=a, log2=r1[:, 0], log=r1[:, 1], exp2=r1[:, 2], exp=r1[:, 3], sin=r2[:, 0], cos=r2[:, 1],
                        pow=r2[:, 2], acos=r2[:, 3], atan2=r3[:, 0], atan=r3[:, 1], asin=r3[:, 2], tan=r3[:, 3])
    print("wrote", dest, dest.stat().st_size, "B")


def _frag(sc, native: bool) -> str:
    """The reference's shader with the scene spliced in: as it stands (`native`: its own tan() in the random stream and the
    camera, raymarcher.frag:46-49) or with tan routed to the portable tangent (glref.PORTABLE_TAN_GLSL)."""
    frag = glref.splice(sc.glsl())
    return frag if native else glref.with_portable_tan(frag)


def configs(native: bool = False, only=()):
    """tests/golden/config_<name>[_native].npz: BASELINE.json's configurations -- the headline C3b (Mandelbulb, full, [256], the
    point light), C3a, C2, C4 and C5 with their own scenes, step counts, lights and cameras (tests/golden_cases.py CONFIGS) --
    through the reference's main() under software GL at 256 x 128 / 128 x 128.  `native` (round 6): from the UNMODIFIED text;
    otherwise tan routed to the portable tangent."""
    import golden_cases as GC

    for name in GC.CONFIGS:
        if only and name not in only:
            continue
        sc, schema, noise = GC.config_case(name)
        w, h = schema["render"]["width"], schema["render"]["height"]
        schema = dict(schema); schema["sdfShaderSource"] = sc.glsl()
        base = glref.uniforms_from_schema(schema, noise[0])
        draws = [{"randNoise": glref.u_float(*x)} for x in noise]
        pl = glref.run_gl(_frag(sc, native), w, h, base, draws=draws, read=(0, 1, 2))["planes"]
        arrays = dict(color=pl[0])
        if schema["render"]["renderMode"] == "full":
            arrays.update(normal_dof=pl[1], albedo_depth=pl[2])
        dest = ROOT / "tests" / "golden" / f"config_{name}{'_native' if native else ''}.npz"
        np.savez_compressed(dest, **arrays)
        print("wrote", dest, dest.stat().st_size, "B, finite", float(np.isfinite(pl[0]).all(-1).mean()))


def fullsize(native: bool = False, only=()):
    """tests/golden/rows_<name>[_native].npz: BASELINE configurations at megapixel size -- the headline C3b at 2048 x 1024 and
    4096 x 2048, C4 at 1024 x 1024 ... -- through the reference's main() under software GL (`native`: the unmodified text;
    otherwise the portable tangent); the planes stay here, the fixture is a CRC-32 per row and plane (tests/golden_cases.py
    row_checksums) plus the share of finite pixels per row."""
    import time

    import golden_cases as GC

    for name in GC.ROW_CHECKSUM_CASES:
        if only and name not in only:
            continue
        sc, schema, noise = GC.row_checksum_case(name)
        w, h = schema["render"]["width"], schema["render"]["height"]
        schema = dict(schema); schema["sdfShaderSource"] = sc.glsl()
        base = glref.uniforms_from_schema(schema, noise[0])
        t0 = time.time()
        out = {}
        for k, plane in enumerate(("color", "normal_dof", "albedo_depth") if schema["render"]["renderMode"] == "full" else ("color",)):  # one plane per run: the read-back travels as text
            pl = glref.run_gl(_frag(sc, native), w, h, base, draws=[{"randNoise": glref.u_float(*noise[0])}], read=(k,))["planes"][k]
            out[plane] = GC.row_checksums(pl)
            if k == 0:
                out["finite_share"] = np.isfinite(pl).all(-1).mean(1).astype(np.float32)
            del pl
        dest = ROOT / "tests" / "golden" / f"rows_{name}{'_native' if native else ''}.npz"
        np.savez_compressed(dest, **out)
        print("wrote", dest, dest.stat().st_size, "B in", round(time.time() - t0), "s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "images":
        images()
    elif len(sys.argv) > 1 and sys.argv[1] == "kinds":
        kinds()
    elif len(sys.argv) > 1 and sys.argv[1] == "jobs":
        jobs()
    elif len(sys.argv) > 1 and sys.argv[1] == "math":
        math()
    elif len(sys.argv) > 1 and sys.argv[1] in ("configs", "configs_native"):  # [case ...]
        configs(sys.argv[1].endswith("_native"), sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] in ("fullsize", "fullsize_native"):  # [case ...]
        fullsize(sys.argv[1].endswith("_native"), sys.argv[2:])
    else:
        main()
