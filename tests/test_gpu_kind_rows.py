"""A kind as a SHAPE of a primitive table on the GPU (RM_PRIM_KIND, round 4): the strict build gives the oracle's bits -- distances,
castRay end points, normals, material lookups, whole frames of main() at a size the goldens do not have --, the fast build its
statistics, the C ABI refuses what a table cannot hold.  (The GL-stack build against the reference's own renders of the two
composed scenes: tests/test_gpu_reference_bits.py runs every case of tests/golden_cases.py.)"""
import ctypes as C

import numpy as np
import pytest

import golden_cases as GC
from oracle import oracle as O
from raymarching_engine_amd import abi, job as J, native, scene as S

pytestmark = pytest.mark.gpu
STRICT, FAST, MK = abi.RM_RENDER_STRICT, abi.RM_RENDER_FAST, abi.RM_RENDER_MEGAKERNEL


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    O.set_tan_mode(O.TAN_PORTABLE)
    yield c
    c.close()


SCENES = {
    "csg_bulb_cut": (0.25, 0.125, -1.625),
    "csg_lattice_ball": (0.25, 0.5, -3.5),
    # a lattice inside a repeated, folded space with a smooth union: kind rows see the domain rows' point and factor like any shape
    "folded": (0.2, 0.1, -2.2),
}


def build(name):
    if name == "folded":
        return S.CsgScene().fold(0.75, (0.5, 0.125, 0.25)).box((0, 0, 0), (0.5, 0.25, 0.25)).smooth_union(0.125).shape(S.SphereLattice(0.5, 0.125), (0.0625, 0.0, 0.0))
    return GC.build_scene(name)


@pytest.mark.parametrize("name", list(SCENES))
def test_strict_build_equals_the_oracle(ctx, name):
    sc, pos = build(name), SCENES[name]
    h = ctx.create_scene(sc)
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-2, 2, (20000, 3)), rng.normal(size=(2000, 3)) * 30.0, [[np.nan, 0, 0], [np.inf, 1, 1], [0, 0, 0]]]).astype(np.float32)
    assert same_bits(ctx.probe(h, abi.RM_PROBE_SDF, pts, 0.0, STRICT), O.eval_sdf(sc, pts)).all()
    assert same_bits(ctx.probe(h, abi.RM_PROBE_MATERIAL, pts[:4000], 0.0, STRICT), O.material(sc, pts[:4000])).all()
    rays = GC.camera_rays(pos, 64, 32)
    end = ctx.probe(h, abi.RM_PROBE_CAST_RAY, rays, 48.0, STRICT)
    assert same_bits(end, O.cast_ray(sc, rays, 48.0)).all()
    hit = end[np.isfinite(end).all(-1)]
    assert same_bits(ctx.probe(h, abi.RM_PROBE_NORMAL, hit, 1e-5, STRICT), O.normal(sc, hit, 1e-5)).all()
    h.destroy()
    # whole frames: full mode with two bounces and the light, three samples; preview
    for kw in (dict(render_mode="full", counts=(40, 20), lights=GC.LIGHT), dict(render_mode="preview", counts=(48,))):
        schema = J.make_schema(sc, 160, 96, position=pos, **kw)
        hh = ctx.create_scene(sc)
        fb = ctx.create_framebuffer(160, 96)
        fr = O.Frame(160, 96)
        for nz in GC.halton_pairs(3):
            u = J.uniforms_from_schema(schema, nz)
            ctx.render_sample(hh, fb, u, None, STRICT)
            O.render(sc, u, fr, threads=min(32, O.host_cores()))
        assert ctx.last_pipeline() == "megakernel"
        for k, want in enumerate((fr.color, fr.normal_dof, fr.albedo_depth)):
            if kw["render_mode"] == "preview" and k:
                break
            got = fb.download(k)
            assert same_bits(got, want).all(), f"{name} {kw['render_mode']} plane {k}: {int((~same_bits(got, want)).sum())} values differ from the oracle"
        assert np.isfinite(fr.albedo_depth[..., 3]).all() and (fr.albedo_depth[..., 3] < 300).mean() > 0.05 if kw["render_mode"] == "full" else True
        fb.destroy()
        hh.destroy()


def test_fast_build_estimates_the_same_image(ctx):
    """the fast policy evaluates the kind row with the fast evaluators (the power-8 Mandelbulb's trig-free form included): means of
    the lit pixels within a few per cent of the strict build's at 32 samples"""
    sc, pos = GC.build_scene("csg_bulb_cut"), SCENES["csg_bulb_cut"]
    schema = J.make_schema(sc, 96, 64, render_mode="full", counts=(48, 24), lights=GC.LIGHT, position=pos)
    means = []
    for flags in (STRICT, FAST):
        h = ctx.create_scene(sc)
        fb = ctx.create_framebuffer(96, 64)
        for nz in GC.halton_pairs(32):
            ctx.render_sample(h, fb, J.uniforms_from_schema(schema, nz), None, flags)
        c = fb.download(0)[..., :3]
        means.append(float(np.nanmean(c[np.isfinite(c).all(-1)])))
        fb.destroy()
        h.destroy()
    assert abs(means[1] / means[0] - 1.0) < 0.05, means


def test_the_c_abi_refuses_what_a_table_cannot_hold(ctx):
    def create(rows, params):
        d = abi.RmSceneDesc()
        d.kind = abi.RM_SCENE_TABLE
        arr = (abi.RmPrim * len(rows))()
        for r, (typ, size0) in zip(arr, rows):
            r.type, r.size[0] = typ, size0
        d.nprims, d.prims = len(rows), C.cast(arr, C.POINTER(abi.RmPrim))
        for i, v in enumerate(params):
            d.params[i] = v
        d.material = S.Material().to_c()
        out = C.c_void_p()
        rc = ctx.lib.rm_scene_create(ctx.h, C.byref(d), C.byref(out))
        if rc == 0:
            ctx.lib.rm_scene_destroy(out)
        return rc, ctx.lib.rm_last_error(ctx.h).decode()

    K = abi.RM_PRIM_KIND
    assert create([(K, float(abi.RM_SCENE_MANDELBULB))], [8.0, 4.0, 2.0])[0] == 0
    rc, why = create([(K, float(abi.RM_SCENE_MENGER))], [4.0])
    assert rc != 0 and "kind row" in why
    rc, why = create([(K, 0.5)], [8.0, 4.0, 2.0])
    assert rc != 0 and "kind row" in why
    rc, why = create([(K, float(abi.RM_SCENE_MANDELBULB)), (K, float(abi.RM_SCENE_SPHERE_LATTICE))], [8.0, 4.0, 2.0])
    assert rc != 0 and "ONE kind" in why
    rc, why = create([(K, float(abi.RM_SCENE_MANDELBULB))], [8.0, 1000.0, 2.0])
    assert rc != 0 and "iterations" in why
    rc, why = create([(K, float(abi.RM_SCENE_SPHERE_LATTICE))], [0.0, 0.25])
    assert rc != 0 and "period" in why
