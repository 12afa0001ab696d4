"""The HIP kernels against the reference's own GLSL, BIT FOR BIT, on the GPU.

tests/test_reference_bits.py shows that the oracle, computing in the arithmetic of the GL stack the goldens were rendered
under (its twelve transcendental functions restated in oracle/ss_math.h, its min / max / fract / unorm conventions),
reproduces every golden of tests/golden/.  rm_ctx_set_gl_stack gives the parity build of the kernels the same arithmetic
(csrc/rm_ss_math.hpp is the same text; rm_glstack.hip).  Here that build reproduces the goldens THEMSELVES through the
C ABI -- distances, marches, normals, the random stream, all 29 whole-main() images, the random tables / example-scene
parameters / materials / render jobs, the present pass byte for byte -- and equals the oracle in that arithmetic on
random jobs no golden holds."""
import os

import numpy as np
import pytest

import golden_cases as GC
from oracle import oracle as O
from raymarching_engine_amd import abi, job as J
from test_gpu_parity import MK, STRICT, _random_scene, load, render_gpu, render_oracle, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def glctx():
    from raymarching_engine_amd import native

    c = native.Context(0)
    c.set_gl_stack(True)
    O.set_tan_mode(O.TAN_PORTABLE)
    yield c
    c.close()


def _planes_equal(z, got, suffix="", full=True):
    for k, name in enumerate(("color", "normal_dof", "albedo_depth") if full else ("color",)):
        eq = same_bits(z[name + suffix], got[k])
        assert eq.all(), f"{name}{suffix}: {int((~eq.all(-1)).sum())} pixels differ from the reference"


@pytest.mark.parametrize("name", list(GC.SCENES))
def test_sdf_is_the_reference_bit_for_bit(glctx, name):
    z = load("sdf_" + name)
    h = glctx.create_scene(GC.build_scene(name))
    assert same_bits(glctx.probe(h, abi.RM_PROBE_SDF, z["points"]), z["sdf"]).all()
    h.destroy()


@pytest.mark.parametrize("name", list(GC.CAST))
def test_cast_ray_and_normal_are_the_reference_bit_for_bit(glctx, name):
    z = load("cast_" + name)
    h = glctx.create_scene(GC.build_scene(name))
    assert same_bits(glctx.probe(h, abi.RM_PROBE_CAST_RAY, z["rays"], float(z["steps"])), z["end"]).all()
    assert same_bits(glctx.probe(h, abi.RM_PROBE_NORMAL, z["normal_points"], 1e-5), z["normal"]).all()
    h.destroy()


@pytest.mark.parametrize("name", abi.RM_MATH_FUNCTIONS)
def test_gl_stack_transcendentals_have_the_oracles_bits_on_the_gpu(glctx, name):
    """The GL stack's functions as the kernels compile them (csrc/rm_ss_math.hpp) against the oracle's (oracle/ss_math.h, itself
    pinned by tests/golden/swiftshader_math.npz), argument by argument (rm_probe_math on a GL-stack context)."""
    from test_gpu_parity import math_arguments

    a, b = math_arguments(name, 200_000, 515 + abi.RM_MATH_FUNCTIONS.index(name))
    O.set_math_mode(O.MATH_SWIFTSHADER)
    try:
        want = O.math(name, a, b)
    finally:
        O.set_math_mode(O.MATH_PORTABLE)
    got = glctx.probe_math(name, a, b)
    eq = same_bits(got, want)
    bad = np.flatnonzero(~eq)
    assert eq.all(), f"{name}: {bad.size} of {a.size} differ, e.g. " + "; ".join(f"f({a[i]!r}, {None if b is None else b[i]!r}) = {got[i]!r} vs {want[i]!r}" for i in bad[:6])


def test_random_stream_is_the_reference_bit_for_bit(glctx):
    r = load("rng_32x32")
    u = J.uniforms_from_schema(J.make_schema(GC.build_scene("sphere"), 32, 32), tuple(r["rand_noise"]))
    assert same_bits(glctx.probe_rng(u, 32, 32, 4), r["uniform4"]).all()


TEXTS = ("portable_tan", "unmodified")


@pytest.fixture
def text_ctx(glctx, request):
    """The GL-stack context set to the tangent of the text a golden was rendered from: rm_ctx_set_gl_stack(ctx, 1) for the
    goldens with the portable tangent injected, (ctx, 2) -- the stack's own tan() -- for raymarcher.frag as it stands."""
    text = request.param
    glctx.set_gl_stack(2 if text == "unmodified" else 1)
    yield glctx, ("_native" if text == "unmodified" else "")
    glctx.set_gl_stack(1)


@pytest.mark.parametrize("text_ctx", TEXTS, indirect=True)
@pytest.mark.parametrize("case", list(GC.IMAGES))
def test_whole_main_image_is_the_reference_bit_for_bit(text_ctx, case):
    """All 29 image cases: every value of every plane the reference's main() wrote under its GL stack, for both texts."""
    ctx, suffix = text_ctx
    sc, samples, schema = GC.image_schema(case)
    z = load("image_" + case + suffix)
    _planes_equal(z, render_gpu(ctx, sc, schema, z["rand_noise"], STRICT | MK), full="normal_dof" in z)


@pytest.mark.parametrize("text_ctx", TEXTS, indirect=True)
@pytest.mark.parametrize("name", list(GC.CONFIGS))
def test_baseline_configurations_are_the_reference_bit_for_bit(text_ctx, name):
    """BASELINE.json's configurations (the headline C3b, C3a, C2, C4, C5: their own scenes, step counts, lights, cameras) at
    256 x 128 / 128 x 128: the GPU reproduces what the reference's main() rendered of them under its GL stack -- from the text
    with the portable tangent and (round 6) from the UNMODIFIED text."""
    ctx, suffix = text_ctx
    sc, schema, noises = GC.config_case(name)
    z = load("config_" + name + suffix)
    _planes_equal(z, render_gpu(ctx, sc, schema, noises, STRICT | MK), full="normal_dof" in z)


@pytest.mark.parametrize("text_ctx", TEXTS, indirect=True)
@pytest.mark.parametrize("name", list(GC.ROW_CHECKSUM_CASES))
def test_megapixel_configurations_every_row_checksum(text_ctx, name):
    """BASELINE's configurations at megapixel size as the reference's GLSL rendered them under software GL -- the headline C3b
    at 2048 x 1024 and at 4096 x 2048 (the pixel count of its own frame), C3a and C2 at 2048 x 1024, C4 and C5 at 1024 x 1024; a
    CRC-32 per row and plane in the fixture: the GPU reproduces every row of every plane, for both texts."""
    ctx, suffix = text_ctx
    sc, schema, noises = GC.row_checksum_case(name)
    z = load("rows_" + name + suffix)
    got = render_gpu(ctx, sc, schema, noises, STRICT | MK)
    for k, plane in enumerate(("color", "normal_dof", "albedo_depth") if schema["render"]["renderMode"] == "full" else ("color",)):
        crc = GC.row_checksums(got[k])
        assert (crc == z[plane]).all(), f"{plane}: {int((crc != z[plane]).sum())} of {len(crc)} rows differ"


def test_random_goldens_are_reproduced_bit_for_bit(glctx):
    """Random tables (sdf, castRay), the example scenes with random parameters, random-material images, random jobs."""
    z = load("random_tables")
    for i in range(int(z["count"])):
        h = glctx.create_scene(GC.table_from_rows(z[f"rows_{i}"]))
        assert same_bits(glctx.probe(h, abi.RM_PROBE_SDF, z[f"points_{i}"]), z[f"sdf_{i}"]).all()
        assert same_bits(glctx.probe(h, abi.RM_PROBE_CAST_RAY, z[f"rays_{i}"], float(z["steps"])), z[f"end_{i}"]).all()
        h.destroy()
    z = load("random_kinds")
    for i in range(int(z["count"])):
        h = glctx.create_scene(GC.random_kind_case(z, i))
        assert same_bits(glctx.probe(h, abi.RM_PROBE_SDF, z[f"points_{i}"]), z[f"sdf_{i}"]).all(), f"kind {i}"
        assert same_bits(glctx.probe(h, abi.RM_PROBE_CAST_RAY, z[f"rays_{i}"], float(z["steps"])), z[f"end_{i}"]).all(), f"kind {i}"
        h.destroy()
    z = load("random_images")
    for i in range(int(z["count"])):
        sc, schema, noises = GC.random_image_case(z, i)
        _planes_equal(z, render_gpu(glctx, sc, schema, noises, STRICT | MK), f"_{i}")
    z = load("random_jobs")
    for i in range(int(z["count"])):
        sc, schema, noises = GC.random_job_case(z, i)
        _planes_equal(z, render_gpu(glctx, sc, schema, noises, STRICT | MK), f"_{i}", schema["render"]["renderMode"] == "full")


def test_present_pass_is_the_reference_byte_for_byte(glctx):
    """rm_present in the GL stack's arithmetic on the planes the reference's own present pass was run on: its canvas,
    every byte (depth-of-field blurs, non-finite colours -- white under this stack -- included)."""
    cases = [(z["color"], z["normal_dof"], int(z["samples"]), z["rgba8"]) for z in (load("display_dof"), load("display_nodof"))]
    z = load("random_jobs")
    cases += [GC.random_job_present_case(z, i)[:4] for i in range(int(z["count"]))]
    for k, (color, ndof, n, ref) in enumerate(cases):
        fb = glctx.create_framebuffer(color.shape[1], color.shape[0])
        fb.upload(0, color)
        fb.upload(1, ndof)
        got = fb.present(n)
        fb.destroy()
        assert np.array_equal(got, ref), f"case {k}: {int((got != ref).sum())} bytes differ"


def test_random_jobs_equal_the_oracle_in_the_gl_stacks_arithmetic(glctx):
    """Random jobs no golden holds (the generator of tests/test_gpu_parity.py: every kind, random parameters and materials,
    cameras, depth of field, fog, lights, bounces, blend modes): the GL-stack build against the oracle computing with the
    same twelve functions and conventions -- every plane bit-identical."""
    rng = np.random.default_rng(4711 + int(os.environ.get("RM_RANDOM_SEED", "0")))
    O.set_math_mode(O.MATH_SWIFTSHADER)
    try:
        for it in range(int(os.environ.get("RM_RANDOM_GL_JOBS", "100"))):
            sc, pos = _random_scene(rng)
            w, h = int(rng.integers(24, 72)), int(rng.integers(16, 56))
            mode = "preview" if rng.random() < 0.25 else "full"
            counts = tuple(int(c) for c in rng.integers(6, 40, size=rng.integers(1, 5)))
            lights = [J.sun_light(tuple(rng.uniform(-4, 4, 3)), color=tuple(rng.uniform(0.3, 1, 3))) if rng.random() < 0.25 else
                      J.point_light(tuple(rng.uniform(-4, 4, 3)), color=tuple(rng.uniform(0.3, 1, 3)), strength=float(rng.uniform(1, 4)), size=float(rng.choice([0.0, 0.0, 0.3, 1.0])))
                      for _ in range(int(rng.integers(0, 4)))]
            cam = ("perspective", "perspective", "orthographic", "panoramic")[rng.integers(0, 4)]
            schema = J.make_schema(sc, w, h, counts=counts, render_mode=mode, position=tuple(np.array(pos) + rng.uniform(-0.2, 0.2, 3)),
                                   rotation=GC.ROT if rng.random() < 0.5 else None, camera=cam, fov=float(rng.uniform(0.8, 1.8)) if cam != "orthographic" else float(rng.uniform(2.0, 5.0)),
                                   lights=lights, blend_mode="mix" if rng.random() < 0.25 else "additive", fog_density=float(rng.choice([0.0, 0.0, 0.05, 0.3])),
                                   dof_amount=float(rng.choice([0.0, 0.0, 0.05])), dof_distance=float(rng.uniform(1.0, 4.0)),
                                   show_focused_area=bool(mode == "preview" and rng.random() < 0.3))
            noises = GC.halton_pairs(int(rng.integers(1, 4)))
            want = render_oracle(sc, schema, noises, nan_mode=O.NAN_X86)
            got = render_gpu(glctx, sc, schema, noises, STRICT | MK)
            for k in range(3 if mode == "full" else 1):
                eq = same_bits(want[k], got[k])
                assert eq.all(), f"job {it}: {type(sc).__name__} {w}x{h} {mode} counts {counts} camera {cam} lights {len(lights)}: plane {k}, {int((~eq).sum())} values differ"
    finally:
        O.set_math_mode(O.MATH_PORTABLE)


def test_the_unmodified_shader_text_bit_for_bit(glctx):
    """rm_ctx_set_gl_stack(ctx, 2): the GL stack's own tan() in the random stream and the camera, i.e. the reference's text as
    it stands.  The two 256-sample goldens (sphere 32 x 16, Mandelbulb with its light 64 x 32: every pixel's sum) and the 8
    random jobs rendered from the unmodified text: every value."""
    glctx.set_gl_stack(2)
    try:
        for name, scene, kw in (("stat_sphere_full_native_tan", "sphere", dict(render_mode="full", counts=(64, 32), exposure=1.0)),
                                ("stat_mandelbulb_full_native_tan", "mandelbulb", dict(render_mode="full", counts=(64,), position=(0, 0, -2.5), lights=GC.LIGHT, exposure=1.0))):
            z = load(name)
            h, w = z["color_sum"].shape[:2]
            sc = GC.build_scene(scene)
            got = render_gpu(glctx, sc, J.make_schema(sc, w, h, **kw), GC.halton_pairs(int(z["samples"])), STRICT | MK)
            assert same_bits(got[0], z["color_sum"]).all(), name
        z = load("random_jobs")
        for i in range(int(z["count"])):
            if f"color_native_{i}" not in z:
                continue
            sc, schema, noises = GC.random_job_case(z, i)
            got = render_gpu(glctx, sc, schema, noises, STRICT | MK)
            for k, name in enumerate(("color", "normal_dof", "albedo_depth") if schema["render"]["renderMode"] == "full" else ("color",)):
                assert same_bits(got[k], z[f"{name}_native_{i}"]).all(), f"job {i} {name}"
    finally:
        glctx.set_gl_stack(1)


def test_gl_stack_arithmetic_is_partition_and_batch_invariant(glctx):
    """The GL-stack arithmetic goes through the same launch machinery as the default one: a row window, a tile, and three
    samples in one launch (rm_render_samples) give the bits of the whole-frame, sample-by-sample render."""
    sc, schema, noises = GC.random_job_case(load("random_jobs"), 6)
    noises = GC.halton_pairs(3)
    whole = render_gpu(glctx, sc, schema, noises, STRICT | MK)
    part = render_gpu(glctx, sc, schema, noises, STRICT | MK, rows=(8, 16))
    for k in range(3):
        assert same_bits(part[k], whole[k][8:24]).all(), f"row window, plane {k}"
    h = glctx.create_scene(sc)
    fb = glctx.create_framebuffer(GC.IMG_W, GC.IMG_H)
    glctx.render_samples(h, fb, J.uniforms_from_schema(schema, noises[0]), noises, None, STRICT | MK)
    for k in range(3):
        assert same_bits(fb.download(k), whole[k]).all(), f"three samples in one launch, plane {k}"
    fb.destroy()
    h.destroy()
