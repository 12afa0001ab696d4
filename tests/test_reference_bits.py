"""The oracle against the reference's own GLSL, BIT FOR BIT, on every golden of tests/golden/ -- transcendental or not.

GLSL leaves the precision of sin / cos / log / exp / pow / asin / acos / atan (and log(0), fract(Inf), min / max of a NaN,
the unorm conversion of the canvas) to the implementation, so "what the reference computes" has one value only together
with a GL stack.  The goldens were rendered with SwiftShader (HeadlessChrome 88 of the kaleido wheel); oracle/ss_math.h
restates that stack's transcendental approximations as IEEE fp32 operation sequences, pinned function by function by
tests/golden/swiftshader_math.npz (the stack's own outputs).  With them (OR_MATH_SWIFTSHADER) and the stack's conventions
(OR_NAN_X86) the oracle -- the same C text, the same main(), only those twelve functions exchanged -- reproduces every
number the reference produced here: distances, marches, normals, whole images from the UNMODIFIED shader text (its own
tan included), 256-sample accumulations, random scenes / materials / jobs, and the canvas of its present pass, byte for
byte.  The other half of the chain is tests/test_gpu_parity.py: the same oracle with the portable transcendentals
(oracle/pm_math.h = csrc/rm_pm_math.hpp) equals the HIP strict build bit for bit.  tests/test_oracle_golden.py keeps the
comparisons of the goldens with the portable / libm modes, i.e. what exchanging the twelve functions does to an image."""
import numpy as np
import pytest

import golden_cases as GC
from oracle import oracle as O
from raymarching_engine_amd import job as J
from test_oracle_golden import load, same_bits


@pytest.fixture(autouse=True)
def _gl_stack_arithmetic():
    O.set_tan_mode(O.TAN_PORTABLE)  # most goldens were rendered with tan routed to the portable tangent; the native ones say so
    O.set_math_mode(O.MATH_SWIFTSHADER)
    yield
    O.set_math_mode(O.MATH_PORTABLE)
    O.set_tan_mode(O.TAN_PORTABLE)


X86 = dict(nan_mode=O.NAN_X86)

SS_ARGUMENTS = {"log2": ("x",), "log": ("x",), "exp2": ("y",), "exp": ("y",), "sin": ("angle",), "cos": ("angle",), "pow": ("x", "y"), "acos": ("a",),
                "atan2": ("y", "x"), "atan": ("y",), "asin": ("a",), "tan": ("angle",)}


@pytest.mark.parametrize("name", list(SS_ARGUMENTS))
def test_gl_stack_transcendentals_restated_bit_for_bit(name):
    """oracle/ss_math.h against the GL stack's own outputs (oracle/gl/gen_random_golden.py math): 16 384 arguments per
    function -- the shaders' ranges and far beyond, both signs, zeros, denormals, infinities, NaN, the arguments where the
    sine leaves [-1, 1] (the cosine is clamped, the sine is not), tiny angles, the ties of the range reduction."""
    z = load("swiftshader_math")
    got = O.ss_math(name, *[z[k] for k in SS_ARGUMENTS[name]])
    assert same_bits(got, z[name]).all(), f"{name}: {int((~same_bits(got, z[name])).sum())} of {got.size} differ"


@pytest.mark.parametrize("name", list(GC.SCENES))
def test_sdf_bit_for_bit(name):
    z = load("sdf_" + name)
    assert same_bits(O.eval_sdf(GC.build_scene(name), z["points"], **X86), z["sdf"]).all()


@pytest.mark.parametrize("name", list(GC.CAST))
def test_cast_ray_and_normal_bit_for_bit(name):
    z = load("cast_" + name)
    sc = GC.build_scene(name)
    assert same_bits(O.cast_ray(sc, z["rays"], float(z["steps"]), **X86), z["end"]).all()
    assert same_bits(O.normal(sc, z["normal_points"], 1e-5, **X86), z["normal"]).all()


def _render(sc, schema, noises, w=GC.IMG_W, h=GC.IMG_H, **kw):
    fr = O.Frame(w, h)
    for n in noises:
        O.render(sc, J.uniforms_from_schema(schema, tuple(n)), fr, **X86, **kw)
    return fr


def _planes_equal(z, fr, suffix="", full=True):
    for name in ("color", "normal_dof", "albedo_depth") if full else ("color",):
        eq = same_bits(z[name + suffix], getattr(fr, name))
        assert eq.all(), f"{name}{suffix}: {int((~eq.all(-1)).sum())} pixels differ"


TEXTS = ("portable_tan", "unmodified")


def _golden_of_text(prefix, name, text):
    """The fixture of a case rendered from the reference's text with tan routed to the portable tangent, or (round 6) from the
    text exactly as it stands; the oracle's tangent is set to match."""
    O.set_tan_mode(O.TAN_SWIFTSHADER if text == "unmodified" else O.TAN_PORTABLE)
    return load(prefix + name + ("_native" if text == "unmodified" else ""))


@pytest.mark.parametrize("text", TEXTS)
@pytest.mark.parametrize("case", list(GC.IMAGES))
def test_whole_main_image_bit_for_bit(case, text):
    """All 29 image cases (13 scenes; preview and full, three cameras, DoF, fog, both blend modes, 1-3 lights, subsurface
    scattering, up to 4 accumulated samples): every value of every plane -- of the goldens rendered with the portable tangent
    injected and of the ones rendered from raymarcher.frag exactly as it stands."""
    sc, samples, schema = GC.image_schema(case)
    z = _golden_of_text("image_", case, text)
    _planes_equal(z, _render(sc, schema, z["rand_noise"]), full="normal_dof" in z)


@pytest.mark.parametrize("name,scene,kw", [
    ("stat_sphere_full_native_tan", "sphere", dict(render_mode="full", counts=(64, 32), exposure=1.0)),
    ("stat_mandelbulb_full_native_tan", "mandelbulb", dict(render_mode="full", counts=(64,), position=(0, 0, -2.5), lights=GC.LIGHT, exposure=1.0)),
])
def test_256_samples_of_the_unmodified_shader_text_bit_for_bit(name, scene, kw):
    """The two goldens rendered from the reference's text as it stands -- its own tan() in the random stream and the
    camera, no substitution -- 256 samples accumulated: every pixel's sum is reproduced to the bit (sphere 32 x 16,
    Mandelbulb with its light 64 x 32)."""
    z = load(name)
    n = int(z["samples"])
    h, w = z["color_sum"].shape[:2]
    sc = GC.build_scene(scene)
    O.set_tan_mode(O.TAN_SWIFTSHADER)
    fr = _render(sc, J.make_schema(sc, w, h, **kw), GC.halton_pairs(n), w, h, threads=min(8, O.host_cores()))
    assert same_bits(fr.color, z["color_sum"]).all()


def test_random_scenes_materials_and_jobs_bit_for_bit():
    """Every randomised golden: 32 random tables (sdf, castRay), the example scenes with 25 random parameter settings,
    12 random-material images, 24 random render jobs -- and 8 of those jobs once more from the unmodified text (native
    tan): all planes, all pixels."""
    z = load("random_tables")
    for i in range(int(z["count"])):
        sc = GC.table_from_rows(z[f"rows_{i}"])
        assert same_bits(O.eval_sdf(sc, z[f"points_{i}"], **X86), z[f"sdf_{i}"]).all()
        assert same_bits(O.cast_ray(sc, z[f"rays_{i}"], float(z["steps"]), **X86), z[f"end_{i}"]).all()
    z = load("random_kinds")
    for i in range(int(z["count"])):
        sc = GC.random_kind_case(z, i)
        assert same_bits(O.eval_sdf(sc, z[f"points_{i}"], **X86), z[f"sdf_{i}"]).all(), f"kind {i}"
        assert same_bits(O.cast_ray(sc, z[f"rays_{i}"], float(z["steps"]), **X86), z[f"end_{i}"]).all(), f"kind {i}"
    z = load("random_images")
    for i in range(int(z["count"])):
        sc, schema, noises = GC.random_image_case(z, i)
        _planes_equal(z, _render(sc, schema, noises), f"_{i}")
    z = load("random_jobs")
    for i in range(int(z["count"])):
        sc, schema, noises = GC.random_job_case(z, i)
        full = schema["render"]["renderMode"] == "full"
        _planes_equal(z, _render(sc, schema, noises), f"_{i}", full)
        if f"color_native_{i}" in z:
            O.set_tan_mode(O.TAN_SWIFTSHADER)
            fr = _render(sc, schema, noises)
            O.set_tan_mode(O.TAN_PORTABLE)
            for name in ("color", "normal_dof", "albedo_depth") if full else ("color",):
                assert same_bits(z[f"{name}_native_{i}"], getattr(fr, name)).all(), f"job {i} from the unmodified text: {name}"


def test_present_pass_byte_for_byte():
    """The reference's present pass (display.frag under the same GL stack, read back as the RGBA8 canvas) on the two
    display goldens and on the planes of the 24 random jobs -- depth-of-field blurs of up to 33 x 33 taps, non-finite
    colours (presented white by this stack) included: every byte."""
    for name in ("display_dof", "display_nodof"):
        z = load(name)
        assert np.array_equal(O.present(z["color"], z["normal_dof"], int(z["samples"])), z["rgba8"])
    z = load("random_jobs")
    for i in range(int(z["count"])):
        color, ndof, n, ref, _ = GC.random_job_present_case(z, i)
        assert np.array_equal(O.present(color, ndof, n), ref), f"job {i}"


def test_helper_functions_bit_for_bit():
    """schlick, invExpDist, rodrigues, the material functions, the random stream: the harness goldens of the reference's
    own functions."""
    z = load("misc_schlick")
    a = z["inputs"].astype(np.float32)
    f = np.float32
    q = ((a[:, 1] - a[:, 2]) / (a[:, 1] + a[:, 2])).astype(f)
    r0 = O.ss_math("pow", np.abs(q), np.full_like(q, 2.0))
    p5 = O.ss_math("pow", np.abs((f(1) - a[:, 0]).astype(f)), np.full_like(q, 5.0))
    assert same_bits((r0 + ((f(1) - r0).astype(f) * p5).astype(f)).astype(f), z["out"][:, 0]).all()
    assert same_bits((-O.ss_math("log", (f(1) - a[:, 0]).astype(f)) / a[:, 3]).astype(f), z["out"][:, 1]).all()
    for name in GC.MATERIAL_SCENES:
        z = load("misc_material_" + name)
        got = O.material(GC.build_scene(name), z["points"], **X86)
        assert same_bits(got[:, 0:3], z["diffuse"]).all() and same_bits(got[:, 3:6], z["specular"]).all()
        assert same_bits(got[:, 6:9], z["emission"]).all()
    r = load("rng_32x32")
    u = J.uniforms_from_schema(J.make_schema(GC.build_scene("sphere"), 32, 32), tuple(r["rand_noise"]))
    assert same_bits(O.rng(u, 32, 32, 4), r["uniform4"]).all()


@pytest.mark.parametrize("text", TEXTS)
@pytest.mark.parametrize("name", list(GC.CONFIGS))
def test_baseline_configurations_bit_for_bit(name, text):
    """BASELINE.json's configurations with their own scenes, step counts, lights and cameras -- the headline C3b (Mandelbulb,
    full mode, 256 steps, the point light; 2 samples), C3a, C2 (preview and lit), C4 (CSG-64, 128 steps), C5 (three bounces,
    the soft light) -- through the reference's main() at 256 x 128 / 128 x 128: every value of every plane.  Twice: the goldens
    rendered with the portable tangent injected, and the ones rendered from raymarcher.frag as it stands (its own tan() in
    gold_noise :46-49 and the camera :186)."""
    sc, schema, noises = GC.config_case(name)
    z = _golden_of_text("config_", name, text)
    r = schema["render"]
    _planes_equal(z, _render(sc, schema, noises, r["width"], r["height"], threads=min(8, O.host_cores())), full="normal_dof" in z)


@pytest.mark.parametrize("text", TEXTS)
@pytest.mark.parametrize("name", list(GC.ROW_CHECKSUM_CASES))
def test_megapixel_configurations_row_checksums(name, text):
    """BASELINE configurations at megapixel size -- the headline C3b (Mandelbulb, full, 256 steps, the light) at 2048 x 1024
    and 4096 x 2048, C3a and C2 at 2048 x 1024, C4 (CSG-64, 128 steps) and C5 (three bounces, the soft light) at 1024 x 1024 --
    rendered by the reference's GLSL under software GL (both texts, as above); the fixture holds a CRC-32 per image row.  Here:
    24 rows spread over the frame (the colour plane; the GPU test checks every row of every plane), each row's checksum."""
    sc, schema, noises = GC.row_checksum_case(name)
    z = _golden_of_text("rows_", name, text)
    w, h = schema["render"]["width"], schema["render"]["height"]
    rows = list(range(h // 48, h, h // 24))
    _, got = O.render_rows(sc, J.uniforms_from_schema(schema, noises[0]), w, h, rows, threads=min(8, O.host_cores()), **X86)
    crc = GC.row_checksums(got)
    assert (crc == z["color"][rows]).all(), f"rows {[r for r, a, b in zip(rows, crc, z['color'][rows]) if a != b]} differ"
    assert 0.0 < float(z["finite_share"].mean()) <= 1.0
