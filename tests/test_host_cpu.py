"""CPU-side checks: host logic of the render-job boundary, the composer's two
back ends, and that the C-ABI library loads and exports every symbol
include/hip_raymarch.h declares (no compute calls without a GPU)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import golden_cases as GC
from oracle import oracle as O
from raymarching_engine_amd import abi
from raymarching_engine_amd import job as J
from raymarching_engine_amd import scene as S

ROOT = Path(__file__).resolve().parents[1]


def reference_halton(b):
    """The generator algorithm of client/src/util/Halton.tsx:1-19, restated."""
    n, d = 0, 1
    while True:
        x = d - n
        if x == 1:
            n, d = 1, d * b
        else:
            y = d
            while x <= y:
                y /= b
            n = (b + 1) * y - x
        yield n / d


def test_halton_matches_reference_generator():
    for b in (2, 3):
        g, r = J.halton(b), reference_halton(b)
        assert [next(g) for _ in range(500)] == [next(r) for _ in range(500)]
    g2, g3 = J.halton(2), J.halton(3)
    assert [next(g2) for _ in range(3)] == [0.5, 0.25, 0.75]
    assert np.allclose([next(g3) for _ in range(3)], [1 / 3, 2 / 3, 1 / 9])


def test_header_symbols_are_exported():
    from raymarching_engine_amd import native

    header = (ROOT / "include" / "hip_raymarch.h").read_text()
    declared = set(re.findall(r"^RM_API (?:int|void|void\*|const char\*)\s+(rm_[a-z_0-9]+)\(", header, re.M))
    assert declared == set(native.EXPORTS)
    lib = native.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rm_abi_version() == abi.RM_ABI_VERSION
    # ... and nothing else: the library is built -fvisibility=hidden, so its cross-unit launchers (rm_gl_launch_*, rm::launch_*) and
    # kernel stubs stay inside (`nm -D` lists the header's entry points and no other function or object of its own)
    import subprocess

    for path in (native.LIB_PATH, native.XCHECK_LIB_PATH):
        if not path.exists():
            continue
        out = subprocess.run(["nm", "-D", "--defined-only", str(path)], capture_output=True, text=True, check=True).stdout
        exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-2] in "TDBVWR"}
        assert exported == declared, (path.name, sorted(exported ^ declared)[:8])


def test_struct_layouts_match_header():
    from raymarching_engine_amd import native

    import subprocess
    import tempfile

    names = ["RmUniforms", "RmPrim", "RmMaterial", "RmSceneDesc", "RmRect"]
    src = '#include <stdio.h>\n#include "hip_raymarch.h"\nint main(void){ printf("' + " ".join(["%zu"] * len(names)) + '\\n", ' + ", ".join(f"sizeof({n})" for n in names) + "); return 0; }\n"
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "s.c").write_text(src)
        subprocess.run(["gcc", "-I", str(ROOT / "include"), str(Path(d) / "s.c"), "-o", str(Path(d) / "s")], check=True)
        sizes = [int(x) for x in subprocess.run([str(Path(d) / "s")], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [C.sizeof(getattr(abi, n)) for n in names]
    assert C.sizeof(abi.RmPrim) == 32
    m = abi.RmMaterial()
    native.load_library().rm_material_default(C.byref(m))
    assert bytes(m) == bytes(S.Material().to_c()) == bytes(O.material_default())


def test_no_gpu_means_loud_failure():
    """The product has no CPU path: without a device, creating a context fails
    with RM_ERR_NO_DEVICE (skipped on a GPU box)."""
    from raymarching_engine_amd import native

    try:
        ctx = native.Context(0)
    except native.RmError as e:
        assert e.code == abi.RM_ERR_NO_DEVICE and "no CPU fallback" in str(e)
    else:
        ctx.close()
        pytest.skip("a GPU is present")


def test_product_does_not_import_the_oracle():
    pkg = ROOT / "raymarching-engine_amd"
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.hpp")) + list(pkg.rglob("*.inc")) + list(pkg.rglob("*.cpp")) + list(pkg.rglob("*.js")):
        text = f.read_text()
        assert "import oracle" not in text and "from oracle" not in text and "rm_oracle" not in text.replace("oracle/rm_oracle.c", ""), f


def test_uniform_derivations():
    """RenderJobExecutor.tsx:212-297."""
    sc = GC.build_scene("sphere")
    schema = J.make_schema(sc, 1920, 1080, counts=(128, 64, 32), render_mode="full", samples_per_pixel=4, exposure=0.5,
                           lights=[J.point_light((1, 2, 3)), {"type": "sun", "direction": [0, 1, 0], "color": [1, 1, 1]}])
    u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
    assert u.reflections == 3.0 and list(u.raymarchingStepCountsArray)[:3] == [128.0, 64.0, 32.0]
    assert u.raymarchingSteps == 128.0 and u.indirectLightingRaymarchingSteps == 64.0
    assert u.aspect == np.float32(1920 / 1080) and u.exposure == 0.125
    assert u.blendMode == 1 and u.renderMode == 0 and u.cameraMode == 0 and u.lightCount == 2
    assert list(u.lightPositions[1]) == [0.0, 1.0, 0.0] and u.lightSizes[1] == 0.0
    assert np.allclose(list(u.lightColors[0]), [255 * 3 / 256] * 3)
    o = J.uniforms_from_schema(J.make_schema(sc, camera="orthographic", fov=2.5), (0, 0))
    assert o.cameraMode == 1 and o.fov == 2.5
    p = J.uniforms_from_schema(J.make_schema(sc, camera="panoramic"), (0, 0))
    assert p.cameraMode == 2 and p.fov == 1.0
    with pytest.raises(ValueError):
        J.uniforms_from_schema(J.make_schema(sc, counts=range(11)), (0, 0))


def test_tile_rects_cover_the_image_once():
    schema = J.make_schema(GC.build_scene("sphere"), 101, 67, subdivisions=3)
    cover = np.zeros((67, 101), int)
    for y in range(3):
        for x in range(3):
            t = J.tile_rect(schema, x, y)
            cover[t.y : t.y + t.h, t.x : t.x + t.w] += 1
    assert cover.min() == 1  # ceil/floor tiles may overlap by a pixel (as in the reference), never leave a gap


def test_composer_emits_consistent_back_ends():
    """One description, two back ends: the GLSL names the same constants the table holds."""
    sc = GC.build_scene("csg_mixed")
    d = sc.desc()
    text = sc.glsl()
    assert d.kind == abi.RM_SCENE_TABLE and d.nprims == 5
    assert text.count("sdBox(") == 2 and text.count("sdfSphere(") == 3 and "rmSmoothUnion(d," in text and "max(d, -" in text
    for n in sc._nodes:
        for v in n.center:
            assert S._f(v) in text
    assert "sceneDiffuseColor" not in text  # defaults are appended by the reference itself
    assert "sceneEmission" in GC.build_scene("lattice").glsl()
    assert S._f(0.1) == "0.10000000149011612" and S._f(2.0) == "2.0" and S._f(1e-7).startswith("1.00000001168")
    b = S.Mandelbulb(power=8, iterations=8, bailout=2.0)
    assert list(b.desc().params)[:3] == [8.0, 8.0, 2.0] and "pow(r, 8.0)" in b.glsl()


def test_oracle_window_and_tile_equal_full_render():
    """Row windows and tiles use global coordinates (what a GPU of a row-sharded run holds)."""
    sc = GC.build_scene("csg_mixed")
    schema = J.make_schema(sc, 64, 48, render_mode="full", counts=(24, 12), position=(0.3, 0.2, -4.0), lights=GC.LIGHT)
    u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
    full = O.Frame(64, 48)
    O.render(sc, u, full, threads=2)
    win = O.Frame(64, 48, 16, 20)
    O.render(sc, u, win)
    assert np.array_equal(win.color, full.color[16:36], equal_nan=True)
    tiled = O.Frame(64, 48)
    for t in ((0, 0, 40, 30), (40, 0, 24, 30), (0, 30, 64, 18)):
        O.render(sc, u, tiled, tile=t)
    assert np.array_equal(tiled.color, full.color, equal_nan=True)


def test_oracle_flop_counter_matches_survey_order_of_magnitude():
    """Instrumented algorithmic flops per pixel-sample: the
    Mandelbulb headline config is ~1e5 flops per pixel."""
    sc = S.Mandelbulb()
    schema = J.make_schema(sc, 32, 16, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
    fr = O.Frame(32, 16)
    flops = O.render(sc, J.uniforms_from_schema(schema, (0.5, 1 / 3)), fr, count_flops=True)
    per_px = flops / (32 * 16)
    assert 5e3 < per_px < 3e5


def test_png_capture_round_trip_and_container():
    """capture.encode_png (the canvas.toDataURL of index.tsx:470-476): valid container
    (signature, IHDR, CRCs), rows flipped from GL order to top-down, lossless."""
    import base64
    import struct
    import zlib

    from raymarching_engine_amd import capture

    rng = np.random.default_rng(7)
    for h, w in ((1, 1), (3, 5), (32, 64), (17, 301)):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        png = capture.encode_png(img)
        assert png[:8] == b"\x89PNG\r\n\x1a\n" and png[12:16] == b"IHDR" and png[-8:-4] == b"IEND"
        assert struct.unpack(">IIBBBBB", png[16:29]) == (w, h, 8, 6, 0, 0, 0)
        assert struct.unpack(">I", png[29:33])[0] == zlib.crc32(png[12:29]) & 0xFFFFFFFF
        back = capture.decode_png(png)
        assert (back == img[::-1]).all()  # PNG row 0 is the TOP row of the GL image
        assert (capture.decode_png(capture.encode_png(img, bottom_up=False)) == img).all()
        url = capture.to_data_url(img)
        assert url.startswith("data:image/png;base64,") and base64.b64decode(url.split(",", 1)[1]) == png
    with pytest.raises(ValueError):
        capture.encode_png(np.zeros((4, 4, 3), np.uint8))
    with pytest.raises(ValueError):
        capture.decode_png(b"not a png at all")
    bad = bytearray(capture.encode_png(np.zeros((2, 2, 4), np.uint8)))
    bad[20] ^= 1
    with pytest.raises(ValueError):
        capture.decode_png(bytes(bad))


def test_reference_scissor_option():
    """gl.scissor(x1, y1, x2, y2) read as (x, y, width, height) (RenderJobExecutor.tsx:182): the intended tile for
    1 and 2 subdivisions, rectangles that reach the far edges from 3 on."""
    for n in (1, 2):
        schema = J.make_schema(None, 100, 60, subdivisions=n)
        for y in range(n):
            for x in range(n):
                a, b = J.tile_rect(schema, x, y), J.tile_rect(schema, x, y, reference_scissor=True)
                assert (a.x, a.y, a.w, a.h) == (b.x, b.y, b.w, b.h)
    schema = J.make_schema(None, 90, 60, subdivisions=3)
    t = J.tile_rect(schema, 1, 1, reference_scissor=True)
    assert (t.x, t.y, t.w, t.h) == (30, 20, 60, 40)  # scissor(30, 20, 60, 40): to the right and top edges
    t = J.tile_rect(schema, 1, 1)
    assert (t.x, t.y, t.w, t.h) == (30, 20, 30, 20)


def test_portable_math_text_is_shared_by_oracle_and_kernels():
    """The transcendentals of the parity arithmetic exist twice -- oracle/pm_math.h for the checker,
    csrc/rm_pm_math.hpp for the HIP kernels (the product links nothing from oracle/) -- and must be the same
    sequence of operations: from the first definition on, the two files are the same text."""
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    a = (root / "oracle" / "pm_math.h").read_text()
    b = (root / "raymarching-engine_amd" / "csrc" / "rm_pm_math.hpp").read_text()
    mark = "#define PM_INF"
    assert mark in a and mark in b
    assert a[a.index(mark):] == b[b.index(mark):]
    assert "PM_FN" in a and "__device__" in b[:b.index(mark)]
    # binary32 sequences since round 5: no double-precision operation on either side
    body = a[a.index(mark):]
    assert "double" not in body and "PM_FMA(" not in body and "pm_log_hl" in body



def test_gl_stack_math_text_is_shared_by_oracle_and_kernels():
    """The GL stack's transcendentals exist twice as well -- oracle/ss_math.h (the oracle's OR_MATH_SWIFTSHADER mode, pinned
    by tests/golden/swiftshader_math.npz) and csrc/rm_ss_math.hpp (the kernels' rm_ctx_set_gl_stack arithmetic): the same
    text from the marker line on."""
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    a = (root / "oracle" / "ss_math.h").read_text()
    b = (root / "raymarching-engine_amd" / "csrc" / "rm_ss_math.hpp").read_text()
    mark = "/* ---- shared text:"
    assert mark in a and mark in b
    assert a[a.index(mark):] == b[b.index(mark):] and "SS_FN float ss_atan2" in a

def test_portable_math_accuracy_and_conventions():
    """oracle/pm_math.h -- fixed sequences of IEEE binary32 operations since round 5 -- against numpy's double-precision
    functions: within 2 ulp (atan2: 3) over the ranges the path uses, most results the correctly rounded float; C's conventions
    at the edges; the same bits with a hardware fused multiply-add and with libm's fmaf (the operation is IEEE's either way);
    denormal-free by definition (an argument or a result below 2^-126 counts as zero)."""
    import ctypes as C
    import subprocess
    import tempfile

    import numpy as np
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    src = r"""
#include <math.h>
#include <string.h>
#define PM_FN static inline
#define PM_FMAF(a, b, c) fmaf((a), (b), (c))
#define PM_SQRTF(x) sqrtf(x)
#define PM_DIV_ORDINARY(a, b) ((a) / (b))
static inline unsigned int PM_F2U(float x) { unsigned int u; memcpy(&u, &x, 4); return u; }
static inline float PM_U2F(unsigned int u) { float x; memcpy(&x, &u, 4); return x; }
#include "pm_math.h"
#define F1(n) void t_##n(const float* x, float* o, int c) { for (int i = 0; i < c; i++) o[i] = pm_##n(x[i]); }
F1(sin) F1(cos) F1(log) F1(exp) F1(acos)
void t_pow(const float* x, const float* y, float* o, int c) { for (int i = 0; i < c; i++) o[i] = pm_pow(x[i], y[i]); }
void t_atan2(const float* y, const float* x, float* o, int c) { for (int i = 0; i < c; i++) o[i] = pm_atan2(y[i], x[i]); }
void t_pow_pair(const float* x, const float* y, float* o, int c) {  /* two powers from one logarithm = two calls of pm_pow */
  for (int i = 0; i < c; i++) { float hi, lo; pm_log_hl(x[i], &hi, &lo); o[i] = pm_pow_from_log(x[i], y[i], hi, lo); }
}
"""
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "t.c").write_text(src)
        libs = []
        hw_fma = "fma" in Path("/proc/cpuinfo").read_text().split() if Path("/proc/cpuinfo").exists() else False
        for name, extra in (("soft", []),) + ((("hard", ["-mfma"]),) if hw_fma else ()):
            subprocess.run(["gcc", "-O2", "-ffp-contract=off", *extra, "-shared", "-fPIC", "-I", str(root / "oracle"), str(Path(d) / "t.c"), "-o",
                            str(Path(d) / f"{name}.so"), "-lm"], check=True)
            libs.append(C.CDLL(str(Path(d) / f"{name}.so")))

        def call(name, *args, lib=libs[-1]):
            args = [np.ascontiguousarray(a, np.float32) for a in args]
            out = np.empty_like(args[0])
            getattr(lib, name)(*[a.ctypes.data_as(C.c_void_p) for a in args], out.ctypes.data_as(C.c_void_p), C.c_int(len(out)))
            return out

        def ulps(got, want):  # error in units of the last place of the correctly rounded float; results below 2^-126 are zero by definition
            w32 = want.astype(np.float32)
            ok = np.isfinite(want) & (np.abs(want) >= 1.17549435e-38)
            return (np.abs(got.astype(np.float64) - want) / np.spacing(np.abs(w32)).astype(np.float64))[ok], np.mean(got[ok] == w32[ok])

        rng = np.random.default_rng(5)
        n = 400_000
        worst = {}

        def check(name, got, want, bar, exact_bar):
            e, exact = ulps(got, want)
            worst[name] = (float(e.max()), float(exact))
            assert e.max() <= bar and exact >= exact_bar, f"{name}: max error {e.max():.3f} ulp (bar {bar}), correctly rounded {exact:.4f} (bar {exact_bar})"

        for lim in (7.0, 60.0, 1e5):
            x = rng.uniform(-lim, lim, n).astype(np.float32)
            check(f"sin {lim:g}", call("t_sin", x), np.sin(x.astype(np.float64)), 2.0, 0.70)
            check(f"cos {lim:g}", call("t_cos", x), np.cos(x.astype(np.float64)), 2.0, 0.70)
        x = np.exp(rng.uniform(-87, 88, n)).astype(np.float32)
        check("log", call("t_log", x), np.log(x.astype(np.float64)), 1.0, 0.98)
        x = (1.0 - rng.uniform(0, 1, n)).astype(np.float32)  # the path's log(1 - u)
        x = x[x > 0]
        check("log(1 - u)", call("t_log", x), np.log(x.astype(np.float64)), 1.0, 0.98)
        x = rng.uniform(-87, 88.7, n).astype(np.float32)
        check("exp", call("t_exp", x), np.exp(x.astype(np.float64)), 1.5, 0.88)
        x = np.concatenate([rng.uniform(-1, 1, n), 1 - np.exp(rng.uniform(-16, 0, n)), np.exp(rng.uniform(-16, 0, n)) - 1]).astype(np.float32)
        check("acos", call("t_acos", x), np.arccos(x.astype(np.float64)), 2.0, 0.75)
        a, b = np.exp(rng.uniform(-3, 3, n)).astype(np.float32), rng.uniform(-10, 10, n).astype(np.float32)
        check("pow", call("t_pow", a, b), np.power(a.astype(np.float64), b.astype(np.float64)), 2.0, 0.85)
        a = rng.uniform(0, 2, n).astype(np.float32)  # the Mandelbulb's r^7 and r^8, schlick's fifth power
        for e in (5.0, 7.0, 8.0):
            check(f"pow {e:g}", call("t_pow", a, np.full(n, e, np.float32)), np.power(a.astype(np.float64), e), 2.0, 0.85)
        a, b = np.exp(rng.uniform(-20, 20, n)).astype(np.float32), rng.uniform(-4, 4, n).astype(np.float32)
        check("pow wide", call("t_pow", a, b), np.power(a.astype(np.float64), b.astype(np.float64)), 2.0, 0.85)
        assert (call("t_pow_pair", a, b) .view(np.uint32) == call("t_pow", a, b).view(np.uint32)).all()
        # large exponents multiply the logarithm's error: the pair (hi, lo) has to be normalised for this (before it was, y lo entered
        # the exponential's polynomial as part of its argument and pow(1.4, 60) was 1 400 ulp off, pow(1.37, -240) came out 0)
        for lo_e, hi_e, bar in ((15, 17, 2.0), (30, 34, 3.5), (60, 64, 6.0), (-64, -60, 6.0), (200, 250, 24.0), (-250, -200, 24.0)):
            a, b = rng.uniform(0.5, 2.0, n).astype(np.float32), rng.uniform(lo_e, hi_e, n).astype(np.float32)
            want = np.power(a.astype(np.float64), b.astype(np.float64))
            keep = (want > 1e-36) & (want < 1e36)
            check(f"pow exponents {lo_e}..{hi_e}", call("t_pow", a[keep], b[keep]), want[keep], bar, 0.2)
        # whole powers of small integers come out exact (the iterated kinds' per-level scale factors)
        assert (call("t_pow", [2, 2, 2, 3, 10, 0.5, 4, 3], [5, 10, -3, 4, 3, 7, 0.5, 0]) == np.array([32, 1024, 0.125, 81, 1000, 0.0078125, 2, 1], np.float32)).all()
        y, x = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
        check("atan2", call("t_atan2", y, x), np.arctan2(y.astype(np.float64), x.astype(np.float64)), 3.0, 0.6)
        y, x = [(rng.normal(size=n) * np.exp(rng.uniform(-30, 30, n))).astype(np.float32) for _ in range(2)]
        check("atan2 wide", call("t_atan2", y, x), np.arctan2(y.astype(np.float64), x.astype(np.float64)), 3.0, 0.6)
        print({k: (round(v[0], 3), round(v[1], 4)) for k, v in worst.items()})
        # a hardware fused multiply-add and libm's fmaf: the same bits (what lets the GPU and any host agree)
        if len(libs) == 2:
            for name, args in (("t_sin", (x,)), ("t_log", (np.abs(x),)), ("t_exp", (rng.uniform(-87, 88, n).astype(np.float32),)),
                               ("t_pow", (np.abs(y) + 0.1, rng.uniform(-3, 3, n).astype(np.float32))), ("t_atan2", (y, x)),
                               ("t_acos", (rng.uniform(-1, 1, n).astype(np.float32),))):
                g0, g1 = call(name, *args, lib=libs[0]), call(name, *args, lib=libs[1])
                assert ((g0.view(np.uint32) == g1.view(np.uint32)) | (np.isnan(g0) & np.isnan(g1))).all(), name
        # the conventions of C's functions at the edges
        same = lambda got, want: ((got == want.astype(np.float32)) | (np.isnan(got) & np.isnan(want))).all()
        inf, nan = np.float32(np.inf), np.float32(np.nan)
        edge = np.array([0.0, -0.0, 1.0, -1.0, inf, -inf, nan, 2.0, 0.5], np.float32)
        ys, xs = [v.ravel() for v in np.meshgrid(edge, edge)]
        got, want = call("t_atan2", ys, xs), np.arctan2(ys.astype(np.float64), xs.astype(np.float64))
        assert (np.isnan(got) == np.isnan(want)).all() and (np.signbit(got) == np.signbit(want))[~np.isnan(want)].all()
        assert ulps(got, want)[0].max() <= 1.0
        with np.errstate(all="ignore"):
            assert same(call("t_log", edge), np.log(edge.astype(np.float64)))
            got, want = call("t_acos", edge), np.arccos(edge.astype(np.float64))
            assert (np.isnan(got) == np.isnan(want)).all() and ulps(got, want)[0].max() <= 1.0 and got[2] == 0.0
            assert same(call("t_pow", np.abs(xs), ys), np.power(np.abs(xs).astype(np.float64), ys.astype(np.float64)))
        assert np.isnan(call("t_sin", np.array([inf, nan, 1e30, -inf], np.float32))).all() and np.isnan(call("t_cos", np.array([inf, nan, 1e30], np.float32))).all()
        big = np.exp(rng.uniform(0, 80, n)).astype(np.float32)  # any argument: a value in [-1, 1] or NaN, never anything else
        for f in ("t_sin", "t_cos"):
            v = call(f, big)
            assert (np.isnan(v) | (np.abs(v) <= 1.0)).all()
        assert (call("t_exp", [inf, -inf, 88.72, 88.73, -87.3, -87.4, -200, 0]) == np.array([inf, 0, 3.3931806e+38, inf, 1.2192433e-38, 0, 0, 1], np.float32)).all()
        assert np.isnan(call("t_exp", [nan]))[0]
        # denormal-free: arguments below 2^-126 are zeros, results below it are zeros
        assert (call("t_log", [1e-40, 1.17549435e-38]) == np.array([-inf, np.log(np.float64(np.float32(1.17549435e-38)))], np.float32)).all()
        assert (call("t_pow", [1e-40, 1e-20, 1e-20], [1.0, 2.5, -2.5]) == np.array([0.0, 0.0, np.inf], np.float32)).all()


def test_halton_equals_the_reference_generator():
    """job.halton against the first 2048 values of the reference's own generator (Halton.tsx:1-19, run under node by
    oracle/ts/gen_halton_golden.py) for bases 2, 3, 5 and 7: the same doubles, bit for bit."""
    import json
    import struct
    from pathlib import Path

    from raymarching_engine_amd import job as J

    fx = json.loads((Path(__file__).parent / "golden" / "halton_reference.json").read_text())["values"]
    for base, vals in fx.items():
        g = J.halton(int(base))
        got = [struct.pack(">d", next(g)).hex() for _ in vals]
        assert got == vals, f"base {base}: first difference at {next(i for i, (a, b) in enumerate(zip(got, vals)) if a != b)}"


def test_render_job_host_replays_the_reference_loop_call_by_call():
    """tests/golden/host_reference.json.gz holds what the reference's OWN render-job generator does
    (RenderJobExecutor.tsx:147-339, run under node against a recording WebGL mock by oracle/ts/gen_host_golden.py) for 40
    random RenderJobSchemas in a row: every present(n), yield, gl.scissor rectangle, uniform value of every draw, the
    fbo.delete and the return value.  job.do_render_job, driven with the same schemas (the Halton pair continuing across
    jobs, as in the page), produces the same events: same presents and yields at the same sample counts, the same
    rectangle (render.referenceScissor: the reference's gl.scissor arguments as GL reads them), the same uniform values
    to the last float32 bit."""
    import ctypes as C
    import gzip
    import json
    from pathlib import Path

    import numpy as np

    from raymarching_engine_amd import abi, job as J, scene as S

    fx = json.loads(gzip.open(Path(__file__).parent / "golden" / "host_reference.json.gz").read())

    class FakeFb:
        def __init__(self): self.cleared = 0
        def clear(self): self.cleared += 1
        def destroy(self): pass

    class FakeNative:
        def __init__(self, events): self.events = events
        def sync(self): pass
        def create_framebuffer(self, *a): return FakeFb()
        def render_sample(self, handle, fb, u, tile, flags):
            self.events.append(("draw", (tile.x, tile.y, tile.w, tile.h), bytes(u)))
        def render_samples(self, handle, fb, u, noises, tile, flags):
            for n in noises:
                v = abi.RmUniforms.from_buffer_copy(bytes(u))
                v.randNoise[0], v.randNoise[1] = n
                self.render_sample(handle, fb, v, tile, flags)

    class FakeContext(J.RenderJobContext):
        def __init__(self, events):
            self.native, self.flags, self.rows = FakeNative(events), 0, None
            self._scenes, self._live, self._purgatory = {}, {}, []
            self.events = events
        def get_scene(self, scene): return object()
        def fbo_delete(self, w, h, frameid):
            self.events.append(("fboDelete", (w, h, frameid)))
            super().fbo_delete(w, h, frameid)

    J.reset_halton()
    sc = S.single_sphere()
    for k, (schema, want) in enumerate(zip(fx["schemas"], fx["events"])):
        schema = dict(schema, sdfScene=sc, render=dict(schema["render"], referenceScissor=True))
        events = []
        ctx = FakeContext(events)
        gen = J.do_render_job(schema, ctx)(lambda s, c, fb, n: events.append(("present", n)))
        try:
            while True:
                next(gen)
                events.append(("yield", 1))
        except StopIteration as stop:
            events.append(("done", stop.value))
        compare_host_events(k, schema, events, want)


def test_pruned_flop_count_is_a_lower_count_of_the_same_image():
    """bench.py's frac_useful prices the arithmetic the marches NEED: the counting oracle with or_set_count_pruned stops counting a march
    at its bitwise fixed point or its certain escape (rm_oracle.c).  The image does not change, the count can only fall; and
    profiles/flops_per_pixel.json holds a pruned entry next to every workload's full one, row for row."""
    import json

    import numpy as np

    from raymarching_engine_amd import job as J, scene as S

    sc = S.Mandelbulb()
    schema = J.make_schema(sc, 48, 24, counts=(64,), render_mode="full", position=(0, 0, -2.5), lights=[J.point_light((2.0, 3.0, -4.0))])
    u = J.uniforms_from_schema(schema, (0.5, 1.0 / 3.0))
    rows = list(range(24))
    try:
        O.set_count_pruned(False)
        full, img_full = O.render_rows(sc, u, 48, 24, rows, threads=2, count_flops=True)
        O.set_count_pruned(True)
        pruned, img_pruned = O.render_rows(sc, u, 48, 24, rows, threads=2, count_flops=True)
    finally:
        O.set_count_pruned(False)
    assert 0 < pruned < full
    a, b = np.asarray(img_full), np.asarray(img_pruned)
    assert ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()
    d = json.loads((Path(__file__).resolve().parents[1] / "profiles" / "flops_per_pixel.json").read_text())
    for w in ("c2", "c3a", "c3b", "c4", "c5"):
        f, p = d[w], d[w + "_pruned"]
        assert (f["row_stride"], f["rows"], f["width"]) == (p["row_stride"], p["rows"], p["width"])
        assert all(pr <= fr for pr, fr in zip(p["flops_per_row"], f["flops_per_row"])) and p["flops_per_pixel_sample"] < f["flops_per_pixel_sample"]


def test_the_wavefront_pipeline_lives_in_the_cross_check_build_only():
    """Round 5: the second implementation of the per-pixel program (csrc/rm_wavefront.inc) is test infrastructure.  The product
    library holds none of its kernels; tests/_xcheck/libhip_raymarch_xcheck.so -- the same sources with -DRM_WITH_WAVEFRONT=1
    (build.py build_crosscheck) -- holds them and exports the same C ABI, so the GPU tests can hold the pixel kernel to it."""
    from raymarching_engine_amd import native

    product = native.LIB_PATH.read_bytes()
    assert product.count(b"wf_march") == 0 and product.count(b"wf_shade") == 0 and product.count(b"rm_pixel_kernel") > 0
    if not native.XCHECK_LIB_PATH.exists():
        pytest.skip("the cross-check build has not been made (python raymarching-engine_amd/build.py)")
    xcheck = native.XCHECK_LIB_PATH.read_bytes()
    assert xcheck.count(b"wf_march") > 0 and xcheck.count(b"rm_pixel_kernel") > 0
    lib = native.load_library(native.XCHECK_LIB_PATH)
    assert all(hasattr(lib, name) for name in native.EXPORTS) and lib.rm_abi_version() == abi.RM_ABI_VERSION


def test_a_suspended_job_keeps_its_scene_through_the_cache_eviction():
    """The scene cache of a context holds 64 scenes, least recently used out first (job.SCENE_CACHE_ENTRIES) -- but a job yields
    between samples, and the scenes other jobs bring in meanwhile must not destroy the one it still renders (round 4 did: the
    next render of the suspended job then passed a destroyed handle).  do_render_job pins its scene while its generator lives;
    the pin goes with the generator, finished or closed."""
    from raymarching_engine_amd import job as J, scene as S

    destroyed, drawn = [], []

    class Handle:
        def __init__(self, name): self.name = name
        def destroy(self): destroyed.append(self.name)

    class FakeFb:
        def destroy(self): pass

    class FakeNative:
        def create_scene(self, scene): return Handle(scene.radius)
        def create_framebuffer(self, w, h, *a): return FakeFb()
        def sync(self): pass
        def render_sample(self, handle, fb, u, tile, flags):
            assert handle.name not in destroyed, "a destroyed scene handle reached the library"
            drawn.append(handle.name)

    class Ctx(J.RenderJobContext):
        def __init__(self):
            self.native, self.flags, self.rows, self.stripes, self.group, self.stream = FakeNative(), 0, None, None, None, None
            from collections import OrderedDict
            self._scenes, self._pins, self._live, self._purgatory = OrderedDict(), {}, {}, []

    # distinct scenes: spheres of different radii (scene_key hashes the description)
    def sphere(radius):
        t = S.CsgScene(); t.sphere((0.0, 0.0, 0.0), float(radius)); t.radius = radius
        return t

    ctx = Ctx()
    schema = J.make_schema(sphere(1.0), 16, 16, counts=(8,), render_mode="preview", samples_per_pixel=3)
    gen = J.do_render_job(schema, ctx)(lambda *a: None)
    next(gen)  # the job's first present: the generator is suspended with its scene pinned
    for i in range(J.SCENE_CACHE_ENTRIES + 8):  # other jobs bring in more scenes than the cache holds
        ctx.get_scene(sphere(2.0 + i))
    assert 1.0 not in destroyed and len(destroyed) == 9 and len(ctx._scenes) == J.SCENE_CACHE_ENTRIES  # the oldest UNPINNED ones went
    result = J.drain(gen)
    assert result == {"success": True} and drawn == [1.0, 1.0, 1.0]
    assert not ctx._pins and 1.0 not in destroyed
    ctx.get_scene(sphere(999.0))  # unpinned, it is the oldest entry: the next scene pushes it out
    assert destroyed[-1] == 1.0 and len(ctx._scenes) == J.SCENE_CACHE_ENTRIES
    # a generator that is closed mid-job lets go of its pin too
    gen = J.do_render_job(J.make_schema(sphere(1.5), 16, 16, counts=(8,), render_mode="preview", samples_per_pixel=3), ctx)(lambda *a: None)
    next(gen)
    assert len(ctx._pins) == 1
    gen.close()
    assert not ctx._pins
    # a failing render is the job's value, not an exception out of the generator (errors are values: RenderJobExecutor.tsx:56-68)
    from raymarching_engine_amd import native as N

    def boom(*a): raise N.RmError(3, "device lost")
    ctx.native.render_sample = boom
    out = J.drain(J.do_render_job(J.make_schema(sphere(1.25), 16, 16, counts=(8,), render_mode="preview", samples_per_pixel=2), ctx)(lambda *a: None))
    assert out["success"] is False and "device lost" in out["why"]["infoLog"] and not ctx._pins
    assert not ctx._live  # the failed job's frame went back to the cache (fbo.delete), as it does on success
    # 64 suspended jobs pin 64 scenes: the 65th scene is the only unpinned entry -- it must not be the one evicted as it is handed out
    # (get_scene returned a destroyed handle); the cache exceeds its bound while everything in it is in use
    ctx.native.render_sample = FakeNative.render_sample.__get__(ctx.native)
    ctx2 = Ctx()
    destroyed.clear()
    gens = []
    for i in range(J.SCENE_CACHE_ENTRIES):
        g = J.do_render_job(J.make_schema(sphere(100.0 + i), 16, 16, counts=(8,), render_mode="preview", samples_per_pixel=2, frameid=i), ctx2)(lambda *a: None)
        next(g)
        gens.append(g)
    extra = ctx2.get_scene(sphere(500.0))
    assert extra.name == 500.0 and not destroyed and len(ctx2._scenes) == J.SCENE_CACHE_ENTRIES + 1
    g = J.do_render_job(J.make_schema(sphere(501.0), 16, 16, counts=(8,), render_mode="preview", samples_per_pixel=2, frameid=900), ctx2)(lambda *a: None)
    assert J.drain(g) == {"success": True} and 501.0 not in destroyed[:-1]
    for g in gens:
        g.close()
    assert not ctx2._pins


def compare_host_events(k, schema, events, want):
    """events: ("present", n) | ("yield", 1) | ("draw", (x, y, w, h), bytes of RmUniforms) | ("fboDelete", (w, h, id)) | ("done", value)
    against the reference's recorded events of the same job (tests/golden/host_reference.json.gz)."""
    import numpy as np

    from raymarching_engine_amd import abi

    f32 = lambda xs: np.array([float(x) if not isinstance(x, str) else float(x.replace("Infinity", "inf")) for x in xs], np.float32)
    want = [e for e in want if "blit" not in e]
    assert len(events) == len(want), f"job {k}: {len(events)} events against the reference's {len(want)}"
    W, H = schema["render"]["width"], schema["render"]["height"]
    for i, (got, ref) in enumerate(zip(events, want)):
        kind = next(iter(ref))
        assert got[0] == kind, f"job {k} event {i}: {got[0]} against {kind}"
        if kind == "present":
            assert got[1] == ref["present"], f"job {k} event {i}"
        elif kind == "fboDelete":
            assert list(got[1]) == ref["fboDelete"]
        elif kind == "done":
            assert got[1] == ref["done"]
        elif kind == "draw":
            x, y, w, h = ref["draw"]["scissor"]  # the reference's arguments as GL reads them: (x, y, width, height), clipped to the image
            assert tuple(got[1]) == (x, y, min(w, W - x), min(h, H - y)), f"job {k} event {i}: tile {got[1]} against scissor {ref['draw']['scissor']}"
            u = abi.RmUniforms.from_buffer_copy(got[2])
            ru = ref["draw"]["uniforms"]
            n_counts, n_lights = len(schema["reflectionIterationCounts"]), len(schema["lights"])
            mine = {"blendWithPreviousFactor": [u.blendWithPreviousFactor], "randNoise": list(u.randNoise), "position": list(u.position), "dofAmount": [u.dofAmount],
                    "dofFocalPlaneDistance": [u.dofFocalPlaneDistance], "cameraMode": [u.cameraMode], "fov": [u.fov], "reflections": [u.reflections],
                    "raymarchingSteps": [u.raymarchingSteps], "indirectLightingRaymarchingSteps": [u.indirectLightingRaymarchingSteps], "aspect": [u.aspect],
                    "fogDensity": [u.fogDensity], "exposure": [u.exposure], "blendMode": [u.blendMode], "renderMode": [u.renderMode], "lightCount": [u.lightCount],
                    "showDofFocalPlane": [u.showDofFocalPlane], "raymarchingStepCountsArray": list(u.raymarchingStepCountsArray)[:n_counts], "rotation": list(u.rotation)}
            if n_lights:
                mine["lightPositions"] = [c for j in range(n_lights) for c in u.lightPositions[j]]
                mine["lightColors"] = [c for j in range(n_lights) for c in u.lightColors[j]]
                mine["lightSizes"] = list(u.lightSizes)[:n_lights]
            for name, vals in mine.items():
                assert name in ru, f"job {k}: the reference does not set {name}"
                a, b = f32(vals), f32(ru[name])
                assert a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all(), f"job {k} event {i}: uniform {name}: {vals} against {ru[name]}"
            extra = set(ru) - set(mine) - {"previousColor", "previousNormalAndDofRadius", "previousAlbedoAndDepth"}
            # with no light the reference leaves the light arrays as an earlier job set them (lightCount = 0 makes them dead)
            assert extra <= ({"lightPositions", "lightColors", "lightSizes"} if not n_lights else set()), f"job {k}: uniforms of the reference not compared: {extra}"


def fbo_events_equal(results, got):
    """got[i] = (uid or None, [("created", uid) | ("cleared", uid) | ("destroyed", uid), ...]) per operation, against the fixture."""
    assert len(got) == len(results)
    for i, (ref, (uid, events)) in enumerate(zip(results, got)):
        assert uid == ref["result"], f"operation {i}: framebuffer set {uid} against the reference's {ref['result']}"
        want = [(k, e[k]) for e in ref["events"] for k in ("created", "cleared", "destroyed") if k in e]
        assert events == want, f"operation {i}: {events} against {want}"


def test_framebuffer_cache_replays_the_reference_cache_operation_by_operation():
    """tests/golden/fbo_reference.json: what the reference's context.fbo (LoadRenderJobContext.tsx:160-250, run under node by
    oracle/ts/gen_fbo_golden.py) does for 600 random create / delete operations over 12 keys: which framebuffer set comes
    back (fresh, still live, or out of the purgatory), when `prev` is cleared (a parked set taken up under a new
    frameid), which sets the purgatory evicts.  job.RenderJobContext.fbo_create / fbo_delete do the same, operation by
    operation."""
    import json
    from pathlib import Path

    from raymarching_engine_amd import job as J

    fx = json.loads((Path(__file__).parent / "golden" / "fbo_reference.json").read_text())
    log = []

    class FakeFb:
        serial = 0
        def __init__(self):
            FakeFb.serial += 1
            self.uid = FakeFb.serial
            log.append(("created", self.uid))
        def clear(self): log.append(("cleared", self.uid))
        def destroy(self): log.append(("destroyed", self.uid))

    class FakeNative:
        def create_framebuffer(self, *a): return FakeFb()

    class Ctx(J.RenderJobContext):
        def __init__(self):
            self.native, self.flags, self.rows = FakeNative(), 0, None
            self._scenes, self._live, self._purgatory = {}, {}, []

    c = Ctx()
    got = []
    for op, w, h, f in fx["ops"]:
        del log[:]
        uid = None
        if op == "create":
            uid = c.fbo_create(w, h, f).uid
        else:
            c.fbo_delete(w, h, f)
        got.append((uid, list(log)))
    fbo_events_equal(fx["results"], got)


def test_public_header_is_self_contained_c_and_cxx(tmp_path):
    """include/hip_raymarch.h compiles on its own as C99 (-pedantic) and as C++11: the boundary is a C ABI."""
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    src = tmp_path / "hdr.c"
    src.write_text('#include "include/hip_raymarch.h"\nint main(void) { return RM_ABI_VERSION > 0 ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", str(root), str(src)], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", "-I", str(root), str(src)], check=True)


def test_bench_counters_fall_back_to_the_replay_file_and_that_file_matches_the_sources(monkeypatch):
    """bench.py measures its hardware counters itself (live_counters: rocprofv3 --pmc child passes); where it cannot -- no rocprofv3, or the
    run is itself under a profiler -- it replays profiles/<round>_counters.json, which is only valid for the kernel sources it was measured on.
    Here, without a GPU: live_counters gives up with a reason when there is no rocprofv3; a run under a profiler is recognised; and the committed
    replay file carries every leg of the bench line and was measured on THESE sources (a kernel edit without `bench.py --dump-counters` fails here)."""
    import importlib.util
    import shutil

    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(shutil, "which", lambda name: None)
    monkeypatch.setattr(bench.os.path, "exists", lambda p: False if "rocprofv3" in str(p) else os.path.lexists(p))
    got, why = bench.live_counters()
    assert got is None and "rocprofv3" in why
    monkeypatch.undo()
    assert not bench.under_a_profiler() or os.environ.get("LD_PRELOAD", "").find("rocprof") >= 0 or os.environ.get("ROCP_TOOL_LIBRARIES")
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.under_a_profiler()
    monkeypatch.delenv("ROCP_TOOL_LIBRARIES")
    for key, build, gl_stack in bench.COUNTER_LEGS:
        ent = bench.counters_entry(key, build == "strict", "megakernel", False, gl_stack)
        assert ent is not None, f"profiles/<round>_counters.json has no entry for {key} {build} measured on the current kernel sources"
        assert ent["hbm_bytes_per_pixel"] > 25.0 and 0.4 < ent["lanes_active"] <= 1.0 and ent["sq_insts_valu_per_frame"] > 1e6
        acc = bench.issue_accounted(ent, 1.0)  # (per millisecond of kernel time: a number, not a claim)
        assert acc is not None and acc > 0.0
