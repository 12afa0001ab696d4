"""Run the reference's own GLSL under software GL (Kaleido's HeadlessChrome + SwiftShader).

TEST INFRASTRUCTURE ONLY -- used to pin ``oracle/rm_oracle.c`` and to generate
the golden vectors under ``tests/golden``.  It works only in the build
container: it reads the reference's shaders from ``/root/reference`` at run
time (nothing of them is stored in this repository) and needs the ``kaleido``
wheel.  Nothing on the product path, in ``bench.py`` or in the ``-m gpu``
tests imports it.

What it reproduces of the reference's host side (restated, not copied):
  * the scene splice at the ``//SCENESDFHERE`` marker and the default
    material functions appended when the scene text does not define them
    (client/src/renderer/RenderJobExecutor.tsx:121-127,
    client/src/settings/shader-editor/Validate.tsx:8-57 -- the reference asks
    a GLSL parser whether the function exists; a name regex is enough here);
  * the uniform derivations of RenderJobExecutor.tsx:212-297.
"""
from __future__ import annotations

import base64
import json
import os
import re
from pathlib import Path

import numpy as np

REFERENCE_ROOT = Path(os.environ.get("RM_REFERENCE_ROOT", "/root/reference"))
_SHADER_DIR = REFERENCE_ROOT / "client" / "public" / "shader"
_RUNNER_JS = Path(__file__).with_name("gl_runner.js")

_scope = None


def available() -> bool:
    try:
        import kaleido  # noqa: F401
    except Exception:
        return False
    return (_SHADER_DIR / "raymarcher.frag").exists()


def _get_scope():
    global _scope
    if _scope is None:
        from kaleido.scopes.plotly import PlotlyScope

        _scope = PlotlyScope(plotlyjs=_RUNNER_JS.as_uri())
    return _scope


def reference_text(name: str) -> str:
    return (_SHADER_DIR / name).read_text()


def example_scene_text(name: str) -> str:
    """Scene text of one of the reference's example scenes (read at run time)."""
    for sub in ("public/examples", "dist/examples"):
        p = REFERENCE_ROOT / "client" / sub / name
        if p.exists():
            return p.read_text()
    raise FileNotFoundError(name)


# default material functions: semantics of Validate.tsx:18-51, text generated
# here from the numbers (the same numbers live in the scene material block of
# the product, include/hip_raymarch.h RmMaterial).
def _default_functions() -> dict:
    return {
        "sceneDiffuseColor": "vec3 sceneDiffuseColor(vec3 p){ if (length(p) > 35.0) return vec3(0.0); return vec3(0.6); }",
        "sceneSpecularColor": "vec3 sceneSpecularColor(vec3 p){ if (length(p) > 35.0) return vec3(0.0); return vec3(0.6); }",
        "sceneSpecularRoughness": "float sceneSpecularRoughness(vec3 p){ return 0.2; }",
        "sceneSubsurfaceScattering": "float sceneSubsurfaceScattering(vec3 p){ return 11111115.0; }",
        "sceneSubsurfaceScatteringColor": "vec3 sceneSubsurfaceScatteringColor(vec3 p){ return vec3(1.0); }",
        "sceneIOR": "float sceneIOR(vec3 p){ return 100.0; }",
        "sceneEmission": (
            "vec3 sceneEmission(vec3 p){ float d = max(normalize(p).y, 0.2);"
            " vec3 b = vec3(0.7, 0.8, 1.0) * d * 1.0;"
            " return (length(p) > 36.0) ? (b * 2.00) : vec3(0.0); }"
        ),
    }


def add_default_functions(scene_src: str) -> str:
    out = scene_src
    for name, text in _default_functions().items():
        if not re.search(r"\b(?:float|vec3)\s+" + name + r"\s*\(", scene_src):
            out += "\n" + text
    return out


def splice(scene_src: str, harness_main: str | None = None) -> str:
    frag = reference_text("raymarcher.frag")
    assert "//SCENESDFHERE" in frag
    frag = frag.replace("//SCENESDFHERE", add_default_functions(scene_src))
    if harness_main is not None:
        # keep every function of the reference, but let a harness drive them
        assert "void main(void)" in frag
        frag = frag.replace("void main(void)", "void reference_main(void)")
        frag += "\n" + harness_main + "\n"
    return frag


# Portable tangent.  The reference's RNG is fract(tan(large)*x) (raymarcher.frag:46-49)
# and GL implementations disagree on tan of arguments of hundreds of radians
# (SwiftShader vs libm: 13 % of samples agree to 1e-3), so
# no two platforms share a random stream.  For whole-image goldens the
# reference's text is kept but its `tan` calls are routed, by a function-like
# macro, to this fixed sequence of IEEE add/mul/div/floor operations, which
# SwiftShader, C (-ffp-contract=off) and gfx950 (__fmul_rn/__fadd_rn) all
# evaluate to the same bits.  It IS a tangent: 3-term Cody-Waite reduction by
# pi/2 and minimax sin/cos polynomials, relative error <= 1.5e-7 on [0, 870].
PORTABLE_TAN_GLSL = """
float rm_tan(float x) {
  float k = floor(x * 0.636619772 + 0.5);
  float r = x - k * 1.5703125;
  r = r - k * 4.83751296997e-4;
  r = r - k * 7.54978995489e-8;
  float r2 = r * r;
  float s = r2 * -1.9515295891e-4 + 8.3321608736e-3;
  s = s * r2 + -1.6666654611e-1;
  s = s * r2 * r + r;
  float c = r2 * 2.443315711809948e-5 + -1.388731625493765e-3;
  c = c * r2 + 4.166664568298827e-2;
  c = c * r2 * r2 + (1.0 - 0.5 * r2);
  float odd = k - 2.0 * floor(k * 0.5);
  return (odd > 0.5) ? (-c / s) : (s / c);
}
#define tan(x) rm_tan(x)
"""


def with_portable_tan(frag: str) -> str:
    assert "precision highp float;" in frag
    return frag.replace("precision highp float;", "precision highp float;\n" + PORTABLE_TAN_GLSL, 1)


def u_float(*xs):
    return {"type": "f", "count": len(xs), "data": [float(x) for x in xs]}


def u_int(x):
    return {"type": "i", "count": 1, "data": [int(x)]}


def u_floats(name_count, xs):
    return {"type": "f", "count": name_count, "data": [float(x) for x in xs]}


def uniforms_from_schema(schema: dict, rand_noise) -> dict:
    """Uniform set for one sample, derived as RenderJobExecutor.tsx:212-297 derives it."""
    cam = schema["camera"]
    mode = cam["mode"]
    r = schema["render"]
    counts = list(schema["reflectionIterationCounts"])
    mode_index = ["perspective", "orthographic", "panoramic"].index(mode["type"])
    fov = mode["fov"] if mode["type"] == "perspective" else mode["size"] if mode["type"] == "orthographic" else 1.0
    u = {
        "blendWithPreviousFactor": u_float(r["blendWithPreviousFrameFactor"]),
        "randNoise": u_float(rand_noise[0], rand_noise[1]),
        "position": u_float(*cam["position"]),
        "dofAmount": u_float(schema["dof"]["amount"]),
        "dofFocalPlaneDistance": u_float(schema["dof"]["distance"]),
        "cameraMode": u_int(mode_index),
        "fov": u_float(fov),
        "reflections": u_float(len(counts)),
        "raymarchingSteps": u_float(counts[0]),
        "indirectLightingRaymarchingSteps": u_float(counts[1] if len(counts) > 1 else counts[0]),
        "aspect": u_float(r["width"] / r["height"]),
        "fogDensity": u_float(schema["fogDensity"]),
        "exposure": u_float(r["exposure"] / r["samplesPerPixel"]),
        "blendMode": u_int(1 if r["blendMode"] == "additive" else 0),
        "renderMode": u_int(1 if r["renderMode"] == "preview" else 0),
        "lightCount": u_int(len(schema["lights"])),
        "showDofFocalPlane": u_int(1 if schema["dof"]["showFocusedArea"] else 0),
        "raymarchingStepCountsArray": u_floats(1, counts),
        "rotation": {"kind": "mat4", "data": [float(x) for x in cam["rotation"]]},
    }
    lights = schema["lights"]
    if lights:
        pos, col, size = [], [], []
        for l in lights:
            pos += list(l["position"] if l["type"] == "point" else l["direction"])
            col += list(l["color"])
            size.append(l["size"] if l["type"] == "point" else 0.0)
        u["lightPositions"] = u_floats(3, pos)
        u["lightColors"] = u_floats(3, col)
        u["lightSizes"] = u_floats(1, size)
    for name, val in schema.get("customShaderParameters", {}).items():
        u[name] = {"type": val["type"], "count": val["count"], "data": list(val["data"])}
    return u


def run_gl(frag: str, width: int, height: int, uniforms: dict, draws=None, read=(0,), time=False, init_prev0=None, display_brightness=None) -> dict:
    job = {
        "vert": reference_text("raymarcher.vert"),
        "frag": frag,
        "width": int(width),
        "height": int(height),
        "uniforms": uniforms,
        "draws": draws if draws is not None else [{}],
        "read": list(read),
        "time": bool(time),
    }
    if display_brightness is not None:  # also run the reference's present pass (display.vert/.frag)
        job["display"] = {"vert": reference_text("display.vert"), "frag": reference_text("display.frag"), "brightness": float(display_brightness)}
    if init_prev0 is not None:
        a = np.ascontiguousarray(init_prev0, np.float32)
        assert a.shape == (height, width, 4)
        job["init_prev0"] = base64.b64encode(a.tobytes()).decode()
    raw = _get_scope().transform({"data": [], "layout": {"oracle": job}}, format="json")
    res = json.loads(raw.decode() if isinstance(raw, (bytes, bytearray)) else raw)
    if isinstance(res, str):
        res = json.loads(res)
    if not res.get("ok"):
        raise RuntimeError("GL oracle failed: " + str(res.get("error")))
    planes = {}
    for key, b in res["planes"].items():
        a = np.frombuffer(base64.b64decode(b), dtype=np.float32).reshape(height, width, 4)
        planes[int(key[len("plane"):])] = a.copy()
    res["planes"] = planes
    if res.get("display"):
        res["display"] = np.frombuffer(base64.b64decode(res["display"]), dtype=np.uint8).reshape(height, width, 4).copy()
    return res


def halton(base: int):
    """Radical-inverse sequence; same values as client/src/util/Halton.tsx:1-19 (0.5, 0.25, 0.75 ...)."""
    i = 0
    while True:
        i += 1
        num, den, k = 0, 1, i
        while k > 0:
            num = num * base + (k % base)
            den *= base
            k //= base
        yield num / den  # one correctly rounded division, like the reference's n / d
