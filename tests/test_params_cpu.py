"""Scene-parameter annotations (CustomShaderParamParser.tsx:8-209, CustomSettings.tsx:148-173)."""
import os

import pytest

from raymarching_engine_amd import abi, params as P, scene as S

SNIPPET = """
// a scene in the reference's annotated-uniform style (text authored for this test)
uniform float shellRadius;
//@name="Shell Radius"
//@min=0 @max=10 @step=0.001 @sensitivity=0.01 @default=2.5 @scale=log
//@tooltip="Radius of the shell."
uniform vec3 tint;
//@name="Tint" @default=0.25,0.5,0.75 @format=color/numerical
uniform int mode;
//@default=1 @format=checkbox

uniform float bare;
float sdf(vec3 p) { return length(p) - shellRadius; }
"""


def _norm(entry):
    """Non-finite numbers as JavaScript's String() writes them (the fixture is JSON), also inside lists."""
    import math

    def one(v):
        if isinstance(v, float) and math.isnan(v):
            return "NaN"
        if isinstance(v, float) and math.isinf(v):
            return "Infinity" if v > 0 else "-Infinity"
        if isinstance(v, list):
            return [one(x) for x in v]
        return v

    return {k: one(v) for k, v in entry.items()}


def _fixture():
    import json
    from pathlib import Path

    return json.loads((Path(__file__).parent / "golden" / "params_reference.json").read_text())


def test_scanner_equals_the_reference_on_synthetic_texts():
    """get_custom_shader_params against the outputs of the reference's own getCustomShaderParams
    (CustomShaderParamParser.tsx:8-209, run under node by oracle/ts/gen_params_golden.py): block comments, quoted
    values, every error entry with its span, non-float types, annotations before the first uniform, `uniform` that is
    not a declaration, two uniforms on a line, the empty text."""
    fx = _fixture()
    assert len(fx["synthetic_texts"]) >= 9
    for name, text in fx["synthetic_texts"].items():
        got = [_norm(e) for e in P.get_custom_shader_params(text)]
        assert got == fx["results"][name]["params"], name


@pytest.mark.skipif(not os.path.exists("/root/reference/client/public/examples"), reason="the reference's example texts are only in the build container (they are not copied into the repo)")
def test_scanner_equals_the_reference_on_its_example_scenes():
    from pathlib import Path

    fx = _fixture()
    seen = 0
    for key, res in fx["results"].items():
        if not key.startswith("example:"):
            continue
        for sub in ("public/examples", "dist/examples"):
            f = Path("/root/reference/client") / sub / key.split(":", 1)[1]
            if f.exists():
                text = f.read_text()
                assert len(text) == res["length"], key
                assert [_norm(e) for e in P.get_custom_shader_params(text)] == res["params"], key
                seen += 1
                break
    assert seen >= 6


def test_annotation_scanner():
    ps = P.get_custom_shader_params(SNIPPET)
    assert [p["internalName"] for p in ps] == ["shellRadius", "tint", "mode", "bare"]
    a, b, c, d = ps
    assert a["type"] == "f" and a["quantity"] == 1 and a["name"] == "Shell Radius" and a["defaultValue"] == [2.5]
    assert (a["min"], a["max"], a["step"], a["sensitivity"], a["scale"]) == (0.0, 10.0, 0.001, 0.01, "log")
    assert a["tooltip"] == "Radius of the shell."
    assert b["quantity"] == 3 and b["defaultValue"] == [0.25, 0.5, 0.75] and b["formats"] == ["color", "numerical"]
    assert c["type"] == "i" and c["defaultValue"] == [1.0] and c["formats"] == ["checkbox"]
    # no @default: the reference's table carries four zeros whatever the quantity (CustomShaderParamParser.tsx:32)
    assert d["name"] == "bare" and d["defaultValue"] == [0, 0, 0, 0] and d["formats"] == ["numerical"]
    vals = P.default_custom_shader_parameters(SNIPPET)
    assert vals["tint"] == {"type": "f", "count": 3, "data": [0.25, 0.5, 0.75]}
    assert vals["mode"] == {"type": "i", "count": 1, "data": [1]} and vals["bare"]["data"] == [0.0]


def test_annotation_errors_are_values():
    # unknown keys are ignored by the reference (its switch has no default); the two malformed values are reported,
    # BEFORE the parameter they belong to (a parameter is emitted when the next `uniform` or the end is reached)
    ps = P.get_custom_shader_params("uniform vec2 a;\n//@default=1 @bogus=3 @min=abc\n")
    assert [p["success"] for p in ps] == [False, False, True]
    assert all("start" in b and "end" in b and b["reason"] for b in ps[:2])
    assert ps[2]["internalName"] == "a" and ps[2]["defaultValue"] == [1]


def test_scene_from_example_text():
    tree = """uniform float fractalIterations;
//@default=6
uniform float scaleFactor;
//@default=0.7
uniform vec3 angles;
//@default=2.9,-0.8,0.4
uniform float offset;
//@default=1.2
float sdf(vec3 p) { float minDist = 9999.0; /* ... */ minDist = min(minDist, 1.0); return minDist; }"""
    sc = P.scene_from_example(tree)
    assert isinstance(sc, S.KifsTree) and sc.desc().kind == abi.RM_SCENE_KIFS_TREE
    assert sc.params()[:6] == [6.0, 0.7, 2.9, -0.8, 0.4, 1.2] and sc.params()[6] == 0.0
    over = P.scene_from_example(tree, {"scaleFactor": {"type": "f", "count": 1, "data": [0.5]}})
    assert over.params()[1] == 0.5
    with pytest.raises(ValueError):
        P.scene_from_example("float sdf(vec3 p) { return p.x; }")


@pytest.mark.skipif(not os.path.exists("/root/reference/client/public/examples"), reason="the reference is only in the build container")
def test_reference_examples_map_to_their_kinds():
    from pathlib import Path

    ex = Path("/root/reference/client/public/examples")
    want = {"fractal1.glsl": (S.SphereGridFractal, [4.0, 8.0]), "guide.glsl": (S.SphereGridFractal, [4.0, 8.0]),
            "menger-sponge.glsl": (S.MengerSponge, [8.0]), "tree.glsl": (S.KifsTree, [8.0, 0.7]),
            "smooth-tree.glsl": (S.KifsTree, [14.0, 0.7]), "rotation-fractal.glsl": (S.KifsBox, [14.0, 0.5])}
    for name, (cls, head) in want.items():
        sc = P.scene_from_example((ex / name).read_text())
        assert type(sc) is cls and sc.params()[: len(head)] == head, name
    assert P.scene_from_example((ex / "smooth-tree.glsl").read_text()).smoothen is True
    assert tuple(P.scene_from_example((ex / "guide.glsl").read_text()).material.diffuse) == (0.5, 0.5, 0.5)
