"""Scene-parameter annotations (next-row N3).

The reference's scenes declare their tunables as GLSL ``uniform``s followed by
``//@key=value`` annotations; ``getCustomShaderParams``
(client/src/settings/shader-editor/CustomShaderParamParser.tsx:8-209) scans the
text into a parameter table and ``CustomSettings.tsx:148-173`` turns the
``@default``s into the job's ``customShaderParameters``.  This module does the
same for the host side here, so that an example scene's text yields the
parameter values its scene kind needs (``scene_from_example``).

Grammar accepted (as the reference's scanner does): ``uniform <type> <name>;``
with type float | int | uint | vec2..4 | ivec2..4 | uvec2..4; then any number of
``//`` comment lines whose ``@key=value`` pairs are separated by whitespace;
values may be double-quoted; ``@default`` is a comma list; ``@format`` a
slash list.  A uniform without annotations is still a parameter.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional

from . import scene as S

_UNIFORM = re.compile(r"^\s*uniform\s+(u?int|float|[iu]?vec[234])\s+([A-Za-z_][A-Za-z_0-9]*)\s*;", re.M)
_PAIR = re.compile(r'@([A-Za-z]+)\s*=\s*("([^"]*)"|[^\s"]+)')
_KEYS = {"name", "min", "max", "step", "sensitivity", "scale", "default", "tooltip", "format"}


def _type_info(t: str):
    kind = "f"
    if t.startswith("i"):
        kind = "i"
    elif t.startswith("u") and t != "uint":
        kind = "ui"
    elif t == "uint":
        kind = "ui"
    elif t == "int":
        kind = "i"
    qty = int(t[-1]) if t[-1] in "234" else 1
    return kind, qty


def get_custom_shader_params(src: str) -> List[dict]:
    """The parameter table of a scene text: one dict per ``uniform`` (internalName, type, quantity,
    name, formats, min/max/step/sensitivity/scale/defaultValue/tooltip when annotated), or
    ``{"success": False, "reason", "start", "end"}`` entries for malformed annotations."""
    out: List[dict] = []
    matches = list(_UNIFORM.finditer(src))
    for n, m in enumerate(matches):
        kind, qty = _type_info(m.group(1))
        p = {"success": True, "type": kind, "quantity": qty, "internalName": m.group(2), "name": m.group(2), "formats": ["numerical"]}
        end = matches[n + 1].start() if n + 1 < len(matches) else len(src)
        # annotation comments directly following the declaration (stop at the first non-comment, non-blank line)
        pos = m.end()
        for line in src[m.end():end].splitlines(keepends=True):
            stripped = line.strip()
            if stripped and not stripped.startswith("//"):
                break
            body_at = pos + line.find("//") + 2 if "//" in line else pos
            if stripped.startswith("//@") or (stripped.startswith("//") and "@" in stripped and stripped[2:].lstrip().startswith("@")):
                for a in _PAIR.finditer(line):
                    key, val = a.group(1), a.group(3) if a.group(3) is not None else a.group(2)
                    if key not in _KEYS:
                        out.append({"success": False, "reason": f"unknown annotation @{key}", "start": pos + a.start(), "end": pos + a.end()})
                        continue
                    try:
                        if key in ("min", "max", "step", "sensitivity"):
                            p[key] = float(val)
                        elif key == "default":
                            vals = [float(x) for x in val.split(",") if x != ""]
                            if len(vals) != qty:
                                raise ValueError(f"@default needs {qty} value(s)")
                            p["defaultValue"] = vals
                        elif key == "format":
                            p["formats"] = val.split("/")
                        elif key == "scale":
                            p["scale"] = val
                        else:
                            p[key] = val
                    except ValueError as e:
                        out.append({"success": False, "reason": str(e), "start": pos + a.start(), "end": pos + a.end()})
            pos += len(line)
            del body_at
        out.append(p)
    return out


def default_custom_shader_parameters(src: str) -> Dict[str, dict]:
    """``customShaderParameters`` from the ``@default``s (CustomSettings.tsx:148-173): missing defaults are zeros."""
    res = {}
    for p in get_custom_shader_params(src):
        if not p.get("success"):
            continue
        data = p.get("defaultValue", [0.0] * p["quantity"])
        if p["type"] != "f":
            data = [int(v) for v in data]
        res[p["internalName"]] = {"type": p["type"], "count": p["quantity"], "data": data}
    return res


def scene_from_example(src: str, overrides: Optional[Dict[str, dict]] = None) -> S.Scene:
    """The scene kind that restates one of the reference's example scenes, with the parameter
    values its text declares (``@default``s, optionally overridden by ``customShaderParameters``).
    The example is recognised by its uniform names; other text raises ValueError (a HIP kernel
    cannot take arbitrary GLSL)."""
    vals = default_custom_shader_parameters(src)
    if overrides:
        vals.update(overrides)
    names = set(vals)

    def f(name, default=None):
        if name not in vals:
            if default is None:
                raise ValueError(f"scene text lacks uniform {name}")
            return default
        d = vals[name]["data"]
        return float(d[0]) if len(d) == 1 else tuple(float(x) for x in d)

    if {"bigSphereSize", "fractalIterations", "gridScaleFactor", "bigSphereCenter"} <= names:
        mat = S.Material(diffuse=f("fractalColor"), ) if "fractalColor" in names else None
        return S.SphereGridFractal(f("bigSphereSize"), f("fractalIterations"), f("gridScaleFactor"), f("bigSphereCenter"), material=mat)
    if {"fractalIterations", "scaleFactor", "angles", "offset"} <= names:
        if "min(minDist" in src or "generalUnion(" in src:  # tree.glsl / smooth-tree.glsl fold a box per level into minDist
            return S.KifsTree(f("fractalIterations"), f("scaleFactor"), f("angles"), f("offset"), smoothen=int(f("smoothen", 0.0)) == 1)
        return S.KifsBox(f("fractalIterations"), f("scaleFactor"), f("angles"), f("offset"))
    if names == {"fractalIterations"}:
        return S.MengerSponge(f("fractalIterations"))
    if not names and "sd_sphere(repeat" in src:
        return S.sphere_lattice_example()
    raise ValueError("unrecognised scene text: compose the scene with raymarching_engine_amd.scene instead")
