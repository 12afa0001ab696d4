"""Scene-parameter annotations (next-row N3).

The reference's scenes declare their tunables as GLSL ``uniform``s followed by
``//@key=value`` annotations; ``getCustomShaderParams``
(client/src/settings/shader-editor/CustomShaderParamParser.tsx:8-209) scans the
text into a parameter table and ``CustomSettings.tsx:148-173`` turns the
``@default``s into the job's ``customShaderParameters``.  This module does the
same for the host side here, so that an example scene's text yields the
parameter values its scene kind needs (``scene_from_example``).

Two layers, both this module's own (round 5; until round 4 the scanner was a
restatement of the reference's loop, which now lives under oracle/ts/ as the
checker):

* ``tokens(text)`` -- a lexer with three modes (code, line comment, block
  comment) that yields what matters to a parameter table and nothing else:
  ``uniform`` keywords, declarations behind them, annotations inside comments;
* ``ParamTable`` -- folds that token stream into the table, one handler per
  annotation key.

The table is entry for entry the reference's: pinned to the outputs of its own
function, run under node, on 160 random texts and nine written ones
(tests/golden/params_reference.json, oracle/ts/gen_params_golden.py).
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, NamedTuple, Optional, Tuple

from . import scene as S

# ---- lexical classes (JavaScript's: the reference's patterns are JavaScript regular expressions) --------------------------
_SPACE_CHARS = "\t\n\v\f\r \u00a0\u1680\u2000-\u200a\u2028\u2029\u202f\u205f\u3000\ufeff"
_SPACE = re.compile(f"[{_SPACE_CHARS}]")
_SPACES = re.compile(f"[{_SPACE_CHARS}]+")
# @key = value, the value quoted (anything but a quote, line breaks included) or a run of non-space characters -- which may
# swallow a comment's end: `/* @min=1*/` has the value `1*/` and the comment goes on (CustomShaderParamParser.tsx:92)
_ANNOTATION = re.compile(f"@(?P<key>[A-Za-z0-9_]+)[{_SPACE_CHARS}]*=[{_SPACE_CHARS}]*(?P<value>\"[^\"]*?\"|[^{_SPACE_CHARS}]+)")
_DECLARATION = re.compile(f"(?P<type>u?int|float|[iu]?vec[234])[{_SPACE_CHARS}]+(?P<name>[a-zA-Z_][a-zA-Z_0-9]*)")  # Validate.tsx:84-85
_STRIP = "\t\n\v\f\r \u00a0\u1680\u2000\u2001\u2002\u2003\u2004\u2005\u2006\u2007\u2008\u2009\u200a\u2028\u2029\u202f\u205f\u3000\ufeff"
_DECIMAL = re.compile(r"[+-]?(?:Infinity|[0-9]+\.?[0-9]*(?:[eE][+-]?[0-9]+)?|\.[0-9]+(?:[eE][+-]?[0-9]+)?)$")  # ASCII digits only: JavaScript's Number("\u0661") is NaN
_RADIX = {"x": 16, "X": 16, "b": 2, "B": 2, "o": 8, "O": 8}
FORMATS = ("numerical", "position", "color", "checkbox")


def js_number(text: str) -> float:
    """What JavaScript's Number(text) gives: the reference converts annotation values with it."""
    t = text.strip(_STRIP)
    if not t:
        return 0.0
    if len(t) > 2 and t[0] == "0" and t[1] in _RADIX:
        try:
            return float(int(t[2:], _RADIX[t[1]])) if re.fullmatch("[0-9a-fA-F]+", t[2:]) else math.nan
        except ValueError:
            return math.nan
    return float(t.replace("Infinity", "inf")) if _DECIMAL.match(t) else math.nan


def _table_number(v: float):
    """As the value stands in the reference's table (JSON): whole numbers without a fraction."""
    return int(v) if math.isfinite(v) and v == int(v) else v


# ---- the lexer ---------------------------------------------------------------------------------------------------------------
class Token(NamedTuple):
    kind: str          # "uniform" | "declaration" | "annotation"
    start: int
    end: int
    a: str = ""        # declaration: type name; annotation: key
    b: str = ""        # declaration: variable name; annotation: value as written (quotes included), up to a second `=`


def tokens(text: str) -> Iterator[Token]:
    """The tokens of a scene text that a parameter table is made of, in order.

    Modes: code, line comment, block comment.  The comment delimiters are lexed first in EVERY mode, as two-character units --
    where they do not act they are inert but still taken whole, so `*//` ends a block comment and leaves one slash, and `//` inside
    a block comment skips both characters (the loop below has the fine print).  A line comment ends in front of its line break (the break itself is code).  In a comment,
    `@key=value` is an annotation (it is not looked for in code, and a quoted value may run over line breaks: the line comment it
    stands in then runs on with it).  In code, `uniform` is a keyword wherever the seven letters stand; behind it -- across white
    space and comments -- come any number of declarations `type name`, until something else does."""
    pos, end = 0, len(text)
    mode = "code"
    after_uniform = False  # code mode: still behind a `uniform`, where declarations are looked for
    # what each delimiter does: (mode it acts in, mode it leads to); anywhere else it is inert -- taken whole, and the scan goes on
    # behind it WITHOUT looking for an earlier delimiter of this list again (so an inert `//` can be followed by an acting or
    # inert `/*`, then `*/`, in that order only; an inert `*/` in code is followed by one skipped character: `*//*` opens nothing)
    delimiters = (("//", "code", "line"), ("/*", "code", "block"), ("*/", "block", "code"))
    while pos < end:
        acted = False
        for mark, acts_in, leads_to in delimiters:
            if text.startswith(mark, pos):
                pos += 2
                if mode == acts_in:
                    mode, acted = leads_to, True
                    break
        if acted:
            continue
        if mode == "line" and text.startswith("\n", pos):
            mode = "code"  # (not consumed: the break is lexed again, as code)
            continue
        if pos >= end:
            break
        if mode != "code":
            m = _ANNOTATION.match(text, pos)
            if m:
                # (the value as the table sees it ends at a second `=`, should the written one hold any: `@min=1=2` is 1)
                yield Token("annotation", pos, m.end(), m.group("key"), m.group(0).split("=")[1].strip(_STRIP))
                pos = m.end()
            else:
                pos += 1
            continue
        if after_uniform:
            m = _SPACE.match(text, pos)  # (one white-space character at a time: a comment delimiter may stand behind any of them)
            if m:
                pos = m.end()
            m = _DECLARATION.match(text, pos)
            if m:
                yield Token("declaration", pos, m.end(), m.group("type"), m.group("name"))
                pos = m.end()
            else:
                after_uniform = False
            continue
        if text.startswith("uniform", pos):
            pos += 7
            yield Token("uniform", pos - 7, pos)
            after_uniform = True
            m = _SPACE.match(text, pos)
            if m:
                pos = m.end()
            continue
        pos += 1


# ---- the table ---------------------------------------------------------------------------------------------------------------
@dataclass
class _Param:
    quantity: int = 1
    type: str = "f"
    name: str = ""
    internal_name: str = ""
    tooltip: Optional[str] = None
    formats: List[str] = field(default_factory=lambda: ["numerical"])
    numbers: Dict[str, object] = field(default_factory=dict)  # min / max / step / sensitivity, in the order they were written
    scale: Optional[str] = None
    default: List[object] = field(default_factory=lambda: [0, 0, 0, 0])  # four zeros whatever the quantity (the reference's initial value)

    def entry(self) -> dict:
        e = {"success": True, "quantity": self.quantity, "type": self.type, "name": self.name}
        if self.tooltip is not None:
            e["tooltip"] = self.tooltip
        e["internalName"] = self.internal_name
        e["formats"] = list(self.formats)
        e.update(self.numbers)
        if self.scale is not None:
            e["scale"] = self.scale
        e["defaultValue"] = self.default
        return e


class ParamTable:
    """Folds tokens() into the reference's table.  A parameter is written out when the NEXT `uniform` (or the end of the text)
    is met, so the error entries of its annotations precede it, and annotations in front of the first `uniform` belong to the
    first parameter; a `uniform` that no declaration follows still makes an entry (with empty names)."""

    def __init__(self):
        self.entries: List[dict] = []
        self.current = _Param()
        self.seen_uniform = False

    def error(self, reason: str, token: Token, value: str):
        # the span of an error is the VALUE's, measured back from the annotation's end -- of the value without its quotes, so for a
        # quoted one it starts a character late and ends on the closing quote (the reference's arithmetic)
        self.entries.append({"success": False, "reason": reason, "start": token.end - len(value), "end": token.end})

    def feed(self, t: Token):
        if t.kind == "uniform":
            if self.seen_uniform:
                self.entries.append(self.current.entry())
                self.current = _Param()
            self.seen_uniform = True
        elif t.kind == "declaration":
            p = self.current
            p.type = {"u": "ui", "i": "i"}.get(t.a[0], "f")
            p.quantity = int(t.a[-1]) if "vec" in t.a else 1
            p.name = p.internal_name = t.b
        else:
            value = t.b[1:-1] if t.b.startswith('"') else t.b
            handler = getattr(self, "_on_" + t.a, None)  # unknown keys are ignored
            if handler is not None:
                handler(t, value)

    def finish(self) -> List[dict]:
        if self.seen_uniform:
            self.entries.append(self.current.entry())
        return self.entries

    # one handler per annotation key
    def _number(self, t: Token, value: str):
        v = js_number(value)
        if math.isnan(v):
            self.error(f"Expected property '{t.a}' to be a number.", t, value)
        self.current.numbers[t.a] = _table_number(v)  # (reported AND stored)

    _on_min = _on_max = _on_step = _on_sensitivity = _number

    def _on_scale(self, t: Token, value: str):
        if value == "log":  # "linear" is the unset state
            self.current.scale = "log"

    def _on_name(self, t: Token, value: str):
        self.current.name = value

    def _on_tooltip(self, t: Token, value: str):
        self.current.tooltip = value

    def _on_format(self, t: Token, value: str):
        chosen: List[str] = []
        for f in value.split("/"):
            if f not in FORMATS:
                self.error(f"Unknown input format '{f}'. Accepted values are \"numerical\", \"position\", \"color\", and \"checkbox\"", t, value)
            elif f not in chosen:
                chosen.append(f)
        self.current.formats = chosen

    def _on_default(self, t: Token, value: str):
        parts = value.split(",")
        if len(parts) != self.current.quantity:  # reported AND taken
            self.error(f"This variable requires {self.current.quantity} default values, but {len(parts)} were supplied. "
                       "Note that you need quotes if a value contains spaces.", t, value)
        self.current.default = [_table_number(js_number(x)) for x in parts]


def get_custom_shader_params(src: str) -> List[dict]:
    """The parameter table of a scene text, entry for entry what the reference's getCustomShaderParams returns
    (CustomShaderParamParser.tsx:8-209): per ``uniform`` a dict with success, quantity, type, name, internalName,
    formats, defaultValue and, when annotated, tooltip / min / max / step / sensitivity / scale; malformed
    annotations give ``{"success": False, "reason", "start", "end"}`` entries, in the order they are met.
    (tokens() and ParamTable say which of the reference's habits that includes.)  Positions are indices into the Python
    string (UTF-16 code units in the reference: they differ only beyond the BMP)."""
    table = ParamTable()
    for t in tokens(src):
        table.feed(t)
    return table.finish()


def default_custom_shader_parameters(src: str) -> Dict[str, dict]:
    """``customShaderParameters`` from the parameter table (CustomSettings.tsx:148-173): per successful entry
    ``{count: quantity, type, data: defaultValue}``.  (The reference's table always carries a defaultValue -- four
    zeros when there is no ``@default`` -- so its ``?? new Array(quantity).fill(0)`` never fires; the data is cut to
    the quantity here, which is what reaches the uniform, RenderJobExecutor.tsx:266 / Uniforms.tsx.)"""
    res = {}
    for p in get_custom_shader_params(src):
        if not p.get("success"):
            continue
        data = [float(v) if not (isinstance(v, float) and math.isnan(v)) else 0.0 for v in p["defaultValue"]][: p["quantity"]]
        data += [0.0] * (p["quantity"] - len(data))
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
