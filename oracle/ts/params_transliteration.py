"""Test infrastructure, not product: a statement-for-statement restatement in Python of the reference's scene-parameter
scanner, getCustomShaderParams (client/src/settings/shader-editor/CustomShaderParamParser.tsx:8-209, with
client/src/util/StringStream.tsx) -- the same character stream, the same state machine, the same quirks.  Until round 4 this
text WAS the product's scanner (raymarching-engine_amd/params.py); the product now has a lexer and a table builder of its own,
and this file is what tests/test_params_cpu.py holds it to on random texts, next to the outputs of the reference's own
function recorded under node (tests/golden/params_reference.json, oracle/ts/gen_params_golden.py).  Only tests import it."""
from __future__ import annotations

import math
import re
from typing import List, Optional

# JavaScript's \s and \w (the scanner's regular expressions are JavaScript's)
_JS_S = "[\\t\\n\\v\\f\\r \u00a0\u1680\u2000-\u200a\u2028\u2029\u202f\u205f\u3000\ufeff]"
_JS_NS = _JS_S.replace("[", "[^", 1)
_WS = re.compile(_JS_S)
_KVP = re.compile("@[A-Za-z0-9_]+" + _JS_S + "*=" + _JS_S + '*("[^"]*?"|' + _JS_NS + "+)")       # CustomShaderParamParser.tsx:92
_DECL = re.compile("(u?int|float|[iu]?vec[234])" + _JS_S + "+[a-zA-Z_][a-zA-Z_0-9]*")          # Validate.tsx:84-85
_SPLIT_WS = re.compile(_JS_S + "+")
_JS_DECIMAL = re.compile(r"[+-]?(Infinity|(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?))$")


def _js_number(text: str) -> float:
    """JavaScript's Number(string)."""
    t = text.strip("\t\n\v\f\r \u00a0\u1680\u2000\u2001\u2002\u2003\u2004\u2005\u2006\u2007\u2008\u2009\u200a\u2028\u2029\u202f\u205f\u3000\ufeff")
    if t == "":
        return 0.0
    if re.match(r"0[xX][0-9a-fA-F]+$", t):
        return float(int(t[2:], 16))
    if re.match(r"0[bB][01]+$", t):
        return float(int(t[2:], 2))
    if re.match(r"0[oO][0-7]+$", t):
        return float(int(t[2:], 8))
    if not _JS_DECIMAL.match(t):
        return math.nan
    return float(t.replace("Infinity", "inf"))


def _num(v: float):
    """As the value appears in the reference's table: integers print without a fraction."""
    return int(v) if isinstance(v, float) and math.isfinite(v) and v == int(v) else v


class _Stream:
    """client/src/util/StringStream.tsx."""

    def __init__(self, text: str):
        self.s, self.pos = text, 0

    def done(self) -> bool:
        return self.pos >= len(self.s)

    def match(self, pattern, no_consume: bool = False) -> Optional[str]:
        if isinstance(pattern, str):
            if self.s.startswith(pattern, self.pos):
                if not no_consume:
                    self.pos += len(pattern)
                return pattern
            return None
        m = pattern.match(self.s, self.pos)  # every pattern of the scanner is anchored (^)
        if m and m.group(0):
            if not no_consume:
                self.pos += len(m.group(0))
            return m.group(0)
        return None

    def next(self, n: int) -> str:
        self.pos += n
        return self.s[self.pos - n: self.pos]


def get_custom_shader_params(src: str) -> List[dict]:
    """The parameter table of a scene text, entry for entry what the reference's getCustomShaderParams returns
    (CustomShaderParamParser.tsx:8-209): per ``uniform`` a dict with success, quantity, type, name, internalName,
    formats, defaultValue and, when annotated, tooltip / min / max / step / sensitivity / scale; malformed
    annotations give ``{"success": False, "reason", "start", "end"}`` entries, in the order they are met.

    Behaviour that looks odd and is the reference's (each pinned by the fixture):
    * a parameter is emitted when the NEXT ``uniform`` keyword (or the end of the text) is reached, so its error
      entries precede it, and annotations placed before the first ``uniform`` attach to the first parameter;
    * ``defaultValue`` is ``[0, 0, 0, 0]`` when there is no ``@default``, whatever the quantity; a ``@default`` of
      the wrong length is reported AND taken;
    * annotations are read inside ``/* */`` comments as well as ``//`` ones; ``//``, ``/*`` and ``*/`` are consumed
      wherever they stand (the comparison with the comment state comes second, :67-83);
    * ``uniform`` is matched as a substring anywhere outside comments; what follows it need not be a declaration
      (then the entry has empty names); ``uint`` and ``uvec*`` are type "ui", ``int`` / ``ivec*`` "i";
    * a value that is not a number is reported and stored as NaN; unknown keys are ignored; ``@scale=linear`` leaves
      scale unset.
    Positions are indices into the Python string (UTF-16 code units in the reference: they differ only beyond the BMP)."""
    st = _Stream(src)
    out: List[dict] = []
    in_comment = None  # None | "line" | "block"
    cur = dict(quantity=1, type="f", name="", internalName="", tooltip=None, formats=["numerical"], other={}, scale=None,
               default=[0, 0, 0, 0])
    state = {"parse": 0, "first": True}

    def add_param():  # :38-64
        state["parse"] = 1
        if not state["first"]:
            e = {"success": True, "quantity": cur["quantity"], "type": cur["type"], "name": cur["name"]}
            if cur["tooltip"] is not None:
                e["tooltip"] = cur["tooltip"]
            e["internalName"] = cur["internalName"]
            e["formats"] = list(cur["formats"])
            e.update(cur["other"])
            if cur["scale"] is not None:
                e["scale"] = cur["scale"]
            e["defaultValue"] = cur["default"]
            out.append(e)
            cur.update(quantity=1, type="f", name="", internalName="", tooltip=None, formats=["numerical"], other={}, scale=None,
                       default=[0, 0, 0, 0])
        state["first"] = False
        st.match(_WS)

    while not st.done():
        # entering / leaving comments, :67-83 (the match is evaluated before the state is looked at)
        if st.match("//") and not in_comment:
            in_comment = "line"
            continue
        if st.match("/*") and not in_comment:
            in_comment = "block"
            continue
        if st.match("*/") and in_comment == "block":
            in_comment = None
            continue
        if st.match("\n", True) and in_comment == "line":
            in_comment = None
            continue
        if in_comment:  # :86-171
            kvp = st.match(_KVP)
            if kvp:
                parts = [e.strip() for e in kvp.split("=")]
                raw_key, raw_value = parts[0], parts[1]
                key = raw_key[1:]
                value = raw_value[1:-1] if raw_value[:1] == '"' else raw_value
                if key in ("min", "max", "step", "sensitivity"):
                    v = _js_number(value)
                    if math.isnan(v):
                        out.append({"success": False, "reason": f"Expected property '{key}' to be a number.",
                                    "start": st.pos - len(value), "end": st.pos})
                    cur["other"][key] = _num(v)
                elif key == "scale":
                    if value == "log":
                        cur["scale"] = "log"
                elif key == "name":
                    cur["name"] = value
                elif key == "tooltip":
                    cur["tooltip"] = value
                elif key == "format":
                    cur["formats"] = []
                    for f in value.split("/"):
                        if f in ("numerical", "position", "color", "checkbox"):
                            if f not in cur["formats"]:
                                cur["formats"].append(f)
                        else:
                            out.append({"success": False,
                                        "reason": f"Unknown input format '{f}'. Accepted values are \"numerical\", \"position\", \"color\", and \"checkbox\"",
                                        "start": st.pos - len(value), "end": st.pos})
                elif key == "default":
                    vals = value.split(",")
                    if len(vals) != cur["quantity"]:
                        out.append({"success": False,
                                    "reason": f"This variable requires {cur['quantity']} default values, but {len(vals)} were supplied. "
                                              "Note that you need quotes if a value contains spaces.",
                                    "start": st.pos - len(value), "end": st.pos})
                    cur["default"] = [_num(_js_number(x)) for x in vals]
                continue
            st.next(1)
        else:  # :173-203
            if state["parse"] == 1:
                st.match(_WS)
                decl = st.match(_DECL)
                if decl:
                    pieces = [e.strip() for e in _SPLIT_WS.split(decl)]
                    typename, var = pieces[0], pieces[1] if len(pieces) > 1 else ""
                    if not typename or not var:
                        continue
                    cur["quantity"], cur["type"] = 1, "f"
                    if typename[0] == "u":
                        cur["type"] = "ui"
                    if typename[0] == "i":
                        cur["type"] = "i"
                    if "vec" in typename:
                        cur["quantity"] = int(typename[-1])
                    cur["name"] = cur["internalName"] = var
                else:
                    state["parse"] = 0
                continue
            if st.match("uniform"):
                add_param()
                continue
            st.next(1)
    add_param()
    return out
