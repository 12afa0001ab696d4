"""Scene-definition / SDF-composition API.

The reference's scene is GLSL text (``sdfShaderSource``) defining ``sdf()`` and
optional material functions (client/src/settings/shader-editor/Validate.tsx:8-57).
A composed scene here has two back ends built from ONE description:

* ``desc()``  -> ``RmSceneDesc`` for the HIP kernel (kind + parameters +
  primitive table + material constants, include/hip_raymarch.h);
* ``glsl()``  -> scene text satisfying the reference's scene contract, so the
  same scene can be fed to the reference's GLSL path (``sdfShaderSource`` keeps
  working, and it is what the golden generator renders).

Helper SDFs the emitted GLSL calls (``sdfSphere``, ``sdBox``) are the ones the
reference's shader already provides above its splice point
(raymarcher.frag:74-76,108-112).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

from . import abi


def _f(x: float) -> str:
    """GLSL float literal that round-trips the fp32 value.  (A conforming GLSL front end parses it to exactly that
    float.  Chrome 88's ANGLE -- the software-GL stack the goldens are made with -- re-emits a shader's constants with 8
    significant digits before its backend compiles it, so about 1 % of arbitrary constants arrive one ulp off there,
    whatever the literal's form: oracle/gl/gen_random_golden.py, which only uses constants that survive that.)"""
    import numpy as np

    v = float(np.float32(x))
    s = repr(v)
    if "e" in s or "E" in s:
        m, e = s.lower().split("e")
        if "." not in m:
            m += ".0"
        return f"{m}e{int(e)}"
    if "." not in s and "inf" not in s and "nan" not in s:
        s += ".0"
    return s


def _v3(v: Sequence[float]) -> str:
    return f"vec3({_f(v[0])}, {_f(v[1])}, {_f(v[2])})"


@dataclass
class Material:
    """Constants of the seven material functions; defaults = Validate.tsx:18-51."""

    diffuse: Sequence[float] = (0.6, 0.6, 0.6)
    diffuse_cutoff: float = 35.0
    specular: Sequence[float] = (0.6, 0.6, 0.6)
    specular_cutoff: float = 35.0
    roughness: float = 0.2
    subsurface: float = 11111115.0
    subsurface_color: Sequence[float] = (1.0, 1.0, 1.0)
    ior: float = 100.0
    sky_color: Sequence[float] = (0.7, 0.8, 1.0)
    sky_floor: float = 0.2
    sky_scale: float = 2.0
    sky_radius: float = 36.0
    sky_axis: int = 1

    def is_default(self) -> bool:
        return self == Material()

    def to_c(self) -> abi.RmMaterial:
        m = abi.RmMaterial()
        m.diffuse[:] = list(self.diffuse)
        m.diffuse_cutoff = self.diffuse_cutoff
        m.specular[:] = list(self.specular)
        m.specular_cutoff = self.specular_cutoff
        m.roughness = self.roughness
        m.subsurface = self.subsurface
        m.subsurface_color[:] = list(self.subsurface_color)
        m.ior = self.ior
        m.sky_color[:] = list(self.sky_color)
        m.sky_floor = self.sky_floor
        m.sky_scale = self.sky_scale
        m.sky_radius = self.sky_radius
        m.sky_axis = self.sky_axis
        return m

    def glsl(self) -> str:
        """Material functions as scene text; empty for the defaults (the
        reference then appends its own, Validate.tsx:18-51)."""
        if self.is_default():
            return ""
        ax = "xyz"[self.sky_axis]
        return "\n".join(
            [
                f"vec3 sceneDiffuseColor(vec3 position) {{ if (length(position) > {_f(self.diffuse_cutoff)}) return vec3(0.0); return {_v3(self.diffuse)}; }}",
                f"vec3 sceneSpecularColor(vec3 position) {{ if (length(position) > {_f(self.specular_cutoff)}) return vec3(0.0); return {_v3(self.specular)}; }}",
                f"float sceneSpecularRoughness(vec3 position) {{ return {_f(self.roughness)}; }}",
                f"float sceneSubsurfaceScattering(vec3 position) {{ return {_f(self.subsurface)}; }}",
                f"vec3 sceneSubsurfaceScatteringColor(vec3 position) {{ return {_v3(self.subsurface_color)}; }}",
                f"float sceneIOR(vec3 position) {{ return {_f(self.ior)}; }}",
                "vec3 sceneEmission(vec3 position) {"
                f" float d = max(normalize(position).{ax}, {_f(self.sky_floor)});"
                f" vec3 brightColor = {_v3(self.sky_color)} * d * 1.0;"
                f" return (length(position) > {_f(self.sky_radius)}) ? (brightColor * {_f(self.sky_scale)}) : vec3(0.0); }}",
            ]
        )


@dataclass
class Surface:
    """The material values a shape of a composed scene can have of its own (RmSurface): the reference's material
    contract is seven functions of POSITION (Validate.tsx:18-51, examples/guide.glsl:51-88), and for a CsgScene whose shapes
    name surfaces they return, at a position, the values of the nearest shape (CsgScene.material_glsl).  The cut-off radii
    and the sky stay the scene's (Material)."""

    diffuse: Sequence[float] = (0.6, 0.6, 0.6)
    specular: Sequence[float] = (0.6, 0.6, 0.6)
    roughness: float = 0.2
    subsurface: float = 11111115.0
    subsurface_color: Sequence[float] = (1.0, 1.0, 1.0)
    ior: float = 100.0

    def to_c(self) -> abi.RmSurface:
        f = abi.RmSurface()
        f.diffuse[:] = list(self.diffuse)
        f.specular[:] = list(self.specular)
        f.roughness = self.roughness
        f.subsurface = self.subsurface
        f.subsurface_color[:] = list(self.subsurface_color)
        f.ior = self.ior
        return f


class Scene:
    """Base: a scene kind with parameters and a material."""

    kind: int = -1
    material: Material

    def params(self) -> List[float]:
        return []

    def prims(self) -> List[abi.RmPrim]:
        return []

    def sdf_glsl(self) -> str:
        raise NotImplementedError

    def glsl(self) -> str:
        mat = self.material.glsl()
        return self.sdf_glsl() + ("\n" + mat if mat else "") + "\n"

    def custom_shader_parameters(self) -> dict:
        """``customShaderParameters`` of the RenderJobSchema for this scene (the
        emitted GLSL bakes its parameters as literals, so: none)."""
        return {}

    def surfaces(self) -> List["Surface"]:
        return []

    def desc(self) -> abi.RmSceneDesc:
        d = abi.RmSceneDesc()
        d.kind = self.kind
        prims = self.prims()
        d.nprims = len(prims)
        if prims:
            arr = (abi.RmPrim * len(prims))(*prims)
            d.prims = C.cast(arr, C.POINTER(abi.RmPrim))
            d._keepalive = arr  # the desc borrows the table
        surfaces = self.surfaces()
        d.nsurfaces = len(surfaces)
        if surfaces:
            sarr = (abi.RmSurface * len(surfaces))(*[f.to_c() for f in surfaces])
            d.surfaces = C.cast(sarr, C.POINTER(abi.RmSurface))
            d._keepalive_surfaces = sarr
        p = self.params()
        for i, v in enumerate(p):
            d.params[i] = v
        d.material = self.material.to_c()
        return d


# ---- primitive table (CSG fold) -------------------------------------------


@dataclass
class _Node:
    prim: int
    op: int
    k: float
    center: Sequence[float]
    size: Sequence[float]
    surface: int = 0  # 0 = the scene's material; k = the k-th Surface given to the scene's shapes


SHAPE_PRIMS = (abi.RM_PRIM_SPHERE, abi.RM_PRIM_BOX, abi.RM_PRIM_KIND, abi.RM_PRIM_TORUS, abi.RM_PRIM_CYLINDER, abi.RM_PRIM_PLANE)  # rows that contribute a distance term
KIND_SHAPES = (abi.RM_SCENE_MANDELBULB, abi.RM_SCENE_SPHERE_LATTICE)      # kinds a RM_PRIM_KIND row can evaluate


class CsgScene(Scene):
    """Left fold of primitives: ``d = prim0; d = op_i(d, prim_i)``.

    ``sphere``/``box`` start or extend the fold with the current operator
    (``union`` by default); ``smooth_union(k)``, ``subtract()``,
    ``intersect()``, ``union()`` select the operator for the primitives that
    follow.  The smooth union is the polynomial smooth-min of
    examples/smooth-tree.glsl:20-22.

    Domain operators transform the point at which the primitives
    that FOLLOW them are evaluated: ``repeat(period)`` tiles space like the
    ``repeat`` of the reference's sphere-grid example
    (dist/examples/sphere-grid.glsl:42-49), ``fold(scale, offset, angles)`` is one
    level of its kaleidoscopic folds (examples/tree.glsl:24-32: divide by the
    scale, mirror, shift, three plane rotations; the distances of the following
    primitives are multiplied back by the scale).  They compose: several folds
    in a row give the levels of a KIFS fractal over any primitives.
    """

    kind = abi.RM_SCENE_TABLE

    def __init__(self, material: Optional[Material] = None):
        self.material = material or Material()
        self._nodes: List[_Node] = []
        self._op = abi.RM_OP_UNION
        self._k = 0.0
        self._surfaces: List[Surface] = []
        self._kind_scene: Optional[Scene] = None  # the scene kind whose estimator the table's RM_PRIM_KIND rows evaluate

    def _surface(self, surface: Optional[Surface]) -> int:
        """The index a shape row carries for `surface` (0 = none: the scene's material block)."""
        if surface is None:
            return 0
        for i, f in enumerate(self._surfaces):
            if f == surface:
                return i + 1
        if len(self._surfaces) >= abi.RM_MAX_SURFACES:
            raise ValueError(f"a scene has at most {abi.RM_MAX_SURFACES} surfaces")
        self._surfaces.append(surface)
        return len(self._surfaces)

    def surfaces(self) -> List[Surface]:
        return list(self._surfaces)

    def union(self):
        self._op, self._k = abi.RM_OP_UNION, 0.0
        return self

    def smooth_union(self, k: float):
        self._op, self._k = abi.RM_OP_SMOOTH_UNION, float(k)
        return self

    def subtract(self):
        self._op, self._k = abi.RM_OP_SUBTRACT, 0.0
        return self

    def intersect(self):
        self._op, self._k = abi.RM_OP_INTERSECT, 0.0
        return self

    def smooth_subtract(self, k: float):
        """The shapes that follow are carved out of the running value with a rounded edge of radius ~k (the polynomial smooth
        maximum: h = clamp(0.5 - 0.5 (d + di) / k, 0, 1); mix(d, -di, h) + k h (1 - h))."""
        self._op, self._k = abi.RM_OP_SMOOTH_SUBTRACT, float(k)
        return self

    def smooth_intersect(self, k: float):
        """... intersected with it, likewise (h = clamp(0.5 - 0.5 (d - di) / k, 0, 1); mix(d, di, h) + k h (1 - h))."""
        self._op, self._k = abi.RM_OP_SMOOTH_INTERSECT, float(k)
        return self

    def sphere(self, center: Sequence[float], radius: float, surface: Optional[Surface] = None):
        """`surface`: material values of this shape's own (Surface); the material functions then depend on the position."""
        self._nodes.append(_Node(abi.RM_PRIM_SPHERE, self._op, self._k, tuple(center), (radius, 0.0, 0.0), self._surface(surface)))
        return self

    def box(self, center: Sequence[float], half_extents: Sequence[float], surface: Optional[Surface] = None):
        self._nodes.append(_Node(abi.RM_PRIM_BOX, self._op, self._k, tuple(center), tuple(half_extents), self._surface(surface)))
        return self

    def torus(self, center: Sequence[float], major_radius: float, minor_radius: float, surface: Optional[Surface] = None):
        """A ring about `center` in its xz plane (axis y)."""
        self._nodes.append(_Node(abi.RM_PRIM_TORUS, self._op, self._k, tuple(center), (float(major_radius), float(minor_radius), 0.0), self._surface(surface)))
        return self

    def cylinder(self, center: Sequence[float], radius: float, half_height: float, surface: Optional[Surface] = None):
        """A capped cylinder about `center`, axis y."""
        self._nodes.append(_Node(abi.RM_PRIM_CYLINDER, self._op, self._k, tuple(center), (float(radius), float(half_height), 0.0), self._surface(surface)))
        return self

    def plane(self, point: Sequence[float], normal: Sequence[float], surface: Optional[Surface] = None):
        """The half space behind the plane through `point` with the UNIT normal `normal` (distance = dot(p - point, normal))."""
        self._nodes.append(_Node(abi.RM_PRIM_PLANE, self._op, self._k, tuple(point), tuple(float(v) for v in normal), self._surface(surface)))
        return self

    def shape(self, scene: "Scene", center: Sequence[float] = (0.0, 0.0, 0.0), surface: Optional[Surface] = None):
        """A shape whose distance term is another scene kind's own estimator at `p - center` (RM_PRIM_KIND): a Mandelbulb
        or a SphereLattice, folded with the current operator like a sphere or a box -- ``CsgScene().shape(Mandelbulb())
        .intersect().box(...)`` is a Mandelbulb cut by a box.  One kind (one parameter set) per table, any number of rows."""
        if scene.kind not in KIND_SHAPES:
            raise ValueError("shape(): the kinds a table row can evaluate are Mandelbulb and SphereLattice")
        if self._kind_scene is not None and (self._kind_scene.kind != scene.kind or list(self._kind_scene.params()) != list(scene.params())):
            raise ValueError("shape(): a table evaluates ONE kind with one set of parameters (they travel in the scene's parameter block)")
        self._kind_scene = scene
        self._nodes.append(_Node(abi.RM_PRIM_KIND, self._op, self._k, tuple(float(v) for v in center), (float(scene.kind), 0.0, 0.0), self._surface(surface)))
        return self

    def params(self):
        return list(self._kind_scene.params()) if self._kind_scene is not None else []

    def repeat(self, period: Sequence[float]):
        """q = mod(q + period / 2, period) - period / 2 for the primitives that follow; every period > 0."""
        self._nodes.append(_Node(abi.RM_PRIM_REPEAT, 0, 0.0, (0.0, 0.0, 0.0), tuple(float(v) for v in period)))
        return self

    def fold(self, scale: float, offset: Sequence[float], angles: Sequence[float] = (0.0, 0.0, 0.0)):
        """q = abs(q / scale) - offset, then the rotations in the xy, yz and xz planes by `angles` (radians)."""
        self._nodes.append(_Node(abi.RM_PRIM_FOLD, 0, float(scale), tuple(float(v) for v in offset), tuple(float(v) for v in angles)))
        return self

    def prims(self) -> List[abi.RmPrim]:
        out = []
        for n in self._nodes:
            p = abi.RmPrim()
            p.type = n.prim | (n.op << 8) | (n.surface << 16)
            p.k = n.k
            p.center[:] = list(n.center)
            p.size[:] = list(n.size)
            out.append(p)
        return out

    def sdf_glsl(self) -> str:
        shapes = [n for n in self._nodes if n.prim in SHAPE_PRIMS]
        if not shapes:
            raise ValueError("a CSG scene needs at least one shape (sphere, box, torus, cylinder, plane or a kind)")
        domain = len(shapes) != len(self._nodes)
        lines = []
        if self._kind_scene is not None:  # the kind's own text, as a function of its own name
            text = self._kind_scene.sdf_glsl()
            assert text.count("float sdf(") == 1
            lines.append(text.replace("float sdf(", "float rmKindSdf("))
        if any(n.op == abi.RM_OP_SMOOTH_UNION for n in shapes[1:]):
            lines.append(
                "float rmSmoothUnion(float d1, float d2, float k) {"
                " float h = clamp(0.5 + 0.5 * (d2 - d1) / k, 0.0, 1.0);"
                " return mix(d2, d1, h) - k * h * (1.0 - h); }"
            )
        if any(n.op == abi.RM_OP_SMOOTH_SUBTRACT for n in shapes[1:]):
            lines.append("float rmSmoothSubtract(float d, float di, float k) {"
                         " float h = clamp(0.5 - 0.5 * (d + di) / k, 0.0, 1.0);"
                         " return mix(d, -di, h) + k * h * (1.0 - h); }")
        if any(n.op == abi.RM_OP_SMOOTH_INTERSECT for n in shapes[1:]):
            lines.append("float rmSmoothIntersect(float d, float di, float k) {"
                         " float h = clamp(0.5 - 0.5 * (d - di) / k, 0.0, 1.0);"
                         " return mix(d, di, h) + k * h * (1.0 - h); }")
        lines += self._shape_helpers()
        if any(n.prim == abi.RM_PRIM_FOLD for n in self._nodes):
            lines.append(  # scalar by scalar, in the order of the oracle / the kernel (tree.glsl:24-32 writes it with mat2)
                "vec3 rmFold(vec3 q, float scale, vec3 off, vec3 ang) {"
                " q = q / scale; q = abs(q) - off; float c; float s; float nx; float ny;"
                " c = cos(ang.x); s = sin(ang.x); nx = q.x * c + q.y * -s; ny = q.x * s + q.y * c; q.x = nx; q.y = ny;"
                " c = cos(ang.y); s = sin(ang.y); nx = q.y * c + q.z * -s; ny = q.y * s + q.z * c; q.y = nx; q.z = ny;"
                " c = cos(ang.z); s = sin(ang.z); nx = q.x * c + q.z * -s; ny = q.x * s + q.z * c; q.x = nx; q.z = ny;"
                " return q; }"
            )
        lines.append("float sdf(vec3 p) {")
        if domain:
            lines.append("  vec3 q = p; float factor = 1.0; float d;")
        first = True
        for n, e in self._terms(lines, domain):
            if first:
                lines.append(f"  d = {e};" if domain else f"  float d = {e};")
                first = False
            elif n.op == abi.RM_OP_UNION:
                lines.append(f"  d = min(d, {e});")
            elif n.op == abi.RM_OP_SMOOTH_UNION:
                lines.append(f"  d = rmSmoothUnion(d, {e}, {_f(n.k)});")
            elif n.op == abi.RM_OP_SUBTRACT:
                lines.append(f"  d = max(d, -{e});")
            elif n.op == abi.RM_OP_SMOOTH_SUBTRACT:
                lines.append(f"  d = rmSmoothSubtract(d, {e}, {_f(n.k)});")
            elif n.op == abi.RM_OP_SMOOTH_INTERSECT:
                lines.append(f"  d = rmSmoothIntersect(d, {e}, {_f(n.k)});")
            else:
                lines.append(f"  d = max(d, {e});")
        lines.append("  return d;")
        lines.append("}")
        return "\n".join(lines)

    def _shape_helpers(self) -> List[str]:
        """The GLSL of the shapes the reference's shader has no helper for (raymarcher.frag has sdfSphere :74 and sdBox :108), scalar
        by scalar in the order of the oracle and the kernels."""
        out = []
        used = {n.prim for n in self._nodes}
        if used & {abi.RM_PRIM_TORUS, abi.RM_PRIM_CYLINDER}:
            out.append("float rmLength2(float x, float y) { return sqrt(x * x + y * y); }")
        if abi.RM_PRIM_TORUS in used:
            out.append("float rmTorus(vec3 p, float R, float r) { return rmLength2(rmLength2(p.x, p.z) - R, p.y) - r; }")
        if abi.RM_PRIM_CYLINDER in used:
            out.append("float rmCylinder(vec3 p, float r, float h) { float dx = rmLength2(p.x, p.z) - r; float dy = abs(p.y) - h;"
                       " return min(max(dx, dy), 0.0) + rmLength2(max(dx, 0.0), max(dy, 0.0)); }")
        if abi.RM_PRIM_PLANE in used:
            out.append("float rmPlane(vec3 p, vec3 n) { return p.x * n.x + p.y * n.y + p.z * n.z; }")
        return out

    def _terms(self, lines: List[str], domain: bool):
        """The shape rows in table order with the GLSL expression of each one's distance term; the statements of the domain
        rows in between are appended to `lines` as they come (they act on `q` / `factor`)."""
        q = "q" if domain else "p"
        for n in self._nodes:
            if n.prim == abi.RM_PRIM_REPEAT:
                lines.append(f"  q = mod(q + 0.5 * {_v3(n.size)}, {_v3(n.size)}) - 0.5 * {_v3(n.size)};")
                continue
            if n.prim == abi.RM_PRIM_FOLD:
                lines.append(f"  q = rmFold(q, {_f(n.k)}, {_v3(n.center)}, {_v3(n.size)}); factor = factor * {_f(n.k)};")
                continue
            if n.prim == abi.RM_PRIM_SPHERE:
                e = f"sdfSphere({q}, {_v3(n.center)}, {_f(n.size[0])})"
            elif n.prim == abi.RM_PRIM_KIND:
                e = f"rmKindSdf({q} - {_v3(n.center)})"
            elif n.prim == abi.RM_PRIM_TORUS:
                e = f"rmTorus({q} - {_v3(n.center)}, {_f(n.size[0])}, {_f(n.size[1])})"
            elif n.prim == abi.RM_PRIM_CYLINDER:
                e = f"rmCylinder({q} - {_v3(n.center)}, {_f(n.size[0])}, {_f(n.size[1])})"
            elif n.prim == abi.RM_PRIM_PLANE:
                e = f"rmPlane({q} - {_v3(n.center)}, {_v3(n.size)})"
            else:
                e = f"sdBox({q} - {_v3(n.center)}, {_v3(n.size)})"
            if domain:
                e = f"({e} * factor)"
            yield n, e

    def material_glsl(self) -> str:
        """The seven material functions of a scene whose shapes name surfaces: at `position` the values of the shape row
        whose distance term there is the smallest (rmSurfaceIndex: the terms of sdf()'s fold before their operators; the
        earliest row on a tie, a NaN never wins) -- the rule the kernel and the oracle implement (rm_device.hpp
        surface_index).  Cut-offs and the sky are the scene's."""
        m = self.material
        shapes = [n for n in self._nodes if n.prim in SHAPE_PRIMS]
        domain = len(shapes) != len(self._nodes)
        all_surfaces = [Surface(m.diffuse, m.specular, m.roughness, m.subsurface, m.subsurface_color, m.ior)] + self._surfaces
        n = len(all_surfaces)
        lines = ["int rmSurfaceIndex(vec3 p) {"]
        if domain:
            lines.append("  vec3 q = p; float factor = 1.0;")
        lines.append("  float best = 0.0; float di; int surface = 0;")
        first = True
        for node, e in self._terms(lines, domain):
            if first:
                lines.append(f"  best = {e}; surface = {node.surface};")
                first = False
            else:
                lines.append(f"  di = {e}; if (di < best) {{ best = di; surface = {node.surface}; }}")
        lines += ["  return surface;", "}"]

        def table(name, typ, values):
            return f"const {typ} {name}[{n}] = {typ}[{n}](" + ", ".join(values) + ");"

        lines += [
            table("rmDiffuse", "vec3", [_v3(f.diffuse) for f in all_surfaces]),
            table("rmSpecular", "vec3", [_v3(f.specular) for f in all_surfaces]),
            table("rmSubsurfaceColor", "vec3", [_v3(f.subsurface_color) for f in all_surfaces]),
            table("rmRoughness", "float", [_f(f.roughness) for f in all_surfaces]),
            table("rmSubsurface", "float", [_f(f.subsurface) for f in all_surfaces]),
            table("rmIor", "float", [_f(f.ior) for f in all_surfaces]),
            f"vec3 sceneDiffuseColor(vec3 position) {{ if (length(position) > {_f(m.diffuse_cutoff)}) return vec3(0.0); return rmDiffuse[rmSurfaceIndex(position)]; }}",
            f"vec3 sceneSpecularColor(vec3 position) {{ if (length(position) > {_f(m.specular_cutoff)}) return vec3(0.0); return rmSpecular[rmSurfaceIndex(position)]; }}",
            "float sceneSpecularRoughness(vec3 position) { return rmRoughness[rmSurfaceIndex(position)]; }",
            "float sceneSubsurfaceScattering(vec3 position) { return rmSubsurface[rmSurfaceIndex(position)]; }",
            "vec3 sceneSubsurfaceScatteringColor(vec3 position) { return rmSubsurfaceColor[rmSurfaceIndex(position)]; }",
            "float sceneIOR(vec3 position) { return rmIor[rmSurfaceIndex(position)]; }",
        ]
        ax = "xyz"[m.sky_axis]
        lines.append("vec3 sceneEmission(vec3 position) {"
                     f" float d = max(normalize(position).{ax}, {_f(m.sky_floor)});"
                     f" vec3 brightColor = {_v3(m.sky_color)} * d * 1.0;"
                     f" return (length(position) > {_f(m.sky_radius)}) ? (brightColor * {_f(m.sky_scale)}) : vec3(0.0); }}")
        return "\n".join(lines)

    def glsl(self) -> str:
        if any(n.surface for n in self._nodes):
            return self.sdf_glsl() + "\n" + self.material_glsl() + "\n"
        return super().glsl()


def single_sphere(center=(0.0, 0.0, 0.0), radius=1.0, material: Optional[Material] = None) -> CsgScene:
    """BASELINE.json configs[0]/[1]: ``sdf = sdfSphere(p, 0, 1)``."""
    return CsgScene(material).sphere(center, radius)


def csg64(material: Optional[Material] = None) -> CsgScene:
    """BASELINE.json configs[3]: 64 spheres on a jittered 4x4x4 lattice
    (spacing 0.9, radii 0.25-0.45, fixed LCG seed 1) folded with smooth unions
    k = 0.2."""
    sc = CsgScene(material).smooth_union(0.2)
    state = 1

    def lcg():
        nonlocal state
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        return (state >> 8) / float(1 << 24)

    for iz in range(4):
        for iy in range(4):
            for ix in range(4):
                c = [(ix - 1.5) * 0.9 + (lcg() - 0.5) * 0.3,
                     (iy - 1.5) * 0.9 + (lcg() - 0.5) * 0.3,
                     (iz - 1.5) * 0.9 + (lcg() - 0.5) * 0.3]
                r = 0.25 + 0.2 * lcg()
                sc.sphere(c, r)
    return sc


def csg_blocks(material: Optional[Material] = None) -> CsgScene:
    """Not a BASELINE configuration: a long table of hard operators -- 180 boxes and spheres on a jittered 6x6x5 lattice under
    unions, then 12 spheres subtracted (fixed LCG seed 7) -- the kind of scene the fast build's row culling is for
    (include/hip_raymarch.h RM_RENDER_NO_CULL)."""
    sc = CsgScene(material).union()
    state = 7

    def lcg():
        nonlocal state
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        return (state >> 8) / float(1 << 24)

    for iz in range(5):
        for iy in range(6):
            for ix in range(6):
                c = [(ix - 2.5) * 0.7 + (lcg() - 0.5) * 0.25, (iy - 2.5) * 0.7 + (lcg() - 0.5) * 0.25, (iz - 2.0) * 0.7 + (lcg() - 0.5) * 0.25]
                if lcg() < 0.5:
                    sc.sphere(c, 0.15 + 0.2 * lcg())
                else:
                    sc.box(c, [0.1 + 0.2 * lcg(), 0.1 + 0.2 * lcg(), 0.1 + 0.2 * lcg()])
    sc.subtract()
    for _ in range(12):
        sc.sphere([(lcg() - 0.5) * 3.5, (lcg() - 0.5) * 3.5, (lcg() - 0.5) * 3.0], 0.3 + 0.4 * lcg())
    return sc


# ---- specialised kinds -----------------------------------------------------


class Mandelbulb(Scene):
    """Power-n Mandelbulb, spherical-coordinate distance estimator
    ``0.5*log(r)*r/dr`` (BASELINE.json configs[2]).  The
    reference has no such scene; this text is the definition both back ends
    implement."""

    kind = abi.RM_SCENE_MANDELBULB

    def __init__(self, power: float = 8.0, iterations: int = 8, bailout: float = 2.0, material: Optional[Material] = None):
        self.power, self.iterations, self.bailout = float(power), int(iterations), float(bailout)
        self.material = material or Material()

    def params(self):
        return [self.power, float(self.iterations), self.bailout]

    def sdf_glsl(self) -> str:
        return f"""float sdf(vec3 pos) {{
  vec3 z = pos;
  float dr = 1.0;
  float r = 0.0;
  for (int i = 0; i < {self.iterations}; i++) {{
    r = length(z);
    if (r > {_f(self.bailout)}) break;
    float theta = acos(z.z / r);
    float phi = atan(z.y, z.x);
    dr = pow(r, {_f(self.power)} - 1.0) * {_f(self.power)} * dr + 1.0;
    float zr = pow(r, {_f(self.power)});
    theta = theta * {_f(self.power)};
    phi = phi * {_f(self.power)};
    z = zr * vec3(sin(theta) * cos(phi), sin(phi) * sin(theta), cos(theta));
    z += pos;
  }}
  return 0.5 * log(r) * r / dr;
}}"""


class _ReferenceExample(Scene):
    """A scene kind that restates one of the reference's example scenes; its
    GLSL back end is the reference's own example text plus uniform values
    (``customShaderParameters``), so ``glsl()`` needs the example file name."""

    example: str = ""

    def sdf_glsl(self) -> str:
        raise NotImplementedError(
            f"the GLSL of this kind is the reference's examples/{self.example}; pass that text as sdfShaderSource"
        )


class SphereGridFractal(_ReferenceExample):
    """examples/guide.glsl:91-102 == examples/fractal1.glsl:23-34 (defaults from their //@default)."""

    kind = abi.RM_SCENE_SPHERE_GRID
    example = "fractal1.glsl"

    def __init__(self, big_sphere_size=4.0, iterations=8.0, grid_scale=0.33333333333, big_sphere_center=(0.0, 0.0, 10.0), material=None):
        self.big, self.iterations, self.scale, self.center = float(big_sphere_size), float(iterations), float(grid_scale), tuple(big_sphere_center)
        self.material = material or Material()

    def params(self):
        return [self.big, self.iterations, self.scale, *self.center]

    def custom_shader_parameters(self):
        return {
            "bigSphereSize": {"type": "f", "count": 1, "data": [self.big]},
            "fractalIterations": {"type": "f", "count": 1, "data": [self.iterations]},
            "gridScaleFactor": {"type": "f", "count": 1, "data": [self.scale]},
            "bigSphereCenter": {"type": "f", "count": 3, "data": list(self.center)},
        }


class SphereLattice(Scene):
    """dist/examples/sphere-grid.glsl:42-49: infinite lattice of spheres."""

    kind = abi.RM_SCENE_SPHERE_LATTICE

    def __init__(self, period=2.0, radius=0.4, material=None):
        self.period, self.radius = float(period), float(radius)
        self.material = material or Material()

    def params(self):
        return [self.period, self.radius]

    def sdf_glsl(self) -> str:
        h = self.period * 0.5
        return (
            f"float sdf(vec3 p) {{ vec3 rep = mod(p + {_f(h)}, vec3({_f(self.period)})) - {_f(h)};"
            f" return length(rep - vec3(0.0)) - {_f(self.radius)}; }}"
        )


def sphere_lattice_example() -> SphereLattice:
    """The reference's sphere-grid example with its own material functions
    (dist/examples/sphere-grid.glsl:1-40)."""
    return SphereLattice(2.0, 0.4, Material(diffuse=(0.5, 0.5, 0.5), specular=(0.9, 0.9, 0.9), roughness=0.01, sky_axis=0, sky_floor=0.0))


class MengerSponge(_ReferenceExample):
    """examples/menger-sponge.glsl:6-23."""

    kind = abi.RM_SCENE_MENGER
    example = "menger-sponge.glsl"

    def __init__(self, iterations=8.0, material=None):
        self.iterations = float(iterations)
        self.material = material or Material()

    def params(self):
        return [self.iterations]

    def custom_shader_parameters(self):
        return {"fractalIterations": {"type": "f", "count": 1, "data": [self.iterations]}}


class KifsTree(_ReferenceExample):
    """examples/tree.glsl:16-36 (smoothen=False) / examples/smooth-tree.glsl:30-56 (smoothen=True)."""

    kind = abi.RM_SCENE_KIFS_TREE

    def __init__(self, iterations=8.0, scale=0.7, angles=(2.9, -0.8, 0.4), offset=1.2, smoothen=False, material=None):
        self.iterations, self.scale, self.angles, self.offset, self.smoothen = float(iterations), float(scale), tuple(angles), float(offset), bool(smoothen)
        self.example = "smooth-tree.glsl" if smoothen else "tree.glsl"
        self.material = material or Material()

    def params(self):
        return [self.iterations, self.scale, *self.angles, self.offset, 1.0 if self.smoothen else 0.0]

    def custom_shader_parameters(self):
        p = {
            "fractalIterations": {"type": "f", "count": 1, "data": [self.iterations]},
            "scaleFactor": {"type": "f", "count": 1, "data": [self.scale]},
            "angles": {"type": "f", "count": 3, "data": list(self.angles)},
            "offset": {"type": "f", "count": 1, "data": [self.offset]},
        }
        if self.smoothen:
            p["smoothen"] = {"type": "i", "count": 1, "data": [1]}
        return p


class KifsBox(_ReferenceExample):
    """examples/rotation-fractal.glsl:16-35."""

    kind = abi.RM_SCENE_KIFS_BOX
    example = "rotation-fractal.glsl"

    def __init__(self, iterations=14.0, scale=0.5, angles=(0.4, 0.4, 0.4), offset=1.2, material=None):
        self.iterations, self.scale, self.angles, self.offset = float(iterations), float(scale), tuple(angles), float(offset)
        self.material = material or Material()

    def params(self):
        return [self.iterations, self.scale, *self.angles, self.offset, 0.0]

    def custom_shader_parameters(self):
        return {
            "fractalIterations": {"type": "f", "count": 1, "data": [self.iterations]},
            "scaleFactor": {"type": "f", "count": 1, "data": [self.scale]},
            "angles": {"type": "f", "count": 3, "data": list(self.angles)},
            "offset": {"type": "f", "count": 1, "data": [self.offset]},
        }
