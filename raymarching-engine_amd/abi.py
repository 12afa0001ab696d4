"""ctypes mirror of include/hip_raymarch.h (POD structs and constants only)."""
from __future__ import annotations

import ctypes as C

RM_ABI_VERSION = 8
RM_MAX_BOUNCES = 10
RM_MAX_LIGHTS = 10
RM_MAX_PRIMS = 256

RM_OK, RM_ERR_INVALID, RM_ERR_DEVICE, RM_ERR_NO_DEVICE = 0, 1, 2, 3

(RM_SCENE_TABLE, RM_SCENE_MANDELBULB, RM_SCENE_SPHERE_GRID, RM_SCENE_SPHERE_LATTICE,
 RM_SCENE_MENGER, RM_SCENE_KIFS_TREE, RM_SCENE_KIFS_BOX) = range(7)
RM_PRIM_SPHERE, RM_PRIM_BOX, RM_PRIM_REPEAT, RM_PRIM_FOLD, RM_PRIM_KIND, RM_PRIM_TORUS, RM_PRIM_CYLINDER, RM_PRIM_PLANE = 0, 1, 2, 3, 4, 5, 6, 7
RM_OP_UNION, RM_OP_SMOOTH_UNION, RM_OP_SUBTRACT, RM_OP_INTERSECT, RM_OP_SMOOTH_SUBTRACT, RM_OP_SMOOTH_INTERSECT = 0, 1, 2, 3, 4, 5

RM_RENDER_STRICT, RM_RENDER_FAST, RM_RENDER_COLOR_ONLY, RM_RENDER_MEGAKERNEL, RM_RENDER_NO_COST_CLASSES, RM_RENDER_WAVEFRONT = 0, 1, 2, 4, 8, 16
RM_RENDER_NO_OVERLAP = 32
RM_RENDER_NO_FAR_JUMP = 64
RM_RENDER_NO_CULL = 128
RM_PIPELINE_NONE, RM_PIPELINE_PIXEL_KERNEL, RM_PIPELINE_WAVEFRONT = 0, 1, 2
RM_PLANE_COLOR, RM_PLANE_NORMAL_DOF, RM_PLANE_ALBEDO_DEPTH = 0, 1, 2
RM_PROBE_SDF, RM_PROBE_CAST_RAY, RM_PROBE_NORMAL, RM_PROBE_MATERIAL, RM_PROBE_CAST_STEPS, RM_PROBE_CAST_SHADOW = 0, 1, 2, 3, 4, 5
RM_MATH_FUNCTIONS = ("sin", "cos", "log", "exp", "pow", "acos", "atan2", "tan", "pow_pair_nm1", "pow_pair_n", "sincos_s", "sincos_c", "sqrt", "div")  # RM_MATH_*


class RmUniforms(C.Structure):
    _fields_ = [
        ("blendWithPreviousFactor", C.c_float),
        ("randNoise", C.c_float * 2),
        ("position", C.c_float * 3),
        ("rotation", C.c_float * 16),
        ("dofAmount", C.c_float),
        ("dofFocalPlaneDistance", C.c_float),
        ("cameraMode", C.c_int32),
        ("fov", C.c_float),
        ("reflections", C.c_float),
        ("raymarchingSteps", C.c_float),
        ("indirectLightingRaymarchingSteps", C.c_float),
        ("aspect", C.c_float),
        ("fogDensity", C.c_float),
        ("exposure", C.c_float),
        ("raymarchingStepCountsArray", C.c_float * RM_MAX_BOUNCES),
        ("blendMode", C.c_int32),
        ("renderMode", C.c_int32),
        ("lightPositions", (C.c_float * 3) * RM_MAX_LIGHTS),
        ("lightColors", (C.c_float * 3) * RM_MAX_LIGHTS),
        ("lightSizes", C.c_float * RM_MAX_LIGHTS),
        ("lightCount", C.c_int32),
        ("showDofFocalPlane", C.c_int32),
    ]


class RmPrim(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("k", C.c_float),
        ("center", C.c_float * 3),
        ("size", C.c_float * 3),
    ]


class RmMaterial(C.Structure):
    _fields_ = [
        ("diffuse", C.c_float * 3),
        ("diffuse_cutoff", C.c_float),
        ("specular", C.c_float * 3),
        ("specular_cutoff", C.c_float),
        ("roughness", C.c_float),
        ("subsurface", C.c_float),
        ("subsurface_color", C.c_float * 3),
        ("ior", C.c_float),
        ("sky_color", C.c_float * 3),
        ("sky_floor", C.c_float),
        ("sky_scale", C.c_float),
        ("sky_radius", C.c_float),
        ("sky_axis", C.c_int32),
        ("reserved", C.c_int32),
    ]


RM_MAX_SURFACES = 15


class RmSurface(C.Structure):
    _fields_ = [
        ("diffuse", C.c_float * 3),
        ("roughness", C.c_float),
        ("specular", C.c_float * 3),
        ("subsurface", C.c_float),
        ("subsurface_color", C.c_float * 3),
        ("ior", C.c_float),
    ]


class RmSceneDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("nprims", C.c_int32),
        ("prims", C.POINTER(RmPrim)),
        ("params", C.c_float * 16),
        ("material", RmMaterial),
        ("nsurfaces", C.c_int32),
        ("reserved", C.c_int32),
        ("surfaces", C.POINTER(RmSurface)),
    ]


class RmRect(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


assert C.sizeof(RmPrim) == 32 and C.sizeof(RmSurface) == 48
