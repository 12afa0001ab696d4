"""ctypes binding of libhip_raymarch.so (include/hip_raymarch.h).

The library is the product's compute path.  If it is missing this module
raises: there is no Python or CPU fallback for rendering.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional

import numpy as np

from . import abi
from .scene import Scene

import os

# Process environment the device path depends on, set where the library is loaded so that EVERY host gets it (bench.py,
# the job API, the tests), unless the user has set it.  Both are read by the HIP / HSA runtime when it starts, so they
# have to be in place before anything in this process touches the GPU:
#  * GPU_MAX_HW_QUEUES=8: the runtime deals a process's streams over 4 hardware queues by default, and streams that share
#    one serialise -- with samples in flight (side streams), torch's stream and RCCL's, a sharded rank's renders stopped
#    overlapping (0.59 instead of 0.42 ms per sample at 8 GPUs).  rm_ctx_set_samples_in_flight warns when it is lower.
#  * HSA_ENABLE_IPC_MODE_LEGACY=0: dmabuf IPC, which RCCL needs across processes on this driver.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# RM_LIB selects an experiment build of the SAME library (tools/); default = the in-tree product
LIB_PATH = Path(os.environ.get("RM_LIB") or Path(__file__).resolve().parent / "libhip_raymarch.so")

# every symbol include/hip_raymarch.h declares
EXPORTS = [
    "rm_abi_version", "rm_material_default", "rm_ctx_create", "rm_ctx_destroy", "rm_last_error", "rm_ctx_set_stream",
    "rm_ctx_set_retire_eps", "rm_ctx_set_samples_in_flight", "rm_ctx_last_warning", "rm_ctx_set_cost_order", "rm_ctx_set_cull_min_pixels", "rm_ctx_set_cull_budget", "rm_ctx_cull_stats", "rm_debug_counters", "rm_device_memory", "rm_sync", "rm_scene_create", "rm_scene_destroy", "rm_fb_create", "rm_fb_create_striped", "rm_fb_rows", "rm_fb_width", "rm_fb_height", "rm_fb_wrap", "rm_fb_clear", "rm_fb_destroy",
    "rm_fb_download", "rm_fb_upload", "rm_fb_device_ptr", "rm_buffer_create", "rm_buffer_destroy", "rm_buffer_download", "rm_buffer_upload", "rm_render_sample", "rm_render_samples", "rm_ctx_set_sample_batch", "rm_ctx_set_gl_stack", "rm_render_timed",
    "rm_probe", "rm_probe_camera", "rm_probe_rng", "rm_probe_math", "rm_assemble_striped", "rm_assemble_striped_bytes", "rm_present", "rm_present_planes", "rm_present_device", "rm_present_rows", "rm_pack_present_rows", "rm_ctx_last_pipeline", "rm_present_sharded", "rm_present_sharded_start", "rm_present_sharded_finish", "rm_present_striped_rows", "rm_debug_cull_cell",
]


def cull_cell(scene, centre, radius: float, margin: float = 0.0):
    """Rows of a primitive table that an evaluation anywhere in the ball (centre, radius) has to fold: a bool per row
    (rm_debug_cull_cell -- the rule behind the culling grid of the fast build; host arithmetic, no GPU)."""
    lib = load_library()
    desc = scene.desc()
    words = (desc.nprims + 63) // 64
    out = (C.c_ulonglong * words)()
    c = (C.c_double * 3)(*[float(v) for v in centre])
    rc = lib.rm_debug_cull_cell(C.byref(desc), c, float(radius), float(margin), out)
    if rc != abi.RM_OK:
        raise RmError(rc, "rm_debug_cull_cell: not a table of shapes only")
    return [bool((out[i >> 6] >> (i & 63)) & 1) for i in range(desc.nprims)]


class RmError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


_lib = None
_libs: dict = {}
# the tests' cross-check build of the same sources (build.py build_crosscheck): the library plus the wavefront pipeline.  NOT what a
# host loads: Context(library=XCHECK_LIB_PATH) is for tests/ and the tools that time both implementations.
XCHECK_LIB_PATH = Path(__file__).resolve().parent.parent / "tests" / "_xcheck" / "libhip_raymarch_xcheck.so"


def _share_torch_hip_runtime():
    """One HIP/HSA runtime per process.  PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 (same sonames as
    /opt/rocm's); if this library pulls in /opt/rocm's copies first and torch initialises its own afterwards, the
    second HSA runtime finds no GPU ("No HIP GPUs are available").  When torch is installed, load ITS runtime first:
    libhip_raymarch.so then binds to it by soname, whichever of the two is imported first.  Hosts without torch (the
    Node addon) use /opt/rocm's."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return  # torch's runtime is already in the process
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = Path(list(spec.submodule_search_locations)[0]) / "lib"
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        p = libdir / name
        if p.exists():
            try:
                C.CDLL(str(p), mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load_library(path=None):
    """dlopen the HIP library (loading needs no GPU; creating a context does).  `path`: another build of the same library (the
    tests' cross-check build); default = the product."""
    global _lib
    path = Path(path) if path is not None else LIB_PATH
    if str(path) in _libs:
        return _libs[str(path)]
    if not path.exists():
        raise RmError(abi.RM_ERR_DEVICE, f"{path} is missing: build it with `python raymarching-engine_amd/build.py` "
                                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _share_torch_hip_runtime()
    lib = C.CDLL(str(path))
    vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.c_int
    sig = {
        "rm_abi_version": (ip, []),
        "rm_material_default": (None, [C.POINTER(abi.RmMaterial)]),
        "rm_ctx_create": (ip, [ip, C.POINTER(vp)]),
        "rm_ctx_destroy": (None, [vp]),
        "rm_last_error": (C.c_char_p, [vp]),
        "rm_ctx_set_stream": (ip, [vp, vp]),
        "rm_ctx_set_retire_eps": (ip, [vp, C.c_float]),
        "rm_ctx_set_samples_in_flight": (ip, [vp, C.c_int]),
        "rm_ctx_last_warning": (C.c_char_p, [vp]),
        "rm_ctx_set_sample_batch": (ip, [vp, C.c_int]),
        "rm_ctx_set_gl_stack": (ip, [vp, C.c_int]),
        "rm_device_memory": (ip, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "rm_buffer_create": (ip, [vp, C.c_size_t, C.POINTER(C.c_void_p)]),
        "rm_buffer_destroy": (ip, [vp, vp]),
        "rm_buffer_download": (ip, [vp, vp, vp, C.c_size_t]),
        "rm_buffer_upload": (ip, [vp, vp, vp, C.c_size_t]),
        "rm_ctx_set_cost_order": (ip, [vp, C.c_int]),
        "rm_ctx_set_cull_min_pixels": (ip, [vp, C.c_longlong]),
        "rm_ctx_set_cull_budget": (ip, [vp, C.c_size_t]),
        "rm_ctx_cull_stats": (ip, [vp, C.POINTER(C.c_ulonglong)]),
        "rm_debug_counters": (ip, [vp, C.POINTER(C.c_ulonglong), ip]),
        "rm_debug_cull_cell": (ip, [C.POINTER(abi.RmSceneDesc), C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(C.c_ulonglong)]),
        "rm_sync": (ip, [vp]),
        "rm_scene_create": (ip, [vp, C.POINTER(abi.RmSceneDesc), C.POINTER(vp)]),
        "rm_scene_destroy": (None, [vp]),
        "rm_fb_create": (ip, [vp, ip, ip, ip, ip, C.POINTER(vp)]),
        "rm_fb_wrap": (ip, [vp, ip, ip, ip, ip, vp, vp, vp, C.POINTER(vp)]),
        "rm_fb_create_striped": (ip, [vp, ip, ip, ip, ip, ip, vp, vp, vp, C.POINTER(vp)]),
        "rm_fb_rows": (ip, [vp]),
        "rm_fb_width": (ip, [vp]),
        "rm_fb_height": (ip, [vp]),
        "rm_fb_clear": (ip, [vp]),
        "rm_fb_destroy": (None, [vp]),
        "rm_fb_download": (ip, [vp, ip, fp]),
        "rm_fb_upload": (ip, [vp, ip, fp]),
        "rm_fb_device_ptr": (vp, [vp, ip]),
        "rm_render_sample": (ip, [vp, vp, vp, C.POINTER(abi.RmUniforms), C.POINTER(abi.RmRect), ip]),
        "rm_render_samples": (ip, [vp, vp, vp, C.POINTER(abi.RmUniforms), fp, ip, C.POINTER(abi.RmRect), ip]),
        "rm_render_timed": (ip, [vp, vp, vp, C.POINTER(abi.RmUniforms), ip, C.POINTER(abi.RmRect), ip, fp]),
        "rm_probe": (ip, [vp, vp, ip, fp, ip, C.c_float, ip, fp]),
        "rm_probe_camera": (ip, [vp, C.POINTER(abi.RmUniforms), ip, ip, fp]),
        "rm_probe_rng": (ip, [vp, C.POINTER(abi.RmUniforms), ip, ip, ip, fp]),
        "rm_probe_math": (ip, [vp, ip, fp, fp, ip, fp]),
        "rm_assemble_striped": (ip, [vp, vp, ip, ip, ip, ip, ip, vp, vp]),
        "rm_assemble_striped_bytes": (ip, [vp, vp, ip, ip, C.c_longlong, ip, ip, vp, vp]),
        "rm_present_device": (ip, [vp, vp, vp, ip, ip, ip, vp, vp]),
        "rm_present_rows": (ip, [vp, vp, ip, vp, vp]),
        "rm_pack_present_rows": (ip, [vp, vp, vp, vp]),
        "rm_ctx_last_pipeline": (ip, [vp]),
        "rm_present_sharded": (ip, [C.POINTER(vp), C.POINTER(vp), ip, ip, ip, C.POINTER(C.c_uint8), C.c_size_t]),
        "rm_present_sharded_start": (ip, [C.POINTER(vp), C.POINTER(vp), ip, ip, ip]),
        "rm_present_sharded_finish": (ip, [C.POINTER(vp), ip, C.POINTER(C.c_uint8), C.c_size_t]),
        "rm_present_striped_rows": (ip, [vp, vp, vp, ip, ip, ip, ip, ip, ip, vp, vp]),
        "rm_present": (ip, [vp, vp, ip, C.POINTER(C.c_uint8)]),
        "rm_present_planes": (ip, [vp, vp, vp, ip, ip, ip, C.POINTER(C.c_uint8)]),
    }
    for name, (res, args) in sig.items():
        if name.startswith("rm_debug_") and not hasattr(lib, name) and os.environ.get("RM_LIB"):
            continue  # an experiment build of an older source tree (tools/): the debug entries are not part of what it measures
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _libs[str(path)] = lib
    if path == LIB_PATH:
        _lib = lib
    return lib


def present_sharded(contexts, framebuffers, samples: int, dof: bool) -> np.ndarray:
    """rm_present_sharded: the canvas [H, W, 4] of a frame whose stripes ONE process renders on several contexts (part p of
    len(contexts) on contexts[p]); the rows travel to contexts[0]'s GPU by peer copies."""
    n = len(contexts)
    assert n == len(framebuffers) and n >= 1
    cs = (C.c_void_p * n)(*[c.h for c in contexts])
    fs = (C.c_void_p * n)(*[f.h for f in framebuffers])
    out = np.empty((framebuffers[0].height, framebuffers[0].width, 4), np.uint8)
    contexts[0]._check(contexts[0].lib.rm_present_sharded(cs, fs, n, int(samples), 1 if dof else 0, out.ctypes.data_as(C.POINTER(C.c_uint8)), out.nbytes))
    return out


def present_sharded_start(contexts, framebuffers, samples: int, dof: bool):
    """rm_present_sharded_start: snapshot and send; returns at once (the next samples can be handed out while the frame travels)."""
    n = len(contexts)
    assert n == len(framebuffers) and n >= 1
    cs = (C.c_void_p * n)(*[c.h for c in contexts])
    fs = (C.c_void_p * n)(*[f.h for f in framebuffers])
    contexts[0]._check(contexts[0].lib.rm_present_sharded_start(cs, fs, n, int(samples), 1 if dof else 0))


def present_sharded_finish(contexts, width: int, height: int) -> np.ndarray:
    """rm_present_sharded_finish: the canvas [H, W, 4] of the present that was started."""
    n = len(contexts)
    cs = (C.c_void_p * n)(*[c.h for c in contexts])
    out = np.empty((height, width, 4), np.uint8)
    contexts[0]._check(contexts[0].lib.rm_present_sharded_finish(cs, n, out.ctypes.data_as(C.POINTER(C.c_uint8)), out.nbytes))
    return out


def scene_key(scene: Scene) -> bytes:
    """Cache key of a scene description (the reference keys its program cache
    by the source string, ShaderCache.tsx:91-119)."""
    d = scene.desc()
    prims = bytes(C.string_at(d.prims, C.sizeof(abi.RmPrim) * d.nprims)) if d.nprims else b""
    surfaces = bytes(C.string_at(d.surfaces, C.sizeof(abi.RmSurface) * d.nsurfaces)) if d.nsurfaces else b""
    return bytes([d.kind & 0xFF]) + bytes(d.params) + bytes(d.material) + prims + surfaces


def _fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


class DeviceBuffer:
    """rm_buffer_*: `nbytes` of device memory owned by the library; .ptr is the device address."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        ctx._check(ctx.lib.rm_buffer_create(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def download(self, dtype=np.uint8) -> np.ndarray:
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype)
        self.ctx._check(self.ctx.lib.rm_buffer_download(self.ctx.h, self.ptr, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def upload(self, a: np.ndarray):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.rm_buffer_upload(self.ctx.h, self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def destroy(self):
        if self.ptr:
            self.ctx._check(self.ctx.lib.rm_buffer_destroy(self.ctx.h, self.ptr))
            self.ptr = None


class Context:
    gl_stack_on = False

    def __init__(self, device: int = 0, library=None):
        self.lib = load_library(library)
        h = C.c_void_p()
        rc = self.lib.rm_ctx_create(device, C.byref(h))
        if rc != abi.RM_OK:
            raise RmError(rc, self.lib.rm_last_error(None).decode())
        self.h = h
        self.device = device
        self.settings = {}  # setter name -> arguments, as last called: what a second context needs to behave like this one (tests)

    def _check(self, rc: int):
        if rc != abi.RM_OK:
            raise RmError(rc, self.lib.rm_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.rm_ctx_destroy(self.h)
            self.h = None

    def set_stream(self, hip_stream: Optional[int]):
        """A hipStream_t (as an integer) for all later launches; None = the context's own stream.  torch's DEFAULT
        stream has the handle 0 (the NULL stream), which cannot be told from None here and with which the context's
        own non-blocking stream is not ordered: give torch work that shares buffers with the renders a stream of its own
        (torch.cuda.Stream) and pass that one."""
        if hip_stream is not None and int(hip_stream) == 0:
            raise RmError(abi.RM_ERR_INVALID, "set_stream(0): the NULL stream is not supported; use a torch.cuda.Stream() "
                                              "(or None for the context's own stream)")
        self._check(self.lib.rm_ctx_set_stream(self.h, C.c_void_p(hip_stream or 0)))

    def set_samples_in_flight(self, n: int):
        """Consecutive full-mode samples that may overlap on the GPU (1 = none); the planes get the same bits."""
        self.settings["set_samples_in_flight"] = (n,)
        self._check(self.lib.rm_ctx_set_samples_in_flight(self.h, int(n)))
        note = self.lib.rm_ctx_last_warning(self.h).decode()
        if note.startswith("warning:"):  # accepted, but the process environment will keep the samples from overlapping
            import warnings

            warnings.warn(note)

    def device_memory(self):
        """(free, total) bytes of the context's GPU."""
        f, t = C.c_size_t(), C.c_size_t()
        self._check(self.lib.rm_device_memory(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def buffer(self, nbytes: int) -> "DeviceBuffer":
        """Zero-filled raw device memory (for rm_present_rows / rm_assemble_striped_bytes on hosts without torch)."""
        return DeviceBuffer(self, nbytes)

    def set_gl_stack(self, on: bool):
        """Parity mode: strict-flag renders, probes and presents in the arithmetic of the GL stack the goldens were
        rendered under (include/hip_raymarch.h rm_ctx_set_gl_stack)."""
        self._check(self.lib.rm_ctx_set_gl_stack(self.h, int(on)))  # True / 1: on; 2: on, with the stack's own tan
        self.gl_stack_on = bool(on)

    def set_sample_batch(self, n: int):
        """Samples per launch of render_samples (0 = automatic, 1 = one launch per sample, up to 8); same bits."""
        self.settings["set_sample_batch"] = (n,)
        self._check(self.lib.rm_ctx_set_sample_batch(self.h, int(n)))

    def assemble_striped(self, src_ptr: int, parts: int, max_rows: int, width: int, height: int, stripe_rows: int, dst_ptr: int,
                         stream: Optional[int] = None):
        """Gathered striped windows (device pointer, parts x max_rows x width float4) -> the frame in image order,
        on `stream` (a hipStream_t as an integer) or the context's stream."""
        self._check(self.lib.rm_assemble_striped(self.h, C.c_void_p(src_ptr), parts, max_rows, width, height, stripe_rows, C.c_void_p(dst_ptr),
                                                 C.c_void_p(stream) if stream else None))

    def assemble_striped_bytes(self, src_ptr: int, parts: int, max_rows: int, row_bytes: int, height: int, stripe_rows: int, dst_ptr: int,
                               stream: Optional[int] = None):
        """assemble_striped for rows of opaque bytes (RGBA8 after present_rows: row_bytes = 4 * width)."""
        self._check(self.lib.rm_assemble_striped_bytes(self.h, C.c_void_p(src_ptr), parts, max_rows, int(row_bytes), height, stripe_rows,
                                                       C.c_void_p(dst_ptr), C.c_void_p(stream) if stream else None))

    def present_device(self, color_ptr: int, normal_dof_ptr: Optional[int], width: int, height: int, samples: int, out_ptr: int,
                       stream: Optional[int] = None):
        """display.frag into DEVICE memory (height*width*4 bytes), asynchronous."""
        self._check(self.lib.rm_present_device(self.h, C.c_void_p(color_ptr), C.c_void_p(normal_dof_ptr or 0), width, height, int(samples),
                                               C.c_void_p(out_ptr), C.c_void_p(stream) if stream else None))

    def present_striped_rows(self, color_ptr: int, normal_dof_ptr: Optional[int], width: int, height: int, samples: int, stripe_rows: int, parts: int,
                             part: int, out_ptr: int, stream: Optional[int] = None):
        """display.frag for the rows part `part` of `parts` holds, read from the WHOLE frame in DEVICE memory, into DEVICE memory
        (that part's packed rows x width x 4 bytes), asynchronous: each rank's share of a sharded frame's blur."""
        self._check(self.lib.rm_present_striped_rows(self.h, C.c_void_p(color_ptr), C.c_void_p(normal_dof_ptr or 0), width, height, int(samples), stripe_rows,
                                                     parts, part, C.c_void_p(out_ptr), C.c_void_p(stream) if stream else None))

    def present_rows(self, fb: "Framebuffer", samples: int, out_ptr: int, stream: Optional[int] = None):
        """Tone-map the rows `fb` holds (no depth of field) into DEVICE memory (rows*width*4 bytes), asynchronous."""
        self._check(self.lib.rm_present_rows(self.h, fb.h, int(samples), C.c_void_p(out_ptr), C.c_void_p(stream) if stream else None))

    def pack_present_rows(self, fb: "Framebuffer", out_ptr: int, stream: Optional[int] = None):
        """(colour.rgb, normal_dof.w) of the rows `fb` holds into DEVICE memory (rows*width float4), asynchronous: what a
        sharded job with depth of field gathers; assembled, it is both planes of present_device."""
        self._check(self.lib.rm_pack_present_rows(self.h, fb.h, C.c_void_p(out_ptr), C.c_void_p(stream) if stream else None))

    def last_pipeline(self) -> str:
        """Which implementation the last render call dispatched: "megakernel" (the pixel kernel) or "wavefront"."""
        return {abi.RM_PIPELINE_PIXEL_KERNEL: "megakernel", abi.RM_PIPELINE_WAVEFRONT: "wavefront"}.get(int(self.lib.rm_ctx_last_pipeline(self.h)), "none")

    def set_cost_order(self, on: bool):
        """Start the tiles of a job most-expensive-first by their cost in the previous sample (scheduling only)."""
        self.settings["set_cost_order"] = (on,)
        self._check(self.lib.rm_ctx_set_cost_order(self.h, 1 if on else 0))

    def set_cull_min_pixels(self, pixels: int):
        """Pixel-samples a scene has to be asked for before its culling grid is built (0: with the first render; same bits)."""
        self.settings["set_cull_min_pixels"] = (pixels,)
        self._check(self.lib.rm_ctx_set_cull_min_pixels(self.h, int(pixels)))

    def set_cull_budget(self, nbytes: int):
        """The bytes this context's culling grids may hold together (default: a sixteenth of the device's memory, at most 1 GiB)."""
        self.settings["set_cull_budget"] = (nbytes,)
        self._check(self.lib.rm_ctx_set_cull_budget(self.h, int(nbytes)))

    def cull_stats(self) -> dict:
        out = (C.c_ulonglong * 4)()
        self._check(self.lib.rm_ctx_cull_stats(self.h, out))
        return dict(built=int(out[0]), bytes=int(out[1]), grids=int(out[2]), budget=int(out[3]))

    def set_retire_eps(self, eps: float):
        self.settings["set_retire_eps"] = (eps,)
        self._check(self.lib.rm_ctx_set_retire_eps(self.h, float(eps)))

    def debug_counters(self, reset: bool = True):
        out = (C.c_ulonglong * 16)()
        self._check(self.lib.rm_debug_counters(self.h, out, 1 if reset else 0))
        return list(out)

    def sync(self):
        self._check(self.lib.rm_sync(self.h))

    def create_scene(self, scene: Scene) -> "SceneHandle":
        return SceneHandle(self, scene)

    def create_framebuffer(self, width: int, height: int, row_begin: int = 0, row_count: Optional[int] = None) -> "Framebuffer":
        return Framebuffer(self, width, height, row_begin, height if row_count is None else row_count)

    def create_striped_framebuffer(self, width, height, stripe_rows, parts, part, color_ptr=None, normal_ptr=None, albedo_ptr=None) -> "Framebuffer":
        """Rows r with (r // stripe_rows) % parts == part, packed (row sharding across GPUs)."""
        return Framebuffer(self, width, height, 0, 0, striped=(stripe_rows, parts, part, color_ptr, normal_ptr, albedo_ptr))

    def wrap_framebuffer(self, width, height, row_begin, row_count, color_ptr, normal_ptr=None, albedo_ptr=None) -> "Framebuffer":
        return Framebuffer(self, width, height, row_begin, row_count, wrap=(color_ptr, normal_ptr, albedo_ptr))

    def render_sample(self, scene: "SceneHandle", fb: "Framebuffer", uniforms: abi.RmUniforms, tile: Optional[abi.RmRect] = None,
                      flags: int = abi.RM_RENDER_STRICT):
        self._check(self.lib.rm_render_sample(self.h, scene.h, fb.h, C.byref(uniforms), C.byref(tile) if tile is not None else None, flags))

    def render_samples(self, scene, fb, uniforms, rand_noise_pairs, tile=None, flags=abi.RM_RENDER_STRICT):
        rn = np.ascontiguousarray(rand_noise_pairs, np.float32).reshape(-1, 2)
        self._check(self.lib.rm_render_samples(self.h, scene.h, fb.h, C.byref(uniforms), _fp(rn), len(rn),
                                               C.byref(tile) if tile is not None else None, flags))

    def render_timed(self, scene, fb, uniforms, count: int, tile=None, flags=abi.RM_RENDER_STRICT) -> float:
        ms = C.c_float(0.0)
        self._check(self.lib.rm_render_timed(self.h, scene.h, fb.h, C.byref(uniforms), count,
                                             C.byref(tile) if tile is not None else None, flags, C.byref(ms)))
        return float(ms.value)

    def probe(self, scene: "SceneHandle", what: int, inputs: np.ndarray, param: float = 0.0, flags: int = abi.RM_RENDER_STRICT) -> np.ndarray:
        in_w = {abi.RM_PROBE_SDF: 3, abi.RM_PROBE_CAST_RAY: 6, abi.RM_PROBE_NORMAL: 3, abi.RM_PROBE_MATERIAL: 3, abi.RM_PROBE_CAST_STEPS: 6, abi.RM_PROBE_CAST_SHADOW: 9}[what]
        out_w = {abi.RM_PROBE_SDF: 1, abi.RM_PROBE_CAST_RAY: 3, abi.RM_PROBE_NORMAL: 3, abi.RM_PROBE_MATERIAL: 12, abi.RM_PROBE_CAST_STEPS: 1, abi.RM_PROBE_CAST_SHADOW: 1}[what]
        a = np.ascontiguousarray(inputs, np.float32).reshape(-1, in_w)
        out = np.empty((len(a), out_w), np.float32)
        self._check(self.lib.rm_probe(self.h, scene.h, what, _fp(a), len(a), float(param), flags, _fp(out)))
        return out[:, 0] if out_w == 1 else out

    def probe_camera(self, uniforms: abi.RmUniforms, width: int, height: int) -> np.ndarray:
        out = np.empty((height, width, 8), np.float32)
        self._check(self.lib.rm_probe_camera(self.h, C.byref(uniforms), width, height, _fp(out)))
        return out

    def probe_rng(self, uniforms: abi.RmUniforms, width: int, height: int, count: int) -> np.ndarray:
        out = np.empty((height, width, count), np.float32)
        self._check(self.lib.rm_probe_rng(self.h, C.byref(uniforms), width, height, count, _fp(out)))
        return out

    def probe_math(self, name: str, a: np.ndarray, b: np.ndarray = None) -> np.ndarray:
        """One transcendental of this context's parity arithmetic on an array (rm_probe_math; name: abi.RM_MATH_FUNCTIONS)."""
        a = np.ascontiguousarray(a, np.float32).ravel()
        b = np.ascontiguousarray(b, np.float32).ravel() if b is not None else None
        out = np.empty_like(a)
        self._check(self.lib.rm_probe_math(self.h, abi.RM_MATH_FUNCTIONS.index(name), _fp(a), _fp(b) if b is not None else None, a.size, _fp(out)))
        return out


class SceneHandle:
    def __init__(self, ctx: Context, scene: Scene):
        self.ctx = ctx
        desc = scene.desc()
        h = C.c_void_p()
        ctx._check(ctx.lib.rm_scene_create(ctx.h, C.byref(desc), C.byref(h)))
        self.h = h

    def destroy(self):
        if self.h:
            self.ctx.lib.rm_scene_destroy(self.h)
            self.h = None


class Framebuffer:
    def __init__(self, ctx: Context, width, height, row_begin, row_count, wrap=None, striped=None):
        self.ctx = ctx
        self.width, self.height, self.row_begin, self.row_count = width, height, row_begin, row_count
        h = C.c_void_p()
        if striped is not None:
            s, n, r, c0, c1, c2 = striped
            ctx._check(ctx.lib.rm_fb_create_striped(ctx.h, width, height, s, n, r, C.c_void_p(c0 or 0), C.c_void_p(c1 or 0),
                                                    C.c_void_p(c2 or 0), C.byref(h)))
            self.row_count = int(ctx.lib.rm_fb_rows(h))
        elif wrap is None:
            ctx._check(ctx.lib.rm_fb_create(ctx.h, width, height, row_begin, row_count, C.byref(h)))
        else:
            ctx._check(ctx.lib.rm_fb_wrap(ctx.h, width, height, row_begin, row_count, C.c_void_p(wrap[0]),
                                          C.c_void_p(wrap[1] or 0), C.c_void_p(wrap[2] or 0), C.byref(h)))
        self.h = h

    def clear(self):
        self.ctx._check(self.ctx.lib.rm_fb_clear(self.h))

    def destroy(self):
        if self.h:
            self.ctx.lib.rm_fb_destroy(self.h)
            self.h = None

    def download(self, plane: int = abi.RM_PLANE_COLOR) -> np.ndarray:
        out = np.empty((self.row_count, self.width, 4), np.float32)
        self.ctx._check(self.ctx.lib.rm_fb_download(self.h, plane, _fp(out)))
        return out

    def upload(self, plane: int, data: np.ndarray):
        a = np.ascontiguousarray(data, np.float32)
        assert a.shape == (self.row_count, self.width, 4)
        self.ctx._check(self.ctx.lib.rm_fb_upload(self.h, plane, _fp(a)))

    def device_ptr(self, plane: int = abi.RM_PLANE_COLOR) -> int:
        return int(self.ctx.lib.rm_fb_device_ptr(self.h, plane) or 0)

    def present(self, samples: int) -> np.ndarray:
        """Tone-mapped RGBA8 image of the whole frame (display.frag:16-64), row 0 = bottom."""
        out = np.empty((self.row_count, self.width, 4), np.uint8)
        self.ctx._check(self.ctx.lib.rm_present(self.ctx.h, self.h, int(samples), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out
