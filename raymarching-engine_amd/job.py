"""Host side of the render-job API: the reference's ``doRenderJob`` boundary.

Mirrors client/src/renderer/RenderJobExecutor.tsx:77-341 and the data model of
client/src/renderer/RenderJobSchema.tsx:17-86 (same field names, as plain
dicts) -- the ``gl.*`` block of the reference (RenderJobExecutor.tsx:181-326)
becomes one native call per sample.

The job description is the reference's, with one addition: ``sdfScene`` (a
``scene.Scene``) next to ``sdfShaderSource`` -- a HIP kernel cannot consume
GLSL text.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Callable, Dict, Generator, Iterable, Optional, Tuple

from . import abi
from .scene import Scene


def halton(base: int) -> Generator[float, None, None]:
    """Radical-inverse sequence, value for value the generator of
    client/src/util/Halton.tsx:1-19 (0.5, 0.25, 0.75, ... / 1/3, 2/3, 1/9, ...):
    integer numerator and denominator, one division."""
    i = 0
    while True:
        i += 1
        num, den, k = 0, 1, i
        while k > 0:
            num = num * base + (k % base)
            den *= base
            k //= base
        yield num / den


# the reference keeps ONE pair of generators for the life of the page
# (RenderJobExecutor.tsx:70-71), so the sequence continues across jobs
_render_job_halton2 = halton(2)
_render_job_halton3 = halton(3)


def reset_halton() -> None:
    global _render_job_halton2, _render_job_halton3
    _render_job_halton2, _render_job_halton3 = halton(2), halton(3)


def next_rand_noise() -> Tuple[float, float]:
    return next(_render_job_halton2), next(_render_job_halton3)


IDENTITY = [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]


def make_schema(
    scene: Optional[Scene] = None,
    width: int = 256,
    height: int = 256,
    counts: Iterable[float] = (128,),
    render_mode: str = "preview",
    position=(0.0, 0.0, -3.0),
    rotation=None,
    fov: float = 1.5,
    camera: str = "perspective",
    lights=(),
    exposure: float = 0.5,
    samples_per_pixel: int = 1,
    blend_mode: str = "additive",
    blend_factor: float = 0.9,
    fog_density: float = 0.0,
    dof_amount: float = 0.0,
    dof_distance: float = 1.5,
    show_focused_area: bool = False,
    subdivisions: int = 1,
    frameid: int = 0,
    sample_yield_interval: int = 1,
    sdf_shader_source: Optional[str] = None,
) -> dict:
    """A RenderJobSchema with the live defaults of client/src/index.tsx:121-182,308-333."""
    if camera == "perspective":
        mode = {"type": "perspective", "fov": fov}
    elif camera == "orthographic":
        mode = {"type": "orthographic", "size": fov}
    else:
        mode = {"type": "panoramic", "angleX": 0.0, "angleY": 0.0}
    if sdf_shader_source is None and scene is not None:
        try:
            sdf_shader_source = scene.glsl()
        except NotImplementedError:
            sdf_shader_source = ""
    return {
        "reflectionIterationCounts": [float(c) for c in counts],
        "normalDelta": 1e-5,
        "sdfShaderSource": sdf_shader_source or "",
        "sdfScene": scene,
        "customShaderParameters": scene.custom_shader_parameters() if scene is not None else {},
        "fogDensity": fog_density,
        "time": 0.0,
        "timeDelta": 0.0,
        "dof": {"amount": dof_amount, "distance": dof_distance, "showFocusedArea": show_focused_area},
        "camera": {
            "position": list(position),
            "motion": [0.0, 0.0, 0.0],
            "rotation": list(rotation) if rotation is not None else list(IDENTITY),
            "mode": mode,
        },
        "render": {
            "samplesPerPixel": samples_per_pixel,
            "exposure": exposure,
            "subdivisions": subdivisions,
            "width": width,
            "height": height,
            "frameid": frameid,
            "blendWithPreviousFrameFactor": blend_factor,
            "sampleYieldInterval": sample_yield_interval,
            "blendMode": blend_mode,
            "renderMode": render_mode,
        },
        "lights": [dict(l) for l in lights],
    }


def point_light(position, color=(1.0, 1.0, 1.0), strength: float = 3.0, size: float = 0.0) -> dict:
    """A light as the UI hands it to the job: colour (0-255 scale 255 = white)
    times strength / 256 (client/src/index.tsx:170-181, LightSettings.tsx:82-87)."""
    return {"type": "point", "position": list(position), "color": [c * 255.0 * strength / 256.0 for c in color], "size": size}


def sun_light(direction, color=(1.0, 1.0, 1.0), strength: float = 3.0) -> dict:
    """A sun: the reference passes its `direction` where a point light has its position, with size 0
    (RenderJobExecutor.tsx:276-291) -- so it is lit like a point at that place."""
    return {"type": "sun", "direction": list(direction), "color": [c * 255.0 * strength / 256.0 for c in color]}


_REQUIRED = ("reflectionIterationCounts", "fogDensity", "dof", "camera", "render", "lights")


def uniforms_from_schema(schema: dict, rand_noise: Tuple[float, float]) -> abi.RmUniforms:
    """The uniform block for one sample, derived exactly as
    RenderJobExecutor.tsx:212-297 derives it."""
    u = abi.RmUniforms()
    cam, r = schema["camera"], schema["render"]
    counts = list(schema["reflectionIterationCounts"])
    if len(counts) > abi.RM_MAX_BOUNCES:
        raise ValueError("at most 10 bounces (raymarchingStepCountsArray[10], raymarcher.frag:31)")
    if len(schema["lights"]) > abi.RM_MAX_LIGHTS:
        raise ValueError("at most 10 lights (raymarcher.frag:37-39)")
    u.blendWithPreviousFactor = r["blendWithPreviousFrameFactor"]
    u.randNoise[0], u.randNoise[1] = rand_noise
    u.position[:] = cam["position"]
    u.rotation[:] = list(cam["rotation"])
    u.dofAmount = schema["dof"]["amount"]
    u.dofFocalPlaneDistance = schema["dof"]["distance"]
    mode = cam["mode"]
    u.cameraMode = ["perspective", "orthographic", "panoramic"].index(mode["type"])  # :228-232
    u.fov = mode["fov"] if mode["type"] == "perspective" else mode["size"] if mode["type"] == "orthographic" else 1.0  # :233-239
    u.reflections = float(len(counts))  # :241
    u.raymarchingSteps = counts[0] if counts else 0.0  # :243 (unused by the shader)
    u.indirectLightingRaymarchingSteps = counts[1] if len(counts) > 1 else (counts[0] if counts else 0.0)  # :245-248
    u.aspect = r["width"] / r["height"]  # :250
    u.fogDensity = schema["fogDensity"]
    u.exposure = r["exposure"] / r["samplesPerPixel"]  # :254-256
    for i, c in enumerate(counts):
        u.raymarchingStepCountsArray[i] = c
    u.blendMode = 1 if r["blendMode"] == "additive" else 0  # :258
    u.renderMode = 1 if r["renderMode"] == "preview" else 0  # :259
    u.lightCount = len(schema["lights"])  # :261
    u.showDofFocalPlane = 1 if schema["dof"]["showFocusedArea"] else 0  # :263
    for i, l in enumerate(schema["lights"]):  # :276-291: a sun is a point at `direction`, size 0
        p = l["position"] if l["type"] == "point" else l["direction"]
        u.lightPositions[i][:] = list(p)
        u.lightColors[i][:] = list(l["color"])
        u.lightSizes[i] = l["size"] if l["type"] == "point" else 0.0
    return u


def tile_rect(schema: dict, x_part: int, y_part: int, reference_scissor: bool = False) -> abi.RmRect:
    """Screen tile of the `subdivisions` loop (RenderJobExecutor.tsx:167-180).
    The reference hands (x1, y1, x2, y2) to gl.scissor, where GL expects (x, y, width, height) (:182): GL then
    takes x2, y2 as the SIZE of the rectangle at (x1, y1).  Clipped to the image that is exactly the intended
    tile for subdivisions 1 and 2; from 3 on the rectangles reach to the right and top edges of the image and
    overlap, so with the reference an additive render is brighter where they do.  The default here is the
    intent -- the tile rectangle; `reference_scissor=True` (job option `render.referenceScissor`) gives the
    reference's rectangle, for a host that has to reproduce upstream images of subdivided renders."""
    r = schema["render"]
    n = r["subdivisions"]
    x1 = math.floor(r["width"] / n * x_part)
    y1 = math.floor(r["height"] / n * y_part)
    x2 = math.ceil(r["width"] / n * (x_part + 1))
    y2 = math.ceil(r["height"] / n * (y_part + 1))
    if reference_scissor or r.get("referenceScissor"):
        return abi.RmRect(x1, y1, min(x2, r["width"] - x1), min(y2, r["height"] - y1))
    return abi.RmRect(x1, y1, x2 - x1, y2 - y1)


SCENE_CACHE_ENTRIES = 64  # scenes a RenderJobContext keeps (least recently used out first)


class RenderJobContext:
    """Counterpart of RenderJobContext (RenderJobExecutor.tsx:32-54) +
    loadRenderJobContext (LoadRenderJobContext.tsx:162-287): the native
    context, a scene cache keyed by scene description (programCache,
    ShaderCache.tsx:91-119 -- errors are cached too) and the framebuffer cache
    keyed (w, h, frameid) with its <= 3 entry "purgatory" of released sets.

    Sharded mode (`group`: a dist.ShardGroup -- one process per GPU, torch.distributed).  The reference's tile loop
    (RenderJobExecutor.tsx:148-182) is the precedent for cutting a job's frame; here the frame's rows are dealt to the
    ranks in 8-row stripes (shard.py), every rank runs the SAME job -- same schema, same do_render_job, same yields --
    on the stripes it holds, and what fbo.create hands out is a dist.ShardedFramebuffer: the `present` callback of every
    rank calls its collective ``present(samples)`` at every yield and rank 0 gets the assembled canvas, with and without
    depth of field (dist.py).  The context makes a torch stream of its own current for the job's device work, so that
    renders, snapshots, the collective and the assembly are ordered on one stream.
    """

    group = None   # a dist.ShardGroup when the context is sharded
    stream = None  # the torch stream a sharded context orders its device work on (GPU)

    stripes = None  # (parts, part): this context holds one part's 8-row stripes of every frame, with no group (measurement: one GPU standing in for a rank)

    def __init__(self, device: int = 0, flags: int = abi.RM_RENDER_STRICT, rows: Optional[Tuple[int, int]] = None, group=None,
                 native_context=None, stripes: Optional[Tuple[int, int]] = None):
        from . import native

        self.native = native_context if native_context is not None else native.Context(device)
        self.flags = flags
        self.rows = rows  # (row_begin, row_count) window of this GPU, None = whole image
        self.stripes = stripes
        self.group = group if (group is not None and group.sharded) else None
        self.stream = None
        if self.group is not None:
            if rows is not None:
                raise ValueError("RenderJobContext: a sharded context holds stripes, not a row window")
            import torch

            dev = torch.device(self.group.device)
            if dev.type == "cuda":
                # not torch's default stream: its handle is NULL, which the context's own non-blocking stream is not ordered with.
                # The stream is made torch's CURRENT stream of this thread -- the collective of a present is enqueued from it -- and
                # close() puts the previous one back: a host with torch work of its own scopes the context (`with ctx:` / close()).
                self._prev_stream = torch.cuda.current_stream(dev)
                self.stream = torch.cuda.Stream(device=dev)
                torch.cuda.set_stream(self.stream)
                self.native.set_stream(self.stream.cuda_stream)
        # programCache: by description; bounded (the reference's cache grows with every edit of the shader text -- a page's lifetime; a
        # long-running host that animates scene parameters would otherwise keep a device table per distinct scene; the culling grids
        # of long CSG tables -- 31.5 MB per 64 rows -- are bounded by the library itself, by bytes: include/hip_raymarch.h
        # rm_ctx_cull_stats).  Least recently used out first, its handle destroyed -- but never a scene a job still renders
        # (do_render_job pins its scene while its generator lives: jobs yield between samples).
        self._scenes: "OrderedDict[bytes, object]" = OrderedDict()
        self._pins: Dict[bytes, int] = {}
        self._live: Dict[Tuple[int, int, int], object] = {}
        self._purgatory: list = []

    # programCache.getProgram
    def get_scene(self, scene: Scene):
        from . import native

        key = native.scene_key(scene)
        hit = self._scenes.get(key)
        if hit is None:
            try:
                hit = self.native.create_scene(scene)
            except native.RmError as e:  # cached like a failed compile
                hit = {"type": "fragment", "infoLog": str(e)}
            self._scenes[key] = hit
            self._evict(keep=key)  # (never the entry being handed out: with 64 pinned scenes it is the only unpinned one)
        else:
            self._scenes.move_to_end(key)
        return hit

    def _evict(self, keep=None):
        """Least recently used out first; pinned scenes and `keep` (the key a caller is about to use) stay -- the cache exceeds its
        64 entries while everything in it is in use."""
        spare = len(self._scenes) - SCENE_CACHE_ENTRIES
        pins = self.__dict__.setdefault("_pins", {})
        for k in [k for k in self._scenes if k not in pins and k != keep][:max(spare, 0)]:
            old = self._scenes.pop(k)
            if not isinstance(old, dict) and hasattr(old, "destroy"):
                old.destroy()  # (rm_scene_destroy waits for the renders that use it)

    def pin_scene(self, scene: Scene):
        """get_scene for the lifetime of a job: the handle (or the cached error value) stays out of the cache's eviction until
        unpin_scene -- a suspended job's scene is not destroyed under it by the 64 scenes other jobs bring in."""
        from . import native

        hit = self.get_scene(scene)
        key = native.scene_key(scene)
        pins = self.__dict__.setdefault("_pins", {})
        pins[key] = pins.get(key, 0) + 1
        return key, hit

    def unpin_scene(self, key):
        pins = self.__dict__.setdefault("_pins", {})
        n = pins.get(key, 0) - 1
        if n > 0:
            pins[key] = n
        else:
            pins.pop(key, None)
            self._evict()

    def close(self):
        """Destroys the cached scenes and framebuffers and, for a sharded context on a GPU, makes the stream that was torch's
        current one before the context was made current again.  (The native context is the caller's to close when it was passed in.)"""
        for h in self._scenes.values():
            if not isinstance(h, dict) and hasattr(h, "destroy"):
                h.destroy()
        self._scenes.clear()
        for fb in list(self._live.values()) + [fb for _, fb in self._purgatory]:
            fb.destroy()
        self._live.clear()
        self._purgatory.clear()
        if self.stream is not None:
            import torch

            self.native.sync()
            torch.cuda.set_stream(self._prev_stream)
            self.stream = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # fbo.create: LoadRenderJobContext.tsx:186-223
    def fbo_create(self, width: int, height: int, frameid: int):
        key = (width, height, frameid)
        fb = self._live.get(key)
        if fb is not None:
            return fb
        for i, (pkey, pfb) in enumerate(self._purgatory):
            if pkey[0] == width and pkey[1] == height:
                self._purgatory.pop(i)
                if pkey[2] != frameid:
                    pfb.clear()  # :196-208: a new frameid restarts the accumulation
                self._live[key] = pfb
                return pfb
        if self.group is not None:
            from . import dist as rmdist

            fb = rmdist.ShardedFramebuffer(self.native, self.group, width, height,
                                           render_stream=self.stream.cuda_stream if self.stream is not None else None)
        elif self.stripes is not None:
            from . import shard

            fb = self.native.create_striped_framebuffer(width, height, shard.STRIPE_ROWS, self.stripes[0], self.stripes[1])
        else:
            rb, rc = self.rows if self.rows is not None else (0, height)
            fb = self.native.create_framebuffer(width, height, rb, rc)
        self._live[key] = fb
        return fb

    # fbo.delete: LoadRenderJobContext.tsx:227-248
    def fbo_delete(self, width: int, height: int, frameid: int) -> None:
        key = (width, height, frameid)
        fb = self._live.pop(key, None)
        if fb is None:
            return
        self._purgatory.append((key, fb))
        while len(self._purgatory) > 3:
            _, old = self._purgatory.pop(0)
            old.destroy()


def do_render_job(schema: dict, context: RenderJobContext):
    """``doRenderJob`` (RenderJobExecutor.tsx:77-341).  Returns a generator
    factory: call it with ``present(schema, context, framebuffer,
    samples_so_far)`` and drain the generator; its return value is
    ``{"success": True}`` or ``{"success": False, "why": {...}}`` -- errors are
    values, never exceptions (:56-68,112-136)."""

    def fail(why):
        def gen(present=None):
            return {"success": False, "why": why}
            yield  # pragma: no cover

        return gen

    for key in _REQUIRED:
        if schema.get(key) is None:
            return fail({"type": "general", "infoLog": f"missing field {key}"})
    r = schema["render"]
    try:
        fb = context.fbo_create(r["width"], r["height"], r["frameid"])
    except Exception as e:  # :112-119
        return fail({"type": "general", "infoLog": "Failed to load framebuffers. " + str(e)})
    scene = schema.get("sdfScene")
    if scene is None:
        return fail({"type": "fragment", "infoLog": "no sdfScene: the HIP back end takes a composed scene, not GLSL text"})
    handle = context.get_scene(scene)
    if isinstance(handle, dict):  # :129-136
        return fail(handle)

    sharded = getattr(fb, "sharded", False)
    if sharded:
        fb.dof = schema["dof"]["amount"] != 0.0  # selects what the ranks gather for a present (dist.ShardedFramebuffer)

    def gen(present: Callable):
        # the scene stays pinned in the context's cache while this generator lives (it yields between samples, and other jobs may
        # bring more scenes than the cache holds in the meantime); looked up again here: the generator may start long after the call
        key, handle = context.pin_scene(scene)
        try:
            if isinstance(handle, dict):
                return {"success": False, "why": handle}
            return (yield from run(present, handle))
        finally:
            context.unpin_scene(key)

    def run(present: Callable, handle):
        from . import native as _native

        samples = 0
        n = r["subdivisions"]
        for y_part in range(n):  # :148-162
            for x_part in range(n):
                tile = tile_rect(schema, x_part, y_part)
                left = r["samplesPerPixel"]
                while left > 0:
                    if samples % r["sampleYieldInterval"] == 0:  # :163-166
                        if not sharded:  # (a sharded present is ordered on the job's stream: the next samples render while the frame travels)
                            context.native.sync()
                        present(schema, context, fb, samples)
                        yield
                    # the samples up to the next yield differ in randNoise only (:219-222): one native call for all of them
                    k = min(left, r["sampleYieldInterval"] - samples % r["sampleYieldInterval"])
                    noise = [next_rand_noise() for _ in range(k)]
                    u = uniforms_from_schema(schema, noise[0])
                    try:
                        if k == 1:
                            context.native.render_sample(handle, fb, u, tile, context.flags)  # :181-326
                        else:
                            context.native.render_samples(handle, fb, u, noise, tile, context.flags)
                    except _native.RmError as e:  # errors are values (RenderJobExecutor.tsx:56-68)
                        context.fbo_delete(r["width"], r["height"], r["frameid"])  # the frame goes back to the cache on this path too
                        return {"success": False, "why": {"type": "general", "infoLog": "render failed: " + str(e)}}
                    samples += k
                    left -= k
        context.fbo_delete(r["width"], r["height"], r["frameid"])  # :333-337
        # The job's last word, sharded or not: wait for the device once, so that a failure of an asynchronous launch becomes THIS
        # job's {success: False} and not an exception out of some later call (errors are values: RenderJobExecutor.tsx:56-68).
        try:
            context.native.sync()
        except Exception as e:
            return {"success": False, "why": {"type": "general", "infoLog": "render failed: " + str(e)}}
        present(schema, context, fb, samples)
        return {"success": True}

    return gen


def drain(generator) -> dict:
    """Run a job generator to completion (what index.tsx:242-264 does inside rAF)."""
    try:
        while True:
            next(generator)
    except StopIteration as stop:
        return stop.value


def collect_presents(frames: list) -> Callable:
    """A ``present`` callback for drain(): appends (samples, canvas) for every present of the job that has samples to
    show -- the reference presents once BEFORE the first sample too (RenderJobExecutor.tsx:163 at samplesRenderedSoFar
    = 0: the previous job's accumulation; its presenter divides by its own running count, index.tsx:25-39), which has
    no brightness here.  `canvas` = framebuffer.present(samples): RGBA8 [H, W, 4], row 0 = bottom; on a sharded job the
    call is the ranks' collective and the canvas is None on every rank but 0."""

    def present(schema, context, fb, samples):
        if samples > 0:
            frames.append((samples, fb.present(samples)))

    return present
