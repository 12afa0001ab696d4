"""A render job's frame across ranks: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on a GPU node, "gloo" in the CPU tests).

Rendering needs no exchange (every pixel-sample depends on its own pixel only:
raymarcher.frag:226,350-351,382); the only collective of the path is a gather of
what is SHOWN to rank 0, once per presented frame -- the
counterpart of the reference's ``present`` callback drawing the accumulated
textures to the canvas (index.tsx:25-59, display.frag:16-64).  Three payloads:

* "rgba8" -- jobs without depth of field (dof.amount = 0).  The blur radius of
  display.frag:24-27 is 0, its one tap is the pixel itself, so every rank
  tone-maps the stripes it holds (rm_present_rows, the bytes rm_present gives) and
  4 bytes per pixel travel.  At 3840x2160 over 8 GPUs rank 0 takes in 7/8 x 33 MB
  per presented frame.
* "f32dof" -- jobs WITH depth of field.  The blur reads up to 16 rows either side
  of a pixel; with 8-row stripes those rows live on other GPUs (a halo would be
  four stripes each way, i.e. most of the frame).  Round 3 gathered the packed
  rows -- (colour.rgb, normalAndDofRadius.w), one float4 per pixel,
  rm_pack_present_rows -- to rank 0 and let rank 0 blur the whole frame alone:
  correct, and serial (18 ms per present at the 16-pixel cap on 4K, ~145 ms at
  8192^2 against a 20 ms shard step).  Since round 4 the packed rows go to EVERY
  rank (all-gather: 16 bytes per pixel into each GPU over its seven links, 1 GB at
  8192^2 = ~1 ms), every rank puts the frame in image order and blurs and
  tone-maps the stripes IT holds (rm_present_striped_rows: 1 / N of the pass),
  and the RGBA8 rows are gathered to rank 0 like those of a job without depth of
  field.  The all-gather travels while the next samples render; the bytes are
  those of rm_present on the unsharded frame.
* "f32" -- the accumulated colour plane itself, for hosts that want the radiance.

The fp32 planes never move otherwise: they stay where they are accumulated.

``ShardedFramebuffer`` is what a sharded ``job.RenderJobContext`` hands out as the
job's framebuffer: this rank's striped window plus the gatherers; its
``present(samples)`` is the collective counterpart of ``native.Framebuffer.present``.
"""
from __future__ import annotations

from typing import Callable, List, Optional

from . import shard


class FrameGatherer:
    """Gathers row-striped planes to `dst` and puts them in image order.

    Every rank passes a window padded to `max_rows` rows (ranks can differ by one
    stripe).  Receive buffers, the assembled frame and two snapshot buffers are
    allocated once.  At most ONE gather is outstanding: start() -> finish().

    Streams on a GPU.  By default everything is enqueued on the CURRENT stream: the snapshot, the collective (RCCL's
    own stream waits for the stream it is called from), the wait for it and rm_assemble_striped(_bytes).  With samples
    in flight that stream only carries the small blend kernels -- the renders run on the context's side streams -- so
    the next samples render while the frame travels, and no stream is added: the HIP runtime deals a process's
    streams over a few hardware queues and streams that share one serialise (with two more streams here rank 0's
    renders stopped overlapping: 0.65 instead of 0.41 ms per step).  `side_stream=True` moves the collective and the
    assembly to a stream of their own (for hosts that render on the current stream itself).

    The current stream must be the one the renders are ordered on (native.Context.set_stream) and not torch's default
    stream: that one is the NULL stream, which the context's own non-blocking stream is not ordered with -- a gather
    could read a half-written snapshot.  start() checks both when it has a context.
    """

    PAYLOADS = ("f32", "rgba8", "f32dof")

    def __init__(self, height: int, width: int, world: int, rank: int, device, dst: int = 0, channels: int = 4,
                 stripe_rows: int = shard.STRIPE_ROWS, force: bool = False, ctx=None, payload: str = "f32", side_stream: bool = False,
                 render_stream: Optional[int] = None, to_all: bool = False):
        import torch

        assert payload in self.PAYLOADS
        self.torch = torch
        self.height, self.width, self.world, self.rank, self.dst = height, width, world, rank, dst
        self.stripe_rows = stripe_rows
        self.force = force  # run the collective even with one rank (testing aid)
        self.to_all = to_all  # all-gather: EVERY rank receives every part and assembles the frame (the depth-of-field present)
        self.payload = payload
        self.dtype = torch.uint8 if payload == "rgba8" else torch.float32
        self.channels = channels if payload == "f32" else 4
        # native.Context: assemble with one launch of rm_assemble_striped(_bytes); None = torch index_copy_ (CPU/gloo tests)
        self.ctx = ctx if self.channels == 4 else None
        self.render_stream = render_stream  # the hipStream_t the renders are ordered on, if the host tells (checked in start)
        self.counts = shard.row_counts(height, world, stripe_rows)
        self.max_rows = max(self.counts)
        self.rows = self.counts[rank]
        self.recv: Optional[List] = None
        self.frame = None
        self.index = None
        self.pending = None
        self.aux = None
        self.snaps = None  # two snapshot buffers, used alternately (GPU)
        self.snap_free = None
        self.turn = 0
        self.device = torch.device(device)
        self.shape = (self.max_rows, width, self.channels)
        self.collective = world > 1 or force  # otherwise nothing is gathered and nothing is allocated
        if (rank == dst or to_all) and self.collective:
            self.recv_all = torch.empty((world,) + self.shape, dtype=self.dtype, device=device)
            self.recv = [self.recv_all[p] for p in range(world)]  # gather's output list: views of one buffer
            self.frame = torch.empty((height, width, self.channels), dtype=self.dtype, device=device)
            self.index = [torch.as_tensor(shard.owned_rows(height, world, p, stripe_rows), device=device) for p in range(world)]
        if self.device.type == "cuda" and self.collective:
            self.aux = torch.cuda.Stream(device=device) if side_stream else None
            self.snaps = [torch.zeros(self.shape, dtype=self.dtype, device=device) for _ in range(2)]
            self.snap_free = [None, None]  # events: the collective that sent snaps[k] is done

    @property
    def receives(self) -> bool:
        return self.to_all or self.rank == self.dst

    def _collect(self, dist, send, recv):
        """the collective: gather to dst, or all-gather (to_all); asynchronous"""
        if self.to_all:
            return dist.all_gather(recv, send, async_op=True)
        return dist.gather(send, recv if self.rank == self.dst else None, dst=self.dst, async_op=True)

    @property
    def row_bytes(self) -> int:
        return self.width * self.channels * (1 if self.payload == "rgba8" else 4)

    def _fill(self, plane, fb, samples) -> Callable:
        """What writes this rank's payload into a snapshot tensor (snap, hipStream_t as an integer or None)."""
        if self.payload == "rgba8":
            return lambda snap, stream: self.ctx.present_rows(fb, samples, snap.data_ptr(), stream)
        if self.payload == "f32dof":
            return lambda snap, stream: self.ctx.pack_present_rows(fb, snap.data_ptr(), stream)
        return lambda snap, stream: snap.copy_(plane)

    # ---- the overlapped form: start(frame n) ... render sample n+1 ... finish(frame n) ----

    def start(self, plane, dist, fb=None, samples: int = 1, fill: Optional[Callable] = None):
        """Begin gathering a SNAPSHOT of this rank's window; the next sample may render at once (the planes are
        accumulated in place).  payload "f32": `plane` is the [max_rows, W, C] colour tensor.  payload "rgba8" /
        "f32dof": `fb` is the native.Framebuffer whose rows are tone-mapped (x 1/samples) / packed into the snapshot.
        `fill(snapshot, stream)` replaces either (hosts with a device side of their own, the CPU tests).
        Returns a handle for finish()."""
        if not self.collective:
            return plane[: self.rows] if plane is not None else None
        assert self.pending is None, "FrameGatherer: finish() the previous gather before starting the next"
        torch = self.torch
        fill = fill or self._fill(plane, fb, samples)
        if self.snaps is not None:
            k = self.turn
            self.turn ^= 1
            snap = self.snaps[k]
            cur = torch.cuda.current_stream()
            if self.ctx is not None or self.render_stream is not None:
                if cur.cuda_stream == 0:
                    raise RuntimeError("FrameGatherer.start: the current stream is torch's default (NULL) stream, which the render context's "
                                       "stream is not ordered with; make a torch.cuda.Stream() current and pass it to Context.set_stream")
                if self.render_stream is not None and cur.cuda_stream != self.render_stream:
                    raise RuntimeError("FrameGatherer.start: the current stream is not the stream the renders are ordered on")
            if self.snap_free[k] is not None:
                cur.wait_event(self.snap_free[k])  # two frames old: long done, no stall in practice
            fill(snap, cur.cuda_stream)
            if self.aux is not None:
                self.aux.wait_stream(cur)  # the snapshot is complete; the previous frame's assembly is earlier on aux
            if dist.get_backend() == "gloo":
                # testing aid (several ranks sharing ONE GPU cannot use RCCL): the rows travel through host memory
                host = snap.cpu()
                recv_host = [torch.empty_like(host) for _ in range(self.world)] if self.receives else None
                work = self._collect(dist, host, recv_host)
                self.pending = (work, k, recv_host)
                return self.pending
            with torch.cuda.stream(self.aux if self.aux is not None else cur):
                work = self._collect(dist, snap, self.recv)
            self.pending = (work, k)
        else:  # CPU (gloo)
            snap = torch.empty(self.shape, dtype=self.dtype)
            fill(snap, None)
            work = self._collect(dist, snap, self.recv)
            self.pending = (work, snap)
        return self.pending

    def finish(self, handle=None):
        """Wait for start()'s gather and put the stripes in image order (on dst -- on every rank with to_all; None elsewhere).  On a GPU the
        returned frame is ordered on the stream the gatherer works on (self.stream(): the current stream, or aux)."""
        if not self.collective:
            return handle
        assert self.pending is not None and (handle is None or handle is self.pending)
        work, k = self.pending[0], self.pending[1]
        recv_host = self.pending[2] if len(self.pending) > 2 else None
        self.pending = None
        torch = self.torch
        if self.snaps is not None:
            on = self.stream()
            with torch.cuda.stream(on):
                work.wait()  # a stream-level wait, not a host one (gloo: a host wait)
                if recv_host is not None:
                    for p in range(self.world):
                        self.recv[p].copy_(recv_host[p], non_blocking=False)
                ev = torch.cuda.Event()
                ev.record(on)
                self.snap_free[k] = ev
                if self.receives:
                    self._assemble(on.cuda_stream)
            return self.frame
        work.wait()
        return self._assemble() if self.receives else None

    def stream(self):
        """The stream the collective and the assembly are ordered on."""
        return self.aux if self.aux is not None else self.torch.cuda.current_stream()

    def _assemble(self, stream=None):
        if self.ctx is not None and self.recv_all.is_cuda:
            self.ctx.assemble_striped_bytes(self.recv_all.data_ptr(), self.world, self.max_rows, self.row_bytes, self.height, self.stripe_rows,
                                            self.frame.data_ptr(), stream)
        else:
            for p in range(self.world):
                self.frame.index_copy_(0, self.index[p], self.recv[p][: self.counts[p]])
        return self.frame

    # ---- the blocking form ----

    def gather(self, plane, dist):
        """plane: [max_rows, W, C] tensor of this rank (first self.rows rows valid).
        Returns the assembled [H, W, C] frame on dst, None elsewhere."""
        if not self.collective:
            return plane[: self.rows]
        dist.gather(plane, self.recv if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        return self._assemble()


class ShardGroup:
    """The ranks that share a render job's frames: what a sharded job.RenderJobContext is constructed with.
    `dist` is torch.distributed (initialised: backend "nccl" on a GPU node -- one process per GPU --, "gloo" in the CPU
    tests and for ranks that share one GPU); `device` the torch device the snapshots, receive buffers and the assembled
    frame live on."""

    def __init__(self, dist, device, world: Optional[int] = None, rank: Optional[int] = None, stripe_rows: int = shard.STRIPE_ROWS,
                 force: bool = False):
        self.dist = dist
        self.device = device
        self.world = dist.get_world_size() if world is None else world
        self.rank = dist.get_rank() if rank is None else rank
        self.stripe_rows = stripe_rows
        self.force = force  # run the collective even with one rank (testing aid: the RCCL path on a one-GPU box)

    @property
    def sharded(self) -> bool:
        return self.world > 1 or self.force


class ShardedFramebuffer:
    """The framebuffer of a sharded render job on one rank: this rank's 8-row stripes of the three accumulation planes
    (rm_fb_create_striped: pixel coordinates, texcoord and aspect stay global, so the assembled frame has the bits of
    the single-GPU one) plus what it takes to show the frame.

    ``present(samples)`` is COLLECTIVE -- every rank of the group calls it, with the same `samples`, as every rank's
    ``present`` callback of the job does (job.do_render_job) -- and returns the RGBA8 canvas [H, W, 4] (row 0 = bottom) on
    rank 0, None on the others: the bytes ``native.Framebuffer.present`` returns for the same samples on one GPU, with
    and without depth of field (``dof``: set by the job from its schema; it selects the payload, see the module text).
    ``start_present`` / ``finish_present`` are the same in two halves: the gather travels while the next samples render,
    and the frame stays on the device.
    """

    sharded = True

    def __init__(self, native_ctx, group: ShardGroup, width: int, height: int, render_stream: Optional[int] = None):
        import torch

        self.torch = torch
        if not group.sharded:  # one rank and no `force`: nothing is gathered (FrameGatherer allocates nothing) and present() would have no frame
            raise ValueError("ShardedFramebuffer: a group of one rank does not shard a frame -- use native.Context.create_framebuffer "
                             "(job.RenderJobContext does), or ShardGroup(force=True) to run the collective path with one rank")
        self.ctx, self.group = native_ctx, group
        self.width, self.height = width, height
        self.fb = native_ctx.create_striped_framebuffer(width, height, group.stripe_rows, group.world, group.rank)
        self.row_count = self.fb.row_count
        self.render_stream = render_stream
        self.dof = False  # does the job that renders into this framebuffer have depth of field (set by do_render_job)
        self._gatherers = {}
        self._pending = None  # (gatherer, samples) of the present in flight

    # what the render calls and the framebuffer cache of job.RenderJobContext use
    @property
    def h(self):
        return self.fb.h

    def clear(self):
        self.fb.clear()

    def destroy(self):
        self.fb.destroy()

    def download(self, plane: int = 0):
        """This rank's rows of a plane, packed in ascending stripe order (shard.owned_rows)."""
        return self.fb.download(plane)

    def gatherer(self, payload: str) -> FrameGatherer:
        g = self._gatherers.get(payload)
        if g is None:
            gr = self.group
            g = FrameGatherer(self.height, self.width, gr.world, gr.rank, gr.device, stripe_rows=gr.stripe_rows, force=gr.force,
                              ctx=self.ctx, payload=payload, render_stream=self.render_stream, to_all=payload == "f32dof")
            assert g.rows == self.row_count
            self._gatherers[payload] = g
        return g

    def gathered_bytes_per_present(self, dof: Optional[bool] = None) -> int:
        """Bytes the collectives of ONE present take into the receiving ranks' memory (padded windows, as they travel): the RGBA8
        rows of every rank into rank 0; with depth of field also the packed float4 rows of every rank into EVERY rank."""
        dof = self.dof if dof is None else dof
        gr = self.group
        window = max(shard.row_counts(self.height, gr.world, gr.stripe_rows)) * self.width
        total = gr.world * window * 4
        if dof:
            total += gr.world * gr.world * window * 16
        return total

    def start_present(self, samples: int, dof: Optional[bool] = None):
        """First half of present(): snapshot this rank's payload and start the gather (asynchronous)."""
        assert self._pending is None, "ShardedFramebuffer: finish_present() before the next start_present()"
        dof = self.dof if dof is None else dof
        g = self.gatherer("f32dof" if dof else "rgba8")
        g.start(None, self.group.dist, fb=self.fb, samples=samples)
        self._pending = (g, int(samples))

    def finish_present(self):
        """Second half: the canvas as a uint8 tensor [H, W, 4] on the group's device on rank 0 (ordered on the current
        stream on a GPU), None on the other ranks."""
        assert self._pending is not None
        g, samples = self._pending
        self._pending = None
        frame = g.finish()
        if g.payload == "rgba8":
            return frame if self.group.rank == g.dst else None
        # depth of field: every rank holds the packed frame now (the all-gather) and runs the present pass for ITS stripes -- the
        # assembled (colour.rgb, dofRadius) buffer is both of its planes (display.frag reads .rgb of one, .w of the other) --
        # straight into the snapshot of an RGBA8 gather, which then is the one of a job without depth of field
        gr = self.group
        if not frame.is_cuda and not getattr(self.ctx, "takes_host_pointers", False):
            # the library's present pass reads DEVICE memory; a CPU group (gloo) only exists in the tests, whose stand-in context says so
            raise RuntimeError("ShardedFramebuffer.finish_present: the gathered frame is in host memory (a CPU group) and the render context "
                               "is a real one: rm_present_striped_rows takes device pointers")
        # (the blur of this rank's stripes and the RGBA8 gather behind it run here, back to back: of a depth-of-field present only the
        # all-gather of the packed rows overlaps the next samples; every rank holds recv_all + frame, two full-frame float4 buffers)
        g8 = self.gatherer("rgba8")
        g8.start(None, gr.dist, fill=lambda snap, stream: self.ctx.present_striped_rows(frame.data_ptr(), frame.data_ptr(), self.width, self.height, samples,
                                                                                        gr.stripe_rows, gr.world, gr.rank, snap.data_ptr(), stream))
        canvas = g8.finish()
        return canvas if gr.rank == g8.dst else None

    def present(self, samples: int, dof: Optional[bool] = None):
        self.start_present(samples, dof)
        canvas = self.finish_present()
        if canvas is None:
            return None
        return canvas.cpu().numpy() if canvas.is_cuda else canvas.numpy().copy()  # (the gatherer's frame is reused by the next present)
