"""Frame assembly across ranks: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on a GPU node, "gloo" in the CPU tests).

The only collective of the path is a gather of what is SHOWN to rank 0 once per
presented frame; rendering itself needs no exchange.  Two
payloads:

* "rgba8" (default on a GPU when the job has no depth of field): every rank
  tone-maps the stripes it holds (rm_present_rows = display.frag with blur
  radius 0, the same bytes rm_present gives) and 4 bytes per pixel travel --
  a quarter of the fp32 colour plane.  At 3840x2160 over 8 GPUs rank 0 takes in
  7/8 x 33 MB per presented frame (83 GB/s with a present after every 0.35-ms
  sample, 13 GB/s with a present per 8 samples, spread over seven xGMI links).
* "f32": the accumulated colour plane itself (16 bytes per pixel), for hosts that
  want the radiance (depth of field on: the blur needs neighbour rows, so the
  frame is gathered and rm_present_planes runs on rank 0), and for the CPU tests.

The fp32 planes never move otherwise: they stay where they are accumulated.
"""
from __future__ import annotations

from typing import List, Optional

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
    """

    def __init__(self, height: int, width: int, world: int, rank: int, device, dst: int = 0, channels: int = 4,
                 stripe_rows: int = shard.STRIPE_ROWS, force: bool = False, ctx=None, payload: str = "f32", side_stream: bool = False):
        import torch

        assert payload in ("f32", "rgba8")
        self.torch = torch
        self.height, self.width, self.world, self.rank, self.dst = height, width, world, rank, dst
        self.stripe_rows = stripe_rows
        self.force = force  # run the collective even with one rank (testing aid)
        self.payload = payload
        self.dtype = torch.uint8 if payload == "rgba8" else torch.float32
        self.channels = 4 if payload == "rgba8" else channels
        # native.Context: assemble with one launch of rm_assemble_striped(_bytes); None = torch index_copy_ (CPU/gloo tests)
        self.ctx = ctx if self.channels == 4 else None
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
        shape = (self.max_rows, width, self.channels)
        if rank == dst:
            self.recv_all = torch.empty((world,) + shape, dtype=self.dtype, device=device)
            self.recv = [self.recv_all[p] for p in range(world)]  # gather's output list: views of one buffer
            self.frame = torch.empty((height, width, self.channels), dtype=self.dtype, device=device)
            self.index = [torch.as_tensor(shard.owned_rows(height, world, p, stripe_rows), device=device) for p in range(world)]
        if torch.device(device).type == "cuda":
            self.aux = torch.cuda.Stream(device=device) if side_stream else None
            self.snaps = [torch.zeros(shape, dtype=self.dtype, device=device) for _ in range(2)]
            self.snap_free = [None, None]  # events: the collective that sent snaps[k] is done

    @property
    def row_bytes(self) -> int:
        return self.width * self.channels * (1 if self.payload == "rgba8" else 4)

    # ---- the overlapped form: start(frame n) ... render sample n+1 ... finish(frame n) ----

    def start(self, plane, dist, fb=None, samples: int = 1):
        """Begin gathering a SNAPSHOT of this rank's window; the next sample may render at once (the planes are
        accumulated in place).  payload "f32": `plane` is the [max_rows, W, C] colour tensor.  payload "rgba8":
        `fb` is the native.Framebuffer whose rows are tone-mapped (x 1/samples) into the snapshot.
        Returns a handle for finish()."""
        if self.world == 1 and not self.force:
            return plane[: self.rows] if plane is not None else None
        assert self.pending is None, "FrameGatherer: finish() the previous gather before starting the next"
        torch = self.torch
        if self.snaps is not None:
            k = self.turn
            self.turn ^= 1
            snap = self.snaps[k]
            cur = torch.cuda.current_stream()
            if self.snap_free[k] is not None:
                cur.wait_event(self.snap_free[k])  # two frames old: long done, no stall in practice
            if self.payload == "rgba8":
                self.ctx.present_rows(fb, samples, snap.data_ptr(), cur.cuda_stream)
            else:
                snap.copy_(plane)
            if self.aux is not None:
                self.aux.wait_stream(cur)  # the snapshot is complete; the previous frame's assembly is earlier on aux
            if dist.get_backend() == "gloo":
                # testing aid (several ranks sharing ONE GPU cannot use RCCL): the rows travel through host memory
                host = snap.cpu()
                recv_host = [torch.empty_like(host) for _ in range(self.world)] if self.rank == self.dst else None
                work = dist.gather(host, recv_host, dst=self.dst, async_op=True)
                self.pending = (work, k, recv_host)
                return self.pending
            with torch.cuda.stream(self.aux if self.aux is not None else cur):
                work = dist.gather(snap, self.recv if self.rank == self.dst else None, dst=self.dst, async_op=True)
            self.pending = (work, k)
        else:  # CPU (gloo)
            assert self.payload == "f32"
            snap = plane.clone()
            work = dist.gather(snap, self.recv if self.rank == self.dst else None, dst=self.dst, async_op=True)
            self.pending = (work, snap)
        return self.pending

    def finish(self, handle=None):
        """Wait for start()'s gather and put the stripes in image order (on dst; None elsewhere).  On a GPU the
        returned frame is ordered on the stream the gatherer works on (self.stream(): the current stream, or aux)."""
        if self.world == 1 and not self.force:
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
                if self.rank == self.dst:
                    self._assemble(on.cuda_stream)
            return self.frame
        work.wait()
        return self._assemble() if self.rank == self.dst else None

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
        if self.world == 1 and not self.force:
            return plane[: self.rows]
        dist.gather(plane, self.recv if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        return self._assemble()
