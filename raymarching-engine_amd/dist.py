"""Frame assembly across ranks: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on a GPU node, "gloo" in the CPU tests).

The only collective of the path is a gather of the colour plane to rank 0 once
per presented frame; rendering itself needs no exchange.
"""
from __future__ import annotations

from typing import List, Optional

from . import shard


class FrameGatherer:
    """Gathers row-striped colour planes to `dst` and puts them in image order.

    Every rank passes a plane padded to `max_rows` rows (ranks can differ by one
    stripe); buffers and the row index tensors are allocated once.
    """

    def __init__(self, height: int, width: int, world: int, rank: int, device, dst: int = 0, channels: int = 4,
                 stripe_rows: int = shard.STRIPE_ROWS, force: bool = False, ctx=None):
        import torch

        self.torch = torch
        self.height, self.width, self.world, self.rank, self.dst = height, width, world, rank, dst
        self.stripe_rows = stripe_rows
        self.force = force  # run the collective even with one rank (testing aid)
        # native.Context: assemble with one launch of rm_assemble_striped on ITS stream (which must be the stream the
        # collective is ordered with, i.e. torch's current stream); None = torch index_copy_ (the CPU/gloo tests)
        self.ctx = ctx if channels == 4 else None
        self.aux = None  # GPU runs: the stream the frame is put together on, so that the render stream never waits for it
        self.counts = shard.row_counts(height, world, stripe_rows)
        self.max_rows = max(self.counts)
        self.rows = self.counts[rank]
        self.recv: Optional[List] = None
        self.frame = None
        self.index = None
        if rank == dst:
            self.recv_all = torch.empty((world, self.max_rows, width, channels), dtype=torch.float32, device=device)
            self.recv = [self.recv_all[p] for p in range(world)]  # gather's output list: views of one buffer
            self.frame = torch.empty((height, width, channels), dtype=torch.float32, device=device)
            self.index = [torch.as_tensor(shard.owned_rows(height, world, p, stripe_rows), device=device) for p in range(world)]

    def start(self, plane, dist):
        """Begin gathering a SNAPSHOT of `plane` (the render of the next sample may start at once:
        the plane is accumulated in place).  Returns a handle for finish()."""
        if self.world == 1 and not self.force:
            return plane[: self.rows]
        snap = plane.clone()
        if self.rank == self.dst and self.aux is not None:
            # the receive buffers are reused: the last frame's assembly (on aux) has to be done before they are overwritten
            self.torch.cuda.current_stream().wait_stream(self.aux)
        work = dist.gather(snap, self.recv if self.rank == self.dst else None, dst=self.dst, async_op=True)
        return (work, snap)

    def finish(self, handle):
        """Wait for start()'s gather and put the stripes in image order (on dst; None elsewhere)."""
        if self.world == 1 and not self.force:
            return handle
        work, _snap = handle
        if self.rank != self.dst:
            work.wait()
            return None
        if self.ctx is not None and self.recv_all.is_cuda:
            if self.aux is None:
                self.aux = self.torch.cuda.Stream(priority=0)
            with self.torch.cuda.stream(self.aux):
                work.wait()  # aux waits for the collective; the render stream does not
                _snap.record_stream(self.aux)
                self._assemble(self.aux.cuda_stream)
            return self.frame  # ordered on self.aux: consumers wait_stream(gatherer.aux) (see drain in bench.py)
        work.wait()
        return self._assemble()

    def _assemble(self, stream=None):
        if self.ctx is not None:
            self.ctx.assemble_striped(self.recv_all.data_ptr(), self.world, self.max_rows, self.width, self.height, self.stripe_rows,
                                      self.frame.data_ptr(), stream)
        else:
            for p in range(self.world):
                self.frame.index_copy_(0, self.index[p], self.recv[p][: self.counts[p]])
        return self.frame

    def gather(self, plane, dist):
        """plane: [max_rows, W, C] tensor of this rank (first self.rows rows valid).
        Returns the assembled [H, W, C] frame on dst, None elsewhere."""
        if self.world == 1:
            return plane[: self.rows]
        dist.gather(plane, self.recv if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        return self._assemble()
