#!/usr/bin/env python3
"""One GPU stands in for every rank of an N-GPU run of the headline frame, one rank at a time: the rank's stripes are
rendered with 3 samples in flight, tone-mapped (rm_present_rows) and -- for rank 0 -- a full RGBA8 frame is put together
from a gathered buffer (rm_assemble_striped_bytes) on a side stream, as dist.FrameGatherer drives it; the collective
itself is replaced by the local copy of the rank's own part.  Prints ms per step per rank and the speed-up the slowest
rank allows: what the compute side of strong scaling can give (the receive side needs the real node)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S, shard

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = native.Context(0); ctx.set_stream(st.cuda_stream)
sc = S.Mandelbulb(); h = ctx.create_scene(sc)
W, H = 3840, 2160
schema = J.make_schema(sc, W, H, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
aux = torch.cuda.Stream()
issue = torch.cuda.Stream()  # the stream the collective is enqueued from (dist.FrameGatherer): the render stream never waits
full = None
for N in (1, 2, 4, 8):
    counts = shard.row_counts(H, N); max_rows = max(counts)
    # EMU_YIELD = render.sampleYieldInterval: 1 = present after every sample (4 samples in flight); 8 (bench.py's value for
    # a sharded frame) = the 8 samples between two presents in one rm_render_samples call, two such batches in flight
    Y = 1 if N == 1 else int(os.environ.get("EMU_YIELD", "8"))
    depth = 1 if N == 1 else int(os.environ.get("EMU_DEPTH", "4" if Y == 1 else "3"))
    ctx.set_samples_in_flight(depth)
    recv = torch.zeros((N, max_rows, W, 4), dtype=torch.uint8, device=dev)
    frame = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
    snaps = [torch.zeros((max_rows, W, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
    per_rank = []
    for r in range(N):
        planes = [torch.zeros((max_rows, W, 4), dtype=torch.float32, device=dev) for _ in range(3)]
        fb = ctx.create_striped_framebuffer(W, H, shard.STRIPE_ROWS, N, r, *(p.data_ptr() for p in planes))
        h2, h3 = J.halton(2), J.halton(3)
        k = [0]
        def step(present):
            if Y == 1:
                u.randNoise[0], u.randNoise[1] = next(h2), next(h3)
                ctx.render_sample(h, fb, u, None, abi.RM_RENDER_FAST)
            else:
                ctx.render_samples(h, fb, u, [(next(h2), next(h3)) for _ in range(Y)], None, abi.RM_RENDER_FAST)
            if present and N > 1:
                snap = snaps[k[0] & 1]; k[0] += 1
                ctx.present_rows(fb, k[0], snap.data_ptr(), st.cuda_stream)
                if r == 0:
                    issue.wait_stream(st)
                    issue.wait_stream(aux)
                    with torch.cuda.stream(issue):
                        recv[0].copy_(snap, non_blocking=True)  # stands in for the collective's local part
                    aux.wait_stream(issue)
                    ctx.assemble_striped_bytes(recv.data_ptr(), N, max_rows, W * 4, H, shard.STRIPE_ROWS, frame.data_ptr(), aux.cuda_stream)
        for _ in range(max(2, 8 // Y)): step(True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); K = max(8, 64 // Y)
        for _ in range(K): step(True)
        torch.cuda.synchronize(); per_rank.append((time.perf_counter() - t0) / (K * Y) * 1e3)
        fb.destroy()
    if N == 1: full = per_rank[0]
    print(f"N={N}: ms per sample per rank (yield interval {Y}, render x{depth} in flight + present rows{' + assemble on rank 0' if N > 1 else ''}): "
          + " ".join(f"{t:.3f}" for t in per_rank) + f"   -> slowest {max(per_rank):.3f} ms, speed-up {full / max(per_rank):.2f}x of {N}")
