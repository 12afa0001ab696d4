"""The N > 1 path on CPU: two processes over gloo shard a frame in 8-row
stripes, "render" their rows with the oracle (this is a test: the oracle is
the checker and here also the stand-in renderer, since there is no GPU), gather
the colour plane to rank 0 with the same FrameGatherer bench.py uses, and rank
0 checks the assembled frame against the single-process render."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, width, height, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import golden_cases as GC
    from oracle import oracle as O
    from raymarching_engine_amd import dist as rmdist, job as J, shard

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = GC.build_scene("csg_mixed")
    schema = J.make_schema(sc, width, height, render_mode="full", counts=(24, 12), position=(0.3, 0.2, -4.0), lights=GC.LIGHT)
    noises = GC.halton_pairs(2)
    g = rmdist.FrameGatherer(height, width, world, rank, torch.device("cpu"))
    plane = torch.zeros((g.max_rows, width, 4), dtype=torch.float32)
    rows = shard.owned_rows(height, world, rank)
    assert len(rows) == g.rows
    # render this rank's stripes (global pixel coordinates, as a GPU of the run would)
    for k in range(0, len(rows), shard.STRIPE_ROWS):
        r0 = int(rows[k])
        n = min(shard.STRIPE_ROWS, height - r0)
        fr = O.Frame(width, height, r0, n)
        for nz in noises:
            O.render(sc, J.uniforms_from_schema(schema, nz), fr)
        plane[k : k + n] = torch.from_numpy(fr.color)
    frame = g.gather(plane, dist)
    first = frame.clone() if rank == 0 else None
    # the overlapped form bench.py uses: snapshot + async gather, finished later
    handle = g.start(plane, dist)
    plane.zero_()  # the next sample may already be overwriting the plane
    frame2 = g.finish(handle)
    if rank == 0:
        assert torch.equal(frame2, first)
    # the all-gather form (to_all: what a job with depth of field uses for its packed rows, round 4): EVERY rank ends up with the
    # assembled frame, bit for bit the one rank 0 gathered
    ga = rmdist.FrameGatherer(height, width, world, rank, torch.device("cpu"), to_all=True)
    plane2 = torch.zeros((ga.max_rows, width, 4), dtype=torch.float32)
    for k in range(0, len(rows), shard.STRIPE_ROWS):
        r0 = int(rows[k])
        n = min(shard.STRIPE_ROWS, height - r0)
        fr = O.Frame(width, height, r0, n)
        for nz in noises:
            O.render(sc, J.uniforms_from_schema(schema, nz), fr)
        plane2[k : k + n] = torch.from_numpy(fr.color)
    everyone = ga.finish(ga.start(plane2, dist))
    assert everyone is not None and everyone.shape == (height, width, 4)
    check = [torch.zeros_like(everyone) for _ in range(world)]
    dist.all_gather(check, everyone.contiguous())
    assert all(torch.equal(torch.nan_to_num(c), torch.nan_to_num(check[0])) for c in check)
    if rank == 0:
        assert torch.equal(torch.nan_to_num(everyone), torch.nan_to_num(first))
    if rank == 0:
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


# 52 rows = ragged: the ranks hold different numbers of rows; 3 ranks: an odd world, rank 2 holds two stripes of 52 rows' seven
@pytest.mark.parametrize("world,height", [(2, 64), (2, 52), (3, 52)])
def test_two_rank_striped_render_gather_assemble(tmp_path, world, height):
    import torch.multiprocessing as mp

    import golden_cases as GC
    from oracle import oracle as O
    from raymarching_engine_amd import job as J

    width = 48
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), width, height, out), nprocs=world, join=True)
    got = np.load(out)
    sc = GC.build_scene("csg_mixed")
    schema = J.make_schema(sc, width, height, render_mode="full", counts=(24, 12), position=(0.3, 0.2, -4.0), lights=GC.LIGHT)
    fr = O.Frame(width, height)
    for nz in GC.halton_pairs(2):
        O.render(sc, J.uniforms_from_schema(schema, nz), fr)
    assert np.array_equal(got, fr.color, equal_nan=True)


def test_shard_rows_partition_the_frame():
    from raymarching_engine_amd import shard

    for height in (2160, 4096, 8192, 13, 8):
        for parts in (1, 2, 4, 8):
            rows = [shard.owned_rows(height, parts, p) for p in range(parts)]
            allrows = np.sort(np.concatenate(rows))
            assert np.array_equal(allrows, np.arange(height))
            assert shard.row_counts(height, parts) == [len(r) for r in rows]
            data = [np.stack([r.astype(np.float32)] * 3, -1)[:, None, :].repeat(5, 1) for r in rows]
            frame = shard.assemble(data, height)
            assert np.array_equal(frame[:, 0, 0], np.arange(height, dtype=np.float32))
