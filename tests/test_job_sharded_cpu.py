"""The sharded render job on CPU: two (three) processes over gloo run the SAME job through job.do_render_job on a
sharded job.RenderJobContext -- the classes a GPU node runs -- and rank 0's presented canvases are compared with the
single-process job's.  There is no GPU here, so the device side is a stand-in built on the oracle (this is a test: the
oracle is the checker and here also the stand-in renderer): a fake native context that renders this rank's stripes with
oracle.render, tone-maps / packs them like rm_present_rows / rm_pack_present_rows and runs the present pass of its own
stripes of the gathered frame with oracle.present (rm_present_striped_rows: with depth of field every rank blurs what it holds).  What is under test is everything above the C ABI: the striped window arithmetic,
the yield cadence, both gather payloads (depth of field off and on), the assembly, ragged heights."""
import ctypes
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


def _at(ptr, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return np.frombuffer((ctypes.c_uint8 * n).from_address(ptr), dtype=dtype).reshape(shape)


class OracleFb:
    """rows of a width x height image held by one rank (packed), rendered stripe by stripe with the oracle"""

    def __init__(self, O, width, height, rows, stripe):
        self.O, self.width, self.height, self.rows, self.stripe = O, width, height, rows, stripe
        self.row_count = len(rows)
        self.h = self
        self.clear()

    def clear(self):
        self.planes = [np.zeros((self.row_count, self.width, 4), np.float32) for _ in range(3)]

    def destroy(self):
        pass

    def render(self, scene, u, tile):
        O = self.O
        for k in range(0, self.row_count, self.stripe):
            r0 = int(self.rows[k])
            n = min(self.stripe, self.row_count - k)
            fr = O.Frame(self.width, self.height, r0, n)
            fr.color[:], fr.normal_dof[:], fr.albedo_depth[:] = (p[k:k + n] for p in self.planes)
            O.render(scene, u, fr, None if tile is None else (tile.x, tile.y, tile.w, tile.h))
            for p, q in zip(self.planes, (fr.color, fr.normal_dof, fr.albedo_depth)):
                p[k:k + n] = q


class OracleNative:
    """the methods of native.Context the sharded job uses, on host memory"""

    takes_host_pointers = True  # (dist.ShardedFramebuffer refuses to hand a real context the host memory of a CPU group)

    def __init__(self, O, shard):
        self.O, self.shard = O, shard
        self.scenes = {}
        self.calls = {"present_striped_rows": [], "present_device": 0}

    def create_scene(self, scene):
        return scene

    def create_framebuffer(self, width, height, rb=0, rc=None):
        rc = height if rc is None else rc
        return OracleFb(self.O, width, height, np.arange(rb, rb + rc), rc)

    def create_striped_framebuffer(self, width, height, stripe_rows, parts, part, *planes):
        return OracleFb(self.O, width, height, self.shard.owned_rows(height, parts, part, stripe_rows), stripe_rows)

    def sync(self):
        pass

    def render_sample(self, handle, fb, u, tile, flags):
        fb.h.render(handle, u, tile)

    def render_samples(self, handle, fb, u, noises, tile, flags):
        for n0, n1 in noises:
            u.randNoise[0], u.randNoise[1] = n0, n1
            fb.h.render(handle, u, tile)

    def present_rows(self, fb, samples, out_ptr, stream=None):  # rm_present_rows: display.frag with blur radius 0 on this rank's rows
        out = _at(out_ptr, (fb.row_count, fb.width, 4), np.uint8)
        out[:] = self.O.present(fb.planes[0], None, samples)

    def pack_present_rows(self, fb, out_ptr, stream=None):  # rm_pack_present_rows
        out = _at(out_ptr, (fb.row_count, fb.width, 4), np.float32)
        out[..., :3] = fb.planes[0][..., :3]
        out[..., 3] = fb.planes[1][..., 3]

    def present_striped_rows(self, color_ptr, normal_dof_ptr, width, height, samples, stripe_rows, parts, part, out_ptr, stream=None):
        # rm_present_striped_rows: display.frag for the rows this part holds, read from the whole (gathered) frame
        color = _at(color_ptr, (height, width, 4), np.float32)
        nd = _at(normal_dof_ptr, (height, width, 4), np.float32)
        rows = self.shard.owned_rows(height, parts, part, stripe_rows)
        _at(out_ptr, (len(rows), width, 4), np.uint8)[:] = self.O.present(color, nd, samples)[rows]
        self.calls["present_striped_rows"].append((parts, part, len(rows)))

    def present_device(self, color_ptr, normal_dof_ptr, width, height, samples, out_ptr, stream=None):  # rm_present_device
        color = _at(color_ptr, (height, width, 4), np.float32)
        nd = _at(normal_dof_ptr, (height, width, 4), np.float32)
        _at(out_ptr, (height, width, 4), np.uint8)[:] = self.O.present(color, nd, samples)
        self.calls["present_device"] += 1


def _schema(J, GC, width, height, dof, spp, interval):
    sc = GC.build_scene("csg_mixed")
    return sc, J.make_schema(sc, width, height, render_mode="full", counts=(24, 12), position=(0.3, 0.2, -4.0), lights=GC.LIGHT,
                             samples_per_pixel=spp, sample_yield_interval=interval, dof_amount=0.05 if dof else 0.0, dof_distance=3.5)


def _worker(rank, world, port, width, height, dof, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import golden_cases as GC
    from oracle import oracle as O
    from raymarching_engine_amd import dist as rmdist, job as J, shard

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = rmdist.ShardGroup(dist, torch.device("cpu"))
    assert group.world == world and group.rank == rank and group.sharded
    stand_in = OracleNative(O, shard)
    ctx = J.RenderJobContext(group=group, native_context=stand_in)
    sc, schema = _schema(J, GC, width, height, dof, spp=3, interval=2)
    J.reset_halton()
    frames = []
    res = J.drain(J.do_render_job(schema, ctx)(J.collect_presents(frames)))
    assert res == {"success": True}
    assert [n for n, _ in frames] == [2, 3]  # a present at the yield after 2 samples and the final one
    if rank == 0:
        np.save(out_path, np.stack([c for _, c in frames]))
    else:
        assert all(c is None for _, c in frames)
    # the same frameid again: the accumulation continues (LoadRenderJobContext.tsx:196-208), the stripes stay where they are
    res = J.drain(J.do_render_job(schema, ctx)(J.collect_presents(frames)))
    assert res == {"success": True} and frames[-1][0] == 3
    if rank == 0:
        np.save(out_path.replace(".npy", "_again.npy"), frames[-1][1])
    # with depth of field EVERY rank ran the present pass, for the rows it holds and no others (1 / world of the blur each; round 3 had
    # rank 0 blur the whole frame); without it nobody blurs
    mine = len(shard.owned_rows(height, world, rank, group.stripe_rows))
    # what bench.py's `collective` object reports for a sharded run: the bytes one present takes into the receiving ranks (padded windows)
    sfb = ctx.fbo_create(width, height, schema["render"]["frameid"])
    window = max(shard.row_counts(height, world, group.stripe_rows)) * width
    assert sfb.gathered_bytes_per_present() == world * window * 4 + (world * world * window * 16 if dof else 0)
    assert stand_in.calls["present_device"] == 0
    assert stand_in.calls["present_striped_rows"] == ([(world, rank, mine)] * 4 if dof else [])
    dist.barrier()
    dist.destroy_process_group()


# 52 rows = ragged (the ranks hold different numbers of rows); 3 ranks: an odd world
@pytest.mark.parametrize("world,height,dof", [(2, 64, False), (2, 52, True), (3, 52, False), (3, 40, True)])
def test_sharded_render_job_presents_what_one_process_presents(tmp_path, world, height, dof):
    import torch.multiprocessing as mp

    import golden_cases as GC
    from oracle import oracle as O
    from raymarching_engine_amd import job as J, shard

    width = 48
    out = str(tmp_path / "frames.npy")
    mp.spawn(_worker, args=(world, _free_port(), width, height, dof, out), nprocs=world, join=True)
    got, again = np.load(out), np.load(out.replace(".npy", "_again.npy"))
    # one process, one framebuffer, the same job twice
    ctx = J.RenderJobContext(native_context=OracleNative(O, shard))
    sc, schema = _schema(J, GC, width, height, dof, spp=3, interval=2)
    J.reset_halton()
    frames = []
    J.drain(J.do_render_job(schema, ctx)(lambda s, c, fb, n: frames.append(O.present(fb.planes[0], fb.planes[1], n)) if n else None))
    assert np.array_equal(got, np.stack(frames))
    if dof:  # the blur is on: the canvas is not what a present without depth of field gives
        fb = ctx.fbo_create(width, height, 0)
        assert not np.array_equal(got[-1], O.present(fb.planes[0], None, 3))
    J.drain(J.do_render_job(schema, ctx)(lambda s, c, fb, n: frames.append(O.present(fb.planes[0], fb.planes[1], n)) if n else None))
    assert np.array_equal(again, frames[-1])
    assert not np.array_equal(again, got[-1])  # six samples' accumulation shown as three: brighter


def test_sharded_context_with_one_rank_is_the_plain_context():
    from raymarching_engine_amd import dist as rmdist, job as J

    class OneRank:
        def get_world_size(self): return 1
        def get_rank(self): return 0

    g = rmdist.ShardGroup(OneRank(), "cpu")
    assert not g.sharded
    ctx = J.RenderJobContext(group=g, native_context=object())
    assert ctx.group is None
