"""rm_present_sharded through ctypes: ONE process, several native contexts (all on GPU 0: a gpurun box has one), each holding
one part of a frame's 8-row stripes; the canvas it assembles -- rows tone-mapped per context and copied to the first one (no depth
of field), or packed, copied to EVERY context, blurred there for the stripes that context holds and then copied to the first one
(depth of field; round 4) -- equals rm_present of the same samples on one framebuffer, byte for byte (display.frag:16-64;
index.tsx:25-59).  And the two halves, rm_present_sharded_start / _finish: the next samples are handed out while the frame travels."""
import numpy as np
import pytest

import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S, shard

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("parts", [2, 5])
@pytest.mark.parametrize("dof", [False, True])
def test_present_sharded_equals_present(parts, dof):
    W, H = 136, 100
    sc = S.Mandelbulb()
    schema = J.make_schema(sc, W, H, counts=(40, 20), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT,
                           dof_amount=0.03 if dof else 0.0, dof_distance=1.6)
    noises = GC.halton_pairs(3)
    one = native.Context(0)
    ctxs = [native.Context(0) for _ in range(parts)]
    try:
        h = one.create_scene(sc)
        fb = one.create_framebuffer(W, H)
        for nz in noises:
            one.render_sample(h, fb, J.uniforms_from_schema(schema, nz), None, abi.RM_RENDER_STRICT)
        want = fb.present(len(noises))
        fbs, hs = [], []
        for p, c in enumerate(ctxs):
            hs.append(c.create_scene(sc))
            fbs.append(c.create_striped_framebuffer(W, H, shard.STRIPE_ROWS, parts, p))
        for nz in noises:  # every sample to each context in turn: the launches are asynchronous
            u = J.uniforms_from_schema(schema, nz)
            for c, hh, f in zip(ctxs, hs, fbs):
                c.render_sample(hh, f, u, None, abi.RM_RENDER_STRICT)
        got = native.present_sharded(ctxs, fbs, len(noises), dof)
        assert got.shape == (H, W, 4) and np.array_equal(got, want)
        assert int(want[..., :3].max()) > 100
        if dof:  # and the blur is on: not the canvas a present without it gives
            assert not np.array_equal(got, native.present_sharded(ctxs, fbs, len(noises), False))
        # argument checks: parts in the wrong order, a window that is not striped
        with pytest.raises(native.RmError):
            native.present_sharded(ctxs[::-1], fbs, len(noises), dof)
        with pytest.raises(native.RmError):
            native.present_sharded([one], [fb], 1, dof)
    finally:
        for c in ctxs + [one]:
            c.close()


@pytest.mark.parametrize("dof", [False, True])
def test_present_sharded_in_two_halves_overlaps_the_next_samples(dof):
    """start() snapshots and returns; samples rendered between start() and finish() do not show in THAT canvas (the snapshot is taken
    on the contexts' streams, behind the samples enqueued before it) and do show in the next one; a second start() before the
    finish() is refused; three presents in a row reuse the buffers."""
    parts, W, H = 3, 160, 84  # 10.5 stripes over 3 contexts: ragged
    sc = S.Mandelbulb()
    schema = J.make_schema(sc, W, H, counts=(40,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT, dof_amount=0.03 if dof else 0.0, dof_distance=1.6)
    noises = GC.halton_pairs(6)
    one = native.Context(0)
    ctxs = [native.Context(0) for _ in range(parts)]
    try:
        h = one.create_scene(sc)
        fb = one.create_framebuffer(W, H)
        want = []
        for k, nz in enumerate(noises):
            one.render_sample(h, fb, J.uniforms_from_schema(schema, nz), None, abi.RM_RENDER_FAST)
            if k % 2 == 1:
                want.append(fb.present(k + 1))
        hs = [c.create_scene(sc) for c in ctxs]
        fbs = [c.create_striped_framebuffer(W, H, shard.STRIPE_ROWS, parts, p) for p, c in enumerate(ctxs)]
        got, pending = [], False
        for k, nz in enumerate(noises):
            u = J.uniforms_from_schema(schema, nz)
            for c, hh, f in zip(ctxs, hs, fbs):
                c.render_sample(hh, f, u, None, abi.RM_RENDER_FAST)
            if k == 2:  # between a start and its finish
                with pytest.raises(native.RmError):
                    native.present_sharded_start(ctxs, fbs, k + 1, dof)
            if k % 2 == 1:
                if pending:
                    got.append(native.present_sharded_finish(ctxs, W, H))
                native.present_sharded_start(ctxs, fbs, k + 1, dof)  # ... and the loop goes on rendering
                pending = True
        got.append(native.present_sharded_finish(ctxs, W, H))
        with pytest.raises(native.RmError):
            native.present_sharded_finish(ctxs, W, H)  # nothing pending
        assert len(got) == len(want) == 3
        for k, (a, b) in enumerate(zip(got, want)):
            assert np.array_equal(a, b), f"present {k}: {int((a != b).sum())} bytes differ"
    finally:
        for c in ctxs + [one]:
            c.close()


@pytest.mark.parametrize("stripe_rows", [8, 16, 4, 6])
def test_present_striped_rows_are_the_rows_of_present(stripe_rows):
    """rm_present_striped_rows -- one part's share of the present pass, read from the whole frame -- for every part of 1, 3 and 8:
    the bytes of rm_present's rows, with a blur radius up to the cap (dof amount 0.3), at an image edge that wraps (REPEAT), for
    stripe heights the kernel stages in LDS (multiples of 4) and one it does not."""
    import torch

    W, H = 200, 90
    sc = S.csg64()
    schema = J.make_schema(sc, W, H, counts=(40,), render_mode="full", position=(0, 0, -5.0), lights=GC.LIGHT, dof_amount=0.3, dof_distance=4.0)
    ctx = native.Context(0)
    try:
        h = ctx.create_scene(sc)
        fb = ctx.create_framebuffer(W, H)
        for nz in GC.halton_pairs(2):
            ctx.render_sample(h, fb, J.uniforms_from_schema(schema, nz), None, abi.RM_RENDER_FAST)
        want = fb.present(2)
        color = torch.from_numpy(fb.download(0)).cuda()
        nd = torch.from_numpy(fb.download(1)).cuda()
        sharp = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        ctx.present_device(color.data_ptr(), None, W, H, 2, sharp.data_ptr())
        ctx.sync()
        assert not np.array_equal(want, sharp.cpu().numpy())  # the blur is on
        for parts in (1, 3, 8):
            for part in range(parts):
                rows = shard.owned_rows(H, parts, part, stripe_rows)
                if len(rows) == 0:
                    continue
                out = torch.zeros((len(rows), W, 4), dtype=torch.uint8, device="cuda")
                ctx.present_striped_rows(color.data_ptr(), nd.data_ptr(), W, H, 2, stripe_rows, parts, part, out.data_ptr())
                ctx.sync()
                assert np.array_equal(out.cpu().numpy(), want[rows]), (stripe_rows, parts, part)
    finally:
        ctx.close()
