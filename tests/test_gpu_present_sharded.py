"""rm_present_sharded through ctypes: ONE process, several native contexts (all on GPU 0: a gpurun box has one), each holding
one part of a frame's 8-row stripes; the canvas it assembles -- rows tone-mapped (no depth of field) or packed (depth of
field) per context, copied to the first context, put in image order, blurred there -- equals rm_present of the same samples
on one framebuffer, byte for byte (display.frag:16-64; index.tsx:25-59)."""
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
