"""The rule behind the row culling of primitive tables (rm_params.hpp rm_cull_cell, the routine the grid-building kernel runs
per cell), checked on the CPU against the fold itself: for random tables and random balls, the fold of the rows that stay equals
the fold of every row at every sampled point of the ball, bit for bit (float64 and float32 restatements of the folds; min and max
are exact in any precision; a far smooth-union row is dropped only where its rounding of the running value is the identity)."""
import numpy as np
import pytest

from raymarching_engine_amd import abi, native, scene as S


def random_table(rng, kind):
    """0: spheres and boxes under unions; 1: under unions, subtractions and intersections; 2: smooth unions among them; 3: mostly
    smooth unions, of several radii"""
    sc = S.CsgScene()
    n = int(rng.integers(12, 100))
    p = [[1.0, 0, 0, 0], [0.6, 0, 0.25, 0.15], [0.45, 0.3, 0.15, 0.1], [0.1, 0.8, 0.05, 0.05]][kind]
    for _ in range(n):
        [sc.union, lambda: sc.smooth_union(float(rng.uniform(0.05, 0.5))), sc.subtract, sc.intersect][int(rng.choice(4, p=p))]()
        c = rng.uniform(-2, 2, 3)
        if rng.uniform() < 0.6:
            sc.sphere(c, float(rng.uniform(0.2, 0.7)))
        else:
            sc.box(c, rng.uniform(0.1, 0.6, 3))
    return sc


def fold(sc, pts, keep, ft=np.float64):
    """fold of the rows with keep[i] at the points [m, 3] (the operators of Sdf<RM_SCENE_TABLE>::eval) in the float type ft, every
    operation rounded once (ft = float32: the parity build's fold; the shapes' distances are correctly rounded -- the rule only
    needs them to be floats within its allowance of the true distance)"""
    d = None
    for i, nd in enumerate(sc._nodes):
        if not keep[i]:
            continue
        q = pts - np.asarray(nd.center, np.float64)[None]
        if nd.prim == abi.RM_PRIM_SPHERE:
            di = np.sqrt((q * q).sum(-1)) - np.float64(np.float32(nd.size[0]))
        else:
            b = np.abs(q) - np.asarray(nd.size, np.float32).astype(np.float64)[None]
            di = np.sqrt((np.maximum(b, 0.0) ** 2).sum(-1)) + np.minimum(b.max(-1), 0.0)
        di = di.astype(ft)
        if d is None:
            d = di
        elif nd.op == abi.RM_OP_UNION:
            d = np.minimum(d, di)
        elif nd.op == abi.RM_OP_SMOOTH_UNION:
            k = ft(np.float32(nd.k))  # examples/smooth-tree.glsl:20-22
            h = np.clip(ft(0.5) + (ft(0.5) * (di - d)) / k, ft(0.0), ft(1.0))
            d = (di + h * (d - di)) - (k * h) * (ft(1.0) - h)
        elif nd.op == abi.RM_OP_SUBTRACT:
            d = np.maximum(d, -di)
        else:
            d = np.maximum(d, di)
    return d


@pytest.mark.parametrize("ft", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_dropped_rows_are_no_ops_everywhere_in_the_ball(kind, ft):
    rng = np.random.default_rng(100 + kind)
    dropped = total = 0
    for trial in range(12):
        sc = random_table(rng, kind)
        # float32 centres / sizes, as the table holds them
        for nd in sc._nodes:
            nd.center = tuple(float(np.float32(v)) for v in nd.center)
        n = len(sc._nodes)
        for _ in range(60):
            scale = float(10.0 ** rng.uniform(-0.5, 2.5))
            c = rng.normal(0, 1, 3) * scale if rng.uniform() < 0.6 else rng.uniform(-2.5, 2.5, 3)
            rad = float(10.0 ** rng.uniform(-2, 0)) * max(1.0, 0.1 * np.linalg.norm(c))
            # float32: with the allowance the grid's build kernel gives the distances' rounding (rm_params.hpp rm_cull_margin)
            keep = native.cull_cell(sc, c, rad, 0.0 if ft is np.float64 else 1e-4 + 1.2e-7 * (n + 8) * (float(np.abs(c).max()) + rad + 3.0))
            assert keep[0]
            u = rng.normal(0, 1, (400, 3))
            u /= np.linalg.norm(u, axis=1, keepdims=True)
            pts = c[None] + u * (rad * rng.uniform(0, 1, (400, 1)) ** (1 / 3))
            pts = np.concatenate([pts, c[None] + u[:100] * rad])  # and on the sphere itself
            if ft is np.float32:
                pts = pts.astype(np.float32).astype(np.float64)
                pts = pts[np.linalg.norm(pts - c[None], axis=1) <= rad]  # (rounded to float32 a point on the sphere may have left the ball)
            full, part = fold(sc, pts, [True] * n, ft), fold(sc, pts, keep, ft)
            assert np.array_equal(full, part), (kind, trial, int((full != part).sum()), float(np.abs(full - part).max()))
            dropped += n - sum(keep)
            total += n
    assert dropped > (0.3 if kind < 2 else 0.15) * total  # and the rule does drop rows


def test_rejects_domain_rows():
    sc = S.CsgScene().smooth_union(0.2)
    sc.repeat((1.0, 1.0, 1.0))
    for _ in range(12):
        sc.sphere((0, 0, 0), 0.3)
    with pytest.raises(native.RmError):
        native.cull_cell(sc, (0, 0, 0), 0.1)


# ---- spheres under ONE smooth-union radius (BASELINE's CSG-64), round 4 ---------------------------------------------------------------
# A far row of a smooth union is not a no-op -- the fast fold's d' = di - fl(di - d) rounds d to the grid of (di - d) -- unless d
# already lies on a grid at least as coarse; the rule tracks that per cell and drops the rows whose rounding is provably the identity.
# Here: the rule against an fp32 restatement of the FAST fold (rm_device.hpp sphere_row1 / smooth_row: v_fma_f32, v_sqrt_f32 taken as
# correctly rounded -- any rounding of the square root leaves the argument intact: it only needs di to be a float), bit for bit.

def _fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)  # (a 2^-29 chance of a double rounding: both folds share it)


def fast_fold_fp32(C, R, k, pts, keep):
    f32 = np.float32
    hik = f32(0.5) * (f32(1.0) / f32(k))

    def row(i):
        q = pts - C[i][None]
        s = _fma32(q[:, 2], q[:, 2], _fma32(q[:, 1], q[:, 1], q[:, 0] * q[:, 0]))
        return (np.sqrt(s.astype(np.float64)).astype(f32) - R[i]).astype(f32)

    d = row(0)
    for i in range(1, len(R)):
        if not keep[i]:
            continue
        di = row(i)
        t = (di - d).astype(f32)
        h = np.minimum(np.maximum(_fma32(np.full_like(t, hik), t, np.full_like(t, 0.5)), f32(0)), f32(1))
        d = _fma32(-h, _fma32(np.full_like(t, f32(k)), (f32(1) - h).astype(f32), t), di)
    return d


def parity_fold_fp32(C, R, k, pts, keep):
    """the parity build's fold (rm_device.hpp op_smooth_union<PM>: h = clamp(0.5 + 0.5 (di - d) / k), mix(di, d, h) - k h (1 - h) with mix
    = x + a (y - x), every operation rounded once; the distance correctly rounded): a far row is d' = fl(di + fl(d - di)) here too"""
    f32 = np.float32
    P64, C64 = pts.astype(np.float64), C.astype(np.float64)

    def row(i):
        return (np.sqrt(((P64 - C64[i][None]) ** 2).sum(1)) - float(R[i])).astype(f32)

    d = row(0)
    for i in range(1, len(R)):
        if not keep[i]:
            continue
        di = row(i)
        h = np.minimum(np.maximum(f32(0.5) + ((f32(0.5) * (di - d)).astype(f32) / f32(k)).astype(f32), f32(0)), f32(1)).astype(f32)
        mix = (di + (h * (d - di).astype(f32)).astype(f32)).astype(f32)
        d = (mix - ((f32(k) * h).astype(f32) * (f32(1) - h).astype(f32)).astype(f32)).astype(f32)
    return d


@pytest.mark.parametrize("fold", ["fast", "parity"])
def test_far_rows_of_a_smooth_union_are_dropped_only_where_their_rounding_is_the_identity(fold):
    fold_fp32 = fast_fold_fp32 if fold == "fast" else parity_fold_fp32
    rng = np.random.default_rng(404)
    dropped = total = 0
    for trial in range(14):
        sc = S.csg64() if trial == 0 else S.CsgScene()
        if trial:
            n, k = int(rng.integers(16, 90)), float(np.float32(rng.uniform(0.05, 0.4)))
            spacing = float(rng.uniform(0.6, 1.2))
            sc.smooth_union(k)
            for _ in range(n):
                sc.sphere(tuple(rng.uniform(-2, 2, 3) * spacing), float(rng.uniform(0.15, 0.5)))
        C = np.array([nd.center for nd in sc._nodes], np.float32)
        R = np.array([nd.size[0] for nd in sc._nodes], np.float32)
        k = np.float32(sc._nodes[1].k)
        for nd, c in zip(sc._nodes, C):
            nd.center = tuple(float(v) for v in c)
        n = len(R)
        for _ in range(40):
            # cells the size the grid has (64^3 over the scene, and the coarser outer levels), near surfaces, inside shapes, far outside
            rad = float(rng.choice([0.02, 0.05, 0.07, 0.15, 0.4]))
            i = int(rng.integers(0, n))
            u = rng.normal(size=3)
            u /= np.linalg.norm(u)
            c = C[i].astype(np.float64) + u * (float(R[i]) + float(rng.choice([-0.3, -0.1, -1e-3, 0.0, 1e-4, 0.01, 0.05, 0.3, 1.0, 4.0, 30.0])))
            margin = 1e-4 + 1.2e-7 * (n + 8) * (float(np.abs(c).max()) + rad + 3.0)  # rm_cull_margin at this cell's magnitude
            keep = native.cull_cell(sc, c, rad, margin)
            assert keep[0]
            v = rng.normal(0, 1, (300, 3))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            pts = (c[None] + v * (rad * rng.uniform(0, 1, (300, 1)) ** (1 / 3))).astype(np.float32)
            full, part = fold_fp32(C, R, k, pts, [True] * n), fold_fp32(C, R, k, pts, keep)
            same = full.view(np.uint32) == part.view(np.uint32)
            assert same.all(), (trial, int((~same).sum()), rad, [j for j in range(n) if not keep[j]][:8])
            dropped += n - sum(keep)
            total += n
    assert dropped > 0.25 * total, (dropped, total)  # the rule does drop rows: about half of them in cells of the grid's size
