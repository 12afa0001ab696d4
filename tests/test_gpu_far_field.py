"""Adversarial tests of the exact far-field exits of the march (round 4; rm_device.hpp far_escape / far_need / far_shadow_escape /
clear_miss, Sdf<RM_SCENE_MANDELBULB>::far_jump_at).

castRay (raymarcher.frag:163-170) marches a fixed number of steps with no distance bound; an escaping ray's end state is the
overflow's.  The kernels set such a ray to that state when hand-derived bounds say the remaining steps reach it.  The tests of
round 3 drew random rays, which sit well inside those bounds; the ones here CONSTRUCT the cases the bounds were derived for, per
kind of scene:
  * scenes of every size -- R' from 1e-3 to 10 (the overflow is absolute, so small scenes need the most steps);
  * rays started tangentially (p . dir = 0, and a hair either side of it) at the jump's radius and at the edges of its two far
    tiers (2 500 x and 2.5e11 x far_r2), a hair inside and outside each;
  * |dir|^2 at the edges the exits admit (0.98 .. 1.02) and just beyond them;
  * step budgets at every tier +- 1;
  * rays that are going to miss the scene's sphere by the margin the miss asks for, +- a hair;
  * shadow rays whose light is as far away as the ray is certain to get, +- one binade, and at the exponent limits.
Every end point must have the bits of the stepwise march (RM_RENDER_NO_FAR_JUMP), in the fast AND the parity build -- for the
shadow rays the outcome of the comparison that reads the end point (raymarcher.frag:362-363).  tests/test_far_bounds_cpu.py
asserts the tiers themselves.  >= 1e6 rays per kind and build (RM_FAR_RAYS scales it; tools/r04_fuzz.sh runs 30 x under three seeds).
"""
import os

import numpy as np
import pytest

from raymarching_engine_amd import abi
from raymarching_engine_amd import scene as S

pytestmark = pytest.mark.gpu

STRICT, FAST, NJ = abi.RM_RENDER_STRICT, abi.RM_RENDER_FAST, abi.RM_RENDER_NO_FAR_JUMP
SEED_OFFSET = int(os.environ.get("RM_RANDOM_SEED", "0"))
SCALE = float(os.environ.get("RM_FAR_RAYS", "1"))
TIER_BUDGETS = (24.0, 49.0, 50.0, 51.0, 62.0, 63.0, 64.0, 70.0, 71.0, 72.0, 89.0, 90.0, 91.0, 99.0, 100.0, 101.0, 128.0)
BULB_BUDGETS = (8.0, 15.0, 16.0, 17.0, 21.0, 22.0, 23.0, 29.0, 30.0, 31.0, 64.0)


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


@pytest.fixture(scope="module")
def ctx():
    from raymarching_engine_amd import native

    c = native.Context(0)
    yield c
    c.close()


def unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def perpendicular(rng, u):
    """a random unit vector at a right angle to each row of u"""
    w = rng.normal(size=u.shape)
    w -= (w * u).sum(1, keepdims=True) * u
    return unit(w)


def tier_rays(rng, jump_r, n):
    """Rays at the radii where a far tier starts (1, 50 and 5e5 times the jump's radius: sqrt of far_r2, 2 500 far_r2, 2.5e11 far_r2),
    a hair either side, tangential or nearly so, |dir|^2 at and around the admitted range."""
    tier = rng.choice([1.0, 1.0, 1.0, 50.0, 50.0, 5e5], size=(n, 1))
    hair = rng.choice([-1e-3, -1e-6, -1e-7, 0.0, 1e-7, 1e-6, 1e-3, 0.1, 1.0], size=(n, 1))
    u = unit(rng.normal(size=(n, 3)))
    o = u * (jump_r * tier * (1.0 + hair))
    lean = rng.choice([-1e-2, -1e-5, -1e-7, 0.0, 0.0, 0.0, 1e-7, 1e-5, 1e-2, 0.3, 1.0], size=(n, 1))  # radial part of the direction
    d = unit(perpendicular(rng, u) + lean * u)
    dd = rng.choice([0.9799, 0.9801, 0.99, 1.0, 1.0, 1.0, 1.01, 1.0199, 1.0201], size=(n, 1))
    d = d * np.sqrt(dd)
    z = rng.random(n) < 0.05  # a zero component (0 x Inf = NaN in the end state; not renormalised)
    d[z, rng.integers(0, 3, int(z.sum()))] = 0.0
    return np.concatenate([o, d], 1).astype(np.float32)


def miss_rays(rng, jump_r, n):
    """Rays whose line ahead passes the origin at about the distance from which a ray counts as going to miss (m^2 = 0.4 far_r2 +
    0.05), a hair either side, started between there and 30 times the jump's radius, moving inward."""
    far_r2 = jump_r * jump_r
    m = np.sqrt(0.4 * far_r2 + 0.05) * (1.0 + rng.choice([-0.05, -1e-3, -1e-6, 0.0, 1e-6, 1e-3, 0.05, 0.5], size=(n, 1)))
    v = unit(rng.normal(size=(n, 3)))
    w = perpendicular(rng, v)
    start = np.maximum(m, jump_r * rng.choice([0.64, 0.8, 1.0, 1.0 + 1e-6, 3.0, 10.0, 31.0, 31.7], size=(n, 1)))
    t = np.sqrt(np.maximum(start * start - m * m, 0.0))
    o = v * m - w * t
    dd = rng.choice([0.9801, 1.0, 1.0, 1.0199], size=(n, 1))
    return np.concatenate([o, w * np.sqrt(dd)], 1).astype(np.float32)


def check_cast(ctx, sc, rays, budgets, build, what):
    h = ctx.create_scene(sc)
    jumped = 0
    for steps in budgets:
        a = ctx.probe(h, abi.RM_PROBE_CAST_RAY, rays, steps, build)
        b = ctx.probe(h, abi.RM_PROBE_CAST_RAY, rays, steps, build | NJ)
        bad = ~same_bits(a, b).all(-1)
        assert not bad.any(), f"{what}, {steps:g} steps: {int(bad.sum())} of {len(rays)} end points differ from the stepwise march, e.g. ray {rays[bad][0]} -> {a[bad][0]} against {b[bad][0]}"
        jumped += int((~np.isfinite(a)).any(-1).sum())
    h.destroy()
    return jumped


def table_scene(rng, scale, rows, smooth):
    """`rows` spheres and boxes within `scale` of the origin; returns the scene and R' as rm_api.hip table_far_field computes it"""
    sc = S.CsgScene()
    reach, kmax = 0.0, 0.0
    for i in range(rows):
        if i:
            op = int(rng.integers(0, 4)) if smooth else int(rng.choice([0, 2, 3]))
            if op == 0: sc.union()
            elif op == 1:
                k = float(rng.uniform(0.05, 0.5) * scale)
                sc.smooth_union(k)
                kmax = max(kmax, k)
            elif op == 2: sc.subtract()
            else: sc.intersect()
        c = rng.uniform(-1.0, 1.0, 3) * scale
        if rng.random() < 0.6:
            r = float(rng.uniform(0.2, 0.9) * scale)
            sc.sphere(tuple(c), r)
            ext = r
        else:
            b = rng.uniform(0.15, 0.8, 3) * scale
            sc.box(tuple(c), tuple(b))
            ext = float(np.linalg.norm(b.astype(np.float32).astype(np.float64)))
        reach = max(reach, float(np.linalg.norm(np.float32(c).astype(np.float64))) + ext)
    return sc, 2.0 * (reach + (1.01 * kmax if kmax > 0 else 0.0)) + 1.0


@pytest.mark.parametrize("build", [FAST, STRICT], ids=["fast", "strict"])
def test_tables_at_the_edges_of_the_far_tiers(ctx, build):
    rng = np.random.default_rng(8101 + SEED_OFFSET)
    n = int(40000 * SCALE)
    total = jumped = 0
    for it in range(12):
        scale = [1e-3, 1e-2, 0.1, 0.3, 1.0, 3.0][it % 6]
        sc, jump_r = table_scene(rng, scale, rows=int(rng.integers(1, 12)) if it % 2 else int(rng.integers(16, 40)), smooth=it % 3 != 0)
        rays = np.concatenate([tier_rays(rng, jump_r, n), miss_rays(rng, jump_r, n // 2)])
        jumped += check_cast(ctx, sc, rays, TIER_BUDGETS, build, f"table {it} (scale {scale:g}, jump radius {jump_r:.4g})")
        total += len(rays) * len(TIER_BUDGETS)
    assert total >= 1e6 * min(SCALE, 1.0) and jumped > 0.2 * total, (total, jumped)


@pytest.mark.parametrize("build", [FAST, STRICT], ids=["fast", "strict"])
def test_mandelbulb_at_the_edges_of_its_jump(ctx, build):
    """The jump asks for r^2 > max(bailout^2, 4) and 30 / 22 / 16 steps from r^2 >= 4 / 1e4 / 1e12: rays on those spheres, a hair
    either side, tangential.  Power 8 and, since round 4, the other powers (whose far field is the same function)."""
    rng = np.random.default_rng(8202 + SEED_OFFSET)
    n = int(30000 * SCALE)
    total = jumped = 0
    for it in range(10):
        bailout = float(rng.choice([0.5, 1.5, 2.0, 2.0, 3.0, 10.0]))
        power = 8.0 if it % 2 == 0 else float(rng.choice([2.0, 3.0, 5.0, 9.0]))
        sc = S.Mandelbulb(power=power, iterations=int(rng.integers(1, 9)), bailout=bailout)
        rays = []
        for radius in (max(bailout, 2.0), 1e2, 1e6):  # sqrt of 4 (or bailout^2), 1e4, 1e12
            u = unit(rng.normal(size=(n, 3)))
            hair = rng.choice([-1e-3, -1e-6, 0.0, 1e-7, 1e-6, 1e-4, 1e-3, 0.1, 1.0], size=(n, 1))
            lean = rng.choice([-1e-2, -1e-6, 0.0, 0.0, 1e-6, 1e-2, 1.0], size=(n, 1))
            d = unit(perpendicular(rng, u) + lean * u) * np.sqrt(rng.choice([0.9799, 0.9801, 1.0, 1.0, 1.0199, 1.0201], size=(n, 1)))
            z = rng.random(n) < 0.05
            d[z, rng.integers(0, 3, int(z.sum()))] = 0.0
            rays.append(np.concatenate([u * radius * (1.0 + hair), d], 1))
        rays = np.concatenate(rays).astype(np.float32)
        jumped += check_cast(ctx, sc, rays, BULB_BUDGETS, build, f"mandelbulb {it} (power {power:g}, bailout {bailout:g})")
        total += len(rays) * len(BULB_BUDGETS)
    assert total >= 1e6 * min(SCALE, 1.0) and jumped > 0.2 * total, (total, jumped)


@pytest.mark.parametrize("build", [FAST, STRICT], ids=["fast", "strict"])
def test_sponge_rotation_fractal_and_sphere_grid_at_the_edges_of_the_far_tiers(ctx, build):
    rng = np.random.default_rng(8303 + SEED_OFFSET)
    n = int(24000 * SCALE)
    for kind in ("menger", "kifs_box", "sphere_grid"):
        total = jumped = 0
        for it in range(5):
            if kind == "menger":
                sc, jump_r = S.MengerSponge(iterations=float(rng.integers(1, 6))), 5.0   # far_jump: far_r2 = 25
            elif kind == "kifs_box":
                s, off = float(rng.uniform(0.35, 0.7)), float(rng.choice([1e-3, 0.05, 0.7, 1.6]))
                sc = S.KifsBox(iterations=float(rng.integers(1, 12)), scale=s, angles=tuple(rng.uniform(-1.5, 1.5, 3)), offset=off)
                jump_r = 2.0 * np.sqrt(3.0) * (off * s / (1.0 - s) + 1.0) + 1.0        # rm_api.hip kifs_far_field
            else:
                big, c = float(rng.choice([1e-3, 0.1, 1.0, 4.0])), rng.uniform(-1, 1, 3) * float(rng.choice([1e-3, 1.0, 8.0]))
                sc = S.SphereGridFractal(big_sphere_size=big, iterations=float(rng.integers(1, 7)), grid_scale=float(rng.uniform(0.2, 0.6)), big_sphere_center=tuple(float(x) for x in c))
                jump_r = 2.0 * (float(np.linalg.norm(np.float32(c).astype(np.float64))) + big) + 1.0  # rm_scene_create
            rays = np.concatenate([tier_rays(rng, jump_r, n), miss_rays(rng, jump_r, n // 2)])
            jumped += check_cast(ctx, sc, rays, TIER_BUDGETS, build, f"{kind} {it} (jump radius {jump_r:.4g})")
            total += len(rays) * len(TIER_BUDGETS)
        assert total >= 1e6 * min(SCALE, 1.0) and jumped > 0.2 * total, (kind, total, jumped)


@pytest.mark.parametrize("build", [FAST, STRICT], ids=["fast", "strict"])
def test_shadow_rays_at_the_limits_of_their_certain_comparison(ctx, build):
    """far_shadow_escape stops a long table's shadow ray once `distance(result, adj) >= distance(pos, adj)` is certain: outside the
    jump's radius, not moving inward, at least 8 steps to go, the end point finite (1.00716 left + log2 r <= 63.3) and far enough
    (log2 r + left - 3 >= need_e, 2^need_e >= distance(pos, adj) + |adj|).  Lights placed as far away as the ray is promised to
    get, +- two binades; lights beyond 2^100 and 1e30 (no promise); scenes of 1e-7 .. 1 with the light inside them (need_e at its
    lower clamp); budgets 6 .. 64; rays aimed at the light (as the kernel casts them) and not.  RM_PROBE_CAST_SHADOW returns the
    comparison: the same with the exit and without it."""
    rng = np.random.default_rng(8404 + SEED_OFFSET)
    n = int(60000 * SCALE)
    total = stopped = 0
    for it in range(8):
        scale = [1e-7, 1e-3, 0.1, 1.0][it % 4]
        sc, jump_r = table_scene(rng, scale, rows=int(rng.integers(16, 48)), smooth=it % 2 == 0)
        h = ctx.create_scene(sc)
        for left in (6.0, 7.0, 8.0, 9.0, 20.0, 40.0, 57.0, 59.0, 62.0, 63.0, 64.0):
            u = unit(rng.normal(size=(n, 3)))
            # radii out to where the finite-end condition cuts in for this budget: log2 r <= 63.3 - 1.00716 left, +- a hair
            edge = 2.0 ** (63.295 - 1.00716 * left)
            r = np.where(rng.random((n, 1)) < 0.5, jump_r * rng.choice([1.0 - 1e-6, 1.0 + 1e-6, 2.0, 50.0, 1e6], size=(n, 1)),
                         np.minimum(edge * rng.choice([0.5, 0.99, 1.0, 1.01, 2.0], size=(n, 1)), 1e29))
            pos = u * r
            lean = rng.choice([-1e-3, -1e-7, 0.0, 1e-7, 1e-3, 0.5, 1.0, 1.0], size=(n, 1))
            d = unit(perpendicular(rng, u) + lean * u)
            # the light: 2^(log2 r + left - 3 + j) away along the ray, j = -2 .. 2, or absurdly far, or inside the scene
            j = rng.choice([-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 2.0], size=(n, 1))
            far = np.minimum(r * 2.0 ** (left - 3.0 + j), 1e37)
            dist = np.where(rng.random((n, 1)) < 0.6, far, rng.choice([1e-9, 1e-3, 1.0, 2.0 ** 99, 2.0 ** 100.5, 1e30, 1e31], size=(n, 1)))
            aimed = rng.random((n, 1)) < 0.7
            adj = np.where(aimed, pos + d * dist, unit(rng.normal(size=(n, 3))) * dist)
            near = rng.random(n) < 0.1
            adj[near] = rng.uniform(-1, 1, (int(near.sum()), 3)) * scale  # the light inside the scene, the ray cast from far away: need_e from |pos|
            rays = np.concatenate([pos, d, adj], 1).astype(np.float32)
            a = ctx.probe(h, abi.RM_PROBE_CAST_SHADOW, rays, left, build)
            b = ctx.probe(h, abi.RM_PROBE_CAST_SHADOW, rays, left, build | NJ)
            bad = a != b
            assert not bad.any(), f"table {it} (scale {scale:g}), {left:g} steps: {int(bad.sum())} of {n} shadow comparisons differ, e.g. {rays[bad][0]}: {a[bad][0]} against {b[bad][0]}"
            total += n
            stopped += int((a == 1.0).sum())
        h.destroy()
    assert total >= 1e6 * min(SCALE, 1.0) and 0.05 * total < stopped < 0.95 * total, (total, stopped)
