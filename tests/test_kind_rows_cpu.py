"""A kind as a SHAPE of a primitive table (RM_PRIM_KIND, round 4; the reference takes any GLSL sdf(): RenderJobExecutor.tsx:121-127).
CPU side: the composer's rows and text, the oracle's fold.  The goldens of the two composed scenes (tests/golden/*csg_bulb_cut*,
*csg_lattice_ball*) come from the reference's own shader with the composer's text spliced in (oracle/gl/gen_golden.py) and are
checked by the generic golden tests."""
import numpy as np
import pytest

import golden_cases as GC
from oracle import oracle as O
from raymarching_engine_amd import abi, scene as S


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def test_a_table_of_one_kind_row_is_that_kind():
    """CsgScene().shape(kind) alone folds to kind(p - 0): the oracle gives the kind's own distances bit for bit, translated rows
    the distances at the translated points."""
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.uniform(-1.5, 1.5, (3000, 3)), rng.normal(size=(500, 3)) * 20.0]).astype(np.float32)
    for kind in (S.Mandelbulb(power=8.0, iterations=6, bailout=2.0), S.Mandelbulb(power=3.0, iterations=3, bailout=2.0), S.SphereLattice(0.75, 0.25)):
        want = O.eval_sdf(kind, pts)
        assert same_bits(O.eval_sdf(S.CsgScene().shape(kind), pts), want).all()
        c = np.array([0.25, -0.5, 0.125], np.float32)
        assert same_bits(O.eval_sdf(S.CsgScene().shape(kind, tuple(c)), pts), O.eval_sdf(kind, pts - c)).all()


def test_kind_rows_fold_like_any_shape():
    """max(bulb, box) and max(that, -sphere): the fold's operators on the kind row's term, in numpy on the oracle's own terms."""
    pts = GC.sdf_points(2000, seed=11)
    bulb = S.Mandelbulb(power=8.0, iterations=5, bailout=2.0)
    sc = GC.build_scene("csg_bulb_cut")
    d_bulb = O.eval_sdf(bulb, pts)
    d_box = O.eval_sdf(S.CsgScene().box((0.0, 0.0, 0.25), (1.25, 1.25, 0.75)), pts)
    d_sph = O.eval_sdf(S.CsgScene().sphere((0.5, 0.375, -0.5), 0.375), pts)
    want = np.maximum(np.maximum(d_bulb, d_box), -d_sph)
    got = O.eval_sdf(sc, pts)
    ok = same_bits(got, want) | np.isnan(d_bulb)  # (a NaN term: the oracle's max keeps the other operand, numpy's hands the NaN on)
    assert ok.all()
    assert [p.type & 0xff for p in sc.prims()] == [abi.RM_PRIM_KIND, abi.RM_PRIM_BOX, abi.RM_PRIM_SPHERE]
    assert sc.params()[:3] == [8.0, 5.0, 2.0] and sc.prims()[0].size[0] == float(abi.RM_SCENE_MANDELBULB)


def test_the_composer_emits_the_kinds_own_text_and_refuses_what_the_table_cannot_hold():
    sc = GC.build_scene("csg_lattice_ball")
    text = sc.glsl()
    assert "float rmKindSdf(vec3 p)" in text and "rmKindSdf(p - vec3(0.125, 0.0, 0.0))" in text and text.count("float sdf(") == 1
    with pytest.raises(ValueError):
        S.CsgScene().shape(S.MengerSponge())  # (its text is the reference's example file: not a shape the composer can emit)
    with pytest.raises(ValueError):
        S.CsgScene().shape(S.Mandelbulb(iterations=4)).shape(S.Mandelbulb(iterations=5))  # one parameter block per table
    with pytest.raises(ValueError):
        S.CsgScene().shape(S.Mandelbulb()).shape(S.SphereLattice())
    # the surface of a kind row: the nearest term rule covers it (rmSurfaceIndex calls rmKindSdf too)
    red = S.Surface(diffuse=(0.9, 0.1, 0.1))
    sc2 = S.CsgScene().sphere((1.875, 0.375, 0.375), 0.5).union().shape(S.SphereLattice(0.75, 0.25), surface=red)
    assert "rmKindSdf" in sc2.material_glsl() and sc2.prims()[1].type >> 16 == 1
    # at a lattice sphere's surface the lattice's term is the smallest; at the big sphere's centre (the middle of a lattice cell) its own
    near_lattice, near_sphere = np.array([[0.0, 0.0, 0.26]], np.float32), np.array([[1.875, 0.375, 0.375]], np.float32)
    assert O.material(sc2, near_lattice)[0, 0] == np.float32(0.9) and O.material(sc2, near_sphere)[0, 0] == np.float32(0.6)
