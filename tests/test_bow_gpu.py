"""GPU parity tests of the bag-of-words kernels through the C ABI against the CPU oracle: the tree walk (word, weight, node per
keypoint: array_equal) and match::bow_tree (match indices and distances: array_equal, ties and taken targets included)."""
import numpy as np
import pytest

from bow_util import read_vocab
from lpslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(hiplib):
    v = read_vocab()
    ctx = hiplib.Context(640, 480, 1000, 1.2, 4, max_images=2)
    voc = hiplib.Vocabulary(ctx, v["k"], v["L"], v["parent"], v["desc"], v["weight"], v["is_leaf"])
    assert (voc.k, voc.L, voc.n_nodes, voc.n_words) == (10, 3, len(v["parent"]), 1000)
    return v, ctx, voc


def test_transform_parity_on_extracted_keypoints(hiplib, oracle, setup):
    v, ctx, voc = setup
    seq = synth.StereoSequence(640, 480, 4, n_points=6000)
    l, r = seq.frame(3)
    ctx.upload(0, l); ctx.upload(1, r)
    ctx.extract(2)
    for slot in (0, 1):
        _, desc = ctx.keypoints(slot)
        assert len(desc) > 500
        for lu in (0, 1, 2, 4):
            w, ww, nd = voc.transform(slot, lu)
            ow, oww, ond = oracle.bow_transform(v, desc, lu)
            assert np.array_equal(w, ow) and np.array_equal(ww, oww) and np.array_equal(nd, ond), (slot, lu)
    # descriptors from host memory, some of them exactly on tree nodes (ties between children)
    rng = np.random.default_rng(3)
    d = rng.integers(0, 256, (777, 32), dtype=np.uint8)
    d[:100] = v["desc"][rng.integers(0, len(v["desc"]), 100)]
    w, ww, nd = voc.transform_host(d, 1)
    ow, oww, ond = oracle.bow_transform(v, d, 1)
    assert np.array_equal(w, ow) and np.array_equal(ww, oww) and np.array_equal(nd, ond)


def test_vocabulary_arguments_are_checked(hiplib, setup):
    v, ctx, _ = setup
    bad_parent = v["parent"].copy(); bad_parent[5] = 900                      # a parent that does not precede its child
    with pytest.raises(hiplib.LpslamHipError):
        hiplib.Vocabulary(ctx, 10, 3, bad_parent, v["desc"], v["weight"], v["is_leaf"])
    bad_leaf = v["is_leaf"].copy(); bad_leaf[0] = 1                            # node 1 has children
    with pytest.raises(hiplib.LpslamHipError):
        hiplib.Vocabulary(ctx, 10, 3, v["parent"], v["desc"], v["weight"], bad_leaf)


@pytest.mark.parametrize("levels_up,ratio,thr", [(1, 0.75, 50), (2, 0.9, 50), (3, 1.0, 100), (0, 0.6, 30)])
def test_bow_tree_match_parity(hiplib, oracle, setup, levels_up, ratio, thr):
    """two consecutive frames: keypoints of the first (a third of them switched off, as keypoints without a landmark are) against
    the second's; with levels_up = 3 the node is the root (L = 3): every query scans every target, lists get exhausted and the
    single-query rescans run"""
    v, ctx, voc = setup
    seq = synth.StereoSequence(640, 480, 4, n_points=6000)
    ctx.upload(0, seq.frame(5)[0]); ctx.upload(1, seq.frame(6)[0])
    ctx.extract(2)
    _, da = ctx.keypoints(0); _, db = ctx.keypoints(1)
    rng = np.random.default_rng(11)
    db = db.copy(); db[40:60] = db[20:40]                                       # duplicated descriptors: ties between targets
    _, _, na = voc.transform_host(da, levels_up); _, _, nb = voc.transform_host(db, levels_up)
    na = na.copy(); na[rng.random(len(na)) < 0.33] = -1
    taken = (rng.random(len(db)) < 0.1).astype(np.uint8)
    gi, gd, gn = hiplib.match_bow_tree(ctx, da, na, db, nb, thr, ratio, taken)
    oi, od, on = oracle.bow_tree_match(da, na, db, nb, thr, ratio, taken)
    assert gn == on and on > 50 and np.array_equal(gi, oi) and np.array_equal(gd[gi >= 0], od[oi >= 0])
    assert np.all(gi[na < 0] == -1) and not np.any(taken[gi[gi >= 0]])
    m = gi[gi >= 0]
    assert len(np.unique(m)) == len(m)                                          # a target is matched once


def test_bow_tree_match_of_several_sets_in_one_call(hiplib, oracle, setup):
    """one keyframe's descriptors against five target sets (different frames, ragged sizes, an empty set, a set without a node in common,
    one with a `taken` mask) in ONE call: every set's result is that of the single-set call and of the oracle"""
    v, ctx, voc = setup
    seq = synth.StereoSequence(640, 480, 4, n_points=6000)
    ctx.upload(0, seq.frame(5)[0])
    ctx.extract(1)
    _, da = ctx.keypoints(0)
    rng = np.random.default_rng(3)
    _, _, na = voc.transform_host(da, 1)
    na = na.copy(); na[rng.random(len(na)) < 0.3] = -1
    tds, tns, tks = [], [], []
    for i, fr in enumerate((6, 7, 9)):
        ctx.upload(1, seq.frame(fr)[0]); ctx.extract_range(1, 1)
        _, db = ctx.keypoints(1)
        db = db[: len(db) - 37 * i].copy()
        _, _, nb = voc.transform_host(db, 1)
        tds.append(db); tns.append(nb.copy()); tks.append((rng.random(len(db)) < 0.1).astype(np.uint8) if i == 1 else None)
    tds.append(np.zeros((0, 32), np.uint8)); tns.append(np.zeros(0, np.int32)); tks.append(None)                 # an empty set
    tds.append(tds[0].copy()); tns.append(np.full(len(tns[0]), -1, np.int32)); tks.append(None)                  # no target is under a node
    multi = hiplib.match_bow_tree_multi(ctx, da, na, tds, tns, 50, 0.75, tks)
    assert len(multi) == 5
    total = 0
    for (gi, gd, gn), td, tn, tk in zip(multi, tds, tns, tks):
        si, sd, sn = hiplib.match_bow_tree(ctx, da, na, td, tn, 50, 0.75, tk) if len(td) else (np.full(len(na), -1, np.int32), None, 0)
        assert gn == sn and np.array_equal(gi, si)
        if len(td):
            oi, od, on = oracle.bow_tree_match(da, na, td, tn, 50, 0.75, tk if tk is not None else np.zeros(len(td), np.uint8))
            assert gn == on and np.array_equal(gi, oi) and np.array_equal(gd[gi >= 0], od[oi >= 0])
        total += gn
    assert total > 100 and multi[3][2] == 0 and multi[4][2] == 0
