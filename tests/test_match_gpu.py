"""GPU parity tests of the matchers (bit-exact index / distance / depth results)."""
import numpy as np
import pytest

from conftest import golden
from lpslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(hiplib):
    return hiplib.Context(320, 240, 400, 1.2, 4, max_images=4)


def test_bf_golden(ctx, oracle):
    g = golden("g4_bf.npz")
    ctx.set_descriptors(0, g["q"]); ctx.set_descriptors(1, g["t"])
    ctx.match_bf(0, 1)
    bi, bd, sd = ctx.bf_knn2(0)
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_dist"]) and np.array_equal(sd, g["second_dist"])
    mq, mt, md = ctx.bf_matches(0, 1, 50, 0.9, True)
    assert np.array_equal(mq, g["mq"]) and np.array_equal(mt, g["mt"]) and np.array_equal(md, g["md"])


@pytest.mark.parametrize("nq,nt", [(1, 1), (63, 65), (64, 256), (257, 300), (411, 412), (5, 0), (0, 7)])
def test_bf_ragged_sizes(ctx, oracle, nq, nt):
    rng = np.random.default_rng(nq * 1000 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8); t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    if nq > 3 and nt > 3:
        t[1] = t[2]; q[0] = t[2]; q[3] = q[0]                       # exact ties: first minimum must win
    ctx.set_descriptors(2, q); ctx.set_descriptors(3, t)
    ctx.match_bf(2, 3)
    g = ctx.bf_knn2(2)
    o = oracle.match_bf_knn2(q, t)
    assert all(np.array_equal(a, b) for a, b in zip(o, g))


@pytest.mark.parametrize("nq,nt", [(1, 1), (63, 65), (257, 300), (411, 412), (5, 0), (0, 7)])
def test_bf_one_call_keyframe_comparison(ctx, oracle, nq, nt):
    """lpslam_hip_match_bf_descriptors (upload + both directions + kernel-delivered read-back, one wait) returns what the three
    calls it replaces return, and what the oracle's filtered matcher returns."""
    rng = np.random.default_rng(nq * 977 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = np.concatenate([q[: nt // 2] ^ (rng.random((min(nt // 2, nq), 32)) < 0.05).astype(np.uint8), rng.integers(0, 256, (nt - min(nt // 2, nq), 32), dtype=np.uint8)])[:nt] if nt else np.zeros((0, 32), np.uint8)
    ctx.set_descriptors(2, q)
    for ratio, cross in ((0.0, False), (0.8, True), (0.75, True)):
        one = ctx.match_bf_descriptors(2, 3, t, 60, ratio, cross)
        ctx.set_descriptors(3, t); ctx.match_bf(2, 3)
        three = ctx.bf_matches(2, 3, 60, ratio, cross)
        assert all(np.array_equal(a, b) for a, b in zip(one, three))
        if nq and nt:
            o = oracle.match_bf(q, t, 60, ratio, cross)
            assert np.array_equal(one[0], o[0]) and np.array_equal(one[1], o[1])


def test_bf_against_stored_descriptor_sets(ctx):
    """lpslam_hip_match_bf_stored: an image slot against many descriptor sets kept on the device, one launch and one wait -- per set
    the matches of lpslam_hip_match_bf_descriptors; sets of different sizes, an empty one, a replaced one, a dropped one"""
    rng = np.random.default_rng(77)
    q = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    sets = {}
    for key, nt in ((5, 280), (9, 0), (2, 411), (40, 17), (41, 300)):
        base = q[rng.permutation(300)[: min(nt, 300)]] ^ (rng.random((min(nt, 300), 32)) < 0.04).astype(np.uint8)
        sets[key] = np.concatenate([base, rng.integers(0, 256, (nt - len(base), 32), dtype=np.uint8)])[:nt]
        ctx.desc_store_put(key, sets[key])
    sets[41] = rng.integers(0, 256, (120, 32), dtype=np.uint8)
    ctx.desc_store_put(41, sets[41])                         # replaces the first set under this key
    ctx.set_descriptors(2, q)
    keys = [2, 9, 41, 5, 40]
    for ratio, cross in ((0.0, False), (0.75, True)):
        got = ctx.match_bf_stored(2, keys, 60, ratio, cross)
        for key, g in zip(keys, got):
            want = ctx.match_bf_descriptors(2, 3, sets[key], 60, ratio, cross)
            assert all(np.array_equal(a, b) for a, b in zip(g, want)), key
    assert sum(len(g[0]) for g in got) > 100
    ctx.desc_store_drop(5)
    with pytest.raises(Exception):
        ctx.match_bf_stored(2, [5], 60, 0.0, False)
    assert ctx.match_bf_stored(2, [], 60, 0.0, False) == []


@pytest.fixture(scope="module")
def big_ctx(hiplib):
    """2100 keypoints / 8 levels: 2124 descriptor slots, i.e. train sets of more than two 1024-descriptor LDS tiles."""
    return hiplib.Context(1280, 720, 2100, 1.2, 8, max_images=2)


@pytest.mark.parametrize("nq,nt", [(2000, 2000), (1025, 1023), (1024, 2048), (1500, 1025), (7, 2049), (2124, 2124)])
def test_bf_multi_tile_sizes(big_ctx, oracle, nq, nt):
    """The shapes the bench and the tracker run: the train set spans several 1024-descriptor LDS tiles, so the cross-tile
    (distance, index) merge decides best index, best distance and second distance.  Exact ties are planted across tile
    boundaries (first minimum must win, the second distance of a tied query is the tie's distance)."""
    rng = np.random.default_rng(nq * 7919 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8); t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    if nt > 1030:
        t[1030] = t[5]; q[0] = t[5]                  # tie across the first tile boundary, exact match: idx 5, dist 0, second 0
        t[1024] = t[1023]; q[1] = t[1023]            # tie between the last descriptor of tile 0 and the first of tile 1
        q[2] = t[1030] ^ np.uint8(1)                 # one bit off a tied pair: best 5 (first minimum), second equal
        t[nt - 1] = t[0]; q[3] = t[0]                # first descriptor against the very last one
    if nt > 2048:
        t[2048] = t[1024]; q[4] = t[1024]; t[nt - 1] = t[0] if nt > 2049 else t[nt - 1]     # three-way tie over three tiles (1023, 1024, 2048)
    big_ctx.set_descriptors(0, q); big_ctx.set_descriptors(1, t)
    big_ctx.match_bf(0, 1)
    g = big_ctx.bf_knn2(0)
    o = oracle.match_bf_knn2(q, t)
    for name, a, b in zip(("best_idx", "best_dist", "second_dist"), o, g):
        assert np.array_equal(a, b), name
    if nt > 1030:
        assert g[0][0] == 5 and g[1][0] == 0 and g[2][0] == 0
        assert g[0][1] == 1023 and g[2][1] == 0
        if nt != 2049:
            assert g[0][3] == 0 and g[2][3] == 0
    if nt > 2048:
        assert g[0][4] == 1023 and g[1][4] == 0 and g[2][4] == 0


def test_bf_golden_full_size(big_ctx):
    """G8: the benchmark's 2000 x 2000 shape against the committed oracle output (tools/make_golden.py)."""
    g = golden("g8_bf2000.npz")
    big_ctx.set_descriptors(0, g["q"]); big_ctx.set_descriptors(1, g["t"])
    big_ctx.match_bf(0, 1)
    bi, bd, sd = big_ctx.bf_knn2(0)
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_dist"]) and np.array_equal(sd, g["second_dist"])
    mq, mt, md = big_ctx.bf_matches(0, 1, 100, 0.9, True)
    assert np.array_equal(mq, g["mq"]) and np.array_equal(mt, g["mt"]) and np.array_equal(md, g["md"])


def test_bf_symmetry_property(hiplib):
    """Full-size 2000 x 2000: distance symmetry d(q,t)=d(t,q) and self-match distance 0."""
    c = hiplib.Context(1280, 720, 2000, 1.2, 8, max_images=2)
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (2000, 32), dtype=np.uint8); b = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    c.set_descriptors(0, a); c.set_descriptors(1, b)
    c.match_bf(0, 1); c.match_bf(1, 0)
    ia, da, _ = c.bf_knn2(0); ib, db, _ = c.bf_knn2(1)
    assert da.min() == db.min()
    mutual = ib[ia] == np.arange(2000)
    assert (db[ia][mutual] == da[mutual]).all()
    c.match_bf(0, 0)
    i0, d0, s0 = c.bf_knn2(0)
    assert (d0 == 0).all() and (i0 == np.arange(2000)).all() and (s0 > 0).all()


def test_stereo_golden_and_oracle(hiplib, oracle):
    g = golden("g6_stereo.npz")
    c = hiplib.Context(320, 240, 400, 1.2, 4, max_images=2)
    c.upload(0, g["left"]); c.upload(1, g["right"])
    c.extract(2)
    k = synth.intrinsics(320, 240)
    c.match_stereo(0, 1, k["fxb"], k["baseline"])
    xr, dep, bi = c.stereo(0)
    assert np.array_equal(xr, g["x_right"]) and np.array_equal(dep, g["depth"]) and np.array_equal(bi, g["best_idx"])


def test_stereo_golden_720(hiplib):
    """G9: 1280x720 / 2000 keypoints / 8 levels against the committed oracle output: keypoints, descriptors, x_right, depth."""
    import hashlib
    g = golden("g9_stereo720.npz")
    l, r = synth.StereoSequence(1280, 720, 9).frame(2)
    assert hashlib.sha256(l.tobytes()).hexdigest() == str(g["sha_left"]) and hashlib.sha256(r.tobytes()).hexdigest() == str(g["sha_right"])
    c = hiplib.Context(1280, 720, 2000, 1.2, 8, max_images=2)
    c.upload(0, l); c.upload(1, r)
    c.extract(2)
    k = synth.intrinsics(1280, 720)
    c.match_stereo(0, 1, k["fxb"], k["baseline"])
    for slot, kk, dd in ((0, g["kl"], g["dl"]), (1, g["kr"], g["dr"])):
        kp, desc = c.keypoints(slot)
        assert len(kp) == len(kk)
        for f in kk.dtype.names:
            assert np.array_equal(kp[f], kk[f]), f
        assert np.array_equal(desc, dd)
    xr, dep, bi = c.stereo(0)
    assert np.array_equal(xr, g["x_right"]) and np.array_equal(dep, g["depth"]) and np.array_equal(bi, g["best_idx"])
    assert int((dep > 0).sum()) == int(g["n_valid"])
    c.close()


def test_stereo_full_size(hiplib, oracle):
    seq = synth.StereoSequence(1280, 720, 2)
    l, r = seq.frame(1)
    p = oracle.params(2000, 1.2, 8)
    kl, dl, _, pl = oracle.extract(l, p, True); kr, dr, _, pr = oracle.extract(r, p, True)
    k = synth.intrinsics(1280, 720)
    oxr, odep, obi, nv = oracle.match_stereo(pl, pr, p, kl, dl, kr, dr, k["fxb"], k["baseline"])
    c = hiplib.Context(1280, 720, 2000, 1.2, 8, max_images=4)
    c.upload(2, l); c.upload(3, r)                                  # non-zero slots
    c.upload(0, r); c.upload(1, l)
    c.extract(4)
    c.match_stereo_strided(0, 1, 2, 2, k["fxb"], k["baseline"])
    xr, dep, bi = c.stereo(2)
    assert np.array_equal(xr, oxr) and np.array_equal(dep, odep) and np.array_equal(bi, obi)
    ok = dep > 0
    assert ok.sum() == nv and ok.sum() > 100
    assert (xr[ok] <= kl["x"][ok]).all()                            # positive disparity
    # swapped pair (right as left): disparities are negative, nothing may survive the range test with depth > 0 ... mostly
    xr2, dep2, _ = c.stereo(0)
    assert (dep2 > 0).sum() < ok.sum()


def test_temporal_match_batch(hiplib, oracle):
    w, h, F = 320, 240, 4
    seq = synth.StereoSequence(w, h, 5, n_points=1500)
    c = hiplib.Context(w, h, 400, 1.2, 4, max_images=2 * F)
    p = oracle.params(400, 1.2, 4)
    descs = []
    for f in range(F):
        l, r = seq.frame(f)
        c.upload(2 * f, l); c.upload(2 * f + 1, r)
        descs.append(oracle.extract(l, p)[1])
    c.extract(2 * F)
    c.match_bf_strided(2, 0, 2, F - 1)
    for f in range(1, F):
        g = c.bf_knn2(2 * f)
        o = oracle.match_bf_knn2(descs[f], descs[f - 1])
        assert all(np.array_equal(a, b) for a, b in zip(o, g))


# ---- projection matching (match::projection) ----------------------------------------------------------------------------
def _proj_case(hiplib, oracle, w, h, kpts, levels, seed, radius_scale=15.0, n_queries=None, jitter=6.0):
    """Frame k+1 is matched from the keypoints of frame k displaced by a known flow (as the motion model would predict)."""
    seq = synth.StereoSequence(w, h, seed)
    l0, r0 = seq.frame(0); l1, r1 = seq.frame(1)
    ctx = hiplib.Context(w, h, kpts, 1.2, levels, max_images=4)
    for i, im in enumerate((l0, r0, l1, r1)):
        ctx.upload(i, im)
    ctx.extract(4)
    k = synth.intrinsics(w, h)
    ctx.match_stereo_strided(0, 1, 2, 2, k["fxb"], k["baseline"])
    kp0, d0 = ctx.keypoints(0); kp1, d1 = ctx.keypoints(2)
    xr1, _, _ = ctx.stereo(2)
    xr0, _, _ = ctx.stereo(0)
    rng = np.random.default_rng(seed)
    n = len(kp0) if n_queries is None else min(n_queries, len(kp0))
    q = np.zeros(n, hiplib.PROJ_QUERY_DTYPE)
    q["x"] = kp0["x"][:n] + rng.normal(0, jitter, n).astype(np.float32); q["y"] = kp0["y"][:n] + rng.normal(0, jitter, n).astype(np.float32)
    q["x_right"] = np.where(xr0[:n] > 0, xr0[:n] + (q["x"] - kp0["x"][:n]), -1.0)
    q["radius"] = (radius_scale * np.float32(1.2) ** kp0["octave"][:n]).astype(np.float32)
    q["min_level"] = kp0["octave"][:n] - 1; q["max_level"] = kp0["octave"][:n] + 1
    return ctx, q, d0[:n], kp0[:n], kp1, d1, xr1


@pytest.mark.parametrize("w,h,kpts,levels,radius", [(320, 240, 400, 4, 15.0), (640, 480, 1000, 8, 15.0), (640, 480, 1000, 8, 60.0)])
def test_projection_match_parity(hiplib, oracle, w, h, kpts, levels, radius):
    ctx, q, qd, kp0, kp1, d1, xr1 = _proj_case(hiplib, oracle, w, h, kpts, levels, 3, radius)
    for use_stereo in (False, True):
        gi, gd, gn = ctx.match_projection(2, q, qd, 100, 0.8, None, use_stereo)
        oi, od, on = oracle.match_projection(kp1, d1, xr1 if use_stereo else None, w, h, q, qd, 100, 0.8)
        assert gn == on and np.array_equal(gi, oi) and np.array_equal(gd[gi >= 0], od[oi >= 0])       # identical match pairs
        assert gn > 0.3 * len(q)
        m = gi[gi >= 0]
        assert len(np.unique(m)) == len(m)                                                         # one landmark per keypoint
    # orientation histogram filter on top
    fg, ng = hiplib.match_orientation_filter(kp0["angle"], kp1["angle"], gi)
    fo, no = oracle.match_orientation_filter(kp0["angle"], kp1["angle"], oi)
    assert ng == no and np.array_equal(fg, fo) and 0 < ng <= gn


def test_projection_match_when_every_keypoint_is_a_candidate(hiplib, oracle):
    """1280 x 720, 2000 keypoints, windows that cover the image and every level: each query's wavefront lists more candidates than its LDS
    list holds at a time (the second pass runs in the middle of the scan as well as at its end)."""
    w, h = 1280, 720
    ctx, q, qd, kp0, kp1, d1, xr1 = _proj_case(hiplib, oracle, w, h, 2000, 8, 11, n_queries=96)
    assert len(kp1) > 1500
    q["radius"] = 4000.0; q["min_level"] = -1; q["max_level"] = -1; q["x_right"] = -1.0
    taken = np.zeros(len(kp1), np.uint8); taken[::7] = 1
    for tk in (None, taken):
        gi, gd, gn = ctx.match_projection(2, q, qd, 100, 0.8, tk, False)
        oi, od, on = oracle.match_projection(kp1, d1, None, w, h, q, qd, 100, 0.8, taken=tk)
        assert gn == on and np.array_equal(gi, oi) and np.array_equal(gd[gi >= 0], od[oi >= 0])
        assert gn > 0


def test_projection_match_sequential_semantics_and_rescans(hiplib, oracle):
    """Many queries compete for the same few keypoints (identical predictions, wide window): later queries must see the
    earlier assignments; short candidate lists get exhausted, which exercises the single-query re-scan."""
    ctx, q, qd, kp0, kp1, d1, xr1 = _proj_case(hiplib, oracle, 320, 240, 400, 4, 5, radius_scale=80.0, n_queries=120, jitter=0.0)
    q["x"][:] = q["x"][0]; q["y"][:] = q["y"][0]; q["min_level"][:] = -1; q["max_level"][:] = -1
    qd = np.repeat(qd[:1], len(q), axis=0)                                      # every query has the same descriptor
    gi, gd, gn = ctx.match_projection(2, q, qd, 256, 1.0, None, False)
    oi, od, on = oracle.match_projection(kp1, d1, None, 320, 240, q, qd, 256, 1.0)
    assert gn == on and np.array_equal(gi, oi)
    m = gi[gi >= 0]
    assert len(m) > 20 and len(np.unique(m)) == len(m) and (np.diff(gd[gi >= 0]) >= 0).all()       # greedy: distances only grow
    taken = np.zeros(len(kp1), np.uint8); taken[m[:10]] = 1
    gi2, _, _ = ctx.match_projection(2, q, qd, 256, 1.0, taken, False)
    oi2, _, _ = oracle.match_projection(kp1, d1, None, 320, 240, q, qd, 256, 1.0, taken)
    assert np.array_equal(gi2, oi2) and not np.isin(gi2[gi2 >= 0], m[:10]).any()
    e_i, _, e_n = ctx.match_projection(2, q[:0], qd[:0])
    assert e_n == 0 and len(e_i) == 0


def test_fuse_match_parity(hiplib, oracle):
    """match::fuse: landmarks projected into a keyframe, chi-square gate of the keypoint's level, no exclusivity."""
    w, h = 640, 480
    ctx, q, qd, kp0, kp1, d1, xr1 = _proj_case(hiplib, oracle, w, h, 1000, 8, 7, radius_scale=3.0, jitter=1.2)
    q["min_level"] = np.maximum(kp0["octave"] - 1, 0); q["max_level"] = kp0["octave"]
    isq = (np.float32(1.0) / (np.array(ctx.scale, np.float32) ** 2)).astype(np.float32)
    for use_stereo in (False, True):
        gi, gd, gn = ctx.match_fuse(2, q, qd, 50, use_stereo)
        oi, od, on = oracle.match_fuse(kp1, d1, xr1 if use_stereo else None, w, h, isq, q, qd, 50)
        assert gn == on > 50 and np.array_equal(gi, oi) and np.array_equal(gd[gi >= 0], od[oi >= 0])
    assert len(np.unique(gi[gi >= 0])) <= (gi >= 0).sum()                     # duplicates are allowed here


def test_area_match_parity_with_stealing(hiplib, oracle):
    """match::area (monocular initialiser): level-0 keypoints, a better later query takes the keypoint from an earlier one."""
    w, h = 640, 480
    ctx, q, qd, kp0, kp1, d1, xr1 = _proj_case(hiplib, oracle, w, h, 1000, 8, 9, radius_scale=40.0, jitter=3.0)
    lvl0 = kp0["octave"] == 0
    q = q[lvl0].copy(); qd = qd[lvl0].copy()
    q["radius"] = 100.0; q["min_level"] = 0; q["max_level"] = 0
    # the first 40 queries get a few bits flipped and come back unspoilt at the end: the later, better ones take the keypoint
    n0 = len(q)
    extra = q[:40].copy(); extra_d = qd[:40].copy()
    qd[:40, 0] ^= 0x15; qd[:40, 7] ^= 0x81
    q = np.concatenate([q, extra]); qd = np.concatenate([qd, extra_d])
    gi, gd, gn = ctx.match_area(2, q, qd, 50, 0.9)
    oi, on = oracle.match_area(kp1, d1, w, h, q, qd, 50, 0.9)
    assert gn == on > 30 and np.array_equal(gi, oi)
    m = gi[gi >= 0]
    assert len(np.unique(m)) == len(m)                                        # one query per keypoint at the end
    assert (kp1["octave"][m] == 0).all()
    stolen = (gi[:40] == -1) & (gi[n0:n0 + 40] >= 0)
    assert stolen.sum() >= 5                                                  # the steal rule was exercised


def test_frame_readback_in_one_round_trip(hiplib):
    """lpslam_hip_get_frame returns what get_keypoints + get_stereo return."""
    w, h = 640, 480
    k = synth.intrinsics(w, h)
    l, r = synth.StereoSequence(w, h, 2, n_points=4000).frame(0)
    ctx = hiplib.Context(w, h, 800, 1.2, 5, max_images=2)
    ctx.upload(0, l); ctx.upload(1, r); ctx.extract(2)
    ctx.match_stereo(0, 1, k["fxb"], k["baseline"])
    kp, desc = ctx.keypoints(0)
    xr, dep, _ = ctx.stereo(0)
    fkp, fdesc, fxr, fdep = ctx.frame(0)
    assert len(fkp) == len(kp) > 100 and fkp.tobytes() == kp.tobytes() and np.array_equal(fdesc, desc)
    assert np.array_equal(fxr, xr) and np.array_equal(fdep, dep) and (fdep > 0).sum() > 20
    # delivered ahead of time behind a prefetched front end (lpslam_hip_prefetch_frame): the same bytes; a copy that the slot's next
    # extraction voids is not served (another frame goes into the slot, the late read-back must show IT)
    l2, r2 = synth.StereoSequence(w, h, 2, n_points=4000).frame(7)
    with ctx.prefetch():
        ctx.upload(0, l); ctx.upload(1, r); ctx.extract(2)
        ctx.match_stereo(0, 1, k["fxb"], k["baseline"])
        ctx.prefetch_frame(0)
    ctx.prefetch_join()
    pkp, pdesc, pxr, pdep = ctx.frame(0)
    assert pkp.tobytes() == kp.tobytes() and np.array_equal(pdesc, desc) and np.array_equal(pxr, xr) and np.array_equal(pdep, dep)
    with ctx.prefetch():
        ctx.prefetch_frame(0)                              # delivered ...
        ctx.upload(0, l2); ctx.upload(1, r2); ctx.extract(2)      # ... and voided: the slot holds another frame now
        ctx.match_stereo(0, 1, k["fxb"], k["baseline"])
    ctx.prefetch_join()
    qkp, qdesc, qxr, qdep = ctx.frame(0)
    kp2, desc2 = ctx.keypoints(0)
    xr2, dep2, _ = ctx.stereo(0)
    assert qkp.tobytes() == kp2.tobytes() and np.array_equal(qdesc, desc2) and np.array_equal(qxr, xr2) and qkp.tobytes() != kp.tobytes()
    ctx.prefetch_frame(0); ctx.prefetch_frame(0)            # two deliveries in a row (main stream): the second waits for the first
    rkp, rdesc, rxr, rdep = ctx.frame(0)
    assert rkp.tobytes() == kp2.tobytes() and np.array_equal(rdep, dep2)
    # the view: the same data without the copy into caller buffers, from a prefetched delivery and from a direct read-back
    for ahead in (True, False):
        if ahead:
            ctx.prefetch_frame(0)
        vkp, vdesc, vxr, vdep = ctx.frame_view(0)
        assert vkp.tobytes() == kp2.tobytes() and np.array_equal(vdesc, desc2) and np.array_equal(vxr, xr2) and np.array_equal(vdep, dep2)
