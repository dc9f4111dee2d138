"""GPU parity tests of the ORB front end, through the C ABI, against the CPU oracle and the golden fixtures.
Bar: bit-exact pyramid bytes, FAST candidates (x, y, score, order), keypoint fields and descriptors."""
import numpy as np
import pytest

from conftest import golden
from lpslam_amd import synth

pytestmark = pytest.mark.gpu


def _assert_same_keypoints(okp, od, gkp, gd):
    assert len(okp) == len(gkp)
    for f in okp.dtype.names:
        assert np.array_equal(okp[f], gkp[f]), "keypoint field %s" % f
    assert np.array_equal(od, gd)


def _run(hiplib, img, kpts, levels, scale=1.2, ini=20, mn=7):
    h, w = img.shape
    ctx = hiplib.Context(w, h, kpts, scale, levels, ini, mn, max_images=1)
    ctx.upload(0, img)
    ctx.extract(1)
    return ctx


@pytest.mark.parametrize("w,h,kpts,levels", [(160, 120, 150, 3), (333, 251, 400, 4), (640, 480, 1000, 8), (1280, 720, 2000, 8), (1920, 1080, 2000, 8)])
def test_extract_parity(hiplib, oracle, w, h, kpts, levels):
    img = synth.random_image(w, h, seed=w)
    p = oracle.params(kpts, 1.2, levels)
    okp, od, occ, opyr = oracle.extract(img, p, True)
    ctx = _run(hiplib, img, kpts, levels)
    for l in range(levels):
        assert np.array_equal(ctx.pyramid_level(0, l), opyr[l]), "pyramid level %d" % l
        oc = oracle.fast_level(opyr[l])
        gc = ctx.candidates(0, l)
        assert len(gc) == occ[l] == len(oc)
        assert np.array_equal(oc, gc), "FAST candidates level %d" % l
    gkp, gd = ctx.keypoints(0)
    _assert_same_keypoints(okp, od, gkp, gd)
    assert len(gkp) >= min(kpts, 50)


def test_golden_vectors(hiplib):
    g = golden("g3_orb.npz")
    ctx = _run(hiplib, g["image"], 150, 3)
    kp, desc = ctx.keypoints(0)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(kp[f], g[f]), f
    assert np.array_equal(desc, g["desc"])
    c = ctx.candidates(0, 0)
    g2 = golden("g2_fast.npz")
    assert np.array_equal(c["x"], g2["x"]) and np.array_equal(c["y"], g2["y"]) and np.array_equal(c["score"], g2["score"])
    g1 = golden("g1_pyramid.npz")
    ctx1 = _run(hiplib, g1["image"], 60, 3)
    assert np.array_equal(ctx1.pyramid_level(0, 1), g1["level1"]) and np.array_equal(ctx1.pyramid_level(0, 2), g1["level2"])


def test_blank_and_weak_images(hiplib, oracle):
    flat = np.full((240, 320), 100, np.uint8)
    ctx = _run(hiplib, flat, 300, 4)
    kp, desc = ctx.keypoints(0)
    assert len(kp) == 0 and len(desc) == 0
    yy, xx = np.mgrid[0:240, 0:320]
    weak = flat.copy(); weak[100:, 150:] = 112
    weak = (weak + (xx * 7 + yy * 13) % 3).astype(np.uint8)           # only min-threshold corners exist
    okp, od, _, _ = oracle.extract(weak, oracle.params(300, 1.2, 4))
    ctx = _run(hiplib, weak, 300, 4)
    gkp, gd = ctx.keypoints(0)
    assert len(okp) > 0 and (okp["response"] < 20).all()
    _assert_same_keypoints(okp, od, gkp, gd)


@pytest.mark.parametrize("ini,mn,scale,levels,kpts", [(30, 5, 1.2, 5, 700), (12, 12, 1.5, 3, 1200), (20, 7, 1.2, 1, 500)])
def test_parameter_variants(hiplib, oracle, ini, mn, scale, levels, kpts):
    """Feature.* are runtime parameters: reference default is 3 levels / 1200 keypoints."""
    img = synth.random_image(480, 360, seed=77)
    okp, od, _, _ = oracle.extract(img, oracle.params(kpts, scale, levels, ini, mn))
    ctx = _run(hiplib, img, kpts, levels, scale, ini, mn)
    gkp, gd = ctx.keypoints(0)
    _assert_same_keypoints(okp, od, gkp, gd)


def test_batch_of_images_and_slots(hiplib, oracle):
    """A batch is processed in one set of launches; every image slot gives what it gives alone."""
    w, h, kpts, levels, n = 320, 240, 400, 4, 6
    seq = synth.StereoSequence(w, h, 9, n_points=1500)
    imgs = [im for k in range(n // 2) for im in seq.frame(k)]
    ctx = hiplib.Context(w, h, kpts, 1.2, levels, max_images=n)
    for i, im in enumerate(imgs):
        ctx.upload(i, im)
    ctx.extract(n)
    p = oracle.params(kpts, 1.2, levels)
    for i, im in enumerate(imgs):
        okp, od, _, _ = oracle.extract(im, p)
        gkp, gd = ctx.keypoints(i)
        _assert_same_keypoints(okp, od, gkp, gd)
    with pytest.raises(hiplib.LpslamHipError):
        ctx.extract(n + 1)


def test_full_size_properties(hiplib):
    """1280x720 / 2000 keypoints (BASELINE configs[1]): size-independent invariants of the result."""
    seq = synth.StereoSequence(1280, 720, 1)
    l, r = seq.frame(3)
    ctx = hiplib.Context(1280, 720, 2000, 1.2, 8, max_images=2)
    ctx.upload(0, l); ctx.upload(1, r)
    ctx.extract(2)
    kp, desc = ctx.keypoints(0)
    assert 2000 <= len(kp) <= 2000 + 3 * 8
    assert (np.diff(kp["octave"]) >= 0).all()                      # levels are emitted in order
    quota = ctx.quota
    counts = np.bincount(kp["octave"], minlength=8)
    assert all(q <= c <= q + 3 for q, c in zip(quota, counts))
    sc = np.array(ctx.scale)[kp["octave"]]
    assert (kp["x"] >= 22 * sc - 1e-3).all() and (kp["x"] <= (np.array(ctx.level_w)[kp["octave"]] - 23) * sc + 1e-3).all()
    assert (kp["angle"] >= 0).all() and (kp["angle"] < 360).all() and (kp["response"] >= 7).all()
    assert len({(a, b, c) for a, b, c in zip(kp["x"], kp["y"], kp["octave"])}) == len(kp)     # no duplicates
    # idempotence: a second extraction of the same resident frame gives the same bytes
    ctx.extract(2)
    kp2, desc2 = ctx.keypoints(0)
    assert np.array_equal(kp, kp2) and np.array_equal(desc, desc2)


@pytest.mark.parametrize("w,h,kpts,levels", [(320, 240, 300, 4), (640, 480, 1000, 8)])
def test_extract_with_camera_mask(hiplib, oracle, w, h, kpts, levels):
    """Camera masks (mask_type Radial / Image of the reference's camera configuration): cells with a masked corner are skipped,
    corners at masked positions dropped after the threshold fallback; per eye (even slots: left mask, odd slots: right mask)."""
    img = synth.random_image(w, h, seed=77)
    yy, xx = np.mgrid[0:h, 0:w]
    radial = np.where((xx - w // 2) ** 2 + (yy - h // 2) ** 2 <= (0.42 * w) ** 2, 255, 0).astype(np.uint8)
    stripes = np.where(((xx // 37) + (yy // 29)) % 3 == 0, 0, 255).astype(np.uint8)
    c = hiplib.Context(w, h, kpts, 1.2, levels, max_images=2)
    c.upload(0, img); c.upload(1, img)
    c.set_mask(0, radial); c.set_mask(1, stripes)
    c.extract(2)
    p = oracle.params(kpts, 1.2, levels)
    for slot, mask in ((0, radial), (1, stripes)):
        ok, od, occ, _ = oracle.extract(img, p, mask=mask)
        gk, gd = c.keypoints(slot)
        assert len(gk) == len(ok) and len(ok) > 20
        for f in ok.dtype.names:
            assert np.array_equal(gk[f], ok[f]), f
        assert np.array_equal(gd, od)
        assert (mask[gk["y"].astype(int), gk["x"].astype(int)] > 0).all()          # nothing survives inside the mask
        for l in range(levels):
            sc = np.float32(1.2) ** l
            oc = oracle.fast_level(oracle_pyramid(oracle, img, p)[l], 20, 7, mask=mask, scale=c.scale[l])
            gc = c.candidates(slot, l)
            assert len(gc) == len(oc) and np.array_equal(np.sort(gc, order=("y", "x")), np.sort(oc, order=("y", "x"))), l
    # removing the masks restores the plain result
    c.set_mask(0, None); c.set_mask(1, None)
    c.extract(2)
    ok, od, _, _ = oracle.extract(img, p)
    gk, gd = c.keypoints(1)
    assert np.array_equal(gd, od) and np.array_equal(gk["x"], ok["x"])
    c.close()


def oracle_pyramid(oracle, img, p):
    return oracle.extract(img, p, True)[3]


def test_prefetch_stream_gives_the_same_frame(hiplib, oracle):
    """lpslam_hip_prefetch_begin / _end / _join: the front end of the NEXT stereo frame is enqueued on the context's second stream
    (from a helper thread, as the tracker does) while the main stream matches the current one; after the join the prefetched slots
    hold exactly what the ordinary path gives.  Misuse is refused."""
    import threading
    w, h, kpts, levels = 640, 480, 1000, 4
    k = synth.intrinsics(w, h)
    seq = synth.StereoSequence(w, h, 5, n_points=4000)
    f0, f1 = seq.frame(0), seq.frame(1)
    ref = hiplib.Context(w, h, kpts, 1.2, levels, max_images=6)
    for s, im in ((0, f0[0]), (1, f0[1]), (2, f1[0]), (3, f1[1])):
        ref.upload(s, im)
    ref.extract_range(0, 4)
    ref.match_stereo(0, 1, k["fxb"], k["baseline"]); ref.match_stereo(2, 3, k["fxb"], k["baseline"])
    want = [ref.frame(0), ref.frame(2)]

    ctx = hiplib.Context(w, h, kpts, 1.2, levels, max_images=6)
    ctx.upload(0, f0[0]); ctx.upload(1, f0[1])
    ctx.extract_range(0, 2); ctx.match_stereo(0, 1, k["fxb"], k["baseline"])
    err = []

    def helper():                                    # the next frame, on the prefetch stream
        try:
            with ctx.prefetch():
                ctx.upload(2, f1[0]); ctx.upload(3, f1[1])
                ctx.extract_range(2, 2); ctx.match_stereo(2, 3, k["fxb"], k["baseline"])
        except Exception as e:                       # noqa: BLE001
            err.append(e)
    th = threading.Thread(target=helper); th.start()
    for _ in range(5):                               # meanwhile the main stream works on the current frame
        ctx.match_bf(0, 1)
    got0 = ctx.frame(0)
    th.join()
    assert not err
    ctx.prefetch_join()
    got1 = ctx.frame(2)
    for g, wnt in ((got0, want[0]), (got1, want[1])):
        for a, b in zip(g, wnt):
            assert np.array_equal(a, b)
    with pytest.raises(hiplib.LpslamHipError):       # end without begin
        hiplib._check(hiplib.load().lpslam_hip_prefetch_end(ctx.h))


def test_async_uploads_from_page_locked_frames(hiplib, oracle):
    """lpslam_hip_upload_images_async: frames in page-locked memory of the context (host_alloc), in the caller's own array pinned in
    place (host_register), with a row stride, and in pageable memory all reach their slots; an extraction waits for its slots'
    copies on the device, and a copy into a slot starts only after the work enqueued before it (which may still read the slot)."""
    w, h, kpts, levels = 640, 480, 800, 6
    imgs = [synth.random_image(w, h, seed=70 + i) for i in range(6)]
    p = oracle.params(kpts, 1.2, levels)
    want = [oracle.extract(im, p, True)[:2] for im in imgs]
    ctx = hiplib.Context(w, h, kpts, 1.2, levels, max_images=6)
    owned = [ctx.host_frame(imgs[0]), ctx.host_frame(imgs[1])]
    mine = imgs[2].copy(); ctx.host_register(mine)
    wide = np.zeros((h, w + 64), np.uint8); wide[:, :w] = imgs[3]; ctx.host_register(wide)
    ctx.upload_async(0, owned)
    ctx.upload_async(2, [mine])
    ctx.upload_async(3, [wide[:, :w]])                       # stride w + 64
    ctx.upload_async(4, [imgs[4], imgs[5]])                  # pageable: staged by the runtime
    ctx.extract_range(0, 6)
    for i in range(6):
        gk, gd = ctx.keypoints(i)
        _assert_same_keypoints(want[i][0], want[i][1], gk, gd)
    # re-use of a slot: the extraction enqueued BEFORE the new upload still sees the old frame, the one after it the new frame
    for rounds in range(3):
        a, b = owned
        ctx.upload_async(0, [a]); ctx.extract_range(0, 1)
        ctx.upload_async(0, [b])                             # ordered behind the extraction above
        ka, da = ctx.keypoints(0)                            # synchronises: results of the FIRST extraction
        _assert_same_keypoints(want[0][0], want[0][1], ka, da)
        ctx.extract_range(0, 1)
        kb, db = ctx.keypoints(0)
        _assert_same_keypoints(want[1][0], want[1][1], kb, db)
    ctx.sync()
    ctx.host_unregister(mine); ctx.host_unregister(wide)
    ctx.host_free(owned[0])
    with pytest.raises(hiplib.LpslamHipError):
        ctx.host_free(mine)                                  # not a block of host_alloc
    ctx.close()                                              # frees owned[1] with the context


def test_mapping_reserve_changes_speed_only(hiplib):
    """lpslam_hip_set_mapping_reserve: the extraction kernels run as queued grids of persistent workgroups that leave the reserved
    compute units of every XCD to the bundle adjustments beside them (no stream carries a CU mask).  Keypoints and descriptors are
    bit for bit the same with and without it; a bundle adjustment created on such a context gives the same result; out-of-range
    values are refused."""
    from lpslam_amd import hip
    w, h = 640, 480
    ctx = hip.Context(w, h, 1000, 1.2, 8, max_images=4)
    seq = synth.StereoSequence(w, h, 3, n_points=5000)
    imgs = [seq.frame(i)[e] for i in range(2) for e in range(2)]

    def run():
        for i, im in enumerate(imgs):
            ctx.upload(i, im)
        ctx.extract_range(0, 4)
        return [ctx.keypoints(i) for i in range(4)]
    p = synth.ba_problem(12, 600, 4000, w, h, seq_id=3)

    def solve():
        b = hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], hip.ba_obs_array(p), p["cam"])
        log = b.optimize(True, 6)
        poses, points = b.state()
        b.close()
        return [tuple(l) for l in log], poses, points
    base, base_ba = run(), solve()
    for r in (4, 8, 12, 16, 0):
        ctx.set_mapping_reserve(r)
        got, got_ba = run(), solve()
        for (k0, d0), (k1, d1) in zip(base, got):
            assert np.array_equal(k0, k1) and np.array_equal(d0, d1)
        assert got_ba[0] == base_ba[0] and np.array_equal(got_ba[1], base_ba[1]) and np.array_equal(got_ba[2], base_ba[2])
    with pytest.raises(hip.LpslamHipError):
        ctx.set_mapping_reserve(17)
    ctx.close()


def test_mapping_reserve_at_the_benchmarked_shape(hiplib, oracle):
    """The configuration `bench.py` quotes its value on: 1280x720, 2000 keypoints, 8 levels, 32 images per extraction launch, the
    front end confined to half of the chip (reserve 16; 12 is the other setting DESIGN.md quotes), a full local window (50 keyframes /
    5000 landmarks / 40 k observations, 10 iterations) solved on the same context.  Bit for bit the results of the unreserved chip --
    and the QUEUED kernels' output (reserve 16: k_fast_cells_q / k_distribute_q / k_describe_q, the ones the headline times) is
    compared with the oracle directly on six of the 32 images, stereo columns included."""
    from lpslam_amd import hip
    w, h, n = 1280, 720, 32
    ctx = hip.Context(w, h, 2000, 1.2, 8, max_images=n)
    seq = synth.StereoSequence(w, h, 0)
    imgs = [seq.frame(i)[e] for i in range(n // 2) for e in range(2)]
    for i, im in enumerate(imgs):
        ctx.upload(i, im)
    p = synth.ba_problem(50, 5000, 40000, w, h, seq_id=0)

    def run():
        ctx.extract_range(0, n)
        for i in range(0, n, 2):
            ctx.match_stereo(i, i + 1, synth.intrinsics(w, h)["fxb"], synth.intrinsics(w, h)["baseline"])
        kd = [ctx.keypoints(i) for i in range(n)]
        st = [ctx.stereo(i) for i in range(0, n, 2)]
        b = hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], hip.ba_obs_array(p), p["cam"])
        log = b.optimize(True, 10)
        poses, points = b.state()
        b.close()
        return kd, st, [tuple(l) for l in log], poses, points
    base = run()
    assert all(len(k) > 1500 for k, _ in base[0])
    k = synth.intrinsics(w, h)
    op = oracle.params(2000, 1.2, 8)
    for r in (16, 12, 0):
        ctx.set_mapping_reserve(r)
        got = run()
        if r == 16:
            for i in (0, 1, 6, 7, 30, 31):
                okp, od, _, _ = oracle.extract(imgs[i], op)
                _assert_same_keypoints(okp, od, got[0][i][0], got[0][i][1])
            for i in (0, 6, 30):
                okl, odl, _, pl = oracle.extract(imgs[i], op, True)
                okr, odr, _, pr = oracle.extract(imgs[i + 1], op, True)
                oxr, odep, obi, _ = oracle.match_stereo(pl, pr, op, okl, odl, okr, odr, k["fxb"], k["baseline"])
                gxr, gdep, gbi = got[1][i // 2]
                assert np.array_equal(obi, gbi) and np.array_equal(oxr, gxr) and np.array_equal(odep, gdep), i
        for (k0, d0), (k1, d1) in zip(base[0], got[0]):
            assert np.array_equal(k0, k1) and np.array_equal(d0, d1), r
        for s0, s1 in zip(base[1], got[1]):
            assert all(np.array_equal(a, b) for a, b in zip(s0, s1)), r
        assert got[2] == base[2] and np.array_equal(got[3], base[3]) and np.array_equal(got[4], base[4]), r
    ctx.close()


@pytest.mark.parametrize("reserve", [8, 16])
def test_mapping_reserve_completes_wherever_the_grid_lands(hiplib, oracle, reserve):
    """The queued extraction must finish its work whatever compute units the dispatcher gives it.  The worst placement is forced:
    a test hook holds the whole LDS of every UNRESERVED compute unit (as a running bundle adjustment, a prefetch stream or another
    session can), so every extraction workgroup lands on a reserved one.  Workgroups there leave only up to a cap (leave tickets,
    frontend.hip); the rest stay and drain the queue.  The slot held other images' results before: stale output cannot pass.
    Checked against the oracle, pyramid and FAST candidates included."""
    w, h, n = 640, 480, 4
    ctx = hiplib.Context(w, h, 1000, 1.2, 8, max_images=n)
    seq = synth.StereoSequence(w, h, 5, n_points=5000)
    old = [seq.frame(10 + i)[e] for i in range(2) for e in range(2)]
    new = [seq.frame(i)[e] for i in range(2) for e in range(2)]
    ctx.set_mapping_reserve(reserve)
    for i, im in enumerate(old):
        ctx.upload(i, im)
    ctx.extract_range(0, n)
    stale = [ctx.keypoints(i) for i in range(n)]
    for i, im in enumerate(new):
        ctx.upload(i, im)
    landed = ctx.debug_occupy_unreserved(20000)                 # 20 ms: several times one extraction on a quarter of the chip
    assert landed >= (256 - 8 * reserve) * 9 // 10, landed      # (practically every unreserved unit is held)
    ctx.extract_range(0, n)
    p = oracle.params(1000, 1.2, 8)
    for i in range(n):
        okp, od, occ, opyr = oracle.extract(new[i], p, True)
        for l in range(8):
            assert np.array_equal(ctx.pyramid_level(i, l), opyr[l]), (i, l)
            assert np.array_equal(oracle.fast_level(opyr[l]), ctx.candidates(i, l)), (i, l)
        gkp, gd = ctx.keypoints(i)
        _assert_same_keypoints(okp, od, gkp, gd)
        assert len(gkp) != len(stale[i][0]) or not np.array_equal(gkp["x"], stale[i][0]["x"])
    ctx.sync()
    ctx.close()


def test_contexts_of_a_flat_priority_process(hiplib, oracle):
    """lpslam_hip_set_flat_priorities(1): five contexts created one after the other (context k puts k mod 4 placeholder streams in front of
    its main stream; prefetch and solves at the default priority) each extract, prefetch and match like a context of the default kind:
    same keypoints and descriptors as the oracle, same frames from the prefetch stream.  The switch is put back afterwards."""
    import threading
    w, h, kpts, levels = 320, 240, 400, 4
    k = synth.intrinsics(w, h)
    seq = synth.StereoSequence(w, h, 6)
    f0, f1 = seq.frame(0), seq.frame(1)
    kp_o, d_o = oracle.extract(f0[0], oracle.params(kpts, 1.2, levels))[:2]
    # (this process has made priority streams by now: the plain switch is refused and says why; the test asks for the late form)
    probe = hiplib.Context(w, h, kpts, 1.2, levels, max_images=2)
    with probe.prefetch():
        pass                                             # a prefetch stream of the low class exists from here on
    with pytest.raises(RuntimeError, match="after a priority stream"):
        hiplib.set_flat_priorities(True)
    probe.close()
    hiplib.set_flat_priorities(True, late_ok=True)
    ctxs = []
    try:
        for i in range(5):
            c = hiplib.Context(w, h, kpts, 1.2, levels, max_images=4)
            ctxs.append(c)
            c.upload(0, f0[0]); c.upload(1, f0[1])
            c.extract_range(0, 2); c.match_stereo(0, 1, k["fxb"], k["baseline"])
        for c in ctxs:
            def helper(cc=c):
                with cc.prefetch():
                    cc.upload(2, f1[0]); cc.upload(3, f1[1])
                    cc.extract_range(2, 2); cc.match_stereo(2, 3, k["fxb"], k["baseline"])
            th = threading.Thread(target=helper); th.start()
            c.match_bf(0, 1)
            th.join(); c.prefetch_join()
        frames = [(c.frame(0), c.frame(2)) for c in ctxs]
        kp0, d0 = ctxs[0].keypoints(0)
        _assert_same_keypoints(kp_o, d_o, kp0, d0)
        for fr in frames[1:]:
            for a, b in zip(fr[0] + fr[1], frames[0][0] + frames[0][1]):
                assert np.array_equal(a, b)
    finally:
        for c in ctxs:
            c.close()
        hiplib.set_flat_priorities(None)
