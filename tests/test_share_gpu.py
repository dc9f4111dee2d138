"""Launches shared by the sessions of a process (lpslam_amd/csrc/share.hip): the front end of several sessions' pending frames as
one launch chain over an image list of the session pool, the window matchers' first scans and the pose optimisations of several
sessions as one launch each -- against the same calls made by a lone context: integer results (keypoints, descriptors, stereo
columns, match lists) are equal, the pose optimiser runs the same device code per request and returns the same bits."""
import threading

import numpy as np
import pytest

from lpslam_amd import synth

pytestmark = pytest.mark.gpu


def _frame_of(ctx, slot):
    kp, desc, xr, depth = ctx.frame_view(slot, True)
    return kp.copy(), desc.copy(), xr.copy(), depth.copy()


@pytest.mark.parametrize("with_upload", [False, True])
@pytest.mark.parametrize("size", [(640, 480, 1000, 4), (1280, 720, 2000, 8)])
def test_shared_front_end_equals_the_unshared_one(hiplib, size, with_upload):
    """Four sessions of one pool submit lpslam_hip_front_end from four threads (sharing forced on, so a lone request is a shared launch of
    one too); every session's frame -- keypoints, descriptors, right-image columns, depths -- is what a private context extracts from the
    same images with extract_range + match_stereo + prefetch_frame; three rounds over the rotating slot pairs."""
    w, h, kpts, levels = size
    k = synth.intrinsics(w, h)
    n_sessions, rounds = 4, 3
    seqs = [synth.StereoSequence(w, h, 4 + s) for s in range(n_sessions)]
    images = [[seqs[s].frame(r) for r in range(rounds)] for s in range(n_sessions)]
    ref_ctx = hiplib.Context(w, h, kpts, 1.2, levels, max_images=2)
    ref = {}
    for s in range(n_sessions):
        for r in range(rounds):
            l, rt = images[s][r]
            ref_ctx.upload(0, l); ref_ctx.upload(1, rt)
            ref_ctx.extract_range(0, 2); ref_ctx.match_stereo(0, 1, k["fxb"], k["baseline"]); ref_ctx.prefetch_frame(0, True)
            ref[(s, r)] = _frame_of(ref_ctx, 0)
    ref_ctx.close()
    got, errors = {}, []
    try:
        hiplib.set_shared_launches(1)
        b0, r0 = hiplib.shared_front_end_counters(0)
        sessions = [hiplib.Context(w, h, kpts, 1.2, levels, max_images=6, session=True) for _ in range(n_sessions)]
        barrier = threading.Barrier(n_sessions)

        def run(s):
            try:
                c = sessions[s]
                for r in range(rounds):
                    slot = 2 * (r % 3)
                    l, rt = images[s][r]
                    barrier.wait()
                    if with_upload:                  # lpslam_hip_front_end_images: the chain uploads the frames itself
                        c.front_end_images(slot, l, rt, k["fxb"], k["baseline"])
                    else:
                        c.upload(slot, l); c.upload(slot + 1, rt)
                        c.front_end(slot, True, k["fxb"], k["baseline"])
                    got[(s, r)] = _frame_of(c, slot)
            except BaseException as e:
                errors.append(e)
                barrier.abort()

        th = [threading.Thread(target=run, args=(s,)) for s in range(n_sessions)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        b1, r1 = hiplib.shared_front_end_counters(0)
        for c in sessions:
            c.close()
    finally:
        hiplib.set_shared_launches(None)
    assert not errors, errors
    assert r1 - r0 == n_sessions * rounds and b1 - b0 <= r1 - r0
    for key, (kp, desc, xr, depth) in ref.items():
        gk, gd, gx, gz = got[key]
        assert len(kp) == len(gk) and len(kp) > 100, (key, len(kp), len(gk))
        for f in kp.dtype.names:
            assert np.array_equal(kp[f], gk[f]), (key, f)
        assert np.array_equal(desc, gd) and np.array_equal(xr, gx) and np.array_equal(depth, gz), key
    print("shared front end %dx%d: %d requests in %d launch chains" % (w, h, r1 - r0, b1 - b0))
