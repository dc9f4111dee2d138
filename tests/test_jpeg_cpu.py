"""The host's JPEG decoder (lpslam_amd/host/jpeg.cpp) against libjpeg's samples: the reference hands its compressed frames to
cv::imdecode (src/Manager/SlamManager.cpp:1139-1146 for LpSlamImageFormat_8UC1_JPEPG frames, src/Manager/ReplayEngine.cpp:123 for
recorded ones).  Fixtures: tests/golden/g17_jpeg.npz (tools/make_golden_jpeg.py: streams written and decoded by Pillow =
libjpeg-turbo).  Bit for bit."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden


def _decode(lib, data):
    data = np.ascontiguousarray(data, np.uint8)
    w, h = C.c_int(0), C.c_int(0)
    out = np.zeros(1 << 20, np.uint8)
    rc = lib.lpslam_jpeg_decode_gray(data.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), out.ctypes.data_as(C.c_void_p), C.c_size_t(len(out)), C.byref(w), C.byref(h))
    return rc, (out[:w.value * h.value].reshape(h.value, w.value).copy() if rc == 0 else None)


@pytest.fixture(scope="module")
def lib():
    from lpslam_amd import _build
    l = C.CDLL(_build.host_library())
    l.lpslam_jpeg_decode_gray.restype = C.c_int
    return l


def test_decoder_gives_libjpegs_samples(lib):
    g = golden("g17_jpeg.npz")
    names = [k[5:] for k in g.files if k.startswith("jpeg_")]
    assert len(names) >= 10
    for name in names:
        rc, img = _decode(lib, g["jpeg_" + name])
        assert rc == 0, name
        want = g["grey_" + name]
        assert img.shape == want.shape, (name, img.shape, want.shape)
        assert np.array_equal(img, want), (name, int(np.abs(img.astype(int) - want).max()), int((img != want).sum()))


def test_unsupported_and_damaged_streams_are_refused(lib):
    g = golden("g17_jpeg.npz")
    assert _decode(lib, g["progressive_refused"])[0] == 2                       # progressive: a stated limit, not a crash
    good = g["jpeg_grey_ramp_123x77_q70"]
    assert _decode(lib, good[:len(good) // 3])[0] in (0, 2)                     # truncated: an image of what was there, or an error
    assert _decode(lib, np.zeros(100, np.uint8))[0] == 2
    assert _decode(lib, good[:2])[0] == 2
    rng = np.random.default_rng(1)
    for _ in range(200):                                                        # random damage never crashes the decoder
        bad = good.copy()
        for i in rng.integers(2, len(bad), 4):
            bad[i] = rng.integers(0, 256)
        assert _decode(lib, bad)[0] in (0, 1, 2)                                # (1: the damaged header asks for more than the test's buffer)


def _segments(data):
    """(marker, offset of the FF, offset behind the segment) of the header segments up to and including SOS"""
    out, pos = [], 2
    raw = bytes(data)
    while pos + 4 <= len(raw):
        assert raw[pos] == 0xFF
        m, ln = raw[pos + 1], (raw[pos + 2] << 8) | raw[pos + 3]
        out.append((m, pos, pos + 2 + ln))
        if m == 0xDA:
            break
        pos += 2 + ln
    return out


def test_crafted_streams_stay_inside_their_buffers(lib):
    """Streams an attacker of the ingest path (addImageFromBuffer, replay records) could hand in.  (i) luma 1x1 beside chroma 2x2: the
    plane of component 0 is a quarter of the frame -- refused (it used to be copied as if it were full size: a heap over-read,
    ADVICE round 3); (ii) a buffer that ends inside the SOS header; (iii) a second scan of component 0; (iv) a scan that names a
    component twice.  All refused with a stated error; run under tools/asan_cpu.sh with the sanitizers."""
    g = golden("g17_jpeg.npz")
    col = g["jpeg_colour_420_97x61_q75"].copy()
    sof = [x for x in _segments(col) if x[0] == 0xC0][0]
    comps = sof[1] + 10                                   # FF C0 len(2) P Y(2) X(2) Nf, then (id, hv, tq) x 3
    assert col[comps + 1] == 0x22 and col[comps + 4] == 0x11 and col[comps + 7] == 0x11
    bad = col.copy(); bad[comps + 1] = 0x11; bad[comps + 4] = 0x22; bad[comps + 7] = 0x22
    assert _decode(lib, bad)[0] == 2
    bad = col.copy(); bad[comps + 1] = 0x21                # luma 2x1 beside ... vmax 1? (chroma 1x1): still h = hmax, v = vmax -> decodes or errors, no crash
    assert _decode(lib, bad)[0] in (0, 2)
    grey = g["jpeg_grey_ramp_123x77_q70"]
    sos = [x for x in _segments(grey) if x[0] == 0xDA][0]
    cut = np.concatenate([grey[:sos[1] + 2], np.array([0, 2], np.uint8)])          # FF DA 00 02 and nothing behind it
    assert _decode(lib, cut)[0] == 2
    assert _decode(lib, grey[:sos[1] + 3])[0] == 2
    twice = np.concatenate([grey[:-2], grey[sos[1]:]])                              # ... scan, scan again, EOI
    assert _decode(lib, twice)[0] == 2
    csos = [x for x in _segments(col) if x[0] == 0xDA][0]
    dup = col.copy(); dup[csos[1] + 7] = dup[csos[1] + 5]                            # second scan component = the first one's id
    assert _decode(lib, dup)[0] == 2


def test_pillow_agrees_when_present(lib):
    """the fixtures again, decoded now by the Pillow of this machine (skipped where there is none)"""
    PIL = pytest.importorskip("PIL.Image")
    import io
    g = golden("g17_jpeg.npz")
    for name in ("grey_320x240_q95", "colour_420_97x61_q75"):
        data = g["jpeg_" + name].tobytes()
        im = PIL.open(io.BytesIO(data)); im.draft("L", im.size); im.load()
        assert np.array_equal(np.asarray(im), _decode(lib, g["jpeg_" + name])[1])


def _encode(lib, img, quality):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(img.size * 2 + 4096, np.uint8)
    lib.lpslam_jpeg_encode_gray.restype = C.c_size_t
    n = lib.lpslam_jpeg_encode_gray(img.ctypes.data_as(C.c_void_p), C.c_int(img.shape[1]), C.c_int(img.shape[0]), C.c_int(quality), out.ctypes.data_as(C.c_void_p), C.c_size_t(len(out)))
    return out[:n].copy()


def test_encoder_writes_what_libjpeg_writes(lib):
    """LpSlamManager::compressImage (src/InterfaceImpl/LpSlamManager.cpp:133-152) = cv::imencode(".jpg", grey): libjpeg, baseline,
    quality 95.  The encoder here makes the same choices (Annex K tables, quality scaling, islow forward DCT, rounding of the
    quantiser, edge expansion): its streams are byte for byte Pillow's (= libjpeg-turbo's) where Pillow is present, and they always
    decode -- by this library's decoder -- to the image within the quantisation error."""
    from lpslam_amd import synth
    frame = synth.StereoSequence(640, 480, 4, n_points=6000).frame(0)[0]
    rng = np.random.default_rng(3)
    cases = [(frame[:240, :320], 95), (frame[3:100, 5:206], 70), (rng.integers(0, 256, (33, 47), dtype=np.uint8), 95), (np.full((16, 16), 200, np.uint8), 50)]
    for img, quality in cases:
        data = _encode(lib, img, quality)
        assert len(data) > 100 and data[0] == 0xFF and data[1] == 0xD8 and data[-2] == 0xFF and data[-1] == 0xD9
        rc, back = _decode(lib, data)
        assert rc == 0 and back.shape == img.shape
        err = np.abs(back.astype(int) - img.astype(int))
        assert err.mean() < (2.5 if quality >= 90 else 12.0), (quality, err.mean())
    try:
        import io
        from PIL import Image
    except ImportError:
        return
    for img, quality in cases:
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", quality=quality)
        theirs = np.frombuffer(buf.getvalue(), np.uint8)
        ours = _encode(lib, img, quality)
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(ours.tobytes()))), np.asarray(Image.open(io.BytesIO(theirs.tobytes()))))      # the same coefficients
        assert len(ours) == len(theirs) and np.array_equal(ours, theirs), (len(ours), len(theirs))                                        # ... and the same bytes
