"""CPU tests of the host mirror (lpslam_amd/host, through its C shim).  The first block restates the reference's own
boundary tests: slam_manager.read_config_file / read_camera_calibration (src/test/SlamManagerTest.cpp:13-197),
interface.type_conversion / add_marker-style smoke (src/test/InterfaceTest.cpp:14-44)."""
import json
import os
import struct
import time

import numpy as np
import pytest


@pytest.fixture(scope="module")
def mgrlib(hiplib):
    from lpslam_amd import _build, manager
    _build.host_library()
    manager.load()
    return manager


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def test_read_config_file(mgrlib, tmp_path):
    m = mgrlib.Manager()
    assert m.read_configuration_file(str(tmp_path / "does_not_exist.json")) is False
    assert m.read_configuration_file(_write(tmp_path, "a.json", '{"datasources": [{"type": "FileSource"}]}')) is True
    assert m.read_configuration_file(_write(tmp_path, "b.json", '{"datasources": [{"type": "FileSource" ')) is False
    assert m.read_configuration_file(_write(tmp_path, "c.json", json.dumps(
        {"datasources": [{"type": "FileSource", "configuration": {"file_names": ["a.jpg", "b.jpg"]}}]}))) is True


def _cam(**over):
    c = {"number": 0, "model": "fisheye", "fx": 286.18, "fy": 286.39, "cx": 416.94, "cy": 403.27,
         "resolution_x": 848, "resolution_y": 800, "distortion": [-0.0115, 0.0479, -0.0451, 0.0083]}
    c.update(over)
    return c


def test_read_camera_calibration(mgrlib, tmp_path):
    m = mgrlib.Manager()
    assert m.read_configuration_file(_write(tmp_path, "ok.json", json.dumps({"cameras": [_cam(), _cam(number=1, model="perspective")]})))
    assert m.read_configuration_file(_write(tmp_path, "nd.json", json.dumps({"cameras": [_cam(model="no_distortion", focal_x_baseline=84.0)]})))
    too_many = _cam(distortion=[0.1] * 9)
    assert not m.read_configuration_file(_write(tmp_path, "d9.json", json.dumps({"cameras": [too_many]})))
    no_res = _cam(distortion=[0.1] * 7); del no_res["resolution_x"]
    assert not m.read_configuration_file(_write(tmp_path, "nr.json", json.dumps({"cameras": [no_res]})))
    assert not m.read_configuration_file(_write(tmp_path, "um.json", json.dumps({"cameras": [_cam(model="pinhole9000")]})))
    assert not m.read_configuration_file(_write(tmp_path, "nm.json", json.dumps({"cameras": [{"number": 0}]})))
    assert m.read_configuration_file(_write(tmp_path, "rv.json", json.dumps({"cameras": [_cam(rotation_vec=[0.0, 0.1, 0.0], translation=[0.1, 0, 0])]})))
    assert not m.read_configuration_file(_write(tmp_path, "rv4.json", json.dumps({"cameras": [_cam(rotation_vec=[0, 0, 0, 1])]})))


def test_type_conversion_roundtrip(mgrlib):
    import ctypes as C
    s = mgrlib.GlobalStateInTime()
    s.timestamp = 1234567890123; s.has_ros_timestamp = 1; s.ros_timestamp.seconds = 12; s.ros_timestamp.nanoseconds = 12000000345
    s.state.valid = True
    s.state.position.x, s.state.position.y, s.state.position.z = 1.5, -2.25, 3.0
    s.state.position.x_sigma, s.state.position.y_sigma, s.state.position.z_sigma = 0.1, 0.2, 0.3
    s.state.orientation.w, s.state.orientation.x, s.state.orientation.y, s.state.orientation.z, s.state.orientation.sigma = 0.5, 0.5, -0.5, 0.5, 0.01
    out = mgrlib.GlobalStateInTime()
    mgrlib.load().lpslam_roundtrip_state(C.byref(s), C.byref(out))
    assert bytes(s)[:8] == bytes(out)[:8] and out.ros_timestamp.nanoseconds == 12000000345 and out.has_ros_timestamp == 1
    for f in ("x", "y", "z", "x_sigma", "y_sigma", "z_sigma"):
        assert getattr(out.state.position, f) == getattr(s.state.position, f)
    for f in ("w", "x", "y", "z", "sigma"):
        assert getattr(out.state.orientation, f) == getattr(s.state.orientation, f)
    assert out.state.valid
    assert C.sizeof(mgrlib.GlobalStateInTime) == 128 and C.sizeof(mgrlib.CameraConfiguration) == 240 and C.sizeof(mgrlib.ImageDescription) == 48
    assert C.sizeof(mgrlib.Status) == 40


def test_plugin_factories_and_config_keys(mgrlib, tmp_path):
    m = mgrlib.Manager()
    assert m.add_tracker("VSLAMStereo", '{"slamKeypoints": 2000, "cameraSetup": "stereo", "numLevels": 8, "_comment": "ignored"}')
    assert m.add_tracker("VSLAMMono", "")
    assert not m.add_tracker("VSLAMStereo", '{"noSuchKey": 1}')             # unknown key -> rejected (ConfigOptions)
    assert not m.add_tracker("VSLAMStereo", '{"slamKeypoints": "many"}')    # wrong type
    assert not m.add_tracker("Marker9000", "")                               # unknown plugin -> false
    assert not m.add_processor("BlackoutImage", "") and not m.add_source("Zed", "")
    cfg = {"manager": {"thread_num": 2, "record": False},
           "trackers": [{"type": "VSLAMStereo", "configuration": {"slamKeypoints": 1000}}, {"_type": "VSLAMMono"}],
           "cameras": [_cam(model="no_distortion", focal_x_baseline=84.0)]}
    assert m.read_configuration_file(_write(tmp_path, "full.json", json.dumps(cfg)))
    assert not m.read_configuration_file(_write(tmp_path, "bad.json", json.dumps({"trackers": [{"configuration": {}}]})))
    assert not m.read_configuration_file(_write(tmp_path, "bad2.json", json.dumps({"trackers": [{"type": "Nope"}]})))


def test_frames_without_odometry_are_skipped_and_reported(mgrlib):
    """No nav callback -> every frame is skipped but the client still gets one (invalid) result per frame
    (src/Manager/SlamManager.cpp:193-196,230-236).  Runs without a GPU: the tracker is never reached."""
    m = mgrlib.Manager()
    assert m.add_tracker("VSLAMStereo", '{"cameraSetup": "stereo"}')
    m.collect_results()
    m.start()                      # tracker start fails without camera configuration / GPU: logged, not fatal
    img = np.zeros((120, 160), np.uint8)
    for k in range(3):
        assert m.add_stereo(1000 * (k + 1), img, img)
    t0 = time.time()
    while len(m.results) < 3 and time.time() - t0 < 5:
        time.sleep(0.01)
    m.stop()
    assert len(m.results) == 3 and not any(r["valid"] for r in m.results)
    assert m.status().localization == 0          # Off


def test_image_callback_receives_every_frame_as_jpeg(mgrlib):
    """OnImageCallback_t (src/Interface/LpSlamTypes.h:233-235; image-callback thread src/Manager/SlamManager.cpp:258-314): every frame
    the worker takes is sent to the client as JPEG of quality 70 -- one buffer, left stream then right stream, desc.imageSize /
    imageSizeSecond the split, structure Stereo_Compressed (one image: OneImage_Compressed), format 8UC1_JPEPG -- on its own thread.
    The reference arms that queue only when the reconstruction callback is set as well (WorkerThreadParams, SlamManager.cpp:532-548).
    The streams are what libjpeg writes for the frame (Pillow where present), and decode back to it.  Runs without a GPU."""
    from lpslam_amd import synth
    frame = synth.StereoSequence(320, 240, 4, n_points=2000).frame(0)
    left, right = np.ascontiguousarray(frame[0]), np.ascontiguousarray(frame[1])
    m = mgrlib.Manager()
    m.collect_results(); m.collect_images()
    m.start()
    assert m.add_stereo(1000, left, right)
    assert m.add_image(2000, left, camera=3)
    t0 = time.time()
    while len(m.images) < 2 and time.time() - t0 < 10:
        time.sleep(0.01)
    m.stop()
    assert len(m.images) == 2
    (ts0, cam0, st0, fmt0, l0, r0), (ts1, cam1, st1, fmt1, l1, r1) = m.images
    assert (ts0, cam0, st0, fmt0) == (1000, 0, 5, 0) and (ts1, cam1, st1, fmt1, r1) == (2000, 3, 4, 0, None)
    assert l0 == l1 and l0[:2] == b"\xff\xd8" and l0[-2:] == b"\xff\xd9" and r0[:2] == b"\xff\xd8" and r0 != l0
    try:
        import io
        from PIL import Image
        for img, ours in ((left, l0), (right, r0)):
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, "JPEG", quality=70)
            assert buf.getvalue() == ours
            back = np.asarray(Image.open(io.BytesIO(ours)))
            assert back.shape == img.shape and np.abs(back.astype(int) - img.astype(int)).mean() < 12
    except ImportError:
        pass
    # without the reconstruction callback nothing is queued (the reference's condition); frames are still consumed
    m2 = mgrlib.Manager()
    m2.collect_images()
    m2.start()
    assert m2.add_stereo(1000, left, right)
    time.sleep(0.3)
    m2.stop()
    assert m2.images == []


def test_compressed_frames_are_decoded_at_ingest(mgrlib):
    """LpSlamImageFormat_8UC1_JPEPG frames (src/Manager/SlamManager.cpp:1139-1146: cv::imdecode, IMREAD_GRAYSCALE): the manager decodes
    baseline JPEG itself (host/jpeg.cpp) and queues the grey image; what is not a decodable stream is refused like any unsupported
    format.  Runs without a GPU: the frames are skipped for lack of odometry, one invalid result each."""
    from conftest import golden
    g = golden("g17_jpeg.npz")
    m = mgrlib.Manager()
    assert m.add_tracker("VSLAMMono", '{"cameraSetup": "monocular"}')
    m.collect_results()
    m.start()
    assert m.add_jpeg(1000, g["jpeg_grey_201x99_q90"].tobytes())
    assert m.add_jpeg(2000, g["jpeg_colour_420_97x61_q75"].tobytes())             # a colour stream: its luma plane
    assert not m.add_jpeg(3000, g["progressive_refused"].tobytes())                # progressive: refused, logged
    assert not m.add_jpeg(4000, b"\xff\xd8\xff\xd9")                              # no frame inside
    assert not m.add_jpeg(5000, b"not a jpeg at all")
    t0 = time.time()
    while len(m.results) < 2 and time.time() - t0 < 5:
        time.sleep(0.01)
    m.stop()
    assert len(m.results) == 2


def test_compress_image_round_trip(mgrlib):
    """LpSlamManager::compressImage (src/InterfaceImpl/LpSlamManager.cpp:133-152): BGRA -> grey (cv::cvtColor's weights) -> JPEG as
    cv::imencode writes it, and back in through addImageFromBuffer as a LpSlamImageFormat_8UC1_JPEPG frame -- what LpGlobalFusion does
    with its Webots frames."""
    rng = np.random.default_rng(2)
    grey = np.clip(np.add.outer(np.arange(96) * 2, np.arange(128)) + rng.integers(0, 20, (96, 128)), 0, 255).astype(np.uint8)
    bgra = np.stack([np.roll(grey, 3, 1), grey, np.roll(grey, -3, 0), np.full_like(grey, 255)], axis=-1)
    data = mgrlib.Manager.compress_image(bgra)
    assert data is not None and data[:2] == b"\xff\xd8" and data[-2:] == b"\xff\xd9" and len(data) < bgra.size
    want = ((bgra[..., 0].astype(np.int64) * 1868 + bgra[..., 1].astype(np.int64) * 9617 + bgra[..., 2].astype(np.int64) * 4899 + (1 << 13)) >> 14).astype(np.uint8)
    try:                                                     # byte for byte what libjpeg writes for that grey image at quality 95
        import io
        from PIL import Image
        buf = io.BytesIO(); Image.fromarray(want).save(buf, "JPEG", quality=95)
        assert data == buf.getvalue()
    except ImportError:
        pass
    m = mgrlib.Manager()
    assert m.add_tracker("VSLAMMono", '{"cameraSetup": "monocular"}')
    m.collect_results()
    m.start()
    assert m.add_jpeg(1000, data)
    t0 = time.time()
    while len(m.results) < 1 and time.time() - t0 < 5:
        time.sleep(0.01)
    m.stop()
    assert len(m.results) == 1


def test_default_camera_configuration(mgrlib):
    c = mgrlib.default_camera()
    assert c.fps == 25.0 and c.distortion_function == mgrlib.NO_DISTORTION and list(c.rotation) == [1, 0, 0, 0, 1, 0, 0, 0, 1]


def test_replay_stream_reader(tmp_path):
    """src/Serialize/ProtoStream.h framing + SlamSerialize.proto CameraImage records (ReplayEngine.cpp:83-242)."""
    import ctypes
    import replay_format as rf
    from lpslam_amd import _build
    lib = ctypes.CDLL(_build.host_library())
    rng = np.random.default_rng(0)
    l = rng.integers(0, 256, (48, 64)).astype(np.uint8); r = rng.integers(0, 256, (48, 64)).astype(np.uint8)
    from conftest import golden
    jpeg = golden("g17_jpeg.npz")["jpeg_grey_201x99_q90"].tobytes()
    imu = rf.f_varint(1, 5) + rf.f_bytes(2, rf.vec3(0, 0, 9.81)) + rf.f_bytes(3, rf.vec3(0.1, 0, 0))
    stream = b"".join([
        rf.record(rf.SENSOR_IMU, imu),
        rf.record(rf.CAMERA_IMAGE, rf.camera_image(1_000_000_123, l, r, cam=4, odom=((1.5, -2.0, 0.25), (0.5, 0.5, -0.5, 0.5)), data_number=7)),
        rf.record(rf.RESULT, rf.f_varint(1, 9)),
        rf.record(rf.CAMERA_IMAGE, rf.camera_image(2_000_000_000, l, None, cam=2, map_=((3, 4, 5), (1, 0, 0, 0)))),
        rf.record(rf.CAMERA_IMAGE, rf.camera_image(3_000_000_000, l, raw_left=b"\\xff\\xd8\\xff\\xe0JFIF-not-decodable")),     # what the recorder writes
        rf.record(rf.SENSOR_GLOBAL_STATE, rf.f_varint(1, 11)), rf.record(rf.SENSOR_FEATURE, rf.f_varint(1, 12)),
        rf.record(rf.CAMERA_IMAGE, rf.camera_image(4_000_000_000, l, raw_left=jpeg)),                                      # a real JPEG record
        struct.pack("<QQ", 77, 3) + b"abc",                                                                                 # corrupt tail
    ])
    path = tmp_path / "rec.pb"; path.write_bytes(stream)
    stats = (ctypes.c_long * 8)(); first = (ctypes.c_long * 6)(); state = (ctypes.c_double * 14)()
    lib.lpslam_replay_probe.restype = ctypes.c_long
    n = lib.lpslam_replay_probe(str(path).encode(), stats, first, state)
    assert n == 3 and list(stats) == [8, 4, 1, 1, 1, 1, 1, 1]                    # the JPEG record decodes (baseline, host/jpeg.cpp), the fake one does not
    assert list(first) == [1_000_000_123, 4, 5, 64, 48, 1]
    assert list(state)[:7] == [1.5, -2.0, 0.25, 0.5, 0.5, -0.5, 0.5] and all(np.isnan(list(state)[7:]))
    assert lib.lpslam_replay_probe(str(tmp_path / "missing.pb").encode(), stats, first, state) == -1
    empty = tmp_path / "empty.pb"; empty.write_bytes(b"")
    assert lib.lpslam_replay_probe(str(empty).encode(), stats, first, state) == 0 and list(stats)[:2] == [0, 0]
