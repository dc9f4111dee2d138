"""ctypes face of the host library (lpslam_amd/liblpslam.so): the plain-C shim over the C++ LpSlamManager mirror
(lpslam_amd/host/interface.cpp).  Used by the tests; C++ clients include include/lpslam_manager.h instead."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblpslam.so")


class ROSTimestamp(C.Structure):
    _fields_ = [("seconds", C.c_int32), ("nanoseconds", C.c_int64)]


class Position(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("x", "y", "z", "x_sigma", "y_sigma", "z_sigma")]


class OrientationS(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("w", "x", "y", "z", "sigma")]


class GlobalState(C.Structure):
    _fields_ = [("position", Position), ("orientation", OrientationS), ("valid", C.c_bool)]


class GlobalStateInTime(C.Structure):
    _fields_ = [("timestamp", C.c_int64), ("ros_timestamp", ROSTimestamp), ("has_ros_timestamp", C.c_uint8), ("state", GlobalState)]


class ImageDescription(C.Structure):
    _fields_ = [("structure", C.c_int), ("format", C.c_int), ("image_conversion", C.c_int), ("height", C.c_uint32),
                ("width", C.c_uint32), ("imageSize", C.c_uint32), ("imageSizeSecond", C.c_uint32),
                ("hasRosTimestamp", C.c_uint8), ("rosTimestamp", ROSTimestamp)]


class CameraConfiguration(C.Structure):
    _fields_ = [("camera_number", C.c_uint32), ("distortion_function", C.c_int), ("f_x", C.c_double), ("f_y", C.c_double),
                ("c_x", C.c_double), ("c_y", C.c_double), ("dist", C.c_double * 8), ("mask_type", C.c_int),
                ("mask_parameter", C.c_double), ("resolution_x", C.c_int), ("resolution_y", C.c_int), ("fps", C.c_double),
                ("focal_x_baseline", C.c_double), ("rotation", C.c_double * 9), ("translation", C.c_double * 3)]


class Status(C.Structure):
    _fields_ = [("localization", C.c_int), ("feature_points", C.c_long), ("key_frames", C.c_long), ("frame_time", C.c_double), ("fps", C.c_double)]


class FeatureEntry(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


RECON_CB = C.CFUNCTYPE(None, C.POINTER(GlobalStateInTime), C.c_void_p)
IMAGE_CB = C.CFUNCTYPE(None, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint8), ImageDescription, C.c_void_p)
NAV_CB = C.CFUNCTYPE(C.c_int, ROSTimestamp, C.POINTER(GlobalStateInTime), C.POINTER(GlobalStateInTime), C.c_void_p)

STEREO_TWO_BUFFER, FORMAT_8UC1, FORMAT_8UC3, NO_DISTORTION, ODOM_ONLY = 3, 1, 2, 3, 1
PINHOLE, FISHEYE, OMNI = 0, 1, 2          # LpSlamCameraDistortionFunction
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run __graft_entry__.build()" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.lpslam_manager_create.restype = C.c_void_p
        for name in ("destroy", "start", "stop"):
            getattr(_lib, "lpslam_manager_" + name).argtypes = [C.c_void_p]
        _lib.lpslam_manager_set_log_level.argtypes = [C.c_void_p, C.c_int]
        for name in ("add_tracker", "add_processor", "add_source"):
            getattr(_lib, "lpslam_manager_" + name).argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        _lib.lpslam_manager_read_configuration_file.argtypes = [C.c_void_p, C.c_char_p]
        _lib.lpslam_manager_log_to_file.argtypes = [C.c_void_p, C.c_char_p]
        _lib.lpslam_manager_read_replay_items.argtypes = [C.c_void_p, C.c_char_p]
        _lib.lpslam_manager_set_camera_configuration.argtypes = [C.c_void_p, C.POINTER(CameraConfiguration)]
        _lib.lpslam_manager_default_camera_configuration.argtypes = [C.POINTER(CameraConfiguration)]
        _lib.lpslam_manager_on_reconstruction.argtypes = [C.c_void_p, RECON_CB, C.c_void_p]
        _lib.lpslam_manager_on_image.argtypes = [C.c_void_p, IMAGE_CB, C.c_void_p]
        _lib.lpslam_manager_request_nav_data.argtypes = [C.c_void_p, NAV_CB, C.c_void_p]
        _lib.lpslam_manager_add_stereo_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(ImageDescription)]
        _lib.lpslam_manager_add_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.POINTER(ImageDescription)]
        _lib.lpslam_manager_status.argtypes = [C.c_void_p, C.POINTER(Status)]
        _lib.lpslam_manager_features.argtypes = [C.c_void_p, C.POINTER(FeatureEntry), C.c_size_t, C.POINTER(C.c_float)]
        _lib.lpslam_manager_features.restype = C.c_size_t
        _lib.lpslam_manager_features_count.argtypes = [C.c_void_p]
        _lib.lpslam_manager_features_count.restype = C.c_size_t
        _lib.lpslam_roundtrip_state.argtypes = [C.POINTER(GlobalStateInTime), C.POINTER(GlobalStateInTime)]
    return _lib


def default_camera():
    c = CameraConfiguration()
    load().lpslam_manager_default_camera_configuration(C.byref(c))
    return c


class Manager:
    def __init__(self, log_level=2):
        self.lib = load()
        self.h = self.lib.lpslam_manager_create()
        self.lib.lpslam_manager_set_log_level(self.h, log_level)
        self.results = []
        self._keep = []

    def close(self):
        if self.h:
            self.lib.lpslam_manager_destroy(self.h)
            self.h = None

    __del__ = close

    def log_to_file(self, path, level=1):
        """LpSlamManager::logToFile + setLogLevel (1 = Info): the tracker's statistics line is read back from it by the tests"""
        self.lib.lpslam_manager_set_log_level(self.h, level)
        self.lib.lpslam_manager_log_to_file(self.h, str(path).encode())

    @staticmethod
    def statistics(path):
        """the last "VSLAM statistics: key=value ..." line of a log file as a dict (counters as ints, ms_* timings as floats)"""
        out = {}
        for line in open(path, errors="replace"):
            if "VSLAM statistics:" in line:
                out = {k: (float(v) if k.startswith("ms_") else int(v)) for k, v in (kv.split("=") for kv in line.split("VSLAM statistics:")[1].split())}
        return out

    def tracker_statistics(self):
        """this manager's own "VSLAM statistics" line of its last stop() as a dict (the log file is process-wide)"""
        f = self.lib.lpslam_manager_tracker_statistics
        f.restype = C.c_size_t; f.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        buf = C.create_string_buffer(4096)
        f(self.h, buf, 4096)
        line = buf.value.decode(errors="replace")
        if "VSLAM statistics:" not in line:
            return {}
        return {k: (float(v) if k.startswith("ms_") else int(v)) for k, v in (kv.split("=") for kv in line.split("VSLAM statistics:")[1].split())}

    def read_configuration_file(self, path):
        return bool(self.lib.lpslam_manager_read_configuration_file(self.h, path.encode()))

    def add_tracker(self, name, cfg=""):
        return bool(self.lib.lpslam_manager_add_tracker(self.h, name.encode(), cfg.encode()))

    def add_processor(self, name, cfg=""):
        return bool(self.lib.lpslam_manager_add_processor(self.h, name.encode(), cfg.encode()))

    def read_replay_items(self, path):
        return bool(self.lib.lpslam_manager_read_replay_items(self.h, str(path).encode()))

    def add_source(self, name, cfg=""):
        return bool(self.lib.lpslam_manager_add_source(self.h, name.encode(), cfg.encode()))

    def set_camera(self, cam):
        self.lib.lpslam_manager_set_camera_configuration(self.h, C.byref(cam))

    def collect_results(self, on_result=None):
        def cb(state, _):
            s = state.contents
            self.results.append(dict(timestamp=s.timestamp, valid=bool(s.state.valid),
                                     p=(s.state.position.x, s.state.position.y, s.state.position.z),
                                     q=(s.state.orientation.w, s.state.orientation.x, s.state.orientation.y, s.state.orientation.z)))
            if on_result is not None:
                on_result()
        f = RECON_CB(cb); self._keep.append(f)
        self.lib.lpslam_manager_on_reconstruction(self.h, f, None)

    def count_results(self):
        """installs the library's compiled counting callback (no interpreter on the notify thread): result_counts() = (results, valid)"""
        self.lib.lpslam_manager_count_results.argtypes = [C.c_void_p]
        self.lib.lpslam_manager_count_results(self.h)

    def result_counts(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.lib.lpslam_manager_result_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        self.lib.lpslam_manager_result_counts(self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def collect_images(self):
        """LpSlamManager::addOnImageCallback: every frame the worker takes comes back as JPEG (quality 70) on the image thread;
        self.images gets (timestamp, camera, structure, format, left stream, right stream or None)"""
        self.images = []

        def cb(ts, cam, buf, desc, _):
            n0, n1 = int(desc.imageSize), int(desc.imageSizeSecond)
            raw = C.string_at(buf, n0 + n1)
            self.images.append((int(ts), int(cam), int(desc.structure), int(desc.format), raw[:n0], raw[n0:] if n1 else None))
        f = IMAGE_CB(cb); self._keep.append(f)
        self.lib.lpslam_manager_on_image(self.h, f, None)

    def provide_odometry(self, native=False):
        """answers every navigation request with a valid identity odometry; native=True installs the library's compiled callback
        (lpslam_manager_request_identity_nav_data) instead of a Python one -- no interpreter on the worker thread's path"""
        if native:
            self.lib.lpslam_manager_request_identity_nav_data.argtypes = [C.c_void_p]
            self.lib.lpslam_manager_request_identity_nav_data(self.h)
            return

        def cb(ts, odom, mp, _):
            odom.contents.state.valid = True
            odom.contents.state.orientation.w = 1.0
            return ODOM_ONLY
        f = NAV_CB(cb); self._keep.append(f)
        self.lib.lpslam_manager_request_nav_data(self.h, f, None)

    def add_stereo(self, ts_ns, left, right, ros=True):
        d = ImageDescription(STEREO_TWO_BUFFER, FORMAT_8UC1, 0, left.shape[0], left.shape[1], left.size, right.size,
                             1 if ros else 0, ROSTimestamp(int(ts_ns // 10**9), int(ts_ns)))
        return bool(self.lib.lpslam_manager_add_stereo_image(self.h, 0, int(ts_ns), left.ctypes.data, right.ctypes.data, C.byref(d)))

    def add_image(self, ts_ns, img, camera=0, ros=True):
        d = ImageDescription(0, FORMAT_8UC1, 0, img.shape[0], img.shape[1], img.size, 0, 1 if ros else 0, ROSTimestamp(int(ts_ns // 10**9), int(ts_ns)))
        return bool(self.lib.lpslam_manager_add_image(self.h, camera, int(ts_ns), img.ctypes.data, C.byref(d)))

    def add_jpeg(self, ts_ns, data, camera=0, ros=True):
        """a compressed frame (LpSlamImageFormat_8UC1_JPEPG, one image): decoded by the manager as the reference does with cv::imdecode"""
        raw = bytes(data)
        buf = C.create_string_buffer(raw, len(raw))
        d = ImageDescription(0, 0, 0, 0, 0, len(raw), 0, 1 if ros else 0, ROSTimestamp(int(ts_ns // 10**9), int(ts_ns)))
        return bool(self.lib.lpslam_manager_add_image(self.h, camera, int(ts_ns), C.cast(buf, C.c_void_p), C.byref(d)))

    @staticmethod
    def compress_image(bgra):
        """LpSlamManager::compressImage: an (h, w, 4) BGRA image -> the JPEG stream of its grey version (bytes), or None"""
        import numpy as np
        bgra = np.ascontiguousarray(bgra, np.uint8)
        h, w = bgra.shape[:2]
        d = ImageDescription(0, 3, 0, h, w, bgra.size, 0, 0, ROSTimestamp(0, 0))          # LpSlamImageFormat_8UC4
        out = np.zeros(bgra.size, np.uint8)
        n = C.c_uint32(0)
        f = load().lpslam_manager_compress_image
        f.restype = C.c_int; f.argtypes = [C.c_void_p, C.POINTER(ImageDescription), C.c_void_p, C.POINTER(C.c_uint32)]
        if not f(bgra.ctypes.data, C.byref(d), out.ctypes.data, C.byref(n)):
            return None
        return out[:n.value].tobytes()

    def start(self):
        self.lib.lpslam_manager_start(self.h)

    def stop(self):
        self.lib.lpslam_manager_stop(self.h)

    def status(self):
        s = Status()
        self.lib.lpslam_manager_status(self.h, C.byref(s))
        return s

    def features(self, cap=100000):
        buf = (FeatureEntry * cap)()
        t = (C.c_float * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1)
        n = self.lib.lpslam_manager_features(self.h, buf, cap, t)
        return [(buf[i].x, buf[i].y, buf[i].z) for i in range(n)]
