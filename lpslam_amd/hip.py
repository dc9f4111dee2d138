"""ctypes binding of the C ABI in include/lpslam_hip.h (lpslam_amd/liblpslam_hip.so).

This is the thin Python face of the HIP library used by the tests and bench.py; the product's host side is the
C++ mirror of the reference interface in lpslam_amd/host/.  There is no CPU fallback: a missing library or a
missing GPU raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LPSLAM_HIP_LIB") or os.path.join(_HERE, "liblpslam_hip.so")      # the override: A/B timing of two builds on one box
MAX_LEVELS = 16

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CORNER_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("score", "<i4")])
BA_OBS_DTYPE = np.dtype([("pose", "<i4"), ("point", "<i4"), ("u", "<f8"), ("v", "<f8"), ("ur", "<f8"),
                         ("inv_sigma2", "<f8")])
SIM3_EDGE_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("meas", "<f8", (8,))])
SIM3_PAIR_DTYPE = np.dtype([("p1c", "<f8", (3,)), ("p2c", "<f8", (3,)), ("obs1", "<f8", (2,)), ("obs2", "<f8", (2,)),
                            ("inv_sigma2_1", "<f8"), ("inv_sigma2_2", "<f8")])
PROJ_QUERY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("x_right", "<f4"), ("radius", "<f4"), ("min_level", "<i4"), ("max_level", "<i4")])
BA_LOG_DTYPE = np.dtype([("chi2_before", "<f8"), ("chi2_after", "<f8"), ("lambda", "<f8"),
                         ("trials", "<i4"), ("status", "<i4")])

# every symbol include/lpslam_hip.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "lpslam_hip_last_error", "lpslam_hip_device_count", "lpslam_hip_create", "lpslam_hip_destroy",
    "lpslam_hip_stream", "lpslam_hip_set_mapping_reserve", "lpslam_hip_set_flat_priorities", "lpslam_hip_debug_occupy_unreserved", "lpslam_hip_sync", "lpslam_hip_timer_begin", "lpslam_hip_timer_end", "lpslam_hip_timer_read", "lpslam_hip_level_info", "lpslam_hip_max_keypoints_per_image",
    "lpslam_hip_image_ptr", "lpslam_hip_upload_image", "lpslam_hip_host_alloc", "lpslam_hip_host_free", "lpslam_hip_host_register", "lpslam_hip_host_unregister", "lpslam_hip_upload_images_async", "lpslam_hip_set_rectify_map", "lpslam_hip_set_mask", "lpslam_hip_upload_raw_image", "lpslam_hip_remap_staged", "lpslam_hip_extract", "lpslam_hip_extract_range", "lpslam_hip_stage_pyramid",
    "lpslam_hip_stage_fast", "lpslam_hip_stage_distribute", "lpslam_hip_stage_describe",
    "lpslam_hip_keypoint_count", "lpslam_hip_get_keypoints", "lpslam_hip_get_frame", "lpslam_hip_get_pyramid_level",
    "lpslam_hip_get_candidates", "lpslam_hip_keypoint_buffers", "lpslam_hip_match_bf", "lpslam_hip_get_bf_knn2",
    "lpslam_hip_get_bf_matches", "lpslam_hip_match_bf_strided", "lpslam_hip_set_descriptors",
    "lpslam_hip_match_stereo", "lpslam_hip_match_stereo_strided", "lpslam_hip_get_stereo",
    "lpslam_hip_vocab_create", "lpslam_hip_vocab_destroy", "lpslam_hip_vocab_info", "lpslam_hip_bow_transform", "lpslam_hip_bow_transform_host", "lpslam_hip_match_bow_tree", "lpslam_hip_match_bow_tree_multi",
    "lpslam_hip_match_projection", "lpslam_hip_match_fuse", "lpslam_hip_match_area", "lpslam_hip_match_orientation_filter",
    "lpslam_hip_ba_create", "lpslam_hip_ba_prepare", "lpslam_hip_ba_build_batch", "lpslam_hip_ba_destroy", "lpslam_hip_ba_set_active", "lpslam_hip_ba_optimize", "lpslam_hip_ba_optimize_begin", "lpslam_hip_ba_optimize_end", "lpslam_hip_ba_graph_replays", "lpslam_hip_ba_wg_factorisations", "lpslam_hip_pose_optimize_passes", "lpslam_hip_match_bf_descriptors", "lpslam_hip_prefetch_frame", "lpslam_hip_get_frame_view", "lpslam_hip_desc_store_put", "lpslam_hip_desc_store_drop", "lpslam_hip_match_bf_stored", "lpslam_hip_ba_set_state_batch", "lpslam_hip_ba_get_batch", "lpslam_hip_ba_get_solver", "lpslam_hip_ba_set_solver", "lpslam_hip_ba_timeouts", "lpslam_hip_ba_counters", "lpslam_hip_set_shared_launches", "lpslam_hip_shared_launch_counters", "lpslam_hip_create_session", "lpslam_hip_front_end", "lpslam_hip_frame_done", "lpslam_hip_front_end_images", "lpslam_hip_shared_front_end_counters", "lpslam_hip_shared_solve_counters", "lpslam_hip_ba_local_window",
    "lpslam_hip_ba_optimize_batch", "lpslam_hip_ba_reset_batch", "lpslam_hip_ba_optimize_profiled", "lpslam_hip_ba_optimize_partitioned", "lpslam_hip_ba_optimize_partitioned_with",
    "lpslam_hip_ba_local", "lpslam_hip_ba_set_points_fixed", "lpslam_hip_ba_pose_optimize", "lpslam_hip_pose_optimize", "lpslam_hip_ba_reset", "lpslam_hip_ba_set_state", "lpslam_hip_prefetch_begin", "lpslam_hip_prefetch_end", "lpslam_hip_prefetch_join", "lpslam_hip_ba_get", "lpslam_hip_ba_chi2", "lpslam_hip_ba_reduced_buffer",
    "lpslam_hip_ba_step_begin", "lpslam_hip_ba_step_lambda0", "lpslam_hip_ba_step_solve", "lpslam_hip_ba_scalar_buffer", "lpslam_hip_ba_step_end", "lpslam_hip_ba_status",
    "lpslam_hip_sim3_create", "lpslam_hip_sim3_destroy", "lpslam_hip_sim3_optimize", "lpslam_hip_sim3_get", "lpslam_hip_sim3_chi2", "lpslam_hip_sim3_transform_optimize",
]


class LpslamHipError(RuntimeError):
    pass


class FrontendConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_keypoints", C.c_int32),
                ("scale_factor", C.c_float), ("num_levels", C.c_int32), ("ini_fast_threshold", C.c_int32),
                ("min_fast_threshold", C.c_int32), ("max_images", C.c_int32), ("device", C.c_int32)]


class BaCamera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("focal_x_baseline", C.c_double), ("huber_mono", C.c_double), ("huber_stereo", C.c_double)]


class BaKernelTimes(C.Structure):
    _fields_ = [("ms", C.c_float * 8), ("launches", C.c_int32 * 8), ("launches_per_mark", C.c_int32 * 8), ("iterations", C.c_int32), ("dim", C.c_int32)]


_lib = None


def load():
    """Loads the in-tree HIP library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LpslamHipError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                 "(hipcc --offload-arch=gfx950); the lpslam hot path has no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.lpslam_hip_last_error.restype = C.c_char_p
        _lib.lpslam_hip_stream.restype = C.c_void_p
        _lib.lpslam_hip_stream.argtypes = [C.c_void_p]
        for name in ("lpslam_hip_match_stereo",):
            getattr(_lib, name).argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        _lib.lpslam_hip_match_stereo_strided.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
        _lib.lpslam_hip_match_bf_descriptors.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        _lib.lpslam_hip_get_bf_matches.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_float, C.c_int32,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    return _lib


def _check(rc):
    if rc != 0:
        raise LpslamHipError("lpslam_hip error %d: %s" % (rc, load().lpslam_hip_last_error().decode()))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def set_flat_priorities(flat, late_ok=False):
    """process-wide: streams of contexts created from now on at the default priority (True) or in the three classes (False); None: the environment.
    True after a priority stream exists in the process is refused (RuntimeError) unless late_ok"""
    _check(load().lpslam_hip_set_flat_priorities(C.c_int32(-1 if flat is None else ((2 if late_ok else 1) if flat else 0))))


def set_shared_launches(mode):
    """process-wide: launches shared by the sessions of a process -- 0 never, 1 always, 2 when two or more sessions are tracking (default), None: the environment"""
    _check(load().lpslam_hip_set_shared_launches(C.c_int32(-1 if mode is None else int(mode))))


def shared_launch_counters(device=0):
    b, r = C.c_int64(0), C.c_int64(0)
    _check(load().lpslam_hip_shared_launch_counters(C.c_int32(device), C.byref(b), C.byref(r)))
    return b.value, r.value


def shared_front_end_counters(device=0):
    b, r = C.c_int64(0), C.c_int64(0)
    _check(load().lpslam_hip_shared_front_end_counters(C.c_int32(device), C.byref(b), C.byref(r)))
    return b.value, r.value


def shared_solve_counters(device=0):
    b, r = C.c_int64(0), C.c_int64(0)
    _check(load().lpslam_hip_shared_solve_counters(C.c_int32(device), C.byref(b), C.byref(r)))
    return b.value, r.value


def device_count():
    n = C.c_int(0)
    rc = load().lpslam_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class Context:
    """One front-end context = one GPU, one stream, `max_images` resident image slots."""

    def __init__(self, width, height, max_keypoints=2000, scale_factor=1.2, num_levels=8, ini_thr=20, min_thr=7,
                 max_images=2, device=0, session=False):
        self.lib = load()
        self.cfg = FrontendConfig(width, height, max_keypoints, scale_factor, num_levels, ini_thr, min_thr, max_images, device)
        h = C.c_void_p()
        # session: a slice of the device's session pool (lpslam_hip_create_session: what a tracker plugin creates)
        _check((self.lib.lpslam_hip_create_session if session else self.lib.lpslam_hip_create)(C.byref(self.cfg), C.byref(h)))
        self.h = h
        self.max_kp = self.lib.lpslam_hip_max_keypoints_per_image(self.h)
        L = num_levels
        w = (C.c_int32 * MAX_LEVELS)(); hh = (C.c_int32 * MAX_LEVELS)(); p = (C.c_int32 * MAX_LEVELS)()
        q = (C.c_int32 * MAX_LEVELS)(); s = (C.c_float * MAX_LEVELS)()
        _check(self.lib.lpslam_hip_level_info(self.h, w, hh, p, q, s))
        self.level_w, self.level_h, self.level_pitch = list(w[:L]), list(hh[:L]), list(p[:L])
        self.quota, self.scale = list(q[:L]), list(s[:L])
        import weakref
        self._children = weakref.WeakSet()      # objects that hand memory back to this context when they are destroyed

    def close(self):
        if getattr(self, "h", None):
            for child in list(getattr(self, "_children", ())):
                child.close()
            self.lib.lpslam_hip_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def stream(self):
        return self.lib.lpslam_hip_stream(self.h)

    def sync(self):
        _check(self.lib.lpslam_hip_sync(self.h))

    def front_end(self, image, stereo, fxb=0.0, baseline=0.0):
        """extraction of the slot (pair), stereo match and delivery of the frame's results as one asynchronous call"""
        f = self.lib.lpslam_hip_front_end
        f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_float, C.c_float]
        _check(f(self.h, int(image), 1 if stereo else 0, float(fxb), float(baseline)))

    def front_end_images(self, image, left, right=None, fxb=0.0, baseline=0.0):
        """upload of the frame into the slot (pair) + front_end, as one asynchronous call"""
        f = self.lib.lpslam_hip_front_end_images
        f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float]
        left = np.ascontiguousarray(left); right = None if right is None else np.ascontiguousarray(right)
        _check(f(self.h, int(image), _p(left), None if right is None else _p(right), left.shape[1], float(fxb), float(baseline)))

    def set_mapping_reserve(self, cus_per_xcd):
        """compute units of every XCD the front end's kernels leave to the bundle adjustments that run beside them (idle context)"""
        f = self.lib.lpslam_hip_set_mapping_reserve
        f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int32]
        _check(f(self.h, int(cus_per_xcd)))

    def debug_occupy_unreserved(self, microseconds):
        """Test hook: every compute unit outside the mapping reserve is held (whole LDS) for `microseconds`; returns how many are."""
        n = C.c_int32(0)
        _check(self.lib.lpslam_hip_debug_occupy_unreserved(self.h, C.c_int32(int(microseconds)), C.byref(n)))
        return n.value

    def ba_counters(self):
        """totals over every bundle-adjustment problem of this context: signatures in the graph cache, graphs built, replays, timed-out hand-overs (band, update)"""
        out = (C.c_int64 * 5)()
        _check(self.lib.lpslam_hip_ba_counters(self.h, out, 5))
        return dict(zip(("signatures", "graphs", "replays", "timeouts_band", "timeouts_update"), (int(v) for v in out)))

    def ba_graph_replays(self):
        f = self.lib.lpslam_hip_ba_graph_replays
        f.restype = C.c_int64; f.argtypes = [C.c_void_p]
        return int(f(self.h))

    def ba_wg_factorisations(self):
        f = self.lib.lpslam_hip_ba_wg_factorisations
        f.restype = C.c_int64; f.argtypes = [C.c_void_p]
        return int(f(self.h))

    def pose_optimize_passes(self):
        f = self.lib.lpslam_hip_pose_optimize_passes
        f.restype = C.c_int32; f.argtypes = [C.c_void_p]
        return int(f(self.h))

    def timer_begin(self, slot):
        _check(self.lib.lpslam_hip_timer_begin(self.h, slot))

    def timer_end(self, slot):
        _check(self.lib.lpslam_hip_timer_end(self.h, slot))

    def timer_ms(self, slot):
        ms = C.c_float()
        _check(self.lib.lpslam_hip_timer_read(self.h, slot, C.byref(ms)))
        return ms.value

    def image_ptr(self, image):
        ptr = C.c_void_p(); pitch = C.c_int32()
        _check(self.lib.lpslam_hip_image_ptr(self.h, image, C.byref(ptr), C.byref(pitch)))
        return ptr.value, pitch.value

    def upload(self, image, arr):
        arr = np.ascontiguousarray(arr, np.uint8)
        assert arr.shape == (self.cfg.height, self.cfg.width), arr.shape
        _check(self.lib.lpslam_hip_upload_image(self.h, image, _p(arr), arr.shape[1]))
        self.sync()      # host array may be freed by the caller

    def host_frame(self, arr=None):
        """a page-locked (height, width) uint8 frame owned by the context (lpslam_hip_host_alloc), optionally filled from `arr`"""
        hgt, wid = self.cfg.height, self.cfg.width
        ptr = C.c_void_p()
        f = self.lib.lpslam_hip_host_alloc
        f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        _check(f(self.h, hgt * wid, C.byref(ptr)))
        buf = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(hgt, wid))
        if arr is not None:
            buf[...] = arr
        return buf

    def host_free(self, frame):
        f = self.lib.lpslam_hip_host_free
        f.argtypes = [C.c_void_p, C.c_void_p]
        _check(f(self.h, frame.ctypes.data))

    def host_register(self, arr):
        f = self.lib.lpslam_hip_host_register
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _check(f(self.h, arr.ctypes.data, arr.nbytes))

    def host_unregister(self, arr):
        f = self.lib.lpslam_hip_host_unregister
        f.argtypes = [C.c_void_p, C.c_void_p]
        _check(f(self.h, arr.ctypes.data))

    def upload_async(self, first, frames):
        """lpslam_hip_upload_images_async: frames[i] -> slot first + i on the copy stream; returns at once (the frames must stay
        alive and unchanged until an extraction of the slots has been synchronised)"""
        n = len(frames)
        ptrs = (C.c_void_p * n)(*[fr.ctypes.data for fr in frames])
        for fr in frames:
            assert fr.dtype == np.uint8 and fr.shape == (self.cfg.height, self.cfg.width) and fr.strides[1] == 1, (fr.dtype, fr.shape)
        f = self.lib.lpslam_hip_upload_images_async
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int32]
        _check(f(self.h, int(first), n, ptrs, int(frames[0].strides[0])))

    def set_rectify_map(self, eye, map_x, map_y):
        mx = np.ascontiguousarray(map_x, np.float32); my = np.ascontiguousarray(map_y, np.float32)
        assert mx.shape == (self.cfg.height, self.cfg.width) and my.shape == mx.shape
        _check(self.lib.lpslam_hip_set_rectify_map(self.h, int(eye), _p(mx), _p(my)))

    def set_mask(self, eye, mask):
        """camera mask of one eye (uint8, image size, 0 = masked out); None removes it"""
        if mask is None:
            _check(self.lib.lpslam_hip_set_mask(self.h, int(eye), None, 0)); return
        m = np.ascontiguousarray(mask, np.uint8)
        assert m.shape == (self.cfg.height, self.cfg.width), m.shape
        _check(self.lib.lpslam_hip_set_mask(self.h, int(eye), _p(m), m.shape[1]))

    def upload_raw(self, image, eye, arr):
        arr = np.ascontiguousarray(arr, np.uint8)
        assert arr.shape == (self.cfg.height, self.cfg.width), arr.shape
        _check(self.lib.lpslam_hip_upload_raw_image(self.h, image, int(eye), _p(arr), arr.shape[1]))
        self.sync()

    def match_projection(self, image, queries, q_desc, hamming_thr=100, lowe_ratio=0.8, taken=None, use_stereo=False):
        q = np.ascontiguousarray(queries, PROJ_QUERY_DTYPE); d = np.ascontiguousarray(q_desc, np.uint8)
        assert d.shape == (len(q), 32)
        t = np.ascontiguousarray(taken, np.uint8) if taken is not None else None
        idx = np.full(max(len(q), 1), -1, np.int32); dist = np.zeros(max(len(q), 1), np.int32); n = C.c_int32()
        f = self.lib.lpslam_hip_match_projection
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_int32,
                      C.c_void_p, C.c_void_p, C.c_void_p]
        _check(f(self.h, image, _p(q), _p(d), len(q), int(hamming_thr), float(lowe_ratio), _p(t), int(use_stereo), _p(idx), _p(dist), C.byref(n)))
        return idx[:len(q)].copy(), dist[:len(q)].copy(), n.value

    def match_fuse(self, image, queries, q_desc, hamming_thr=50, use_stereo=False):
        q = np.ascontiguousarray(queries, PROJ_QUERY_DTYPE); d = np.ascontiguousarray(q_desc, np.uint8)
        idx = np.full(max(len(q), 1), -1, np.int32); dist = np.zeros(max(len(q), 1), np.int32); n = C.c_int32()
        f = self.lib.lpslam_hip_match_fuse
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(f(self.h, image, _p(q), _p(d), len(q), int(hamming_thr), int(use_stereo), _p(idx), _p(dist), C.byref(n)))
        return idx[:len(q)].copy(), dist[:len(q)].copy(), n.value

    def match_area(self, image, queries, q_desc, hamming_thr=50, lowe_ratio=0.9):
        q = np.ascontiguousarray(queries, PROJ_QUERY_DTYPE); d = np.ascontiguousarray(q_desc, np.uint8)
        idx = np.full(max(len(q), 1), -1, np.int32); dist = np.zeros(max(len(q), 1), np.int32); n = C.c_int32()
        f = self.lib.lpslam_hip_match_area
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(f(self.h, image, _p(q), _p(d), len(q), int(hamming_thr), float(lowe_ratio), _p(idx), _p(dist), C.byref(n)))
        return idx[:len(q)].copy(), dist[:len(q)].copy(), n.value

    def remap_staged(self, image, eye):
        _check(self.lib.lpslam_hip_remap_staged(self.h, image, int(eye)))

    def prefetch(self):
        """context manager: upload / remap / extract / stereo calls of THIS thread go to the context's prefetch stream"""
        ctx = self
        class _Section:
            def __enter__(self_inner):
                _check(ctx.lib.lpslam_hip_prefetch_begin(ctx.h))
            def __exit__(self_inner, *exc):
                _check(ctx.lib.lpslam_hip_prefetch_end(ctx.h))
                return False
        return _Section()

    def prefetch_join(self):
        _check(self.lib.lpslam_hip_prefetch_join(self.h))

    def prefetch_frame(self, image, with_stereo=True):
        """queues the read-back of `image` behind what this thread has enqueued for it: frame(image) then finds the results delivered"""
        _check(self.lib.lpslam_hip_prefetch_frame(self.h, image, 1 if with_stereo else 0))

    def extract(self, n_images):
        _check(self.lib.lpslam_hip_extract(self.h, n_images))

    def extract_range(self, first, n_images):
        _check(self.lib.lpslam_hip_extract_range(self.h, first, n_images))

    def stage(self, name, n_images):
        _check(getattr(self.lib, "lpslam_hip_stage_" + name)(self.h, n_images))

    def keypoints(self, image):
        kp = np.zeros(self.max_kp, KP_DTYPE); desc = np.zeros((self.max_kp, 32), np.uint8)
        n = C.c_int32()
        _check(self.lib.lpslam_hip_get_keypoints(self.h, image, _p(kp), _p(desc), self.max_kp, C.byref(n)))
        return kp[:n.value].copy(), desc[:n.value].copy()

    def frame(self, image):
        """keypoints, descriptors, x_right, depth of one slot in one round trip (lpslam_hip_get_frame)"""
        kp = np.zeros(self.max_kp, KP_DTYPE); desc = np.zeros((self.max_kp, 32), np.uint8)
        xr = np.zeros(self.max_kp, np.float32); dep = np.zeros(self.max_kp, np.float32)
        n = C.c_int32()
        _check(self.lib.lpslam_hip_get_frame(self.h, image, _p(kp), _p(desc), _p(xr), _p(dep), self.max_kp, C.byref(n)))
        return kp[:n.value].copy(), desc[:n.value].copy(), xr[:n.value].copy(), dep[:n.value].copy()

    def frame_view(self, image, with_stereo=True):
        """lpslam_hip_get_frame_view: copies made from the context's page-locked block (valid until the next frame call)"""
        kp = C.c_void_p(); desc = C.c_void_p(); xr = C.c_void_p(); dep = C.c_void_p(); n = C.c_int32()
        f = self.lib.lpslam_hip_get_frame_view
        f.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(f(self.h, image, 1 if with_stereo else 0, C.byref(kp), C.byref(desc), C.byref(xr), C.byref(dep), C.byref(n)))
        m = n.value
        def arr(ptr, dtype, count):
            return np.frombuffer(C.string_at(ptr.value, count * np.dtype(dtype).itemsize), dtype).copy() if (ptr.value and count) else np.zeros(0, dtype)
        return arr(kp, KP_DTYPE, m), arr(desc, np.uint8, 32 * m).reshape(-1, 32), arr(xr, np.float32, m if with_stereo else 0), arr(dep, np.float32, m if with_stereo else 0)

    def pyramid_level(self, image, level):
        out = np.zeros((self.level_h[level], self.level_w[level]), np.uint8)
        _check(self.lib.lpslam_hip_get_pyramid_level(self.h, image, level, _p(out), out.shape[1]))
        return out

    def candidates(self, image, level):
        n = C.c_int32()
        _check(self.lib.lpslam_hip_get_candidates(self.h, image, level, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), CORNER_DTYPE)
        _check(self.lib.lpslam_hip_get_candidates(self.h, image, level, _p(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def set_descriptors(self, image, desc):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        _check(self.lib.lpslam_hip_set_descriptors(self.h, image, _p(desc), len(desc)))

    def match_bf(self, query, train):
        _check(self.lib.lpslam_hip_match_bf(self.h, query, train))

    def match_bf_strided(self, q0, t0, stride, n):
        _check(self.lib.lpslam_hip_match_bf_strided(self.h, q0, t0, stride, n))

    def bf_knn2(self, query):
        bi = np.zeros(self.max_kp, np.int32); bd = np.zeros(self.max_kp, np.int32); sd = np.zeros(self.max_kp, np.int32)
        n = C.c_int32()
        _check(self.lib.lpslam_hip_get_bf_knn2(self.h, query, _p(bi), _p(bd), _p(sd), self.max_kp, C.byref(n)))
        return bi[:n.value].copy(), bd[:n.value].copy(), sd[:n.value].copy()

    def bf_matches(self, query, train, max_dist=50, ratio=0.0, cross_check=False):
        oq = np.zeros(self.max_kp, np.int32); ot = np.zeros(self.max_kp, np.int32); od = np.zeros(self.max_kp, np.int32)
        n = C.c_int32()
        _check(self.lib.lpslam_hip_get_bf_matches(self.h, query, train, int(max_dist), float(ratio), int(cross_check),
                                                  _p(oq), _p(ot), _p(od), self.max_kp, C.addressof(n)))
        return oq[:n.value].copy(), ot[:n.value].copy(), od[:n.value].copy()

    def match_bf_descriptors(self, query, scratch, desc, max_dist=50, ratio=0.0, cross_check=False):
        """set_descriptors(scratch, desc) + match_bf(query, scratch) + bf_matches(query, scratch, ...) as one call with one wait"""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        oq = np.zeros(self.max_kp, np.int32); ot = np.zeros(self.max_kp, np.int32); od = np.zeros(self.max_kp, np.int32)
        n = C.c_int32()
        _check(self.lib.lpslam_hip_match_bf_descriptors(self.h, query, scratch, _p(desc), len(desc), int(max_dist), float(ratio), int(cross_check),
                                                        _p(oq), _p(ot), _p(od), self.max_kp, C.addressof(n)))
        return oq[:n.value].copy(), ot[:n.value].copy(), od[:n.value].copy()

    def desc_store_put(self, key, desc):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        f = self.lib.lpslam_hip_desc_store_put; f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        _check(f(self.h, int(key), _p(desc), len(desc)))

    def desc_store_drop(self, key):
        f = self.lib.lpslam_hip_desc_store_drop; f.argtypes = [C.c_void_p, C.c_int32]
        _check(f(self.h, int(key)))

    def match_bf_stored(self, query, keys, max_dist=50, ratio=0.0, cross_check=False):
        """the slot `query` against every stored descriptor set in `keys`, one call: [(mq, mt, md)] per key"""
        keys = np.ascontiguousarray(keys, np.int32); n = len(keys); cap = self.max_kp
        oq = np.zeros((max(n, 1), cap), np.int32); ot = np.zeros_like(oq); od = np.zeros_like(oq); cnt = np.zeros(max(n, 1), np.int32)
        f = self.lib.lpslam_hip_match_bf_stored
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        _check(f(self.h, int(query), _p(keys), n, int(max_dist), float(ratio), int(cross_check), _p(oq), _p(ot), _p(od), cap, _p(cnt)))
        return [(oq[k, :cnt[k]].copy(), ot[k, :cnt[k]].copy(), od[k, :cnt[k]].copy()) for k in range(n)]

    def match_stereo(self, left, right, fxb, baseline):
        _check(self.lib.lpslam_hip_match_stereo(self.h, left, right, float(fxb), float(baseline)))

    def match_stereo_strided(self, l0, r0, stride, n, fxb, baseline):
        _check(self.lib.lpslam_hip_match_stereo_strided(self.h, l0, r0, stride, n, float(fxb), float(baseline)))

    def stereo(self, left):
        xr = np.zeros(self.max_kp, np.float32); dep = np.zeros(self.max_kp, np.float32); bi = np.zeros(self.max_kp, np.int32)
        n = C.c_int32()
        _check(self.lib.lpslam_hip_get_stereo(self.h, left, _p(xr), _p(dep), _p(bi), self.max_kp, C.byref(n)))
        return xr[:n.value].copy(), dep[:n.value].copy(), bi[:n.value].copy()


class BundleAdjuster:
    """Device-resident bundle-adjustment problem (lpslam_hip_ba_*)."""

    def __init__(self, ctx, poses, fixed, points, obs, cam, robust_kernel=True, build=True):
        self.ctx = ctx
        self.lib = ctx.lib
        poses = np.ascontiguousarray(poses, np.float64); points = np.ascontiguousarray(points, np.float64)
        fixed = np.ascontiguousarray(fixed, np.uint8); obs = np.ascontiguousarray(obs, BA_OBS_DTYPE)
        self.n_poses, self.n_points, self.n_obs = len(poses), len(points), len(obs)
        c = BaCamera(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["fxb"],
                     float(np.sqrt(5.991)) if robust_kernel else 0.0, float(np.sqrt(7.815)) if robust_kernel else 0.0)
        h = C.c_void_p()
        # build=False: the host half alone (lpslam_hip_ba_prepare: no kernel, thread safe); ba_build_batch enqueues the device half of many
        _check((self.lib.lpslam_hip_ba_create if build else self.lib.lpslam_hip_ba_prepare)(ctx.h, _p(poses), _p(fixed), self.n_poses, _p(points), self.n_points,
                                                                                           _p(obs), self.n_obs, C.byref(c), C.byref(h)))
        self.h = h
        ctx._children.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.lib.lpslam_hip_ba_destroy(self.h)
            self.h = None

    __del__ = close

    def solver(self):
        """('band' | 'dense', block half-bandwidth found at creation or -1)"""
        sv, hb = C.c_int32(), C.c_int32()
        _check(self.lib.lpslam_hip_ba_get_solver(self.h, C.byref(sv), C.byref(hb)))
        return ("band" if sv.value == 2 else "dense"), hb.value

    def timeouts(self):
        """(band, update): hand-overs between workgroups that timed out on this problem so far -- expected (0, 0)"""
        a, b = C.c_int32(0), C.c_int32(0)
        _check(self.lib.lpslam_hip_ba_timeouts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_solver(self, name):
        _check(self.lib.lpslam_hip_ba_set_solver(self.h, {"auto": 0, "dense": 1, "band": 2}[name]))

    def set_active(self, active=None):
        a = np.ascontiguousarray(active, np.uint8) if active is not None else None
        _check(self.lib.lpslam_hip_ba_set_active(self.h, _p(a)))

    def optimize(self, robust=True, iters=10):
        log = np.zeros(max(iters, 1), BA_LOG_DTYPE); done = C.c_int32()
        _check(self.lib.lpslam_hip_ba_optimize(self.h, int(robust), int(iters), _p(log), C.byref(done)))
        return log[:done.value].copy()

    def optimize_begin(self, robust=True, iters=10):
        self._iters = int(iters)
        _check(self.lib.lpslam_hip_ba_optimize_begin(self.h, int(robust), int(iters)))

    def optimize_end(self):
        log = np.zeros(max(self._iters, 1), BA_LOG_DTYPE); done = C.c_int32()
        _check(self.lib.lpslam_hip_ba_optimize_end(self.h, _p(log), C.byref(done)))
        return log[:done.value].copy()

    def reset(self):
        _check(self.lib.lpslam_hip_ba_reset(self.h))

    def set_state(self, poses=None, points=None):
        """new creation-time poses / landmarks for the same observation graph (None keeps them); resets the LM state"""
        ps = None if poses is None else np.ascontiguousarray(poses, np.float64)
        pt = None if points is None else np.ascontiguousarray(points, np.float64)
        assert ps is None or ps.shape == (self.n_poses, 7)
        assert pt is None or pt.shape == (self.n_points, 3)
        _check(self.lib.lpslam_hip_ba_set_state(self.h, _p(ps) if ps is not None else None, _p(pt) if pt is not None else None))

    def optimize_profiled(self, robust=True, iters=10):
        """per-kernel HIP-event times of one optimize() call: {kernel: (ms summed, marks, launches per mark)}, iterations"""
        t = BaKernelTimes()
        _check(self.lib.lpslam_hip_ba_optimize_profiled(self.h, int(robust), int(iters), C.byref(t)))
        band = self.solver()[0] == "band"
        one_pass = t.launches[5] == 0 and t.launches[6] > 0      # back substitution + trial + linearisation in one launch (ba_update.inl)
        names = ["k_ba_lin", "k_ba_point_sum", "k_schur_group" if band else "k_ba_schur", "chol", "k_chol_xsolve", "k_ba_backsub", "k_ba_update" if one_pass else "k_ba_trial", "k_schur_band_reduce"]
        return {n: (t.ms[i], t.launches[i], t.launches_per_mark[i]) for i, n in enumerate(names) if t.launches[i] or n == "chol"}, t.iterations, t.dim

    def pose_optimize(self):
        out = np.zeros(max(self.n_obs, 1), np.uint8); n = C.c_int32()
        _check(self.lib.lpslam_hip_ba_pose_optimize(self.h, _p(out), C.byref(n)))
        return out[:self.n_obs], n.value

    def local(self, first=5, second=10):
        out = np.zeros(max(self.n_obs, 1), np.uint8)
        _check(self.lib.lpslam_hip_ba_local(self.h, int(first), int(second), _p(out)))
        return out[:self.n_obs]

    def state(self):
        poses = np.zeros((self.n_poses, 7)); points = np.zeros((self.n_points, 3))
        _check(self.lib.lpslam_hip_ba_get(self.h, _p(poses), _p(points)))
        return poses, points

    def chi2(self):
        chi = np.zeros(max(self.n_obs, 1)); pos = np.zeros(max(self.n_obs, 1), np.uint8)
        _check(self.lib.lpslam_hip_ba_chi2(self.h, _p(chi), _p(pos)))
        return chi[:self.n_obs], pos[:self.n_obs]


    def optimize_partitioned(self, comm, robust=True, iters=10):
        """lpslam_hip_ba_optimize_partitioned: `comm` is an RcclComm (or a raw ncclComm_t value)"""
        log = np.zeros(max(iters, 1), BA_LOG_DTYPE); done = C.c_int32()
        f = self.lib.lpslam_hip_ba_optimize_partitioned
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        _check(f(self.h, getattr(comm, "comm", comm), int(robust), int(iters), _p(log), C.addressof(done)))
        return log[:done.value].copy()

    # ---- partitioned (multi-GPU) solve: phases of one LM trial (see include/lpslam_hip.h) ----
    def reduced_buffer(self):
        ptr = C.c_void_p(); n = C.c_int64()
        _check(self.lib.lpslam_hip_ba_reduced_buffer(self.h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def scalar_buffer(self):
        ptr = C.c_void_p(); n = C.c_int64()
        _check(self.lib.lpslam_hip_ba_scalar_buffer(self.h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def step_begin(self, robust, first):
        _check(self.lib.lpslam_hip_ba_step_begin(self.h, int(robust), int(first)))

    def step_lambda0(self):
        _check(self.lib.lpslam_hip_ba_step_lambda0(self.h))

    def step_solve(self):
        _check(self.lib.lpslam_hip_ba_step_solve(self.h))

    def step_end(self):
        a = C.c_int32(); f = C.c_int32()
        _check(self.lib.lpslam_hip_ba_step_end(self.h, C.byref(a), C.byref(f)))
        return bool(a.value), bool(f.value)

    def status(self):
        o = C.c_int32(); st = C.c_int32(); lam = C.c_double(); chi = C.c_double()
        _check(self.lib.lpslam_hip_ba_status(self.h, C.byref(o), C.byref(st), C.byref(lam), C.byref(chi)))
        return dict(outer_done=o.value, stopped=bool(st.value), lam=lam.value, chi2=chi.value)


class Vocabulary:
    """lpslam_hip_vocab: a DBoW2 vocabulary tree in HBM.  nodes: structured array (parent, desc[32], weight, is_leaf) in file order."""

    def __init__(self, ctx, k, L, parent, desc, weight, is_leaf):
        self.ctx, self.lib = ctx, ctx.lib
        parent = np.ascontiguousarray(parent, np.int32); desc = np.ascontiguousarray(desc, np.uint8)
        weight = np.ascontiguousarray(weight, np.float32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        assert desc.shape == (len(parent), 32)
        h = C.c_void_p()
        f = self.lib.lpslam_hip_vocab_create
        f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        _check(f(ctx.h, int(k), int(L), len(parent), _p(parent), _p(desc), _p(weight), _p(is_leaf), C.byref(h)))
        self.h = h
        ctx._children.add(self)
        k_, L_, nn, nw = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.lpslam_hip_vocab_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 4
        _check(self.lib.lpslam_hip_vocab_info(self.h, C.byref(k_), C.byref(L_), C.byref(nn), C.byref(nw)))
        self.k, self.L, self.n_nodes, self.n_words = k_.value, L_.value, nn.value, nw.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.lpslam_hip_vocab_destroy.argtypes = [C.c_void_p]
            self.lib.lpslam_hip_vocab_destroy(self.h)
            self.h = None

    __del__ = close

    def transform(self, image, levels_up=4):
        """word id, word weight, node id per keypoint of an image slot"""
        cap = self.ctx.max_kp
        w = np.zeros(cap, np.int32); wt = np.zeros(cap, np.float32); nd = np.zeros(cap, np.int32); n = C.c_int32()
        f = self.lib.lpslam_hip_bow_transform
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        _check(f(self.ctx.h, self.h, int(image), int(levels_up), _p(w), _p(wt), _p(nd), cap, C.byref(n)))
        return w[:n.value].copy(), wt[:n.value].copy(), nd[:n.value].copy()

    def transform_host(self, desc, levels_up=4):
        d = np.ascontiguousarray(desc, np.uint8)
        n = len(d)
        w = np.zeros(max(n, 1), np.int32); wt = np.zeros(max(n, 1), np.float32); nd = np.zeros(max(n, 1), np.int32)
        f = self.lib.lpslam_hip_bow_transform_host
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(f(self.ctx.h, self.h, _p(d), n, int(levels_up), _p(w), _p(wt), _p(nd)))
        return w[:n], wt[:n], nd[:n]


def match_bow_tree(ctx, q_desc, q_node, t_desc, t_node, hamming_thr=50, lowe_ratio=0.75, t_taken=None):
    qd = np.ascontiguousarray(q_desc, np.uint8); qn = np.ascontiguousarray(q_node, np.int32)
    td = np.ascontiguousarray(t_desc, np.uint8); tn = np.ascontiguousarray(t_node, np.int32)
    tk = np.ascontiguousarray(t_taken, np.uint8) if t_taken is not None else None
    idx = np.full(max(len(qn), 1), -1, np.int32); dist = np.zeros(max(len(qn), 1), np.int32); n = C.c_int32()
    f = ctx.lib.lpslam_hip_match_bow_tree
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    _check(f(ctx.h, _p(qd), _p(qn), len(qn), _p(td), _p(tn), len(tn), _p(tk), int(hamming_thr), float(lowe_ratio), _p(idx), _p(dist), C.byref(n)))
    return idx[:len(qn)], dist[:len(qn)], n.value


def match_bow_tree_multi(ctx, q_desc, q_node, t_descs, t_nodes, hamming_thr=50, lowe_ratio=0.75, t_takens=None):
    """one query set against several target sets in one round trip (lpslam_hip_match_bow_tree_multi); a list of (idx, dist, n) per set"""
    qd = np.ascontiguousarray(q_desc, np.uint8); qn = np.ascontiguousarray(q_node, np.int32)
    ns = len(t_descs)
    tds = [np.ascontiguousarray(t, np.uint8) for t in t_descs]; tns = [np.ascontiguousarray(t, np.int32) for t in t_nodes]
    tks = [np.ascontiguousarray(t, np.uint8) if t is not None else None for t in (t_takens or [None] * ns)]
    idx = [np.full(max(len(qn), 1), -1, np.int32) for _ in range(ns)]; dist = [np.zeros(max(len(qn), 1), np.int32) for _ in range(ns)]
    arr = lambda xs: (C.c_void_p * max(ns, 1))(*[x.ctypes.data if x is not None and x.size else None for x in xs])
    nts = (C.c_int32 * max(ns, 1))(*[len(t) for t in tns]); nm = (C.c_int32 * max(ns, 1))()
    f = ctx.lib.lpslam_hip_match_bow_tree_multi
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(f(ctx.h, _p(qd), _p(qn), len(qn), ns, arr(tds), arr(tns), nts, arr(tks) if t_takens is not None else None, int(hamming_thr), float(lowe_ratio), arr(idx), arr(dist), nm))
    return [(idx[i][:len(qn)], dist[i][:len(qn)], int(nm[i])) for i in range(ns)]


class RcclComm:
    """An ncclComm_t made through ctypes (tests and bench.py: the product's caller is C++ and links RCCL itself).  `unique_id` is the
    128-byte id of rank 0 (RcclComm.unique_id()), handed to the other ranks by the caller's own means."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            last = None
            for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"):
                try:
                    cls._lib = C.CDLL(name, mode=C.RTLD_GLOBAL); break
                except OSError as e:
                    last = e
            if cls._lib is None:
                raise LpslamHipError("RCCL not found: %s" % last)
        return cls._lib

    @classmethod
    def unique_id(cls):
        buf = C.create_string_buffer(128)
        rc = cls.lib().ncclGetUniqueId(buf)
        if rc != 0:
            raise LpslamHipError("ncclGetUniqueId failed: %d" % rc)
        return buf.raw

    def __init__(self, unique_id, world, rank):
        class Uid(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid = Uid(); C.memmove(C.byref(uid), unique_id, 128)
        comm = C.c_void_p()
        f = self.lib().ncclCommInitRank
        f.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
        rc = f(C.byref(comm), int(world), uid, int(rank))
        if rc != 0:
            raise LpslamHipError("ncclCommInitRank failed: %d" % rc)
        self.comm = comm

    def close(self):
        if getattr(self, "comm", None):
            f = self.lib().ncclCommDestroy; f.argtypes = [C.c_void_p]
            f(self.comm); self.comm = None


def ba_factor_kernel_name(dim, band=False):
    """name of the kernel that factors a reduced system of `dim` unknowns (rule of enqueue_solve in csrc/ba.hip)"""
    return "k_chol_band" if band else "k_chol_pair"          # k_chol_wg takes over only in batches of 40 dense problems and more


def ba_build_batch(problems):
    """the device half of creation (structure build) for problems made with build=False: one launch chain for all of them"""
    lib = load()
    arr = (C.c_void_p * len(problems))(*[p.h for p in problems])
    _check(lib.lpslam_hip_ba_build_batch(arr, len(problems)))


def ba_set_state_batch(problems, poses, points):
    """lpslam_hip_ba_set_state of every problem in one call (poses[i] / points[i]: arrays or None)"""
    n = len(problems)
    ps = [None if a is None else np.ascontiguousarray(a, np.float64) for a in poses]
    pt = [None if a is None else np.ascontiguousarray(a, np.float64) for a in points]
    hs = (C.c_void_p * n)(*[p.h for p in problems])
    pa = (C.c_void_p * n)(*[None if a is None else a.ctypes.data for a in ps])
    ta = (C.c_void_p * n)(*[None if a is None else a.ctypes.data for a in pt])
    f = problems[0].lib.lpslam_hip_ba_set_state_batch
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    _check(f(hs, n, pa, ta))


def ba_state_batch(problems):
    """lpslam_hip_ba_get of every problem in one call: [(poses, points)]"""
    n = len(problems)
    out = [(np.empty((p.n_poses, 7)), np.empty((p.n_points, 3))) for p in problems]
    hs = (C.c_void_p * n)(*[p.h for p in problems])
    pa = (C.c_void_p * n)(*[o[0].ctypes.data for o in out])
    ta = (C.c_void_p * n)(*[o[1].ctypes.data for o in out])
    f = problems[0].lib.lpslam_hip_ba_get_batch
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    _check(f(hs, n, pa, ta))
    return out


def ba_optimize_batch(problems, robust=True, iters=10):
    """lpslam_hip_ba_optimize_batch: the problems (BundleAdjuster objects on one device) advanced by one launch chain.
    Returns one iteration log per problem."""
    n = len(problems)
    lib = problems[0].lib
    hs = (C.c_void_p * n)(*[p.h for p in problems])
    stride = max(int(iters), 1)
    logs = np.zeros((n, stride), BA_LOG_DTYPE); done = np.zeros(n, np.int32)
    _check(lib.lpslam_hip_ba_optimize_batch(hs, n, int(robust), int(iters), _p(logs), stride, _p(done)))
    return [logs[i, :done[i]].copy() for i in range(n)]


def ba_reset_batch(problems):
    n = len(problems)
    hs = (C.c_void_p * n)(*[p.h for p in problems])
    _check(problems[0].lib.lpslam_hip_ba_reset_batch(hs, n))


def ba_obs_array(prob):
    o = np.zeros(len(prob["obs_pose"]), BA_OBS_DTYPE)
    o["pose"] = prob["obs_pose"]; o["point"] = prob["obs_point"]
    o["u"] = prob["obs_uvr"][:, 0]; o["v"] = prob["obs_uvr"][:, 1]; o["ur"] = prob["obs_uvr"][:, 2]
    o["inv_sigma2"] = prob["obs_inv_sigma2"]
    return o


def ba_local_window(ctx, poses, fixed, points, obs, cam, first=5, second=10):
    """a keyframe's local bundle adjustment in one call (lpslam_hip_ba_local_window): returns (poses, points, outlier mask)"""
    po = np.ascontiguousarray(poses, np.float64).copy(); pt = np.ascontiguousarray(points, np.float64).copy()
    fx = np.ascontiguousarray(fixed, np.uint8); o = np.ascontiguousarray(obs, BA_OBS_DTYPE)
    c = BaCamera(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["fxb"], float(np.sqrt(5.991)), float(np.sqrt(7.815)))
    out = np.zeros(max(len(o), 1), np.uint8)
    f = ctx.lib.lpslam_hip_ba_local_window
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    _check(f(ctx.h, _p(po), _p(fx), len(po), _p(pt), len(pt), _p(o), len(o), C.byref(c), int(first), int(second), _p(out)))
    return po, pt, out[:len(o)].astype(bool)


def sim3_edges(ei, ej, meas):
    e = np.zeros(len(ei), SIM3_EDGE_DTYPE)
    e["i"] = ei; e["j"] = ej; e["meas"] = meas
    return e


class PoseGraph:
    """Device-resident Sim3 pose graph (lpslam_hip_sim3_*): vertices n x 8 (qw qx qy qz tx ty tz s)."""

    def __init__(self, ctx, verts, fixed, edges, fix_scale=True):
        self.ctx = ctx
        self.lib = ctx.lib
        verts = np.ascontiguousarray(verts, np.float64); fixed = np.ascontiguousarray(fixed, np.uint8)
        edges = np.ascontiguousarray(edges, SIM3_EDGE_DTYPE)
        self.n, self.n_edges = len(verts), len(edges)
        h = C.c_void_p()
        _check(self.lib.lpslam_hip_sim3_create(ctx.h, _p(verts), _p(fixed), self.n, _p(edges), self.n_edges, int(fix_scale), C.byref(h)))
        self.h = h
        ctx._children.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.lib.lpslam_hip_sim3_destroy(self.h)
            self.h = None

    __del__ = close

    def optimize(self, iters=50):
        log = np.zeros(max(iters, 1), BA_LOG_DTYPE); done = C.c_int32()
        _check(self.lib.lpslam_hip_sim3_optimize(self.h, int(iters), _p(log), C.byref(done)))
        return log[:done.value].copy()

    def get(self):
        v = np.zeros((self.n, 8))
        _check(self.lib.lpslam_hip_sim3_get(self.h, _p(v)))
        return v

    def chi2(self):
        c = np.zeros(self.n_edges)
        _check(self.lib.lpslam_hip_sim3_chi2(self.h, _p(c)))
        return c


def sim3_pairs(prob):
    p = np.zeros(len(prob["p1c"]), SIM3_PAIR_DTYPE)
    for k in ("p1c", "p2c", "obs1", "obs2", "inv_sigma2_1", "inv_sigma2_2"):
        p[k] = prob[k]
    return p


def sim3_transform_optimize(ctx, s12, pairs_list, cam1, cam2, chi_sq=10.0, fix_scale=True):
    """Batch of loop candidates (lpslam_hip_sim3_transform_optimize): s12 (n x 8), one pair array per candidate.
    Returns (s12 n x 8, list of inlier masks, inlier counts)."""
    s = np.ascontiguousarray(np.atleast_2d(s12), np.float64).copy()
    n = len(pairs_list)
    start = np.zeros(n + 1, np.int32); start[1:] = np.cumsum([len(p) for p in pairs_list])
    pairs = np.ascontiguousarray(np.concatenate(pairs_list) if n else np.zeros(0, SIM3_PAIR_DTYPE), SIM3_PAIR_DTYPE)
    c1 = np.ascontiguousarray(cam1, np.float64); c2 = np.ascontiguousarray(cam2, np.float64)
    inl = np.zeros(max(len(pairs), 1), np.uint8); cnt = np.zeros(max(n, 1), np.int32)
    f = ctx.lib.lpslam_hip_sim3_transform_optimize
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]
    _check(f(ctx.h, n, _p(s), _p(pairs), _p(start), _p(c1), _p(c2), float(chi_sq), int(fix_scale), _p(inl), _p(cnt)))
    return s, [inl[start[i]:start[i + 1]].astype(bool) for i in range(n)], cnt[:n].copy()


def match_orientation_filter(angle_q, angle_t, match_idx):
    aq = np.ascontiguousarray(angle_q, np.float32); at = np.ascontiguousarray(angle_t, np.float32)
    m = np.ascontiguousarray(match_idx, np.int32).copy(); n = C.c_int32()
    _check(load().lpslam_hip_match_orientation_filter(_p(aq), _p(at), _p(m), len(m), C.byref(n)))
    return m, n.value


def pose_optimize(ctx, pose7, points, obs, cam, robust_kernel=True):
    """optimize::pose_optimizer in one launch (lpslam_hip_pose_optimize): returns (pose7, outlier mask, inliers)."""
    pose = np.ascontiguousarray(pose7, np.float64).copy(); pts = np.ascontiguousarray(points, np.float64)
    o = np.ascontiguousarray(obs, BA_OBS_DTYPE)
    c = BaCamera(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["fxb"],
                 float(np.sqrt(5.991)) if robust_kernel else 0.0, float(np.sqrt(7.815)) if robust_kernel else 0.0)
    out = np.zeros(max(len(o), 1), np.uint8); n = C.c_int32()
    _check(ctx.lib.lpslam_hip_pose_optimize(ctx.h, _p(pose), _p(pts), len(pts), _p(o), len(o), C.byref(c), _p(out), C.byref(n)))
    return pose, out[:len(o)].astype(bool), n.value
