"""Writer of lpslam's recording stream for the tests (the reader under test is lpslam_amd/host/replay.cpp).
Record = u64 type | u64 size | proto3 message of the reference's src/Serialize/SlamSerialize.proto (field numbers below),
written field by field with a few lines of protobuf wire encoding.  Images are binary PGM, which cv::imdecode reads."""
import struct


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _tag(n, wire):
    return _varint((n << 3) | wire)


def f_varint(n, v):
    return _tag(n, 0) + _varint(int(v))


def f_double(n, v):
    return _tag(n, 1) + struct.pack("<d", float(v))


def f_bytes(n, b):
    return _tag(n, 2) + _varint(len(b)) + bytes(b)


def vec3(x, y, z):
    return f_double(1, x) + f_double(2, y) + f_double(3, z)


def orientation(w, x, y, z):
    return f_double(1, w) + f_double(2, x) + f_double(3, y) + f_double(4, z)


def global_state(pos, quat):
    return f_bytes(1, vec3(*pos)) + f_bytes(2, orientation(*quat))


def pgm(img):
    h, w = img.shape
    return b"P5\n# lpslam test\n%d %d\n255\n" % (w, h) + img.tobytes()


def camera_image(ts, left, right=None, cam=0, odom=None, map_=None, data_number=0, raw_left=None):
    m = f_varint(1, ts) + f_varint(2, data_number) + f_bytes(3, raw_left if raw_left is not None else pgm(left))
    if odom is not None:
        m += f_bytes(4, global_state(*odom))
    if map_ is not None:
        m += f_bytes(5, global_state(*map_))
    m += f_varint(6, cam)
    if right is not None:
        m += f_bytes(7, pgm(right)) + f_varint(8, cam + 1) + f_bytes(10, global_state((0, 0, 0), (1, 0, 0, 0)))
    m += f_bytes(9, global_state((0, 0, 0), (1, 0, 0, 0)))
    if odom is not None:
        m += f_varint(11, 1)
    if map_ is not None:
        m += f_varint(12, 1)
    return m


def record(msg_type, payload):
    return struct.pack("<QQ", msg_type, len(payload)) + payload


CAMERA_IMAGE, SENSOR_IMU, SENSOR_GLOBAL_STATE, RESULT, SENSOR_FEATURE = 1, 2, 3, 4, 5
