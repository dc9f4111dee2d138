// replay.cpp -- see replay.h
#include "replay.h"
#include "jpeg.h"
#include <cstring>
#include <fstream>

namespace LpSlam {
namespace {

struct Cursor { const uint8_t* p; const uint8_t* end; bool ok = true; };

uint64_t varint(Cursor& c)
{
    uint64_t v = 0;
    for (int shift = 0; shift < 70; shift += 7) {
        if (c.p >= c.end) { c.ok = false; return 0; }
        const uint8_t b = *c.p++;
        v |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) return v;
    }
    c.ok = false;
    return 0;
}
// one field: number, wire type, and either the scalar value or the byte range of a length-delimited field
struct Field { uint32_t number = 0; int wire = 0; uint64_t value = 0; const uint8_t* data = nullptr; size_t size = 0; };

bool next_field(Cursor& c, Field& f)
{
    if (c.p >= c.end) return false;
    const uint64_t key = varint(c);
    if (!c.ok) return false;
    f.number = (uint32_t)(key >> 3); f.wire = (int)(key & 7);
    switch (f.wire) {
    case 0: f.value = varint(c); return c.ok;
    case 1: if (c.end - c.p < 8) { c.ok = false; return false; } std::memcpy(&f.value, c.p, 8); c.p += 8; return true;
    case 2: { const uint64_t n = varint(c); if (!c.ok || (uint64_t)(c.end - c.p) < n) { c.ok = false; return false; } f.data = c.p; f.size = (size_t)n; c.p += n; return true; }
    case 5: if (c.end - c.p < 4) { c.ok = false; return false; } { uint32_t v; std::memcpy(&v, c.p, 4); f.value = v; } c.p += 4; return true;
    default: c.ok = false; return false;     // groups (3, 4) do not occur in proto3 files
    }
}
double as_double(const Field& f) { double d; std::memcpy(&d, &f.value, 8); return d; }

// Position / Orientation: doubles x y z (1..3) resp. w x y z (1..4)
void parse_vec(const uint8_t* d, size_t n, double* out, int count)
{
    Cursor c{d, d + n};
    Field f;
    while (next_field(c, f)) if (f.wire == 1 && f.number >= 1 && (int)f.number <= count) out[f.number - 1] = as_double(f);
}
void parse_state(const uint8_t* d, size_t n, ReplayState& s)      // GlobalState / TrackerCoordinateSystem: position = 1, orientation = 2
{
    Cursor c{d, d + n};
    Field f;
    s.orientation[0] = 0;                                           // proto3 default of an absent double
    while (next_field(c, f)) {
        if (f.wire != 2) continue;
        if (f.number == 1) parse_vec(f.data, f.size, s.position, 3);
        else if (f.number == 2) parse_vec(f.data, f.size, s.orientation, 4);
    }
}

}  // namespace

bool decode_pgm(const uint8_t* d, size_t size, GrayImage& out)
{
    size_t i = 0;
    auto token = [&](long& v) {
        for (;;) {                                                  // whitespace and comments
            while (i < size && (d[i] == ' ' || d[i] == '\t' || d[i] == '\r' || d[i] == '\n')) ++i;
            if (i < size && d[i] == '#') { while (i < size && d[i] != '\n') ++i; continue; }
            break;
        }
        if (i >= size || d[i] < '0' || d[i] > '9') return false;
        v = 0;
        while (i < size && d[i] >= '0' && d[i] <= '9') { v = v * 10 + (d[i] - '0'); if (v > (1L << 30)) return false; ++i; }
        return true;
    };
    if (size < 7 || d[0] != 'P' || d[1] != '5') return false;
    i = 2;
    long w = 0, h = 0, maxv = 0;
    if (!token(w) || !token(h) || !token(maxv)) return false;
    if (w <= 0 || h <= 0 || maxv <= 0 || maxv > 255 || i >= size) return false;
    ++i;                                                            // the single whitespace byte after maxval
    if ((size_t)w * (size_t)h > size - i) return false;
    out.width = (int)w; out.height = (int)h;
    out.pixels.assign(d + i, d + i + (size_t)w * (size_t)h);
    return true;
}

// an image payload of a record: baseline JPEG (what the reference's recorder writes: cv::imencode(".jpg"), RecordEngine.cpp:93) or PGM
bool decode_image(const uint8_t* d, size_t size, GrayImage& out)
{
    return looks_like_jpeg(d, size) ? decode_jpeg_gray(d, size, out) : decode_pgm(d, size, out);
}

bool ReplayReader::open(const std::string& path, std::string* err)
{
    m_in.close(); m_in.clear();
    m_in.open(path, std::ios::binary);
    m_done = false; m_stats = ReplayStats{};
    if (!m_in) { if (err) *err = "cannot open " + path; m_done = true; return false; }
    return true;
}

// The next decodable camera frame; false at the end of the stream (clean, or the first record the format does not know).
bool ReplayReader::next(ReplayFrame& out)
{
    constexpr uint64_t kMaxMessage = 1ull << 30;                    // ProtoStream refuses larger buffers as corrupt
    std::ifstream& in = m_in;
    ReplayStats& stats = m_stats;
    std::vector<uint8_t>& buf = m_buf;
    while (!m_done) {
        uint64_t type = 0, size = 0;
        if (!in.read(reinterpret_cast<char*>(&type), 8)) break;     // clean end of file
        if (!in.read(reinterpret_cast<char*>(&size), 8) || size > kMaxMessage) { stats.truncated = true; break; }
        if (type < 1 || type > 5) { stats.truncated = true; break; }
        buf.resize((size_t)size);
        if (size && !in.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)size)) { stats.truncated = true; break; }
        ++stats.records;
        if (type == 2) { ++stats.imu; continue; }
        if (type == 3) { ++stats.global_state; continue; }
        if (type == 4) { ++stats.result; continue; }
        if (type == 5) { ++stats.feature; continue; }
        ++stats.camera;
        ReplayFrame fr;
        const uint8_t *img = nullptr, *img2 = nullptr;
        size_t img_n = 0, img2_n = 0;
        bool has_base2 = false, has_odom = false, has_map = false, flag_odom = false, flag_map = false;
        ReplayState odom, map;
        Cursor c{buf.data(), buf.data() + buf.size()};
        Field f;
        while (next_field(c, f)) {
            switch (f.number) {
            case 1: if (f.wire == 0) fr.timestamp = (int64_t)f.value; break;
            case 2: if (f.wire == 0) fr.data_number = (int64_t)f.value; break;
            case 3: if (f.wire == 2) { img = f.data; img_n = f.size; } break;
            case 4: if (f.wire == 2) { parse_state(f.data, f.size, odom); has_odom = true; } break;
            case 5: if (f.wire == 2) { parse_state(f.data, f.size, map); has_map = true; } break;
            case 6: if (f.wire == 0) fr.camera = (int32_t)f.value; break;
            case 7: if (f.wire == 2) { img2 = f.data; img2_n = f.size; } break;
            case 8: if (f.wire == 0) fr.camera_second = (int32_t)f.value; break;
            case 10: if (f.wire == 2) has_base2 = true; break;     // imageBase_second marks a stereo record (ReplayEngine.cpp:150)
            case 11: if (f.wire == 0) flag_odom = f.value != 0; break;
            case 12: if (f.wire == 0) flag_map = f.value != 0; break;
            default: break;
            }
        }
        if (!c.ok) { stats.truncated = true; break; }
        if (!img || !decode_image(img, img_n, fr.image)) { ++stats.undecodable_images; continue; }
        if (has_base2) {
            GrayImage second;
            if (!img2 || !decode_image(img2, img2_n, second)) { ++stats.undecodable_images; continue; }
            fr.image_second = std::move(second);
        }
        if (has_odom && flag_odom) fr.odom = odom;
        if (has_map && flag_map) fr.map = map;
        out = std::move(fr);
        return true;
    }
    m_done = true;
    return false;
}

bool read_replay_file(const std::string& path, std::vector<ReplayFrame>& frames, ReplayStats& stats, std::string* err)
{
    ReplayReader r;
    if (!r.open(path, err)) return false;
    ReplayFrame f;
    while (r.next(f)) frames.push_back(std::move(f));
    stats = r.stats();
    return true;
}

}  // namespace LpSlam

// C shim for the tests: stats[0..7] = records, camera, imu, global_state, result, feature, undecodable, truncated; returns the
// number of decodable frames; first[0..5] = timestamp, camera, camera_second, width, height, stereo of the first frame,
// first_state[0..13] = odom position(3) orientation(4), map position(3) orientation(4) of the first frame (NaN when absent)
extern "C" __attribute__((visibility("default"))) long lpslam_replay_probe(const char* path, long* stats, long* first, double* first_state)
{
    std::vector<LpSlam::ReplayFrame> frames;
    LpSlam::ReplayStats st;
    if (!LpSlam::read_replay_file(path, frames, st, nullptr)) return -1;
    const long v[8] = {(long)st.records, (long)st.camera, (long)st.imu, (long)st.global_state, (long)st.result, (long)st.feature, (long)st.undecodable_images, st.truncated ? 1 : 0};
    for (int i = 0; i < 8; ++i) stats[i] = v[i];
    if (!frames.empty()) {
        const LpSlam::ReplayFrame& f = frames[0];
        first[0] = (long)f.timestamp; first[1] = f.camera; first[2] = f.camera_second; first[3] = f.image.width; first[4] = f.image.height; first[5] = f.image_second ? 1 : 0;
        for (int i = 0; i < 14; ++i) first_state[i] = __builtin_nan("");
        if (f.odom) { for (int i = 0; i < 3; ++i) first_state[i] = f.odom->position[i]; for (int i = 0; i < 4; ++i) first_state[3 + i] = f.odom->orientation[i]; }
        if (f.map) { for (int i = 0; i < 3; ++i) first_state[7 + i] = f.map->position[i]; for (int i = 0; i < 4; ++i) first_state[10 + i] = f.map->orientation[i]; }
    }
    return (long)frames.size();
}
