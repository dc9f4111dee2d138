// replay.h -- reader of lpslam's recording stream, the data format on the input side of the path.
// Wire format (reference: src/Serialize/ProtoStream.h:15-60, src/Serialize/MessageTypes.h:5-11): a sequence of records
//     u64 message type | u64 payload size | payload = protobuf (proto3) message of src/Serialize/SlamSerialize.proto
// with types CameraImage = 1, SensorImu = 2, SensorGlobalState = 3, Result = 4, SensorFeatureList = 5.  ReplayEngine
// (src/Manager/ReplayEngine.cpp:83-242) turns CameraImage records into camera-queue entries -- image(s) decoded with
// cv::imdecode, camera numbers, the odometry / map state stored with the frame -- and the others into sensor-queue entries.
// This reader walks the records with a hand-written protobuf varint / length-delimited parser (no protoc in the image) and
// decodes image payloads that are baseline JPEG (what the reference's recorder writes: cv::imencode(".jpg"), RecordEngine.cpp:93;
// jpeg.h -- libjpeg's samples, grey; a colour record gives its luma plane) or binary PGM ("P5", maxval 255), which cv::imdecode
// reads natively too; progressive JPEG / PNG payloads are counted and skipped.
#pragma once
#include <cstdint>
#include <fstream>
#include <optional>
#include <string>
#include <vector>
#include "core.h"

namespace LpSlam {

struct ReplayState {                     // message GlobalState (position, orientation; velocity is not used on the path)
    double position[3] = {0, 0, 0};
    double orientation[4] = {1, 0, 0, 0};    // w x y z
};

struct ReplayFrame {
    int64_t timestamp = 0;               // nanoseconds since the epoch (CameraImage.timeStamp)
    int64_t data_number = 0;
    int32_t camera = 0, camera_second = 0;
    GrayImage image;
    std::optional<GrayImage> image_second;
    std::optional<ReplayState> odom, map;
};

struct ReplayStats {
    size_t records = 0, camera = 0, imu = 0, global_state = 0, result = 0, feature = 0, undecodable_images = 0;
    bool truncated = false;              // the reference also stops at the first unknown record type (corrupt tails happen)
};

// Walks the stream one camera frame at a time, so a long recording is never resident as a whole (the reference reads
// `replay_chunks` camera records at a time as its camera queue drains, ReplayEngine.cpp:77-100).
class ReplayReader {
public:
    bool open(const std::string& path, std::string* err);
    bool next(ReplayFrame& out);
    bool done() const { return m_done; }
    const ReplayStats& stats() const { return m_stats; }
private:
    std::ifstream m_in;
    std::vector<uint8_t> m_buf;
    ReplayStats m_stats;
    bool m_done = true;
};

bool read_replay_file(const std::string& path, std::vector<ReplayFrame>& frames, ReplayStats& stats, std::string* err);
// binary PGM (P5, maxval <= 255) -> gray image; false for anything else
bool decode_pgm(const uint8_t* data, size_t size, GrayImage& out);
bool decode_image(const uint8_t* data, size_t size, GrayImage& out);     // baseline JPEG (jpeg.h) or PGM

}  // namespace LpSlam
