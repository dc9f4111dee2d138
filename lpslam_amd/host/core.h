// core.h -- host-side data types and plugin surfaces of the drop-in runtime (C++17, no third-party dependencies).
//
// Mirrors the shapes of the reference's runtime types with plain buffers in place of cv::Mat / Eigen / TBB:
//   CameraQueueEntry, SensorQueueEntry, ResultQueueEntry   src/DataTypes/{CameraQueue,SensorQueue,ResultQueue}.h
//   Position3 / Orientation / GlobalState(InTime)           src/DataTypes/Space.h:14-197
//   TrackerBase / TrackerResult                             src/Trackers/TrackerBase.h:34-150
//   ProcessorBase                                           src/Processor/ProcessorBase.h:10-23
//   CameraRegistry                                          src/Manager/CameraRegistry.h:11-20
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <vector>

#include "../../include/lpslam_types.h"
#include "json_min.h"

namespace LpSlam {

using TimeStamp = std::chrono::time_point<std::chrono::high_resolution_clock>;
inline TimeStamp int64ToTimeStamp(int64_t ns) { return TimeStamp(std::chrono::duration_cast<TimeStamp::duration>(std::chrono::nanoseconds(ns))); }
inline int64_t timeStampToInt64(TimeStamp t) { return std::chrono::duration_cast<std::chrono::nanoseconds>(t.time_since_epoch()).count(); }

struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double w = 1, x = 0, y = 0, z = 0; };
struct Position3 { Vector3 value; Vector3 sigma; };
struct Orientation { Quaternion value; double sigma = 0; };
struct GlobalState { Position3 position; Orientation orientation; bool stateValid = true; };
struct CompositeTimestamp { TimeStamp system_time{}; std::optional<LpSlamROSTimestamp> ros_timestamp; };
struct GlobalStateInTime { CompositeTimestamp first; GlobalState second; };

// POD <-> internal conversion (src/InterfaceImpl/LpSlamConversion.h:11-75)
namespace conversion {
inline GlobalStateInTime gsInTimeInterfaceToInternal(const LpSlamGlobalStateInTime& s) {
    GlobalStateInTime g;
    g.first.system_time = int64ToTimeStamp(s.timestamp);
    if (s.has_ros_timestamp) g.first.ros_timestamp = s.ros_timestamp;
    g.second.position.value = {s.state.position.x, s.state.position.y, s.state.position.z};
    g.second.position.sigma = {s.state.position.x_sigma, s.state.position.y_sigma, s.state.position.z_sigma};
    g.second.orientation.value = {s.state.orientation.w, s.state.orientation.x, s.state.orientation.y, s.state.orientation.z};
    g.second.orientation.sigma = s.state.orientation.sigma;
    g.second.stateValid = s.state.valid;
    return g;
}
inline LpSlamGlobalStateInTime gsInTimeInternalToInterface(const GlobalStateInTime& g) {
    LpSlamGlobalStateInTime s{};
    s.timestamp = timeStampToInt64(g.first.system_time);
    s.has_ros_timestamp = g.first.ros_timestamp.has_value() ? 1 : 0;
    if (g.first.ros_timestamp) s.ros_timestamp = *g.first.ros_timestamp;
    s.state.position = {g.second.position.value.x, g.second.position.value.y, g.second.position.value.z,
                        g.second.position.sigma.x, g.second.position.sigma.y, g.second.position.sigma.z};
    s.state.orientation = {g.second.orientation.value.w, g.second.orientation.value.x, g.second.orientation.value.y,
                           g.second.orientation.value.z, g.second.orientation.sigma};
    s.state.valid = g.second.stateValid;
    return s;
}
}  // namespace conversion

// One grey image owned by the queue entry (the reference aliases caller memory for 8UC1 stereo,
// src/Manager/SlamManager.cpp:1082-1085; here the frame is copied at enqueue so the caller may reuse its buffers).
struct GrayImage {
    int width = 0, height = 0;
    std::vector<uint8_t> pixels;      // tightly packed rows
    bool empty() const { return pixels.empty(); }
};

struct CameraQueueEntry {
    bool valid = false;               // false = exit signal for the worker
    TimeStamp timestamp{};
    uint32_t cameraNumber = 0, cameraNumberSecond = 0;
    GrayImage image;
    std::optional<GrayImage> image_second;
    std::optional<LpSlamROSTimestamp> ros_timestamp;
};

struct SensorQueueEntry { TimeStamp timestamp{}; bool valid = true; };
struct ResultQueueEntry { GlobalStateInTime globalStateInTime; bool exitSignal = false; };

template <class T>
class BlockingQueue {      // stands in for tbb::concurrent_bounded_queue with default (unbounded) capacity
public:
    void push(T v) { { std::lock_guard<std::mutex> l(m_); q_.push_back(std::move(v)); } cv_.notify_one(); }
    void pop(T& out) { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return !q_.empty(); }); out = std::move(q_.front()); q_.pop_front(); }
    bool try_pop(T& out) { std::lock_guard<std::mutex> l(m_); if (q_.empty()) return false; out = std::move(q_.front()); q_.pop_front(); return true; }
    size_t size() { std::lock_guard<std::mutex> l(m_); return q_.size(); }
    template <class F> void with_front(F&& f) { std::lock_guard<std::mutex> l(m_); if (!q_.empty()) f(q_.front()); }   // the entry stays queued
    void clear() { std::lock_guard<std::mutex> l(m_); q_.clear(); }
private:
    std::mutex m_; std::condition_variable cv_; std::deque<T> q_;
};
using CameraQueue = BlockingQueue<CameraQueueEntry>;
using SensorQueue = BlockingQueue<SensorQueueEntry>;
using ResultQueue = BlockingQueue<ResultQueueEntry>;

class CameraRegistry {
public:
    void setConfiguration(const LpSlamCameraConfiguration& c) { std::lock_guard<std::mutex> l(m_); cams_[c.camera_number] = c; }
    std::optional<LpSlamCameraConfiguration> getConfiguration(LpSlamCameraNumber n) {
        std::lock_guard<std::mutex> l(m_);
        auto it = cams_.find(n);
        if (it == cams_.end()) return std::nullopt;
        return it->second;
    }
private:
    std::mutex m_; std::map<LpSlamCameraNumber, LpSlamCameraConfiguration> cams_;
};

enum class ResultType { TrackedMarker, TrackedVehicle };
using MarkerId = uint32_t;

struct TrackerResult {
    ResultType type = ResultType::TrackedMarker;
    MarkerId id = 0;
    Position3 position;
    Orientation orientation;
    CompositeTimestamp timestamp;
};

void logMessage(LpSlamLogLevel level, const std::string& msg);     // slam_manager.cpp

class TrackerBase {
public:
    virtual ~TrackerBase() = default;
    typedef std::vector<TrackerResult> ProcessImageResult;

    virtual ProcessImageResult processImage(CameraQueueEntry& cam, std::optional<GlobalStateInTime> navResultOdom = std::nullopt,
                                            std::optional<GlobalStateInTime> navResultMap = std::nullopt,
                                            std::vector<SensorQueueEntry> const& sensorValues = {}) = 0;
    // The frame that will be processed after the one handed to the next processImage call (nullptr: none is queued yet).  A tracker
    // may start that frame's device-side front end once its own has finished, so that it runs beside the tracking of the current
    // frame; the pointer is valid during that processImage call only.
    virtual void setNextFrame(CameraQueueEntry const*) {}
    virtual void addRequestNavTransformationCallback(RequestNavTransformationCallback_t, void*) {}
    virtual std::optional<unsigned long> mappingGetMapRawSize() { return std::nullopt; }
    virtual std::optional<LpMapInfo> mappingGetMapRaw(int8_t*, std::size_t) { return std::nullopt; }
    virtual void addLaserScan(GlobalStateInTime, float*, size_t, float, float, float, float, float, float) {}

    bool setConfig(std::string const& jsonConfig) {
        auto newConfig = m_config;
        try { newConfig.parse(jsonConfig); }
        catch (std::exception& ex) { logMessage(LpSlamLogLevel_Error, std::string("Cannot parse config due to error: ") + ex.what()); return false; }
        m_config = newConfig;
        OnConfigurationUpdate();
        return true;
    }
    virtual bool start(SensorQueue&) { return true; }
    virtual bool stop() { return true; }
    virtual void OnConfigurationUpdate() {}
    virtual std::string type() = 0;
    void setCameraRegistry(CameraRegistry* r) { m_camReg = r; }
    CameraRegistry* getCameraRegistry() { return m_camReg; }

protected:
    ConfigOptions& getConfigOptions() { return m_config; }

private:
    ConfigOptions m_config;
    CameraRegistry* m_camReg = nullptr;
};

class ProcessorBase {
public:
    virtual ~ProcessorBase() = default;
    virtual void processImage(CameraQueueEntry&) {}
    virtual void processSensorValuesAndResults(std::vector<SensorQueueEntry> const&, GlobalStateInTime const&) {}
    virtual std::string type() = 0;
    void setConfig(std::string const&) {}
};

}  // namespace LpSlam
