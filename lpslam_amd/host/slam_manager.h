// slam_manager.h -- runtime behind LpSlamManager: plugin factories, configuration file, frame ingest, the worker and
// notify threads (mirror of /root/reference/src/Manager/SlamManager.{h,cpp}; call stack in SURVEY.md section 3.1).
#pragma once
#include "core.h"
#include "hip_tracker.h"
#include "replay.h"

#include <atomic>
#include <thread>

namespace LpSlam {

class SlamManager {
public:
    SlamManager();
    ~SlamManager();

    void logToFile(std::string const& filename);
    void setLogLevel(LpSlamLogLevel l);

    void addOnReconstructionCallback(OnReconstructionCallback_t cb, void* ud) { m_onReconstruction = cb; m_onReconstructionData = ud; }
    void addRequestNavDataCallback(RequestNavDataCallback_t cb, void* ud) { m_requestNavData = cb; m_requestNavDataData = ud; }
    void addRequestNavTransformation(RequestNavTransformationCallback_t cb, void* ud) { m_requestNavTransformation = cb; m_requestNavTransformationData = ud; }
    void addOnImageCallback(OnImageCallback_t cb, void* ud) { m_onImage = cb; m_onImageData = ud; }

    bool addImageFromBuffer(uint32_t cameraNumber, LpSlamTimestamp timestamp, uint8_t* buffer, LpSlamImageDescription desc);
    bool addStereoImageFromBuffer(uint32_t cameraNumber, LpSlamTimestamp timestamp, uint8_t* left, uint8_t* right, LpSlamImageDescription desc);

    void setCameraConfiguration(LpSlamCameraConfiguration const& c) { m_camRegistry.setConfiguration(c); }
    bool readConfigurationFile(std::string const& filename);
    bool addSource(std::string const& name, std::string const& jsonConfig);
    bool loadReplayItems(std::string const& filename);      // SlamManager::loadReplayItems (reference SlamManager.cpp:503-505)
    bool addTracker(std::string const& name, std::string const& jsonConfig);
    bool addProcessor(std::string const& name, std::string const& jsonConfig);

    void start();
    void stop();
    LpSlamStatus getSlamStatus();

    std::size_t mappingGetFeatures(LpSlamMapBoundary b, LpSlamFeatureEntry* e, std::size_t n, LpSlamMatrix9x9 t);
    std::size_t mappingGetFeaturesCount(LpSlamMapBoundary b);
    std::string trackerStatistics();                    // the VSLAM tracker's last statistics line (empty before its first stop)
    bool mappingSetMode(bool enable);
    bool mappingSetFilename(std::string const& f);
    bool mappingExportCSV(std::string const& f);

    // test hooks
    size_t trackerCount() const { return m_trackers.size(); }
    CameraRegistry& cameraRegistry() { return m_camRegistry; }
    uint64_t framesProcessed() const { return m_framesProcessed.load(); }
    uint64_t framesSkipped() const { return m_framesSkipped.load(); }
    uint64_t imagesSent() const { return m_imagesSent.load(); }

private:
    void streamMoreReplayItems();
    bool workerStep();
    bool notifyStep();
    bool imageCallbackStep();

    CameraRegistry m_camRegistry;
    CameraQueue m_camQueue;
    SensorQueue m_sensorQueue;
    ResultQueue m_resultQueue;
    CameraQueue m_imageCallbackQueue;    // frames on their way to OnImageCallback_t (reference SlamManager.h:183)
    std::vector<std::unique_ptr<TrackerBase>> m_trackers;
    std::vector<std::unique_ptr<ProcessorBase>> m_processors;
    HipVslamTrackerBase* m_vslamTracker = nullptr;
    std::thread m_worker, m_notifyWorker, m_imageCallbackWorker;
    bool m_pushToImageCallbackQueue = false;      // fixed at start(), as the reference copies its thread parameters there
    bool m_running = false;
    bool m_requireOdometry = true;       // reference behaviour: frames without odometry are skipped (SlamManager.cpp:193-196)
    int m_thread_num = -1;
    std::atomic<uint64_t> m_framesProcessed{0}, m_framesSkipped{0}, m_imagesSent{0};
    std::atomic<double> m_currentFps{0.0};
    double m_secondsInWorker = 0, m_secondsInTrackers = 0;      // worker thread only (logged by stop()): a frame from the moment it is taken / inside processImage
    std::atomic<bool> m_stopRequested{false};
    std::optional<CameraQueueEntry> m_lookahead;       // worker thread only: the frame after the one being processed
    ReplayReader m_replay;
    std::mutex m_replayMutex;
    size_t m_replayChunk = 500;          // ReplayEngine.h:53
    std::optional<std::chrono::steady_clock::time_point> m_lastFrame;

    OnReconstructionCallback_t m_onReconstruction = nullptr; void* m_onReconstructionData = nullptr;
    RequestNavDataCallback_t m_requestNavData = nullptr; void* m_requestNavDataData = nullptr;
    RequestNavTransformationCallback_t m_requestNavTransformation = nullptr; void* m_requestNavTransformationData = nullptr;
    OnImageCallback_t m_onImage = nullptr; void* m_onImageData = nullptr;
};

}  // namespace LpSlam
