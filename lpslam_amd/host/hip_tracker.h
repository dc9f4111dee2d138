// hip_tracker.h -- tracker plugins "VSLAMStereo" / "VSLAMMono" backed by the MI355X hot path.
//
// Drop-in for the reference's OpenVSLAM adapters (src/Trackers/OpenVSLAMTrackerBase.{h,cpp},
// src/Trackers/OpenVSLAMStereoTracker.{h,cpp}, src/Trackers/OpenVSLAMTracker.{h,cpp}): same type() strings, same JSON
// configuration keys (OpenVSLAMTrackerBase.cpp:31-50), same pose conventions on the way out (createTrackerResult,
// :307-329).  Where the reference holds an openvslam::system, this class holds a lpslam_hip_ctx: per frame it runs the
// HIP front end (ORB on both eyes, stereo match, brute-force match against the previous frame), a motion-only pose
// optimisation, and on keyframes a local bundle adjustment over a sliding window (include/lpslam_hip.h).
#pragma once
#include "core.h"
#include "bow.h"
#include "../../include/lpslam_hip.h"

#include <array>
#include <condition_variable>
#include <memory>
#include <thread>
#include <unordered_map>

namespace LpSlam {

enum class TrackerState { NotInitialized, Initializing, Tracking, Lost };     // openvslam::tracker_state_t

class HipVslamTrackerBase : public TrackerBase {
public:
    HipVslamTrackerBase();
    ~HipVslamTrackerBase() override;
    void OnConfigurationUpdate() override;
    bool stop() override;

    // what the manager down-casts for (src/Manager/SlamManager.cpp:1316-1366)
    bool mappingSetMode(bool enableMapping) { m_enableMapping = enableMapping; return true; }
    bool mappingSetFilename(std::string const& filename) { m_mapFilename = filename; return true; }
    bool mappingExportCSV(std::string csv_filename);
    std::size_t mappingGetFeatures(LpSlamMapBoundary boundary, LpSlamFeatureEntry* entry, std::size_t entry_count, LpSlamMatrix9x9 transform);
    std::size_t mappingGetFeaturesCount(LpSlamMapBoundary boundary);
    std::string lastStatistics() { std::scoped_lock lock(m_slamLock); return m_lastStatistics; }      // "VSLAM statistics: key=value ..." of the last stop(), this tracker's own (the log is process-wide)
    LpSlamStatus getSlamStatus();

    TrackerState state() const { return m_state; }
    std::array<double, 16> currentCamPose();          // T_cw, row-major 4x4 (get_current_cam_pose)

protected:
    struct Pose { double q[4] = {1, 0, 0, 0}; double t[3] = {0, 0, 0}; };   // world -> camera
    struct FrameData {
        std::vector<lpslam_hip_keypoint> kpts;
        std::vector<uint8_t> desc;                    // 32 bytes per keypoint
        std::vector<float> x_right, depth;
        std::vector<int> landmark;                    // landmark id per keypoint or -1
        Pose pose;
        int slot = 0;
    };
    // The map ([UPSTREAM] data::map_database): every keyframe ever inserted with its keypoints, descriptors, stereo columns and
    // the landmark of every keypoint; every landmark with the list of its observations.  Covisibility ([UPSTREAM]
    // data::graph_node: keyframes weighted by the landmarks they share) is computed from the observation lists when it is needed.
    struct Keyframe {
        Pose pose;
        std::vector<lpslam_hip_keypoint> kpts; std::vector<uint8_t> desc; std::vector<float> x_right, depth;
        std::vector<int> landmark;                    // per keypoint: landmark id or -1
        int segment = 0;                              // map segment: a re-initialisation after a loss opens a new one
        bool erased = false;                          // culled as redundant: holds no keypoints and no observations any more (the index stays)
        bool desc_on_device = false;                  // lpslam_hip_desc_store_put succeeded: the loop-candidate search may name it in a batched comparison
        BowVector bow;                                // with a vocabulary: the keyframe's BoW vector and, per keypoint, the tree node it falls under
        std::vector<int32_t> node;
    };
    struct Landmark {
        double p[3]; 
        // what local-map tracking needs ([UPSTREAM] data::landmark): the descriptor, the viewing direction and the valid
        // distance range of the observation that created it (scale prediction: level = ceil(log(max_valid / d) / log s))
        uint8_t desc[32] = {0};
        double normal[3] = {0, 0, 1};
        double max_valid = 0, min_valid = 0;
        int ref_kf = -1;                              // the keyframe that created it (loop correction moves it with that keyframe)
        // [UPSTREAM] data::landmark num_observable_ / num_observed_ (both start at 1): how often tracking expected to see the landmark
        // and how often it did -- local_map_cleaner::remove_redundant_landmarks drops the ones seen less than 30 % of the time
        int n_observable = 1, n_observed = 1;
        std::vector<std::pair<int, int>> obs;         // (keyframe, keypoint) in insertion order
    };
    struct Statistics {                               // logged at stop() ("VSLAM statistics: ..."): what the tracker did, for logs and tests
        long frames = 0, motion_tracked = 0, bf_tracked = 0, local_map_joined = 0, keyframes = 0, fused_added = 0, fused_merged = 0;
        long culled_landmarks = 0, culled_keyframes = 0;
        long ba_failed = 0;                           // windows whose solve failed on the device (result dropped)
        long local_ba = 0, loops_closed = 0, loop_fused = 0, global_ba = 0, lost = 0, relocalised = 0, reinitialised = 0, nav_priors = 0, prefetched = 0;
        // where the frames' time went (seconds, summed): front end (upload, extraction, stereo, read-back), tracking against the
        // previous frame, local-map tracking, keyframe work on the tracking thread (insertion, fusion, loop search, BA set-up / wait)
        double t_front = 0, t_track = 0, t_local = 0, t_keyframe = 0, t_total = 0;
        double t_kf_wait = 0, t_kf_apply = 0, t_kf_insert = 0, t_kf_loop = 0, t_kf_prepare = 0, t_map_solve = 0;      // keyframe path: waiting for the previous window's solve, applying it, the new keyframe, the next window's set-up; the mapping thread's solve
        double t_prefetch_wait = 0, t_prefetch_busy = 0;       // this thread waiting for the prefetch thread at the end of a frame / that thread's own time per frame
        double t_dev_match = 0, t_dev_pose = 0, t_dev_upload = 0, t_dev_extract = 0, t_dev_get = 0;      // inside the above: device calls
    };

    bool startContext(bool stereo);
    void warmUpContext(bool stereo);                  // one frame's worth of every per-frame call on blank images: lazily created resources exist before the first frame
    ProcessImageResult trackFrame(CameraQueueEntry& cam, bool stereo, const std::optional<GlobalStateInTime>& navOdom);
    TrackerResult createTrackerResult(const Pose& pose_cw, TimeStamp timestamp) const;
    bool initializeMap(FrameData& f, const Pose& at);
    bool poseFromMatches(FrameData& cur, const std::vector<int>& cur_idx, const std::vector<int>& lm_ids, const Pose& init, int& n_inliers, int min_inliers = 10);
    bool predictedPose(Pose& init) const;
    bool navDelta(Pose& v) const;                     // motion of the navigation prior between the last two frames (world -> camera)
    static void movePose(const Pose& v, const Pose& from, Pose& to);
    bool trackWithMotionModel(FrameData& cur, int& n_inliers);
    bool trackAgainstPrevious(FrameData& cur, int& n_inliers);
    bool trackLocalMap(FrameData& cur, int& n_inliers);
    bool relocalise(FrameData& cur);
    // monocular: map initialisation from two views ([UPSTREAM] module::initializer + initialize::perspective) and new landmarks
    // by triangulation between consecutive keyframes ([UPSTREAM] mapping_module::create_new_landmarks, previous keyframe only)
    bool monoInitialize(FrameData& cur);
    void monoTriangulate(int prev_kf, Keyframe& kf, FrameData& f);
    bool detectAndCloseLoop(FrameData& cur, int c);
    void initLandmarkView(Landmark& lm, const Pose& pose, const lpslam_hip_keypoint& kp, const uint8_t* desc32) const;
    int insertKeyframe(FrameData& f);
    int resolve(int id) const;                        // follows the replacements of merged landmarks
    std::vector<int> covisible(int kf, int top_n, int min_weight) const;
    // [UPSTREAM] mapping_module::fuse_landmark_duplication / match::fuse: the given landmarks are projected into keyframe `c`
    // (whose keypoints sit in image slot `slot`); a match with a keypoint that has no landmark adds an observation, a match with
    // one that has another landmark merges the two (the one with more observations stays)
    void fuseInto(int c, int slot, const std::vector<int>& landmark_ids, FrameData& f, long& added, long& merged);
    void mergeLandmarks(int keep, int drop, FrameData* f);
    bool keyframeNeeded(int inliers) const;
    // Local bundle adjustment ([UPSTREAM] mapping_module -> optimize::local_bundle_adjuster): the new keyframe and its covisible
    // keyframes are free, every other keyframe that sees their landmarks is held fixed.  As in the reference it runs beside
    // tracking: the problem is copied when a keyframe is inserted, a mapping thread solves it on the GPU (the BA object has its own
    // stream) while the tracking thread takes the next frames, and the result enters the map right before the next keyframe is
    // inserted -- a fixed point of the frame sequence, so a run is reproducible.
    struct MappingJob {
        std::vector<double> poses, pts;
        std::vector<uint8_t> fixed, outlier;
        std::vector<lpslam_hip_ba_obs> obs;
        std::vector<std::pair<int, int>> origin;      // (keyframe, keypoint) of every BA observation
        std::vector<int> ids;                         // landmark id of every BA point
        std::vector<int> kfs;                         // keyframe of every BA pose
        bool global = false;                          // loop-time global BA: plain robust iterations, no outlier pass
        int keyframe = -1;                            // local BA: the keyframe it was started for (redundant-keyframe culling follows it)
        bool solved = false;
    };
    std::unique_ptr<MappingJob> prepareMapping(int c);
    std::unique_ptr<MappingJob> prepareBundle(const std::vector<int>& free_kfs, const std::vector<int>& fixed_kfs);
    void solveMapping(MappingJob& job) const;
    void applyMapping(const MappingJob& job);
    void startMapping(int c);                         // prepare + solve on the mapping thread (or inline when asyncMapping is off)
    void finishMapping();                             // wait for the mapping thread and apply its result
    void logStatistics();
    std::string m_lastStatistics;                       // the line logStatistics() logged last (kept past stop(): lastStatistics())
    void storeDescriptors(int key, Keyframe& kf);     // keeps the keyframe's descriptors on the device for the batched loop-candidate search

    // configuration (names as in the reference tracker)
    bool m_useLiveView = false, m_useMapDb = true, m_forwardNavState = true, m_forwardImu = true, m_emitMap = false;
    bool m_enableMapping = true, m_waitForNavigation = false, m_forwardHighResNav = false, m_loopClosure = true;
    bool m_useOpenCL = false, m_useCUDA = false, m_relocWithNavigation = true, m_asyncMapping = true, m_prefetch = true;
    int m_mappingReserve = 0;                      // compute units of every XCD the front end leaves to the mapping thread's solves
    std::string m_configFromFile, m_cameraSetup = "monocular", m_vocabFile = "orb_vocab.dbow2", m_mapFilename = "map.db";
    int m_slamKeypoints = 1200, m_viewerFps = 10;
    double m_baselineDistThresh = 0.1, m_maxLaserAge = 1.0;
    // MI355X additions: the generated reference config hard-codes these
    int m_numLevels = 3, m_iniFastThr = 20, m_minFastThr = 7, m_device = 0, m_keyframeInterval = 6, m_localWindow = 10;
    double m_scaleFactor = 1.2;

    std::mutex m_slamLock;
    lpslam_hip_ctx* m_ctx = nullptr;
    LpSlamCameraConfiguration m_cam{};
    bool m_started = false, m_stereo = false, m_rectify = false;
    TrackerState m_state = TrackerState::NotInitialized;
    std::optional<TimeStamp> m_firstImageTimestamp;
    uint64_t m_imageTracked = 0;
    // Image slots: a ring of three slot pairs -- the frame being tracked, the previous one (brute-force fallback, scratch for
    // relocalisation / loop candidates once it is no longer needed) and the next one, whose front end is prefetched.
    static int slotOf(uint64_t frame_index) { return (int)(frame_index % 3) * 2; }
    static int previousSlot(int slot) { return ((slot / 2 + 2) % 3) * 2; }
    // map maintenance ([UPSTREAM] module::local_map_cleaner, run by the mapping module around every local BA)
    bool m_mapCulling = true;
    std::vector<int> m_freshLandmarks;                  // landmarks younger than three keyframes
    void cullLandmarks(int cur_kf);
    void cullKeyframes(int cur_kf);
    void eraseLandmark(int id);
    // bag-of-words place recognition ([UPSTREAM] data::bow_vocabulary / bow_database): loaded from vocabFile when that file exists
    lpslam_hip_vocab* m_vocab = nullptr;
    int m_bowLevelsUp = 4;                              // FeatureVector level: 4 levels above the leaves as upstream, less for shallow trees
    BowDatabase m_bowDb;
    void computeBow(Keyframe& kf) const;
    bool frameNodes(const FrameData& f, std::vector<int32_t>& node, BowVector* bow) const;
    struct Prefetched { bool valid = false; bool issued = false /* work may sit on the prefetch stream, valid or not */; const uint8_t* data = nullptr; TimeStamp timestamp{}; int slot = 0; bool stereo = false; } m_prefetched;
    void prefetchFrame(CameraQueueEntry const& cam, bool stereo);      // m_slamLock held
    CameraQueueEntry const* m_nextFrame = nullptr;
    void setNextFrame(CameraQueueEntry const* next) override { m_nextFrame = next; }
    bool frontEnd(CameraQueueEntry const& cam, bool stereo, int slot);
    double m_lastFrameSeconds = 0;
    int m_maxKp = 0;
    float m_scales[LPSLAM_HIP_MAX_LEVELS] = {0};
    FrameData m_prev;
    bool m_havePrev = false;
    Pose m_velocity;                                  // last inter-frame motion (constant-velocity prediction)
    bool m_haveVelocity = false;
    int m_framesSinceKeyframe = 0;
    std::vector<Keyframe> m_kfs;                      // the map's keyframes, index = id
    std::vector<std::pair<std::vector<int>, int>> m_loopSets;      // (keyframe set, continuity) of the loop candidates detected at the previous keyframe ([UPSTREAM] cont_detected_keyfrm_sets_)
    std::unordered_map<int, Landmark> m_landmarks;
    // Landmark ids are dense (0 .. m_nextLandmarkId): an index of the map's nodes by id (a node stays where it is until it is erased) and
    // scratch marks by id replace the hash lookups of the per-frame and per-keyframe loops (covisibility, local landmarks, window assembly:
    // 0.10 of a 0.50 ms frame were hash operations)
    std::vector<Landmark*> m_lmIndex;
    Landmark* lm(int id) const { return id >= 0 && (size_t)id < m_lmIndex.size() ? m_lmIndex[(size_t)id] : nullptr; }
    void indexLandmark(int id) { if ((size_t)id >= m_lmIndex.size()) m_lmIndex.resize((size_t)id + 256, nullptr); m_lmIndex[(size_t)id] = &m_landmarks.find(id)->second; }
    void unindexLandmark(int id) { if (id >= 0 && (size_t)id < m_lmIndex.size()) m_lmIndex[(size_t)id] = nullptr; }
    struct IdMarks {                                  // value per id, valid for the current round only (begin() starts a round in O(1))
        std::vector<uint32_t> round; std::vector<int32_t> val; uint32_t cur = 0;
        void begin(size_t n) { if (round.size() < n) { round.resize(n + 256, 0); val.resize(n + 256, 0); } if (++cur == 0) { std::fill(round.begin(), round.end(), 0u); cur = 1; } }
        bool has(int id) const { return id >= 0 && (size_t)id < round.size() && round[(size_t)id] == cur; }
        int32_t& at(int id) { if (round[(size_t)id] != cur) { round[(size_t)id] = cur; val[(size_t)id] = 0; } return val[(size_t)id]; }
        int32_t get(int id) const { return has(id) ? val[(size_t)id] : 0; }
    };
    mutable IdMarks m_marksA, m_marksB, m_marksC, m_marksKf;      // tracking thread only
    std::unordered_map<int, int> m_replaced;          // merged landmark -> the one that took its observations
    int m_nextLandmarkId = 0;
    int m_refKf = -1;                                 // reference keyframe of tracking (the last one inserted)
    int m_refTracked = 0;                             // landmarks the reference keyframe held when it was inserted
    int m_segment = 0, m_segmentStart = 0;            // current map segment and its first keyframe
    // loss and relocalisation (the reference forwards time_to_relocalize = 3.0 s, src/Trackers/OpenVSLAMTrackerBase.cpp:210-211)
    Pose m_lastGoodPose;
    TimeStamp m_lostSince{};
    double m_timeToRelocalize = 3.0;
    // Initializer.num_min_triangulated_pts / parallax_deg_threshold as the reference sets them (src/Trackers/OpenVSLAMTrackerBase.cpp:181-182)
    int m_initMinTriangulated = 40;
    double m_initParallaxDeg = 0.2;
    // navigation prior (src/Trackers/OpenVSLAMStereoTracker.cpp:70-179): camera pose of the odometry, optical axes, world -> camera
    std::optional<Pose> m_navPrev, m_navCur;
    Statistics m_stats;
    FrameData m_monoRef;                              // monocular initialisation: the reference frame ...
    bool m_haveMonoRef = false;
    std::vector<float> m_monoPrevMatched;             // ... and where each of its keypoints was last matched (x, y)
    // the mapping thread (one per tracker, started with the context): a one-slot mailbox each way
    // the prefetch helper: ONE persistent thread (a std::async per frame spawned and joined a thread every frame)
    std::thread m_pfThread;
    std::mutex m_pfMutex;
    std::condition_variable m_pfCv;
    const CameraQueueEntry* m_pfJob = nullptr; bool m_pfStereo = false, m_pfBusy = false, m_pfQuit = false;
    void prefetchLoop();
    void prefetchSubmit(const CameraQueueEntry* next, bool stereo);
    void prefetchWait();
    void stopPrefetchThread();
    std::thread m_mapThread;
    std::mutex m_mapMutex;
    std::condition_variable m_mapCv;
    std::unique_ptr<MappingJob> m_mapIn, m_mapOut;
    bool m_mapBusy = false, m_mapQuit = false;
    void mappingLoop();
    void stopMappingThread();
};

class HipStereoTracker : public HipVslamTrackerBase {
public:
    ProcessImageResult processImage(CameraQueueEntry& cam, std::optional<GlobalStateInTime> navResultOdom = std::nullopt,
                                    std::optional<GlobalStateInTime> navResultMap = std::nullopt,
                                    std::vector<SensorQueueEntry> const& sensorValues = {}) override;
    bool start(SensorQueue&) override;
    std::string type() override { return "VSLAMStereo"; }       // src/Trackers/OpenVSLAMStereoTracker.h:35-39
};

class HipMonoTracker : public HipVslamTrackerBase {
public:
    ProcessImageResult processImage(CameraQueueEntry& cam, std::optional<GlobalStateInTime> navResultOdom = std::nullopt,
                                    std::optional<GlobalStateInTime> navResultMap = std::nullopt,
                                    std::vector<SensorQueueEntry> const& sensorValues = {}) override;
    bool start(SensorQueue&) override;
    std::string type() override { return "VSLAMMono"; }
};

}  // namespace LpSlam
