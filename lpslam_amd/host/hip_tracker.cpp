// hip_tracker.cpp -- see hip_tracker.h.  Host-side tracking glue around the HIP C ABI (the arithmetic runs on the GPU).
#include "hip_tracker.h"
#include "rectify.h"
#include "two_view.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <limits>

#include <atomic>
static std::atomic<long> g_motion_tracked{0};      // process-wide count of frames tracked by the motion model (introspection for the tests)
extern "C" __attribute__((visibility("default"))) long lpslam_debug_motion_tracked(void) { return g_motion_tracked.load(); }
static std::atomic<long> g_loops_closed{0};       // loops closed by any tracker of this process
extern "C" __attribute__((visibility("default"))) long lpslam_debug_loops_closed(void) { return g_loops_closed.load(); }
static std::atomic<long> g_local_map_joined{0};   // landmarks local-map tracking brought back into frames
extern "C" __attribute__((visibility("default"))) long lpslam_debug_local_map_joined(void) { return g_local_map_joined.load(); }

namespace LpSlam {

namespace {

struct Mat3 { double m[9]; };

Mat3 quatToRot(const double* q)
{
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    return {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
             2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
             2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
}

void quatMul(const double* a, const double* b, double* o)
{
    const double r[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                         a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
    const double n = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    for (int i = 0; i < 4; ++i) o[i] = r[i] / n;
}

// Eigen::Quaterniond(R) for a rotation matrix
void rotToQuat(const Mat3& R, double* q)
{
    const double* m = R.m;
    const double tr = m[0] + m[4] + m[8];
    if (tr > 0) {
        const double s = std::sqrt(tr + 1.0) * 2;
        q[0] = 0.25 * s; q[1] = (m[7] - m[5]) / s; q[2] = (m[2] - m[6]) / s; q[3] = (m[3] - m[1]) / s;
    } else if (m[0] > m[4] && m[0] > m[8]) {
        const double s = std::sqrt(1.0 + m[0] - m[4] - m[8]) * 2;
        q[0] = (m[7] - m[5]) / s; q[1] = 0.25 * s; q[2] = (m[1] + m[3]) / s; q[3] = (m[2] + m[6]) / s;
    } else if (m[4] > m[8]) {
        const double s = std::sqrt(1.0 + m[4] - m[0] - m[8]) * 2;
        q[0] = (m[2] - m[6]) / s; q[1] = (m[1] + m[3]) / s; q[2] = 0.25 * s; q[3] = (m[5] + m[7]) / s;
    } else {
        const double s = std::sqrt(1.0 + m[8] - m[0] - m[4]) * 2;
        q[0] = (m[3] - m[1]) / s; q[1] = (m[2] + m[6]) / s; q[2] = (m[5] + m[7]) / s; q[3] = 0.25 * s;
    }
}

}  // namespace

HipVslamTrackerBase::HipVslamTrackerBase()
{
    auto& o = getConfigOptions();
    // keys and defaults of the reference tracker (src/Trackers/OpenVSLAMTrackerBase.cpp:31-50)
    o.optional("liveView", false); o.optional("useMapDb", true); o.optional("configFromFile", "");
    o.optional("cameraSetup", "monocular"); o.optional("slamKeypoints", 1200); o.optional("vocabFile", m_vocabFile);
    o.optional("forwardNavState", true); o.optional("forwardImu", true); o.optional("emitMap", false);
    o.optional("enableMapping", true); o.optional("waitForNavigation", false); o.optional("viewerFps", 10);
    o.optional("forwardHighResNav", false); o.optional("loopClosure", true); o.optional("useOpenCL", false);
    o.optional("useCUDA", false); o.optional("relocWithNavigation", true); o.optional("baselineDistThresh", 0.1);
    o.optional("mapFilename", "map.db"); o.optional("maxLaserAge", 1.0);
    // runtime ORB parameters the reference hard-codes in its generated YAML (:193-198), plus device selection
    o.optional("numLevels", 3); o.optional("scaleFactor", 1.2); o.optional("iniFastThr", 20); o.optional("minFastThr", 7);
    o.optional("device", 0); o.optional("keyframeInterval", 6); o.optional("localWindow", 10); o.optional("asyncMapping", true);
}

HipVslamTrackerBase::~HipVslamTrackerBase() { stop(); }

void HipVslamTrackerBase::OnConfigurationUpdate()
{
    auto& o = getConfigOptions();
    m_useLiveView = o.getBool("liveView"); m_useMapDb = o.getBool("useMapDb"); m_configFromFile = o.getString("configFromFile");
    m_slamKeypoints = o.getInteger("slamKeypoints"); m_cameraSetup = o.getString("cameraSetup"); m_vocabFile = o.getString("vocabFile");
    m_forwardNavState = o.getBool("forwardNavState"); m_forwardImu = o.getBool("forwardImu"); m_emitMap = o.getBool("emitMap");
    m_enableMapping = o.getBool("enableMapping"); m_mapFilename = o.getString("mapFilename");
    m_waitForNavigation = o.getBool("waitForNavigation"); m_viewerFps = o.getInteger("viewerFps");
    m_forwardHighResNav = o.getBool("forwardHighResNav"); m_loopClosure = o.getBool("loopClosure");
    m_useOpenCL = o.getBool("useOpenCL"); m_useCUDA = o.getBool("useCUDA"); m_relocWithNavigation = o.getBool("relocWithNavigation");
    m_baselineDistThresh = o.getDouble("baselineDistThresh"); m_maxLaserAge = o.getDouble("maxLaserAge");
    m_numLevels = o.getInteger("numLevels"); m_scaleFactor = o.getDouble("scaleFactor");
    m_iniFastThr = o.getInteger("iniFastThr"); m_minFastThr = o.getInteger("minFastThr"); m_device = o.getInteger("device");
    m_keyframeInterval = std::max(1, o.getInteger("keyframeInterval")); m_localWindow = std::max(2, o.getInteger("localWindow"));
    m_asyncMapping = o.getBool("asyncMapping");
}

bool HipVslamTrackerBase::startContext(bool stereo)
{
    std::scoped_lock lock(m_slamLock);
    if (m_ctx) return true;
    if (!m_configFromFile.empty()) {
        logMessage(LpSlamLogLevel_Error, "configFromFile (raw OpenVSLAM YAML) is not supported; use the tracker's JSON keys");
        return false;
    }
    CameraRegistry* reg = getCameraRegistry();
    if (!reg) { logMessage(LpSlamLogLevel_Error, "Cannot process image without camera registry"); return false; }
    auto left = reg->getConfiguration(0);
    if (!left) { logMessage(LpSlamLogLevel_Error, "Cannot load camera configuration for camera with number 0"); return false; }
    if (stereo && !reg->getConfiguration(1)) { logMessage(LpSlamLogLevel_Error, "Cannot load camera configuration for right camera with number 1"); return false; }
    m_cam = *left;
    // ImageProcessing::Undistort (reference: src/Utils/ImageProcessing.h:134-250): the maps are built once from the camera
    // pair and every frame is remapped -- here on the device (lpslam_hip_upload_raw_image).  The reference does this for the
    // stereo tracker only (src/Trackers/OpenVSLAMStereoTracker.cpp:198-213); the monocular one feeds frames as they come.
    m_rectify = stereo && m_cam.distortion_function != LpSlamCameraDistortionFunction_NoDistortion;
    RectifyMaps maps[2];
    if (m_rectify) {
        auto right = reg->getConfiguration(1);
        std::string err;
        for (int eye = 0; eye < 2; ++eye)
            if (!build_rectify_maps(*left, *right, eye == 0, maps[eye], &err)) {
                logMessage(LpSlamLogLevel_Error, "Cannot build the rectification maps: " + err);
                return false;
            }
    }
    if (m_cam.resolution_x <= 0 || m_cam.resolution_y <= 0 || !(m_cam.f_x > 0) || (stereo && !(m_cam.focal_x_baseline > 0))) {
        logMessage(LpSlamLogLevel_Error, "Camera configuration incomplete (resolution, focal length, focal_x_baseline)");
        return false;
    }
    lpslam_hip_frontend_config cfg{};
    cfg.width = m_cam.resolution_x; cfg.height = m_cam.resolution_y; cfg.max_keypoints = m_slamKeypoints;
    cfg.scale_factor = (float)m_scaleFactor; cfg.num_levels = m_numLevels; cfg.ini_fast_threshold = m_iniFastThr;
    cfg.min_fast_threshold = m_minFastThr; cfg.max_images = 4; cfg.device = m_device;
    if (lpslam_hip_create(&cfg, &m_ctx) != LPSLAM_HIP_OK) {
        logMessage(LpSlamLogLevel_Error, std::string("Cannot create the HIP context: ") + lpslam_hip_last_error());
        m_ctx = nullptr;
        return false;
    }
    for (int eye = 0; m_rectify && eye < 2; ++eye)
        if (lpslam_hip_set_rectify_map(m_ctx, eye, maps[eye].map_x.data(), maps[eye].map_y.data()) != LPSLAM_HIP_OK) {
            logMessage(LpSlamLogLevel_Error, std::string("Cannot upload the rectification maps: ") + lpslam_hip_last_error());
            lpslam_hip_destroy(m_ctx); m_ctx = nullptr;
            return false;
        }
    m_maxKp = lpslam_hip_max_keypoints_per_image(m_ctx);
    m_stereo = stereo;
    m_state = TrackerState::NotInitialized;
    m_started = true;
    return true;
}

bool HipVslamTrackerBase::stop()
{
    std::scoped_lock lock(m_slamLock);
    stopMappingThread();                               // the mapping thread uses the context
    if (m_ctx) { lpslam_hip_destroy(m_ctx); m_ctx = nullptr; }
    m_started = false;
    return true;
}

std::array<double, 16> HipVslamTrackerBase::currentCamPose()
{
    std::scoped_lock lock(m_slamLock);
    const Mat3 R = quatToRot(m_prev.pose.q);
    return {R.m[0], R.m[1], R.m[2], m_prev.pose.t[0], R.m[3], R.m[4], R.m[5], m_prev.pose.t[1],
            R.m[6], R.m[7], R.m[8], m_prev.pose.t[2], 0, 0, 0, 1};
}

// T_cw -> camera centre, optical axes (x right, y down, z forward) -> lpslam axes: p_lp = (-y, x, z), q_lp = (w, -y, x, z)
// (src/Trackers/OpenVSLAMTrackerBase.cpp:307-329)
TrackerResult HipVslamTrackerBase::createTrackerResult(const Pose& p, TimeStamp timestamp) const
{
    const Mat3 R = quatToRot(p.q);
    const double cx = -(R.m[0] * p.t[0] + R.m[3] * p.t[1] + R.m[6] * p.t[2]);
    const double cy = -(R.m[1] * p.t[0] + R.m[4] * p.t[1] + R.m[7] * p.t[2]);
    const double cz = -(R.m[2] * p.t[0] + R.m[5] * p.t[1] + R.m[8] * p.t[2]);
    double q[4];
    rotToQuat(R, q);
    TrackerResult t;
    t.type = ResultType::TrackedVehicle;
    t.id = 0;
    t.position.value = {-cy, cx, cz};
    t.orientation.value = {q[0], -q[2], q[1], q[3]};
    t.timestamp.system_time = timestamp;
    return t;
}

bool HipVslamTrackerBase::initializeMap(FrameData& f)
{
    // stereo initialisation: every keypoint with a valid depth becomes a landmark once at least
    // Initializer.num_min_triangulated_pts = 40 exist (src/Trackers/OpenVSLAMTrackerBase.cpp:181)
    int n = 0;
    for (size_t i = 0; i < f.kpts.size(); ++i) if (f.depth[i] > 0) ++n;
    if (n < 40) return false;
    { std::unique_lock<std::mutex> lk(m_mapMutex); m_mapCv.wait(lk, [this] { return !m_mapBusy; }); m_mapOut.reset(); }   // a solve of the map that is being dropped
    m_landmarks.clear(); m_keyframes.clear(); m_archive.clear(); m_nextLandmarkId = 0;
    f.pose = Pose();
    insertKeyframe(f);
    return true;
}

// descriptor, viewing direction and valid distance range of the observation that creates a landmark ([UPSTREAM] data::landmark::
// update_normal_and_depth / compute_descriptor, reference observation only)
void HipVslamTrackerBase::initLandmarkView(Landmark& lm, const Pose& pose, const lpslam_hip_keypoint& kp, const uint8_t* desc32) const
{
    float scales[LPSLAM_HIP_MAX_LEVELS];
    lpslam_hip_level_info(m_ctx, nullptr, nullptr, nullptr, nullptr, scales);
    const Mat3 R = quatToRot(pose.q);
    // camera centre C = -R^T t
    const double C[3] = {-(R.m[0] * pose.t[0] + R.m[3] * pose.t[1] + R.m[6] * pose.t[2]), -(R.m[1] * pose.t[0] + R.m[4] * pose.t[1] + R.m[7] * pose.t[2]),
                         -(R.m[2] * pose.t[0] + R.m[5] * pose.t[1] + R.m[8] * pose.t[2])};
    const double ray[3] = {lm.p[0] - C[0], lm.p[1] - C[1], lm.p[2] - C[2]};
    const double dist = std::sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
    for (int a = 0; a < 3; ++a) lm.normal[a] = dist > 0 ? ray[a] / dist : 0.0;
    const int lvl = std::min(std::max(kp.octave, 0), m_numLevels - 1);
    lm.max_valid = dist * scales[lvl];
    lm.min_valid = lm.max_valid / scales[std::max(m_numLevels - 1, 0)];
    std::copy(desc32, desc32 + 32, lm.desc);
}

void HipVslamTrackerBase::insertKeyframe(FrameData& f)
{
    const double baseline = m_cam.focal_x_baseline / m_cam.f_x;
    const double depth_thr = 40.0 * baseline;
    const Mat3 R = quatToRot(f.pose.q);
    float scales[LPSLAM_HIP_MAX_LEVELS];
    lpslam_hip_level_info(m_ctx, nullptr, nullptr, nullptr, nullptr, scales);
    Keyframe kf;
    kf.pose = f.pose;
    // new landmarks: unmatched keypoints closer than depth_threshold (40 baselines, OpenVSLAMTrackerBase.cpp:200); when fewer
    // than 100 landmarks would result, the closest ones beyond the threshold are taken too; the first keyframe takes all
    std::vector<std::pair<float, int>> by_depth;
    for (size_t i = 0; i < f.kpts.size(); ++i) if (f.landmark[i] < 0 && f.depth[i] > 0) by_depth.emplace_back(f.depth[i], (int)i);
    std::sort(by_depth.begin(), by_depth.end());
    std::vector<char> create(f.kpts.size(), 0);
    int tracked = 0;
    for (size_t i = 0; i < f.kpts.size(); ++i) tracked += f.landmark[i] >= 0;
    for (size_t r = 0; r < by_depth.size(); ++r)
        if (m_keyframes.empty() || by_depth[r].first < depth_thr || tracked + (int)r < 100) create[by_depth[r].second] = 1;
    for (size_t i = 0; i < f.kpts.size(); ++i) {
        int id = f.landmark[i];
        if (id < 0 && create[i]) {
            // back-project into the world: X_w = R^T (X_c - t)
            const double z = f.depth[i];
            const double xc = (f.kpts[i].x - m_cam.c_x) * z / m_cam.f_x, yc = (f.kpts[i].y - m_cam.c_y) * z / m_cam.f_y;
            const double d[3] = {xc - f.pose.t[0], yc - f.pose.t[1], z - f.pose.t[2]};
            Landmark lm;
            lm.p[0] = R.m[0] * d[0] + R.m[3] * d[1] + R.m[6] * d[2];
            lm.p[1] = R.m[1] * d[0] + R.m[4] * d[1] + R.m[7] * d[2];
            lm.p[2] = R.m[2] * d[0] + R.m[5] * d[1] + R.m[8] * d[2];
            initLandmarkView(lm, f.pose, f.kpts[i], f.desc.data() + 32 * i);
            lm.ref_kf = (long)m_archive.size();
            id = m_nextLandmarkId++;
            m_landmarks[id] = lm;
            f.landmark[i] = id;
        }
        if (id >= 0) {
            const double s = scales[f.kpts[i].octave];
            kf.obs.push_back({id, f.kpts[i].x, f.kpts[i].y, f.x_right[i] >= 0 ? (double)f.x_right[i] : -1.0, 1.0 / (s * s)});
            m_landmarks[id].n_obs++;
        }
    }
    if (!m_stereo) {
        if (!m_keyframes.empty()) monoTriangulate(m_keyframes.back(), kf, f);
        kf.kpts = f.kpts; kf.desc = f.desc; kf.landmark = f.landmark;
    }
    if (m_stereo && m_loopClosure) archiveKeyframe(kf, f);
    m_keyframes.push_back(std::move(kf));
    ++m_keyframeCount;
    while ((int)m_keyframes.size() > m_localWindow) {
        for (auto& o : m_keyframes.front().obs) {
            auto it = m_landmarks.find(o.landmark);
            if (it != m_landmarks.end() && --it->second.n_obs <= 0) m_landmarks.erase(it);
        }
        m_keyframes.pop_front();
    }
    m_framesSinceKeyframe = 0;
}

// motion-only pose optimisation of `cur` against the landmarks seen in the previous frame
// motion-only pose optimisation ([UPSTREAM] optimize::pose_optimizer) over keypoint <-> landmark associations
bool HipVslamTrackerBase::poseFromMatches(FrameData& cur, const std::vector<int>& cur_idx, const std::vector<int>& lm_ids, const Pose& init, int& n_inliers)
{
    n_inliers = 0;
    float scales[LPSLAM_HIP_MAX_LEVELS];
    lpslam_hip_level_info(m_ctx, nullptr, nullptr, nullptr, nullptr, scales);
    std::vector<double> pts;
    std::vector<lpslam_hip_ba_obs> obs;
    std::vector<int> kept_idx, kept_lm;
    for (size_t k = 0; k < cur_idx.size(); ++k) {
        auto it = m_landmarks.find(lm_ids[k]);
        if (it == m_landmarks.end()) continue;
        const int i = cur_idx[k];
        const double s = scales[cur.kpts[i].octave];
        lpslam_hip_ba_obs o{};
        o.pose = 0; o.point = (int32_t)kept_idx.size();
        o.u = cur.kpts[i].x; o.v = cur.kpts[i].y; o.ur = cur.x_right[i] >= 0 ? (double)cur.x_right[i] : -1.0; o.inv_sigma2 = 1.0 / (s * s);
        obs.push_back(o);
        pts.insert(pts.end(), it->second.p, it->second.p + 3);
        kept_idx.push_back(i);
        kept_lm.push_back(lm_ids[k]);
    }
    if (obs.size() < 10) return false;
    double pose7[7] = {init.q[0], init.q[1], init.q[2], init.q[3], init.t[0], init.t[1], init.t[2]};
    lpslam_hip_ba_camera cam{m_cam.f_x, m_cam.f_y, m_cam.c_x, m_cam.c_y, m_cam.focal_x_baseline, std::sqrt(5.991), std::sqrt(7.815)};
    std::vector<uint8_t> outlier(obs.size());
    int32_t inl = 0;
    // one launch: the whole 4 x 10 iteration flow runs in one workgroup on the device
    if (lpslam_hip_pose_optimize(m_ctx, pose7, pts.data(), (int32_t)kept_idx.size(), obs.data(), (int32_t)obs.size(), &cam, outlier.data(), &inl) != LPSLAM_HIP_OK) return false;
    n_inliers = inl;
    if (inl < 10) return false;
    for (int k = 0; k < 4; ++k) cur.pose.q[k] = pose7[k];
    for (int k = 0; k < 3; ++k) cur.pose.t[k] = pose7[4 + k];
    std::fill(cur.landmark.begin(), cur.landmark.end(), -1);
    for (size_t k = 0; k < kept_idx.size(); ++k) cur.landmark[kept_idx[k]] = outlier[k] ? -1 : kept_lm[k];   // inliers keep their landmark
    return true;
}

// [UPSTREAM] frame_tracker::motion_based_track: pose predicted with the constant-velocity model, the last frame's landmarks are
// projected into the current frame and matched inside a window around the prediction (match::projection::
// match_current_and_last_frames: margin 10 px x scale factor for stereo, doubled once if fewer than 20 matches), matches
// with an inconsistent keypoint rotation are dropped (angle_checker), then the motion-only pose optimiser runs.
bool HipVslamTrackerBase::trackWithMotionModel(FrameData& cur, int& n_inliers)
{
    n_inliers = 0;
    if (!m_haveVelocity) return false;
    Pose init;
    quatMul(m_velocity.q, m_prev.pose.q, init.q);
    const Mat3 Rv = quatToRot(m_velocity.q);
    for (int r = 0; r < 3; ++r) init.t[r] = Rv.m[r * 3] * m_prev.pose.t[0] + Rv.m[r * 3 + 1] * m_prev.pose.t[1] + Rv.m[r * 3 + 2] * m_prev.pose.t[2] + m_velocity.t[r];
    const Mat3 R = quatToRot(init.q);
    float scales[LPSLAM_HIP_MAX_LEVELS];
    const int32_t n_levels = m_numLevels;               // (the first array argument of level_info is the widths, not a count)
    lpslam_hip_level_info(m_ctx, nullptr, nullptr, nullptr, nullptr, scales);
    std::vector<lpslam_hip_proj_query> q;
    std::vector<uint8_t> qd;
    std::vector<float> q_angle;
    std::vector<int> q_lm;
    for (size_t i = 0; i < m_prev.kpts.size(); ++i) {
        const int id = m_prev.landmark[i];
        if (id < 0) continue;
        auto it = m_landmarks.find(id);
        if (it == m_landmarks.end()) continue;
        const double* X = it->second.p;
        const double pc[3] = {R.m[0] * X[0] + R.m[1] * X[1] + R.m[2] * X[2] + init.t[0], R.m[3] * X[0] + R.m[4] * X[1] + R.m[5] * X[2] + init.t[1],
                              R.m[6] * X[0] + R.m[7] * X[1] + R.m[8] * X[2] + init.t[2]};
        if (!(pc[2] > 0)) continue;
        const double u = m_cam.f_x * pc[0] / pc[2] + m_cam.c_x, v = m_cam.f_y * pc[1] / pc[2] + m_cam.c_y;
        if (u < 0 || v < 0 || u >= m_cam.resolution_x || v >= m_cam.resolution_y) continue;
        const int lvl = m_prev.kpts[i].octave;
        lpslam_hip_proj_query e{};
        e.x = (float)u; e.y = (float)v; e.x_right = m_stereo ? (float)(u - m_cam.focal_x_baseline / pc[2]) : -1.0f;
        e.radius = (m_stereo ? 10.0f : 20.0f) * scales[lvl];       // match_current_and_last_frames: margin 10 (stereo) / 20 (monocular)
        e.min_level = std::max(0, lvl - 1); e.max_level = std::min(n_levels - 1, lvl + 1);
        q.push_back(e);
        qd.insert(qd.end(), m_prev.desc.begin() + 32 * i, m_prev.desc.begin() + 32 * (i + 1));
        q_angle.push_back(m_prev.kpts[i].angle);
        q_lm.push_back(id);
    }
    if (q.size() < 20) return false;
    std::vector<int32_t> idx(q.size()), dist(q.size());
    std::vector<float> cur_angle(cur.kpts.size());
    for (size_t i = 0; i < cur.kpts.size(); ++i) cur_angle[i] = cur.kpts[i].angle;
    int32_t n_m = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (lpslam_hip_match_projection(m_ctx, cur.slot, q.data(), qd.data(), (int32_t)q.size(), 100 /* HAMMING_DIST_THR_HIGH */, 1.0f, nullptr, m_stereo ? 1 : 0,
                                        idx.data(), dist.data(), &n_m) != LPSLAM_HIP_OK) return false;
        lpslam_hip_match_orientation_filter(q_angle.data(), cur_angle.data(), idx.data(), (int32_t)q.size(), &n_m);
        if (n_m >= 20) break;
        for (auto& e : q) e.radius *= 2.0f;
    }
    if (n_m < 20) return false;
    std::vector<int> cur_idx, lm_ids;
    for (size_t k = 0; k < q.size(); ++k) if (idx[k] >= 0) { cur_idx.push_back(idx[k]); lm_ids.push_back(q_lm[k]); }
    return poseFromMatches(cur, cur_idx, lm_ids, init, n_inliers);
}

// [UPSTREAM] tracking_module::optimize_current_frame_with_local_map: the landmarks of the local keyframes that the frame does not
// hold yet are projected with the pose just found and searched in a window of margin x scale factor of the predicted level
// (match::projection::match_frame_and_landmarks: margin 5 px for stereo, levels [predicted - 1, predicted], Lowe ratio 0.8 between
// candidates of one level, right-image check); then the motion-only optimiser runs again over all associations.
bool HipVslamTrackerBase::trackLocalMap(FrameData& cur, int& n_inliers)
{
    float scales[LPSLAM_HIP_MAX_LEVELS];
    const int32_t n_levels = m_numLevels;               // (the first array argument of level_info is the widths, not a count)
    lpslam_hip_level_info(m_ctx, nullptr, nullptr, nullptr, nullptr, scales);
    const double log_sf = std::log((double)m_scaleFactor);
    const Mat3 R = quatToRot(cur.pose.q);
    const double C[3] = {-(R.m[0] * cur.pose.t[0] + R.m[3] * cur.pose.t[1] + R.m[6] * cur.pose.t[2]),
                         -(R.m[1] * cur.pose.t[0] + R.m[4] * cur.pose.t[1] + R.m[7] * cur.pose.t[2]),
                         -(R.m[2] * cur.pose.t[0] + R.m[5] * cur.pose.t[1] + R.m[8] * cur.pose.t[2])};
    std::vector<uint8_t> taken(cur.kpts.size(), 0);
    std::unordered_map<int, char> held;
    size_t held_on_entry = 0; int n_new = 0;
    for (size_t i = 0; i < cur.kpts.size(); ++i) if (cur.landmark[i] >= 0) { taken[i] = 1; held[cur.landmark[i]] = 1; ++held_on_entry; }
    std::vector<lpslam_hip_proj_query> q;
    std::vector<uint8_t> qd;
    std::vector<int> q_lm;
    for (auto& kf : m_keyframes) {                       // local landmarks in keyframe / observation order (deterministic)
        for (auto& o : kf.obs) {
            if (held.count(o.landmark)) continue;
            held[o.landmark] = 1;
            auto it = m_landmarks.find(o.landmark);
            if (it == m_landmarks.end()) continue;
            const Landmark& lm = it->second;
            const double* X = lm.p;
            const double pc[3] = {R.m[0] * X[0] + R.m[1] * X[1] + R.m[2] * X[2] + cur.pose.t[0], R.m[3] * X[0] + R.m[4] * X[1] + R.m[5] * X[2] + cur.pose.t[1],
                                  R.m[6] * X[0] + R.m[7] * X[1] + R.m[8] * X[2] + cur.pose.t[2]};
            if (!(pc[2] > 0)) continue;
            const double u = m_cam.f_x * pc[0] / pc[2] + m_cam.c_x, v = m_cam.f_y * pc[1] / pc[2] + m_cam.c_y;
            if (u < 0 || v < 0 || u >= m_cam.resolution_x || v >= m_cam.resolution_y) continue;
            const double ray[3] = {X[0] - C[0], X[1] - C[1], X[2] - C[2]};
            const double dist = std::sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
            if (!(dist > 0) || dist < 0.8 * lm.min_valid || dist > 1.2 * lm.max_valid) continue;           // can_observe: scale range
            if ((ray[0] * lm.normal[0] + ray[1] * lm.normal[1] + ray[2] * lm.normal[2]) / dist < 0.5) continue;    // viewing angle < 60 deg
            const int lvl = std::min(std::max((int)std::ceil(std::log(lm.max_valid / dist) / log_sf), 0), n_levels - 1);
            lpslam_hip_proj_query e{};
            e.x = (float)u; e.y = (float)v; e.x_right = m_stereo ? (float)(u - m_cam.focal_x_baseline / pc[2]) : -1.0f;
            e.radius = 5.0f * scales[lvl];
            e.min_level = std::max(0, lvl - 1); e.max_level = lvl;
            q.push_back(e);
            qd.insert(qd.end(), lm.desc, lm.desc + 32);
            q_lm.push_back(o.landmark);
        }
    }
    if (!q.empty()) {
        std::vector<int32_t> idx(q.size()), dist(q.size());
        int32_t n_m = 0;
        if (lpslam_hip_match_projection(m_ctx, cur.slot, q.data(), qd.data(), (int32_t)q.size(), 100 /* HAMMING_DIST_THR_HIGH */, 0.8f, taken.data(), m_stereo ? 1 : 0,
                                        idx.data(), dist.data(), &n_m) != LPSLAM_HIP_OK) return false;
        for (size_t k = 0; k < q.size(); ++k) if (idx[k] >= 0) { cur.landmark[idx[k]] = q_lm[k]; ++n_new; }
    }
    g_local_map_joined += n_new;
    if (n_new == 0) { n_inliers = (int)held_on_entry; return true; }      // nothing joined: the pose found from the same associations stands
    std::vector<int> cur_idx, lm_ids;
    for (size_t i = 0; i < cur.kpts.size(); ++i) if (cur.landmark[i] >= 0) { cur_idx.push_back((int)i); lm_ids.push_back(cur.landmark[i]); }
    const Pose init = cur.pose;
    const std::vector<int> before = cur.landmark;
    for (size_t i = 0; i < cur.kpts.size(); ++i) if (!taken[i]) const_cast<std::vector<int>&>(before)[i] = -1;    // what the frame held on entry
    if (poseFromMatches(cur, cur_idx, lm_ids, init, n_inliers)) return true;
    cur.pose = init; cur.landmark = before;
    return false;
}

bool HipVslamTrackerBase::trackAgainstPrevious(FrameData& cur, int& n_inliers)
{
    // motion model first; descriptor matching against the whole previous frame is the fallback (upstream falls back to
    // BoW / robust matching, frame_tracker::bow_match_based_track / robust_match_based_track)
    if (trackWithMotionModel(cur, n_inliers)) { ++m_motionTracked; ++g_motion_tracked; return true; }
    n_inliers = 0;
    if (lpslam_hip_match_bf(m_ctx, cur.slot, m_prev.slot) != LPSLAM_HIP_OK) return false;
    std::vector<int32_t> mq(m_maxKp), mt(m_maxKp), md(m_maxKp);
    int32_t nm = 0;
    // HAMMING_DIST_THR_LOW = 50, Lowe ratio 0.9, mutual best
    if (lpslam_hip_get_bf_matches(m_ctx, cur.slot, m_prev.slot, 50, 0.9f, 1, mq.data(), mt.data(), md.data(), m_maxKp, &nm) != LPSLAM_HIP_OK) return false;
    std::vector<int> cur_idx, lm_ids;
    for (int k = 0; k < nm; ++k) {
        const int id = m_prev.landmark[mt[k]];
        if (id < 0) continue;
        cur_idx.push_back(mq[k]); lm_ids.push_back(id);
    }
    // prediction: constant velocity, else the previous pose
    Pose init = m_prev.pose;
    if (m_haveVelocity) {
        quatMul(m_velocity.q, m_prev.pose.q, init.q);
        const Mat3 Rv = quatToRot(m_velocity.q);
        for (int r = 0; r < 3; ++r) init.t[r] = Rv.m[r * 3] * m_prev.pose.t[0] + Rv.m[r * 3 + 1] * m_prev.pose.t[1] + Rv.m[r * 3 + 2] * m_prev.pose.t[2] + m_velocity.t[r];
    }
    return poseFromMatches(cur, cur_idx, lm_ids, init, n_inliers);
}

std::unique_ptr<HipVslamTrackerBase::MappingJob> HipVslamTrackerBase::prepareMapping()
{
    if (!m_enableMapping || m_keyframes.size() < 2) return nullptr;
    auto job = std::make_unique<MappingJob>();
    // landmarks observed by at least two keyframes of the window
    std::unordered_map<int, int> seen, index;
    for (auto& kf : m_keyframes) for (auto& o : kf.obs) seen[o.landmark]++;
    for (auto& kv : seen) {
        if (kv.second < 2) continue;
        auto it = m_landmarks.find(kv.first);
        if (it == m_landmarks.end()) continue;
        index[kv.first] = (int)job->ids.size(); job->ids.push_back(kv.first);
        job->pts.insert(job->pts.end(), it->second.p, it->second.p + 3);
    }
    if (job->ids.size() < 20) return nullptr;
    job->n_keyframes = (int)m_keyframes.size();
    for (size_t f = 0; f < m_keyframes.size(); ++f) {
        const Pose& p = m_keyframes[f].pose;
        job->poses.insert(job->poses.end(), {p.q[0], p.q[1], p.q[2], p.q[3], p.t[0], p.t[1], p.t[2]});
        job->fixed.push_back(f == 0 ? 1 : 0);         // the oldest keyframe anchors the gauge
        for (size_t k = 0; k < m_keyframes[f].obs.size(); ++k) {
            const KeyframeObs& o = m_keyframes[f].obs[k];
            auto it = index.find(o.landmark);
            if (it == index.end()) continue;
            job->obs.push_back({(int32_t)f, it->second, o.u, o.v, o.ur, o.inv_sigma2});
            job->origin.emplace_back((int)f, (int)k);
        }
    }
    job->outlier.assign(job->obs.size(), 0);
    return job;
}

// runs on the mapping thread: touches the job and the GPU only
void HipVslamTrackerBase::solveMapping(MappingJob& job) const
{
    lpslam_hip_ba_camera cam{m_cam.f_x, m_cam.f_y, m_cam.c_x, m_cam.c_y, m_cam.focal_x_baseline, std::sqrt(5.991), std::sqrt(7.815)};
    lpslam_hip_ba* ba = nullptr;
    if (lpslam_hip_ba_create(m_ctx, job.poses.data(), job.fixed.data(), job.n_keyframes, job.pts.data(), (int32_t)job.ids.size(), job.obs.data(),
                             (int32_t)job.obs.size(), &cam, &ba) != LPSLAM_HIP_OK) return;
    job.solved = lpslam_hip_ba_local(ba, 5, 10, job.outlier.data()) == LPSLAM_HIP_OK &&
                 lpslam_hip_ba_get(ba, job.poses.data(), job.pts.data()) == LPSLAM_HIP_OK;
    lpslam_hip_ba_destroy(ba);
}

void HipVslamTrackerBase::applyMapping(const MappingJob& job)
{
    if (!job.solved || job.n_keyframes != (int)m_keyframes.size()) return;
    for (size_t f = 0; f < m_keyframes.size(); ++f) {
        for (int k = 0; k < 4; ++k) m_keyframes[f].pose.q[k] = job.poses[7 * f + k];
        for (int k = 0; k < 3; ++k) m_keyframes[f].pose.t[k] = job.poses[7 * f + 4 + k];
        const long ai = m_keyframes[f].archive_index;
        if (ai >= 0 && (size_t)ai < m_archive.size()) m_archive[(size_t)ai].pose = m_keyframes[f].pose;
    }
    for (size_t j = 0; j < job.ids.size(); ++j) {
        auto it = m_landmarks.find(job.ids[j]);
        if (it != m_landmarks.end()) { it->second.p[0] = job.pts[3 * j]; it->second.p[1] = job.pts[3 * j + 1]; it->second.p[2] = job.pts[3 * j + 2]; }
    }
    // erase outlier observations (back to front so indices stay valid)
    for (size_t k = job.obs.size(); k-- > 0;) {
        if (!job.outlier[k]) continue;
        auto& v = m_keyframes[job.origin[k].first].obs;
        auto it = m_landmarks.find(v[job.origin[k].second].landmark);
        if (it != m_landmarks.end() && --it->second.n_obs <= 0) m_landmarks.erase(it);
        v.erase(v.begin() + job.origin[k].second);
    }
}

// ---- monocular initialisation ([UPSTREAM] module::initializer::initialize for Monocular setups) ---------------------------------
// The first frame with enough keypoints becomes the reference.  Every later frame is matched against it in a 100-px window
// around where each level-0 reference keypoint was last matched (match::area::match_in_consistent_area, Hamming <= 50, ratio 0.9,
// orientation check); with fewer than 100 matches the reference is dropped.  Two-view geometry (two_view.h) decides whether the
// pair has enough parallax; on success the map is the two keyframes and the triangulated landmarks, scaled to a median scene
// depth of 1 and refined by a global bundle adjustment of 20 iterations.
bool HipVslamTrackerBase::monoInitialize(FrameData& cur)
{
    const int n_cur = (int)cur.kpts.size();
    if (!m_haveMonoRef) {
        if (n_cur < 100) return false;
        m_monoRef = cur; m_haveMonoRef = true;
        m_monoPrevMatched.resize(2 * (size_t)n_cur);
        for (int i = 0; i < n_cur; ++i) { m_monoPrevMatched[2 * i] = cur.kpts[i].x; m_monoPrevMatched[2 * i + 1] = cur.kpts[i].y; }
        return false;
    }
    if (n_cur < 100) { m_haveMonoRef = false; return false; }
    const FrameData& ref = m_monoRef;
    std::vector<lpslam_hip_proj_query> q;
    std::vector<uint8_t> qd;
    std::vector<int> q_ref;
    std::vector<float> q_angle;
    for (size_t i = 0; i < ref.kpts.size(); ++i) {
        if (ref.kpts[i].octave > 0) continue;
        lpslam_hip_proj_query e{};
        e.x = m_monoPrevMatched[2 * i]; e.y = m_monoPrevMatched[2 * i + 1]; e.x_right = -1.0f; e.radius = 100.0f;
        e.min_level = 0; e.max_level = 0;
        q.push_back(e);
        qd.insert(qd.end(), ref.desc.begin() + 32 * i, ref.desc.begin() + 32 * (i + 1));
        q_ref.push_back((int)i); q_angle.push_back(ref.kpts[i].angle);
    }
    if (q.size() < 100) { m_haveMonoRef = false; return false; }
    std::vector<int32_t> idx(q.size()), dist(q.size());
    int32_t n_m = 0;
    if (lpslam_hip_match_area(m_ctx, cur.slot, q.data(), qd.data(), (int32_t)q.size(), 50 /* HAMMING_DIST_THR_LOW */, 0.9f, idx.data(), dist.data(), &n_m) != LPSLAM_HIP_OK) return false;
    std::vector<float> cur_angle(cur.kpts.size());
    for (size_t i = 0; i < cur.kpts.size(); ++i) cur_angle[i] = cur.kpts[i].angle;
    lpslam_hip_match_orientation_filter(q_angle.data(), cur_angle.data(), idx.data(), (int32_t)q.size(), &n_m);
    if (n_m < 100) { m_haveMonoRef = false; return false; }          // too few: a new reference with the next frame
    std::vector<int32_t> matches;
    for (size_t k = 0; k < q.size(); ++k) {
        if (idx[k] < 0) continue;
        matches.push_back(q_ref[k]); matches.push_back(idx[k]);
        m_monoPrevMatched[2 * (size_t)q_ref[k]] = cur.kpts[idx[k]].x; m_monoPrevMatched[2 * (size_t)q_ref[k] + 1] = cur.kpts[idx[k]].y;
    }
    std::vector<float> kr(2 * ref.kpts.size()), kc(2 * cur.kpts.size());
    for (size_t i = 0; i < ref.kpts.size(); ++i) { kr[2 * i] = ref.kpts[i].x; kr[2 * i + 1] = ref.kpts[i].y; }
    for (size_t i = 0; i < cur.kpts.size(); ++i) { kc[2 * i] = cur.kpts[i].x; kc[2 * i + 1] = cur.kpts[i].y; }
    const double K[4] = {m_cam.f_x, m_cam.f_y, m_cam.c_x, m_cam.c_y};
    TwoViewParams prm;
    TwoViewResult tv;
    if (!two_view_initialize(K, kr.data(), kc.data(), matches.data(), (int)(matches.size() / 2), prm, tv)) return false;

    // ---- the initial map: reference keyframe at the origin, current keyframe at (R, t), scale: median depth in the reference = 1
    std::vector<double> depths;
    for (size_t m = 0; m < tv.triangulated.size(); ++m) if (tv.triangulated[m]) depths.push_back(tv.points[3 * m + 2]);
    if (depths.size() < 50) return false;
    std::nth_element(depths.begin(), depths.begin() + depths.size() / 2, depths.end());
    const double median = depths[depths.size() / 2];
    if (!(median > 0)) return false;
    const double inv = 1.0 / median;
    if (m_mapThread.joinable()) finishMapping();
    m_landmarks.clear(); m_keyframes.clear(); m_nextLandmarkId = 0;
    FrameData reff = ref;
    reff.pose = Pose();
    reff.landmark.assign(reff.kpts.size(), -1);
    Mat3 Rm; std::copy(tv.R, tv.R + 9, Rm.m);
    rotToQuat(Rm, cur.pose.q);
    for (int a = 0; a < 3; ++a) cur.pose.t[a] = tv.t[a] * inv;
    cur.landmark.assign(cur.kpts.size(), -1);
    float scales[LPSLAM_HIP_MAX_LEVELS];
    lpslam_hip_level_info(m_ctx, nullptr, nullptr, nullptr, nullptr, scales);
    Keyframe k0, k1;
    k0.pose = reff.pose; k1.pose = cur.pose;
    for (size_t m = 0; m < tv.triangulated.size(); ++m) {
        if (!tv.triangulated[m]) continue;
        const int ir = matches[2 * m], ic = matches[2 * m + 1];
        Landmark lm;
        for (int a = 0; a < 3; ++a) lm.p[a] = tv.points[3 * m + a] * inv;
        initLandmarkView(lm, reff.pose, reff.kpts[ir], reff.desc.data() + 32 * (size_t)ir);
        lm.n_obs = 2;
        const int id = m_nextLandmarkId++;
        m_landmarks[id] = lm;
        reff.landmark[ir] = id; cur.landmark[ic] = id;
        const double s0 = scales[reff.kpts[ir].octave], s1 = scales[cur.kpts[ic].octave];
        k0.obs.push_back({id, reff.kpts[ir].x, reff.kpts[ir].y, -1.0, 1.0 / (s0 * s0)});
        k1.obs.push_back({id, cur.kpts[ic].x, cur.kpts[ic].y, -1.0, 1.0 / (s1 * s1)});
    }
    k0.kpts = reff.kpts; k0.desc = reff.desc; k0.landmark = reff.landmark;
    k1.kpts = cur.kpts; k1.desc = cur.desc; k1.landmark = cur.landmark;
    m_keyframes.push_back(std::move(k0)); m_keyframes.push_back(std::move(k1));
    m_keyframeCount += 2;
    m_framesSinceKeyframe = 0;
    // global bundle adjustment of the two-keyframe map (20 iterations, Huber), inline: nothing can be tracked before it
    {
        const bool keep_async = m_asyncMapping;
        m_asyncMapping = false;
        auto job = prepareMapping();
        m_asyncMapping = keep_async;
        if (job) {
            lpslam_hip_ba_camera cam{m_cam.f_x, m_cam.f_y, m_cam.c_x, m_cam.c_y, 0.0, std::sqrt(5.991), std::sqrt(7.815)};
            lpslam_hip_ba* ba = nullptr;
            if (lpslam_hip_ba_create(m_ctx, job->poses.data(), job->fixed.data(), job->n_keyframes, job->pts.data(), (int32_t)job->ids.size(), job->obs.data(),
                                     (int32_t)job->obs.size(), &cam, &ba) == LPSLAM_HIP_OK) {
                int32_t done = 0;
                job->solved = lpslam_hip_ba_optimize(ba, 1, 20, nullptr, &done) == LPSLAM_HIP_OK && lpslam_hip_ba_get(ba, job->poses.data(), job->pts.data()) == LPSLAM_HIP_OK;
                lpslam_hip_ba_destroy(ba);
                if (job->solved) applyMapping(*job);
            }
        }
    }
    cur.pose = m_keyframes.back().pose;
    m_haveMonoRef = false;
    logMessage(LpSlamLogLevel_Info, "VSLAM monocular map initialised: " + std::to_string(m_landmarks.size()) + " landmarks, model " + (tv.model == 0 ? "H" : "F"));
    return m_landmarks.size() >= 50;
}

// New landmarks for a monocular keyframe: keypoints without a landmark are matched by descriptor against the previous keyframe's
// (brute force on the device, mutual best, ratio 0.8), kept when they satisfy the epipolar constraint of the two poses, and
// triangulated; the checks are upstream's (positive depth in both views, reprojection chi2 <= 5.991 per view, parallax, scale
// consistency of the distances with the pyramid levels).
void HipVslamTrackerBase::monoTriangulate(Keyframe& prev, Keyframe& kf, FrameData& f)
{
    if (prev.kpts.empty() || f.kpts.empty()) return;
    const int scratch = f.slot ^ 1;                      // monocular frames use slots 0 / 2
    if (lpslam_hip_set_descriptors(m_ctx, scratch, prev.desc.data(), (int32_t)prev.kpts.size()) != LPSLAM_HIP_OK) return;
    if (lpslam_hip_match_bf(m_ctx, f.slot, scratch) != LPSLAM_HIP_OK) return;
    std::vector<int32_t> mq(m_maxKp), mt(m_maxKp), md(m_maxKp);
    int32_t nm = 0;
    if (lpslam_hip_get_bf_matches(m_ctx, f.slot, scratch, 50, 0.8f, 1, mq.data(), mt.data(), md.data(), m_maxKp, &nm) != LPSLAM_HIP_OK) return;
    float scales[LPSLAM_HIP_MAX_LEVELS];
    lpslam_hip_level_info(m_ctx, nullptr, nullptr, nullptr, nullptr, scales);
    const double fx = m_cam.f_x, fy = m_cam.f_y, cx = m_cam.c_x, cy = m_cam.c_y;
    const Mat3 R1 = quatToRot(prev.pose.q), R2 = quatToRot(f.pose.q);
    auto proj = [&](const Mat3& R, const double* t, double* P) {
        for (int c = 0; c < 3; ++c) { P[c] = fx * R.m[c] + cx * R.m[6 + c]; P[4 + c] = fy * R.m[3 + c] + cy * R.m[6 + c]; P[8 + c] = R.m[6 + c]; }
        P[3] = fx * t[0] + cx * t[2]; P[7] = fy * t[1] + cy * t[2]; P[11] = t[2];
    };
    double P1[12], P2[12];
    proj(R1, prev.pose.t, P1); proj(R2, f.pose.t, P2);
    auto centre = [](const Mat3& R, const double* t, double* C) { for (int a = 0; a < 3; ++a) C[a] = -(R.m[a] * t[0] + R.m[3 + a] * t[1] + R.m[6 + a] * t[2]); };
    double C1[3], C2[3];
    centre(R1, prev.pose.t, C1); centre(R2, f.pose.t, C2);
    // relative pose prev -> cur and the fundamental matrix x2^T F x1 = 0
    double R21[9], t21[3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R21[r * 3 + c] = R2.m[r * 3] * R1.m[c * 3] + R2.m[r * 3 + 1] * R1.m[c * 3 + 1] + R2.m[r * 3 + 2] * R1.m[c * 3 + 2];
    for (int r = 0; r < 3; ++r) t21[r] = f.pose.t[r] - (R21[r * 3] * prev.pose.t[0] + R21[r * 3 + 1] * prev.pose.t[1] + R21[r * 3 + 2] * prev.pose.t[2]);
    const double tx[9] = {0, -t21[2], t21[1], t21[2], 0, -t21[0], -t21[1], t21[0], 0};
    double E[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) E[r * 3 + c] = tx[r * 3] * R21[c] + tx[r * 3 + 1] * R21[3 + c] + tx[r * 3 + 2] * R21[6 + c];
    const double ratio_factor = 1.5 * m_scaleFactor;
    int created = 0;
    for (int k = 0; k < nm; ++k) {
        const int ic = mq[k], ip = mt[k];
        if (f.landmark[ic] >= 0 || prev.landmark[(size_t)ip] >= 0) continue;
        const lpslam_hip_keypoint& k1 = prev.kpts[(size_t)ip]; const lpslam_hip_keypoint& k2 = f.kpts[(size_t)ic];
        const double x1n[3] = {(k1.x - cx) / fx, (k1.y - cy) / fy, 1.0}, x2n[3] = {(k2.x - cx) / fx, (k2.y - cy) / fy, 1.0};
        // epipolar line of x1 in the current image (normalised coordinates -> pixels: a / fx, b / fy)
        const double l[3] = {E[0] * x1n[0] + E[1] * x1n[1] + E[2], E[3] * x1n[0] + E[4] * x1n[1] + E[5], E[6] * x1n[0] + E[7] * x1n[1] + E[8]};
        const double num = l[0] * x2n[0] + l[1] * x2n[1] + l[2];
        const double a = l[0] / fx, b = l[1] / fy;
        const double s2 = scales[k2.octave] * scales[k2.octave];
        if (num * num / (a * a + b * b) > 3.84 * s2) continue;
        // parallax of the two rays (world frame)
        double r1[3], r2[3];
        for (int a2 = 0; a2 < 3; ++a2) { r1[a2] = R1.m[a2] * x1n[0] + R1.m[3 + a2] * x1n[1] + R1.m[6 + a2]; r2[a2] = R2.m[a2] * x2n[0] + R2.m[3 + a2] * x2n[1] + R2.m[6 + a2]; }
        const double cosr = (r1[0] * r2[0] + r1[1] * r2[1] + r1[2] * r2[2]) / (std::sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]) * std::sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]));
        if (!(cosr > 0 && cosr < 0.9998)) continue;
        const double p1[2] = {k1.x, k1.y}, p2[2] = {k2.x, k2.y};
        double X[3];
        if (!triangulate_point(P1, P2, p1, p2, X)) continue;
        const double Xc1[3] = {R1.m[0] * X[0] + R1.m[1] * X[1] + R1.m[2] * X[2] + prev.pose.t[0], R1.m[3] * X[0] + R1.m[4] * X[1] + R1.m[5] * X[2] + prev.pose.t[1],
                               R1.m[6] * X[0] + R1.m[7] * X[1] + R1.m[8] * X[2] + prev.pose.t[2]};
        const double Xc2[3] = {R2.m[0] * X[0] + R2.m[1] * X[1] + R2.m[2] * X[2] + f.pose.t[0], R2.m[3] * X[0] + R2.m[4] * X[1] + R2.m[5] * X[2] + f.pose.t[1],
                               R2.m[6] * X[0] + R2.m[7] * X[1] + R2.m[8] * X[2] + f.pose.t[2]};
        if (!(Xc1[2] > 0) || !(Xc2[2] > 0)) continue;
        const double s1 = scales[k1.octave] * scales[k1.octave];
        const double e1x = fx * Xc1[0] / Xc1[2] + cx - k1.x, e1y = fy * Xc1[1] / Xc1[2] + cy - k1.y;
        if (e1x * e1x + e1y * e1y > 5.991 * s1) continue;
        const double e2x = fx * Xc2[0] / Xc2[2] + cx - k2.x, e2y = fy * Xc2[1] / Xc2[2] + cy - k2.y;
        if (e2x * e2x + e2y * e2y > 5.991 * s2) continue;
        const double d1 = std::sqrt((X[0] - C1[0]) * (X[0] - C1[0]) + (X[1] - C1[1]) * (X[1] - C1[1]) + (X[2] - C1[2]) * (X[2] - C1[2]));
        const double d2 = std::sqrt((X[0] - C2[0]) * (X[0] - C2[0]) + (X[1] - C2[1]) * (X[1] - C2[1]) + (X[2] - C2[2]) * (X[2] - C2[2]));
        if (!(d1 > 0) || !(d2 > 0)) continue;
        const double ratio_d = d2 / d1, ratio_o = (double)scales[k1.octave] / (double)scales[k2.octave];
        if (ratio_d * ratio_factor < ratio_o || ratio_d > ratio_o * ratio_factor) continue;
        Landmark lm;
        lm.p[0] = X[0]; lm.p[1] = X[1]; lm.p[2] = X[2];
        initLandmarkView(lm, f.pose, k2, f.desc.data() + 32 * (size_t)ic);
        lm.n_obs = 2;
        const int id = m_nextLandmarkId++;
        m_landmarks[id] = lm;
        f.landmark[ic] = id; prev.landmark[(size_t)ip] = id;
        prev.obs.push_back({id, k1.x, k1.y, -1.0, 1.0 / s1});
        kf.obs.push_back({id, k2.x, k2.y, -1.0, 1.0 / s2});
        ++created;
    }
    (void)created;
}

// ---- loop closing ([UPSTREAM] module::loop_detector + loop_bundle_adjuster's pose-graph stage, global_optimization_module) -----
// Candidates: archived keyframes that are old enough and whose camera centre lies near the current estimate (the vocabulary of
// the reference is replaced by position gating + brute-force descriptor matching on the device).  A candidate's mutual matches
// with landmarks on both sides feed the Sim3 optimiser of the loop detector (lpslam_hip_sim3_transform_optimize, scale fixed for
// stereo); with >= 20 inliers the loop is closed: pose graph over the keyframes of the loop (consecutive edges + the loop edge,
// lpslam_hip_sim3_optimize, 50 iterations), landmarks move with the keyframe that created them.
void HipVslamTrackerBase::archiveKeyframe(Keyframe& kf, const FrameData& f)
{
    ArchivedKeyframe a;
    a.pose = kf.pose; a.kpts = f.kpts; a.desc = f.desc;
    a.pc.assign(3 * f.kpts.size(), std::numeric_limits<double>::quiet_NaN());
    const Mat3 R = quatToRot(kf.pose.q);
    for (size_t i = 0; i < f.kpts.size(); ++i) {
        if (f.landmark[i] < 0) continue;
        auto it = m_landmarks.find(f.landmark[i]);
        if (it == m_landmarks.end()) continue;
        const double* X = it->second.p;
        for (int r = 0; r < 3; ++r) a.pc[3 * i + r] = R.m[r * 3] * X[0] + R.m[r * 3 + 1] * X[1] + R.m[r * 3 + 2] * X[2] + kf.pose.t[r];
    }
    kf.archive_index = (long)m_archive.size();
    m_archive.push_back(std::move(a));
}

namespace {
// SE3 as (q, t): x_c = R x_w + t
struct Se3 { double q[4]; double t[3]; };
Se3 se3_mul(const Se3& a, const Se3& b)            // a after b
{
    Se3 o;
    quatMul(a.q, b.q, o.q);
    const Mat3 Ra = quatToRot(a.q);
    for (int r = 0; r < 3; ++r) o.t[r] = Ra.m[r * 3] * b.t[0] + Ra.m[r * 3 + 1] * b.t[1] + Ra.m[r * 3 + 2] * b.t[2] + a.t[r];
    return o;
}
Se3 se3_inv(const Se3& a)
{
    Se3 o;
    o.q[0] = a.q[0]; o.q[1] = -a.q[1]; o.q[2] = -a.q[2]; o.q[3] = -a.q[3];
    const Mat3 R = quatToRot(a.q);
    for (int r = 0; r < 3; ++r) o.t[r] = -(R.m[r] * a.t[0] + R.m[3 + r] * a.t[1] + R.m[6 + r] * a.t[2]);
    return o;
}
}  // namespace

bool HipVslamTrackerBase::detectAndCloseLoop(FrameData& cur)
{
    const long cur_ai = m_keyframes.back().archive_index;
    if (cur_ai < 0) return false;
    const long newest_candidate = cur_ai - 2L * m_localWindow;          // well outside the local window
    if (newest_candidate < 0) return false;
    const ArchivedKeyframe& ca = m_archive[(size_t)cur_ai];
    auto centre = [](const Pose& p, double* C) { const Mat3 R = quatToRot(p.q); for (int a = 0; a < 3; ++a) C[a] = -(R.m[a] * p.t[0] + R.m[3 + a] * p.t[1] + R.m[6 + a] * p.t[2]); };
    double Cc[3];
    centre(ca.pose, Cc);
    const double radius = 10.0 * m_cam.focal_x_baseline / m_cam.f_x + 1.0;              // metres: ten baselines + 1
    std::vector<std::pair<double, long>> near;
    for (long a = 0; a <= newest_candidate; ++a) {
        double C[3];
        centre(m_archive[(size_t)a].pose, C);
        const double d = std::sqrt((C[0] - Cc[0]) * (C[0] - Cc[0]) + (C[1] - Cc[1]) * (C[1] - Cc[1]) + (C[2] - Cc[2]) * (C[2] - Cc[2]));
        if (d < radius) near.emplace_back(d, a);
    }
    if (near.empty()) return false;
    std::sort(near.begin(), near.end());
    if (near.size() > 3) near.resize(3);
    float scales[LPSLAM_HIP_MAX_LEVELS];
    lpslam_hip_level_info(m_ctx, nullptr, nullptr, nullptr, nullptr, scales);
    const int scratch = cur.slot ^ 2;                  // the previous frame's slot pair is free for the descriptors of a candidate ...
    std::vector<lpslam_hip_sim3_pair> pairs;
    std::vector<int32_t> start{0};
    std::vector<double> s12;
    std::vector<long> cand;
    std::vector<int32_t> mq(m_maxKp), mt(m_maxKp), md(m_maxKp);
    const Se3 Tc{{ca.pose.q[0], ca.pose.q[1], ca.pose.q[2], ca.pose.q[3]}, {ca.pose.t[0], ca.pose.t[1], ca.pose.t[2]}};
    for (auto& nc : near) {
        const ArchivedKeyframe& ka = m_archive[(size_t)nc.second];
        if (lpslam_hip_set_descriptors(m_ctx, scratch, ka.desc.data(), (int32_t)ka.kpts.size()) != LPSLAM_HIP_OK) continue;
        if (lpslam_hip_match_bf(m_ctx, cur.slot, scratch) != LPSLAM_HIP_OK) continue;
        int32_t nm = 0;
        if (lpslam_hip_get_bf_matches(m_ctx, cur.slot, scratch, 50, 0.75f, 1, mq.data(), mt.data(), md.data(), m_maxKp, &nm) != LPSLAM_HIP_OK) continue;
        const size_t first = pairs.size();
        for (int k = 0; k < nm; ++k) {
            const size_t ic = (size_t)mq[k], ia = (size_t)mt[k];
            if (!std::isfinite(ca.pc[3 * ic]) || !std::isfinite(ka.pc[3 * ia])) continue;
            lpslam_hip_sim3_pair pr{};
            for (int r = 0; r < 3; ++r) { pr.p1c[r] = ca.pc[3 * ic + r]; pr.p2c[r] = ka.pc[3 * ia + r]; }
            pr.obs1[0] = ca.kpts[ic].x; pr.obs1[1] = ca.kpts[ic].y; pr.obs2[0] = ka.kpts[ia].x; pr.obs2[1] = ka.kpts[ia].y;
            const double s1 = scales[ca.kpts[ic].octave], s2 = scales[ka.kpts[ia].octave];
            pr.inv_sigma2_1 = 1.0 / (s1 * s1); pr.inv_sigma2_2 = 1.0 / (s2 * s2);
            pairs.push_back(pr);
        }
        if (pairs.size() - first < 20) { pairs.resize(first); continue; }               // [UPSTREAM] num_matches >= 20 to try a candidate
        start.push_back((int32_t)pairs.size());
        cand.push_back(nc.second);
        const Se3 Ta{{ka.pose.q[0], ka.pose.q[1], ka.pose.q[2], ka.pose.q[3]}, {ka.pose.t[0], ka.pose.t[1], ka.pose.t[2]}};
        const Se3 T12 = se3_mul(Tc, se3_inv(Ta));                                        // candidate camera -> current camera
        s12.insert(s12.end(), {T12.q[0], T12.q[1], T12.q[2], T12.q[3], T12.t[0], T12.t[1], T12.t[2], 1.0});
    }
    // the scratch slot's keypoint count no longer describes an extracted image; the next extraction into it rewrites it
    if (cand.empty()) return false;
    const double cam[4] = {m_cam.f_x, m_cam.f_y, m_cam.c_x, m_cam.c_y};
    std::vector<uint8_t> inl(pairs.size());
    std::vector<int32_t> n_inl(cand.size(), 0);
    if (lpslam_hip_sim3_transform_optimize(m_ctx, (int32_t)cand.size(), s12.data(), pairs.data(), start.data(), cam, cam, 10.0, 1, inl.data(), n_inl.data()) != LPSLAM_HIP_OK) return false;
    int best = -1;
    for (size_t i = 0; i < cand.size(); ++i) if (n_inl[i] >= 20 && (best < 0 || n_inl[i] > n_inl[(size_t)best])) best = (int)i;
    if (best < 0) return false;

    // ---- pose graph over the keyframes of the loop: a = candidate (fixed) ... cur
    const long a0 = cand[(size_t)best];
    const int n = (int)(cur_ai - a0 + 1);
    std::vector<double> verts(8 * (size_t)n);
    std::vector<uint8_t> fixed((size_t)n, 0);
    fixed[0] = 1;
    std::vector<Se3> old((size_t)n);
    for (int v = 0; v < n; ++v) {
        const Pose& p = m_archive[(size_t)(a0 + v)].pose;
        old[(size_t)v] = Se3{{p.q[0], p.q[1], p.q[2], p.q[3]}, {p.t[0], p.t[1], p.t[2]}};
        const double row[8] = {p.q[0], p.q[1], p.q[2], p.q[3], p.t[0], p.t[1], p.t[2], 1.0};
        std::copy(row, row + 8, verts.begin() + 8 * (size_t)v);
    }
    std::vector<lpslam_hip_sim3_edge> edges;
    for (int v = 0; v + 1 < n; ++v) {                    // consecutive keyframes: measurement = S_j S_i^-1 of the current estimates
        const Se3 m = se3_mul(old[(size_t)v + 1], se3_inv(old[(size_t)v]));
        lpslam_hip_sim3_edge e{};
        e.i = v; e.j = v + 1;
        const double row[8] = {m.q[0], m.q[1], m.q[2], m.q[3], m.t[0], m.t[1], m.t[2], 1.0};
        std::copy(row, row + 8, e.meas);
        edges.push_back(e);
    }
    {   // the loop edge: current <- candidate as the transform optimiser found it
        lpslam_hip_sim3_edge e{};
        e.i = 0; e.j = n - 1;
        std::copy(s12.begin() + 8 * (size_t)best, s12.begin() + 8 * (size_t)best + 8, e.meas);
        edges.push_back(e);
    }
    lpslam_hip_sim3* graph = nullptr;
    if (lpslam_hip_sim3_create(m_ctx, verts.data(), fixed.data(), n, edges.data(), (int32_t)edges.size(), 1, &graph) != LPSLAM_HIP_OK) return false;
    int32_t done = 0;
    const bool ok = lpslam_hip_sim3_optimize(graph, 50, nullptr, &done) == LPSLAM_HIP_OK && lpslam_hip_sim3_get(graph, verts.data()) == LPSLAM_HIP_OK;
    lpslam_hip_sim3_destroy(graph);
    if (!ok) return false;
    finishMapping();                                     // no window solve may be in flight while the map moves
    std::vector<Se3> neu((size_t)n);
    for (int v = 0; v < n; ++v) {
        const double* r = &verts[8 * (size_t)v];
        const double qn = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]), s = r[7] > 0 ? r[7] : 1.0;
        neu[(size_t)v] = Se3{{r[0] / qn, r[1] / qn, r[2] / qn, r[3] / qn}, {r[4] / s, r[5] / s, r[6] / s}};
        Pose& p = m_archive[(size_t)(a0 + v)].pose;
        for (int k = 0; k < 4; ++k) p.q[k] = neu[(size_t)v].q[k];
        for (int k = 0; k < 3; ++k) p.t[k] = neu[(size_t)v].t[k];
    }
    for (auto& kv : m_landmarks) {                       // X_new = T_ref_new^-1 (T_ref_old X)
        const long rk = kv.second.ref_kf;
        if (rk < a0 || rk > cur_ai) continue;
        const Se3& To = old[(size_t)(rk - a0)]; const Se3 Tni = se3_inv(neu[(size_t)(rk - a0)]);
        const Mat3 Ro = quatToRot(To.q), Rn = quatToRot(Tni.q);
        double xc[3], xw[3];
        for (int r = 0; r < 3; ++r) xc[r] = Ro.m[r * 3] * kv.second.p[0] + Ro.m[r * 3 + 1] * kv.second.p[1] + Ro.m[r * 3 + 2] * kv.second.p[2] + To.t[r];
        for (int r = 0; r < 3; ++r) xw[r] = Rn.m[r * 3] * xc[0] + Rn.m[r * 3 + 1] * xc[1] + Rn.m[r * 3 + 2] * xc[2] + Tni.t[r];
        for (int r = 0; r < 3; ++r) kv.second.p[r] = xw[r];
    }
    for (auto& kf : m_keyframes) if (kf.archive_index >= a0 && kf.archive_index <= cur_ai) kf.pose = m_archive[(size_t)kf.archive_index].pose;
    cur.pose = m_archive[(size_t)cur_ai].pose;
    ++m_loopsClosed; ++g_loops_closed;
    logMessage(LpSlamLogLevel_Info, "VSLAM loop closed: keyframe " + std::to_string(cur_ai) + " with " + std::to_string(a0) + ", " + std::to_string(n_inl[(size_t)best]) + " inliers");
    return true;
}

void HipVslamTrackerBase::mappingLoop()
{
    std::unique_lock<std::mutex> lk(m_mapMutex);
    for (;;) {
        m_mapCv.wait(lk, [this] { return m_mapQuit || m_mapIn; });
        if (m_mapQuit) return;
        std::unique_ptr<MappingJob> job = std::move(m_mapIn);
        lk.unlock();
        solveMapping(*job);
        lk.lock();
        m_mapOut = std::move(job);
        m_mapBusy = false;
        m_mapCv.notify_all();
    }
}

void HipVslamTrackerBase::stopMappingThread()
{
    if (!m_mapThread.joinable()) return;
    {
        std::unique_lock<std::mutex> lk(m_mapMutex);
        m_mapCv.wait(lk, [this] { return !m_mapBusy; });
        m_mapQuit = true;
        m_mapCv.notify_all();
    }
    m_mapThread.join();
    m_mapQuit = false; m_mapOut.reset();
}

void HipVslamTrackerBase::startMapping()
{
    auto job = prepareMapping();
    if (!job) return;
    if (!m_asyncMapping) { solveMapping(*job); applyMapping(*job); return; }
    if (!m_mapThread.joinable()) m_mapThread = std::thread([this] { mappingLoop(); });
    std::lock_guard<std::mutex> lk(m_mapMutex);
    m_mapIn = std::move(job); m_mapBusy = true;
    m_mapCv.notify_all();
}

void HipVslamTrackerBase::finishMapping()
{
    std::unique_ptr<MappingJob> job;
    {
        std::unique_lock<std::mutex> lk(m_mapMutex);
        m_mapCv.wait(lk, [this] { return !m_mapBusy; });
        job = std::move(m_mapOut);
    }
    if (job) applyMapping(*job);
}

TrackerBase::ProcessImageResult HipVslamTrackerBase::trackFrame(CameraQueueEntry& cam, bool stereo)
{
    ProcessImageResult res;
    std::scoped_lock lock(m_slamLock);
    if (!m_ctx) { logMessage(LpSlamLogLevel_Error, "VSLAM instance not created"); return res; }
    if (stereo && !cam.image_second.has_value()) { logMessage(LpSlamLogLevel_Error, "VSLAM stereo needs two images"); return res; }
    if (cam.image.width != m_cam.resolution_x || cam.image.height != m_cam.resolution_y ||
        (stereo && (cam.image_second->width != cam.image.width || cam.image_second->height != cam.image.height))) {
        logMessage(LpSlamLogLevel_Error, "Image size does not match the camera configuration");
        return res;
    }
    if (!m_firstImageTimestamp) m_firstImageTimestamp = cam.timestamp;
    const auto t0 = std::chrono::steady_clock::now();

    FrameData cur;
    cur.slot = (int)(m_imageTracked % 2) * 2;
    bool ok;
    if (m_rectify) {       // raw frames: undistort + rectify on the device
        ok = lpslam_hip_upload_raw_image(m_ctx, cur.slot, 0, cam.image.pixels.data(), cam.image.width) == LPSLAM_HIP_OK;
        if (ok && stereo) ok = lpslam_hip_upload_raw_image(m_ctx, cur.slot + 1, 1, cam.image_second->pixels.data(), cam.image.width) == LPSLAM_HIP_OK;
    } else {
        ok = lpslam_hip_upload_image(m_ctx, cur.slot, cam.image.pixels.data(), cam.image.width) == LPSLAM_HIP_OK;
        if (ok && stereo) ok = lpslam_hip_upload_image(m_ctx, cur.slot + 1, cam.image_second->pixels.data(), cam.image.width) == LPSLAM_HIP_OK;
    }
    if (ok) ok = lpslam_hip_extract_range(m_ctx, cur.slot, stereo ? 2 : 1) == LPSLAM_HIP_OK;
    if (ok && stereo) {
        const float baseline = (float)(m_cam.focal_x_baseline / m_cam.f_x);
        ok = lpslam_hip_match_stereo(m_ctx, cur.slot, cur.slot + 1, (float)m_cam.focal_x_baseline, baseline) == LPSLAM_HIP_OK;
    }
    int32_t n = 0;
    cur.kpts.resize(m_maxKp); cur.desc.resize((size_t)m_maxKp * 32);
    cur.x_right.assign(m_maxKp, -1.0f); cur.depth.assign(m_maxKp, -1.0f);
    if (ok) ok = lpslam_hip_get_frame(m_ctx, cur.slot, cur.kpts.data(), cur.desc.data(), stereo ? cur.x_right.data() : nullptr,
                                      stereo ? cur.depth.data() : nullptr, m_maxKp, &n) == LPSLAM_HIP_OK;
    if (!ok) { logMessage(LpSlamLogLevel_Error, std::string("HIP front end failed: ") + lpslam_hip_last_error()); return res; }
    cur.kpts.resize(n); cur.desc.resize((size_t)n * 32);
    cur.x_right.resize(n); cur.depth.resize(n); cur.landmark.assign(n, -1);
    ++m_imageTracked;

    if (!stereo && m_state != TrackerState::Tracking) {
        // monocular: no map until two views with enough parallax have been found
        m_state = TrackerState::Initializing;
        if (monoInitialize(cur)) m_state = TrackerState::Tracking;
        m_haveVelocity = false;
        m_prev = std::move(cur); m_havePrev = true;
    } else if (m_state != TrackerState::Tracking) {
        m_state = TrackerState::Initializing;
        if (initializeMap(cur)) m_state = TrackerState::Tracking;
        m_haveVelocity = false;
        m_prev = std::move(cur); m_havePrev = true;
    } else {
        int inliers = 0;
        if (trackAgainstPrevious(cur, inliers)) {
            int with_local_map = 0;
            if (trackLocalMap(cur, with_local_map)) inliers = with_local_map;      // else: the motion-model result stands
            // velocity = T_cur * T_prev^-1
            const Mat3 Rc = quatToRot(cur.pose.q), Rp = quatToRot(m_prev.pose.q);
            Mat3 Rv;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rv.m[r * 3 + c] = Rc.m[r * 3] * Rp.m[c * 3] + Rc.m[r * 3 + 1] * Rp.m[c * 3 + 1] + Rc.m[r * 3 + 2] * Rp.m[c * 3 + 2];
            rotToQuat(Rv, m_velocity.q);
            for (int r = 0; r < 3; ++r) m_velocity.t[r] = cur.pose.t[r] - (Rv.m[r * 3] * m_prev.pose.t[0] + Rv.m[r * 3 + 1] * m_prev.pose.t[1] + Rv.m[r * 3 + 2] * m_prev.pose.t[2]);
            m_haveVelocity = true;
            ++m_framesSinceKeyframe;
            if (m_framesSinceKeyframe >= m_keyframeInterval || inliers < 50) {
                finishMapping();                    // the previous keyframe's solve enters the map before the window moves
                insertKeyframe(cur);
                if (m_stereo && m_loopClosure) detectAndCloseLoop(cur);
                startMapping();
                if (!m_asyncMapping) cur.pose = m_keyframes.back().pose;
            }
            m_prev = std::move(cur);
        } else {
            finishMapping();
            m_haveMonoRef = false;
            m_state = TrackerState::Lost;
            logMessage(LpSlamLogLevel_Info, "VSLAM tracking lost; re-initialising from the next stereo frame");
            m_prev = std::move(cur);
            m_haveVelocity = false;
        }
    }
    m_lastFrameSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (m_state == TrackerState::Tracking) {
        TrackerResult tres = createTrackerResult(m_prev.pose, cam.timestamp);
        tres.timestamp.ros_timestamp = cam.ros_timestamp;
        res.push_back(tres);
    }
    return res;
}

std::size_t HipVslamTrackerBase::mappingGetFeatures(LpSlamMapBoundary, LpSlamFeatureEntry* entry, std::size_t entry_count, LpSlamMatrix9x9 transform)
{
    std::scoped_lock lock(m_slamLock);
    std::size_t copied = 0;
    for (auto const& kv : m_landmarks) {
        if (copied >= entry_count) break;
        const float p[3] = {(float)-kv.second.p[1], (float)kv.second.p[0], (float)kv.second.p[2]};     // optical -> lpslam axes
        entry[copied].position = {transform[0] * p[0] + transform[1] * p[1] + transform[2] * p[2],
                                  transform[3] * p[0] + transform[4] * p[1] + transform[5] * p[2],
                                  transform[6] * p[0] + transform[7] * p[1] + transform[8] * p[2]};
        ++copied;
    }
    return copied;
}

std::size_t HipVslamTrackerBase::mappingGetFeaturesCount(LpSlamMapBoundary)
{
    std::scoped_lock lock(m_slamLock);
    return m_landmarks.size();
}

bool HipVslamTrackerBase::mappingExportCSV(std::string csv_filename)
{
    std::scoped_lock lock(m_slamLock);
    std::ofstream f(csv_filename);
    if (!f) return false;
    for (auto const& kv : m_landmarks) f << kv.first << "," << kv.second.p[0] << "," << kv.second.p[1] << "," << kv.second.p[2] << "\n";
    return true;
}

LpSlamStatus HipVslamTrackerBase::getSlamStatus()
{
    std::scoped_lock lock(m_slamLock);
    LpSlamStatus s{};
    s.localization = LpSlamLocalization_Off; s.fps = 0.0;
    if (!m_ctx) return s;
    switch (m_state) {
    case TrackerState::Tracking: s.localization = LpSlamLocalization_Tracking; break;
    case TrackerState::Lost: s.localization = LpSlamLocalization_Lost; break;
    default: s.localization = LpSlamLocalization_Initializing; break;
    }
    s.frame_time = m_lastFrameSeconds;
    s.key_frames = m_keyframeCount;
    s.feature_points = (long)m_landmarks.size();
    return s;
}

TrackerBase::ProcessImageResult HipStereoTracker::processImage(CameraQueueEntry& cam, std::optional<GlobalStateInTime>, std::optional<GlobalStateInTime>,
                                                               std::vector<SensorQueueEntry> const&)
{
    return trackFrame(cam, true);
}
bool HipStereoTracker::start(SensorQueue&) { return startContext(true); }

TrackerBase::ProcessImageResult HipMonoTracker::processImage(CameraQueueEntry& cam, std::optional<GlobalStateInTime>, std::optional<GlobalStateInTime>,
                                                             std::vector<SensorQueueEntry> const&)
{
    return trackFrame(cam, false);
}
bool HipMonoTracker::start(SensorQueue&) { return startContext(false); }

}  // namespace LpSlam
