// hip_tracker.cpp -- see hip_tracker.h.  Host-side tracking glue around the HIP C ABI (the arithmetic runs on the GPU).
#include "hip_tracker.h"
#include <future>
#include "rectify.h"
#include "two_view.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <limits>


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
    o.optional("device", 0); o.optional("keyframeInterval", 6); o.optional("localWindow", 10); o.optional("asyncMapping", true); o.optional("prefetch", true); o.optional("mapCulling", true); o.optional("mappingReserve", 0);
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
    m_mappingReserve = std::min(16, std::max(0, o.getInteger("mappingReserve")));
    m_prefetch = o.getBool("prefetch"); m_mapCulling = o.getBool("mapCulling");
}

bool HipVslamTrackerBase::startContext(bool stereo)
{
    m_prefetched.valid = false; m_nextFrame = nullptr;
    std::scoped_lock lock(m_slamLock);
    if (m_ctx) return true;
    // configFromFile: the OpenVSLAM configuration as a YAML file instead of the one the adapter generates
    // (src/Trackers/OpenVSLAMTrackerBase.cpp:114-123: a file that cannot be loaded fails the start).  The keys this path has a use for
    // are read -- Camera.{fx, fy, cx, cy, cols, rows, focal_x_baseline}, Feature.{max_num_keypoints, scale_factor, num_levels,
    // ini_fast_threshold, min_fast_threshold}, Initializer.{num_min_triangulated_pts, parallax_deg_threshold}, time_to_relocalize,
    // relocalize_with_nav_data -- nested ("Camera:" + indented "fx: ...") or flat ("Camera.fx: ..."); the rest is ignored.
    std::unordered_map<std::string, std::string> yaml;
    if (!m_configFromFile.empty()) {
        std::ifstream f(m_configFromFile);
        if (!f) { logMessage(LpSlamLogLevel_Error, "Failed to load OpenVSLAM config file " + m_configFromFile); return false; }
        std::string line, section;
        auto trim = [](std::string v) { const size_t a = v.find_first_not_of(" \t\r\"'"), b = v.find_last_not_of(" \t\r\"'"); return a == std::string::npos ? std::string() : v.substr(a, b - a + 1); };
        while (std::getline(f, line)) {
            const size_t hash = line.find('#');
            if (hash != std::string::npos) line.erase(hash);
            if (trim(line).empty() || line[0] == '%' || line.rfind("---", 0) == 0) continue;
            const size_t colon = line.find(':');
            if (colon == std::string::npos) { logMessage(LpSlamLogLevel_Error, "OpenVSLAM config file " + m_configFromFile + ": cannot parse \"" + trim(line) + "\""); return false; }
            const bool indented = line[0] == ' ' || line[0] == '\t';
            const std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
            if (!indented) section.clear();
            if (val.empty()) { if (!indented) section = key; continue; }
            yaml[(indented && !section.empty()) ? section + "." + key : key] = val;
        }
        logMessage(LpSlamLogLevel_Info, "VSLAM config loaded from file " + m_configFromFile + " (" + std::to_string(yaml.size()) + " keys)");
    }
    auto ynum = [&yaml](const char* k, double& out) { auto it = yaml.find(k); if (it == yaml.end()) return false; char* e = nullptr; const double v = std::strtod(it->second.c_str(), &e); if (e == it->second.c_str()) return false; out = v; return true; };
    CameraRegistry* reg = getCameraRegistry();
    if (!reg) { logMessage(LpSlamLogLevel_Error, "Cannot process image without camera registry"); return false; }
    auto left = reg->getConfiguration(0);
    if (!left) { logMessage(LpSlamLogLevel_Error, "Cannot load camera configuration for camera with number 0"); return false; }
    if (stereo && !reg->getConfiguration(1)) { logMessage(LpSlamLogLevel_Error, "Cannot load camera configuration for right camera with number 1"); return false; }
    m_cam = *left;
    if (!yaml.empty()) {
        double v;
        if (ynum("Camera.fx", v)) m_cam.f_x = v;
        if (ynum("Camera.fy", v)) m_cam.f_y = v;
        if (ynum("Camera.cx", v)) m_cam.c_x = v;
        if (ynum("Camera.cy", v)) m_cam.c_y = v;
        if (ynum("Camera.cols", v)) m_cam.resolution_x = (int)v;
        if (ynum("Camera.rows", v)) m_cam.resolution_y = (int)v;
        if (ynum("Camera.focal_x_baseline", v)) m_cam.focal_x_baseline = v;
        if (ynum("Feature.max_num_keypoints", v)) m_slamKeypoints = (int)v;
        if (ynum("Feature.scale_factor", v)) m_scaleFactor = v;
        if (ynum("Feature.num_levels", v)) m_numLevels = (int)v;
        if (ynum("Feature.ini_fast_threshold", v)) m_iniFastThr = (int)v;
        if (ynum("Feature.min_fast_threshold", v)) m_minFastThr = (int)v;
        if (ynum("Initializer.num_min_triangulated_pts", v)) m_initMinTriangulated = (int)v;
        if (ynum("Initializer.parallax_deg_threshold", v)) m_initParallaxDeg = v;
        if (ynum("time_to_relocalize", v)) m_timeToRelocalize = v;
        auto it = yaml.find("relocalize_with_nav_data");
        if (it != yaml.end()) m_relocWithNavigation = it->second == "true" || it->second == "1";
        if (m_numLevels < 1 || m_numLevels > 16 || !(m_scaleFactor > 1.0) || m_slamKeypoints < 1) {
            logMessage(LpSlamLogLevel_Error, "OpenVSLAM config file " + m_configFromFile + ": Feature.* values out of range");
            return false;
        }
    }
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
    cfg.min_fast_threshold = m_minFastThr; cfg.max_images = 6; cfg.device = m_device;
    // a session of the device's pool: when several managers of a process track at once their per-frame launches are shared
    if (lpslam_hip_create_session(&cfg, &m_ctx) != LPSLAM_HIP_OK) {
        logMessage(LpSlamLogLevel_Error, std::string("Cannot create the HIP context: ") + lpslam_hip_last_error());
        m_ctx = nullptr;
        return false;
    }
    // mappingReserve > 0: keep compute units of every XCD free of the front end's workgroups for the mapping thread's solves
    // (lpslam_hip.h).  Off by default: it pays where the front end runs in chip-filling batches (bench.py: +4 % frames/s at 16 frames
    // per launch); this tracker's per-frame launches are small and every one of them is slower on a masked stream (1250 -> 950 frames/s)
    if (m_asyncMapping && m_mappingReserve > 0) (void)lpslam_hip_set_mapping_reserve(m_ctx, m_mappingReserve);
    for (int eye = 0; m_rectify && eye < 2; ++eye)
        if (lpslam_hip_set_rectify_map(m_ctx, eye, maps[eye].map_x.data(), maps[eye].map_y.data()) != LPSLAM_HIP_OK) {
            logMessage(LpSlamLogLevel_Error, std::string("Cannot upload the rectification maps: ") + lpslam_hip_last_error());
            lpslam_hip_destroy(m_ctx); m_ctx = nullptr;
            return false;
        }
    // camera masks (OpenVSLAMTrackerBase::configureMasks, src/Trackers/OpenVSLAMTrackerBase.cpp:331-380): a filled circle around the
    // image centre (mask_type Radial, radius mask_parameter) or camera_mask_left.bmp / camera_mask_right.bmp (mask_type Image)
    for (int eye = 0; eye < (stereo ? 2 : 1); ++eye) {
        auto cfg_eye = reg->getConfiguration((LpSlamCameraNumber)eye);
        if (!cfg_eye || cfg_eye->mask_type == LpSlamCameraMaskType_None) continue;
        std::vector<uint8_t> mask;
        std::string err;
        if (!build_camera_mask(*cfg_eye, eye == 0, mask, &err)) { logMessage(LpSlamLogLevel_Error, "Cannot build the camera mask: " + err); continue; }
        if (lpslam_hip_set_mask(m_ctx, eye, mask.data(), cfg_eye->resolution_x) != LPSLAM_HIP_OK)
            logMessage(LpSlamLogLevel_Error, std::string("Cannot upload the camera mask: ") + lpslam_hip_last_error());
        else logMessage(LpSlamLogLevel_Info, std::string("Camera mask set for the ") + (eye == 0 ? "left" : "right") + " camera");
    }
    m_maxKp = lpslam_hip_max_keypoints_per_image(m_ctx);
    lpslam_hip_level_info(m_ctx, nullptr, nullptr, nullptr, nullptr, m_scales);
    // the DBoW2 vocabulary (reference: src/Trackers/OpenVSLAMTrackerBase.cpp:224-227 refuses to start without the file).  Here a
    // missing file is not fatal: relocalisation and loop detection then pick their candidates by position.
    m_bowDb.clear();
    if (!m_vocabFile.empty()) {
        Vocabulary voc; std::string err;
        std::ifstream probe(m_vocabFile, std::ios::binary);
        if (!probe) logMessage(LpSlamLogLevel_Info, "Vocab file " + m_vocabFile + " not present: place recognition by position instead of bag of words");
        else if (!voc.load(m_vocabFile, err)) { logMessage(LpSlamLogLevel_Error, "Vocab file " + m_vocabFile + ": " + err); lpslam_hip_destroy(m_ctx); m_ctx = nullptr; return false; }
        else if (lpslam_hip_vocab_create(m_ctx, voc.k, voc.L, voc.nodes(), voc.parent.data(), voc.desc.data(), voc.weight.data(), voc.is_leaf.data(), &m_vocab) != LPSLAM_HIP_OK) {
            logMessage(LpSlamLogLevel_Error, std::string("Cannot load the vocabulary onto the device: ") + lpslam_hip_last_error());
            lpslam_hip_destroy(m_ctx); m_ctx = nullptr; return false;
        } else {
            m_bowLevelsUp = voc.L - std::min(2, voc.L - 1);       // the FeatureVector groups at tree level 2 (upstream: L = 6, levels_up = 4)
            int32_t nw = 0;
            lpslam_hip_vocab_info(m_vocab, nullptr, nullptr, nullptr, &nw);
            logMessage(LpSlamLogLevel_Info, "VSLAM vocabulary " + m_vocabFile + ": k=" + std::to_string(voc.k) + " L=" + std::to_string(voc.L) + " nodes=" + std::to_string(voc.nodes()) + " words=" + std::to_string(nw));
        }
    }
    m_stereo = stereo;
    warmUpContext(stereo);
    m_stats = Statistics{};
    m_kfs.clear(); m_landmarks.clear(); m_lmIndex.clear(); m_replaced.clear(); m_freshLandmarks.clear(); m_nextLandmarkId = 0; m_refKf = -1; m_segment = 0; m_segmentStart = 0;
    m_state = TrackerState::NotInitialized;
    m_started = true;
    return true;
}

// The context creates its streams, page-locked staging blocks, pool blocks and completion counters on first use, and the runtime
// loads a kernel's code object at its first launch: 4 ms on the first tracked frame of a session, 2.5 ms on its first keyframe
// (measured, 0.45 ms per frame otherwise).  start() pays that instead: one blank frame goes through every call a tracked frame and a
// keyframe make -- upload / extraction / stereo on both streams, frame read-back, a window match, a pose optimisation, a keyframe
// comparison and a small local bundle adjustment.  Nothing of it reaches the map; every slot it wrote is rewritten by the first frames.
void HipVslamTrackerBase::warmUpContext(bool stereo)
{
    if (!m_ctx) return;
    CameraQueueEntry blank;
    blank.valid = true;
    blank.image.width = m_cam.resolution_x; blank.image.height = m_cam.resolution_y;
    blank.image.pixels.assign((size_t)m_cam.resolution_x * m_cam.resolution_y, 0);
    if (stereo) blank.image_second = blank.image;
    bool ok = frontEnd(blank, stereo, 0);
    if (ok && m_prefetch && lpslam_hip_prefetch_begin(m_ctx) == LPSLAM_HIP_OK) {
        ok = frontEnd(blank, stereo, 2);
        ok = lpslam_hip_prefetch_end(m_ctx) == LPSLAM_HIP_OK && ok;
        ok = lpslam_hip_prefetch_join(m_ctx) == LPSLAM_HIP_OK && ok;
    }
    std::vector<lpslam_hip_keypoint> kp((size_t)m_maxKp);
    std::vector<uint8_t> desc((size_t)m_maxKp * 32);
    std::vector<float> xr((size_t)m_maxKp), dep((size_t)m_maxKp);
    int32_t n = 0;
    if (ok) ok = lpslam_hip_get_frame(m_ctx, 0, kp.data(), desc.data(), stereo ? xr.data() : nullptr, stereo ? dep.data() : nullptr, m_maxKp, &n) == LPSLAM_HIP_OK;
    if (ok) {
        // sixteen queries / observations of a made-up scene: the calls only have to run
        std::vector<lpslam_hip_proj_query> q(16);
        std::vector<uint8_t> qd(16 * 32, 0x5a);
        std::vector<int32_t> mi(16), md(16);
        for (int i = 0; i < 16; ++i) { q[(size_t)i] = lpslam_hip_proj_query{}; q[(size_t)i].x = 40.0f + 30.0f * (float)i; q[(size_t)i].y = 50.0f + 20.0f * (float)i; q[(size_t)i].x_right = -1.0f; q[(size_t)i].radius = 15.0f; q[(size_t)i].min_level = -1; q[(size_t)i].max_level = -1; }
        int32_t found = 0;
        ok = lpslam_hip_match_projection(m_ctx, 0, q.data(), qd.data(), 16, 100, 0.9f, nullptr, stereo ? 1 : 0, mi.data(), md.data(), &found) == LPSLAM_HIP_OK;
        std::vector<int32_t> mq((size_t)m_maxKp), mt((size_t)m_maxKp), mdist((size_t)m_maxKp);
        if (ok) ok = lpslam_hip_match_bf_descriptors(m_ctx, 0, 1, qd.data(), 16, 50, 0.75f, 1, mq.data(), mt.data(), mdist.data(), m_maxKp, &found) == LPSLAM_HIP_OK;
        if (ok && !m_vocab && m_loopClosure) {             // the loop-candidate search's call (and its page-locked block, sized for 48 candidates)
            const int32_t key = -1; int32_t cnt = 0;
            ok = lpslam_hip_desc_store_put(m_ctx, key, qd.data(), 16) == LPSLAM_HIP_OK &&
                 lpslam_hip_match_bf_stored(m_ctx, 0, &key, 1, 50, 0.75f, 1, mq.data(), mt.data(), mdist.data(), m_maxKp, &cnt) == LPSLAM_HIP_OK;
            (void)lpslam_hip_desc_store_drop(m_ctx, key);
        }
    }
    if (ok) {
        const lpslam_hip_ba_camera cam{m_cam.f_x, m_cam.f_y, m_cam.c_x, m_cam.c_y, stereo ? m_cam.focal_x_baseline : 0.0, std::sqrt(5.991), std::sqrt(7.815)};
        // a 4 x 4 grid of landmarks five metres ahead, seen from the origin and from a camera 0.1 m to the side
        std::vector<double> pts, poses = {1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, -0.1, 0, 0};
        std::vector<lpslam_hip_ba_obs> obs;
        for (int i = 0; i < 16; ++i) { pts.push_back(-1.5 + (i % 4)); pts.push_back(-1.5 + (i / 4)); pts.push_back(5.0 + 0.1 * i); }
        for (int p = 0; p < 2; ++p)
            for (int i = 0; i < 16; ++i) {
                const double X = pts[3 * (size_t)i] + poses[7 * (size_t)p + 4], Y = pts[3 * (size_t)i + 1], Z = pts[3 * (size_t)i + 2];
                lpslam_hip_ba_obs o{};
                o.pose = p; o.point = i; o.u = cam.fx * X / Z + cam.cx; o.v = cam.fy * Y / Z + cam.cy; o.ur = stereo ? o.u - cam.focal_x_baseline / Z : -1.0; o.inv_sigma2 = 1.0;
                obs.push_back(o);
            }
        double pose7[7] = {1, 0, 0, 0, 0.01, 0, 0};
        std::vector<uint8_t> outl(32);
        int32_t inl = 0;
        ok = lpslam_hip_pose_optimize(m_ctx, pose7, pts.data(), 16, obs.data(), 16, &cam, outl.data(), &inl) == LPSLAM_HIP_OK;
        const uint8_t fixed[2] = {1, 0};
        lpslam_hip_ba* ba = nullptr;
        if (ok && lpslam_hip_ba_create(m_ctx, poses.data(), fixed, 2, pts.data(), 16, obs.data(), 32, &cam, &ba) == LPSLAM_HIP_OK) {
            ok = lpslam_hip_ba_local(ba, 1, 1, outl.data()) == LPSLAM_HIP_OK && lpslam_hip_ba_get(ba, poses.data(), pts.data()) == LPSLAM_HIP_OK;
            lpslam_hip_ba_destroy(ba);
        }
    }
    if (!ok) logMessage(LpSlamLogLevel_Info, std::string("VSLAM warm-up of the HIP context incomplete: ") + lpslam_hip_last_error());
}

bool HipVslamTrackerBase::stop()
{
    std::scoped_lock lock(m_slamLock);
    stopMappingThread();                               // the mapping thread uses the context
    stopPrefetchThread();
    if (m_vocab) { lpslam_hip_vocab_destroy(m_vocab); m_vocab = nullptr; }
    if (m_ctx) { logStatistics(); lpslam_hip_destroy(m_ctx); m_ctx = nullptr; }
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

int HipVslamTrackerBase::resolve(int id) const
{
    for (int guard = 0; id >= 0 && guard < 64; ++guard) {
        auto it = m_replaced.find(id);
        if (it == m_replaced.end()) break;
        id = it->second;
    }
    return id;
}

// [UPSTREAM] graph_node::get_top_n_covisibilities: the keyframes that share landmarks with `kf`, by weight (number of shared
// landmarks) descending, ties to the newer keyframe; at most top_n with weight >= min_weight, returned in ascending id order
std::vector<int> HipVslamTrackerBase::covisible(int kf, int top_n, int min_weight) const
{
    IdMarks& w = m_marksKf;                              // shared observations per keyframe
    w.begin(m_kfs.size());
    std::vector<int> touched;
    for (int id : m_kfs[(size_t)kf].landmark) {
        const Landmark* l = lm(id);
        if (!l) continue;
        for (auto& o : l->obs) if (o.first != kf) { if (!w.has(o.first)) touched.push_back(o.first); w.at(o.first)++; }
    }
    std::vector<std::pair<int, int>> v;
    for (int k : touched) if (w.get(k) >= min_weight) v.emplace_back(w.get(k), k);
    std::sort(v.begin(), v.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first != b.first ? a.first > b.first : a.second > b.second; });
    if ((int)v.size() > top_n) v.resize((size_t)std::max(top_n, 0));
    std::vector<int> out;
    for (auto& e : v) out.push_back(e.second);
    std::sort(out.begin(), out.end());
    return out;
}

bool HipVslamTrackerBase::initializeMap(FrameData& f, const Pose& at)
{
    // stereo initialisation: every keypoint with a valid depth becomes a landmark once at least
    // Initializer.num_min_triangulated_pts = 40 exist (src/Trackers/OpenVSLAMTrackerBase.cpp:181).  The map that exists stays:
    // after a loss the new keyframes open a new segment at the pose handed in, and a loop closure can join the segments later.
    int n = 0;
    for (size_t i = 0; i < f.kpts.size(); ++i) if (f.depth[i] > 0) ++n;
    if (n < m_initMinTriangulated) return false;
    finishMapping();
    f.pose = at;
    std::fill(f.landmark.begin(), f.landmark.end(), -1);
    m_segmentStart = (int)m_kfs.size();
    insertKeyframe(f);
    return true;
}

// descriptor, viewing direction and valid distance range of the observation that creates a landmark ([UPSTREAM] data::landmark::
// update_normal_and_depth / compute_descriptor, reference observation only)
void HipVslamTrackerBase::initLandmarkView(Landmark& lm, const Pose& pose, const lpslam_hip_keypoint& kp, const uint8_t* desc32) const
{
    const Mat3 R = quatToRot(pose.q);
    // camera centre C = -R^T t
    const double C[3] = {-(R.m[0] * pose.t[0] + R.m[3] * pose.t[1] + R.m[6] * pose.t[2]), -(R.m[1] * pose.t[0] + R.m[4] * pose.t[1] + R.m[7] * pose.t[2]),
                         -(R.m[2] * pose.t[0] + R.m[5] * pose.t[1] + R.m[8] * pose.t[2])};
    const double ray[3] = {lm.p[0] - C[0], lm.p[1] - C[1], lm.p[2] - C[2]};
    const double dist = std::sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
    for (int a = 0; a < 3; ++a) lm.normal[a] = dist > 0 ? ray[a] / dist : 0.0;
    const int lvl = std::min(std::max(kp.octave, 0), m_numLevels - 1);
    lm.max_valid = dist * m_scales[lvl];
    lm.min_valid = lm.max_valid / m_scales[std::max(m_numLevels - 1, 0)];
    std::copy(desc32, desc32 + 32, lm.desc);
}

// the landmark `drop` becomes `keep` everywhere ([UPSTREAM] landmark::replace): its observations move over unless the keyframe
// already sees `keep`; frames in flight follow through m_replaced
void HipVslamTrackerBase::mergeLandmarks(int keep, int drop, FrameData* f)
{
    auto ik = m_landmarks.find(keep), id = m_landmarks.find(drop);
    if (ik == m_landmarks.end() || id == m_landmarks.end() || keep == drop) return;
    for (auto& o : id->second.obs) {
        Keyframe& kf = m_kfs[(size_t)o.first];
        bool sees_keep = false;
        for (auto& ko : ik->second.obs) if (ko.first == o.first) { sees_keep = true; break; }
        if (sees_keep) kf.landmark[(size_t)o.second] = -1;
        else { kf.landmark[(size_t)o.second] = keep; ik->second.obs.push_back(o); }
    }
    unindexLandmark(drop); m_landmarks.erase(id);
    m_replaced[drop] = keep;
    if (f) for (auto& l : f->landmark) if (l == drop) l = keep;
}

void HipVslamTrackerBase::fuseInto(int c, int slot, const std::vector<int>& landmark_ids, FrameData& f, long& added, long& merged)
{
    Keyframe& kc = m_kfs[(size_t)c];
    const Mat3 R = quatToRot(kc.pose.q);
    const double C[3] = {-(R.m[0] * kc.pose.t[0] + R.m[3] * kc.pose.t[1] + R.m[6] * kc.pose.t[2]), -(R.m[1] * kc.pose.t[0] + R.m[4] * kc.pose.t[1] + R.m[7] * kc.pose.t[2]),
                         -(R.m[2] * kc.pose.t[0] + R.m[5] * kc.pose.t[1] + R.m[8] * kc.pose.t[2])};
    const double log_sf = std::log((double)m_scaleFactor);
    std::vector<lpslam_hip_proj_query> q;
    std::vector<uint8_t> qd;
    std::vector<int> q_lm;
    for (int id : landmark_ids) {
        auto it = m_landmarks.find(id);
        if (it == m_landmarks.end()) continue;
        const Landmark& lm = it->second;
        const double* X = lm.p;
        const double pc[3] = {R.m[0] * X[0] + R.m[1] * X[1] + R.m[2] * X[2] + kc.pose.t[0], R.m[3] * X[0] + R.m[4] * X[1] + R.m[5] * X[2] + kc.pose.t[1],
                              R.m[6] * X[0] + R.m[7] * X[1] + R.m[8] * X[2] + kc.pose.t[2]};
        if (!(pc[2] > 0)) continue;
        const double u = m_cam.f_x * pc[0] / pc[2] + m_cam.c_x, v = m_cam.f_y * pc[1] / pc[2] + m_cam.c_y;
        if (u < 0 || v < 0 || u >= m_cam.resolution_x || v >= m_cam.resolution_y) continue;
        const double ray[3] = {X[0] - C[0], X[1] - C[1], X[2] - C[2]};
        const double dist = std::sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
        if (!(dist > 0) || dist < 0.8 * lm.min_valid || dist > 1.2 * lm.max_valid) continue;
        if ((ray[0] * lm.normal[0] + ray[1] * lm.normal[1] + ray[2] * lm.normal[2]) / dist < 0.5) continue;
        const int lvl = std::min(std::max((int)std::ceil(std::log(lm.max_valid / dist) / log_sf), 0), m_numLevels - 1);
        lpslam_hip_proj_query e{};
        e.x = (float)u; e.y = (float)v; e.x_right = m_stereo ? (float)(u - m_cam.focal_x_baseline / pc[2]) : -1.0f;
        e.radius = 3.0f * m_scales[lvl];                 // match::fuse: margin 3 px x scale factor of the predicted level
        e.min_level = std::max(0, lvl - 1); e.max_level = lvl;
        q.push_back(e);
        qd.insert(qd.end(), lm.desc, lm.desc + 32);
        q_lm.push_back(id);
    }
    if (q.empty()) return;
    std::vector<int32_t> idx(q.size()), dist(q.size());
    int32_t n_m = 0;
    if (lpslam_hip_match_fuse(m_ctx, slot, q.data(), qd.data(), (int32_t)q.size(), 50 /* HAMMING_DIST_THR_LOW */, m_stereo ? 1 : 0, idx.data(), dist.data(), &n_m) != LPSLAM_HIP_OK) return;
    for (size_t k = 0; k < q.size(); ++k) {
        if (idx[k] < 0) continue;
        const int kp = idx[k];
        const int id = resolve(q_lm[k]);
        auto it = m_landmarks.find(id);
        if (it == m_landmarks.end()) continue;
        bool seen = false;                               // the keyframe may see this landmark already (through another keypoint)
        for (auto& o : it->second.obs) if (o.first == c) { seen = true; break; }
        const int have = kc.landmark[(size_t)kp];
        if (have < 0) {
            if (seen) continue;
            kc.landmark[(size_t)kp] = id; it->second.obs.emplace_back(c, kp);
            if ((size_t)kp < f.landmark.size()) f.landmark[(size_t)kp] = id;
            ++added;
        } else if (have != id) {
            if (seen) continue;
            auto ih = m_landmarks.find(have);
            if (ih == m_landmarks.end()) continue;
            // the one with more observations stays (ties: the older landmark)
            const bool keep_have = ih->second.obs.size() > it->second.obs.size() || (ih->second.obs.size() == it->second.obs.size() && have < id);
            mergeLandmarks(keep_have ? have : id, keep_have ? id : have, &f);
            for (auto& l : m_prev.landmark) if (l == (keep_have ? id : have)) l = keep_have ? have : id;
            ++merged;
        }
    }
}

// [UPSTREAM] data::keyframe::compute_bow: the BoW vector and the FeatureVector (node per keypoint) of a keyframe, on the device
void HipVslamTrackerBase::computeBow(Keyframe& kf) const
{
    kf.bow.clear(); kf.node.clear();
    if (!m_vocab || kf.kpts.empty()) return;
    const int n = (int)kf.kpts.size();
    std::vector<int32_t> word((size_t)n), node((size_t)n);
    std::vector<float> weight((size_t)n);
    if (lpslam_hip_bow_transform_host(m_ctx, m_vocab, kf.desc.data(), n, m_bowLevelsUp, word.data(), weight.data(), node.data()) != LPSLAM_HIP_OK) return;
    kf.bow = make_bow_vector(word.data(), weight.data(), n);
    kf.node = std::move(node);
}

// the same for the frame being tracked (its descriptors are on the device already)
bool HipVslamTrackerBase::frameNodes(const FrameData& f, std::vector<int32_t>& node, BowVector* bow) const
{
    if (!m_vocab) return false;
    std::vector<int32_t> word((size_t)m_maxKp), nd((size_t)m_maxKp);
    std::vector<float> weight((size_t)m_maxKp);
    int32_t n = 0;
    static const bool trace = getenv("LPSLAM_HIP_MATCH_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (lpslam_hip_bow_transform(m_ctx, m_vocab, f.slot, m_bowLevelsUp, word.data(), weight.data(), nd.data(), m_maxKp, &n) != LPSLAM_HIP_OK) return false;
    if (n != (int32_t)f.kpts.size()) return false;
    const auto t1 = std::chrono::steady_clock::now();
    nd.resize((size_t)n);
    node = std::move(nd);
    if (bow) *bow = make_bow_vector(word.data(), weight.data(), n);
    if (trace) fprintf(stderr, "frame_nodes: device %.1f us, vector %.1f us\n", 1e6 * std::chrono::duration<double>(t1 - t0).count(), 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    return true;
}

int HipVslamTrackerBase::insertKeyframe(FrameData& f)
{
    const int c = (int)m_kfs.size();
    const double baseline = m_cam.focal_x_baseline / m_cam.f_x;
    const double depth_thr = 40.0 * baseline;
    const Mat3 R = quatToRot(f.pose.q);
    Keyframe kf;
    kf.pose = f.pose; kf.segment = m_segment;
    for (auto& l : f.landmark) { l = resolve(l); if (l >= 0 && !m_landmarks.count(l)) l = -1; }
    // new landmarks: unmatched keypoints closer than depth_threshold (40 baselines, OpenVSLAMTrackerBase.cpp:200); when fewer
    // than 100 landmarks would result, the closest ones beyond the threshold are taken too; a segment's first keyframe takes all
    std::vector<std::pair<float, int>> by_depth;
    for (size_t i = 0; i < f.kpts.size(); ++i) if (f.landmark[i] < 0 && f.depth[i] > 0) by_depth.emplace_back(f.depth[i], (int)i);
    std::sort(by_depth.begin(), by_depth.end());
    std::vector<char> create(f.kpts.size(), 0);
    int tracked = 0;
    for (size_t i = 0; i < f.kpts.size(); ++i) tracked += f.landmark[i] >= 0;
    const bool first = c == m_segmentStart;
    for (size_t r = 0; r < by_depth.size(); ++r)
        if (first || by_depth[r].first < depth_thr || tracked + (int)r < 100) create[(size_t)by_depth[r].second] = 1;
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
            lm.ref_kf = c;
            id = m_nextLandmarkId++;
            lm.obs.emplace_back(c, (int)i);
            m_landmarks[id] = std::move(lm); indexLandmark(id);
            m_freshLandmarks.push_back(id);
            f.landmark[i] = id;
        } else if (id >= 0) {
            // (the id was resolved and validated at the top of this function; a landmark that is gone all the same -- erased or merged while
            // the frame was in flight -- must not be re-created by operator[] behind the index's back)
            if (Landmark* l = lm(id)) l->obs.emplace_back(c, (int)i);
            else f.landmark[i] = -1;
        }
    }
    if (!m_stereo && c > m_segmentStart) monoTriangulate(c - 1, kf, f);
    kf.kpts = f.kpts; kf.desc = f.desc; kf.x_right = f.x_right; kf.depth = f.depth; kf.landmark = f.landmark;
    if (m_vocab) {
        static const bool trace = getenv("LPSLAM_HIP_MATCH_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        if (!frameNodes(f, kf.node, &kf.bow)) computeBow(kf);      // the frame's descriptors are still in its slot: no upload
        const auto t1 = std::chrono::steady_clock::now();
        m_bowDb.add(c, kf.bow);
        if (trace) fprintf(stderr, "kf_bow: transform %.1f us, database %.1f us\n", 1e6 * std::chrono::duration<double>(t1 - t0).count(), 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    }
    else if (m_loopClosure) storeDescriptors(c, kf);      // without a vocabulary the loop-candidate search matches descriptors: they stay on the device
    m_kfs.push_back(std::move(kf));
    if (m_mapCulling) {
        cullLandmarks(c);                                 // [UPSTREAM] mapping_module: remove_redundant_landmarks once the new keyframe is stored
        for (size_t i = 0; i < f.landmark.size(); ++i) if (f.landmark[i] >= 0 && !m_landmarks.count(f.landmark[i])) f.landmark[i] = -1;
    }
    // duplicates: the landmarks of the covisible keyframes that this keyframe does not hold are searched in it (match::fuse)
    {
        const std::vector<int> nb = covisible(c, m_localWindow - 1, 15);
        IdMarks& held = m_marksA;
        held.begin((size_t)m_nextLandmarkId);
        for (int id : m_kfs[(size_t)c].landmark) if (id >= 0) held.at(id) = 1;
        std::vector<int> ids;
        for (int k : nb) for (int id : m_kfs[(size_t)k].landmark) if (id >= 0 && !held.has(id)) { held.at(id) = 1; ids.push_back(id); }
        fuseInto(c, f.slot, ids, f, m_stats.fused_added, m_stats.fused_merged);
    }
    m_refKf = c;
    m_refTracked = 0;
    for (int id : m_kfs[(size_t)c].landmark) m_refTracked += id >= 0;
    m_framesSinceKeyframe = 0;
    ++m_stats.keyframes;
    return c;
}

// motion-only pose optimisation ([UPSTREAM] optimize::pose_optimizer) over keypoint <-> landmark associations
namespace {
struct ScopedSeconds {                                   // adds the scope's wall time to a statistics field
    double& acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit ScopedSeconds(double& a) : acc(a) {}
    ~ScopedSeconds() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

bool HipVslamTrackerBase::poseFromMatches(FrameData& cur, const std::vector<int>& cur_idx, const std::vector<int>& lm_ids, const Pose& init, int& n_inliers, int min_inliers)
{
    n_inliers = 0;
    std::vector<double> pts;
    std::vector<lpslam_hip_ba_obs> obs;
    std::vector<int> kept_idx, kept_lm;
    for (size_t k = 0; k < cur_idx.size(); ++k) {
        const Landmark* l = lm(lm_ids[k]);
        if (!l) continue;
        const int i = cur_idx[k];
        const double s = m_scales[cur.kpts[(size_t)i].octave];
        lpslam_hip_ba_obs o{};
        o.pose = 0; o.point = (int32_t)kept_idx.size();
        o.u = cur.kpts[(size_t)i].x; o.v = cur.kpts[(size_t)i].y; o.ur = cur.x_right[(size_t)i] >= 0 ? (double)cur.x_right[(size_t)i] : -1.0; o.inv_sigma2 = 1.0 / (s * s);
        obs.push_back(o);
        pts.insert(pts.end(), l->p, l->p + 3);
        kept_idx.push_back(i);
        kept_lm.push_back(lm_ids[k]);
    }
    if (obs.size() < 10) return false;
    double pose7[7] = {init.q[0], init.q[1], init.q[2], init.q[3], init.t[0], init.t[1], init.t[2]};
    lpslam_hip_ba_camera cam{m_cam.f_x, m_cam.f_y, m_cam.c_x, m_cam.c_y, m_cam.focal_x_baseline, std::sqrt(5.991), std::sqrt(7.815)};
    std::vector<uint8_t> outlier(obs.size());
    int32_t inl = 0;
    // one launch: the whole 4 x 10 iteration flow runs in one workgroup on the device
    ScopedSeconds timed(m_stats.t_dev_pose);
    if (lpslam_hip_pose_optimize(m_ctx, pose7, pts.data(), (int32_t)kept_idx.size(), obs.data(), (int32_t)obs.size(), &cam, outlier.data(), &inl) != LPSLAM_HIP_OK) return false;
    n_inliers = inl;
    if (inl < min_inliers) return false;
    for (int k = 0; k < 4; ++k) cur.pose.q[k] = pose7[k];
    for (int k = 0; k < 3; ++k) cur.pose.t[k] = pose7[4 + k];
    std::fill(cur.landmark.begin(), cur.landmark.end(), -1);
    for (size_t k = 0; k < kept_idx.size(); ++k) cur.landmark[(size_t)kept_idx[k]] = outlier[k] ? -1 : kept_lm[k];   // inliers keep their landmark
    return true;
}

// prediction of the current pose: constant velocity from the last two tracked frames; when that is unknown (first frame of a
// segment, first frame after a loss) the relative motion of the navigation prior, if the host hands one in
// (src/Trackers/OpenVSLAMStereoTracker.cpp:70-179: odometry forwarded with feed_stereo_frame)
bool HipVslamTrackerBase::navDelta(Pose& v) const
{
    if (!(m_forwardNavState && m_navPrev && m_navCur)) return false;
    // v = T_nav_cur * T_nav_prev^-1
    const Mat3 Rc = quatToRot(m_navCur->q), Rp = quatToRot(m_navPrev->q);
    Mat3 Rv;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rv.m[r * 3 + c] = Rc.m[r * 3] * Rp.m[c * 3] + Rc.m[r * 3 + 1] * Rp.m[c * 3 + 1] + Rc.m[r * 3 + 2] * Rp.m[c * 3 + 2];
    rotToQuat(Rv, v.q);
    for (int r = 0; r < 3; ++r) v.t[r] = m_navCur->t[r] - (Rv.m[r * 3] * m_navPrev->t[0] + Rv.m[r * 3 + 1] * m_navPrev->t[1] + Rv.m[r * 3 + 2] * m_navPrev->t[2]);
    return true;
}

void HipVslamTrackerBase::movePose(const Pose& v, const Pose& from, Pose& to)        // to = v * from
{
    double q[4];
    quatMul(v.q, from.q, q);
    const Mat3 Rv = quatToRot(v.q);
    double t[3];
    for (int r = 0; r < 3; ++r) t[r] = Rv.m[r * 3] * from.t[0] + Rv.m[r * 3 + 1] * from.t[1] + Rv.m[r * 3 + 2] * from.t[2] + v.t[r];
    for (int k = 0; k < 4; ++k) to.q[k] = q[k];
    for (int k = 0; k < 3; ++k) to.t[k] = t[k];
}

bool HipVslamTrackerBase::predictedPose(Pose& init) const
{
    Pose v;
    if (m_haveVelocity) v = m_velocity;
    else if (!navDelta(v)) return false;
    movePose(v, m_prev.pose, init);
    return true;
}

// [UPSTREAM] frame_tracker::motion_based_track: pose predicted with the constant-velocity model, the last frame's landmarks are
// projected into the current frame and matched inside a window around the prediction (match::projection::
// match_current_and_last_frames: margin 10 px x scale factor for stereo, doubled once if fewer than 20 matches), matches
// with an inconsistent keypoint rotation are dropped (angle_checker), then the motion-only pose optimiser runs.
bool HipVslamTrackerBase::trackWithMotionModel(FrameData& cur, int& n_inliers)
{
    n_inliers = 0;
    Pose init;
    if (!predictedPose(init)) return false;
    if (!m_haveVelocity) ++m_stats.nav_priors;
    const Mat3 R = quatToRot(init.q);
    const int32_t n_levels = m_numLevels;
    std::vector<lpslam_hip_proj_query> q;
    std::vector<uint8_t> qd;
    std::vector<float> q_angle;
    std::vector<int> q_lm;
    for (size_t i = 0; i < m_prev.kpts.size(); ++i) {
        const int id = resolve(m_prev.landmark[i]);
        if (id < 0) continue;
        const Landmark* l = lm(id);
        if (!l) continue;
        const double* X = l->p;
        const double pc[3] = {R.m[0] * X[0] + R.m[1] * X[1] + R.m[2] * X[2] + init.t[0], R.m[3] * X[0] + R.m[4] * X[1] + R.m[5] * X[2] + init.t[1],
                              R.m[6] * X[0] + R.m[7] * X[1] + R.m[8] * X[2] + init.t[2]};
        if (!(pc[2] > 0)) continue;
        const double u = m_cam.f_x * pc[0] / pc[2] + m_cam.c_x, v = m_cam.f_y * pc[1] / pc[2] + m_cam.c_y;
        if (u < 0 || v < 0 || u >= m_cam.resolution_x || v >= m_cam.resolution_y) continue;
        const int lvl = m_prev.kpts[i].octave;
        lpslam_hip_proj_query e{};
        e.x = (float)u; e.y = (float)v; e.x_right = m_stereo ? (float)(u - m_cam.focal_x_baseline / pc[2]) : -1.0f;
        e.radius = (m_stereo ? 10.0f : 20.0f) * m_scales[lvl];       // match_current_and_last_frames: margin 10 (stereo) / 20 (monocular)
        e.min_level = std::max(0, lvl - 1); e.max_level = std::min(n_levels - 1, lvl + 1);
        q.push_back(e);
        qd.insert(qd.end(), m_prev.desc.begin() + 32 * (long)i, m_prev.desc.begin() + 32 * (long)(i + 1));
        q_angle.push_back(m_prev.kpts[i].angle);
        q_lm.push_back(id);
    }
    if (q.size() < 20) return false;
    std::vector<int32_t> idx(q.size()), dist(q.size());
    std::vector<float> cur_angle(cur.kpts.size());
    for (size_t i = 0; i < cur.kpts.size(); ++i) cur_angle[i] = cur.kpts[i].angle;
    int32_t n_m = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        ScopedSeconds timed(m_stats.t_dev_match);
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

// [UPSTREAM] tracking_module::optimize_current_frame_with_local_map: the landmarks of the local keyframes (the reference keyframe
// and its covisible keyframes) that the frame does not hold yet are projected with the pose just found and searched in a window
// of margin x scale factor of the predicted level (match::projection::match_frame_and_landmarks: margin 5 px for stereo, levels
// [predicted - 1, predicted], Lowe ratio 0.8 between candidates of one level, right-image check); then the motion-only optimiser
// runs again over all associations.
bool HipVslamTrackerBase::trackLocalMap(FrameData& cur, int& n_inliers)
{
    const int32_t n_levels = m_numLevels;
    const double log_sf = std::log((double)m_scaleFactor);
    const Mat3 R = quatToRot(cur.pose.q);
    const double C[3] = {-(R.m[0] * cur.pose.t[0] + R.m[3] * cur.pose.t[1] + R.m[6] * cur.pose.t[2]),
                         -(R.m[1] * cur.pose.t[0] + R.m[4] * cur.pose.t[1] + R.m[7] * cur.pose.t[2]),
                         -(R.m[2] * cur.pose.t[0] + R.m[5] * cur.pose.t[1] + R.m[8] * cur.pose.t[2])};
    std::vector<uint8_t> taken(cur.kpts.size(), 0);
    IdMarks& held = m_marksA;
    held.begin((size_t)m_nextLandmarkId);
    size_t held_on_entry = 0; int n_new = 0;
    for (size_t i = 0; i < cur.kpts.size(); ++i) if (cur.landmark[i] >= 0) {
        taken[i] = 1; ++held_on_entry;
        if (cur.landmark[i] < m_nextLandmarkId) held.at(cur.landmark[i]) = 1;
        Landmark* l = lm(cur.landmark[i]);
        if (l) ++l->n_observable;                        // [UPSTREAM] search_local_landmarks: the frame's own landmarks are observable
    }
    std::vector<lpslam_hip_proj_query> q;
    std::vector<uint8_t> qd;
    std::vector<int> q_lm;
    if (m_refKf < 0) { n_inliers = (int)held_on_entry; return true; }
    std::vector<int> local = covisible(m_refKf, m_localWindow - 1, 15);
    local.push_back(m_refKf);
    std::sort(local.begin(), local.end());
    for (int kfi : local) {                              // local landmarks in keyframe / keypoint order (deterministic)
        for (int lid : m_kfs[(size_t)kfi].landmark) {
            if (lid < 0 || held.has(lid)) continue;
            held.at(lid) = 1;
            Landmark* lp = this->lm(lid);
            if (!lp) continue;
            Landmark& lm = *lp;
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
            ++lm.n_observable;                           // in the frustum at a plausible scale: tracking expects to find it
            lpslam_hip_proj_query e{};
            e.x = (float)u; e.y = (float)v; e.x_right = m_stereo ? (float)(u - m_cam.focal_x_baseline / pc[2]) : -1.0f;
            e.radius = 5.0f * m_scales[lvl];
            e.min_level = std::max(0, lvl - 1); e.max_level = lvl;
            q.push_back(e);
            qd.insert(qd.end(), lm.desc, lm.desc + 32);
            q_lm.push_back(lid);
        }
    }
    if (!q.empty()) {
        std::vector<int32_t> idx(q.size()), dist(q.size());
        int32_t n_m = 0;
        ScopedSeconds timed(m_stats.t_dev_match);
        if (lpslam_hip_match_projection(m_ctx, cur.slot, q.data(), qd.data(), (int32_t)q.size(), 100 /* HAMMING_DIST_THR_HIGH */, 0.8f, taken.data(), m_stereo ? 1 : 0,
                                        idx.data(), dist.data(), &n_m) != LPSLAM_HIP_OK) return false;
        for (size_t k = 0; k < q.size(); ++k) if (idx[k] >= 0) { cur.landmark[(size_t)idx[k]] = q_lm[k]; ++n_new; }
    }
    m_stats.local_map_joined += n_new;
    if (n_new == 0) { n_inliers = (int)held_on_entry; return true; }      // nothing joined: the pose found from the same associations stands
    std::vector<int> cur_idx, lm_ids;
    for (size_t i = 0; i < cur.kpts.size(); ++i) if (cur.landmark[i] >= 0) { cur_idx.push_back((int)i); lm_ids.push_back(cur.landmark[i]); }
    const Pose init = cur.pose;
    std::vector<int> before = cur.landmark;
    for (size_t i = 0; i < cur.kpts.size(); ++i) if (!taken[i]) before[i] = -1;    // what the frame held on entry
    if (poseFromMatches(cur, cur_idx, lm_ids, init, n_inliers)) return true;
    cur.pose = init; cur.landmark = before;
    return false;
}

bool HipVslamTrackerBase::trackAgainstPrevious(FrameData& cur, int& n_inliers)
{
    // motion model first; descriptor matching against the whole previous frame is the fallback (upstream falls back to
    // BoW / robust matching, frame_tracker::bow_match_based_track / robust_match_based_track)
    if (trackWithMotionModel(cur, n_inliers)) { ++m_stats.motion_tracked; return true; }
    n_inliers = 0;
    if (lpslam_hip_match_bf(m_ctx, cur.slot, m_prev.slot) != LPSLAM_HIP_OK) return false;
    std::vector<int32_t> mq((size_t)m_maxKp), mt((size_t)m_maxKp), md((size_t)m_maxKp);
    int32_t nm = 0;
    // HAMMING_DIST_THR_LOW = 50, Lowe ratio 0.9, mutual best
    if (lpslam_hip_get_bf_matches(m_ctx, cur.slot, m_prev.slot, 50, 0.9f, 1, mq.data(), mt.data(), md.data(), m_maxKp, &nm) != LPSLAM_HIP_OK) return false;
    std::vector<int> cur_idx, lm_ids;
    for (int k = 0; k < nm; ++k) {
        const int id = resolve(m_prev.landmark[(size_t)mt[k]]);
        if (id < 0) continue;
        cur_idx.push_back(mq[k]); lm_ids.push_back(id);
    }
    Pose init = m_prev.pose;
    (void)predictedPose(init);                           // constant velocity / navigation prior, else the previous pose
    if (!poseFromMatches(cur, cur_idx, lm_ids, init, n_inliers)) return false;
    ++m_stats.bf_tracked;
    return true;
}

// [UPSTREAM] keyframe_inserter::new_keyframe_is_needed, reduced to what this tracker knows: the interval, too few tracked
// landmarks in absolute terms or against what the reference keyframe held when it was inserted
bool HipVslamTrackerBase::keyframeNeeded(int inliers) const
{
    if (m_framesSinceKeyframe >= m_keyframeInterval) return true;
    if (inliers < 50) return true;
    if (m_refTracked > 0 && inliers < m_refTracked / 4) return true;
    if (m_framesSinceKeyframe >= std::max(1, m_keyframeInterval / 2) && m_refTracked > 0 && 10 * inliers < 6 * m_refTracked) return true;
    return false;
}

// ---- map maintenance ([UPSTREAM] module::local_map_cleaner) ---------------------------------------------------------------------
// What keeps a long session bounded: landmarks that tracking keeps missing or that never gained a second (stereo) / third
// (monocular) keyframe are dropped two keyframes after their creation; a keyframe 90 % of whose landmarks are seen by three other
// keyframes at the same or a finer scale is redundant and leaves the map.  The reference reports the resulting counts
// (/root/reference/src/Trackers/OpenVSLAMTrackerBase.cpp:438-452).
void HipVslamTrackerBase::eraseLandmark(int id)
{
    auto it = m_landmarks.find(id);
    if (it == m_landmarks.end()) return;
    for (auto& o : it->second.obs) {
        Keyframe& kf = m_kfs[(size_t)o.first];
        if ((size_t)o.second < kf.landmark.size() && kf.landmark[(size_t)o.second] == id) kf.landmark[(size_t)o.second] = -1;
    }
    unindexLandmark(id); m_landmarks.erase(it);
    ++m_stats.culled_landmarks;
}

void HipVslamTrackerBase::cullLandmarks(int cur_kf)
{
    const int num_obs_thr = m_stereo ? 3 : 2;
    std::vector<int> keep;
    for (int id : m_freshLandmarks) {
        auto it = m_landmarks.find(id);
        if (it == m_landmarks.end()) continue;                          // merged away or erased already
        const Landmark& lm = it->second;
        int n_obs = 0;
        for (auto& o : lm.obs) { const Keyframe& kf = m_kfs[(size_t)o.first]; n_obs += (!kf.x_right.empty() && kf.x_right[(size_t)o.second] >= 0) ? 2 : 1; }
        if ((double)lm.n_observed / (double)lm.n_observable < 0.3) eraseLandmark(id);
        else if (lm.ref_kf + 2 <= cur_kf && n_obs <= num_obs_thr) eraseLandmark(id);
        else if (lm.ref_kf + 3 <= cur_kf) continue;                   // reliable from now on
        else keep.push_back(id);
    }
    m_freshLandmarks.swap(keep);
}

void HipVslamTrackerBase::cullKeyframes(int cur_kf)
{
    if (!m_mapCulling || cur_kf < 0 || cur_kf >= (int)m_kfs.size()) return;
    const double depth_thr = m_stereo ? 40.0 * m_cam.focal_x_baseline / m_cam.f_x : 0.0;
    for (int k : covisible(cur_kf, (int)m_kfs.size(), 15)) {
        Keyframe& kf = m_kfs[(size_t)k];
        if (kf.erased || k == 0 || k == m_refKf || m_kfs[(size_t)k - 1].segment != kf.segment) continue;      // the origin of a segment stays
        int n_valid = 0, n_redundant = 0;
        for (size_t i = 0; i < kf.landmark.size(); ++i) {
            const int id = kf.landmark[i];
            if (id < 0) continue;
            auto it = m_landmarks.find(id);
            if (it == m_landmarks.end()) continue;
            if (m_stereo && (kf.depth[i] > depth_thr || kf.depth[i] < 0)) continue;
            ++n_valid;
            int n_obs = 0;
            for (auto& o : it->second.obs) { const Keyframe& ko = m_kfs[(size_t)o.first]; n_obs += (!ko.x_right.empty() && ko.x_right[(size_t)o.second] >= 0) ? 2 : 1; }
            if (n_obs <= 3) continue;
            const int level = kf.kpts[i].octave;
            int better = 0;
            for (auto& o : it->second.obs) {
                if (o.first == k) continue;
                if (m_kfs[(size_t)o.first].kpts[(size_t)o.second].octave <= level + 1 && ++better >= 3) break;
            }
            if (better >= 3) ++n_redundant;
        }
        if (n_valid == 0 || (double)n_redundant < 0.9 * (double)n_valid) continue;
        // erase: its observations leave the landmarks (a landmark left with two observation counts or fewer goes with it)
        for (size_t i = 0; i < kf.landmark.size(); ++i) {
            const int id = kf.landmark[i];
            if (id < 0) continue;
            kf.landmark[i] = -1;
            auto it = m_landmarks.find(id);
            if (it == m_landmarks.end()) continue;
            auto& ob = it->second.obs;
            for (size_t o = 0; o < ob.size(); ++o) if (ob[o].first == k && ob[o].second == (int)i) { ob.erase(ob.begin() + (long)o); break; }
            int n_obs = 0;
            for (auto& o : ob) { const Keyframe& ko = m_kfs[(size_t)o.first]; n_obs += (!ko.x_right.empty() && ko.x_right[(size_t)o.second] >= 0) ? 2 : 1; }
            if (n_obs <= 2) eraseLandmark(id);
        }
        kf.erased = true;
        (void)lpslam_hip_desc_store_drop(m_ctx, k); kf.desc_on_device = false;
        kf.kpts.clear(); kf.kpts.shrink_to_fit(); kf.desc.clear(); kf.desc.shrink_to_fit(); kf.x_right.clear(); kf.depth.clear(); kf.landmark.clear(); kf.node.clear(); kf.bow.clear();
        m_bowDb.remove(k);
        ++m_stats.culled_keyframes;
    }
}

// a bundle-adjustment problem over the given keyframes: landmarks seen by the free keyframes, observed at least twice among all
std::unique_ptr<HipVslamTrackerBase::MappingJob> HipVslamTrackerBase::prepareBundle(const std::vector<int>& free_kfs, const std::vector<int>& fixed_kfs)
{
    auto job = std::make_unique<MappingJob>();
    std::vector<int> all = free_kfs;
    all.insert(all.end(), fixed_kfs.begin(), fixed_kfs.end());
    std::sort(all.begin(), all.end());
    std::unordered_map<int, int> kf_index, fixed_set;
    for (int k : fixed_kfs) fixed_set[k] = 1;
    for (size_t i = 0; i < all.size(); ++i) kf_index[all[i]] = (int)i;
    IdMarks &seen = m_marksA, &index = m_marksB, &of_free = m_marksC;        // index: BA point + 1 of a landmark
    seen.begin((size_t)m_nextLandmarkId); index.begin((size_t)m_nextLandmarkId); of_free.begin((size_t)m_nextLandmarkId);
    for (int k : free_kfs) for (int id : m_kfs[(size_t)k].landmark) if (id >= 0) of_free.at(id) = 1;
    for (int k : all) for (int id : m_kfs[(size_t)k].landmark) if (id >= 0 && of_free.has(id)) seen.at(id)++;
    for (int k : all)                                    // point order: first appearance in keyframe / keypoint order
        for (int id : m_kfs[(size_t)k].landmark) {
            if (id < 0 || index.has(id)) continue;
            if (seen.get(id) < 2) continue;
            const Landmark* l = lm(id);
            if (!l) continue;
            index.at(id) = (int)job->ids.size() + 1; job->ids.push_back(id);
            job->pts.insert(job->pts.end(), l->p, l->p + 3);
        }
    if (job->ids.size() < 20 || all.size() < 2) return nullptr;
    job->kfs = all;
    bool any_fixed = false;
    for (size_t f = 0; f < all.size(); ++f) {
        const Keyframe& kf = m_kfs[(size_t)all[f]];
        const Pose& p = kf.pose;
        job->poses.insert(job->poses.end(), {p.q[0], p.q[1], p.q[2], p.q[3], p.t[0], p.t[1], p.t[2]});
        bool fx = fixed_set.count(all[f]) != 0;
        // the first keyframe of a segment anchors the gauge
        if (all[f] == 0 || m_kfs[(size_t)all[f] - 1].segment != kf.segment) fx = true;
        job->fixed.push_back(fx ? 1 : 0);
        any_fixed = any_fixed || fx;
        for (size_t k = 0; k < kf.landmark.size(); ++k) {
            if (kf.landmark[k] < 0) continue;
            if (!index.has(kf.landmark[k])) continue;
            const double s = m_scales[kf.kpts[k].octave];
            const double ur = (!kf.x_right.empty() && kf.x_right[k] >= 0) ? (double)kf.x_right[k] : -1.0;
            job->obs.push_back({(int32_t)f, index.get(kf.landmark[k]) - 1, kf.kpts[k].x, kf.kpts[k].y, ur, 1.0 / (s * s)});
            job->origin.emplace_back(all[f], (int)k);
        }
    }
    if (!any_fixed) job->fixed[0] = 1;                   // no anchor in the set: the oldest keyframe holds the gauge
    job->outlier.assign(job->obs.size(), 0);
    return job;
}

std::unique_ptr<HipVslamTrackerBase::MappingJob> HipVslamTrackerBase::prepareMapping(int c)
{
    if (!m_enableMapping || m_kfs.size() < 2 || c < 0) return nullptr;
    // [UPSTREAM] local_bundle_adjuster: the keyframe and its covisible keyframes move, every other keyframe that sees one of
    // their landmarks is held fixed (here: the localWindow keyframes with the most such observations)
    std::vector<int> local = covisible(c, m_localWindow - 1, 15);
    local.push_back(c);
    std::sort(local.begin(), local.end());
    IdMarks &cnt = m_marksKf, &done = m_marksA;          // cnt: -1 for a local keyframe, else its shared observations
    cnt.begin(m_kfs.size()); done.begin((size_t)m_nextLandmarkId);
    for (int k : local) cnt.at(k) = -1;
    std::vector<int> touched;
    for (int k : local)
        for (int id : m_kfs[(size_t)k].landmark) {
            if (id < 0 || done.has(id)) continue;
            done.at(id) = 1;
            const Landmark* l = lm(id);
            if (!l) continue;
            for (auto& o : l->obs) {
                if (cnt.get(o.first) < 0) continue;
                if (!cnt.has(o.first)) touched.push_back(o.first);
                cnt.at(o.first)++;
            }
        }
    std::vector<std::pair<int, int>> v;
    for (int k : touched) v.emplace_back(cnt.get(k), k);
    std::sort(v.begin(), v.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first != b.first ? a.first > b.first : a.second > b.second; });
    if ((int)v.size() > m_localWindow) v.resize((size_t)m_localWindow);
    std::vector<int> fixed;
    for (auto& e : v) fixed.push_back(e.second);
    auto job = prepareBundle(local, fixed);
    if (job) job->keyframe = c;
    return job;
}

// runs on the mapping thread: touches the job and the GPU only
void HipVslamTrackerBase::solveMapping(MappingJob& job) const
{
    lpslam_hip_ba_camera cam{m_cam.f_x, m_cam.f_y, m_cam.c_x, m_cam.c_y, m_stereo ? m_cam.focal_x_baseline : 0.0, std::sqrt(5.991), std::sqrt(7.815)};
    lpslam_hip_ba* ba = nullptr;
    if (!job.global) {
        // the keyframe's window: created, solved (5 robust + 10 plain iterations around the outlier classification), read back and destroyed
        // by one call -- which the windows of the other sessions' mapping threads join (one launch chain for all of them)
        job.solved = lpslam_hip_ba_local_window(m_ctx, job.poses.data(), job.fixed.data(), (int32_t)job.kfs.size(), job.pts.data(), (int32_t)job.ids.size(), job.obs.data(),
                                                (int32_t)job.obs.size(), &cam, 5, 10, job.outlier.data()) == LPSLAM_HIP_OK;
        return;
    }
    if (lpslam_hip_ba_create(m_ctx, job.poses.data(), job.fixed.data(), (int32_t)job.kfs.size(), job.pts.data(), (int32_t)job.ids.size(), job.obs.data(),
                             (int32_t)job.obs.size(), &cam, &ba) != LPSLAM_HIP_OK) return;
    if (job.global) {
        int32_t done = 0;
        job.solved = lpslam_hip_ba_optimize(ba, 1, 10, nullptr, &done) == LPSLAM_HIP_OK && lpslam_hip_ba_get(ba, job.poses.data(), job.pts.data()) == LPSLAM_HIP_OK;
    } else {
        job.solved = lpslam_hip_ba_local(ba, 5, 10, job.outlier.data()) == LPSLAM_HIP_OK &&
                     lpslam_hip_ba_get(ba, job.poses.data(), job.pts.data()) == LPSLAM_HIP_OK;
    }
    lpslam_hip_ba_destroy(ba);
}

void HipVslamTrackerBase::applyMapping(const MappingJob& job)
{
    if (!job.solved) return;
    for (size_t f = 0; f < job.kfs.size(); ++f) {
        if (job.fixed[f]) continue;
        Pose& p = m_kfs[(size_t)job.kfs[f]].pose;
        for (int k = 0; k < 4; ++k) p.q[k] = job.poses[7 * f + (size_t)k];
        for (int k = 0; k < 3; ++k) p.t[k] = job.poses[7 * f + 4 + (size_t)k];
    }
    for (size_t j = 0; j < job.ids.size(); ++j) {
        auto it = m_landmarks.find(resolve(job.ids[j]));
        if (it != m_landmarks.end()) { it->second.p[0] = job.pts[3 * j]; it->second.p[1] = job.pts[3 * j + 1]; it->second.p[2] = job.pts[3 * j + 2]; }
    }
    // outlier observations leave the map ([UPSTREAM] local_bundle_adjuster: erase_observation on both sides)
    for (size_t k = 0; k < job.obs.size(); ++k) {
        if (!job.outlier[k]) continue;
        Keyframe& kf = m_kfs[(size_t)job.origin[k].first];
        const int kp = job.origin[k].second;
        const int id = kf.landmark[(size_t)kp];
        if (id < 0 || id != resolve(job.ids[(size_t)job.obs[k].point])) continue;        // changed since the problem was copied
        kf.landmark[(size_t)kp] = -1;
        auto it = m_landmarks.find(id);
        if (it == m_landmarks.end()) continue;
        auto& ob = it->second.obs;
        for (size_t o = 0; o < ob.size(); ++o) if (ob[o].first == job.origin[k].first && ob[o].second == kp) { ob.erase(ob.begin() + (long)o); break; }
        if (ob.empty()) { unindexLandmark(id); m_landmarks.erase(it); }
    }
    ++m_stats.local_ba;
    if (!job.global && job.keyframe >= 0) cullKeyframes(job.keyframe);      // [UPSTREAM] mapping_module: remove_redundant_keyframes after the local BA
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
        for (int i = 0; i < n_cur; ++i) { m_monoPrevMatched[2 * (size_t)i] = cur.kpts[(size_t)i].x; m_monoPrevMatched[2 * (size_t)i + 1] = cur.kpts[(size_t)i].y; }
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
        qd.insert(qd.end(), ref.desc.begin() + 32 * (long)i, ref.desc.begin() + 32 * (long)(i + 1));
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
        m_monoPrevMatched[2 * (size_t)q_ref[k]] = cur.kpts[(size_t)idx[k]].x; m_monoPrevMatched[2 * (size_t)q_ref[k] + 1] = cur.kpts[(size_t)idx[k]].y;
    }
    std::vector<float> kr(2 * ref.kpts.size()), kc(2 * cur.kpts.size());
    for (size_t i = 0; i < ref.kpts.size(); ++i) { kr[2 * i] = ref.kpts[i].x; kr[2 * i + 1] = ref.kpts[i].y; }
    for (size_t i = 0; i < cur.kpts.size(); ++i) { kc[2 * i] = cur.kpts[i].x; kc[2 * i + 1] = cur.kpts[i].y; }
    const double K[4] = {m_cam.f_x, m_cam.f_y, m_cam.c_x, m_cam.c_y};
    TwoViewParams prm;
    prm.min_triangulated = m_initMinTriangulated;       // the reference's Initializer.* values (src/Trackers/OpenVSLAMTrackerBase.cpp:181-182)
    prm.parallax_deg_thr = m_initParallaxDeg;
    TwoViewResult tv;
    if (!two_view_initialize(K, kr.data(), kc.data(), matches.data(), (int)(matches.size() / 2), prm, tv)) return false;

    // ---- the initial map: reference keyframe at the last good pose (the origin for the first segment), current keyframe at (R, t)
    // relative to it, scale: median depth in the reference = 1
    std::vector<double> depths;
    for (size_t m = 0; m < tv.triangulated.size(); ++m) if (tv.triangulated[m]) depths.push_back(tv.points[3 * m + 2]);
    if ((int)depths.size() < m_initMinTriangulated) return false;
    std::nth_element(depths.begin(), depths.begin() + (long)(depths.size() / 2), depths.end());
    const double median = depths[depths.size() / 2];
    if (!(median > 0)) return false;
    const double inv = 1.0 / median;
    finishMapping();
    m_segmentStart = (int)m_kfs.size();
    FrameData reff = ref;
    reff.pose = Pose();
    reff.landmark.assign(reff.kpts.size(), -1);
    Mat3 Rm; std::copy(tv.R, tv.R + 9, Rm.m);
    rotToQuat(Rm, cur.pose.q);
    for (int a = 0; a < 3; ++a) cur.pose.t[a] = tv.t[a] * inv;
    cur.landmark.assign(cur.kpts.size(), -1);
    Keyframe k0, k1;
    const int i0 = m_segmentStart, i1 = m_segmentStart + 1;
    k0.pose = reff.pose; k1.pose = cur.pose; k0.segment = k1.segment = m_segment;
    for (size_t m = 0; m < tv.triangulated.size(); ++m) {
        if (!tv.triangulated[m]) continue;
        const int ir = matches[2 * m], ic = matches[2 * m + 1];
        Landmark lm;
        for (int a = 0; a < 3; ++a) lm.p[a] = tv.points[3 * m + (size_t)a] * inv;
        initLandmarkView(lm, reff.pose, reff.kpts[(size_t)ir], reff.desc.data() + 32 * (size_t)ir);
        lm.ref_kf = i0;
        lm.obs.emplace_back(i0, ir); lm.obs.emplace_back(i1, ic);
        const int id = m_nextLandmarkId++;
        m_landmarks[id] = std::move(lm); indexLandmark(id);
        reff.landmark[(size_t)ir] = id; cur.landmark[(size_t)ic] = id;
    }
    k0.kpts = reff.kpts; k0.desc = reff.desc; k0.landmark = reff.landmark; k0.x_right.assign(reff.kpts.size(), -1.0f); k0.depth.assign(reff.kpts.size(), -1.0f);
    k1.kpts = cur.kpts; k1.desc = cur.desc; k1.landmark = cur.landmark; k1.x_right.assign(cur.kpts.size(), -1.0f); k1.depth.assign(cur.kpts.size(), -1.0f);
    if (m_vocab) { computeBow(k0); m_bowDb.add((int)m_kfs.size(), k0.bow); computeBow(k1); m_bowDb.add((int)m_kfs.size() + 1, k1.bow); }
    else if (m_loopClosure) {
        storeDescriptors((int)m_kfs.size(), k0);
        storeDescriptors((int)m_kfs.size() + 1, k1);
    }
    m_kfs.push_back(std::move(k0)); m_kfs.push_back(std::move(k1));
    m_stats.keyframes += 2;
    m_framesSinceKeyframe = 0;
    m_refKf = i1;
    m_refTracked = 0;
    for (int id : m_kfs[(size_t)i1].landmark) m_refTracked += id >= 0;
    // global bundle adjustment of the two-keyframe map (20 iterations, Huber), inline: nothing can be tracked before it
    {
        auto job = prepareBundle({i0, i1}, {});
        if (job) {
            lpslam_hip_ba_camera cam{m_cam.f_x, m_cam.f_y, m_cam.c_x, m_cam.c_y, 0.0, std::sqrt(5.991), std::sqrt(7.815)};
            lpslam_hip_ba* ba = nullptr;
            if (lpslam_hip_ba_create(m_ctx, job->poses.data(), job->fixed.data(), (int32_t)job->kfs.size(), job->pts.data(), (int32_t)job->ids.size(), job->obs.data(),
                                     (int32_t)job->obs.size(), &cam, &ba) == LPSLAM_HIP_OK) {
                int32_t done = 0;
                job->solved = lpslam_hip_ba_optimize(ba, 1, 20, nullptr, &done) == LPSLAM_HIP_OK && lpslam_hip_ba_get(ba, job->poses.data(), job->pts.data()) == LPSLAM_HIP_OK;
                lpslam_hip_ba_destroy(ba);
                if (job->solved) applyMapping(*job);
            }
        }
    }
    cur.pose = m_kfs.back().pose;
    m_haveMonoRef = false;
    size_t n_seg = 0;
    for (int id : m_kfs[(size_t)i1].landmark) n_seg += id >= 0;
    logMessage(LpSlamLogLevel_Info, "VSLAM monocular map initialised: " + std::to_string(n_seg) + " landmarks, model " + (tv.model == 0 ? "H" : "F"));
    return (int)n_seg >= m_initMinTriangulated;
}

// New landmarks for a monocular keyframe: keypoints without a landmark are matched by descriptor against the previous keyframe's
// (brute force on the device, mutual best, ratio 0.8), kept when they satisfy the epipolar constraint of the two poses, and
// triangulated; the checks are upstream's (positive depth in both views, reprojection chi2 <= 5.991 per view, parallax, scale
// consistency of the distances with the pyramid levels).  `kf` is the keyframe being inserted (id = m_kfs.size()).
void HipVslamTrackerBase::monoTriangulate(int prev_kf, Keyframe& kf, FrameData& f)
{
    Keyframe& prev = m_kfs[(size_t)prev_kf];
    const int c = (int)m_kfs.size();
    if (prev.kpts.empty() || f.kpts.empty()) return;
    const int scratch = f.slot ^ 1;                      // monocular frames use slots 0 / 2
    std::vector<int32_t> mq((size_t)m_maxKp), mt((size_t)m_maxKp), md((size_t)m_maxKp);
    int32_t nm = 0;
    if (lpslam_hip_match_bf_descriptors(m_ctx, f.slot, scratch, prev.desc.data(), (int32_t)prev.kpts.size(), 50, 0.8f, 1, mq.data(), mt.data(), md.data(), m_maxKp, &nm) != LPSLAM_HIP_OK) return;
    const float* scales = m_scales;
    const double fx = m_cam.f_x, fy = m_cam.f_y, cx = m_cam.c_x, cy = m_cam.c_y;
    const Mat3 R1 = quatToRot(prev.pose.q), R2 = quatToRot(f.pose.q);
    auto proj = [&](const Mat3& R, const double* t, double* P) {
        for (int c2 = 0; c2 < 3; ++c2) { P[c2] = fx * R.m[c2] + cx * R.m[6 + c2]; P[4 + c2] = fy * R.m[3 + c2] + cy * R.m[6 + c2]; P[8 + c2] = R.m[6 + c2]; }
        P[3] = fx * t[0] + cx * t[2]; P[7] = fy * t[1] + cy * t[2]; P[11] = t[2];
    };
    double P1[12], P2[12];
    proj(R1, prev.pose.t, P1); proj(R2, f.pose.t, P2);
    auto centre = [](const Mat3& R, const double* t, double* C) { for (int a = 0; a < 3; ++a) C[a] = -(R.m[a] * t[0] + R.m[3 + a] * t[1] + R.m[6 + a] * t[2]); };
    double C1[3], C2[3];
    centre(R1, prev.pose.t, C1); centre(R2, f.pose.t, C2);
    // relative pose prev -> cur and the fundamental matrix x2^T F x1 = 0
    double R21[9], t21[3];
    for (int r = 0; r < 3; ++r) for (int c2 = 0; c2 < 3; ++c2) R21[r * 3 + c2] = R2.m[r * 3] * R1.m[c2 * 3] + R2.m[r * 3 + 1] * R1.m[c2 * 3 + 1] + R2.m[r * 3 + 2] * R1.m[c2 * 3 + 2];
    for (int r = 0; r < 3; ++r) t21[r] = f.pose.t[r] - (R21[r * 3] * prev.pose.t[0] + R21[r * 3 + 1] * prev.pose.t[1] + R21[r * 3 + 2] * prev.pose.t[2]);
    const double tx[9] = {0, -t21[2], t21[1], t21[2], 0, -t21[0], -t21[1], t21[0], 0};
    double E[9];
    for (int r = 0; r < 3; ++r) for (int c2 = 0; c2 < 3; ++c2) E[r * 3 + c2] = tx[r * 3] * R21[c2] + tx[r * 3 + 1] * R21[3 + c2] + tx[r * 3 + 2] * R21[6 + c2];
    const double ratio_factor = 1.5 * m_scaleFactor;
    for (int k = 0; k < nm; ++k) {
        const int ic = mq[k], ip = mt[k];
        if (f.landmark[(size_t)ic] >= 0 || prev.landmark[(size_t)ip] >= 0) continue;
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
        lm.ref_kf = c;
        lm.obs.emplace_back(prev_kf, ip); lm.obs.emplace_back(c, ic);
        const int id = m_nextLandmarkId++;
        m_landmarks[id] = std::move(lm); indexLandmark(id);
        m_freshLandmarks.push_back(id);
        f.landmark[(size_t)ic] = id; prev.landmark[(size_t)ip] = id;
    }
    (void)kf;
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
void pose_centre(const double* q, const double* t, double* C)
{
    const Mat3 R = quatToRot(q);
    for (int a = 0; a < 3; ++a) C[a] = -(R.m[a] * t[0] + R.m[3 + a] * t[1] + R.m[6 + a] * t[2]);
}
}  // namespace

// ---- relocalisation ([UPSTREAM] module::relocalizer, reached while the tracker state is Lost) -----------------------------------
// Candidates: the keyframes nearest to the last pose that was tracked (the reference picks them from the BoW database; with
// relocWithNavigation the navigation prior moves that pose along).  Each candidate's descriptors are matched against the frame
// on the device (mutual best, Hamming <= 50, ratio 0.75); the matched keypoints' landmarks and the candidate's pose start the
// motion-only optimiser, and 30 inliers bring the tracker back.
bool HipVslamTrackerBase::relocalise(FrameData& cur)
{
    if (m_kfs.empty()) return false;
    double Cl[3];
    pose_centre(m_lastGoodPose.q, m_lastGoodPose.t, Cl);
    std::vector<std::pair<double, int>> near;
    for (size_t k = 0; k < m_kfs.size(); ++k) {
        if (m_kfs[k].erased) continue;
        double C[3];
        pose_centre(m_kfs[k].pose.q, m_kfs[k].pose.t, C);
        near.emplace_back(std::sqrt((C[0] - Cl[0]) * (C[0] - Cl[0]) + (C[1] - Cl[1]) * (C[1] - Cl[1]) + (C[2] - Cl[2]) * (C[2] - Cl[2])), (int)k);
    }
    std::sort(near.begin(), near.end());
    if (near.size() > 8) near.resize(8);
    // with a vocabulary the candidates come from the BoW database instead ([UPSTREAM] bow_database::acquire_relocalization_candidates):
    // the keyframes that share the most words with the frame, best L1 score first -- wherever they are in the map
    std::vector<int32_t> cur_node;
    BowVector cur_bow;
    const bool use_bow = m_vocab && !cur.kpts.empty() && frameNodes(cur, cur_node, &cur_bow);
    if (use_bow) {
        near.clear();
        for (auto& sc : m_bowDb.query(cur_bow, {}, -1.0, std::numeric_limits<int>::max())) { near.emplace_back(-sc.first, sc.second); if (near.size() >= 8) break; }
    }
    const int scratch = previousSlot(cur.slot);                    // the previous frame's slot pair: its device data is not needed while lost
    std::vector<int32_t> mq((size_t)m_maxKp), mt((size_t)m_maxKp), md((size_t)m_maxKp);
    for (auto& nc : near) {
        const Keyframe& kf = m_kfs[(size_t)nc.second];
        if (kf.kpts.empty()) continue;
        std::vector<int> cur_idx, lm_ids;
        if (use_bow && kf.node.size() == kf.kpts.size()) {
            // [UPSTREAM] match::bow_tree::match_frame_and_keyframe: the keyframe's keypoints that carry a landmark against the
            // frame's keypoints under the same vocabulary node (Hamming <= 50, ratio 0.75), then the orientation check
            std::vector<int32_t> qn(kf.node);
            for (size_t i = 0; i < qn.size(); ++i) if (resolve(kf.landmark[i]) < 0) qn[i] = -1;
            std::vector<int32_t> idx(kf.kpts.size(), -1);
            int32_t nm = 0;
            if (lpslam_hip_match_bow_tree(m_ctx, kf.desc.data(), qn.data(), (int32_t)qn.size(), cur.desc.data(), cur_node.data(), (int32_t)cur_node.size(), nullptr, 50, 0.75f,
                                          idx.data(), nullptr, &nm) != LPSLAM_HIP_OK) continue;
            std::vector<float> aq(kf.kpts.size()), at(cur.kpts.size());
            for (size_t i = 0; i < aq.size(); ++i) aq[i] = kf.kpts[i].angle;
            for (size_t i = 0; i < at.size(); ++i) at[i] = cur.kpts[i].angle;
            int32_t kept = 0;
            (void)lpslam_hip_match_orientation_filter(aq.data(), at.data(), idx.data(), (int32_t)idx.size(), &kept);
            for (size_t i = 0; i < idx.size(); ++i) if (idx[i] >= 0) { cur_idx.push_back(idx[i]); lm_ids.push_back(resolve(kf.landmark[i])); }
        } else {
            int32_t nm = 0;
            if (lpslam_hip_match_bf_descriptors(m_ctx, cur.slot, scratch, kf.desc.data(), (int32_t)kf.kpts.size(), 50, 0.75f, 1, mq.data(), mt.data(), md.data(), m_maxKp, &nm) != LPSLAM_HIP_OK) continue;
            for (int k = 0; k < nm; ++k) {
                const int id = resolve(kf.landmark[(size_t)mt[k]]);
                if (id < 0) continue;
                cur_idx.push_back(mq[k]); lm_ids.push_back(id);
            }
        }
        if (cur_idx.size() < 15) continue;
        // [UPSTREAM] solve::pnp_solver: the pose from the matches alone -- no prior, the camera may be anywhere that sees these
        // landmarks -- refined by the pose optimiser ([UPSTREAM] min_num_inliers 10 for the solver)
        Pose seed = kf.pose;
        {
            std::vector<double> pw, ob, w;
            for (size_t k = 0; k < cur_idx.size(); ++k) {
                auto it = m_landmarks.find(lm_ids[k]);
                if (it == m_landmarks.end()) continue;               // culled since the keyframe saw it (poseFromMatches skips it too)
                const lpslam_hip_keypoint& kp = cur.kpts[(size_t)cur_idx[k]];
                pw.insert(pw.end(), it->second.p, it->second.p + 3);
                ob.push_back(kp.x); ob.push_back(kp.y);
                const double sc = m_scales[kp.octave];
                w.push_back(1.0 / (sc * sc));
            }
            const size_t nm2 = w.size();
            const double cam4[4] = {m_cam.f_x, m_cam.f_y, m_cam.c_x, m_cam.c_y};
            double p7[7];
            std::vector<uint8_t> pnp_inl(std::max<size_t>(nm2, 1));
            if (nm2 < 4 || pnp_solve_ransac(pw.data(), ob.data(), w.data(), (int)nm2, cam4, 100, 0x9E3779B9u, p7, pnp_inl.data()) < 10) continue;
            for (int a = 0; a < 4; ++a) seed.q[a] = p7[a];
            for (int a = 0; a < 3; ++a) seed.t[a] = p7[4 + a];
        }
        int inl = 0;
        FrameData trial = cur;
        if (!poseFromMatches(trial, cur_idx, lm_ids, seed, inl, 30)) continue;
        cur.pose = trial.pose; cur.landmark = trial.landmark;
        logMessage(LpSlamLogLevel_Info, "VSLAM relocalised against keyframe " + std::to_string(nc.second) + " with " + std::to_string(inl) + " inliers");
        return true;
    }
    return false;
}

// ---- loop closing ([UPSTREAM] module::loop_detector + loop_bundle_adjuster, global_optimization_module) -------------------------
// Candidates: keyframes well outside the covisibility of the new keyframe, chosen by descriptor voting -- every candidate's
// descriptors are matched against the keyframe's on the device (mutual best, Hamming <= 50, ratio 0.75) and the three with the
// most matches that carry landmarks on both sides go on (the reference asks its DBoW2 vocabulary, which is absent; when the map
// holds more than 48 such keyframes only the 48 nearest to the current estimate are asked).  The Sim3 optimiser of the loop
// detector verifies a candidate (>= 20 inliers, scale fixed for stereo); the loop is closed by the Sim3 pose graph over the
// keyframes in between (consecutive edges + the loop edge, 50 iterations), landmarks move with the keyframe that created them,
// the candidate's landmarks are fused into the new keyframe (duplicates of the revisited structure merge), and a global bundle
// adjustment over the keyframes of the loop (10 iterations, the candidate fixed) refines the result.
bool HipVslamTrackerBase::detectAndCloseLoop(FrameData& cur, int c)
{
    const int newest_candidate = c - 2 * m_localWindow;                 // well outside the local window
    if (newest_candidate < 0) return false;
    const Keyframe& kc = m_kfs[(size_t)c];
    std::unordered_map<int, char> covis;
    for (int k : covisible(c, (int)m_kfs.size(), 15)) covis[k] = 1;      // the covisibility graph's neighbours (edges of weight >= 15) are no loop
    double Cc[3];
    pose_centre(kc.pose.q, kc.pose.t, Cc);
    std::vector<std::pair<double, int>> cands;
    for (int a = 0; a <= newest_candidate; ++a) {
        if (covis.count(a) || m_kfs[(size_t)a].erased) continue;
        double C[3];
        pose_centre(m_kfs[(size_t)a].pose.q, m_kfs[(size_t)a].pose.t, C);
        cands.emplace_back(std::sqrt((C[0] - Cc[0]) * (C[0] - Cc[0]) + (C[1] - Cc[1]) * (C[1] - Cc[1]) + (C[2] - Cc[2]) * (C[2] - Cc[2])), a);
    }
    if (cands.empty()) { m_loopSets.clear(); return false; }
    std::sort(cands.begin(), cands.end());
    if (cands.size() > 48) cands.resize(48);
    // with a vocabulary: [UPSTREAM] loop_detector::detect_loop_candidates -- keyframes outside the covisibility that share words with
    // the new keyframe and score (L1) at least as well as its worst covisible neighbour, best first, wherever the estimate puts them
    const bool use_bow = m_vocab && kc.node.size() == kc.kpts.size() && !kc.bow.empty();
    if (use_bow) {
        double min_score = 1.0;
        for (auto& kv : covis) { const BowVector* nb = m_bowDb.vector_of(kv.first); if (nb) min_score = std::min(min_score, bow_score_l1(kc.bow, *nb)); }
        if (covis.empty()) min_score = 0.0;
        std::unordered_map<int, char> exclude = covis;
        exclude[c] = 1;
        cands.clear();
        // [UPSTREAM] data::bow_database::acquire_loop_candidates, last two steps (ORB-SLAM's DetectLoopCandidates): a candidate's score is
        // ACCUMULATED over its covisibility group -- itself plus those of its ten best covisible keyframes that are candidates too -- the
        // group is represented by its best-scoring member, and the groups whose total reaches 0.75 of the best total are kept.
        const auto scored = m_bowDb.query(kc.bow, exclude, min_score, newest_candidate);
        std::unordered_map<int, double> score_of;
        for (auto& sc : scored) score_of[sc.second] = sc.first;
        std::vector<std::pair<double, int>> groups;            // (accumulated score, best keyframe of the group)
        double best_total = 0;
        for (auto& sc : scored) {
            double total = sc.first, best_sc = sc.first;
            int best_kf = sc.second;
            for (int nb : covisible(sc.second, 10, 15)) {
                auto it = score_of.find(nb);
                if (it == score_of.end()) continue;
                total += it->second;
                if (it->second > best_sc) { best_sc = it->second; best_kf = nb; }
            }
            groups.emplace_back(total, best_kf);
            best_total = std::max(best_total, total);
        }
        std::unordered_map<int, char> kept;
        std::sort(groups.begin(), groups.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first != b.first ? a.first > b.first : a.second > b.second; });
        for (auto& g : groups) if (g.first > 0.75 * best_total && !kept.count(g.second)) { kept[g.second] = 1; cands.emplace_back(-g.first, g.second); }
        if (cands.empty()) { m_loopSets.clear(); return false; }
    }
    // [UPSTREAM] loop_detector::find_continuously_detected_keyframe_sets (min_continuity_ = 3; ORB-SLAM's covisibility consistency,
    // toggled with the detector at src/Trackers/OpenVSLAMTrackerBase.cpp:250-255), on the RAW candidates of the query, before any
    // descriptor is matched: a candidate stands for the set of itself and its covisibility neighbours; its continuity is one more than
    // that of a set detected at the PREVIOUS keyframe that shares a keyframe with it (0 when there is none); only candidates whose
    // continuity has reached 3 -- detected at four keyframes in a row -- go on to the matching and the Sim3 verification.  The chain
    // breaks only at a keyframe whose query returns no candidate (above); the 20-match test comes later, as upstream's does.
    {
        std::vector<std::pair<std::vector<int>, int>> sets_now;
        std::vector<std::pair<double, int>> continuous;
        for (auto& cd : cands) {
            std::vector<int> group = covisible(cd.second, (int)m_kfs.size(), 15);
            group.push_back(cd.second);
            std::sort(group.begin(), group.end());
            int cont = 0;
            for (auto& prev : m_loopSets) {
                bool shared = false;
                for (size_t i = 0, j = 0; i < group.size() && j < prev.first.size() && !shared;) {
                    if (group[i] == prev.first[j]) shared = true;
                    else if (group[i] < prev.first[j]) ++i; else ++j;
                }
                if (shared) cont = std::max(cont, prev.second + 1);
            }
            sets_now.emplace_back(std::move(group), cont);
            if (cont >= 3) continuous.push_back(cd);
        }
        m_loopSets = std::move(sets_now);
        cands = std::move(continuous);
        if (cands.empty()) return false;
    }
    const int scratch = previousSlot(cur.slot);                  // the previous frame's slot pair is free for the descriptors of a candidate
    std::vector<int32_t> mq((size_t)m_maxKp), mt((size_t)m_maxKp), md((size_t)m_maxKp);
    struct Vote { int kf; std::vector<std::pair<int, int>> pairs; };        // (keypoint of c, keypoint of the candidate), landmarks on both sides
    std::vector<Vote> votes;
    // without a vocabulary every candidate is matched descriptor against descriptor: all of them in ONE call against the keyframe
    // descriptors kept on the device (lpslam_hip_match_bf_stored; candidate by candidate this was an upload and a wait each -- 2 ms
    // per keyframe once a session has 100 keyframes)
    std::vector<int32_t> bq, bt, bd, bn, bkeys;
    bool batched = false;
    if (!use_bow) {
        // (only keyframes whose descriptors did reach the device: one missing key would make the call refuse the whole batch; the others
        // are compared one by one below)
        for (auto& cd : cands) if (!m_kfs[(size_t)cd.second].kpts.empty() && m_kfs[(size_t)cd.second].desc_on_device) bkeys.push_back(cd.second);
        bq.resize(bkeys.size() * (size_t)m_maxKp); bt.resize(bq.size()); bd.resize(bq.size()); bn.assign(bkeys.size(), 0);
        batched = !bkeys.empty() && lpslam_hip_match_bf_stored(m_ctx, cur.slot, bkeys.data(), (int32_t)bkeys.size(), 50, 0.75f, 1, bq.data(), bt.data(), bd.data(), m_maxKp, bn.data()) == LPSLAM_HIP_OK;
    }
    // with a vocabulary: [UPSTREAM] match::bow_tree::match_keyframes per candidate -- landmark-carrying keypoints of both keyframes under the
    // same vocabulary node -- for ALL candidates in one call (lpslam_hip_match_bow_tree_multi: one upload, one wait; candidate by candidate
    // it was 47 us each, eight times per keyframe)
    std::vector<std::vector<int32_t>> bow_idx;
    std::vector<int> bow_slot(cands.size(), -1);
    if (use_bow) {
        std::vector<int32_t> qn(kc.node);
        for (size_t i = 0; i < qn.size(); ++i) if (kc.landmark[i] < 0) qn[i] = -1;
        std::vector<std::vector<int32_t>> tns;
        std::vector<const uint8_t*> tdp; std::vector<const int32_t*> tnp; std::vector<int32_t> tnt; std::vector<int32_t*> idp;
        for (size_t ci = 0; ci < cands.size(); ++ci) {
            const Keyframe& ka = m_kfs[(size_t)cands[ci].second];
            if (ka.kpts.empty() || ka.node.size() != ka.kpts.size()) continue;
            bow_slot[ci] = (int)tns.size();
            tns.emplace_back(ka.node);
            std::vector<int32_t>& tn = tns.back();
            for (size_t i = 0; i < tn.size(); ++i) if (ka.landmark[i] < 0) tn[i] = -1;
            bow_idx.emplace_back(kc.kpts.size(), -1);
        }
        for (size_t ci = 0; ci < cands.size(); ++ci) {
            if (bow_slot[ci] < 0) continue;
            const Keyframe& ka = m_kfs[(size_t)cands[ci].second];
            tdp.push_back(ka.desc.data()); tnp.push_back(tns[(size_t)bow_slot[ci]].data()); tnt.push_back((int32_t)ka.kpts.size()); idp.push_back(bow_idx[(size_t)bow_slot[ci]].data());
        }
        if (!tdp.empty() && lpslam_hip_match_bow_tree_multi(m_ctx, kc.desc.data(), qn.data(), (int32_t)qn.size(), (int32_t)tdp.size(), tdp.data(), tnp.data(), tnt.data(), nullptr, 50, 0.75f,
                                                            idp.data(), nullptr, nullptr) != LPSLAM_HIP_OK) {
            logMessage(LpSlamLogLevel_Error, std::string("loop candidates: ") + lpslam_hip_last_error());
            for (auto& b : bow_slot) b = -2;                // every candidate of this keyframe is skipped
        }
    }
    size_t b_at = 0;
    for (size_t ci = 0; ci < cands.size(); ++ci) {
        auto& cd = cands[ci];
        const Keyframe& ka = m_kfs[(size_t)cd.second];
        if (ka.kpts.empty()) continue;
        int32_t nm = 0;
        if (use_bow && ka.node.size() == ka.kpts.size()) {
            if (bow_slot[ci] < 0) continue;
            std::vector<int32_t>& idx = bow_idx[(size_t)bow_slot[ci]];
            std::vector<float> aq(kc.kpts.size()), at(ka.kpts.size());
            for (size_t i = 0; i < aq.size(); ++i) aq[i] = kc.kpts[i].angle;
            for (size_t i = 0; i < at.size(); ++i) at[i] = ka.kpts[i].angle;
            int32_t kept = 0;
            (void)lpslam_hip_match_orientation_filter(aq.data(), at.data(), idx.data(), (int32_t)idx.size(), &kept);
            for (size_t i = 0; i < idx.size(); ++i) if (idx[i] >= 0) { mq[(size_t)nm] = (int32_t)i; mt[(size_t)nm] = idx[i]; ++nm; }
        } else if (batched && ka.desc_on_device) {
            const size_t at = b_at++ * (size_t)m_maxKp;
            nm = bn[b_at - 1];
            std::copy(bq.begin() + (long)at, bq.begin() + (long)at + nm, mq.begin());
            std::copy(bt.begin() + (long)at, bt.begin() + (long)at + nm, mt.begin());
        } else {
            if (lpslam_hip_match_bf_descriptors(m_ctx, cur.slot, scratch, ka.desc.data(), (int32_t)ka.kpts.size(), 50, 0.75f, 1, mq.data(), mt.data(), md.data(), m_maxKp, &nm) != LPSLAM_HIP_OK) continue;
        }
        Vote v; v.kf = cd.second;
        for (int k = 0; k < nm; ++k)
            if (kc.landmark[(size_t)mq[k]] >= 0 && ka.landmark[(size_t)mt[k]] >= 0 && resolve(kc.landmark[(size_t)mq[k]]) != resolve(ka.landmark[(size_t)mt[k]])) v.pairs.emplace_back(mq[k], mt[k]);
        if (v.pairs.size() >= 20) votes.push_back(std::move(v));         // [UPSTREAM] num_matches >= 20 to try a candidate
    }
    if (votes.empty()) return false;
    std::sort(votes.begin(), votes.end(), [](const Vote& a, const Vote& b) { return a.pairs.size() != b.pairs.size() ? a.pairs.size() > b.pairs.size() : a.kf > b.kf; });
    if (votes.size() > 3) votes.resize(3);
    // [UPSTREAM] solve::sim3_solver: the similarity candidate camera -> current camera from the matched landmarks alone (Horn on
    // 3-match samples, RANSAC over the reprojection error in both images), independent of the drifted estimates; the Sim3 optimiser
    // on the device refines it.  Monocular maps drift in scale, so the scale is free there and fixed for stereo.
    const bool fix_scale = m_stereo;
    std::vector<lpslam_hip_sim3_pair> pairs;
    std::vector<int32_t> start{0};
    std::vector<double> s12;
    std::vector<int> vote_kf;
    const Mat3 Rc = quatToRot(kc.pose.q);
    const double cam[4] = {m_cam.f_x, m_cam.f_y, m_cam.c_x, m_cam.c_y};
    for (auto& v : votes) {
        const Keyframe& ka = m_kfs[(size_t)v.kf];
        const Mat3 Ra = quatToRot(ka.pose.q);
        const size_t first = pairs.size(), np_ = v.pairs.size();
        std::vector<double> p1((size_t)3 * np_), p2((size_t)3 * np_), o1((size_t)2 * np_), o2((size_t)2 * np_), w1(np_), w2(np_);
        for (size_t n_ = 0; n_ < np_; ++n_) {
            const size_t ic = (size_t)v.pairs[n_].first, ia = (size_t)v.pairs[n_].second;
            const Landmark& lc = m_landmarks.at(resolve(kc.landmark[ic]));
            const Landmark& la = m_landmarks.at(resolve(ka.landmark[ia]));
            lpslam_hip_sim3_pair pr{};
            for (int r = 0; r < 3; ++r) {
                pr.p1c[r] = Rc.m[r * 3] * lc.p[0] + Rc.m[r * 3 + 1] * lc.p[1] + Rc.m[r * 3 + 2] * lc.p[2] + kc.pose.t[r];
                pr.p2c[r] = Ra.m[r * 3] * la.p[0] + Ra.m[r * 3 + 1] * la.p[1] + Ra.m[r * 3 + 2] * la.p[2] + ka.pose.t[r];
                p1[3 * n_ + (size_t)r] = pr.p1c[r]; p2[3 * n_ + (size_t)r] = pr.p2c[r];
            }
            pr.obs1[0] = kc.kpts[ic].x; pr.obs1[1] = kc.kpts[ic].y; pr.obs2[0] = ka.kpts[ia].x; pr.obs2[1] = ka.kpts[ia].y;
            const double s1 = m_scales[kc.kpts[ic].octave], s2 = m_scales[ka.kpts[ia].octave];
            pr.inv_sigma2_1 = 1.0 / (s1 * s1); pr.inv_sigma2_2 = 1.0 / (s2 * s2);
            o1[2 * n_] = pr.obs1[0]; o1[2 * n_ + 1] = pr.obs1[1]; o2[2 * n_] = pr.obs2[0]; o2[2 * n_ + 1] = pr.obs2[1];
            w1[n_] = pr.inv_sigma2_1; w2[n_] = pr.inv_sigma2_2;
            pairs.push_back(pr);
        }
        double seed12[8];
        std::vector<uint8_t> seed_inl(np_);
        const int found = sim3_solve_ransac(p1.data(), p2.data(), o1.data(), o2.data(), w1.data(), w2.data(), (int)np_, cam, cam, fix_scale, 200, 0x9E3779B9u, seed12, seed_inl.data());
        // a seed needs 12 inliers ([UPSTREAM] asks the solver for 20 of its several hundred bag-of-words matches); the count that
        // decides stays the optimiser's: 20
        if (found < 12) { pairs.resize(first); continue; }
        start.push_back((int32_t)pairs.size());
        s12.insert(s12.end(), seed12, seed12 + 8);
        vote_kf.push_back(v.kf);
    }
    if (vote_kf.empty()) return false;
    std::vector<uint8_t> inl(pairs.size());
    std::vector<int32_t> n_inl(vote_kf.size(), 0);
    if (lpslam_hip_sim3_transform_optimize(m_ctx, (int32_t)vote_kf.size(), s12.data(), pairs.data(), start.data(), cam, cam, 10.0, fix_scale ? 1 : 0, inl.data(), n_inl.data()) != LPSLAM_HIP_OK) return false;
    int best = -1;
    for (size_t i = 0; i < vote_kf.size(); ++i) if (n_inl[i] >= 20 && (best < 0 || n_inl[i] > n_inl[(size_t)best])) best = (int)i;
    if (best < 0) return false;

    // ---- pose graph over the keyframes of the loop: a0 = candidate (fixed) ... c
    const int a0 = vote_kf[(size_t)best];
    const int n = c - a0 + 1;
    std::vector<double> verts(8 * (size_t)n);
    std::vector<uint8_t> fixed((size_t)n, 0);
    fixed[0] = 1;
    std::vector<Se3> old((size_t)n);
    for (int v = 0; v < n; ++v) {
        const Pose& p = m_kfs[(size_t)(a0 + v)].pose;
        old[(size_t)v] = Se3{{p.q[0], p.q[1], p.q[2], p.q[3]}, {p.t[0], p.t[1], p.t[2]}};
        const double row[8] = {p.q[0], p.q[1], p.q[2], p.q[3], p.t[0], p.t[1], p.t[2], 1.0};
        std::copy(row, row + 8, verts.begin() + 8 * (long)v);
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
        std::copy(s12.begin() + 8 * (long)best, s12.begin() + 8 * (long)best + 8, e.meas);
        edges.push_back(e);
    }
    lpslam_hip_sim3* graph = nullptr;
    if (lpslam_hip_sim3_create(m_ctx, verts.data(), fixed.data(), n, edges.data(), (int32_t)edges.size(), fix_scale ? 1 : 0, &graph) != LPSLAM_HIP_OK) return false;
    int32_t done = 0;
    const bool ok = lpslam_hip_sim3_optimize(graph, 50, nullptr, &done) == LPSLAM_HIP_OK && lpslam_hip_sim3_get(graph, verts.data()) == LPSLAM_HIP_OK;
    lpslam_hip_sim3_destroy(graph);
    if (!ok) return false;
    finishMapping();                                     // no window solve may be in flight while the map moves
    std::vector<Se3> neu((size_t)n);
    std::vector<double> scale((size_t)n, 1.0);
    for (int v = 0; v < n; ++v) {
        const double* r = &verts[8 * (size_t)v];
        const double qn = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]), s = r[7] > 0 ? r[7] : 1.0;
        scale[(size_t)v] = s;                            // Sim3 (R, t, s) -> SE3 (R, t / s): camera coordinates shrink by s
        neu[(size_t)v] = Se3{{r[0] / qn, r[1] / qn, r[2] / qn, r[3] / qn}, {r[4] / s, r[5] / s, r[6] / s}};
        Pose& p = m_kfs[(size_t)(a0 + v)].pose;
        for (int k = 0; k < 4; ++k) p.q[k] = neu[(size_t)v].q[k];
        for (int k = 0; k < 3; ++k) p.t[k] = neu[(size_t)v].t[k];
    }
    for (auto& kv : m_landmarks) {                       // X_new = T_ref_new^-1 (T_ref_old X)
        const int rk = kv.second.ref_kf;
        if (rk < a0 || rk > c) continue;
        const Se3& To = old[(size_t)(rk - a0)]; const Se3 Tni = se3_inv(neu[(size_t)(rk - a0)]);
        const Mat3 Ro = quatToRot(To.q), Rn = quatToRot(Tni.q);
        double xc[3], xw[3];
        const double sc = scale[(size_t)(rk - a0)];
        for (int r = 0; r < 3; ++r) xc[r] = (Ro.m[r * 3] * kv.second.p[0] + Ro.m[r * 3 + 1] * kv.second.p[1] + Ro.m[r * 3 + 2] * kv.second.p[2] + To.t[r]) / sc;
        for (int r = 0; r < 3; ++r) xw[r] = Rn.m[r * 3] * xc[0] + Rn.m[r * 3 + 1] * xc[1] + Rn.m[r * 3 + 2] * xc[2] + Tni.t[r];
        for (int r = 0; r < 3; ++r) kv.second.p[r] = xw[r];
    }
    cur.pose = m_kfs[(size_t)c].pose;
    // ---- the revisited structure exists twice: the landmarks of the candidate and of its covisible keyframes are searched in the
    // new keyframe and merged with what it holds ([UPSTREAM] loop_detector -> fuse with the loop's landmarks)
    {
        std::vector<int> nb = covisible(a0, m_localWindow - 1, 15);
        nb.push_back(a0);
        std::sort(nb.begin(), nb.end());
        IdMarks& held = m_marksA;
        held.begin((size_t)m_nextLandmarkId);
        for (int id : m_kfs[(size_t)c].landmark) if (id >= 0) held.at(id) = 1;
        std::vector<int> ids;
        for (int k : nb) for (int id : m_kfs[(size_t)k].landmark) if (id >= 0 && !held.has(id)) { held.at(id) = 1; ids.push_back(id); }
        long added = 0, merged = 0;
        fuseInto(c, cur.slot, ids, cur, added, merged);
        m_stats.loop_fused += added + merged;
    }
    // ---- global bundle adjustment over the keyframes of the loop ([UPSTREAM] loop_bundle_adjuster: 10 iterations), inline
    {
        std::vector<int> free_kfs, fixed_kfs{a0};
        for (int k = a0 + 1; k <= c; ++k) if (!m_kfs[(size_t)k].erased) free_kfs.push_back(k);
        auto job = prepareBundle(free_kfs, fixed_kfs);
        if (job) {
            job->global = true;
            solveMapping(*job);
            if (job->solved) { applyMapping(*job); --m_stats.local_ba; ++m_stats.global_ba; cur.pose = m_kfs[(size_t)c].pose; }
        }
    }
    ++m_stats.loops_closed;
    m_loopSets.clear();
    logMessage(LpSlamLogLevel_Info, "VSLAM loop closed: keyframe " + std::to_string(c) + " with " + std::to_string(a0) + ", " + std::to_string(n_inl[(size_t)best]) + " inliers");
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
        const auto t_solve = std::chrono::steady_clock::now();
        solveMapping(*job);
        const double solved_in = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_solve).count();
        lk.lock();
        m_stats.t_map_solve += solved_in;
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

void HipVslamTrackerBase::startMapping(int c)
{
    std::unique_ptr<MappingJob> job;
    { ScopedSeconds timed(m_stats.t_kf_prepare); job = prepareMapping(c); }
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
        ScopedSeconds timed(m_stats.t_kf_wait);
        std::unique_lock<std::mutex> lk(m_mapMutex);
        m_mapCv.wait(lk, [this] { return !m_mapBusy; });
        job = std::move(m_mapOut);
    }
    if (job) {
        if (!job->solved) {      // the window's solve failed on the device (its result is dropped, the map stays as it was): counted, and said once
            if (m_stats.ba_failed++ == 0) logMessage(LpSlamLogLevel_Error, std::string("VSLAM bundle adjustment failed: ") + lpslam_hip_last_error());
        }
        ScopedSeconds timed(m_stats.t_kf_apply); applyMapping(*job);
    }
}

void HipVslamTrackerBase::storeDescriptors(int key, Keyframe& kf)
{
    kf.desc_on_device = lpslam_hip_desc_store_put(m_ctx, key, kf.desc.data(), (int32_t)kf.kpts.size()) == LPSLAM_HIP_OK;
    if (!kf.desc_on_device)      // (the loop-candidate search then uploads this keyframe's descriptors whenever it is a candidate: slower, same matches)
        logMessage(LpSlamLogLevel_Error, std::string("keyframe ") + std::to_string(key) + ": descriptors not kept on the device (" + lpslam_hip_last_error() + "); compared one by one from now on");
}

void HipVslamTrackerBase::logStatistics()
{
    const Statistics& s = m_stats;
    char buf[2048];
    const double per = s.frames > 0 ? 1e3 / (double)s.frames : 0.0;
    int64_t dev[LPSLAM_HIP_BA_COUNTERS] = {0};        // what happened to the windows on the device: graph cache, replays, timed-out hand-overs
    if (m_ctx) (void)lpslam_hip_ba_counters(m_ctx, dev, LPSLAM_HIP_BA_COUNTERS);
    std::snprintf(buf, sizeof(buf), "VSLAM statistics: frames=%ld motion_tracked=%ld bf_tracked=%ld local_map_joined=%ld keyframes=%ld fused_added=%ld fused_merged=%ld "
                  "local_ba=%ld loops_closed=%ld loop_fused=%ld global_ba=%ld lost=%ld relocalised=%ld reinitialised=%ld nav_priors=%ld landmarks=%zu "
                  "culled_landmarks=%ld culled_keyframes=%ld live_keyframes=%ld prefetched=%ld ba_failed=%ld ba_signatures=%ld ba_graphs=%ld ba_replays=%ld ba_timeouts=%ld ms_per_frame=%.4f ms_front_end=%.4f ms_track=%.4f ms_local_map=%.4f ms_keyframe=%.4f ms_dev_upload=%.4f ms_dev_extract=%.4f ms_dev_get=%.4f ms_dev_match=%.4f ms_dev_pose=%.4f ms_prefetch_wait=%.4f ms_prefetch_busy=%.4f ms_kf_wait=%.4f ms_kf_apply=%.4f ms_kf_insert=%.4f ms_kf_loop=%.4f ms_kf_prepare=%.4f ms_map_solve=%.4f",
                  s.frames, s.motion_tracked, s.bf_tracked, s.local_map_joined, s.keyframes, s.fused_added, s.fused_merged, s.local_ba, s.loops_closed, s.loop_fused,
                  s.global_ba, s.lost, s.relocalised, s.reinitialised, s.nav_priors, m_landmarks.size(), s.culled_landmarks, s.culled_keyframes,
                  (long)std::count_if(m_kfs.begin(), m_kfs.end(), [](const Keyframe& k) { return !k.erased; }), s.prefetched,
                  s.ba_failed, (long)dev[LPSLAM_HIP_BA_COUNTER_SIGNATURES], (long)dev[LPSLAM_HIP_BA_COUNTER_GRAPHS], (long)dev[LPSLAM_HIP_BA_COUNTER_REPLAYS],
                  (long)(dev[LPSLAM_HIP_BA_COUNTER_TIMEOUTS_BAND] + dev[LPSLAM_HIP_BA_COUNTER_TIMEOUTS_UPDATE]),
                  s.t_total * per, s.t_front * per, s.t_track * per, s.t_local * per, s.t_keyframe * per,
                  s.t_dev_upload * per, s.t_dev_extract * per, s.t_dev_get * per, s.t_dev_match * per, s.t_dev_pose * per, s.t_prefetch_wait * per, s.t_prefetch_busy * per, s.t_kf_wait * per, s.t_kf_apply * per, s.t_kf_insert * per, s.t_kf_loop * per, s.t_kf_prepare * per, s.t_map_solve * per);
    m_lastStatistics = buf;
    logMessage(LpSlamLogLevel_Info, buf);
}

// upload (raw frames: + undistortion / rectification on the device), extraction, stereo matching of one frame into a slot pair;
// everything is only enqueued
bool HipVslamTrackerBase::frontEnd(CameraQueueEntry const& cam, bool stereo, int slot)
{
    bool ok;
    static const bool readback_ahead_fused = std::getenv("LPSLAM_HIP_NO_PREFETCH_READBACK") == nullptr;
    if (!m_rectify && readback_ahead_fused) {
        // upload, extraction, stereo match and the read-back that rides behind them: ONE call, which the frames that other sessions of the
        // process have pending join (one upload + launch chain for all of them, share.hip)
        const float baseline = (float)(m_cam.focal_x_baseline / m_cam.f_x);
        return lpslam_hip_front_end_images(m_ctx, slot, cam.image.pixels.data(), stereo ? cam.image_second->pixels.data() : nullptr, cam.image.width,
                                           (float)m_cam.focal_x_baseline, baseline) == LPSLAM_HIP_OK;
    }
    if (m_rectify) {       // raw frames: undistort + rectify on the device
        ok = lpslam_hip_upload_raw_image(m_ctx, slot, 0, cam.image.pixels.data(), cam.image.width) == LPSLAM_HIP_OK;
        if (ok && stereo) ok = lpslam_hip_upload_raw_image(m_ctx, slot + 1, 1, cam.image_second->pixels.data(), cam.image.width) == LPSLAM_HIP_OK;
    } else {
        ok = lpslam_hip_upload_image(m_ctx, slot, cam.image.pixels.data(), cam.image.width) == LPSLAM_HIP_OK;
        if (ok && stereo) ok = lpslam_hip_upload_image(m_ctx, slot + 1, cam.image_second->pixels.data(), cam.image.width) == LPSLAM_HIP_OK;
    }
    static const bool readback_ahead = std::getenv("LPSLAM_HIP_NO_PREFETCH_READBACK") == nullptr;      // (development switch)
    const float baseline = (float)(m_cam.focal_x_baseline / m_cam.f_x);
    // extraction, stereo match and the read-back that rides behind them: one call (shared with the other sessions' pending frames when there are any)
    if (ok && readback_ahead) return lpslam_hip_front_end(m_ctx, slot, stereo ? 1 : 0, (float)m_cam.focal_x_baseline, baseline) == LPSLAM_HIP_OK;
    if (ok) ok = lpslam_hip_extract_range(m_ctx, slot, stereo ? 2 : 1) == LPSLAM_HIP_OK;
    if (ok && stereo) ok = lpslam_hip_match_stereo(m_ctx, slot, slot + 1, (float)m_cam.focal_x_baseline, baseline) == LPSLAM_HIP_OK;
    return ok;
}

// The front end of the frame that comes next, on the context's prefetch stream: issued right after this frame's own front end has
// come back, it runs on the GPU while this frame is matched and optimised (LpSlamManager hands the queued frame over with
// setNextFrame).
void HipVslamTrackerBase::prefetchFrame(CameraQueueEntry const& cam, bool stereo)
{
    m_prefetched.valid = false;
    if (!m_ctx || !m_prefetch) return;
    if (stereo && !cam.image_second.has_value()) return;
    if (cam.image.width != m_cam.resolution_x || cam.image.height != m_cam.resolution_y ||
        (stereo && (cam.image_second->width != cam.image.width || cam.image_second->height != cam.image.height))) return;
    const int slot = slotOf(m_imageTracked);              // the current frame has been counted already
    if (lpslam_hip_prefetch_begin(m_ctx) != LPSLAM_HIP_OK) { logMessage(LpSlamLogLevel_Error, std::string("prefetch_begin: ") + lpslam_hip_last_error()); return; }
    m_prefetched.issued = true;       // from here on uploads / kernels may be queued on the prefetch stream, even if a later step fails
    const bool ok = frontEnd(cam, stereo, slot);
    if (!ok) logMessage(LpSlamLogLevel_Error, std::string("prefetch front end: ") + lpslam_hip_last_error());
    if (lpslam_hip_prefetch_end(m_ctx) != LPSLAM_HIP_OK || !ok) return;
    m_prefetched.valid = true; m_prefetched.data = cam.image.pixels.data(); m_prefetched.timestamp = cam.timestamp;
    m_prefetched.slot = slot; m_prefetched.stereo = stereo;
}

void HipVslamTrackerBase::prefetchLoop()
{
    std::unique_lock<std::mutex> lk(m_pfMutex);
    for (;;) {
        m_pfCv.wait(lk, [this] { return m_pfQuit || m_pfJob; });
        if (m_pfQuit) return;
        const CameraQueueEntry* job = m_pfJob;
        const bool stereo = m_pfStereo;
        lk.unlock();
        const auto t_job = std::chrono::steady_clock::now();
        prefetchFrame(*job, stereo);
        const double busy = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_job).count();
        lk.lock();
        m_stats.t_prefetch_busy += busy;
        m_pfJob = nullptr; m_pfBusy = false;
        m_pfCv.notify_all();
    }
}

void HipVslamTrackerBase::prefetchSubmit(const CameraQueueEntry* next, bool stereo)
{
    if (!m_pfThread.joinable()) m_pfThread = std::thread([this] { prefetchLoop(); });
    std::lock_guard<std::mutex> lk(m_pfMutex);
    m_pfJob = next; m_pfStereo = stereo; m_pfBusy = true;
    m_pfCv.notify_all();
}

void HipVslamTrackerBase::prefetchWait()
{
    const auto t0 = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> lk(m_pfMutex);
    m_pfCv.wait(lk, [this] { return !m_pfBusy; });
    m_stats.t_prefetch_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void HipVslamTrackerBase::stopPrefetchThread()
{
    if (!m_pfThread.joinable()) return;
    {
        std::unique_lock<std::mutex> lk(m_pfMutex);
        m_pfCv.wait(lk, [this] { return !m_pfBusy; });
        m_pfQuit = true;
        m_pfCv.notify_all();
    }
    m_pfThread.join();
    m_pfQuit = false;
}

TrackerBase::ProcessImageResult HipVslamTrackerBase::trackFrame(CameraQueueEntry& cam, bool stereo, const std::optional<GlobalStateInTime>& navOdom)
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
    // navigation prior: lpslam axes -> optical axes (q_ov = (w, y, -x, z), p_ov = (y, -x, z): src/Trackers/OpenVSLAMStereoTracker.cpp:
    // 117-135), kept as world -> camera
    m_navPrev = m_navCur;
    m_navCur.reset();
    if (navOdom && m_forwardNavState && navOdom->second.stateValid) {
        const Quaternion& ql = navOdom->second.orientation.value; const Vector3& pl = navOdom->second.position.value;
        const double qwc[4] = {ql.w, ql.y, -ql.x, ql.z};            // camera orientation in the world
        const double Cw[3] = {pl.y, -pl.x, pl.z};
        const double nn = std::sqrt(qwc[0] * qwc[0] + qwc[1] * qwc[1] + qwc[2] * qwc[2] + qwc[3] * qwc[3]);
        if (nn > 0) {
            Pose p;
            p.q[0] = qwc[0] / nn; p.q[1] = -qwc[1] / nn; p.q[2] = -qwc[2] / nn; p.q[3] = -qwc[3] / nn;      // world -> camera rotation
            const Mat3 R = quatToRot(p.q);
            for (int r = 0; r < 3; ++r) p.t[r] = -(R.m[r * 3] * Cw[0] + R.m[r * 3 + 1] * Cw[1] + R.m[r * 3 + 2] * Cw[2]);
            m_navCur = p;
        }
    }

    FrameData cur;
    cur.slot = slotOf(m_imageTracked);
    bool ok;
    auto t_dev = std::chrono::steady_clock::now();
    auto dev_lap = [&t_dev](double& acc) { const auto now = std::chrono::steady_clock::now(); acc += std::chrono::duration<double>(now - t_dev).count(); t_dev = now; };
    if (m_prefetched.valid && m_prefetched.data == cam.image.pixels.data() && m_prefetched.timestamp == cam.timestamp &&
        m_prefetched.slot == cur.slot && m_prefetched.stereo == stereo) {
        // this frame's front end was started while the previous frame was tracked: the main stream waits for it on the device
        ok = lpslam_hip_prefetch_join(m_ctx) == LPSLAM_HIP_OK;
        ++m_stats.prefetched;
        m_prefetched.valid = false; m_prefetched.issued = false;
    } else {
        if (m_prefetched.issued) {
            // a prefetch for a frame that did not come next (skipped, queue cleared) or one that failed part way: whatever it
            // queued writes the slot pair and the shared raw-image staging this frame's front end is about to use -- the main
            // stream waits for it first
            (void)lpslam_hip_prefetch_join(m_ctx);
            m_prefetched.valid = false; m_prefetched.issued = false;
        }
        ok = frontEnd(cam, stereo, cur.slot);
    }
    dev_lap(m_stats.t_dev_extract);
    int32_t n = 0;
    // the frame's results straight out of the context's page-locked block (delivered behind the prefetched front end): one copy,
    // into vectors of the frame's own size
    const lpslam_hip_keypoint* v_kp = nullptr; const uint8_t* v_desc = nullptr; const float* v_xr = nullptr; const float* v_dep = nullptr;
    if (ok) ok = lpslam_hip_get_frame_view(m_ctx, cur.slot, stereo ? 1 : 0, &v_kp, &v_desc, &v_xr, &v_dep, &n) == LPSLAM_HIP_OK && n <= m_maxKp;
    if (!ok) { dev_lap(m_stats.t_dev_get); logMessage(LpSlamLogLevel_Error, std::string("HIP front end failed: ") + lpslam_hip_last_error()); return res; }
    cur.kpts.assign(v_kp, v_kp + n); cur.desc.assign(v_desc, v_desc + (size_t)n * 32);
    if (stereo) { cur.x_right.assign(v_xr, v_xr + n); cur.depth.assign(v_dep, v_dep + n); }
    else { cur.x_right.assign((size_t)n, -1.0f); cur.depth.assign((size_t)n, -1.0f); }
    cur.landmark.assign((size_t)n, -1);
    dev_lap(m_stats.t_dev_get);
    ++m_imageTracked; ++m_stats.frames;
    // The next frame's upload + front end: a helper thread stages and enqueues it (0.2 ms of host time at 1280x720 stereo, mostly the
    // copy of the cold frame into page-locked memory) on the context's prefetch stream while this thread goes on tracking; the
    // future joins before this call returns (the frame is only valid that long) and on every early return.
    struct PrefetchJoin { HipVslamTrackerBase* t; ~PrefetchJoin() { t->prefetchWait(); } } prefetching{this};
    if (m_nextFrame && m_prefetch) prefetchSubmit(m_nextFrame, stereo);
    auto t_mark = std::chrono::steady_clock::now();
    auto lap = [&t_mark](double& acc) { const auto now = std::chrono::steady_clock::now(); acc += std::chrono::duration<double>(now - t_mark).count(); t_mark = now; };
    m_stats.t_front += std::chrono::duration<double>(t_mark - t0).count();

    if (m_state == TrackerState::Lost) {
        // the map stays; every frame tries to relocalise against the keyframes, and the state stays Lost (no pose goes out) until
        // that succeeds or time_to_relocalize has passed -- then a new map segment starts at the pose the tracker last believed
        // in, moved along by the navigation prior where there is one
        Pose nav_step;
        if (m_relocWithNavigation && navDelta(nav_step)) movePose(nav_step, m_lastGoodPose, m_lastGoodPose);
        if (relocalise(cur)) {
            m_state = TrackerState::Tracking;
            ++m_stats.relocalised;
            const int c = insertKeyframe(cur);
            startMapping(c);
            m_haveVelocity = false;
            m_prev = std::move(cur); m_havePrev = true;
        } else {
            const double lost_for = std::chrono::duration<double>(cam.timestamp - m_lostSince).count();
            if (lost_for > m_timeToRelocalize) {
                bool started = false;
                if (stereo) { ++m_segment; started = initializeMap(cur, m_lastGoodPose); if (!started) --m_segment; }
                else { m_state = TrackerState::Initializing; }            // monocular: the two-view initialiser takes over (scale is lost)
                if (started) {
                    m_state = TrackerState::Tracking; ++m_stats.reinitialised;
                    logMessage(LpSlamLogLevel_Info, "VSLAM not relocalised within " + std::to_string(m_timeToRelocalize) + " s: new map segment at the last known pose");
                }
            }
            m_haveVelocity = false;
            m_prev = std::move(cur); m_havePrev = true;
        }
    } else if (!stereo && m_state != TrackerState::Tracking) {
        // monocular: no map until two views with enough parallax have been found
        m_state = TrackerState::Initializing;
        if (monoInitialize(cur)) m_state = TrackerState::Tracking;
        m_haveVelocity = false;
        m_prev = std::move(cur); m_havePrev = true;
    } else if (m_state != TrackerState::Tracking) {
        m_state = TrackerState::Initializing;
        if (initializeMap(cur, Pose())) m_state = TrackerState::Tracking;
        m_haveVelocity = false;
        m_prev = std::move(cur); m_havePrev = true;
    } else {
        int inliers = 0;
        const bool tracked = trackAgainstPrevious(cur, inliers);
        lap(m_stats.t_track);
        if (tracked) {
            int with_local_map = 0;
            if (trackLocalMap(cur, with_local_map)) inliers = with_local_map;      // else: the motion-model result stands
            (void)lpslam_hip_frame_done(m_ctx);            // the frame's pose is final: other sessions' shared launches need not wait for this one
            for (int id : cur.landmark) if (id >= 0) { auto it = m_landmarks.find(id); if (it != m_landmarks.end()) ++it->second.n_observed; }      // inliers of the final pose optimisation
            lap(m_stats.t_local);
            // velocity = T_cur * T_prev^-1
            const Mat3 Rc = quatToRot(cur.pose.q), Rp = quatToRot(m_prev.pose.q);
            Mat3 Rv;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rv.m[r * 3 + c] = Rc.m[r * 3] * Rp.m[c * 3] + Rc.m[r * 3 + 1] * Rp.m[c * 3 + 1] + Rc.m[r * 3 + 2] * Rp.m[c * 3 + 2];
            rotToQuat(Rv, m_velocity.q);
            for (int r = 0; r < 3; ++r) m_velocity.t[r] = cur.pose.t[r] - (Rv.m[r * 3] * m_prev.pose.t[0] + Rv.m[r * 3 + 1] * m_prev.pose.t[1] + Rv.m[r * 3 + 2] * m_prev.pose.t[2]);
            m_haveVelocity = true;
            ++m_framesSinceKeyframe;
            if (keyframeNeeded(inliers)) {
                finishMapping();                    // the previous keyframe's solve enters the map before the next one is inserted
                int c;
                { ScopedSeconds timed(m_stats.t_kf_insert); c = insertKeyframe(cur); }
                if (m_loopClosure) { ScopedSeconds timed(m_stats.t_kf_loop); detectAndCloseLoop(cur, c); }
                startMapping(c);
                if (!m_asyncMapping) cur.pose = m_kfs[(size_t)c].pose;
                lap(m_stats.t_keyframe);
            }
            m_lastGoodPose = cur.pose;
            m_prev = std::move(cur);
        } else {
            (void)lpslam_hip_frame_done(m_ctx);
            finishMapping();
            m_haveMonoRef = false;
            m_state = TrackerState::Lost;
            m_lostSince = cam.timestamp;
            ++m_stats.lost;
            // the pose the tracker believes in for this frame: the prediction
            Pose pred = m_prev.pose;
            (void)predictedPose(pred);
            m_lastGoodPose = pred;
            logMessage(LpSlamLogLevel_Info, "VSLAM tracking lost; the map is kept, relocalising");
            m_prev = std::move(cur);
            m_haveVelocity = false;
        }
    }
    (void)lpslam_hip_frame_done(m_ctx);                  // (the branches that do not track: initialisation, relocalisation)
    m_lastFrameSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    m_stats.t_total += m_lastFrameSeconds;
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
    s.key_frames = (long)std::count_if(m_kfs.begin(), m_kfs.end(), [](const Keyframe& k) { return !k.erased; });      // redundant keyframes leave the map
    s.feature_points = (long)m_landmarks.size();
    return s;
}

TrackerBase::ProcessImageResult HipStereoTracker::processImage(CameraQueueEntry& cam, std::optional<GlobalStateInTime> navResultOdom, std::optional<GlobalStateInTime>,
                                                               std::vector<SensorQueueEntry> const&)
{
    return trackFrame(cam, true, navResultOdom);
}
bool HipStereoTracker::start(SensorQueue&) { return startContext(true); }

TrackerBase::ProcessImageResult HipMonoTracker::processImage(CameraQueueEntry& cam, std::optional<GlobalStateInTime> navResultOdom, std::optional<GlobalStateInTime>,
                                                             std::vector<SensorQueueEntry> const&)
{
    return trackFrame(cam, false, navResultOdom);
}
bool HipMonoTracker::start(SensorQueue&) { return startContext(false); }

}  // namespace LpSlam
