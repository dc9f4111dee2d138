// slam_manager.cpp -- see slam_manager.h.
#include "slam_manager.h"
#include "jpeg.h"
#include "replay.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

namespace LpSlam {

namespace {
std::mutex g_logMutex;
LpSlamLogLevel g_logLevel = LpSlamLogLevel_Error;
std::ofstream g_logFile;
}

// console pattern of the reference: "*** message ***" (src/Manager/SlamManager.cpp:319-325)
void logMessage(LpSlamLogLevel level, const std::string& msg)
{
    std::lock_guard<std::mutex> l(g_logMutex);
    if (level < g_logLevel) return;
    std::cerr << "*** " << msg << " ***" << std::endl;
    if (g_logFile.is_open()) { g_logFile << msg << std::endl; g_logFile.flush(); }
}

SlamManager::SlamManager() {}
SlamManager::~SlamManager() { stop(); }

void SlamManager::logToFile(std::string const& filename)
{
    std::lock_guard<std::mutex> l(g_logMutex);
    if (g_logFile.is_open()) g_logFile.close();
    g_logFile.open(filename, std::ios::app);
}
void SlamManager::setLogLevel(LpSlamLogLevel l) { std::lock_guard<std::mutex> g(g_logMutex); g_logLevel = l; }

// ---- plugin factories: string compare against type(), as in the reference (src/Manager/SlamManager.cpp:393-501) -------
bool SlamManager::addTracker(std::string const& name, std::string const& jsonConfig)
{
    std::unique_ptr<TrackerBase> tracker;
    if (name == "VSLAMStereo") tracker = std::make_unique<HipStereoTracker>();
    if (name == "VSLAMMono") tracker = std::make_unique<HipMonoTracker>();
    if (tracker) {
        if (!tracker->setConfig(jsonConfig)) { logMessage(LpSlamLogLevel_Error, "Cannot parse config for tracker " + tracker->type()); return false; }
        tracker->setCameraRegistry(&m_camRegistry);
        m_vslamTracker = static_cast<HipVslamTrackerBase*>(tracker.get());
        logMessage(LpSlamLogLevel_Info, "Tracker " + tracker->type() + " added");
        m_trackers.emplace_back(std::move(tracker));
        return true;
    }
    logMessage(LpSlamLogLevel_Error, "Tracker with name " + name + " not found");
    return false;
}

bool SlamManager::addProcessor(std::string const& name, std::string const&)
{
    // the reference's processors (BlackoutImage, AdjustIntensity, CameraCalibration) are outside the accelerated path;
    // the ProcessorBase hook is kept for client-side plugins
    logMessage(LpSlamLogLevel_Error, "Processor with name " + name + " not found");
    return false;
}

bool SlamManager::addSource(std::string const& name, std::string const&)
{
    // camera / file / simulator sources are outside the accelerated path: frames arrive through add*ImageFromBuffer
    logMessage(LpSlamLogLevel_Error, "Source with name " + name + " not found");
    return false;
}

// ---- configuration file (schema of src/Manager/SlamManager.cpp:613-1003) -----------------------------------------------
static bool rodrigues(const double v[3], double R[9])
{
    const double th = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (th < 1e-12) { const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; std::memcpy(R, I, sizeof(I)); return true; }
    const double k[3] = {v[0] / th, v[1] / th, v[2] / th}, c = std::cos(th), s = std::sin(th), C = 1 - c;
    const double Rm[9] = {c + k[0] * k[0] * C, k[0] * k[1] * C - k[2] * s, k[0] * k[2] * C + k[1] * s,
                          k[1] * k[0] * C + k[2] * s, c + k[1] * k[1] * C, k[1] * k[2] * C - k[0] * s,
                          k[2] * k[0] * C - k[1] * s, k[2] * k[1] * C + k[0] * s, c + k[2] * k[2] * C};
    std::memcpy(R, Rm, sizeof(Rm));
    return true;
}

LpSlamCameraConfiguration defaultCameraConfiguration()
{
    LpSlamCameraConfiguration c{};
    c.distortion_function = LpSlamCameraDistortionFunction_NoDistortion;
    c.mask_type = LpSlamCameraMaskType_None;
    c.fps = 25.0;                                   // src/InterfaceImpl/LpSlamManager.cpp:35
    c.rotation[0] = c.rotation[4] = c.rotation[8] = 1.0;
    return c;
}

bool SlamManager::readConfigurationFile(std::string const& filename)
{
    logMessage(LpSlamLogLevel_Info, "Loading configuration from " + filename);
    std::ifstream ifs(filename);
    if (!ifs) { logMessage(LpSlamLogLevel_Error, "Cannot open config file " + filename); return false; }
    std::stringstream ss; ss << ifs.rdbuf();
    Json j;
    try { j = Json::parse(ss.str()); }
    catch (JsonError& e) { logMessage(LpSlamLogLevel_Error, "Cannot parse config file " + filename + " because: " + e.what()); return false; }
    try {
        if (const Json* m = j.find("manager")) {
            if (const Json* t = m->find("thread_num")) m_thread_num = (int)t->asNumber();
            if (const Json* r = m->find("require_odometry")) m_requireOdometry = r->asBool();
            if (const Json* c = m->find("replay_chunks")) m_replayChunk = (size_t)std::max(1.0, c->asNumber());   // SlamManager.cpp:668-671
            // record / show_live / record_raw / replay_chunks belong to subsystems outside the accelerated path: accepted, ignored
        }
        auto plugin_list = [&](const char* section, const char* what, auto add) -> bool {
            const Json* list = j.find(section);
            if (!list) return true;
            for (const Json& e : list->arr) {
                if (e.find("_type")) continue;              // disabled entry
                const Json* type = e.find("type");
                if (!type) { logMessage(LpSlamLogLevel_Error, std::string(what) + " entry is missing required field 'type'"); return false; }
                const Json* cfg = e.find("configuration");
                const std::string cfgJson = cfg ? cfg->dump() : std::string();
                if (!add(type->asString(), cfgJson)) {
                    logMessage(LpSlamLogLevel_Error, std::string("Adding ") + what + " of type " + type->asString() + " failed");
                    return false;
                }
            }
            return true;
        };
        if (!plugin_list("trackers", "Tracker", [&](const std::string& n, const std::string& c) { return addTracker(n, c); })) return false;
        if (!plugin_list("processors", "Processor", [&](const std::string& n, const std::string& c) { return addProcessor(n, c); })) return false;
        if (const Json* ds = j.find("datasources")) {
            for (const Json& e : ds->arr) {
                if (e.find("_type")) continue;
                const Json* type = e.find("type");
                if (!type) { logMessage(LpSlamLogLevel_Error, "Datasource entry is missing required field 'type'"); return false; }
                // sources are not part of this library: the entry is validated and skipped (frames come in through the buffer API)
                logMessage(LpSlamLogLevel_Info, "Datasource " + type->asString() + " ignored: frames are fed through add*ImageFromBuffer");
            }
        }
        if (const Json* cams = j.find("cameras")) {
            for (const Json& c : cams->arr) {
                for (const char* req : {"model", "number", "fx", "fy", "cx", "cy", "resolution_x", "resolution_y"})
                    if (!c.find(req)) { logMessage(LpSlamLogLevel_Error, std::string("Camera entry is missing required field '") + req + "'"); return false; }
                LpSlamCameraConfiguration cc = defaultCameraConfiguration();
                cc.f_x = c.find("fx")->asNumber(); cc.f_y = c.find("fy")->asNumber(); cc.c_x = c.find("cx")->asNumber(); cc.c_y = c.find("cy")->asNumber();
                cc.resolution_x = (int)c.find("resolution_x")->asNumber(); cc.resolution_y = (int)c.find("resolution_y")->asNumber();
                if (const Json* f = c.find("fps")) cc.fps = f->asNumber();
                if (const Json* f = c.find("focal_x_baseline")) cc.focal_x_baseline = f->asNumber();
                cc.camera_number = (uint32_t)c.find("number")->asNumber();
                const std::string model = c.find("model")->asString();
                if (model == "fisheye") cc.distortion_function = LpSlamCameraDistortionFunction_Fisheye;
                else if (model == "perspective") cc.distortion_function = LpSlamCameraDistortionFunction_Pinhole;
                else if (model == "omni") cc.distortion_function = LpSlamCameraDistortionFunction_Omni;
                else if (model == "no_distortion") cc.distortion_function = LpSlamCameraDistortionFunction_NoDistortion;
                else { logMessage(LpSlamLogLevel_Error, "Camera model " + model + " not supported"); return false; }
                if (const Json* d = c.find("distortion")) {
                    if (d->arr.size() > LpSlamMaxDistortion) { logMessage(LpSlamLogLevel_Error, "Too many distortion parameter in config file"); return false; }
                    for (size_t i = 0; i < d->arr.size(); ++i) cc.dist[i] = d->arr[i].asNumber();
                }
                if (const Json* r = c.find("rotation")) {
                    if (r->arr.size() > 9) { logMessage(LpSlamLogLevel_Error, "Too many rotation parameter in config file"); return false; }
                    for (size_t i = 0; i < r->arr.size(); ++i) cc.rotation[i] = r->arr[i].asNumber();
                }
                if (const Json* r = c.find("rotation_vec")) {
                    if (r->arr.size() > 3) { logMessage(LpSlamLogLevel_Error, "Too many rotation vector parameter in config file"); return false; }
                    double v[3] = {0, 0, 0};
                    for (size_t i = 0; i < r->arr.size(); ++i) v[i] = r->arr[i].asNumber();
                    rodrigues(v, cc.rotation);
                }
                if (const Json* t = c.find("translation")) {
                    if (t->arr.size() > 3) { logMessage(LpSlamLogLevel_Error, "Too many translation vector parameter in config file"); return false; }
                    for (size_t i = 0; i < t->arr.size(); ++i) cc.translation[i] = t->arr[i].asNumber();
                }
                if (const Json* m = c.find("mask")) {
                    if (m->asString() == "image") cc.mask_type = LpSlamCameraMaskType_Image;
                    else logMessage(LpSlamLogLevel_Error, "Camera mask type " + m->asString() + " not supported");
                }
                setCameraConfiguration(cc);
            }
        }
        if (const Json* markers = j.find("markers"))
            for (const Json& m : markers->arr)
                if (!m.find("type")) { logMessage(LpSlamLogLevel_Error, "Marker entry is missing required field 'type'"); return false; }
    } catch (JsonError& e) {
        logMessage(LpSlamLogLevel_Error, "Cannot read config file " + filename + " because: " + e.what());
        return false;
    }
    return true;
}

// ---- frame ingest (src/Manager/SlamManager.cpp:1038-1297): frames are copied at enqueue --------------------------------
// The copies live in recycled buffers: a megabyte-sized std::vector is an mmap at every enqueue (plus its page faults and the
// zero fill of resize) and a munmap on the worker thread at the end of every frame -- 0.1 ms per stereo frame on either side at
// 1280 x 720.  The worker hands the buffers of a finished frame back (recycleFrame), the next enqueue of the same size takes them.
namespace {
std::mutex g_framePoolMutex;
std::vector<std::vector<uint8_t>> g_framePool;
constexpr size_t kFramePoolMax = 16;
std::vector<uint8_t> takeFrameBuffer(size_t n)
{
    {
        std::lock_guard<std::mutex> l(g_framePoolMutex);
        for (size_t i = g_framePool.size(); i-- > 0;)
            if (g_framePool[i].size() == n) { std::vector<uint8_t> v = std::move(g_framePool[i]); g_framePool.erase(g_framePool.begin() + (long)i); return v; }
    }
    return std::vector<uint8_t>(n);
}
void recycleFrameBuffer(std::vector<uint8_t>&& v)
{
    if (v.size() < (64u << 10)) return;      // small frames: the allocator's own free lists do
    std::lock_guard<std::mutex> l(g_framePoolMutex);
    if (g_framePool.size() < kFramePoolMax) g_framePool.push_back(std::move(v));
}
void recycleFrame(CameraQueueEntry& e)
{
    recycleFrameBuffer(std::move(e.image.pixels));
    if (e.image_second) recycleFrameBuffer(std::move(e.image_second->pixels));
}
}  // namespace

static GrayImage toGray(const uint8_t* buf, const LpSlamImageDescription& d)
{
    GrayImage g; g.width = (int)d.width; g.height = (int)d.height;
    const size_t n = (size_t)d.width * d.height;
    g.pixels = takeFrameBuffer(n);
    if (d.format == LpSlamImageFormat_8UC1) std::memcpy(g.pixels.data(), buf, n);
    else {      // 8UC3, RGB -> grey with cv::cvtColor's fixed-point weights (R 4899, G 9617, B 1868, >> 14)
        const bool bgr = d.image_conversion == LpSlamImageConversion_BGR2RGB;
        for (size_t i = 0; i < n; ++i) {
            const int r = buf[3 * i + (bgr ? 2 : 0)], gg = buf[3 * i + 1], b = buf[3 * i + (bgr ? 0 : 2)];
            g.pixels[i] = (uint8_t)((r * 4899 + gg * 9617 + b * 1868 + (1 << 13)) >> 14);
        }
    }
    return g;
}

bool SlamManager::addStereoImageFromBuffer(uint32_t cameraNumber, LpSlamTimestamp timestamp, uint8_t* left, uint8_t* right, LpSlamImageDescription desc)
{
    if (desc.structure != LpSlamImageStructure_Stereo_TwoBuffer) {
        logMessage(LpSlamLogLevel_Error, "Image structure not supported");
        return true;       // the reference logs and still returns true (SlamManager.cpp:1106-1110)
    }
    if (desc.format != LpSlamImageFormat_8UC3 && desc.format != LpSlamImageFormat_8UC1) { logMessage(LpSlamLogLevel_Error, "Image format not supported"); return false; }
    if (!left || !right || desc.width == 0 || desc.height == 0) { logMessage(LpSlamLogLevel_Error, "Empty stereo frame"); return false; }
    CameraQueueEntry q;
    q.valid = true; q.timestamp = int64ToTimeStamp((int64_t)timestamp);
    q.cameraNumber = cameraNumber; q.cameraNumberSecond = cameraNumber + 1;
    q.image = toGray(left, desc); q.image_second = toGray(right, desc);
    if (desc.hasRosTimestamp > 0) q.ros_timestamp = desc.rosTimestamp;
    m_camQueue.push(std::move(q));
    return true;
}

bool SlamManager::addImageFromBuffer(uint32_t cameraNumber, LpSlamTimestamp timestamp, uint8_t* buffer, LpSlamImageDescription desc)
{
    if (!buffer || (desc.format != LpSlamImageFormat_8UC1_JPEPG && (desc.width == 0 || desc.height == 0))) { logMessage(LpSlamLogLevel_Error, "Empty frame"); return false; }
    CameraQueueEntry q;
    q.valid = true; q.timestamp = int64ToTimeStamp((int64_t)timestamp); q.cameraNumber = cameraNumber;
    if (desc.hasRosTimestamp > 0) q.ros_timestamp = desc.rosTimestamp;
    if (desc.format == LpSlamImageFormat_8UC1_JPEPG) {
        // a compressed frame (LpGlobalFusion / Webots / a recording): cv::imdecode(..., IMREAD_GRAYSCALE) in the reference
        // (SlamManager.cpp:1139-1146), only as LpSlamImageStructure_OneImage (:1138-1152)
        if (desc.structure != LpSlamImageStructure_OneImage) { logMessage(LpSlamLogLevel_Error, "Image format not supported"); return false; }
        std::string why;
        if (desc.imageSize == 0 || !decode_jpeg_gray(buffer, desc.imageSize, q.image, &why)) {
            logMessage(LpSlamLogLevel_Error, "Cannot decode the compressed frame: " + (desc.imageSize ? why : std::string("imageSize is 0")));
            return false;
        }
        m_camQueue.push(std::move(q));
        return true;
    }
    if (desc.format != LpSlamImageFormat_8UC1 && desc.format != LpSlamImageFormat_8UC3) { logMessage(LpSlamLogLevel_Error, "Image format not supported"); return false; }
    const size_t px = (desc.format == LpSlamImageFormat_8UC3) ? 3 : 1;
    if (desc.structure == LpSlamImageStructure_OneImage) q.image = toGray(buffer, desc);
    else if (desc.structure == LpSlamImageStructure_Stereo_LeftTop_RightBottom) {
        LpSlamImageDescription half = desc; half.height = desc.height / 2;
        q.image = toGray(buffer, half);
        q.image_second = toGray(buffer + (size_t)half.height * desc.width * px, half);
        q.cameraNumberSecond = cameraNumber + 1;
    } else if (desc.structure == LpSlamImageStructure_Stereo_LeftLeft_RightRight) {
        // de-interleave the two halves of every row
        LpSlamImageDescription half = desc; half.width = desc.width / 2;
        std::vector<uint8_t> l((size_t)half.width * desc.height * px), r(l.size());
        for (uint32_t y = 0; y < desc.height; ++y) {
            std::memcpy(&l[(size_t)y * half.width * px], buffer + (size_t)y * desc.width * px, (size_t)half.width * px);
            std::memcpy(&r[(size_t)y * half.width * px], buffer + ((size_t)y * desc.width + half.width) * px, (size_t)half.width * px);
        }
        q.image = toGray(l.data(), half); q.image_second = toGray(r.data(), half);
        q.cameraNumberSecond = cameraNumber + 1;
    } else { logMessage(LpSlamLogLevel_Error, "Image structure not supported"); return false; }
    m_camQueue.push(std::move(q));
    return true;
}

// ---- replay (src/Manager/ReplayEngine.cpp:61-242) ---------------------------------------------------------------------------
// As in the reference, the file is streamed: `replay_chunks` camera records (manager section of the configuration, 500 by
// default) are decoded now, and the worker asks for the next chunk whenever the camera queue has drained below half a chunk
// (ReplayEngine.cpp:77-100), so a long recording is never resident as a whole.  Replayed frames carry no ROS time stamp, so the
// navigation callback is not asked for them (SlamManager.cpp:148): trackers see them only with "require_odometry": false.
bool SlamManager::loadReplayItems(std::string const& filename)
{
    logMessage(LpSlamLogLevel_Info, "Loading replay items from file " + filename);
    {
        ReplayReader r;
        std::string err;
        if (!r.open(filename, &err)) { logMessage(LpSlamLogLevel_Info, "Cannot load replay from file " + filename); return false; }
        std::lock_guard<std::mutex> l(m_replayMutex);
        m_replay = std::move(r);
    }
    streamMoreReplayItems();
    return true;
}

void SlamManager::streamMoreReplayItems()
{
    std::lock_guard<std::mutex> l(m_replayMutex);
    if (m_replay.done() || m_camQueue.size() >= (m_replayChunk + 1) / 2) return;
    size_t loaded = 0;
    ReplayFrame f;
    while (loaded < m_replayChunk && m_replay.next(f)) {
        CameraQueueEntry q;
        q.valid = true; q.timestamp = int64ToTimeStamp(f.timestamp);
        q.cameraNumber = (uint32_t)f.camera; q.cameraNumberSecond = (uint32_t)f.camera_second;
        q.image = std::move(f.image);
        if (f.image_second) q.image_second = std::move(*f.image_second);
        m_camQueue.push(std::move(q));
        ++loaded;
    }
    const ReplayStats& st = m_replay.stats();
    logMessage(LpSlamLogLevel_Info, "Loaded " + std::to_string(loaded) + " replay items (" + std::to_string(st.records) + " records so far, " +
               std::to_string(st.undecodable_images) + " frames with an image codec this library does not carry" +
               (st.truncated ? ", stream truncated)" : m_replay.done() ? ", end of stream)" : ")"));
}

// ---- threads (src/Manager/SlamManager.cpp:54-257) ----------------------------------------------------------------------
bool SlamManager::workerStep()
{
    streamMoreReplayItems();                           // SlamManager.cpp:56-57
    CameraQueueEntry cam;
    if (m_lookahead) { cam = std::move(*m_lookahead); m_lookahead.reset(); }
    else m_camQueue.pop(cam);
    const auto t_taken = std::chrono::steady_clock::now();
    if (!cam.valid || m_stopRequested.load()) return false;   // exit signal; a stop abandons the backlog (SlamManager::stop)
    // every frame the worker takes also goes to the image-callback thread (SlamManager.cpp:64-66); a copy: the tracker consumes `cam`
    if (m_pushToImageCallbackQueue) {
        CameraQueueEntry copy;
        copy.valid = true; copy.timestamp = cam.timestamp; copy.cameraNumber = cam.cameraNumber; copy.cameraNumberSecond = cam.cameraNumberSecond;
        copy.image = cam.image; copy.image_second = cam.image_second;
        m_imageCallbackQueue.push(std::move(copy));
    }
    // one frame of lookahead, owned by this thread: if another frame is already queued the trackers learn about it, and may start
    // its upload and extraction on the GPU beside the tracking of this frame
    {
        CameraQueueEntry next;
        if (m_camQueue.try_pop(next)) m_lookahead = std::move(next);
    }
    const CameraQueueEntry* next_frame = (m_lookahead && m_lookahead->valid) ? &*m_lookahead : nullptr;
    const auto now = std::chrono::steady_clock::now();
    if (m_lastFrame) {                                  // m_lastFrame belongs to this thread; the rate is read by getSlamStatus
        const double dt = std::chrono::duration<double>(now - *m_lastFrame).count();
        if (dt > 0) m_currentFps.store(0.9 * m_currentFps.load() + 0.1 / dt);
    }
    m_lastFrame = now;

    std::vector<SensorQueueEntry> sensors;
    SensorQueueEntry se;
    while (m_sensorQueue.try_pop(se)) { sensors.push_back(se); if (se.timestamp > cam.timestamp) break; }
    for (auto& p : m_processors) p->processImage(cam);

    std::optional<GlobalStateInTime> odom, map;
    if (m_requestNavData != nullptr && cam.ros_timestamp.has_value()) {
        LpSlamGlobalStateInTime lpOdom{}, lpMap{};
        const auto r = m_requestNavData(*cam.ros_timestamp, &lpOdom, &lpMap, m_requestNavDataData);
        if (r == LpSlamRequestNavDataResult_OdomOnly || r == LpSlamRequestNavDataResult_OdomAndMap) odom = conversion::gsInTimeInterfaceToInternal(lpOdom);
        if (r == LpSlamRequestNavDataResult_MapOnly || r == LpSlamRequestNavDataResult_OdomAndMap) map = conversion::gsInTimeInterfaceToInternal(lpMap);
    }
    bool resultSent = false;
    for (auto& tracker : m_trackers) {
        if (m_requireOdometry && !odom.has_value()) {
            logMessage(LpSlamLogLevel_Info, "No vehicle odometry, skipping frame");
            ++m_framesSkipped;
            break;
        }
        tracker->setNextFrame(next_frame);
        const auto t_in = std::chrono::steady_clock::now();
        auto results = tracker->processImage(cam, odom, map, sensors);
        m_secondsInTrackers += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in).count();
        tracker->setNextFrame(nullptr);
        for (auto& tr : results) {
            GlobalStateInTime st;
            st.first = tr.timestamp;
            st.second.position = tr.position; st.second.orientation = tr.orientation; st.second.stateValid = true;
            for (auto& p : m_processors) p->processSensorValuesAndResults(sensors, st);
            ResultQueueEntry re; re.globalStateInTime = st;
            m_resultQueue.push(re);
            resultSent = true;
        }
    }
    if (!resultSent) {      // clients learn that the frame was consumed
        ResultQueueEntry re; re.globalStateInTime.second.stateValid = false; re.globalStateInTime.first.system_time = cam.timestamp;
        m_resultQueue.push(re);
    }
    ++m_framesProcessed;
    recycleFrame(cam);
    m_secondsInWorker += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_taken).count();
    return true;
}

bool SlamManager::notifyStep()
{
    ResultQueueEntry r;
    m_resultQueue.pop(r);
    if (r.exitSignal) return false;
    if (m_onReconstruction != nullptr) m_onReconstruction(conversion::gsInTimeInternalToInterface(r.globalStateInTime), m_onReconstructionData);
    return true;
}

// The image-callback thread (src/Manager/SlamManager.cpp:258-314): every frame the worker took is compressed -- cv::imencode(".jpg",
// IMWRITE_JPEG_QUALITY 70), the two eyes of a stereo frame side by side in two threads -- and handed to OnImageCallback_t as ONE buffer
// (left stream, then right stream; desc.imageSize / imageSizeSecond give the split; structure OneImage_Compressed / Stereo_Compressed,
// format 8UC1_JPEPG, conversion None).  The reference leaves the other members of `desc` uninitialised; here width / height carry the
// frame's size and the rest is zero.
bool SlamManager::imageCallbackStep()
{
    CameraQueueEntry q;
    m_imageCallbackQueue.pop(q);
    if (!q.valid) return false;
    if (m_onImage == nullptr || q.image.empty()) return true;
    std::vector<uint8_t> left, right;
    const int quality = 70;
    std::thread t_left([&] { encode_jpeg_gray(q.image, quality, left); });
    if (q.image_second.has_value()) {
        std::thread t_right([&] { encode_jpeg_gray(*q.image_second, quality, right); });
        t_right.join();
    }
    t_left.join();
    LpSlamImageDescription desc{};
    desc.width = (uint32_t)q.image.width; desc.height = (uint32_t)q.image.height;
    desc.imageSize = (uint32_t)left.size(); desc.imageSizeSecond = 0;
    desc.image_conversion = LpSlamImageConversion_None;
    desc.structure = LpSlamImageStructure_OneImage_Compressed;
    desc.format = LpSlamImageFormat_8UC1_JPEPG;
    if (q.image_second.has_value()) {
        desc.imageSizeSecond = (uint32_t)right.size();
        desc.structure = LpSlamImageStructure_Stereo_Compressed;
        left.insert(left.end(), right.begin(), right.end());
    }
    m_onImage((LpSlamTimestamp)timeStampToInt64(q.timestamp), q.cameraNumber, left.data(), desc, m_onImageData);
    ++m_imagesSent;
    return true;
}

void SlamManager::start()
{
    if (m_running) return;
    for (auto& t : m_trackers) {
        if (!t->start(m_sensorQueue)) logMessage(LpSlamLogLevel_Error, "Tracker " + t->type() + " failed to start");
        t->addRequestNavTransformationCallback(m_requestNavTransformation, m_requestNavTransformationData);
    }
    m_running = true;
    m_stopRequested.store(false);
    // the reference arms the image queue with `m_onRecoCallback != nullptr` (the WorkerThreadParams initialiser, SlamManager.cpp:532-548,
    // puts that expression into pushToImageCallbackQueue) and its thread calls the image callback when that one is set: images reach a
    // client that has set BOTH callbacks.  Same condition here, without queueing frames nobody will be handed.
    m_pushToImageCallbackQueue = m_onReconstruction != nullptr && m_onImage != nullptr;
    m_worker = std::thread([this] { while (workerStep()) {} });
    m_notifyWorker = std::thread([this] { while (notifyStep()) {} });
    m_imageCallbackWorker = std::thread([this] { while (imageCallbackStep()) {} });
}

void SlamManager::stop()
{
    if (!m_running) return;
    // the reference stops its worker and clears the camera queue (SlamManager::stop -> stopAsync + m_camQueue.clear()): frames
    // still queued are dropped, not tracked
    m_stopRequested.store(true);
    m_camQueue.clear();
    CameraQueueEntry poison; poison.valid = false;
    m_camQueue.push(std::move(poison));
    if (m_worker.joinable()) m_worker.join();
    m_lookahead.reset();                                // the frame the worker had taken ahead is part of the abandoned backlog
    m_camQueue.clear();                                 // ... and so is the exit signal if the worker left on the stop flag
    ResultQueueEntry rp; rp.exitSignal = true;
    m_resultQueue.push(rp);
    if (m_notifyWorker.joinable()) m_notifyWorker.join();
    CameraQueueEntry ip; ip.valid = false;              // SlamManager.cpp:586-589: frames already handed to the image thread are still sent
    m_imageCallbackQueue.push(std::move(ip));
    if (m_imageCallbackWorker.joinable()) m_imageCallbackWorker.join();
    m_imageCallbackQueue.clear();
    for (auto& t : m_trackers) t->stop();
    if (const uint64_t n = m_framesProcessed.load())
        logMessage(LpSlamLogLevel_Info, "Worker statistics: frames=" + std::to_string(n) + " ms_per_frame=" + std::to_string(1e3 * m_secondsInWorker / (double)n) +
                   " ms_in_trackers=" + std::to_string(1e3 * m_secondsInTrackers / (double)n));
    m_running = false;
}

LpSlamStatus SlamManager::getSlamStatus()
{
    LpSlamStatus s{};
    s.localization = LpSlamLocalization_Off;
    if (m_vslamTracker) s = m_vslamTracker->getSlamStatus();
    s.fps = m_currentFps.load();
    return s;
}

std::size_t SlamManager::mappingGetFeatures(LpSlamMapBoundary b, LpSlamFeatureEntry* e, std::size_t n, LpSlamMatrix9x9 t) { return m_vslamTracker ? m_vslamTracker->mappingGetFeatures(b, e, n, t) : 0; }
std::string SlamManager::trackerStatistics() { return m_vslamTracker ? m_vslamTracker->lastStatistics() : std::string(); }
std::size_t SlamManager::mappingGetFeaturesCount(LpSlamMapBoundary b) { return m_vslamTracker ? m_vslamTracker->mappingGetFeaturesCount(b) : 0; }
bool SlamManager::mappingSetMode(bool enable) { return m_vslamTracker ? m_vslamTracker->mappingSetMode(enable) : false; }
bool SlamManager::mappingSetFilename(std::string const& f) { return m_vslamTracker ? m_vslamTracker->mappingSetFilename(f) : false; }
bool SlamManager::mappingExportCSV(std::string const& f) { return m_vslamTracker ? m_vslamTracker->mappingExportCSV(f) : false; }

}  // namespace LpSlam
