// interface.cpp -- LpSlamManager (include/lpslam_manager.h): one-line forwards to LpSlam::SlamManager, like the reference's
// pimpl (/root/reference/src/InterfaceImpl/LpSlamManager.cpp:52-243), plus a plain-C shim of the same calls for
// non-C++ clients and the Python tests.
#include <atomic>
#include "../../include/lpslam_manager.h"
#include "slam_manager.h"
#include "jpeg.h"
#include <cstring>
#include <vector>

namespace LpSlam { LpSlamCameraConfiguration defaultCameraConfiguration(); }

LpSlamCameraConfiguration LpSlamConfiguration::createDefaultCameraConfiguration() { return LpSlam::defaultCameraConfiguration(); }

LpSlamManager::LpSlamManager() { m_impl = new LpSlam::SlamManager(); }
LpSlamManager::~LpSlamManager() { delete m_impl; }
void LpSlamManager::logToFile(char const* filename) { m_impl->logToFile(filename ? filename : ""); }
void LpSlamManager::setLogLevel(LpSlamLogLevel l) { m_impl->setLogLevel(l); }
void LpSlamManager::addOnReconstructionCallback(OnReconstructionCallback_t cb, void* ud) { m_impl->addOnReconstructionCallback(cb, ud); }
void LpSlamManager::addRequestNavDataCallback(RequestNavDataCallback_t cb, void* ud) { m_impl->addRequestNavDataCallback(cb, ud); }
void LpSlamManager::addRequestNavTransformation(RequestNavTransformationCallback_t cb, void* ud) { m_impl->addRequestNavTransformation(cb, ud); }
void LpSlamManager::addOnImageCallback(OnImageCallback_t cb, void* ud) { m_impl->addOnImageCallback(cb, ud); }
void LpSlamManager::updateGlobalReferenceState(LpSlamGlobalStateInTime) {}
void LpSlamManager::addImageFromFile(char const*) {}
void LpSlamManager::addStereoImageFromFiles(char const*, char const*) {}
void LpSlamManager::addMarker(LpSlamMarkerIdentifier, LpSlamMarkerState) {}      // no-op in the reference too (SlamManager.cpp:1311-1312)
bool LpSlamManager::addImageFromBuffer(uint32_t n, LpSlamTimestamp t, uint8_t* b, LpSlamImageDescription d) { return m_impl->addImageFromBuffer(n, t, b, d); }
bool LpSlamManager::addStereoImageFromBuffer(uint32_t n, LpSlamTimestamp t, uint8_t* l, uint8_t* r, LpSlamImageDescription d) { return m_impl->addStereoImageFromBuffer(n, t, l, r, d); }
// src/InterfaceImpl/LpSlamManager.cpp:133-152: the buffer is taken as BGRA (Webots), turned grey with cv::cvtColor(COLOR_BGRA2GRAY) -- fixed-point weights
// B 1868, G 9617, R 4899, >> 14 -- and written as cv::imencode(".jpg") writes it: libjpeg, baseline, quality 95 (host/jpeg.cpp; the
// stream is libjpeg's byte for byte).  8UC1 / 8UC3 buffers are taken as what they say they are.  `bufferOut` is as large as the input.
bool LpSlamManager::compressImage(uint8_t* buffer, LpSlamImageDescription desc, uint8_t* bufferOut, uint32_t* bufferOutSize)
{
    if (!buffer || !bufferOut || !bufferOutSize || desc.width == 0 || desc.height == 0) return false;
    LpSlam::GrayImage g; g.width = (int)desc.width; g.height = (int)desc.height;
    const size_t n = (size_t)desc.width * desc.height;
    g.pixels.resize(n);
    size_t in_bytes = 4 * n;
    if (desc.format == LpSlamImageFormat_8UC1) { std::memcpy(g.pixels.data(), buffer, n); in_bytes = n; }
    else if (desc.format == LpSlamImageFormat_8UC3) {
        const bool bgr = desc.image_conversion == LpSlamImageConversion_BGR2RGB;
        for (size_t i = 0; i < n; ++i) g.pixels[i] = (uint8_t)((buffer[3 * i + (bgr ? 2 : 0)] * 4899 + buffer[3 * i + 1] * 9617 + buffer[3 * i + (bgr ? 0 : 2)] * 1868 + (1 << 13)) >> 14);
        in_bytes = 3 * n;
    } else {
        for (size_t i = 0; i < n; ++i) g.pixels[i] = (uint8_t)((buffer[4 * i] * 1868 + buffer[4 * i + 1] * 9617 + buffer[4 * i + 2] * 4899 + (1 << 13)) >> 14);
    }
    std::vector<uint8_t> out;
    if (!LpSlam::encode_jpeg_gray(g, 95, out) || out.size() > in_bytes) return false;      // (a stream larger than its image: tiny noise images only)
    std::memcpy(bufferOut, out.data(), out.size());
    *bufferOutSize = (uint32_t)out.size();
    return true;
}
void LpSlamManager::setCameraConfiguration(LpSlamCameraConfiguration c) { m_impl->setCameraConfiguration(c); }
bool LpSlamManager::readConfigurationFile(char const* f) { return m_impl->readConfigurationFile(f ? f : ""); }
bool LpSlamManager::readReplayItems(char const* f) { return m_impl->loadReplayItems(f ? f : ""); }
bool LpSlamManager::addSource(char const* n, char const* c) { return m_impl->addSource(n ? n : "", c ? c : ""); }
bool LpSlamManager::addTracker(char const* n, char const* c) { return m_impl->addTracker(n ? n : "", c ? c : ""); }
bool LpSlamManager::addProcessor(char const* n, char const* c) { return m_impl->addProcessor(n ? n : "", c ? c : ""); }
void LpSlamManager::setShowLiveStream(bool) {}
void LpSlamManager::setWriteImageFiles(bool) {}
void LpSlamManager::setRecord(bool) {}
void LpSlamManager::setRecordImages(bool) {}
void LpSlamManager::start() { m_impl->start(); }
void LpSlamManager::stop() { m_impl->stop(); }
LpSlamStatus LpSlamManager::getSlamStatus() { return m_impl->getSlamStatus(); }
void LpSlamManager::mappingAddLaserScan(LpSlamGlobalStateInTime, float*, size_t, float, float, float, float, float, float) {}
unsigned long LpSlamManager::mappingGetMapRawSize() { return 0; }
LpMapInfo LpSlamManager::mappingGetMapRaw(int8_t*, std::size_t) { return LpMapInfo{}; }
std::size_t LpSlamManager::mappingGetFeatures(LpSlamMapBoundary b, LpSlamFeatureEntry* e, std::size_t n, LpSlamMatrix9x9 t) { return m_impl->mappingGetFeatures(b, e, n, t); }
std::size_t LpSlamManager::mappingGetFeaturesCount(LpSlamMapBoundary b) { return m_impl->mappingGetFeaturesCount(b); }
bool LpSlamManager::mappingSetMode(bool e) { return m_impl->mappingSetMode(e); }
bool LpSlamManager::mappingSetFilename(const char* f) { return m_impl->mappingSetFilename(f ? f : ""); }
bool LpSlamManager::mappingExportCSV(const char* f) { return m_impl->mappingExportCSV(f ? f : ""); }

// ---- plain-C shim ----------------------------------------------------------------------------------------------------
extern "C" {
#define LPS_API __attribute__((visibility("default")))
typedef void (*lpslam_c_reconstruction_cb)(const LpSlamGlobalStateInTime* state, void* user);
struct lpslam_c_manager { LpSlamManager mgr; lpslam_c_reconstruction_cb cb = nullptr; void* user = nullptr; std::atomic<uint64_t> n_results{0}, n_valid{0}; };
static void c_trampoline(LpSlamGlobalStateInTime const& s, void* p) { auto* m = static_cast<lpslam_c_manager*>(p); if (m->cb) m->cb(&s, m->user); }

LPS_API lpslam_c_manager* lpslam_manager_create(void) { return new lpslam_c_manager(); }
LPS_API void lpslam_manager_destroy(lpslam_c_manager* m) { delete m; }
LPS_API void lpslam_manager_set_log_level(lpslam_c_manager* m, int level) { m->mgr.setLogLevel((LpSlamLogLevel)level); }
LPS_API void lpslam_manager_log_to_file(lpslam_c_manager* m, const char* f) { m->mgr.logToFile(f); }
LPS_API int lpslam_manager_read_configuration_file(lpslam_c_manager* m, const char* f) { return m->mgr.readConfigurationFile(f); }
LPS_API int lpslam_manager_add_tracker(lpslam_c_manager* m, const char* n, const char* c) { return m->mgr.addTracker(n, c); }
LPS_API int lpslam_manager_add_processor(lpslam_c_manager* m, const char* n, const char* c) { return m->mgr.addProcessor(n, c); }
LPS_API int lpslam_manager_add_source(lpslam_c_manager* m, const char* n, const char* c) { return m->mgr.addSource(n, c); }
LPS_API void lpslam_manager_set_camera_configuration(lpslam_c_manager* m, const LpSlamCameraConfiguration* c) { m->mgr.setCameraConfiguration(*c); }
LPS_API void lpslam_manager_default_camera_configuration(LpSlamCameraConfiguration* out) { *out = LpSlamConfiguration().createDefaultCameraConfiguration(); }
LPS_API void lpslam_manager_on_reconstruction(lpslam_c_manager* m, lpslam_c_reconstruction_cb cb, void* user) { m->cb = cb; m->user = user; m->mgr.addOnReconstructionCallback(c_trampoline, m); }
LPS_API void lpslam_manager_on_image(lpslam_c_manager* m, OnImageCallback_t cb, void* user) { m->mgr.addOnImageCallback(cb, user); }
LPS_API void lpslam_manager_request_nav_data(lpslam_c_manager* m, RequestNavDataCallback_t cb, void* user) { m->mgr.addRequestNavDataCallback(cb, user); }
// A compiled RequestNavDataCallback_t that answers every request with a valid identity odometry (what Manager.provide_odometry's
// Python callback answers): for throughput measurements, where a Python callback per frame on the worker thread costs more than the
// tracker's own host work.
static LpSlamRequestNavDataResult identity_odometry(LpSlamROSTimestamp, LpSlamGlobalStateInTime* odom, LpSlamGlobalStateInTime*, void*) {
    odom->state.valid = true;
    odom->state.orientation.w = 1.0;
    return LpSlamRequestNavDataResult_OdomOnly;
}
LPS_API void lpslam_manager_request_identity_nav_data(lpslam_c_manager* m) { m->mgr.addRequestNavDataCallback(identity_odometry, nullptr); }
// A compiled OnReconstructionCallback_t that only counts (results, valid results): sixteen managers at 2000 results per second each are
// 32 000 interpreter entries per second from sixteen notify threads otherwise -- the measurement would be of the interpreter lock.
static void counting_result(LpSlamGlobalStateInTime const& s, void* p) { auto* m = static_cast<lpslam_c_manager*>(p); m->n_results.fetch_add(1); if (s.state.valid) m->n_valid.fetch_add(1); }
LPS_API void lpslam_manager_count_results(lpslam_c_manager* m) { m->n_results = 0; m->n_valid = 0; m->mgr.addOnReconstructionCallback(counting_result, m); }
LPS_API void lpslam_manager_result_counts(lpslam_c_manager* m, uint64_t* results, uint64_t* valid) { if (results) *results = m->n_results.load(); if (valid) *valid = m->n_valid.load(); }
LPS_API int lpslam_manager_add_stereo_image(lpslam_c_manager* m, uint32_t cam, uint64_t ts, uint8_t* l, uint8_t* r, const LpSlamImageDescription* d) { return m->mgr.addStereoImageFromBuffer(cam, ts, l, r, *d); }
LPS_API int lpslam_manager_add_image(lpslam_c_manager* m, uint32_t cam, uint64_t ts, uint8_t* b, const LpSlamImageDescription* d) { return m->mgr.addImageFromBuffer(cam, ts, b, *d); }
LPS_API int lpslam_manager_compress_image(uint8_t* b, const LpSlamImageDescription* d, uint8_t* out, uint32_t* out_size) { return LpSlamManager::compressImage(b, *d, out, out_size); }
LPS_API int lpslam_manager_read_replay_items(lpslam_c_manager* m, const char* f) { return m->mgr.readReplayItems(f); }
LPS_API void lpslam_manager_start(lpslam_c_manager* m) { m->mgr.start(); }
LPS_API void lpslam_manager_stop(lpslam_c_manager* m) { m->mgr.stop(); }
LPS_API void lpslam_manager_status(lpslam_c_manager* m, LpSlamStatus* out) { *out = m->mgr.getSlamStatus(); }
LPS_API size_t lpslam_manager_features(lpslam_c_manager* m, LpSlamFeatureEntry* e, size_t n, const float* t9) {
    LpSlamMatrix9x9 t; for (int i = 0; i < 9; ++i) t[i] = t9[i];
    return m->mgr.mappingGetFeatures(LpSlamMapBoundary{}, e, n, t);
}
// this manager's own tracker statistics (the log file is process-wide: with several managers in a process their lines share it);
// returns the length of the line, copies at most cap - 1 characters
static_assert(sizeof(LpSlamManager) == sizeof(void*), "LpSlamManager holds m_impl and nothing else (src/Interface/LpSlamManager.h:120)");
LPS_API size_t lpslam_manager_tracker_statistics(lpslam_c_manager* m, char* out, size_t cap) {
    LpSlam::SlamManager* impl = *reinterpret_cast<LpSlam::SlamManager**>(&m->mgr);
    const std::string s = impl ? impl->trackerStatistics() : std::string();
    if (out && cap) { const size_t n = std::min(s.size(), cap - 1); memcpy(out, s.data(), n); out[n] = 0; }
    return s.size();
}
LPS_API size_t lpslam_manager_features_count(lpslam_c_manager* m) { return m->mgr.mappingGetFeaturesCount(LpSlamMapBoundary{}); }
// interface.type_conversion of the reference's tests (src/test/InterfaceTest.cpp:14-33): POD -> internal -> POD
LPS_API void lpslam_roundtrip_state(const LpSlamGlobalStateInTime* in, LpSlamGlobalStateInTime* out) {
    *out = LpSlam::conversion::gsInTimeInternalToInterface(LpSlam::conversion::gsInTimeInterfaceToInternal(*in));
}
}
