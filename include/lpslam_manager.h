/*
 * lpslam_manager.h -- the public object of the drop-in boundary.
 *
 * Same class name, method names, signatures and single data member as the reference's exported class
 * (/root/reference/src/Interface/LpSlamManager.h:17-121, C++ ABI, pimpl), so existing client code compiles unchanged;
 * behind it sits LpSlam::SlamManager of lpslam_amd/host/ whose trackers run on the MI355X through include/lpslam_hip.h.
 * Methods that belong to subsystems outside the accelerated path (file/replay sources, recording, laser/occupancy map,
 * live view) keep their signatures and behave as the reference does when the backing plugin is absent: they return
 * false / 0 / do nothing (cf. src/Manager/SlamManager.cpp:1311-1312,1368-1395).
 */
#ifndef LPSLAM_AMD_MANAGER_H
#define LPSLAM_AMD_MANAGER_H

#include "lpslam_types.h"
#include <string>

#ifndef LPSLAM_EXPORT
#define LPSLAM_EXPORT __attribute__((visibility("default")))
#endif

namespace LpSlam { class SlamManager; }

class LPSLAM_EXPORT LpSlamConfiguration {
public:
    LpSlamCameraConfiguration createDefaultCameraConfiguration();       /* src/Interface/LpSlamConfiguration.h:11-14 */
};

class LPSLAM_EXPORT LpSlamManager {
public:
    LpSlamManager();
    ~LpSlamManager();

    void logToFile(char const* filename);
    void setLogLevel(LpSlamLogLevel);

    void addOnReconstructionCallback(OnReconstructionCallback_t callback, void* userData);      /* notify thread */
    void addRequestNavDataCallback(RequestNavDataCallback_t callback, void* userData);          /* worker thread, per frame */
    void addRequestNavTransformation(RequestNavTransformationCallback_t callback, void* userData);
    void addOnImageCallback(OnImageCallback_t callback, void* userData);
    void updateGlobalReferenceState(LpSlamGlobalStateInTime globalStateInTime);

    void addImageFromFile(char const* filename);
    void addStereoImageFromFiles(char const* filename_left, char const* filename_right);
    void addMarker(LpSlamMarkerIdentifier id, LpSlamMarkerState state);

    /* stereo: the second camera is cameraNumber + 1 */
    bool addImageFromBuffer(uint32_t cameraNumber, LpSlamTimestamp timestamp, uint8_t* buffer, LpSlamImageDescription desc);
    bool addStereoImageFromBuffer(uint32_t cameraNumber, LpSlamTimestamp timestamp, uint8_t* buffer_left, uint8_t* buffer_right,
                                  LpSlamImageDescription desc);
    static bool compressImage(uint8_t* buffer, LpSlamImageDescription desc, uint8_t* bufferOut, uint32_t* bufferOutSize);

    void setCameraConfiguration(LpSlamCameraConfiguration conf);
    bool readConfigurationFile(char const* filename);
    bool readReplayItems(char const* filename);
    bool addSource(char const* name, char const* config);
    bool addTracker(char const* name, char const* config);
    bool addProcessor(char const* name, char const* config);
    void setShowLiveStream(bool b);
    void setWriteImageFiles(bool b);
    void setRecord(bool b);
    void setRecordImages(bool b);

    void start();
    void stop();
    LpSlamStatus getSlamStatus();

    void mappingAddLaserScan(LpSlamGlobalStateInTime origin, float* ranges, size_t rangeCount, float start_range, float end_range,
                             float start_angle, float end_angle, float increment, float range_threshold);
    unsigned long mappingGetMapRawSize();
    LpMapInfo mappingGetMapRaw(int8_t* map, std::size_t mapSize);
    std::size_t mappingGetFeatures(LpSlamMapBoundary boundary, LpSlamFeatureEntry* entry, std::size_t entry_count, LpSlamMatrix9x9 transform);
    std::size_t mappingGetFeaturesCount(LpSlamMapBoundary boundary);
    bool mappingSetMode(bool enableMapping);
    bool mappingSetFilename(const char* filename);
    bool mappingExportCSV(const char* csv_filename);

private:
    LpSlam::SlamManager* m_impl;
};

#endif
