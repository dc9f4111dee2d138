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
    /* life cycle: configure, start(), feed frames from one producer thread, stop()  (reference :19-20, :79-80) */
    LpSlamManager();  ~LpSlamManager();
    void start();  void stop();

    /* configuration: cameras 0 / 1 (stereo: left = n, right = n + 1), JSON file with the reference's schema, plugins by name
       (reference :62-71; factories src/Manager/SlamManager.cpp:393-501) */
    void setCameraConfiguration(LpSlamCameraConfiguration camera);
    bool readConfigurationFile(char const* json_path);
    bool addTracker(char const* plugin, char const* json);   bool addProcessor(char const* plugin, char const* json);
    bool addSource(char const* plugin, char const* json);    bool readReplayItems(char const* recording_path);
    void setLogLevel(LpSlamLogLevel level);                   void logToFile(char const* path);

    /* callbacks, one slot each, set before start(): poses leave on the notify thread, navigation data is asked for on the
       worker thread once per frame (reference :27-42; src/Manager/SlamManager.cpp:148-156,240-257) */
    void addOnReconstructionCallback(OnReconstructionCallback_t on_pose, void* user);
    void addRequestNavDataCallback(RequestNavDataCallback_t nav_request, void* user);
    void addRequestNavTransformation(RequestNavTransformationCallback_t transform_request, void* user);
    void addOnImageCallback(OnImageCallback_t on_image, void* user);
    void updateGlobalReferenceState(LpSlamGlobalStateInTime reference_state);

    /* frames in: pixel buffers are copied at the call (the reference aliases 8UC1 stereo buffers until processed) */
    bool addStereoImageFromBuffer(uint32_t left_camera, LpSlamTimestamp t_ns, uint8_t* left, uint8_t* right, LpSlamImageDescription layout);
    bool addImageFromBuffer(uint32_t camera, LpSlamTimestamp t_ns, uint8_t* pixels, LpSlamImageDescription layout);

    /* state out: tracker status, landmarks of the map in lpslam axes */
    LpSlamStatus getSlamStatus();
    std::size_t mappingGetFeaturesCount(LpSlamMapBoundary region);
    std::size_t mappingGetFeatures(LpSlamMapBoundary region, LpSlamFeatureEntry* out, std::size_t capacity, LpSlamMatrix9x9 rotation);
    bool mappingExportCSV(const char* csv_path);

    /* belong to subsystems outside the accelerated path; signatures kept, behaviour of the reference without the backing plugin */
    void addImageFromFile(char const* image_path);  void addStereoImageFromFiles(char const* left_path, char const* right_path);
    void addMarker(LpSlamMarkerIdentifier marker, LpSlamMarkerState state);
    static bool compressImage(uint8_t* pixels, LpSlamImageDescription layout, uint8_t* jpeg_out, uint32_t* jpeg_size);
    void setShowLiveStream(bool on);  void setWriteImageFiles(bool on);  void setRecord(bool on);  void setRecordImages(bool on);
    bool mappingSetMode(bool mapping_on);  bool mappingSetFilename(const char* map_path);
    unsigned long mappingGetMapRawSize();  LpMapInfo mappingGetMapRaw(int8_t* cells, std::size_t capacity);
    void mappingAddLaserScan(LpSlamGlobalStateInTime origin, float* ranges, size_t n_ranges, float range_min, float range_max,
                             float angle_min, float angle_max, float angle_step, float range_threshold);

private:
    LpSlam::SlamManager* m_impl;       /* the single data member, as in the reference (pimpl) */
};

#endif
