/*
 * lpslam_types.h -- interface PODs of the drop-in boundary.
 *
 * Layout- and name-compatible re-statement of the reference's public types (/root/reference/src/Interface/LpSlamTypes.h:6-245):
 * a client compiled against the reference header can be relinked against this library.  The layouts are pinned by the
 * static_asserts at the end (x86-64 Linux, values from SURVEY.md Appendix A).  Non-C features of the reference header are
 * kept on purpose (default member initialisers, a `using` alias, a reference parameter in one callback); the plain-C
 * layer beneath is include/lpslam_hip.h.
 */
#ifndef LPSLAM_AMD_TYPES_H
#define LPSLAM_AMD_TYPES_H

#include <cstddef>
#include <cstdint>

/* ---- geometry ------------------------------------------------------------------------------------------------- */
struct LpSlamPosition { double x, y, z; double x_sigma, y_sigma, z_sigma; };          /* :6-9  x up/fwd, y right, z fwd (see tracker) */
struct LpSlamMapPosition { double y; double z; };                                       /* :11-16 +y right, +z forward */
struct LpSlamFeaturePosition { float x; float y; float z; };                            /* :18-24 */
typedef float LpSlamMatrix9x9[9];                                                        /* :26 (3x3 row-major despite the name) */
struct LpSlamMapBoundary { LpSlamMapPosition top_left; LpSlamMapPosition bottom_right; };
struct LpSlamFeatureEntry { LpSlamFeaturePosition position; };
struct LpSlamMapEntry { LpSlamMapPosition position; float occupancy; };                  /* 0 free .. 1 occupied */
struct LpSlamOrientation { double w, x, y, z; double sigma; };                           /* :89-92 */

/* ---- enums (values = declaration order from 0) ------------------------------------------------------------------ */
enum LpSlamLocalization { LpSlamLocalization_Off, LpSlamLocalization_Initializing, LpSlamLocalization_Tracking, LpSlamLocalization_Lost };
enum LpSlamRequestNavDataResult { LpSlamRequestNavDataResult_None, LpSlamRequestNavDataResult_OdomOnly,
                                  LpSlamRequestNavDataResult_MapOnly, LpSlamRequestNavDataResult_OdomAndMap };
enum LpSlamNavDataFrame { LpSlamNavDataFrame_Camera, LpSlamNavDataFrame_Laser, LpSlamNavDataFrame_Odometry };
enum LpSlamLogLevel { LpSlamLogLevel_Debug, LpSlamLogLevel_Info, LpSlamLogLevel_Error };
enum LpSlamImageStructure {
    LpSlamImageStructure_OneImage,
    LpSlamImageStructure_Stereo_LeftTop_RightBottom,   /* left image above the right one */
    LpSlamImageStructure_Stereo_LeftLeft_RightRight,   /* side by side */
    LpSlamImageStructure_Stereo_TwoBuffer,             /* two separate buffers (the path this library accelerates) */
    LpSlamImageStructure_OneImage_Compressed,
    LpSlamImageStructure_Stereo_Compressed             /* imageSize + imageSizeSecond bytes back to back */
};
enum LpSlamImageFormat { LpSlamImageFormat_8UC1_JPEPG, LpSlamImageFormat_8UC1, LpSlamImageFormat_8UC3, LpSlamImageFormat_8UC4,
                         LpSlamImageFormat_NV12, LpSlamImageFormat_YUV16 };
enum LpSlamImageConversion { LpSlamImageConversion_None, LpSlamImageConversion_BGR2RGB };
enum LpSlamCameraDistortionFunction { LpSlamCameraDistortionFunction_Pinhole, LpSlamCameraDistortionFunction_Fisheye,
                                      LpSlamCameraDistortionFunction_Omni, LpSlamCameraDistortionFunction_NoDistortion };
enum LpSlamCameraMaskType { LpSlamCameraMaskType_None, LpSlamCameraMaskType_Radial, LpSlamCameraMaskType_Image };

/* ---- status / state ---------------------------------------------------------------------------------------------- */
struct LpMapInfo { float x_cell_size; float y_cell_size; uint32_t x_cell_count; uint32_t y_cell_count; float x_origin; float y_origin; };
struct LpSlamStatus { LpSlamLocalization localization; long feature_points; long key_frames; double frame_time /* s */; double fps; };
typedef uint64_t LpSlamTimestamp;
struct LpSlamROSTimestamp { int32_t seconds = 0; int64_t nanoseconds = 0; };             /* ROS2 representation */
struct LpSlamGlobalState { LpSlamPosition position; LpSlamOrientation orientation; bool valid; };
struct LpSlamGlobalStateInTime { int64_t timestamp; LpSlamROSTimestamp ros_timestamp; uint8_t has_ros_timestamp; LpSlamGlobalState state; };
using LpSlamRequestNavTransformation = LpSlamGlobalState;

/* ---- frames and cameras -------------------------------------------------------------------------------------------- */
struct LpSlamImageDescription {
    LpSlamImageStructure structure; LpSlamImageFormat format; LpSlamImageConversion image_conversion;
    uint32_t height; uint32_t width;
    uint32_t imageSize; uint32_t imageSizeSecond;      /* compressed payload sizes */
    uint8_t hasRosTimestamp; LpSlamROSTimestamp rosTimestamp;
};
typedef uint32_t LpSlamMarkerIdentifier;
struct LpSlamMarkerState { LpSlamPosition position; LpSlamOrientation orientation; };
typedef uint32_t LpSlamCameraNumber;
const uint32_t LpSlamMaxDistortion = 8;
struct LpSlamCameraConfiguration {
    LpSlamCameraNumber camera_number; LpSlamCameraDistortionFunction distortion_function;
    double f_x, f_y, c_x, c_y; double dist[LpSlamMaxDistortion];
    LpSlamCameraMaskType mask_type; double mask_parameter;
    int resolution_x, resolution_y; double fps;
    double focal_x_baseline;           /* -P2[0][3] of the stereo calibration; true baseline = focal_x_baseline / f_x (:219-222) */
    double rotation[9]; double translation[3];   /* camera pose relative to the rectified stereo plane */
};

/* ---- callbacks (one slot each; set before start(); threads: see DESIGN.md) ------------------------------------------ */
typedef void (*OnReconstructionCallback_t)(LpSlamGlobalStateInTime const& reconstructedState, void*);
typedef void (*OnImageCallback_t)(LpSlamTimestamp timestamp, uint32_t cameraNumber, uint8_t* buffer, LpSlamImageDescription desc, void*);
typedef LpSlamRequestNavDataResult (*RequestNavDataCallback_t)(LpSlamROSTimestamp for_ros_time, LpSlamGlobalStateInTime* odometry,
                                                               LpSlamGlobalStateInTime* map, void*);
typedef LpSlamRequestNavTransformation (*RequestNavTransformationCallback_t)(LpSlamROSTimestamp ros_time, LpSlamNavDataFrame from_frame,
                                                                             LpSlamNavDataFrame to_frame, void*);

/* ---- ABI pins --------------------------------------------------------------------------------- */
static_assert(sizeof(LpSlamPosition) == 48 && sizeof(LpSlamOrientation) == 40, "pose PODs");
static_assert(sizeof(LpSlamGlobalState) == 96 && offsetof(LpSlamGlobalState, orientation) == 48 && offsetof(LpSlamGlobalState, valid) == 88, "LpSlamGlobalState");
static_assert(sizeof(LpSlamROSTimestamp) == 16 && offsetof(LpSlamROSTimestamp, nanoseconds) == 8, "LpSlamROSTimestamp");
static_assert(sizeof(LpSlamGlobalStateInTime) == 128 && offsetof(LpSlamGlobalStateInTime, ros_timestamp) == 8 &&
              offsetof(LpSlamGlobalStateInTime, has_ros_timestamp) == 24 && offsetof(LpSlamGlobalStateInTime, state) == 32, "LpSlamGlobalStateInTime");
static_assert(sizeof(LpSlamImageDescription) == 48 && offsetof(LpSlamImageDescription, height) == 12 && offsetof(LpSlamImageDescription, imageSize) == 20 &&
              offsetof(LpSlamImageDescription, hasRosTimestamp) == 28 && offsetof(LpSlamImageDescription, rosTimestamp) == 32, "LpSlamImageDescription");
static_assert(sizeof(LpSlamCameraConfiguration) == 240 && offsetof(LpSlamCameraConfiguration, f_x) == 8 && offsetof(LpSlamCameraConfiguration, dist) == 40 &&
              offsetof(LpSlamCameraConfiguration, mask_type) == 104 && offsetof(LpSlamCameraConfiguration, resolution_x) == 120 &&
              offsetof(LpSlamCameraConfiguration, fps) == 128 && offsetof(LpSlamCameraConfiguration, focal_x_baseline) == 136 &&
              offsetof(LpSlamCameraConfiguration, rotation) == 144 && offsetof(LpSlamCameraConfiguration, translation) == 216, "LpSlamCameraConfiguration");
static_assert(sizeof(LpSlamStatus) == 40 && offsetof(LpSlamStatus, feature_points) == 8 && offsetof(LpSlamStatus, fps) == 32, "LpSlamStatus");
static_assert(sizeof(LpMapInfo) == 24 && sizeof(LpSlamMapEntry) == 24 && sizeof(LpSlamFeatureEntry) == 12 && sizeof(LpSlamMapBoundary) == 32 &&
              sizeof(LpSlamMarkerState) == 88 && sizeof(LpSlamMatrix9x9) == 36, "map PODs");
static_assert(LpSlamImageStructure_Stereo_TwoBuffer == 3 && LpSlamImageFormat_8UC1 == 1 && LpSlamImageFormat_8UC3 == 2 &&
              LpSlamRequestNavDataResult_OdomOnly == 1 && LpSlamCameraDistortionFunction_NoDistortion == 3, "enum values");

#endif
