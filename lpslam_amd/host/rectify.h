// rectify.h -- undistort / rectify map generation on the host (init only), for the on-device remap (lpslam_hip_set_rectify_map).
// Mirrors the first-call branch of the reference's ImageProcessing::Undistort::undistort
// (/root/reference/src/Utils/ImageProcessing.h:139-243): cv::stereoRectify(CALIB_ZERO_DISPARITY, alpha 0) on the camera pair, then
// cv::initUndistortRectifyMap (pinhole) or cv::fisheye::initUndistortRectifyMap (fisheye) with the eye's own K, D and its R, P.
// The OpenCV routines are restated from their published 4.x sources in FP64 (points OpenCV keeps as CV_32F are rounded to
// float at the same places); two deliberate simplifications, both below 1e-12 on valid input: rotation matrices are taken as
// orthonormal (cvRodrigues2 re-orthogonalises them by SVD first) and 3x3 inverses use the adjugate (fisheye upstream: SVD).
#pragma once
#include <string>
#include <vector>
#include "../../include/lpslam_types.h"

namespace LpSlam {

struct RectifyMaps {
    int width = 0, height = 0;
    std::vector<float> map_x, map_y;     // CV_32FC1, row-major
};

// K row-major 3x3, D distortion coefficients (k1 k2 p1 p2 k3 [k4 k5 k6]), R 3x3, T 3; outputs R1, R2 (3x3), P1, P2 (3x4)
void stereo_rectify(const double* K1, const double* D1, int n1, const double* K2, const double* D2, int n2, int width, int height,
                    const double* R, const double* T, double* R1, double* R2, double* P1, double* P2);
void init_undistort_rectify_map(const double* K, const double* D, int nd, const double* R, const double* P, int width, int height,
                                float* map_x, float* map_y);
void fisheye_init_undistort_rectify_map(const double* K, const double* D4, const double* R, const double* P, int width, int height,
                                        float* map_x, float* map_y);
// The reference's dispatch on leftCam.distortion_function; false (with *err) for no_distortion (nothing to do), omni (the
// reference's branch is commented out there as well) and inconsistent configurations.
bool build_rectify_maps(const LpSlamCameraConfiguration& left, const LpSlamCameraConfiguration& right, bool is_left,
                        RectifyMaps& out, std::string* err);


// Camera mask of one eye (OpenVSLAMTrackerBase::configureMasks, /root/reference/src/Trackers/OpenVSLAMTrackerBase.cpp:331-380):
// mask_type Radial = filled circle of radius int(mask_parameter) around the image centre (cv::circle, FILLED), 255 inside, 0 outside;
// mask_type Image = camera_mask_left.bmp / camera_mask_right.bmp in the working directory, read as grey (cv::imread GRAYSCALE:
// uncompressed 8 / 24 / 32-bit BMP here), must have the camera's resolution.  Row-major width x height, 0 = masked out.
bool build_camera_mask(const LpSlamCameraConfiguration& cam, bool is_left, std::vector<uint8_t>& mask, std::string* err);
}  // namespace LpSlam
