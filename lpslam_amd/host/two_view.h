// two_view.h -- monocular map initialisation from two views (host, init only).
// What openvslam::system does inside feed_monocular_frame until a map exists (reference call site:
// /root/reference/src/Trackers/OpenVSLAMTracker.cpp:120; parameters of the generated configuration,
// src/Trackers/OpenVSLAMTrackerBase.cpp:161-201): [UPSTREAM] initialize::perspective -- homography and fundamental matrix by
// RANSAC over 8-match samples on Hartley-normalised keypoints (solve::homography_solver / solve::fundamental_solver), model
// choice by the score ratio (H if S_H / (S_H + S_F) > 0.40), motion hypotheses from the chosen model (4 from E = K^T F K, 8 from
// H by Faugeras' decomposition), linear triangulation of the inlier matches and the plausibility test of
// initialize::base::find_most_plausible_pose (positive depth, reprojection error, parallax, one clear winner).
// The upstream sources are absent from the reference tree; this restates the published algorithm (ORB-SLAM's Initializer, which
// OpenVSLAM follows) in FP64 with a deterministic sampler.
#pragma once
#include <cstdint>
#include <vector>

namespace LpSlam {

struct TwoViewParams {
    double sigma = 1.0;                  // keypoint standard deviation in pixels (level 0)
    int ransac_iters = 100;              // Initializer.num_ransac_iterations
    int min_triangulated = 50;           // Initializer.num_min_triangulated_pts
    double parallax_deg_thr = 1.0;       // Initializer.parallax_deg_threshold
    double reproj_err_thr = 4.0;         // Initializer.reprojection_error_threshold
    uint32_t seed = 0x9E3779B9u;         // sampler (xorshift32); upstream seeds from std::random_device
};

struct TwoViewResult {
    bool ok = false;
    int model = -1;                      // 0 = homography, 1 = fundamental matrix
    double score_h = 0, score_f = 0;
    double H[9] = {0}, F[9] = {0};       // reference -> current, row-major
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};     // reference -> current (|t| = 1)
    double parallax_deg = 0;
    int n_inliers = 0, n_valid = 0;
    std::vector<uint8_t> inlier;         // per match: inlier of the chosen model
    std::vector<uint8_t> triangulated;   // per match: landmark accepted
    std::vector<double> points;          // per match: X Y Z in the reference camera frame (valid where triangulated)
};

// K = (fx, fy, cx, cy), shared by both views; keypoints (x, y) in pixels; matches = (index in ref, index in cur) pairs
bool two_view_initialize(const double* K, const float* kp_ref, const float* kp_cur, const int32_t* matches, int n_matches,
                         const TwoViewParams& prm, TwoViewResult& out);

// building blocks (exposed for the tests)
void sym_eigen_jacobi(const double* A, int n, double* eigval, double* eigvec);        // A symmetric n x n; eigvec columns, ascending eigval
void homography_from_matches(const double* x1, const double* x2, int n, double* H);   // DLT, x2 ~ H x1 (already normalised points)
void fundamental_from_matches(const double* x1, const double* x2, int n, double* F);  // 8-point + rank 2, x2^T F x1 = 0
bool triangulate_point(const double* P1, const double* P2, const double* x1, const double* x2, double* X);

// [UPSTREAM] solve::sim3_solver (ORB-SLAM Sim3Solver): the similarity (or rigid, fix_scale) transform between two keyframes from
// matched landmarks given in each keyframe's camera frame -- x1 = s R x2 + t -- by Horn's closed-form absolute orientation on
// 3-match samples (RANSAC, xorshift32 sampler as in two_view_initialize), scored by the reprojection error in BOTH images
// (chi-square 9.210 x the keypoint's level sigma^2).  This is what gives a loop candidate a transform WITHOUT trusting the current
// (drifted) pose estimates; the Sim3 optimiser on the device refines it.  p1c / p2c: n x 3, obs1 / obs2: n x 2 pixels,
// inv_sigma2_*: n.  Returns the number of inliers of the best hypothesis (0: none found); s12 = (qw qx qy qz tx ty tz s).
int sim3_solve_ransac(const double* p1c, const double* p2c, const double* obs1, const double* obs2, const double* inv_sigma2_1, const double* inv_sigma2_2,
                      int n, const double* cam1 /* fx fy cx cy */, const double* cam2, bool fix_scale, int iterations, uint32_t seed, double* s12, uint8_t* inlier);
// EPnP (Lepetit, Moreno-Noguer, Fua 2009): world -> camera (R row major, t) from n >= 4 landmark / pixel matches; false for degenerate input
bool epnp_solve(const double* pw, const double* uv, int n, const double* cam /* fx fy cx cy */, double* R, double* t);
// [UPSTREAM] solve::pnp_solver: world -> camera pose from n >= 4 landmark / keypoint matches, no prior: EPnP on 4-match samples (RANSAC),
// inliers by reprojection error (chi-square 5.991 at the keypoint's level), then EPnP once more over the inliers of the best sample.
// pw: n x 3 world points, obs: n x 2 pixels.  Returns the inlier count of the result (0: none); pose7 = qw qx qy qz tx ty tz.
int pnp_solve_ransac(const double* pw, const double* obs, const double* inv_sigma2, int n, const double* cam /* fx fy cx cy */, int iterations, uint32_t seed,
                     double* pose7, uint8_t* inlier);
// Horn's absolute orientation of n >= 3 matched points: x1 = s R x2 + t (R row major)
bool horn_absolute_orientation(const double* x1, const double* x2, int n, bool fix_scale, double* R, double* t, double* s);

}  // namespace LpSlam
