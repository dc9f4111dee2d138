/*
 * oracle/ora.h -- CPU restatement ("oracle") of the lpslam hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (lpslam_amd/, include/) never includes, links or calls anything in oracle/.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in third-party code that is absent from
 * /root/reference (empty submodule external/openvslam, .gitmodules:1-3; g2o @ 691dc51ac7c7,
 * conan-packages/g2o-conan/conanfile.py:6,24-27; OpenCV >= 4.2, CMakeLists.txt:53-58), the
 * reference has no test, fixture or golden vector for keypoints / descriptors / matches / BA
 * (the gtest files under src/test/), and neither the reference nor its dependencies can be
 * compiled in this image.  The functions below restate the *published* algorithms of those
 * projects (OpenVSLAM feature::orb_extractor, match::stereo, optimize::local_bundle_adjuster;
 * OpenCV FAST_t<16>, resize INTER_LINEAR 8u, GaussianBlur 8u fixed point, fastAtan2; g2o
 * OptimizationAlgorithmLevenberg + BlockSolver Schur), anchored on the reference's call sites:
 *   src/Trackers/OpenVSLAMStereoTracker.cpp:293-295  (feed_stereo_frame: the entry to this path)
 *   src/Trackers/OpenVSLAMTracker.cpp:120            (feed_monocular_frame)
 *   src/Trackers/OpenVSLAMTrackerBase.cpp:193-201    (Feature.* / depth_threshold parameters)
 *   src/Interface/LpSlamTypes.h:219-222              (focal_x_baseline semantics)
 */
#ifndef LPSLAM_ORACLE_H
#define LPSLAM_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_MAX_LEVELS 16

/* Same field order as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct {
    float x, y;
    float size;
    float angle;     /* degrees [0,360) */
    float response;  /* FAST score */
    int32_t octave;
    int32_t class_id;
} ora_keypoint;

/* Feature.* of the tracker YAML: src/Trackers/OpenVSLAMTrackerBase.cpp:193-198 */
typedef struct {
    int32_t max_num_keypts;   /* Feature.max_num_keypoints (slamKeypoints, default 1200)  */
    float scale_factor;       /* Feature.scale_factor      (1.2)                          */
    int32_t num_levels;       /* Feature.num_levels        (3 generated; 8 in BASELINE)   */
    int32_t ini_fast_thr;     /* Feature.ini_fast_threshold (20)                          */
    int32_t min_fast_thr;     /* Feature.min_fast_threshold (7)                           */
} ora_orb_params;

/* candidate corner in border-relative coordinates of one pyramid level */
typedef struct {
    int32_t x, y;     /* relative to (19,19) = (min_border_x, min_border_y) */
    int32_t score;
} ora_corner;

/* ---- front end ------------------------------------------------------------------------------ */
void ora_scale_factors(const ora_orb_params* p, float* scale, float* inv_scale);
void ora_pyramid_sizes(int w, int h, const ora_orb_params* p, int* lw, int* lh);
void ora_keypts_per_level(const ora_orb_params* p, int* quota);
void ora_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride);
/* cv::FAST(img, thr, nms=true, TYPE_9_16) on a sub-image; returns count (x,y,score) */
int ora_fast9_16(const uint8_t* img, int w, int h, int stride, int thr, int nms,
                 ora_corner* out, int max_out);
/* all cells of one level (64-px cells, 6-px overlap, 19-px border, ini->min threshold fallback) */
int ora_fast_level(const uint8_t* img, int w, int h, int stride, int ini_thr, int min_thr,
                   ora_corner* out, int max_out);
/* the same with a level-0 mask (0 = masked out): [UPSTREAM] is_in_mask on the cell corners and on every corner found */
int ora_fast_level_masked(const uint8_t* img, int w, int h, int stride, int ini_thr, int min_thr,
                          const uint8_t* mask, int mw, int mh, int mstride, float scale,
                          ora_corner* out, int max_out);
/* quad-tree distribution; writes indices into `cand` of the selected corners, in result order */
int ora_distribute(const ora_corner* cand, int n, int min_x, int max_x, int min_y, int max_y,
                   int num_keypts, int32_t* out_idx, int max_out);
float ora_fast_atan2(float y, float x);
float ora_ic_angle(const uint8_t* img, int stride, int x, int y);
void ora_gauss7x7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
void ora_sincos_deg(float angle_deg, float* s, float* c);
void ora_brief256(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t* desc32);

/* Full extractor (feature::orb_extractor::extract without mask).  pyr_out (optional) receives the
 * pyramid levels tightly packed one after another.  cand_count (optional, num_levels ints) receives
 * the number of FAST candidates per level.  Returns the number of keypoints written. */
int ora_orb_extract(const uint8_t* img, int w, int h, int stride, const ora_orb_params* p,
                    ora_keypoint* kpts, uint8_t* descs, int max_out,
                    uint8_t* pyr_out, int32_t* cand_count);
/* the same with a mask image of the level-0 size (0 = masked out; NULL = none) */
int ora_orb_extract_masked(const uint8_t* img, int w, int h, int stride, const ora_orb_params* p,
                           const uint8_t* mask, int mask_stride,
                           ora_keypoint* kpts, uint8_t* descs, int max_out,
                           uint8_t* pyr_out, int32_t* cand_count);

/* ---- matching ------------------------------------------------------------------------------- */
int ora_hamming256(const uint8_t* a, const uint8_t* b);
/* For every query: best train index (first minimum), best and second-best distance. */
void ora_match_bf_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt,
                       int32_t* best_idx, int32_t* best_dist, int32_t* second_dist);
/* knn2 + max distance + Lowe ratio + optional cross check; returns number of (q,t,dist) triples */
int ora_match_bf(const uint8_t* q, int nq, const uint8_t* t, int nt, int max_dist, float ratio,
                 int cross_check, int32_t* out_q, int32_t* out_t, int32_t* out_d);
/* match::stereo::compute.  pyr_l / pyr_r: arrays of level pointers; returns #valid depths */
int ora_match_stereo(const uint8_t* const* pyr_l, const uint8_t* const* pyr_r,
                     const int* lw, const int* lh, const ora_orb_params* p,
                     const ora_keypoint* kl, const uint8_t* dl, int nl,
                     const ora_keypoint* kr, const uint8_t* dr, int nr,
                     float focal_x_baseline, float true_baseline,
                     float* stereo_x_right, float* depths, int32_t* best_right_idx);

/* projection matching (match::projection): query = predicted position, right-image x (< 0: none), search radius (margin x
 * scale factor of the predicted level) and level range (-1: open) */
typedef struct {
    float x, y, x_right, radius;
    int32_t min_level, max_level;
} ora_proj_query;
int ora_match_projection(const ora_keypoint* kp, const uint8_t* desc, const float* stereo_x_right, int n_kp, int width, int height,
                         const ora_proj_query* q, const uint8_t* q_desc, int nq, int hamming_thr, float lowe_ratio,
                         uint8_t* taken, int32_t* match_idx, int32_t* match_dist);
int ora_match_orientation_filter(const float* angle_q, const float* angle_t, int32_t* match_idx, int nq);
/* match::fuse: best keypoint per projected landmark inside the chi-square gate (no exclusivity) */
int ora_match_fuse(const ora_keypoint* kp, const uint8_t* desc, const float* stereo_x_right, int n_kp, int width, int height,
                   const float* inv_level_sigma_sq, const ora_proj_query* q, const uint8_t* q_desc, int nq, int hamming_thr,
                   int32_t* match_idx, int32_t* match_dist);
/* match::area::match_in_consistent_area: window match with "a better later query takes the keypoint" */
int ora_match_area(const ora_keypoint* kp2, const uint8_t* desc2, int n2, int width, int height,
                   const ora_proj_query* q, const uint8_t* q_desc, int nq, int hamming_thr, float lowe_ratio, int32_t* match_idx);

/* ---- bundle adjustment ---------------------------------------------------------------------- */
typedef struct {
    int32_t pose;      /* index into poses   */
    int32_t point;     /* index into points  */
    double u, v, ur;   /* ur < 0 => monocular (2 rows), else stereo (3 rows) */
    double inv_sigma2; /* information = inv_sigma2 * I */
} ora_ba_obs;

typedef struct {
    double fx, fy, cx, cy, fxb;   /* fxb = focal_x_baseline */
    double huber_mono, huber_stereo; /* sqrt(5.991), sqrt(7.815); <= 0 disables the kernel */
} ora_ba_cam;

typedef struct {
    double chi2_before, chi2_after; /* robust chi2 */
    double lambda;
    int32_t trials;                 /* inner LM trials this iteration */
    int32_t status;                 /* 0 OK, 1 terminate */
} ora_ba_iter_log;

/* poses: n_poses x 7 doubles (qw,qx,qy,qz,tx,ty,tz), world->camera.  fixed[i] != 0 => not optimised.
 * points: n_points x 3.  active[k] == 0 => observation ignored (level 1).  robust != 0 => Huber.
 * Runs `iters` iterations of g2o-style Levenberg-Marquardt with landmark Schur complement.
 * Returns iterations performed. */
int ora_ba_optimize(double* poses, const uint8_t* fixed, int n_poses, double* points, int n_points,
                    const ora_ba_obs* obs, const uint8_t* active, int n_obs, const ora_ba_cam* cam,
                    int robust, int iters, ora_ba_iter_log* log);
/* chi2 (non-robust, = e^T Omega e) and depth sign per observation */
void ora_ba_chi2(const double* poses, const double* points, const ora_ba_obs* obs, int n_obs,
                 const ora_ba_cam* cam, double* chi2, uint8_t* depth_positive);
/* local_bundle_adjuster flow: first_iters robust, classify outliers, second_iters plain;
 * outlier[k] set for observations rejected at the end. */
int ora_ba_local(double* poses, const uint8_t* fixed, int n_poses, double* points, int n_points,
                 const ora_ba_obs* obs, int n_obs, const ora_ba_cam* cam,
                 int first_iters, int second_iters, uint8_t* outlier);
/* motion-only pose optimiser (optimize::pose_optimizer): 4 rounds x 10 iterations, unary edges */
int ora_pose_optimize(double* pose7, const double* points, const ora_ba_obs* obs, int n_obs,
                      const ora_ba_cam* cam, uint8_t* outlier);

/* ---- Sim3 pose graph (g2o types_sim3 + OpenVSLAM graph_optimizer; ora_sim3.c) ----------------- */
typedef struct {
    int32_t i, j;      /* vertices: error = log(meas * S_i * S_j^-1) */
    double meas[8];    /* Sim3 as qw qx qy qz tx ty tz s */
} ora_sim3_edge;
void ora_sim3_exp(const double* update7, double* sim3_out);
void ora_sim3_log(const double* sim3_in, double* log7);
void ora_sim3_mul(const double* a, const double* b, double* out);
void ora_sim3_inv(const double* a, double* out);
double ora_sim3_graph_chi2(const double* verts, const ora_sim3_edge* edges, int n_edges);
/* verts: n x 8 (in/out); fixed[i] != 0 => not optimised; fix_scale != 0 => the scale component of every update is zeroed
 * (stereo / RGBD).  g2o Levenberg, numeric Jacobians, dense solve.  Returns iterations performed. */
int ora_sim3_graph_optimize(double* verts, const uint8_t* fixed, int n, const ora_sim3_edge* edges, int n_edges,
                            int fix_scale, int iters, ora_ba_iter_log* log);

/* Sim3 between two keyframes (optimize::transform_optimizer): matched landmark pair k = landmark 1 in camera-1 coordinates,
 * landmark 2 in camera-2 coordinates, their keypoints and 1/sigma^2 of the keypoints' pyramid levels. */
typedef struct {
    double p1c[3], p2c[3];
    double obs1[2], obs2[2];
    double inv_sigma2_1, inv_sigma2_2;
} ora_sim3_pair;
/* s12: Sim3 camera 2 -> camera 1 (in/out); cam = fx fy cx cy; returns the number of inlier pairs (0 = rejected). */
int ora_sim3_transform_optimize(double* s12, const ora_sim3_pair* pairs, int n, const double* cam1, const double* cam2,
                                double chi_sq, int fix_scale, uint8_t* inlier);

#ifdef __cplusplus
}
#endif
#endif
