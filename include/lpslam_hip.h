/*
 * lpslam_hip.h -- C ABI of the MI355X (gfx950) hot path of lpslam: ORB extraction, descriptor matching and
 * SE3 bundle adjustment.  This is the "thin HIP C-ABI layer" beneath the C++ host mirror of the reference's
 * Interface / Tracker plugin surface (lpslam_amd/host/).  Plain pointers and sizes only; every entry point
 * returns an int status (0 = LPSLAM_HIP_OK) and never throws; lpslam_hip_last_error() gives the message.
 *
 * What each group replaces in the reference (paths relative to /root/reference):
 *   context / frames    openvslam::system construction + feed_stereo_frame / feed_monocular_frame
 *                       (src/Trackers/OpenVSLAMTrackerBase.cpp:238-239,
 *                        src/Trackers/OpenVSLAMStereoTracker.cpp:293-295, src/Trackers/OpenVSLAMTracker.cpp:120)
 *   front-end config    the Feature.* block of the generated tracker YAML
 *                       (src/Trackers/OpenVSLAMTrackerBase.cpp:193-198) and Camera.* (:161-190)
 *   keypoint readback   get_frame_state().curr_keypts (src/Trackers/OpenVSLAMStereoTracker.cpp:302-317)
 *   stereo matching     depth_threshold / focal_x_baseline parameters (src/Trackers/OpenVSLAMTrackerBase.cpp:188-201,
 *                       src/Interface/LpSlamTypes.h:219-222)
 *   bundle adjustment   the mapping / global-optimisation threads started by startup()
 *                       (src/Trackers/OpenVSLAMTrackerBase.cpp:239,250-255)
 * The arithmetic itself lives in absent third-party code (OpenVSLAM fork, g2o@691dc51, OpenCV); see DESIGN.md.
 */
#ifndef LPSLAM_HIP_H
#define LPSLAM_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPSLAM_HIP_OK 0
#define LPSLAM_HIP_ERR_INVALID 1      /* bad argument / unsupported configuration */
#define LPSLAM_HIP_ERR_DEVICE 2       /* HIP runtime error (no device, allocation, launch) */
#define LPSLAM_HIP_ERR_CAPACITY 3     /* caller buffer too small / batch exceeds context capacity */

#define LPSLAM_HIP_MAX_LEVELS 16

/* Memory layout of cv::KeyPoint (what get_frame_state().curr_keypts holds). */
typedef struct lpslam_hip_keypoint {
    float x, y;        /* level-0 pixel coordinates                          */
    float size;        /* 31 * scale_factor^octave, truncated                */
    float angle;       /* degrees [0,360)                                    */
    float response;    /* FAST score                                         */
    int32_t octave;
    int32_t class_id;  /* -1                                                 */
} lpslam_hip_keypoint;

/* FAST candidate in border-relative coordinates of its pyramid level (parity/debug readback). */
typedef struct lpslam_hip_corner {
    int32_t x, y, score;
} lpslam_hip_corner;

typedef struct lpslam_hip_frontend_config {
    int32_t width, height;        /* Camera.cols / Camera.rows                               */
    int32_t max_keypoints;        /* Feature.max_num_keypoints  (slamKeypoints)              */
    float   scale_factor;         /* Feature.scale_factor       (1.2)                        */
    int32_t num_levels;           /* Feature.num_levels         (3 generated, 8 in BASELINE) */
    int32_t ini_fast_threshold;   /* Feature.ini_fast_threshold (20)                         */
    int32_t min_fast_threshold;   /* Feature.min_fast_threshold (7)                          */
    int32_t max_images;           /* images resident per batch (a stereo frame is 2 images)  */
    int32_t device;               /* HIP device ordinal                                      */
} lpslam_hip_frontend_config;

typedef struct lpslam_hip_ctx lpslam_hip_ctx;

const char* lpslam_hip_last_error(void);
int lpslam_hip_device_count(int* count);

/* ---- context ------------------------------------------------------------------------------------------- */
int lpslam_hip_create(const lpslam_hip_frontend_config* cfg, lpslam_hip_ctx** out);
void lpslam_hip_destroy(lpslam_hip_ctx* ctx);
/* hipStream_t all work of this context is enqueued on (for event timing / interop). */
void* lpslam_hip_stream(lpslam_hip_ctx* ctx);
/* Keep `cus_per_xcd` compute units of every XCD free of the front end's kernels (0 = none, the default; at most 16): a bundle
 * adjustment that runs beside the front end -- the reference's mapping thread beside its tracking thread,
 * src/Trackers/OpenVSLAMTrackerBase.cpp:238 (openvslam::system starts both) -- otherwise stands still while front-end workgroups
 * fill every compute unit's LDS.  The context's streams are re-created: call it on an idle context, outside a prefetch section. */
int lpslam_hip_set_mapping_reserve(lpslam_hip_ctx* ctx, int32_t cus_per_xcd);
/* Test hook for the reserve (no reference counterpart): for `microseconds` (<= 50 000) one workgroup holds the whole LDS of every
 * compute unit that is NOT reserved, on a stream of its own; *landed (may be NULL) = how many sit.  Extraction launched meanwhile
 * finds room on the reserved compute units only -- the placement under which it must still complete (tests/test_frontend_gpu.py). */
int lpslam_hip_debug_occupy_unreserved(lpslam_hip_ctx* ctx, int32_t microseconds, int32_t* landed);
/* Process-wide: create the streams of contexts made FROM NOW ON at the default priority (1) or in the three classes (0, the default: main /
 * low-priority prefetch / high-priority solves; -1: back to the environment, LPSLAM_HIP_FLAT_PRIORITIES).  A process that hosts many
 * sessions per GPU (one LpSlamManager per sequence, src/Manager/SlamManager.cpp:54-61) should set 1 FIRST, before anything in it has
 * created a stream: every priority class takes its own set of hardware queues, beyond a few the command processor time-slices them, and
 * one high-priority stream ever created in the process halves the aggregate of the sessions that follow.  In a flat
 * process context k puts k mod 4 placeholder streams in front of its main stream, so that the contexts' main streams spread over the
 * process' four hardware queues (LPSLAM_HIP_NO_QUEUE_SPREAD=1 switches that off: measurements).  Called with 1 AFTER a priority stream has
 * been created in the process it changes nothing, says so on stderr and returns LPSLAM_HIP_ERR_INVALID (the caller must not believe it
 * runs flat); 2 = flat from now on although it is late (tests and measurements of that case). */
int lpslam_hip_set_flat_priorities(int32_t flat);
/* Process-wide: launches shared by the sessions of a process.  The reference runs one manager per sequence, one frame in flight each
 * (src/Manager/SlamManager.cpp:54-61,191-201); N of them on one GPU issue N chains of small latency-bound launches.  With sharing the
 * window matchers' first scan and the pose optimiser of every session that is tracking at that moment go out as ONE launch each
 * (blockIdx = request; per-request result blocks and completion flags as in the unshared call, same device code, same bits).
 * mode 2 (default): when two or more contexts of the device have made such calls within the last few milliseconds; 1: always, a lone
 * session too (tests); 0: never; -1: back to the environment (LPSLAM_HIP_SHARED_LAUNCHES).  LPSLAM_HIP_SHARE_WINDOW_US (30) /
 * LPSLAM_HIP_SHARE_QUIET_US (6) bound how long a request waits for the other sessions'. */
int lpslam_hip_set_shared_launches(int32_t mode);
/* How many shared launches a device has seen and how many requests they carried (measurement / test hook). */
int lpslam_hip_shared_launch_counters(int32_t device, int64_t* batches, int64_t* requests);
/* A context for one SESSION of a process that may host several (what a tracker plugin creates: one per LpSlamManager,
 * src/Manager/SlamManager.cpp:54-61).  Same calls, same results as a context of lpslam_hip_create; its per-image arrays are slices of a
 * pool that the sessions of one device and front-end configuration share (LPSLAM_HIP_POOL_SESSIONS, default 16; max_images <= 8), so
 * that lpslam_hip_front_end can put the frames several sessions have pending through ONE launch chain.  A session beyond the pool's
 * capacity silently gets arrays of its own. */
int lpslam_hip_create_session(const lpslam_hip_frontend_config* cfg, lpslam_hip_ctx** out);
/* One frame's front end behind its uploads, as one asynchronous call: lpslam_hip_extract_range(image, stereo ? 2 : 1), for stereo
 * lpslam_hip_match_stereo(image, image + 1, fxb, baseline), then lpslam_hip_prefetch_frame(image, stereo) -- the per-frame work of
 * feed_stereo_frame / feed_monocular_frame up to the tracking step (src/Trackers/OpenVSLAMStereoTracker.cpp:293-295).  Inside or outside
 * a prefetch section.  With shared launches on and two or more sessions of a pool submitting, their frames are extracted together. */
int lpslam_hip_front_end(lpslam_hip_ctx* ctx, int image, int32_t stereo, float focal_x_baseline, float baseline);
/* The same with the frame itself: lpslam_hip_upload_image of `left` into `image` (and of `right` into image + 1; NULL = monocular),
 * then lpslam_hip_front_end.  A shared front end uploads the frames at the head of its launch chain. */
int lpslam_hip_front_end_images(lpslam_hip_ctx* ctx, int image, const uint8_t* left, const uint8_t* right, int32_t stride,
                                float focal_x_baseline, float baseline);
/* A hint for shared launches: this session will make no more latency-bound requests (window matchers, pose optimisations) for the frame
 * it collected with lpslam_hip_get_frame / _view -- the other sessions' requests stop waiting for it.  Optional; a tracker calls it
 * when the frame's pose is final, before its keyframe work. */
int lpslam_hip_frame_done(lpslam_hip_ctx* ctx);
int lpslam_hip_shared_front_end_counters(int32_t device, int64_t* batches, int64_t* requests);
int lpslam_hip_shared_solve_counters(int32_t device, int64_t* batches, int64_t* requests);
int lpslam_hip_sync(lpslam_hip_ctx* ctx);
/* Geometry derived from the configuration (pyramid sizes, per-level keypoint quota). */
int lpslam_hip_level_info(lpslam_hip_ctx* ctx, int32_t* widths, int32_t* heights, int32_t* pitches,
                          int32_t* quotas, float* scale_factors);
/* Upper bound of keypoints one image can yield (sum of quota+3 over levels). */
int lpslam_hip_max_keypoints_per_image(lpslam_hip_ctx* ctx);

/* Event timers on the context stream (HIP events; used by bench.py to time individual kernels in place). */
#define LPSLAM_HIP_MAX_TIMERS 64
int lpslam_hip_timer_begin(lpslam_hip_ctx* ctx, int slot);
int lpslam_hip_timer_end(lpslam_hip_ctx* ctx, int slot);
/* Waits for the end event of `slot` and returns the elapsed milliseconds. */
int lpslam_hip_timer_read(lpslam_hip_ctx* ctx, int slot, float* ms);

/* ---- frames: images resident in HBM --------------------------------------------------------------------- */
/* Device address and row pitch of level 0 of image slot `image` (write frames there directly, e.g. from a
 * capture DMA or another kernel), or upload from host memory (tightly packed rows of `stride` bytes). */
int lpslam_hip_image_ptr(lpslam_hip_ctx* ctx, int image, void** dev_ptr, int32_t* pitch);
int lpslam_hip_upload_image(lpslam_hip_ctx* ctx, int image, const uint8_t* host, int32_t stride);
/* Page-locked frames and asynchronous uploads.  The reference aliases the caller's 8-bit frame (zero copy, the caller keeps it
 * alive until it is consumed: src/Manager/SlamManager.cpp:1082-1085); the counterpart on a discrete GPU is a frame in page-locked
 * host memory -- lpslam_hip_host_alloc (what a capture layer fills) or the caller's own buffer pinned in place with
 * lpslam_hip_host_register -- copied by the DMA engines on a copy stream of the context while kernels of earlier frames run.
 * lpslam_hip_upload_images_async enqueues the copies of n frames (hosts[i] -> slot first + i; every frame `stride` bytes per row)
 * and returns at once.  Ordering: the copies start after the work enqueued on the context BEFORE this call (it may still read the
 * slots); lpslam_hip_extract* / stage_pyramid of a slot wait for its copy on the device.  The frames must stay valid and unchanged
 * until such an extraction has been synchronised (lpslam_hip_sync, a get_* call).  Frames in pageable memory work but are staged by
 * the runtime inside the call. */
int lpslam_hip_host_alloc(lpslam_hip_ctx* ctx, size_t bytes, void** out);
int lpslam_hip_host_free(lpslam_hip_ctx* ctx, void* p);
int lpslam_hip_host_register(lpslam_hip_ctx* ctx, void* p, size_t bytes);
int lpslam_hip_host_unregister(lpslam_hip_ctx* ctx, void* p);
int lpslam_hip_upload_images_async(lpslam_hip_ctx* ctx, int first, int n, const uint8_t* const* hosts, int32_t stride);
/* On-device undistort / rectify: replaces the per-frame cv::remap(INTER_LINEAR) of ImageProcessing::Undistort::undistort
 * (reference: src/Utils/ImageProcessing.h:245-249; called for both eyes per frame, src/Trackers/OpenVSLAMStereoTracker.cpp:
 * 198-213).  map_x / map_y are the CV_32FC1 maps of one eye (0 = left, 1 = right), width x height floats, as
 * cv::initUndistortRectifyMap / cv::fisheye::initUndistortRectifyMap produce them (the host library builds them from the
 * LpSlamCameraConfiguration pair, lpslam_amd/host/rectify.h).  upload_raw_image copies the distorted frame to HBM and remaps it
 * into level 0 of the image slot, bit-identical to cv::remap with BORDER_CONSTANT 0. */
int lpslam_hip_set_rectify_map(lpslam_hip_ctx* ctx, int32_t eye, const float* map_x, const float* map_y);
int lpslam_hip_upload_raw_image(lpslam_hip_ctx* ctx, int image, int32_t eye, const uint8_t* host, int32_t stride);
/* remaps the raw frame already staged in HBM (the last upload_raw_image) into another slot: the device-side step alone */
int lpslam_hip_remap_staged(lpslam_hip_ctx* ctx, int image, int32_t eye);

/* Prefetch: between prefetch_begin and prefetch_end every upload / remap / extract / stereo-match call of this context is enqueued
 * on a second stream, so the front end of the NEXT frame (into image slots nothing else touches) runs on the GPU beside the
 * matching and pose optimisation of the current one; prefetch_join makes the context's main stream wait (on the device, not the
 * host) for the last prefetch section before the first call that reads those slots.  The reference overlaps the same way with
 * its frame queue and worker thread (src/Manager/SlamManager.cpp:54-61: the next frame is decoded while the tracker runs). */
int lpslam_hip_prefetch_begin(lpslam_hip_ctx* ctx);
int lpslam_hip_prefetch_end(lpslam_hip_ctx* ctx);
int lpslam_hip_prefetch_join(lpslam_hip_ctx* ctx);

/* Camera mask of one eye (0 = left: even image slots, 1 = right: odd image slots): a width x height byte image, 0 = masked out,
 * NULL removes it.  Replaces the `mask` argument of feed_stereo_frame / feed_monocular_frame
 * (src/Trackers/OpenVSLAMStereoTracker.cpp:293, masks built by OpenVSLAMTrackerBase::configureMasks, OpenVSLAMTrackerBase.cpp:331-380):
 * [UPSTREAM] orb_extractor skips a FAST cell with a corner in the mask and drops every corner whose own position is masked. */
int lpslam_hip_set_mask(lpslam_hip_ctx* ctx, int32_t eye, const uint8_t* mask, int32_t stride);

/* ---- ORB front end (asynchronous on the context stream) -------------------------------------------------- */
/* pyramid -> FAST (64-px cells, ini/min threshold) -> quad-tree distribution -> orientation + rBRIEF,
 * for image slots [0, n_images). */
int lpslam_hip_extract(lpslam_hip_ctx* ctx, int n_images);
/* Same for image slots [first, first + n_images): lets a tracker keep the previous frame resident (ping-pong). */
int lpslam_hip_extract_range(lpslam_hip_ctx* ctx, int first, int n_images);
/* Individual stages (lpslam_hip_extract = these four in order); exposed for profiling and parity tests. */
int lpslam_hip_stage_pyramid(lpslam_hip_ctx* ctx, int n_images);
int lpslam_hip_stage_fast(lpslam_hip_ctx* ctx, int n_images);
int lpslam_hip_stage_distribute(lpslam_hip_ctx* ctx, int n_images);
int lpslam_hip_stage_describe(lpslam_hip_ctx* ctx, int n_images);

/* Synchronising readbacks (device -> caller memory). */
int lpslam_hip_keypoint_count(lpslam_hip_ctx* ctx, int image, int32_t* count);
int lpslam_hip_get_keypoints(lpslam_hip_ctx* ctx, int image, lpslam_hip_keypoint* kpts, uint8_t* desc32,
                             int32_t capacity, int32_t* count);
/* Everything the tracker reads back per frame in one round trip: keypoints, descriptors and (after lpslam_hip_match_stereo on
 * this slot as the left image) the stereo columns of lpslam_hip_get_stereo.  Any output pointer may be NULL.  What
 * openvslam::data::frame holds after extraction: keypoints, descriptors, stereo_x_right, depths ([UPSTREAM] data/frame.h;
 * reached through feed_stereo_frame, src/Trackers/OpenVSLAMStereoTracker.cpp:293-295).  Not re-entrant per context. */
int lpslam_hip_get_frame(lpslam_hip_ctx* ctx, int image, lpslam_hip_keypoint* kpts, uint8_t* desc32, float* stereo_x_right,
                         float* depths, int32_t capacity, int32_t* count);
/* Queues the read-back of slot `image` behind whatever the calling thread has enqueued for it (inside a prefetch section: behind the
 * next frame's extraction and stereo match): the results are in page-locked memory before lpslam_hip_get_frame(image, ...) asks for
 * them, which then only checks a flag.  Anything that rewrites the slot's results afterwards voids the copy (get_frame then reads
 * back as usual). */
int lpslam_hip_prefetch_frame(lpslam_hip_ctx* ctx, int image, int32_t with_stereo);
/* lpslam_hip_get_frame without the copy out: pointers into the context's page-locked block that holds the slot's results (first
 * *count entries each), valid until the next lpslam_hip_get_frame / _view / lpslam_hip_prefetch_frame call on this context.  The
 * caller copies what it keeps, once, into storage of the right size. */
int lpslam_hip_get_frame_view(lpslam_hip_ctx* ctx, int image, int32_t with_stereo, const lpslam_hip_keypoint** kpts, const uint8_t** desc32,
                              const float** stereo_x_right, const float** depths, int32_t* count);
int lpslam_hip_get_pyramid_level(lpslam_hip_ctx* ctx, int image, int level, uint8_t* out, int32_t out_stride);
int lpslam_hip_get_candidates(lpslam_hip_ctx* ctx, int image, int level, lpslam_hip_corner* out,
                              int32_t capacity, int32_t* count);
/* Device-resident results (valid until the next extract on this context): keypoints [max_per_image],
 * descriptors [max_per_image][32], count (int32). */
int lpslam_hip_keypoint_buffers(lpslam_hip_ctx* ctx, int image, void** kpts_dev, void** desc_dev, void** count_dev);

/* ---- matching -------------------------------------------------------------------------------------------- */
/* Brute-force Hamming 2-NN of the descriptors of image slot `query` against those of slot `train`
 * (first minimum wins).  Results stay on the device until fetched. */
int lpslam_hip_match_bf(lpslam_hip_ctx* ctx, int query, int train);
int lpslam_hip_get_bf_knn2(lpslam_hip_ctx* ctx, int query, int32_t* best_idx, int32_t* best_dist,
                           int32_t* second_dist, int32_t capacity, int32_t* count);
/* knn2 + max distance + Lowe ratio (<= 0 disables) + optional cross check -> (query, train, distance). */
int lpslam_hip_get_bf_matches(lpslam_hip_ctx* ctx, int query, int train, int32_t max_dist, float ratio,
                              int32_t cross_check, int32_t* out_q, int32_t* out_t, int32_t* out_d,
                              int32_t capacity, int32_t* count);
/* The three calls a keyframe comparison makes -- lpslam_hip_set_descriptors(scratch, train_desc32, n_train), lpslam_hip_match_bf(query,
 * scratch), lpslam_hip_get_bf_matches(query, scratch, ...) -- as ONE call with ONE wait: the descriptors go up, both directions are
 * matched, the result arrays are delivered by a kernel into page-locked memory.  Same matches as the three calls; the scratch slot's
 * descriptors and count are overwritten the same way. */
int lpslam_hip_match_bf_descriptors(lpslam_hip_ctx* ctx, int query, int scratch, const uint8_t* train_desc32, int32_t n_train,
                                    int32_t max_dist, float ratio, int32_t cross_check, int32_t* out_q, int32_t* out_t,
                                    int32_t* out_d, int32_t capacity, int32_t* count);
/* Descriptor sets kept on the device under a caller's key -- a tracker's keyframes -- and the comparison of an image slot with MANY
 * of them in one call: one launch matches the slot against every listed set, both directions, one kernel delivers all result arrays,
 * the caller waits once.  out_q / out_t / out_d hold capacity_per_key entries per key (key k's matches start at k * capacity_per_key),
 * counts[k] their number; per key the matches are those of lpslam_hip_match_bf_descriptors against that set.  A loop-candidate search
 * over 48 old keyframes is 48 uploads and 48 waits otherwise.  put replaces an existing key; unknown keys in a match are an error. */
int lpslam_hip_desc_store_put(lpslam_hip_ctx* ctx, int32_t key, const uint8_t* desc32, int32_t n);
int lpslam_hip_desc_store_drop(lpslam_hip_ctx* ctx, int32_t key);
int lpslam_hip_match_bf_stored(lpslam_hip_ctx* ctx, int query, const int32_t* keys, int32_t n_keys, int32_t max_dist, float ratio,
                               int32_t cross_check, int32_t* out_q, int32_t* out_t, int32_t* out_d, int32_t capacity_per_key,
                               int32_t* counts);
/* Batched form: pairs (query0 + i*stride, train0 + i*stride), i in [0, n_pairs), in one launch. */
int lpslam_hip_match_bf_strided(lpslam_hip_ctx* ctx, int query0, int train0, int stride, int n_pairs);
/* Loads a caller-provided descriptor set (host memory, n x 32 bytes) into image slot `image`, replacing the
 * slot's extraction result: matching of descriptors that were produced elsewhere (map points, other frames). */
int lpslam_hip_set_descriptors(lpslam_hip_ctx* ctx, int image, const uint8_t* desc32, int32_t n);

/* Stereo: left slot vs right slot; row-band candidates, Hamming < 75, 11x11 SAD sub-pixel refinement,
 * 2 x median correlation cut.  focal_x_baseline as in LpSlamCameraConfiguration. */
int lpslam_hip_match_stereo(lpslam_hip_ctx* ctx, int left, int right, float focal_x_baseline, float true_baseline);
int lpslam_hip_match_stereo_strided(lpslam_hip_ctx* ctx, int left0, int right0, int stride, int n_pairs,
                                    float focal_x_baseline, float true_baseline);
int lpslam_hip_get_stereo(lpslam_hip_ctx* ctx, int left, float* stereo_x_right, float* depths,
                          int32_t* best_right_idx, int32_t capacity, int32_t* count);

/* ---- bundle adjustment (FP64) ------------------------------------------------------------------------------ */
typedef struct lpslam_hip_ba_obs {
    int32_t pose, point;
    double u, v, ur;          /* ur < 0: monocular edge (2 rows); else stereo (3 rows) */
    double inv_sigma2;        /* information = inv_sigma2 * I (1 / 1.2^(2*octave))      */
} lpslam_hip_ba_obs;

typedef struct lpslam_hip_ba_camera {
    double fx, fy, cx, cy, focal_x_baseline;
    double huber_mono, huber_stereo;   /* sqrt(5.991) / sqrt(7.815); <= 0: no robust kernel */
} lpslam_hip_ba_camera;

typedef struct lpslam_hip_ba_iter_log {
    double chi2_before, chi2_after, lambda;
    int32_t trials, status;            /* status 0 = OK, 1 = terminate */
} lpslam_hip_ba_iter_log;

typedef struct lpslam_hip_ba lpslam_hip_ba;

/* Builds the device-side problem (structure phase = g2o buildStructure): poses n x 7 (qw qx qy qz tx ty tz,
 * world->camera), fixed flags, points n x 3, observations.  One copy takes the inputs to HBM; the index structures (storage
 * order, CSR by landmark, pair lists of the Schur complement) are built there by kernels.  Asynchronous: returns once the work
 * is enqueued on the problem's stream. */
int lpslam_hip_ba_create(lpslam_hip_ctx* ctx, const double* poses, const uint8_t* fixed, int32_t n_poses,
                         const double* points, int32_t n_points, const lpslam_hip_ba_obs* obs, int32_t n_obs,
                         const lpslam_hip_ba_camera* cam, lpslam_hip_ba** out);
/* The same in two halves, for a host that serves many sessions.  lpslam_hip_ba_prepare is the HOST half -- validation, the window's
 * shape, the block carved out of the context's cache, the inputs copied into page-locked staging -- and enqueues nothing: every
 * session's mapping thread prepares its window on its own ([UPSTREAM] mapping_module::run of each session; reached through
 * feed_stereo_frame, src/Trackers/OpenVSLAMStereoTracker.cpp:293-295), thread safe per context.  lpslam_hip_ba_build_batch is the DEVICE
 * half for any number of prepared problems of one device: the structure kernels once for all of them (blockIdx.y = problem, ~16
 * launches whatever n is) on the first problem's stream; the others' streams wait for it on the device.  A prepared problem accepts
 * no other call before it has been built (LPSLAM_HIP_ERR_INVALID).  lpslam_hip_ba_create = prepare + build_batch of one. */
int lpslam_hip_ba_prepare(lpslam_hip_ctx* ctx, const double* poses, const uint8_t* fixed, int32_t n_poses,
                          const double* points, int32_t n_points, const lpslam_hip_ba_obs* obs, int32_t n_obs,
                          const lpslam_hip_ba_camera* cam, lpslam_hip_ba** out);
int lpslam_hip_ba_build_batch(lpslam_hip_ba* const* problems, int32_t n);
void lpslam_hip_ba_destroy(lpslam_hip_ba* ba);
/* Observation activity mask (0 = level 1 / ignored); NULL = all active. */
int lpslam_hip_ba_set_active(lpslam_hip_ba* ba, const uint8_t* active);
/* Levenberg-Marquardt with landmark Schur complement, `iters` outer iterations (g2o semantics: lambda_0 =
 * 1e-5 max diag H, rho-controlled lambda, <= 10 trials).  log may be NULL.  Returns iterations run in *done. */
int lpslam_hip_ba_optimize(lpslam_hip_ba* ba, int32_t robust, int32_t iters, lpslam_hip_ba_iter_log* log,
                           int32_t* done);
/* The same call in two halves.  begin enqueues the work on the problem's own stream and returns at once; end waits
 * for it, runs the extra trials rejected steps need and fetches the log.  Between the two the caller may enqueue
 * front-end work on the context: that is the reference's mapping thread running local BA beside tracking
 * ([UPSTREAM] mapping_module::run; lpslam only sees it through feed_stereo_frame,
 * src/Trackers/OpenVSLAMStereoTracker.cpp:293-295).  No other call on `ba` is allowed in between. */
int lpslam_hip_ba_optimize_begin(lpslam_hip_ba* ba, int32_t robust, int32_t iters);
int lpslam_hip_ba_optimize_end(lpslam_hip_ba* ba, lpslam_hip_ba_iter_log* log, int32_t* done);
/* How many solves of this context were started by replaying a captured launch graph (optimize_begin captures the launch chain
 * of a (stream, launch extents, robust, iters) signature the second time it sees it and replays it from then on, also for OTHER
 * problems with that signature on that stream).  Measurement / test hook; no reference counterpart. */
int64_t lpslam_hip_ba_graph_replays(lpslam_hip_ctx* ctx);
/* Totals over every bundle-adjustment problem this context has solved (the tracker's mapping thread makes a problem per keyframe and
 * destroys it: what happened to them is only visible here).  SIGNATURES: entries of the (stream, signature) cache, bounded at 256 and
 * never evicted -- a signature beyond the bound runs on direct launches; GRAPHS: graphs instantiated; REPLAYS: as
 * lpslam_hip_ba_graph_replays; TIMEOUTS_*: hand-overs between workgroups that timed out (lpslam_hip_ba_timeouts, summed -- any value
 * but 0 means a solve of this context failed).  Writes min(n, LPSLAM_HIP_BA_COUNTERS) values.  Measurement / test hook. */
enum { LPSLAM_HIP_BA_COUNTER_SIGNATURES = 0, LPSLAM_HIP_BA_COUNTER_GRAPHS = 1, LPSLAM_HIP_BA_COUNTER_REPLAYS = 2,
       LPSLAM_HIP_BA_COUNTER_TIMEOUTS_BAND = 3, LPSLAM_HIP_BA_COUNTER_TIMEOUTS_UPDATE = 4, LPSLAM_HIP_BA_COUNTERS = 5 };
int lpslam_hip_ba_counters(lpslam_hip_ctx* ctx, int64_t* out, int32_t n);
/* How many launches of the one-workgroup factorisation (k_chol_wg: batches of LPSLAM_HIP_CW_MIN_BATCH = 40 problems and more) this
 * context has enqueued.  Test hook: proves which of the two factorisations a batch went through.  No reference counterpart. */
int64_t lpslam_hip_ba_wg_factorisations(lpslam_hip_ctx* ctx);
/* Which linear solver a problem takes.  DENSE: Schur complement by pair lists, dense panel-pair Cholesky (any window).  BAND: when
 * no landmark of the window spans more than 10 free keyframes the reduced system is block-banded -- the shape a tracker's windows have,
 * and why the reference's local bundle adjuster solves with g2o's LinearSolverCSparse (SURVEY.md a21;
 * conan-packages/g2o-conan/conanfile.py:117-124) -- and the problem takes the landmark-group Schur complement on the FP64 matrix
 * cores and a band Cholesky in one workgroup (chosen at creation; LPSLAM_HIP_BA_SOLVER=dense in the environment keeps every problem
 * dense).  get_solver reports the solver in use and the block half-bandwidth found at creation (-1: not banded); set_solver
 * switches (BAND on a window that is not banded: LPSLAM_HIP_ERR_INVALID).  Both give the same optimum; their sums differ in order
 * (chi2 agrees to ~1e-12 relative).  The partitioned solve always works on the dense buffer. */
#define LPSLAM_HIP_BA_SOLVER_AUTO 0
#define LPSLAM_HIP_BA_SOLVER_DENSE 1
#define LPSLAM_HIP_BA_SOLVER_BAND 2
int lpslam_hip_ba_get_solver(lpslam_hip_ba* ba, int32_t* solver, int32_t* block_half_bandwidth);
/* Hand-overs between workgroups that timed out on this problem so far: `band` -- the two chains of the twisted band factorisation
 * (k_chol_band), `update` -- the keyframe blocks of the one-launch update waiting for the trial landmarks (k_ba_update).  Both waits
 * are bounded so that a grid always drains; a time-out makes the optimize call that meets it fail with LPSLAM_HIP_ERR_DEVICE (its
 * result is not valid) and is counted here.  Expected: 0, always (asserted by the batch tests).  No reference counterpart. */
int lpslam_hip_ba_timeouts(lpslam_hip_ba* ba, int32_t* band, int32_t* update);
int lpslam_hip_ba_set_solver(lpslam_hip_ba* ba, int32_t solver);
/* Batched solve -- north star: "a batched Levenberg-Marquardt local-BA".  n independent problems (the keyframe windows of
 * several SLAM sessions served by one GPU, [UPSTREAM] mapping_module::run of each session; lpslam reaches it through
 * feed_stereo_frame, src/Trackers/OpenVSLAMStereoTracker.cpp:293-295) are advanced by ONE launch chain: every kernel runs once
 * for the whole batch (blockIdx.y = problem) and each problem follows its own lambda control.  Results are identical to n calls of
 * lpslam_hip_ba_optimize for batches of fewer than 40 problems; from 40 on the reduced systems of local windows are factored by one
 * workgroup each (another summation order: equal within rounding, chi2 to 1e-12 relative).  All problems must live on one device; logs (may be NULL) receives log_stride entries per problem,
 * done (may be NULL) the iterations each problem ran.  reset_batch = lpslam_hip_ba_reset of every problem in one launch. */
int lpslam_hip_ba_optimize_batch(lpslam_hip_ba* const* problems, int32_t n, int32_t robust, int32_t iters,
                                 lpslam_hip_ba_iter_log* logs, int32_t log_stride, int32_t* done);
int lpslam_hip_ba_reset_batch(lpslam_hip_ba* const* problems, int32_t n);
/* lpslam_hip_ba_set_state / lpslam_hip_ba_get for every problem of a batch in one call (poses[i] / points[i] may be NULL as in the
 * single calls): the per-problem kernels are all enqueued before the first wait, and a server that runs its sessions' windows from a
 * scripting language crosses the boundary once per round instead of once per window. */
int lpslam_hip_ba_set_state_batch(lpslam_hip_ba* const* problems, int32_t n, const double* const* poses, const double* const* points);
int lpslam_hip_ba_get_batch(lpslam_hip_ba* const* problems, int32_t n, double* const* poses, double* const* points);
/* lpslam_hip_ba_optimize with a HIP event after every launch of the chain, on the problem's stream: time per kernel, summed over
 * the call (measurement hook of bench.py, like lpslam_hip_timer_*; no reference counterpart). */
#define LPSLAM_HIP_BA_KERNELS 8
#define LPSLAM_HIP_BA_K_LIN 0         /* k_ba_lin       (first unit of a call only; later linearisations ride in k_ba_trial) */
#define LPSLAM_HIP_BA_K_POINT_SUM 1   /* k_ba_point_sum */
#define LPSLAM_HIP_BA_K_SCHUR 2       /* k_ba_schur     */
#define LPSLAM_HIP_BA_K_CHOL 3        /* the factorisation: k_chol_pair x ceil(panels / 2), or k_chol_wg (one launch) */
#define LPSLAM_HIP_BA_K_XSOLVE 4      /* k_chol_xsolve  */
#define LPSLAM_HIP_BA_K_BACKSUB 5     /* k_ba_backsub   */
#define LPSLAM_HIP_BA_K_TRIAL 6       /* k_ba_trial     */
#define LPSLAM_HIP_BA_K_BAND_REDUCE 7 /* k_schur_band_reduce (band path only; there K_SCHUR is k_schur_group and K_CHOL is k_chol_band, one launch) */
typedef struct lpslam_hip_ba_kernel_times {
    float ms[LPSLAM_HIP_BA_KERNELS];                  /* summed over the call                                    */
    int32_t launches[LPSLAM_HIP_BA_KERNELS];          /* marks (= LM trials the kernel ran in)                   */
    int32_t launches_per_mark[LPSLAM_HIP_BA_KERNELS]; /* kernel launches behind one mark (factorisation: several) */
    int32_t iterations, dim;
} lpslam_hip_ba_kernel_times;
int lpslam_hip_ba_optimize_profiled(lpslam_hip_ba* ba, int32_t robust, int32_t iters, lpslam_hip_ba_kernel_times* out);
/* Motion-only mode: landmarks are held fixed (unary edges), only the free poses move. */
int lpslam_hip_ba_set_points_fixed(lpslam_hip_ba* ba, int32_t points_fixed);
/* optimize::pose_optimizer flow on a problem created with ONE free pose: 4 rounds x 10 iterations, outliers
 * (chi2 > 5.991 mono / 7.815 stereo) re-classified after every round, Huber dropped after the third round.
 * outlier may be NULL; *n_inliers receives the number of inlier observations. */
int lpslam_hip_ba_pose_optimize(lpslam_hip_ba* ba, uint8_t* outlier, int32_t* n_inliers);
/* The same flow for the tracker's per-frame call, without a problem object: one launch, one workgroup, the whole 4 x 10
 * iteration flow on the device (pose7 in / out; obs[k].point indexes `points`, obs[k].pose is ignored). */
int lpslam_hip_pose_optimize(lpslam_hip_ctx* ctx, double* pose7, const double* points, int32_t n_points, const lpslam_hip_ba_obs* obs,
                             int32_t n_obs, const lpslam_hip_ba_camera* cam, uint8_t* outlier, int32_t* n_inliers);
/* Diagnostic: passes over the observations the last lpslam_hip_pose_optimize on this context made (one per round + one per
 * Levenberg trial; the call's duration is this number times the latency of a trial). */
int32_t lpslam_hip_pose_optimize_passes(lpslam_hip_ctx* ctx);
/* local_bundle_adjuster flow: first_iters robust, outlier classification, second_iters plain. */
int lpslam_hip_ba_local(lpslam_hip_ba* ba, int32_t first_iters, int32_t second_iters, uint8_t* outlier);
/* A keyframe's local bundle adjustment in ONE call ([UPSTREAM] mapping_module -> optimize::local_bundle_adjuster::optimize, started per
 * keyframe by the mapping thread of startup(), src/Trackers/OpenVSLAMTrackerBase.cpp:239): lpslam_hip_ba_create of the window,
 * lpslam_hip_ba_local, lpslam_hip_ba_get into `poses` / `points` (in: the window's state, out: the solved one), destroy.  When several
 * sessions' mapping threads call it at the same time (shared launches on), their windows are built and solved together. */
int lpslam_hip_ba_local_window(lpslam_hip_ctx* ctx, double* poses, const uint8_t* fixed, int32_t n_poses, double* points, int32_t n_points,
                               const lpslam_hip_ba_obs* obs, int32_t n_obs, const lpslam_hip_ba_camera* cam, int32_t first_iters,
                               int32_t second_iters, uint8_t* outlier);
/* Restores the poses / points / activity mask given at creation (kept in HBM) and clears the LM state. */
int lpslam_hip_ba_reset(lpslam_hip_ba* ba);
/* Replaces the creation-time poses (n_poses x 7) and / or landmarks (n_points x 3) of an existing problem (NULL keeps them) and
 * resets it: the observation graph and its device-built structure stay.  A mapping thread creates the next window while the
 * previous one is being solved (lpslam_hip_ba_create only enqueues) and hands it the state that solve produced
 * ([UPSTREAM] local_bundle_adjuster reads the map after the previous run has written it back). */
int lpslam_hip_ba_set_state(lpslam_hip_ba* ba, const double* poses, const double* points);
int lpslam_hip_ba_get(lpslam_hip_ba* ba, double* poses, double* points);
int lpslam_hip_ba_chi2(lpslam_hip_ba* ba, double* chi2, uint8_t* depth_positive);

/* Partitioned (multi-GPU) global BA: every rank holds a landmark partition (its observations) and all poses.  One LM
 * trial is three device phases with the caller's collectives in between (each phase ends synchronised):
 *   step_begin(robust, first=1)   linearise                -> all-reduce SUM reduced_buffer, MAX scalar_buffer[4]
 *   step_lambda0()                lambda_0 from the reduced diagonals (first trial of an optimisation only)
 *   step_begin(robust, first=0)   (linearise if the state changed) + partial Schur complement -> all-reduce SUM reduced_buffer
 *   step_solve()                  factor, solve, update, trial chi2  -> all-reduce SUM scalar_buffer[1..2]
 *   step_end()                    accept / reject (identical on every rank) */
int lpslam_hip_ba_reduced_buffer(lpslam_hip_ba* ba, void** dev_ptr, int64_t* n_doubles);
int lpslam_hip_ba_scalar_buffer(lpslam_hip_ba* ba, void** dev_ptr, int64_t* n_doubles);
int lpslam_hip_ba_step_begin(lpslam_hip_ba* ba, int32_t robust, int32_t first);
int lpslam_hip_ba_step_lambda0(lpslam_hip_ba* ba);
int lpslam_hip_ba_step_solve(lpslam_hip_ba* ba);
int lpslam_hip_ba_step_end(lpslam_hip_ba* ba, int32_t* accepted, int32_t* iteration_finished);
/* The same partitioned solve driven from C++ with RCCL (ncclAllReduce of the packed triangle of S + b per LM
 * trial, a 2-double all-reduce for the trial chi2): `nccl_comm` is this rank's ncclComm_t, created by the caller
 * (ncclCommInitRank / ncclCommInitAll), one rank per GPU.  All collectives are enqueued on the problem's own stream between the
 * kernels -- no host synchronisation inside a trial; the host looks at the control block once per call (once more per batch of
 * rejected trials).  RCCL is bound at run time (dlopen).  Every rank must call with the same robust / iters. */
int lpslam_hip_ba_optimize_partitioned(lpslam_hip_ba* ba, void* nccl_comm, int32_t robust, int32_t iters,
                                       lpslam_hip_ba_iter_log* log, int32_t* done);
/* The same driver over the caller's own collective (an MPI / shared-memory host, or a test that runs several ranks on one
 * device): `allreduce(user, buf, count, op, stream)` must enqueue an IN-PLACE all-reduce of `count` doubles at device address
 * `buf` on HIP stream `stream` (a hipStream_t), ordered after the work already on that stream, op = LPSLAM_HIP_REDUCE_SUM or
 * LPSLAM_HIP_REDUCE_MAX, and return 0 on success.  Every rank must produce bit-identical results (a fixed summation order):
 * the ranks take their accept / reject decisions separately from the reduced values.  lpslam_hip_ba_optimize_partitioned is
 * this call with ncclAllReduce behind the callback. */
#define LPSLAM_HIP_REDUCE_SUM 0
#define LPSLAM_HIP_REDUCE_MAX 2
typedef int (*lpslam_hip_allreduce_fn)(void* user, void* buf, size_t count, int32_t op, void* stream);
int lpslam_hip_ba_optimize_partitioned_with(lpslam_hip_ba* ba, lpslam_hip_allreduce_fn allreduce, void* user, int32_t robust,
                                            int32_t iters, lpslam_hip_ba_iter_log* log, int32_t* done);
/* Control state after the last optimize / step_end: finished outer iterations, g2o "Terminate", lambda, robust chi2. */
int lpslam_hip_ba_status(lpslam_hip_ba* ba, int32_t* outer_done, int32_t* stopped, double* lambda, double* chi2);

/* ---- bag-of-words vocabulary and match::bow_tree ------------------------------------------------------------------------------
 * [UPSTREAM] DBoW2 TemplatedVocabulary<ORB> (shinsumicco/DBoW2 @ e8cc74d: conan-packages/dbow2-conan/conanfile.py:30-31), which the
 * reference requires at start-up (src/Trackers/OpenVSLAMTrackerBase.cpp:224-227: "Vocab file ... not present" -> start fails;
 * handed to openvslam::system at :238), and [UPSTREAM] openvslam match::bow_tree, which the relocaliser and the loop detector
 * (toggled at :250-255) match keypoints with.
 * lpslam_hip_vocab_create takes the tree as DBoW2 stores it -- nodes 1 .. n_nodes in file order, each naming its parent (0 = the
 * root, which precedes every node), its 32-byte descriptor, its weight and whether it is a leaf (= a word; word ids count the
 * leaves in node order) -- and keeps it in HBM.  The file formats are read by the host mirror (lpslam_amd/host/bow.h).
 * lpslam_hip_bow_transform walks the descriptors of an image slot down the tree on the device (TemplatedVocabulary::transform with
 * a FeatureVector: at every level the child at the smallest Hamming distance, the first on ties): per keypoint the word id, the
 * word's weight and the id of the node `levels_up` levels above the leaves (0 = the root when the tree is not that deep).  The
 * result buffers must hold lpslam_hip_max_keypoints_per_image() entries.  _host: the same for descriptors in host memory. */
typedef struct lpslam_hip_vocab lpslam_hip_vocab;
int lpslam_hip_vocab_create(lpslam_hip_ctx* ctx, int32_t k, int32_t L, int32_t n_nodes, const int32_t* parent, const uint8_t* desc32,
                            const float* weight, const uint8_t* is_leaf, lpslam_hip_vocab** out);
void lpslam_hip_vocab_destroy(lpslam_hip_vocab* vocab);
int lpslam_hip_vocab_info(lpslam_hip_vocab* vocab, int32_t* k, int32_t* L, int32_t* n_nodes, int32_t* n_words);
int lpslam_hip_bow_transform(lpslam_hip_ctx* ctx, lpslam_hip_vocab* vocab, int image, int32_t levels_up, int32_t* word_id, float* word_weight,
                             int32_t* node_id, int32_t capacity, int32_t* count);
int lpslam_hip_bow_transform_host(lpslam_hip_ctx* ctx, lpslam_hip_vocab* vocab, const uint8_t* desc32, int32_t n, int32_t levels_up,
                                  int32_t* word_id, float* word_weight, int32_t* node_id);
/* match::bow_tree (match_frame_and_keyframe / match_keyframes): query k (descriptor + the node it falls under, q_node[k] < 0 =
 * not a query, e.g. a keypoint without a landmark) is compared with the targets under the same node.  Queries are served node by
 * node (ascending id) and in keypoint order inside a node; a target matched by an earlier query (or flagged in t_taken, may be
 * NULL) is invisible to later ones; the nearest free target wins (the first on ties), accepted when best <= hamming_thr and
 * best <= lowe_ratio * second (second = 256 when there is no other).  match_idx[k] = target index or -1.  The orientation check
 * of upstream is lpslam_hip_match_orientation_filter on the result. */
int lpslam_hip_match_bow_tree(lpslam_hip_ctx* ctx, const uint8_t* q_desc32, const int32_t* q_node, int32_t nq, const uint8_t* t_desc32,
                              const int32_t* t_node, int32_t nt, const uint8_t* t_taken, int32_t hamming_thr, float lowe_ratio,
                              int32_t* match_idx, int32_t* match_dist, int32_t* n_matches);
/* The same query set against n_sets target sets (a loop-candidate search: the new keyframe against up to eight keyframes,
 * [UPSTREAM] loop_detector -> match::bow_tree::match_keyframes per candidate) in ONE round trip to the device; set i's results in
 * match_idx[i][0 .. nq), match_dist[i] (match_dist or its entries may be NULL), n_matches[i]; t_taken (or its entries) may be NULL.
 * Results are those of n_sets calls of lpslam_hip_match_bow_tree. */
int lpslam_hip_match_bow_tree_multi(lpslam_hip_ctx* ctx, const uint8_t* q_desc32, const int32_t* q_node, int32_t nq, int32_t n_sets,
                                    const uint8_t* const* t_desc32, const int32_t* const* t_node, const int32_t* nt, const uint8_t* const* t_taken,
                                    int32_t hamming_thr, float lowe_ratio, int32_t* const* match_idx, int32_t* const* match_dist, int32_t* n_matches);

/* ---- projection matching ---------------------------------------------------------------------------------------------
 * [UPSTREAM] match::projection::match_frame_and_landmarks (local-map tracking) and match_current_and_last_frames (motion-model
 * tracking), which openvslam::system runs inside feed_stereo_frame / feed_monocular_frame
 * (reference call sites: src/Trackers/OpenVSLAMStereoTracker.cpp:293-295, src/Trackers/OpenVSLAMTracker.cpp:120).
 * Query k = predicted pixel position, predicted right-image x (< 0: none), search radius (margin x scale factor of the
 * predicted level) and level range (-1: open) + its 32-byte descriptor.  Queries are matched IN ORDER against the keypoints of
 * image slot `image`: a keypoint taken by an earlier query (or flagged in `taken`, may be NULL) is invisible to later ones;
 * best distance <= hamming_thr; Lowe ratio only between candidates of the same level.  use_stereo != 0 applies the right-image
 * check against the slot's last stereo match.  match_idx[k] = keypoint index or -1. */
typedef struct lpslam_hip_proj_query {
    float x, y, x_right, radius;
    int32_t min_level, max_level;
} lpslam_hip_proj_query;
int lpslam_hip_match_projection(lpslam_hip_ctx* ctx, int image, const lpslam_hip_proj_query* queries, const uint8_t* q_desc32, int32_t nq,
                                int32_t hamming_thr, float lowe_ratio, const uint8_t* taken, int32_t use_stereo,
                                int32_t* match_idx, int32_t* match_dist, int32_t* n_matches);
/* [UPSTREAM] match::fuse (mapping module: detect / replace duplication): for every landmark projected into a keyframe the best
 * keypoint inside the window whose reprojection error passes the chi-square gate of the keypoint's level (5.99146 / 7.81473 with
 * the right-image x, use_stereo != 0), Hamming distance <= hamming_thr (50).  No exclusivity: what happens to a keypoint that
 * already carries a landmark is the caller's rule. */
int lpslam_hip_match_fuse(lpslam_hip_ctx* ctx, int image, const lpslam_hip_proj_query* queries, const uint8_t* q_desc32, int32_t nq,
                          int32_t hamming_thr, int32_t use_stereo, int32_t* match_idx, int32_t* match_dist, int32_t* n_matches);
/* [UPSTREAM] match::area::match_in_consistent_area (monocular initialiser): queries = level-0 keypoints of frame 1 at the position
 * they were matched to before (radius = margin, levels 0..0), image = frame 2.  A keypoint already matched at an equal or
 * smaller distance is not a candidate; best <= hamming_thr and best < lowe_ratio x second; a better later query takes the
 * keypoint away from the earlier one (whose match_idx becomes -1). */
int lpslam_hip_match_area(lpslam_hip_ctx* ctx, int image, const lpslam_hip_proj_query* queries, const uint8_t* q_desc32, int32_t nq,
                          int32_t hamming_thr, float lowe_ratio, int32_t* match_idx, int32_t* match_dist, int32_t* n_matches);
/* match::angle_checker: drops the matches whose angle difference (query - keypoint, degrees) lies outside the three most
 * populated 30-degree bins.  Host-side (a few thousand matches). */
int lpslam_hip_match_orientation_filter(const float* angle_q, const float* angle_t, int32_t* match_idx, int32_t nq, int32_t* n_kept);

/* ---- Sim3 pose graph ---------------------------------------------------------------------------------------------
 * Replaces what openvslam::system runs on its global-optimisation thread after a loop closure while the loop detector
 * is enabled (reference: src/Trackers/OpenVSLAMTrackerBase.cpp:250-255): [UPSTREAM] optimize::graph_optimizer on
 * g2o@691dc51 types_sim3 (VertexSim3Expmap / EdgeSim3, numeric Jacobians, Levenberg, BlockSolver_7_3); SURVEY.md 8(a) a23.
 * A Sim3 is 8 doubles qw qx qy qz tx ty tz s (world -> camera: x_c = s R x_w + t). */
typedef struct lpslam_hip_sim3_edge {
    int32_t i, j;          /* vertices: error = log(meas * S_i * S_j^-1), information = identity */
    double meas[8];
} lpslam_hip_sim3_edge;

typedef struct lpslam_hip_sim3 lpslam_hip_sim3;

/* verts n x 8, fixed flags (NULL = none fixed; the loop keyframe should be), edges; fix_scale != 0 zeroes the scale
 * component of every update (stereo / RGBD).  Copies everything to HBM. */
int lpslam_hip_sim3_create(lpslam_hip_ctx* ctx, const double* verts, const uint8_t* fixed, int32_t n,
                           const lpslam_hip_sim3_edge* edges, int32_t n_edges, int32_t fix_scale, lpslam_hip_sim3** out);
void lpslam_hip_sim3_destroy(lpslam_hip_sim3* graph);
/* `iters` outer Levenberg iterations (graph_optimizer: 50), g2o lambda control; log may be NULL. */
int lpslam_hip_sim3_optimize(lpslam_hip_sim3* graph, int32_t iters, lpslam_hip_ba_iter_log* log, int32_t* done);
int lpslam_hip_sim3_get(lpslam_hip_sim3* graph, double* verts);
/* chi2 (= |error|^2) per edge of the current estimate */
int lpslam_hip_sim3_chi2(lpslam_hip_sim3* graph, double* chi2);

/* Sim3 between two keyframes ([UPSTREAM] optimize::transform_optimizer, called by the loop detector for every loop candidate):
 * matched landmark pairs k = landmark 1 in camera-1 coordinates, landmark 2 in camera-2 coordinates, both keypoints and
 * 1 / sigma^2 of their pyramid levels.  A batch of candidates is solved in one launch (one workgroup each): problem i owns
 * pairs [pair_start[i], pair_start[i+1]) and s12[8 i .. 8 i + 7] (Sim3 camera 2 -> camera 1, in / out).  Flow per problem:
 * 5 Levenberg iterations (Huber sqrt(chi_sq)), pairs with chi2 > chi_sq on either edge dropped, 10 more iterations if any was
 * dropped (else 5); n_inliers[i] = 0 when fewer than 10 pairs survive the first cut.  cam = fx fy cx cy.  inlier may be NULL. */
typedef struct lpslam_hip_sim3_pair {
    double p1c[3], p2c[3];
    double obs1[2], obs2[2];
    double inv_sigma2_1, inv_sigma2_2;
} lpslam_hip_sim3_pair;
int lpslam_hip_sim3_transform_optimize(lpslam_hip_ctx* ctx, int32_t n_problems, double* s12, const lpslam_hip_sim3_pair* pairs,
                                       const int32_t* pair_start, const double* cam1, const double* cam2, double chi_sq,
                                       int32_t fix_scale, uint8_t* inlier, int32_t* n_inliers);

#ifdef __cplusplus
}
#endif
#endif /* LPSLAM_HIP_H */
