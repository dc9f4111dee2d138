// internal.h -- shared host-side definitions of the lpslam HIP library (gfx950 only).
#pragma once
#include <sched.h>
#include <time.h>
#include <sys/prctl.h>
#include <hip/hip_runtime.h>
#include <array>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <vector>
#include <utility>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/lpslam_hip.h"

namespace lpslam {

constexpr int kMaxLevels = LPSLAM_HIP_MAX_LEVELS;
constexpr int kEdge = 19;          // orb_patch_radius_: cells start 19 px inside the level
constexpr int kCell = 64;          // FAST cell size
constexpr int kOverlap = 6;        // FAST cell overlap (2 x 3-px FAST border)
constexpr int kCellSlots = 1024;   // max NMS survivors in a 64x64 cell (no two 8-adjacent)
constexpr int kQuotaMax = 2000;    // per-level quota supported by the distribution kernel (node index < 2048)

// Per-level geometry, passed to kernels by value.
constexpr int kPyrMaxBands = 32;   // banded pyramid kernel: up to 32 bands per image

// Which images a front-end launch works on: `n` consecutive slots from `first`, or -- the pending frames of several sessions, whose
// slots lie anywhere in the session pool -- a list (share.hip).  Passed by value; image k of the launch is lp_image(sel, k).
constexpr int kMaxListed = 64;
struct ImgSel { int first; int listed; uint16_t list[kMaxListed]; };
#ifdef __HIPCC__
__device__ __forceinline__ int lp_image(const ImgSel& s, int k) { return s.listed ? (int)s.list[k] : s.first + k; }
#endif
inline ImgSel lp_img_sel(int first, int n, const uint16_t* list)
{
    ImgSel s{};
    s.first = first; s.listed = list ? 1 : 0;
    if (list) for (int i = 0; i < n && i < kMaxListed; ++i) s.list[i] = list[i];
    return s;
}

struct LevelTable {
    int n_levels;
    int w[kMaxLevels], h[kMaxLevels], pitch[kMaxLevels];
    unsigned off[kMaxLevels];          // byte offset of the level inside one image's pyramid slab
    int cells_x[kMaxLevels], cells_y[kMaxLevels];
    int cell_start[kMaxLevels + 1];    // prefix of cells over levels (per image)
    int quota[kMaxLevels];
    int slot_start[kMaxLevels + 1];    // prefix of (quota+3) over levels: selected-keypoint slots per image
    int cand_start[kMaxLevels + 1];    // prefix of candidate capacity (cells*kCellSlots) over levels
    float scale[kMaxLevels], inv_scale[kMaxLevels];
    int xtab_start[kMaxLevels], ytab_start[kMaxLevels];  // offsets into the host-side resize tables (level >= 1)
    int dx_start[kMaxLevels], dy_start[kMaxLevels];      // offsets into the device copy: x as four planes (below), y linear
    // quad-tree roots (initialize_nodes): grid and patch size per level, node capacity Q = max(quota, 4*roots) + 4
    int nxg[kMaxLevels], nyg[kMaxLevels], qcap[kMaxLevels];
    double delta_x[kMaxLevels], delta_y[kMaxLevels];
};

void set_error(const char* fmt, ...);
// Between lpslam_hip_prefetch_begin and _end the CALLING THREAD's upload / remap / extraction / stereo launches go to the context's
// prefetch stream; other threads (the tracking thread) keep using the main stream of the same context.
extern thread_local hipStream_t lp_tls_stream;
extern thread_local bool lp_tls_stream_used;       // something was enqueued on it since lpslam_hip_prefetch_begin (lp_fe_stream hands it out)
int hip_fail(hipError_t e, const char* what);

#define LP_HIP(call)                                              \
    do {                                                          \
        hipError_t e_ = (call);                                   \
        if (e_ != hipSuccess) return ::lpslam::hip_fail(e_, #call); \
    } while (0)

}  // namespace lpslam

struct lpslam_hip_ctx {
    lpslam_hip_frontend_config cfg{};
    lpslam::LevelTable lt{};
    hipStream_t stream = nullptr;      // the stream every entry point enqueues on
    uint32_t* d_cu_table = nullptr;    // mapping reserve in software: [8 XCC][8 words] bit per compute unit the extraction kernels leave alone (frontend.hip)
    unsigned* d_done = nullptr;        // arrival counters of the kernels that deliver results to page-locked memory (lp_signal_done), 128 bytes apart
    int done_seq = 0;                  // sequence numbers those kernels release into their done flags
    int* d_fe_counters = nullptr;      // work-queue counters of the queued extraction launches (a ring of 64 per stream, 128 bytes apart)
    std::atomic<unsigned> fe_counter_next{0}, fe_counter_next_prefetch{0};
    hipStream_t debug_stream = nullptr; // lpslam_hip_debug_occupy_unreserved (test hook)
    std::vector<hipStream_t> pad_streams; void* d_pad = nullptr;      // flat priorities: streams created in front of the main one, so that the main streams of a process' contexts spread over its hardware queues (api.hip)
    int reserve_cus = 0;               // lpslam_hip_set_mapping_reserve: CUs of every XCD the front end's streams leave to the mapping solves
    hipStream_t fe_stream = nullptr;   // prefetch: front end of the NEXT frame beside the tracking of this one (lp_fe_stream)
    hipEvent_t fe_done = nullptr;
    bool fe_join_needed = false;       // the last prefetch section enqueued work on fe_stream and recorded fe_done: lpslam_hip_prefetch_join has something to wait for
    // asynchronous uploads from the caller's page-locked frames (lpslam_hip_upload_images_async): a copy stream of its own, one event
    // per call that the main stream waits for before it first reads one of the call's slots
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copy_mark = nullptr;           // main stream -> copy stream: work enqueued before the call may still read the slots
    std::vector<hipEvent_t> ev_copy_pool;        // events of the calls (ring)
    size_t ev_copy_next = 0;
    hipEvent_t ev_copy_mark_fe = nullptr;        // the same for the prefetch stream when one exists
    std::mutex copy_mutex;                       // slot_copy_event / the event ring: uploads and extractions may come from different threads
    std::vector<hipEvent_t> slot_copy_event;     // per image slot: event of the upload the main stream has not yet waited for (or nullptr)
    std::vector<std::pair<void*, size_t>> host_allocs;   // lpslam_hip_host_alloc blocks (freed with the context if the caller forgets)
    std::vector<uint8_t*> h_upload;    // per image slot: page-locked staging of the last uploaded frame (uploads are asynchronous)
    std::vector<hipEvent_t> ev_upload; // ... and the event after its copy
    size_t image_slab = 0;             // bytes of one image's pyramid (all levels, pitched)
    int slots_per_image = 0;           // sum(quota+3)
    int cells_per_image = 0;
    int cand_per_image = 0;

    uint8_t* d_pyr = nullptr;          // [max_images][image_slab]
    // resize tables: xofs/yofs (int16) and 11-bit coefficient pairs (int16 x2) per output column/row
    int2* d_rs_pack = nullptr;         // resize tables, entries (s0 | s1 << 16, w0 | w1 << 16); per level: the columns as FOUR PLANES
                                       // [k][q] = column min(4 q + k, w - 1) (a lane produces the four pixels of dword q: neighbouring lanes
                                       // read neighbouring entries of a plane -- the linear table was an 8-way LDS bank conflict), then the rows
    int rs_entries = 0;
    uint8_t* h_stage = nullptr;        // pinned host staging of lpslam_hip_get_frame (one frame's results)
    size_t h_stage_bytes = 0;
    // a frame's results delivered ahead of time (lpslam_hip_prefetch_frame): block of its own, the image it holds (-1: none), what it holds
    uint8_t* h_stage_pf = nullptr; size_t h_stage_pf_bytes = 0; std::atomic<int> pf_image{-1}; int pf_seq = 0, pf_seq_next = 0, pf_fields = 0; bool pf_in_flight = false; hipStream_t pf_stream = nullptr;
    // host mirror of d_kp_count: valid after any call that fetched it, invalidated by whatever rewrites it on the device
    std::vector<int32_t> h_kp_count; std::vector<uint8_t> h_kp_valid;
    // descriptor sets kept on the device under a caller's key (lpslam_hip_desc_store_put: a tracker's keyframes), blocks of the pool
    struct StoredDesc { void* blk = nullptr; size_t cap = 0; int n = 0; };
    std::map<int, StoredDesc> desc_store;
    std::vector<void*> pin_free;       // page-locked 8 KB blocks handed to bundle-adjustment objects (lp_pin_alloc / lp_pin_free)
    std::vector<hipStream_t> ba_streams;   // idle high-priority streams of destroyed bundle-adjustment problems (lp_stream_acquire / release)
    // Captured launch chains of the bundle adjustment, shared by every problem that runs on a stream: the graph's kernels read their
    // view from a per-stream device slot, the problem's own view is copied there (device to device) in front of the launch.  A
    // mapping thread makes a NEW problem per keyframe: with per-problem graphs each ran on direct launches (2.8 us per dependent
    // kernel against 1.7 us inside a graph, 120+ kernels per solve).
    std::map<hipStream_t, void*> ba_view_slot;
    std::map<std::pair<hipStream_t, std::array<int, 24>>, hipGraphExec_t> ba_graphs;      // nullptr = signature seen once
    std::atomic<long> ba_wg_launches{0};     // k_chol_wg launches so far (lpslam_hip_ba_wg_factorisations: a test sees which factorisation a batch took)
    std::atomic<long> ba_graphs_built{0};    // graphs instantiated so far (at most 256: the signature cache is bounded)
    std::atomic<long> ba_timeouts_band{0}, ba_timeouts_update{0};      // timed-out hand-overs of every problem of this context (report_faults)
    std::atomic<long> ba_graph_replays{0};   // hipGraphLaunch calls so far (lpslam_hip_ba_graph_replays: lets a test see that it exercised the replay path)
    std::vector<std::pair<size_t, void*>> pin_big;   // idle page-locked staging blocks (capacity, block) of lp_pin_big_alloc / free
    uint8_t* h_match = nullptr;        // pinned host staging of the window matchers (queries in, candidate lists out)
    size_t h_match_bytes = 0;
    // A session on the role streams uploads its frames on a stream of ITS OWN that carries copies and their events only: the sessions'
    // copies run side by side on the copy engines and beside the chains on the front-end stream (16 images one after the other on one
    // stream: 0.41 ms, as long as their extraction), and what such a stream can hold up on a hardware queue it shares is bounded by a copy.
    hipStream_t up_stream = nullptr; std::vector<uint8_t> up_pending;      // per image slot: uploaded on up_stream, not yet waited for by the front-end stream
    bool owns_streams = true;          // false: a session of a pool -- stream / fe_stream / the solves' stream are the device's role streams (share.hip)
    hipStream_t role_solve = nullptr;  // session: the role stream its bundle adjustments run on
    hipStream_t role_aux = nullptr;    // session: the role stream of its loop-candidate search and descriptor-store copies (the matchers' unless a fifth queue exists)
    int share_slot = -1;               // this context's entry in its device's session table (share.hip), -1: never shared
    // Session pool (lpslam_hip_create_session): the per-image arrays of the sessions of one device and front-end configuration are
    // slices of ONE set of arrays -- those of a pool context that no caller sees -- so that a front-end launch can work on the pending
    // frames of several sessions at once (image lists, ImgSel).  A session's slot i is image pool_first + i of the pool.
    lpslam_hip_ctx* sess_pool = nullptr;    // session: its pool; nullptr: a context with arrays of its own
    int pool_first = 0, pool_slot = -1;
    bool is_pool = false; int pool_per = 0; std::vector<uint8_t> pool_used; int pool_refs = 0;      // pool: images per session, which session slots are taken
    hipEvent_t ev_fe_ready = nullptr;  // session: "this frame's uploads are in the slots" for a shared front end (recorded on the session's stream)
    std::atomic<int> share_fe_pending{0};      // session: a shared front end has been launched and its delivery not yet collected
    int po_passes = 0;                 // passes the last pose optimisation made (diagnostic)
    int po_seq = 0;                    // sequence number the pose optimiser's kernel releases into its done flag (h_match + 64)
    int2* d_band_rows = nullptr;       // [band count 0..32][levels][bands]: rows of each level a band work-group computes

    // FAST output: per cell fixed slots + counts
    uint32_t* d_cell_keys = nullptr;   // [max_images][cells_per_image][kCellSlots]  score<<24 | y<<12 | x
    int32_t* d_cell_count = nullptr;   // [max_images][cells_per_image]
    // distribution scratch
    uint32_t* d_cand_key = nullptr;    // [max_images][cand_per_image]
    uint32_t* d_cand_node = nullptr;   // [max_images][cand_per_image]
    int32_t* d_cand_count = nullptr;   // [max_images][levels]
    uint32_t* d_sel_key = nullptr;     // [max_images][slots_per_image] selected corners (level slots)
    int32_t* d_sel_count = nullptr;    // [max_images][levels]
    // final keypoints
    lpslam_hip_keypoint* d_kpts = nullptr;  // [max_images][slots_per_image]
    uint8_t* d_desc = nullptr;              // [max_images][slots_per_image][32]
    int32_t* d_kp_count = nullptr;          // [max_images]
    // matching results
    int32_t* d_bf = nullptr;           // [max_images][3][slots_per_image] best idx, best dist, second dist
    float* d_stereo = nullptr;         // [max_images][2][slots_per_image] x_right, depth
    int32_t* d_stereo_idx = nullptr;   // [max_images][slots_per_image]
    int32_t* d_stereo_corr = nullptr;  // [max_images][slots_per_image]
    int32_t* d_st_row_start = nullptr; // [max_images][H + 1]: per image row, where its candidate list starts (stereo matcher)
    int32_t* d_st_row_list = nullptr;  // [max_images][st_row_cap]: right-image keypoints whose row band covers the row
    int st_row_cap = 0;
    // on-device undistort / rectify: per eye the fixed-point map (cv::convertMaps form) and one raw-frame staging buffer
    short2* d_map_xy[2] = {nullptr, nullptr};      // [h][w] integer source coordinate (sx >> 5, sy >> 5)
    uint16_t* d_map_frac[2] = {nullptr, nullptr};  // [h][w] (sy & 31) * 32 + (sx & 31)
    uint8_t* d_raw = nullptr;                      // [h][w] distorted frame of the upload in flight
    uint8_t* d_mask[2] = {nullptr, nullptr};       // camera masks (level-0 size, pitch = width, 0 = masked out), left / right eye; nullptr = none
    // cache of device blocks for the short-lived objects the tracker makes every frame / keyframe (bundle-adjustment problems, pose
    // optimiser and projection-matcher staging): hipMalloc / hipFree cost ~50-100 us each and a problem needs ~40 buffers
    std::vector<std::pair<size_t, void*>> pool;        // (capacity, block), free blocks only
    size_t pool_bytes = 0;
    size_t pool_cap = 0;                               // bytes the cache may hold (0: not yet derived from the device's memory, api.hip)
    std::mutex pool_mutex;
    // staging for *_host convenience calls
    uint8_t* d_tmp_desc = nullptr; size_t tmp_desc_bytes = 0;
    int32_t* d_tmp_res = nullptr;  size_t tmp_res_bytes = 0;
    size_t distribute_lds = 0;
    hipEvent_t ev_begin[LPSLAM_HIP_MAX_TIMERS] = {}, ev_end[LPSLAM_HIP_MAX_TIMERS] = {};
};

// kernel launchers (frontend.hip / match.hip)
// block cache (api.hip): capacity-rounded first fit; *capacity receives the size to hand back to lp_pool_free
inline hipStream_t lp_fe_stream(lpslam_hip_ctx* c) { if (lpslam::lp_tls_stream) { lpslam::lp_tls_stream_used = true; return lpslam::lp_tls_stream; } return c->stream; }
int lp_pool_alloc(lpslam_hip_ctx* c, size_t bytes, void** out, size_t* capacity);
void lp_pool_free(lpslam_hip_ctx* c, void* p, size_t capacity);
hipStream_t lp_stream_acquire(lpslam_hip_ctx* c);   // high-priority non-blocking stream from the context's cache (nullptr on failure)
void lp_stream_release(lpslam_hip_ctx* c, hipStream_t s);
void* lp_pin_big_alloc(lpslam_hip_ctx* c, size_t bytes, size_t* capacity);      // page-locked staging of any size, recycled through the context
void lp_pin_big_free(lpslam_hip_ctx* c, void* p, size_t capacity);
void* lp_pin_alloc(lpslam_hip_ctx* c);          // 8 KB of page-locked host memory, recycled through the context (nullptr on failure)
void lp_pin_free(lpslam_hip_ctx* c, void* p);
int lp_launch_pyramid(lpslam_hip_ctx* c, int first, int n_images, const uint16_t* list = nullptr);      // list: n_images slots in any order instead of first ..
bool lp_flat_priorities();             // several contexts live in the process: new streams at the default priority (api.hip)
int lp_fe_calibrate(lpslam_hip_ctx* c, int reserve_cus_per_xcd);
int lp_fe_occupy_unreserved(lpslam_hip_ctx* c, int microseconds, int* landed);
int lp_launch_remap(lpslam_hip_ctx* c, int image, int eye);
int lp_launch_fast(lpslam_hip_ctx* c, int first, int n_images, const uint16_t* list = nullptr);
int lp_launch_distribute(lpslam_hip_ctx* c, int first, int n_images, const uint16_t* list = nullptr);
int lp_launch_describe(lpslam_hip_ctx* c, int first, int n_images, const uint16_t* list = nullptr);
int lp_launch_bf_strided(lpslam_hip_ctx* c, int q0, int t0, int stride, int n_pairs);
int lp_launch_stereo_strided(lpslam_hip_ctx* c, int left0, int right0, int stride, int n_pairs, float fxb, float baseline, const uint16_t* pair_list = nullptr);
size_t lp_distribute_lds_bytes(int qcap_max, int ncell_max);

// ---- launches shared by the sessions of a process (share.hip) ----------------------------------------------------------------
// The reference's deployment unit is one manager per sequence (src/Manager/SlamManager.cpp:54-61,191-201); a process that serves N
// sequences on one GPU then issues N independent chains of small latency-bound launches per frame, and a HIP process has four hardware
// queues: beyond four the chains wait for each other (8 managers ran at 1.5 x one).  When two or more contexts of a
// device are tracking at the same time, the per-frame calls that end in a wait -- the window matchers' first scan, the pose optimiser
// -- become REQUESTS: the calling thread publishes its request and whichever caller finds the combiner free gathers what is pending
// (a few microseconds: callers in lockstep arrive together) and issues ONE launch for all of it, blockIdx = request; every request
// keeps its own page-locked result block and completion flag, and its caller polls that flag as before.  The kernels' code per request
// is the code of the unshared launch (same device function), so results are the same bits either way.
struct LpProjGate { int mode; float inv_sigma_sq[LPSLAM_HIP_MAX_LEVELS]; };
struct LpProjReq {                      // one window-matcher call (k_proj_topk_req); lives in the combiner's page-locked table
    const void* kp; const uint8_t* desc; const float* stereo_xr; const int32_t* kp_count;
    const void* queries; const uint8_t* q_desc; const int16_t* best_so_far;
    unsigned long long* out_keys; int* out_count; unsigned* done_counter; int* done_flag;
    int nq, grid_x, done_seq; float inv_w, inv_h; LpProjGate gate;
};
struct LpPoseReq { uint8_t* blk; int n, seq; };        // one pose optimisation: the caller's page-locked block (ba.hip, PO_BLK_*), observation count, flag value
int lp_launch_proj_batch(hipStream_t s, const LpProjReq* table, int n, int grid_x_max);      // match.hip
int lp_launch_pose_batch(hipStream_t s, const LpPoseReq* reqs, int n);                        // ba.hip (splits into launches of <= 32 requests)
enum { LP_SHARE_DONE = 0, LP_SHARE_DIRECT = 1 };        // (negative: minus an LPSLAM_HIP_ERR_* code)
// Publishes the request and returns LP_SHARE_DONE when ITS flag has arrived, LP_SHARE_DIRECT when sharing does not apply to this call
// (one session tracking alone, sharing switched off, the context's stream still has work the request must follow): the caller then
// launches itself, as before.
int lp_share_pose(lpslam_hip_ctx* c, const LpPoseReq& r, int* flag);
int lp_share_proj(lpslam_hip_ctx* c, const LpProjReq& r);
void lp_share_forget(lpslam_hip_ctx* c);                // the context is being destroyed
void lp_share_frame(lpslam_hip_ctx* c, int inside);     // the session has collected a frame (1) / will make no more latency-bound requests for it (0)
// The four role streams of a device (made and probed at the first call): [0] pose batches, [1] matcher batches = a session's main
// stream, [2] front-end chains = a session's prefetch stream, [3] the windows' solves = a session's bundle-adjustment stream.  The
// sessions of a pool own NO stream: whatever they enqueue goes to the role stream of its kind, so N sessions keep four hardware queues
// busy, not 3 N streams spread over them at the runtime's discretion (a latency-bound launch behind another session's chain on the same
// queue waited for all of it).  false: the streams could not be made.
// A window's local bundle adjustment as a request (lpslam_hip_ba_local_window): the windows that several sessions' mapping threads have
// pending are built and solved together (lp_ba_local_batch, ba.hip) by one of those threads, on the solves' role stream.
struct lpslam_hip_ba;
int lp_share_ba_local(lpslam_hip_ctx* c, lpslam_hip_ba* b, int first_iters, int second_iters, uint8_t* outlier, double* poses, double* points);
int lp_ba_local_batch(lpslam_hip_ba* const* ps, int n, int first_iters, int second_iters, uint8_t* const* outliers, double* const* poses_out, double* const* points_out);
enum { LP_ROLE_POSE = 0, LP_ROLE_MAIN = 1, LP_ROLE_FRONT = 2, LP_ROLE_SOLVE = 3, LP_ROLE_AUX = 4 };
bool lp_share_role_streams(int device, hipStream_t out[5]);
// One frame's front end (extraction of 1 or 2 slots, stereo match, delivery into the session's page-locked block) as a request: the
// frames that several sessions have pending go through ONE launch chain on their pool's stream.  DONE: enqueued; DIRECT: not shared.
struct LpDeliverReq { const int* d_count; const uint32_t* kp; const uint32_t* desc; const uint32_t* xr; const uint32_t* dep; uint32_t* st; unsigned* counter; int* flag; int seq, blocks; };
int lp_share_front_end(lpslam_hip_ctx* c, int slot, int stereo, float fxb, float baseline);
void lp_share_front_end_collected(lpslam_hip_ctx* c);   // the session has its frame (or gave up on it)
hipStream_t lp_aux_stream(lpslam_hip_ctx* c);           // api.hip
int lp_wait_own_uploads(lpslam_hip_ctx* c, int first, int n, hipStream_t s);      // api.hip
int lp_prepare_delivery(lpslam_hip_ctx* c, int image, int with_stereo, LpDeliverReq* out);      // api.hip: what lpslam_hip_prefetch_frame sets up, without the launch
void lp_commit_delivery(lpslam_hip_ctx* c, int image, int with_stereo, const LpDeliverReq& r, hipStream_t s);
int lp_launch_deliver_batch(hipStream_t s, const LpDeliverReq* reqs, int n, int slots_per_image, const lpslam_hip_ctx* layout);

// ---- results delivered by the kernel itself ---------------------------------------------------------------------------------
// A small read-back through the copy engines costs a packet round trip per transfer plus the wait for the stream (15 - 25 us for
// the three or five copies of a tracker call).  The kernels that end such a call instead write their results into page-locked host
// memory and release a sequence number into a flag there as their very last store; the calling thread polls the flag.
// lp_signal_done: every thread of every workgroup calls it at the end of the kernel (after its last store to host memory).
#ifdef __HIPCC__
__device__ __forceinline__ void lp_signal_done(unsigned* counter, int* flag, int seq)
{
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if (total == 1 || atomicAdd(counter, 1u) == total - 1) {
            if (total > 1) { *counter = 0; __threadfence(); }
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
#endif
inline void lp_pf_invalidate(lpslam_hip_ctx* c, int first, int n) { const int p = c->pf_image.load(std::memory_order_relaxed); if (p >= first && p < first + n) c->pf_image.store(-1, std::memory_order_relaxed); }      // the slot's results are being rewritten
unsigned* lp_done_counter(lpslam_hip_ctx* c, int which);      // arrival counter number `which` (0 .. 7) of the context, nullptr on failure
bool lp_wait_recover(lpslam_hip_ctx* c, int which, hipStream_t s);      // after a failed lp_wait_done: counter re-zeroed; false = the stream is dead, leak what its kernels touch
// the next sequence number of a completion flag: positive, never 0 (0 is what the flag is reset to before a launch)
inline int lp_next_seq(int& s) { s = s >= 0x7ffffff0 ? 1 : s + 1; return s; }
// One step of a host-side poll: a pause while the wait is young (a tracker call's kernel is back in 20 - 130 us), then the core is
// OFFERED to other threads between looks (sched_yield returns at once when nobody else is runnable: a lone session keeps its latency).
// Sixteen sessions in one process are sixteen workers and sixteen prefetch threads polling: spinning without yielding, they held every
// core of the GPU's CPU share and starved the threads that feed them (tracker_multi: 16 managers slower than one).
// A process is usually given a CPU quota (a container's cpu.max), not cores of its own: threads that spin through a wait of a hundred
// microseconds spend the quota of every session of the process, and when it runs out the kernel stops ALL of its threads until the next
// period (measured on the GPU boxes: 16 CPUs' worth per 100 ms for 256 visible CPUs; 16 managers = 48 polling threads ran slower than 8).
// So after the young phase of a wait (~10 us of pauses) the thread SLEEPS between looks: a few microseconds each (timer slack 1 us).
void lp_poll_sleep();
inline void lp_poll_pause(int spin) { if (spin < 256) __builtin_ia32_pause(); else lp_poll_sleep(); }
// host side: true when the flag arrived; after ~20 ms without it the stream is synchronised and the flag checked once more
inline bool lp_wait_done(int* flag, int seq, hipStream_t s)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0; ; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return true;
        lp_poll_pause(spin);
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return false;
    return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq;
}
