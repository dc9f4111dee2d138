// api.hip -- C ABI of the lpslam HIP library: context life cycle, geometry tables, frame upload, stage launches and
// readbacks.  Declarations and the reference interfaces they replace: include/lpslam_hip.h.
#include "internal.h"
#include <cstring>
#include <vector>
#include <cmath>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <algorithm>

namespace lpslam {

static thread_local char g_err[512] = "";
thread_local hipStream_t lp_tls_stream = nullptr;
thread_local bool lp_tls_stream_used = false;

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace lpslam
std::atomic<int> g_lp_live_contexts{0};               // contexts of the process (every one of them may have threads polling)
void lp_poll_sleep()
{
    // 5 us between looks; 20 us once more contexts are alive than a 16-CPU quota carries polling threads for (16 managers: 5.4 k -> 6.0 k frames/s)
    static const long sleep_env = [] { const char* e = getenv("LPSLAM_HIP_POLL_SLEEP_US"); return e ? 1000l * std::max(atoi(e), 0) : -1l; }();
    const long sleep_ns = sleep_env >= 0 ? sleep_env : (g_lp_live_contexts.load(std::memory_order_relaxed) > 10 ? 20000l : 5000l);
    if (sleep_ns <= 0) { sched_yield(); return; }
    static thread_local bool slack_set = false;
    if (!slack_set) { (void)prctl(PR_SET_TIMERSLACK, 1000ul, 0ul, 0ul, 0ul); slack_set = true; }      // (the default slack of 50 us would turn a 5 us sleep into 55)
    const struct timespec ts{0, sleep_ns};
    (void)nanosleep(&ts, nullptr);
}
namespace lpslam {

int hip_fail(hipError_t e, const char* what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return LPSLAM_HIP_ERR_DEVICE;
}

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

// cv::resize INTER_LINEAR coefficient table for one axis: offset and two 11-bit weights per destination index
// ([UPSTREAM] resizeGeneric_: fx = (float)((d + 0.5) * scale - 0.5); weights = cvRound(w * 2048)).
static void resize_axis(int ssize, int dsize, std::vector<int16_t>& ofs, std::vector<int16_t>& coef)
{
    const double scale = 1.0 / ((double)dsize / ssize);
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= (float)s;
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
        ofs.push_back((int16_t)s);
        coef.push_back((int16_t)std::nearbyint((1.f - f) * 2048.f));
        coef.push_back((int16_t)std::nearbyint(f * 2048.f));
    }
}

// Geometry of the path: pyramid sizes, FAST cell grid, per-level quota, quad-tree roots.
static int build_level_table(const lpslam_hip_frontend_config& cfg, LevelTable& lt, size_t& image_slab,
                             std::vector<int16_t>& ofs, std::vector<int16_t>& coef)
{
    const int L = cfg.num_levels;
    lt.n_levels = L;
    lt.scale[0] = 1.0f;
    for (int l = 1; l < L; ++l) lt.scale[l] = cfg.scale_factor * lt.scale[l - 1];
    for (int l = 0; l < L; ++l) lt.inv_scale[l] = 1.0f / lt.scale[l];
    unsigned off = 0;
    lt.cell_start[0] = 0; lt.slot_start[0] = 0; lt.cand_start[0] = 0;
    // quota per level: geometric share, remainder on the last level
    {
        const double f = 1.0 / (double)cfg.scale_factor;
        double desired = cfg.max_keypoints * (1.0 - f) / (1.0 - std::pow(f, (double)L));
        int total = 0;
        for (int l = 0; l < L - 1; ++l) { lt.quota[l] = (int)std::round(desired); total += lt.quota[l]; desired *= f; }
        lt.quota[L - 1] = std::max(cfg.max_keypoints - total, 0);
    }
    for (int l = 0; l < L; ++l) {
        if (l == 0) { lt.w[0] = cfg.width; lt.h[0] = cfg.height; }
        else {
            const double s = lt.scale[l];
            lt.w[l] = (int)std::round(cfg.width * 1.0 / s);
            lt.h[l] = (int)std::round(cfg.height * 1.0 / s);
        }
        if (lt.w[l] <= 2 * kEdge + 7 || lt.h[l] <= 2 * kEdge + 7) {
            set_error("level %d (%dx%d) is too small for the 19-px border; reduce num_levels", l, lt.w[l], lt.h[l]);
            return LPSLAM_HIP_ERR_INVALID;
        }
        lt.pitch[l] = align_up(lt.w[l], 64);
        lt.off[l] = off;
        off += (unsigned)lt.pitch[l] * (unsigned)align_up(lt.h[l], 4);
        // FAST cells: rows/cols whose start lies more than `overlap` px before the far border
        const int max_bx = lt.w[l] - kEdge, max_by = lt.h[l] - kEdge;
        const int ncols = (max_bx - kEdge) / kCell + 1, nrows = (max_by - kEdge) / kCell + 1;
        int cx = 0, cy = 0;
        for (int j = 0; j < ncols; ++j) if (!(max_bx - kOverlap <= kEdge + j * kCell)) cx = j + 1;
        for (int i = 0; i < nrows; ++i) if (!(max_by - kOverlap <= kEdge + i * kCell)) cy = i + 1;
        lt.cells_x[l] = cx; lt.cells_y[l] = cy;
        lt.cell_start[l + 1] = lt.cell_start[l] + cx * cy;
        lt.cand_start[l + 1] = lt.cand_start[l] + cx * cy * kCellSlots;
        // quad-tree roots
        const int width = max_bx - kEdge, height = max_by - kEdge;
        const double ratio = (double)width / (double)height;
        if (ratio > 1) { lt.nxg[l] = (int)std::round(ratio); lt.nyg[l] = 1; lt.delta_x[l] = (double)width / lt.nxg[l]; lt.delta_y[l] = height; }
        else { lt.nxg[l] = 1; lt.nyg[l] = (int)std::round(1 / ratio); lt.delta_x[l] = width; lt.delta_y[l] = (double)height / lt.nyg[l]; }
        const int roots = lt.nxg[l] * lt.nyg[l];
        lt.qcap[l] = std::max(lt.quota[l], 4 * roots) + 4;
        lt.slot_start[l + 1] = lt.slot_start[l] + std::max(lt.quota[l] + 3, 4 * roots);
        if (lt.quota[l] > kQuotaMax || lt.qcap[l] >= 2048) {
            set_error("per-level keypoint quota %d exceeds the supported %d", lt.quota[l], kQuotaMax);
            return LPSLAM_HIP_ERR_INVALID;
        }
        if (cx * cy * kCellSlots >= (1 << 21) || lt.w[l] >= 4096 + 2 * kEdge || lt.h[l] >= 4096 + 2 * kEdge) {
            set_error("level %d (%dx%d) exceeds the supported image size", l, lt.w[l], lt.h[l]);
            return LPSLAM_HIP_ERR_INVALID;
        }
        if (l >= 1) {
            lt.xtab_start[l] = (int)ofs.size();
            resize_axis(lt.w[l - 1], lt.w[l], ofs, coef);
            lt.ytab_start[l] = (int)ofs.size();
            resize_axis(lt.h[l - 1], lt.h[l], ofs, coef);
        } else { lt.xtab_start[0] = lt.ytab_start[0] = 0; }
    }
    image_slab = (size_t)align_up((int)off, 256);
    return LPSLAM_HIP_OK;
}

}  // namespace lpslam

using namespace lpslam;

// ---- block cache ---------------------------------------------------------------------------------------------------------
// Every context keeps the device blocks it has released (bundle-adjustment problems, matcher scratch) for the next request of that
// size: a hipMalloc is 0.1 - 1 ms.  The cache is bounded per context (lp_pool_cap: 1/32 of the device's memory, at most 16 GB;
// LPSLAM_HIP_POOL_CAP_MB overrides) and gives way under pressure: an allocation that fails flushes the cached blocks of EVERY
// context of the process on that device and is tried once more, so a server with one context per session does not fail an
// allocation while gigabytes sit idle in its other sessions' caches.
namespace {
std::mutex g_ctx_mutex;
std::vector<lpslam_hip_ctx*> g_contexts;              // live contexts of the process (registered at creation)
size_t lp_pool_flush_locked(lpslam_hip_ctx* c)        // c->pool_mutex held
{
    size_t freed = 0;
    for (auto& blk : c->pool) { (void)hipFree(blk.second); freed += blk.first; }
    c->pool.clear(); c->pool_bytes = 0;
    return freed;
}
}
extern std::atomic<int> g_lp_live_contexts;
void lp_ctx_register(lpslam_hip_ctx* c, bool add)
{
    g_lp_live_contexts.fetch_add(add ? 1 : -1);
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    if (add) g_contexts.push_back(c);
    else g_contexts.erase(std::remove(g_contexts.begin(), g_contexts.end(), c), g_contexts.end());
}
// Stream priorities cost hardware queues: every priority class a process uses takes its own set (four per class), and a process
// that hosts several sessions -- a main, a low-priority prefetch and high-priority solve streams each -- then keeps twelve queues busy
// that the command processor time-slices.  A single session wants the classes (the current frame's matchers and pose optimisations win
// the compute units from the background extraction and never share a queue with it); a process with MANY sessions per
// GPU does better with every stream at the default priority: lpslam_hip_set_flat_priorities(1) before the sessions are created (or
// LPSLAM_HIP_FLAT_PRIORITIES=1).  Measured (tools/dev_tracker_multi.py, MI355X): 8 managers 2800 -> 3380 frames/s aggregate, 16 managers
// 1980 -> 2400.  Mixing both kinds of streams in one process is worse than either (a switch at the third context: 2470 / 1590).
static std::atomic<int> g_flat_priorities{-1};          // -1: the environment decides
bool lp_flat_priorities()
{
    const int v = g_flat_priorities.load();
    if (v >= 0) return v != 0;
    static const bool env = [] { const char* e = getenv("LPSLAM_HIP_FLAT_PRIORITIES"); return e && atoi(e) != 0; }();
    return env;
}
static std::atomic<bool> g_priority_stream_made{false};      // a stream outside the default priority has been created in this process (it keeps its hardware queues for good)
extern "C" int lpslam_hip_set_flat_priorities(int32_t flat)
{
    if (flat == 1 && g_priority_stream_made.load()) {      // (2: flat although late -- tests and measurements of exactly that case)
        // too late: the process already holds hardware queues of another priority class, and every session created from now on would run
        // beside them at half the aggregate -- the caller is told instead of being left to find out
        set_error("lpslam_hip_set_flat_priorities(1) after a priority stream was created in this process: call it before the first context");
        fprintf(stderr, "lpslam_hip: %s\n", lpslam_hip_last_error());
        return LPSLAM_HIP_ERR_INVALID;
    }
    g_flat_priorities.store(flat < 0 ? -1 : (flat ? 1 : 0));
    return LPSLAM_HIP_OK;
}
size_t lp_pool_flush_device(int device)
{
    size_t freed = 0;
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    for (lpslam_hip_ctx* o : g_contexts) {
        if (o->cfg.device != device) continue;
        std::lock_guard<std::mutex> l2(o->pool_mutex);
        freed += lp_pool_flush_locked(o);
    }
    return freed;
}
static size_t lp_pool_cap(lpslam_hip_ctx* c)
{
    if (c->pool_cap) return c->pool_cap;
    size_t cap = 16ull << 30;
    if (const char* e = getenv("LPSLAM_HIP_POOL_CAP_MB")) cap = (size_t)std::max(0l, atol(e)) << 20;
    else { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) cap = std::min(cap, tot / 32); else (void)hipGetLastError(); }
    c->pool_cap = std::max<size_t>(cap, 1);
    return c->pool_cap;
}

int lp_pool_alloc(lpslam_hip_ctx* c, size_t bytes, void** out, size_t* capacity)
{
    const size_t want = ((std::max<size_t>(bytes, 1) + 4095) / 4096) * 4096;
    {
        std::lock_guard<std::mutex> lock(c->pool_mutex);
        size_t best = c->pool.size();
        for (size_t i = 0; i < c->pool.size(); ++i)
            if (c->pool[i].first >= want && c->pool[i].first <= 2 * want + 65536 && (best == c->pool.size() || c->pool[i].first < c->pool[best].first)) best = i;
        if (best != c->pool.size()) {
            *out = c->pool[best].second; *capacity = c->pool[best].first;
            c->pool_bytes -= c->pool[best].first;
            c->pool[best] = c->pool.back(); c->pool.pop_back();
            return LPSLAM_HIP_OK;
        }
    }
    hipError_t e = hipMalloc(out, want);
    if (e == hipErrorOutOfMemory) {          // idle blocks of this and the sibling contexts go back to the device, then once more
        (void)hipGetLastError();
        if (lp_pool_flush_device(c->cfg.device) > 0) e = hipMalloc(out, want);
    }
    LP_HIP(e);
    *capacity = want;
    return LPSLAM_HIP_OK;
}

// A front-end stream of the context (never CU-masked: the mapping reserve is kept by the kernels themselves, frontend.hip).  The
// prefetch stream is created at the LOWEST priority: it carries the next frame's extraction -- 0.2 ms of chip-filling kernels --
// beside this frame's matchers and pose optimisations on the main stream, which are a few workgroups each and latency bound.  With
// equal priorities the two streams can end up on one hardware queue (a process that has created many streams: measured in bench.py,
// every local-map matcher call waited 0.2 ms behind the prefetched extraction); different priorities never share a queue.
static hipError_t lp_fe_stream_create(hipStream_t* s, bool background)
{
    if (background && !lp_flat_priorities()) {
        int prio_least = 0, prio_greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess && prio_least != prio_greatest &&
            hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio_least) == hipSuccess) { g_priority_stream_made.store(true); return hipSuccess; }
        (void)hipGetLastError();
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

hipStream_t lp_stream_acquire(lpslam_hip_ctx* c)
{
    if (c->sess_pool && c->sess_pool->pool_refs > 1) {   // several sessions: the solves' role stream, shared with the other sessions' windows
        if (!c->role_solve) { hipStream_t roles[5]; if (lp_share_role_streams(c->cfg.device, roles)) c->role_solve = roles[LP_ROLE_SOLVE]; }
        if (c->role_solve) return c->role_solve;
    }
    if (!c->owns_streams) return c->role_solve;
    {
        std::lock_guard<std::mutex> lock(c->pool_mutex);
        if (!c->ba_streams.empty()) { hipStream_t s = c->ba_streams.back(); c->ba_streams.pop_back(); return s; }
    }
    // a bundle adjustment runs beside the front end of later frames (the reference's mapping thread), at the highest priority: its
    // kernels are small and latency bound, the front end's fill the chip for 100 us at a time
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const bool flat = lp_flat_priorities();
    hipStream_t s = nullptr;
    if ((flat ? hipStreamCreateWithFlags(&s, hipStreamNonBlocking) : hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio_greatest)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (!flat && prio_greatest != prio_least) g_priority_stream_made.store(true);
    return s;
}

void lp_stream_release(lpslam_hip_ctx* c, hipStream_t s)
{
    if (!s || s == c->role_solve) return;
    std::lock_guard<std::mutex> lock(c->pool_mutex);
    c->ba_streams.push_back(s);
}

void* lp_pin_big_alloc(lpslam_hip_ctx* c, size_t bytes, size_t* capacity)
{
    const size_t want = ((std::max<size_t>(bytes, 1) + 65535) / 65536) * 65536;
    {
        std::lock_guard<std::mutex> lock(c->pool_mutex);
        for (size_t i = 0; i < c->pin_big.size(); ++i)
            if (c->pin_big[i].first >= want && c->pin_big[i].first <= 4 * want) {
                void* p = c->pin_big[i].second; *capacity = c->pin_big[i].first;
                c->pin_big[i] = c->pin_big.back(); c->pin_big.pop_back();
                return p;
            }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *capacity = want;
    return p;
}

void lp_pin_big_free(lpslam_hip_ctx* c, void* p, size_t capacity)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lock(c->pool_mutex);
        if (c->pin_big.size() < 256) { c->pin_big.emplace_back(capacity, p); return; }      // (16 made a 16-session batch -- a staging and an exchange block per window -- free and re-allocate page-locked memory every round: milliseconds)
    }
    (void)hipHostFree(p);
}

void* lp_pin_alloc(lpslam_hip_ctx* c)
{
    {
        std::lock_guard<std::mutex> lock(c->pool_mutex);
        if (!c->pin_free.empty()) { void* p = c->pin_free.back(); c->pin_free.pop_back(); return p; }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, 8192, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void lp_pin_free(lpslam_hip_ctx* c, void* p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(c->pool_mutex);
    c->pin_free.push_back(p);
}

void lp_pool_free(lpslam_hip_ctx* c, void* p, size_t capacity)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lock(c->pool_mutex);
        // (a batch of 16 local windows holds 0.7 GB; a 2 GB bound made a multi-session server free and re-allocate its blocks every round)
        if (c->pool_bytes + capacity <= lp_pool_cap(c)) { c->pool.emplace_back(capacity, p); c->pool_bytes += capacity; return; }
    }
    (void)hipFree(p);
}

extern "C" {

const char* lpslam_hip_last_error(void) { return g_err; }

int lpslam_hip_device_count(int* count)
{
    if (!count) { set_error("null argument"); return LPSLAM_HIP_ERR_INVALID; }
    *count = 0;
    LP_HIP(hipGetDeviceCount(count));
    return LPSLAM_HIP_OK;
}

static int ctx_alloc(lpslam_hip_ctx* c)
{
    const size_t B = (size_t)c->cfg.max_images;
    const int L = c->lt.n_levels;
    c->st_row_cap = c->slots_per_image * ((int)(4.0f * c->lt.scale[c->lt.n_levels - 1]) + 4);      // rows per keypoint <= 4 * scale + 3
    if (c->sess_pool) {
        // a session: every per-image array is the pool's, from image pool_first on (same configuration: same per-image sizes)
        const lpslam_hip_ctx* p = c->sess_pool;
        const size_t f = (size_t)c->pool_first;
        c->d_pyr = p->d_pyr + f * c->image_slab;
        c->d_cell_keys = p->d_cell_keys + f * c->cells_per_image * kCellSlots; c->d_cell_count = p->d_cell_count + f * c->cells_per_image;
        c->d_cand_key = p->d_cand_key + f * c->cand_per_image; c->d_cand_node = p->d_cand_node + f * c->cand_per_image; c->d_cand_count = p->d_cand_count + f * L;
        c->d_sel_key = p->d_sel_key + f * c->slots_per_image; c->d_sel_count = p->d_sel_count + f * L;
        c->d_kpts = p->d_kpts + f * c->slots_per_image; c->d_desc = p->d_desc + f * c->slots_per_image * 32; c->d_kp_count = p->d_kp_count + f;
        c->d_bf = p->d_bf + f * 3 * c->slots_per_image; c->d_stereo = p->d_stereo + f * 2 * c->slots_per_image;
        c->d_stereo_idx = p->d_stereo_idx + f * c->slots_per_image; c->d_stereo_corr = p->d_stereo_corr + f * c->slots_per_image;
        c->d_st_row_start = p->d_st_row_start + f * (size_t)(c->lt.h[0] + 1); c->d_st_row_list = p->d_st_row_list + f * (size_t)c->st_row_cap;
        LP_HIP(hipMemsetAsync(c->d_pyr, 0, B * c->image_slab, c->stream));
        LP_HIP(hipMemsetAsync(c->d_kp_count, 0, B * sizeof(int32_t), c->stream));
        LP_HIP(hipMemsetAsync(c->d_sel_count, 0, B * L * sizeof(int32_t), c->stream));
        LP_HIP(hipMemsetAsync(c->d_cand_count, 0, B * L * sizeof(int32_t), c->stream));
        return LPSLAM_HIP_OK;
    }
    LP_HIP(hipMalloc((void**)&c->d_pyr, B * c->image_slab));
    LP_HIP(hipMemsetAsync(c->d_pyr, 0, B * c->image_slab, c->stream));
    LP_HIP(hipMalloc((void**)&c->d_cell_keys, B * c->cells_per_image * kCellSlots * sizeof(uint32_t)));
    LP_HIP(hipMalloc((void**)&c->d_cell_count, B * c->cells_per_image * sizeof(int32_t)));
    LP_HIP(hipMalloc((void**)&c->d_cand_key, B * c->cand_per_image * sizeof(uint32_t)));
    LP_HIP(hipMalloc((void**)&c->d_cand_node, B * c->cand_per_image * sizeof(uint32_t)));
    LP_HIP(hipMalloc((void**)&c->d_cand_count, B * L * sizeof(int32_t)));
    LP_HIP(hipMalloc((void**)&c->d_sel_key, B * c->slots_per_image * sizeof(uint32_t)));
    LP_HIP(hipMalloc((void**)&c->d_sel_count, B * L * sizeof(int32_t)));
    LP_HIP(hipMalloc((void**)&c->d_kpts, B * c->slots_per_image * sizeof(lpslam_hip_keypoint)));
    LP_HIP(hipMalloc((void**)&c->d_desc, B * c->slots_per_image * 32));
    LP_HIP(hipMalloc((void**)&c->d_kp_count, B * sizeof(int32_t)));
    LP_HIP(hipMemsetAsync(c->d_kp_count, 0, B * sizeof(int32_t), c->stream));
    LP_HIP(hipMemsetAsync(c->d_sel_count, 0, B * L * sizeof(int32_t), c->stream));
    LP_HIP(hipMemsetAsync(c->d_cand_count, 0, B * L * sizeof(int32_t), c->stream));
    LP_HIP(hipMalloc((void**)&c->d_bf, B * 3 * c->slots_per_image * sizeof(int32_t)));
    LP_HIP(hipMalloc((void**)&c->d_stereo, B * 2 * c->slots_per_image * sizeof(float)));
    LP_HIP(hipMalloc((void**)&c->d_stereo_idx, B * c->slots_per_image * sizeof(int32_t)));
    LP_HIP(hipMalloc((void**)&c->d_stereo_corr, B * c->slots_per_image * sizeof(int32_t)));
    LP_HIP(hipMalloc((void**)&c->d_st_row_start, B * (size_t)(c->lt.h[0] + 1) * sizeof(int32_t)));
    LP_HIP(hipMalloc((void**)&c->d_st_row_list, B * (size_t)c->st_row_cap * sizeof(int32_t)));
    return LPSLAM_HIP_OK;
}

static bool ensure_upload_staging(lpslam_hip_ctx* c, int image);
int lp_wait_uploads(lpslam_hip_ctx* c, int first, int n);

static int create_impl(const lpslam_hip_frontend_config* cfg, lpslam_hip_ctx* pool, int pool_slot, lpslam_hip_ctx** out, bool* attached = nullptr);

// ---- session pools ---------------------------------------------------------------------------------------------------------
// One pool per (device, front-end configuration), created with the first session that asks for it and destroyed with the last one
// that leaves.  LPSLAM_HIP_POOL_SESSIONS (16) sessions fit; the next one gets arrays of its own (and launches of its own).
namespace {
std::mutex g_pools_mutex;
std::vector<lpslam_hip_ctx*> g_pools;
bool same_front_end(const lpslam_hip_frontend_config& a, const lpslam_hip_frontend_config& b)
{
    return a.device == b.device && a.width == b.width && a.height == b.height && a.max_keypoints == b.max_keypoints && a.scale_factor == b.scale_factor &&
           a.num_levels == b.num_levels && a.ini_fast_threshold == b.ini_fast_threshold && a.min_fast_threshold == b.min_fast_threshold;
}
}
static void lp_pool_session_release(lpslam_hip_ctx* pool, int slot)
{
    bool last = false;
    {
        std::lock_guard<std::mutex> lock(g_pools_mutex);
        if (slot >= 0 && (size_t)slot < pool->pool_used.size()) pool->pool_used[(size_t)slot] = 0;
        if (--pool->pool_refs == 0) { g_pools.erase(std::remove(g_pools.begin(), g_pools.end(), pool), g_pools.end()); last = true; }
    }
    if (last) lpslam_hip_destroy(pool);
}

int lpslam_hip_create_session(const lpslam_hip_frontend_config* cfg, lpslam_hip_ctx** out)
{
    if (!cfg || !out) { set_error("null argument"); return LPSLAM_HIP_ERR_INVALID; }
    static const int sessions_env = [] { const char* e = getenv("LPSLAM_HIP_POOL_SESSIONS"); return e ? std::min(std::max(atoi(e), 0), 64) : 16; }();
    const int per = (cfg->max_images + 1) & ~1;          // stereo pairs keep their parity (left even, right odd: the eyes' masks)
    if (sessions_env < 2 || cfg->max_images < 1 || cfg->max_images > 8 || per * sessions_env > 65535) return lpslam_hip_create(cfg, out);
    lpslam_hip_ctx* pool = nullptr;
    int slot = -1;
    {
        std::lock_guard<std::mutex> lock(g_pools_mutex);
        for (lpslam_hip_ctx* p : g_pools) if (same_front_end(p->cfg, *cfg) && p->pool_per == per) { pool = p; break; }
        if (!pool) {
            lpslam_hip_frontend_config pc = *cfg;
            pc.max_images = per * sessions_env;
            lpslam_hip_ctx* p = nullptr;
            if (create_impl(&pc, nullptr, -1, &p) != LPSLAM_HIP_OK) return lpslam_hip_create(cfg, out);      // (no room for a pool: a context of its own)
            p->is_pool = true; p->pool_per = per; p->pool_used.assign((size_t)sessions_env, 0);
            // (a mapping reserve on the pool's chains -- compute units of every XCD left to the matchers, pose optimisers and windows beside them
            // -- was measured at 2 / 4 / 8 / 16: 8 managers 4694 / 5159 / 4927 / 4469 against 5045 without; the pool runs without one)
            g_pools.push_back(p);
            pool = p;
        }
        for (size_t i = 0; i < pool->pool_used.size() && slot < 0; ++i) if (!pool->pool_used[i]) slot = (int)i;
        if (slot >= 0) { pool->pool_used[(size_t)slot] = 1; ++pool->pool_refs; }
    }
    if (slot < 0) return lpslam_hip_create(cfg, out);    // the pool is full
    bool attached = false;
    const int rc = create_impl(cfg, pool, slot, out, &attached);
    if (rc != LPSLAM_HIP_OK && !attached) lp_pool_session_release(pool, slot);      // (a creation that failed later handed the slot back in lpslam_hip_destroy)
    return rc;
}

int lpslam_hip_create(const lpslam_hip_frontend_config* cfg, lpslam_hip_ctx** out) { return create_impl(cfg, nullptr, -1, out); }

static int create_impl(const lpslam_hip_frontend_config* cfg, lpslam_hip_ctx* pool, int pool_slot, lpslam_hip_ctx** out, bool* attached)
{
    if (!cfg || !out) { set_error("null argument"); return LPSLAM_HIP_ERR_INVALID; }
    *out = nullptr;
    if (cfg->num_levels < 1 || cfg->num_levels > kMaxLevels || cfg->width < 64 || cfg->height < 64 ||
        cfg->max_keypoints < 1 || cfg->max_images < 1 || !(cfg->scale_factor > 1.0f) ||
        cfg->ini_fast_threshold < 0 || cfg->min_fast_threshold < 0) {
        set_error("invalid front-end configuration (%dx%d, %d levels, scale %.3f, %d keypoints, %d images)", cfg->width,
                  cfg->height, cfg->num_levels, (double)cfg->scale_factor, cfg->max_keypoints, cfg->max_images);
        return LPSLAM_HIP_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available: the lpslam hot path requires an MI355X (gfx950) GPU; there is no CPU fallback");
        return LPSLAM_HIP_ERR_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { set_error("device %d out of range (%d devices)", cfg->device, ndev); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(cfg->device));

    lpslam_hip_ctx* c = new lpslam_hip_ctx();
    c->cfg = *cfg;
    c->h_kp_count.assign((size_t)cfg->max_images, 0); c->h_kp_valid.assign((size_t)cfg->max_images, 0);
    std::vector<int16_t> ofs, coef;
    int rc = build_level_table(*cfg, c->lt, c->image_slab, ofs, coef);
    if (rc != LPSLAM_HIP_OK) { delete c; return rc; }
    const int L = c->lt.n_levels;
    c->cells_per_image = c->lt.cell_start[L];
    c->cand_per_image = c->lt.cand_start[L];
    c->slots_per_image = c->lt.slot_start[L];
    // keypoint indices travel as 16 bits in the window matchers (the candidate key's index field, the survivor list and the best-so-far
    // table of k_proj_topk, match.hip)
    if (c->slots_per_image > 65535) { set_error("%d keypoint slots per image exceed the matchers' 16-bit keypoint index", c->slots_per_image); delete c; return LPSLAM_HIP_ERR_CAPACITY; }
    size_t lds = 0;
    for (int l = 0; l < L; ++l) {
        lds = std::max(lds, lp_distribute_lds_bytes(c->lt.qcap[l], c->lt.cells_x[l] * c->lt.cells_y[l]));
    }
    c->distribute_lds = lds;
    if (lds > 160 * 1024) { set_error("distribution kernel needs %zu B of LDS (> 160 KiB)", lds); delete c; return LPSLAM_HIP_ERR_INVALID; }

    // Many sessions in one process (flat priorities): a HIP process has four hardware queues per priority and a new stream takes the least
    // used one, so the contexts' streams -- created main first -- would put every main stream (the latency-bound matchers and pose
    // optimisations of a tracked frame) on the SAME queue, behind one another: two managers ran at 1.24 x one.  Context k of the process
    // puts k mod 4 placeholder streams in front of its main stream.
    hipStream_t roles[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // (the FIRST session of a pool keeps streams of its own, in the three priority classes: a session alone is what it always was; those that
    // join later -- a process that hosts several -- use the role streams)
    const bool session_streams = pool && pool->pool_refs > 1 && lp_share_role_streams(cfg->device, roles);
    if (session_streams) {
        // a session of a pool: its streams are the device's role streams (share.hip)
        c->owns_streams = false; c->stream = roles[LP_ROLE_MAIN]; c->fe_stream = roles[LP_ROLE_FRONT]; c->role_solve = roles[LP_ROLE_SOLVE]; c->role_aux = roles[LP_ROLE_AUX];
        if (hipEventCreateWithFlags(&c->fe_done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); delete c; set_error("hipEventCreate failed"); return LPSLAM_HIP_ERR_DEVICE; }
    } else
    // (best effort: it leans on how this runtime binds streams to its four queues -- counted per device, and left alone when the caller has
    // changed the number of queues; sessions that join a pool use the measured role streams instead, share.hip)
    if (lp_flat_priorities() && !getenv("LPSLAM_HIP_NO_QUEUE_SPREAD") && !getenv("GPU_MAX_HW_QUEUES")) {
        static std::atomic<int> ctx_seq[64];
        const int k = ctx_seq[cfg->device & 63].fetch_add(1) % 4;
        if (k > 0 && hipMalloc(&c->d_pad, 256) == hipSuccess) {
            for (int i = 0; i < k; ++i) {
                hipStream_t d = nullptr;
                if (hipStreamCreateWithFlags(&d, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
                (void)hipMemsetAsync(c->d_pad, 0, 4, d);           // (a stream takes its queue at its first use)
                (void)hipStreamSynchronize(d);
                c->pad_streams.push_back(d);
            }
        }
    }
    hipError_t e = session_streams ? hipSuccess : lp_fe_stream_create(&c->stream, false);
    if (e != hipSuccess) {
        for (hipStream_t d : c->pad_streams) (void)hipStreamDestroy(d);
        if (c->d_pad) (void)hipFree(c->d_pad);
        delete c;
        return hip_fail(e, "hipStreamCreate");
    }
    if (c->d_pad) { (void)hipMemsetAsync(c->d_pad, 0, 4, c->stream); (void)hipStreamSynchronize(c->stream); }
    if (pool) { c->sess_pool = pool; c->pool_slot = pool_slot; c->pool_first = pool_slot * pool->pool_per; if (attached) *attached = true; }      // from here on lpslam_hip_destroy hands the pool's slot back
    lp_ctx_register(c, true);
    rc = ctx_alloc(c);
    if (rc == LPSLAM_HIP_OK && !ofs.empty()) {
        std::vector<int2> pack;
        pack.reserve(2 * ofs.size());
        for (int l = 1; l < L; ++l) {
            auto entry = [&](int axis, int i) {
                const int start = axis ? c->lt.ytab_start[l] : c->lt.xtab_start[l];
                const int src_n = axis ? c->lt.h[l - 1] : c->lt.w[l - 1];
                const int s0 = ofs[start + i], s1 = std::min(s0 + 1, src_n - 1);
                return make_int2(s0 | (s1 << 16), (int)(uint16_t)coef[2 * (start + i)] | ((int)coef[2 * (start + i) + 1] << 16));
            };
            const int per_row = c->lt.pitch[l] >> 2;
            c->lt.dx_start[l] = (int)pack.size();
            for (int k = 0; k < 4; ++k)
                for (int q = 0; q < per_row; ++q) pack.push_back(entry(0, std::min(4 * q + k, c->lt.w[l] - 1)));
            c->lt.dy_start[l] = (int)pack.size();
            for (int i = 0; i < c->lt.h[l]; ++i) pack.push_back(entry(1, i));
        }
        c->lt.dx_start[0] = c->lt.dy_start[0] = 0;
        c->rs_entries = (int)pack.size();
        e = hipMalloc((void**)&c->d_rs_pack, pack.size() * sizeof(int2));
        if (e == hipSuccess) e = hipMemcpy(c->d_rs_pack, pack.data(), pack.size() * sizeof(int2), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = hip_fail(e, "resize tables");
    }
    if (rc == LPSLAM_HIP_OK && L > 1) {
        // Row ranges of the banded pyramid kernel: band b owns rows [h*b/B, h*(b+1)/B) of every level and also computes the rows
        // of the finer levels its own coarser rows are interpolated from, so a band never waits for a neighbour.
        std::vector<int2> rows((size_t)(kPyrMaxBands + 1) * kMaxLevels * kPyrMaxBands, make_int2(0, -1));
        for (int s = 1; s <= kPyrMaxBands; ++s) {      // one table per band count
            const int B = s;
            for (int b = 0; b < B; ++b) {
                int lo = 0, hi = -1;
                for (int l = L - 1; l >= 1; --l) {
                    const int h = c->lt.h[l];
                    int own_lo = (int)((long)h * b / B), own_hi = (int)((long)h * (b + 1) / B) - 1;
                    if (own_hi >= own_lo) {
                        if (hi < lo) { lo = own_lo; hi = own_hi; } else { lo = std::min(lo, own_lo); hi = std::max(hi, own_hi); }
                    }
                    rows[((size_t)s * kMaxLevels + l) * kPyrMaxBands + b] = make_int2(lo, hi);
                    if (hi >= lo && l > 1) {      // rows of level l-1 these rows read
                        const int yt = c->lt.ytab_start[l];
                        const int nlo = ofs[yt + lo], nhi = std::min(ofs[yt + hi] + 1, c->lt.h[l - 1] - 1);
                        lo = nlo; hi = nhi;
                    }
                }
            }
        }
        e = hipMalloc((void**)&c->d_band_rows, rows.size() * sizeof(int2));
        if (e == hipSuccess) e = hipMemcpy(c->d_band_rows, rows.data(), rows.size() * sizeof(int2), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = hip_fail(e, "pyramid band table");
    }
    if (rc == LPSLAM_HIP_OK) {
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize");
    }
    // a tracker-sized context gets its upload staging now (page-locked allocations cost ~0.5 ms each: not inside the first frames);
    // a large resident ring allocates per slot on first use
    if (rc == LPSLAM_HIP_OK && cfg->max_images <= 8)
        for (int i = 0; i < cfg->max_images && rc == LPSLAM_HIP_OK; ++i) if (!ensure_upload_staging(c, i)) rc = LPSLAM_HIP_ERR_DEVICE;
    if (rc != LPSLAM_HIP_OK) { lpslam_hip_destroy(c); return rc; }
    *out = c;
    return LPSLAM_HIP_OK;
}

void lpslam_hip_destroy(lpslam_hip_ctx* c)
{
    if (!c) return;
    lp_ctx_register(c, false);
    lp_share_forget(c);
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->fe_stream) (void)hipStreamSynchronize(c->fe_stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (auto& kv : c->desc_store) if (kv.second.blk) (void)hipFree(kv.second.blk);
    c->desc_store.clear();
    for (auto& blk : c->pool) (void)hipFree(blk.second);
    c->pool.clear();
    lpslam_hip_ctx* const pool_of = c->sess_pool;
    const int pool_slot_of = c->pool_slot;
    if (pool_of) {
        // a session: its per-image arrays are the pool's.  A shared front end of this session may still run on the pool's stream.
        if (pool_of->stream) (void)hipStreamSynchronize(pool_of->stream);
        c->d_pyr = nullptr; c->d_cell_keys = nullptr; c->d_cell_count = nullptr; c->d_cand_key = nullptr; c->d_cand_node = nullptr; c->d_cand_count = nullptr;
        c->d_sel_key = nullptr; c->d_sel_count = nullptr; c->d_kpts = nullptr; c->d_desc = nullptr; c->d_kp_count = nullptr; c->d_bf = nullptr; c->d_stereo = nullptr;
        c->d_stereo_idx = nullptr; c->d_stereo_corr = nullptr; c->d_st_row_start = nullptr; c->d_st_row_list = nullptr;
    }
    if (c->ev_fe_ready) (void)hipEventDestroy(c->ev_fe_ready);
    if (c->up_stream) { (void)hipStreamSynchronize(c->up_stream); (void)hipStreamDestroy(c->up_stream); }
    void* bufs[] = {c->d_pyr, c->d_band_rows, c->d_rs_pack, c->d_cell_keys, c->d_cell_count, c->d_cand_key, c->d_cand_node,
                    c->d_cand_count, c->d_sel_key, c->d_sel_count, c->d_kpts, c->d_desc,
                    c->d_kp_count, c->d_bf, c->d_stereo, c->d_stereo_idx, c->d_stereo_corr, c->d_st_row_start, c->d_st_row_list, c->d_tmp_desc, c->d_tmp_res,
                    c->d_map_xy[0], c->d_map_xy[1], c->d_map_frac[0], c->d_map_frac[1], c->d_raw, c->d_mask[0], c->d_mask[1]};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (c->d_cu_table) (void)hipFree(c->d_cu_table);
    if (c->d_fe_counters) (void)hipFree(c->d_fe_counters);
    if (c->d_done) (void)hipFree(c->d_done);
    for (int i = 0; i < LPSLAM_HIP_MAX_TIMERS; ++i) { if (c->ev_begin[i]) (void)hipEventDestroy(c->ev_begin[i]); if (c->ev_end[i]) (void)hipEventDestroy(c->ev_end[i]); }
    for (void* p : c->pin_free) (void)hipHostFree(p);
    c->pin_free.clear();
    for (auto& pb : c->pin_big) (void)hipHostFree(pb.second);
    c->pin_big.clear();
    for (auto& g : c->ba_graphs) if (g.second && g.second != (hipGraphExec_t)(uintptr_t)1) (void)hipGraphExecDestroy(g.second);      // 1 = the failed-capture sentinel (ba.hip)
    c->ba_graphs.clear();
    for (auto& v : c->ba_view_slot) if (v.second) (void)hipFree(v.second);
    c->ba_view_slot.clear();
    if (!c->owns_streams) { c->ba_streams.clear(); c->fe_stream = nullptr; c->stream = nullptr; }      // (the device's role streams: synchronised above, not this context's to destroy)
    for (hipStream_t st : c->ba_streams) (void)hipStreamDestroy(st);
    c->ba_streams.clear();
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_stage_pf) (void)hipHostFree(c->h_stage_pf);
    if (c->h_match) (void)hipHostFree(c->h_match);
    for (uint8_t* b : c->h_upload) if (b) (void)hipHostFree(b);
    for (hipEvent_t e : c->ev_upload) if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->ev_copy_mark) (void)hipEventDestroy(c->ev_copy_mark);
    if (c->ev_copy_mark_fe) (void)hipEventDestroy(c->ev_copy_mark_fe);
    for (hipEvent_t e : c->ev_copy_pool) if (e) (void)hipEventDestroy(e);
    for (auto& a : c->host_allocs) if (a.first) (void)hipHostFree(a.first);
    c->host_allocs.clear();
    if (c->fe_stream) { (void)hipStreamSynchronize(c->fe_stream); (void)hipStreamDestroy(c->fe_stream); }
    if (c->fe_done) (void)hipEventDestroy(c->fe_done);
    if (c->debug_stream) { (void)hipStreamSynchronize(c->debug_stream); (void)hipStreamDestroy(c->debug_stream); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (hipStream_t d : c->pad_streams) (void)hipStreamDestroy(d);
    if (c->d_pad) (void)hipFree(c->d_pad);
    delete c;
    if (pool_of) lp_pool_session_release(pool_of, pool_slot_of);
}

// The front end's kernels fill every compute unit's LDS with their workgroups, and a panel-pair workgroup of a running bundle
// adjustment (128 KB of LDS; the band factorisation 130 KB) then finds no room until the kernel drains, stream priority or not: a mapping
// solve beside the front end stands still for the front end's kernels.  With a reserve the extraction kernels leave `cus_per_xcd`
// compute units of every XCD alone -- in SOFTWARE (frontend.hip, FeQueue): persistent workgroups that find themselves on a reserved
// compute unit leave at once, the others take their work from a queue.  (Round 3 did it with CU masks on the context's streams; a
// masked queue in the process makes every solve beside uploads slower, DESIGN.md section 10, so no stream is masked any more.)
// Call it on an idle context.
int lpslam_hip_set_mapping_reserve(lpslam_hip_ctx* c, int32_t cus_per_xcd)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    if (cus_per_xcd < 0 || cus_per_xcd > 16) { set_error("mapping reserve %d out of range (0 .. 16 CUs per XCD)", cus_per_xcd); return LPSLAM_HIP_ERR_INVALID; }
    if (lp_tls_stream) { set_error("set_mapping_reserve inside a prefetch section"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    if (cus_per_xcd == c->reserve_cus) return LPSLAM_HIP_OK;
    LP_HIP(hipStreamSynchronize(c->stream));
    if (c->fe_stream) LP_HIP(hipStreamSynchronize(c->fe_stream));
    if (c->copy_stream) LP_HIP(hipStreamSynchronize(c->copy_stream));
    const int rc = lp_fe_calibrate(c, cus_per_xcd);
    if (rc) return rc;
    c->reserve_cus = cus_per_xcd;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_debug_occupy_unreserved(lpslam_hip_ctx* c, int32_t microseconds, int32_t* landed)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    int n = 0;
    const int rc = lp_fe_occupy_unreserved(c, microseconds, landed ? &n : nullptr);
    if (landed) *landed = n;
    return rc;
}

void* lpslam_hip_stream(lpslam_hip_ctx* c) { return c ? (void*)c->stream : nullptr; }

int lpslam_hip_set_mask(lpslam_hip_ctx* c, int32_t eye, const uint8_t* mask, int32_t stride)
{
    if (!c || eye < 0 || eye > 1) { set_error("invalid mask arguments"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    LP_HIP(hipStreamSynchronize(c->stream));             // no extraction may be reading the old mask,
    if (c->fe_stream) LP_HIP(hipStreamSynchronize(c->fe_stream));      // a prefetched one neither
    if (!mask) {
        if (c->d_mask[eye]) { (void)hipFree(c->d_mask[eye]); c->d_mask[eye] = nullptr; }
        return LPSLAM_HIP_OK;
    }
    if (stride < c->cfg.width) { set_error("mask stride %d smaller than the image width %d", stride, c->cfg.width); return LPSLAM_HIP_ERR_INVALID; }
    if (!c->d_mask[eye]) LP_HIP(hipMalloc((void**)&c->d_mask[eye], (size_t)c->cfg.width * c->cfg.height));
    LP_HIP(hipMemcpy2D(c->d_mask[eye], (size_t)c->cfg.width, mask, (size_t)stride, (size_t)c->cfg.width, (size_t)c->cfg.height, hipMemcpyHostToDevice));
    return LPSLAM_HIP_OK;
}

static int timer_slot(lpslam_hip_ctx* c, int slot)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    if (slot < 0 || slot >= LPSLAM_HIP_MAX_TIMERS) { set_error("timer slot %d out of range", slot); return LPSLAM_HIP_ERR_INVALID; }
    if (!c->ev_begin[slot]) { LP_HIP(hipEventCreate(&c->ev_begin[slot])); LP_HIP(hipEventCreate(&c->ev_end[slot])); }
    return LPSLAM_HIP_OK;
}
int lpslam_hip_timer_begin(lpslam_hip_ctx* c, int slot)
{
    int rc = timer_slot(c, slot); if (rc) return rc;
    LP_HIP(hipEventRecord(c->ev_begin[slot], c->stream));
    return LPSLAM_HIP_OK;
}
int lpslam_hip_timer_end(lpslam_hip_ctx* c, int slot)
{
    int rc = timer_slot(c, slot); if (rc) return rc;
    LP_HIP(hipEventRecord(c->ev_end[slot], c->stream));
    return LPSLAM_HIP_OK;
}
int lpslam_hip_timer_read(lpslam_hip_ctx* c, int slot, float* ms)
{
    int rc = timer_slot(c, slot); if (rc) return rc;
    if (!ms) { set_error("null argument"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipEventSynchronize(c->ev_end[slot]));
    LP_HIP(hipEventElapsedTime(ms, c->ev_begin[slot], c->ev_end[slot]));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_sync(lpslam_hip_ctx* c)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipStreamSynchronize(c->stream));
    if (c->fe_stream) LP_HIP(hipStreamSynchronize(c->fe_stream));
    if (c->copy_stream) LP_HIP(hipStreamSynchronize(c->copy_stream));
    return LPSLAM_HIP_OK;
}

// ---- prefetch: the front end of the next frame on a stream of its own ------------------------------------------------------
int lpslam_hip_prefetch_begin(lpslam_hip_ctx* c)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    if (lp_tls_stream) { set_error("prefetch_begin: this thread is already inside a prefetch section"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    if (!c->fe_stream) {
        LP_HIP(lp_fe_stream_create(&c->fe_stream, true));
        LP_HIP(hipEventCreateWithFlags(&c->fe_done, hipEventDisableTiming));
    }
    lp_tls_stream = c->fe_stream;
    lp_tls_stream_used = false;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_prefetch_end(lpslam_hip_ctx* c)
{
    if (!c || !lp_tls_stream || lp_tls_stream != c->fe_stream) { set_error("prefetch_end without prefetch_begin on this thread"); return LPSLAM_HIP_ERR_INVALID; }
    // (a section whose front end went out as a shared launch enqueued nothing here: the session's streams stay untouched -- an event
    // record and the join's wait are packets on hardware queues that other sessions' chains may be occupying)
    const hipError_t e = lp_tls_stream_used ? hipEventRecord(c->fe_done, c->fe_stream) : hipSuccess;
    if (lp_tls_stream_used) c->fe_join_needed = true;
    lp_tls_stream = nullptr;
    if (e != hipSuccess) { set_error("hipEventRecord failed"); return LPSLAM_HIP_ERR_DEVICE; }
    return LPSLAM_HIP_OK;
}

int lpslam_hip_prefetch_join(lpslam_hip_ctx* c)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    if (lp_tls_stream) { set_error("prefetch_join inside a prefetch section"); return LPSLAM_HIP_ERR_INVALID; }
    if (c->fe_done && c->fe_join_needed) { c->fe_join_needed = false; LP_HIP(hipStreamWaitEvent(c->stream, c->fe_done, 0)); }
    return LPSLAM_HIP_OK;
}

int lpslam_hip_level_info(lpslam_hip_ctx* c, int32_t* widths, int32_t* heights, int32_t* pitches, int32_t* quotas,
                          float* scale_factors)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    for (int l = 0; l < c->lt.n_levels; ++l) {
        if (widths) widths[l] = c->lt.w[l];
        if (heights) heights[l] = c->lt.h[l];
        if (pitches) pitches[l] = c->lt.pitch[l];
        if (quotas) quotas[l] = c->lt.quota[l];
        if (scale_factors) scale_factors[l] = c->lt.scale[l];
    }
    return LPSLAM_HIP_OK;
}

int lpslam_hip_max_keypoints_per_image(lpslam_hip_ctx* c) { return c ? c->slots_per_image : 0; }

static int check_image(lpslam_hip_ctx* c, int image)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    if (image < 0 || image >= c->cfg.max_images) { set_error("image slot %d out of range [0,%d)", image, c->cfg.max_images); return LPSLAM_HIP_ERR_CAPACITY; }
    return LPSLAM_HIP_OK;
}
static int check_batch(lpslam_hip_ctx* c, int n)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    if (n < 1 || n > c->cfg.max_images) { set_error("batch of %d images exceeds the context capacity %d", n, c->cfg.max_images); return LPSLAM_HIP_ERR_CAPACITY; }
    LP_HIP(hipSetDevice(c->cfg.device));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_image_ptr(lpslam_hip_ctx* c, int image, void** dev_ptr, int32_t* pitch)
{
    int rc = check_image(c, image); if (rc) return rc;
    if (dev_ptr) *dev_ptr = c->d_pyr + (size_t)image * c->image_slab;
    if (pitch) *pitch = c->lt.pitch[0];
    return LPSLAM_HIP_OK;
}

// The caller's frame goes through a page-locked buffer of its image slot (one row-compacting memcpy): the copy to the device is then
// a real asynchronous DMA, where a copy from pageable memory is staged by the runtime inside the call (measured: ~0.1 ms of the
// calling thread per 1280x720 stereo frame).  The buffer is reused when the slot is uploaded again; its previous copy is awaited.
static bool ensure_upload_staging(lpslam_hip_ctx* c, int image)
{
    const size_t w = (size_t)c->lt.w[0], h = (size_t)c->lt.h[0];
    if (c->h_upload.size() < (size_t)c->cfg.max_images) { c->h_upload.resize((size_t)c->cfg.max_images, nullptr); c->ev_upload.resize((size_t)c->cfg.max_images, nullptr); }
    uint8_t*& buf = c->h_upload[(size_t)image];
    if (buf) return true;
    if (hipHostMalloc((void**)&buf, w * h, hipHostMallocDefault) != hipSuccess) { buf = nullptr; set_error("page-locked upload staging of %zu bytes failed", w * h); return false; }
    if (hipEventCreateWithFlags(&c->ev_upload[(size_t)image], hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return false; }
    return true;
}
static const uint8_t* stage_upload(lpslam_hip_ctx* c, int image, const uint8_t* host, int32_t stride)
{
    const size_t w = (size_t)c->lt.w[0], h = (size_t)c->lt.h[0];
    const bool fresh = c->h_upload.size() <= (size_t)image || !c->h_upload[(size_t)image];
    if (!ensure_upload_staging(c, image)) return nullptr;
    uint8_t* buf = c->h_upload[(size_t)image];
    if (!fresh && hipEventSynchronize(c->ev_upload[(size_t)image]) != hipSuccess) { set_error("hipEventSynchronize failed"); return nullptr; }
    if ((size_t)stride == w) memcpy(buf, host, w * h);
    else for (size_t r = 0; r < h; ++r) memcpy(buf + r * w, host + r * (size_t)stride, w);
    return buf;
}

}  // extern "C"

// where a context's frame uploads go: its front-end stream, or -- a session on the role streams -- a copy-only stream of its own
static hipStream_t lp_upload_stream(lpslam_hip_ctx* c)
{
    if (c->owns_streams) return lp_fe_stream(c);
    if (!c->up_stream) {
        if (hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->up_stream = nullptr; return lp_fe_stream(c); }
        c->up_pending.assign((size_t)c->cfg.max_images, 0);
    }
    return c->up_stream;
}
// A session's stream for work that is neither latency bound nor part of a shared launch -- the loop-candidate search's brute-force
// matching against stored keyframe descriptors (100-150 us per keyframe), the descriptor uploads of new keyframes.  On the matchers'
// role stream the eight sessions' searches of a keyframe round stand in a row in front of the window matchers; on any other of the
// four queues of a priority they stand in front of that queue's role instead (measured: what the matchers gain the pose optimiser loses).
hipStream_t lp_aux_stream(lpslam_hip_ctx* c)
{
    if (c->owns_streams) return c->stream;
    // 0 main role stream (measured best without priorities: 8 managers 6004 / 5938 frames/s), 1 the session's copy stream (5115 / 5598),
    // 2 the front-end role stream (5184 / 5441), 3 the auxiliary role stream (a queue of its own when the process has one to spare)
    static const int where = [] { const char* e = getenv("LPSLAM_HIP_AUX_STREAM"); return e ? atoi(e) : 3; }();
    if (where == 3) return c->role_aux ? c->role_aux : c->stream;
    return where == 0 ? c->stream : (where == 1 ? lp_upload_stream(c) : c->fe_stream);
}

// stream `s` of the context waits (on the device) for the uploads of slots [first, first + n) that went through the copy-only stream
int lp_wait_own_uploads(lpslam_hip_ctx* c, int first, int n, hipStream_t s)
{
    if (c->up_pending.empty()) return LPSLAM_HIP_OK;
    for (int i = first; i < first + n && (size_t)i < c->up_pending.size(); ++i) {
        if (!c->up_pending[(size_t)i]) continue;
        c->up_pending[(size_t)i] = 0;
        LP_HIP(hipStreamWaitEvent(s, c->ev_upload[(size_t)i], 0));
    }
    return LPSLAM_HIP_OK;
}

extern "C" {

int lpslam_hip_upload_image(lpslam_hip_ctx* c, int image, const uint8_t* host, int32_t stride)
{
    int rc = check_image(c, image); if (rc) return rc;
    if (!host || stride < c->lt.w[0]) { set_error("bad host image (stride %d < width %d)", stride, c->lt.w[0]); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    if (c->cfg.max_images > 8) {
        // a large resident ring (bench, batch callers): the runtime's own staged copy of pageable memory moves more bytes per second
        // than a single-threaded memcpy into page-locked buffers (measured: 2571 against 2177 frames/s with uploads in the loop)
        LP_HIP(hipMemcpy2DAsync(c->d_pyr + (size_t)image * c->image_slab, c->lt.pitch[0], host, stride, c->lt.w[0], c->lt.h[0],
                                hipMemcpyHostToDevice, lp_fe_stream(c)));
        return LPSLAM_HIP_OK;
    }
    const uint8_t* src = stage_upload(c, image, host, stride);
    if (!src) return LPSLAM_HIP_ERR_DEVICE;
    hipStream_t us = lp_upload_stream(c);
    if ((size_t)c->lt.pitch[0] == (size_t)c->lt.w[0]) LP_HIP(hipMemcpyAsync(c->d_pyr + (size_t)image * c->image_slab, src, (size_t)c->lt.w[0] * c->lt.h[0], hipMemcpyHostToDevice, us));
    else LP_HIP(hipMemcpy2DAsync(c->d_pyr + (size_t)image * c->image_slab, c->lt.pitch[0], src, c->lt.w[0], c->lt.w[0], c->lt.h[0], hipMemcpyHostToDevice, us));
    LP_HIP(hipEventRecord(c->ev_upload[(size_t)image], us));
    if (us == c->up_stream && (size_t)image < c->up_pending.size()) c->up_pending[(size_t)image] = 1;
    return LPSLAM_HIP_OK;
}

// ---- page-locked frames and asynchronous uploads ----------------------------------------------------------------------------
// The reference hands the tracker an alias of the caller's 8-bit frame (src/Manager/SlamManager.cpp:1082-1085: zero copy, the
// caller keeps it alive until consumed).  On a discrete GPU the equivalent is a frame in page-locked host memory -- the capture
// layer's DMA target, or the caller's own buffer registered in place -- copied by the DMA engines on a stream of its own while the
// kernels of earlier frames run.
int lpslam_hip_host_alloc(lpslam_hip_ctx* c, size_t bytes, void** out)
{
    if (!c || !out || !bytes) { set_error("invalid host_alloc arguments"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("page-locked allocation of %zu bytes failed", bytes); return LPSLAM_HIP_ERR_DEVICE; }
    { std::lock_guard<std::mutex> lock(c->pool_mutex); c->host_allocs.emplace_back(p, bytes); }
    *out = p;
    return LPSLAM_HIP_OK;
}
int lpslam_hip_host_free(lpslam_hip_ctx* c, void* p)
{
    if (!c || !p) return LPSLAM_HIP_OK;
    LP_HIP(hipSetDevice(c->cfg.device));
    if (c->copy_stream) LP_HIP(hipStreamSynchronize(c->copy_stream));      // no copy may still be reading it
    {
        std::lock_guard<std::mutex> lock(c->pool_mutex);
        auto it = std::find_if(c->host_allocs.begin(), c->host_allocs.end(), [p](const std::pair<void*, size_t>& a) { return a.first == p; });
        if (it == c->host_allocs.end()) { set_error("host_free: not a block of lpslam_hip_host_alloc of this context"); return LPSLAM_HIP_ERR_INVALID; }
        c->host_allocs.erase(it);
    }
    LP_HIP(hipHostFree(p));
    return LPSLAM_HIP_OK;
}
int lpslam_hip_host_register(lpslam_hip_ctx* c, void* p, size_t bytes)
{
    if (!c || !p || !bytes) { set_error("invalid host_register arguments"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); set_error("hipHostRegister of %zu bytes failed", bytes); return LPSLAM_HIP_ERR_DEVICE; }
    return LPSLAM_HIP_OK;
}
int lpslam_hip_host_unregister(lpslam_hip_ctx* c, void* p)
{
    if (!c || !p) return LPSLAM_HIP_OK;
    LP_HIP(hipSetDevice(c->cfg.device));
    if (c->copy_stream) LP_HIP(hipStreamSynchronize(c->copy_stream));
    LP_HIP(hipHostUnregister(p));
    return LPSLAM_HIP_OK;
}

// main stream waits (on the device) for uploads of slots [first, first + n) it has not waited for yet
int lp_wait_uploads(lpslam_hip_ctx* c, int first, int n)
{
    std::lock_guard<std::mutex> lock(c->copy_mutex);
    if (c->slot_copy_event.empty()) return LPSLAM_HIP_OK;
    hipStream_t s = lp_fe_stream(c);
    hipEvent_t last = nullptr;
    for (int i = first; i < first + n; ++i) {
        hipEvent_t e = c->slot_copy_event[(size_t)i];
        if (!e) continue;
        c->slot_copy_event[(size_t)i] = nullptr;
        if (e != last) { LP_HIP(hipStreamWaitEvent(s, e, 0)); last = e; }
    }
    return LPSLAM_HIP_OK;
}

int lpslam_hip_upload_images_async(lpslam_hip_ctx* c, int first, int n, const uint8_t* const* hosts, int32_t stride)
{
    if (!c || !hosts) { set_error("null argument"); return LPSLAM_HIP_ERR_INVALID; }
    if (first < 0 || n < 1 || first + n > c->cfg.max_images) { set_error("image slots [%d,%d) exceed the context capacity %d", first, first + n, c->cfg.max_images); return LPSLAM_HIP_ERR_CAPACITY; }
    if (stride < c->lt.w[0]) { set_error("bad host image (stride %d < width %d)", stride, c->lt.w[0]); return LPSLAM_HIP_ERR_INVALID; }
    for (int i = 0; i < n; ++i) if (!hosts[i]) { set_error("null frame %d", i); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    std::lock_guard<std::mutex> lock(c->copy_mutex);
    if (!c->copy_stream) {
        LP_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        LP_HIP(hipEventCreateWithFlags(&c->ev_copy_mark, hipEventDisableTiming));
        LP_HIP(hipEventCreateWithFlags(&c->ev_copy_mark_fe, hipEventDisableTiming));
        c->ev_copy_pool.assign(16, nullptr);
        for (auto& e : c->ev_copy_pool) LP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->slot_copy_event.assign((size_t)c->cfg.max_images, nullptr);
    }
    // whatever has been enqueued on the context's stream so far may still read these slots: the copies start behind it
    LP_HIP(hipEventRecord(c->ev_copy_mark, c->stream));
    LP_HIP(hipStreamWaitEvent(c->copy_stream, c->ev_copy_mark, 0));
    if (c->fe_stream) {                                 // ... and so may the prefetch stream's (extraction of the next frame)
        LP_HIP(hipEventRecord(c->ev_copy_mark_fe, c->fe_stream));
        LP_HIP(hipStreamWaitEvent(c->copy_stream, c->ev_copy_mark_fe, 0));
    }
    const size_t w = (size_t)c->lt.w[0], h = (size_t)c->lt.h[0], pitch = (size_t)c->lt.pitch[0];
    for (int i = 0; i < n; ++i) {
        uint8_t* dst = c->d_pyr + (size_t)(first + i) * c->image_slab;
        if (pitch == w && (size_t)stride == w) LP_HIP(hipMemcpyAsync(dst, hosts[i], w * h, hipMemcpyHostToDevice, c->copy_stream));
        else LP_HIP(hipMemcpy2DAsync(dst, pitch, hosts[i], (size_t)stride, w, h, hipMemcpyHostToDevice, c->copy_stream));
    }
    // an event of the ring: one still referenced by a slot (an upload nobody extracted) is waited for by the main stream first
    hipEvent_t e = c->ev_copy_pool[c->ev_copy_next % c->ev_copy_pool.size()];
    ++c->ev_copy_next;
    for (size_t i = 0; i < c->slot_copy_event.size(); ++i)
        if (c->slot_copy_event[i] == e) { LP_HIP(hipStreamWaitEvent(c->stream, e, 0)); for (auto& x : c->slot_copy_event) if (x == e) x = nullptr; break; }
    LP_HIP(hipEventRecord(e, c->copy_stream));
    for (int i = 0; i < n; ++i) c->slot_copy_event[(size_t)(first + i)] = e;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_set_rectify_map(lpslam_hip_ctx* c, int32_t eye, const float* map_x, const float* map_y)
{
    if (!c || eye < 0 || eye > 1 || !map_x || !map_y) { set_error("bad rectify map arguments"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    const int w = c->lt.w[0], h = c->lt.h[0];
    const size_t n = (size_t)w * h;
    // cv::remap's own conversion of CV_32FC1 maps: sx = cvRound(x * INTER_TAB_SIZE) in float, round half to even,
    // integer part saturated to short (imgwarp.cpp, remap: "sx = cvRound(sX[x1]*INTER_TAB_SIZE)")
    std::vector<short2> xy(n);
    std::vector<uint16_t> frac(n);
    for (size_t i = 0; i < n; ++i) {
        const float fx = map_x[i] * 32.0f, fy = map_y[i] * 32.0f;
        auto rnd = [](float v) -> int { if (!(v > -2.0e9f)) return INT32_MIN; if (!(v < 2.0e9f)) return INT32_MIN; return (int)nearbyintf(v); };   // cvRound of NaN / overflow: INT_MIN as _mm_cvtss_si32
        const int sx = rnd(fx), sy = rnd(fy);
        auto sat = [](int v) -> short { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); };
        xy[i] = make_short2(sat(sx >> 5), sat(sy >> 5));
        frac[i] = (uint16_t)((sy & 31) * 32 + (sx & 31));
    }
    if (!c->d_map_xy[eye]) { LP_HIP(hipMalloc((void**)&c->d_map_xy[eye], n * sizeof(short2))); LP_HIP(hipMalloc((void**)&c->d_map_frac[eye], n * sizeof(uint16_t))); }
    if (!c->d_raw) LP_HIP(hipMalloc((void**)&c->d_raw, n));
    LP_HIP(hipStreamSynchronize(c->stream));              // a remap using the old map may be in flight,
    if (c->fe_stream) LP_HIP(hipStreamSynchronize(c->fe_stream));      // on the prefetch stream too
    LP_HIP(hipMemcpy(c->d_map_xy[eye], xy.data(), n * sizeof(short2), hipMemcpyHostToDevice));
    LP_HIP(hipMemcpy(c->d_map_frac[eye], frac.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_upload_raw_image(lpslam_hip_ctx* c, int image, int32_t eye, const uint8_t* host, int32_t stride)
{
    int rc = check_image(c, image); if (rc) return rc;
    if (eye < 0 || eye > 1 || !c->d_map_xy[eye]) { set_error("no rectify map set for eye %d", eye); return LPSLAM_HIP_ERR_INVALID; }
    if (!host || stride < c->lt.w[0]) { set_error("bad host image (stride %d < width %d)", stride, c->lt.w[0]); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    // the staging buffer is reused by every upload: copy and remap are ordered on the context stream
    const uint8_t* src = stage_upload(c, image, host, stride);
    if (!src) return LPSLAM_HIP_ERR_DEVICE;
    LP_HIP(hipMemcpy2DAsync(c->d_raw, c->lt.w[0], src, c->lt.w[0], c->lt.w[0], c->lt.h[0], hipMemcpyHostToDevice, lp_fe_stream(c)));
    LP_HIP(hipEventRecord(c->ev_upload[(size_t)image], lp_fe_stream(c)));
    return lp_launch_remap(c, image, eye);
}

int lpslam_hip_remap_staged(lpslam_hip_ctx* c, int image, int32_t eye)
{
    int rc = check_image(c, image); if (rc) return rc;
    if (eye < 0 || eye > 1 || !c->d_map_xy[eye] || !c->d_raw) { set_error("no rectify map / staged frame for eye %d", eye); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    return lp_launch_remap(c, image, eye);
}

static int check_range(lpslam_hip_ctx* c, int first, int n)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    if (first < 0 || n < 1 || first + n > c->cfg.max_images) {
        set_error("image slots [%d,%d) exceed the context capacity %d", first, first + n, c->cfg.max_images); return LPSLAM_HIP_ERR_CAPACITY;
    }
    LP_HIP(hipSetDevice(c->cfg.device));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_stage_pyramid(lpslam_hip_ctx* c, int n) { int rc = check_batch(c, n); if (!rc) rc = lp_wait_uploads(c, 0, n); return rc ? rc : lp_launch_pyramid(c, 0, n); }
int lpslam_hip_stage_fast(lpslam_hip_ctx* c, int n) { int rc = check_batch(c, n); return rc ? rc : lp_launch_fast(c, 0, n); }
int lpslam_hip_stage_distribute(lpslam_hip_ctx* c, int n) { int rc = check_batch(c, n); return rc ? rc : lp_launch_distribute(c, 0, n); }
int lpslam_hip_stage_describe(lpslam_hip_ctx* c, int n) { int rc = check_batch(c, n); return rc ? rc : lp_launch_describe(c, 0, n); }

int lpslam_hip_extract_range(lpslam_hip_ctx* c, int first, int n)
{
    int rc = check_range(c, first, n); if (rc) return rc;
    if ((rc = lp_wait_uploads(c, first, n))) return rc;
    if ((rc = lp_wait_own_uploads(c, first, n, lp_fe_stream(c)))) return rc;
    if ((rc = lp_launch_pyramid(c, first, n))) return rc;
    if ((rc = lp_launch_fast(c, first, n))) return rc;
    if ((rc = lp_launch_distribute(c, first, n))) return rc;
    return lp_launch_describe(c, first, n);
}

int lpslam_hip_extract(lpslam_hip_ctx* c, int n) { return lpslam_hip_extract_range(c, 0, n); }

int lpslam_hip_keypoint_count(lpslam_hip_ctx* c, int image, int32_t* count)
{
    int rc = check_image(c, image); if (rc) return rc;
    if (!count) { set_error("null argument"); return LPSLAM_HIP_ERR_INVALID; }
    if ((size_t)image < c->h_kp_valid.size() && c->h_kp_valid[(size_t)image]) { *count = c->h_kp_count[(size_t)image]; return LPSLAM_HIP_OK; }
    LP_HIP(hipMemcpyAsync(count, c->d_kp_count + image, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    LP_HIP(hipStreamSynchronize(c->stream));
    if ((size_t)image < c->h_kp_valid.size()) { c->h_kp_count[(size_t)image] = *count; c->h_kp_valid[(size_t)image] = 1; }
    return LPSLAM_HIP_OK;
}

int lpslam_hip_get_keypoints(lpslam_hip_ctx* c, int image, lpslam_hip_keypoint* kpts, uint8_t* desc32, int32_t capacity,
                             int32_t* count)
{
    int32_t n = 0;
    int rc = lpslam_hip_keypoint_count(c, image, &n); if (rc) return rc;
    if (count) *count = n;
    if (n > capacity) { set_error("keypoint buffer too small (%d < %d)", capacity, n); return LPSLAM_HIP_ERR_CAPACITY; }
    const size_t o = (size_t)image * c->slots_per_image;
    if (kpts && n) LP_HIP(hipMemcpyAsync(kpts, c->d_kpts + o, (size_t)n * sizeof(lpslam_hip_keypoint), hipMemcpyDeviceToHost, c->stream));
    if (desc32 && n) LP_HIP(hipMemcpyAsync(desc32, c->d_desc + o * 32, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
    LP_HIP(hipStreamSynchronize(c->stream));
    return LPSLAM_HIP_OK;
}

// One frame's results in one round trip: count, keypoints, descriptors and the stereo columns are copied (whole slots, the count
// is not known yet) into a pinned staging block of the context with a single synchronisation, then the first `count` entries go
// to the caller's arrays.  The separate getters cost a synchronisation each and stage pageable memory copy by copy.
}  // extern "C"

unsigned* lp_done_counter(lpslam_hip_ctx* c, int which)
{
    std::lock_guard<std::mutex> lock(c->pool_mutex);       // (a tracking thread and its prefetch thread may both arrive here first)
    if (!c->d_done) {
        if (hipMalloc((void**)&c->d_done, 8 * 32 * sizeof(unsigned)) != hipSuccess) { c->d_done = nullptr; return nullptr; }
        // (the context's streams do not synchronise with the null stream: the zeroes must be there before any of them runs a kernel)
        if (hipMemset(c->d_done, 0, 8 * 32 * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(c->d_done); c->d_done = nullptr; return nullptr; }
    }
    return c->d_done + 32 * (which & 7);
}

// After lp_wait_done reported failure on stream `s` (the flag did not arrive within 20 ms and was not there after the synchronisation
// either).  Two cases.  The stream synchronises: its kernels are complete, so the device block they wrote may be handed back -- but
// arrival counter `which` was left between 0 and total - 1 by whatever went wrong, and every later delivery through it would count
// to the wrong total and never release its flag: it is zeroed here (true: safe to release).  The stream does NOT synchronise (a
// device fault): nothing the kernels touch may be reused -- false, the caller leaks its block.
bool lp_wait_recover(lpslam_hip_ctx* c, int which, hipStream_t s)
{
    if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return false; }
    unsigned* counter = lp_done_counter(c, which);
    if (counter && (hipMemsetAsync(counter, 0, sizeof(unsigned), s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) { (void)hipGetLastError(); return false; }
    return true;
}

namespace {
// One frame's results into page-locked host memory: count, then the first `count` keypoints / descriptors / stereo columns / depths
// as 32-bit words (the device knows the count; the copy engines would have to move all the slots, in five packets).
__device__ __forceinline__ void frame_to_host_body(const int* __restrict__ d_count, const uint32_t* __restrict__ kp, const uint32_t* __restrict__ desc,
                                                   const uint32_t* __restrict__ xr, const uint32_t* __restrict__ dep, uint32_t* __restrict__ st,
                                                   int o_kp, int o_desc, int o_xr, int o_dep, int slots, int n_blocks)
{
    const int n = min(max(*d_count, 0), slots);
    const int w_kp = kp ? 7 * n : 0, w_desc = desc ? 8 * n : 0, w_xr = xr ? n : 0, w_dep = dep ? n : 0;
    const int total = w_kp + w_desc + w_xr + w_dep;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += n_blocks * blockDim.x) {
        if (i < w_kp) st[o_kp + i] = kp[i];
        else if (i < w_kp + w_desc) st[o_desc + i - w_kp] = desc[i - w_kp];
        else if (i < w_kp + w_desc + w_xr) st[o_xr + i - w_kp - w_desc] = xr[i - w_kp - w_desc];
        else st[o_dep + i - w_kp - w_desc - w_xr] = dep[i - w_kp - w_desc - w_xr];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st[0] = (uint32_t)*d_count;
}
__global__ __launch_bounds__(256) void k_frame_to_host(const int* __restrict__ d_count, const uint32_t* __restrict__ kp, const uint32_t* __restrict__ desc,
                                                       const uint32_t* __restrict__ xr, const uint32_t* __restrict__ dep, uint32_t* __restrict__ st,
                                                       int o_kp, int o_desc, int o_xr, int o_dep, int slots, unsigned* counter, int* flag, int seq)
{
    frame_to_host_body(d_count, kp, desc, xr, dep, st, o_kp, o_desc, o_xr, o_dep, slots, (int)gridDim.x);
    lp_signal_done(counter, flag, seq);
}
// the deliveries of several sessions' frames in one launch: blockIdx.y = request, every request releases its own flag
constexpr int kDeliverBatch = 32;
struct DeliverBatch { LpDeliverReq r[kDeliverBatch]; };
__global__ __launch_bounds__(256) void k_frame_to_host_req(DeliverBatch b, int o_kp, int o_desc, int o_xr, int o_dep, int slots)
{
    const LpDeliverReq& r = b.r[blockIdx.y];
    const int n_blocks = r.blocks;
    if ((int)blockIdx.x >= n_blocks) return;
    frame_to_host_body(r.d_count, r.kp, r.desc, r.xr, r.dep, r.st, o_kp, o_desc, o_xr, o_dep, slots, n_blocks);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (n_blocks == 1 || atomicAdd(r.counter, 1u) == (unsigned)n_blocks - 1) {
            if (n_blocks > 1) { *r.counter = 0; __threadfence(); }
            __hip_atomic_store(r.flag, r.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
}  // namespace

extern "C" {

// the page-locked block of a frame's results: count | done flag | keypoints | descriptors | stereo columns | depths
struct FrameStage { size_t o_kp, o_desc, o_xr, o_dep, need; };
static FrameStage frame_stage(const lpslam_hip_ctx* c)
{
    const size_t S = (size_t)c->slots_per_image;
    FrameStage f;
    f.o_kp = 64; f.o_desc = f.o_kp + ((S * sizeof(lpslam_hip_keypoint) + 63) / 64) * 64; f.o_xr = f.o_desc + S * 32; f.o_dep = f.o_xr + ((S * 4 + 63) / 64) * 64;
    f.need = f.o_dep + S * 4;
    return f;
}
enum { FRAME_KPTS = 1, FRAME_DESC = 2, FRAME_XR = 4, FRAME_DEPTH = 8 };

// launches the delivery of image slot `image` into the block `st` on stream `s`; returns the sequence number the kernel will release
static int frame_deliver(lpslam_hip_ctx* c, int image, int fields, uint8_t* st, int counter_id, hipStream_t s, int& seq_counter, int* seq_out)
{
    static_assert(sizeof(lpslam_hip_keypoint) == 28, "k_frame_to_host moves keypoints as seven words");
    const FrameStage f = frame_stage(c);
    const size_t S = (size_t)c->slots_per_image;
    unsigned* counter = lp_done_counter(c, counter_id);
    if (!counter) { set_error("device memory for the completion counters"); return LPSLAM_HIP_ERR_DEVICE; }
    int* flag = (int*)(st + 32);
    const int seq = lp_next_seq(seq_counter);      // (the prefetch thread's deliveries count on their own: two threads never share a counter)
    __atomic_store_n(flag, 0, __ATOMIC_RELAXED);
    const size_t o = (size_t)image * S;
    const float* fs = c->d_stereo + o * 2;
    const int want_words = (int)S * (((fields & FRAME_KPTS) ? 7 : 0) + ((fields & FRAME_DESC) ? 8 : 0) + ((fields & FRAME_XR) ? 1 : 0) + ((fields & FRAME_DEPTH) ? 1 : 0));
    const int blocks = std::max(1, std::min(64, (want_words + 1023) / 1024));
    hipLaunchKernelGGL(k_frame_to_host, dim3(blocks), dim3(256), 0, s, (const int*)(c->d_kp_count + image),
                       (fields & FRAME_KPTS) ? (const uint32_t*)(c->d_kpts + o) : nullptr, (fields & FRAME_DESC) ? (const uint32_t*)(c->d_desc + o * 32) : nullptr,
                       (fields & FRAME_XR) ? (const uint32_t*)fs : nullptr, (fields & FRAME_DEPTH) ? (const uint32_t*)(fs + S) : nullptr, (uint32_t*)st,
                       (int)(f.o_kp / 4), (int)(f.o_desc / 4), (int)(f.o_xr / 4), (int)(f.o_dep / 4), (int)S, counter, flag, seq);
    LP_HIP(hipGetLastError());
    *seq_out = seq;
    return LPSLAM_HIP_OK;
}

// The read-back of a frame whose front end runs ahead (inside a prefetch section, after its extraction and stereo match): the delivery
// kernel is queued behind them on the same stream, so the results are in page-locked memory before the caller asks --
// lpslam_hip_get_frame for that slot then only checks the flag.  Whatever rewrites the slot's results afterwards voids the copy.
int lpslam_hip_prefetch_frame(lpslam_hip_ctx* c, int image, int32_t with_stereo)
{
    int rc = check_image(c, image); if (rc) return rc;
    const FrameStage f = frame_stage(c);
    hipStream_t s = lp_fe_stream(c);
    if (c->h_stage_pf_bytes < f.need) {
        if (c->h_stage_pf) { if (c->pf_stream) LP_HIP(hipStreamSynchronize(c->pf_stream)); (void)hipHostFree(c->h_stage_pf); }
        c->h_stage_pf = nullptr; c->h_stage_pf_bytes = 0; c->pf_image = -1; c->pf_in_flight = false;
        LP_HIP(hipHostMalloc((void**)&c->h_stage_pf, f.need, hipHostMallocDefault));
        c->h_stage_pf_bytes = f.need;
    }
    const int fields = FRAME_KPTS | FRAME_DESC | (with_stereo ? (FRAME_XR | FRAME_DEPTH) : 0);
    int seq = 0;
    c->pf_image = -1;
    // one delivery into the block at a time: a copy that was voided instead of collected may still be on its way (another stream)
    if (c->pf_in_flight && !lp_wait_done((int*)(c->h_stage_pf + 32), c->pf_seq, c->pf_stream)) { c->pf_in_flight = !lp_wait_recover(c, 3, c->pf_stream); set_error("lpslam_hip_prefetch_frame: the previous read-back did not complete"); return LPSLAM_HIP_ERR_DEVICE; }
    c->pf_in_flight = false;
    if ((rc = frame_deliver(c, image, fields, c->h_stage_pf, 3, s, c->pf_seq_next, &seq))) return rc;
    c->pf_seq = seq; c->pf_fields = fields; c->pf_stream = s; c->pf_in_flight = true;
    c->pf_image = image;
    return LPSLAM_HIP_OK;
}

}  // extern "C"

// What lpslam_hip_prefetch_frame sets up, without its launch: the request of a delivery that a shared launch will carry (share.hip).
int lp_prepare_delivery(lpslam_hip_ctx* c, int image, int with_stereo, LpDeliverReq* out)
{
    const FrameStage f = frame_stage(c);
    if (c->h_stage_pf_bytes < f.need) {
        if (c->h_stage_pf) { if (c->pf_stream) LP_HIP(hipStreamSynchronize(c->pf_stream)); (void)hipHostFree(c->h_stage_pf); }
        c->h_stage_pf = nullptr; c->h_stage_pf_bytes = 0; c->pf_image = -1; c->pf_in_flight = false;
        LP_HIP(hipHostMalloc((void**)&c->h_stage_pf, f.need, hipHostMallocDefault));
        c->h_stage_pf_bytes = f.need;
    }
    c->pf_image = -1;
    if (c->pf_in_flight && !lp_wait_done((int*)(c->h_stage_pf + 32), c->pf_seq, c->pf_stream)) { c->pf_in_flight = !lp_wait_recover(c, 3, c->pf_stream); set_error("front end: the previous read-back did not complete"); return LPSLAM_HIP_ERR_DEVICE; }
    c->pf_in_flight = false;
    unsigned* counter = lp_done_counter(c, 3);
    if (!counter) { set_error("device memory for the completion counters"); return LPSLAM_HIP_ERR_DEVICE; }
    const size_t S = (size_t)c->slots_per_image, o = (size_t)image * S;
    const float* fs = c->d_stereo + o * 2;
    const int words = (int)S * (7 + 8 + (with_stereo ? 2 : 0));
    int* flag = (int*)(c->h_stage_pf + 32);
    __atomic_store_n(flag, 0, __ATOMIC_RELAXED);
    *out = LpDeliverReq{(const int*)(c->d_kp_count + image), (const uint32_t*)(c->d_kpts + o), (const uint32_t*)(c->d_desc + o * 32),
                        with_stereo ? (const uint32_t*)fs : nullptr, with_stereo ? (const uint32_t*)(fs + S) : nullptr, (uint32_t*)c->h_stage_pf, counter, flag,
                        lp_next_seq(c->pf_seq_next), std::max(1, std::min(64, (words + 1023) / 1024))};
    return LPSLAM_HIP_OK;
}
void lp_commit_delivery(lpslam_hip_ctx* c, int image, int with_stereo, const LpDeliverReq& r, hipStream_t s)
{
    c->pf_seq = r.seq; c->pf_fields = FRAME_KPTS | FRAME_DESC | (with_stereo ? (FRAME_XR | FRAME_DEPTH) : 0); c->pf_stream = s; c->pf_in_flight = true;
    c->pf_image = image;
}
int lp_launch_deliver_batch(hipStream_t s, const LpDeliverReq* reqs, int n, int slots_per_image, const lpslam_hip_ctx* layout)
{
    const FrameStage f = frame_stage(layout);
    for (int i0 = 0; i0 < n; i0 += kDeliverBatch) {
        const int m = std::min(n - i0, (int)kDeliverBatch);
        DeliverBatch b{};
        int gx = 1;
        for (int i = 0; i < m; ++i) { b.r[i] = reqs[i0 + i]; gx = std::max(gx, reqs[i0 + i].blocks); }
        hipLaunchKernelGGL(k_frame_to_host_req, dim3((unsigned)gx, (unsigned)m), dim3(256), 0, s, b, (int)(f.o_kp / 4), (int)(f.o_desc / 4), (int)(f.o_xr / 4), (int)(f.o_dep / 4), slots_per_image);
        LP_HIP(hipGetLastError());
    }
    return LPSLAM_HIP_OK;
}

extern "C" {

// One frame's front end behind its uploads: extraction of the slot (stereo: the slot pair), stereo match, delivery of the results into the
// context's page-locked block -- what a tracker enqueues per frame (extract_range + match_stereo + prefetch_frame), as ONE call, so that
// the frames several sessions of a pool have pending can go through one launch chain (share.hip).  Asynchronous, like its parts.
int lpslam_hip_front_end(lpslam_hip_ctx* c, int image, int32_t stereo, float fxb, float baseline)
{
    int rc = check_image(c, image); if (rc) return rc;
    if (stereo && (rc = check_image(c, image + 1))) return rc;
    if (stereo && (!(baseline > 0.f) || !(fxb > 0.f))) { set_error("focal_x_baseline and baseline must be positive"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    const int shared = lp_share_front_end(c, image, stereo ? 1 : 0, fxb, baseline);
    if (shared < 0) return -shared;
    if (shared == LP_SHARE_DONE) return LPSLAM_HIP_OK;
    if ((rc = lpslam_hip_extract_range(c, image, stereo ? 2 : 1))) return rc;
    if (stereo && (rc = lpslam_hip_match_stereo(c, image, image + 1, fxb, baseline))) return rc;
    return lpslam_hip_prefetch_frame(c, image, stereo ? 1 : 0);
}

// lpslam_hip_front_end with the frame itself: upload of the slot (stereo: the pair) + front end.
int lpslam_hip_front_end_images(lpslam_hip_ctx* c, int image, const uint8_t* left, const uint8_t* right, int32_t stride, float fxb, float baseline)
{
    int rc = check_image(c, image); if (rc) return rc;
    const bool stereo = right != nullptr;
    if (stereo && (rc = check_image(c, image + 1))) return rc;
    if (!left || stride < c->lt.w[0]) { set_error("bad host image (stride %d < width %d)", stride, c->lt.w[0]); return LPSLAM_HIP_ERR_INVALID; }
    if (stereo && (!(baseline > 0.f) || !(fxb > 0.f))) { set_error("focal_x_baseline and baseline must be positive"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    // the uploads are enqueued at once, by the calling thread, on the context's front-end stream -- for a session that joined a pool that
    // IS the front-end role stream the shared chain will run on: the copies of the sessions' frames proceed while the gather waits for the
    // last session, and the chain's kernels queue behind them
    if ((rc = lpslam_hip_upload_image(c, image, left, stride))) return rc;
    if (stereo && (rc = lpslam_hip_upload_image(c, image + 1, right, stride))) return rc;
    const int shared = lp_share_front_end(c, image, stereo ? 1 : 0, fxb, baseline);
    if (shared < 0) return -shared;
    if (shared == LP_SHARE_DONE) return LPSLAM_HIP_OK;
    if ((rc = lpslam_hip_extract_range(c, image, stereo ? 2 : 1))) return rc;
    if (stereo && (rc = lpslam_hip_match_stereo(c, image, image + 1, fxb, baseline))) return rc;
    return lpslam_hip_prefetch_frame(c, image, stereo ? 1 : 0);
}

// the block that holds `fields` of slot `image` once this returns (delivered ahead of time, or read back now)
static int frame_collect(lpslam_hip_ctx* c, int image, int fields, uint8_t** block)
{
    const FrameStage f = frame_stage(c);
    uint8_t* st = nullptr;
    int rc = 0;
    if (c->pf_image == image && c->h_stage_pf && (fields & ~c->pf_fields) == 0) {
        // delivered ahead of time by lpslam_hip_prefetch_frame
        st = c->h_stage_pf;
        c->pf_image = -1;
        const bool arrived = lp_wait_done((int*)(st + 32), c->pf_seq, c->pf_stream);
        lp_share_front_end_collected(c);
        if (!arrived) { c->pf_in_flight = !lp_wait_recover(c, 3, c->pf_stream); set_error("lpslam_hip_get_frame: the prefetched read-back did not complete"); return LPSLAM_HIP_ERR_DEVICE; }
        c->pf_in_flight = false;
    } else {
        if (c->h_stage_bytes < f.need) {
            if (c->h_stage) { LP_HIP(hipStreamSynchronize(c->stream)); (void)hipHostFree(c->h_stage); }
            c->h_stage = nullptr; c->h_stage_bytes = 0;
            LP_HIP(hipHostMalloc((void**)&c->h_stage, f.need, hipHostMallocDefault));
            c->h_stage_bytes = f.need;
        }
        st = c->h_stage;
        int seq = 0;
        if ((rc = frame_deliver(c, image, fields, st, 0, c->stream, c->done_seq, &seq))) return rc;
        if (!lp_wait_done((int*)(st + 32), seq, c->stream)) { (void)lp_wait_recover(c, 0, c->stream); set_error("lpslam_hip_get_frame: the read-back kernel did not complete"); return LPSLAM_HIP_ERR_DEVICE; }
    }
    int32_t n = 0;
    memcpy(&n, st, sizeof(n));
    if ((size_t)image < c->h_kp_valid.size()) { c->h_kp_count[(size_t)image] = n; c->h_kp_valid[(size_t)image] = 1; }
    *block = st;
    lp_share_frame(c, 1);                                // the session has its frame: its matcher and pose-optimiser requests are about to come (share.hip)
    return LPSLAM_HIP_OK;
}

int lpslam_hip_get_frame(lpslam_hip_ctx* c, int image, lpslam_hip_keypoint* kpts, uint8_t* desc32, float* stereo_x_right, float* depths,
                         int32_t capacity, int32_t* count)
{
    int rc = check_image(c, image); if (rc) return rc;
    const FrameStage f = frame_stage(c);
    const int fields = (kpts ? FRAME_KPTS : 0) | (desc32 ? FRAME_DESC : 0) | (stereo_x_right ? FRAME_XR : 0) | (depths ? FRAME_DEPTH : 0);
    uint8_t* st = nullptr;
    if ((rc = frame_collect(c, image, fields, &st))) return rc;
    int32_t n = 0;
    memcpy(&n, st, sizeof(n));
    if (count) *count = n;
    if (n > capacity) { set_error("frame buffers too small (%d < %d)", capacity, n); return LPSLAM_HIP_ERR_CAPACITY; }
    if (kpts && n) memcpy(kpts, st + f.o_kp, (size_t)n * sizeof(lpslam_hip_keypoint));
    if (desc32 && n) memcpy(desc32, st + f.o_desc, (size_t)n * 32);
    if (stereo_x_right && n) memcpy(stereo_x_right, st + f.o_xr, (size_t)n * sizeof(float));
    if (depths && n) memcpy(depths, st + f.o_dep, (size_t)n * sizeof(float));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_get_frame_view(lpslam_hip_ctx* c, int image, int32_t with_stereo, const lpslam_hip_keypoint** kpts, const uint8_t** desc32,
                              const float** stereo_x_right, const float** depths, int32_t* count)
{
    int rc = check_image(c, image); if (rc) return rc;
    if (!kpts || !desc32 || !count || (with_stereo && (!stereo_x_right || !depths))) { set_error("null argument"); return LPSLAM_HIP_ERR_INVALID; }
    const FrameStage f = frame_stage(c);
    uint8_t* st = nullptr;
    if ((rc = frame_collect(c, image, FRAME_KPTS | FRAME_DESC | (with_stereo ? (FRAME_XR | FRAME_DEPTH) : 0), &st))) return rc;
    int32_t n = 0;
    memcpy(&n, st, sizeof(n));
    *count = n;
    *kpts = (const lpslam_hip_keypoint*)(st + f.o_kp); *desc32 = st + f.o_desc;
    if (stereo_x_right) *stereo_x_right = with_stereo ? (const float*)(st + f.o_xr) : nullptr;
    if (depths) *depths = with_stereo ? (const float*)(st + f.o_dep) : nullptr;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_get_pyramid_level(lpslam_hip_ctx* c, int image, int level, uint8_t* out, int32_t out_stride)
{
    int rc = check_image(c, image); if (rc) return rc;
    if (level < 0 || level >= c->lt.n_levels || !out || out_stride < c->lt.w[level]) { set_error("bad level/stride"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipMemcpy2DAsync(out, out_stride, c->d_pyr + (size_t)image * c->image_slab + c->lt.off[level], c->lt.pitch[level],
                            c->lt.w[level], c->lt.h[level], hipMemcpyDeviceToHost, c->stream));
    LP_HIP(hipStreamSynchronize(c->stream));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_get_candidates(lpslam_hip_ctx* c, int image, int level, lpslam_hip_corner* out, int32_t capacity, int32_t* count)
{
    int rc = check_image(c, image); if (rc) return rc;
    if (level < 0 || level >= c->lt.n_levels) { set_error("bad level"); return LPSLAM_HIP_ERR_INVALID; }
    int32_t n = 0;
    LP_HIP(hipMemcpyAsync(&n, c->d_cand_count + image * c->lt.n_levels + level, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    LP_HIP(hipStreamSynchronize(c->stream));
    if (count) *count = n;
    if (!out) return LPSLAM_HIP_OK;
    if (n > capacity) { set_error("candidate buffer too small (%d < %d)", capacity, n); return LPSLAM_HIP_ERR_CAPACITY; }
    std::vector<uint32_t> keys((size_t)n);
    if (n) LP_HIP(hipMemcpy(keys.data(), c->d_cand_key + (size_t)image * c->cand_per_image + c->lt.cand_start[level],
                            (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) { out[i].x = keys[i] & 0xFFF; out[i].y = (keys[i] >> 12) & 0xFFF; out[i].score = keys[i] >> 24; }
    return LPSLAM_HIP_OK;
}

int lpslam_hip_keypoint_buffers(lpslam_hip_ctx* c, int image, void** kpts_dev, void** desc_dev, void** count_dev)
{
    int rc = check_image(c, image); if (rc) return rc;
    const size_t o = (size_t)image * c->slots_per_image;
    if (kpts_dev) *kpts_dev = c->d_kpts + o;
    if (desc_dev) *desc_dev = c->d_desc + o * 32;
    if (count_dev) { *count_dev = c->d_kp_count + image; if ((size_t)image < c->h_kp_valid.size()) c->h_kp_valid[(size_t)image] = 0; }
    lp_pf_invalidate(c, image, 1);
    return LPSLAM_HIP_OK;
}

int lpslam_hip_set_descriptors(lpslam_hip_ctx* c, int image, const uint8_t* desc32, int32_t n)
{
    int rc = check_image(c, image); if (rc) return rc;
    if (n < 0 || n > c->slots_per_image || (n && !desc32)) { set_error("descriptor count %d out of range [0,%d]", n, c->slots_per_image); return LPSLAM_HIP_ERR_CAPACITY; }
    LP_HIP(hipSetDevice(c->cfg.device));
    if (n) LP_HIP(hipMemcpyAsync(c->d_desc + (size_t)image * c->slots_per_image * 32, desc32, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
    LP_HIP(hipMemcpyAsync(c->d_kp_count + image, &n, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    LP_HIP(hipStreamSynchronize(c->stream));
    if ((size_t)image < c->h_kp_valid.size()) { c->h_kp_count[(size_t)image] = n; c->h_kp_valid[(size_t)image] = 1; }
    lp_pf_invalidate(c, image, 1);
    return LPSLAM_HIP_OK;
}

}  // extern "C"
