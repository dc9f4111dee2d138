// share.hip -- launches shared by the sessions (contexts) of a process on one device (gfx950 only).  See internal.h for the why.
//
// No thread of its own: a caller publishes its request, and whichever caller finds the combiner free gathers what is pending and
// launches it (flat combining).  The gather ends when every session that is tracking and not already waiting for a launched batch has a
// request in, when no request has arrived for `quiet` microseconds, or `window` microseconds after the oldest request came -- callers
// that run in lockstep (they were released by the same launch) arrive within a few microseconds of each other.
#include "internal.h"
#include <algorithm>
#include <condition_variable>

using namespace lpslam;

namespace {

using Clock = std::chrono::steady_clock;
inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now().time_since_epoch()).count(); }

enum { T_PENDING = 0, T_LAUNCHED = 1, T_FAILED = 2 };
struct Ticket {                                         // lives in the caller's frame until its request is done
    std::atomic<int> state{T_PENDING};
    hipStream_t stream = nullptr;                       // where the batch went (T_LAUNCHED)
    std::atomic<int>* table_users = nullptr;            // the table block of a matcher batch: released when the request is done
    std::atomic<int>* owner_done = nullptr;             // the request's entry of its batch (InFlight): set when the owner has seen it complete
};
// The batch of a kind that is on its stream: done when every request's flag holds its sequence number -- looked at by whoever wants
// to launch the next batch, not left to the requests' owners to report (an owner that the scheduler has taken off its core for a
// while would keep the stream idle for everybody; it only marks its entry before it re-uses the flag for its next request).
struct InFlight {
    int n = 0;
    int* flag[64]; int seq[64]; std::atomic<int> owner_done[64];
    bool busy() const
    {
        for (int i = 0; i < n; ++i) if (!owner_done[i].load(std::memory_order_acquire) && __atomic_load_n(flag[i], __ATOMIC_ACQUIRE) != seq[i]) return true;
        return false;
    }
};
struct PendingPose { LpPoseReq req; Ticket* t; int* flag; };
struct PendingProj { LpProjReq req; Ticket* t; };      // (flag and sequence number are in the request)
struct PendingSolve { lpslam_hip_ba* b; int first, second; uint8_t* outlier; double* poses; double* points; std::atomic<int>* state; int* rc; };
struct PendingFront { lpslam_hip_ctx* c; int slot, stereo; float fxb, baseline; hipStream_t own; LpDeliverReq deliver; Ticket* t; };      // own: the stream the session enqueued its uploads on

constexpr int kMaxDevices = 16, kMaxSessions = 256, kTableBlocks = 32, kTableEntries = 64;

struct Share {
    std::mutex m;                                       // the pending lists
    std::mutex combiner;                                // held by the caller that gathers and launches
    std::vector<PendingPose> pose;
    std::vector<PendingProj> proj;
    int64_t pose_oldest_ns = 0, pose_newest_ns = 0, proj_oldest_ns = 0, proj_newest_ns = 0;      // arrival of the oldest / newest pending request of a kind
    InFlight pose_fl[2], proj_fl[2];                    // the batch on each stream (written under the combiner lock); two generations, alternating: an owner's late mark lands in the one that is not current
    int pose_gen = 0, proj_gen = 0;
    std::atomic<int> in_flight{0};                      // requests launched whose callers have not seen them complete
    std::atomic<int64_t> last_ns[kMaxSessions];         // per session: its last request (0: free entry)
    std::atomic<int> n_sessions{0};                     // high-water mark of the table
    std::atomic<int> in_frame[kMaxSessions];            // per session: between the collection of a frame and lpslam_hip_frame_done -- it WILL make the frame's latency-bound requests
    bool ready = false, broken = false;
    // The streams of the shared launches, one per ROLE, each on a hardware queue of its own (lp_share_streams): the pose optimiser's
    // batches, the matchers' batches, the front-end chains (the pool context's stream), the windows' solves.
    hipStream_t s_pose = nullptr, s_proj = nullptr, s_front = nullptr, s_solve = nullptr, s_aux = nullptr;      // (s_aux: a fifth queue when there is one, else the matchers')
    int distinct_queues = 0;                            // how many of the four roles got a hardware queue to themselves (diagnostic)
    LpProjReq* table = nullptr;                         // page-locked: kTableBlocks blocks of kTableEntries requests
    std::atomic<int> table_users[kTableBlocks];         // requests of the block's last batch that are not done yet
    unsigned table_next = 0;
    // front ends: a combiner of their own (a chain of ten launches takes the calling thread ~50 us: the matchers' gather does not wait for it)
    std::mutex combiner_fe;
    std::vector<PendingFront> front;
    int64_t fe_oldest_ns = 0, fe_newest_ns = 0;
    std::atomic<int> fe_in_flight{0};                   // sessions whose shared front end has been launched and not collected
    hipEvent_t ev_fe_chain = nullptr; bool fe_chain_in_flight = false;      // behind the last chain on the front-end stream (combiner_fe held)
    std::atomic<int64_t> fe_last_ns[kMaxSessions];      // per session: its last front-end request
    std::atomic<long> batches{0}, requests{0};          // statistics (lpslam_hip_shared_launch_counters)
    std::atomic<long> fe_batches{0}, fe_requests{0};
    // windows: solved by one of the submitting mapping threads at a time (the flow has host steps between its launch chains)
    std::mutex solver;
    std::vector<PendingSolve> solves;
    int64_t solve_oldest_ns = 0, solve_newest_ns = 0;
    std::atomic<int> solve_in_flight{0};
    std::atomic<int64_t> solve_last_ns[kMaxSessions];
    std::atomic<long> solve_batches{0}, solve_requests{0};
    Share() { for (auto& x : in_frame) x.store(0); for (auto& x : last_ns) x.store(0); for (auto& x : fe_last_ns) x.store(0); for (auto& x : solve_last_ns) x.store(0); for (auto& x : table_users) x.store(0); }
};
Share g_share[kMaxDevices];

std::atomic<int> g_mode{-1};                            // -1: the environment decides (default: automatic)
int share_mode()
{
    const int v = g_mode.load(std::memory_order_relaxed);
    if (v >= 0) return v;
    static const int env = [] { const char* e = getenv("LPSLAM_HIP_SHARED_LAUNCHES"); return e ? std::min(std::max(atoi(e), 0), 2) : 2; }();
    return env;                                         // 0: never, 1: always (a lone session too: tests), 2: when two or more sessions are tracking
}
int env_us(const char* name, int dflt) { const char* e = getenv(name); return e ? std::max(atoi(e), 0) : dflt; }
int64_t window_ns() { static const int64_t v = 1000ll * env_us("LPSLAM_HIP_SHARE_WINDOW_US", 30); return v; }
int64_t pose_quiet_ns() { static const int64_t v = 1000ll * env_us("LPSLAM_HIP_SHARE_POSE_QUIET_US", 2); return v; }      // (a pose batch runs ~110 us: a request that just misses one waits that long)
int64_t frame_window_ns() { static const int64_t v = 1000ll * env_us("LPSLAM_HIP_SHARE_FRAME_WINDOW_US", 25); return v; }      // how long a request waits for a session that is inside its frame and has not asked yet
int64_t quiet_ns() { static const int64_t v = 1000ll * env_us("LPSLAM_HIP_SHARE_QUIET_US", 2); return v; }
int64_t fe_window_ns() { static const int64_t v = 1000ll * env_us("LPSLAM_HIP_SHARE_FE_WINDOW_US", 300); return v; }
int64_t fe_quiet_ns() { static const int64_t v = 1000ll * env_us("LPSLAM_HIP_SHARE_FE_QUIET_US", 40); return v; }
int64_t active_ns() { static const int64_t v = 1000ll * env_us("LPSLAM_HIP_SHARE_ACTIVE_US", 3000); return v; }

// ---- one hardware queue per role ---------------------------------------------------------------------------------------------
// A HIP process has four hardware queues per priority; a new stream is bound to one of them by the runtime, and streams that share a
// queue run IN ORDER: a 25 us matcher launch behind another stream's 0.5 ms extraction chain waits for all of it (measured: 8 sessions,
// every shared matcher launch took 0.3-0.4 ms; with 16 queues 0.16 ms, but then the hardware scheduler time-slices the queues and
// everything else slows down -- GPU_MAX_HW_QUEUES=32: 7x slower).  So the shared launches use FOUR streams, one per role, chosen from a
// set of candidates by MEASUREMENT: a spinning kernel on one candidate, a stamp kernel on another -- if the stamp comes before the
// spin ends, the two do not share a queue.  No assumption about the runtime's binding policy.
__global__ void k_probe_spin(unsigned long long ticks, unsigned long long* out)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) out[0] = wall_clock64();
}
__global__ void k_probe_stamp(unsigned long long* out) { if (threadIdx.x == 0) out[0] = wall_clock64(); }

// true: a launch on `b` does not wait for a kernel running on `a`
bool probe_independent(hipStream_t a, hipStream_t b, unsigned long long* d_stamps, unsigned long long* h_stamps)
{
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, a, 15000ull /* 150 us of the 100 MHz clock */, d_stamps);
    hipLaunchKernelGGL(k_probe_stamp, dim3(1), dim3(64), 0, b, d_stamps + 1);
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipMemcpy(h_stamps, d_stamps, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return false; }
    return h_stamps[1] + 2000ull < h_stamps[0];          // stamped at least 20 us before the spin ended
}

bool share_trace() { static const bool on = getenv("LPSLAM_HIP_SHARE_TRACE") != nullptr; return on; }      // development: one stderr line per shared launch

bool share_init(Share& sh)                              // sh.m held
{
    if (sh.ready) return true;
    if (sh.broken) return false;
    bool ok = hipHostMalloc((void**)&sh.table, sizeof(LpProjReq) * kTableBlocks * kTableEntries, hipHostMallocDefault) == hipSuccess;
    constexpr int kCand = 12;
    hipStream_t cand[kCand] = {};
    int n_cand = 0;
    // LPSLAM_HIP_SHARE_PRIO=1 (measurements): the two latency-bound roles (pose optimiser, matchers) on high-priority streams -- a
    // process has four hardware queues PER PRIORITY, so they cannot share a queue with the chains of the front end, the solves or a
    // session's loop-candidate search, and the dispatcher serves them first
    static const int prio_mode = [] { const char* e = getenv("LPSLAM_HIP_SHARE_PRIO"); return e ? atoi(e) : 0; }();
    int prio_least = 0, prio_greatest = 0;
    if (prio_mode && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { (void)hipGetLastError(); prio_greatest = 0; }
    const int n_high = (prio_mode && prio_greatest < 0) ? 5 : 0;
    for (; ok && n_cand < kCand; ++n_cand)
        if (hipStreamCreateWithPriority(&cand[n_cand], hipStreamNonBlocking, n_cand < n_high ? prio_greatest : 0) != hipSuccess) { (void)hipGetLastError(); break; }
    ok = ok && n_cand >= 5 && n_cand > n_high + 2;
    unsigned long long* d_stamps = nullptr;
    unsigned long long h_stamps[2] = {0, 0};
    ok = ok && hipMalloc((void**)&d_stamps, 2 * sizeof(unsigned long long)) == hipSuccess;
    constexpr int kRoles = 5;                              // pose, matchers, front end, solves, auxiliary (a session's loop-candidate search)
    int picked[kRoles] = {-1, -1, -1, -1, -1}, n_picked = 0;
    static const bool no_probe = getenv("LPSLAM_HIP_SHARE_NO_PROBE") != nullptr;      // measurements: the first candidates as they come
    auto independent = [&](int i) {
        for (int k = 0; k < n_picked; ++k)
            if (!(probe_independent(cand[picked[k]], cand[i], d_stamps, h_stamps) && probe_independent(cand[i], cand[picked[k]], d_stamps, h_stamps))) return false;
        return true;
    };
    // roles in order; with priorities the first two come from the high-priority candidates, the others from the rest
    for (int role = 0; ok && !no_probe && role < kRoles; ++role) {
        const int lo = (n_high && role >= 2) ? n_high : 0, hi = (n_high && role < 2) ? n_high : n_cand;
        for (int i = lo; i < hi; ++i) {
            if (std::find(picked, picked + n_picked, i) != picked + n_picked) continue;
            if (independent(i)) { picked[n_picked++] = i; break; }
        }
        if (n_picked != role + 1) break;
    }
    sh.distinct_queues = n_picked;
    // fewer independent candidates than roles (four queues per priority: the fifth role shares; GPU_MAX_HW_QUEUES < 4; the probe off):
    // roles share, the latency-critical ones last
    for (int next = 0; n_picked < kRoles && ok;) {
        const int role = n_picked;
        if (role == 4) { picked[n_picked++] = picked[1]; break; }      // no queue of its own: the auxiliary work stays with the matchers (measured best)
        const int lo = (n_high && role >= 2) ? n_high : 0;
        next = std::max(next, lo);
        while (std::find(picked, picked + n_picked, next) != picked + n_picked) ++next;
        picked[n_picked++] = next < n_cand ? next : 0;
    }
    if (d_stamps) (void)hipFree(d_stamps);
    if (!ok) { for (int i = 0; i < n_cand; ++i) (void)hipStreamDestroy(cand[i]); (void)hipGetLastError(); sh.broken = true; return false; }
    sh.s_pose = cand[picked[0]]; sh.s_proj = cand[picked[1]]; sh.s_front = cand[picked[2]]; sh.s_solve = cand[picked[3]]; sh.s_aux = cand[picked[4]];
    for (int i = 0; i < n_cand; ++i) if (std::find(picked, picked + kRoles, i) == picked + kRoles) (void)hipStreamDestroy(cand[i]);
    if (share_trace()) fprintf(stderr, "share: role streams from candidates %d %d %d %d %d (%d independent, %d high-priority candidates)\n", picked[0], picked[1], picked[2], picked[3], picked[4], sh.distinct_queues, n_high);
    sh.ready = true;
    return true;
}

// sessions that made a request within the last few milliseconds (this one included)
int active_sessions(Share& sh, int64_t now)
{
    int n = 0;
    const int hi = sh.n_sessions.load(std::memory_order_relaxed);
    for (int i = 0; i < hi; ++i) { const int64_t t = sh.last_ns[i].load(std::memory_order_relaxed); if (t && now - t < active_ns()) ++n; }
    return n;
}

// this context's entry in the session table, stamped with `now`; false: table full (the call stays unshared)
bool touch_session(Share& sh, lpslam_hip_ctx* c, int64_t now)
{
    if (c->share_slot < 0) {
        std::lock_guard<std::mutex> lock(sh.m);
        int slot = -1;
        const int hi = sh.n_sessions.load();
        for (int i = 0; i < hi && slot < 0; ++i) if (sh.last_ns[i].load() == 0) slot = i;
        if (slot < 0) { if (hi >= kMaxSessions) return false; slot = hi; sh.n_sessions.store(hi + 1); }
        c->share_slot = slot;
    }
    sh.last_ns[c->share_slot].store(now, std::memory_order_relaxed);
    return true;
}

// With the combiner held.  A kind's launches go to ONE stream, so a batch would only queue behind the one that is running: while a
// kind's batch is in flight its requests are left to accumulate, and the batch that follows carries all of them -- sessions that were
// served by one launch come back together (this is what puts sessions in step; a gather window alone never did: kernels finishing one
// after the other on a stream keep their callers apart by exactly a kernel's duration).  On an idle stream the gather ends when every
// session that can submit has, when nothing new arrived for `quiet`, or `window` after the oldest request.  Returns when the caller's
// own request has been launched (or failed).
void combine(Share& sh, Ticket& mine)
{
    for (;;) {
        std::vector<PendingPose> pose;
        std::vector<PendingProj> proj;
        const int64_t now = now_ns();
        {
            std::lock_guard<std::mutex> lock(sh.m);
            if (sh.pose.empty() && sh.proj.empty()) return;
            // Who is still to come?  Sessions that say where they are in their frame (lpslam_hip_get_frame_view ... lpslam_hip_frame_done) are
            // waited for while they are inside it and have no request in flight -- up to `frame_window`; without such hints: every session
            // that made a request lately, and then only for `quiet` (a session in its keyframe work or waiting for its frame is not coming).
            int framed = 0;
            { const int hi = sh.n_sessions.load(std::memory_order_relaxed); for (int i = 0; i < hi; ++i) framed += sh.in_frame[i].load(std::memory_order_relaxed); }
            const int expected = std::max(1, (framed ? framed : active_sessions(sh, now)) - sh.in_flight.load(std::memory_order_relaxed));
            const int64_t pose_q = framed ? frame_window_ns() : pose_quiet_ns(), proj_q = framed ? frame_window_ns() : quiet_ns();
            const bool pose_free = !sh.pose_fl[sh.pose_gen].busy();
            if (!sh.pose.empty() && pose_free &&
                ((int)sh.pose.size() >= expected || now - sh.pose_newest_ns >= pose_q || now - sh.pose_oldest_ns >= window_ns())) pose.swap(sh.pose);
            if (!sh.proj.empty() && !sh.proj_fl[sh.proj_gen].busy() &&
                ((int)sh.proj.size() >= expected || now - sh.proj_newest_ns >= proj_q || now - sh.proj_oldest_ns >= window_ns() || (int)sh.proj.size() >= kTableEntries)) proj.swap(sh.proj);
        }
        if (pose.empty() && proj.empty()) {
            if (mine.state.load(std::memory_order_acquire) != T_PENDING) return;
            __builtin_ia32_pause();
            continue;
        }
        bool ok;
        { std::lock_guard<std::mutex> lock(sh.m); ok = share_init(sh); }
        if (!pose.empty()) {
            hipStream_t s = sh.s_pose;
            sh.pose_gen ^= 1;
            InFlight& fl = sh.pose_fl[sh.pose_gen];
            if (pose.size() > 64) { std::lock_guard<std::mutex> lock(sh.m); sh.pose.insert(sh.pose.begin(), pose.begin() + 64, pose.end()); pose.resize(64); }      // (more than 64 sessions: the rest with the next batch)
            std::vector<LpPoseReq> reqs(pose.size());
            for (size_t i = 0; i < pose.size(); ++i) reqs[i] = pose[i].req;
            fl.n = 0;
            for (size_t i = 0; i < pose.size(); ++i) { fl.flag[i] = pose[i].flag; fl.seq[i] = pose[i].req.seq; fl.owner_done[i].store(0, std::memory_order_relaxed); }
            const bool launched = ok && lp_launch_pose_batch(s, reqs.data(), (int)reqs.size()) == LPSLAM_HIP_OK;
            if (share_trace()) fprintf(stderr, "share %.3f pose %d\n", 1e-6 * (double)(now_ns() % 100000000000ll), (int)pose.size());
            if (launched) { fl.n = (int)pose.size(); sh.in_flight.fetch_add((int)pose.size()); }
            for (size_t i = 0; i < pose.size(); ++i) { auto& p = pose[i]; p.t->stream = s; p.t->owner_done = launched ? &fl.owner_done[i] : nullptr; p.t->state.store(launched ? T_LAUNCHED : T_FAILED, std::memory_order_release); }
            sh.requests.fetch_add((long)pose.size()); sh.batches.fetch_add(1);
        }
        if (!proj.empty()) {
            // a table block nobody reads any more (its last batch's requests are all done)
            int blk = -1;
            for (int k = 0; ok && k < kTableBlocks && blk < 0; ++k) { const int b = (int)((sh.table_next + k) % kTableBlocks); if (sh.table_users[b].load(std::memory_order_acquire) == 0) blk = b; }
            bool launched = false;
            hipStream_t s = sh.s_proj;
            if (proj.size() > (size_t)kTableEntries) { std::lock_guard<std::mutex> lock(sh.m); sh.proj.insert(sh.proj.begin(), proj.begin() + kTableEntries, proj.end()); proj.resize((size_t)kTableEntries); }
            sh.proj_gen ^= 1;
            InFlight& fl = sh.proj_fl[sh.proj_gen];
            fl.n = 0;
            for (size_t i = 0; i < proj.size(); ++i) { fl.flag[i] = proj[i].req.done_flag; fl.seq[i] = proj[i].req.done_seq; fl.owner_done[i].store(0, std::memory_order_relaxed); }
            if (blk >= 0) {
                sh.table_next = (unsigned)blk + 1;
                LpProjReq* tab = sh.table + (size_t)blk * kTableEntries;
                int gx = 1;
                for (size_t i = 0; i < proj.size(); ++i) { tab[i] = proj[i].req; gx = std::max(gx, proj[i].req.grid_x); }
                sh.table_users[blk].store((int)proj.size(), std::memory_order_release);
                launched = lp_launch_proj_batch(s, tab, (int)proj.size(), gx) == LPSLAM_HIP_OK;
                if (share_trace()) fprintf(stderr, "share %.3f proj %d\n", 1e-6 * (double)(now_ns() % 100000000000ll), (int)proj.size());
                if (!launched) sh.table_users[blk].store(0, std::memory_order_release);
            }
            if (launched) { fl.n = (int)proj.size(); sh.in_flight.fetch_add((int)proj.size()); }
            for (size_t i = 0; i < proj.size(); ++i) {
                auto& p = proj[i];
                p.t->stream = s; p.t->table_users = launched ? &sh.table_users[blk] : nullptr; p.t->owner_done = launched ? &fl.owner_done[i] : nullptr;
                p.t->state.store(launched ? T_LAUNCHED : T_FAILED, std::memory_order_release);
            }
            sh.requests.fetch_add((long)proj.size()); sh.batches.fetch_add(1);
        }
        if (mine.state.load(std::memory_order_acquire) != T_PENDING) return;
    }
}

// the caller's side after publishing: become the combiner while the request is pending, then poll the request's own flag
int wait_request(Share& sh, Ticket& t, int* flag, int seq, const char* what)
{
    const auto t0 = Clock::now();
    int rc = LP_SHARE_DONE;
    int64_t t_launched = 0;
    for (int spin = 0; ; ++spin) {
        const int st = t.state.load(std::memory_order_acquire);
        if (share_trace() && st == T_LAUNCHED && !t_launched) t_launched = now_ns();
        if (st == T_PENDING) {
            if (sh.combiner.try_lock()) {
                if (t.state.load(std::memory_order_acquire) == T_PENDING) combine(sh, t);
                sh.combiner.unlock();
                continue;
            }
        } else if (st == T_FAILED) {
            set_error("%s: the shared launch failed", what);
            return -LPSLAM_HIP_ERR_DEVICE;
        } else if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
        lp_poll_pause(spin);
        if ((spin & 1023) == 1023 && st == T_LAUNCHED && Clock::now() - t0 > std::chrono::milliseconds(50)) {
            if (hipStreamSynchronize(t.stream) != hipSuccess || __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) { (void)hipGetLastError(); set_error("%s: the shared launch did not complete", what); rc = -LPSLAM_HIP_ERR_DEVICE; }
            break;
        }
    }
    if (share_trace()) {
        const int64_t t_end = now_ns(), t_beg = std::chrono::duration_cast<std::chrono::nanoseconds>(t0.time_since_epoch()).count();
        fprintf(stderr, "req %s: to launch %.0f us, launch to done %.0f us\n", what, 1e-3 * (double)((t_launched ? t_launched : t_end) - t_beg), 1e-3 * (double)(t_end - (t_launched ? t_launched : t_end)));
    }
    sh.in_flight.fetch_sub(1);
    if (t.owner_done) t.owner_done->store(1, std::memory_order_release);
    if (t.table_users) t.table_users->fetch_sub(1, std::memory_order_release);
    return rc;
}

// does sharing apply to a call of this context right now?
Share* share_for(lpslam_hip_ctx* c)
{
    const int mode = share_mode();
    if (mode == 0 || !c || c->cfg.device < 0 || c->cfg.device >= kMaxDevices) return nullptr;
    Share& sh = g_share[c->cfg.device];
    if (sh.broken) return nullptr;
    const int64_t now = now_ns();
    if (!touch_session(sh, c, now)) return nullptr;
    if (mode == 2 && active_sessions(sh, now) < 2) return nullptr;
    // a request runs on the combiner's stream: whatever the context's own stream still holds (a caller that enqueued an extraction and
    // did not wait for it) would not be ordered in front of it -- such a call stays on its stream
    // (a session of a pool has the matchers' role stream as its main stream: ordered by construction)
    if (c->owns_streams && hipStreamQuery(c->stream) != hipSuccess) { (void)hipGetLastError(); if (hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; } }
    return &sh;
}

int active_fe_sessions(Share& sh, int64_t now)
{
    int n = 0;
    const int hi = sh.n_sessions.load(std::memory_order_relaxed);
    for (int i = 0; i < hi; ++i) { const int64_t t = sh.fe_last_ns[i].load(std::memory_order_relaxed); if (t && now - t < active_ns()) ++n; }
    return n;
}

// with combiner_fe held: gather the pending front ends, one launch chain per pool (and chunk of 64 images) on the pool's stream
void combine_front(Share& sh)
{
    std::vector<PendingFront> reqs;
    int64_t t_oldest = 0, t_taken = 0;
    // (the calling thread may be inside ITS session's prefetch section: these launches go to the front-end role's stream)
    { std::lock_guard<std::mutex> lock(sh.m); if (!share_init(sh)) { std::vector<PendingFront> all; all.swap(sh.front); for (auto& r : all) r.t->state.store(T_FAILED, std::memory_order_release); return; } }
    struct TlsGuard { hipStream_t keep; bool used; explicit TlsGuard(hipStream_t s) : keep(lp_tls_stream), used(lp_tls_stream_used) { lp_tls_stream = s; } ~TlsGuard() { lp_tls_stream = keep; lp_tls_stream_used = used; } } tls_guard(sh.s_front);      // the pool's launchers enqueue on lp_fe_stream()
    for (;;) {
        const int64_t now = now_ns();
        {
            std::lock_guard<std::mutex> lock(sh.m);
            const int np = (int)sh.front.size();
            if (np == 0) return;
            // A chain would only queue behind the one that is running on the front-end stream: while that one is in flight the requests
            // accumulate, and the next chain carries all of them.  On an idle stream: every session that is submitting front ends and has
            // none in flight (one whose frames came out of the last chain with this caller's is about to arrive; one in flight cannot
            // before its chain is through), or nothing new for `quiet`, or `window` after the oldest request.
            bool busy = false;
            if (sh.fe_chain_in_flight) { busy = hipEventQuery(sh.ev_fe_chain) == hipErrorNotReady; if (!busy) sh.fe_chain_in_flight = false; (void)hipGetLastError(); }
            const int expected = std::max(1, active_fe_sessions(sh, now) - sh.fe_in_flight.load(std::memory_order_relaxed));
            if ((!busy && (np >= expected || now - sh.fe_newest_ns >= fe_quiet_ns() || now - sh.fe_oldest_ns >= fe_window_ns())) || 2 * np >= kMaxListed) {
                reqs.swap(sh.front);
                t_oldest = sh.fe_oldest_ns; t_taken = now;
                sh.fe_oldest_ns = sh.fe_newest_ns = 0;
                break;
            }
        }
        __builtin_ia32_pause();
    }
    // requests of one pool with the same stereo parameters go together (a process usually has one pool and one camera model)
    std::vector<char> taken(reqs.size(), 0);
    for (size_t i0 = 0; i0 < reqs.size(); ++i0) {
        if (taken[i0]) continue;
        lpslam_hip_ctx* pool = reqs[i0].c->sess_pool;
        std::vector<size_t> grp;
        int images = 0;
        for (size_t i = i0; i < reqs.size(); ++i) {
            const PendingFront& r = reqs[i];
            if (taken[i] || r.c->sess_pool != pool || r.stereo != reqs[i0].stereo || r.fxb != reqs[i0].fxb || r.baseline != reqs[i0].baseline) continue;
            if (images + (r.stereo ? 2 : 1) > kMaxListed) break;
            grp.push_back(i); taken[i] = 1; images += r.stereo ? 2 : 1;
        }
        uint16_t list[kMaxListed];
        std::vector<LpDeliverReq> deliver;
        int n = 0;
        hipStream_t s = sh.s_front;
        bool ok = true;
        for (size_t gi : grp) {
            const PendingFront& r = reqs[gi];
            list[n++] = (uint16_t)(r.c->pool_first + r.slot);
            if (r.stereo) list[n++] = (uint16_t)(r.c->pool_first + r.slot + 1);
            deliver.push_back(r.deliver);
            // the session's uploads are in the slots before the chain reads them
            if (r.own == nullptr) ok = ok && lp_wait_own_uploads(r.c, r.slot, r.stereo ? 2 : 1, s) == LPSLAM_HIP_OK;      // (its copy-only stream: the events of the copies)
            else if (r.own != s) ok = ok && hipStreamWaitEvent(s, r.c->ev_fe_ready, 0) == hipSuccess;
        }
        const int64_t t_copied = now_ns();
        ok = ok && lp_launch_pyramid(pool, 0, n, list) == LPSLAM_HIP_OK && lp_launch_fast(pool, 0, n, list) == LPSLAM_HIP_OK &&
             lp_launch_distribute(pool, 0, n, list) == LPSLAM_HIP_OK && lp_launch_describe(pool, 0, n, list) == LPSLAM_HIP_OK;
        if (ok && reqs[i0].stereo) ok = lp_launch_stereo_strided(pool, 0, 0, 0, n / 2, reqs[i0].fxb, reqs[i0].baseline, list) == LPSLAM_HIP_OK;
        ok = ok && lp_launch_deliver_batch(s, deliver.data(), (int)deliver.size(), pool->slots_per_image, pool) == LPSLAM_HIP_OK;
        if (ok) {
            if (!sh.ev_fe_chain && hipEventCreateWithFlags(&sh.ev_fe_chain, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); sh.ev_fe_chain = nullptr; }
            if (sh.ev_fe_chain && hipEventRecord(sh.ev_fe_chain, s) == hipSuccess) sh.fe_chain_in_flight = true;
        }
        if (!ok) (void)hipGetLastError();
        if (share_trace()) fprintf(stderr, "share %.3f front %d (gathered for %.0f us, launched in %.0f us, of which waits %.0f)\n", 1e-6 * (double)(now_ns() % 100000000000ll), (int)grp.size(), 1e-3 * (double)(t_taken - t_oldest), 1e-3 * (double)(now_ns() - t_taken), 1e-3 * (double)(t_copied - t_taken));
        for (size_t gi : grp) {
            PendingFront& r = reqs[gi];
            r.t->stream = s;
            if (ok) { sh.fe_in_flight.fetch_add(1); r.c->share_fe_pending.store(1); }
            r.t->state.store(ok ? T_LAUNCHED : T_FAILED, std::memory_order_release);
        }
        sh.fe_requests.fetch_add((long)grp.size()); sh.fe_batches.fetch_add(1);
    }
}

}  // namespace

int lp_share_ba_local(lpslam_hip_ctx* c, lpslam_hip_ba* b, int first_iters, int second_iters, uint8_t* outlier, double* poses, double* points)
{
    if (share_mode() == 0 || !c || c->cfg.device < 0 || c->cfg.device >= kMaxDevices) return LP_SHARE_DIRECT;
    Share& sh = g_share[c->cfg.device];
    int64_t now = now_ns();
    if (!touch_session(sh, c, now)) return LP_SHARE_DIRECT;
    sh.solve_last_ns[c->share_slot].store(now, std::memory_order_relaxed);
    std::atomic<int> state{T_PENDING};
    int rc = LPSLAM_HIP_OK;
    {
        std::lock_guard<std::mutex> lock(sh.m);
        if (sh.solves.empty()) sh.solve_oldest_ns = now;
        sh.solve_newest_ns = now;
        sh.solves.push_back(PendingSolve{b, first_iters, second_iters, outlier, poses, points, &state, &rc});
    }
    static const int64_t quiet = 1000ll * env_us("LPSLAM_HIP_SHARE_SOLVE_QUIET_US", 40), window = 1000ll * env_us("LPSLAM_HIP_SHARE_SOLVE_WINDOW_US", 200), active_for = 30000000ll;
    for (;;) {
        if (state.load(std::memory_order_acquire) != T_PENDING) break;
        if (sh.solver.try_lock()) {
            // the solver: whatever is pending when the previous batch is through (it ran under this lock) goes together; on an idle stream a
            // short gather for the other sessions' mapping threads (keyframes of sessions in step come together)
            std::vector<PendingSolve> take;
            while (state.load(std::memory_order_acquire) == T_PENDING && take.empty()) {
                now = now_ns();
                std::lock_guard<std::mutex> lock(sh.m);
                int active = 0;
                const int hi = sh.n_sessions.load(std::memory_order_relaxed);
                for (int i = 0; i < hi; ++i) { const int64_t t = sh.solve_last_ns[i].load(std::memory_order_relaxed); if (t && now - t < active_for) ++active; }
                if ((int)sh.solves.size() >= std::max(1, active) || now - sh.solve_newest_ns >= quiet || now - sh.solve_oldest_ns >= window) {
                    // (windows with the same iteration counts: every tracker's are)
                    for (size_t i = 0; i < sh.solves.size();) {
                        if (sh.solves[i].first == first_iters && sh.solves[i].second == second_iters && take.size() < 64) { take.push_back(sh.solves[i]); sh.solves.erase(sh.solves.begin() + (long)i); }
                        else ++i;
                    }
                    if (!sh.solves.empty()) sh.solve_oldest_ns = sh.solve_newest_ns = now;
                }
            }
            if (!take.empty()) {
                std::vector<lpslam_hip_ba*> ps; std::vector<uint8_t*> outs; std::vector<double*> po, pt;
                for (auto& t : take) { ps.push_back(t.b); outs.push_back(t.outlier); po.push_back(t.poses); pt.push_back(t.points); }
                const int brc = lp_ba_local_batch(ps.data(), (int)ps.size(), first_iters, second_iters, outs.data(), po.data(), pt.data());
                if (share_trace()) fprintf(stderr, "share %.3f solve %d (rc %d)\n", 1e-6 * (double)(now_ns() % 100000000000ll), (int)take.size(), brc);
                sh.solve_batches.fetch_add(1); sh.solve_requests.fetch_add((long)take.size());
                for (auto& t : take) { *t.rc = brc; t.state->store(brc == LPSLAM_HIP_OK ? T_LAUNCHED : T_FAILED, std::memory_order_release); }
            }
            sh.solver.unlock();
            continue;
        }
        const struct timespec ts{0, 20000};
        (void)nanosleep(&ts, nullptr);                    // a mapping thread is in no hurry: it sleeps while another one solves its window
    }
    return rc == LPSLAM_HIP_OK ? LP_SHARE_DONE : -rc;
}

bool lp_share_role_streams(int device, hipStream_t out[5])
{
    if (device < 0 || device >= kMaxDevices) return false;
    Share& sh = g_share[device];
    std::lock_guard<std::mutex> lock(sh.m);
    if (!share_init(sh)) return false;
    out[LP_ROLE_POSE] = sh.s_pose; out[LP_ROLE_MAIN] = sh.s_proj; out[LP_ROLE_FRONT] = sh.s_front; out[LP_ROLE_SOLVE] = sh.s_solve; out[LP_ROLE_AUX] = sh.s_aux;
    return true;
}

int lp_share_front_end(lpslam_hip_ctx* c, int slot, int stereo, float fxb, float baseline)
{
    // a session of a pool, no masks (the launch takes the pool's), no copy-stream uploads pending, sharing wanted, somebody to share with
    const int mode = share_mode();
    if (mode == 0 || !c || !c->sess_pool || c->d_mask[0] || c->d_mask[1] || !c->slot_copy_event.empty() || c->cfg.device < 0 || c->cfg.device >= kMaxDevices) return LP_SHARE_DIRECT;
    Share& sh = g_share[c->cfg.device];
    const int64_t now = now_ns();
    if (!touch_session(sh, c, now)) return LP_SHARE_DIRECT;
    sh.fe_last_ns[c->share_slot].store(now, std::memory_order_relaxed);
    if (mode == 2 && active_fe_sessions(sh, now) < 2) return LP_SHARE_DIRECT;
    hipStream_t own = c->up_stream ? nullptr : lp_fe_stream(c);      // nullptr: the uploads went through the session's copy-only stream
    if (!c->ev_fe_ready && hipEventCreateWithFlags(&c->ev_fe_ready, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); c->ev_fe_ready = nullptr; return LP_SHARE_DIRECT; }
    lp_share_front_end_collected(c);                    // (a delivery nobody collected: a prefetch for a frame that did not come next)
    PendingFront r{c, slot, stereo, fxb, baseline, own, LpDeliverReq{}, nullptr};
    int rc = lp_prepare_delivery(c, slot, stereo, &r.deliver);
    if (rc) return -rc;
    { std::lock_guard<std::mutex> lock(sh.m); if (!share_init(sh)) return LP_SHARE_DIRECT; }
    if (own && own != sh.s_front && hipEventRecord(c->ev_fe_ready, own) != hipSuccess) { (void)hipGetLastError(); return LP_SHARE_DIRECT; }
    // the slots' results are about to be rewritten: the session's host mirrors go stale now
    for (int i = slot; i < slot + (stereo ? 2 : 1) && (size_t)i < c->h_kp_valid.size(); ++i) c->h_kp_valid[(size_t)i] = 0;
    Ticket t;
    r.t = &t;
    {
        std::lock_guard<std::mutex> lock(sh.m);
        if (sh.front.empty()) sh.fe_oldest_ns = now;
        sh.fe_newest_ns = now_ns();
        sh.front.push_back(r);
    }
    // asynchronous: wait until SOMEBODY has launched it (this thread, if the combiner is free), not until the device is done
    for (int spin = 0; ; ++spin) {
        const int st = t.state.load(std::memory_order_acquire);
        if (st == T_LAUNCHED) break;
        if (st == T_FAILED) { set_error("front end: the shared launch failed"); return -LPSLAM_HIP_ERR_DEVICE; }
        if (sh.combiner_fe.try_lock()) {
            if (t.state.load(std::memory_order_acquire) == T_PENDING) combine_front(sh);
            sh.combiner_fe.unlock();
            continue;
        }
        lp_poll_pause(spin);
    }
    lp_commit_delivery(c, slot, stereo, r.deliver, t.stream);
    return LP_SHARE_DONE;
}

void lp_share_front_end_collected(lpslam_hip_ctx* c)
{
    if (!c || c->cfg.device < 0 || c->cfg.device >= kMaxDevices) return;
    if (c->share_fe_pending.exchange(0)) g_share[c->cfg.device].fe_in_flight.fetch_sub(1);
}

int lp_share_pose(lpslam_hip_ctx* c, const LpPoseReq& r, int* flag)
{
    Share* sh = share_for(c);
    if (!sh) return LP_SHARE_DIRECT;
    Ticket t;
    {
        std::lock_guard<std::mutex> lock(sh->m);
        const int64_t now = now_ns();
        if (sh->pose.empty()) sh->pose_oldest_ns = now;
        sh->pose_newest_ns = now;
        sh->pose.push_back(PendingPose{r, &t, flag});
    }
    return wait_request(*sh, t, flag, r.seq, "pose optimiser");
}

int lp_share_proj(lpslam_hip_ctx* c, const LpProjReq& r)
{
    Share* sh = share_for(c);
    if (!sh) return LP_SHARE_DIRECT;
    Ticket t;
    {
        std::lock_guard<std::mutex> lock(sh->m);
        const int64_t now = now_ns();
        if (sh->proj.empty()) sh->proj_oldest_ns = now;
        sh->proj_newest_ns = now;
        sh->proj.push_back(PendingProj{r, &t});
    }
    return wait_request(*sh, t, r.done_flag, r.done_seq, "window matcher");
}

void lp_share_frame(lpslam_hip_ctx* c, int inside)
{
    if (!c || !c->sess_pool || share_mode() == 0 || c->cfg.device < 0 || c->cfg.device >= kMaxDevices) return;
    Share& sh = g_share[c->cfg.device];
    if (c->share_slot < 0 && (!inside || !touch_session(sh, c, now_ns()))) return;
    static const bool off = getenv("LPSLAM_HIP_SHARE_NO_FRAME_HINTS") != nullptr;      // measurements
    if (!off) sh.in_frame[c->share_slot].store(inside ? 1 : 0, std::memory_order_relaxed);
}

void lp_share_forget(lpslam_hip_ctx* c)
{
    if (!c || c->share_slot < 0 || c->cfg.device < 0 || c->cfg.device >= kMaxDevices) return;
    lp_share_front_end_collected(c);
    g_share[c->cfg.device].in_frame[c->share_slot].store(0);
    g_share[c->cfg.device].fe_last_ns[c->share_slot].store(0);
    g_share[c->cfg.device].last_ns[c->share_slot].store(0);
    c->share_slot = -1;
}

extern "C" {

int lpslam_hip_frame_done(lpslam_hip_ctx* c)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    lp_share_frame(c, 0);
    return LPSLAM_HIP_OK;
}

int lpslam_hip_set_shared_launches(int32_t mode)
{
    if (mode < -1 || mode > 2) { set_error("shared launches: mode %d (-1 environment, 0 never, 1 always, 2 automatic)", mode); return LPSLAM_HIP_ERR_INVALID; }
    g_mode.store(mode);
    return LPSLAM_HIP_OK;
}

int lpslam_hip_shared_launch_counters(int32_t device, int64_t* batches, int64_t* requests)
{
    if (device < 0 || device >= kMaxDevices) { set_error("device %d out of range", device); return LPSLAM_HIP_ERR_INVALID; }
    if (batches) *batches = g_share[device].batches.load();
    if (requests) *requests = g_share[device].requests.load();
    return LPSLAM_HIP_OK;
}

int lpslam_hip_shared_solve_counters(int32_t device, int64_t* batches, int64_t* requests)
{
    if (device < 0 || device >= kMaxDevices) { set_error("device %d out of range", device); return LPSLAM_HIP_ERR_INVALID; }
    if (batches) *batches = g_share[device].solve_batches.load();
    if (requests) *requests = g_share[device].solve_requests.load();
    return LPSLAM_HIP_OK;
}

int lpslam_hip_shared_front_end_counters(int32_t device, int64_t* batches, int64_t* requests)
{
    if (device < 0 || device >= kMaxDevices) { set_error("device %d out of range", device); return LPSLAM_HIP_ERR_INVALID; }
    if (batches) *batches = g_share[device].fe_batches.load();
    if (requests) *requests = g_share[device].fe_requests.load();
    return LPSLAM_HIP_OK;
}

}  // extern "C"
