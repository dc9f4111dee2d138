// Test driver of lpslam_hip_ba_optimize_partitioned_with: R ranks (host threads, one context and one BA stream each) on ONE device,
// landmarks dealt round robin, with an in-process all-reduce behind the callback.  A one-rank RCCL communicator reduces nothing
// (every all-reduce is the identity), so the C++ driver's launch order, its in-place reduced buffers and its device-side lambda
// control are only exercised for real with R >= 2; this binary does that on a one-GPU box.
//
// The all-reduce is stream ordered like RCCL's (no host synchronisation of the data path): every rank records an event behind
// the work already on its stream, the host threads meet, every rank makes its stream wait for the peers' events and sums all R
// buffers IN RANK ORDER into a private scratch (bit-identical on every rank), records a second event, the threads meet again,
// every stream waits until all peers have read its buffer and copies the scratch over it.
//
// usage: partitioned_shared_device <problem.bin> <result.bin> <ranks> <iters>
//   problem.bin : int32 n_poses, n_points, n_obs, robust | double cam[7] | double poses[n_poses*7] | uint8 fixed[n_poses] (padded to 8)
//                 | double points[n_points*3] | lpslam_hip_ba_obs obs[n_obs]            (as tests/cpp/partitioned_rccl_main.cpp)
//   result.bin  : int32 ranks, done, calls | per rank: lpslam_hip_ba_iter_log log[done] | double poses[n_poses*7]
//                 | then double points[n_points*3] gathered from the owning ranks
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/lpslam_hip.h"

namespace {

constexpr int MAX_RANKS = 8;
struct Ptrs { const double* p[MAX_RANKS]; };

__global__ void k_reduce(double* out, Ptrs in, int ranks, size_t count, int op)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double acc = in.p[0][i];
    for (int r = 1; r < ranks; ++r) { const double x = in.p[r][i]; acc = op == LPSLAM_HIP_REDUCE_MAX ? (x > acc ? x : acc) : acc + x; }
    out[i] = acc;
}

struct Reducer {
    int ranks = 0;
    std::mutex m; std::condition_variable cv;
    int arrived = 0; long generation = 0; bool broken = false;
    double* buf[MAX_RANKS] = {}; size_t count[MAX_RANKS] = {}; int op[MAX_RANKS] = {};
    hipEvent_t ev_in[MAX_RANKS] = {}, ev_out[MAX_RANKS] = {};
    double* scratch[MAX_RANKS] = {}; size_t scratch_cap[MAX_RANKS] = {};
    long calls[MAX_RANKS] = {};

    // all ranks or nobody: a rank that never arrives (the ranks took different decisions) breaks the meeting after 30 s
    bool meet()
    {
        std::unique_lock<std::mutex> lock(m);
        if (broken) return false;
        const long g = generation;
        if (++arrived == ranks) { arrived = 0; ++generation; cv.notify_all(); return true; }
        if (!cv.wait_for(lock, std::chrono::seconds(30), [&] { return generation != g || broken; })) { broken = true; cv.notify_all(); return false; }
        return !broken;
    }
    void abandon() { std::lock_guard<std::mutex> lock(m); broken = true; cv.notify_all(); }
};
struct Handle { Reducer* R; int rank; };

int allreduce_cb(void* user, void* vbuf, size_t count, int32_t op, void* vstream)
{
    Handle* h = (Handle*)user; Reducer& R = *h->R; const int r = h->rank;
    hipStream_t s = (hipStream_t)vstream;
    R.buf[r] = (double*)vbuf; R.count[r] = count; R.op[r] = op; ++R.calls[r];
    if (R.scratch_cap[r] < count) {
        if (R.scratch[r]) { (void)hipStreamSynchronize(s); (void)hipFree(R.scratch[r]); }
        if (hipMalloc((void**)&R.scratch[r], count * sizeof(double) * 2) != hipSuccess) { R.abandon(); return 2; }
        R.scratch_cap[r] = 2 * count;
    }
    if (hipEventRecord(R.ev_in[r], s) != hipSuccess) { R.abandon(); return 3; }
    if (!R.meet()) return 4;
    Ptrs in{};
    for (int p = 0; p < R.ranks; ++p) {
        if (R.count[p] != count || R.op[p] != op) { fprintf(stderr, "rank %d: all-reduce %ld mismatched (count %zu/%zu op %d/%d)\n", r, R.calls[r], count, R.count[p], op, R.op[p]); R.abandon(); return 5; }
        in.p[p] = R.buf[p];
        if (p != r && hipStreamWaitEvent(s, R.ev_in[p], 0) != hipSuccess) { R.abandon(); return 6; }
    }
    hipLaunchKernelGGL(k_reduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, R.scratch[r], in, R.ranks, count, (int)op);
    if (hipGetLastError() != hipSuccess || hipEventRecord(R.ev_out[r], s) != hipSuccess) { R.abandon(); return 7; }
    if (!R.meet()) return 8;
    for (int p = 0; p < R.ranks; ++p) if (p != r && hipStreamWaitEvent(s, R.ev_out[p], 0) != hipSuccess) { R.abandon(); return 9; }
    if (hipMemcpyAsync(vbuf, R.scratch[r], count * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) { R.abandon(); return 10; }
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s problem.bin result.bin ranks iters\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("problem file"); return 2; }
    int32_t hdr[4];
    if (fread(hdr, 4, 4, f) != 4) return 2;
    const int n_poses = hdr[0], n_points = hdr[1], n_obs = hdr[2], robust = hdr[3];
    lpslam_hip_ba_camera cam;
    std::vector<double> poses((size_t)n_poses * 7), points((size_t)n_points * 3);
    std::vector<uint8_t> fixed(((size_t)n_poses + 7) / 8 * 8);
    std::vector<lpslam_hip_ba_obs> obs((size_t)n_obs);
    if (fread(&cam, sizeof(cam), 1, f) != 1 || fread(poses.data(), 8, poses.size(), f) != poses.size() || fread(fixed.data(), 1, fixed.size(), f) != fixed.size() ||
        fread(points.data(), 8, points.size(), f) != points.size() || fread(obs.data(), sizeof(lpslam_hip_ba_obs), obs.size(), f) != obs.size()) { fprintf(stderr, "short problem file\n"); return 2; }
    fclose(f);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { fprintf(stderr, "no HIP device\n"); return 3; }
    const int ranks = atoi(argv[3]), iters = atoi(argv[4]);
    if (ranks < 1 || ranks > MAX_RANKS) { fprintf(stderr, "ranks must be in [1,%d]\n", MAX_RANKS); return 2; }
    Reducer R; R.ranks = ranks;
    for (int r = 0; r < ranks; ++r)
        if (hipEventCreateWithFlags(&R.ev_in[r], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&R.ev_out[r], hipEventDisableTiming) != hipSuccess) return 3;
    std::vector<std::vector<double>> out_poses(ranks, std::vector<double>((size_t)n_poses * 7));
    std::vector<double> out_points((size_t)n_points * 3, 0.0);
    std::vector<std::vector<lpslam_hip_ba_iter_log>> logs(ranks, std::vector<lpslam_hip_ba_iter_log>((size_t)std::max(iters, 1)));
    std::vector<int> done(ranks, 0), status(ranks, 0);
    auto worker = [&](int rank) {
        lpslam_hip_frontend_config cfg{};
        cfg.width = 640; cfg.height = 480; cfg.max_keypoints = 500; cfg.scale_factor = 1.2f; cfg.num_levels = 4; cfg.ini_fast_threshold = 20;
        cfg.min_fast_threshold = 7; cfg.max_images = 1; cfg.device = 0;
        lpslam_hip_ctx* ctx = nullptr;
        if (lpslam_hip_create(&cfg, &ctx)) { status[rank] = 10; R.abandon(); return; }
        std::vector<int> remap((size_t)n_points, -1), mine;
        std::vector<double> my_points;
        for (int j = rank; j < n_points; j += ranks) { remap[j] = (int)mine.size(); mine.push_back(j); for (int k = 0; k < 3; ++k) my_points.push_back(points[3 * (size_t)j + k]); }
        std::vector<lpslam_hip_ba_obs> my_obs;
        for (const auto& o : obs) if (remap[o.point] >= 0) { lpslam_hip_ba_obs m = o; m.point = remap[o.point]; my_obs.push_back(m); }
        const int my_n = (int)mine.size();
        if (my_points.empty()) my_points.resize(3, 0.0);
        lpslam_hip_ba* ba = nullptr;
        if (lpslam_hip_ba_create(ctx, poses.data(), fixed.data(), n_poses, my_points.data(), std::max(my_n, 1), my_obs.data(), (int)my_obs.size(), &cam, &ba)) {
            fprintf(stderr, "rank %d: %s\n", rank, lpslam_hip_last_error()); status[rank] = 11; R.abandon(); return;
        }
        Handle h{&R, rank};
        if (lpslam_hip_ba_optimize_partitioned_with(ba, allreduce_cb, &h, robust, iters, logs[rank].data(), &done[rank])) {
            fprintf(stderr, "rank %d: %s\n", rank, lpslam_hip_last_error()); status[rank] = 12; R.abandon();
        } else {
            std::vector<double> px(std::max((size_t)my_n, (size_t)1) * 3);
            if (lpslam_hip_ba_get(ba, out_poses[rank].data(), px.data())) status[rank] = 13;
            for (int i = 0; i < my_n; ++i) for (int k = 0; k < 3; ++k) out_points[3 * (size_t)mine[i] + k] = px[3 * (size_t)i + k];
        }
        lpslam_hip_ba_destroy(ba);
        lpslam_hip_destroy(ctx);
    };
    std::vector<std::thread> th;
    for (int r = 0; r < ranks; ++r) th.emplace_back(worker, r);
    for (auto& t : th) t.join();
    for (int r = 0; r < ranks; ++r) if (status[r]) { fprintf(stderr, "rank %d failed with %d\n", r, status[r]); return 5; }
    for (int r = 1; r < ranks; ++r) if (R.calls[r] != R.calls[0]) { fprintf(stderr, "rank %d made %ld all-reduce calls, rank 0 %ld\n", r, R.calls[r], R.calls[0]); return 6; }
    FILE* g = fopen(argv[2], "wb");
    if (!g) { perror("result file"); return 2; }
    int32_t oh[3] = {ranks, done[0], (int32_t)R.calls[0]};
    for (int r = 1; r < ranks; ++r) if (done[r] != done[0]) { fprintf(stderr, "rank %d ran %d iterations, rank 0 %d\n", r, done[r], done[0]); return 6; }
    fwrite(oh, 4, 3, g);
    for (int r = 0; r < ranks; ++r) {
        fwrite(logs[r].data(), sizeof(lpslam_hip_ba_iter_log), (size_t)done[0], g);
        fwrite(out_poses[r].data(), 8, out_poses[r].size(), g);
    }
    fwrite(out_points.data(), 8, out_points.size(), g);
    fclose(g);
    printf("partitioned_shared_device: %d rank(s), %d iterations, %ld all-reduces, chi2 %.6f\n", ranks, done[0], R.calls[0], done[0] ? logs[0][done[0] - 1].chi2_after : 0.0);
    return 0;
}
