// Multi-GPU test driver of lpslam_hip_ba_optimize_partitioned: one rank per device in one process (ncclCommInitAll, one host
// thread per rank), landmarks dealt round robin.  Reads the problem from a flat binary file written by tests/test_dist_gpu.py,
// writes every rank's poses and rank 0's iteration log.  usage: partitioned_rccl <problem.bin> <result.bin> <max_ranks> <iters>
//   problem.bin : int32 n_poses, n_points, n_obs, robust | double cam[7] | double poses[n_poses*7] | uint8 fixed[n_poses] (padded to 8)
//                 | double points[n_points*3] | lpslam_hip_ba_obs obs[n_obs]
//   result.bin  : int32 ranks, done | lpslam_hip_ba_iter_log log[done] | double poses[ranks][n_poses*7]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../../include/lpslam_hip.h"

int main(int argc, char** argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s problem.bin result.bin max_ranks iters\n", argv[0]); return 2; }
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
    const int ranks = std::min(ndev, atoi(argv[3])), iters = atoi(argv[4]);
    std::vector<ncclComm_t> comms(ranks);
    std::vector<int> devs(ranks);
    for (int i = 0; i < ranks; ++i) devs[i] = i;
    if (ncclCommInitAll(comms.data(), ranks, devs.data()) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 4; }
    std::vector<std::vector<double>> out_poses(ranks, std::vector<double>((size_t)n_poses * 7));
    std::vector<lpslam_hip_ba_iter_log> log((size_t)std::max(iters, 1));
    std::vector<int> done(ranks, 0), status(ranks, 0);
    auto worker = [&](int rank) {
        lpslam_hip_frontend_config cfg{};
        cfg.width = 640; cfg.height = 480; cfg.max_keypoints = 500; cfg.scale_factor = 1.2f; cfg.num_levels = 4; cfg.ini_fast_threshold = 20;
        cfg.min_fast_threshold = 7; cfg.max_images = 1; cfg.device = devs[rank];
        lpslam_hip_ctx* ctx = nullptr;
        if (lpslam_hip_create(&cfg, &ctx)) { status[rank] = 10; return; }
        // this rank's landmarks: j % ranks == rank, renumbered; all poses
        std::vector<int> remap((size_t)n_points, -1);
        std::vector<double> my_points;
        for (int j = rank; j < n_points; j += ranks) { remap[j] = (int)(my_points.size() / 3); for (int k = 0; k < 3; ++k) my_points.push_back(points[3 * (size_t)j + k]); }
        std::vector<lpslam_hip_ba_obs> my_obs;
        for (const auto& o : obs) if (remap[o.point] >= 0) { lpslam_hip_ba_obs m = o; m.point = remap[o.point]; my_obs.push_back(m); }
        if (my_points.empty()) my_points.resize(3, 0.0);
        lpslam_hip_ba* ba = nullptr;
        if (lpslam_hip_ba_create(ctx, poses.data(), fixed.data(), n_poses, my_points.data(), (int)(my_points.size() / 3), my_obs.data(), (int)my_obs.size(), &cam, &ba)) {
            fprintf(stderr, "rank %d: %s\n", rank, lpslam_hip_last_error()); status[rank] = 11; return;
        }
        std::vector<lpslam_hip_ba_iter_log> my_log((size_t)std::max(iters, 1));
        if (lpslam_hip_ba_optimize_partitioned(ba, comms[rank], robust, iters, my_log.data(), &done[rank])) {
            fprintf(stderr, "rank %d: %s\n", rank, lpslam_hip_last_error()); status[rank] = 12; return;
        }
        if (rank == 0) log = my_log;
        if (lpslam_hip_ba_get(ba, out_poses[rank].data(), nullptr)) status[rank] = 13;
        lpslam_hip_ba_destroy(ba);
        lpslam_hip_destroy(ctx);
    };
    std::vector<std::thread> th;
    for (int r = 0; r < ranks; ++r) th.emplace_back(worker, r);
    for (auto& t : th) t.join();
    for (int r = 0; r < ranks; ++r) if (status[r]) { fprintf(stderr, "rank %d failed with %d\n", r, status[r]); return 5; }
    for (int r = 1; r < ranks; ++r)
        if (done[r] != done[0] || memcmp(out_poses[r].data(), out_poses[0].data(), out_poses[0].size() * 8) != 0) { fprintf(stderr, "rank %d disagrees with rank 0\n", r); return 6; }
    for (auto& c : comms) ncclCommDestroy(c);
    FILE* g = fopen(argv[2], "wb");
    if (!g) { perror("result file"); return 2; }
    int32_t oh[2] = {ranks, done[0]};
    fwrite(oh, 4, 2, g);
    fwrite(log.data(), sizeof(lpslam_hip_ba_iter_log), (size_t)done[0], g);
    for (int r = 0; r < ranks; ++r) fwrite(out_poses[r].data(), 8, out_poses[r].size(), g);
    fclose(g);
    printf("partitioned_rccl: %d rank(s), %d iterations, chi2 %.6f\n", ranks, done[0], done[0] ? log[done[0] - 1].chi2_after : 0.0);
    return 0;
}
