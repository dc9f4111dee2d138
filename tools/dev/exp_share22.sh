cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_match_gpu.py tests/test_host_gpu.py tests/test_track_gpu.py -x -q 2>&1 | tail -3
for i in 1 2 3; do LPSLAM_DEV_FLAT=1 LPSLAM_DEV_STATS=1 timeout -k 10 200 python tools/dev_tracker_multi.py 1,8,16 120 2>&1 | grep "managers:\|mean over" | grep -v "^   per-manager statistics (mean over 1 managers): ms_per_frame 0.6" | cut -c1-420; done
