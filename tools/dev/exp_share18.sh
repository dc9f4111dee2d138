cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_share_gpu.py tests/test_host_gpu.py tests/test_track_gpu.py tests/test_match_gpu.py -x -q -m gpu 2>&1 | tail -2
run() { echo "== $*"; env "$@" LPSLAM_DEV_FLAT=1 LPSLAM_DEV_STATS=1 timeout -k 10 200 python tools/dev_tracker_multi.py ${N:-1,8,16} ${F:-200} 2>&1 | grep "managers:\|mean over [0-9]* managers" | tail -6 | cut -c1-330; }
run A=1
run A=2
