cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_track_gpu.py tests/test_frontend_gpu.py tests/test_host_gpu.py tests/test_share_gpu.py -x -q -m gpu 2>&1 | grep "^E  \|passed\|failed" | head -20
run() { echo "== $*"; env "$@" LPSLAM_DEV_FLAT=1 timeout -k 10 200 python tools/dev_tracker_multi.py ${N:-8,16} ${F:-200} 2>&1 | grep "managers:" | tail -3; }
run A=1
run LPSLAM_HIP_NO_QUEUE_SPREAD=1
run A=2
run LPSLAM_HIP_NO_QUEUE_SPREAD=1
