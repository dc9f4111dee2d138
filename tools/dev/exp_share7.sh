cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_share_gpu.py tests/test_host_gpu.py tests/test_track_gpu.py -x -q -m gpu 2>&1 | grep "^E  \|passed\|failed" | head -20
run() { echo "== $*"; env "$@" LPSLAM_DEV_FLAT=1 LPSLAM_DEV_STATS=1 timeout -k 10 200 python tools/dev_tracker_multi.py ${N:-8} 200 2>&1 | grep "managers:\|mean over [0-9]* managers" | tail -2 | cut -c1-420; }
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false'
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false, "loopClosure": false'
run A=1
N=1,2,4,16 run A=1
