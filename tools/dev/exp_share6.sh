cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_host_gpu.py -x -q -m gpu -k shared_launches 2>&1 | grep "^E  \|passed\|failed" | head -20
run() { echo "== $*"; env "$@" LPSLAM_DEV_FLAT=1 LPSLAM_DEV_STATS=1 timeout -k 10 200 python tools/dev_tracker_multi.py ${N:-8} 200 2>&1 | grep "managers:\|mean over [0-9]* managers" | tail -2 | cut -c1-420; }
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false'
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false' LPSLAM_HIP_SHARE_FE_WINDOW_US=500
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false' LPSLAM_HIP_SHARE_NO_PROBE=1
run A=1
