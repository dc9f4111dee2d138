cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" LPSLAM_DEV_FLAT=1 LPSLAM_DEV_STATS=1 timeout -k 10 200 python tools/dev_tracker_multi.py ${N:-8} 200 2>&1 | grep "managers:\|mean over [0-9]* managers" | tail -2 | cut -c1-420; }
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false'
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false' GPU_MAX_HW_QUEUES=8
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false' GPU_MAX_HW_QUEUES=16
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false' GPU_MAX_HW_QUEUES=32
run GPU_MAX_HW_QUEUES=16
run GPU_MAX_HW_QUEUES=32
