cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" LPSLAM_DEV_FLAT=1 timeout -k 10 200 python tools/dev_tracker_multi.py 8 200 2>&1 | grep "managers:" ; }
run A=share
run LPSLAM_HIP_SHARED_LAUNCHES=0
run LPSLAM_HIP_SHARE_PRIORITY=1
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false'
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false' LPSLAM_HIP_SHARED_LAUNCHES=0
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false' LPSLAM_HIP_SHARE_PRIORITY=1
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false, "prefetch": false'
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false, "prefetch": false' LPSLAM_HIP_SHARED_LAUNCHES=0
run LPSLAM_DEV_TRACKER_CFG=', "prefetch": false'
run LPSLAM_HIP_BA_GRAPH=0
run LPSLAM_HIP_BA_GRAPH=0 LPSLAM_HIP_SHARE_PRIORITY=1
