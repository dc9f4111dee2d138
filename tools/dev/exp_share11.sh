cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" LPSLAM_DEV_FLAT=1 LPSLAM_DEV_STATS=1 timeout -k 10 200 python tools/dev_tracker_multi.py ${N:-8} ${F:-200} 2>&1 | grep "managers:\|mean over [0-9]* managers" | tail -${T:-2} | cut -c1-460; }
run LPSLAM_HIP_SHARE_POSE_QUIET_US=10
run LPSLAM_HIP_SHARE_POSE_QUIET_US=25
run LPSLAM_HIP_SHARE_POSE_QUIET_US=50
run LPSLAM_DEV_TRACKER_CFG=', "loopClosure": false'
run LPSLAM_HIP_SHARE_POSE_QUIET_US=2
