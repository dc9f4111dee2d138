cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" LPSLAM_DEV_STATS=1 LPSLAM_DEV_FLAT=1 timeout -k 10 200 python tools/dev_tracker_multi.py ${N:-8} 200 2>&1 | grep "managers:\|mean over [0-9]* managers" | tail -2 ; }
N=1,2,4,8,16 run A=share
N=8,16 run LPSLAM_HIP_SHARED_LAUNCHES=0
run LPSLAM_DEV_TRACKER_CFG=', "enableMapping": false'
run LPSLAM_HIP_SHARE_FE_QUIET_US=30 LPSLAM_HIP_SHARE_FE_WINDOW_US=150
run LPSLAM_HIP_SHARE_QUIET_US=15 LPSLAM_HIP_SHARE_WINDOW_US=60
