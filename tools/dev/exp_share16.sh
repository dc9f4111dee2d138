cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" LPSLAM_DEV_FLAT=1 LPSLAM_DEV_STATS=1 timeout -k 10 200 python tools/dev_tracker_multi.py ${N:-8,16} ${F:-200} 2>&1 | grep "managers:\|mean over [0-9]* managers" | tail -4 | cut -c1-290; }
run A=1
run LPSLAM_HIP_POOL_RESERVE=2
run LPSLAM_HIP_POOL_RESERVE=4
run LPSLAM_HIP_POOL_RESERVE=8
run LPSLAM_HIP_POOL_RESERVE=16
