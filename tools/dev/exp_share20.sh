cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" LPSLAM_DEV_FLAT=1 LPSLAM_DEV_STATS=1 timeout -k 10 200 python tools/dev_tracker_multi.py ${N:-8,16} ${F:-200} 2>&1 | grep "share: role\|managers:\|mean over [0-9]* managers" | tail -5 | cut -c1-330; }
for i in 1 2; do
run LPSLAM_HIP_SHARE_PRIO=0 LPSLAM_HIP_SHARE_TRACE=
run LPSLAM_HIP_SHARE_PRIO=1
run LPSLAM_HIP_SHARE_PRIO=1 LPSLAM_HIP_AUX_STREAM=0
done
