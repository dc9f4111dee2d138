cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" LPSLAM_DEV_FLAT=1 LPSLAM_DEV_STATS=1 timeout -k 10 200 python tools/dev_tracker_multi.py ${N:-8} ${F:-200} 2>&1 | grep "managers:\|mean over [0-9]* managers\|cpu.stat" | tail -${T:-2} | cut -c1-420; }
N=1,2,4,8,12,16 T=30 run A=1
N=8,16 run LPSLAM_HIP_POLL_SLEEP_US=0
N=8,16 run LPSLAM_HIP_POLL_SLEEP_US=20
N=8 run LPSLAM_HIP_SHARE_QUIET_US=15
N=8 run LPSLAM_HIP_SHARE_QUIET_US=2
N=8 F=400 run A=1
