cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_share_gpu.py tests/test_host_gpu.py -x -q -m gpu 2>&1 | grep "^E  \|passed\|failed" | head -20
LPSLAM_HIP_SHARE_TRACE=1 LPSLAM_DEV_FLAT=1 LPSLAM_DEV_STATS=1 timeout -k 10 300 python3 tools/dev_tracker_multi.py 8 150 > gpurun_out/strace.log 2>&1; grep "managers" gpurun_out/strace.log | tail -2 | cut -c1-420; python3 tools/dev/share_trace_summary.py gpurun_out/strace.log 200 3
