# usage (on the GPU box, through gpurun): bash tools/refresh_profiles.sh r02x
# bench line, rocprofv3 kernel statistics of the same command, and four separate PMC passes (the guide's HBM / rocprofv3 recipe:
# FETCH_SIZE and WRITE_SIZE each in a pass of their own, never combined with tracing domains) -> gpurun_out/<tag>_*
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench.log 2>&1
grep "^{\"metric\"" gpurun_out/${TAG}_bench.log | tail -1 > gpurun_out/${TAG}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ks -o ks --output-format csv -- python3 bench.py --no-cpu --no-extras > gpurun_out/ks.log 2>&1
cp $(find gpurun_out/ks -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv
python3 tools/kstats.py gpurun_out/ks > gpurun_out/${TAG}_bench_kernel_stats.txt
rm -rf gpurun_out/ks
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d gpurun_out/p1 -o p --output-format csv -- python3 bench.py --no-cpu --no-extras --steps 4 --warmup 1 > gpurun_out/p1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/p2 -o p --output-format csv -- python3 bench.py --no-cpu --no-extras --steps 4 --warmup 1 > gpurun_out/p2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/p3 -o p --output-format csv -- python3 bench.py --no-cpu --no-extras --steps 4 --warmup 1 > gpurun_out/p3.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAVES -d gpurun_out/p4 -o p --output-format csv -- python3 bench.py --no-cpu --no-extras --steps 4 --warmup 1 > gpurun_out/p4.log 2>&1
python3 tools/pmc_summary.py gpurun_out/p1 gpurun_out/p2 gpurun_out/p3 gpurun_out/p4 > gpurun_out/${TAG}_pmc_kernels.json
rm -rf gpurun_out/p1 gpurun_out/p2 gpurun_out/p3 gpurun_out/p4
echo REFRESH-OK
