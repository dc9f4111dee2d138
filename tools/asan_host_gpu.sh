#!/bin/bash
# The host mirror (manager, trackers, replay, rectification) built with AddressSanitizer + UBSan, run through its GPU tests on an
# MI355X box (the HIP library itself stays the normal build: GPU sanitizers are not available on the pool).  Covers the threads of
# the host side: worker / notify / mapping thread, the prefetch helper, two managers in one process.
# Usage (through gpurun): bash tools/asan_host_gpu.sh
set -e
cd "$(dirname "$0")/.."
TMP=$(mktemp -d)
cp lpslam_amd/liblpslam.so "$TMP/host.so"
restore() { cp "$TMP/host.so" lpslam_amd/liblpslam.so; touch lpslam_amd/liblpslam.so; rm -rf "$TMP"; }
trap restore EXIT
SAN="-O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ $SAN -std=c++17 -shared -fvisibility=hidden -pthread -o lpslam_amd/liblpslam.so lpslam_amd/host/*.cpp -Llpslam_amd -llpslam_hip -Wl,-rpath,"$PWD/lpslam_amd"
touch lpslam_amd/liblpslam.so
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:protect_shadow_gap=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    timeout -k 10 900 python -m pytest tests/test_host_gpu.py tests/test_track_gpu.py tests/test_rectify_gpu.py -q -p no:cacheprovider
