#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (host mirror and oracle) with the CPU test suite.  GPU sanitizers are not
# available on the MI355X pool; this covers the host logic, the replay / JSON parsers, the rectification and two-view code.
# Usage: bash tools/asan_cpu.sh        (restores the normal builds afterwards)
set -e
cd "$(dirname "$0")/.."
TMP=$(mktemp -d)
cp lpslam_amd/liblpslam.so "$TMP/host.so"; cp oracle/liblpslam_oracle.so "$TMP/oracle.so"
restore() { cp "$TMP/host.so" lpslam_amd/liblpslam.so; cp "$TMP/oracle.so" oracle/liblpslam_oracle.so; touch lpslam_amd/liblpslam.so oracle/liblpslam_oracle.so; rm -rf "$TMP"; }
trap restore EXIT
SAN="-O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ $SAN -std=c++17 -shared -fvisibility=hidden -pthread -o lpslam_amd/liblpslam.so lpslam_amd/host/*.cpp -Llpslam_amd -llpslam_hip -Wl,-rpath,"$PWD/lpslam_amd"
(cd oracle && gcc $SAN -std=c11 -ffp-contract=off -shared -o liblpslam_oracle.so ora_orb.c ora_match.c ora_ba.c ora_sim3.c ora_bow.c -lm)
touch lpslam_amd/liblpslam.so oracle/liblpslam_oracle.so
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests -q -m "not gpu" -p no:cacheprovider
