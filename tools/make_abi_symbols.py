#!/usr/bin/env python3
"""Writes tests/golden/abi_symbols.txt: the mangled names a client of the REFERENCE's exported class needs from `liblpslam.so`.

A small client (below; it calls every public member of `LpSlamManager` and `LpSlamConfiguration`) is compiled against the
reference's own interface headers (/root/reference/src/Interface, read in place: nothing of them is copied), and the undefined
`LpSlamManager` / `LpSlamConfiguration` symbols of its object file are the list.  The list is DATA (names only); it travels
with the repository, the headers do not.  `tests/test_abi_cpu.py::test_reference_header_client_links` checks that
`nm -D lpslam_amd/liblpslam.so` covers every line.  Run in the build container:  python tools/make_abi_symbols.py
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INTERFACE = "/root/reference/src/Interface"
OUT = os.path.join(ROOT, "tests", "golden", "abi_symbols.txt")

CLIENT = r"""
#include "LpSlamManager.h"
#include "LpSlamConfiguration.h"
static void on_pose(LpSlamGlobalStateInTime const&, void*) {}
static void on_image(LpSlamTimestamp, uint32_t, uint8_t*, LpSlamImageDescription, void*) {}
int main(int argc, char** argv)
{
    LpSlamConfiguration conf;
    LpSlamCameraConfiguration cam = conf.createDefaultCameraConfiguration();
    LpSlamManager m;
    m.logToFile("x.log"); m.setLogLevel(LpSlamLogLevel_Info);
    m.addOnReconstructionCallback(&on_pose, nullptr);
    m.addRequestNavDataCallback(nullptr, nullptr);
    m.addRequestNavTransformation(nullptr, nullptr);
    m.addOnImageCallback(&on_image, nullptr);
    LpSlamGlobalStateInTime st{};
    m.updateGlobalReferenceState(st);
    m.addImageFromFile("a.png"); m.addStereoImageFromFiles("l.png", "r.png");
    m.addMarker(LpSlamMarkerIdentifier{}, LpSlamMarkerState{});
    LpSlamImageDescription d{};
    uint8_t px[16] = {0}; uint32_t n = 0;
    m.addImageFromBuffer(0, 0, px, d); m.addStereoImageFromBuffer(0, 0, px, px, d);
    LpSlamManager::compressImage(px, d, px, &n);
    m.setCameraConfiguration(cam);
    m.readConfigurationFile(argv[argc - 1]); m.readReplayItems("r.pb");
    m.addSource("s", "{}"); m.addTracker("t", "{}"); m.addProcessor("p", "{}");
    m.setShowLiveStream(false); m.setWriteImageFiles(false); m.setRecord(false); m.setRecordImages(false);
    m.start(); m.stop();
    (void)m.getSlamStatus();
    float r[2] = {0, 0};
    m.mappingAddLaserScan(st, r, 2, 0.f, 1.f, 0.f, 1.f, 0.5f, 10.f);
    (void)m.mappingGetMapRawSize();
    int8_t cells[4];
    (void)m.mappingGetMapRaw(cells, 4);
    LpSlamFeatureEntry fe[2]; LpSlamMatrix9x9 t{};
    (void)m.mappingGetFeatures(LpSlamMapBoundary{}, fe, 2, t);
    (void)m.mappingGetFeaturesCount(LpSlamMapBoundary{});
    m.mappingSetMode(true); m.mappingSetFilename("map.db"); m.mappingExportCSV("f.csv");
    return 0;
}
"""


def main():
    if not os.path.isdir(REF_INTERFACE):
        sys.exit("the reference tree is not present: %s" % REF_INTERFACE)
    with tempfile.TemporaryDirectory() as td:
        src, obj = os.path.join(td, "client.cpp"), os.path.join(td, "client.o")
        open(src, "w").write(CLIENT)
        subprocess.check_call(["g++", "-std=c++17", "-O0", "-c", "-I" + REF_INTERFACE, "-o", obj, src])
        nm = subprocess.check_output(["nm", "-u", obj], text=True)
        names = sorted({ln.split()[-1] for ln in nm.splitlines() if "LpSlamManager" in ln or "LpSlamConfiguration" in ln})
        # the client must also LINK and RUN against the product library (it fails cleanly without a GPU: no tracker is added)
        lib_dir = os.path.join(ROOT, "lpslam_amd")
        exe = os.path.join(td, "client")
        subprocess.check_call(["g++", "-o", exe, obj, "-L" + lib_dir, "-llpslam", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link," + lib_dir, "-pthread"])
    with open(OUT, "w") as f:
        f.write("\n".join(names) + "\n")
    print("%d symbols -> %s" % (len(names), OUT))


if __name__ == "__main__":
    main()
