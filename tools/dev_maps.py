"""development: library layout of a process that looks like tools/dev_tracker_multi.py (same imports, one context, one small
bundle adjustment so that every runtime library is mapped), every mapping relative to libc's base.  Run under the profiler command a
recorded stack came from: the relative layout of the libraries loaded at start-up is the same from run to run (only the base is
randomised), so the frames of the record resolve to library + offset.
usage: dev_maps.py [0x<address of __restore_rt in the record> 0x<frame> ...]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpslam_amd import manager, synth, _build, hip  # noqa: F401
_build.host_library()
ctx = hip.Context(640, 480, 1000, 1.2, 8, max_images=2)
p = synth.ba_problem(6, 120, 600, 640, 480)
b = hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], hip.ba_obs_array(p), p["cam"]); b.optimize(True, 3); b.close()
libs = {}
for line in open("/proc/self/maps"):
    m = re.match(r"([0-9a-f]+)-([0-9a-f]+) \S+ \S+ \S+ \S+\s+(/\S+)", line)
    if not m: continue
    lo, hi = int(m.group(1), 16), int(m.group(2), 16)
    a = libs.setdefault(m.group(3), [lo, hi]); a[0] = min(a[0], lo); a[1] = max(a[1], hi)
libc = [v for k, v in libs.items() if "/libc.so" in k][0][0]
print("libc base %x" % libc)
for k, v in sorted(libs.items(), key=lambda kv: -kv[1][0]):
    print("%+14x %+14x %s" % (v[0] - libc, v[1] - libc, k))
if len(sys.argv) > 2:
    rec_libc = int(sys.argv[1], 16) - 0x42520            # __restore_rt of this image's glibc 2.35
    for a in sys.argv[2:]:
        rel = int(a, 16) - rec_libc
        hit = [(k, rel - (v[0] - libc)) for k, v in libs.items() if v[0] - libc <= rel < v[1] - libc]
        print("%s -> libc%+x -> %s" % (a, rel, ", ".join("%s+0x%x" % h for h in hit) or "?"))
ctx.close()
