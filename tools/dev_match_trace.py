"""development: where a window-matcher call of the tracker spends its time (LPSLAM_HIP_MATCH_TRACE) and what the pose optimiser's calls look like
(LPSLAM_HIP_PO_TRACE), medians over one tracked sequence.  usage: dev_match_trace.py [frames]"""
import os, re, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, LPSLAM_HIP_MATCH_TRACE="1", LPSLAM_HIP_PO_TRACE="1", FRAMES=sys.argv[1] if len(sys.argv) > 1 else "90")
if len(sys.argv) > 2: env["TRACKER_CFG"] = ', "vocabFile": "%s"' % os.path.abspath(sys.argv[2])
out = subprocess.run([sys.executable, os.path.join(root, "tools", "dev_tracker_time.py")], env=env, capture_output=True, text=True)
rows = {}
for l in out.stderr.splitlines():
    m = re.match(r"window_match: policy (\d), (\d+) queries, (\d+) matches, (\d+) rescans; us: setup ([\d.]+), staged \+ launched ([\d.]+), lists back ([\d.]+), replayed ([\d.]+)", l)
    if m:
        p = int(m.group(1)); big = int(m.group(2)) > 600
        rows.setdefault(("policy %d %s" % (p, "many queries" if big else "few queries")), []).append([float(x) for x in m.groups()[1:]])
    m = re.match(r"bow_tree_match: (\d+) queries, (\d+) targets, (\d+) matches; us: sorted ([\d.]+), staged ([\d.]+), lists back ([\d.]+), replayed ([\d.]+)", l)
    if m:
        rows.setdefault("bow_tree_match", []).append([float(x) for x in m.groups()])
    m = re.match(r"pose_optimize: (\d+) observations, (\d+) inliers, (\d+) passes, ([\d.]+) us", l)
    if m:
        rows.setdefault("pose_optimize", []).append([float(x) for x in m.groups()])
for k, a in sorted(rows.items()):
    a = np.array(a)
    if k == "pose_optimize":
        print("%-28s %4d calls; medians: observations %.0f, inliers %.0f, passes %.0f, %.1f us (%.2f us per pass); observations <= 64: %.0f %%, <= 128: %.0f %%" % (
            k, len(a), *np.median(a, axis=0), np.median(a[:, 3] / a[:, 2]), 100 * np.mean(a[:, 0] <= 64), 100 * np.mean(a[:, 0] <= 128)))
    elif k == "bow_tree_match":
        md = np.median(a, axis=0)
        print("%-28s %4d calls; medians: queries %.0f, targets %.0f, matches %.0f; us: sorted %.1f, staged %.1f, lists back %.1f, replayed %.1f" % (k, len(a), *md))
    else:
        md = np.median(a, axis=0)
        print("%-28s %4d calls; medians: queries %.0f, matches %.0f, rescans %.2f (mean %.2f); us: setup %.1f, launched %.1f, lists back %.1f, replayed %.1f" % (
            k, len(a), md[0], md[1], md[2], np.mean(a[:, 2]), md[3], md[4], md[5], md[6]))
print(out.stdout[-600:])
