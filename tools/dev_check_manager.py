import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import _build, manager, synth
w, h, n_frames = 640, 480, 24
k = synth.intrinsics(w, h)
seq = synth.StereoSequence(w, h, 4, n_points=6000)
m = manager.Manager(log_level=0)
for num in (0, 1):
    c = manager.default_camera()
    c.camera_number = num; c.f_x = k["fx"]; c.f_y = k["fy"]; c.c_x = k["cx"]; c.c_y = k["cy"]
    c.resolution_x = w; c.resolution_y = h; c.focal_x_baseline = k["fxb"]
    m.set_camera(c)
print(m.add_tracker("VSLAMStereo", '{"cameraSetup": "stereo", "slamKeypoints": 1000, "numLevels": 4, "keyframeInterval": 4}'))
m.collect_results(); m.provide_odometry()
m.start()
for i in range(n_frames):
    l, r = seq.frame(i)
    m.add_stereo((i + 1) * 40_000_000, l, r)
t0 = time.time()
while len(m.results) < n_frames and time.time() - t0 < 60:
    time.sleep(0.01)
st = m.status()
print("status", st.localization, st.key_frames, st.feature_points, st.frame_time)
for r in m.results:
    print(r["valid"], np.round(r["p"], 4), np.round(r["q"], 5))
m.stop()
