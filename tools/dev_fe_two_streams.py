"""development: does the front end of 32 images finish sooner as two halves on two streams (tails of the low-occupancy stages overlap
with the chip-filling ones) than as one launch chain?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
n = 32
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=n)
seq = synth.StereoSequence(1280, 720, 0)
for i in range(n // 2):
    l, r = seq.frame(i % 8)
    ctx.upload(2 * i, l); ctx.upload(2 * i + 1, r)
def one():
    ctx.extract_range(0, n); ctx.sync()
def two():
    with ctx.prefetch():
        ctx.extract_range(0, n // 2)
    ctx.extract_range(n // 2, n // 2)
    ctx.prefetch_join(); ctx.sync()
for name, f in (("one chain", one), ("two streams", two), ("one chain", one), ("two streams", two)):
    for _ in range(3): f()
    t = time.perf_counter()
    for _ in range(20): f()
    print("%s: %.1f us per %d images" % (name, 1e6 * (time.perf_counter() - t) / 20, n))
kp_a = [ctx.keypoints(i)[0].tobytes() for i in range(n)]
one()
kp_b = [ctx.keypoints(i)[0].tobytes() for i in range(n)]
print("same keypoints:", kp_a == kp_b)
