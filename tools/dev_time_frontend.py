"""Scratch: per-stage HIP-event times of the front end, 32 images per launch (1280x720, 2000 kpts, 8 levels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=n)
seq = synth.StereoSequence(1280, 720, 0)
for i in range(n // 2):
    l, r = seq.frame(i % 8)
    ctx.upload(2 * i, l); ctx.upload(2 * i + 1, r)
names = ["pyramid", "fast", "distribute", "describe"]
acc = np.zeros(4)
for rep in range(6):
    for s, nm in enumerate(names):
        ctx.timer_begin(s); ctx.stage(nm, n); ctx.timer_end(s)
    ctx.sync()
    if rep:
        acc += [ctx.timer_ms(s) for s in range(4)]
print({nm: round(1e3 * a / 5, 1) for nm, a in zip(names, acc)}, "us per", n, "images; total", round(1e3 * acc.sum() / 5, 1))
