#!/usr/bin/env python3
"""Generates tests/golden/g17_jpeg.npz: JPEG streams written by Pillow (libjpeg-turbo) and the grey samples libjpeg decodes from
them -- the fixtures of host/jpeg.cpp (the reference decodes its compressed frames with cv::imdecode = libjpeg,
src/Manager/SlamManager.cpp:1139-1146, src/Manager/ReplayEngine.cpp:123).  Grey files: the samples; colour files: the luma plane
libjpeg delivers for greyscale output (Image.draft("L") = JCS_GRAYSCALE, what IMREAD_GRAYSCALE asks for)."""
import io
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpslam_amd import synth            # noqa: E402


def grey_of(data):
    im = Image.open(io.BytesIO(data))
    im.draft("L", im.size)
    im.load()
    assert im.mode == "L", im.mode
    return np.asarray(im).copy()


def main():
    rng = np.random.default_rng(17)
    frame = synth.StereoSequence(640, 480, 4, n_points=6000).frame(0)[0]
    ramp = (np.add.outer(np.arange(77), np.arange(123)) * 255 // (76 + 122)).astype(np.uint8)
    noise = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    colour = np.stack([np.roll(frame[:61, :97], s, axis=1) for s in (0, 5, 11)], axis=-1)
    cases = {}

    def add(name, img, **kw):
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", **kw)
        data = buf.getvalue()
        cases[name] = (np.frombuffer(data, np.uint8).copy(), grey_of(data))
    add("grey_320x240_q95", frame[100:340, 160:480], quality=95)                 # what the recorder writes (cv::imencode default: 95)
    add("grey_201x99_q90", frame[7:106, 11:212], quality=90)
    add("grey_ramp_123x77_q70", ramp, quality=70)                               # not a multiple of 8 either way
    add("grey_noise_64x48_q100", noise, quality=100)                            # every coefficient alive, range limiting
    add("grey_noise_64x48_q5", noise, quality=5)                                # large quantisation steps
    add("grey_optimised_huffman", frame[:200, :312], quality=85, optimize=True)  # tables of the file, not the standard ones
    add("grey_restart_markers", frame[:120, :200], quality=80, restart_marker_blocks=7)
    add("colour_420_97x61_q75", colour, quality=75, subsampling=2)              # luma of an interleaved 4:2:0 file
    add("colour_444_97x61_q92", colour, quality=92, subsampling=0)
    add("colour_422_restart", colour, quality=60, subsampling=1, restart_marker_rows=1)
    buf = io.BytesIO()
    Image.fromarray(frame[:64, :64]).save(buf, "JPEG", quality=80, progressive=True)
    out = {"progressive_refused": np.frombuffer(buf.getvalue(), np.uint8).copy()}
    for name, (data, grey) in cases.items():
        out["jpeg_" + name] = data
        out["grey_" + name] = grey
        print("%-28s %6d bytes -> %dx%d" % (name, len(data), grey.shape[1], grey.shape[0]))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g17_jpeg.npz"), **out)


if __name__ == "__main__":
    main()
