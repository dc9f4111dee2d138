"""Test helper: reads a DBoW2 binary vocabulary (the layout tools/train_vocabulary.py writes and lpslam_amd/host/bow.cpp reads)."""
import os
import struct

import numpy as np

VOCAB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vocab_k10_L3.dbow2")


def read_vocab(path=VOCAB):
    raw = open(path, "rb").read()
    n, size, k, L, scoring, weighting = struct.unpack_from("<6I", raw, 0)
    assert size == 41
    n = (len(raw) - 24) // 41            # the header counts the records, or the records and the root (DBoW2): the file decides
    rec = np.frombuffer(raw, np.uint8, n * 41, 24).reshape(n, 41)
    return dict(k=k, L=L, parent=rec[:, :4].copy().view("<u4").reshape(n).astype(np.int32), desc=rec[:, 4:36].copy(),
                weight=rec[:, 36:40].copy().view("<f4").reshape(n), is_leaf=rec[:, 40].copy())
