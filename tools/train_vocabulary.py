#!/usr/bin/env python3
"""Trains the small test vocabulary tests/golden/vocab_k10_L3.dbow2 (DBoW2 binary layout) from descriptors of the committed synthetic
sequences: hierarchical k-majority clustering of 256-bit ORB descriptors (k = 10 children, depth L = 3 -> at most 1110 nodes),
TF-IDF word weights log(N / N_i) over the N training images.  Deterministic (fixed seeds); the descriptors come from the CPU oracle's
extractor.  The product reads such files with lpslam_amd/host/bow.cpp (Vocabulary::load).  [UPSTREAM] DBoW2 TemplatedVocabulary::create
(k-means++ seeding and binary medians there; a plain seeded k-majority here -- any tree is a valid vocabulary)."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from lpslam_amd import synth            # noqa: E402

K, L = 10, 3
OUT = os.path.join(ROOT, "tests", "golden", "vocab_k10_L3.dbow2")


def hamming(a, b):
    """a [n, 32] uint8 against b [m, 32] -> [n, m]"""
    x = np.bitwise_xor(a[:, None, :], b[None, :, :])
    return np.unpackbits(x, axis=2).sum(axis=2)


def k_majority(desc, k, rng, iters=8):
    if len(desc) <= k:
        return [desc[i:i + 1] for i in range(len(desc))], desc.copy()
    cent = desc[rng.choice(len(desc), k, replace=False)].copy()
    for _ in range(iters):
        d = np.stack([np.unpackbits(np.bitwise_xor(desc, c[None, :]), axis=1).sum(axis=1) for c in cent], axis=1)
        lab = d.argmin(axis=1)
        for c in range(k):
            m = desc[lab == c]
            if len(m):
                bits = np.unpackbits(m, axis=1).mean(axis=0) >= 0.5
                cent[c] = np.packbits(bits)
    d = np.stack([np.unpackbits(np.bitwise_xor(desc, c[None, :]), axis=1).sum(axis=1) for c in cent], axis=1)
    lab = d.argmin(axis=1)
    keep = [c for c in range(k) if (lab == c).any()]
    return [desc[lab == c] for c in keep], cent[keep]


def main():
    O.build()
    p = O.params(1000, 1.2, 4)
    images = []
    seq = synth.StereoSequence(640, 480, 4, n_points=6000)
    images += [seq.frame(i)[0] for i in range(0, 40, 4)]
    turn, _ = synth.turning_sequence(640, 480, 132)
    images += [turn[i][0] for i in range(0, 132, 6)]
    per_image = [O.extract(img, p)[1] for img in images]
    desc = np.concatenate(per_image)
    print("training on %d descriptors of %d images" % (len(desc), len(images)))
    rng = np.random.default_rng(12345)
    parent, node_desc, is_leaf = [], [], []

    def grow(members, parent_id, level):
        groups, cents = k_majority(members, K, rng)
        for g, c in zip(groups, cents):
            parent.append(parent_id); node_desc.append(c); is_leaf.append(0)
            nid = len(parent)
            if level < L and len(g) > 1:
                grow(g, nid, level + 1)
            else:
                is_leaf[nid - 1] = 1
    grow(desc, 0, 1)
    vocab = dict(k=K, L=L, parent=np.array(parent, np.int32), desc=np.array(node_desc, np.uint8), weight=np.ones(len(parent), np.float32),
                 is_leaf=np.array(is_leaf, np.uint8))
    n_words = int(vocab["is_leaf"].sum())
    seen = np.zeros(n_words)
    for d in per_image:
        w, _, _ = O.bow_transform(vocab, d, 0)
        seen[np.unique(w)] += 1
    idf = np.log(len(images) / np.maximum(seen, 1.0))
    word_of_node = np.cumsum(vocab["is_leaf"]) - 1
    weight = np.where(vocab["is_leaf"] == 1, idf[word_of_node], 0.0).astype(np.float32)
    with open(OUT, "wb") as f:
        f.write(struct.pack("<6I", len(parent), 41, K, L, 0, 0))          # scoring L1_NORM = 0, weighting TF_IDF = 0
        for i in range(len(parent)):
            f.write(struct.pack("<I", parent[i])); f.write(vocab["desc"][i].tobytes()); f.write(struct.pack("<f", weight[i])); f.write(struct.pack("<B", is_leaf[i]))
    print("wrote %s: %d nodes, %d words, %d bytes" % (OUT, len(parent), n_words, os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
