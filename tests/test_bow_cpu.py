"""CPU tests of the bag-of-words pieces: the oracle's tree walk against a brute-force numpy restatement, the host mirror's vocabulary
reader (binary and text layouts of DBoW2) and its BoW vector / L1 score against the oracle's.  [UPSTREAM] DBoW2 TemplatedVocabulary,
BowVector, L1Scoring (shinsumicco/DBoW2 @ e8cc74d); the reference refuses to start without a vocabulary
(src/Trackers/OpenVSLAMTrackerBase.cpp:224-227)."""
import ctypes as C
import struct

import numpy as np
import pytest

from bow_util import VOCAB, read_vocab


def _hamming(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def test_tree_walk_against_brute_force(oracle):
    v = read_vocab()
    assert v["k"] == 10 and v["L"] == 3 and v["is_leaf"].sum() == 1000
    rng = np.random.default_rng(5)
    desc = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    desc[:50] = v["desc"][rng.integers(0, len(v["desc"]), 50)]                  # some exactly on nodes: ties between children are real
    children = {}
    for i, p in enumerate(v["parent"]):
        children.setdefault(int(p), []).append(i + 1)
    word_of = np.cumsum(v["is_leaf"]) - 1
    for lu in (0, 1, 2, 4):
        w, ww, nd = oracle.bow_transform(v, desc, lu)
        for f in range(len(desc)):
            cur, level, nid = 0, 0, 0
            while cur == 0 or not v["is_leaf"][cur - 1]:
                level += 1
                ds = [_hamming(desc[f], v["desc"][c - 1]) for c in children[cur]]
                cur = children[cur][int(np.argmin(ds))]                          # argmin: the first minimum
                if level == v["L"] - lu:
                    nid = cur
            assert w[f] == word_of[cur - 1] and ww[f] == v["weight"][cur - 1] and nd[f] == nid


def test_bow_tree_match_semantics(oracle):
    """nearest free target under the same node, first on ties, ratio against the second, a matched target is gone"""
    q = np.zeros((3, 32), np.uint8); t = np.zeros((4, 32), np.uint8)
    t[0, 0] = 0b1; t[1, 0] = 0b1; t[2, 0] = 0b111; t[3, :4] = 255                 # distances from q0 = 0: 1, 1, 3, 32
    idx, dist, n = oracle.bow_tree_match(q, [7, 7, 9], t, [7, 7, 7, 9], 50, 1.0)
    assert list(idx) == [0, 1, 3] and list(dist) == [1, 1, 32] and n == 3      # q0 takes t0 (first of the tie), q1 the next free one
    idx, _, n = oracle.bow_tree_match(q, [7, 7, 9], t, [7, 7, 7, 9], 50, 0.75)
    assert list(idx) == [-1, -1, 3] and n == 1                                   # 1 > 0.75 * 1 for q0; q1 likewise (t0 still free); node 9 has one target
    idx, _, n = oracle.bow_tree_match(q, [7, -1, 9], t, [7, 7, 7, 9], 20, 1.0, t_taken=[1, 0, 0, 0])
    assert list(idx) == [1, -1, -1] and n == 1                                   # t0 was taken beforehand, q1 is no query, 32 > 20


@pytest.fixture(scope="module")
def hostlib():
    from lpslam_amd import _build
    l = C.CDLL(_build.host_library())
    l.lpslam_bow_vocab_load.restype = C.c_int
    l.lpslam_bow_vocab_load.argtypes = [C.c_char_p] + [C.c_void_p] * 7 + [C.c_int32]
    l.lpslam_bow_score.restype = C.c_double
    l.lpslam_bow_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    return l


def _load(hostlib, path):
    k, L, n = C.c_int32(), C.c_int32(), C.c_int32()
    cap = 2000
    parent = np.zeros(cap, np.int32); desc = np.zeros((cap, 32), np.uint8); weight = np.zeros(cap, np.float32); leaf = np.zeros(cap, np.uint8)
    r = hostlib.lpslam_bow_vocab_load(str(path).encode(), C.byref(k), C.byref(L), C.byref(n), parent.ctypes.data, desc.ctypes.data, weight.ctypes.data, leaf.ctypes.data, cap)
    return r, k.value, L.value, parent[:max(r, 0)], desc[:max(r, 0)], weight[:max(r, 0)], leaf[:max(r, 0)]


def test_host_reads_binary_and_text_vocabularies(hostlib, tmp_path):
    v = read_vocab()
    r, k, L, parent, desc, weight, leaf = _load(hostlib, VOCAB)
    assert r == len(v["parent"]) and (k, L) == (10, 3)
    assert np.array_equal(parent, v["parent"]) and np.array_equal(desc, v["desc"]) and np.array_equal(weight, v["weight"]) and np.array_equal(leaf, v["is_leaf"])
    # the text layout of TemplatedVocabulary::loadFromTextFile (ORB-SLAM's ORBvoc.txt): "k L scoring weighting", then one node per line
    txt = tmp_path / "vocab.txt"
    with open(txt, "w") as f:
        f.write("10 3 0 0\n")
        for i in range(len(v["parent"])):
            f.write("%d %d %s %.9g\n" % (v["parent"][i], v["is_leaf"][i], " ".join(str(int(b)) for b in v["desc"][i]), v["weight"][i]))
    r2, k2, L2, parent2, desc2, weight2, leaf2 = _load(hostlib, txt)
    assert r2 == r and (k2, L2) == (10, 3) and np.array_equal(parent2, parent) and np.array_equal(desc2, desc) and np.array_equal(leaf2, leaf)
    assert np.allclose(weight2, weight, rtol=1e-7)
    # the header's node count: DBoW2 counts the root (records + 1), the fixture's trainer did not -- both are read, anything else is not
    raw = bytearray(open(VOCAB, "rb").read())
    n_rec = (len(raw) - 24) // 41
    assert struct.unpack_from("<I", raw, 0)[0] == n_rec
    for count, ok in ((n_rec + 1, True), (n_rec, True), (n_rec + 2, False), (n_rec - 1, False)):
        struct.pack_into("<I", raw, 0, count)
        f2 = tmp_path / ("count_%d.dbow2" % count)
        f2.write_bytes(bytes(raw))
        r3 = _load(hostlib, f2)
        assert (r3[0] == r and np.array_equal(r3[3], parent) and np.array_equal(r3[4], desc)) if ok else r3[0] == -1, count
    struct.pack_into("<I", raw, 0, n_rec + 1); struct.pack_into("<I", raw, 8, 1)    # k = 1: refused by the header check
    f3 = tmp_path / "k1.dbow2"; f3.write_bytes(bytes(raw))
    assert _load(hostlib, f3)[0] == -1
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"\x00" * 100)
    assert _load(hostlib, bad)[0] == -1 and _load(hostlib, tmp_path / "missing")[0] == -1


def test_host_bow_vector_and_score_follow_the_oracle(hostlib, oracle):
    v = read_vocab()
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, (500, 32), dtype=np.uint8); b = a.copy()
    b[250:] = rng.integers(0, 256, (250, 32), dtype=np.uint8)                    # half the descriptors in common
    wa, xa, _ = oracle.bow_transform(v, a, 2); wb, xb, _ = oracle.bow_transform(v, b, 2)
    va, vb = oracle.bow_vector(wa, xa), oracle.bow_vector(wb, xb)
    assert abs(va[1].sum() - 1.0) < 1e-12 and np.all(np.diff(va[0]) > 0)
    want = oracle.bow_score_l1(va, vb)
    got = hostlib.lpslam_bow_score(wa.ctypes.data, xa.ctypes.data, len(wa), wb.ctypes.data, xb.ctypes.data, len(wb))
    assert 0.2 < want < 0.95 and abs(got - want) < 1e-12
    assert abs(hostlib.lpslam_bow_score(wa.ctypes.data, xa.ctypes.data, len(wa), wa.ctypes.data, xa.ctypes.data, len(wa)) - 1.0) < 1e-12
