"""include/se2lam_amd/ORBVocabulary.h - the DBoW2 vocabulary as se2lam uses it (loadFromBinaryFile OdoSLAM.cpp:45,
transform(.., 4) KeyFrame.cpp:251, score GlobalMapper.cpp:237) - against an independent numpy walk of the same tree.
The vocabulary file is written here, byte for byte in the layout TemplatedVocabulary::saveToBinaryFile produces
(/root/reference/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1526-1546); the reference ships no vocabulary and no
golden BoW vectors, so the pin is the definition.  Host code only: runs without a GPU."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POP = np.array([bin(i).count("1") for i in range(256)], np.int32)


def _random_vocabulary(rng, k, L, scoring, weighting, stop_frac=0.05):
    """nodes in breadth-first order (a parent precedes its children, as k-means tree building leaves them), leaves only at
    depth L except for a few early ones; returns parent, descriptor, weight, leaf arrays incl. the root at index 0"""
    parent, depth = [0], [0]
    frontier = [0]
    for d in range(1, L + 1):
        nxt = []
        for p in frontier:
            for _ in range(int(rng.integers(2, k + 1))):
                parent.append(p); depth.append(d); nxt.append(len(parent) - 1)
        frontier = nxt
    n = len(parent)
    parent = np.array(parent, np.int32); depth = np.array(depth)
    has_child = np.zeros(n, bool); has_child[parent[1:]] = True
    leaf = ~has_child
    leaf[0] = False
    desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    # children resemble their parent (as cluster centres do), so that the walk is not a coin toss at every level
    for i in range(1, n):
        flip = (rng.random((32, 8)) < 0.08 * (1 + depth[i]))
        desc[i] = desc[parent[i]] ^ np.packbits(flip, axis=1).reshape(32)
    weight = np.where(leaf, rng.uniform(0.1, 9.0, n), 0.0).astype(np.float32)
    weight[leaf & (rng.random(n) < stop_frac)] = 0.0                     # stopped words
    if weighting in (1, 3):                                              # TF / BINARY vocabularies store 1
        weight[leaf & (weight > 0)] = 1.0
    return parent, desc, weight, leaf


def _write(path, k, L, scoring, weighting, parent, desc, weight, leaf):
    with open(path, "wb") as f:
        f.write(struct.pack("<IIiiii", len(parent), 41, k, L, scoring, weighting))
        for i in range(1, len(parent)):
            f.write(struct.pack("<i", int(parent[i]))); f.write(desc[i].tobytes())
            f.write(struct.pack("<f", float(weight[i]))); f.write(b"\x01" if leaf[i] else b"\x00")


def _numpy_transform(parent, desc, weight, leaf, L, scoring, weighting, feats, levelsup):
    n = len(parent)
    children = [[] for _ in range(n)]
    for i in range(1, n):
        children[parent[i]].append(i)
    word_of = -np.ones(n, int); word_of[np.nonzero(leaf)[0]] = np.arange(int(leaf.sum()))
    nid_level = L - levelsup
    bow, fv = {}, {}
    for i, f in enumerate(feats):
        node, level, nid = 0, 0, (0 if nid_level <= 0 else None)
        while True:
            level += 1
            ch = children[node]
            d = POP[desc[ch] ^ f].sum(1)
            node = ch[int(np.argmin(d))]                                  # first minimum
            if level == nid_level:
                nid = node
            if leaf[node]:
                break
        if nid is None:
            nid = node
        w = float(weight[node])
        if not w > 0:
            continue
        wid = int(word_of[node])
        if weighting in (0, 1):
            bow[wid] = bow.get(wid, 0.0) + w
        else:
            bow.setdefault(wid, w)
        fv.setdefault(int(nid), []).append(i)
    words = sorted(bow)
    vals = np.array([bow[w] for w in words], np.float64)
    if scoring == 5:
        if weighting in (0, 1) and len(vals):
            vals = vals / len(vals)
    else:
        norm = np.sqrt((vals ** 2).sum()) if scoring == 1 else np.abs(vals).sum()
        if norm > 0:
            vals = vals / norm
    return words, vals, fv


def _score(scoring, wa, va, wb, vb):
    a, b = dict(zip(wa, va)), dict(zip(wb, vb))
    common = sorted(set(a) & set(b))
    if scoring == 0:
        return -sum(abs(a[w] - b[w]) - abs(a[w]) - abs(b[w]) for w in common) / 2.0
    if scoring == 1:
        s = sum(a[w] * b[w] for w in common)
        return 1.0 if s >= 1 else 1.0 - np.sqrt(1.0 - s)
    if scoring == 2:
        return 2.0 * sum(a[w] * b[w] / (a[w] + b[w]) for w in common)
    if scoring == 3:
        le = np.log(np.finfo(np.float64).eps)
        return sum(a[w] * np.log(a[w] / b[w]) if w in b else a[w] * (np.log(a[w]) - le) for w in sorted(a))
    if scoring == 4:
        return sum(np.sqrt(a[w] * b[w]) for w in common)
    return sum(a[w] * b[w] for w in common)


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("voc") / "cpp_vocabulary"
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp_vocabulary.cpp"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return str(out)


def _parse(stdout):
    rows = {l.split()[0]: l.split()[1:] for l in stdout.splitlines() if l}
    bows, fvs = [], []
    for s in (0, 1):
        items = [x.split(":") for x in rows[f"BOW{s}"]]
        bows.append(([int(a) for a, _ in items], np.array([float(b) for _, b in items])))
        fvs.append({int(x.split(":")[0]): [int(t) for t in x.split(":")[1].split(",")] for x in rows[f"FV{s}"]})
    return rows["VOC"], bows, fvs, [float(x) for x in rows["SCORE"]]


@pytest.mark.parametrize("k,L,scoring,weighting,levelsup,seed", [
    (10, 4, 0, 0, 2, 0),     # the ORB-SLAM vocabulary's shape in small: L1 scoring, TF-IDF
    (6, 6, 0, 0, 4, 1),      # transform(.., 4) on a 6-level tree (KeyFrame.cpp:249-251)
    (4, 3, 1, 1, 4, 2),      # L2 / TF, levelsup beyond the depth: every feature files under the root
    (5, 4, 5, 0, 1, 3),      # dot product: no normalisation, divided by the number of words
    (3, 5, 2, 2, 0, 4),      # chi-square / IDF, levelsup 0: the feature vector is keyed by the leaves
    (8, 3, 3, 3, 1, 5),      # KL / BINARY
    (7, 4, 4, 0, 3, 6),      # Bhattacharyya
])
def test_transform_and_score_equal_an_independent_tree_walk(exe, tmp_path, k, L, scoring, weighting, levelsup, seed):
    rng = np.random.default_rng(seed)
    parent, desc, weight, leaf = _random_vocabulary(rng, k, L, scoring, weighting)
    voc = tmp_path / "voc.bin"
    _write(voc, k, L, scoring, weighting, parent, desc, weight, leaf)
    sets = []
    for s in range(2):
        base = desc[rng.choice(np.nonzero(leaf)[0], 400)]                 # features near words, with repeats
        noise = np.packbits(rng.random((400, 32, 8)) < 0.06, axis=2).reshape(400, 32)
        f = base ^ noise
        f[:40] = sets[0][:40] if s else f[:40]                            # the two key frames share some features
        sets.append(f)
        (tmp_path / f"d{s}.bin").write_bytes(f.tobytes())
    r = subprocess.run([exe, str(voc), str(tmp_path / "d0.bin"), str(tmp_path / "d1.bin"), str(levelsup),
                        str(tmp_path / "resaved.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    head, bows, fvs, scores = _parse(r.stdout)
    assert [int(x) for x in head] == [k, L, len(parent), int(leaf.sum()), scoring, weighting]
    ref = [_numpy_transform(parent, desc, weight, leaf, L, scoring, weighting, f, levelsup) for f in sets]
    for s in range(2):
        words, vals, fv = ref[s]
        assert bows[s][0] == words and len(words) > 20
        assert np.allclose(bows[s][1], vals, rtol=1e-14, atol=0)
        assert fvs[s] == fv
        if levelsup >= L:
            assert list(fv) == [0]
    want = _score(scoring, ref[0][0], ref[0][1], ref[1][0], ref[1][1])
    assert scores[0] == pytest.approx(want, rel=1e-12, abs=1e-15)
    if scoring == 0:
        assert scores[1] == pytest.approx(1.0, abs=1e-12) and 0.0 < scores[0] < 1.0      # a vector scores 1 against itself
    # saveToBinaryFile writes the file that was read
    assert (tmp_path / "resaved.bin").read_bytes() == voc.read_bytes()


def test_malformed_files_are_refused(exe, tmp_path):
    rng = np.random.default_rng(9)
    parent, desc, weight, leaf = _random_vocabulary(rng, 4, 3, 0, 0)
    good = tmp_path / "voc.bin"
    _write(good, 4, 3, 0, 0, parent, desc, weight, leaf)
    (tmp_path / "d.bin").write_bytes(desc[1:20].tobytes())
    blob = good.read_bytes()
    cases = {"truncated": blob[:-17], "node size": blob[:4] + struct.pack("<I", 40) + blob[8:],
             "forward parent": blob[:24] + struct.pack("<i", 5) + blob[28:], "empty": b""}
    for name, data in cases.items():
        p = tmp_path / (name.replace(" ", "_") + ".bin")
        p.write_bytes(data)
        r = subprocess.run([exe, str(p), str(tmp_path / "d.bin"), str(tmp_path / "d.bin"), "4", str(tmp_path / "o.bin")],
                           capture_output=True, text=True)
        assert r.returncode == 1 and "LOAD failed" in r.stdout, name
    r = subprocess.run([exe, str(tmp_path / "missing.bin"), str(tmp_path / "d.bin"), str(tmp_path / "d.bin"), "4",
                        str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 1


def test_feature_vectors_drive_search_by_bow(exe, tmp_path, oracle, synth):
    """KeyFrame::ComputeBoW -> ORBmatcher::SearchByBoW (ORBmatcher.cpp:128-276): the CSR feature vectors of the header are
    what the matcher takes (here the oracle's, which the device path equals bit for bit): features are only compared under
    a common node, so every match joins two features of the same node - and two views of one scene match well."""
    k1, d1 = oracle.orb_extract(synth.frame(0))
    k2, d2 = oracle.orb_extract(synth.frame(1))
    rng = np.random.default_rng(4)
    # a small vocabulary "trained" on the scene: node descriptors are descriptors of frame 0, children near their parent
    k, L = 8, 3
    parent, depth, frontier = [0], [0], [0]
    for lv in range(1, L + 1):
        nxt = []
        for p in frontier:
            for _ in range(k):
                parent.append(p); depth.append(lv); nxt.append(len(parent) - 1)
        frontier = nxt
    n = len(parent)
    parent = np.array(parent, np.int32)
    desc = np.zeros((n, 32), np.uint8)
    for i in range(1, n):
        if parent[i] == 0:
            desc[i] = d1[rng.integers(0, len(d1))]
        else:
            desc[i] = desc[parent[i]] ^ np.packbits(rng.random((32, 8)) < 0.05 * depth[i], axis=1).reshape(32)
    leaf = np.array(depth) == L
    weight = np.where(leaf, 1.0, 0.0).astype(np.float32)
    voc = tmp_path / "voc.bin"
    _write(voc, k, L, 0, 0, parent, desc, weight, leaf)
    (tmp_path / "a.bin").write_bytes(d1.tobytes()); (tmp_path / "b.bin").write_bytes(d2.tobytes())
    r = subprocess.run([exe, str(voc), str(tmp_path / "a.bin"), str(tmp_path / "b.bin"), "2", str(tmp_path / "o.bin")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    _, bows, fvs, scores = _parse(r.stdout)

    def csr(fv):
        nodes = sorted(fv)
        ptr = np.concatenate([[0], np.cumsum([len(fv[x]) for x in nodes])]).astype(np.int32)
        return np.array(nodes, np.int32), ptr, np.concatenate([fv[x] for x in nodes]).astype(np.int32)
    fv1, fv2 = csr(fvs[0]), csr(fvs[1])
    assert sorted(fv1[2].tolist()) == list(range(len(k1)))                 # no stopped words: every feature is filed once
    node1 = np.zeros(len(k1), int); node2 = np.zeros(len(k2), int)
    for nd, feats in fvs[0].items():
        node1[feats] = nd
    for nd, feats in fvs[1].items():
        node2[feats] = nd
    h1 = np.ones(len(k1), np.uint8); h2 = np.ones(len(k2), np.uint8)
    m12, nm = oracle.search_by_bow(k1, d1, fv1, h1, k2, d2, fv2, h2, False, 0.9, True)
    i1 = np.nonzero(m12 >= 0)[0]
    assert nm == len(i1) > 150
    assert (node1[i1] == node2[m12[i1]]).all()
    assert 0.2 < scores[0] < 1.0 and scores[1] == pytest.approx(1.0, abs=1e-12)
