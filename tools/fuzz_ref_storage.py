"""Random maps through the compiled MapStorage.cpp / OdoSLAM.cpp (oracle/_ref) against include/se2lam_amd/MapStorage.h: the mirror writes a map
and its trajectory (tests/cpp_mapstorage.cpp gen), the reference loads the map file's node structure into its own Map, saves it again and
writes the trajectory.  CPU only.   python tools/fuzz_ref_storage.py"""
import os, sys, subprocess, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_ref_compiled as T
from test_mapstorage import _parse
from oracle import ref
exe = '/tmp/cpp_mapstorage_fz'
subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp_mapstorage.cpp"), "-o", exe], check=True)
rng = np.random.default_rng(7)
n_ok = n_nodes = 0
bTc = np.eye(4, dtype=np.float32); bTc[:3, :3] = [[0, 0, 1], [-1, 0, 0], [0, -1, 0]]; bTc[0, 3] = 100; bTc[2, 3] = 300
for seed in range(100, 260):
    nkf, nmp = int(rng.integers(2, 30)), int(rng.integers(1, 80))
    with tempfile.TemporaryDirectory() as tmp:
        a = tmp + "/a/"; os.makedirs(a)
        subprocess.run([exe, "gen", a, str(seed), str(nkf), str(nmp)], check=True, capture_output=True)
        docs, top = _parse(open(a + "se2lam.map").read())
        ev = T._fs_events(docs)
        m = ref.RefMap(np.eye(3), np.eye(4), 2.0)
        nk = m.storage_load("\n".join(ev) + "\n")
        got, none = T._fs_canonical(m.storage_save().split("\n"))
        want, dups = T._fs_canonical(ev)
        assert none == 0 and got == want, (seed, nkf, nmp)
        alive = [i for i in range(nkf) if i % 5 != 3]
        t = tmp + "/t"; os.makedirs(t)
        txt = m.save_trajectory(bTc, t, frame_ids=[7 * i + 3 for i in alive])
        assert txt == open(a + "se2lam_kf_trajectory.txt").read(), (seed, "trajectory")
        n_ok += 1; n_nodes += len(want)
print(f"fuzz_storage: {n_ok} random maps (2-29 key frames, 1-79 map points): the reference's loadMap + saveMap reproduce the mirror's file node for node ({n_nodes} nodes), its trajectory file byte for byte")
