"""Map::updateLocalGraph (/root/reference/src/Map.cpp:285-331) on the CSR view: the library's
flat-array implementation against the reference's loop restated with Python sets of (id, object) exactly as it is
written - three covisibility hops, getAllObsMPs(false), the reference key frames.  Host code: runs without a GPU."""
import numpy as np


def _reference_sets(kf_id, covisible, kf_obs, mp_id, mp_obs, cur, level=3):
    local = {cur}
    for _ in range(level):                      # Map.cpp:298-308
        for a in list(local):
            local |= set(covisible[a])
    mps = set()
    for a in local:                             # :310-315
        mps |= set(kf_obs[a])
    refs = set()
    for p in mps:                               # :317-326
        for b in mp_obs[p]:
            if b not in local:
                refs.add(b)
    return (sorted(local, key=lambda a: kf_id[a]), sorted(refs, key=lambda a: kf_id[a]), sorted(mps, key=lambda p: mp_id[p]))


def _random_map(rng, K, M, reach):
    kf_id = rng.permutation(10 * K)[:K]
    mp_id = rng.permutation(10 * M)[:M]
    mp_obs = []
    for j in range(M):
        c = rng.integers(0, K)
        who = np.unique(np.clip(c + rng.integers(-reach, reach + 1, rng.integers(1, 6)), 0, K - 1))
        mp_obs.append([int(a) for a in who])
    kf_obs = [[] for _ in range(K)]
    for j, who in enumerate(mp_obs):
        for a in who:
            kf_obs[a].append(j)
    share = np.zeros((K, K), int)
    for who in mp_obs:
        for a in who:
            for b in who:
                share[a, b] += 1
    covisible = [[int(b) for b in np.nonzero(share[a] >= 2)[0] if b != a] for a in range(K)]
    return kf_id, covisible, kf_obs, mp_id, mp_obs


def test_update_local_graph_matches_the_reference_loop():
    from se2lam_amd.mapview import updateLocalGraph
    rng = np.random.default_rng(11)
    for K, M, reach in ((6, 40, 2), (60, 900, 3), (300, 6000, 4), (40, 0, 1)):
        m = _random_map(rng, K, M, reach)
        for cur in (0, K // 2, K - 1):
            for level in (3, 1, 0):
                lk, rk, lm = updateLocalGraph(*m, cur, level)
                want = _reference_sets(*m, cur, level)
                assert lk.tolist() == want[0] and rk.tolist() == want[1] and lm.tolist() == want[2]
                assert not set(lk) & set(rk)


def test_isolated_current_key_frame_and_errors():
    import pytest
    from se2lam_amd import capi
    from se2lam_amd.mapview import updateLocalGraph
    lk, rk, lm = updateLocalGraph([5, 3], [[], []], [[0], [0]], [9], [[0, 1]], 0)
    assert lk.tolist() == [0] and rk.tolist() == [1] and lm.tolist() == [0]      # KF 1 only observes a local map point
    with pytest.raises(capi.Se2GpuError):
        updateLocalGraph([5, 3], [[], []], [[0], [0]], [9], [[0, 1]], 7)
    with pytest.raises(capi.Se2GpuError):
        updateLocalGraph([5, 3], [[4], []], [[0], [0]], [9], [[0, 1]], 0)
