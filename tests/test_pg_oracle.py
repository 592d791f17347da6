"""CPU tests of the pose-graph restatement (oracle/pg_ref.cpp: GlobalMapper::GlobalBA) against things
that are not the restatement: a numpy / scipy model of the cost (toVectorMQT through scipy's quaternions), numeric
derivatives of the edge error through the vertex update estimate * fromVectorMQT(d), the numpy plane-motion prior."""
import numpy as np
import pytest


def _cost_np(synth, g, X):
    chi, ec = 0.0, []
    for a in range(g.P):
        if g.has_prior[a]:
            e = synth.mqt_np(np.linalg.inv(g.prior_meas[a]) @ X[a])
            chi += e @ g.prior_info[a] @ e
    for k in range(g.O):
        e = synth.mqt_np(np.linalg.inv(g.o_meas[k]) @ np.linalg.inv(X[g.o_i[k]]) @ X[g.o_j[k]])
        ec.append(e @ g.o_info[k] @ e)
    return chi + sum(ec), np.array(ec)


@pytest.mark.parametrize("P", [12, 60])
def test_cost_and_convergence_against_the_numpy_model(oracle, synth, P):
    g = synth.pose_graph(P)
    c0, ec0 = oracle.pg_chi2(g)
    cn, ecn = _cost_np(synth, g, g.poses)
    assert c0 == pytest.approx(cn, rel=1e-10) and np.allclose(ec0, ecn, rtol=1e-9, atol=1e-12)
    X, ec, st = oracle.pg_optimize(g, 15)
    cn1, _ = _cost_np(synth, g, X)
    assert st["chi2_final"] == pytest.approx(cn1, rel=1e-9)
    assert st["chi2_final"] < 0.5 * st["chi2_init"] and st["chi2_final"] < 2.0 * 6 * g.O      # down to the noise floor (6 dof per edge)
    assert np.array_equal(X[0], g.poses[0])
    # the drift is taken out: positions come back to the truth within the measurement noise
    err0 = np.linalg.norm(g.poses[:, :3, 3] - g.poses_true[:, :3, 3], axis=1).max()
    err1 = np.linalg.norm(X[:, :3, 3] - g.poses_true[:, :3, 3], axis=1).max()
    assert err1 < 0.5 * err0


def test_edge_jacobians_are_the_derivatives_of_the_error(oracle, synth):
    g = synth.pose_graph(12)
    rng = np.random.default_rng(2)
    for k in (0, 5, g.O - 1):
        Xi, Xj, Z = g.poses[g.o_i[k]], g.poses[g.o_j[k]], g.o_meas[k]
        if k == 5:   # a rotation past 180 degrees between the frames: the quaternion sign normalisation is exercised
            Z = Z @ synth.from_mqt_np(np.array([0, 0, 0, 0, 0, 0.9999]))
        e, Ji, Jj = oracle.pg_edge(Xi, Xj, Z)
        assert np.allclose(e, synth.mqt_np(np.linalg.inv(Z) @ np.linalg.inv(Xi) @ Xj), atol=1e-10)
        h = 1e-6
        for c in range(6):
            d = np.zeros(6); d[c] = h
            di = (oracle.pg_edge(oracle.pg_oplus(Xi, d), Xj, Z)[0] - oracle.pg_edge(oracle.pg_oplus(Xi, -d), Xj, Z)[0]) / (2 * h)
            dj = (oracle.pg_edge(Xi, oracle.pg_oplus(Xj, d), Z)[0] - oracle.pg_edge(Xi, oracle.pg_oplus(Xj, -d), Z)[0]) / (2 * h)
            assert np.allclose(Ji[:, c], di, atol=2e-6 * max(1.0, np.abs(di).max())), (k, c)
            assert np.allclose(Jj[:, c], dj, atol=2e-6 * max(1.0, np.abs(dj).max())), (k, c)
        assert np.allclose(oracle.pg_oplus(Xi, rng.normal(0, 1e-2, 6) * 0), Xi)


def test_plane_motion_prior_matches_numpy(oracle, synth):
    g = synth.pose_graph(12)
    Tbc = np.eye(4); Tbc[:3, :3] = synth.RBC; Tbc[:3, 3] = synth.TBC
    for a in (0, 3, 11):
        m, w = oracle.pg_plane_motion_prior(g.poses[a], Tbc)
        assert np.allclose(m, g.prior_meas[a], atol=1e-9) and np.allclose(w, g.prior_info[a], rtol=1e-12, atol=1e-12)
        # the measurement is on the plane: body height 0, no roll / pitch
        Twb = m @ np.linalg.inv(Tbc)
        assert abs(Twb[2, 3]) < 1e-9 and np.allclose(Twb[2, :3], [0, 0, 1], atol=1e-12)


def test_full_system_is_consistent_with_the_cost(oracle, synth):
    """b = -J' W e is minus half the gradient of the cost along the vertex updates, H = J' W J is symmetric positive
    definite on the free poses."""
    g = synth.pose_graph(12)
    H, b = oracle.pg_system(g)
    assert np.abs(H - H.T).max() <= 1e-9 * np.abs(H).max()
    free = np.repeat(g.fixed == 0, 6)
    assert np.linalg.eigvalsh(H[free][:, free]).min() > 0
    h = 1e-6
    for a, c in ((1, 0), (4, 3), (9, 5), (7, 2)):
        d = np.zeros(6); d[c] = h
        Xp, Xm = g.poses.copy(), g.poses.copy()
        Xp[a] = oracle.pg_oplus(g.poses[a], d); Xm[a] = oracle.pg_oplus(g.poses[a], -d)
        grad = (oracle.pg_chi2(g, Xp)[0] - oracle.pg_chi2(g, Xm)[0]) / (2 * h)
        assert -2 * b[6 * a + c] == pytest.approx(grad, rel=1e-4, abs=1e-3 * np.abs(b).max())
