"""CPU tests of the oracle's solver algebra (the reference holds no stored outputs for it: "parity unpinned" by
fixtures, so it is cross-checked here against an independent numpy restatement of SinglePhase.cpp:323-391 /
145-178 and by the properties the algorithm must have)."""
import numpy as np
import pytest

from conftest import pkg


def _solver(oracle_lib, phases, x0):
    s = pkg.Solver(oracle_lib, phases, batch=x0.shape[0])
    for i, p in enumerate(phases):
        s.set_nominal(i, p["Xbar"], p["Ubar"])
    s.set_initial_condition(x0)
    return s


def numpy_backward_sweep(s, nph, reg=0.0):
    """Independent Riccati recursion (numpy) over the oracle's own LQ data, phases last -> first, with the
    impact-aware step (MultiPhaseDDP.cpp:196-201) taken from the oracle's dX/Px-free identity: here single phase only."""
    assert nph == 1
    A = s.field(0, "A")[0]; B = s.field(0, "B")[0]; C = s.field(0, "C")[0]; D = s.field(0, "D")[0]
    lx = s.field(0, "LX")[0]; lu = s.field(0, "LU")[0]; ly = s.field(0, "LY")[0]
    lxx = s.field(0, "LXX")[0]; luu = s.field(0, "LUU")[0]; lyy = s.field(0, "LYY")[0]
    Df = s.field(0, "DEFECT")[0]
    G = s.field(0, "PHIX")[0, 0].copy(); H = s.field(0, "PHIXX")[0, 0].copy()
    h = A.shape[0]; Ks = np.zeros((h, 12, 36)); dUs = np.zeros((h, 12)); dV = 0.0
    for k in range(h - 1, -1, -1):
        Gn = G + H @ Df[k + 1]
        Qx = lx[k] + A[k].T @ Gn + C[k].T @ ly[k]
        Qu = lu[k] + B[k].T @ Gn + D[k].T @ ly[k]
        Qxx = lxx[k] + A[k].T @ H @ A[k] + C[k].T @ lyy[k] @ C[k] + reg * np.eye(36)
        Quu = luu[k] + B[k].T @ H @ B[k] + D[k].T @ lyy[k] @ D[k] + reg * np.eye(12)
        Qux = B[k].T @ H @ A[k] + D[k].T @ lyy[k] @ C[k]
        Qi = np.linalg.inv(Quu - 1e-9 * np.eye(12))
        Qxx = 0.5 * (Qxx + Qxx.T)
        dUs[k] = -Qi @ Qu; Ks[k] = -Qi @ Qux
        G = Qx - Qux.T @ Qi @ Qu; H = Qxx - Qux.T @ Qi @ Qux
        dV += Qu @ dUs[k]
    return Ks, dUs, dV


def test_riccati_against_numpy(oracle_lib):
    phases = pkg.problems.wb_stance_problem(horizon=10)
    s = _solver(oracle_lib, phases, pkg.problems.wb_ensemble_x0(1, 3))
    opt = pkg.mhpc_ddp_setting()
    s.hybrid_rollout(0.0, opt); s.update_nominal_trajectory(); s.compute_cost(opt); s.LQ_approximation(opt)
    assert s.backward_sweep(0.0).all()
    Ks, dUs, dV = numpy_backward_sweep(s, 1)
    assert np.abs(s.field(0, "K")[0] - Ks).max() < 1e-8
    assert np.abs(s.field(0, "DU")[0] - dUs).max() < 1e-9
    dV1, dV2 = s.get_exp_cost_change()
    assert abs(dV1[0] - dV) < 1e-9 * max(1, abs(dV)) and abs(dV2[0] + dV) < 1e-9 * max(1, abs(dV))


def test_linear_rollout_closes_defects_to_first_order(oracle_lib):
    """dX from linear_rollout(1) is the Newton step of the multiple-shooting defects: after the full step the
    dynamics infeasibility must drop several-fold on a mild problem (the remainder is the nonlinearity)."""
    phases = pkg.problems.wb_stance_problem(horizon=20)
    s = _solver(oracle_lib, phases, pkg.problems.wb_ensemble_x0(1, 5))
    opt = pkg.mhpc_ddp_setting()
    s.hybrid_rollout(0.0, opt); s.update_nominal_trajectory(); s.compute_cost(opt)
    f0 = s.measure_dynamics_feasibility()[0]
    s.LQ_approximation(opt); s.backward_sweep(0.0); s.linear_rollout(1.0, opt); s.hybrid_rollout(1.0, opt)
    f1 = s.measure_dynamics_feasibility()[0]
    assert f1 < 0.3 * f0


def test_solve_converges_and_al_reduces_touchdown_violation(oracle_lib):
    phases = pkg.problems.wb_trot_problem(horizons=(15, 15, 15, 15))
    s = _solver(oracle_lib, phases, pkg.problems.wb_ensemble_x0(2, 20241222))
    opt1 = pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=6)
    s.solve(opt1); a = s.info_arrays()
    s2 = _solver(oracle_lib, phases, pkg.problems.wb_ensemble_x0(2, 20241222))
    s2.solve(pkg.mhpc_ddp_setting(max_AL_iter=5, max_DDP_iter=6)); b = s2.info_arrays()
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    assert (b["dyn_feas"] < 1e-2).all()
    assert (b["max_tconstr"] < a["max_tconstr"]).all()          # AL outer loop tightens the touchdown constraint
    assert (b["max_pconstr"] > -1e-3).all()                      # path constraints hold (ReB)


def test_zero_torque_start_needs_backtracking(oracle_lib):
    """BASELINE config 1 literal (Ubar = 0): the first line search needs several trials (alpha = 0.5)."""
    phases = pkg.problems.wb_stance_problem(horizon=50, ubar_mode="zero")
    s = _solver(oracle_lib, phases, pkg.problems.wb_nominal_state()[None])
    s.solve(pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=1))
    info = s.info_arrays()
    assert info["n_iters"][0] == 1 and info["n_ls_iters"][0] >= 3


def test_regularisation_retry_on_indefinite_quu(oracle_lib):
    """Negative control weights make Quu indefinite at reg = 0: backward_sweep must fail there and the
    regularised sweep must raise reg until it passes (MultiPhaseDDP.cpp:136-165)."""
    phases = pkg.problems.wb_stance_problem(horizon=6)
    for i in range(12):
        phases[0]["desc"].r[i] = -50.0
    s = _solver(oracle_lib, phases, pkg.problems.wb_ensemble_x0(1, 9))
    opt = pkg.mhpc_ddp_setting(ReB_active=0)
    s.hybrid_rollout(0.0, opt); s.update_nominal_trajectory(); s.compute_cost(opt); s.LQ_approximation(opt)
    assert not s.backward_sweep(0.0).any()
    assert s.backward_sweep(10.0).all()
    s.solve(pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=1, ReB_active=0))
    assert s.info_arrays()["n_reg_iters"][0] > 1
