"""Host-side problem builder (cafe_mpc_amd.builder): MHPCProblem::initialization + QuadReference restated, driven by the gait
files and settings the reference ships (trimmed copies under tests/golden/cafe_tree, made by tests/golden/make_gait_fixture.py)."""
import importlib
import os

import numpy as np
import pytest

from conftest import pkg, ROOT
import parity_common as pc

builder = importlib.import_module(pkg.__name__ + ".builder")
TREE = os.path.join(ROOT, "tests", "golden", "cafe_tree")


def test_quad_reference_loader_semantics():
    ref = builder.QuadReference(os.path.join(TREE, "Reference/Data/bound/quad_reference.csv"))
    assert len(ref) == 130 and ref.dt == np.float32(0.01)
    # std::stof parsing: values are float32 numbers widened to double (QuadReference.cpp:170-332)
    bs = ref.tp["body_state"]
    assert np.array_equal(bs, bs.astype(np.float32).astype(np.float64))
    # reorder_body_states: file order [eul, pos, omega, vWorld] -> [pos, eul, vWorld, omega]; the file's 6th number is the body height
    assert abs(bs[0, 2] - np.float32(0.146)) < 1e-12 or bs[0, 2] > 0.1
    ref.initialize(0.75)
    assert ref.sz == 76
    # nearest-sample lookup in float arithmetic (QuadReference.cpp:63-76)
    assert ref.index(0.0) == 0 and ref.index(0.014) == 1 and ref.index(0.016) == 2 and ref.index(10.0) == 75
    swapped = builder.QuadReference(os.path.join(TREE, "Reference/Data/bound/quad_reference.csv"), reorder=True)
    assert np.array_equal(swapped.tp["contact"], ref.tp["contact"][:, [1, 0, 3, 2]]) and not swapped.tp["jnt_vel"].any()


def test_mhpc_problem_from_shipped_bound_gait():
    phases, info, cfg = builder.build_from_tree(TREE)         # mhpc_config.info: bound, 0.25 s whole-body + 0.5 s SRB
    assert cfg["referenceFile"] == "bound" and info["horizons"] == [6, 10, 9] and info["srb_horizon"] == 10
    assert info["contacts"] == [[1, 1, 1, 1], [1, 1, 0, 0], [0, 0, 0, 0]]
    d = [p["desc"] for p in phases]
    assert [x.model for x in d] == [0, 0, 0, 1] and [x.next_model for x in d][:3] == [0, 0, 1]
    assert list(d[2].next_contact) == [0, 0, 1, 1]             # hind feet land 0.27 s into the plan: touchdown constraint + impact + projection
    assert abs(d[0].reb_torque.delta - 1.0) < 1e-15 and abs(d[0].reb_grf.eps - 0.05) < 1e-15 and d[0].al_td.sigma == 20.0
    assert list(d[0].q)[:6] == [0, 0, 10, 1, 2, 2] and abs(d[3].r[0] - 0.01) < 1e-15 and d[3].dt == 0.05
    assert np.allclose(phases[0]["bufs"]["yr"][0], [0, 0, 22.5] * 4)       # GRF reference of the first sample
    if os.path.isdir("/root/reference/Reference/Data"):        # the trimmed fixture reproduces what the full tree gives
        p2, i2, _ = builder.build_from_tree("/root/reference")
        assert i2["horizons"] == info["horizons"]
        for a, b in zip(phases, p2):
            for k in a["bufs"]:
                assert np.array_equal(a["bufs"][k], b["bufs"][k]), k


@pytest.mark.parametrize("gait", ["bound", "trot/dynfeas"])
def test_oracle_solves_shipped_gaits(oracle_lib, gait):
    phases, info, cfg = builder.build_from_tree(TREE, gait=gait, ubar_mode="gravity_comp")
    opt = builder.load_ddp_setting(os.path.join(TREE, "MHPC/settings/ddp_setting.info"))
    s = pkg.Solver(oracle_lib, phases, batch=1)
    for i, p in enumerate(phases):
        s.set_nominal(i, p["Xbar"], p["Ubar"])
    s.set_initial_condition(info["x0"][None])
    s.hybrid_rollout(0.0, opt); s.compute_cost(opt)
    f0 = s.measure_dynamics_feasibility()[0]
    s.solve(opt)
    ia = s.info_arrays()
    assert ia["status"][0] == 0 and ia["n_iters"][0] >= 2
    assert ia["dyn_feas"][0] < 0.2 * f0 and ia["max_tconstr"][0] < 5e-3


def test_hkd_problem_from_shipped_bound_gait(oracle_lib):
    """HKDProblem::initialization + HKDMPCSolver constants on the gait file HKDMPC.h:30 names, solved by the oracle; foot-placement
    extraction (HKDMPC.cpp:207-240) returns the landing spots the reset map projected on the ground."""
    ref = builder.QuadReference(os.path.join(TREE, "Reference/Data/bound/quad_reference.csv"), reorder=True)
    cpar = builder.load_hkd_constraint_params(os.path.join(TREE, "HKDMPC/settings/constraint_params.info"))
    phases, info = builder.build_hkd_problem(ref, cpar)
    assert sum(info["horizons"]) == 60 and info["contacts"][0] == [1, 1, 1, 1] and info["contacts"][1] == [1, 1, 0, 0]      # FR FL HR HL: front stance
    d0 = phases[0]["desc"]
    assert d0.model == pkg.MODEL_HKD and abs(d0.reb_grf.eps - 0.5) < 1e-15 and d0.al_td.sigma == 20.0 and abs(d0.mu - 0.7) < 1e-15
    opt = builder.load_ddp_setting(os.path.join(TREE, "HKDMPC/settings/ddp_setting.info"))
    s = pkg.Solver(oracle_lib, phases, batch=1)
    for i, p in enumerate(phases):
        s.set_nominal(i, p["Xbar"], p["Ubar"])
    s.set_initial_condition(info["x0"][None])
    s.hybrid_rollout(0.0, opt); s.compute_cost(opt); f0 = s.measure_dynamics_feasibility()[0]
    s.solve(opt)
    ia = s.info_arrays()
    # cold start on a kinematic (not dynamically consistent) reference with 5 x 10 iterations: the defects close, the plan is not converged
    assert ia["status"][0] == 0 and ia["n_iters"][0] == 50 and ia["dyn_feas"][0] < 0.2 * f0
    pf = builder.hkd_next_footholds(s, info["contacts"])
    assert set(pf) >= {2, 3} and all(abs(v[2]) < 1e-12 for v in pf.values())       # hind legs land first; cmap = (1,1,0) puts the foothold on z = 0
