"""CPU test: the HIP kernel programs (cafe-mpc_amd/csrc/wb_knot.hpp, sweep.hpp) compiled for the host by the
test-only lane emulator tests/_emu, checked against the oracle.  Catches indexing / phase-order / per-lane math
errors without a GPU; real HIP execution is covered by the -m gpu tests."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import pkg, ROOT
import parity_common as pc


@pytest.fixture(scope="module")
def emu_lib():
    d = os.path.join(ROOT, "tests", "_emu")
    subprocess.check_call(["make", "-C", d, "-s"])
    return pkg._abi.bind(ctypes.CDLL(os.path.join(d, "libhsddp_emu.so")))


@pytest.mark.parametrize("program", ["wave", "quad"])
@pytest.mark.parametrize("which", ["stance", "trot", "mhpc", "srb_only", "barrel_roll", "hkd", "mpc_tick"])
def test_kernel_programs_match_oracle(emu_lib, oracle_lib, oracle_ld_lib, which, program, monkeypatch):
    """program: which rollout program evaluates the whole-body running knots - the one-wave knot (wb_knot.hpp) or the lane-quad knot
    (wb_quad.hpp: one lane per leg; here its four lanes run as a four-wide value).  Both write the trajectories AND the contact-solve cache the
    LQ knot of the next step fetches, so the per-iterate comparison covers the cache layout too."""
    if program == "quad":
        if which in ("srb_only", "hkd"):
            pytest.skip("no whole-body knot in this problem")
        monkeypatch.setenv("HSDDP_EMU_QUAD", "1")
    if which == "mhpc":    # whole-body phases + single-rigid-body tail: mixed state dimension across the phase boundary
        phases = pkg.problems.mhpc_problem(wb_horizons=(4, 3), srb_horizons=(3, 2))
    else:
        phases = pkg.problems.wb_stance_problem(horizon=5) if which == "stance" else pkg.problems.wb_trot_problem(horizons=(4, 3, 3, 3))
    x0 = pkg.problems.wb_ensemble_x0(2, 20241222)
    if which == "srb_only":
        phases = pkg.problems.mhpc_problem(wb_schedule=(), wb_horizons=(), srb_horizons=(4, 3))
        x0 = np.ascontiguousarray(x0[:, list(range(6)) + list(range(18, 24))])
    opt = pkg.mhpc_ddp_setting()
    if which == "barrel_roll":   # BarrelRollTO.cpp shape at short phase durations: flight phases, 4-foot touchdown, joint-speed barrier
        phases, xinit = pkg.problems.barrel_roll_problem(switching_times=(0.0, 0.03, 0.06, 0.10, 0.13, 0.16, 0.19))
        x0 = np.vstack([xinit, xinit + 0.01 * (x0[0] - pkg.problems.wb_nominal_state())])
        opt = pkg.problems.br_ddp_setting()
    if which == "mpc_tick":      # window after one receding-horizon update: a young single-shooting phase (h = 1) in front of the SRB tail
        import importlib, os
        builder = importlib.import_module(pkg.__name__ + ".builder")
        tree = os.path.join(ROOT, "tests", "golden", "cafe_tree")
        cfg = builder.load_mhpc_config(tree + "/MHPC/settings/mhpc_config.info")
        pd = builder.MHPCProblemData(builder.QuadReference(tree + "/Reference/Data/bound/quad_reference.csv"), cfg,
                                     builder.load_cost_weights(tree + "/" + cfg["costFile"]), builder.load_constraint_params(tree + "/" + cfg["constraintParamFile"]))
        pd.update()
        phases, info = pd.describe(ubar_mode="gravity_comp")
        assert info["shooting"] == [1, 1, 1, 0] and info["horizons"] == [4, 10, 10, 1]
        x0 = np.vstack([info["x0"], info["x0"] + 0.01 * (x0[0] - pkg.problems.wb_nominal_state())])
    if which == "hkd":           # HKD-MPC trot: 24/24/0 phases, lift-off / touchdown reset maps, touchdown constraint from leg kinematics
        phases = pkg.problems.hkd_trot_problem(horizons=(3, 4, 3, 3))
        x0 = pkg.problems.hkd_ensemble_x0(2, 11, phases)
        opt = pkg.problems.hkd_ddp_setting()
    so, se = pc.make_pair(pkg, oracle_lib, emu_lib, phases, x0)
    # the barrel-roll iterate after a full step from the zero-torque start is badly conditioned: the fp64 oracle itself sits ~1e-5 from
    # the exact (long-double) gains at |K| ~ 650, so the long-double run arbitrates there (parity_common.compare)
    exact = pc.make_exact(pkg, oracle_ld_lib, phases, x0) if which == "barrel_roll" else None
    pc.run_steps(pkg, so, se, phases, opt, n_iter=2, rtol=1e-8, exact=exact, rtol_scalar=1e-8 if exact is not None else None,
                 cap={"K": 1.5e-5, "DU": 1.2e-5, "QU": 2e-4, "G": 3e-3, "H0": 8e-5, "DX": 3e-5})      # measured grants of this iterate x 2 (|K| = 646, |G| = 1.2e5)


def test_one_wave_lq_variant_matches_oracle(oracle_lib):
    """The LQ knot also exists as a one-wave program (LQ_NT=64: tangent rounds one after the other, no wave-level phase sequences);
    the product builds the two-wave one (LQ_NT=128, what every other test here emulates)."""
    d = os.path.join(ROOT, "tests", "_emu")
    subprocess.check_call(["make", "-C", d, "-s", "libhsddp_emu64.so"])
    emu64 = pkg._abi.bind(ctypes.CDLL(os.path.join(d, "libhsddp_emu64.so")))
    phases = pkg.problems.wb_trot_problem(horizons=(4, 3, 3, 3))
    x0 = pkg.problems.wb_ensemble_x0(2, 20241222)
    so, se = pc.make_pair(pkg, oracle_lib, emu64, phases, x0)
    pc.run_steps(pkg, so, se, phases, pkg.mhpc_ddp_setting(), n_iter=2, rtol=1e-8)


@pytest.mark.parametrize("which", ["trot", "mhpc", "hkd"])
def test_single_shooting_programs_match_oracle(emu_lib, oracle_lib, which):
    """option.MS = false: the single-shooting chain (one wave walks every phase, SRB / HKD knots included) against the oracle."""
    if which == "hkd":
        phases = pkg.problems.hkd_trot_problem(horizons=(3, 4, 3, 3)); x0 = pkg.problems.hkd_ensemble_x0(2, 11, phases)
        opt = pkg.problems.hkd_ddp_setting(MS=0)
    else:
        phases = pkg.problems.wb_trot_problem(horizons=(4, 3, 3, 3)) if which == "trot" else pkg.problems.mhpc_problem(wb_horizons=(4, 3), srb_horizons=(3, 2))
        x0 = pkg.problems.wb_ensemble_x0(2, 20241227); opt = pkg.mhpc_ddp_setting(MS=0)
    so, se = pc.make_pair(pkg, oracle_lib, emu_lib, phases, x0)
    for s_ in (so, se):
        s_.hybrid_rollout(0.0, opt); s_.compute_cost(opt); s_.update_nominal_trajectory(); s_.LQ_approximation(opt)
        assert s_.backward_sweep(0.0).all()
    pc.compare(so, se, pc.STEP_FIELDS["rollout"] + pc.STEP_FIELDS["lq"] + pc.STEP_FIELDS["sweep"], len(phases), 1e-8, "ss0", atol_K=1e-6)
    assert np.abs(se.field(0, "DEFECT")).max() == 0.0
    for s_ in (so, se):
        s_.hybrid_rollout(0.5, opt); s_.compute_cost(opt)
    pc.compare(so, se, pc.STEP_FIELDS["rollout"], len(phases), 1e-8, "ss1")


def test_quad_probe_matches_the_one_wave_knot_slot_by_slot(emu_lib):
    """The lane-quad program as a line-search PROBE (nothing written) against the one-wave knot: per (problem, knot) the three merit partials
    cost, squared defect, min g at several step lengths, on schedules with 2 / 3 / 4 / 0 contact feet and the barrel roll's joint-speed barrier."""
    raw = ctypes.CDLL(os.path.join(ROOT, "tests", "_emu", "libhsddp_emu.so"))

    def partials(s, fn, *a):
        nsl = sum(p["desc"].horizon + 1 for p in s.phases)
        out = np.zeros((s.batch, nsl, 3))
        assert getattr(raw, fn)(s.h, *a, out.ctypes.data_as(ctypes.c_void_p)) == 0
        return out
    x2 = pkg.problems.wb_ensemble_x0(2, 20241222)
    cases = [(pkg.problems.wb_trot_problem(horizons=(4, 3, 3, 3)), x2, pkg.mhpc_ddp_setting()),
             (pkg.problems.wb_trot_problem(schedule=((1, 1, 1, 1), (0, 0, 0, 0), (1, 1, 0, 1)), horizons=(3, 3, 3), last_next=(1, 1, 1, 1)), x2, pkg.mhpc_ddp_setting())]
    phases, xinit = pkg.problems.barrel_roll_problem(switching_times=(0.0, 0.03, 0.06, 0.10, 0.13, 0.16, 0.19))
    cases.append((phases, np.vstack([xinit, xinit + 0.01 * (x2[0] - pkg.problems.wb_nominal_state())]), pkg.problems.br_ddp_setting()))
    for phases, x0, opt in cases:
        s = pkg.Solver(emu_lib, phases, batch=2)
        for i, p in enumerate(phases):
            s.set_nominal(i, p["Xbar"], p["Ubar"])
        s.set_initial_condition(x0)
        for it, eps in enumerate((0.0, 1.0, 0.25)):
            s.hybrid_rollout(eps, opt)
            a = partials(s, "hsddp_debug_slot_partials")
            b = partials(s, "hsddp_debug_quad_probe", ctypes.c_double(eps), ctypes.byref(opt))
            m = ~np.isnan(b[..., 0])
            assert m.sum() == 2 * sum(p["desc"].horizon for p in phases)
            for q in range(3):
                assert np.abs(a[..., q][m] - b[..., q][m]).max() <= 1e-11 * max(1.0, np.abs(a[..., q][m]).max()), (it, q)
            if it == 0:
                s.update_nominal_trajectory()
            s.LQ_approximation(opt); assert s.backward_sweep(0.0).all(); s.linear_rollout(1.0, opt)
        s.close()
