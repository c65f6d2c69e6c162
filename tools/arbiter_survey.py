#!/usr/bin/env python3
"""What the long-double arbiter of tests/parity_common.py WOULD grant on every GPU parity test that passes `exact=`: the fp64 oracle
against its own long-double build on the same problems, CPU only (no GPU backend involved: sb := sa).  The per-test caps written into
tests/test_gpu_parity.py come from this table (the largest grant per field, rounded up by about a factor 2).

    python tools/arbiter_survey.py [case ...]
"""
import ctypes, importlib, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import parity_common as pc    # noqa: E402

pkg = ge.load_package()
builder = importlib.import_module(pkg.__name__ + ".builder")
olib = pkg._abi.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle_hsddp.so")))
xlib = pkg._abi.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle_hsddp_ld.so")))
tree = os.path.join(ROOT, "tests", "golden", "cafe_tree")
BIG = {"*": 1e30}


def pair(phases, x0):
    so = pc.make_pair(pkg, olib, olib, phases, x0)[0]
    return so, pc.make_exact(pkg, xlib, phases, x0)


def solve_case(name, phases, x0, opt):
    so, sx = pair(phases, x0)
    so.solve(opt); sx.solve(opt)
    n0 = len(pc.GRANTS)
    pc.compare_solve(so, so, len(phases), exact=sx, cap=BIG, tag=name)
    return pc.GRANTS[n0:]


def cases():
    x3 = pkg.problems.wb_ensemble_x0(3, 20241222)
    def per_iterate_barrel():
        phases, xinit = pkg.problems.barrel_roll_problem(switching_times=(0.0, 0.05, 0.11, 0.18, 0.23, 0.29, 0.34))
        x0 = np.vstack([xinit, xinit + 0.01 * (x3[:2] - pkg.problems.wb_nominal_state())])
        so, sx = pair(phases, x0)
        so2 = pair(phases, x0)[0]      # (run_steps steps both of its backends: the stand-in for the GPU must be a second oracle instance)
        n0 = len(pc.GRANTS)
        pc.run_steps(pkg, so, so2, phases, pkg.problems.br_ddp_setting(), n_iter=2, exact=sx, rtol_scalar=1e-8, cap=BIG)
        return pc.GRANTS[n0:]
    yield "per_iterate[barrel_roll]", per_iterate_barrel
    for gait in ("bound", "trot/dynfeas"):
        def shipped(gait=gait):
            phases, info, cfg = builder.build_from_tree(tree, gait=gait, ubar_mode="gravity_comp")
            opt = builder.load_ddp_setting(os.path.join(tree, "MHPC/settings/ddp_setting.info"))
            x0 = np.vstack([info["x0"], info["x0"] + 0.01 * (pkg.problems.wb_ensemble_x0(2, 3) - pkg.problems.wb_nominal_state())])
            return solve_case(f"shipped[{gait}]", phases, x0, opt)
        yield f"shipped[{gait}]", shipped
    def hkd_shipped():
        ref = builder.QuadReference(os.path.join(tree, "Reference/Data/bound/quad_reference.csv"), reorder=True)
        phases, info = builder.build_hkd_problem(ref, builder.load_hkd_constraint_params(os.path.join(tree, "HKDMPC/settings/constraint_params.info")))
        opt = builder.load_ddp_setting(os.path.join(tree, "HKDMPC/settings/ddp_setting.info")); opt.max_AL_iter, opt.max_DDP_iter = 2, 4
        x0 = np.vstack([info["x0"], info["x0"]]); x0[1, :12] += 0.01
        return solve_case("hkd_shipped", phases, x0, opt)
    yield "hkd_shipped", hkd_shipped
    def mpc_loop():
        cfg = builder.load_mhpc_config(tree + "/MHPC/settings/mhpc_config.info")
        pd = builder.MHPCProblemData(builder.QuadReference(tree + "/Reference/Data/bound/quad_reference.csv"), cfg,
                                     builder.load_cost_weights(tree + "/" + cfg["costFile"]), builder.load_constraint_params(tree + "/" + cfg["constraintParamFile"]))
        opt0 = builder.load_ddp_setting(tree + "/MHPC/settings/ddp_setting.info"); opt_rt = builder.load_ddp_setting(tree + "/MHPC/settings/ddp_setting.info")
        opt_rt.max_AL_iter, opt_rt.max_DDP_iter = opt_rt.max_AL_iter_runtime, opt_rt.max_DDP_iter_runtime
        phases, info = pd.describe(ubar_mode="gravity_comp")
        x0 = np.vstack([info["x0"], info["x0"] + 0.005 * (pkg.problems.wb_ensemble_x0(1, 3)[0] - pkg.problems.wb_nominal_state())])
        so, sx = pair(phases, x0)
        so.solve(opt0); sx.solve(opt0)
        n0 = len(pc.GRANTS)
        pc.compare_solve(so, so, len(phases), exact=sx, cap=BIG, tag="mpc_loop tick 0")
        nst = int(round(float(cfg["dt_mpc"]) / cfg["dt_wb"]))
        for tick in range(1, 9):
            m = pd.update()
            xg = so.field(0, "XBAR")
            x0n = np.ascontiguousarray(xg[:, nst] if xg.shape[1] > nst else so.field(1, "XBAR")[:, nst - xg.shape[1] + 1])
            old = phases
            builder.shift_solver_in_place(sx, old, pd, m)
            phases, _ = builder.shift_solver_in_place(so, old, pd, m)
            for s_ in (so, sx):
                s_.set_initial_condition(x0n); s_.solve(opt_rt)
            pc.compare_solve(so, so, len(phases), exact=sx, cap=BIG, tag=f"mpc_loop tick {tick}")
        return pc.GRANTS[n0:]
    yield "mpc_loop", mpc_loop
    def barrel_full():
        phases, xinit = pkg.problems.barrel_roll_problem()
        x0 = np.vstack([xinit, xinit]); x0[1, 6:18] += 0.02
        return solve_case("barrel_roll_full", phases, x0, pkg.problems.br_ddp_setting(max_AL_iter=1, max_DDP_iter=4))
    yield "barrel_roll_full", barrel_full
    def zero_torque():
        phases = pkg.problems.wb_stance_problem(horizon=50, ubar_mode="zero")
        x0 = np.vstack([pkg.problems.wb_nominal_state()[None], pkg.problems.wb_ensemble_x0(2, 7)])
        return solve_case("zero_torque", phases, x0, pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=1))
    yield "zero_torque", zero_torque
    def config4():
        B = 1024
        phases, xinit = pkg.problems.barrel_roll_running_problem()
        x0 = pkg.problems.barrel_roll_ensemble_x0(B, 20241220 + 4, xinit)
        xs = np.ascontiguousarray(x0[[0, 7, B // 3]])
        return solve_case("config4", phases, xs, pkg.problems.br_ddp_setting(max_AL_iter=1, max_DDP_iter=2))
    yield "config4", config4
    for which in ("trot", "mhpc", "hkd"):
        def ss(which=which):
            if which == "hkd":
                phases = pkg.problems.hkd_trot_problem(horizons=(6, 7, 6, 5)); x0 = pkg.problems.hkd_ensemble_x0(3, 11, phases)
                opt = pkg.problems.hkd_ddp_setting(max_AL_iter=2, max_DDP_iter=3, MS=0)
            else:
                phases = pkg.problems.wb_trot_problem(horizons=(7, 6, 5, 6)) if which == "trot" else pkg.problems.mhpc_problem(wb_horizons=(7, 6), srb_horizons=(5, 4))
                x0 = pkg.problems.wb_ensemble_x0(3, 20241227); opt = pkg.mhpc_ddp_setting(max_AL_iter=2, max_DDP_iter=3, MS=0)
            so, sx = pair(phases, x0)
            n0 = len(pc.GRANTS)
            for s_ in (so, sx):
                s_.hybrid_rollout(0.0, opt); s_.compute_cost(opt); s_.update_nominal_trajectory(); s_.LQ_approximation(opt); s_.backward_sweep(0.0)
            pc.compare(so, so, pc.STEP_FIELDS["rollout"] + pc.STEP_FIELDS["lq"] + pc.STEP_FIELDS["sweep"], len(phases), 1e-8, f"ss[{which}] ss0", atol_K=1e-6, exact=sx, cap=BIG)
            for s_ in (so, sx):
                s_.hybrid_rollout(0.5, opt); s_.compute_cost(opt)
            pc.compare(so, so, pc.STEP_FIELDS["rollout"], len(phases), 1e-8, f"ss[{which}] ss1", exact=sx, cap=BIG)
            so.close(); sx.close()
            return pc.GRANTS[n0:] + solve_case(f"ss[{which}] solve", phases, x0, opt)
        yield f"single_shooting[{which}]", ss


if __name__ == "__main__":
    want = sys.argv[1:]
    for name, fn in cases():
        if want and not any(w in name for w in want):
            continue
        import io, contextlib
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            g = fn()
        flow = [l for l in buf.getvalue().splitlines() if "other decisions" in l]
        per = {}
        for (tag, f, i, tol, granted, own, err, sc) in g:
            per[f] = max(per.get(f, 0.0), granted)
        print(f"{name:32s} grants: " + (", ".join(f"{f} {v:.2e}" for f, v in sorted(per.items())) if per else "none") + ("   [" + flow[0][:160] + "]" if flow else ""), flush=True)
