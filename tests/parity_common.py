"""Shared driver: run the same HS-DDP steps on two backends of the include/hsddp.h ABI and compare fields."""
import numpy as np

STEP_FIELDS = {
    "rollout": ["X", "U", "Y", "XSIM", "DEFECT", "L", "PHI"],
    "lq": ["A", "B", "C", "D", "LX", "LU", "LY", "LXX", "LUU", "LYY", "PHIX", "PHIXX"],
    "sweep": ["K", "DU", "QU", "QUU", "QUX", "G", "H0"],
    "linear": ["DX"],
}


def make_pair(pkg, lib_a, lib_b, phases, x0, **kw):
    out = []
    for lib in (lib_a, lib_b):
        s = pkg.Solver(lib, phases, batch=x0.shape[0], **kw)
        for i, p in enumerate(phases):
            s.set_nominal(i, p["Xbar"], p["Ubar"])
        s.set_initial_condition(x0)
        out.append(s)
    return out


def compare(sa, sb, fields, nph, rtol, tag, atol_K=None):
    worst = {}
    for f in fields:
        for i in range(nph):
            a = sa.field(i, f); b = sb.field(i, f)
            if a.size == 0:
                continue
            err = np.abs(a - b).max(); sc = max(1.0, np.abs(a).max())
            worst[(f, i)] = (err, sc)
            tol = rtol * sc
            if f == "K" and atol_K is not None:
                tol = atol_K     # north_star: ||K_gpu - K_cpu||_inf < 1e-6 (absolute)
            assert err <= tol, f"{tag}: field {f} phase {i}: |diff|={err:.3e} > {tol:.3e} (scale {sc:.3e})"
    return worst


def run_steps(pkg, sa, sb, phases, opt, n_iter=2, rtol=1e-8, atol_K=1e-6):
    """Per-iterate parity: rollout -> LQ -> backward sweep -> linear rollout -> rollout(eps=1) ..."""
    nph = len(phases)
    for it in range(n_iter):
        eps = 0.0 if it == 0 else 1.0
        for s in (sa, sb):
            s.hybrid_rollout(eps, opt); s.compute_cost(opt)
        compare(sa, sb, STEP_FIELDS["rollout"], nph, rtol, f"rollout{it}")
        fa, fb = sa.measure_dynamics_feasibility(), sb.measure_dynamics_feasibility()
        rs = max(1e-10, 1e-2 * rtol)     # scalars: 1e-10 at the default per-iterate tolerance
        assert np.allclose(fa, fb, rtol=rs, atol=1e-12), (fa, fb)
        ia, ib = sa.info_arrays(), sb.info_arrays()
        assert np.allclose(ia["actual_cost"], ib["actual_cost"], rtol=rs, atol=1e-10), (ia["actual_cost"], ib["actual_cost"])
        if it == 0:
            for s in (sa, sb):
                s.update_nominal_trajectory()
        for s in (sa, sb):
            s.LQ_approximation(opt)
        compare(sa, sb, STEP_FIELDS["lq"], nph, rtol, f"lq{it}")
        oka, okb = sa.backward_sweep(0.0), sb.backward_sweep(0.0)
        assert np.array_equal(oka, okb) and oka.all()
        compare(sa, sb, STEP_FIELDS["sweep"], nph, rtol, f"sweep{it}", atol_K=atol_K)
        da, db = sa.get_exp_cost_change(), sb.get_exp_cost_change()
        assert np.allclose(da[0], db[0], rtol=1e-8, atol=1e-10) and np.allclose(da[1], db[1], rtol=1e-8, atol=1e-10)
        for s in (sa, sb):
            s.linear_rollout(1.0, opt)
        compare(sa, sb, STEP_FIELDS["linear"], nph, rtol, f"linear{it}")
        da, db = sa.get_exp_cost_change(), sb.get_exp_cost_change()
        assert np.allclose(da[0], db[0], rtol=1e-8, atol=1e-10) and np.allclose(da[1], db[1], rtol=1e-8, atol=1e-10)


def compare_solve(sa, sb, nph, rtol=1e-6, atol_K=1e-6):
    ia, ib = sa.info_arrays(), sb.info_arrays()
    for k in ("n_iters", "n_ls_iters", "n_reg_iters", "status"):
        assert np.array_equal(ia[k], ib[k]), (k, ia[k], ib[k])
    for k in ("actual_cost", "dyn_feas", "max_tconstr", "max_pconstr"):
        assert np.allclose(ia[k], ib[k], rtol=rtol, atol=1e-8), (k, ia[k], ib[k])
    compare(sa, sb, ["XBAR", "UBAR", "X", "U", "Y", "K", "DU", "QU", "QUU", "QUX"], nph, rtol, "solve", atol_K=atol_K)
