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


def make_exact(pkg, ld_lib, phases, x0, **kw):
    """The long-double build of the oracle on the same problem: the arbiter of the conditioning-limited cases (see compare)."""
    return make_pair(pkg, ld_lib, ld_lib, phases, x0, **kw)[0]


COND_FACTOR = 10.0     # rounding-error realisations of two fp64 implementations of one algebra scatter by about this much


def compare(sa, sb, fields, nph, rtol, tag, atol_K=None, exact=None):
    """sa: fp64 oracle, sb: backend under test.  Every field must agree to rtol x its scale (K: atol_K absolute, north_star's
    1e-6).  `exact` (optional): the long-double oracle run through the same steps.  Where the problem is so badly conditioned that
    the fp64 ORACLE ITSELF sits farther than the tolerance from the exact iterate, no fp64 implementation can meet the tolerance
    against it; the bound is then COND_FACTOR x the oracle's own distance from the exact iterate (measured here, per field)."""
    worst = {}
    for f in fields:
        for i in range(nph):
            a = sa.field(i, f); b = sb.field(i, f)
            if a.size == 0:
                continue
            err = np.abs(a - b).max(); sc = max(1.0, np.abs(a).max())
            tol = rtol * sc
            if f == "K" and atol_K is not None:
                tol = atol_K     # north_star: ||K_gpu - K_cpu||_inf < 1e-6 (absolute)
            own = None
            if exact is not None:
                own = np.abs(a - exact.field(i, f)).max()
                tol = max(tol, COND_FACTOR * own)
            worst[(f, i)] = (err, sc, own)
            assert err <= tol, f"{tag}: field {f} phase {i}: |diff|={err:.3e} > {tol:.3e} (scale {sc:.3e}, oracle-vs-exact {own})"
    return worst


def run_steps(pkg, sa, sb, phases, opt, n_iter=2, rtol=1e-8, atol_K=1e-6, exact=None, rtol_scalar=None):
    """Per-iterate parity: rollout -> LQ -> backward sweep -> linear rollout -> rollout(eps=1) ..."""
    nph = len(phases)
    every = (sa, sb) if exact is None else (sa, sb, exact)
    for it in range(n_iter):
        eps = 0.0 if it == 0 else 1.0
        for s in every:
            s.hybrid_rollout(eps, opt); s.compute_cost(opt)
        compare(sa, sb, STEP_FIELDS["rollout"], nph, rtol, f"rollout{it}", exact=exact)
        fa, fb = sa.measure_dynamics_feasibility(), sb.measure_dynamics_feasibility()
        rs = rtol_scalar if rtol_scalar is not None else max(1e-10, 1e-2 * rtol)     # scalars: 1e-10 at the default per-iterate tolerance
        if exact is not None:
            rs = max(rs, COND_FACTOR * float(np.max(np.abs(fa - exact.measure_dynamics_feasibility()) / np.maximum(np.abs(fa), 1e-300))))
        assert np.allclose(fa, fb, rtol=rs, atol=1e-12), (fa, fb)
        ia, ib = sa.info_arrays(), sb.info_arrays()
        rc = rtol_scalar if rtol_scalar is not None else max(1e-10, 1e-2 * rtol)
        if exact is not None:
            ix = exact.info_arrays()
            rc = max(rc, COND_FACTOR * float(np.max(np.abs(ia["actual_cost"] - ix["actual_cost"]) / np.maximum(np.abs(ia["actual_cost"]), 1e-300))))
        assert np.allclose(ia["actual_cost"], ib["actual_cost"], rtol=rc, atol=1e-10), (ia["actual_cost"], ib["actual_cost"])
        if it == 0:
            for s in every:
                s.update_nominal_trajectory()
        for s in every:
            s.LQ_approximation(opt)
        compare(sa, sb, STEP_FIELDS["lq"], nph, rtol, f"lq{it}", exact=exact)
        oks = [s.backward_sweep(0.0) for s in every]
        assert np.array_equal(oks[0], oks[1]) and oks[0].all()
        compare(sa, sb, STEP_FIELDS["sweep"], nph, rtol, f"sweep{it}", atol_K=atol_K, exact=exact)

        def dv_close(tag):
            da, db = sa.get_exp_cost_change(), sb.get_exp_cost_change()
            for q in (0, 1):
                r = 1e-8
                if exact is not None:
                    dx = exact.get_exp_cost_change()
                    r = max(r, COND_FACTOR * float(np.max(np.abs(da[q] - dx[q]) / np.maximum(np.abs(da[q]), 1e-300))))
                assert np.allclose(da[q], db[q], rtol=r, atol=1e-10), (tag, q, da[q], db[q])
        dv_close("sweep")
        for s in every:
            s.linear_rollout(1.0, opt)
        compare(sa, sb, STEP_FIELDS["linear"], nph, rtol, f"linear{it}", exact=exact)
        dv_close("linear")


def compare_solve(sa, sb, nph, rtol=1e-6, atol_K=1e-6, exact=None):
    ia, ib = sa.info_arrays(), sb.info_arrays()
    for k in ("n_iters", "n_ls_iters", "n_reg_iters", "status"):
        assert np.array_equal(ia[k], ib[k]), (k, ia[k], ib[k])
    ix = exact.info_arrays() if exact is not None else None
    for k in ("actual_cost", "dyn_feas", "max_tconstr", "max_pconstr"):
        r, at = rtol, 1e-8
        if ix is not None:
            own = np.abs(ia[k] - ix[k]); at = max(at, COND_FACTOR * float(own.max()))
        assert np.allclose(ia[k], ib[k], rtol=r, atol=at), (k, ia[k], ib[k])
    return compare(sa, sb, ["XBAR", "UBAR", "X", "U", "Y", "K", "DU", "QU", "QUU", "QUX"], nph, rtol, "solve", atol_K=atol_K, exact=exact)
