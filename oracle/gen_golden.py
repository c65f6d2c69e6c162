"""ORACLE tooling: generate tests/golden/casadi_ref.npz — input/output vectors of the reference's
CasADi-generated functions at seeded random inputs.  Run in the build container (needs oracle/_ref, i.e.
/root/reference):   python oracle/gen_golden.py
The fixture holds DATA only (inputs + the reference code's outputs); no reference source is copied."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_casadi  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(n=16, seed=20241220):
    R = ref_casadi.Ref()
    rng = np.random.default_rng(seed)
    out = {}
    q = rng.uniform(-1, 1, (n, 18)); v = rng.uniform(-2, 2, (n, 18)); a = rng.uniform(-20, 20, (n, 18)); F = rng.uniform(-30, 30, (n, 12))
    out.update(wb_q=q, wb_v=v, wb_a=a, wb_F=F)
    out["footVelPartialDq"] = np.array([R.call("footVelPartialDq", q[i], v[i]) for i in range(n)])
    out["footAccPartialDq"] = np.array([R.call("footAccPartialDq", q[i], v[i], a[i]) for i in range(n)])
    out["footAccPartialDv"] = np.array([R.call("footAccPartialDv", q[i], v[i], a[i]) for i in range(n)])
    out["footForcePartialDq"] = np.array([R.call("footForcePartialDq", q[i], F[i]) for i in range(n)])
    x = rng.uniform(-0.5, 0.5, (n, 12)); x[:, 2] += 0.3; u = rng.uniform(-20, 40, (n, 12)); pf = rng.uniform(-0.3, 0.3, (n, 12))
    c = rng.integers(0, 2, (n, 4)).astype(np.float64)
    out.update(srb_x=x, srb_u=u, srb_pf=pf, srb_c=c)
    out["SRBDynamics"] = np.array([R.call("SRBDynamics", x[i], u[i], pf[i], c[i])[0].ravel() for i in range(n)])
    d = [R.call("SRBDynamicsDerivatives", x[i], u[i], pf[i], c[i]) for i in range(n)]
    out["SRB_Ac"] = np.array([k[0] for k in d]); out["SRB_Bc"] = np.array([k[1] for k in d])
    xh = rng.uniform(-0.4, 0.4, (n, 24)); uh = rng.uniform(-10, 30, (n, 24)); dt = np.full((n, 1), 0.01)
    out.update(hkd_x=xh, hkd_u=uh, hkd_dt=dt, hkd_c=c)
    out["hkinodyn"] = np.array([R.call("hkinodyn", xh[i], uh[i], dt[i], c[i])[0].ravel() for i in range(n)])
    d = [R.call("hkinodyn_par", xh[i], uh[i], dt[i], c[i]) for i in range(n)]
    out["hkd_A"] = np.array([k[0] for k in d]); out["hkd_B"] = np.array([k[1] for k in d])
    pos = rng.uniform(-0.2, 0.2, (n, 3)); eul = rng.uniform(-0.5, 0.5, (n, 3)); ql = rng.uniform(-1.5, 1.5, (n, 3))
    out.update(fk_pos=pos, fk_eul=eul, fk_qleg=ql)
    out["compute_foot_position"] = np.array([[R.call("compute_foot_position", pos[i], eul[i], ql[i], np.array([float(l)]))[0].ravel() for l in (1, 2, 3, 4)] for i in range(n)])
    out["comp_foot_jacob"] = np.array([[R.call(f"comp_foot_jacob_{l}", pos[i], eul[i], ql[i])[0] for l in (1, 2, 3, 4)] for i in range(n)])
    path = os.path.join(ROOT, "tests", "golden", "casadi_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
