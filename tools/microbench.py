#!/usr/bin/env python3
"""Per-kernel timing through the step API (HIP events inside the library): rollout, LQ, Riccati sweep, linear rollout."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--hkd", choices=["f32", "f64"], default=None, help="the kinodynamic bound-gait problem (config 5) instead of the whole-body trot")
a = ap.parse_args()
pkg = ge.load_package()
if a.hkd:
    ph = pkg.problems.hkd_bound_problem()
    s = pkg.MultiPhaseDDP(ph, batch=a.batch, precision=(pkg.PREC_F32 if a.hkd == "f32" else pkg.PREC_F64))
    s.set_initial_condition(pkg.problems.hkd_ensemble_x0(a.batch, 1, ph))
    opt = pkg.problems.hkd_ddp_setting()
else:
    ph = pkg.problems.wb_trot_problem()
    s = pkg.MultiPhaseDDP(ph, batch=a.batch)
    s.set_initial_condition(pkg.problems.wb_ensemble_x0(a.batch, 1))
    opt = pkg.mhpc_ddp_setting()
s.hybrid_rollout(0.0, opt); s.update_nominal_trajectory()
for _ in range(a.reps):
    s.LQ_approximation(opt); s.backward_sweep(0.0); s.linear_rollout(1.0, opt); s.hybrid_rollout(1.0, opt)
kt = s.kernel_times()
for k, (ms, n) in sorted(kt.items()):
    print(f"{k:18s} {ms / n:10.3f} ms/launch  ({n} launches)")
import ctypes
lib = pkg.load_hip_library()
if hasattr(lib, "hsddp_debug_sweep_prof"):
    buf = (ctypes.c_ulonglong * 48)()
    lib.hsddp_debug_sweep_prof(buf, 1)
    s.backward_sweep(0.0)
    lib.hsddp_debug_sweep_prof(buf, 0)
    for p, nm in enumerate(["phase 1", "phase 2", "K, dU", "H, G update"]):
        print(f"  waves' own time in {nm:12s}: " + "  ".join(f"w{w} {buf[16 + 4 * p + w] / 200:7.0f}" for w in range(4)) + "  cycles/knot")
    names = ["commit + next fetch", "phase 1: HA, HB, lC, lD", "phase 2: Qxx, Qux, Quu", "reg + store Qu/Quu/Qux", "chol + K, dU solves", "symmetrise Qxx", "ok check",
             "H, G update", "store K", "  LDLT: pivot order", "  LDLT: permuted row", "  LDLT: factorisation", "  LDLT: forward solves", "  LDLT: scale, backward solves, store"]
    tot = sum(buf[:16])
    for i, n in enumerate(names):
        print(f"  stamp {i} {n:26s} {buf[i] / 200:10.0f} cycles/knot ({100.0 * buf[i] / max(tot, 1):5.1f} %)")
if hasattr(lib, "hsddp_debug_quad_prof"):
    buf = (ctypes.c_ulonglong * 24)()
    names = ["first reads", "trig", "composite inertias", "D, Ct columns", "base block", "bias pass", "Jacobian", "chol3, E, Schur, chol6", "y, X", "Gram, rhs",
             "block Cholesky, lam", "qdd", "tail reads", "integrate, defect", "running cost", "constraints, barriers"]
    for label, eps_list in (("ordinary rollout (writes trajectories + cache)", None),):
        lib.hsddp_debug_quad_prof(buf, 1)
        s.hybrid_rollout(1.0, opt)
        lib.hsddp_debug_quad_prof(buf, 0)
        tot = sum(buf[:16])
        print(f"  quad kernel, {label}: {tot} cycles for one wave (16 knots)")
        for i, n in enumerate(names):
            print(f"    quad stamp {i:2d} {n:28s} {buf[i]:8d} cycles ({100.0 * buf[i] / max(tot, 1):5.1f} %)")
    # a probe launch (nothing written): through a short fixed-work solve past convergence
    s2 = pkg.MultiPhaseDDP(ph, batch=a.batch); s2.set_initial_condition(pkg.problems.wb_ensemble_x0(a.batch, 1))
    o2 = pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=9, cost_thresh=0.0)
    s2.solve(o2)
    lib.hsddp_debug_quad_prof(buf, 1)
    o3 = pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=1, cost_thresh=0.0); s2.solve(o3)
    lib.hsddp_debug_quad_prof(buf, 0)
    tot = sum(buf[:16]); kt2 = s2.kernel_times()
    print(f"  quad kernel, one more iteration past convergence (initial rollout + full step + probe launch: stamps of the same block id summed over the launches): {tot} cycles; kernel times {kt2}")
    for i, n in enumerate(names):
        print(f"    quad stamp {i:2d} {n:28s} {buf[i]:8d} cycles ({100.0 * buf[i] / max(tot, 1):5.1f} %)")
if hasattr(lib, "hsddp_debug_lq_prof") and os.environ.get("ROLL_PROF"):
    buf = (ctypes.c_ulonglong * 16)()
    lib.hsddp_debug_lq_prof(buf, 1)
    s.hybrid_rollout(1.0, opt)
    lib.hsddp_debug_lq_prof(buf, 0)
    for i, n in enumerate(["first reads, x", "terms", "lam, qdd substitution", "cache store", "integrate, constraints", "sums", "", "K store, mat-vec, select", "chol M", "X, y, Gram", "chol G"]):
        print(f"  rollout stamp {i} {n:20s} {buf[i]:10d} cycles")
elif hasattr(lib, "hsddp_debug_lq_prof"):
    buf = (ctypes.c_ulonglong * 16)()
    lib.hsddp_debug_lq_prof(buf, 1)
    s.LQ_approximation(opt)
    lib.hsddp_debug_lq_prof(buf, 0)
    names = ["terms(P pass)", "kkt_direct + keep", "dpass", "A/C columns", "store A,B,C,D", "cost partials lxx (two-wave knot: wave 1, all its cost partials)", "lu/luu/ly/lyy (two-wave knot: wave 1, kinematic round + Schur factor)",
             "kkt: select", "kkt: chol M", "kkt: X, y, gram", "kkt: chol G", "load phase"]
    tot = sum(buf)
    for i, n in enumerate(names):
        print(f"  lq stamp {i} {n:20s} {buf[i]:10d} cycles ({100.0 * buf[i] / max(tot, 1):5.1f} %)")
