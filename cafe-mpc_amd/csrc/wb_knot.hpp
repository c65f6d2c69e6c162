// Whole-body per-knot programs: ONE WAVEFRONT PER KNOT (workgroup = 64 lanes), knot state in LDS.
//
//   wb_rollout_knot     replaces SinglePhase::hybrid_rollout body (SinglePhase.cpp:197-224) + compute_defect
//                       (TrajectoryManagement.cpp:231) + compute_cost running part (SinglePhase.cpp:240-251)
//                       -> WBM::dynamics / KKTContactDynamics (WBM.cpp:17-57, 368-424)
//   wb_rollout_terminal terminal constraint/cost (SinglePhase.cpp:227, 254-261) + reset map to the next phase
//                       (MultiPhaseDDP.cpp:70-78, MHPCReset.cpp:4-28, WBM::impact WBM.cpp:178-206,427-456)
//   wb_lq_knot          SinglePhase::LQ_approximation knot body (SinglePhase.cpp:287-306):
//                       WBM::dynamics_partial (WBM.cpp:60-139, 459-505) + cost partials (MHPCCost.cpp) + ReB fold
//   wb_lq_terminal      terminal partials + AL (SinglePhase.cpp:311-319) + reset-map partial Px
//                       (MHPCReset.cpp:31-52, WBM::impact_partial WBM.cpp:225-254, 508-543)
// Multiple shooting with every knot a shooting node (update_SS_config(h+1), MHPCProblem.cpp:209) makes all
// knots independent in the rollout, which is what lets the grid be batch x knots.
#pragma once
#include "hs_types.hpp"
#include "wb_model.hpp"

namespace hs {

struct WbLds {
    double x[36], xb[36], u[12], acc[18], tau[18], fext[12];
    double M[18 * 18], Lm[18 * 18], Minv[18 * 18], h[18];
    double Jall[12 * 18], Jdv[12], fpos[12], fvel[12];
    double Jc[12 * 18], gam[12];
    double Xm[18 * 12], Ym[18 * 12], G[144], LG[144], Lam[144];
    double a0[18], rhs[12], lam[12], qdd[18], grf[12], tmp[64];
    double Kinv[30 * 30];
    double dtau[18 * 64], dacc[12 * 64], dvel[12 * 64];
    double out[36 * 36];
    double Jv[12 * 18];
    double gval[MAXG], bar[MAXG], bd[MAXG], bdd[MAXG];
    double red[64];
    int flag;
};

// ---- wave-cooperative dense helpers (row-major n x n in LDS) -------------------------------------------
// Cholesky A = L L^T (lower, row-major ld), left-looking; tmp holds the column being built.  Every lane calls.
template <int NT>
HD void chol_lds(const double* A, double* Lo, int n, int ld, double* tmp, double diag_add) {
    for (int j = 0; j < n; j++) {
        HS_PHASE(NT, if (tid >= j && tid < n) {
            double s = A[tid * ld + j] + ((tid == j) ? diag_add : 0.0);
            for (int k = 0; k < j; k++) s -= Lo[tid * ld + k] * Lo[j * ld + k];
            tmp[tid] = s;
        })
        HS_PHASE(NT, if (tid >= j && tid < n) {
            double d = sqrt(tmp[j]);
            Lo[tid * ld + j] = (tid == j) ? d : tmp[tid] / d;
        })
    }
}
// column c of the inverse from a Cholesky factor, written into Inv[:, c] (row-major ld). Called by lane c.
HD void chol_inv_col(const double* Lo, double* Inv, int n, int ld, int c) {
    for (int i = 0; i < n; i++) {
        double s = (i == c) ? 1.0 : 0.0;
        for (int k = 0; k < i; k++) s -= Lo[i * ld + k] * Inv[k * ld + c];
        Inv[i * ld + c] = s / Lo[i * ld + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = Inv[i * ld + c];
        for (int k = i + 1; k < n; k++) s -= Lo[k * ld + i] * Inv[k * ld + c];
        Inv[i * ld + c] = s / Lo[i * ld + i];
    }
}

HD LaneCfg lane_cfg(const ModelDev& md, bool kin, double mscale, double grav, double fscale, double vscale, double ascale, int aunit, int tq, int tv) {
    LaneCfg c;
    c.cpsi = kin ? md.cpsi_kin : md.cpsi_dyn; c.spsi = kin ? md.spsi_kin : md.spsi_dyn;
    c.mscale = mscale; c.grav = grav; c.fscale = fscale; c.vscale = vscale; c.ascale = ascale; c.aunit = aunit; c.tq = tq; c.tv = tv;
    return c;
}

// Phase: M, h, all-foot Jacobians, Jdot*v, foot pos/vel at L.x (psi_dyn).  lanes 0..17: columns, lane 18: bias terms.
template <int NT>
HD void wb_terms(WbLds& L, const ModelDev& md, bool need_cols) {
    HS_PHASE(NT, if (tid < 19 && (need_cols || tid == 18)) {
        LaneCfg c = (tid < 18) ? lane_cfg(md, false, 1.0, 0.0, 0.0, 0.0, 0.0, tid, -1, -1)
                               : lane_cfg(md, false, 1.0, GRAV, 0.0, 1.0, 0.0, -1, -1, -1);
        PassOut<double> o;
        wb_pass<double>(c, L.x, L.x + 18, L.acc, L.fext, o);
        if (tid < 18) {
            for (int i = 0; i < 18; i++) L.M[i * 18 + tid] = o.tau[i];
            for (int f = 0; f < 4; f++) { L.Jall[(3 * f) * 18 + tid] = o.facc[f].x; L.Jall[(3 * f + 1) * 18 + tid] = o.facc[f].y; L.Jall[(3 * f + 2) * 18 + tid] = o.facc[f].z; }
        } else {
            for (int i = 0; i < 18; i++) L.h[i] = o.tau[i];
            for (int f = 0; f < 4; f++) {
                L.Jdv[3 * f] = o.facc[f].x; L.Jdv[3 * f + 1] = o.facc[f].y; L.Jdv[3 * f + 2] = o.facc[f].z;
                L.fpos[3 * f] = o.fpos[f].x; L.fpos[3 * f + 1] = o.fpos[f].y; L.fpos[3 * f + 2] = o.fpos[f].z;
                L.fvel[3 * f] = o.fvel[f].x; L.fvel[3 * f + 1] = o.fvel[f].y; L.fvel[3 * f + 2] = o.fvel[f].z;
            }
        }
    })
}

// Phase group: contact KKT solve.  mode 0: forward dynamics (gam = Jdv + 2 alpha J v, damping 1e-12, rhs uses tau-h)
//                                  mode 1: impulse (rhs = -J v, damping 0, "a0" = v)
// Results: L.qdd (or v+), L.lam (compact), L.grf (scattered 12).  need_kinv: also L.Kinv ((18+m)^2, ld 30, damping 0).
template <int NT>
HD void wb_kkt(WbLds& L, int nc, const int* feet, int mode, double alpha, bool need_kinv) {
    const int m = 3 * nc;
    HS_PHASE(NT, if (tid < m) {
        int f = feet[tid / 3], r = tid % 3;
        for (int j = 0; j < 18; j++) L.Jc[tid * 18 + j] = L.Jall[(3 * f + r) * 18 + j];
        L.gam[tid] = (mode == 0) ? (L.Jdv[3 * f + r] + 2.0 * alpha * L.fvel[3 * f + r]) : 0.0;
    })
    chol_lds<NT>(L.M, L.Lm, 18, 18, L.tmp, 0.0);
    HS_PHASE(NT, if (tid < 18) chol_inv_col(L.Lm, L.Minv, 18, 18, tid);)
    HS_PHASE(NT, if (tid < 18) {
        double s = 0;
        if (mode == 0) { for (int j = 0; j < 18; j++) s += L.Minv[tid * 18 + j] * (L.tau[j] - L.h[j]); }
        else s = L.x[18 + tid];
        L.a0[tid] = s;
    } if (tid < 12) { L.grf[tid] = 0.0; L.lam[tid] = 0.0; })
    if (m > 0) {
        HS_PHASE(NT, if (tid < m) {
            for (int i = 0; i < 18; i++) { double s = 0; for (int j = 0; j < 18; j++) s += L.Minv[i * 18 + j] * L.Jc[tid * 18 + j]; L.Xm[i * 12 + tid] = s; }
            for (int a = 0; a < m; a++) { double s = 0; for (int i = 0; i < 18; i++) s += L.Jc[a * 18 + i] * L.Xm[i * 12 + tid]; L.G[a * 12 + tid] = s; }
            double s = 0; for (int i = 0; i < 18; i++) s += L.Jc[tid * 18 + i] * L.a0[i];
            L.rhs[tid] = -s - L.gam[tid];
        })
        chol_lds<NT>(L.G, L.LG, m, 12, L.tmp, (mode == 0) ? 1e-12 : 0.0);
        HS_PHASE(NT, if (tid == 0) {   // lambda = G^-1 rhs  (forward / backward substitution, m <= 12)
            for (int i = 0; i < m; i++) { double s = L.rhs[i]; for (int k = 0; k < i; k++) s -= L.LG[i * 12 + k] * L.lam[k]; L.lam[i] = s / L.LG[i * 12 + i]; }
            for (int i = m - 1; i >= 0; i--) { double s = L.lam[i]; for (int k = i + 1; k < m; k++) s -= L.LG[k * 12 + i] * L.lam[k]; L.lam[i] = s / L.LG[i * 12 + i]; }
        })
    }
    HS_PHASE(NT, if (tid < 18) {
        double s = L.a0[tid];
        // qdd = Minv (tau - h + J^T lam) = a0 + (Minv J^T) lam   (Pinocchio forwardDynamics / impulseDynamics)
        for (int a = 0; a < m; a++) s += L.Xm[tid * 12 + a] * L.lam[a];
        L.qdd[tid] = s;
    } if (tid < m) L.grf[3 * feet[tid / 3] + tid % 3] = L.lam[tid];)
    if (need_kinv) {
        if (m > 0) {
            if (mode == 0) chol_lds<NT>(L.G, L.LG, m, 12, L.tmp, 0.0);   // undamped (computeKKTContactDynamicMatrixInverse)
            HS_PHASE(NT, if (tid < m) chol_inv_col(L.LG, L.Lam, m, 12, tid);)
            HS_PHASE(NT, if (tid < m) {
                for (int i = 0; i < 18; i++) { double s = 0; for (int b = 0; b < m; b++) s += L.Xm[i * 12 + b] * L.Lam[b * 12 + tid]; L.Ym[i * 12 + tid] = s; }
            })
        }
        HS_PHASE(NT, if (tid < 18) {
            for (int i = 0; i < 18; i++) { double s = L.Minv[i * 18 + tid]; for (int a = 0; a < m; a++) s -= L.Ym[i * 12 + a] * L.Xm[tid * 12 + a]; L.Kinv[i * 30 + tid] = s; }
            for (int a = 0; a < m; a++) { L.Kinv[(18 + a) * 30 + tid] = L.Ym[tid * 12 + a]; L.Kinv[tid * 30 + 18 + a] = L.Ym[tid * 12 + a]; }
        } else if (tid >= 18 && tid < 18 + m) {
            int a = tid - 18; for (int b = 0; b < m; b++) L.Kinv[(18 + b) * 30 + 18 + a] = -L.Lam[b * 12 + a];
        })
    }
}

// Dual pass: lanes 0..35: d ID(q,v,acc)/dx_lane (psi_dyn, gravity `grav`); lanes 36..53: massless, foot forces L.fext,
// psi_kin, tangent on q_(lane-36): tau tangent = -d(J^T F)/dq, foot acc / vel tangents.  vscale/ascale scale v and acc.
template <int NT>
HD void wb_dpass(WbLds& L, const ModelDev& md, double grav, double vscale_dyn, double vscale_kin, double ascale_kin, bool q_only) {
    HS_PHASE(NT, if (tid < 54 && !(q_only && tid >= 18 && tid < 36)) {
        LaneCfg c = (tid < 36) ? lane_cfg(md, false, 1.0, grav, 0.0, vscale_dyn, 1.0, -1, tid < 18 ? tid : -1, tid >= 18 ? tid - 18 : -1)
                               : lane_cfg(md, true, 0.0, 0.0, 1.0, vscale_kin, ascale_kin, -1, tid - 36, -1);
        PassOut<Dual> o;
        wb_pass<Dual>(c, L.x, L.x + 18, L.acc, L.fext, o);
        for (int i = 0; i < 18; i++) L.dtau[i * 64 + tid] = o.tau[i].d;
        if (tid >= 36) for (int f = 0; f < 4; f++) {
            L.dacc[(3 * f) * 64 + tid] = o.facc[f].x.d; L.dacc[(3 * f + 1) * 64 + tid] = o.facc[f].y.d; L.dacc[(3 * f + 2) * 64 + tid] = o.facc[f].z.d;
            L.dvel[(3 * f) * 64 + tid] = o.fvel[f].x.d; L.dvel[(3 * f + 1) * 64 + tid] = o.fvel[f].y.d; L.dvel[(3 * f + 2) * 64 + tid] = o.fvel[f].z.d;
        }
    })
}

HD double reb_barrier(double g, double delta) {   // ConstraintsBase.h:238-245
    if (g > delta) return -log(g);
    double t = (g - 2 * delta) / delta;
    return .5 * (t * t - 1) - log(delta);
}

// path-constraint value c (order: torque 24, joint 24, height 1, grf 5/foot) — MHPCConstraint.cpp
HD double wb_constraint(const PhaseDev& P, const WbLds& L, int c) {
    if (P.go_torque >= 0 && c >= P.go_torque && c < P.go_torque + 24) { int i = c - P.go_torque; return i < 12 ? -L.u[i] + P.torque_limit : L.u[i - 12] + P.torque_limit; }
    if (P.go_joint >= 0 && c >= P.go_joint && c < P.go_joint + 24) { int i = c - P.go_joint; return i < 12 ? L.x[6 + i] - P.joint_lb[i % 3] : -L.x[6 + i - 12] + P.joint_ub[i % 3]; }
    if (P.go_height >= 0 && c == P.go_height) return L.x[2] - P.h_min;
    int i = c - P.go_grf, a = i / 5, r = i % 5, f = P.feet[a];
    double fx = L.grf[3 * f], fy = L.grf[3 * f + 1], fz = L.grf[3 * f + 2];
    if (r == 0) return fz;
    if (r == 1) return -fx + P.mu * fz;
    if (r == 2) return fx + P.mu * fz;
    if (r == 3) return -fy + P.mu * fz;
    return fy + P.mu * fz;
}
HD int wb_group_of(const PhaseDev& P, int c) {
    if (P.go_torque >= 0 && c >= P.go_torque && c < P.go_torque + 24) return 0;
    if (P.go_joint >= 0 && c >= P.go_joint && c < P.go_joint + 24) return 1;
    if (P.go_height >= 0 && c == P.go_height) return 2;
    return 3;
}

// running cost without barrier terms (tracking + foot costs), evaluated by one lane.  MHPCCost.cpp:4-245
HD double wb_running_cost_base(const PhaseDev& P, const WbLds& L, int k) {
    const double dt = P.dt;
    double lq = 0, lr = 0;
    for (int i = 0; i < 36; i++) { double d = L.x[i] - P.xr[(size_t)k * 36 + i]; lq += d * P.q[i] * d; }
    for (int i = 0; i < 12; i++) { double d = L.u[i] - P.ur[(size_t)k * 12 + i]; lr += d * P.r[i] * d; }
    double l = 0.5 * lq; l += 0.5 * lr; l *= dt;
    const int* rc = P.ref_contact + (size_t)k * 4; const double* fp = P.foot_pos + (size_t)k * 12; const double* bp = P.body_pos + (size_t)k * 3;
    double l2 = 0, l3 = 0, l4 = 0;
    for (int f = 0; f < 4; f++) {
        double d[3]; for (int a = 0; a < 3; a++) d[a] = (L.fpos[3 * f + a] - L.x[a]) - (fp[3 * f + a] - bp[a]);
        if (rc[f] > 0 && P.w_foot_reg[0] >= 0) { double s = 0; for (int a = 0; a < 3; a++) s += d[a] * P.w_foot_reg[a] * d[a]; l2 += 0.5 * s * dt; }
        if (rc[f] == 0 && P.w_swing_pos[0] >= 0) { double s = 0; for (int a = 0; a < 3; a++) s += d[a] * P.w_swing_pos[a] * d[a]; l3 += 0.5 * s * dt; }
        if (rc[f] == 0 && P.w_swing_vel[0] >= 0) { double s = 0; for (int a = 0; a < 3; a++) { double dv = L.fvel[3 * f + a] - P.foot_vel[(size_t)k * 12 + 3 * f + a]; s += dv * P.w_swing_vel[a] * dv; } l4 += 0.5 * s * dt; }
    }
    l += l2; l += l3; l += l4;
    return l;
}

struct SlotOut { double* cost; double* dsq; double* ming; double* maxh; };   // per (problem, slot) partials

// -------------------------------------------------------------------------------------------------------
// Rollout of one knot k < h of problem b.   eps: line-search step.
template <int NT>
HD void wb_rollout_knot(WbLds& L, const PhaseDev& P, const ModelDev& md, int b, int k, double eps, int reb_active,
                        const double* x0, SlotOut so, size_t slot, int* fail_flag) {
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + k) * 36, ku = ((size_t)b * h + k) * 12;
    HS_PHASE(NT, if (tid < 36) {
        double xb = P.Xbar[kx + tid], x = xb + eps * P.dX[kx + tid];
        L.xb[tid] = xb; L.x[tid] = x; P.X[kx + tid] = x;
    } if (tid < 18) L.acc[tid] = 0.0; if (tid < 12) L.fext[tid] = 0.0; L.red[tid] = 0.0;)
    // K (12x36 column-major) staged through L.out with unit-stride loads
    HS_PHASE(NT, for (int i = tid; i < 432; i += NT) L.out[i] = P.K[((size_t)b * h + k) * 432 + i];)
    HS_PHASE(NT, if (tid < 12) {
        double s = 0; for (int j = 0; j < 36; j++) s += L.out[tid + 12 * j] * (L.x[j] - L.xb[j]);
        double u = P.Ubar[ku + tid] + eps * P.dU[ku + tid] + s;
        L.u[tid] = u; P.U[ku + tid] = u;
    } if (tid < 18) L.tau[tid] = 0.0;)
    HS_PHASE(NT, if (tid < 12) L.tau[6 + tid] = L.u[tid];)
    wb_terms<NT>(L, md, true);
    wb_kkt<NT>(L, P.nc, P.feet, 0, P.bg_alpha, false);
    // integrate, defect of knot k+1 (and of knot 0 for the very first knot of phase 0), outputs
    HS_PHASE(NT, if (tid < 36) {
        double xs = (tid < 18) ? L.x[tid] + L.x[18 + tid] * P.dt : L.x[tid] + L.qdd[tid - 18] * P.dt;
        P.Xsim[kx + 36 + tid] = xs;
        double xn = P.Xbar[kx + 36 + tid] + eps * P.dX[kx + 36 + tid];
        double d = xs - xn; P.Defect[kx + 36 + tid] = d;
        double dsq = d * d, nsq = xs * xs;
        if (x0 != nullptr && k == 0) { double d0 = x0[(size_t)b * 36 + tid] - L.x[tid]; P.Xsim[kx + tid] = x0[(size_t)b * 36 + tid]; P.Defect[kx + tid] = d0; dsq += d0 * d0; }
        L.red[tid] = dsq; L.tmp[tid] = nsq;
    } if (tid < 12) P.Y[((size_t)b * h + k) * 12 + tid] = L.grf[tid];)
    // constraints + barrier
    HS_PHASE(NT, for (int c = tid; c < P.ng; c += NT) {
        double g = wb_constraint(P, L, c); L.gval[c] = g;
        size_t gi = ((size_t)b * h + k) * P.ng + c; P.g[gi] = g;
        L.bar[c] = P.eps[gi] * reb_barrier(g, P.delta[gi]);
    })
    HS_PHASE(NT, if (tid == 0) {
        double lb = wb_running_cost_base(P, L, k);
        P.lbase[(size_t)b * h + k] = lb;
        double l = lb;
        if (reb_active) {   // per constraint object: ReB_cost then l += dt*ReB_cost (SinglePhase.cpp:394-402)
            int offs[4] = {P.go_torque, P.go_joint, P.go_height, P.go_grf}; int sz[4] = {24, 24, 1, 5 * P.nc};
            for (int gI = 0; gI < 4; gI++) if (offs[gI] >= 0) { double c = 0; for (int i = 0; i < sz[gI]; i++) c += L.bar[offs[gI] + i]; l += P.dt * c; }
        }
        P.l[(size_t)b * h + k] = l;
        double ming = 0; for (int c = 0; c < P.ng; c++) ming = fmin(ming, L.gval[c]);
        double dsq = 0, nsq = 0; for (int i = 0; i < 36; i++) { dsq += L.red[i]; nsq += L.tmp[i]; }
        so.cost[slot] = l; so.dsq[slot] = dsq; so.ming[slot] = ming; so.maxh[slot] = 0.0;
        if (sqrt(nsq) > 1e6 || !(nsq == nsq)) fail_flag[b] = 1;   // SinglePhase.cpp:205
    })
}

// terminal cost without AL (tracking + foot-place reg (x1) + touchdown-velocity penalty). MHPCCost.cpp:67-87,255-268
HD double wb_terminal_cost_base(const PhaseDev& P, const WbLds& L) {
    const int h = P.h;
    double s = 0; for (int i = 0; i < 36; i++) { double d = L.x[i] - P.xr[(size_t)h * 36 + i]; s += d * P.qf[i] * d; }
    double Phi = 0.5 * s;
    const int* rc = P.ref_contact + (size_t)h * 4; const double* fp = P.foot_pos + (size_t)h * 12; const double* bp = P.body_pos + (size_t)h * 3;
    double l2 = 0, l5 = 0;
    for (int f = 0; f < 4; f++) {
        if (rc[f] > 0 && P.w_foot_reg[0] >= 0) { double t = 0; for (int a = 0; a < 3; a++) { double d = (L.fpos[3 * f + a] - L.x[a]) - (fp[3 * f + a] - bp[a]); t += d * P.w_foot_reg[a] * d; } l2 += 0.5 * t; }
        if (P.td[f] && P.n_td > 0 && P.w_td_vel >= 0) { double vz = L.fvel[3 * f + 2]; l5 += 0.5 * vz * P.w_td_vel * vz; }
    }
    Phi += l2; Phi += l5;
    return Phi;
}

// Terminal knot (k = h) of a phase: terminal constraint + cost, then the reset map into the next phase.
template <int NT>
HD void wb_rollout_terminal(WbLds& L, const PhaseDev& P, const PhaseDev* Pn, const ModelDev& md, int b, double eps, int al_active,
                            SlotOut so, size_t slot) {
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + h) * 36;
    HS_PHASE(NT, if (tid < 36) { double x = P.Xbar[kx + tid] + eps * P.dX[kx + tid]; L.x[tid] = x; P.X[kx + tid] = x; }
             if (tid < 18) { L.acc[tid] = 0.0; L.tau[tid] = 0.0; } if (tid < 12) L.fext[tid] = 0.0;)
    const bool impact = (Pn != nullptr) && P.has_impact;
    wb_terms<NT>(L, md, impact);
    HS_PHASE(NT, if (tid == 0) {
        double pb = wb_terminal_cost_base(P, L); P.Phibase[b] = pb;
        double maxh = 0, c = 0; int i = 0;
        for (int f = 0; f < 4; f++) if (P.td[f] && P.nt > 0) {
            double hh = L.fpos[3 * f + 2] - P.ground_height; P.th[(size_t)b * P.nt + i] = hh; maxh = fmax(maxh, fabs(hh));
            double sg = P.sigma[(size_t)b * P.nt + i], lm = P.lambda[(size_t)b * P.nt + i];
            c += 0.5 * sg * hh * hh; c += lm * hh; i++;
        }
        double Phi = pb; if (al_active && P.nt > 0) Phi += c;
        P.Phi[b] = Phi;
        so.cost[slot] = Phi; so.ming[slot] = 0.0; so.maxh[slot] = maxh; so.dsq[slot] = 0.0;
    })
    if (Pn == nullptr) return;
    // reset map (MHPCReset.cpp:4-28): impact if any touchdown, then optional WB->SRB projection
    if (impact) {
        int tdfeet[4]; int ntd = 0; for (int f = 0; f < 4; f++) if (P.td[f]) tdfeet[ntd++] = f;
        wb_kkt<NT>(L, ntd, tdfeet, 1, 0.0, false);
    } else { HS_PHASE(NT, if (tid < 18) L.qdd[tid] = L.x[18 + tid];) }
    const int nn = Pn->n;
    const size_t nx = ((size_t)b * (Pn->h + 1)) * nn;
    HS_PHASE(NT, if (tid < nn) {
        double xi;
        if (nn == 36) xi = (tid < 18) ? L.x[tid] : L.qdd[tid - 18];
        else xi = (tid < 6) ? L.x[tid] : L.qdd[tid - 6];          // StateProjection: x[0:6], x[18:24] (MHPCReset.h:24-26)
        Pn->Xsim[nx + tid] = xi;
        double d = xi - (Pn->Xbar[nx + tid] + eps * Pn->dX[nx + tid]);
        Pn->Defect[nx + tid] = d; L.red[tid] = d * d;
    })
    HS_PHASE(NT, if (tid == 0) { double s = 0; for (int i = 0; i < nn; i++) s += L.red[i]; so.dsq[slot] = s; })
}

// -------------------------------------------------------------------------------------------------------
// coalesced copy LDS -> global
template <int NT> HD void store_block(double* dst, const double* src, int n) { HS_PHASE(NT, for (int i = tid; i < n; i += NT) dst[i] = src[i];) }

// LQ approximation of knot k < h (reads X,U,Y? no: recomputes the contact solve at the stored X,U like WBM.cpp:463)
template <int NT>
HD void wb_lq_knot(WbLds& L, const PhaseDev& P, const ModelDev& md, int b, int k, int reb_active) {
    const int h = P.h; const double dt = P.dt;
    const size_t kx = ((size_t)b * (h + 1) + k) * 36, ku = ((size_t)b * h + k) * 12, kk = (size_t)b * h + k;
    HS_PHASE(NT, if (tid < 36) L.x[tid] = P.X[kx + tid]; if (tid < 12) { L.u[tid] = P.U[ku + tid]; L.fext[tid] = 0.0; }
             if (tid < 18) { L.acc[tid] = 0.0; L.tau[tid] = 0.0; })
    HS_PHASE(NT, if (tid < 12) L.tau[6 + tid] = L.u[tid];)
    wb_terms<NT>(L, md, true);
    wb_kkt<NT>(L, P.nc, P.feet, 0, P.bg_alpha, true);
    const int m = 3 * P.nc;
    HS_PHASE(NT, if (tid < 18) L.acc[tid] = L.qdd[tid]; if (tid < 12) L.fext[tid] = L.grf[tid];)
    wb_dpass<NT>(L, md, GRAV, 1.0, 1.0, 1.0, false);
    // rhs columns in place: top[:,d] = dtau - dJTF ; bot rows (compact active) into dacc[a*64 + d], d < 36
    HS_PHASE(NT, if (tid < 36) {
        const int d = tid;
        if (d < 18) for (int i = 0; i < 18; i++) L.dtau[i * 64 + d] += L.dtau[i * 64 + 36 + d];   // lanes 36+: tau tangent = -dJTF
    })
    HS_PHASE(NT, if (tid < 36) {
        const int d = tid; double bot[12];
        for (int a = 0; a < m; a++) {
            int f = P.feet[a / 3], r = 3 * f + a % 3;
            if (d < 18) bot[a] = L.dacc[r * 64 + 36 + d] + 2.0 * P.bg_alpha * L.dvel[r * 64 + 36 + d];
            else bot[a] = 2.0 * L.dvel[r * 64 + 36 + (d - 18)] + 2.0 * P.bg_alpha * L.Jall[r * 18 + (d - 18)];   // footAccPartialDv == 2 footVelPartialDq
        }
        // column d of A (36) and C (12) into L.out (A: 36x36 col-major at 0)
        for (int i = 0; i < 18; i++) {
            double s = 0;
            for (int j = 0; j < 18; j++) s -= L.Kinv[i * 30 + j] * L.dtau[j * 64 + d];
            for (int a = 0; a < m; a++) s -= L.Kinv[i * 30 + 18 + a] * bot[a];
            L.out[(18 + i) + 36 * d] = s * dt + ((d == 18 + i) ? 1.0 : 0.0);
            L.out[i + 36 * d] = ((d == i) ? 1.0 : 0.0) + ((d == 18 + i) ? dt : 0.0);
        }
        for (int a = 0; a < 12; a++) L.dacc[a * 64 + d] = 0.0;   // reuse lanes<36 part of dacc as C staging (rows = grf index)
        for (int a = 0; a < m; a++) {
            double s = 0;
            for (int j = 0; j < 18; j++) s += L.Kinv[(18 + a) * 30 + j] * L.dtau[j * 64 + d];
            for (int b2 = 0; b2 < m; b2++) s += L.Kinv[(18 + a) * 30 + 18 + b2] * bot[b2];
            L.dacc[(3 * P.feet[a / 3] + a % 3) * 64 + d] = s;
        }
    })
    store_block<NT>(P.A + kk * 1296, L.out, 1296);
    // C (12x36 col-major), B (36x12), D (12x12) staged in L.out
    HS_PHASE(NT, for (int i = tid; i < 432; i += NT) { int r = i % 12, d = i / 12; L.out[i] = L.dacc[r * 64 + d]; }
             for (int i = tid; i < 432; i += NT) { int r = i % 36, j = i / 36; L.out[432 + i] = (r < 18) ? 0.0 : L.Kinv[(r - 18) * 30 + 6 + j] * dt; }
             for (int i = tid; i < 144; i += NT) { int r = i % 12, j = i / 12, f = r / 3; int a = -1; for (int t = 0; t < P.nc; t++) if (P.feet[t] == f) a = 3 * t + r % 3;
                 L.out[864 + i] = (a >= 0) ? -L.Kinv[(18 + a) * 30 + 6 + j] : 0.0; })
    store_block<NT>(P.C + kk * 432, L.out, 432);
    store_block<NT>(P.B + kk * 432, L.out + 432, 432);
    store_block<NT>(P.D + kk * 144, L.out + 864, 144);
    // ---------------- cost partials.  Jv = d(foot vel)/dq (psi_kin) sits in dvel lanes 36..53
    HS_PHASE(NT, for (int i = tid; i < 216; i += NT) { int r = i / 18, j = i % 18; L.Jv[i] = L.dvel[r * 64 + 36 + j]; })
    // constraint values (from the rollout) and barrier derivatives
    HS_PHASE(NT, for (int c = tid; c < P.ng; c += NT) {
        size_t gi = kk * P.ng + c; double g = P.g[gi], delta = P.delta[gi], e = P.eps[gi], bd, bdd;
        if (g > delta) { bd = -1.0 / g; bdd = 1.0 / (g * g); } else { bd = (g - 2 * delta) / delta / delta; bdd = 1.0 / (delta * delta); }
        L.bd[c] = reb_active ? e * bd : 0.0; L.bdd[c] = reb_active ? e * bdd : 0.0;
    })
    const int* rc = P.ref_contact + (size_t)k * 4; const double* fp = P.foot_pos + (size_t)k * 12; const double* bp = P.body_pos + (size_t)k * 3;
    // lxx column d (36 rows) + lx[d]
    HS_PHASE(NT, if (tid < 36) {
        const int d = tid;
        double col[36]; for (int i = 0; i < 36; i++) col[i] = 0.0;
        double lxd = dt * P.q[d] * (L.x[d] - P.xr[(size_t)k * 36 + d]);
        col[d] += dt * P.q[d];
        for (int pass = 0; pass < 3; pass++) {   // foot-place reg, swing pos, swing vel: each its own cost object
            double tx = 0.0, tcol[36]; for (int i = 0; i < 36; i++) tcol[i] = 0.0;
            for (int f = 0; f < 4; f++) {
                const double* w = pass == 0 ? P.w_foot_reg : pass == 1 ? P.w_swing_pos : P.w_swing_vel;
                bool on = (pass == 0 ? rc[f] > 0 : rc[f] == 0) && w[0] >= 0;
                if (!on) continue;
                if (pass < 2) {
                    if (d >= 18 || d < 3) continue;   // J_foot.leftCols<3>().setZero(); only the q block
                    double dd[3], jd[3];
                    for (int a = 0; a < 3; a++) { dd[a] = (L.fpos[3 * f + a] - L.x[a]) - (fp[3 * f + a] - bp[a]); jd[a] = L.Jall[(3 * f + a) * 18 + d]; }
                    double s = 0; for (int a = 0; a < 3; a++) s += jd[a] * w[a] * dd[a]; tx += s * dt;
                    for (int i = 3; i < 18; i++) { double t = 0; for (int a = 0; a < 3; a++) t += L.Jall[(3 * f + a) * 18 + i] * w[a] * jd[a]; tcol[i] += t * dt; }
                } else {
                    double dv[3], jd[3];
                    for (int a = 0; a < 3; a++) { dv[a] = L.fvel[3 * f + a] - P.foot_vel[(size_t)k * 12 + 3 * f + a]; jd[a] = d < 18 ? L.Jv[(3 * f + a) * 18 + d] : L.Jall[(3 * f + a) * 18 + d - 18]; }
                    double s = 0; for (int a = 0; a < 3; a++) s += jd[a] * w[a] * dv[a]; tx += s * dt;
                    for (int i = 0; i < 36; i++) { double t = 0; for (int a = 0; a < 3; a++) { double ji = i < 18 ? L.Jv[(3 * f + a) * 18 + i] : L.Jall[(3 * f + a) * 18 + i - 18]; t += ji * w[a] * jd[a]; } tcol[i] += t * dt; }
                }
            }
            lxd += tx; for (int i = 0; i < 36; i++) col[i] += tcol[i];
        }
        // ReB fold on x (joint limits: x[6+i], height: x[2]) — rank-1 updates on the diagonal (ConstraintsBase.h:282-287)
        if (P.go_joint >= 0 && d >= 6 && d < 18) {
            int i = d - 6; double gsum = L.bd[P.go_joint + i] * 1.0 + L.bd[P.go_joint + 12 + i] * (-1.0);
            double hsum = L.bdd[P.go_joint + i] + L.bdd[P.go_joint + 12 + i];
            lxd += dt * gsum; col[d] += dt * hsum;
        }
        if (P.go_height >= 0 && d == 2) { lxd += dt * L.bd[P.go_height]; col[d] += dt * L.bdd[P.go_height]; }
        for (int i = 0; i < 36; i++) L.out[i + 36 * d] = col[i];
        P.lx[kk * 36 + d] = lxd;
    })
    store_block<NT>(P.lxx + kk * 1296, L.out, 1296);
    // lu, luu (diag + torque barrier), ly, lyy (grf barrier 3x3 blocks)
    HS_PHASE(NT, for (int i = tid; i < 144; i += NT) { L.out[i] = 0.0; L.out[144 + i] = 0.0; })
    HS_PHASE(NT, if (tid < 12) {
        const int i = tid;
        double lu = dt * P.r[i] * (L.u[i] - P.ur[(size_t)k * 12 + i]), luu = dt * P.r[i];
        if (P.go_torque >= 0) { lu += dt * (L.bd[P.go_torque + i] * (-1.0) + L.bd[P.go_torque + 12 + i]); luu += dt * (L.bdd[P.go_torque + i] + L.bdd[P.go_torque + 12 + i]); }
        P.lu[kk * 12 + i] = lu; L.out[i + 12 * i] = luu;
        // y: grf pyramid rows [0 0 1; -1 0 mu; 1 0 mu; 0 -1 mu; 0 1 mu] for foot f = i/3
        double ly = 0.0; int f = i / 3, r = i % 3, a = -1; for (int t = 0; t < P.nc; t++) if (P.feet[t] == f) a = t;
        if (P.go_grf >= 0 && a >= 0) {
            const double mu = P.mu; const double rows[5][3] = {{0, 0, 1}, {-1, 0, mu}, {1, 0, mu}, {0, -1, mu}, {0, 1, mu}};
            for (int c = 0; c < 5; c++) {
                ly += L.bd[P.go_grf + 5 * a + c] * rows[c][r];
                for (int r2 = 0; r2 < 3; r2++) L.out[144 + (3 * f + r2) + 12 * i] += dt * L.bdd[P.go_grf + 5 * a + c] * rows[c][r2] * rows[c][r];
            }
        }
        P.ly[kk * 12 + i] = dt * ly;
    })
    store_block<NT>(P.luu + kk * 144, L.out, 144);
    store_block<NT>(P.lyy + kk * 144, L.out + 144, 144);
}

// Terminal partials of a phase (+ AL) and the reset-map partial Px (next_n x 36, column-major) if a phase follows.
template <int NT>
HD void wb_lq_terminal(WbLds& L, const PhaseDev& P, const PhaseDev* Pn, const ModelDev& md, int b, int al_active) {
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + h) * 36;
    HS_PHASE(NT, if (tid < 36) L.x[tid] = P.X[kx + tid]; if (tid < 18) { L.acc[tid] = 0.0; L.tau[tid] = 0.0; } if (tid < 12) L.fext[tid] = 0.0;)
    wb_terms<NT>(L, md, true);
    // Jv = d(J v)/dq with psi_kin at (q, v): dual kinematic lanes only
    wb_dpass<NT>(L, md, 0.0, 1.0, 1.0, 0.0, true);
    HS_PHASE(NT, for (int i = tid; i < 216; i += NT) { int r = i / 18, j = i % 18; L.Jv[i] = L.dvel[r * 64 + 36 + j]; })
    const int* rc = P.ref_contact + (size_t)h * 4; const double* fp = P.foot_pos + (size_t)h * 12; const double* bp = P.body_pos + (size_t)h * 3;
    HS_PHASE(NT, if (tid < 36) {
        const int d = tid; double col[36]; for (int i = 0; i < 36; i++) col[i] = 0.0;
        double px = P.qf[d] * (L.x[d] - P.xr[(size_t)h * 36 + d]); col[d] += P.qf[d];
        {   // WBFootPlaceReg::terminal_cost_par: x2 (MHPCCost.cpp:114-115)
            double tx = 0, tcol[36]; for (int i = 0; i < 36; i++) tcol[i] = 0.0;
            if (d >= 3 && d < 18) for (int f = 0; f < 4; f++) if (rc[f] > 0 && P.w_foot_reg[0] >= 0) {
                const double* w = P.w_foot_reg; double dd[3], jd[3];
                for (int a = 0; a < 3; a++) { dd[a] = (L.fpos[3 * f + a] - L.x[a]) - (fp[3 * f + a] - bp[a]); jd[a] = L.Jall[(3 * f + a) * 18 + d]; }
                double s = 0; for (int a = 0; a < 3; a++) s += jd[a] * w[a] * dd[a]; tx += 2 * s;
                for (int i = 3; i < 18; i++) { double t = 0; for (int a = 0; a < 3; a++) t += L.Jall[(3 * f + a) * 18 + i] * w[a] * jd[a]; tcol[i] += 2 * t; }
            }
            px += tx; for (int i = 0; i < 36; i++) col[i] += tcol[i];
        }
        if (P.n_td > 0 && P.w_td_vel >= 0) {   // TDVelocityPenalty::terminal_cost_par (MHPCCost.cpp:271-291)
            double tx = 0, tcol[36]; for (int i = 0; i < 36; i++) tcol[i] = 0.0;
            for (int f = 0; f < 4; f++) if (P.td[f]) {
                double vz = L.fvel[3 * f + 2]; double jd = d < 18 ? L.Jv[(3 * f + 2) * 18 + d] : L.Jall[(3 * f + 2) * 18 + d - 18];
                tx += jd * P.w_td_vel * vz;
                for (int i = 0; i < 36; i++) { double ji = i < 18 ? L.Jv[(3 * f + 2) * 18 + i] : L.Jall[(3 * f + 2) * 18 + i - 18]; tcol[i] += ji * P.w_td_vel * jd; }
            }
            px += tx; for (int i = 0; i < 36; i++) col[i] += tcol[i];
        }
        if (al_active && P.nt > 0) {   // compute_AL_partials (ConstraintsBase.h:412-425); hx[0:18] = J_foot,z
            double ag = 0, acol[36]; for (int i = 0; i < 36; i++) acol[i] = 0.0; int t = 0;
            for (int f = 0; f < 4; f++) if (P.td[f]) {
                double sg = P.sigma[(size_t)b * P.nt + t], lm = P.lambda[(size_t)b * P.nt + t], hh = P.th[(size_t)b * P.nt + t];
                double hd = d < 18 ? L.Jall[(3 * f + 2) * 18 + d] : 0.0;
                ag += (sg * hh + lm) * hd;
                for (int i = 0; i < 18; i++) acol[i] += (sg * (1 + hh) + lm) * (L.Jall[(3 * f + 2) * 18 + i] * hd);
                t++;
            }
            px += ag; for (int i = 0; i < 36; i++) col[i] += acol[i];
        }
        P.Phix[(size_t)b * 36 + d] = px;
        for (int i = 0; i < 36; i++) L.out[i + 36 * d] = col[i];
    })
    store_block<NT>(P.Phixx + (size_t)b * 1296, L.out, 1296);
    if (Pn == nullptr) return;
    const int nn = Pn->n;
    if (!P.has_impact) {
        HS_PHASE(NT, for (int i = tid; i < nn * 36; i += NT) { int r = i % nn, c = i / nn; int src = (nn == 36) ? r : (r < 6 ? r : r + 12); L.out[i] = (src == c) ? 1.0 : 0.0; })
        store_block<NT>(P.Px + (size_t)b * nn * 36, L.out, nn * 36);
        return;
    }
    // ---- impact partial (WBM.cpp:508-543)
    int tdfeet[4]; int ntd = 0; for (int f = 0; f < 4; f++) if (P.td[f]) tdfeet[ntd++] = f;
    const int m = 3 * ntd;
    wb_kkt<NT>(L, ntd, tdfeet, 1, 0.0, true);    // L.qdd = v+, L.lam = impulse_c (compact), Kinv
    // pass A: d(M dv)/dq  (v = 0, acc = v+ - v, gravity off) on lanes 0..17 ; d(J^T imp)/dq with the mis-sliced impulse (quirk v)
    HS_PHASE(NT, if (tid < 18) L.acc[tid] = L.qdd[tid] - L.x[18 + tid];
             if (tid < 12) { L.fext[tid] = 0.0; })
    HS_PHASE(NT, if (tid == 0) { double pad[16]; for (int i = 0; i < 16; i++) pad[i] = i < m ? L.lam[i] : 0.0;
                 for (int i = 0; i < ntd; i++) for (int d = 0; d < 3; d++) L.fext[3 * tdfeet[i] + d] = pad[i + d]; })   // WBM.cpp:454: offset i, not 3i
    wb_dpass<NT>(L, md, 0.0, 0.0, 0.0, 0.0, true);
    HS_PHASE(NT, if (tid < 18) for (int i = 0; i < 18; i++) L.dtau[i * 64 + tid] += L.dtau[i * 64 + 36 + tid];)
    // pass B: d(J v+)/dq with psi_kin: overwrite velocity with v+ (x is no longer needed as pre-impact v except for Px assembly -> keep copy in xb)
    HS_PHASE(NT, if (tid < 18) { L.xb[tid] = L.x[18 + tid]; })
    HS_PHASE(NT, if (tid < 18) { L.x[18 + tid] = L.qdd[tid]; })
    HS_PHASE(NT, if (tid >= 36 && tid < 54) {
        LaneCfg c = lane_cfg(md, true, 0.0, 0.0, 0.0, 1.0, 0.0, -1, tid - 36, -1);
        PassOut<Dual> o; wb_pass<Dual>(c, L.x, L.x + 18, L.acc, L.fext, o);
        for (int f = 0; f < 4; f++) { L.dvel[(3 * f) * 64 + tid] = o.fvel[f].x.d; L.dvel[(3 * f + 1) * 64 + tid] = o.fvel[f].y.d; L.dvel[(3 * f + 2) * 64 + tid] = o.fvel[f].z.d; }
    })
    // Px = [I 0; dv+/dq dv+/dv] , dv+/dq = -TL*dtau_dq - TR*dv_dq ; dv+/dv = TL*M
    HS_PHASE(NT, if (tid < 36) {
        const int d = tid;
        for (int i = 0; i < 18; i++) {
            double s = 0;
            if (d < 18) {
                for (int j = 0; j < 18; j++) s -= L.Kinv[i * 30 + j] * L.dtau[j * 64 + d];
                for (int a = 0; a < m; a++) s -= L.Kinv[i * 30 + 18 + a] * L.dvel[(3 * tdfeet[a / 3] + a % 3) * 64 + 36 + d];
            } else { for (int j = 0; j < 18; j++) s += L.Kinv[i * 30 + j] * L.M[j * 18 + (d - 18)]; }
            int r = 18 + i;
            if (nn == 36) { L.out[r + 36 * d] = s; L.out[i + 36 * d] = (i == d) ? 1.0 : 0.0; }
            else { if (i < 6) { L.out[(6 + i) + 12 * d] = s; L.out[i + 12 * d] = (i == d) ? 1.0 : 0.0; } }
        }
    })
    store_block<NT>(P.Px + (size_t)b * nn * 36, L.out, nn * 36);
}

}  // namespace hs
