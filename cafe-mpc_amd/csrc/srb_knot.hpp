// Single-rigid-body phases (n = 12, m = 12, p = 0) of the MHPC problem: per-knot rollout and LQ approximation, one
// wavefront per knot.  The reference evaluates this model through CasADi-generated code
// (MHPC/MHPC-Trajopt/SRBM.h:43-93: SRBDynamics / SRBDynamicsDerivatives); here the dynamics is ONE closed-form
// function templated on the scalar, and the 24 columns of [A B] come from 24 lanes running it on forward-mode duals.
//   x = [p(3), (yaw, pitch, roll)(3), v(3), Euler rates(3)],  u = 4 world-frame foot forces,
//   per-knot data: planned foot positions and the contact flags of the reference (MHPCProblem.cpp:233-252).
#pragma once
#include "hs_types.hpp"
#include "wb_knot.hpp"   // SlotOut, reb_barrier

namespace hs {

constexpr double SRB_MASS = 8.912;                       // constants of the generated code
constexpr double SRB_IXX = 0.061578036, SRB_IYY = 0.2207093, SRB_IZZ = 0.272612336, SRB_IXZ = 5.38e-5;

HD double srb_recip(double a) { return 1.0 / a; }
HD Dual srb_recip(Dual a) { const double r = 1.0 / a.v; return Dual(r, -a.d * r * r); }

// continuous-time xdot = f(x, u; pf, contact)
template <class S>
HD void srb_xdot(const S* x, const S* u, const double* pf, const int* contact, S* xd) {
    S sy, cy, st, ct, sp, cp;
    sincos_(x[3], sy, cy); sincos_(x[4], st, ct); sincos_(x[5], sp, cp);
    const S dyaw = x[9], dth = x[10], dph = x[11];
    V3<S> F{S(0.0), S(0.0), S(0.0)}, tw{S(0.0), S(0.0), S(0.0)};
    for (int l = 0; l < 4; l++) if (contact[l]) {
        const V3<S> f{u[3 * l], u[3 * l + 1], u[3 * l + 2]};
        const V3<S> r{S(pf[3 * l]) - x[0], S(pf[3 * l + 1]) - x[1], S(pf[3 * l + 2]) - x[2]};
        F = F + f; tw = tw + cross(r, f);
    }
    for (int i = 0; i < 6; i++) xd[i] = x[6 + i];
    xd[6] = F.x * (1.0 / SRB_MASS); xd[7] = F.y * (1.0 / SRB_MASS); xd[8] = F.z * (1.0 / SRB_MASS) - GRAV;
    // body-frame torque: R^T tw with R = Rz(yaw) Ry(pitch) Rx(roll)
    const V3<S> tb = rotT<0, S>(cp, sp, rotT<1, S>(ct, st, rotT<2, S>(cy, sy, tw)));
    // body rates w = T(eul) eul_dot
    const V3<S> w{dph - st * dyaw, sp * ct * dyaw + cp * dth, cp * ct * dyaw - sp * dth};
    const V3<S> Iw{SRB_IXX * w.x + SRB_IXZ * w.z, SRB_IYY * w.y, SRB_IXZ * w.x + SRB_IZZ * w.z};
    const V3<S> rhs = tb - cross(w, Iw);
    constexpr double det = SRB_IXX * SRB_IZZ - SRB_IXZ * SRB_IXZ;
    const V3<S> wd{(SRB_IZZ * rhs.x - SRB_IXZ * rhs.z) * (1.0 / det), rhs.y * (1.0 / SRB_IYY), (SRB_IXX * rhs.z - SRB_IXZ * rhs.x) * (1.0 / det)};
    // Tdot eul_dot, column by column (the roll column of T is constant)
    const V3<S> c0d{-(ct * dth), cp * ct * dph - sp * st * dth, -(sp * ct * dph) - cp * st * dth};
    const V3<S> c1d{S(0.0), -(sp * dph), -(cp * dph)};
    const V3<S> bb = wd - (scale(dyaw, c0d) + scale(dth, c1d));
    const S ddyaw = (sp * bb.y + cp * bb.z) * srb_recip(ct);
    xd[9] = ddyaw; xd[10] = cp * bb.y - sp * bb.z; xd[11] = bb.x + st * ddyaw;
}

struct SrbLds {
    double x[12], xb[12], u[12], xd[12], red[12], tmp[12];
    double K[144], AB[288];
    double g, bar, bd, bdd;
};

// Rollout of one SRB knot k < h of problem b (SinglePhase::forward_sweep body with the SRB callbacks).
template <int NT>
HD void srb_rollout_knot(SrbLds& L, PhaseC& P, int b, int k, double eps, int reb_active, const double* x0, SlotOut so, size_t slot, int* fail_flag, bool ss = false, bool wr = true) {
    // ss: single shooting (MS = false, MultiPhaseDDP.cpp:65-68): X[k] is the state the previous knot of this wave simulated (Xsim[k]), no defect
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + k) * 12, ku = ((size_t)b * h + k) * 12, kk = (size_t)b * h + k;
    HS_PHASE(NT, if (tid < 12) { double xb = P.Xbar[kx + tid], x = ss ? P.Xsim[kx + tid] : xb + eps * P.dX[kx + tid]; L.xb[tid] = xb; L.x[tid] = x; if (wr) { P.X[kx + tid] = x; if (ss && k == 0) P.Defect[kx + tid] = 0.0; } }
             for (int i = tid; i < 144; i += NT) L.K[i] = P.K[kk * 144 + i];)
    HS_PHASE(NT, if (tid < 12) {
        double s = 0; for (int j = 0; j < 12; j++) s += L.K[tid + 12 * j] * (L.x[j] - L.xb[j]);
        const double u = P.Ubar[ku + tid] + eps * P.dU[ku + tid] + s;
        L.u[tid] = u; if (wr) P.U[ku + tid] = u;
    })
    HS_PHASE(NT, if (tid == 0) srb_xdot<double>(L.x, L.u, P.foot_pos + (size_t)k * 12, P.ref_contact + (size_t)k * 4, L.xd);)
    HS_PHASE(NT, if (tid < 12) {
        const double xs = L.x[tid] + L.xd[tid] * P.dt;
        if (wr) P.Xsim[kx + 12 + tid] = xs;
        const double d = ss ? 0.0 : xs - (P.Xbar[kx + 12 + tid] + eps * P.dX[kx + 12 + tid]);
        if (wr) P.Defect[kx + 12 + tid] = d;
        double dsq = d * d;
        if (x0 != nullptr && k == 0) { const double d0 = x0[(size_t)b * 12 + tid] - L.x[tid]; if (wr) { P.Xsim[kx + tid] = x0[(size_t)b * 12 + tid]; P.Defect[kx + tid] = d0; } dsq += d0 * d0; }
        L.red[tid] = dsq; L.tmp[tid] = xs * xs;
    })
    HS_PHASE(NT, if (tid == 0) {
        double lq = 0, lr = 0;
        for (int i = 0; i < 12; i++) { const double d = L.x[i] - P.xr[(size_t)k * 12 + i]; lq += d * P.q[i] * d; }
        for (int i = 0; i < 12; i++) { const double d = L.u[i] - P.ur[(size_t)k * 12 + i]; lr += d * P.r[i] * d; }
        double l = 0.5 * lq; l += 0.5 * lr; l *= P.dt;
        if (wr) P.lbase[kk] = l;
        double ming = 0;
        if (P.go_height >= 0) {   // MinimumHeight on the body (MHPCConstraint.cpp:207-250): the only SRB path constraint
            const size_t gi = kk * P.ng + P.go_height; const double g = L.x[2] - P.h_min;
            if (wr) P.g[gi] = g; ming = fmin(ming, g);
            if (reb_active) l += P.dt * (P.eps[gi] * reb_barrier(g, P.delta[gi]));
        }
        if (wr) P.l[kk] = l;
        double dsq = 0, nsq = 0; for (int i = 0; i < 12; i++) { dsq += L.red[i]; nsq += L.tmp[i]; }
        so.cost[slot] = l; so.dsq[slot] = dsq; so.ming[slot] = ming; so.maxh[slot] = 0.0;
        if (sqrt(nsq) > 1e6 || !(nsq == nsq)) fail_flag[b] = 1;
    })
}

// Terminal knot of an SRB phase: quadratic terminal cost; the reset map to a following SRB phase is the identity.
template <int NT>
HD void srb_rollout_terminal(SrbLds& L, PhaseC& P, PhaseC* Pn, int b, double eps, SlotOut so, size_t slot, bool ss = false, bool wr = true) {
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + h) * 12;
    HS_PHASE(NT, if (tid < 12) { const double x = ss ? P.Xsim[kx + tid] : P.Xbar[kx + tid] + eps * P.dX[kx + tid]; L.x[tid] = x; if (wr) P.X[kx + tid] = x; })
    HS_PHASE(NT, if (tid == 0) {
        double s = 0; for (int i = 0; i < 12; i++) { const double d = L.x[i] - P.xr[(size_t)h * 12 + i]; s += d * P.qf[i] * d; }
        const double Phi = 0.5 * s;
        if (wr) { P.Phibase[b] = Phi; P.Phi[b] = Phi; }
        so.cost[slot] = Phi; so.ming[slot] = 0.0; so.maxh[slot] = 0.0; so.dsq[slot] = 0.0;
    })
    if (Pn == nullptr) return;
    const size_t nx = ((size_t)b * (Pn->h + 1)) * 12;
    HS_PHASE(NT, if (tid < 12) {
        const double xi = L.x[tid];
        if (wr) Pn->Xsim[nx + tid] = xi;
        const double d = (ss || !Pn->shooting) ? 0.0 : xi - (Pn->Xbar[nx + tid] + eps * Pn->dX[nx + tid]);     // no shooting node at the start of the next phase: X[0] = x_init
        if (wr) Pn->Defect[nx + tid] = d; L.red[tid] = d * d;
    })
    HS_PHASE(NT, if (tid == 0) { double s = 0; for (int i = 0; i < 12; i++) s += L.red[i]; so.dsq[slot] = s; })
}

// LQ approximation of SRB knot k < h: A = I + dt df/dx, B = dt df/du (SRBM.h:70-93), tracking cost + height barrier.
template <int NT>
HD void srb_lq_knot(SrbLds& L, PhaseC& P, int b, int k, int reb_active) {
    const int h = P.h; const double dt = P.dt;
    const size_t kx = ((size_t)b * (h + 1) + k) * 12, ku = ((size_t)b * h + k) * 12, kk = (size_t)b * h + k;
    HS_PHASE(NT, if (tid < 12) { L.x[tid] = P.X[kx + tid]; L.u[tid] = P.U[ku + tid]; })
    HS_PHASE(NT, if (tid < 24) {   // lane d: column d of [df/dx | df/du]
        Dual xs[12], us[12], out[12];
        for (int i = 0; i < 12; i++) { xs[i] = Dual(L.x[i], (tid == i) ? 1.0 : 0.0); us[i] = Dual(L.u[i], (tid == 12 + i) ? 1.0 : 0.0); }
        srb_xdot<Dual>(xs, us, P.foot_pos + (size_t)k * 12, P.ref_contact + (size_t)k * 4, out);
        for (int i = 0; i < 12; i++) L.AB[i + 12 * tid] = out[i].d * dt + ((tid == i) ? 1.0 : 0.0);
    } if (tid == 32) {
        double bd = 0, bdd = 0;
        if (P.go_height >= 0 && reb_active) {
            const size_t gi = kk * P.ng + P.go_height; const double g = P.g[gi], delta = P.delta[gi], e = P.eps[gi];
            if (g > delta) { bd = -1.0 / g; bdd = 1.0 / (g * g); } else { bd = (g - 2 * delta) / delta / delta; bdd = 1.0 / (delta * delta); }
            bd *= e; bdd *= e;
        }
        L.bd = bd; L.bdd = bdd;
    })
    HS_PHASE(NT, for (int e = tid; e < 144; e += NT) {
        const int r = e % 12, c = e / 12;
        rec_put(P, kk, P.oA + e, L.AB[e]); rec_put(P, kk, P.oB + e, L.AB[144 + e]);
        double qd = 0, rd = 0;
        if (r == c) { qd = dt * P.q[r]; rd = dt * P.r[r]; if (r == 2) qd += dt * L.bdd; }
        rec_put(P, kk, P.oLxx + e, qd); rec_put(P, kk, P.oLuu + e, rd);
    } if (tid < 12) {
        double lx = dt * P.q[tid] * (L.x[tid] - P.xr[(size_t)k * 12 + tid]); if (tid == 2) lx += dt * L.bd;
        rec_put(P, kk, P.oLx + tid, lx);
        rec_put(P, kk, P.oLu + tid, dt * P.r[tid] * (L.u[tid] - P.ur[(size_t)k * 12 + tid]));
    })
}

// Terminal partials of an SRB phase; Px = I (12 x 12) when another SRB phase follows.
template <int NT>
HD void srb_lq_terminal(SrbLds& L, PhaseC& P, PhaseC* Pn, int b) {
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + h) * 12;
    HS_PHASE(NT, if (tid < 12) P.Phix[(size_t)b * 12 + tid] = P.qf[tid] * (P.X[kx + tid] - P.xr[(size_t)h * 12 + tid]);
             for (int e = tid; e < 144; e += NT) { P.Phixx[(size_t)b * 144 + e] = (e % 12 == e / 12) ? P.qf[e % 12] : 0.0; if (Pn != nullptr) P.Px[(size_t)b * 144 + e] = (e % 12 == e / 12) ? 1.0 : 0.0; })
    (void)L;
}

}  // namespace hs
