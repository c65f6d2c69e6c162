// Hybrid kinodynamic phases (n = 24, m = 24, p = 0) of the HKD-MPC problem: per-knot rollout and LQ approximation, one
// wavefront per knot.  The reference evaluates this model through CasADi-generated code (HKDMPC/HKD-TrajOpt/HKDModel.h:33-61
// hkinodyn / hkinodyn_par; HKDReset.h:41-136 and HKDConstraints.cpp:75-170 compute_foot_position / comp_foot_jacob_1..4);
// here the body part of the Euler step and the leg kinematics are closed forms templated on the scalar, and Jacobians
// come from forward-mode lanes.
//   x = [eul=(yaw,pitch,roll), pos, omega_body, v, qdummy(12)], u = [F(12), qJdot(12)], legs FR, FL, HR, HL;
//   qdummy_l = planned foothold (stance: only its x, y enter the dynamics, the foot is taken on z = 0) or joint angles (swing).
#pragma once
#include "hs_types.hpp"
#include "srb_knot.hpp"   // srb_recip, SlotOut, reb_barrier

namespace hs {

constexpr double HKD_MASS = 8.912;
constexpr double HKD_IXX = 0.02746078, HKD_IYY = 0.2425157968, HKD_IZZ = 0.2651935768;   // generated code

// body rows of x+ = x + dt f(x, u): xb = x[0:12], pxy[2l + a] = stance-foot x, y, F = u[0:12]
template <class S>
HD void hkd_body_step(const S* xb, const S* pxy, const S* F, double dt, const int* contact, S* out) {
    S sy, cy, st, ct, sp, cp;
    sincos_(xb[0], sy, cy); sincos_(xb[1], st, ct); sincos_(xb[2], sp, cp);
    const V3<S> w{xb[6], xb[7], xb[8]};
    const S dyaw = (sp * w.y + cp * w.z) * srb_recip(ct);
    const S dth = cp * w.y - sp * w.z;
    const S dph = w.x + st * dyaw;
    V3<S> Fs{S(0.0), S(0.0), S(0.0)}, tw{S(0.0), S(0.0), S(0.0)};
    for (int l = 0; l < 4; l++) if (contact[l]) {
        const V3<S> f{F[3 * l], F[3 * l + 1], F[3 * l + 2]};
        const V3<S> r{pxy[2 * l] - xb[3], pxy[2 * l + 1] - xb[4], -xb[5]};
        Fs = Fs + f; tw = tw + cross(r, f);
    }
    const V3<S> tb = rotT<0, S>(cp, sp, rotT<1, S>(ct, st, rotT<2, S>(cy, sy, tw)));
    const V3<S> Iw{HKD_IXX * w.x, HKD_IYY * w.y, HKD_IZZ * w.z};
    const V3<S> rhs = tb - cross(w, Iw);
    out[0] = xb[0] + dt * dyaw; out[1] = xb[1] + dt * dth; out[2] = xb[2] + dt * dph;
    for (int i = 0; i < 3; i++) out[3 + i] = xb[3 + i] + dt * xb[9 + i];
    out[6] = xb[6] + dt * (rhs.x * (1.0 / HKD_IXX)); out[7] = xb[7] + dt * (rhs.y * (1.0 / HKD_IYY)); out[8] = xb[8] + dt * (rhs.z * (1.0 / HKD_IZZ));
    out[9] = xb[9] + dt * (Fs.x * (1.0 / HKD_MASS)); out[10] = xb[10] + dt * (Fs.y * (1.0 / HKD_MASS)); out[11] = xb[11] + dt * (Fs.z * (1.0 / HKD_MASS) - GRAV);
}

// foot position of HKD leg l for body pose (pos, eul) and leg joint angles ql; same tree as the whole-body model with the
// thigh yaw offset of the kinematic terms (cpsi, spsi)
template <class S>
HD V3<S> hkd_foot(const S* pos, const S* eul, const S* ql, int leg, double cpsi, double spsi) {
    const double sx = leg < 2 ? 1.0 : -1.0, sgy = (leg % 2 == 0) ? -1.0 : 1.0;
    S s0, c0, s1, c1, s2, c2, sy, cy, st, ct, sp, cp;
    sincos_(ql[0], s0, c0); sincos_(ql[1], s1, c1); sincos_(ql[2], s2, c2);
    sincos_(eul[0], sy, cy); sincos_(eul[1], st, ct); sincos_(eul[2], sp, cp);
    V3<S> w{S(0.0), S(0.0), S(-0.195)};
    w = rot<1, S>(c2, s2, w); w.z = w.z - 0.209;
    w = rot<1, S>(c1, s1, w); w = rot<2, S>(cpsi, spsi, w); w.y = w.y + sgy * 0.062;
    w = rot<0, S>(c0, s0, w); w.x = w.x + sx * 0.19; w.y = w.y + sgy * 0.049;
    w = rot<2, S>(cy, sy, rot<1, S>(ct, st, rot<0, S>(cp, sp, w)));
    return {w.x + pos[0], w.y + pos[1], w.z + pos[2]};
}

struct HkdLds {
    double x[24], xb[24], u[24], xn[24], red[24], tmp[24];
    double K[576];                   // rollout: feedback gain ; LQ: AB[12][32] body-row Jacobian columns
    double gval[20], bar[20], bd[20], bdd[20];
    double J[4 * 27];                // terminal: d foot / d [pos, eul, ql] per touchdown foot, [l][a][9]
    double pf[12], hx[4 * 24], coefg[4], coefh[4];
};

// value of GRF-pyramid row r (0..4) of stance foot f on the force part of u (HKDConstraints.cpp:17-33)
HD double hkd_grf_row(const double* u, int f, int r, double mu) {
    const double fx = u[3 * f], fy = u[3 * f + 1], fz = u[3 * f + 2];
    return r == 0 ? fz : r == 1 ? -fx + mu * fz : r == 2 ? fx + mu * fz : r == 3 ? -fy + mu * fz : fy + mu * fz;
}
// sum_f,a d (c_f w_a) d of the foot-placement regulariser (HKDCost.cpp:4-19) at knot k
HD double hkd_footreg(PhaseC& P, const double* x, int k) {
    const double* fp = P.foot_pos + (size_t)k * 12; const double* bp = P.body_pos + (size_t)k * 3;
    double s = 0;
    for (int f = 0; f < 4; f++) for (int a = 0; a < 3; a++) { const double d = (x[12 + 3 * f + a] - x[3 + a]) - (fp[3 * f + a] - bp[a]); s += d * (P.contact[f] * P.w_foot_reg[a]) * d; }
    return s;
}

template <int NT>
HD void hkd_rollout_knot(HkdLds& L, PhaseC& P, int b, int k, double eps, int reb_active, const double* x0, SlotOut so, size_t slot, int* fail_flag, bool ss = false, bool wr = true) {
    // ss: single shooting (MS = false, MultiPhaseDDP.cpp:65-68): X[k] is the state the previous knot of this wave simulated (Xsim[k]), no defect
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + k) * 24, ku = ((size_t)b * h + k) * 24, kk = (size_t)b * h + k;
    HS_PHASE(NT, if (tid < 24) { double xb = P.Xbar[kx + tid], x = ss ? P.Xsim[kx + tid] : xb + eps * P.dX[kx + tid]; L.xb[tid] = xb; L.x[tid] = x; if (wr) { P.X[kx + tid] = x; if (ss && k == 0) P.Defect[kx + tid] = 0.0; } }
             for (int i = tid; i < 576; i += NT) L.K[i] = P.K[kk * 576 + i];)
    HS_PHASE(NT, if (tid < 24) {
        double s = 0; for (int j = 0; j < 24; j++) s += L.K[tid + 24 * j] * (L.x[j] - L.xb[j]);
        const double u = P.Ubar[ku + tid] + eps * P.dU[ku + tid] + s;
        L.u[tid] = u; if (wr) P.U[ku + tid] = u;
    })
    HS_PHASE(NT, if (tid == 0) {
        double pxy[8]; for (int l = 0; l < 4; l++) { pxy[2 * l] = L.x[12 + 3 * l]; pxy[2 * l + 1] = L.x[13 + 3 * l]; }
        hkd_body_step<double>(L.x, pxy, L.u, P.dt, P.contact, L.xn);
    } else if (tid >= 32 && tid < 44) { const int j = tid - 32; L.xn[12 + j] = P.contact[j / 3] ? L.x[12 + j] : L.x[12 + j] + P.dt * L.u[12 + j]; })
    HS_PHASE(NT, if (tid < 24) {
        const double xs = L.xn[tid];
        if (wr) P.Xsim[kx + 24 + tid] = xs;
        const double d = ss ? 0.0 : xs - (P.Xbar[kx + 24 + tid] + eps * P.dX[kx + 24 + tid]);
        if (wr) P.Defect[kx + 24 + tid] = d;
        double dsq = d * d;
        if (x0 != nullptr && k == 0) { const double d0 = x0[(size_t)b * 24 + tid] - L.x[tid]; if (wr) { P.Xsim[kx + tid] = x0[(size_t)b * 24 + tid]; P.Defect[kx + tid] = d0; } dsq += d0 * d0; }
        L.red[tid] = dsq; L.tmp[tid] = xs * xs;
    } else if (tid >= 32 && tid < 32 + P.ng) {
        const int c = tid - 32; const size_t gi = kk * P.ng + c;
        const double g = hkd_grf_row(L.u, P.feet[c / 5], c % 5, P.mu);
        L.gval[c] = g; if (wr) P.g[gi] = g; L.bar[c] = P.eps[gi] * reb_barrier(g, P.delta[gi]);
    })
    HS_PHASE(NT, if (tid == 0) {
        double lq = 0, lr = 0;
        for (int i = 0; i < 24; i++) { const double d = L.x[i] - P.xr[(size_t)k * 24 + i]; lq += d * P.q[i] * d; }
        for (int i = 0; i < 24; i++) { const double d = L.u[i] - P.ur[(size_t)k * 24 + i]; lr += d * P.r[i] * d; }
        double l = 0.5 * lq; l += 0.5 * lr; l *= P.dt;
        if (P.w_foot_reg[0] >= 0) { double t = .5 * hkd_footreg(P, L.x, k); t *= P.dt; l += t; }
        if (wr) P.lbase[kk] = l;
        double ming = 0;
        if (P.ng > 0) {
            double c = 0; for (int i = 0; i < P.ng; i++) { c += L.bar[i]; ming = fmin(ming, L.gval[i]); }
            if (reb_active) l += P.dt * c;
        }
        if (wr) P.l[kk] = l;
        double dsq = 0, nsq = 0; for (int i = 0; i < 24; i++) { dsq += L.red[i]; nsq += L.tmp[i]; }
        so.cost[slot] = l; so.dsq[slot] = dsq; so.ming[slot] = ming; so.maxh[slot] = 0.0;
        if (sqrt(nsq) > 1e6 || !(nsq == nsq)) fail_flag[b] = 1;
    })
}

// Terminal knot of an HKD phase: terminal cost, touchdown constraint (foot height of the legs about to land), reset map
// (HKDReset.h:41-76: lift-off -> default joint angles, touchdown -> foot projected on the ground) into the next phase.
template <int NT>
HD void hkd_rollout_terminal(HkdLds& L, PhaseC& P, PhaseC* Pn, const ModelDev& md, int b, double eps, int al_active, SlotOut so, size_t slot, bool ss = false, bool wr = true) {
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + h) * 24;
    HS_PHASE(NT, if (tid < 24) { const double x = ss ? P.Xsim[kx + tid] : P.Xbar[kx + tid] + eps * P.dX[kx + tid]; L.x[tid] = x; if (wr) P.X[kx + tid] = x; })
    HS_PHASE(NT, if (tid < 4 && P.td[tid]) {
        const V3<double> f = hkd_foot<double>(L.x + 3, L.x, L.x + 12 + 3 * tid, tid, md.cpsi_kin, md.spsi_kin);
        L.pf[3 * tid] = f.x; L.pf[3 * tid + 1] = f.y; L.pf[3 * tid + 2] = f.z;
    })
    HS_PHASE(NT, if (tid == 0) {
        double s = 0; for (int i = 0; i < 24; i++) { const double d = L.x[i] - P.xr[(size_t)h * 24 + i]; s += d * P.qf[i] * d; }
        double pb = 0.5 * s;
        if (P.w_foot_reg[0] >= 0) pb += 10 * hkd_footreg(P, L.x, h);
        if (wr) P.Phibase[b] = pb;
        double maxh = 0, c = 0; int i = 0;
        for (int f = 0; f < 4; f++) if (P.td[f] && P.nt > 0) {
            const double hh = L.pf[3 * f + 2] - P.ground_height; if (wr) P.th[(size_t)b * P.nt + i] = hh; maxh = fmax(maxh, fabs(hh));
            const double sg = P.sigma[(size_t)b * P.nt + i], lm = P.lambda[(size_t)b * P.nt + i];
            c += 0.5 * sg * hh * hh; c += lm * hh; i++;
        }
        double Phi = pb; if (al_active && P.nt > 0) Phi += c;
        if (wr) P.Phi[b] = Phi;
        so.cost[slot] = Phi; so.ming[slot] = 0.0; so.maxh[slot] = maxh; so.dsq[slot] = 0.0;
    })
    if (Pn == nullptr) return;
    const size_t nx = ((size_t)b * (Pn->h + 1)) * 24;
    HS_PHASE(NT, if (tid < 24) {
        double xi = L.x[tid];
        if (tid >= 12) {
            const int l = (tid - 12) / 3, a = (tid - 12) % 3;
            if (P.contact[l] && !P.next_contact[l]) xi = (a == 0) ? 0.0 : (a == 1) ? -0.8 : 1.7;
            if (!P.contact[l] && P.next_contact[l]) xi = (a < 2) ? L.pf[3 * l + a] : 0.0;
        }
        if (wr) Pn->Xsim[nx + tid] = xi;
        const double d = (ss || !Pn->shooting) ? 0.0 : xi - (Pn->Xbar[nx + tid] + eps * Pn->dX[nx + tid]);     // no shooting node at the start of the next phase: X[0] = x_init
        if (wr) Pn->Defect[nx + tid] = d; L.red[tid] = d * d;
    })
    HS_PHASE(NT, if (tid == 0) { double s = 0; for (int i = 0; i < 24; i++) s += L.red[i]; so.dsq[slot] = s; })
}

// LQ approximation of HKD knot k < h.  32 forward-mode lanes (12 body states, 8 stance-foot x/y, 12 forces) give the body
// rows of [A B]; the qdummy rows are the identity plus dt (1 - c_l) on the joint-velocity inputs.
template <int NT>
HD void hkd_lq_knot(HkdLds& L, PhaseC& P, int b, int k, int reb_active) {
    const int h = P.h; const double dt = P.dt;
    const size_t kx = ((size_t)b * (h + 1) + k) * 24, ku = ((size_t)b * h + k) * 24, kk = (size_t)b * h + k;
    double* AB = L.K;   // [row 0..11][lane 0..31]
    HS_PHASE(NT, if (tid < 24) { L.x[tid] = P.X[kx + tid]; L.u[tid] = P.U[ku + tid]; })
    HS_PHASE(NT, if (tid < 32) {
        Dual xb[12], pxy[8], F[12], out[12];
        for (int i = 0; i < 12; i++) { xb[i] = Dual(L.x[i], (tid == i) ? 1.0 : 0.0); F[i] = Dual(L.u[i], (tid == 20 + i) ? 1.0 : 0.0); }
        for (int i = 0; i < 8; i++) pxy[i] = Dual(L.x[12 + 3 * (i / 2) + (i % 2)], (tid == 12 + i) ? 1.0 : 0.0);
        hkd_body_step<Dual>(xb, pxy, F, dt, P.contact, out);
        for (int i = 0; i < 12; i++) AB[i * 32 + tid] = out[i].d;
    } else if (tid >= 32 && tid < 32 + P.ng) {
        const int c = tid - 32; const size_t gi = kk * P.ng + c;
        const double g = P.g[gi], delta = P.delta[gi], e = P.eps[gi]; double bd, bdd;
        if (g > delta) { bd = -1.0 / g; bdd = 1.0 / (g * g); } else { bd = (g - 2 * delta) / delta / delta; bdd = 1.0 / (delta * delta); }
        L.bd[c] = reb_active ? e * bd : 0.0; L.bdd[c] = reb_active ? e * bdd : 0.0;
    })
    const double fr = (P.w_foot_reg[0] >= 0) ? 1.0 : 0.0;
    HS_PHASE(NT, for (int e = tid; e < 576; e += NT) {
        const int r = e % 24, c = e / 24;
        double a, bb;
        if (r < 12) {
            if (c < 12) a = AB[r * 32 + c];
            else { const int l = (c - 12) / 3, ax = (c - 12) % 3; a = (ax < 2) ? AB[r * 32 + 12 + 2 * l + ax] : 0.0; }
            bb = (c < 12) ? AB[r * 32 + 20 + c] : 0.0;
        } else {
            a = (r == c) ? 1.0 : 0.0;
            bb = (r == c && !P.contact[(r - 12) / 3]) ? dt : 0.0;
        }
        rec_put(P, kk, P.oA + e, a); rec_put(P, kk, P.oB + e, bb);
        // lxx: tracking diagonal + foot-placement regulariser dprel_dx' Q dprel_dx (HKDCost.cpp:22-35)
        double q = (r == c) ? dt * P.q[r] : 0.0;
        if (fr != 0.0) {
            if (r == c && r >= 3 && r < 6) { for (int f = 0; f < 4; f++) q += dt * P.contact[f] * P.w_foot_reg[r - 3]; }
            else if (r == c && r >= 12) q += dt * P.contact[(r - 12) / 3] * P.w_foot_reg[(r - 12) % 3];
            else if (r >= 3 && r < 6 && c >= 12 && (c - 12) % 3 == r - 3) q -= dt * P.contact[(c - 12) / 3] * P.w_foot_reg[r - 3];
            else if (c >= 3 && c < 6 && r >= 12 && (r - 12) % 3 == c - 3) q -= dt * P.contact[(r - 12) / 3] * P.w_foot_reg[c - 3];
        }
        rec_put(P, kk, P.oLxx + e, q);
        // luu: tracking diagonal + GRF barrier 3x3 block of a stance foot (rows [0 0 1; -1 0 mu; 1 0 mu; 0 -1 mu; 0 1 mu])
        double uu = (r == c) ? dt * P.r[r] : 0.0;
        if (P.go_grf >= 0 && r < 12 && c < 12 && r / 3 == c / 3) {
            const int f = r / 3; int a2 = -1; for (int t = 0; t < P.nc; t++) if (P.feet[t] == f) a2 = t;
            if (a2 >= 0) for (int cc = 0; cc < 5; cc++) {
                const double r0 = (cc == 1) ? -1.0 : (cc == 2) ? 1.0 : 0.0, r1 = (cc == 3) ? -1.0 : (cc == 4) ? 1.0 : 0.0, r2 = (cc == 0) ? 1.0 : P.mu;
                const double vr = (r % 3 == 0) ? r0 : (r % 3 == 1) ? r1 : r2, vc = (c % 3 == 0) ? r0 : (c % 3 == 1) ? r1 : r2;
                uu += dt * L.bdd[P.go_grf + 5 * a2 + cc] * vr * vc;
            }
        }
        rec_put(P, kk, P.oLuu + e, uu);
    } if (tid < 24) {
        const int i = tid;
        double lx = dt * P.q[i] * (L.x[i] - P.xr[(size_t)k * 24 + i]);
        if (fr != 0.0) {
            const double* fp = P.foot_pos + (size_t)k * 12; const double* bp = P.body_pos + (size_t)k * 3;
            if (i >= 3 && i < 6) { for (int f = 0; f < 4; f++) lx -= dt * P.contact[f] * P.w_foot_reg[i - 3] * ((L.x[12 + 3 * f + i - 3] - L.x[i]) - (fp[3 * f + i - 3] - bp[i - 3])); }
            else if (i >= 12) { const int f = (i - 12) / 3, a = (i - 12) % 3; lx += dt * P.contact[f] * P.w_foot_reg[a] * ((L.x[i] - L.x[3 + a]) - (fp[3 * f + a] - bp[a])); }
        }
        rec_put(P, kk, P.oLx + i, lx);
        double lu = dt * P.r[i] * (L.u[i] - P.ur[(size_t)k * 24 + i]);
        if (P.go_grf >= 0 && i < 12) {
            const int f = i / 3; int a2 = -1; for (int t = 0; t < P.nc; t++) if (P.feet[t] == f) a2 = t;
            if (a2 >= 0) for (int cc = 0; cc < 5; cc++) {
                const double r0 = (cc == 1) ? -1.0 : (cc == 2) ? 1.0 : 0.0, r1 = (cc == 3) ? -1.0 : (cc == 4) ? 1.0 : 0.0, r2 = (cc == 0) ? 1.0 : P.mu;
                lu += dt * L.bd[P.go_grf + 5 * a2 + cc] * ((i % 3 == 0) ? r0 : (i % 3 == 1) ? r1 : r2);
            }
        }
        rec_put(P, kk, P.oLu + i, lu);
    })
}

// Terminal partials of an HKD phase (+ AL on the touchdown heights) and the reset-map partial Px (HKDReset.h:78-136).
template <int NT>
HD void hkd_lq_terminal(HkdLds& L, PhaseC& P, PhaseC* Pn, const ModelDev& md, int b, int al_active) {
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + h) * 24;
    HS_PHASE(NT, if (tid < 24) L.x[tid] = P.X[kx + tid]; for (int i = tid; i < 96; i += NT) L.hx[i] = 0.0; for (int i = tid; i < 108; i += NT) L.J[i] = 0.0;)
    // lanes 6 l + d (d: eul 0..2, ql 0..2) of the touchdown legs: one column of the foot Jacobian each; d/dpos = I
    HS_PHASE(NT, if (tid < 24 && P.td[tid / 6]) {
        const int l = tid / 6, d = tid % 6;
        Dual pos[3], eul[3], ql[3];
        for (int i = 0; i < 3; i++) { pos[i] = Dual(L.x[3 + i]); eul[i] = Dual(L.x[i], (d == i) ? 1.0 : 0.0); ql[i] = Dual(L.x[12 + 3 * l + i], (d == 3 + i) ? 1.0 : 0.0); }
        const V3<Dual> f = hkd_foot<Dual>(pos, eul, ql, l, md.cpsi_kin, md.spsi_kin);
        L.J[l * 27 + 0 * 9 + 3 + d] = f.x.d; L.J[l * 27 + 1 * 9 + 3 + d] = f.y.d; L.J[l * 27 + 2 * 9 + 3 + d] = f.z.d;
        if (d < 3) L.J[l * 27 + d * 9 + d] = 1.0;
    })
    // AL gradient / curvature coefficients and hx rows of the touchdown constraints (HKDConstraints.cpp:113-170, ConstraintsBase.h:412-425)
    HS_PHASE(NT, if (tid == 0) {
        int t = 0;
        for (int f = 0; f < 4; f++) if (P.td[f] && P.nt > 0) {
            const double sg = P.sigma[(size_t)b * P.nt + t], lm = P.lambda[(size_t)b * P.nt + t], hh = P.th[(size_t)b * P.nt + t];
            L.coefg[t] = al_active ? sg * hh + lm : 0.0; L.coefh[t] = al_active ? sg * (1 + hh) + lm : 0.0;
            for (int j = 0; j < 3; j++) { L.hx[t * 24 + j] = L.J[f * 27 + 2 * 9 + 3 + j]; L.hx[t * 24 + 3 + j] = L.J[f * 27 + 2 * 9 + j]; L.hx[t * 24 + 12 + 3 * f + j] = L.J[f * 27 + 2 * 9 + 6 + j]; }
            t++;
        }
    })
    const double fr = (P.w_foot_reg[0] >= 0) ? 20.0 : 0.0;
    HS_PHASE(NT, for (int e = tid; e < 576; e += NT) {
        const int r = e % 24, c = e / 24;
        double q = (r == c) ? P.qf[r] : 0.0;
        if (fr != 0.0) {
            if (r == c && r >= 3 && r < 6) { for (int f = 0; f < 4; f++) q += fr * P.contact[f] * P.w_foot_reg[r - 3]; }
            else if (r == c && r >= 12) q += fr * P.contact[(r - 12) / 3] * P.w_foot_reg[(r - 12) % 3];
            else if (r >= 3 && r < 6 && c >= 12 && (c - 12) % 3 == r - 3) q -= fr * P.contact[(c - 12) / 3] * P.w_foot_reg[r - 3];
            else if (c >= 3 && c < 6 && r >= 12 && (r - 12) % 3 == c - 3) q -= fr * P.contact[(r - 12) / 3] * P.w_foot_reg[c - 3];
        }
        for (int t = 0; t < P.nt; t++) q += L.coefh[t] * (L.hx[t * 24 + r] * L.hx[t * 24 + c]);
        P.Phixx[(size_t)b * 576 + e] = q;
        if (Pn != nullptr) {   // Px
            double v = (r == c) ? 1.0 : 0.0;
            if (r >= 12) {
                const int l = (r - 12) / 3, a = (r - 12) % 3;
                if (P.contact[l] && !P.next_contact[l]) v = 0.0;
                if (!P.contact[l] && P.next_contact[l]) {
                    if (a == 2) v = 0.0;
                    else if (c < 3) v = L.J[l * 27 + a * 9 + 3 + c];
                    else if (c < 6) v = L.J[l * 27 + a * 9 + (c - 3)];
                    else if (c >= 12 && (c - 12) / 3 == l) v = L.J[l * 27 + a * 9 + 6 + (c - 12) % 3];
                    else v = 0.0;
                }
            }
            P.Px[(size_t)b * 576 + e] = v;
        }
    } if (tid < 24) {
        const int i = tid;
        double px = P.qf[i] * (L.x[i] - P.xr[(size_t)h * 24 + i]);
        if (fr != 0.0) {
            const double* fp = P.foot_pos + (size_t)h * 12; const double* bp = P.body_pos + (size_t)h * 3;
            if (i >= 3 && i < 6) { for (int f = 0; f < 4; f++) px -= fr * P.contact[f] * P.w_foot_reg[i - 3] * ((L.x[12 + 3 * f + i - 3] - L.x[i]) - (fp[3 * f + i - 3] - bp[i - 3])); }
            else if (i >= 12) { const int f = (i - 12) / 3, a = (i - 12) % 3; px += fr * P.contact[f] * P.w_foot_reg[a] * ((L.x[i] - L.x[3 + a]) - (fp[3 * f + a] - bp[a])); }
        }
        for (int t = 0; t < P.nt; t++) px += L.coefg[t] * L.hx[t * 24 + i];
        P.Phix[(size_t)b * 24 + i] = px;
    })
}

}  // namespace hs
