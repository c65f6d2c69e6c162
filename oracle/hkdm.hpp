// ORACLE — TEST INFRASTRUCTURE ONLY.
// Closed-form restatement of the reference's hybrid kinodynamic (HKD) model, which the reference evaluates through
// CasADi-generated code:
//   HKDMPC/HKD-TrajOpt/HKDModel.h:33-61            hkinodyn / hkinodyn_par (the generated function IS the Euler step)
//   HKDMPC/HKD-TrajOpt/HKDReset.h:41-136           reset map + partial (compute_foot_position, comp_foot_jacob_1..4)
//   HKDMPC/HKD-TrajOpt/HKDConstraints.cpp:61-170   touchdown constraint (foot height + its z-row Jacobian)
// Constants were read out of the generated code by numerical probing.
// Pinned against that generated code (oracle/_ref): tests/golden/casadi_ref.npz, keys hkd_*.
//   x = [eul=(yaw,pitch,roll), pos, omega_body, v, qdummy(12)], u = [F(12), qJdot(12)], legs FR, FL, HR, HL;
//   qdummy_l = foot position (stance) or the leg's joint angles (swing).
#pragma once
#include "srbm.hpp"

namespace orc {

constexpr double HKD_MASS = 8.912;
constexpr double HKD_I[3] = {0.02746078, 0.2425157968, 0.2651935768};

template <class S>
inline void hkd_step(const S* x, const S* u, double dt, const int* c, S* xn) {
    S yaw = x[0], th = x[1], ph = x[2];
    S cy = cos(yaw), sy = sin(yaw), ct = cos(th), st = sin(th), cp = cos(ph), sp = sin(ph);
    V3<S> w{x[6], x[7], x[8]}, p{x[3], x[4], x[5]};
    // Euler rates from body rates (inverse of the map used by the SRB model)
    S dyaw = (sp * w.y + cp * w.z) / ct;
    S dth = cp * w.y - sp * w.z;
    S dph = w.x + st * dyaw;
    V3<S> F{S(0.0), S(0.0), S(0.0)}, tw{S(0.0), S(0.0), S(0.0)};
    for (int l = 0; l < 4; l++) if (c[l]) {
        V3<S> f{u[3 * l], u[3 * l + 1], u[3 * l + 2]};
        V3<S> r{x[12 + 3 * l] - p.x, x[13 + 3 * l] - p.y, S(0.0) - p.z};   // the stance foot is taken ON the ground plane z = 0
        F = F + f; tw = tw + cross(r, f);
    }
    M3<S> R;
    R.m[0][0] = cy * ct; R.m[0][1] = cy * st * sp - sy * cp; R.m[0][2] = cy * st * cp + sy * sp;
    R.m[1][0] = sy * ct; R.m[1][1] = sy * st * sp + cy * cp; R.m[1][2] = sy * st * cp - cy * sp;
    R.m[2][0] = -st;     R.m[2][1] = ct * sp;                R.m[2][2] = ct * cp;
    V3<S> tb = mulT(R, tw);
    V3<S> Iw{S(HKD_I[0]) * w.x, S(HKD_I[1]) * w.y, S(HKD_I[2]) * w.z};
    V3<S> rhs = tb - cross(w, Iw);
    S d = S(dt);
    xn[0] = x[0] + d * dyaw; xn[1] = x[1] + d * dth; xn[2] = x[2] + d * dph;
    for (int i = 0; i < 3; i++) xn[3 + i] = x[3 + i] + d * x[9 + i];
    xn[6] = x[6] + d * (rhs.x * S(1.0 / HKD_I[0])); xn[7] = x[7] + d * (rhs.y * S(1.0 / HKD_I[1])); xn[8] = x[8] + d * (rhs.z * S(1.0 / HKD_I[2]));
    xn[9] = x[9] + d * (F.x * S(1.0 / HKD_MASS)); xn[10] = x[10] + d * (F.y * S(1.0 / HKD_MASS)); xn[11] = x[11] + d * (F.z * S(1.0 / HKD_MASS) - S(GRAV));
    for (int l = 0; l < 4; l++) for (int a = 0; a < 3; a++)
        xn[12 + 3 * l + a] = c[l] ? x[12 + 3 * l + a] : x[12 + 3 * l + a] + d * u[12 + 3 * l + a];
}
inline void hkd_dynamics(const double* x, const double* u, const int* c, double dt, double* xn) { hkd_step<double>(x, u, dt, c, xn); }
// A, B 24x24 column-major
inline void hkd_dynamics_partial(const double* x, const double* u, const int* c, double dt, double* A, double* B) {
    Dual xd[24], ud[24], out[24];
    for (int d = 0; d < 48; d++) {
        for (int i = 0; i < 24; i++) { xd[i] = Dual(x[i]); ud[i] = Dual(u[i]); }
        if (d < 24) xd[d].d = 1; else ud[d - 24].d = 1;
        hkd_step<Dual>(xd, ud, dt, c, out);
        for (int i = 0; i < 24; i++) (d < 24 ? A[i + 24 * d] : B[i + 24 * (d - 24)]) = out[i].d;
    }
}

// foot position of HKD leg l (0..3 = FR, FL, HR, HL) for body pose (pos, eul) and leg joint angles ql
// (compute_foot_position of the generated code; same tree as the whole-body model, thigh yaw offset psi)
template <class S>
inline V3<S> hkd_foot(const S* pos, const S* eul, const S* ql, int leg, double psi) {
    const double sx = leg < 2 ? 1.0 : -1.0, sy = (leg % 2 == 0) ? -1.0 : 1.0;
    auto rx = [](S a, V3<S> w) { S c = cos(a), s = sin(a); return V3<S>{w.x, c * w.y - s * w.z, s * w.y + c * w.z}; };
    auto ry = [](S a, V3<S> w) { S c = cos(a), s = sin(a); return V3<S>{c * w.x + s * w.z, w.y, c * w.z - s * w.x}; };
    auto rz = [](S a, V3<S> w) { S c = cos(a), s = sin(a); return V3<S>{c * w.x - s * w.y, s * w.x + c * w.y, w.z}; };
    V3<S> w{S(0.0), S(0.0), S(-0.195)};
    w = ry(ql[2], w); w.z = w.z - S(0.209);
    w = ry(ql[1], w); w = rz(S(psi), w); w.y = w.y + S(sy * 0.062);
    w = rx(ql[0], w); w.x = w.x + S(sx * 0.19); w.y = w.y + S(sy * 0.049);
    w = rz(eul[0], ry(eul[1], rx(eul[2], w)));
    return {w.x + pos[0], w.y + pos[1], w.z + pos[2]};
}
// pf and its Jacobian w.r.t. [pos(3), eul(3), ql(3)]  (3 x 9, row-major J[a][j])
inline void hkd_foot_jac(const double* pos, const double* eul, const double* ql, int leg, double psi, double* pf, double J[3][9]) {
    for (int d = -1; d < 9; d++) {
        Dual p[3], e[3], q[3];
        for (int i = 0; i < 3; i++) { p[i] = Dual(pos[i], d == i); e[i] = Dual(eul[i], d == 3 + i); q[i] = Dual(ql[i], d == 6 + i); }
        V3<Dual> f = hkd_foot<Dual>(p, e, q, leg, psi);
        if (d < 0) { if (pf) { pf[0] = f.x.v; pf[1] = f.y.v; pf[2] = f.z.v; } }
        else { J[0][d] = f.x.d; J[1][d] = f.y.d; J[2][d] = f.z.d; }
    }
}

// HKDReset::resetmap (HKDReset.h:41-76)
inline void hkd_resetmap(const double* x, const int* c, const int* cn, double psi, double* xn) {
    std::memcpy(xn, x, sizeof(double) * 24);
    for (int l = 0; l < 4; l++) {
        if (c[l] && !cn[l]) { xn[12 + 3 * l] = 0.0; xn[13 + 3 * l] = -0.8; xn[14 + 3 * l] = 1.7; }
        if (!c[l] && cn[l]) {
            double pf[3]; double J[3][9]; hkd_foot_jac(x + 3, x, x + 12 + 3 * l, l, psi, pf, J);
            xn[12 + 3 * l] = pf[0]; xn[13 + 3 * l] = pf[1]; xn[14 + 3 * l] = 0.0;
        }
    }
}
// HKDReset::resetmap_partial (HKDReset.h:78-136); Px 24x24 column-major
inline void hkd_resetmap_partial(const double* x, const int* c, const int* cn, double psi, double* Px) {
    std::memset(Px, 0, sizeof(double) * 576);
    for (int i = 0; i < 24; i++) Px[i + 24 * i] = 1.0;
    for (int l = 0; l < 4; l++) {
        if (c[l] && !cn[l]) for (int a = 0; a < 3; a++) for (int j = 0; j < 24; j++) Px[(12 + 3 * l + a) + 24 * j] = 0.0;
        if (!c[l] && cn[l]) {
            double J[3][9]; hkd_foot_jac(x + 3, x, x + 12 + 3 * l, l, psi, nullptr, J);
            for (int a = 0; a < 3; a++) {
                const double cm = a < 2 ? 1.0 : 0.0;   // cmap = (1,1,0)
                const int r = 12 + 3 * l + a;
                for (int j = 0; j < 3; j++) { Px[r + 24 * j] = cm * J[a][3 + j]; Px[r + 24 * (3 + j)] = cm * J[a][j]; }
                for (int j = 0; j < 12; j++) Px[r + 24 * (12 + j)] = (j / 3 == l) ? cm * J[a][6 + j % 3] : 0.0;
            }
        }
    }
}

}  // namespace orc
