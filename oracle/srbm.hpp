// ORACLE — TEST INFRASTRUCTURE ONLY.
// Closed-form restatement of the reference's single-rigid-body model, which the reference evaluates
// through CasADi-generated code: MHPC/MHPC-Trajopt/SRBM.h:43-93 (SRBDynamics / SRBDynamicsDerivatives,
// MHPC/MHPC-Trajopt/CasadiGen/source/SRBDynamics.cpp).  Constants (m, I) were read out of the generated
// code.  Pinned against that generated code: tests/golden/casadi_ref.npz.
// State x = [p(3), eul=(yaw,pitch,roll)(3), v(3), eul-rates(3)], u = 4 world-frame foot forces.
#pragma once
#include "wbm.hpp"

namespace orc {

inline Dual operator/(Dual a, Dual b) { return {a.v / b.v, (a.d * b.v - a.v * b.d) / (b.v * b.v)}; }

constexpr double SRB_M = 8.912;
constexpr double SRB_I[3][3] = {{0.061578036, 0, 5.38e-5}, {0, 0.2207093, 0}, {5.38e-5, 0, 0.272612336}};

template <class S>
inline void srb_xdot(const S* x, const S* u, const double* pf, const int* c, S* xd) {
    S yaw = x[3], th = x[4], ph = x[5];
    S dyaw = x[9], dth = x[10], dph = x[11];
    S cy = cos(yaw), sy = sin(yaw), ct = cos(th), st = sin(th), cp = cos(ph), sp = sin(ph);
    for (int i = 0; i < 3; i++) { xd[i] = x[6 + i]; xd[3 + i] = x[9 + i]; }
    V3<S> F{S(0.0), S(0.0), S(0.0)}, tauw{S(0.0), S(0.0), S(0.0)}, p{x[0], x[1], x[2]};
    for (int l = 0; l < 4; l++) if (c[l]) {
        V3<S> f{u[3 * l], u[3 * l + 1], u[3 * l + 2]};
        V3<S> r{S(pf[3 * l]) - p.x, S(pf[3 * l + 1]) - p.y, S(pf[3 * l + 2]) - p.z};
        F = F + f; tauw = tauw + cross(r, f);
    }
    xd[6] = F.x * S(1.0 / SRB_M); xd[7] = F.y * S(1.0 / SRB_M); xd[8] = F.z * S(1.0 / SRB_M) - S(GRAV);
    // R = Rz Ry Rx
    M3<S> R;
    R.m[0][0] = cy * ct; R.m[0][1] = cy * st * sp - sy * cp; R.m[0][2] = cy * st * cp + sy * sp;
    R.m[1][0] = sy * ct; R.m[1][1] = sy * st * sp + cy * cp; R.m[1][2] = sy * st * cp - cy * sp;
    R.m[2][0] = -st;     R.m[2][1] = ct * sp;                R.m[2][2] = ct * cp;
    V3<S> taub = mulT(R, tauw);
    // body rates w = T(eul) * eul_dot ; T = [(-st, sp ct, cp ct), (0, cp, -sp), (1,0,0)]
    V3<S> w{dph - st * dyaw, sp * ct * dyaw + cp * dth, cp * ct * dyaw - sp * dth};
    auto Imul = [&](V3<S> a) {
        return V3<S>{S(SRB_I[0][0]) * a.x + S(SRB_I[0][1]) * a.y + S(SRB_I[0][2]) * a.z,
                     S(SRB_I[1][0]) * a.x + S(SRB_I[1][1]) * a.y + S(SRB_I[1][2]) * a.z,
                     S(SRB_I[2][0]) * a.x + S(SRB_I[2][1]) * a.y + S(SRB_I[2][2]) * a.z};
    };
    V3<S> rhs = taub - cross(w, Imul(w));
    // solve I wd = rhs (I has only the xz coupling)
    double det = SRB_I[0][0] * SRB_I[2][2] - SRB_I[0][2] * SRB_I[0][2];
    V3<S> wd{(S(SRB_I[2][2]) * rhs.x - S(SRB_I[0][2]) * rhs.z) * S(1.0 / det), rhs.y * S(1.0 / SRB_I[1][1]),
             (S(SRB_I[0][0]) * rhs.z - S(SRB_I[0][2]) * rhs.x) * S(1.0 / det)};
    // Tdot * eul_dot
    V3<S> c0d{-(ct * dth), cp * ct * dph - sp * st * dth, -(sp * ct * dph) - cp * st * dth};
    V3<S> c1d{S(0.0), -(sp * dph), -(cp * dph)};
    V3<S> b = wd - (dyaw * c0d + dth * c1d);
    // eul_ddot = T^-1 b
    S ddyaw = (sp * b.y + cp * b.z) / ct;
    S ddth = cp * b.y - sp * b.z;
    S ddph = b.x + st * ddyaw;
    xd[9] = ddyaw; xd[10] = ddth; xd[11] = ddph;
}

inline void srb_dynamics(const double* x, const double* u, const double* pf, const int* c, double dt, double* xnext) {
    double xd[12]; srb_xdot<double>(x, u, pf, c, xd);
    for (int i = 0; i < 12; i++) xnext[i] = x[i] + xd[i] * dt;
}
// continuous-time Jacobians, column-major 12x12
inline void srb_partials_ct(const double* x, const double* u, const double* pf, const int* c, double* Ac, double* Bc) {
    Dual xdv[12], udv[12], out[12];
    for (int d = 0; d < 24; d++) {
        for (int i = 0; i < 12; i++) { xdv[i] = Dual(x[i]); udv[i] = Dual(u[i]); }
        if (d < 12) xdv[d].d = 1; else udv[d - 12].d = 1;
        srb_xdot<Dual>(xdv, udv, pf, c, out);
        for (int i = 0; i < 12; i++) (d < 12 ? Ac[i + 12 * d] : Bc[i + 12 * (d - 12)]) = out[i].d;
    }
}
inline void srb_dynamics_partial(const double* x, const double* u, const double* pf, const int* c, double dt, double* A, double* B) {
    srb_partials_ct(x, u, pf, c, A, B);
    for (int i = 0; i < 144; i++) { A[i] *= dt; B[i] *= dt; }
    for (int i = 0; i < 12; i++) A[i + 12 * i] += 1.0;
}

}  // namespace orc
