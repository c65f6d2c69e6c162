// Mini-Cheetah whole-body model, one Newton-Euler pass PER LANE (gfx950 wave64).
//
// Replaces, for the reference, the Pinocchio calls of MHPC/MHPC-Trajopt/WBM.cpp:368-543 (forwardKinematics,
// computeJointJacobians, crba, nonLinearEffects, computeRNEADerivatives, getFrame*) and the CasADi functions
// footVel/Acc/ForcePartial* (WBM.cpp:565-675).  Tree constants: urdf/mini_cheetah_simple_correctedInertia.urdf,
// joint order PinocchioInteface.cpp:17-56 (PX,PY,PZ,RZ,RY,RX + 4 x (RX,RY,RY)).
//
// MI355X mapping: the pass is linear in the joint acceleration and exact under forward-mode tangents, so a
// 64-lane wave evaluates MANY passes at once, each lane with its own (velocity scale, unit acceleration,
// gravity, tangent seed, thigh-yaw constant) and the shared knot state read from LDS by broadcast:
//   * lanes 0..17, S=double, v=0, a=e_lane, g=0      -> column `lane` of M(q) and of every foot Jacobian
//   * lane 18,     S=double, v=v,  a=0,     g on     -> h(q,v), J̇v, foot positions / velocities
//   * lanes 0..35, S=Dual, tangent seed on x_lane    -> d ID(q,v,qdd)/dx  (computeRNEADerivatives)
//   * lanes 36..53, S=Dual, massless, foot forces on -> d(J^T F)/dq, d(foot acc)/dq, d(foot vel)/dq
// No lane keeps more than one leg's chain live (register budget), nothing is spilled to scratch by design.
#pragma once
#include "hs_common.hpp"

namespace hs {

struct LaneCfg {
    double cpsi, spsi;   // thigh fixed yaw (quirk xii: 3.1415 for Pinocchio-equivalent terms, pi for CasADi-equivalent)
    double mscale;       // 1: link inertias on, 0: massless (pure kinematics / external-force torques)
    double grav;         // fictitious base acceleration (+9.81 or 0)
    double fscale;       // scale of the external foot forces
    double vscale;       // multiplies the knot velocity (0 for the M columns)
    double ascale;       // multiplies the knot acceleration input
    int aunit;           // >=0: acceleration = e_aunit (ascale ignored)
    int tq, tv;          // tangent seed index into q or v (-1: none)
    int l0 = 0, l1 = 4;  // legs this lane walks (a pass restricted to ONE leg is a per-leg TASK: its base wrench is a partial sum)
    bool body = true;    // include the trunk's own inertial force (exactly one task per seed does)
    bool partial = false; // true: hand the accumulated base wrench to sink.base() instead of running the base joints backward
};

// Outputs leave the pass through a SINK (sk.tau(i, value), sk.foot(l, pos, vel, acc)) that stores straight into LDS,
// so no lane holds an output array (dynamic indexing of per-lane arrays would go to scratch memory).
// sin/cos of the 18 joint angles are computed ONCE per knot into LDS (cs/sn) and shared by every lane and pass.
// link inertial constants (link frame, about COM): m, c, Ixx Ixy Ixz Iyy Iyz Izz  — sy = +1 left, -1 right

template <class S>
HD void inertia_force(double ms, double m0, double cx, double cy, double cz, double Ixx0, double Ixy0, double Ixz0, double Iyy0, double Iyz0, double Izz0,
                      const V3<S>& om, const V3<S>& vl, const V3<S>& al, const V3<S>& aa, V3<S>& n, V3<S>& f) {
    // spatial momentum h = I v ; force = I a + v x* h   (link-local Pluecker coordinates); ms = 0 switches the link off
    const double m = ms * m0, Ixx = ms * Ixx0, Ixy = ms * Ixy0, Ixz = ms * Ixz0, Iyy = ms * Iyy0, Iyz = ms * Iyz0, Izz = ms * Izz0;
    V3<S> hl = scaled(m, vl + crossc(om, cx, cy, cz));
    V3<S> Iw = {Ixx * om.x + Ixy * om.y + Ixz * om.z, Ixy * om.x + Iyy * om.y + Iyz * om.z, Ixz * om.x + Iyz * om.y + Izz * om.z};
    V3<S> ha = Iw + ccross(cx, cy, cz, hl);
    V3<S> ff = scaled(m, al + crossc(aa, cx, cy, cz));
    V3<S> Ia = {Ixx * aa.x + Ixy * aa.y + Ixz * aa.z, Ixy * aa.x + Iyy * aa.y + Iyz * aa.z, Ixz * aa.x + Iyz * aa.y + Izz * aa.z};
    V3<S> nn = Ia + ccross(cx, cy, cz, ff);
    f = ff + cross(om, hl);
    n = nn + cross(om, ha) + cross(vl, hl);
}

// child-frame motion after a revolute joint about AX with angle (c,s), rate qd, acceleration qdd.
// (om,vl,aa,al) come in parent coordinates ALREADY translated to the joint origin.
template <int AX, class S>
HD void rev_joint(S c, S s, S qd, S qdd, V3<S>& om, V3<S>& vl, V3<S>& aa, V3<S>& al) {
    V3<S> o = rotT<AX>(c, s, om), v = rotT<AX>(c, s, vl), a2 = rotT<AX>(c, s, aa), a1 = rotT<AX>(c, s, al);
    // S*qd = (e*qd, 0):  alpha += e*qdd + om_new x e*qd ; a_lin += v x e*qd
    if (AX == 0) { o.x = o.x + qd; a2.x = a2.x + qdd; a2.y = a2.y + o.z * qd; a2.z = a2.z - o.y * qd; a1.y = a1.y + v.z * qd; a1.z = a1.z - v.y * qd; }
    if (AX == 1) { o.y = o.y + qd; a2.y = a2.y + qdd; a2.x = a2.x - o.z * qd; a2.z = a2.z + o.x * qd; a1.x = a1.x - v.z * qd; a1.z = a1.z + v.x * qd; }
    if (AX == 2) { o.z = o.z + qd; a2.z = a2.z + qdd; a2.x = a2.x + o.y * qd; a2.y = a2.y - o.x * qd; a1.x = a1.x + v.y * qd; a1.y = a1.y - v.x * qd; }
    om = o; vl = v; aa = a2; al = a1;
}

template <class S> HD S mkt(double v, double t);
template <> HD double mkt<double>(double v, double) { return v; }
template <> HD Dual mkt<Dual>(double v, double t) { return Dual(v, t); }

// One pass.  qs/vs/as: knot q(18), v(18), acceleration input(18); cs/sn: cos/sin of q (all LDS, broadcast reads);
// fext: 12 world foot forces.
// MODE specialises the instruction stream of a whole round of lanes: 0 generic; 1 DYNAMIC (inertias on, no foot kinematics, no
// external forces: what a d ID/dx task needs); 2 KINEMATIC (massless: no inertial forces, foot kinematics and external forces on).
template <class S, class SK, int MODE = 0>
HD void wb_pass(const LaneCfg& L, const double* qs, const double* vs, const double* as, const double* cs, const double* sn, const double* fext, const SK& out) {
    constexpr bool MASS = MODE != 2, FEET = MODE != 1;
    const V3<S> zero3 = {S(0.0), S(0.0), S(0.0)};
    auto Q = [&](int i) { return mk<S>(qs[i], L.tq == i); };
    auto SC = [&](int i, S& s_, S& c_) { const double c0 = cs[i], s0 = sn[i]; const bool sd = (L.tq == i); s_ = mkt<S>(s0, sd ? c0 : 0.0); c_ = mkt<S>(c0, sd ? -s0 : 0.0); };
    auto Vv = [&](int i) { return mk<S>(L.vscale * vs[i], L.tv == i); };
    auto Aa = [&](int i) { return S(L.aunit >= 0 ? (L.aunit == i ? 1.0 : 0.0) : L.ascale * as[i]); };
    const double ms = L.mscale;
    // ---- base: PX,PY,PZ (world axes) then RZ, RY, RX at the same origin
    V3<S> om = {S(0.0), S(0.0), S(0.0)}, aa = om;
    V3<S> vl = {Vv(0), Vv(1), Vv(2)};
    V3<S> al = {Aa(0), Aa(1), Aa(2) + L.grav};
    S c3, s3, c4, s4, c5, s5;
    SC(3, s3, c3); SC(4, s4, c4); SC(5, s5, c5);
    rev_joint<2>(c3, s3, Vv(3), Aa(3), om, vl, aa, al);
    rev_joint<1>(c4, s4, Vv(4), Aa(4), om, vl, aa, al);
    rev_joint<0>(c5, s5, Vv(5), Aa(5), om, vl, aa, al);
    V3<S> nb = {S(0.0), S(0.0), S(0.0)}, fb = nb;
    if (MASS && L.body) inertia_force<S>(ms, 3.3, 0, 0, 0, 0.011253, 0, 0, 0.036203, 0, 0.042673, om, vl, al, aa, nb, fb);
    V3<S> ob = {Q(0), Q(1), Q(2)};
#pragma unroll 1
    for (int l = L.l0; l < L.l1; l++) {
        const double sx = (l < 2) ? 1.0 : -1.0, sy = (l & 1) ? -1.0 : 1.0;
        const int j0 = 6 + 3 * l;
        S ca, sa, ch, sh, ck, sk;
        SC(j0, sa, ca); SC(j0 + 1, sh, ch); SC(j0 + 2, sk, ck);
        // abad: origin (sx*.19, sy*.049, 0) in body, axis x
        V3<S> o1 = om, a1 = aa;
        V3<S> v1 = vl + crossc(om, sx * 0.19, sy * 0.049, 0.0), l1 = al + crossc(aa, sx * 0.19, sy * 0.049, 0.0);
        rev_joint<0>(ca, sa, Vv(j0), Aa(j0), o1, v1, a1, l1);
        V3<S> n1 = zero3, f1 = zero3;
        if (MASS) inertia_force<S>(ms, 0.54, 0, sy * 0.036, 0, 0.000381, sy * 0.000058, 0.00000045, 0.000560, sy * 0.00000095, 0.000444, o1, v1, l1, a1, n1, f1);
        // hip: origin (0, sy*.062, 0) in abad, fixed yaw psi, axis y
        V3<S> o2 = o1, a2 = a1;
        V3<S> v2 = v1 + crossc(o1, 0.0, sy * 0.062, 0.0), l2 = l1 + crossc(a1, 0.0, sy * 0.062, 0.0);
        o2 = rotT<2>(L.cpsi, L.spsi, o2); v2 = rotT<2>(L.cpsi, L.spsi, v2); a2 = rotT<2>(L.cpsi, L.spsi, a2); l2 = rotT<2>(L.cpsi, L.spsi, l2);
        rev_joint<1>(ch, sh, Vv(j0 + 1), Aa(j0 + 1), o2, v2, a2, l2);
        V3<S> n2 = zero3, f2 = zero3;
        if (MASS) inertia_force<S>(ms, 0.634, 0, sy * 0.016, -0.02, 0.001983, sy * 0.000245, 0.000013, 0.002103, sy * 0.0000015, 0.000408, o2, v2, l2, a2, n2, f2);
        // knee: origin (0,0,-.209) in thigh, axis y
        V3<S> o3 = o2, a3 = a2;
        V3<S> v3 = v2 + crossc(o2, 0.0, 0.0, -0.209), l3 = l2 + crossc(a2, 0.0, 0.0, -0.209);
        rev_joint<1>(ck, sk, Vv(j0 + 2), Aa(j0 + 2), o3, v3, a3, l3);
        V3<S> n3 = zero3, f3 = zero3;
        if (MASS) inertia_force<S>(ms, 0.064, 0, 0, -0.061, 0.000245, 0, 0, 0.000248, 0, 0.000006, o3, v3, l3, a3, n3, f3);
        if (FEET) {
        // foot point r = (0,0,-.195) in shank
        V3<S> vp = v3 + crossc(o3, 0.0, 0.0, -0.195);
        V3<S> ap = l3 + crossc(a3, 0.0, 0.0, -0.195) + cross(o3, vp);
        // up-rotation shank -> world:  Rwb * Rx(qa) * Rz(psi) * Ry(qh) * Ry(qk)
        auto up = [&](V3<S> w) {
            w = rot<1>(ck, sk, w); w = rot<1>(ch, sh, w); w = rot<2>(L.cpsi, L.spsi, w); w = rot<0>(ca, sa, w);
            w = rot<0>(c5, s5, w); w = rot<1>(c4, s4, w); w = rot<2>(c3, s3, w); return w;
        };
        V3<S> vw = up(vp);
        V3<S> aw = up(ap); aw.z = aw.z - L.grav;
        V3<S> pw;
        {   // position: o_b + Rwb (p0a + Rx (p0h + Rz Ry (p0k + Ry r)))
            V3<S> w = {S(0.0), S(0.0), S(-0.195)};
            w = rot<1>(ck, sk, w); w.z = w.z - 0.209;
            w = rot<1>(ch, sh, w); w = rot<2>(L.cpsi, L.spsi, w); w.y = w.y + sy * 0.062;
            w = rot<0>(ca, sa, w); w.x = w.x + sx * 0.19; w.y = w.y + sy * 0.049;
            w = rot<0>(c5, s5, w); w = rot<1>(c4, s4, w); w = rot<2>(c3, s3, w);
            pw = ob + w;
        }
        out.foot(l, pw, vw, aw);
        }
        if (FEET && L.fscale != 0.0) {   // external world force at the foot -> shank coordinates, subtract
            V3<S> F = {S(L.fscale * fext[3 * l]), S(L.fscale * fext[3 * l + 1]), S(L.fscale * fext[3 * l + 2])};
            F = rotT<2>(c3, s3, F); F = rotT<1>(c4, s4, F); F = rotT<0>(c5, s5, F);
            F = rotT<0>(ca, sa, F); F = rotT<2>(L.cpsi, L.spsi, F); F = rotT<1>(ch, sh, F); F = rotT<1>(ck, sk, F);
            f3 = f3 - F;
            n3 = n3 - ccross(0.0, 0.0, -0.195, F);
        }
        // backward through the leg
        out.tau(j0 + 2, n3.y);
        V3<S> fu = rot<1>(ck, sk, f3), nu = rot<1>(ck, sk, n3);
        f2 = f2 + fu; n2 = n2 + nu + ccross(0.0, 0.0, -0.209, fu);
        out.tau(j0 + 1, n2.y);
        fu = rot<2>(L.cpsi, L.spsi, rot<1>(ch, sh, f2)); nu = rot<2>(L.cpsi, L.spsi, rot<1>(ch, sh, n2));
        f1 = f1 + fu; n1 = n1 + nu + ccross(0.0, sy * 0.062, 0.0, fu);
        out.tau(j0, n1.x);
        fu = rot<0>(ca, sa, f1); nu = rot<0>(ca, sa, n1);
        fb = fb + fu; nb = nb + nu + ccross(sx * 0.19, sy * 0.049, 0.0, fu);
    }
    if (L.partial) { out.base(fb, nb); return; }
    // backward through the base
    out.tau(5, nb.x);
    V3<S> f = rot<0>(c5, s5, fb), n = rot<0>(c5, s5, nb);
    out.tau(4, n.y);
    f = rot<1>(c4, s4, f); n = rot<1>(c4, s4, n);
    out.tau(3, n.z);
    f = rot<2>(c3, s3, f);
    out.tau(0, f.x); out.tau(1, f.y); out.tau(2, f.z);
}

}  // namespace hs
