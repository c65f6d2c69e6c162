// Whole-body rollout knot on a LANE QUAD: one lane per LEG, sixteen knots per wavefront.
//
// The one-wave-per-knot programs of wb_knot.hpp spend 3 251 fp64 VALU instructions per rollout knot, most of them in phases that keep
// 12-23 of the 64 lanes busy (18-row factor, 12 X columns, 23-lane sums; profiles/r03a_lane_occupancy.json).  This file is the other
// mapping of the same arithmetic (SinglePhase::hybrid_rollout knot body, SinglePhase.cpp:197-224; WBM::dynamics ->
// KKTContactDynamics, WBM.cpp:17-57, 368-424; running cost + ReB cost, SinglePhase.cpp:240-251; defect, TrajectoryManagement.cpp:231):
//
//   * a wave holds SIXTEEN knot evaluations (sixteen problems at the same knot and step length: identical control flow), each on a
//     quad of lanes; lane l of a quad owns LEG l (FL, FR, HL, HR: joints 6+3l .. 8+3l, foot l) and a replica of the floating base;
//   * the robot is a star: four 3-joint chains on one floating body.  In the order (legs, base) the mass matrix is
//         M = [ D  C ; C^T  B ],  D = blockdiag(D_l) (3x3 per leg),  C_l (3x6) leg-base coupling,  B (6x6),
//     so everything about a leg (composite inertias, D_l, C_l, its Cholesky block, the foot Jacobian, X_l = L^-1 J_l^T, the foot's
//     constraints and costs) lives in the registers of ITS lane, and the only things that cross lanes are sums over the four legs
//     (composite inertia of the trunk, Schur complement of the base block, base right-hand sides: two DPP adds per value) and the
//     3x3 blocks of the contact Gram matrix (DPP quad broadcasts);
//   * M comes from the composite-rigid-body algorithm on each leg (10-parameter inertias carried leg -> trunk), the bias forces
//     from one Newton-Euler pass per leg, the foot Jacobians from the geometric (axis x arm) form - no unit-acceleration passes;
//   * the contact solve is the block form of the one in wb_knot.hpp (M = L L^T legs first, X = L^-1 Jc^T, G = X^T X + 1e-12 I,
//     lam = G^-1 (-X^T y - gam), qdd = L^-T (y + X lam)): the same factorisation, so the same numbers up to summation order.
// No LDS, no barrier.  Everything is a template over the scalar S: `double` on the GPU (cross-lane steps = DPP quad_perm), a
// four-wide value on the host (tests/_emu: the four lanes of a quad evaluated together), so the kernel logic is checked against the
// oracle in a container that has no GPU.
#pragma once
#include "hs_types.hpp"

namespace hs {

// ------------------------------------------------------------------------------------------------ lane-quad scalar
#ifdef HS_HOST_EMU
struct Q4 {
    double v[4];
    Q4() : v{0, 0, 0, 0} {}
    Q4(double a) : v{a, a, a, a} {}
    Q4(double a, double b, double c, double d) : v{a, b, c, d} {}
};
#define Q4_OP(OP) inline Q4 operator OP(Q4 a, Q4 b) { return Q4(a.v[0] OP b.v[0], a.v[1] OP b.v[1], a.v[2] OP b.v[2], a.v[3] OP b.v[3]); } \
                  inline Q4 operator OP(Q4 a, double b) { return a OP Q4(b); } inline Q4 operator OP(double a, Q4 b) { return Q4(a) OP b; }
Q4_OP(+) Q4_OP(-) Q4_OP(*)
#undef Q4_OP
inline Q4 operator-(Q4 a) { return Q4(-a.v[0], -a.v[1], -a.v[2], -a.v[3]); }
struct Q4b { bool v[4]; };
struct QH {      // host: the four lanes of ONE quad together
    using S = Q4; using B = Q4b;
    static S legc(double a, double b, double c, double d) { return Q4(a, b, c, d); }
    static S sum(S x) { const double s = (x.v[0] + x.v[1]) + (x.v[2] + x.v[3]); return Q4(s); }
    static S vmin(S x) { return Q4(fmin(fmin(x.v[0], x.v[1]), fmin(x.v[2], x.v[3]))); }
    template <int J> static S get(S x) { return Q4(x.v[J]); }
    static S ld(const double* p, size_t off, int stride) { return Q4(p[off], p[off + stride], p[off + 2 * stride], p[off + 3 * stride]); }
    static S ldv(const double* p, size_t off, S idx) { return Q4(p[off + (int)idx.v[0]], p[off + (int)idx.v[1]], p[off + (int)idx.v[2]], p[off + (int)idx.v[3]]); }      // per-lane index
    static void st(double* p, size_t off, int stride, S x) { for (int l = 0; l < 4; l++) p[off + (size_t)l * stride] = x.v[l]; }
    static void stv(double* p, size_t off, S idx, B on, S x) { for (int l = 0; l < 4; l++) if (on.v[l]) p[off + (int)idx.v[l]] = x.v[l]; }
    static void st0(double* p, size_t off, S x) { p[off] = x.v[0]; }
    static S rsqrt(S x) { return Q4(1.0 / std::sqrt(x.v[0]), 1.0 / std::sqrt(x.v[1]), 1.0 / std::sqrt(x.v[2]), 1.0 / std::sqrt(x.v[3])); }
    static S rcp(S x) { return Q4(1.0 / x.v[0], 1.0 / x.v[1], 1.0 / x.v[2], 1.0 / x.v[3]); }
    static void sincos(S a, S& s, S& c) { for (int l = 0; l < 4; l++) { s.v[l] = std::sin(a.v[l]); c.v[l] = std::cos(a.v[l]); } }
    static S log(S x) { return Q4(std::log(x.v[0]), std::log(x.v[1]), std::log(x.v[2]), std::log(x.v[3])); }
    static S min(S a, S b) { return Q4(fmin(a.v[0], b.v[0]), fmin(a.v[1], b.v[1]), fmin(a.v[2], b.v[2]), fmin(a.v[3], b.v[3])); }
    static B gt(S a, S b) { return Q4b{{a.v[0] > b.v[0], a.v[1] > b.v[1], a.v[2] > b.v[2], a.v[3] > b.v[3]}}; }
    static S sel(B m, S a, S b) { return Q4(m.v[0] ? a.v[0] : b.v[0], m.v[1] ? a.v[1] : b.v[1], m.v[2] ? a.v[2] : b.v[2], m.v[3] ? a.v[3] : b.v[3]); }
    static bool any(B m) { return m.v[0] || m.v[1] || m.v[2] || m.v[3]; }
    static B lnot(B m) { return Q4b{{!m.v[0], !m.v[1], !m.v[2], !m.v[3]}}; }
    static bool any_bad(S x, double lim) { for (int l = 0; l < 4; l++) if (x.v[l] > lim || !(x.v[l] == x.v[l])) return true; return false; }
    static double lane0(S x) { return x.v[0]; }
};
#else
template <int CTRL> HD double q_dpp(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
struct QD {      // GPU: S = one lane's double; the quad's other lanes are reached by DPP quad_perm (every lane of a quad must be active)
    using S = double; using B = bool;
    static HD int lane() { return threadIdx.x & 3; }
    static HD S legc(double a, double b, double c, double d) { const int l = lane(); return l == 0 ? a : l == 1 ? b : l == 2 ? c : d; }
    static HD S sum(S x) { const S t = x + q_dpp<0xB1>(x); return t + q_dpp<0x4E>(t); }     // (x0+x1)+(x2+x3) in every lane, bit-identical across the quad
    static HD S vmin(S x) { const S t = fmin(x, q_dpp<0xB1>(x)); return fmin(t, q_dpp<0x4E>(t)); }
    template <int J> static HD S get(S x) { return q_dpp<J * 0x55>(x); }
    template <class PT> static HD S ld(PT p, size_t off, int stride) { return p[off + (size_t)(lane() * stride)]; }
    template <class PT> static HD S ldv(PT p, size_t off, S idx) { return p[off + (size_t)(int)idx]; }      // per-lane index
    template <class PT> static HD void st(PT p, size_t off, int stride, S x) { p[off + (size_t)(lane() * stride)] = x; }
    template <class PT> static HD void stv(PT p, size_t off, S idx, B on, S x) { if (on) p[off + (size_t)(int)idx] = x; }
    template <class PT> static HD void st0(PT p, size_t off, S x) { if (lane() == 0) p[off] = x; }
    static HD S rsqrt(S x) { return ::rsqrt(x); }
    static HD S rcp(S x) { return 1.0 / x; }
    static HD void sincos(S a, S& s, S& c) { ::sincos(a, &s, &c); }
    static HD S log(S x) { return ::log(x); }
    static HD S min(S a, S b) { return fmin(a, b); }
    static HD B gt(S a, S b) { return a > b; }
    static HD S sel(B m, S a, S b) { return m ? a : b; }
    static HD bool any(B m) { return __any(m) != 0; }      // over the WAVE (uniform): only ever used to skip work no lane needs
    static HD B lnot(B m) { return !m; }
    static HD double lane0(S x) { return x; }
};
#endif

// ------------------------------------------------------------------------------------------------ small fixed-size algebra over S
template <class S> struct Sym3 { S xx, xy, xz, yy, yz, zz; };
template <class S> HD V3<S> symmul(const Sym3<S>& A, const V3<S>& w) { return {A.xx * w.x + A.xy * w.y + A.xz * w.z, A.xy * w.x + A.yy * w.y + A.yz * w.z, A.xz * w.x + A.yz * w.y + A.zz * w.z}; }
// rigid-body inertia about the ORIGIN of its frame: mass, first moment h = m c, rotational inertia I_o.
//   force of a motion (angular aa, linear al at the origin):  n = I_o aa + h x al ,  f = m al + aa x h
template <class S> struct RBI { S m; V3<S> h; Sym3<S> I; };
template <class S> HD void rbi_apply(const RBI<S>& R, const V3<S>& aa, const V3<S>& al, V3<S>& n, V3<S>& f) {
    n = symmul(R.I, aa) + cross(R.h, al);
    f = scale(R.m, al) + cross(aa, R.h);
}
// link inertia from the URDF form (mass, COM c, inertia about the COM): I_o = I_c + m ((c.c) 1 - c c^T)
template <class S> HD RBI<S> rbi_link(double m, S cx, S cy, S cz, S ixx, S ixy, S ixz, S iyy, S iyz, S izz) {
    RBI<S> R; R.m = S(m); R.h = {m * cx, m * cy, m * cz};
    const S cc = cx * cx + cy * cy + cz * cz;
    R.I = {ixx + m * (cc - cx * cx), ixy - m * (cx * cy), ixz - m * (cx * cz), iyy + m * (cc - cy * cy), iyz - m * (cy * cz), izz + m * (cc - cz * cz)};
    return R;
}
template <class S> HD RBI<S> rbi_add(const RBI<S>& a, const RBI<S>& b) {
    return {a.m + b.m, a.h + b.h, {a.I.xx + b.I.xx, a.I.xy + b.I.xy, a.I.xz + b.I.xz, a.I.yy + b.I.yy, a.I.yz + b.I.yz, a.I.zz + b.I.zz}};
}
// R A R^T for an axis rotation R = rot<AX>(c, s)
template <int AX, class S> HD Sym3<S> sym_rot(S c, S s, const Sym3<S>& A) {
    const V3<S> a0 = rot<AX>(c, s, V3<S>{A.xx, A.xy, A.xz}), a1 = rot<AX>(c, s, V3<S>{A.xy, A.yy, A.yz}), a2 = rot<AX>(c, s, V3<S>{A.xz, A.yz, A.zz});   // columns of R A
    const V3<S> r0 = rot<AX>(c, s, V3<S>{a0.x, a1.x, a2.x}), r1 = rot<AX>(c, s, V3<S>{a0.y, a1.y, a2.y}), r2 = rot<AX>(c, s, V3<S>{a0.z, a1.z, a2.z});   // rows of (R A) R^T
    return {r0.x, r0.y, r0.z, r1.y, r1.z, r2.z};
}
// the inertia of a child frame (already rotated into the parent's axes: mass m, first moment g, rotational inertia A) seen from the
// parent's origin, the child's origin sitting at p:  h' = g + m p ,  I' = A + 2 (p.g) 1 - p g^T - g p^T + m ((p.p) 1 - p p^T)
template <class S> HD RBI<S> rbi_shift(const S& m, const V3<S>& g, const Sym3<S>& A, const V3<S>& p) {
    const S pg = p.x * g.x + p.y * g.y + p.z * g.z, pp = p.x * p.x + p.y * p.y + p.z * p.z;
    const S d = 2.0 * pg + m * pp;
    RBI<S> R; R.m = m; R.h = g + scale(m, p);
    R.I = {A.xx + d - 2.0 * (p.x * g.x) - m * (p.x * p.x), A.xy - (p.x * g.y + p.y * g.x) - m * (p.x * p.y), A.xz - (p.x * g.z + p.z * g.x) - m * (p.x * p.z),
           A.yy + d - 2.0 * (p.y * g.y) - m * (p.y * p.y), A.yz - (p.y * g.z + p.z * g.y) - m * (p.y * p.z), A.zz + d - 2.0 * (p.z * g.z) - m * (p.z * p.z)};
    return R;
}
// a force (n about the child's origin, f) moved to the parent's origin: n' = R n + p x (R f), f' = R f   (R applied by the caller)
template <class S> HD void force_shift(const V3<S>& p, V3<S>& n, const V3<S>& f) { n = n + cross(p, f); }

// lower Cholesky factor of a symmetric 3 x 3 block: L = [l00 0 0; l10 l11 0; l20 l21 l22] with the RECIPROCAL diagonal r0..r2 kept beside it
template <class S> struct Chol3 { S l10, l20, l21, r0, r1, r2; };
template <class Q, class S> HD Chol3<S> chol3(const S& a00, const S& a10, const S& a11, const S& a20, const S& a21, const S& a22) {
    Chol3<S> L;
    L.r0 = Q::rsqrt(a00); L.l10 = a10 * L.r0; L.l20 = a20 * L.r0;
    L.r1 = Q::rsqrt(a11 - L.l10 * L.l10); L.l21 = (a21 - L.l20 * L.l10) * L.r1;
    L.r2 = Q::rsqrt(a22 - L.l20 * L.l20 - L.l21 * L.l21);
    return L;
}
template <class S> HD V3<S> fwd3(const Chol3<S>& L, const V3<S>& b) {      // L^-1 b
    V3<S> x; x.x = b.x * L.r0; x.y = (b.y - L.l10 * x.x) * L.r1; x.z = (b.z - L.l20 * x.x - L.l21 * x.y) * L.r2; return x;
}
template <class S> HD V3<S> bwd3(const Chol3<S>& L, const V3<S>& b) {      // L^-T b
    V3<S> x; x.z = b.z * L.r2; x.y = (b.y - L.l21 * x.z) * L.r1; x.x = (b.x - L.l10 * x.y - L.l20 * x.z) * L.r0; return x;
}
// 3 x 3 general block, rows r[0..2]
template <class S> struct M33 { V3<S> r[3]; };
template <class S> HD V3<S> m33_mul(const M33<S>& A, const V3<S>& w) { return {A.r[0].x * w.x + A.r[0].y * w.y + A.r[0].z * w.z, A.r[1].x * w.x + A.r[1].y * w.y + A.r[1].z * w.z, A.r[2].x * w.x + A.r[2].y * w.y + A.r[2].z * w.z}; }
template <class S> HD V3<S> m33_mulT(const M33<S>& A, const V3<S>& w) { return {A.r[0].x * w.x + A.r[1].x * w.y + A.r[2].x * w.z, A.r[0].y * w.x + A.r[1].y * w.y + A.r[2].y * w.z, A.r[0].z * w.x + A.r[1].z * w.y + A.r[2].z * w.z}; }
template <class S> HD S dot3(const V3<S>& a, const V3<S>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// dense symmetric 6 x 6 in packed lower storage p[i (i+1)/2 + j], j <= i, and its Cholesky factor (reciprocal diagonal in rd)
HD constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }
template <class Q, class S> HD void chol6(S (&a)[21], S (&rd)[6]) {
    _Pragma("unroll")
    for (int j = 0; j < 6; j++) {
        S d = a[tri(j, j)];
        _Pragma("unroll") for (int k = 0; k < j; k++) d = d - a[tri(j, k)] * a[tri(j, k)];
        const S r = Q::rsqrt(d); rd[j] = r;
        _Pragma("unroll")
        for (int i = j + 1; i < 6; i++) {
            S s = a[tri(i, j)];
            _Pragma("unroll") for (int k = 0; k < j; k++) s = s - a[tri(i, k)] * a[tri(j, k)];
            a[tri(i, j)] = s * r;
        }
    }
}
template <class S> HD void fwd6(const S (&a)[21], const S (&rd)[6], S (&x)[6]) {
    _Pragma("unroll")
    for (int i = 0; i < 6; i++) { S s = x[i]; _Pragma("unroll") for (int k = 0; k < i; k++) s = s - a[tri(i, k)] * x[k]; x[i] = s * rd[i]; }
}
template <class S> HD void bwd6(const S (&a)[21], const S (&rd)[6], S (&x)[6]) {
    _Pragma("unroll")
    for (int i = 5; i >= 0; i--) { S s = x[i]; _Pragma("unroll") for (int k = i + 1; k < 6; k++) s = s - a[tri(k, i)] * x[k]; x[i] = s * rd[i]; }
}

// a force at the trunk origin (body axes) walked back through the base joints RX, RY, RZ, PZ, PY, PX (all at the same origin): the six
// generalised forces of the floating base, in joint order 0..5 (wb_model.hpp, end of wb_pass)
template <class S> struct Base3 { S c3, s3, c4, s4, c5, s5; };
template <class S> HD void base_walk(const Base3<S>& T, const V3<S>& fb, const V3<S>& nb, S (&t)[6]) {
    t[5] = nb.x;
    V3<S> f = rot<0>(T.c5, T.s5, fb), n = rot<0>(T.c5, T.s5, nb);
    t[4] = n.y;
    f = rot<1>(T.c4, T.s4, f); n = rot<1>(T.c4, T.s4, n);
    t[3] = n.z;
    f = rot<2>(T.c3, T.s3, f);
    t[0] = f.x; t[1] = f.y; t[2] = f.z;
}

// reb barrier with one logarithm (ConstraintsBase.h:238-245), branch-free over the lanes of a quad
template <class Q, class S> HD S q_barrier(const S& g, const S& delta) {
    const typename Q::B above = Q::gt(g, delta);
    const S lg = Q::log(Q::sel(above, g, delta));
    if (!Q::any(Q::lnot(above))) return -lg;      // no lane of the wave sits below its delta (the usual case): no division
    const S t = (g - 2.0 * delta) * Q::rcp(delta);
    return Q::sel(above, -lg, 0.5 * (t * t - 1.0) - lg);
}

struct QuadOut { double cost, dsq, ming; bool bad; };
// diagnostic stamps (-DQUAD_PROF, tools/microbench.py): cycles of one wave from the middle of a launch between consecutive marks
#if defined(QUAD_PROF) && !defined(HS_HOST_EMU)
__device__ unsigned long long g_quad_prof[24];
#define QP0() unsigned long long qp_t_ = 0; const bool qp_on_ = (blockIdx.x == HS_QUAD_PROF_BLOCK && threadIdx.x == 0); if (qp_on_) qp_t_ = clock64();
#define QP(i) if (qp_on_) { const unsigned long long t_ = clock64(); atomicAdd(&g_quad_prof[i], t_ - qp_t_); qp_t_ = t_; }
#ifndef HS_QUAD_PROF_BLOCK
#define HS_QUAD_PROF_BLOCK 30011
#endif
#else
#define QP0()
#define QP(i)
#endif
template <int I> struct IC { static constexpr int value = I; };

// One whole-body rollout knot k < h of problem b of a phase WITH shooting nodes, step length eps, evaluated by a lane quad.
//   wr = false: a probe - only (cost, defect^2, min g, divergence) come back;  wr = true additionally stores everything wb_rollout_knot
//   stores: X, U, Y, Xsim, Defect, g, lbase, l and the contact-solve cache of the knot in the layout the LQ knot fetches (hs_types.hpp KC_*).
//   wr must be uniform over the wave.
template <class Q>
HD QuadOut wbq_rollout_knot(PhaseC& P, const ModelDev& md, int b, int k, double eps, int reb_active, const double* x0, bool wr) {
    using S = typename Q::S;
    const bool WR = wr;
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + k) * 36, ku = ((size_t)b * h + k) * 12, kk = (size_t)b * h + k;
    QP0()
    // ---- state of the knot: the floating base replicated in every lane, the lane's own leg
    S qb[6], vb[6], ql[3], vl_[3], ul[3];
    _Pragma("unroll") for (int i = 0; i < 6; i++) { qb[i] = Q::ld(P.Xbar, kx + i, 0) + eps * Q::ld(P.dX, kx + i, 0); vb[i] = Q::ld(P.Xbar, kx + 18 + i, 0) + eps * Q::ld(P.dX, kx + 18 + i, 0); }
    _Pragma("unroll") for (int j = 0; j < 3; j++) {
        ql[j] = Q::ld(P.Xbar, kx + 6 + j, 3) + eps * Q::ld(P.dX, kx + 6 + j, 3); vl_[j] = Q::ld(P.Xbar, kx + 24 + j, 3) + eps * Q::ld(P.dX, kx + 24 + j, 3);
        ul[j] = Q::ld(P.Ubar, ku + j, 3) + eps * (Q::ld(P.dU, ku + j, 3) + Q::ld(P.KdX, ku + j, 3));      // u = ubar + eps (dU + K dX), wb_knot.hpp / DESIGN section 4
    }
    if (WR) {
        _Pragma("unroll") for (int j = 0; j < 3; j++) {
            Q::st(P.X, kx + 6 + j, 3, ql[j]); Q::st(P.X, kx + 24 + j, 3, vl_[j]); Q::st(P.U, ku + j, 3, ul[j]);
            Q::st0(P.X, kx + j, qb[j]); Q::st0(P.X, kx + 3 + j, qb[3 + j]); Q::st0(P.X, kx + 18 + j, vb[j]); Q::st0(P.X, kx + 21 + j, vb[3 + j]);
        }
    }
    // Everything of the knot that needs (x, u) only - the tracking cost, the torque / joint-speed / joint / height barriers - is evaluated HERE, its
    // references and barrier parameters fetched with the state (one exposed round trip for all of them) and dead before the contact solve's
    // register peak; what needs the dynamics (defect, foot costs, friction pyramid) reads later, ahead of the contact solve (see below).
    const double dt = P.dt;
    const HS_GLOBAL double* rr = P.rref + (size_t)k * 80;
    const size_t gk = kk * P.ng;
    const S zero = S(0.0);
    const S w0 = Q::legc(1.0, 0.0, 0.0, 0.0);      // the replicated base entries are counted once
    S lq, accT = zero, accS = zero, accJ = zero, accH = zero, gmin = zero;
    {
        S xrb[6], vrb[6], xrl[3], vrl[3], url[3];
        S eT[6], dT[6], eS[6], dS[6], eJ[6], dJ[6], eH = zero, dH_ = S(1.0);
        _Pragma("unroll") for (int i = 0; i < 6; i++) { xrb[i] = Q::ld(rr, i, 0); vrb[i] = Q::ld(rr, 18 + i, 0); }
        _Pragma("unroll") for (int j = 0; j < 3; j++) { xrl[j] = Q::ld(rr, 6 + j, 3); vrl[j] = Q::ld(rr, 24 + j, 3); url[j] = Q::ld(rr, 36 + j, 3); }
        // entries 0..2: lower-bound half of the lane's three joints, 3..5: upper-bound half (12 constraints further).  One uniform branch per
        // constraint object AROUND its twelve loads (a branch per load would put a wait between every pair)
        _Pragma("unroll") for (int j = 0; j < 6; j++) { eT[j] = zero; dT[j] = S(1.0); eS[j] = zero; dS[j] = S(1.0); eJ[j] = zero; dJ[j] = S(1.0); }
        if (P.go_torque >= 0) { _Pragma("unroll") for (int j = 0; j < 6; j++) { const int o = (j < 3) ? j : 12 + j - 3; eT[j] = Q::ld(P.eps, gk + P.go_torque + o, 3); dT[j] = Q::ld(P.delta, gk + P.go_torque + o, 3); } }
        if (P.go_jspeed >= 0) { _Pragma("unroll") for (int j = 0; j < 6; j++) { const int o = (j < 3) ? j : 12 + j - 3; eS[j] = Q::ld(P.eps, gk + P.go_jspeed + o, 3); dS[j] = Q::ld(P.delta, gk + P.go_jspeed + o, 3); } }
        if (P.go_joint >= 0) { _Pragma("unroll") for (int j = 0; j < 6; j++) { const int o = (j < 3) ? j : 12 + j - 3; eJ[j] = Q::ld(P.eps, gk + P.go_joint + o, 3); dJ[j] = Q::ld(P.delta, gk + P.go_joint + o, 3); } }
        if (P.go_height >= 0) { eH = Q::ld(P.eps, gk + P.go_height, 0); dH_ = Q::ld(P.delta, gk + P.go_height, 0); }
        QP(0)      // first reads issued
        // running cost, tracking part (QuadraticTrackingCost): 0.5 (sum_x + sum_u) dt, formed like the wave kernel: l = 0.5 sx; l += 0.5 su; l *= dt
        S sxq = zero, suq = zero;
        _Pragma("unroll")
        for (int i = 0; i < 6; i++) {
            const S dq = qb[i] - xrb[i], dv = vb[i] - vrb[i];
            sxq = sxq + w0 * (dq * P.q[i] * dq + dv * P.q[18 + i] * dv);
        }
        _Pragma("unroll")
        for (int j = 0; j < 3; j++) {
            const S dq = ql[j] - xrl[j], dv = vl_[j] - vrl[j], du = ul[j] - url[j];
            // the phase's weights of the lane's leg: scalars of the descriptor picked by the lane (a per-lane index would be a vector load from constant memory)
            const S wq = Q::legc(P.q[6 + j], P.q[9 + j], P.q[12 + j], P.q[15 + j]), wv = Q::legc(P.q[24 + j], P.q[27 + j], P.q[30 + j], P.q[33 + j]), wu = Q::legc(P.r[j], P.r[3 + j], P.r[6 + j], P.r[9 + j]);
            sxq = sxq + (dq * wq * dq + dv * wv * dv);
            suq = suq + du * wu * du;
        }
        sxq = Q::sum(sxq); suq = Q::sum(suq);
        lq = 0.5 * sxq; lq = lq + 0.5 * suq; lq = lq * dt;
        // path constraints on (x, u) of the lane's leg (MHPCConstraint.cpp:77-204): values, ReB cost per constraint object, minimum
        auto six = [&](const S (&g)[6], const S (&e)[6], const S (&dl)[6], int c0, S& acc) {      // one constraint object: entries j / 12 + j of the lane's three joints
            _Pragma("unroll") for (int j = 0; j < 6; j++) { acc = acc + e[j] * q_barrier<Q, S>(g[j], dl[j]); gmin = Q::min(gmin, g[j]); }
            if (WR) { _Pragma("unroll") for (int j = 0; j < 6; j++) Q::st(P.g, gk + c0 + ((j < 3) ? j : 12 + j - 3), 3, g[j]); }
            acc = Q::sum(acc);
        };
        if (P.go_torque >= 0) { const S g[6] = {-ul[0] + P.torque_limit, -ul[1] + P.torque_limit, -ul[2] + P.torque_limit, ul[0] + P.torque_limit, ul[1] + P.torque_limit, ul[2] + P.torque_limit}; six(g, eT, dT, P.go_torque, accT); }
        if (P.go_jspeed >= 0) { const S g[6] = {vl_[0] - P.jspeed_lb, vl_[1] - P.jspeed_lb, vl_[2] - P.jspeed_lb, -vl_[0] + P.jspeed_ub, -vl_[1] + P.jspeed_ub, -vl_[2] + P.jspeed_ub}; six(g, eS, dS, P.go_jspeed, accS); }
        if (P.go_joint >= 0) { const S g[6] = {ql[0] - P.joint_lb[0], ql[1] - P.joint_lb[1], ql[2] - P.joint_lb[2], -ql[0] + P.joint_ub[0], -ql[1] + P.joint_ub[1], -ql[2] + P.joint_ub[2]}; six(g, eJ, dJ, P.go_joint, accJ); }
        if (P.go_height >= 0) {      // one constraint: evaluated by every lane on the same data, counted once
            const S g = qb[2] - P.h_min;
            if (WR) Q::st0(P.g, gk + P.go_height, g);
            gmin = Q::min(gmin, g);
            accH = eH * q_barrier<Q, S>(g, dH_);
        }
    }
    const S sx = Q::legc(1.0, 1.0, -1.0, -1.0), sy = Q::legc(1.0, -1.0, 1.0, -1.0);
    const double cps = md.cpsi_dyn, sps = md.spsi_dyn;      // every term of the rollout is a Pinocchio-equivalent one (quirk xii)
    Base3<S> T; S ca, sa, ch, sh, ck, sk;
    {   // the three base angles: lane l < 3 evaluates angle 3 + l (lane 3 repeats the last), the quad reads the results
        const S ang = Q::legc(1, 0, 0, 0) * qb[3] + Q::legc(0, 1, 0, 0) * qb[4] + Q::legc(0, 0, 1, 1) * qb[5];
        S sb, cb_; Q::sincos(ang, sb, cb_);
        T.s3 = Q::template get<0>(sb); T.c3 = Q::template get<0>(cb_); T.s4 = Q::template get<1>(sb); T.c4 = Q::template get<1>(cb_); T.s5 = Q::template get<2>(sb); T.c5 = Q::template get<2>(cb_);
    }
    Q::sincos(ql[0], sa, ca); Q::sincos(ql[1], sh, ch); Q::sincos(ql[2], sk, ck);
    QP(1)      // trig
    const V3<S> pa = {sx * 0.19, sy * 0.049, S(0.0)}, ph = {S(0.0), sy * 0.062, S(0.0)}, pk = {S(0.0), S(0.0), S(-0.209)};

    // ---- composite inertias, leg -> trunk (frames: K shank, H thigh, A abad, B trunk; v_H = Ry(qk) v_K, v_A = Rz(psi) Ry(qh) v_H, v_B = Rx(qa) v_A)
    const RBI<S> IK = rbi_link<S>(0.064, zero, zero, S(-0.061), S(0.000245), zero, zero, S(0.000248), zero, S(0.000006));
    RBI<S> IH = rbi_link<S>(0.634, zero, sy * 0.016, S(-0.02), S(0.001983), sy * 0.000245, S(0.000013), S(0.002103), sy * 0.0000015, S(0.000408));
    IH = rbi_add(IH, rbi_shift(IK.m, rot<1>(ck, sk, IK.h), sym_rot<1>(ck, sk, IK.I), pk));
    RBI<S> IA = rbi_link<S>(0.54, zero, sy * 0.036, zero, S(0.000381), sy * 0.000058, S(0.00000045), S(0.000560), sy * 0.00000095, S(0.000444));
    IA = rbi_add(IA, rbi_shift(IH.m, rot<2>(S(cps), S(sps), rot<1>(ch, sh, IH.h)), sym_rot<2>(S(cps), S(sps), sym_rot<1>(ch, sh, IH.I)), ph));
    const RBI<S> IBl = rbi_shift(IA.m, rot<0>(ca, sa, IA.h), sym_rot<0>(ca, sa, IA.I), pa);
    // whole robot about the trunk origin: trunk + the four legs (sums over the quad)
    RBI<S> IT;
    IT.m = Q::sum(IBl.m) + 3.3; IT.h = {Q::sum(IBl.h.x), Q::sum(IBl.h.y), Q::sum(IBl.h.z)};
    IT.I = {Q::sum(IBl.I.xx) + 0.011253, Q::sum(IBl.I.xy), Q::sum(IBl.I.xz), Q::sum(IBl.I.yy) + 0.036203, Q::sum(IBl.I.yz), Q::sum(IBl.I.zz) + 0.042673};

    QP(2)      // composite inertias
    // ---- mass-matrix blocks of the leg: D (3x3, joints abad / hip / knee) and Ct[c][j] = M(base joint c, leg joint j)
    S Ct[6][3], d_aa, d_ha, d_hh, d_ka, d_kh, d_kk;
    {
        // knee: unit acceleration about y of K
        V3<S> n = {IK.I.xy, IK.I.yy, IK.I.yz}, f = {IK.h.z, zero, -IK.h.x};
        d_kk = n.y;
        f = rot<1>(ck, sk, f); n = rot<1>(ck, sk, n); force_shift(pk, n, f);                              // -> H
        d_kh = n.y;
        f = rot<2>(S(cps), S(sps), rot<1>(ch, sh, f)); n = rot<2>(S(cps), S(sps), rot<1>(ch, sh, n)); force_shift(ph, n, f);      // -> A
        d_ka = n.x;
        f = rot<0>(ca, sa, f); n = rot<0>(ca, sa, n); force_shift(pa, n, f);                              // -> B
        S t[6]; base_walk(T, f, n, t);
        _Pragma("unroll") for (int c = 0; c < 6; c++) Ct[c][2] = t[c];
    }
    {
        V3<S> n = {IH.I.xy, IH.I.yy, IH.I.yz}, f = {IH.h.z, zero, -IH.h.x};                               // hip: about y of H
        d_hh = n.y;
        f = rot<2>(S(cps), S(sps), rot<1>(ch, sh, f)); n = rot<2>(S(cps), S(sps), rot<1>(ch, sh, n)); force_shift(ph, n, f);
        d_ha = n.x;
        f = rot<0>(ca, sa, f); n = rot<0>(ca, sa, n); force_shift(pa, n, f);
        S t[6]; base_walk(T, f, n, t);
        _Pragma("unroll") for (int c = 0; c < 6; c++) Ct[c][1] = t[c];
    }
    {
        V3<S> n = {IA.I.xx, IA.I.xy, IA.I.xz}, f = {zero, -IA.h.z, IA.h.y};                               // abad: about x of A
        d_aa = n.x;
        f = rot<0>(ca, sa, f); n = rot<0>(ca, sa, n); force_shift(pa, n, f);
        S t[6]; base_walk(T, f, n, t);
        _Pragma("unroll") for (int c = 0; c < 6; c++) Ct[c][0] = t[c];
    }
    QP(3)      // D, Ct columns
    // ---- base block B (6x6, replicated): unit accelerations of the base joints seen in trunk axes, force of the WHOLE robot, walked back
    S Bm[21];
    {
        V3<S> acc[6][2];      // [c][0] angular, [c][1] linear
        const V3<S> z3 = {zero, zero, zero};
        const V3<S> ex = {S(1.0), zero, zero}, ey = {zero, S(1.0), zero}, ez = {zero, zero, S(1.0)};
        auto w2b = [&](const V3<S>& w) { return rotT<0>(T.c5, T.s5, rotT<1>(T.c4, T.s4, rotT<2>(T.c3, T.s3, w))); };
        acc[0][0] = z3; acc[0][1] = w2b(ex); acc[1][0] = z3; acc[1][1] = w2b(ey); acc[2][0] = z3; acc[2][1] = w2b(ez);
        acc[3][0] = rotT<0>(T.c5, T.s5, rotT<1>(T.c4, T.s4, ez)); acc[3][1] = z3;
        acc[4][0] = rotT<0>(T.c5, T.s5, ey); acc[4][1] = z3;
        acc[5][0] = ex; acc[5][1] = z3;
        _Pragma("unroll")
        for (int c = 0; c < 6; c++) {
            V3<S> n, f; rbi_apply(IT, acc[c][0], acc[c][1], n, f);
            S t[6]; base_walk(T, f, n, t);
            _Pragma("unroll") for (int i = c; i < 6; i++) Bm[tri(i, c)] = t[i];
        }
    }
    QP(4)      // base block
    // ---- bias forces: one Newton-Euler pass down and up the leg with the knot's velocities, zero acceleration, gravity as a base acceleration
    S hl[3], hb[6]; V3<S> fpos, fvel, jdv;
    V3<S> rB;      // foot relative to the trunk origin, trunk axes (for the Jacobian below)
    V3<S> dA, dH;  // foot relative to the abad origin (trunk axes after Rx: see below) / hip origin, A axes
    {
        V3<S> om = {zero, zero, zero}, aa = om, vl = {vb[0], vb[1], vb[2]}, al = {zero, zero, S(GRAV)};
        auto revj = [&](auto AXT, const S& c, const S& s, const S& qd, V3<S>& om_, V3<S>& vl_2, V3<S>& aa_, V3<S>& al_) {
            constexpr int AX = decltype(AXT)::value;
            V3<S> o = rotT<AX>(c, s, om_), v = rotT<AX>(c, s, vl_2), a2 = rotT<AX>(c, s, aa_), a1 = rotT<AX>(c, s, al_);
            if (AX == 0) { o.x = o.x + qd; a2.y = a2.y + o.z * qd; a2.z = a2.z - o.y * qd; a1.y = a1.y + v.z * qd; a1.z = a1.z - v.y * qd; }
            if (AX == 1) { o.y = o.y + qd; a2.x = a2.x - o.z * qd; a2.z = a2.z + o.x * qd; a1.x = a1.x - v.z * qd; a1.z = a1.z + v.x * qd; }
            if (AX == 2) { o.z = o.z + qd; a2.x = a2.x + o.y * qd; a2.y = a2.y - o.x * qd; a1.x = a1.x + v.y * qd; a1.y = a1.y - v.x * qd; }
            om_ = o; vl_2 = v; aa_ = a2; al_ = a1;
        };
        using A0 = IC<0>; using A1 = IC<1>; using A2 = IC<2>;
        revj(A2{}, T.c3, T.s3, vb[3], om, vl, aa, al); revj(A1{}, T.c4, T.s4, vb[4], om, vl, aa, al); revj(A0{}, T.c5, T.s5, vb[5], om, vl, aa, al);
        // force of a link: I a + v x* I v with the link's inertia about its origin
        auto link_force = [&](const RBI<S>& I, const V3<S>& o, const V3<S>& v, const V3<S>& a2, const V3<S>& a1, V3<S>& n, V3<S>& f) {
            const V3<S> hl_ = scale(I.m, v) + cross(o, I.h);                 // linear momentum  m v + om x h
            const V3<S> ha = symmul(I.I, o) + cross(I.h, v);                 // angular momentum about the origin
            rbi_apply(I, a2, a1, n, f);
            f = f + cross(o, hl_);
            n = n + cross(o, ha) + cross(v, hl_);
        };
        // trunk (own link only; its wrench joins the legs' in the quad sum, so only lane 0's copy is counted)
        const RBI<S> Itr = rbi_link<S>(3.3, zero, zero, zero, S(0.011253), zero, zero, S(0.036203), zero, S(0.042673));
        V3<S> nb, fb; link_force(Itr, om, vl, aa, al, nb, fb);
        const S l0 = Q::legc(1.0, 0.0, 0.0, 0.0);
        nb = scale(l0, nb); fb = scale(l0, fb);
        // down the leg
        V3<S> o1 = om, a1 = aa, v1 = vl + cross(om, pa), l1 = al + cross(aa, pa);
        revj(A0{}, ca, sa, vl_[0], o1, v1, a1, l1);
        const RBI<S> Iab = rbi_link<S>(0.54, zero, sy * 0.036, zero, S(0.000381), sy * 0.000058, S(0.00000045), S(0.000560), sy * 0.00000095, S(0.000444));
        V3<S> n1, f1; link_force(Iab, o1, v1, a1, l1, n1, f1);
        V3<S> o2 = o1, a2 = a1, v2 = v1 + cross(o1, ph), l2 = l1 + cross(a1, ph);
        o2 = rotT<2>(S(cps), S(sps), o2); v2 = rotT<2>(S(cps), S(sps), v2); a2 = rotT<2>(S(cps), S(sps), a2); l2 = rotT<2>(S(cps), S(sps), l2);
        revj(A1{}, ch, sh, vl_[1], o2, v2, a2, l2);
        const RBI<S> Ith = rbi_link<S>(0.634, zero, sy * 0.016, S(-0.02), S(0.001983), sy * 0.000245, S(0.000013), S(0.002103), sy * 0.0000015, S(0.000408));
        V3<S> n2, f2; link_force(Ith, o2, v2, a2, l2, n2, f2);
        V3<S> o3 = o2, a3 = a2, v3 = v2 + cross(o2, pk), l3 = l2 + cross(a2, pk);
        revj(A1{}, ck, sk, vl_[2], o3, v3, a3, l3);
        V3<S> n3, f3; link_force(IK, o3, v3, a3, l3, n3, f3);
        // foot point (0, 0, -0.195) in K: velocity, classical acceleration, position
        const V3<S> rf = {zero, zero, S(-0.195)};
        const V3<S> vp = v3 + cross(o3, rf), ap = l3 + cross(a3, rf) + cross(o3, vp);
        auto up = [&](V3<S> w) {      // K -> world
            w = rot<1>(ck, sk, w); w = rot<1>(ch, sh, w); w = rot<2>(S(cps), S(sps), w); w = rot<0>(ca, sa, w);
            w = rot<0>(T.c5, T.s5, w); w = rot<1>(T.c4, T.s4, w); w = rot<2>(T.c3, T.s3, w); return w;
        };
        fvel = up(vp); jdv = up(ap); jdv.z = jdv.z - GRAV;
        const V3<S> rK = rot<1>(ck, sk, rf);                            // foot relative to the knee origin, H axes
        const V3<S> rH = pk + rK;                                       // ... relative to the hip origin, H axes
        dH = rot<2>(S(cps), S(sps), rot<1>(ch, sh, rH));                // ... relative to the hip origin, A axes
        const V3<S> rA = ph + dH;                                       // ... relative to the abad origin, A axes
        dA = rot<0>(ca, sa, rA);                                        // ... relative to the abad origin, trunk axes
        rB = pa + dA;
        const V3<S> rW = rot<2>(T.c3, T.s3, rot<1>(T.c4, T.s4, rot<0>(T.c5, T.s5, rB)));
        fpos = V3<S>{qb[0], qb[1], qb[2]} + rW;
        // back up the leg
        hl[2] = n3.y;
        V3<S> fu = rot<1>(ck, sk, f3), nu = rot<1>(ck, sk, n3);
        f2 = f2 + fu; n2 = n2 + nu + cross(pk, fu);
        hl[1] = n2.y;
        fu = rot<2>(S(cps), S(sps), rot<1>(ch, sh, f2)); nu = rot<2>(S(cps), S(sps), rot<1>(ch, sh, n2));
        f1 = f1 + fu; n1 = n1 + nu + cross(ph, fu);
        hl[0] = n1.x;
        fu = rot<0>(ca, sa, f1); nu = rot<0>(ca, sa, n1);
        fb = fb + fu; nb = nb + nu + cross(pa, fu);
        fb = {Q::sum(fb.x), Q::sum(fb.y), Q::sum(fb.z)}; nb = {Q::sum(nb.x), Q::sum(nb.y), Q::sum(nb.z)};
        base_walk(T, fb, nb, hb);
    }
    QP(5)      // bias pass
    // ---- foot Jacobian of the leg, world axes: Ja (3 x 3 over abad, hip, knee), Jb (3 x 6 over the base joints) - geometric form axis x arm
    const S cl = Q::legc(P.contact[0] > 0 ? 1.0 : 0.0, P.contact[1] > 0 ? 1.0 : 0.0, P.contact[2] > 0 ? 1.0 : 0.0, P.contact[3] > 0 ? 1.0 : 0.0);      // contact flag of the lane's leg
    M33<S> Ja; S Jb[3][6];
    {
        auto b2w = [&](const V3<S>& w) { return rot<2>(T.c3, T.s3, rot<1>(T.c4, T.s4, rot<0>(T.c5, T.s5, w))); };
        const V3<S> ex = {S(1.0), zero, zero}, ey = {zero, S(1.0), zero}, ez = {zero, zero, S(1.0)};
        const V3<S> jk = b2w(rot<0>(ca, sa, rot<2>(S(cps), S(sps), rot<1>(ch, sh, cross(ey, rot<1>(ck, sk, V3<S>{zero, zero, S(-0.195)}))))));     // knee axis y (H axes) x arm from the knee
        const V3<S> jh = b2w(rot<0>(ca, sa, cross(rot<2>(S(cps), S(sps), ey), dH)));                                                              // hip axis: Rz(psi) e_y in A axes
        const V3<S> ja = b2w(cross(ex, dA));                                                                                                       // abad axis x of the trunk
        Ja.r[0] = {ja.x, jh.x, jk.x}; Ja.r[1] = {ja.y, jh.y, jk.y}; Ja.r[2] = {ja.z, jh.z, jk.z};
        const V3<S> rW = b2w(rB);
        const V3<S> a3 = ez, a4 = rot<2>(T.c3, T.s3, ey), a5 = rot<2>(T.c3, T.s3, rot<1>(T.c4, T.s4, ex));
        const V3<S> j3 = cross(a3, rW), j4 = cross(a4, rW), j5 = cross(a5, rW);
        Jb[0][0] = S(1.0); Jb[0][1] = zero; Jb[0][2] = zero; Jb[1][0] = zero; Jb[1][1] = S(1.0); Jb[1][2] = zero; Jb[2][0] = zero; Jb[2][1] = zero; Jb[2][2] = S(1.0);
        Jb[0][3] = j3.x; Jb[1][3] = j3.y; Jb[2][3] = j3.z; Jb[0][4] = j4.x; Jb[1][4] = j4.y; Jb[2][4] = j4.z; Jb[0][5] = j5.x; Jb[1][5] = j5.y; Jb[2][5] = j5.z;
    }

    QP(6)      // Jacobian
    // Contact-solve cache of the knot (wr), in the layout of the one-wave programs (hs_types.hpp KC_*: factor in the legs-first order, X and the
    // Gram matrix over the compact contact columns), every piece stored as soon as it is final.  Column block of a foot = its rank among the
    // contact feet; the swing feet fill the padding blocks behind them.
    const auto kc = P.kc + kk * KC_SIZE;
    const S lane = Q::legc(0, 1, 2, 3);
    const S cf0 = Q::template get<0>(cl), cf1 = Q::template get<1>(cl), cf2 = Q::template get<2>(cl), cf3 = Q::template get<3>(cl);
    const S before = Q::legc(0, 1, 0, 0) * cf0 + Q::legc(0, 0, 1, 0) * (cf0 + cf1) + Q::legc(0, 0, 0, 1) * (cf0 + cf1 + cf2);      // contact feet in front of the lane's
    const typename Q::B on = Q::gt(cl, S(0.5)), all = Q::gt(S(1.0), S(0.0));
    const S cb = Q::sel(on, before, (cf0 + cf1 + cf2 + cf3) + (lane - before));
    // (Structural zeros of the cache image - the other legs' columns of a foot's Jacobian rows, the other feet's columns of a leg's rows of X - are
    // never stored: the buffer is zero-filled when it is laid out, hsddp_create / hsddp_reconfigure, the contact pattern of a phase never changes,
    // and the one-wave program writes exact zeros there.)
    if (WR) {      // all-feet Jacobian (12 x 18, joint-order columns): the foot's rows - base columns and its own leg's columns
        _Pragma("unroll") for (int r = 0; r < 3; r++) {
            _Pragma("unroll") for (int c = 0; c < 6; c++) Q::st(kc, KC_J + 18 * r + c, 54, Jb[r][c]);
            Q::st(kc, KC_J + 18 * r + 6, 57, Ja.r[r].x); Q::st(kc, KC_J + 18 * r + 7, 57, Ja.r[r].y); Q::st(kc, KC_J + 18 * r + 8, 57, Ja.r[r].z);
        }
        Q::st(kc, KC_FP, 3, fpos.x); Q::st(kc, KC_FP + 1, 3, fpos.y); Q::st(kc, KC_FP + 2, 3, fpos.z);
        Q::st(kc, KC_FV, 3, fvel.x); Q::st(kc, KC_FV + 1, 3, fvel.y); Q::st(kc, KC_FV + 2, 3, fvel.z);
    }
    // ---- the reads of the tail (what needs the dynamics: next knot's state for the defect, foot-cost references, friction-pyramid parameters)
    // are issued HERE: their round trip runs under the contact solve instead of being exposed at the end
    S xnb[6], vnb[6], xnl[3], vnl[3];
    auto read_next = [&]() {
        _Pragma("unroll") for (int i = 0; i < 6; i++) { xnb[i] = Q::ld(P.Xbar, kx + 36 + i, 0) + eps * Q::ld(P.dX, kx + 36 + i, 0); vnb[i] = Q::ld(P.Xbar, kx + 54 + i, 0) + eps * Q::ld(P.dX, kx + 54 + i, 0); }
        _Pragma("unroll") for (int j = 0; j < 3; j++) { xnl[j] = Q::ld(P.Xbar, kx + 42 + j, 3) + eps * Q::ld(P.dX, kx + 42 + j, 3); vnl[j] = Q::ld(P.Xbar, kx + 60 + j, 3) + eps * Q::ld(P.dX, kx + 60 + j, 3); }
    };
#ifdef QUAD_NEXT_EARLY      // (measured: the 18 values held through the factorisation cost 24 spilled registers)
    read_next();
#endif
    // ---- contact solve, block form.  L = [ blockdiag(L_l) 0 ; E  L_S ],  E_l = Ct L_l^-T (6 x 3),  S = B - sum_l E_l E_l^T
    const Chol3<S> Ll = chol3<Q, S>(d_aa, d_ha, d_hh, d_ka, d_kh, d_kk);
    S E[6][3];
    _Pragma("unroll")
    for (int c = 0; c < 6; c++) {      // row c of E: E L^T = Ct  ->  forward substitution along the row
        E[c][0] = Ct[c][0] * Ll.r0; E[c][1] = (Ct[c][1] - E[c][0] * Ll.l10) * Ll.r1; E[c][2] = (Ct[c][2] - E[c][0] * Ll.l20 - E[c][1] * Ll.l21) * Ll.r2;
    }
    S LS[21], rdS[6];
    _Pragma("unroll")
    for (int i = 0; i < 6; i++) _Pragma("unroll") for (int j = 0; j <= i; j++) LS[tri(i, j)] = Bm[tri(i, j)] - Q::sum(E[i][0] * E[j][0] + E[i][1] * E[j][1] + E[i][2] * E[j][2]);
    chol6<Q, S>(LS, rdS);
    if (WR) {      // factor of M: the leg's 3 x 3 block, its columns of the base rows, the base block (lane 0), reciprocal diagonals
        Q::st(kc, KC_M + 18, 57, Ll.l10); Q::st(kc, KC_M + 36, 57, Ll.l20); Q::st(kc, KC_M + 37, 57, Ll.l21);
        Q::st(kc, KC_RDM, 3, Ll.r0); Q::st(kc, KC_RDM + 1, 3, Ll.r1); Q::st(kc, KC_RDM + 2, 3, Ll.r2);
        _Pragma("unroll") for (int c = 0; c < 6; c++) {
            _Pragma("unroll") for (int j = 0; j < 3; j++) Q::st(kc, KC_M + 216 + 18 * c + j, 3, E[c][j]);
            _Pragma("unroll") for (int j = 0; j < c; j++) Q::st0(kc, KC_M + (12 + c) * 18 + 12 + j, LS[tri(c, j)]);
            Q::st0(kc, KC_RDM + 12 + c, rdS[c]);
        }
    }
    QP(7)      // chol3, E, Schur complement, chol6 (+ cache stores)
    // y = L^-1 (tau - h): leg part in the lane, base part replicated
    const V3<S> yl = fwd3(Ll, V3<S>{ul[0] - hl[0], ul[1] - hl[1], ul[2] - hl[2]});
    S yb[6];
    _Pragma("unroll") for (int c = 0; c < 6; c++) yb[c] = -hb[c] - Q::sum(E[c][0] * yl.x + E[c][1] * yl.y + E[c][2] * yl.z);
    fwd6(LS, rdS, yb);
    // X = L^-1 Jc^T for the lane's foot (zero for a swing leg): Xt (3 leg rows x 3 force directions), Xb (6 base rows x 3)
    M33<S> Xt; S Xb[6][3];      // Xt.r[d] = column d (force direction d) as a 3-vector over the leg rows ; Xb[c][d]
    _Pragma("unroll")
    for (int d = 0; d < 3; d++) {
        const V3<S> xt = scale(cl, fwd3(Ll, Ja.r[d]));      // L_l^-1 (row d of Ja)^T
        Xt.r[d] = xt;
        S w[6];
        _Pragma("unroll") for (int c = 0; c < 6; c++) w[c] = cl * Jb[d][c] - (E[c][0] * xt.x + E[c][1] * xt.y + E[c][2] * xt.z);
        fwd6(LS, rdS, w);
        _Pragma("unroll") for (int c = 0; c < 6; c++) Xb[c][d] = w[c];
    }
    if (WR) {      // X = L^-1 Jc^T (18 x 12, row-major over the legs-first rows): the leg's rows hold its own block only, the base rows one column block per lane
        _Pragma("unroll") for (int d = 0; d < 3; d++) {      // rows 3 lane + j, columns 3 cb + d: the lane's own 3 x 3 block (a swing lane's block is zero)
            const S ix = 36.0 * lane + 3.0 * cb + (double)d;
            Q::stv(kc, KC_X, ix, on, Xt.r[d].x); Q::stv(kc, KC_X + 12, ix, on, Xt.r[d].y); Q::stv(kc, KC_X + 24, ix, on, Xt.r[d].z);
        }
        _Pragma("unroll") for (int c = 0; c < 6; c++) _Pragma("unroll") for (int d = 0; d < 3; d++) Q::stv(kc, KC_X + (12 + c) * 12 + d, 3.0 * cb, all, Xb[c][d]);
    }
    QP(8)      // y, X
    // Gram matrix G = X^T X in 3 x 3 blocks: lane f holds block row f (blocks g <= f), a swing leg's diagonal block is the identity
    // (its multiplier is zero); right-hand side  -X^T y - gam,  gam = Jdot v + 2 alpha J v (WBM.cpp:392-408)
    M33<S> G[4];
    auto xb_of = [&](auto JT, int c, int d) { constexpr int J = decltype(JT)::value; return Q::template get<J>(Xb[c][d]); };
    using J0 = IC<0>; using J1 = IC<1>; using J2 = IC<2>; using J3 = IC<3>;
    auto gram_block = [&](auto JT, M33<S>& Gb) {
        S o[6][3];
        _Pragma("unroll") for (int c = 0; c < 6; c++) _Pragma("unroll") for (int d = 0; d < 3; d++) o[c][d] = xb_of(JT, c, d);
        _Pragma("unroll")
        for (int r = 0; r < 3; r++) {
            S e[3];
            _Pragma("unroll") for (int d = 0; d < 3; d++) { S s = Xb[0][r] * o[0][d]; _Pragma("unroll") for (int c = 1; c < 6; c++) s = s + Xb[c][r] * o[c][d]; e[d] = s; }
            Gb.r[r] = {e[0], e[1], e[2]};
        }
    };
    gram_block(J0{}, G[0]); gram_block(J1{}, G[1]); gram_block(J2{}, G[2]); gram_block(J3{}, G[3]);
    // the lane's own diagonal block: + Xt^T Xt + damping (contact) / identity (swing)
    M33<S> Gd;
    {
        const S dg = 1.0 - cl;      // (the damping 1e-12 of forwardDynamics, WBM.cpp:411, enters at the factorisation: the cache keeps the undamped Gram matrix)
        _Pragma("unroll")
        for (int r = 0; r < 3; r++) {
            const S e0 = dot3(Xt.r[r], Xt.r[0]), e1 = dot3(Xt.r[r], Xt.r[1]), e2 = dot3(Xt.r[r], Xt.r[2]);
            Gd.r[r] = {e0 + (r == 0 ? dg : zero), e1 + (r == 1 ? dg : zero), e2 + (r == 2 ? dg : zero)};
        }
        // added to G[own lane]: by selects on the lane's leg
        const S m0 = Q::legc(1, 0, 0, 0), m1 = Q::legc(0, 1, 0, 0), m2 = Q::legc(0, 0, 1, 0), m3 = Q::legc(0, 0, 0, 1);
        _Pragma("unroll")
        for (int r = 0; r < 3; r++) {
            G[0].r[r] = G[0].r[r] + scale(m0, Gd.r[r]); G[1].r[r] = G[1].r[r] + scale(m1, Gd.r[r]);
            G[2].r[r] = G[2].r[r] + scale(m2, Gd.r[r]); G[3].r[r] = G[3].r[r] + scale(m3, Gd.r[r]);
        }
    }
    if (WR) {      // Gram matrix (undamped; identity on a swing foot's block)
        const S cbg[4] = {Q::template get<0>(cb), Q::template get<1>(cb), Q::template get<2>(cb), Q::template get<3>(cb)};
        _Pragma("unroll") for (int g = 0; g < 4; g++) _Pragma("unroll") for (int r = 0; r < 3; r++) {
            const S base = 36.0 * cb + (double)(12 * r) + 3.0 * cbg[g];
            Q::stv(kc, KC_LG, base, all, G[g].r[r].x); Q::stv(kc, KC_LG + 1, base, all, G[g].r[r].y); Q::stv(kc, KC_LG + 2, base, all, G[g].r[r].z);
        }
    }
    V3<S> rhs;
    {
        S xy[3];
        _Pragma("unroll") for (int d = 0; d < 3; d++) { S s = dot3(Xt.r[d], yl); _Pragma("unroll") for (int c = 0; c < 6; c++) s = s + Xb[c][d] * yb[c]; xy[d] = s; }
        const double a2 = 2.0 * P.bg_alpha;
        rhs = {cl * (-xy[0] - (jdv.x + a2 * fvel.x)), cl * (-xy[1] - (jdv.y + a2 * fvel.y)), cl * (-xy[2] - (jdv.z + a2 * fvel.z))};
    }
    QP(9)      // Gram blocks, rhs
    // block Cholesky of G over the quad: block column kc is finished by lane kc (its diagonal block), then lanes f > kc form L_f,kc.
    // Lane f keeps its block row Lg[0..f]; blocks right of the diagonal are never used.
    M33<S> Lg[4]; Chol3<S> Ld;      // Lg[g]: block (own lane, g) of the factor, g < own lane ; Ld: the own diagonal block's factor
    {
        const S lane = Q::legc(0, 1, 2, 3);
        auto bcast33 = [&](auto JT, const M33<S>& A) { constexpr int J = decltype(JT)::value; M33<S> o; _Pragma("unroll") for (int r = 0; r < 3; r++) o.r[r] = {Q::template get<J>(A.r[r].x), Q::template get<J>(A.r[r].y), Q::template get<J>(A.r[r].z)}; return o; };
        auto bcastL = [&](auto JT, const Chol3<S>& A) { constexpr int J = decltype(JT)::value; Chol3<S> o; o.l10 = Q::template get<J>(A.l10); o.l20 = Q::template get<J>(A.l20); o.l21 = Q::template get<J>(A.l21); o.r0 = Q::template get<J>(A.r0); o.r1 = Q::template get<J>(A.r1); o.r2 = Q::template get<J>(A.r2); return o; };
        // A <- A - P Q^T (3x3 blocks, rows)
        auto sub_abt = [&](M33<S>& A, const M33<S>& Pm, const M33<S>& Qm) { _Pragma("unroll") for (int r = 0; r < 3; r++) A.r[r] = A.r[r] - V3<S>{dot3(Pm.r[r], Qm.r[0]), dot3(Pm.r[r], Qm.r[1]), dot3(Pm.r[r], Qm.r[2])}; };
        // rows of A <- (rows of A) L^-T : solve x L^T = a per row
        auto right_solve = [&](M33<S>& A, const Chol3<S>& Lk) { _Pragma("unroll") for (int r = 0; r < 3; r++) { V3<S> a = A.r[r], x; x.x = a.x * Lk.r0; x.y = (a.y - x.x * Lk.l10) * Lk.r1; x.z = (a.z - x.x * Lk.l20 - x.y * Lk.l21) * Lk.r2; A.r[r] = x; } };
        // own diagonal block = G[own]: picked by selects (each lane needs ITS block at ITS step; every lane runs every step)
        const S dmp = cl * 1e-12;
        auto own_diag = [&](const M33<S>& Gk) { return chol3<Q, S>(Gk.r[0].x + dmp, Gk.r[1].x, Gk.r[1].y + dmp, Gk.r[2].x, Gk.r[2].y, Gk.r[2].z + dmp); };
        // step 0: lane 0's diagonal block is final
        Chol3<S> L0 = own_diag(G[0]);                       // meaningful in lane 0
        const Chol3<S> L00 = bcastL(J0{}, L0);
        Lg[0] = G[0]; right_solve(Lg[0], L00);              // lanes 1..3: L_f0 (lane 0's own copy is not used)
        // step 1
        const M33<S> L10 = bcast33(J1{}, Lg[0]);
        M33<S> A1 = G[1]; sub_abt(A1, Lg[0], L10);          // lanes >= 1: G_f1 - L_f0 L_10^T
        Chol3<S> L1 = own_diag(A1);                         // meaningful in lane 1
        const Chol3<S> L11 = bcastL(J1{}, L1);
        Lg[1] = A1; right_solve(Lg[1], L11);                // lanes 2..3: L_f1
        // step 2
        const M33<S> L20 = bcast33(J2{}, Lg[0]), L21 = bcast33(J2{}, Lg[1]);
        M33<S> A2 = G[2]; sub_abt(A2, Lg[0], L20); sub_abt(A2, Lg[1], L21);
        Chol3<S> L2 = own_diag(A2);                         // meaningful in lane 2
        const Chol3<S> L22 = bcastL(J2{}, L2);
        Lg[2] = A2; right_solve(Lg[2], L22);                // lane 3: L_32
        // step 3
        M33<S> A3 = G[3]; sub_abt(A3, Lg[0], Lg[0]); sub_abt(A3, Lg[1], Lg[1]); sub_abt(A3, Lg[2], Lg[2]);      // lane 3: G_33 - sum L_3g L_3g^T
        Chol3<S> L3 = own_diag(A3);                         // meaningful in lane 3
        // every lane keeps its own diagonal factor
        const typename Q::B is0 = Q::gt(S(0.5), lane), is1 = Q::gt(S(1.5), lane), is2 = Q::gt(S(2.5), lane);
        auto pick = [&](const S& a0, const S& a1, const S& a2, const S& a3) { return Q::sel(is0, a0, Q::sel(is1, a1, Q::sel(is2, a2, a3))); };
        Ld.l10 = pick(L0.l10, L1.l10, L2.l10, L3.l10); Ld.l20 = pick(L0.l20, L1.l20, L2.l20, L3.l20); Ld.l21 = pick(L0.l21, L1.l21, L2.l21, L3.l21);
        Ld.r0 = pick(L0.r0, L1.r0, L2.r0, L3.r0); Ld.r1 = pick(L0.r1, L1.r1, L2.r1, L3.r1); Ld.r2 = pick(L0.r2, L1.r2, L2.r2, L3.r2);
        // lam = G^-1 rhs: forward over block rows 0..3, backward 3..0.  z_f lives in lane f.
        V3<S> z = rhs;
        auto bc3 = [&](auto JT, const V3<S>& w) { constexpr int J = decltype(JT)::value; return V3<S>{Q::template get<J>(w.x), Q::template get<J>(w.y), Q::template get<J>(w.z)}; };
        V3<S> z0 = fwd3(Ld, z);                                               // valid in lane 0
        const V3<S> Z0 = bc3(J0{}, z0);
        V3<S> t1 = z - m33_mul(Lg[0], Z0); V3<S> z1 = fwd3(Ld, t1);            // valid in lane 1
        const V3<S> Z1 = bc3(J1{}, z1);
        V3<S> t2 = t1 - m33_mul(Lg[1], Z1); V3<S> z2 = fwd3(Ld, t2);           // valid in lane 2
        const V3<S> Z2 = bc3(J2{}, z2);
        V3<S> t3 = t2 - m33_mul(Lg[2], Z2); V3<S> z3 = fwd3(Ld, t3);           // valid in lane 3
        z = {pick(z0.x, z1.x, z2.x, z3.x), pick(z0.y, z1.y, z2.y, z3.y), pick(z0.z, z1.z, z2.z, z3.z)};
        // backward: lam_3 = L_33^-T z_3 ; lam_k = L_kk^-T (z_k - sum_{f>k} L_fk^T lam_f) - the product L_fk^T lam_f is formed in lane f and read by lane k
        V3<S> lam3 = bwd3(Ld, z);                                             // valid in lane 3
        const V3<S> w32 = bc3(J3{}, m33_mulT(Lg[2], lam3)), w31 = bc3(J3{}, m33_mulT(Lg[1], lam3)), w30 = bc3(J3{}, m33_mulT(Lg[0], lam3));
        V3<S> lam2 = bwd3(Ld, z - w32);                                       // valid in lane 2
        const V3<S> w21 = bc3(J2{}, m33_mulT(Lg[1], lam2)), w20 = bc3(J2{}, m33_mulT(Lg[0], lam2));
        V3<S> lam1 = bwd3(Ld, z - w31 - w21);                                 // valid in lane 1
        const V3<S> w10 = bc3(J1{}, m33_mulT(Lg[0], lam1));
        V3<S> lam0 = bwd3(Ld, z - w30 - w20 - w10);                           // valid in lane 0
        rhs = {pick(lam0.x, lam1.x, lam2.x, lam3.x), pick(lam0.y, lam1.y, lam2.y, lam3.y), pick(lam0.z, lam1.z, lam2.z, lam3.z)};
    }
    QP(10)     // block Cholesky + lam
    // (second half of the tail's reads - foot-cost references, friction-pyramid parameters - behind the register peak of the factorisation)
    S rel[3], fvr[3], rc, eG[5], dG[5];
#ifndef QUAD_NEXT_EARLY
    read_next();
#endif
    {
        _Pragma("unroll") for (int j = 0; j < 3; j++) { rel[j] = Q::ld(rr, 64 + j, 3); fvr[j] = Q::ld(rr, 48 + j, 3); }
        rc = Q::ld(rr, 60, 1);
        _Pragma("unroll") for (int r = 0; r < 5; r++) { eG[r] = zero; dG[r] = S(1.0); }
        if (P.go_grf >= 0) {      // per-lane constraint index go_grf + 5 slot + r: a swing lane reads a valid dummy (slot 0) and contributes nothing
            const S ci0 = Q::sel(on, 5.0 * before, S(0.0));      // (one per-lane base, constant offsets r: the five loads share an address register)
            _Pragma("unroll") for (int r = 0; r < 5; r++) { eG[r] = Q::ldv(P.eps, gk + P.go_grf + r, ci0); dG[r] = Q::ldv(P.delta, gk + P.go_grf + r, ci0); }
        }
    }
    const V3<S> lam = scale(cl, rhs);      // contact force of the lane's foot (world axes); zero for a swing leg
    // qdd = L^-T (y + X lam): base part replicated, leg part in the lane
    S qddb[6]; V3<S> qddl;
    {
        _Pragma("unroll") for (int c = 0; c < 6; c++) qddb[c] = yb[c] + Q::sum(Xb[c][0] * lam.x + Xb[c][1] * lam.y + Xb[c][2] * lam.z);
        bwd6(LS, rdS, qddb);
        V3<S> zl = yl + V3<S>{Xt.r[0].x * lam.x + Xt.r[1].x * lam.y + Xt.r[2].x * lam.z, Xt.r[0].y * lam.x + Xt.r[1].y * lam.y + Xt.r[2].y * lam.z, Xt.r[0].z * lam.x + Xt.r[1].z * lam.y + Xt.r[2].z * lam.z};
        S et[3];
        _Pragma("unroll") for (int j = 0; j < 3; j++) { S s = E[0][j] * qddb[0]; _Pragma("unroll") for (int c = 1; c < 6; c++) s = s + E[c][j] * qddb[c]; et[j] = s; }
        qddl = bwd3(Ll, zl - V3<S>{et[0], et[1], et[2]});
    }

    QP(11)     // qdd
    QP(12)     // (tail reads: issued ahead of the contact solve)
    if (WR) {
        _Pragma("unroll") for (int c = 0; c < 6; c++) Q::st0(kc, KC_QDD + c, qddb[c]);
        Q::st(kc, KC_QDD + 6, 3, qddl.x); Q::st(kc, KC_QDD + 7, 3, qddl.y); Q::st(kc, KC_QDD + 8, 3, qddl.z);
        Q::st(kc, KC_GRF, 3, lam.x); Q::st(kc, KC_GRF + 1, 3, lam.y); Q::st(kc, KC_GRF + 2, 3, lam.z);
        Q::stv(kc, KC_LAM, 3.0 * cb, all, lam.x); Q::stv(kc, KC_LAM + 1, 3.0 * cb, all, lam.y); Q::stv(kc, KC_LAM + 2, 3.0 * cb, all, lam.z);
    }
    // ---- integrate (forward Euler, WBM.cpp:25-26), defect of knot k+1, divergence norm
    S dsq = zero, nsq = zero;
    {
        _Pragma("unroll")
        for (int i = 0; i < 6; i++) {
            const S xs = qb[i] + vb[i] * dt, vs = vb[i] + qddb[i] * dt;
            const S d0 = xs - xnb[i], d1 = vs - vnb[i];
            dsq = dsq + w0 * (d0 * d0 + d1 * d1); nsq = nsq + w0 * (xs * xs + vs * vs);
            if (WR) { Q::st0(P.Xsim, kx + 36 + i, xs); Q::st0(P.Xsim, kx + 54 + i, vs); Q::st0(P.Defect, kx + 36 + i, d0); Q::st0(P.Defect, kx + 54 + i, d1); }
        }
        const S qd3[3] = {qddl.x, qddl.y, qddl.z};
        _Pragma("unroll")
        for (int j = 0; j < 3; j++) {
            const S xs = ql[j] + vl_[j] * dt, vs = vl_[j] + qd3[j] * dt;
            const S d0 = xs - xnl[j], d1 = vs - vnl[j];
            dsq = dsq + (d0 * d0 + d1 * d1); nsq = nsq + (xs * xs + vs * vs);
            if (WR) { Q::st(P.Xsim, kx + 42 + j, 3, xs); Q::st(P.Xsim, kx + 60 + j, 3, vs); Q::st(P.Defect, kx + 42 + j, 3, d0); Q::st(P.Defect, kx + 60 + j, 3, d1); }
        }
        if (x0 != nullptr && k == 0) {      // very first knot of the horizon: the defect against the initial condition (SinglePhase.cpp:185, compute_defect)
            _Pragma("unroll")
            for (int i = 0; i < 6; i++) {
                const S a = Q::ld(x0, (size_t)b * 36 + i, 0), c = Q::ld(x0, (size_t)b * 36 + 18 + i, 0);
                const S d0 = a - qb[i], d1 = c - vb[i];
                dsq = dsq + w0 * (d0 * d0 + d1 * d1);
                if (WR) { Q::st0(P.Xsim, kx + i, a); Q::st0(P.Xsim, kx + 18 + i, c); Q::st0(P.Defect, kx + i, d0); Q::st0(P.Defect, kx + 18 + i, d1); }
            }
            _Pragma("unroll")
            for (int j = 0; j < 3; j++) {
                const S a = Q::ld(x0, (size_t)b * 36 + 6 + j, 3), c = Q::ld(x0, (size_t)b * 36 + 24 + j, 3);
                const S d0 = a - ql[j], d1 = c - vl_[j];
                dsq = dsq + (d0 * d0 + d1 * d1);
                if (WR) { Q::st(P.Xsim, kx + 6 + j, 3, a); Q::st(P.Xsim, kx + 24 + j, 3, c); Q::st(P.Defect, kx + 6 + j, 3, d0); Q::st(P.Defect, kx + 24 + j, 3, d1); }
            }
        }
        dsq = Q::sum(dsq); nsq = Q::sum(nsq);
    }
    if (WR) { Q::st(P.Y, kk * 12, 3, lam.x); Q::st(P.Y, kk * 12 + 1, 3, lam.y); Q::st(P.Y, kk * 12 + 2, 3, lam.z); }
    QP(13)     // integrate, defect
    // ---- foot costs of the lane's foot (MHPCCost.cpp:4-245); the running cost in the reference's order of additions: tracking, foot-place
    // regulariser, swing position, swing velocity, then dt x the ReB cost of each constraint object (SinglePhase.cpp:394-402)
    S l = lq;
    {
        const V3<S> d = {(fpos.x - qb[0]) - rel[0], (fpos.y - qb[1]) - rel[1], (fpos.z - qb[2]) - rel[2]};
        const V3<S> dv = {fvel.x - fvr[0], fvel.y - fvr[1], fvel.z - fvr[2]};
        const typename Q::B stance = Q::gt(rc, S(0.0)), swing = Q::gt(S(0.5), rc);      // (rc == 0 <=> swing: the flags are 0 / 1)
        const S wr0 = P.w_foot_reg[0], wr1 = P.w_foot_reg[1], wr2 = P.w_foot_reg[2], wp0 = P.w_swing_pos[0], wp1 = P.w_swing_pos[1], wp2 = P.w_swing_pos[2];
        const S wv0 = P.w_swing_vel[0], wv1 = P.w_swing_vel[1], wv2 = P.w_swing_vel[2];
        const S l2 = 0.5 * (d.x * wr0 * d.x + d.y * wr1 * d.y + d.z * wr2 * d.z) * dt, l3 = 0.5 * (d.x * wp0 * d.x + d.y * wp1 * d.y + d.z * wp2 * d.z) * dt;
        const S l4 = 0.5 * (dv.x * wv0 * dv.x + dv.y * wv1 * dv.y + dv.z * wv2 * dv.z) * dt;
        l = l + Q::sum(Q::sel(stance, (P.w_foot_reg[0] >= 0) ? l2 : zero, zero));
        l = l + Q::sum(Q::sel(swing, (P.w_swing_pos[0] >= 0) ? l3 : zero, zero));
        l = l + Q::sum(Q::sel(swing, (P.w_swing_vel[0] >= 0) ? l4 : zero, zero));
    }
    if (WR) Q::st0(P.lbase, kk, l);
    QP(14)     // foot costs
    if (reb_active) {
        if (P.go_torque >= 0) l = l + dt * accT;
        if (P.go_jspeed >= 0) l = l + dt * accS;
        if (P.go_joint >= 0) l = l + dt * accJ;
        if (P.go_height >= 0) l = l + dt * accH;
    }
    if (P.go_grf >= 0) {      // friction pyramid of the lane's foot (MHPCConstraint.cpp:9-70); its slot among the contact feet = number of contact feet before it
        S acc = zero;
        const double mu = P.mu;
        const S gs[5] = {lam.z, -lam.x + mu * lam.z, lam.x + mu * lam.z, -lam.y + mu * lam.z, lam.y + mu * lam.z};
        _Pragma("unroll")
        for (int r = 0; r < 5; r++) {
            const S g = Q::sel(on, gs[r], S(1.0));
            if (WR) Q::stv(P.g, gk + P.go_grf, Q::sel(on, 5.0 * before + (double)r, S(0.0)), on, g);
            acc = acc + Q::sel(on, eG[r] * q_barrier<Q, S>(g, dG[r]), zero);
            gmin = Q::min(gmin, Q::sel(on, g, zero));
        }
        if (reb_active) l = l + dt * Q::sum(acc);
    }
    gmin = Q::vmin(gmin);
    QP(15)     // constraints + barriers
    if (WR) Q::st0(P.l, kk, l);
    QuadOut o;
    o.cost = Q::lane0(l); o.dsq = Q::lane0(dsq); o.ming = Q::lane0(gmin);
    const double ns = Q::lane0(nsq);
    o.bad = (ns > 1e12) || !(ns == ns);      // ||Xsim|| > 1e6 (SinglePhase.cpp:205)
    return o;
}

}  // namespace hs
