// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/load this.
//
// CPU restatement (plain C++17, no Eigen / Pinocchio) of the Mini-Cheetah whole-body model
// of the reference:
//   MHPC/MHPC-Trajopt/WBM.cpp:17-139   dynamics / dynamics_partial (forward Euler, C,D unscaled)
//   MHPC/MHPC-Trajopt/WBM.cpp:368-424  KKTContactDynamics (Pinocchio forwardDynamics, damping 1e-12)
//   MHPC/MHPC-Trajopt/WBM.cpp:459-505  KKTContactDynamicsDerivatives
//   MHPC/MHPC-Trajopt/WBM.cpp:427-456, 508-543  KKTImpact / KKTImpactDerivatives
//   MHPC/MHPC-Trajopt/PinocchioInteface.cpp:17-56  joint order PX,PY,PZ,RZ,RY,RX + 4x(RX,RY,RY)
//   urdf/mini_cheetah_simple_correctedInertia.urdf  (inertial + kinematic constants)
// Pinocchio 2.6.10 itself is a third-party dependency that is NOT under /root/reference; its
// published algorithms (RNEA, CRBA-equivalent, contact KKT inverse) are restated here and pinned by
// the golden vectors of MHPC/MHPC-Trajopt/test/testKKTDynamics.cpp:97-121 and by the reference's
// CasADi-generated kinematic derivative functions (tests/golden/casadi_wb_*.npz).
//
// Method: one generic recursive Newton-Euler pass templated on the scalar type. With S=double it
// yields M (column by column), h, J, Jdot*v.  With S=Dual (forward-mode tangent) it yields exact
// analytic derivatives column by column, which is what computeRNEADerivatives and the CasADi
// functions footVel/Acc/ForcePartial* return in the reference.
#pragma once
#include <cmath>
#include <cstring>

namespace orc {

struct Dual {
    double v, d;
    Dual() : v(0), d(0) {}
    Dual(double a) : v(a), d(0) {}
    Dual(double a, double b) : v(a), d(b) {}
};
inline Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
inline Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
inline Dual operator-(Dual a) { return {-a.v, -a.d}; }
inline Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.v * b.d + a.d * b.v}; }
inline Dual& operator+=(Dual& a, Dual b) { a = a + b; return a; }
inline Dual& operator-=(Dual& a, Dual b) { a = a - b; return a; }
inline Dual sin(Dual a) { return {std::sin(a.v), std::cos(a.v) * a.d}; }
inline Dual cos(Dual a) { return {std::cos(a.v), -std::sin(a.v) * a.d}; }
inline double val(double a) { return a; }
inline double val(Dual a) { return a.v; }
inline double tan_(double) { return 0; }
inline double tan_(Dual a) { return a.d; }
using std::sin;
using std::cos;

template <class S> struct V3 { S x, y, z; };
template <class S> inline V3<S> operator+(V3<S> a, V3<S> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class S> inline V3<S> operator-(V3<S> a, V3<S> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class S> inline V3<S> operator*(S s, V3<S> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class S> inline V3<S> cross(V3<S> a, V3<S> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class S> inline S dot(V3<S> a, V3<S> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class S> struct M3 { S m[3][3]; };  // m[row][col]
template <class S> inline V3<S> mul(const M3<S>& R, V3<S> a) {
    return {R.m[0][0] * a.x + R.m[0][1] * a.y + R.m[0][2] * a.z, R.m[1][0] * a.x + R.m[1][1] * a.y + R.m[1][2] * a.z,
            R.m[2][0] * a.x + R.m[2][1] * a.y + R.m[2][2] * a.z};
}
template <class S> inline V3<S> mulT(const M3<S>& R, V3<S> a) {
    return {R.m[0][0] * a.x + R.m[1][0] * a.y + R.m[2][0] * a.z, R.m[0][1] * a.x + R.m[1][1] * a.y + R.m[2][1] * a.z,
            R.m[0][2] * a.x + R.m[1][2] * a.y + R.m[2][2] * a.z};
}
template <class S> inline M3<S> mul(const M3<S>& A, const M3<S>& B) {
    M3<S> C;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return C;
}

// ---- model constants (urdf/mini_cheetah_simple_correctedInertia.urdf:7-140 and mirrored legs) ----
constexpr int NQ = 18, NX = 36, NU = 12, NY = 12;
constexpr double GRAV = 9.81;  // Pinocchio default gravity (0,0,-9.81)
struct Link { double m, c[3], I[6]; };  // I = xx,xy,xz,yy,yz,zz about COM in link axes
struct WbConst {
    // joint i: type 0 prismatic / 1 revolute; axis 0,1,2; parent; fixed placement p0 and yaw R0=Rz(psi0)
    int type[NQ], axis[NQ], parent[NQ];
    double p0[NQ][3];
    bool has_psi[NQ];
    Link link[NQ];  // mass 0 for the 5 virtual base joints
    double foot_r[3];
    int foot_joint[4];
    WbConst() {
        std::memset(this, 0, sizeof(*this));
        const int btype[6] = {0, 0, 0, 1, 1, 1}, baxis[6] = {0, 1, 2, 2, 1, 0};
        for (int i = 0; i < 6; i++) { type[i] = btype[i]; axis[i] = baxis[i]; parent[i] = i - 1; }
        link[5] = {3.3, {0, 0, 0}, {0.011253, 0, 0, 0.036203, 0, 0.042673}};
        const double sxs[4] = {1, 1, -1, -1}, sys[4] = {1, -1, 1, -1};  // FL, FR, HL, HR
        for (int l = 0; l < 4; l++) {
            double sx = sxs[l], sy = sys[l];
            int a = 6 + 3 * l, h = a + 1, k = a + 2;
            type[a] = type[h] = type[k] = 1;
            axis[a] = 0; axis[h] = 1; axis[k] = 1;
            parent[a] = 5; parent[h] = a; parent[k] = h;
            p0[a][0] = sx * 0.19; p0[a][1] = sy * 0.049; p0[a][2] = 0;
            p0[h][0] = 0; p0[h][1] = sy * 0.062; p0[h][2] = 0;
            has_psi[h] = true;
            p0[k][0] = 0; p0[k][1] = 0; p0[k][2] = -0.209;
            link[a] = {0.54, {0, sy * 0.036, 0}, {0.000381, sy * 0.000058, 0.00000045, 0.000560, sy * 0.00000095, 0.000444}};
            link[h] = {0.634, {0, sy * 0.016, -0.02}, {0.001983, sy * 0.000245, 0.000013, 0.002103, sy * 0.0000015, 0.000408}};
            link[k] = {0.064, {0, 0, -0.061}, {0.000245, 0, 0, 0.000248, 0, 0.000006}};
            foot_joint[l] = k;
        }
        foot_r[0] = 0; foot_r[1] = 0; foot_r[2] = -0.195;
    }
};
inline const WbConst& wbc() { static WbConst c; return c; }

template <class S> struct PassOut {
    S tau[NQ];
    V3<S> foot_pos[4], foot_vel[4], foot_acc[4];
};

// Generic recursive Newton-Euler pass (link-local Pluecker coordinates), one-dof joints.
//  psi        : thigh-joint fixed yaw offset used for THIS pass (quirk xii: 3.1415 vs pi)
//  inertia_on : include link inertias (off -> pure kinematics + external-force torques)
//  gravity_on : fictitious base acceleration +g (foot_acc is corrected for it)
//  fext       : world-frame force applied AT each foot point (nullptr -> none)
template <class S>
inline void wb_pass(double psi, bool inertia_on, bool gravity_on, const S* q, const S* v, const S* a,
                    const double (*fext)[3], PassOut<S>& out) {
    const WbConst& c = wbc();
    M3<S> Rrel[NQ], Rw[NQ];
    V3<S> prel[NQ], ow[NQ], om[NQ], vl[NQ], al[NQ], aa[NQ], ff[NQ], nn[NQ];
    const double cps = std::cos(psi), sps = std::sin(psi);
    for (int i = 0; i < NQ; i++) {
        int p = c.parent[i];
        V3<S> omp{S(0), S(0), S(0)}, vp = omp, alp = omp, ap = omp, op = omp;
        M3<S> Rwp;
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) Rwp.m[r][cc] = S(r == cc ? 1.0 : 0.0);
        if (p >= 0) { omp = om[p]; vp = vl[p]; alp = aa[p]; ap = al[p]; op = ow[p]; Rwp = Rw[p]; }
        else if (gravity_on) ap.z = S(GRAV);
        V3<S> e{S(c.axis[i] == 0 ? 1.0 : 0.0), S(c.axis[i] == 1 ? 1.0 : 0.0), S(c.axis[i] == 2 ? 1.0 : 0.0)};
        V3<S> p0{S(c.p0[i][0]), S(c.p0[i][1]), S(c.p0[i][2])};
        if (c.type[i] == 0) {  // prismatic along e, no fixed rotation for the base joints
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) Rrel[i].m[r][cc] = S(r == cc ? 1.0 : 0.0);
            prel[i] = p0 + q[i] * e;
            om[i] = omp;
            vl[i] = vp + cross(omp, prel[i]) + v[i] * e;
            aa[i] = alp;
            al[i] = ap + cross(alp, prel[i]) + a[i] * e + cross(om[i], v[i] * e);
        } else {
            S cq = cos(q[i]), sq = sin(q[i]);
            M3<S> Rj;
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) Rj.m[r][cc] = S(r == cc ? 1.0 : 0.0);
            int a1 = (c.axis[i] + 1) % 3, a2 = (c.axis[i] + 2) % 3;
            Rj.m[a1][a1] = cq; Rj.m[a1][a2] = -sq; Rj.m[a2][a1] = sq; Rj.m[a2][a2] = cq;
            if (c.has_psi[i]) {
                M3<S> R0;
                for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) R0.m[r][cc] = S(r == cc ? 1.0 : 0.0);
                R0.m[0][0] = S(cps); R0.m[0][1] = S(-sps); R0.m[1][0] = S(sps); R0.m[1][1] = S(cps);
                Rrel[i] = mul(R0, Rj);
            } else Rrel[i] = Rj;
            prel[i] = p0;
            om[i] = mulT(Rrel[i], omp) + v[i] * e;
            vl[i] = mulT(Rrel[i], vp + cross(omp, p0));
            aa[i] = mulT(Rrel[i], alp) + a[i] * e + cross(om[i], v[i] * e);
            al[i] = mulT(Rrel[i], ap + cross(alp, p0)) + cross(vl[i], v[i] * e);
        }
        Rw[i] = mul(Rwp, Rrel[i]);
        ow[i] = op + mul(Rwp, prel[i]);
        ff[i] = {S(0), S(0), S(0)};
        nn[i] = ff[i];
        const Link& L = c.link[i];
        if (inertia_on && L.m > 0) {
            V3<S> cm{S(L.c[0]), S(L.c[1]), S(L.c[2])};
            auto Imul = [&](V3<S> w) {
                return V3<S>{S(L.I[0]) * w.x + S(L.I[1]) * w.y + S(L.I[2]) * w.z,
                             S(L.I[1]) * w.x + S(L.I[3]) * w.y + S(L.I[4]) * w.z,
                             S(L.I[2]) * w.x + S(L.I[4]) * w.y + S(L.I[5]) * w.z};
            };
            V3<S> hl = S(L.m) * (vl[i] + cross(om[i], cm));
            V3<S> ha = Imul(om[i]) + cross(cm, hl);
            V3<S> f = S(L.m) * (al[i] + cross(aa[i], cm));
            V3<S> n = Imul(aa[i]) + cross(cm, f);
            ff[i] = f + cross(om[i], hl);
            nn[i] = n + cross(om[i], ha) + cross(vl[i], hl);
        }
    }
    V3<S> r{S(c.foot_r[0]), S(c.foot_r[1]), S(c.foot_r[2])};
    for (int l = 0; l < 4; l++) {
        int k = c.foot_joint[l];
        out.foot_pos[l] = ow[k] + mul(Rw[k], r);
        V3<S> vp_ = vl[k] + cross(om[k], r);
        out.foot_vel[l] = mul(Rw[k], vp_);
        V3<S> acc = al[k] + cross(aa[k], r) + cross(om[k], vp_);
        out.foot_acc[l] = mul(Rw[k], acc);
        if (gravity_on) out.foot_acc[l].z -= S(GRAV);
        if (fext) {
            V3<S> Fl = mulT(Rw[k], V3<S>{S(fext[l][0]), S(fext[l][1]), S(fext[l][2])});
            ff[k] = ff[k] - Fl;
            nn[k] = nn[k] - cross(r, Fl);
        }
    }
    for (int i = NQ - 1; i >= 0; i--) {
        S t;
        if (c.type[i] == 0) t = (c.axis[i] == 0 ? ff[i].x : c.axis[i] == 1 ? ff[i].y : ff[i].z);
        else t = (c.axis[i] == 0 ? nn[i].x : c.axis[i] == 1 ? nn[i].y : nn[i].z);
        out.tau[i] = t;
        int p = c.parent[i];
        if (p >= 0) {
            V3<S> fp = mul(Rrel[i], ff[i]);
            ff[p] = ff[p] + fp;
            nn[p] = nn[p] + mul(Rrel[i], nn[i]) + cross(prel[i], fp);
        }
    }
}

// ---------------- small dense helpers (row-major scratch, column-major only at the API) ------------
// Cholesky A = L L^T in place (lower). returns false if not PD.
inline bool chol(double* A, int n) {
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return false;
        d = std::sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return true;
}
inline void chol_solve(const double* L, int n, double* b) {  // in place
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k]; b[i] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * b[k]; b[i] = s / L[i * n + i]; }
}

struct WbParams { double psi_dyn = 3.1415, psi_kin = M_PI, bg_alpha = 10.0; bool impulse_quirk = true; /* quirk v: WBM.cpp:454 */ };

// Result of the contact KKT evaluation (WBM.cpp:368-424).  All matrices row-major here.
struct WbKKT {
    int nc; int feet[4];
    double M[NQ * NQ], h[NQ], J[12 * NQ], gamma[12];
    double qdd[NQ], lam[12], grf[12];
    V3<double> foot_pos[4], foot_vel[4];
};

inline void wb_terms(const WbParams& P, const double* q, const double* v, WbKKT& K) {
    PassOut<double> o;
    double z[NQ] = {0}, e[NQ];
    V3<double> Jc[4][NQ];
    for (int j = 0; j < NQ; j++) {
        std::memset(e, 0, sizeof(e)); e[j] = 1;
        wb_pass<double>(P.psi_dyn, true, false, q, z, e, nullptr, o);
        for (int i = 0; i < NQ; i++) K.M[i * NQ + j] = o.tau[i];
        for (int l = 0; l < 4; l++) Jc[l][j] = o.foot_acc[l];
    }
    wb_pass<double>(P.psi_dyn, true, true, q, v, z, nullptr, o);
    for (int i = 0; i < NQ; i++) K.h[i] = o.tau[i];
    for (int l = 0; l < 4; l++) { K.foot_pos[l] = o.foot_pos[l]; K.foot_vel[l] = o.foot_vel[l]; }
    for (int i = 0; i < K.nc; i++) {
        int l = K.feet[i];
        for (int j = 0; j < NQ; j++) {
            K.J[(3 * i + 0) * NQ + j] = Jc[l][j].x; K.J[(3 * i + 1) * NQ + j] = Jc[l][j].y; K.J[(3 * i + 2) * NQ + j] = Jc[l][j].z;
        }
        // gamma = Jdot*v (classical, qdd=0) + 2*alpha*J*v   (WBM.cpp:392-408)
        K.gamma[3 * i + 0] = o.foot_acc[l].x + 2 * P.bg_alpha * o.foot_vel[l].x;
        K.gamma[3 * i + 1] = o.foot_acc[l].y + 2 * P.bg_alpha * o.foot_vel[l].y;
        K.gamma[3 * i + 2] = o.foot_acc[l].z + 2 * P.bg_alpha * o.foot_vel[l].z;
    }
}

inline void set_contacts(WbKKT& K, const int* contact) {
    K.nc = 0;
    for (int l = 0; l < 4; l++) if (contact[l] > 0) K.feet[K.nc++] = l;
}

// Forward contact dynamics.  tau is the full 18-vector (S*u).  (Pinocchio forwardDynamics semantics:
// lambda = (J Minv J^T + damping I)^-1 (-J Minv (tau-h) - gamma); qdd = Minv (tau - h + J^T lambda))
inline void wb_forward(const WbParams& P, const double* q, const double* v, const double* tau, const int* contact, WbKKT& K) {
    set_contacts(K, contact);
    wb_terms(P, q, v, K);
    double L[NQ * NQ]; std::memcpy(L, K.M, sizeof(L)); chol(L, NQ);
    double r[NQ];
    for (int i = 0; i < NQ; i++) r[i] = tau[i] - K.h[i];
    double a0[NQ]; std::memcpy(a0, r, sizeof(r)); chol_solve(L, NQ, a0);
    std::memset(K.grf, 0, sizeof(K.grf)); std::memset(K.lam, 0, sizeof(K.lam));
    int m = 3 * K.nc;
    if (m > 0) {
        double MiJt[NQ * 12], G[12 * 12], rhs[12];
        for (int c = 0; c < m; c++) {
            double col[NQ]; for (int i = 0; i < NQ; i++) col[i] = K.J[c * NQ + i];
            chol_solve(L, NQ, col);
            for (int i = 0; i < NQ; i++) MiJt[i * 12 + c] = col[i];
        }
        for (int a = 0; a < m; a++) for (int b = 0; b < m; b++) {
            double s = 0; for (int i = 0; i < NQ; i++) s += K.J[a * NQ + i] * MiJt[i * 12 + b];
            G[a * m + b] = s + (a == b ? 1e-12 : 0.0);
        }
        chol(G, m);
        for (int a = 0; a < m; a++) { double s = 0; for (int i = 0; i < NQ; i++) s += K.J[a * NQ + i] * a0[i]; rhs[a] = -s - K.gamma[a]; }
        chol_solve(G, m, rhs);
        for (int a = 0; a < m; a++) K.lam[a] = rhs[a];
        for (int i = 0; i < NQ; i++) { double s = r[i]; for (int a = 0; a < m; a++) s += K.J[a * NQ + i] * rhs[a]; r[i] = s; }
        chol_solve(L, NQ, r);
        std::memcpy(K.qdd, r, sizeof(r));
        for (int i = 0; i < K.nc; i++) for (int d = 0; d < 3; d++) K.grf[3 * K.feet[i] + d] = K.lam[3 * i + d];
    } else std::memcpy(K.qdd, a0, sizeof(a0));
}

// KKT matrix inverse blocks (Pinocchio getKKTContactDynamicMatrixInverse; damping 0).
// Kinv is (18+m)x(18+m) row-major with leading dimension 30.
inline void wb_kkt_inverse(const WbKKT& K, double* Kinv) {
    const int LD = 30; int m = 3 * K.nc;
    double L[NQ * NQ]; std::memcpy(L, K.M, sizeof(L)); chol(L, NQ);
    double Mi[NQ * NQ];
    for (int j = 0; j < NQ; j++) { double col[NQ] = {0}; col[j] = 1; chol_solve(L, NQ, col); for (int i = 0; i < NQ; i++) Mi[i * NQ + j] = col[i]; }
    std::memset(Kinv, 0, sizeof(double) * LD * LD);
    if (m == 0) { for (int i = 0; i < NQ; i++) for (int j = 0; j < NQ; j++) Kinv[i * LD + j] = Mi[i * NQ + j]; return; }
    double JMi[12 * NQ], G[144], Lam[144];
    for (int a = 0; a < m; a++) for (int j = 0; j < NQ; j++) { double s = 0; for (int i = 0; i < NQ; i++) s += K.J[a * NQ + i] * Mi[i * NQ + j]; JMi[a * NQ + j] = s; }
    for (int a = 0; a < m; a++) for (int b = 0; b < m; b++) { double s = 0; for (int i = 0; i < NQ; i++) s += JMi[a * NQ + i] * K.J[b * NQ + i]; G[a * m + b] = s; }
    chol(G, m);
    for (int b = 0; b < m; b++) { double col[12] = {0}; col[b] = 1; chol_solve(G, m, col); for (int a = 0; a < m; a++) Lam[a * m + b] = col[a]; }
    double LJMi[12 * NQ];
    for (int a = 0; a < m; a++) for (int j = 0; j < NQ; j++) { double s = 0; for (int b = 0; b < m; b++) s += Lam[a * m + b] * JMi[b * NQ + j]; LJMi[a * NQ + j] = s; }
    for (int i = 0; i < NQ; i++) for (int j = 0; j < NQ; j++) {
        double s = Mi[i * NQ + j]; for (int a = 0; a < m; a++) s -= JMi[a * NQ + i] * LJMi[a * NQ + j];
        Kinv[i * LD + j] = s;
    }
    for (int a = 0; a < m; a++) for (int j = 0; j < NQ; j++) { Kinv[(NQ + a) * LD + j] = LJMi[a * NQ + j]; Kinv[j * LD + NQ + a] = LJMi[a * NQ + j]; }
    for (int a = 0; a < m; a++) for (int b = 0; b < m; b++) Kinv[(NQ + a) * LD + NQ + b] = -Lam[a * m + b];
}

// Column-wise derivative data at (q,v,acc) for a given foot-force vector F(12, world, per foot):
//  dtau[i][d]  = d ID(q,v,acc)_i / d x_d                       (psi_dyn;  computeRNEADerivatives)
//  dJTF[i][d]  = d (sum_feet J_f^T F_f)_i / d q_d               (psi_kin;  footForcePartialDq; zero for v cols)
//  dacc[l][.][d] = d foot classical acceleration / d x_d        (psi_kin;  footAccPartialDq / Dv)
//  dvel[l][.][d] = d foot velocity / d x_d  (only q columns are used by the reference: footVelPartialDq)
struct WbDeriv { double dtau[NQ][NX], dJTF[NQ][NX], dacc[4][3][NX], dvel[4][3][NX]; };

inline void wb_derivs(const WbParams& P, const double* q, const double* v, const double* acc, bool gravity,
                      const double* F12, const int* feet_mask, WbDeriv& D) {
    double fext[4][3];
    for (int l = 0; l < 4; l++) for (int d = 0; d < 3; d++) fext[l][d] = feet_mask[l] ? F12[3 * l + d] : 0.0;
    Dual qd[NQ], vd[NQ], ad[NQ];
    PassOut<Dual> o;
    for (int d = 0; d < NX; d++) {
        for (int i = 0; i < NQ; i++) { qd[i] = Dual(q[i]); vd[i] = Dual(v[i]); ad[i] = Dual(acc[i]); }
        if (d < NQ) qd[d].d = 1; else vd[d - NQ].d = 1;
        wb_pass<Dual>(P.psi_dyn, true, gravity, qd, vd, ad, nullptr, o);
        for (int i = 0; i < NQ; i++) D.dtau[i][d] = o.tau[i].d;
        wb_pass<Dual>(P.psi_kin, false, false, qd, vd, ad, fext, o);
        for (int i = 0; i < NQ; i++) D.dJTF[i][d] = -o.tau[i].d;  // tau = -J^T F
        for (int l = 0; l < 4; l++) {
            D.dacc[l][0][d] = o.foot_acc[l].x.d; D.dacc[l][1][d] = o.foot_acc[l].y.d; D.dacc[l][2][d] = o.foot_acc[l].z.d;
            D.dvel[l][0][d] = o.foot_vel[l].x.d; D.dvel[l][1][d] = o.foot_vel[l].y.d; D.dvel[l][2][d] = o.foot_vel[l].z.d;
        }
    }
}

// Continuous-time partials (WBM.cpp:108-139, 459-505).  Outputs column-major like Eigen:
//  Ac 36x36, Bc 36x12, C 12x36, D 12x12.  Also returns qdd, grf of the point.
inline void wb_partials_ct(const WbParams& P, const double* x, const double* u, const int* contact,
                           double* Ac, double* Bc, double* C, double* Dm, double* qdd_out, double* grf_out) {
    const double* q = x; const double* v = x + NQ;
    double tau[NQ] = {0}; for (int i = 0; i < NU; i++) tau[6 + i] = u[i];
    WbKKT K; wb_forward(P, q, v, tau, contact, K);
    if (qdd_out) std::memcpy(qdd_out, K.qdd, sizeof(K.qdd));
    if (grf_out) std::memcpy(grf_out, K.grf, sizeof(K.grf));
    static thread_local double Kinv[900];
    wb_kkt_inverse(K, Kinv);
    const int LD = 30; int m = 3 * K.nc;
    int mask[4] = {0, 0, 0, 0}; for (int i = 0; i < K.nc; i++) mask[K.feet[i]] = 1;
    static thread_local WbDeriv D;
    wb_derivs(P, q, v, K.qdd, true, K.grf, mask, D);
    // rhs columns: top = dtau - dJTF (18), bottom = da + 2 alpha dv (m) ; da_dv += 2 alpha J
    std::memset(Ac, 0, sizeof(double) * NX * NX); std::memset(Bc, 0, sizeof(double) * NX * NU);
    std::memset(C, 0, sizeof(double) * NY * NX); std::memset(Dm, 0, sizeof(double) * NY * NU);
    for (int i = 0; i < NQ; i++) Ac[i + NX * (NQ + i)] = 1.0;
    for (int d = 0; d < NX; d++) {
        double top[NQ], bot[12];
        for (int i = 0; i < NQ; i++) top[i] = D.dtau[i][d] - D.dJTF[i][d];
        for (int a = 0; a < K.nc; a++) for (int r = 0; r < 3; r++) {
            int l = K.feet[a];
            double da = D.dacc[l][r][d];
            if (d < NQ) da += 2 * P.bg_alpha * D.dvel[l][r][d];      // 2*alpha*dv_dq   (WBM.cpp:488)
            else da += 2 * P.bg_alpha * K.J[(3 * a + r) * NQ + (d - NQ)];  // 2*alpha*J  (WBM.cpp:489)
            bot[3 * a + r] = da;
        }
        for (int i = 0; i < NQ; i++) {
            double s = 0;
            for (int j = 0; j < NQ; j++) s -= Kinv[i * LD + j] * top[j];
            for (int a = 0; a < m; a++) s -= Kinv[i * LD + NQ + a] * bot[a];
            Ac[(NQ + i) + NX * d] = s;
        }
        for (int a = 0; a < m; a++) {
            double s = 0;
            for (int j = 0; j < NQ; j++) s += Kinv[(NQ + a) * LD + j] * top[j];
            for (int b = 0; b < m; b++) s += Kinv[(NQ + a) * LD + NQ + b] * bot[b];
            int row = 3 * K.feet[a / 3] + a % 3;
            C[row + NY * d] = s;
        }
    }
    for (int j = 0; j < NU; j++) {
        for (int i = 0; i < NQ; i++) Bc[(NQ + i) + NX * j] = Kinv[i * LD + 6 + j];
        for (int a = 0; a < m; a++) { int row = 3 * K.feet[a / 3] + a % 3; Dm[row + NY * j] = -Kinv[(NQ + a) * LD + 6 + j]; }
    }
}

// Discrete dynamics (WBM.cpp:17-32): forward Euler.
inline void wb_dynamics(const WbParams& P, const double* x, const double* u, const int* contact, double dt,
                        double* xnext, double* y) {
    const double* q = x; const double* v = x + NQ;
    double tau[NQ] = {0}; for (int i = 0; i < NU; i++) tau[6 + i] = u[i];
    WbKKT K; wb_forward(P, q, v, tau, contact, K);
    for (int i = 0; i < NQ; i++) { xnext[i] = q[i] + v[i] * dt; xnext[NQ + i] = v[i] + K.qdd[i] * dt; }
    std::memcpy(y, K.grf, sizeof(K.grf));
}
inline void wb_dynamics_partial(const WbParams& P, const double* x, const double* u, const int* contact, double dt,
                                double* A, double* B, double* C, double* D) {
    wb_partials_ct(P, x, u, contact, A, B, C, D, nullptr, nullptr);
    for (int i = 0; i < NX * NX; i++) A[i] *= dt;
    for (int i = 0; i < NX; i++) A[i + NX * i] += 1.0;
    for (int i = 0; i < NX * NU; i++) B[i] *= dt;   // C, D not scaled (quirk ix)
}

// Impact (WBM.cpp:178-206, 427-456): feet with contact 0->1.  impulse12 is the reference's
// mis-sliced `impulse` member (quirk v, WBM.cpp:454), returned for the derivative code.
inline void wb_impact_core(const WbParams& P, const double* x, const int* cur, const int* nxt, WbKKT& K,
                           double* vpost, double* impulse12, double* lam_true) {
    int st[4]; for (int l = 0; l < 4; l++) st[l] = (cur[l] == 0 && nxt[l] == 1) ? 1 : 0;
    const double* q = x; const double* v = x + NQ;
    set_contacts(K, st);
    double z[NQ] = {0};
    wb_terms(P, q, z, K);  // M and J only depend on q
    int m = 3 * K.nc;
    double L[NQ * NQ]; std::memcpy(L, K.M, sizeof(L)); chol(L, NQ);
    double MiJt[NQ * 12], G[144], rhs[12];
    for (int c = 0; c < m; c++) { double col[NQ]; for (int i = 0; i < NQ; i++) col[i] = K.J[c * NQ + i]; chol_solve(L, NQ, col); for (int i = 0; i < NQ; i++) MiJt[i * 12 + c] = col[i]; }
    for (int a = 0; a < m; a++) for (int b = 0; b < m; b++) { double s = 0; for (int i = 0; i < NQ; i++) s += K.J[a * NQ + i] * MiJt[i * 12 + b]; G[a * m + b] = s; }
    if (m > 0) chol(G, m);
    for (int a = 0; a < m; a++) { double s = 0; for (int i = 0; i < NQ; i++) s += K.J[a * NQ + i] * v[i]; rhs[a] = -s; }
    if (m > 0) chol_solve(G, m, rhs);
    for (int i = 0; i < NQ; i++) { double s = v[i]; for (int a = 0; a < m; a++) s += MiJt[i * 12 + a] * rhs[a]; vpost[i] = s; }
    std::memset(impulse12, 0, sizeof(double) * 12);
    double pad[16] = {0}; for (int a = 0; a < m; a++) pad[a] = rhs[a];
    for (int i = 0; i < K.nc; i++) for (int d = 0; d < 3; d++) impulse12[3 * K.feet[i] + d] = pad[(P.impulse_quirk ? i : 3 * i) + d];  // reference: offset i, not 3i
    if (lam_true) { std::memset(lam_true, 0, sizeof(double) * 12); for (int a = 0; a < m; a++) lam_true[a] = rhs[a]; }
}
inline void wb_impact(const WbParams& P, const double* x, const int* cur, const int* nxt, double* xnext) {
    WbKKT K; double vpost[NQ], imp[12];
    wb_impact_core(P, x, cur, nxt, K, vpost, imp, nullptr);
    for (int i = 0; i < NQ; i++) { xnext[i] = x[i]; xnext[NQ + i] = vpost[i]; }
}
// Px 36x36 column-major (WBM.cpp:225-254, 508-543)
inline void wb_impact_partial(const WbParams& P, const double* x, const int* cur, const int* nxt, double* Px) {
    WbKKT K; double vpost[NQ], imp[12];
    wb_impact_core(P, x, cur, nxt, K, vpost, imp, nullptr);
    const double* q = x; const double* v = x + NQ;
    static thread_local double Kinv[900];
    wb_kkt_inverse(K, Kinv);
    const int LD = 30; int m = 3 * K.nc;
    int mask[4] = {0, 0, 0, 0}; for (int i = 0; i < K.nc; i++) mask[K.feet[i]] = 1;
    double dv[NQ], z[NQ] = {0}; for (int i = 0; i < NQ; i++) dv[i] = vpost[i] - v[i];
    static thread_local WbDeriv D, D2;
    wb_derivs(P, q, z, dv, false, imp, mask, D);      // d(M dv)/dq (gravity removed) and d(J^T imp)/dq
    wb_derivs(P, q, vpost, z, false, imp, mask, D2);  // d(J vpost)/dq via dvel
    std::memset(Px, 0, sizeof(double) * NX * NX);
    for (int i = 0; i < NQ; i++) Px[i + NX * i] = 1.0;
    for (int d = 0; d < NQ; d++) {
        for (int i = 0; i < NQ; i++) {
            double s = 0;
            for (int j = 0; j < NQ; j++) s -= Kinv[i * LD + j] * (D.dtau[j][d] - D.dJTF[j][d]);
            for (int a = 0; a < m; a++) s -= Kinv[i * LD + NQ + a] * D2.dvel[K.feet[a / 3]][a % 3][d];
            Px[(NQ + i) + NX * d] = s;
        }
    }
    for (int d = 0; d < NQ; d++) for (int i = 0; i < NQ; i++) {
        double s = 0; for (int j = 0; j < NQ; j++) s += Kinv[i * LD + j] * K.M[j * NQ + d];
        Px[(NQ + i) + NX * (NQ + d)] = s;
    }
}

// Foot kinematics used by costs/constraints (WBM.cpp:260-364, 565-611):
//  pos, vel, J (3x18 each, psi_dyn = Pinocchio) and Jv = d(vel)/dq (psi_kin = CasADi footVelPartialDq)
struct WbFootKin { double pos[4][3], vel[4][3], J[4][3][NQ], Jv[4][3][NQ]; };
inline void wb_foot_kin(const WbParams& P, const double* x, WbFootKin& F, bool need_J = true) {
    const double* q = x; const double* v = x + NQ;
    PassOut<double> o; double z[NQ] = {0}, e[NQ];
    wb_pass<double>(P.psi_dyn, false, false, q, v, z, nullptr, o);
    for (int l = 0; l < 4; l++) {
        F.pos[l][0] = o.foot_pos[l].x; F.pos[l][1] = o.foot_pos[l].y; F.pos[l][2] = o.foot_pos[l].z;
        F.vel[l][0] = o.foot_vel[l].x; F.vel[l][1] = o.foot_vel[l].y; F.vel[l][2] = o.foot_vel[l].z;
    }
    if (!need_J) return;
    for (int j = 0; j < NQ; j++) {
        std::memset(e, 0, sizeof(e)); e[j] = 1;
        wb_pass<double>(P.psi_dyn, false, false, q, e, z, nullptr, o);
        for (int l = 0; l < 4; l++) { F.J[l][0][j] = o.foot_vel[l].x; F.J[l][1][j] = o.foot_vel[l].y; F.J[l][2][j] = o.foot_vel[l].z; }
    }
    Dual qd[NQ], vd[NQ], ad[NQ]; PassOut<Dual> od;
    for (int d = 0; d < NQ; d++) {
        for (int i = 0; i < NQ; i++) { qd[i] = Dual(q[i]); vd[i] = Dual(v[i]); ad[i] = Dual(0.0); }
        qd[d].d = 1;
        wb_pass<Dual>(P.psi_kin, false, false, qd, vd, ad, nullptr, od);
        for (int l = 0; l < 4; l++) { F.Jv[l][0][d] = od.foot_vel[l].x.d; F.Jv[l][1][d] = od.foot_vel[l].y.d; F.Jv[l][2][d] = od.foot_vel[l].z.d; }
    }
}

}  // namespace orc
