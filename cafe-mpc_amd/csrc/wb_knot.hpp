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
//
// LDS budget: the rollout program uses WbCore only (16 KB; two waves per SIMD at 254 registers); the LQ program (two waves per
// knot) adds WbDeriv: the tangent columns T[18][54], the result rows R[18][36] / a dense 36x36 staging tile, the staging of C, B, D,
// the UNDAMPED Schur factor for the derivative columns and the barrier derivative tables (40.5 KB per knot).  Neither program forms
// M^-1 or the KKT inverse: the factors of the forward solve (M = L L^T, X = L^-1 Jc^T, X^T X = L_G L_G^T) serve every right-hand side.
// A rollout knot also runs as a PROBE of a line-search candidate (wr = false): nothing but the merit partials leaves the wave.
#pragma once
#include "hs_types.hpp"
#include "wb_model.hpp"
#include "hs_mfma.hpp"

namespace hs {

struct WbCore {
    double x[36], xb[36], u[12], acc[18], tau[18], fext[12], cs[18], sn[18];
    double M[18 * 18];                 // mass matrix (rollout: overwritten by its Cholesky factor)
    double h[18], Jall[12 * 18], Jdv[12], fpos[12], fvel[12];
    double JX[432];                    // Jc (12x18 compact active Jacobian) | Xm (18x12: L^-1 Jc^T or Minv Jc^T); also K / C staging
    double gam[12];
    double GG[288 + 2 * MAXG];         // G (144) | LG (144) | gval (MAXG) | bar (MAXG); the LQ program reuses it for the foot acc / vel tangents
    HD double* Jc() { return JX; }
    HD double* Xm() { return JX + 216; }
    HD double* G() { return GG; }
    HD double* LG() { return GG + 144; }
    HD double* gval() { return GG + 288; }
    HD double* bar() { return GG + 288 + MAXG; }
    HD double* dacc() { return GG; }          // LQ program: [3f+r][18] tangents of the foot accelerations
    HD double* dvel() { return GG + 216; }    // LQ program: [3f+r][18] tangents of the foot velocities (= footVelPartialDq)
    double a0[18], rhs[12], lam[12], qdd[18], grf[12], tmp[64], red[64], rdM[18], rdG[12];
    double wq[48];                     // running-cost weights q (36) | r (12) of the phase, fetched with the knot's other global reads
    double xnext[36];                  // single-shooting chain: the state handed from one knot to the next inside the wave
    unsigned long long tstamp;
};
struct WbDeriv {   // LQ-only LDS; several short-lived matrices share storage (see accessors)
    double JPW[792];                   // column solve: staging of C (432) | B rows 18..35 (216) | D (144)
    double W[18 * 54 + 18 * 36];       // T[18][54] tangent columns + R[18][36] result rows ; finally a dense 36x36 staging tile for lxx / Phixx
    double wp[12], wv[12], ep[12], ev[12];
    double LGs[144], rdGs[12];         // Schur factor of the contact solve, kept while GG holds the foot tangents
    double bdt[2 * MAXG];              // barrier derivative tables
    unsigned long long tstamp1;        // (diagnostic stamps of the second wave)
    HD double* stC() { return JPW; }
    HD double* stB() { return JPW + 432; }
    HD double* stD() { return JPW + 648; }
    HD double* bd() { return bdt; }
    HD double* bdd() { return bdt + MAXG; }
};
constexpr int WT = 54, WR0 = 18 * 54;   // T(i,lane) = W[i*WT + lane] ; R(i,d) = W[WR0 + i*36 + d]
struct WbLqLds { WbCore c; WbDeriv d; };
// the workgroup whose phases the diagnostic stamps time: one from the middle of a launch (the first workgroups start on cold caches)
#ifndef HS_PROF_BLOCK
#define HS_PROF_BLOCK 100007
#endif
#if defined(ROLL_PROF_EXTERNAL)
// (the including file defines RL_STAMP / RL_STAMP0 / KK_STAMP itself: tools/count_instructions.sh places assembler marks there)
#elif defined(ROLL_PROF) && !defined(HS_HOST_EMU)
#define RL_STAMP(i) { if (blockIdx.x == HS_PROF_BLOCK && threadIdx.x == 0) { unsigned long long t_ = clock64(); atomicAdd(&g_lq_prof[i], t_ - L.tstamp); L.tstamp = t_; } }
#define RL_STAMP0() { if (blockIdx.x == HS_PROF_BLOCK && threadIdx.x == 0) L.tstamp = clock64(); }
#else
#define RL_STAMP(i)
#define RL_STAMP0()
#endif
#if (defined(LQ_PROF) || defined(ROLL_PROF)) && !defined(HS_HOST_EMU)
__device__ unsigned long long g_lq_prof[16];
#endif
#if defined(LQ_PROF_EXTERNAL)
#elif defined(LQ_PROF) && !defined(HS_HOST_EMU)
#define LQ_STAMP(i) { if (blockIdx.x == HS_PROF_BLOCK && threadIdx.x == 0) { unsigned long long t_ = clock64(); atomicAdd(&g_lq_prof[i], t_ - L.tstamp); L.tstamp = t_; } }
#define LQ_STAMP0() { if (blockIdx.x == HS_PROF_BLOCK && threadIdx.x == 0) L.tstamp = clock64(); }
#elif defined(ROLL_PROF) && !defined(HS_HOST_EMU)
// rollout profile: the stamps inside the contact solve (7..10) split its share of the knot
#define LQ_STAMP(i) { if ((i) >= 7 && (i) <= 10) RL_STAMP(i) }
#define LQ_STAMP0()
#else
#define LQ_STAMP(i)
#define LQ_STAMP0()
#endif
// the second wave of the two-wave LQ knot times its own spans (thread 64)
#if defined(LQ_PROF) && !defined(HS_HOST_EMU) && !defined(LQ_PROF_EXTERNAL)
#define WB_W1_STAMP_BEGIN() if (blockIdx.x == HS_PROF_BLOCK && threadIdx.x == 64) D.tstamp1 = clock64();
#define WB_W1_STAMP_END(i) if (blockIdx.x == HS_PROF_BLOCK && threadIdx.x == 64) atomicAdd(&g_lq_prof[i], clock64() - D.tstamp1);
#else
#define WB_W1_STAMP_BEGIN()
#define WB_W1_STAMP_END(i)
#endif
// LQ-program aliases into WbCore: Cst = [Jc|Xm] (432), dacc = G.., dvel = G+216 (G,LG,gval,bar = 432)

// ---- wave-cooperative dense helpers (row-major in LDS), COMPILE-TIME sizes ------------------------------
// Every inner loop is fully unrolled and every private vector lives in registers: the serial recurrences of a
// factorisation then depend only on FMA latency, not on an LDS round trip per step.
// Cholesky A = L L^T (lower), one phase per column; Lo may alias A.  rd[i] = 1 / L[i][i].
template <int NT, int N, int LD>
HD void chol_s(const double* A, double* Lo, double* rd, double diag_add) {
    _Pragma("unroll")
    for (int j = 0; j < N; j++) {
        HS_PHASE(NT, if (tid >= j && tid < N) {
            double sj = A[j * LD + j] + diag_add, st = A[tid * LD + j];
            _Pragma("unroll")
            for (int k = 0; k < j; k++) { const double ljk = Lo[j * LD + k]; sj -= ljk * ljk; st -= Lo[tid * LD + k] * ljk; }
            const double r = hs_rsqrt(sj);
            // the diagonal of Lo is never written (it may alias A, whose pivot other lanes still read) nor read: rd[] carries it
            if (tid == j) rd[j] = r; else Lo[tid * LD + j] = st * r;
        })
    }
}
template <int NT, int N, int LD, class NZ = DenseNZ>
HD void chol_f(const double* A, double* Lo, double* rd, double diag_add) {
#ifdef HS_HOST_EMU
    chol_s<NT, N, LD>(A, Lo, rd, diag_add);      // (structural zeros come out as exact zeros of the dense recurrence)
#else
    HS_PHASE(NT, chol_r<N, LD, NZ>(A, LD, 1, Lo, rd, diag_add, tid);)
#endif
}
// in-place factorisation of a 12x12 matrix by ONE wave (tid = lane): registers + lane broadcasts on the GPU, plain loops in the emulator
#ifdef HS_HOST_EMU
#define WB_CHOL_G(G_, rd_, tid_) { if ((tid_) == 0) { for (int j_ = 0; j_ < 12; j_++) { double sj_ = (G_)[j_ * 12 + j_]; for (int k_ = 0; k_ < j_; k_++) sj_ -= (G_)[j_ * 12 + k_] * (G_)[j_ * 12 + k_]; \
    const double r_ = hs_rsqrt(sj_); (rd_)[j_] = r_; for (int i_ = j_ + 1; i_ < 12; i_++) { double st_ = (G_)[i_ * 12 + j_]; for (int k_ = 0; k_ < j_; k_++) st_ -= (G_)[i_ * 12 + k_] * (G_)[j_ * 12 + k_]; (G_)[i_ * 12 + j_] = st_ * r_; } } } }
#else
#define WB_CHOL_G(G_, rd_, tid_) chol_r<12, 12>((G_), 12, 1, (G_), (rd_), 0.0, (tid_))
#endif
#ifndef HS_SOLVE_CBAR
#define HS_SOLVE_CBAR 100    // rows of a triangular solve between two scheduling fences (100: none; 1 keeps every row's loads behind the previous row)
#endif
// x = L^-1 b (forward) in registers; b/x are private arrays.  The factor row of step i+1 is fetched from LDS (a broadcast read per
// entry) into registers BEFORE the dependent multiply-add chain of row i runs: a single wave has no other wave to hide the LDS
// latency behind, so the loads are batched and pipelined by hand (the fence keeps the compiler from sinking them back).
template <int N, int LD, class NZ = DenseNZ> HD void fwd_s(const double* Lo, const double* rd, double* x) {
    double lr[2][N];
    _Pragma("unroll")
    for (int i = 0; i < N; i++) {
        if (i + 1 < N) { _Pragma("unroll") for (int k = 0; k <= i; k++) if (NZ::nz(i + 1, k)) lr[(i + 1) & 1][k] = Lo[(i + 1) * LD + k]; }
        HS_CBAR();
        double s = x[i];
        _Pragma("unroll")
        for (int k = 0; k < i; k++) if (NZ::nz(i, k)) s -= lr[i & 1][k] * x[k];
        x[i] = s * rd[i];
    }
}
// x = L^-T b (backward) in registers
template <int N, int LD, class NZ = DenseNZ> HD void bwd_s(const double* Lo, const double* rd, double* x) {
    double lr[2][N];
    _Pragma("unroll")
    for (int i = N - 1; i >= 0; i--) {
        if (i > 0) { _Pragma("unroll") for (int k = i; k < N; k++) if (NZ::nz(k, i - 1)) lr[(i - 1) & 1][k] = Lo[k * LD + (i - 1)]; }
        HS_CBAR();
        double s = x[i];
        _Pragma("unroll")
        for (int k = i + 1; k < N; k++) if (NZ::nz(k, i)) s -= lr[i & 1][k] * x[k];
        x[i] = s * rd[i];
    }
}

// The mass matrix is factored LEGS FIRST: position i < 12 of the permuted order is leg joint 6 + i, positions 12..17 are the floating base.
// A leg joint only couples with its own leg and the base, so the factor has no entries between different legs (and none fills in): 99
// instead of 153 multiplier entries in the factorisation and in every triangular solve.  Vectors between a forward and a backward solve
// (and the rows of X = L^-1 Jc^T) live in the permuted order; wb_pi / wb_pj translate at the loads and stores.
HD constexpr int wb_pi(int i) { return i < 12 ? i + 6 : i - 12; }     // permuted position -> joint
HD constexpr int wb_pj(int j) { return j < 6 ? j + 12 : j - 6; }      // joint -> permuted position
struct WbNZ { static constexpr bool nz(int i, int k) { return i >= 12 || (k / 3 == i / 3); } };

HD LaneCfg lane_cfg(const ModelDev& md, bool kin, double mscale, double grav, double fscale, double vscale, double ascale, int aunit, int tq, int tv) {
    LaneCfg c;
    c.cpsi = kin ? md.cpsi_kin : md.cpsi_dyn; c.spsi = kin ? md.spsi_kin : md.spsi_dyn;
    c.mscale = mscale; c.grav = grav; c.fscale = fscale; c.vscale = vscale; c.ascale = ascale; c.aunit = aunit; c.tq = tq; c.tv = tv;
    return c;
}

// sinks: where a pass stores its outputs
struct PSink {   // value pass: lanes 0..17 -> column `lane` of M and of all foot Jacobians; lane 18 -> bias terms
    WbCore* L; int lane; int task;
    HD void base(const V3<double>& f, const V3<double>& n) const {   // per-leg task: partial base wrench (GG is free while the terms are formed)
        double* p = L->GG + 6 * task; p[0] = f.x; p[1] = f.y; p[2] = f.z; p[3] = n.x; p[4] = n.y; p[5] = n.z;
    }
    HD void tau(int i, double v) const { if (lane < 18) L->M[wb_pj(i) * 18 + wb_pj(lane)] = v; else L->h[i] = v; }     // M in the legs-first order
    HD void foot(int f, const V3<double>& p, const V3<double>& v, const V3<double>& a) const {
        if (lane < 18) { L->Jall[(3 * f) * 18 + lane] = a.x; L->Jall[(3 * f + 1) * 18 + lane] = a.y; L->Jall[(3 * f + 2) * 18 + lane] = a.z; }
        else {
            L->Jdv[3 * f] = a.x; L->Jdv[3 * f + 1] = a.y; L->Jdv[3 * f + 2] = a.z;
            L->fpos[3 * f] = p.x; L->fpos[3 * f + 1] = p.y; L->fpos[3 * f + 2] = p.z;
            L->fvel[3 * f] = v.x; L->fvel[3 * f + 1] = v.y; L->fvel[3 * f + 2] = v.z;
        }
    }
};
struct DSink {   // tangent pass: d tau -> W[i][lane]; kinematic lanes (>=36) also store foot acc / vel tangents
    WbDeriv* D; WbCore* C; int lane; int task;
    HD void base(const V3<Dual>& f, const V3<Dual>& n) const {   // per-leg task of a base seed: partial base wrench, value and tangent (JPW is free during the pass)
        double* p = D->JPW + 12 * task;
        p[0] = f.x.v; p[1] = f.x.d; p[2] = f.y.v; p[3] = f.y.d; p[4] = f.z.v; p[5] = f.z.d;
        p[6] = n.x.v; p[7] = n.x.d; p[8] = n.y.v; p[9] = n.y.d; p[10] = n.z.v; p[11] = n.z.d;
    }
    HD void tau(int i, const Dual& v) const { D->W[i * WT + lane] = v.d; }
    HD void foot(int f, const V3<Dual>&, const V3<Dual>& v, const V3<Dual>& a) const {
        if (lane >= 36) {
            const int j = lane - 36;
            C->dacc()[(3 * f) * 18 + j] = a.x.d; C->dacc()[(3 * f + 1) * 18 + j] = a.y.d; C->dacc()[(3 * f + 2) * 18 + j] = a.z.d;
            C->dvel()[(3 * f) * 18 + j] = v.x.d; C->dvel()[(3 * f + 1) * 18 + j] = v.y.d; C->dvel()[(3 * f + 2) * 18 + j] = v.z.d;
        }
    }
};

// trig table of the knot state (18 lanes in parallel)
template <int NT> HD void wb_trig(WbCore& L) { HS_PHASE(NT, if (tid < 18) { double sv, cv; sincos_(L.x[tid], sv, cv); L.cs[tid] = cv; L.sn[tid] = sv; }) }

// Phase: M, h, all-foot Jacobians, Jdot*v, foot pos/vel at L.x (psi_dyn), as 40 per-leg TASKS of about a third of a pass each:
//   lanes 0..11  : column 6+lane (a leg joint) — only its own leg moves, one task
//   lanes 12..35 : base column (lane-12)/4, leg (lane-12)%4 ; lanes 36..39: bias terms (h, Jdot v, foot pos/vel), leg lane-36
// A task walks the base joints forward, ONE leg down and up, and leaves the base wrench of that leg (plus the trunk's own
// inertial force in the leg-0 task) as a partial sum; 19 lanes then add the partials in leg order and run the base joints backward.
template <int NT>
HD void wb_terms(WbCore& L, const ModelDev& md, bool need_cols) {
    wb_trig<NT>(L);
    HS_PHASE(NT, if (tid < 40 && (need_cols || tid >= 36)) {
        const int col = (tid < 12) ? 6 + tid : (tid < 36) ? (tid - 12) / 4 : 18;
        const int leg = (tid < 12) ? tid / 3 : (tid < 36) ? (tid - 12) % 4 : tid - 36;
        LaneCfg c = (col < 18) ? lane_cfg(md, false, 1.0, 0.0, 0.0, 0.0, 0.0, col, -1, -1)
                               : lane_cfg(md, false, 1.0, GRAV, 0.0, 1.0, 0.0, -1, -1, -1);
        c.l0 = leg; c.l1 = leg + 1; c.body = (tid >= 12) && (leg == 0); c.partial = true;
        PSink sk{&L, col, tid};
        wb_pass<double>(c, L.x, L.x + 18, L.acc, L.cs, L.sn, L.fext, sk);
        if (tid < 12) {    // a leg-joint column is zero on the other legs' rows and feet
            for (int f = 0; f < 4; f++) if (f != leg) {
                for (int r = 0; r < 3; r++) { L.M[(3 * f + r) * 18 + wb_pj(col)] = 0.0; L.Jall[(3 * f + r) * 18 + col] = 0.0; }
            }
        }
    })
    HS_PHASE(NT, if (tid < 19 && (need_cols || tid == 18)) {
        const int t0 = (tid < 6) ? 12 + 4 * tid : (tid < 18) ? tid - 6 : 36, nt = (tid < 6 || tid == 18) ? 4 : 1;
        V3<double> fb = {0.0, 0.0, 0.0}, nb = fb;
        for (int t = 0; t < nt; t++) { const double* p = L.GG + 6 * (t0 + t); fb = fb + V3<double>{p[0], p[1], p[2]}; nb = nb + V3<double>{p[3], p[4], p[5]}; }
        PSink sk{&L, tid, 0};
        const double c3 = L.cs[3], s3 = L.sn[3], c4 = L.cs[4], s4 = L.sn[4], c5 = L.cs[5], s5 = L.sn[5];
        sk.tau(5, nb.x);
        V3<double> f = rot<0, double>(c5, s5, fb), n = rot<0, double>(c5, s5, nb);
        sk.tau(4, n.y);
        f = rot<1, double>(c4, s4, f); n = rot<1, double>(c4, s4, n);
        sk.tau(3, n.z);
        f = rot<2, double>(c3, s3, f);
        sk.tau(0, f.x); sk.tau(1, f.y); sk.tau(2, f.z);
    })
}

// A per-lane index into a small descriptor array would be a VECTOR load from the descriptor in the middle of a knot (an exposed
// round trip for the single wave); the elements are fetched as scalars instead and picked by selects.
struct Feet4 { int f0, f1, f2, f3; HD int operator[](int i) const { return i == 0 ? f0 : i == 1 ? f1 : i == 2 ? f2 : f3; }
    // per-LANE slot i (0..3): shifts on one packed scalar - the compiler turns a select chain over the four fields back into branches
    HD int pick(int i) const { return (((f0 & 3) | (f1 & 3) << 2 | (f2 & 3) << 4 | (f3 & 3) << 6) >> (2 * i)) & 3; } };
HD Feet4 feet_of(PhaseC& P) { int f0 = P.feet[0], f1 = P.feet[1], f2 = P.feet[2], f3 = P.feet[3]; HS_PIN_S(f0); HS_PIN_S(f1); HS_PIN_S(f2); HS_PIN_S(f3); return Feet4{f0, f1, f2, f3}; }     // pinned: the compiler would otherwise turn the selects back into ONE indexed vector load
template <class T> HD double pick3(const T& w, int a) { const double w0 = w[0], w1 = w[1], w2 = w[2]; return a == 0 ? w0 : a == 1 ? w1 : w2; }

// compact active Jacobian + drift, PADDED to 12 rows (rows >= 3*nc are zero) so that every later loop has a
// compile-time trip count (mode 0: gam = Jdot v + 2 alpha J v ; mode 1: 0)
template <int NT>
HD void wb_select(WbCore& L, int nc, const Feet4& feet, int mode, double alpha) {
    HS_PHASE(NT,
        for (int e = tid; e < 216; e += NT) {
            const int a = e / 18, j = e % 18; const bool act = a < 3 * nc;
            L.Jc()[e] = act ? L.Jall[(3 * feet.pick(act ? a / 3 : 0) + a % 3) * 18 + wb_pi(j)] : 0.0;      // columns in the legs-first order
        }
        for (int e = tid; e < 144; e += NT) L.G()[e] = (e % 13 == 0) ? 1.0 : 0.0;     // identity: the padding of the Gram matrix beyond the active block
        if (tid < 12) {
            const bool act = tid < 3 * nc;
            const int f = act ? feet.pick(tid / 3) : 0, r = tid % 3;
            L.gam[tid] = (act && mode == 0) ? (L.Jdv[3 * f + r] + 2.0 * alpha * L.fvel[3 * f + r]) : 0.0;
            L.grf[tid] = 0.0;
        })
}
// G = X^T X, the leading NM x NM block of the 12 x 12 matrix (rows / columns >= m keep the identity wb_select put there), dealt over the wave ; X = Xm (18x12)
template <int NM> HD void wb_gram(WbCore& L, int m, int tid, int nt) {
    for (int e = tid; e < NM * NM; e += nt) {
        const int a = e / NM, b = e % NM;
        double s = 0;
        _Pragma("unroll")
        for (int i = 0; i < 18; i++) s += L.Xm()[i * 12 + a] * L.Xm()[i * 12 + b];
        if (a < m && b < m) L.G()[a * 12 + b] = s;
    }
}

// Second half of the contact solve on the leading NM x NM block of the Gram matrix (NM = 6: at most two feet in contact, half the columns
// of every phase below): G = X^T X, its factor, lam, qdd.
template <int NT, int NM>
HD void wb_kkt_tail(WbCore& L, int m, const Feet4& feet, int mode) {
    HS_PHASE(NT, wb_gram<NM>(L, m, tid, NT);
        if (tid >= 48 && tid < 60) {
            const int a = tid - 48; double s = 0;
            if (mode == 0) { _Pragma("unroll") for (int i = 0; i < 18; i++) s += L.Xm()[i * 12 + a] * L.a0[i]; }
            else { _Pragma("unroll") for (int i = 0; i < 18; i++) s += L.Jc()[a * 18 + i] * L.x[18 + wb_pi(i)]; }
            L.rhs[a] = (a < m) ? (-s - L.gam[a]) : 0.0;
        })
    LQ_STAMP(9)
    chol_f<NT, NM, 12>(L.G(), L.LG(), L.rdG, (mode == 0) ? 1e-12 : 0.0);
    LQ_STAMP(10)
#ifdef HS_HOST_EMU
    HS_PHASE(NT, if (tid == 0) {             // lam = G^-1 rhs ; then z = y + X lam ; back substitution L^T qdd = z
        double lam[12];
        _Pragma("unroll")
        for (int i = 0; i < 12; i++) lam[i] = L.rhs[i];
        fwd_s<NM, 12>(L.LG(), L.rdG, lam); bwd_s<NM, 12>(L.LG(), L.rdG, lam);
        double z[18];
        _Pragma("unroll")
        for (int i = 0; i < 18; i++) { double s = L.a0[i]; _Pragma("unroll") for (int a = 0; a < NM; a++) s += L.Xm()[i * 12 + a] * lam[a]; z[i] = s; }
        bwd_s<18, 18, WbNZ>(L.M, L.rdM, z);
        _Pragma("unroll")
        for (int i = 0; i < 18; i++) L.qdd[wb_pi(i)] = z[i] + ((mode == 1) ? L.x[18 + wb_pi(i)] : 0.0);
        _Pragma("unroll")
        for (int a = 0; a < 12; a++) { L.lam[a] = lam[a]; if (a < m) L.grf[3 * feet[a / 3] + a % 3] = lam[a]; }
    })
#else
    // lam = G^-1 rhs ; z = y + X lam ; back substitution L^T qdd = z — across the lanes of the wave: lane i owns entry i, the
    // entry just finished travels by lane broadcast, the factor entries each lane needs are preloaded from LDS
    HS_PHASE(NT, {
        const int i12 = tid < NM ? tid : NM - 1, i18 = tid < 18 ? tid : 17;
        double lgr[NM], lgc[NM], xr[NM], lmc[6], lleg[3];
        _Pragma("unroll") for (int k = 0; k < NM; k++) { lgr[k] = L.LG()[i12 * 12 + k]; lgc[k] = L.LG()[k * 12 + i12]; xr[k] = L.Xm()[i18 * 12 + k]; }
        const int lg = i18 < 12 ? i18 / 3 : 3, jl = i18 - 3 * lg;            // leg and joint of the lane's row (base rows: lg = 3 keeps the addresses valid, jl >= 3)
        _Pragma("unroll") for (int k = 0; k < 6; k++) lmc[k] = L.M[(12 + k) * 18 + i18];            // column i18 of the factor: base rows ...
        _Pragma("unroll") for (int j = 0; j < 3; j++) lleg[j] = L.M[(3 * lg + j) * 18 + i18];        // ... and the rows of the lane's own leg
        const double rg = L.rdG[i12], rm = L.rdM[i18];
        double v = L.rhs[i12];
        _Pragma("unroll") for (int k = 0; k < NM; k++) { const double xk = hs_readlane(v * rg, k); v = (tid == k) ? xk : ((tid > k) ? v - lgr[k] * xk : v); }
        _Pragma("unroll") for (int k = NM - 1; k >= 0; k--) { const double xk = hs_readlane(v * rg, k); v = (tid == k) ? xk : ((tid < k) ? v - lgc[k] * xk : v); }
        double z = L.a0[i18];
        _Pragma("unroll") for (int a = 0; a < NM; a++) z += xr[a] * hs_readlane(v, a);
        // L^T qdd = z: the base rows one by one, then the third, second, first joint of EVERY leg at once (the factor has no entries between
        // legs; the finished entry travels inside the leg by ds_bpermute) - 9 dependent steps instead of 18
        _Pragma("unroll") for (int k = 17; k >= 12; k--) { const double xk = hs_readlane(z * rm, k); z = (tid == k) ? xk : ((tid < k) ? z - lmc[k - 12] * xk : z); }
        _Pragma("unroll") for (int j = 2; j >= 0; j--) { const double xk = hs_bperm(z * rm, 3 * lg + j); z = (tid < 12 && jl == j) ? xk : ((tid < 12 && jl < j) ? z - lleg[j] * xk : z); }
        if (tid < 18) L.qdd[wb_pi(tid)] = z + ((mode == 1) ? L.x[18 + wb_pi(tid)] : 0.0);
        if (tid < 12) { const double lm = tid < NM ? v : 0.0; L.lam[tid] = lm; if (tid < m) L.grf[3 * feet.pick(tid / 3) + tid % 3] = lm; }
    })
#endif
}

// Contact solve WITHOUT forming M^-1 (rollout): M = L L^T in place, X = L^-1 Jc^T, G = X^T X (+damping),
//   mode 0 (Pinocchio forwardDynamics): y = L^-1 (tau - h), lam = G^-1 (-X^T y - gam), qdd = L^-T (y + X lam)
//   mode 1 (Pinocchio impulseDynamics): lam = G^-1 (-Jc v), v+ = v + L^-T X lam
template <int NT>
HD void wb_kkt_direct(WbCore& L, int nc, const Feet4& feet, int mode, double alpha) {
    const int m = 3 * nc;
    wb_select<NT>(L, nc, feet, mode, alpha);
    LQ_STAMP(7)
#ifdef HS_HOST_EMU
    chol_f<NT, 18, 18, WbNZ>(L.M, L.M, L.rdM, 0.0);
#else
    HS_PHASE(NT, chol_wb18(L.M, L.rdM, tid);)
#endif
    LQ_STAMP(8)
    // X[:, tid] = L^-1 Jc[tid, :]^T on lanes 0..11 and y = L^-1 (tau - h) (mode 0; 0 in mode 1) on lane 12: ONE instruction stream, the right-hand
    // side and the destination picked per lane
    HS_PHASE(NT, if (tid < 13) {
        const bool isx = tid < 12;
        double* w = reinterpret_cast<double*>(&L);
        constexpr int OXM = offsetof(WbCore, JX) / 8 + 216, OA0 = offsetof(WbCore, a0) / 8;
        double x[18];
        _Pragma("unroll")
        for (int i = 0; i < 18; i++) {
            const double jc = L.Jc()[(isx ? tid : 0) * 18 + i], yv = (mode == 0) ? (L.tau[wb_pi(i)] - L.h[wb_pi(i)]) : 0.0;
            x[i] = isx ? jc : yv;
        }
        fwd_s<18, 18, WbNZ>(L.M, L.rdM, x);
        const int o0 = isx ? OXM + tid : OA0, st = isx ? 12 : 1;
        _Pragma("unroll")
        for (int i = 0; i < 18; i++) w[o0 + i * st] = x[i];
    })
    if (m <= 6) wb_kkt_tail<NT, 6>(L, m, feet, mode); else wb_kkt_tail<NT, 12>(L, m, feet, mode);      // uniform over the wave
}

// One column of the KKT-inverse products the LQ approximation needs, WITHOUT forming M^-1 or the KKT inverse (the
// reference asks Pinocchio for the 30x30 inverse, WBM.cpp:463,512, and multiplies): with M = L L^T, X = L^-1 Jc^T, G = X^T X
//   [M Jc^T; Jc 0]^-1 [top; bot] = [ L^-T (y - X nu) ; nu ],   y = L^-1 top,  nu = G^-1 (X^T y - bot)
// every lane solves its own right-hand side in registers; factors are read from LDS as broadcasts.
// in: top[18] (or y directly if top_is_y), bot[12] (entries >= m zero) ; out: top <- upper part, bot <- nu
template <int NM> HD void wb_kkt_column_mid(const WbCore& L, const WbDeriv& D, double* top, double* bot) {
    _Pragma("unroll")
    for (int a = 0; a < NM; a++) {
        double s = -bot[a];
        _Pragma("unroll")
        for (int i = 0; i < 18; i++) s += L.JX[216 + i * 12 + a] * top[i];
        bot[a] = s;
    }
    fwd_s<NM, 12>(D.LGs, D.rdGs, bot); bwd_s<NM, 12>(D.LGs, D.rdGs, bot);
    _Pragma("unroll")
    for (int i = 0; i < 18; i++) {
        double s = top[i];
        _Pragma("unroll")
        for (int a = 0; a < NM; a++) s -= L.JX[216 + i * 12 + a] * bot[a];
        top[i] = s;
    }
}
// m = 3 nc (uniform over the wave): with at most two feet in contact the constraint part runs on the leading 6 x 6 block (entries of bot
// beyond it are zero on entry and stay untouched)
HD void wb_kkt_column(const WbCore& L, const WbDeriv& D, double* top, double* bot, bool top_is_y, int m) {
    if (!top_is_y) fwd_s<18, 18, WbNZ>(L.M, L.rdM, top);
    if (m <= 6) wb_kkt_column_mid<6>(L, D, top, bot); else wb_kkt_column_mid<12>(L, D, top, bot);
    bwd_s<18, 18, WbNZ>(L.M, L.rdM, top);
}
// Schur factor for the derivative columns: Pinocchio's computeKKTContactDynamicMatrixInverse runs with damping 0 (WBM.cpp:467), unlike
// the forward solve (1e-12, WBM.cpp:411): G = X^T X is factored once more without the damping, into LDS that survives the tangent pass
// (GG is about to receive the foot tangents).  wb_kkt_direct leaves G in L.G().
template <int NT> HD void wb_keep_schur(WbCore& L, WbDeriv& D) {
    chol_f<NT, 12, 12>(L.G(), D.LGs, D.rdGs, 0.0);
}

// Tangent pass.  SEEDS (the W column / foot-tangent column a result goes to): 0..35: d ID(q,v,acc)/dx_seed (psi_dyn, gravity `grav`);
// 36..53: massless, foot forces L.fext, psi_kin, tangent on q_(seed-36): tau tangent = -d(J^T F)/dq, foot acc / vel tangents.
// Run as 84 per-leg TASKS of about a quarter of a pass each, in two rounds of the wave:
//   * a seed on a leg joint only moves its own leg: one task (other legs' rows / feet are zero, the base rows see this leg's wrench only);
//   * a seed on a base rotation or a base velocity moves every leg: four tasks that leave their base wrench (value and tangent; the
//     trunk's own inertial force rides with leg 0) as partial sums, added in leg order and walked back through the base joints afterwards;
//   * a seed on the base position moves nothing that is stored (only foot positions depend on it): zeros.
template <int NT>
HD void wb_dpass(WbCore& L, WbDeriv& D, const ModelDev& md, double grav, double vscale_dyn, double vscale_kin, double ascale_kin, bool q_only, bool chol_g = false) {
    // base seeds, q first so that q_only drops the tail: dyn q3..5 | kin q3..5 | dyn v0..5
    auto base_seed = [](int bs) { return bs < 3 ? 3 + bs : bs < 6 ? 36 + bs : 12 + bs; };
    // round A: the 60 DYNAMIC tasks (base q3..5 and v0..5 seeds x 4 legs, leg-joint q and v seeds); round B: the 24 KINEMATIC tasks
    // (base q3..5 seeds x 4 legs, leg-joint q seeds) - each round runs an instruction stream without the other kind's dead work
    auto zero_rest = [&](int seed, int leg) {
        for (int f = 0; f < 4; f++) if (f != leg) for (int r = 0; r < 3; r++) {
            D.W[(6 + 3 * f + r) * WT + seed] = 0.0;
            if (seed >= 36) { L.dacc()[(3 * f + r) * 18 + seed - 36] = 0.0; L.dvel()[(3 * f + r) * 18 + seed - 36] = 0.0; }
        }
    };
    HS_PHASE(NT,
        WB_W1_STAMP_BEGIN()
        if (tid < 60 && !(q_only && ((tid >= 12 && tid < 36) || tid >= 48))) {
            const int t = tid; int seed, leg, slot = 0; bool part = t < 36;
            if (t < 12) { seed = base_seed(t >> 2); leg = t & 3; slot = t; }
            else if (t < 36) { const int bs = 6 + ((t - 12) >> 2); seed = base_seed(bs); leg = t & 3; slot = 4 * bs + leg; }
            else { const int jl = (t - 36) % 12; seed = (t < 48 ? 6 : 24) + jl; leg = jl / 3; }
            LaneCfg c = lane_cfg(md, false, 1.0, grav, 0.0, vscale_dyn, 1.0, -1, seed < 18 ? seed : -1, seed >= 18 ? seed - 18 : -1);
            c.l0 = leg; c.l1 = leg + 1; c.body = part && leg == 0; c.partial = part;
            DSink sk{&D, &L, seed, slot};
            wb_pass<Dual, DSink, 1>(c, L.x, L.x + 18, L.acc, L.cs, L.sn, L.fext, sk);
            if (!part) zero_rest(seed, leg);
        }
        const int kt = (NT >= 128) ? tid - 64 : tid;      // a 128-thread workgroup runs the kinematic round on its second wave, next to the dynamic round
        if (kt >= 0 && kt < 24) {
            const int t = kt; int seed, leg, slot = 0; const bool part = t < 12;
            if (part) { const int bs = 3 + (t >> 2); seed = base_seed(bs); leg = t & 3; slot = 4 * bs + leg; }
            else { const int jl = t - 12; seed = 42 + jl; leg = jl / 3; }
            LaneCfg c = lane_cfg(md, true, 0.0, 0.0, 1.0, vscale_kin, ascale_kin, -1, seed - 36, -1);
            c.l0 = leg; c.l1 = leg + 1; c.body = false; c.partial = part;
            DSink sk{&D, &L, seed, slot};
            wb_pass<Dual, DSink, 2>(c, L.x, L.x + 18, L.acc, L.cs, L.sn, L.fext, sk);
            if (!part) zero_rest(seed, leg);
        }
        // base-position seeds (0..2 and 36..38): zero columns
        for (int e = tid; e < 108; e += NT) { const int sd = e / 18, i = e % 18; D.W[i * WT + (sd < 3 ? sd : 33 + sd)] = 0.0; }
        for (int e = tid; e < 36; e += NT) { const int j = e / 12, r = e % 12; L.dacc()[r * 18 + j] = 0.0; L.dvel()[r * 18 + j] = 0.0; }
        // two-wave LQ knot: wave 1 (done with the shorter round) factors the Gram matrix the cache delivered, for the column solves
        if (NT >= 128 && chol_g && tid >= 64) WB_CHOL_G(D.LGs, D.rdGs, tid - 64);
        WB_W1_STAMP_END(6))
    HS_PHASE(NT, if (tid < (q_only ? 6 : 12)) {
        const int seed = base_seed(tid), tq = seed < 18 ? seed : seed >= 36 ? seed - 36 : -1;
        V3<Dual> fb = {Dual(0.0), Dual(0.0), Dual(0.0)}, nb = fb;
        for (int l = 0; l < 4; l++) {
            const double* p = D.JPW + 12 * (4 * tid + l);
            fb = fb + V3<Dual>{Dual(p[0], p[1]), Dual(p[2], p[3]), Dual(p[4], p[5])}; nb = nb + V3<Dual>{Dual(p[6], p[7]), Dual(p[8], p[9]), Dual(p[10], p[11])};
        }
        auto SC = [&](int i, Dual& s_, Dual& c_) { const double c0 = L.cs[i], s0 = L.sn[i]; const bool sd = (tq == i); s_ = Dual(s0, sd ? c0 : 0.0); c_ = Dual(c0, sd ? -s0 : 0.0); };
        Dual c3, s3, c4, s4, c5, s5;
        SC(3, s3, c3); SC(4, s4, c4); SC(5, s5, c5);
        DSink sk{&D, &L, seed, 0};
        sk.tau(5, nb.x);
        V3<Dual> f = rot<0>(c5, s5, fb), n = rot<0>(c5, s5, nb);
        sk.tau(4, n.y);
        f = rot<1>(c4, s4, f); n = rot<1>(c4, s4, n);
        sk.tau(3, n.z);
        f = rot<2>(c3, s3, f);
        sk.tau(0, f.x); sk.tau(1, f.y); sk.tau(2, f.z);
    })
}

HD double reb_barrier(double g, double delta) {   // ConstraintsBase.h:238-245
    if (g > delta) return -log(g);
    double t = (g - 2 * delta) / delta;
    return .5 * (t * t - 1) - log(delta);
}

// path-constraint value c (order: torque 24, joint 24, height 1, grf 5/foot) — MHPCConstraint.cpp
HD double wb_constraint(PhaseC& P, const WbCore& L, int c) {
    if (P.go_torque >= 0 && c >= P.go_torque && c < P.go_torque + 24) { int i = c - P.go_torque; return i < 12 ? -L.u[i] + P.torque_limit : L.u[i - 12] + P.torque_limit; }
    if (P.go_jspeed >= 0 && c >= P.go_jspeed && c < P.go_jspeed + 24) { int i = c - P.go_jspeed; return i < 12 ? L.x[24 + i] - P.jspeed_lb : -L.x[24 + i - 12] + P.jspeed_ub; }
    if (P.go_joint >= 0 && c >= P.go_joint && c < P.go_joint + 24) { int i = c - P.go_joint; return i < 12 ? L.x[6 + i] - pick3(P.joint_lb, i % 3) : -L.x[6 + i - 12] + pick3(P.joint_ub, i % 3); }
    if (P.go_height >= 0 && c == P.go_height) return L.x[2] - P.h_min;
    int i = c - P.go_grf, a = i / 5, r = i % 5, f = feet_of(P)[a];
    double fx = L.grf[3 * f], fy = L.grf[3 * f + 1], fz = L.grf[3 * f + 2];
    if (r == 0) return fz;
    if (r == 1) return -fx + P.mu * fz;
    if (r == 2) return fx + P.mu * fz;
    if (r == 3) return -fy + P.mu * fz;
    return fy + P.mu * fz;
}

// the same value by selects instead of one divergent region per constraint family (a single wave pays every region's LDS round trip):
//   g = fma(a, v1, t2)   torque / speed / joint / height: a = +-1, v1 the state or control entry, t2 the bound;  GRF: a = mu (1 for the
//   normal force), v1 = fz, t2 = +-fx / +-fy (0) — the same roundings as the expressions of wb_constraint
HD double wb_constraint_sel(PhaseC& P, const WbCore& L, int c) {
    const double* w = reinterpret_cast<const double*>(&L);
    constexpr int OX = offsetof(WbCore, x) / 8, OU = offsetof(WbCore, u) / 8, OG = offsetof(WbCore, grf) / 8;
    const int it = c - P.go_torque, is = c - P.go_jspeed, ij = c - P.go_joint;
    const bool bt = P.go_torque >= 0 && it >= 0 && it < 24, bs = P.go_jspeed >= 0 && is >= 0 && is < 24, bj = P.go_joint >= 0 && ij >= 0 && ij < 24;
    const bool bh = P.go_height >= 0 && c == P.go_height, lin = bt || bs || bj || bh;
    const int i = bt ? it : bs ? is : ij, lo = i < 12 ? i : i - 12;           // entry inside a 24-block: lower-bound half / upper-bound half
    const bool up = i >= 12;
    const int a3 = lo % 3;
    const double cj = up ? pick3(P.joint_ub, a3) : -pick3(P.joint_lb, a3);
    const double c0 = bt ? P.torque_limit : bs ? (up ? P.jspeed_ub : -P.jspeed_lb) : bj ? cj : -P.h_min;
    const double sg = bt ? (up ? 1.0 : -1.0) : (bh || !up) ? 1.0 : -1.0;
    const int i1l = bt ? OU + lo : bs ? OX + 24 + lo : bj ? OX + 6 + lo : OX + 2;
    const int ig = lin ? 0 : c - P.go_grf, ga = ig / 5, gr = ig - 5 * ga, f = feet_of(P).pick(ga & 3);
    const int i1 = lin ? i1l : OG + 3 * f + 2, i2 = lin ? i1l : OG + 3 * f + (gr >= 3 ? 1 : 0);
    const double v1 = w[i1], v2 = w[i2];
    const double a = lin ? sg : (gr == 0 ? 1.0 : P.mu);
    const double t2 = lin ? c0 : (gr == 0 ? 0.0 : ((gr == 1 || gr == 3) ? -v2 : v2));
    return hs_fma(a, v1, t2);
}
// reb_barrier with ONE logarithm: -log(g) above delta, the quadratic extension minus log(delta) below (ConstraintsBase.h:238-245)
HD double reb_barrier1(double g, double delta) {
    const bool above = g > delta;
    const double lg = log(above ? g : delta);
    double q = 0.0;
    if (!above) { const double t = (g - 2 * delta) / delta; q = .5 * (t * t - 1); }     // (a wave whose constraints all sit above their delta skips the division)
    return above ? -lg : q - lg;
}

struct SlotOut { double* cost; double* dsq; double* ming; double* maxh; };   // per (problem, slot) partials

// Copy-out of a CNT-element image whose element e = r + ROWS*c is produced by f(e, r, c) (LDS reads / constants): fully unrolled, every
// read is issued before the first store, (r, c) follow e = tid + NT*q without a division.  One wave; no barrier inside.
template <int NT, int CNT, int ROWS, class DST, class F>
HD void store_image(DST dst, int tid, F f) {
    constexpr int R = (CNT + NT - 1) / NT;
    const int c0 = tid / ROWS, r0 = tid - ROWS * c0;
    double v[R];
    _Pragma("unroll")
    for (int q = 0; q < R; q++) {
        const int e = tid + NT * q; int r = r0 + (NT * q) % ROWS, c = c0 + (NT * q) / ROWS;
        if (r >= ROWS) { r -= ROWS; c++; }
        v[q] = 0.0;
        if (R * NT == CNT || e < CNT) v[q] = f(e, r, c);
    }
    _Pragma("unroll")
    for (int q = 0; q < R; q++) { const int e = tid + NT * q; if (R * NT == CNT || e < CNT) dst[e] = v[q]; }
}

// -------------------------------------------------------------------------------------------------------
// Rollout of one knot k < h of problem b.   eps: line-search step.
template <int NT>
HD void wb_rollout_knot(WbCore& L, PhaseC& P, const ModelDev& md, int b, int k, double eps, int reb_active,
                        const double* x0, SlotOut so, size_t slot, int* fail_flag, bool ss = false, bool wr = true) {
    // wr = false: a PROBE of the step length eps (one candidate of a batched line-search launch): only the per-slot partials of the merit
    // function (cost, defect^2, min g, max |h|) and the divergence flag leave the wave, no trajectory / cache store
    // ss: knot of a phase WITHOUT shooting nodes (SS_set empty, a phase the receding-horizon update has just created,
    // MHPCProblem.cpp:340-351): X[k] is the simulated state handed over in L.xnext, X[k+1] = Xsim[k+1], the defect is zero
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + k) * 36, ku = ((size_t)b * h + k) * 12;
    double* Kst = L.Jc();   // 432 doubles: [Jc | Xm] are free until the contact solve
    RL_STAMP0()
    // every global read of the knot is issued in this first phase (one exposed HBM latency instead of five).  The control:
    //   multiple shooting: x - xbar = eps dX, so  u = ubar + eps dU + K (x - xbar) = ubar + eps (dU + K dX)  with K dX as the linear rollout
    //   left it (P.KdX, 12 values) - the knot does not read the 432 entries of K at all (equal to the reference's expression up
    //   to the rounding of x - xbar);
    //   single shooting (ss): x is the simulated state, K (x - xbar) is formed here; K travels in registers past the dynamics terms, which
    //   do not need the control.
    const size_t kk = (size_t)b * h + k;
    struct Late { double kr[7]; } late[HS_NLANES(NT)];
    HS_PHASE(NT,
        // ONE instruction stream for every lane group (a branch per group would wait for its own loads where the groups merge): the
        // addresses are picked per lane, idle lanes read a valid dummy
        const int g = tid < 36 ? 0 : tid < 48 ? 1 : tid < 60 ? 2 : 3, i = tid < 36 ? tid : tid < 48 ? tid - 36 : tid < 60 ? tid - 48 : 0;
        const double* pa = g == 0 ? P.Xbar + kx + i : (g == 1 && !ss) ? P.KdX + ku + i : P.Xbar + kx; const double* pb = P.dX + kx + (g == 0 ? i : 0);
        const double* rr = P.rref + (size_t)k * 80;          // the knot's references: one record, lane tid takes entry tid (+ the relative foot position)
        const double* pc = rr + tid;
        const double* pd = g == 0 ? P.Xbar + kx + (ss ? 0 : 36) + i : g == 1 ? P.Ubar + ku + i : rr + 64 + i;
        const double* pe = g == 0 ? P.dX + kx + (ss ? 0 : 36) + i : g == 1 ? P.dU + ku + i : rr;
        double va = *pa, vb = *pb, vc = *pc, vd = *pd, ve = *pe;
        const double* pw = tid < 36 ? &P.q[tid] : &P.r[tid < 48 ? tid - 36 : 0];      // (one load: two would wait for each other where they merge)
        const double vw = *pw;
        double er[2], dr[2];
        _Pragma("unroll") for (int q = 0; q < 2; q++) { const int c = q * NT + tid; er[q] = (c < P.ng) ? P.eps[kk * P.ng + c] : 0.0; dr[q] = (c < P.ng) ? P.delta[kk * P.ng + c] : 0.0; }
        if (ss) { Late& lt = late[HS_LANE(tid)]; _Pragma("unroll") for (int q = 0; q < 7; q++) { const int e = q * NT + tid; lt.kr[q] = (e < 432) ? P.K[kk * 432 + e] : 0.0; } }
        HS_CBAR();
        if (tid < 36) {
            const double xb = va, x = ss ? L.xnext[tid] : xb + eps * vb;
            L.xb[tid] = xb; L.x[tid] = x; if (wr) P.X[kx + tid] = x;
            L.tmp[tid] = vc; L.red[tid] = ss ? 0.0 : vd + eps * ve;
        } else if (tid < 48) {
            L.tmp[tid] = vc;
            if (ss) L.red[tid] = vd + eps * ve;
            else { const double u = vd + eps * (ve + va); L.u[i] = u; if (wr) P.U[ku + i] = u; L.tau[6 + i] = u; }
        }
        else if (tid < 60) { L.tmp[tid] = vc; L.red[tid] = vd; }          // reference foot position relative to the body (foot costs below)
        else if (tid < 64) L.red[tid] = vc;                               // reference contact flags
        if (tid < 18) { L.acc[tid] = 0.0; if (ss || tid < 6) L.tau[tid] = 0.0; } if (tid < 12) L.fext[tid] = 0.0; if (tid < 48) L.wq[tid] = vw;
        _Pragma("unroll") for (int q = 0; q < 2; q++) { const int c = q * NT + tid; if (c < P.ng) { L.gval()[c] = er[q]; L.bar()[c] = dr[q]; } })
    RL_STAMP(0)
    wb_terms<NT>(L, md, true);
    RL_STAMP(1)
    if (ss) {
        HS_PHASE(NT,
            const Late& lt = late[HS_LANE(tid)];
            _Pragma("unroll") for (int q = 0; q < 7; q++) { const int e = q * NT + tid; if (e < 432) Kst[e] = lt.kr[q]; })
        HS_PHASE(NT, if (tid < 12) {
            double s = 0; for (int j = 0; j < 36; j++) s += Kst[tid + 12 * j] * (L.x[j] - L.xb[j]);
            double u = L.red[36 + tid] + s;
            L.u[tid] = u; if (wr) P.U[ku + tid] = u; L.tau[6 + tid] = u;
        })
    }
    wb_kkt_direct<NT>(L, P.nc, feet_of(P), 0, P.bg_alpha);
    RL_STAMP(2)
    if (wr) {   // contact-solve cache for the LQ approximation of this knot (hs_types.hpp KC_*): fire-and-forget stores
        double* kc = P.kc + kk * KC_SIZE;
        HS_PHASE_L(NT,
            store_image<NT, 324, 324>(kc + KC_M, tid, [&](int e, int, int) { return L.M[e]; });
            store_image<NT, 216, 216>(kc + KC_X, tid, [&](int e, int, int) { return L.Xm()[e]; });
            store_image<NT, 216, 216>(kc + KC_J, tid, [&](int e, int, int) { return L.Jall[e]; });
            store_image<NT, 144, 144>(kc + KC_LG, tid, [&](int e, int, int) { return L.G()[e]; });       // the Gram matrix itself: the LQ knot factors it WITHOUT the damping
            if (tid < 18) { kc[KC_RDM + tid] = L.rdM[tid]; kc[KC_QDD + tid] = L.qdd[tid]; }
            if (tid < 12) { kc[KC_GRF + tid] = L.grf[tid]; kc[KC_LAM + tid] = L.lam[tid]; kc[KC_FP + tid] = L.fpos[tid]; kc[KC_FV + tid] = L.fvel[tid]; })
    }
    RL_STAMP(3)
    // integrate, defect of knot k+1 (and of knot 0 for the very first knot of phase 0), constraint values + barrier, cost terms:
    // one phase, every lane its own entries; the sums are taken afterwards in the reference's order by four lanes in parallel
    double* S = L.JX;      // scratch (the contact solve is done): [0,36) x-terms | [36,48) u-terms | [48,60) foot terms | [64,100) defect^2 | [100,136) xsim^2
    HS_PHASE(NT, if (tid < 36) {
        const double xs = (tid < 18) ? L.x[tid] + L.x[18 + tid] * P.dt : L.x[tid] + L.qdd[tid - 18] * P.dt;
        if (wr) P.Xsim[kx + 36 + tid] = xs;
        const double xn = ss ? xs : L.red[tid];
        const double d = xs - xn; if (wr) P.Defect[kx + 36 + tid] = d;
        if (ss) { L.xnext[tid] = xs; if (k == 0 && wr) { P.Xsim[kx + tid] = L.x[tid]; P.Defect[kx + tid] = 0.0; } }
        double dsq = d * d;
        if (x0 != nullptr && k == 0) { const double d0 = x0[(size_t)b * 36 + tid] - L.x[tid]; if (wr) { P.Xsim[kx + tid] = x0[(size_t)b * 36 + tid]; P.Defect[kx + tid] = d0; } dsq += d0 * d0; }
        S[64 + tid] = dsq; S[100 + tid] = xs * xs;
        const double dx = L.x[tid] - L.tmp[tid]; S[tid] = dx * L.wq[tid] * dx;
    } else if (tid < 48) {
        const int i = tid - 36; if (wr) P.Y[kk * 12 + i] = L.grf[i];
        const double du = L.u[i] - L.tmp[tid]; S[tid] = du * L.wq[36 + i] * du;
    } else if (tid < 52) {     // foot costs of foot f: place regulariser (stance), swing position, swing velocity (MHPCCost.cpp:4-245)
        const int f = tid - 48; const int rc = (int)L.red[60 + f];
        const double d0 = (L.fpos[3 * f] - L.x[0]) - L.red[48 + 3 * f], d1 = (L.fpos[3 * f + 1] - L.x[1]) - L.red[48 + 3 * f + 1], d2 = (L.fpos[3 * f + 2] - L.x[2]) - L.red[48 + 3 * f + 2];
        double l2 = 0, l3 = 0, l4 = 0;
        if (rc > 0 && P.w_foot_reg[0] >= 0) l2 = 0.5 * (d0 * P.w_foot_reg[0] * d0 + d1 * P.w_foot_reg[1] * d1 + d2 * P.w_foot_reg[2] * d2) * P.dt;
        if (rc == 0 && P.w_swing_pos[0] >= 0) l3 = 0.5 * (d0 * P.w_swing_pos[0] * d0 + d1 * P.w_swing_pos[1] * d1 + d2 * P.w_swing_pos[2] * d2) * P.dt;
        if (rc == 0 && P.w_swing_vel[0] >= 0) {
            const double v0 = L.fvel[3 * f] - L.tmp[48 + 3 * f], v1 = L.fvel[3 * f + 1] - L.tmp[48 + 3 * f + 1], v2 = L.fvel[3 * f + 2] - L.tmp[48 + 3 * f + 2];
            l4 = 0.5 * (v0 * P.w_swing_vel[0] * v0 + v1 * P.w_swing_vel[1] * v1 + v2 * P.w_swing_vel[2] * v2) * P.dt;
        }
        S[48 + f] = l2; S[52 + f] = l3; S[56 + f] = l4;
    }
    double gmin = 0.0;     // this lane's share of min(0, min_c g_c); the partial minima are folded below (a minimum does not depend on the order)
    for (int c = tid; c < P.ng; c += NT) {
        const double g = wb_constraint_sel(P, L, c), e = L.gval()[c], dl = L.bar()[c];
        if (wr) P.g[kk * P.ng + c] = g;
        L.bar()[c] = e * reb_barrier1(g, dl); gmin = fmin(gmin, g);
    }
    if (tid < 64) S[200 + tid] = gmin;)
    RL_STAMP(4)
    // every sum of the knot in ONE instruction stream of twelve steps: 23 lanes each add a run of at most twelve consecutive LDS entries -
    //   lanes 0..2 the state terms (three runs of 12), 3 the control terms, 4..6 the foot terms, 7..9 the squared defect, 10..12 the squared norm
    //   of the simulated state, 13 + 2 o + half the ReB cost of constraint object o (SinglePhase.cpp:394-402; an object has at most 24 constraints)
    // - (first entry, length) of a run come out of packed constants by shifts, not out of a branch per role; entries past a run's end are other
    // data of the knot (in bounds) and add 0.  The partial sums of a quantity are added in run order in the next phase (the reference adds the
    // terms one by one: same terms, a different association - 1e-16-level).  Lanes 16..19 also fold a quarter of the partial minima.
    HS_PHASE(NT, {
        const double* w = reinterpret_cast<const double*>(&L);
        constexpr int OS = offsetof(WbCore, JX) / 8, OBAR = offsetof(WbCore, GG) / 8 + 288 + MAXG;
        constexpr unsigned long long RUN_OFF0 = 0ull | 12ull << 8 | 24ull << 16 | 36ull << 24 | 48ull << 32 | 52ull << 40 | 56ull << 48 | 64ull << 56,      // lanes 0..7
                                     RUN_OFF1 = 76ull | 88ull << 8 | 100ull << 16 | 112ull << 24 | 124ull << 32,                                       // lanes 8..12
                                     RUN_LEN = 0xCCCCCC444CCCCull;                                                                                   // 4 bits per lane 0..12
        unsigned long long po = P.obj_off, pl = P.obj_sz; HS_PIN_S(po); HS_PIN_S(pl);
        const bool fixed = tid < 13; const int ob = tid < 23 ? tid - 13 : 0, o = ob >> 1, half = ob & 1;
        const int foff = (int)(((tid < 8 ? RUN_OFF0 : RUN_OFF1) >> (8 * (tid & 7))) & 255), flen = (int)((RUN_LEN >> (4 * (fixed ? tid : 0))) & 15);
        const int osz = (int)((pl >> (8 * o)) & 255) - 12 * half, olen = (tid >= 13 && tid < 23) ? (osz < 0 ? 0 : osz > 12 ? 12 : osz) : 0;
        const int off = fixed ? OS + foff : OBAR + (int)((po >> (8 * o)) & 255) + 12 * half, n = fixed ? flen : olen;
        double sum = 0;
        _Pragma("unroll") for (int i = 0; i < 12; i++) { const double v = w[off + i]; sum += (i < n) ? v : 0.0; }
        double sm = 0; const int mo = OS + 200 + 16 * (tid & 3);
        _Pragma("unroll") for (int i = 0; i < 16; i++) sm = fmin(sm, w[mo + i]);
        if (tid < 23) S[140 + tid] = sum;
        if (tid >= 16 && tid < 20) S[180 + tid] = sm;
    })
    HS_PHASE(NT, if (tid == 0) {          // running cost: the reference's order of additions between the quantities
        const double lq = (S[140] + S[141]) + S[142];
        double l = 0.5 * lq; l += 0.5 * S[143]; l *= P.dt;
        l += S[144]; l += S[145]; l += S[146];
        if (wr) P.lbase[kk] = l;
        if (reb_active) { _Pragma("unroll") for (int gI = 0; gI < 5; gI++) if (gI < P.nobj) l += P.dt * (S[153 + 2 * gI] + S[154 + 2 * gI]); }
        if (wr) P.l[kk] = l; so.cost[slot] = l;
    } else if (tid == 16) {
        so.ming[slot] = fmin(fmin(S[196], S[197]), fmin(S[198], S[199])); so.maxh[slot] = 0.0;
    } else if (tid == 32) {
        so.dsq[slot] = (S[147] + S[148]) + S[149];
    } else if (tid == 48) {
        const double nsq = (S[150] + S[151]) + S[152];
        if (nsq > 1e12 || !(nsq == nsq)) fail_flag[b] = 1;   // ||Xsim|| > 1e6 (SinglePhase.cpp:205)
    })
    RL_STAMP(5)
}

// terminal cost without AL (tracking + foot-place reg (x1) + touchdown-velocity penalty). MHPCCost.cpp:67-87,255-268
HD double wb_terminal_cost_base(PhaseC& P, const WbCore& L) {
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
HD void wb_rollout_terminal(WbCore& L, PhaseC& P, PhaseC* Pn, const ModelDev& md, int b, double eps, int al_active,
                            SlotOut so, size_t slot, bool ss = false, bool wr = true) {
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + h) * 36;
    HS_PHASE(NT, if (tid < 36) { double x = ss ? L.xnext[tid] : P.Xbar[kx + tid] + eps * P.dX[kx + tid]; L.x[tid] = x; if (wr) P.X[kx + tid] = x; }
             if (tid < 18) { L.acc[tid] = 0.0; L.tau[tid] = 0.0; } if (tid < 12) L.fext[tid] = 0.0;)
    const bool impact = (Pn != nullptr) && P.has_impact;
    wb_terms<NT>(L, md, impact);
    HS_PHASE(NT, if (tid == 0) {
        double pb = wb_terminal_cost_base(P, L); if (wr) P.Phibase[b] = pb;
        double maxh = 0, c = 0; int i = 0;
        for (int f = 0; f < 4; f++) if (P.td[f] && P.nt > 0) {
            double hh = L.fpos[3 * f + 2] - P.ground_height; if (wr) P.th[(size_t)b * P.nt + i] = hh; maxh = fmax(maxh, fabs(hh));
            double sg = P.sigma[(size_t)b * P.nt + i], lm = P.lambda[(size_t)b * P.nt + i];
            c += 0.5 * sg * hh * hh; c += lm * hh; i++;
        }
        double Phi = pb; if (al_active && P.nt > 0) Phi += c;
        if (wr) P.Phi[b] = Phi;
        so.cost[slot] = Phi; so.ming[slot] = 0.0; so.maxh[slot] = maxh; so.dsq[slot] = 0.0;
    })
    if (Pn == nullptr) return;
    // reset map (MHPCReset.cpp:4-28): impact if any touchdown, then optional WB->SRB projection
    if (impact) {
        int tdfeet[4] = {0, 0, 0, 0}; int ntd = 0; for (int f = 0; f < 4; f++) if (P.td[f]) tdfeet[ntd++] = f;
        wb_kkt_direct<NT>(L, ntd, Feet4{tdfeet[0], tdfeet[1], tdfeet[2], tdfeet[3]}, 1, 0.0);
    } else { HS_PHASE(NT, if (tid < 18) L.qdd[tid] = L.x[18 + tid];) }
    const int nn = Pn->n;
    const size_t nx = ((size_t)b * (Pn->h + 1)) * nn;
    HS_PHASE(NT, if (tid < nn) {
        double xi;
        if (nn == 36) xi = (tid < 18) ? L.x[tid] : L.qdd[tid - 18];
        else xi = (tid < 6) ? L.x[tid] : L.qdd[tid - 6];          // StateProjection: x[0:6], x[18:24] (MHPCReset.h:24-26)
        if (wr) Pn->Xsim[nx + tid] = xi;
        double d = Pn->shooting ? xi - (Pn->Xbar[nx + tid] + eps * Pn->dX[nx + tid]) : 0.0;     // no shooting node at the start of a young phase: X[0] = x_init
        if (wr) Pn->Defect[nx + tid] = d; L.red[tid] = d * d;
        if (nn == 36) L.xnext[tid] = xi;
    })
    HS_PHASE(NT, if (tid == 0) { double s = 0; for (int i = 0; i < nn; i++) s += L.red[i]; so.dsq[slot] = s; })
}

// -------------------------------------------------------------------------------------------------------
// coalesced copy LDS -> global
template <int NT> HD void store_block(double* dst, const double* src, int n) { HS_PHASE_L(NT, for (int i = tid; i < n; i += NT) dst[i] = src[i];) }

// References of knot k for the cost partials, fetched with the state so that no later phase waits on HBM:
//   tmp[0,36) xr | tmp[36,48) ur | red[0,12) foot_pos | red[12,24) foot_vel | red[24,27) body_pos | red[28,32) ref_contact
HD void wb_cost_prefetch(WbCore& L, PhaseC& P, int k, int tid) {
    if (tid < 36) L.tmp[tid] = P.xr[(size_t)k * 36 + tid];
    else if (tid < 48) L.tmp[tid] = P.ur[(size_t)k * 12 + tid - 36];
    if (tid < 12) { L.red[tid] = P.foot_pos[(size_t)k * 12 + tid]; L.red[12 + tid] = P.foot_vel[(size_t)k * 12 + tid]; }
    else if (tid < 15) L.red[12 + tid] = P.body_pos[(size_t)k * 3 + tid - 12];
    else if (tid < 19) L.red[13 + tid] = (double)P.ref_contact[(size_t)k * 4 + tid - 15];
}
// foot-cost Jacobian blocks of the knot: JP (position-type rows, base-translation and velocity columns zero),
// JW (velocity-type rows [d vel/dq | J]), with per-row weights (dt folded in) and residuals.  `terminal` selects the
// terminal cost objects (foot-place reg x2, touchdown velocity) instead of the running ones.
struct WbLqLds;
HD void wb_cost_blocks_lane(WbLqLds& S, PhaseC& P, bool terminal, int tid);
template <int NT>
HD void wb_cost_blocks(WbLqLds& S, PhaseC& P, int k, bool terminal) {
    (void)k;
    HS_PHASE(NT, wb_cost_blocks_lane(S, P, terminal, tid);)
}
HD void wb_cost_blocks_lane(WbLqLds& S, PhaseC& P, bool terminal, int tid) {
    WbCore& L = S.c; WbDeriv& D = S.d;
    const double* rc = L.red + 28; const double* fp = L.red; const double* bp = L.red + 24;     // wb_cost_prefetch
    {
        if (tid < 12) {
            const int f = tid / 3, a = tid % 3;
            double wpos = 0, wvel = 0;
            if (!terminal) {
                if (rc[f] > 0 && P.w_foot_reg[0] >= 0) wpos = pick3(P.w_foot_reg, a) * P.dt;
                if (rc[f] == 0 && P.w_swing_pos[0] >= 0) wpos = pick3(P.w_swing_pos, a) * P.dt;
                if (rc[f] == 0 && P.w_swing_vel[0] >= 0) wvel = pick3(P.w_swing_vel, a) * P.dt;
            } else {
                if (rc[f] > 0 && P.w_foot_reg[0] >= 0) wpos = 2.0 * pick3(P.w_foot_reg, a);
                if (P.td[f] && P.n_td > 0 && P.w_td_vel >= 0 && a == 2) wvel = P.w_td_vel;
            }
            D.wp[tid] = wpos; D.wv[tid] = wvel;
            D.ep[tid] = (L.fpos[tid] - L.x[a]) - (fp[tid] - bp[a]);
            D.ev[tid] = terminal ? L.fvel[tid] : (L.fvel[tid] - L.red[12 + tid]);
        }
    }
}
// column d of  JP^T diag(wp) JP + JW^T diag(wv) JW  written to out[0..35] (stride 36), and the gradient entry.  The blocks are
// read in place:  JP = [0 | J(:,3:18) | 0] (position-type rows: base-translation and velocity columns zero, MHPCCost.cpp:54-59),
// JW = [d(foot vel)/dq | J] (velocity-type rows)
HD double wb_cost_column(const WbDeriv& D, const double* Jall, const double* dvel, int d, double* colout /* stride 36 */) {
    // foot by foot; a foot's position-type rows (foot-place reg / swing pos) only touch columns 3..17, its velocity-type rows
    // (swing vel / touchdown vel) exist only for swing or touchdown feet: the activity tests are uniform over the wave, so
    // a stance knot skips the 36x36 velocity blocks entirely.
    double acc[36];
    _Pragma("unroll")
    for (int i = 0; i < 36; i++) acc[i] = 0.0;
    double g = 0;
    for (int f = 0; f < 4; f++) {
        const double* wp = D.wp + 3 * f; const double* wv = D.wv + 3 * f;
        const double* J0 = Jall + (3 * f) * 18; const double* J1 = J0 + 18; const double* J2 = J1 + 18;
        if (wp[0] != 0.0 || wp[1] != 0.0 || wp[2] != 0.0) {
            const bool in = d >= 3 && d < 18; const int dd = in ? d : 3;
            const double t0 = in ? wp[0] * J0[dd] : 0.0, t1 = in ? wp[1] * J1[dd] : 0.0, t2 = in ? wp[2] * J2[dd] : 0.0;
            g += t0 * D.ep[3 * f] + t1 * D.ep[3 * f + 1] + t2 * D.ep[3 * f + 2];
            _Pragma("unroll")
            for (int i = 3; i < 18; i++) acc[i] += J0[i] * t0 + J1[i] * t1 + J2[i] * t2;
        }
        if (wv[0] != 0.0 || wv[1] != 0.0 || wv[2] != 0.0) {
            const double* V0 = dvel + (3 * f) * 18; const double* V1 = V0 + 18; const double* V2 = V1 + 18;
            const int dd = d < 18 ? d : d - 18;
            const double t0 = wv[0] * (d < 18 ? V0[dd] : J0[dd]), t1 = wv[1] * (d < 18 ? V1[dd] : J1[dd]), t2 = wv[2] * (d < 18 ? V2[dd] : J2[dd]);
            g += t0 * D.ev[3 * f] + t1 * D.ev[3 * f + 1] + t2 * D.ev[3 * f + 2];
            _Pragma("unroll")
            for (int i = 0; i < 18; i++) { acc[i] += V0[i] * t0 + V1[i] * t1 + V2[i] * t2; acc[18 + i] += J0[i] * t0 + J1[i] * t1 + J2[i] * t2; }
        }
    }
    _Pragma("unroll")
    for (int i = 0; i < 36; i++) colout[i * 36] = acc[i];
    return g;
}

#ifndef HS_HOST_EMU
// the 54 MFMAs of the foot-cost product E^T diag(w) [E | e] of one wave: lane (li = lane & 15, lk = lane >> 4) ends up with
// c[ti][tj][q] = entry (row 16 ti + lk + 4 q, column 16 tj + li)
HD void wb_cost_gram_acc(WbCore& L, WbDeriv& D, int lane, d4_t (&c)[3][3]) {
    const int li = lane & 15, lk = lane >> 4;
    bool anyv = false;
    _Pragma("unroll") for (int a = 0; a < 12; a++) anyv = anyv || (D.wv[a] != 0.0);
    _Pragma("unroll") for (int ti = 0; ti < 3; ti++) _Pragma("unroll") for (int tj = 0; tj < 3; tj++) c[ti][tj] = d4_t{0.0, 0.0, 0.0, 0.0};
    _Pragma("unroll") for (int kg = 0; kg < 6; kg++) {
        if (kg >= 3 && !anyv) break;
        const int k = 4 * kg + lk, r = (kg < 3) ? k : k - 12;
        const double w = (kg < 3) ? D.wp[r] : D.wv[r], res = (kg < 3) ? D.ep[r] : D.ev[r];
        double e[3];
        _Pragma("unroll") for (int t = 0; t < 3; t++) {
            const int col = 16 * t + li, cc = col < 36 ? col : 35, lo = cc < 18, c18 = lo ? cc : cc - 18;
            const double vj = L.Jall[r * 18 + c18];
            if (kg < 3) e[t] = (cc >= 3 && lo && col < 36) ? vj : 0.0;
            else { const double vd = L.dvel()[r * 18 + c18]; e[t] = col < 36 ? (lo ? vd : vj) : 0.0; }
        }
        _Pragma("unroll") for (int ti = 0; ti < 3; ti++) {
            const double a = w * e[ti];
            _Pragma("unroll") for (int tj = 0; tj < 3; tj++) {
                const double bb = (tj == 2 && li == 4) ? res : e[tj];      // column 36: the residual -> gradient
                c[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, c[ti][tj], 0, 0, 0);
            }
        }
    }
}
#endif
// Foot-cost Hessian and gradient of a knot on the fp64 matrix cores (one wave).  With E the 24 x 36 matrix of cost rows
//   rows 0..11  (position type, weight wp, residual ep): [0 0 0 | J(3f+r, 3:18) | 0]      (MHPCCost.cpp:54-59)
//   rows 12..23 (velocity type, weight wv, residual ev): [d(foot vel)/dq | J(3f+r, :)]
// the result is  E^T diag(w) [E | e]  (36 x 37): columns 0..35 -> W[row*36 + col] (dense staging tile of lxx), column 36 -> gvec.
// 9 output tiles x 6 k-steps of v_mfma_f64_16x16x4; the three operand values of a k-step serve as A (scaled by the row weight) and
// as B; a knot without velocity-type rows (all feet in stance, no touchdown) skips their three k-steps.
template <int NT>
HD void wb_cost_gram(WbLqLds& S, double* gvec) {
    WbCore& L = S.c; WbDeriv& D = S.d;
#ifdef HS_HOST_EMU
    HS_PHASE_L(NT, if (tid < 37) {
        const int j = tid;
        auto E = [&](int k, int c) { if (k < 12) return (c >= 3 && c < 18) ? L.Jall[k * 18 + c] : 0.0; const int r = k - 12; return c < 18 ? L.dvel()[r * 18 + c] : L.Jall[r * 18 + c - 18]; };
        for (int i = 0; i < 36; i++) {
            double s = 0;
            for (int k = 0; k < 24; k++) { const double w = k < 12 ? D.wp[k] : D.wv[k - 12]; s += w * E(k, i) * (j < 36 ? E(k, j) : (k < 12 ? D.ep[k] : D.ev[k - 12])); }
            if (j < 36) D.W[i * 36 + j] = s; else gvec[i] = s;
        }
    })
#else
    HS_PHASE_L(NT, if (tid < 64) {
        const int li = tid & 15, lk = tid >> 4;
        d4_t c[3][3];
        wb_cost_gram_acc(L, D, tid, c);
        _Pragma("unroll") for (int ti = 0; ti < 3; ti++) _Pragma("unroll") for (int tj = 0; tj < 3; tj++) _Pragma("unroll") for (int q = 0; q < 4; q++) {
            const int row = 16 * ti + lk + 4 * q, col = 16 * tj + li;
            if (row < 36) { if (col < 36) D.W[row * 36 + col] = c[ti][tj][q]; else if (col == 36) gvec[row] = c[ti][tj][q]; }
        }
    })
#endif
}

// cache commit: which regions a round of NT elements (q) touches is known at compile time: no comparison chain per element
#define KC_REGION(s_, e_, ptr_) if (NT * q + NT - 1 >= (s_) && NT * q < (e_)) { if (NT * q >= (s_) && NT * q + NT - 1 < (e_)) (ptr_)[i - (s_)] = v; else if (i >= (s_) && i < (e_)) (ptr_)[i - (s_)] = v; }
// LQ approximation of knot k < h (recomputes the contact solve at the stored X,U like WBM.cpp:463)
template <int NT>
HD void wb_lq_knot(WbLqLds& S, PhaseC& P, const ModelDev& md, int b, int k, int reb_active, bool cached = false) {
    WbCore& L = S.c; WbDeriv& D = S.d;
    const int h = P.h; const double dt = P.dt;
    const size_t kx = ((size_t)b * (h + 1) + k) * 36, ku = ((size_t)b * h + k) * 12, kk = (size_t)b * h + k;
    // the barrier derivative tables only need g, delta, eps of the rollout: their global loads are issued together with x, u
    LQ_STAMP0()
    HS_PHASE(NT,
        // ---- every global read of the knot first (one exposed HBM round trip), the LDS stores afterwards
        const double vx = (tid < 36) ? P.X[kx + tid] : 0.0, vu = (tid < 12) ? P.U[ku + tid] : 0.0;
        const double vt = (tid < 36) ? P.xr[(size_t)k * 36 + tid] : (tid < 48) ? P.ur[(size_t)k * 12 + tid - 36] : 0.0;
        const double vf = (tid < 12) ? P.foot_pos[(size_t)k * 12 + tid] : (tid < 15) ? P.body_pos[(size_t)k * 3 + tid - 12] : (tid < 19) ? (double)P.ref_contact[(size_t)k * 4 + tid - 15] : 0.0;
        const double vv = (tid < 12) ? P.foot_vel[(size_t)k * 12 + tid] : 0.0;
        const double vw = (tid < 36) ? P.q[tid] : (tid < 48) ? P.r[tid - 36] : 0.0;
        double gr[2], dr[2], er[2];
        _Pragma("unroll") for (int q = 0; q < 2; q++) { const int c = q * NT + tid; const size_t gi = kk * P.ng + c; const bool in = c < P.ng; gr[q] = in ? P.g[gi] : 1.0; dr[q] = in ? P.delta[gi] : 1.0; er[q] = in ? P.eps[gi] : 0.0; }
        double r[KC_SIZE / NT];
        if (cached) {   // the rollout that produced X[k], U[k] left its contact solve behind: fetch it instead of repeating the terms and the factorisations
            const double* kc = P.kc + kk * KC_SIZE;
            _Pragma("unroll") for (int q = 0; q < KC_SIZE / NT; q++) r[q] = kc[q * NT + tid];
        }
        HS_CBAR();
        if (tid < 36) L.x[tid] = vx; if (tid < 12) { L.u[tid] = vu; L.fext[tid] = 0.0; }
        if (tid < 18) { L.acc[tid] = 0.0; L.tau[tid] = 0.0; }
        // cost references (layout of wb_cost_prefetch): tmp[0,36) xr | tmp[36,48) ur | red[0,12) foot_pos | red[12,24) foot_vel | red[24,27) body_pos | red[28,32) ref_contact
        if (tid < 48) { L.tmp[tid] = vt; L.wq[tid] = vw; }
        if (tid < 12) { L.red[tid] = vf; L.red[12 + tid] = vv; } else if (tid < 15) L.red[12 + tid] = vf; else if (tid < 19) L.red[13 + tid] = vf;
        _Pragma("unroll") for (int q = 0; q < 2; q++) {
            const int c = q * NT + tid; const double g = gr[q], delta = dr[q], e = er[q]; double bd, bdd;
            if (g > delta) { bd = -1.0 / g; bdd = 1.0 / (g * g); } else { bd = (g - 2 * delta) / delta / delta; bdd = 1.0 / (delta * delta); }
            if (c < P.ng) { D.bd()[c] = reb_active ? e * bd : 0.0; D.bdd()[c] = reb_active ? e * bdd : 0.0; }
        }
        if (cached) {
            // trig table straight from the registers (same values wb_trig would read back from LDS), while the cache reads are in flight
            if (tid < 18) { double sv, cv; sincos_(vx, sv, cv); L.cs[tid] = cv; L.sn[tid] = sv; }
            _Pragma("unroll") for (int q = 0; q < KC_SIZE / NT; q++) {
                const int i = q * NT + tid; const double v = r[q];
                KC_REGION(KC_M, KC_X, L.M) KC_REGION(KC_X, KC_LG, L.Xm()) KC_REGION(KC_LG, KC_RDM, D.LGs) KC_REGION(KC_RDM, KC_RDG, L.rdM)
                KC_REGION(KC_QDD, KC_GRF, L.qdd) KC_REGION(KC_GRF, KC_LAM, L.grf) KC_REGION(KC_LAM, KC_J, L.lam)
                KC_REGION(KC_J, KC_FP, L.Jall) KC_REGION(KC_FP, KC_FV, L.fpos) KC_REGION(KC_FV, KC_FV + 12, L.fvel)
            }
        })
    LQ_STAMP(11)
    if (cached) {
        // the cache holds the Gram matrix G = X^T X: its factor WITHOUT the damping of the forward solve (see wb_keep_schur); in the two-wave
        // knot this runs on wave 1 behind its tangent round, the shorter one (wb_dpass, chol_g)
        if (NT < 128) chol_f<NT, 12, 12>(D.LGs, D.LGs, D.rdGs, 0.0);
        LQ_STAMP(0)
    } else {
        HS_PHASE(NT, if (tid < 12) L.tau[6 + tid] = L.u[tid];)
        wb_terms<NT>(L, md, true);
        LQ_STAMP(0)
        wb_kkt_direct<NT>(L, P.nc, feet_of(P), 0, P.bg_alpha);
        wb_keep_schur<NT>(L, D);
    }
    LQ_STAMP(1)
    const int m = 3 * P.nc;
    HS_PHASE(NT, if (tid < 18) L.acc[tid] = L.qdd[tid]; if (tid < 12) L.fext[tid] = L.grf[tid];)
    wb_dpass<NT>(L, D, md, GRAV, 1.0, 1.0, 1.0, false, cached);
    LQ_STAMP(2)
    if (NT >= 128) {
        // ---- two waves, two jobs, no workgroup barrier in between (disjoint LDS): wave 0 solves the 48 KKT columns and copies A, B, C, D
        // out; wave 1 forms every cost partial of the knot meanwhile and stores lxx / lx straight from the matrix-core accumulators
        HS_WPHASE_W(0, if (tid < 48) {
        const int d = tid;
        double top[18], bot[12];
        {   // right-hand sides: every LDS read of the lane first, unconditionally (clamped addresses), then selects - a branch per entry would
            // pay one exposed LDS round trip per entry (a single wave has nothing to hide it behind)
            const int dc = d < 36 ? d : 35, c = dc < 18 ? dc : dc - 18, d2 = d < 18 ? d : 0;
            double w1[18], w2[18], vg[12], vd[12], vj[12];
            _Pragma("unroll") for (int i = 0; i < 18; i++) { w1[i] = D.W[wb_pi(i) * WT + dc]; w2[i] = D.W[wb_pi(i) * WT + 36 + d2]; }
            const Feet4 ft = feet_of(P);
            _Pragma("unroll") for (int a = 0; a < 12; a++) {
                const int r = 3 * ft[a / 3] + a % 3;        // (uniform: scalar selects)
                vg[a] = L.G()[r * 18 + c]; vd[a] = L.dvel()[r * 18 + c]; vj[a] = L.Jall[r * 18 + c];
            }
            HS_CBAR();
            _Pragma("unroll") for (int i = 0; i < 18; i++) top[i] = (d < 36) ? (w1[i] + ((d < 18) ? w2[i] : 0.0)) : ((wb_pi(i) == 6 + d - 36) ? 1.0 : 0.0);   // lanes 36+: tau tangent = -dJTF; entries in the legs-first order
            _Pragma("unroll") for (int a = 0; a < 12; a++) {
                const double t1 = dc < 18 ? vg[a] : 2.0 * vd[a], t2 = dc < 18 ? vd[a] : vj[a];       // footAccPartialDv == 2 footVelPartialDq
                bot[a] = (a < m && d < 36) ? t1 + 2.0 * P.bg_alpha * t2 : 0.0;
            }
        }
        wb_kkt_column(L, D, top, bot, false, m);
        if (d < 36) {
            _Pragma("unroll")
            for (int i = 0; i < 18; i++) D.W[WR0 + wb_pi(i) * 36 + d] = -top[i] * dt + ((d == 18 + wb_pi(i)) ? 1.0 : 0.0);     // rows 18..35 of A
            _Pragma("unroll")
            for (int a = 0; a < 12; a++) D.stC()[a + 12 * d] = 0.0;
            for (int a = 0; a < m; a++) D.stC()[(3 * P.feet[a / 3] + a % 3) + 12 * d] = bot[a];
        } else {
            const int j = d - 36;
            _Pragma("unroll")
            for (int i = 0; i < 18; i++) D.stB()[wb_pi(i) + 18 * j] = top[i] * dt;
            _Pragma("unroll")
            for (int a = 0; a < 12; a++) D.stD()[a + 12 * j] = 0.0;
            for (int a = 0; a < m; a++) D.stD()[(3 * P.feet[a / 3] + a % 3) + 12 * j] = -bot[a];
        }
        })
        LQ_STAMP(3)
        // wave 1.  Scratch in the free Jc block of JX (the columns only read Xm): [0,36) lx without the foot terms | [36,72) diagonal
        // additions | [72,84) luu diagonal | [84,120) lyy 3x3 blocks (r' + 3 column) | [120,156) foot-cost gradient (emulator only)
        double* const T1 = L.Jc();
        WB_W1_STAMP_BEGIN()
        HS_WPHASE_W(1, wb_cost_blocks_lane(S, P, false, tid);
            if (tid < 36) {
                const int d = tid;
                double lxd = dt * L.wq[d] * (L.x[d] - L.tmp[d]);
                double diag = dt * L.wq[d];
                if (P.go_joint >= 0 && d >= 6 && d < 18) {
                    const int i = d - 6;
                    lxd += dt * (D.bd()[P.go_joint + i] - D.bd()[P.go_joint + 12 + i]); diag += dt * (D.bdd()[P.go_joint + i] + D.bdd()[P.go_joint + 12 + i]);
                }
                if (P.go_height >= 0 && d == 2) { lxd += dt * D.bd()[P.go_height]; diag += dt * D.bdd()[P.go_height]; }
                if (P.go_jspeed >= 0 && d >= 24) {
                    const int i = d - 24;
                    lxd += dt * (D.bd()[P.go_jspeed + i] - D.bd()[P.go_jspeed + 12 + i]); diag += dt * (D.bdd()[P.go_jspeed + i] + D.bdd()[P.go_jspeed + 12 + i]);
                }
                T1[d] = lxd; T1[36 + d] = diag;
            })
#ifdef HS_HOST_EMU
        // (the emulator has no accumulator registers: a host-side tile stands in for them; same transposed placement as the GPU path)
        static double emu_tile[1296];
        HS_WPHASE_W(1, if (tid < 37) {
            const int j = tid;
            auto E = [&](int k2, int c) { if (k2 < 12) return (c >= 3 && c < 18) ? L.Jall[k2 * 18 + c] : 0.0; const int r = k2 - 12; return c < 18 ? L.dvel()[r * 18 + c] : L.Jall[r * 18 + c - 18]; };
            for (int i = 0; i < 36; i++) {
                double sm = 0;
                for (int k2 = 0; k2 < 24; k2++) { const double w = k2 < 12 ? D.wp[k2] : D.wv[k2 - 12]; sm += w * E(k2, i) * (j < 36 ? E(k2, j) : (k2 < 12 ? D.ep[k2] : D.ev[k2 - 12])); }
                if (j < 36) emu_tile[i * 36 + j] = sm; else T1[120 + i] = sm;
            }
        })
        HS_WPHASE_W(1, if (tid < 36) { emu_tile[tid * 36 + tid] += T1[36 + tid]; P.lx[kk * P.rs + tid] = T1[tid] + T1[120 + tid]; })
        HS_WPHASE_W(1, store_image<64, 1296, 36>(P.lxx + kk * P.rs, tid, [&](int, int r, int c) { return emu_tile[c * 36 + r]; });)
#else
        HS_WPHASE_W(1, {
            const int li = tid & 15, lk = tid >> 4;
            d4_t c[3][3];
            wb_cost_gram_acc(L, D, tid, c);
            // entry (row, col) goes to the TRANSPOSED position col + 36 row (lxx is symmetric; the 16 lanes of a row write 128 contiguous bytes)
            _Pragma("unroll") for (int ti = 0; ti < 3; ti++) _Pragma("unroll") for (int tj = 0; tj < 3; tj++) _Pragma("unroll") for (int q = 0; q < 4; q++) {
                const int row = 16 * ti + lk + 4 * q, col = 16 * tj + li;
                if (row < 36) {
                    if (col < 36) { double v = c[ti][tj][q]; if (row == col) v += T1[36 + row]; P.lxx[kk * P.rs + col + 36 * row] = v; }
                    else if (col == 36) P.lx[kk * P.rs + row] = T1[row] + c[ti][tj][q];
                }
            }
        })
#endif
        // lu, luu (diagonal + torque barrier), ly, lyy (grf barrier: one 3x3 block per foot)
        HS_WPHASE_W(1, if (tid < 12) {
            const int i = tid;
            double lu = dt * L.wq[36 + i] * (L.u[i] - L.tmp[36 + i]), luu = dt * L.wq[36 + i];
            if (P.go_torque >= 0) { lu += dt * (-D.bd()[P.go_torque + i] + D.bd()[P.go_torque + 12 + i]); luu += dt * (D.bdd()[P.go_torque + i] + D.bdd()[P.go_torque + 12 + i]); }
            P.lu[kk * P.rs + i] = lu; T1[72 + i] = luu;
            double ly = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0; const int f = i / 3, r = i % 3; int a = -1; for (int t = 0; t < P.nc; t++) if (P.feet[t] == f) a = t;
            if (P.go_grf >= 0 && a >= 0) {
                const double mu = P.mu;
                for (int c = 0; c < 5; c++) {
                    const double r0 = (c == 1) ? -1.0 : (c == 2) ? 1.0 : 0.0, r1 = (c == 3) ? -1.0 : (c == 4) ? 1.0 : 0.0, r2 = (c == 0) ? 1.0 : mu;
                    const double rr = (r == 0) ? r0 : (r == 1) ? r1 : r2;
                    const double hb = dt * D.bdd()[P.go_grf + 5 * a + c] * rr;
                    ly += D.bd()[P.go_grf + 5 * a + c] * rr;
                    b0 += hb * r0; b1 += hb * r1; b2 += hb * r2;
                }
            }
            T1[84 + 3 * i] = b0; T1[85 + 3 * i] = b1; T1[86 + 3 * i] = b2;
            P.ly[kk * P.rs + i] = dt * ly;
        })
        HS_WPHASE_W(1, store_image<64, 144, 12>(P.luu + kk * P.rs, tid, [&](int, int r, int c) { return r == c ? T1[72 + r] : 0.0; });
                       store_image<64, 144, 12>(P.lyy + kk * P.rs, tid, [&](int, int r, int c) { return (r / 3 == c / 3) ? T1[84 + (r % 3) + 3 * c] : 0.0; });)
        // wave 1 has been done for a while when wave 0 leaves the column solves: both copy A, B, C, D out
        WB_W1_STAMP_END(5)      // wave 1: its cost partials, start to end
        hs_phase_sync_all<NT>();
        HS_PHASE_L(NT,
            store_image<NT, 648, 18>(P.A + kk * P.rs, tid, [&](int, int r, int c) { return D.W[WR0 + r * 36 + c]; });     /* rows 18..35 of A (the upper rows are [I, dt I]) */
            store_image<NT, 432, 36>(P.C + kk * P.rs, tid, [&](int e, int, int) { return D.stC()[e]; });
            store_image<NT, 216, 18>(P.B + kk * P.rs, tid, [&](int e, int, int) { return D.stB()[e]; });     /* rows 18..35 of B (the upper rows are zero) */
            store_image<NT, 144, 12>(P.D + kk * P.rs, tid, [&](int e, int, int) { return D.stD()[e]; });)
        LQ_STAMP(4)
        return;
    }
    // lane d < 36: column d of the continuous partials, right-hand side top = d tau - d(J^T F) (18), bot = d(foot acc) + Baumgarte
    // terms (m) -> column d of A (rows 18..35) and of C ; lanes 36..47: unit torque j -> column j of B and of D
    HS_PHASE(NT, if (tid < 48) {
        const int d = tid;
        double top[18], bot[12];
        {   // right-hand sides: every LDS read of the lane first, unconditionally (clamped addresses), then selects - a branch per entry would
            // pay one exposed LDS round trip per entry (a single wave has nothing to hide it behind)
            const int dc = d < 36 ? d : 35, c = dc < 18 ? dc : dc - 18, d2 = d < 18 ? d : 0;
            double w1[18], w2[18], vg[12], vd[12], vj[12];
            _Pragma("unroll") for (int i = 0; i < 18; i++) { w1[i] = D.W[wb_pi(i) * WT + dc]; w2[i] = D.W[wb_pi(i) * WT + 36 + d2]; }
            const Feet4 ft = feet_of(P);
            _Pragma("unroll") for (int a = 0; a < 12; a++) {
                const int r = 3 * ft[a / 3] + a % 3;        // (uniform: scalar selects)
                vg[a] = L.G()[r * 18 + c]; vd[a] = L.dvel()[r * 18 + c]; vj[a] = L.Jall[r * 18 + c];
            }
            HS_CBAR();
            _Pragma("unroll") for (int i = 0; i < 18; i++) top[i] = (d < 36) ? (w1[i] + ((d < 18) ? w2[i] : 0.0)) : ((wb_pi(i) == 6 + d - 36) ? 1.0 : 0.0);   // lanes 36+: tau tangent = -dJTF; entries in the legs-first order
            _Pragma("unroll") for (int a = 0; a < 12; a++) {
                const double t1 = dc < 18 ? vg[a] : 2.0 * vd[a], t2 = dc < 18 ? vd[a] : vj[a];       // footAccPartialDv == 2 footVelPartialDq
                bot[a] = (a < m && d < 36) ? t1 + 2.0 * P.bg_alpha * t2 : 0.0;
            }
        }
        wb_kkt_column(L, D, top, bot, false, m);
        if (d < 36) {
            _Pragma("unroll")
            for (int i = 0; i < 18; i++) D.W[WR0 + wb_pi(i) * 36 + d] = -top[i] * dt + ((d == 18 + wb_pi(i)) ? 1.0 : 0.0);     // rows 18..35 of A
            _Pragma("unroll")
            for (int a = 0; a < 12; a++) D.stC()[a + 12 * d] = 0.0;
            for (int a = 0; a < m; a++) D.stC()[(3 * P.feet[a / 3] + a % 3) + 12 * d] = bot[a];
        } else {
            const int j = d - 36;
            _Pragma("unroll")
            for (int i = 0; i < 18; i++) D.stB()[wb_pi(i) + 18 * j] = top[i] * dt;
            _Pragma("unroll")
            for (int a = 0; a < 12; a++) D.stD()[a + 12 * j] = 0.0;
            for (int a = 0; a < m; a++) D.stD()[(3 * P.feet[a / 3] + a % 3) + 12 * j] = -bot[a];
        }
    })
    LQ_STAMP(3)
    // A = [I, dt I; dt*dqdd_dq, I + dt*dqdd_dv]  (WBM.cpp:68, 122-125): the lower half is the data (RecLayout), coalesced store
    HS_PHASE_L(NT,
        store_image<NT, 648, 18>(P.A + kk * P.rs, tid, [&](int, int r, int c) { return D.W[WR0 + r * 36 + c]; });     /* rows 18..35 of A (the upper rows are [I, dt I]) */
        store_image<NT, 432, 36>(P.C + kk * P.rs, tid, [&](int e, int, int) { return D.stC()[e]; });
        store_image<NT, 216, 18>(P.B + kk * P.rs, tid, [&](int e, int, int) { return D.stB()[e]; });     /* rows 18..35 of B (the upper rows are zero) */
        store_image<NT, 144, 12>(P.D + kk * P.rs, tid, [&](int e, int, int) { return D.stD()[e]; });)
    LQ_STAMP(4)
    // ---------------- cost partials
    wb_cost_blocks<NT>(S, P, k, false);
    wb_cost_gram<NT>(S, D.JPW);          // the column-solve staging has gone out to memory: JPW[0,36) takes the gradient
    HS_PHASE_L(NT, if (tid < 36) {
        const int d = tid;
        double lxd = dt * L.wq[d] * (L.x[d] - L.tmp[d]);
        double diag = dt * L.wq[d];
        lxd += D.JPW[d];
        // ReB fold on x (joint limits: x[6+i], height: x[2]) — rank-1 updates on the diagonal (ConstraintsBase.h:282-287)
        if (P.go_joint >= 0 && d >= 6 && d < 18) {
            const int i = d - 6;
            lxd += dt * (D.bd()[P.go_joint + i] - D.bd()[P.go_joint + 12 + i]); diag += dt * (D.bdd()[P.go_joint + i] + D.bdd()[P.go_joint + 12 + i]);
        }
        if (P.go_height >= 0 && d == 2) { lxd += dt * D.bd()[P.go_height]; diag += dt * D.bdd()[P.go_height]; }
        if (P.go_jspeed >= 0 && d >= 24) {
            const int i = d - 24;
            lxd += dt * (D.bd()[P.go_jspeed + i] - D.bd()[P.go_jspeed + 12 + i]); diag += dt * (D.bdd()[P.go_jspeed + i] + D.bdd()[P.go_jspeed + 12 + i]);
        }
        D.W[d * 36 + d] += diag;
        P.lx[kk * P.rs + d] = lxd;
    })
    HS_PHASE_L(NT, store_image<NT, 1296, 36>(P.lxx + kk * P.rs, tid, [&](int, int r, int c) { return D.W[r * 36 + c]; });)
    LQ_STAMP(5)
    // lu, luu (diag + torque barrier), ly, lyy (grf barrier 3x3 blocks) staged in Cst (288 of 432)
    HS_PHASE_L(NT, for (int i = tid; i < 288; i += NT) L.Jc()[i] = 0.0;)
    HS_PHASE_L(NT, if (tid < 12) {
        const int i = tid;
        double lu = dt * L.wq[36 + i] * (L.u[i] - L.tmp[36 + i]), luu = dt * L.wq[36 + i];
        if (P.go_torque >= 0) { lu += dt * (-D.bd()[P.go_torque + i] + D.bd()[P.go_torque + 12 + i]); luu += dt * (D.bdd()[P.go_torque + i] + D.bdd()[P.go_torque + 12 + i]); }
        P.lu[kk * P.rs + i] = lu; L.Jc()[i + 12 * i] = luu;
        // y: grf pyramid rows [0 0 1; -1 0 mu; 1 0 mu; 0 -1 mu; 0 1 mu] for foot f = i/3
        double ly = 0.0; const int f = i / 3, r = i % 3; int a = -1; for (int t = 0; t < P.nc; t++) if (P.feet[t] == f) a = t;
        if (P.go_grf >= 0 && a >= 0) {
            const double mu = P.mu;
            for (int c = 0; c < 5; c++) {
                const double r0 = (c == 1) ? -1.0 : (c == 2) ? 1.0 : 0.0, r1 = (c == 3) ? -1.0 : (c == 4) ? 1.0 : 0.0, r2 = (c == 0) ? 1.0 : mu;
                const double rr = (r == 0) ? r0 : (r == 1) ? r1 : r2;
                const double hb = dt * D.bdd()[P.go_grf + 5 * a + c] * rr;
                ly += D.bd()[P.go_grf + 5 * a + c] * rr;
                L.Jc()[144 + (3 * f) + 12 * i] += hb * r0; L.Jc()[144 + (3 * f + 1) + 12 * i] += hb * r1; L.Jc()[144 + (3 * f + 2) + 12 * i] += hb * r2;
            }
        }
        P.ly[kk * P.rs + i] = dt * ly;
    })
    HS_PHASE_L(NT, store_image<NT, 144, 12>(P.luu + kk * P.rs, tid, [&](int e, int, int) { return L.Jc()[e]; });
                   store_image<NT, 144, 12>(P.lyy + kk * P.rs, tid, [&](int e, int, int) { return L.Jc()[144 + e]; });)
    LQ_STAMP(6)
}

// Terminal partials of a phase (+ AL) and the reset-map partial Px (next_n x 36, column-major) if a phase follows.
template <int NT>
HD void wb_lq_terminal(WbLqLds& S, PhaseC& P, PhaseC* Pn, const ModelDev& md, int b, int al_active) {
    WbCore& L = S.c; WbDeriv& D = S.d;
    const int h = P.h;
    const size_t kx = ((size_t)b * (h + 1) + h) * 36;
    HS_PHASE(NT, if (tid < 36) L.x[tid] = P.X[kx + tid]; if (tid < 18) { L.acc[tid] = 0.0; L.tau[tid] = 0.0; } if (tid < 12) L.fext[tid] = 0.0;
             wb_cost_prefetch(L, P, h, tid);)
    wb_terms<NT>(L, md, true);
    // d(J v)/dq with psi_kin at (q, v): kinematic lanes only
    wb_dpass<NT>(L, D, md, 0.0, 1.0, 1.0, 0.0, true);
    wb_cost_blocks<NT>(S, P, h, true);
    HS_PHASE(NT, if (tid < 36) {
        const int d = tid;
        double px = P.qf[d] * (L.x[d] - L.tmp[d]);
        px += wb_cost_column(D, L.Jall, L.dvel(), d, D.W + d);
        double diag = P.qf[d];
        if (al_active && P.nt > 0) {   // compute_AL_partials (ConstraintsBase.h:412-425); hx[0:18] = J_foot,z (psi_dyn)
            int t = 0;
            for (int f = 0; f < 4; f++) if (P.td[f]) {
                const double sg = P.sigma[(size_t)b * P.nt + t], lm = P.lambda[(size_t)b * P.nt + t], hh = P.th[(size_t)b * P.nt + t];
                const double hd = d < 18 ? L.Jall[(3 * f + 2) * 18 + d] : 0.0;
                px += (sg * hh + lm) * hd;
                for (int i = 0; i < 18; i++) D.W[i * 36 + d] += (sg * (1 + hh) + lm) * (L.Jall[(3 * f + 2) * 18 + i] * hd);
                t++;
            }
        }
        D.W[d * 36 + d] += diag;
        P.Phix[(size_t)b * 36 + d] = px;
    })
    HS_PHASE(NT, for (int e = tid; e < 1296; e += NT) { const int r = e % 36, c = e / 36; P.Phixx[(size_t)b * 1296 + e] = D.W[r * 36 + c]; })
    if (Pn == nullptr) return;
    const int nn = Pn->n;
    if (!P.has_impact) {
        HS_PHASE(NT, for (int i = tid; i < nn * 36; i += NT) { int r = i % nn, c = i / nn; int src = (nn == 36) ? r : (r < 6 ? r : r + 12); P.Px[(size_t)b * nn * 36 + i] = (src == c) ? 1.0 : 0.0; })
        return;
    }
    // ---- impact partial (WBM.cpp:508-543)
    int tdfeet[4] = {0, 0, 0, 0}; int ntd = 0; for (int f = 0; f < 4; f++) if (P.td[f]) tdfeet[ntd++] = f;
    const int m = 3 * ntd;
    wb_kkt_direct<NT>(L, ntd, Feet4{tdfeet[0], tdfeet[1], tdfeet[2], tdfeet[3]}, 1, 0.0);    // L.qdd = v+, L.lam = impulse_c (compact); factors L (in M), X, Schur factor
    wb_keep_schur<NT>(L, D);
    // pass A: d(M dv)/dq (v = 0, acc = v+ - v, gravity off) on lanes 0..17; d(J^T imp)/dq with the mis-sliced impulse (quirk v)
    HS_PHASE(NT, if (tid < 18) L.acc[tid] = L.qdd[tid] - L.x[18 + tid]; if (tid < 12) L.fext[tid] = 0.0;)
    HS_PHASE(NT, if (tid == 0) { double pad[16]; for (int i = 0; i < 16; i++) pad[i] = i < m ? L.lam[i] : 0.0;
                 for (int i = 0; i < ntd; i++) for (int d = 0; d < 3; d++) L.fext[3 * tdfeet[i] + d] = pad[i + d]; })   // WBM.cpp:454: offset i, not 3i
    wb_dpass<NT>(L, D, md, 0.0, 0.0, 0.0, 0.0, true);
    HS_PHASE(NT, if (tid < 18) for (int i = 0; i < 18; i++) D.W[i * WT + tid] += D.W[i * WT + 36 + tid];)
    // pass B: d(J v+)/dq with psi_kin: kinematic lanes with the velocity replaced by v+  (pre-impact v kept in xb)
    HS_PHASE(NT, if (tid < 18) { L.xb[tid] = L.x[18 + tid]; })
    HS_PHASE(NT, if (tid < 18) { L.x[18 + tid] = L.qdd[tid]; })
    HS_PHASE(NT, if (tid >= 36 && tid < 54) {
        LaneCfg c = lane_cfg(md, true, 0.0, 0.0, 0.0, 1.0, 0.0, -1, tid - 36, -1);
        struct VSink { WbCore* C; int j; HD void tau(int, const Dual&) const {} HD void base(const V3<Dual>&, const V3<Dual>&) const {} HD void foot(int f, const V3<Dual>&, const V3<Dual>& v, const V3<Dual>&) const {
            C->dvel()[(3 * f) * 18 + j] = v.x.d; C->dvel()[(3 * f + 1) * 18 + j] = v.y.d; C->dvel()[(3 * f + 2) * 18 + j] = v.z.d; } } sk{&L, tid - 36};
        wb_pass<Dual>(c, L.x, L.x + 18, L.acc, L.cs, L.sn, L.fext, sk);
    })
    // Px = [I 0; dv+/dq dv+/dv] , dv+/dq = -(K^-1 [dtau_dq; dv_dq])_top ; dv+/dv = (K^-1 [M; 0])_top, where the column M e_j = L L^T e_j
    // enters the solve directly as y = L^T e_j ; staged in W rows (column d per lane)
    HS_PHASE(NT, if (tid < 36) {
        const int d = tid;
        double top[18], bot[12];
        _Pragma("unroll")
        for (int i = 0; i < 18; i++) {
            if (d < 18) top[i] = D.W[wb_pi(i) * WT + d];
            else { const int j = wb_pj(d - 18); top[i] = (i < j) ? L.M[j * 18 + i] : (i == j) ? 1.0 / L.rdM[j] : 0.0; }     // M e_(d-18) in the permuted order = L L^T e_j
        }
        _Pragma("unroll")
        for (int a = 0; a < 12; a++) bot[a] = (d < 18 && a < m) ? L.dvel()[(3 * tdfeet[a / 3] + a % 3) * 18 + d] : 0.0;
        wb_kkt_column(L, D, top, bot, d >= 18, m);
        // results into rows 18..35 of W (rows 0..17 still hold the tangents other lanes read)
        _Pragma("unroll")
        for (int i = 0; i < 18; i++) D.W[WR0 + wb_pi(i) * 36 + d] = (d < 18) ? -top[i] : top[i];
    })
    HS_PHASE(NT, for (int e = tid; e < nn * 36; e += NT) {
        const int r = e % nn, c = e / nn;
        double v;
        if (nn == 36) v = (r < 18) ? ((r == c) ? 1.0 : 0.0) : D.W[WR0 + (r - 18) * 36 + c];
        else v = (r < 6) ? ((r == c) ? 1.0 : 0.0) : D.W[WR0 + (r - 6) * 36 + c];
        P.Px[(size_t)b * nn * 36 + e] = v;
    })
}

}  // namespace hs
