// Device-visible descriptors and per-problem solver state (shared by all kernels and the host driver).
#pragma once
#include "hs_common.hpp"

namespace hs {

constexpr int MAXN = 36, MAXM = 24, MAXP = 12, MAXG = 96;   // MAXG: 24 torque + 24 joint speed + 24 joint + 1 height + 20 GRF
// LQ record layout (doubles) of one knot for a model with dims (N, M, PY): every sub-array starts at a multiple of 256
// doubles so that a 256-thread workgroup streams the record in rounds that each lie inside ONE sub-array.
//   A | lxx | B | C | D | luu | lyy | [lx lu ly]
constexpr int rec_rnd(int x) { return (x + 255) / 256 * 256; }
// Rows of A and B that are DATA.  The whole-body discretisation is A = [I, dt I; A21, A22], B = [0; B2] (WBM.cpp:68, 122-125: forward Euler of
// q' = v): only the lower 18 rows are stored (column-major, 18 rows per column) and the sweeps use the structure (half the inner dimension of
// every product with A or B).  SRB / HKD: dense.
constexpr int rec_arows(int N) { return N == 36 ? 18 : N; }
template <int N, int M, int PY> struct RecLayout {
    static constexpr int AR = rec_arows(N);
    static constexpr int oA = 0, oLxx = oA + rec_rnd(AR * N), oB = oLxx + rec_rnd(N * N), oC = oB + rec_rnd(AR * M), oD = oC + rec_rnd(PY * N),
                         oLuu = oD + rec_rnd(PY * M), oLyy = oLuu + rec_rnd(M * M), oLx = oLyy + rec_rnd(PY * PY), oLu = oLx + N, oLy = oLu + M,
                         size = oLx + rec_rnd(N + M + PY);
    static constexpr int rA = rec_rnd(AR * N) / 256, rQ = rec_rnd(N * N) / 256, rB = rec_rnd(AR * M) / 256, rC = rec_rnd(PY * N) / 256, rD = rec_rnd(PY * M) / 256,
                         rLuu = rec_rnd(M * M) / 256, rLyy = rec_rnd(PY * PY) / 256;
    static constexpr int rounds = rA + rQ + rB + rC + rD + rLuu + rLyy;   // + 1 round for the vectors
};

// Contact-solve cache (whole-body knots): what the LAST rollout of a knot computed at (X[k], U[k]) and the LQ approximation would
// recompute bit for bit at the same point (the reference does recompute it, WBM.cpp:463): the factor of M, X = L^-1 Jc^T, the Schur
// factor, their reciprocal diagonals, qdd, the contact forces, all foot Jacobians, foot positions / velocities.
constexpr int KC_M = 0, KC_X = 324, KC_LG = 540, KC_RDM = 684, KC_RDG = 702, KC_QDD = 714, KC_GRF = 732, KC_LAM = 744, KC_J = 756, KC_FP = 972, KC_FV = 984, KC_SIZE = 1024;

// Per-phase device descriptor.  Trajectory arrays are [batch][count][elems] (problem-major, horizon-major,
// element-contiguous, matrices column-major) so that one wave reads/writes a knot's record with unit stride.
struct PhaseDev {
    int model, n, m, p, h;
    double dt, bg_alpha;
    int contact[4], next_contact[4], td[4], feet[4];
    int nc, n_td, has_impact, next_model, next_n, shooting, is_last;
    double q[MAXN], r[MAXM], qf[MAXN], w_foot_reg[3], w_swing_pos[3], w_swing_vel[3], w_td_vel;
    int c_torque, c_joint, c_minheight, c_grf, c_touchdown, c_jspeed;
    double torque_limit, joint_lb[3], joint_ub[3], h_min, mu, ground_height, jspeed_lb, jspeed_ub;
    double reb_init[5][3];   // torque, joint, minheight, grf, joint speed : delta, delta_min, eps
    double al_init[3];       // sigma, lambda, sigma_max
    int ng, go_torque, go_joint, go_height, go_grf, go_jspeed;   // path-constraint count and group offsets (-1: absent)
    int nt;                                           // terminal constraints (touchdown feet)
    int slot0;                                        // first global slot of this phase (slots = h+1 per phase)
    int nobj; unsigned long long obj_off, obj_sz;     // constraint objects in the order the reference adds their ReB cost (constraint_objects): first constraint / count of object o in bits [8o, 8o+8)
    // reference arrays shared by the batch: (h+1) x width
    const HS_GLOBAL double *xr, *ur, *yr, *foot_pos, *foot_vel, *body_pos;
    const HS_GLOBAL int* ref_contact;
    // whole-body phases: the references of knot k in ONE record of 80 doubles (one base pointer for the rollout knot's first reads):
    // [0,36) xr | [36,48) ur | [48,60) foot_vel | [60,64) ref_contact as doubles | [64,76) foot_pos - body_pos | [76,80) pad
    const HS_GLOBAL double* rref;
    // trajectories
    HS_GLOBAL double *X, *Xbar, *Xsim, *Defect, *Defect_bar, *dX, *G;           // (h+1) x n
    HS_GLOBAL double *U, *Ubar, *dU, *Qu;                                        // h x m
    HS_GLOBAL double *KdX;                                                       // h x m: K[k] dX[k] as the last linear rollout formed it (the whole-body rollout knot takes it instead of re-reading K)
    HS_GLOBAL double *Y;                                                         // h x p
    HS_GLOBAL double *K, *Qux, *Quu;                                             // h x (m*n), h x (m*m)
    HS_GLOBAL double *A, *B, *C, *D;                                             // h x ...
    // LQ record of a knot: ONE contiguous block per (problem, knot) of `rs` doubles holding A | lxx | B | C | D | luu | lyy |
    // lx lu ly, each sub-array starting at a multiple of 256 doubles (REC_* offsets), so that the Riccati workgroup
    // streams a knot with one base pointer and unit stride.  A, lxx, ... below point INTO rec: element e of knot kk is P.A[kk*rs + e].
    HS_GLOBAL double* rec; int rs;
    // fp32 handles (hsddp_create_ex, HSDDP_PREC_F32): the record holds floats (same element offsets), rec is null and the A ... ly pointers
    // below are unset; knot programs store through rec_put, the sweep reads rec32
    HS_GLOBAL float* rec32;
    int oA, oLxx, oB, oC, oD, oLuu, oLyy, oLx, oLu, oLy;                         // element offsets of the sub-arrays inside a knot's record
    HS_GLOBAL double *l, *lbase, *lx, *lu, *ly, *lxx, *luu, *lyy;                // running cost data (lux == 0 for every shipped cost)
    HS_GLOBAL double *Phi, *Phibase, *Phix, *Phixx, *H0, *Px;                    // per problem: 1, 1, n, n*n, n*n, next_n*n
    HS_GLOBAL double *g, *delta, *eps;                                           // h x ng
    HS_GLOBAL double* kc;                                                        // whole-body phases: contact-solve cache of the last rollout, h x KC_SIZE
    HS_GLOBAL double *th, *sigma, *lambda;                                       // nt
};

// store into the LQ record of knot kk in the precision the handle keeps it in
template <class PD> HD void rec_put(PD& P, size_t kk, int off, double v) {
    if (P.rec32 != nullptr) P.rec32[kk * P.rs + off] = (float)v; else P.rec[kk * P.rs + off] = v;
}

// how device code sees a descriptor: constant memory (scalar loads, values survive memory clobbers)
using PhaseC = const HS_CONST PhaseDev;

// ReB parameter group of path constraint c (index into reb_init)
HDH int constraint_group(const PhaseDev& P, int c) {
    if (P.go_torque >= 0 && c >= P.go_torque && c < P.go_torque + 24) return 0;
    if (P.go_jspeed >= 0 && c >= P.go_jspeed && c < P.go_jspeed + 24) return 4;
    if (P.go_joint >= 0 && c >= P.go_joint && c < P.go_joint + 24) return 1;
    if (P.go_height >= 0 && c == P.go_height) return 2;
    return 3;
}
// constraint objects of a phase in the order the reference adds them (one `l += dt * ReB_cost` each, SinglePhase.cpp:394-402)
HDH int constraint_objects(const PhaseDev& P, int* offs, int* sz) {
    int n = 0;
    if (P.go_torque >= 0) { offs[n] = P.go_torque; sz[n++] = 24; }
    if (P.go_jspeed >= 0) { offs[n] = P.go_jspeed; sz[n++] = 24; }
    if (P.go_joint >= 0) { offs[n] = P.go_joint; sz[n++] = 24; }
    if (P.go_height >= 0) { offs[n] = P.go_height; sz[n++] = 1; }
    if (P.go_grf >= 0) { offs[n] = P.go_grf; sz[n++] = 5 * P.nc; }
    return n;
}

// control flags of the per-problem state machine (MultiPhaseDDP::solve as masks over the batch)
struct ProbState {
    double actual_cost, merit, feas, dV_1, dV_2, merit_rho;
    double cost_prev, merit_prev, feas_prev, reg;
    double max_tconstr, max_pconstr, max_tconstr_prev, max_pconstr_prev;
    double info_tconstr, info_pconstr;   // last buffered values (eqn/ineq_feas_buffer.back())
    double ls_eps;
    int iter, ls_total, reg_total, status;
    int outer_active, inner_active, ls_active, ls_success, rollout_ok, bs_ok;
    int iter_in, iter_ou;
    int hist_n, push_pending;            // entries in the history buffers; an entry waits for the next control step (see k_eval)
    int need_commit, pad_;               // batched line search: the trajectories of step ls_eps still have to be written (k_ls_pick)
};

struct OptDev {
    double alpha, gamma, update_penalty, update_relax, update_regularization, update_ReB;
    int max_DDP_iter, max_AL_iter;
    double cost_thresh, tconstr_thresh, pconstr_thresh, dynamics_feas_thresh, merit_scale, merit_offset;
    int AL_active, ReB_active, MS;
};

struct ModelDev { double cpsi_dyn, spsi_dyn, cpsi_kin, spsi_kin; };

}  // namespace hs
