// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked/imported by the product path (cafe-mpc_amd/).
//
// CPU restatement of the reference's HS-DDP solve path, function by function:
//   MultiPhaseDDP<T>::solve & friends      HSDDPSolver/source/MultiPhaseDDP.cpp:12-548
//   SinglePhase<T,xs,us,ys>                HSDDPSolver/source/SinglePhase.cpp:145-450
//   Trajectory::compute_defect / feasibility / update_nominal_vals   TrajectoryManagement.cpp:122-259
//   PathConstraintBase / TerminalConstraintBase (ReB, AL)            header/ConstraintsBase.h:194-425
//   QuadraticTrackingCost / CostContainer  source/SinglePhaseInterface.cpp:21-181
//   WB costs / constraints / reset         MHPC/MHPC-Trajopt/MHPCCost.cpp, MHPCConstraint.cpp, MHPCReset.cpp
// The reference takes std::function closures; this restatement takes the POD phase descriptors of
// include/hsddp.h (what those closures capture).  Parity status: the solver algebra has NO golden
// data in the reference's own tests ("parity unpinned by fixtures" — SURVEY 8c); the WB dynamics is
// pinned by testKKTDynamics.cpp golden vectors and the CasADi kinematic functions (tests/).
#pragma once
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <omp.h>
#include <cstdlib>
#include "hsddp.h"
#include "linalg.hpp"
#include "wbm.hpp"
#include "srbm.hpp"
#include "hkdm.hpp"

namespace orc {

inline void model_dims(int model, int& n, int& m, int& p) {
    if (model == HSDDP_MODEL_WB) { n = 36; m = 12; p = 12; }
    else if (model == HSDDP_MODEL_SRB) { n = 12; m = 12; p = 0; }
    else { n = 24; m = 24; p = 0; }
}

// one scalar linear inequality g = sum coef*z[idx] + b >= 0 over z in {x,u,y}
struct LinCon { int kind; int nnz; int idx[3]; double coef[3]; double b; };
// a PathConstraintBase object = contiguous group of scalar constraints sharing initial ReB params
struct ConGroup { int first, size; hsddp_reb_t init; };

struct PhaseDef {
    hsddp_phase_desc_t d;
    int n, m, p, h;
    std::vector<double> xr, ur, yr, foot_pos, foot_vel, body_pos;
    std::vector<int> ref_contact;
    std::vector<LinCon> cons; std::vector<ConGroup> groups;
    int td[4]; int n_td;          // touchdown status (MHPCProblem.cpp:568-574)
    bool has_impact;
};

struct Traj {  // Trajectory<T,xs,us,ys> (TrajectoryManagement.h:54-84) + constraint data/params
    std::vector<double> X, Xbar, Xsim, Defect, Defect_bar, dX, G, H;
    std::vector<double> U, Ubar, dU, Qu, Y, K, Qux, Quu, A, B, C, D;
    std::vector<double> l, lx, lu, ly, lxx, lux, luu, lyy;
    double Phi; std::vector<double> Phix, Phixx;
    std::vector<double> g, delta, eps;                 // h x ng
    std::vector<double> th, thx, sigma, lambda;        // terminal constraints
    double max_pviol, max_tviol;
    double x_init[36], dx_init[36];
    double dV_1, dV_2, actual_cost;
    bool shooting;
};

struct Problem {
    std::vector<Traj> tr;
    double x0[36];
    double actual_cost = 0, merit = 0, feas = 0, dV_1 = 0, dV_2 = 0;
    double max_tconstr = 0, max_pconstr = 0, max_tconstr_prev = 0, max_pconstr_prev = 0, merit_rho = 0;
    int iter_ = 0, ls_iter_total_ = 0, reg_iter_total_ = 0, status = 0;
    float solve_time_ = 0;
    std::vector<float> cost_buffer, dyn_feas_buffer, eqn_feas_buffer, ineq_feas_buffer;
};

struct Solver {
    std::vector<PhaseDef> ph;
    std::vector<Problem> pb;
    WbParams wp;
    int batch = 0;
    int lq_threads = 1;      // OpenMP threads over knots inside LQ_approximation (SinglePhase.cpp:277)
    int problem_threads = 1; // independent problems spread over cores (throughput-shaped CPU baseline)
    float solve_ms = 0;
    bool trace = getenv("HSDDP_ORACLE_TRACE") != nullptr;

    // ------------------------------------------------------------------ setup
    void build_constraints(PhaseDef& P) {
        const hsddp_phase_desc_t& d = P.d;
        auto add_group = [&](int first, hsddp_reb_t r) { P.groups.push_back({first, (int)P.cons.size() - first, r}); };
        if (P.d.model == HSDDP_MODEL_WB) {
            if (d.c_torque) {  // TorqueLimit (MHPCConstraint.cpp:77-112): C=[-I; I], b=-limit
                int f = P.cons.size();
                for (int i = 0; i < 12; i++) P.cons.push_back({1, 1, {i, 0, 0}, {-1, 0, 0}, d.torque_limit});
                for (int i = 0; i < 12; i++) P.cons.push_back({1, 1, {i, 0, 0}, {1, 0, 0}, d.torque_limit});
                add_group(f, d.reb_torque);
            }
            if (d.c_jointspeed) {   // BarrelRoll::JointSpeedLimit (BarrelRoll/BarrelRollConstraints.cpp:143-186), added right after the torque limit
                int f = P.cons.size();
                for (int i = 0; i < 12; i++) P.cons.push_back({0, 1, {24 + i, 0, 0}, {1, 0, 0}, -d.jointspeed_lb});
                for (int i = 0; i < 12; i++) P.cons.push_back({0, 1, {24 + i, 0, 0}, {-1, 0, 0}, d.jointspeed_ub});
                add_group(f, d.reb_jointspeed);
            }
            if (d.c_joint) {   // JointLimit (MHPCConstraint.cpp:163-204)
                int f = P.cons.size();
                for (int i = 0; i < 12; i++) P.cons.push_back({0, 1, {6 + i, 0, 0}, {1, 0, 0}, -d.joint_lb[i % 3]});
                for (int i = 0; i < 12; i++) P.cons.push_back({0, 1, {6 + i, 0, 0}, {-1, 0, 0}, d.joint_ub[i % 3]});
                add_group(f, d.reb_joint);
            }
            if (d.c_minheight) { int f = P.cons.size(); P.cons.push_back({0, 1, {2, 0, 0}, {1, 0, 0}, -d.h_min}); add_group(f, d.reb_minheight); }
            bool any = false; for (int l = 0; l < 4; l++) any |= d.contact[l] == 1;
            if (d.c_grf && any) {  // WBGRF (MHPCConstraint.cpp:9-70)
                int f = P.cons.size(); double mu = d.mu;
                for (int l = 0; l < 4; l++) if (d.contact[l] > 0) {
                    P.cons.push_back({2, 1, {3 * l + 2, 0, 0}, {1, 0, 0}, 0});
                    P.cons.push_back({2, 2, {3 * l, 3 * l + 2, 0}, {-1, mu, 0}, 0});
                    P.cons.push_back({2, 2, {3 * l, 3 * l + 2, 0}, {1, mu, 0}, 0});
                    P.cons.push_back({2, 2, {3 * l + 1, 3 * l + 2, 0}, {-1, mu, 0}, 0});
                    P.cons.push_back({2, 2, {3 * l + 1, 3 * l + 2, 0}, {1, mu, 0}, 0});
                }
                add_group(f, d.reb_grf);
            }
        } else if (P.d.model == HSDDP_MODEL_SRB) {
            if (d.c_minheight) { int f = P.cons.size(); P.cons.push_back({0, 1, {2, 0, 0}, {1, 0, 0}, -d.h_min}); add_group(f, d.reb_minheight); }
        } else if (P.d.model == HSDDP_MODEL_HKD) {
            bool any = false; for (int l = 0; l < 4; l++) any |= d.contact[l] == 1;
            if (d.c_grf && any) {   // GRFConstraint on u[0:12] (HKDConstraints.cpp:7-63), legs FR FL HR HL
                int f = P.cons.size(); double mu = d.mu;
                for (int l = 0; l < 4; l++) if (d.contact[l] > 0) {
                    P.cons.push_back({1, 1, {3 * l + 2, 0, 0}, {1, 0, 0}, 0});
                    P.cons.push_back({1, 2, {3 * l, 3 * l + 2, 0}, {-1, mu, 0}, 0});
                    P.cons.push_back({1, 2, {3 * l, 3 * l + 2, 0}, {1, mu, 0}, 0});
                    P.cons.push_back({1, 2, {3 * l + 1, 3 * l + 2, 0}, {-1, mu, 0}, 0});
                    P.cons.push_back({1, 2, {3 * l + 1, 3 * l + 2, 0}, {1, mu, 0}, 0});
                }
                add_group(f, d.reb_grf);
            }
        }
        P.n_td = 0; P.has_impact = false;
        for (int l = 0; l < 4; l++) { P.td[l] = (d.contact[l] == 0 && d.next_contact[l] == 1) ? 1 : 0; P.n_td += P.td[l]; }
        P.has_impact = P.n_td > 0;
    }

    int create(int n_phases, const hsddp_phase_desc_t* phases, const hsddp_model_param_t* mp, int batch_) {
        batch = batch_;
        if (mp) { wp.psi_dyn = mp->psi_dyn; wp.psi_kin = mp->psi_kin; }
        ph.resize(n_phases);
        for (int i = 0; i < n_phases; i++) {
            PhaseDef& P = ph[i]; P.d = phases[i];
            model_dims(P.d.model, P.n, P.m, P.p); P.h = P.d.horizon;
            int h1 = P.h + 1;
            auto cp = [&](const auto* src, int w, std::vector<double>& dst) { dst.assign((size_t)h1 * w, 0.0); if (src && w) std::copy(src, src + (size_t)h1 * w, dst.begin()); };   // (element-wise: the ABI side is always fp64, see ORC_LONG_DOUBLE)
            cp(P.d.xr, P.n, P.xr); cp(P.d.ur, P.m, P.ur); cp(P.d.yr, P.p, P.yr);
            cp(P.d.foot_pos, 12, P.foot_pos); cp(P.d.foot_vel, 12, P.foot_vel); cp(P.d.body_pos, 3, P.body_pos);
            P.ref_contact.assign((size_t)h1 * 4, 0);
            if (P.d.ref_contact) std::memcpy(P.ref_contact.data(), P.d.ref_contact, sizeof(int) * h1 * 4);
            P.d.xr = P.d.ur = P.d.yr = P.d.foot_pos = P.d.foot_vel = P.d.body_pos = nullptr; P.d.ref_contact = nullptr;
            build_constraints(P);
        }
        pb.resize(batch);
        for (auto& q : pb) {
            q.tr.resize(n_phases);
            for (int i = 0; i < n_phases; i++) alloc_traj(ph[i], q.tr[i]);
        }
        return HSDDP_OK;
    }
    void alloc_traj(const PhaseDef& P, Traj& T) {
        int n = P.n, m = P.m, p = P.p, h = P.h, h1 = h + 1;
        auto z = [](std::vector<double>& v, size_t s) { v.assign(s, 0.0); };
        z(T.X, h1 * n); z(T.Xbar, h1 * n); z(T.Xsim, h1 * n); z(T.Defect, h1 * n); z(T.Defect_bar, h1 * n); z(T.dX, h1 * n);
        z(T.G, h1 * n); z(T.H, (size_t)h1 * n * n);
        z(T.U, h * m); z(T.Ubar, h * m); z(T.dU, h * m); z(T.Qu, h * m); z(T.Y, h * p); z(T.K, (size_t)h1 * m * n);
        z(T.Qux, (size_t)h * m * n); z(T.Quu, (size_t)h * m * m);
        z(T.A, (size_t)h * n * n); z(T.B, (size_t)h * n * m); z(T.C, (size_t)h * p * n); z(T.D, (size_t)h * p * m);
        z(T.l, h); z(T.lx, h * n); z(T.lu, h * m); z(T.ly, h * p); z(T.lxx, (size_t)h * n * n); z(T.lux, (size_t)h * m * n);
        z(T.luu, (size_t)h * m * m); z(T.lyy, (size_t)h * p * p);
        T.Phi = 0; z(T.Phix, n); z(T.Phixx, n * n);
        int ng = P.cons.size();
        z(T.g, (size_t)h * ng); z(T.delta, (size_t)h * ng); z(T.eps, (size_t)h * ng);
        for (const auto& gr : P.groups) for (int k = 0; k < h; k++) for (int i = 0; i < gr.size; i++) {
            T.delta[(size_t)k * ng + gr.first + i] = gr.init.delta; T.eps[(size_t)k * ng + gr.first + i] = gr.init.eps;
        }
        int nt = (P.d.c_touchdown && P.d.model != HSDDP_MODEL_SRB) ? P.n_td : 0;
        z(T.th, nt); z(T.thx, (size_t)nt * n); T.sigma.assign(nt, P.d.al_td.sigma); T.lambda.assign(nt, P.d.al_td.lambda);
        T.max_pviol = T.max_tviol = 0; std::memset(T.x_init, 0, sizeof(T.x_init)); std::memset(T.dx_init, 0, sizeof(T.dx_init));
        T.dV_1 = T.dV_2 = T.actual_cost = 0; T.shooting = P.d.shooting != 0;
    }

    // ------------------------------------------------------------------ models
    void dynamics(const PhaseDef& P, int k, const double* x, const double* u, double* xnext, double* y) const {
        if (P.d.model == HSDDP_MODEL_WB) { WbParams w = wp; w.bg_alpha = P.d.BG_alpha; wb_dynamics(w, x, u, P.d.contact, P.d.dt, xnext, y); }
        else if (P.d.model == HSDDP_MODEL_HKD) hkd_dynamics(x, u, P.d.contact, P.d.dt, xnext);
        else srb_dynamics(x, u, &P.foot_pos[(size_t)k * 12], &P.ref_contact[(size_t)k * 4], P.d.dt, xnext);
    }
    void dynamics_partial(const PhaseDef& P, int k, const double* x, const double* u, double* A, double* B, double* C, double* D) const {
        if (P.d.model == HSDDP_MODEL_WB) { WbParams w = wp; w.bg_alpha = P.d.BG_alpha; wb_dynamics_partial(w, x, u, P.d.contact, P.d.dt, A, B, C, D); }
        else if (P.d.model == HSDDP_MODEL_HKD) hkd_dynamics_partial(x, u, P.d.contact, P.d.dt, A, B);
        else srb_dynamics_partial(x, u, &P.foot_pos[(size_t)k * 12], &P.ref_contact[(size_t)k * 4], P.d.dt, A, B);
    }
    // MHPCReset::reset_map (MHPCReset.cpp:4-28); returns dim of xnext
    int resetmap(const PhaseDef& P, const double* x, double* xnext) const {
        if (P.d.model == HSDDP_MODEL_HKD) { hkd_resetmap(x, P.d.contact, P.d.next_contact, wp.psi_kin, xnext); return 24; }   // HKDReset.h:41-76
        double tmp[36];
        if (P.d.model == HSDDP_MODEL_WB && P.has_impact) wb_impact(wp, x, P.d.contact, P.d.next_contact, tmp);
        else std::memcpy(tmp, x, sizeof(double) * P.n);
        if (P.d.model == HSDDP_MODEL_WB && P.d.next_model == HSDDP_MODEL_SRB) {
            for (int i = 0; i < 6; i++) { xnext[i] = tmp[i]; xnext[6 + i] = tmp[18 + i]; }
            return 12;
        }
        std::memcpy(xnext, tmp, sizeof(double) * P.n);
        return P.n;
    }
    // Px: n_next x n column-major (MHPCReset.cpp:31-52)
    int resetmap_partial(const PhaseDef& P, const double* x, double* Px) const {
        if (P.d.model == HSDDP_MODEL_HKD) { hkd_resetmap_partial(x, P.d.contact, P.d.next_contact, wp.psi_kin, Px); return 24; }   // HKDReset.h:78-136
        int n = P.n;
        std::vector<double> full((size_t)n * n, 0.0);
        if (P.d.model == HSDDP_MODEL_WB && P.has_impact) wb_impact_partial(wp, x, P.d.contact, P.d.next_contact, full.data());
        else for (int i = 0; i < n; i++) full[i + n * i] = 1.0;
        if (P.d.model == HSDDP_MODEL_WB && P.d.next_model == HSDDP_MODEL_SRB) {
            for (int c = 0; c < 36; c++) for (int i = 0; i < 6; i++) { Px[i + 12 * c] = full[i + 36 * c]; Px[6 + i + 12 * c] = full[18 + i + 36 * c]; }
            return 12;
        }
        std::memcpy(Px, full.data(), sizeof(double) * n * n);
        return n;
    }

    // ------------------------------------------------------------------ constraints
    // ConstraintContainer::compute_path_constraints (ConstraintsBase.h:470) + update_max_violation (:217)
    void path_constraints(const PhaseDef& P, Traj& T, int k) const {
        int ng = P.cons.size(); const double* x = &T.X[(size_t)k * P.n]; const double* u = &T.U[(size_t)k * P.m]; const double* y = P.p ? &T.Y[(size_t)k * P.p] : nullptr;
        if (k == 0) T.max_pviol = 0;
        double mk = 0;
        for (int i = 0; i < ng; i++) {
            const LinCon& c = P.cons[i]; const double* z = c.kind == 0 ? x : c.kind == 1 ? u : y;
            double g = 0; for (int t = 0; t < c.nnz; t++) g += c.coef[t] * z[c.idx[t]];
            g += c.b;   // reference: A.row(i)*z - b
            T.g[(size_t)k * ng + i] = g; mk = std::min(mk, g);
        }
        T.max_pviol = std::min(T.max_pviol, mk);
    }
    // WBTouchDown::compute_violation (MHPCConstraint.cpp:254-280)
    void terminal_constraints(const PhaseDef& P, Traj& T) const {
        int nt = T.th.size(); T.max_tviol = 0;
        if (!nt) return;
        if (P.d.model == HSDDP_MODEL_HKD) {   // TouchDownConstraint::compute_violation (HKDConstraints.cpp:75-111)
            const double* x = &T.X[(size_t)P.h * P.n]; int i = 0;
            for (int l = 0; l < 4; l++) if (P.td[l]) {
                double pf[3], J[3][9]; hkd_foot_jac(x + 3, x, x + 12 + 3 * l, l, wp.psi_kin, pf, J);
                T.th[i] = pf[2] - P.d.ground_height; T.max_tviol = std::max(T.max_tviol, std::fabs(T.th[i])); i++;
            }
            return;
        }
        WbFootKin F; WbParams w = wp; wb_foot_kin(w, &T.X[(size_t)P.h * P.n], F, false);
        int i = 0;
        for (int l = 0; l < 4; l++) if (P.td[l]) { T.th[i] = F.pos[l][2] - P.d.ground_height; T.max_tviol = std::max(T.max_tviol, std::fabs(T.th[i])); i++; }
    }

    // ------------------------------------------------------------------ costs
    // CostContainer::running_cost (SinglePhaseInterface.cpp:136-146) for the cost list of
    // MHPCProblem::create_problem_one_phase (MHPCProblem.cpp:425-437); returns l (zeroes the partials: quirk ii)
    void running_cost(const PhaseDef& P, Traj& T, int k, const WbFootKin* F) const {
        int n = P.n, m = P.m, p = P.p; double dt = P.d.dt;
        const double* x = &T.X[(size_t)k * n]; const double* u = &T.U[(size_t)k * m];
        std::fill_n(&T.lx[(size_t)k * n], n, 0.0); std::fill_n(&T.lu[(size_t)k * m], m, 0.0); if (p) std::fill_n(&T.ly[(size_t)k * p], p, 0.0);
        std::fill_n(&T.lxx[(size_t)k * n * n], n * n, 0.0); std::fill_n(&T.lux[(size_t)k * m * n], m * n, 0.0);
        std::fill_n(&T.luu[(size_t)k * m * m], m * m, 0.0); if (p) std::fill_n(&T.lyy[(size_t)k * p * p], p * p, 0.0);
        double l = 0;
        {   // QuadraticTrackingCost::running_cost (SinglePhaseInterface.cpp:70-85, 21-33); S = 0
            double lq = 0, lr = 0;
            for (int i = 0; i < n; i++) { double dx = x[i] - P.xr[(size_t)k * n + i]; lq += dx * P.d.q[i] * dx; }
            for (int i = 0; i < m; i++) { double du = u[i] - P.ur[(size_t)k * m + i]; lr += du * P.d.r[i] * du; }
            double t = 0.5 * lq; t += 0.5 * lr; t += 0.0; t *= dt; l += t;
        }
        if (P.d.model == HSDDP_MODEL_WB) {
            const int* rc = &P.ref_contact[(size_t)k * 4]; const double* fp = &P.foot_pos[(size_t)k * 12]; const double* bp = &P.body_pos[(size_t)k * 3];
            double l2 = 0, l3 = 0, l4 = 0;
            for (int f = 0; f < 4; f++) {
                double d[3]; for (int a = 0; a < 3; a++) d[a] = (F->pos[f][a] - x[a]) - (fp[3 * f + a] - bp[a]);
                if (rc[f] > 0 && P.d.w_foot_reg[0] >= 0) { double s = 0; for (int a = 0; a < 3; a++) s += d[a] * P.d.w_foot_reg[a] * d[a]; l2 += 0.5 * s * dt; }
                if (rc[f] == 0 && P.d.w_swing_pos[0] >= 0) { double s = 0; for (int a = 0; a < 3; a++) s += d[a] * P.d.w_swing_pos[a] * d[a]; l3 += 0.5 * s * dt; }
                if (rc[f] == 0 && P.d.w_swing_vel[0] >= 0) { double s = 0; for (int a = 0; a < 3; a++) { double dv = F->vel[f][a] - P.foot_vel[(size_t)k * 12 + 3 * f + a]; s += dv * P.d.w_swing_vel[a] * dv; } l4 += 0.5 * s * dt; }
            }
            l += l2; l += l3; l += l4;
        } else if (P.d.model == HSDDP_MODEL_HKD && P.d.w_foot_reg[0] >= 0) {   // HKDFootPlaceReg::running_cost (HKDCost.cpp:4-19): Qfoot = diag(c_l * w)
            const double* fp = &P.foot_pos[(size_t)k * 12]; const double* bp = &P.body_pos[(size_t)k * 3];
            double s = 0;
            for (int f = 0; f < 4; f++) for (int a = 0; a < 3; a++) { double d = (x[12 + 3 * f + a] - x[3 + a]) - (fp[3 * f + a] - bp[a]); s += d * (P.d.contact[f] * P.d.w_foot_reg[a]) * d; }
            double t = .5 * s; t *= dt; l += t;
        }
        T.l[k] = l;
    }
    // PathConstraintBase::compute_ReB_cost (ConstraintsBase.h:230-248) via SinglePhase.cpp:394-402
    void add_reb_cost(const PhaseDef& P, Traj& T, int k) const {
        int ng = P.cons.size();
        for (const auto& gr : P.groups) {
            double c = 0;
            for (int i = gr.first; i < gr.first + gr.size; i++) {
                double g = T.g[(size_t)k * ng + i], delta = T.delta[(size_t)k * ng + i], eps = T.eps[(size_t)k * ng + i], barr;
                if (g > delta) barr = -std::log(g);
                else { barr = .5 * (((g - 2 * delta) / delta) * ((g - 2 * delta) / delta) - 1); barr -= std::log(delta); }
                c += eps * barr;
            }
            T.l[k] += P.d.dt * c;
        }
    }
    void terminal_cost(const PhaseDef& P, Traj& T, const WbFootKin* F) const {
        int n = P.n, h = P.h; const double* x = &T.X[(size_t)h * n];
        std::fill(T.Phix.begin(), T.Phix.end(), 0.0); std::fill(T.Phixx.begin(), T.Phixx.end(), 0.0);
        double Phi = 0;
        { double s = 0; for (int i = 0; i < n; i++) { double dx = x[i] - P.xr[(size_t)h * n + i]; s += dx * P.d.qf[i] * dx; } Phi += 0.5 * s; }
        if (P.d.model == HSDDP_MODEL_WB) {
            const int* rc = &P.ref_contact[(size_t)h * 4]; const double* fp = &P.foot_pos[(size_t)h * 12]; const double* bp = &P.body_pos[(size_t)h * 3];
            double l2 = 0, l5 = 0;
            for (int f = 0; f < 4; f++) {
                if (rc[f] > 0 && P.d.w_foot_reg[0] >= 0) {  // WBFootPlaceReg::terminal_cost (MHPCCost.cpp:67-87): value x1
                    double s = 0; for (int a = 0; a < 3; a++) { double d = (F->pos[f][a] - x[a]) - (fp[3 * f + a] - bp[a]); s += d * P.d.w_foot_reg[a] * d; } l2 += 0.5 * s;
                }
                if (P.td[f] && P.n_td > 0 && P.d.w_td_vel >= 0) { double vz = F->vel[f][2]; l5 += 0.5 * vz * P.d.w_td_vel * vz; }  // TDVelocityPenalty (MHPCCost.cpp:255-268)
            }
            Phi += l2; Phi += l5;
        } else if (P.d.model == HSDDP_MODEL_HKD && P.d.w_foot_reg[0] >= 0) {   // HKDFootPlaceReg::terminal_cost (HKDCost.cpp:37-49): 10 d'Qd
            const double* fp = &P.foot_pos[(size_t)h * 12]; const double* bp = &P.body_pos[(size_t)h * 3];
            double s = 0;
            for (int f = 0; f < 4; f++) for (int a = 0; a < 3; a++) { double d = (x[12 + 3 * f + a] - x[3 + a]) - (fp[3 * f + a] - bp[a]); s += d * (P.d.contact[f] * P.d.w_foot_reg[a]) * d; }
            Phi += 10 * s;
        }
        T.Phi = Phi;
    }
    // SinglePhase::compute_cost (SinglePhase.cpp:236-262)
    void compute_cost_phase(const PhaseDef& P, Traj& T, const hsddp_option_t& opt) const {
        T.actual_cost = 0; WbFootKin F;
        for (int k = 0; k < P.h; k++) {
            if (P.d.model == HSDDP_MODEL_WB) wb_foot_kin(wp, &T.X[(size_t)k * P.n], F, false);
            running_cost(P, T, k, &F);
            if (opt.ReB_active) add_reb_cost(P, T, k);
            T.actual_cost += T.l[k];
        }
        if (P.d.model == HSDDP_MODEL_WB) wb_foot_kin(wp, &T.X[(size_t)P.h * P.n], F, false);
        terminal_cost(P, T, &F);
        if (opt.AL_active) {  // TerminalConstraintBase::compute_AL_cost (ConstraintsBase.h:400-411)
            double c = 0; for (size_t i = 0; i < T.th.size(); i++) { c += 0.5 * T.sigma[i] * T.th[i] * T.th[i]; c += T.lambda[i] * T.th[i]; }
            if (!T.th.empty()) T.Phi += c;
        }
        T.actual_cost += T.Phi;
    }

    // cost + constraint partials of one knot, accumulated on top of rcostData (quirk ii)
    void running_cost_par(const PhaseDef& P, Traj& T, int k, const hsddp_option_t& opt) const {
        int n = P.n, m = P.m, p = P.p; double dt = P.d.dt;
        const double* x = &T.X[(size_t)k * n]; const double* u = &T.U[(size_t)k * m];
        double* lx = &T.lx[(size_t)k * n]; double* lu = &T.lu[(size_t)k * m]; double* ly = p ? &T.ly[(size_t)k * p] : nullptr;
        double* lxx = &T.lxx[(size_t)k * n * n]; double* luu = &T.luu[(size_t)k * m * m]; double* lyy = p ? &T.lyy[(size_t)k * p * p] : nullptr;
        for (int i = 0; i < n; i++) { lx[i] += dt * P.d.q[i] * (x[i] - P.xr[(size_t)k * n + i]); lxx[i + n * i] += dt * P.d.q[i]; }
        for (int i = 0; i < m; i++) { lu[i] += dt * P.d.r[i] * (u[i] - P.ur[(size_t)k * m + i]); luu[i + m * i] += dt * P.d.r[i]; }
        if (P.d.model == HSDDP_MODEL_WB) {
            WbFootKin F; wb_foot_kin(wp, x, F, true);
            const int* rc = &P.ref_contact[(size_t)k * 4]; const double* fp = &P.foot_pos[(size_t)k * 12]; const double* bp = &P.body_pos[(size_t)k * 3];
            // three cost objects, each accumulated separately then added (CostContainer::running_cost_par)
            for (int pass = 0; pass < 3; pass++) {
                std::vector<double> tx(n, 0.0), txx((size_t)n * n, 0.0);
                for (int f = 0; f < 4; f++) {
                    if (pass < 2) {
                        const auto* w = pass == 0 ? P.d.w_foot_reg : P.d.w_swing_pos;
                        bool on = (pass == 0 ? rc[f] > 0 : rc[f] == 0) && w[0] >= 0;
                        if (!on) continue;
                        double d[3]; for (int a = 0; a < 3; a++) d[a] = (F.pos[f][a] - x[a]) - (fp[3 * f + a] - bp[a]);
                        double J[3][18]; for (int a = 0; a < 3; a++) for (int j = 0; j < 18; j++) J[a][j] = j < 3 ? 0.0 : F.J[f][a][j];  // leftCols<3>().setZero()
                        for (int i = 0; i < 18; i++) { double s = 0; for (int a = 0; a < 3; a++) s += J[a][i] * w[a] * d[a]; tx[i] += s * dt; }
                        for (int j = 0; j < 18; j++) for (int i = 0; i < 18; i++) { double s = 0; for (int a = 0; a < 3; a++) s += J[a][i] * w[a] * J[a][j]; txx[i + n * j] += s * dt; }
                    } else {
                        const auto* w = P.d.w_swing_vel;
                        if (!(rc[f] == 0 && w[0] >= 0)) continue;
                        double dv[3]; for (int a = 0; a < 3; a++) dv[a] = F.vel[f][a] - P.foot_vel[(size_t)k * 12 + 3 * f + a];
                        double J[3][36]; for (int a = 0; a < 3; a++) for (int j = 0; j < 18; j++) { J[a][j] = F.Jv[f][a][j]; J[a][18 + j] = F.J[f][a][j]; }
                        for (int i = 0; i < 36; i++) { double s = 0; for (int a = 0; a < 3; a++) s += J[a][i] * w[a] * dv[a]; tx[i] += s * dt; }
                        for (int j = 0; j < 36; j++) for (int i = 0; i < 36; i++) { double s = 0; for (int a = 0; a < 3; a++) s += J[a][i] * w[a] * J[a][j]; txx[i + n * j] += s * dt; }
                    }
                }
                for (int i = 0; i < n; i++) lx[i] += tx[i];
                for (int i = 0; i < n * n; i++) lxx[i] += txx[i];
            }
        }
        if (P.d.model == HSDDP_MODEL_HKD && P.d.w_foot_reg[0] >= 0) hkd_footreg_par(P, x, &P.foot_pos[(size_t)k * 12], &P.body_pos[(size_t)k * 3], dt, lx, lxx);
        if (opt.ReB_active) {  // compute_ReB_partials (ConstraintsBase.h:250-289) + SinglePhase.cpp:404-418
            int ng = P.cons.size();
            for (const auto& gr : P.groups) {
                std::vector<double> gx(n, 0.0), gu(m, 0.0), gy(p, 0.0), hx((size_t)n * n, 0.0), hu((size_t)m * m, 0.0), hy((size_t)p * p, 0.0);
                for (int i = gr.first; i < gr.first + gr.size; i++) {
                    const LinCon& c = P.cons[i];
                    double g = T.g[(size_t)k * ng + i], delta = T.delta[(size_t)k * ng + i], eps = T.eps[(size_t)k * ng + i], bd, bdd;
                    if (g > delta) { bd = -1.0 / g; bdd = std::pow(g, -2); } else { bd = (g - 2 * delta) / delta / delta; bdd = std::pow(delta, -2); }
                    double* gv = c.kind == 0 ? gx.data() : c.kind == 1 ? gu.data() : gy.data();
                    double* hv = c.kind == 0 ? hx.data() : c.kind == 1 ? hu.data() : hy.data();
                    int dim = c.kind == 0 ? n : c.kind == 1 ? m : p;
                    for (int a = 0; a < c.nnz; a++) gv[c.idx[a]] += eps * bd * c.coef[a];
                    for (int a = 0; a < c.nnz; a++) for (int b = 0; b < c.nnz; b++) hv[c.idx[a] + dim * c.idx[b]] += eps * (bdd * c.coef[a] * c.coef[b]);
                }
                for (int i = 0; i < m; i++) lu[i] += dt * gu[i];
                for (int i = 0; i < n; i++) lx[i] += dt * gx[i];
                for (int i = 0; i < p; i++) ly[i] += dt * gy[i];
                for (int i = 0; i < m * m; i++) luu[i] += dt * hu[i];
                for (int i = 0; i < n * n; i++) lxx[i] += dt * hx[i];
                for (int i = 0; i < p * p; i++) lyy[i] += dt * hy[i];
            }
        }
    }
    // HKDFootPlaceReg::running_cost_par / terminal_cost_par (HKDCost.cpp:22-35, 52-65): scale * dprel_dx' Qfoot (d | dprel_dx),
    // dprel_dx row (f,a) = contact_f * (e_{12+3f+a} - e_{3+a})
    void hkd_footreg_par(const PhaseDef& P, const double* x, const double* fp, const double* bp, double scale, double* gx, double* gxx) const {
        const int n = 24;
        for (int f = 0; f < 4; f++) for (int a = 0; a < 3; a++) {
            const double cf = P.d.contact[f], q = cf * P.d.w_foot_reg[a];
            const double d = (x[12 + 3 * f + a] - x[3 + a]) - (fp[3 * f + a] - bp[a]);
            const int i0 = 12 + 3 * f + a, i1 = 3 + a;
            gx[i0] += scale * cf * q * d; gx[i1] -= scale * cf * q * d;
            gxx[i0 + n * i0] += scale * cf * q * cf; gxx[i1 + n * i1] += scale * cf * q * cf;
            gxx[i0 + n * i1] -= scale * cf * q * cf; gxx[i1 + n * i0] -= scale * cf * q * cf;
        }
    }
    void terminal_cost_par(const PhaseDef& P, Traj& T, const hsddp_option_t& opt) const {
        int n = P.n, h = P.h; const double* x = &T.X[(size_t)h * n];
        for (int i = 0; i < n; i++) { T.Phix[i] += P.d.qf[i] * (x[i] - P.xr[(size_t)h * n + i]); T.Phixx[i + n * i] += P.d.qf[i]; }
        if (P.d.model == HSDDP_MODEL_HKD) {
            if (P.d.w_foot_reg[0] >= 0) hkd_footreg_par(P, x, &P.foot_pos[(size_t)h * 12], &P.body_pos[(size_t)h * 3], 20.0, T.Phix.data(), T.Phixx.data());
            if (opt.AL_active && !T.th.empty()) {   // TouchDownConstraint::compute_partial (HKDConstraints.cpp:113-170) + compute_AL_partials
                int i = 0; std::vector<double> ag(n, 0.0), ah((size_t)n * n, 0.0);
                for (int l = 0; l < 4; l++) if (P.td[l]) {
                    double J[3][9]; hkd_foot_jac(x + 3, x, x + 12 + 3 * l, l, wp.psi_kin, nullptr, J);
                    double* hx = &T.thx[(size_t)i * n]; std::fill_n(hx, n, 0.0);
                    for (int j = 0; j < 3; j++) { hx[j] = J[2][3 + j]; hx[3 + j] = J[2][j]; hx[12 + 3 * l + j] = J[2][6 + j]; }
                    double sg = T.sigma[i], lm = T.lambda[i], hh = T.th[i];
                    for (int a = 0; a < n; a++) ag[a] += (sg * hh + lm) * hx[a];
                    for (int b = 0; b < n; b++) for (int a = 0; a < n; a++) ah[a + n * b] += (sg * (1 + hh) + lm) * (hx[a] * hx[b]);
                    i++;
                }
                for (int a = 0; a < n; a++) T.Phix[a] += ag[a];
                for (int a = 0; a < n * n; a++) T.Phixx[a] += ah[a];
            }
            return;
        }
        if (P.d.model != HSDDP_MODEL_WB) return;
        WbFootKin F; wb_foot_kin(wp, x, F, true);
        const int* rc = &P.ref_contact[(size_t)h * 4]; const double* fp = &P.foot_pos[(size_t)h * 12]; const double* bp = &P.body_pos[(size_t)h * 3];
        {   // WBFootPlaceReg::terminal_cost_par (MHPCCost.cpp:90-117): x2
            std::vector<double> tx(n, 0.0), txx((size_t)n * n, 0.0);
            for (int f = 0; f < 4; f++) if (rc[f] > 0 && P.d.w_foot_reg[0] >= 0) {
                const auto* w = P.d.w_foot_reg; double d[3]; for (int a = 0; a < 3; a++) d[a] = (F.pos[f][a] - x[a]) - (fp[3 * f + a] - bp[a]);
                double J[3][18]; for (int a = 0; a < 3; a++) for (int j = 0; j < 18; j++) J[a][j] = j < 3 ? 0.0 : F.J[f][a][j];
                for (int i = 0; i < 18; i++) { double s = 0; for (int a = 0; a < 3; a++) s += J[a][i] * w[a] * d[a]; tx[i] += 2 * s; }
                for (int j = 0; j < 18; j++) for (int i = 0; i < 18; i++) { double s = 0; for (int a = 0; a < 3; a++) s += J[a][i] * w[a] * J[a][j]; txx[i + n * j] += 2 * s; }
            }
            for (int i = 0; i < n; i++) T.Phix[i] += tx[i];
            for (int i = 0; i < n * n; i++) T.Phixx[i] += txx[i];
        }
        if (P.n_td > 0 && P.d.w_td_vel >= 0) {  // TDVelocityPenalty::terminal_cost_par (MHPCCost.cpp:271-291)
            std::vector<double> tx(n, 0.0), txx((size_t)n * n, 0.0);
            for (int f = 0; f < 4; f++) if (P.td[f]) {
                double vz = F.vel[f][2], J[36]; for (int j = 0; j < 18; j++) { J[j] = F.Jv[f][2][j]; J[18 + j] = F.J[f][2][j]; }
                for (int i = 0; i < 36; i++) tx[i] += J[i] * P.d.w_td_vel * vz;
                for (int j = 0; j < 36; j++) for (int i = 0; i < 36; i++) txx[i + n * j] += J[i] * P.d.w_td_vel * J[j];
            }
            for (int i = 0; i < n; i++) T.Phix[i] += tx[i];
            for (int i = 0; i < n * n; i++) T.Phixx[i] += txx[i];
        }
        if (opt.AL_active && !T.th.empty()) {  // WBTouchDown::compute_partial + compute_AL_partials (ConstraintsBase.h:412-425)
            int i = 0; std::vector<double> ag(n, 0.0), ah((size_t)n * n, 0.0);
            for (int f = 0; f < 4; f++) if (P.td[f]) {
                double* hx = &T.thx[(size_t)i * n]; for (int j = 0; j < 18; j++) hx[j] = F.J[f][2][j];
                double sg = T.sigma[i], lm = T.lambda[i], hh = T.th[i];
                for (int a = 0; a < n; a++) ag[a] += (sg * hh + lm) * hx[a];
                for (int b = 0; b < n; b++) for (int a = 0; a < n; a++) ah[a + n * b] += (sg * (1 + hh) + lm) * (hx[a] * hx[b]);
                i++;
            }
            for (int a = 0; a < n; a++) T.Phix[a] += ag[a];
            for (int a = 0; a < n * n; a++) T.Phixx[a] += ah[a];
        }
    }

    // ------------------------------------------------------------------ SinglePhase methods
    // SinglePhase::hybrid_rollout (SinglePhase.cpp:182-233)
    bool hybrid_rollout_phase(const PhaseDef& P, Traj& T, double eps, const hsddp_option_t& opt) const {
        int n = P.n, m = P.m, p = P.p, h = P.h;
        std::memcpy(&T.Xsim[0], T.x_init, sizeof(double) * n);
        if (T.shooting) for (int i = 0; i < n; i++) T.X[i] = T.Xbar[i] + eps * T.dX[i];
        else std::memcpy(&T.X[0], T.x_init, sizeof(double) * n);
        double ydummy[12];
        for (int k = 0; k < h; k++) {
            const double* x = &T.X[(size_t)k * n]; double* u = &T.U[(size_t)k * m];
            for (int i = 0; i < m; i++) {
                double s = 0; for (int j = 0; j < n; j++) s += T.K[(size_t)k * m * n + i + m * j] * (x[j] - T.Xbar[(size_t)k * n + j]);
                u[i] = T.Ubar[(size_t)k * m + i] + eps * T.dU[(size_t)k * m + i] + s;
            }
            dynamics(P, k, x, u, &T.Xsim[(size_t)(k + 1) * n], p ? &T.Y[(size_t)k * p] : ydummy);
            double nn = 0; for (int i = 0; i < n; i++) nn += T.Xsim[(size_t)(k + 1) * n + i] * T.Xsim[(size_t)(k + 1) * n + i];
            if (std::sqrt(nn) > 1e6) return false;
            if (opt.MS && T.shooting) for (int i = 0; i < n; i++) T.X[(size_t)(k + 1) * n + i] = T.Xbar[(size_t)(k + 1) * n + i] + eps * T.dX[(size_t)(k + 1) * n + i];
            else std::memcpy(&T.X[(size_t)(k + 1) * n], &T.Xsim[(size_t)(k + 1) * n], sizeof(double) * n);
            path_constraints(P, T, k);
        }
        terminal_constraints(P, T);
        for (size_t i = 0; i < T.Defect.size(); i++) T.Defect[i] = T.Xsim[i] - T.X[i];
        return true;
    }
    // SinglePhase::LQ_approximation (SinglePhase.cpp:265-320)
    void LQ_phase(const PhaseDef& P, Traj& T, const hsddp_option_t& opt) const {
        int n = P.n, m = P.m, p = P.p, h = P.h;
        std::vector<double> Cd((size_t)12 * 36), Dd((size_t)12 * 12);
#pragma omp parallel for num_threads(lq_threads) schedule(static) if (lq_threads > 1)
        for (int k = 0; k < h; k++) {
            double Cl[12 * 36], Dl[12 * 12];
            dynamics_partial(P, k, &T.X[(size_t)k * n], &T.U[(size_t)k * m], &T.A[(size_t)k * n * n], &T.B[(size_t)k * n * m],
                             p ? &T.C[(size_t)k * p * n] : Cl, p ? &T.D[(size_t)k * p * m] : Dl);
        }
        for (int k = 0; k < h; k++) running_cost_par(P, T, k, opt);
        terminal_cost_par(P, T, opt);
    }
    // SinglePhase::backward_sweep (SinglePhase.cpp:323-391)
    bool backward_sweep_phase(const PhaseDef& P, Traj& T, double reg, const double* Gprime, const double* Hprime) const {
        int n = P.n, m = P.m, p = P.p, h = P.h;
        for (int i = 0; i < n; i++) T.G[(size_t)h * n + i] = T.Phix[i] + Gprime[i];
        for (int i = 0; i < n * n; i++) T.H[(size_t)h * n * n + i] = T.Phixx[i] + Hprime[i];
        T.dV_1 = 0; T.dV_2 = 0;
        std::vector<double> Gn(n), Qx(n), Qxx((size_t)n * n), HA((size_t)n * n), HB((size_t)n * m), tmp((size_t)std::max(n, p) * std::max(n, m)),
            Quu_s((size_t)m * m), Quu_inv((size_t)m * m), Iuu((size_t)m * m, 0.0), QiQu(m), QiQux((size_t)m * n);
        for (int i = 0; i < m; i++) Iuu[i + m * i] = 1.0;
        LDLT chol;
        for (int k = h - 1; k >= 0; k--) {
            const double* A = &T.A[(size_t)k * n * n]; const double* B = &T.B[(size_t)k * n * m];
            const double* C = p ? &T.C[(size_t)k * p * n] : nullptr; const double* D = p ? &T.D[(size_t)k * p * m] : nullptr;
            const double* Hn = &T.H[(size_t)(k + 1) * n * n];
            double* Qu = &T.Qu[(size_t)k * m]; double* Quu = &T.Quu[(size_t)k * m * m]; double* Qux = &T.Qux[(size_t)k * m * n];
            for (int i = 0; i < n; i++) Gn[i] = T.G[(size_t)(k + 1) * n + i];
            gemv(false, n, n, Hn, n, &T.Defect[(size_t)(k + 1) * n], Gn.data(), 1.0, 1.0);
            gemv(true, n, n, A, n, Gn.data(), Qx.data()); for (int i = 0; i < n; i++) Qx[i] += T.lx[(size_t)k * n + i];
            gemv(true, n, m, B, n, Gn.data(), Qu); for (int i = 0; i < m; i++) Qu[i] += T.lu[(size_t)k * m + i];
            gemm(false, false, n, n, n, Hn, n, A, n, HA.data(), n);
            gemm(false, false, n, m, n, Hn, n, B, n, HB.data(), n);
            gemm(true, false, n, n, n, A, n, HA.data(), n, Qxx.data(), n); for (int i = 0; i < n * n; i++) Qxx[i] += T.lxx[(size_t)k * n * n + i];
            gemm(true, false, m, m, n, B, n, HB.data(), n, Quu, m); for (int i = 0; i < m * m; i++) Quu[i] += T.luu[(size_t)k * m * m + i];
            gemm(true, false, m, n, n, B, n, HA.data(), n, Qux, m); for (int i = 0; i < m * n; i++) Qux[i] += T.lux[(size_t)k * m * n + i];
            if (p > 0) {
                const double* ly = &T.ly[(size_t)k * p]; const double* lyy = &T.lyy[(size_t)k * p * p];
                gemv(true, p, n, C, p, ly, Qx.data(), 1.0, 1.0);
                gemv(true, p, m, D, p, ly, Qu, 1.0, 1.0);
                std::vector<double> lC((size_t)p * n), lD((size_t)p * m);
                gemm(false, false, p, n, p, lyy, p, C, p, lC.data(), p);
                gemm(false, false, p, m, p, lyy, p, D, p, lD.data(), p);
                gemm(true, false, n, n, p, C, p, lC.data(), p, Qxx.data(), n, 1.0, 1.0);
                gemm(true, false, m, m, p, D, p, lD.data(), p, Quu, m, 1.0, 1.0);
                gemm(true, false, m, n, p, D, p, lC.data(), p, Qux, m, 1.0, 1.0);
            }
            for (int i = 0; i < n; i++) Qxx[i + n * i] += reg;
            for (int i = 0; i < m; i++) Quu[i + m * i] += reg;
            for (int i = 0; i < m * m; i++) Quu_s[i] = Quu[i];
            for (int i = 0; i < m; i++) Quu_s[i + m * i] -= 1e-9;
            chol.compute(Quu_s.data(), m);
            if (!chol.isPositive()) return false;
            chol.solve(Iuu.data(), m, Quu_inv.data());
            for (int j = 0; j < n; j++) for (int i = 0; i <= j; i++) { double s = (Qxx[i + n * j] + Qxx[j + n * i]) / 2; Qxx[i + n * j] = s; Qxx[j + n * i] = s; }
            gemv(false, m, m, Quu_inv.data(), m, Qu, QiQu.data());
            gemm(false, false, m, n, m, Quu_inv.data(), m, Qux, m, QiQux.data(), m);
            double* dU = &T.dU[(size_t)k * m]; double* Kk = &T.K[(size_t)k * m * n];
            for (int i = 0; i < m; i++) dU[i] = -QiQu[i];
            for (int i = 0; i < m * n; i++) Kk[i] = -QiQux[i];
            double* G = &T.G[(size_t)k * n]; double* H = &T.H[(size_t)k * n * n];
            gemv(true, m, n, Qux, m, QiQu.data(), G, -1.0, 0.0); for (int i = 0; i < n; i++) G[i] += Qx[i];
            gemm(true, false, n, n, m, Qux, m, QiQux.data(), m, H, n, -1.0, 0.0); for (int i = 0; i < n * n; i++) H[i] += Qxx[i];
            double dVk = -dotv(m, Qu, dU);
            T.dV_1 -= dVk; T.dV_2 += dVk;
        }
        gemv(false, n, n, &T.H[0], n, &T.Defect[0], &T.G[0], 1.0, 1.0);
        return true;
    }
    // SinglePhase::linear_rollout (SinglePhase.cpp:145-178)
    void linear_rollout_phase(const PhaseDef& P, Traj& T, double eps) const {
        int n = P.n, m = P.m, h = P.h; std::vector<double> du(m), t1(std::max(n, m));
        T.dV_1 = 0; T.dV_2 = 0;
        for (int i = 0; i < n; i++) T.dX[i] = T.dx_init[i] + eps * T.Defect[i];
        for (int k = 0; k < h; k++) {
            const double* dx = &T.dX[(size_t)k * n]; const double* A = &T.A[(size_t)k * n * n]; const double* B = &T.B[(size_t)k * n * m];
            gemv(false, m, n, &T.K[(size_t)k * m * n], m, dx, du.data()); for (int i = 0; i < m; i++) du[i] += eps * T.dU[(size_t)k * m + i];
            double* dxn = &T.dX[(size_t)(k + 1) * n];
            gemv(false, n, n, A, n, dx, dxn); gemv(false, n, m, B, n, du.data(), dxn, 1.0, 1.0);
            for (int i = 0; i < n; i++) dxn[i] += eps * T.Defect[(size_t)(k + 1) * n + i];
            T.dV_1 += dotv(n, &T.lx[(size_t)k * n], dx) + dotv(m, &T.lu[(size_t)k * m], du.data());
            gemv(false, n, n, &T.lxx[(size_t)k * n * n], n, dx, t1.data()); T.dV_2 += dotv(n, dx, t1.data());
            gemv(false, m, m, &T.luu[(size_t)k * m * m], m, du.data(), t1.data()); T.dV_2 += dotv(m, du.data(), t1.data());
            gemv(false, m, n, &T.lux[(size_t)k * m * n], m, dx, t1.data()); T.dV_2 += dotv(m, du.data(), t1.data());
        }
        const double* dx = &T.dX[(size_t)h * n];
        T.dV_1 += dotv(n, T.Phix.data(), dx);
        gemv(false, n, n, T.Phixx.data(), n, dx, t1.data()); T.dV_2 += dotv(n, dx, t1.data());
    }

    // ------------------------------------------------------------------ MultiPhaseDDP methods
    // MultiPhaseDDP::hybrid_rollout (MultiPhaseDDP.cpp:49-92)
    bool hybrid_rollout(Problem& q, double eps, const hsddp_option_t& opt) {
        q.actual_cost = 0; q.max_pconstr = 0; q.max_tconstr = 0;
        double xinit[36]; std::memcpy(xinit, q.x0, sizeof(xinit));
        bool success = true;
        for (size_t i = 0; i < ph.size(); i++) {
            if (!opt.MS) q.tr[i].shooting = false;
            if (i > 0) resetmap(ph[i - 1], &q.tr[i - 1].X[(size_t)ph[i - 1].h * ph[i - 1].n], xinit);
            std::memcpy(q.tr[i].x_init, xinit, sizeof(double) * ph[i].n);
            if (!hybrid_rollout_phase(ph[i], q.tr[i], eps, opt)) { success = false; break; }
            q.max_pconstr = std::min(q.max_pconstr, q.tr[i].max_pviol);
            q.max_tconstr = std::max(q.max_tconstr, q.tr[i].max_tviol);
        }
        return success;
    }
    void compute_cost(Problem& q, const hsddp_option_t& opt) {
        q.actual_cost = 0;
        for (size_t i = 0; i < ph.size(); i++) { compute_cost_phase(ph[i], q.tr[i], opt); q.actual_cost += q.tr[i].actual_cost; }
    }
    void LQ_approximation(Problem& q, const hsddp_option_t& opt) { for (size_t i = 0; i < ph.size(); i++) LQ_phase(ph[i], q.tr[i], opt); }
    double measure_dynamics_feasibility(Problem& q) {  // norm_id = 2 (MultiPhaseDDP.cpp:533-548)
        double f = 0; for (auto& T : q.tr) { double s = 0; for (size_t k = 0; k < T.Defect.size(); k++) s += T.Defect[k] * T.Defect[k]; f += s; }
        return std::sqrt(f);
    }
    // MultiPhaseDDP::backward_sweep (MultiPhaseDDP.cpp:174-213) + impact_aware_step (:499-503)
    bool backward_sweep(Problem& q, double reg) {
        int np = ph.size(); q.dV_1 = 0; q.dV_2 = 0;
        for (int i = np - 1; i >= 0; i--) {
            int xs = ph[i].n, xsn = (i < np - 1) ? ph[i + 1].n : xs;
            std::vector<double> Gp(xs, 0.0), Hp((size_t)xs * xs, 0.0);
            if (i <= np - 2) {
                std::vector<double> Px((size_t)xsn * xs), t((size_t)xsn * xs);
                resetmap_partial(ph[i], &q.tr[i].X[(size_t)ph[i].h * xs], Px.data());
                const double* G0 = &q.tr[i + 1].G[0]; const double* H0 = &q.tr[i + 1].H[0];
                gemv(true, xsn, xs, Px.data(), xsn, G0, Gp.data());
                gemm(false, false, xsn, xs, xsn, H0, xsn, Px.data(), xsn, t.data(), xsn);
                gemm(true, false, xs, xs, xsn, Px.data(), xsn, t.data(), xsn, Hp.data(), xs);
            }
            if (!backward_sweep_phase(ph[i], q.tr[i], reg, Gp.data(), Hp.data())) return false;
            q.dV_1 += q.tr[i].dV_1; q.dV_2 += q.tr[i].dV_2;
        }
        return true;
    }
    // MultiPhaseDDP::backward_sweep_regularized (MultiPhaseDDP.cpp:136-165)
    bool backward_sweep_regularized(Problem& q, double& reg, const hsddp_option_t& opt, int& iter) {
        bool success = false; iter = 0;
        while (!success) {
            iter++;
            success = backward_sweep(q, reg);
            if (success) break;
            reg = std::max<double>(reg * opt.update_regularization, 1e-03);
            if (reg > 1e2) break;
        }
        reg = reg / 20; if (reg < 1e-06) reg = 0;
        return success;
    }
    // MultiPhaseDDP::linear_rollout (MultiPhaseDDP.cpp:12-42)
    void linear_rollout(Problem& q, double eps) {
        double dx_init[36] = {0}; q.dV_1 = 0; q.dV_2 = 0;
        for (size_t i = 0; i < ph.size(); i++) {
            if (i > 0) {
                int xs = ph[i - 1].n, xsn = ph[i].n; std::vector<double> Px((size_t)xsn * xs);
                resetmap_partial(ph[i - 1], &q.tr[i - 1].X[(size_t)ph[i - 1].h * xs], Px.data());
                gemv(false, xsn, xs, Px.data(), xsn, &q.tr[i - 1].dX[(size_t)ph[i - 1].h * xs], dx_init);
            }
            std::memcpy(q.tr[i].dx_init, dx_init, sizeof(double) * ph[i].n);
            linear_rollout_phase(ph[i], q.tr[i], eps);
            q.dV_1 += q.tr[i].dV_1; q.dV_2 += q.tr[i].dV_2;
        }
    }
    void update_nominal_trajectory(Problem& q) { for (auto& T : q.tr) { T.Xbar = T.X; T.Ubar = T.U; T.Defect_bar = T.Defect; } }
    // MultiPhaseDDP::line_search (MultiPhaseDDP.cpp:95-133)
    bool line_search(Problem& q, const hsddp_option_t& opt, int& iter) {
        double eps = 1, merit_prev = q.merit, feas_prev = q.feas; bool success = false; iter = 0;
        while (eps > 1e-3) {
            iter++;
            bool rollout_success = hybrid_rollout(q, eps, opt);
            compute_cost(q, opt);
            q.feas = measure_dynamics_feasibility(q);
            q.merit = q.actual_cost + q.merit_rho * q.feas;
            double exp_cost_change = eps * q.dV_1 + 0.5 * eps * eps * q.dV_2;
            double exp_merit_change = exp_cost_change - eps * q.merit_rho * feas_prev;
            if ((q.merit <= merit_prev + opt.gamma * exp_merit_change) && rollout_success) { success = true; break; }
            eps *= opt.alpha;
        }
        return success;
    }
    void update_AL_params(Problem& q, const hsddp_option_t& opt) {  // TerminalConstraintBase::update_params (ConstraintsBase.h:375-391)
        for (size_t i = 0; i < ph.size(); i++) { Traj& T = q.tr[i];
            for (size_t c = 0; c < T.th.size(); c++) {
                if (std::fabs(T.th[c]) < opt.tconstr_thresh) continue;
                if (std::fabs(T.th[c]) > 0.005) { T.sigma[c] *= opt.update_penalty; T.sigma[c] = std::min<double>(T.sigma[c], ph[i].d.al_td.sigma_max); }
                else T.lambda[c] += T.th[c] * T.sigma[c];
            } }
    }
    void update_REB_params(Problem& q, const hsddp_option_t& opt) {  // PathConstraintBase::update_params (ConstraintsBase.h:194-209)
        for (size_t i = 0; i < ph.size(); i++) { Traj& T = q.tr[i]; int ng = ph[i].cons.size();
            for (const auto& gr : ph[i].groups) for (int k = 0; k < ph[i].h; k++) for (int c = gr.first; c < gr.first + gr.size; c++) {
                size_t id = (size_t)k * ng + c;
                if (T.g[id] > -opt.pconstr_thresh) continue;
                T.eps[id] *= opt.update_ReB;
                T.delta[id] *= opt.update_relax; T.delta[id] = std::fmax(T.delta[id], gr.init.delta_min);
            } }
    }

    // MultiPhaseDDP::solve (MultiPhaseDDP.cpp:216-447); printf traces removed
    void solve_one(Problem& q, const hsddp_option_t& opt, float max_cputime) {
        using clk = std::chrono::high_resolution_clock;
        auto t0 = clk::now();
        auto elapsed = [&]() { return std::chrono::duration<float, std::milli>(clk::now() - t0).count(); };
        auto timeup = [&]() { float e = elapsed(); return e > max_cputime || std::fabs(e - max_cputime) <= 1e-6f; };
        q.iter_ = 0; q.ls_iter_total_ = 0; q.status = 0;
        int iter_ou = 0, iter_in = 0; double cost_prev = 0, merit_prev = 0; bool success = true;
        q.cost_buffer.clear(); q.dyn_feas_buffer.clear(); q.eqn_feas_buffer.clear(); q.ineq_feas_buffer.clear();
        bool max_cputime_reached = false;
        hybrid_rollout(q, 0, opt);
        update_nominal_trajectory(q);
        compute_cost(q, opt);
        q.feas = measure_dynamics_feasibility(q);
        auto push = [&]() { q.cost_buffer.push_back(q.actual_cost); q.dyn_feas_buffer.push_back(q.feas); q.eqn_feas_buffer.push_back(q.max_tconstr); q.ineq_feas_buffer.push_back(q.max_pconstr); };
        push();
        double regularization = 0;
        while (iter_ou < opt.max_AL_iter && !max_cputime_reached) {
            iter_ou++;
            q.max_tconstr_prev = q.max_tconstr; q.max_pconstr_prev = q.max_pconstr;
            regularization = 0; iter_in = 0;
            while (iter_in < opt.max_DDP_iter && !max_cputime_reached) {
                compute_cost(q, opt);
                q.feas = measure_dynamics_feasibility(q);
                iter_in++; q.iter_++;
                if (timeup()) { max_cputime_reached = true; break; }
                LQ_approximation(q, opt);
                if (timeup()) { max_cputime_reached = true; break; }
                int reg_iter = 0;
                success = backward_sweep_regularized(q, regularization, opt, reg_iter);
                q.reg_iter_total_ += reg_iter;
                if (!success) goto bad_solve;
                if (timeup()) { max_cputime_reached = true; break; }
                if (opt.MS) linear_rollout(q, 1.0);
                {
                    double dV_abs = std::fabs(q.dV_1 + 0.5 * q.dV_2);
                    q.merit_rho = (q.feas > opt.dynamics_feas_thresh) ? dV_abs / ((1 - opt.merit_scale) * q.feas) + opt.merit_offset : 0;
                    q.merit = q.actual_cost + q.merit_rho * q.feas;
                    cost_prev = q.actual_cost; merit_prev = q.merit;
                    if ((dV_abs < opt.cost_thresh) && (q.feas <= opt.dynamics_feas_thresh)) break;
                }
                {
                    int ls_iter = 0; bool ls_success = line_search(q, opt, ls_iter);
                    q.ls_iter_total_ += ls_iter;
                    if (ls_success) update_nominal_trajectory(q);
                    else { q.actual_cost = cost_prev; q.merit = merit_prev; }
                    if (trace) printf("[oracle] ou %d in %d cost %.6f feas %.3e dV1 %.3e dV2 %.3e rho %.3e ls %d ok %d reg %.1e pc %.3e tc %.3e\n", iter_ou, iter_in, q.actual_cost, q.feas, q.dV_1, q.dV_2, q.merit_rho, ls_iter, (int)ls_success, regularization, q.max_pconstr, q.max_tconstr);
                }
                if ((std::fabs((cost_prev - q.actual_cost) / cost_prev) < opt.cost_thresh) && (q.feas <= opt.dynamics_feas_thresh)) break;
                if (timeup()) { max_cputime_reached = true; break; }
                push();
            }
            if (q.max_tconstr < opt.tconstr_thresh && std::fabs(q.max_pconstr) < opt.pconstr_thresh && q.feas <= opt.dynamics_feas_thresh) break;
            if (std::fabs(q.max_tconstr - q.max_tconstr_prev) < 0.0001 && std::fabs(q.max_pconstr - q.max_pconstr_prev) < 0.0001 && q.feas <= opt.dynamics_feas_thresh) break;
            if (max_cputime_reached) { q.status = 2; break; }
            if (opt.AL_active) update_AL_params(q, opt);
            if (opt.ReB_active) update_REB_params(q, opt);
            if (iter_ou >= opt.max_AL_iter) break;
        }
    bad_solve:
        if (!success) q.status = 1;
        q.solve_time_ = elapsed();
    }
    void solve(const hsddp_option_t& opt, float max_cputime) {
        auto t0 = std::chrono::high_resolution_clock::now();
#pragma omp parallel for num_threads(problem_threads) schedule(dynamic) if (problem_threads > 1)
        for (int b = 0; b < batch; b++) solve_one(pb[b], opt, max_cputime);
        solve_ms = std::chrono::duration<float, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    }
};

}  // namespace orc
