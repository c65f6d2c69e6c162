// ORACLE — TEST INFRASTRUCTURE ONLY.  C-ABI of the CPU restatement: the same hsddp_* entry points
// as include/hsddp.h (so parity tests drive both backends with one harness) plus model-level
// oracle_* probes used to pin the restatement against the reference's golden vectors.
//
// -DORC_LONG_DOUBLE builds the SAME restatement with every internal scalar an 80-bit long double (liboracle_hsddp_ld.so): the
// "exact" side of the conditioning tests (tests/test_gpu_parity.py): where fp64 implementations of the same algebra differ by
// more than north_star's 1e-6 on K, it says which of them is nearer the true iterate.  The ABI stays fp64 on both builds.
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <limits>
#include <cstdlib>
#include <new>
#include <omp.h>
#include "hsddp.h"
typedef double abi_f64;
#ifdef ORC_LONG_DOUBLE
#define double long double
#define orc orc_ld      // both libraries may be loaded into one process: keep their (weak, inline) C++ symbols apart
#endif
#include "hsddp_oracle.hpp"

struct hsddp_handle { orc::Solver s; };
using namespace orc;

extern "C" {

const char* hsddp_backend_name(void) { return "cpu-oracle"; }

int hsddp_create(hsddp_handle_t** out, int n_phases, const hsddp_phase_desc_t* phases, const hsddp_model_param_t* mp, int batch, int device) {
    (void)device;
    if (!out || n_phases <= 0 || !phases || batch <= 0) return HSDDP_EINVAL;
    hsddp_handle* h = new (std::nothrow) hsddp_handle();
    if (!h) return HSDDP_ENOMEM;
    int rc = h->s.create(n_phases, phases, mp, batch);
    if (rc != HSDDP_OK) { delete h; return rc; }
    *out = h; return HSDDP_OK;
}
// (the precision flag selects the arithmetic of the HIP backend; the checker is the fp64 reference algorithm whatever it says)
int hsddp_create_ex(hsddp_handle_t** out, int n_phases, const hsddp_phase_desc_t* phases, const hsddp_model_param_t* mp, int batch, int device, int) { return hsddp_create(out, n_phases, phases, mp, batch, device); }
int hsddp_precision(hsddp_handle_t*) { return HSDDP_PREC_F64; }
void hsddp_destroy(hsddp_handle_t* h) { delete h; }

int oracle_set_threads(hsddp_handle_t* h, int lq_threads, int problem_threads) {
    h->s.lq_threads = lq_threads > 0 ? lq_threads : 1; h->s.problem_threads = problem_threads > 0 ? problem_threads : 1; return 0;
}

int hsddp_set_initial_condition(hsddp_handle_t* h, const abi_f64* x0) {
    if (!h || !x0) return HSDDP_EINVAL;
    int n0 = h->s.ph[0].n;
    for (int b = 0; b < h->s.batch; b++) { std::memset(h->s.pb[b].x0, 0, sizeof(h->s.pb[b].x0)); std::copy(x0 + (size_t)b * n0, x0 + (size_t)(b + 1) * n0, h->s.pb[b].x0); }
    return HSDDP_OK;
}
int hsddp_set_nominal(hsddp_handle_t* h, int phase, const abi_f64* Xbar, const abi_f64* Ubar, int per_problem) {
    if (!h || phase < 0 || phase >= (int)h->s.ph.size()) return HSDDP_EINVAL;
    const PhaseDef& P = h->s.ph[phase]; size_t sx = (size_t)(P.h + 1) * P.n, su = (size_t)P.h * P.m;
    for (int b = 0; b < h->s.batch; b++) {
        Traj& T = h->s.pb[b].tr[phase];
        if (Xbar) { const abi_f64* s = Xbar + (per_problem ? b * sx : 0); std::copy(s, s + sx, T.Xbar.begin()); std::copy(s, s + sx, T.X.begin()); }
        if (Ubar) { const abi_f64* s = Ubar + (per_problem ? b * su : 0); std::copy(s, s + su, T.Ubar.begin()); std::copy(s, s + su, T.U.begin()); }
        for (auto& v : T.K) v = 0; for (auto& v : T.dU) v = 0; for (auto& v : T.dX) v = 0;
    }
    return HSDDP_OK;
}
int hsddp_set_control_knot(hsddp_handle_t* h, int phase, int k, const abi_f64* u) {      // Trajectory::Ubar[k] written by the caller (HKDProblem.cpp:220)
    if (!h || phase < 0 || phase >= (int)h->s.ph.size() || k < 0 || k >= h->s.ph[phase].h) return HSDDP_EINVAL;
    const int m = h->s.ph[phase].m;
    for (int b = 0; b < h->s.batch; b++) {
        Traj& T = h->s.pb[b].tr[phase];
        for (int i = 0; i < m; i++) { const abi_f64 v = u ? u[(size_t)b * m + i] : 0; T.Ubar[(size_t)k * m + i] = v; T.U[(size_t)k * m + i] = v; }
    }
    return HSDDP_OK;
}
int hsddp_solve(hsddp_handle_t* h, const hsddp_option_t* opt, float max_cputime_ms) {
    if (!h || !opt) return HSDDP_EINVAL;
    h->s.solve(*opt, max_cputime_ms); return HSDDP_OK;
}
int hsddp_hybrid_rollout(hsddp_handle_t* h, abi_f64 eps, const hsddp_option_t* opt) { for (auto& q : h->s.pb) h->s.hybrid_rollout(q, eps, *opt); return 0; }
int hsddp_compute_cost(hsddp_handle_t* h, const hsddp_option_t* opt) { for (auto& q : h->s.pb) h->s.compute_cost(q, *opt); return 0; }
int hsddp_LQ_approximation(hsddp_handle_t* h, const hsddp_option_t* opt) { for (auto& q : h->s.pb) h->s.LQ_approximation(q, *opt); return 0; }
int hsddp_backward_sweep(hsddp_handle_t* h, abi_f64 reg, int* success) {
    for (int b = 0; b < h->s.batch; b++) { bool ok = h->s.backward_sweep(h->s.pb[b], reg); if (success) success[b] = ok; }
    return 0;
}
int hsddp_linear_rollout(hsddp_handle_t* h, abi_f64 eps, const hsddp_option_t* opt) { (void)opt; for (auto& q : h->s.pb) h->s.linear_rollout(q, eps); return 0; }
int hsddp_update_nominal_trajectory(hsddp_handle_t* h) { for (auto& q : h->s.pb) h->s.update_nominal_trajectory(q); return 0; }
int hsddp_get_exp_cost_change(hsddp_handle_t* h, abi_f64* dV_1, abi_f64* dV_2) {
    for (int b = 0; b < h->s.batch; b++) { dV_1[b] = h->s.pb[b].dV_1; dV_2[b] = h->s.pb[b].dV_2; } return 0;
}
int hsddp_measure_dynamics_feasibility(hsddp_handle_t* h, abi_f64* feas) { for (int b = 0; b < h->s.batch; b++) feas[b] = h->s.measure_dynamics_feasibility(h->s.pb[b]); return 0; }

int hsddp_get_info(hsddp_handle_t* h, hsddp_info_t* info) {
    for (int b = 0; b < h->s.batch; b++) {
        const Problem& q = h->s.pb[b];
        info[b].actual_cost = q.actual_cost; info[b].dyn_feas = q.feas;
        info[b].max_tconstr = q.eqn_feas_buffer.empty() ? q.max_tconstr : q.eqn_feas_buffer.back();
        info[b].max_pconstr = q.ineq_feas_buffer.empty() ? q.max_pconstr : q.ineq_feas_buffer.back();
        info[b].n_iters = q.iter_; info[b].n_ls_iters = q.ls_iter_total_; info[b].n_reg_iters = q.reg_iter_total_; info[b].status = q.status;
    }
    return 0;
}
static const std::vector<double>* field_ptr(const PhaseDef& P, const Traj& T, int f, int& count, int& elems) {
    int n = P.n, m = P.m, p = P.p, h = P.h;
    switch (f) {
        case HSDDP_F_X: count = h + 1; elems = n; return &T.X;
        case HSDDP_F_XBAR: count = h + 1; elems = n; return &T.Xbar;
        case HSDDP_F_XSIM: count = h + 1; elems = n; return &T.Xsim;
        case HSDDP_F_DEFECT: count = h + 1; elems = n; return &T.Defect;
        case HSDDP_F_DX: count = h + 1; elems = n; return &T.dX;
        case HSDDP_F_G: count = h + 1; elems = n; return &T.G;
        case HSDDP_F_U: count = h; elems = m; return &T.U;
        case HSDDP_F_UBAR: count = h; elems = m; return &T.Ubar;
        case HSDDP_F_DU: count = h; elems = m; return &T.dU;
        case HSDDP_F_QU: count = h; elems = m; return &T.Qu;
        case HSDDP_F_Y: count = h; elems = p; return &T.Y;
        case HSDDP_F_K: count = h; elems = m * n; return &T.K;
        case HSDDP_F_QUX: count = h; elems = m * n; return &T.Qux;
        case HSDDP_F_QUU: count = h; elems = m * m; return &T.Quu;
        case HSDDP_F_A: count = h; elems = n * n; return &T.A;
        case HSDDP_F_B: count = h; elems = n * m; return &T.B;
        case HSDDP_F_C: count = h; elems = p * n; return &T.C;
        case HSDDP_F_D: count = h; elems = p * m; return &T.D;
        case HSDDP_F_L: count = h; elems = 1; return &T.l;
        case HSDDP_F_LX: count = h; elems = n; return &T.lx;
        case HSDDP_F_LU: count = h; elems = m; return &T.lu;
        case HSDDP_F_LY: count = h; elems = p; return &T.ly;
        case HSDDP_F_LXX: count = h; elems = n * n; return &T.lxx;
        case HSDDP_F_LUX: count = h; elems = m * n; return &T.lux;
        case HSDDP_F_LUU: count = h; elems = m * m; return &T.luu;
        case HSDDP_F_LYY: count = h; elems = p * p; return &T.lyy;
        case HSDDP_F_PHIX: count = 1; elems = n; return &T.Phix;
        case HSDDP_F_PHIXX: count = 1; elems = n * n; return &T.Phixx;
        case HSDDP_F_H0: count = 1; elems = n * n; return &T.H;
        case HSDDP_F_PHI: count = 1; elems = 1; return nullptr;
        case HSDDP_F_REB_EPS: count = h; elems = (int)P.cons.size(); return &T.eps;
        case HSDDP_F_REB_DELTA: count = h; elems = (int)P.cons.size(); return &T.delta;
        case HSDDP_F_AL_SIGMA: count = 1; elems = (int)T.sigma.size(); return &T.sigma;
        case HSDDP_F_AL_LAMBDA: count = 1; elems = (int)T.lambda.size(); return &T.lambda;
        default: count = 0; elems = 0; return nullptr;
    }
}
int hsddp_field_shape(hsddp_handle_t* h, int phase, int field, int* count, int* elems) {
    if (!h || phase < 0 || phase >= (int)h->s.ph.size() || field < 0 || field >= HSDDP_F_COUNT) return HSDDP_EINVAL;
    field_ptr(h->s.ph[phase], h->s.pb[0].tr[phase], field, *count, *elems); return 0;
}
int hsddp_get_field(hsddp_handle_t* h, int phase, int field, int b0, int nb, abi_f64* dst) {
    if (!h || phase < 0 || phase >= (int)h->s.ph.size() || field < 0 || field >= HSDDP_F_COUNT || b0 < 0 || b0 + nb > h->s.batch) return HSDDP_EINVAL;
    for (int b = 0; b < nb; b++) {
        int count, elems; const Traj& T = h->s.pb[b0 + b].tr[phase];
        const std::vector<double>* v = field_ptr(h->s.ph[phase], T, field, count, elems);
        size_t sz = (size_t)count * elems;
        if (field == HSDDP_F_PHI) dst[b] = T.Phi;
        else if (v) std::copy(v->begin(), v->begin() + sz, dst + b * sz);
    }
    return 0;
}
float hsddp_get_solve_time_ms(hsddp_handle_t* h) { return h->s.solve_ms; }
// solver_info_lcmt as MHPCLocomotion fills it (MHPC/MHPCLocomotion.cpp:74-79)
int hsddp_export_solver_info(hsddp_handle_t* h, int problem, unsigned int* out) {
    if (!h || problem < 0 || problem >= h->s.batch || !out) return HSDDP_EINVAL;
    hsddp_info_t info[1]; const Problem& q = h->s.pb[problem];
    const int iv[3] = {q.iter_, q.ls_iter_total_, q.reg_iter_total_};
    const float fv[5] = {h->s.solve_ms, (float)q.actual_cost, (float)q.feas, (float)(q.ineq_feas_buffer.empty() ? q.max_pconstr : q.ineq_feas_buffer.back()),
                         (float)(q.eqn_feas_buffer.empty() ? q.max_tconstr : q.eqn_feas_buffer.back())};
    (void)info; std::memcpy(out, iv, sizeof(iv)); std::memcpy(out + 3, fv, sizeof(fv));
    return HSDDP_OK;
}
// SinglePhase::pop_front x shift + push_back_default for the rest (SinglePhase.cpp:513-528, TrajectoryManagement.cpp:130-228); the ReB parameters
// travel with their knots and a pushed knot copies the last knot's (PathConstraintBase::pop_front / push_back, ConstraintsBase.h:296-306;
// reset_params() is a no-op, :192), the AL parameters of the terminal constraint stay with the phase (:375)
static void warm_start_traj(const PhaseDef& PD, Traj& D, const PhaseDef* PS, const Traj* S, int shift) {
    const int n = PD.n, m = PD.m, hd = PD.h; const bool has = S != nullptr; const int hs = has ? PS->h : 0;
    for (int k = 0; k <= hd; k++) for (int i = 0; i < n; i++) {
        const int ks = k + shift; double v = 0.0;
        if (has) v = (ks <= hs) ? S->Xbar[(size_t)ks * n + i] : S->X[(size_t)hs * n + i];
        D.Xbar[(size_t)k * n + i] = v; D.X[(size_t)k * n + i] = v; D.dX[(size_t)k * n + i] = 0.0;
    }
    const int ng = (int)PD.cons.size(); const bool pg = has && (int)PS->cons.size() == ng && hs > 0;
    for (int k = 0; k < hd; k++) {
        const int ks = k + shift; const bool cs = has && ks < hs;
        for (int i = 0; i < m; i++) { const double v = cs ? S->Ubar[(size_t)ks * m + i] : 0.0; D.Ubar[(size_t)k * m + i] = v; D.U[(size_t)k * m + i] = v; D.dU[(size_t)k * m + i] = 0.0; }
        for (int i = 0; i < m * n; i++) D.K[(size_t)k * m * n + i] = cs ? S->K[(size_t)ks * m * n + i] : 0.0;
        if (pg) { const int kg = ks < hs ? ks : hs - 1; for (int i = 0; i < ng; i++) { D.eps[(size_t)k * ng + i] = S->eps[(size_t)kg * ng + i]; D.delta[(size_t)k * ng + i] = S->delta[(size_t)kg * ng + i]; } }
    }
    if (has && S->sigma.size() == D.sigma.size()) { D.sigma = S->sigma; D.lambda = S->lambda; }
}
int hsddp_warm_start_phase(hsddp_handle_t* dst, int dphase, hsddp_handle_t* src, int sphase, int shift) {
    if (!dst || dphase < 0 || dphase >= (int)dst->s.ph.size() || shift < 0) return HSDDP_EINVAL;
    const bool has = src != nullptr && sphase >= 0;
    if (has && (sphase >= (int)src->s.ph.size() || src->s.batch != dst->s.batch || src->s.ph[sphase].d.model != dst->s.ph[dphase].d.model)) return HSDDP_EINVAL;
    for (int b = 0; b < dst->s.batch; b++)
        warm_start_traj(dst->s.ph[dphase], dst->s.pb[b].tr[dphase], has ? &src->s.ph[sphase] : nullptr, has ? &src->s.pb[b].tr[sphase] : nullptr, shift);
    return 0;
}
// the same update inside one handle (include/hsddp.h hsddp_reconfigure): the solver object survives, so its counters keep counting (quirk xi)
int hsddp_reconfigure(hsddp_handle_t* h, int n_phases, const hsddp_phase_desc_t* phases, const int* src_phase, const int* shift) {
    if (!h || n_phases <= 0 || !phases || !src_phase || !shift) return HSDDP_EINVAL;
    for (int i = 0; i < n_phases; i++) if (src_phase[i] >= (int)h->s.ph.size() || shift[i] < 0 || (src_phase[i] >= 0 && h->s.ph[src_phase[i]].d.model != phases[i].model)) return HSDDP_EINVAL;
    hsddp_handle* nh = new (std::nothrow) hsddp_handle(); if (!nh) return HSDDP_ENOMEM;
    hsddp_model_param_t mp; mp.psi_dyn = (abi_f64)h->s.wp.psi_dyn; mp.psi_kin = (abi_f64)h->s.wp.psi_kin;
    int rc = nh->s.create(n_phases, phases, &mp, h->s.batch);
    if (rc != HSDDP_OK) { delete nh; return rc; }
    nh->s.lq_threads = h->s.lq_threads; nh->s.problem_threads = h->s.problem_threads;
    for (int b = 0; b < h->s.batch; b++) {
        Problem& q = nh->s.pb[b]; const Problem& o = h->s.pb[b];
        std::copy(o.x0, o.x0 + 36, q.x0); q.reg_iter_total_ = o.reg_iter_total_;
        for (int i = 0; i < n_phases; i++) { const bool has = src_phase[i] >= 0; warm_start_traj(nh->s.ph[i], q.tr[i], has ? &h->s.ph[src_phase[i]] : nullptr, has ? &o.tr[src_phase[i]] : nullptr, shift[i]); }
    }
    std::swap(h->s, nh->s); delete nh;
    return 0;
}
// MHPCLocomotion::publish_mpc_cmd (MHPC/MHPCLocomotion.cpp:190-287) restated: field order of MHPC_Command_lcmt.lcm, fp32 casts
int hsddp_export_mpc_command(hsddp_handle_t* h, int problem, int n_steps, abi_f64 mpc_time, abi_f64 dt, const float* status_times, unsigned int* out) {
    if (!h || problem < 0 || problem >= h->s.batch || n_steps <= 0 || !out) return HSDDP_EINVAL;
    std::vector<std::pair<int, int>> idx;
    for (int i = 0; i < (int)h->s.ph.size() && (int)idx.size() < n_steps; i++) {
        if (h->s.ph[i].d.model != HSDDP_MODEL_WB) break;
        for (int k = 0; k < h->s.ph[i].h && (int)idx.size() < n_steps; k++) idx.push_back({i, k});
    }
    if ((int)idx.size() < n_steps) return HSDDP_EINVAL;
    auto putf = [](unsigned int* p, abi_f64 v) { float f = (float)v; std::memcpy(p, &f, 4); };
    out[0] = (unsigned int)n_steps; unsigned int* p = out + 1;
    const Problem& pb = h->s.pb[problem];
    auto rows = [&](int w, auto get) { for (int s = 0; s < n_steps; s++) for (int e = 0; e < w; e++) get(p++, s, e); };
    rows(1, [&](unsigned int* q, int s, int) { putf(q, mpc_time + s * dt); });
    rows(12, [&](unsigned int* q, int s, int e) { putf(q, pb.tr[idx[s].first].Ubar[(size_t)idx[s].second * 12 + e]); });
    const int xo[6] = {3, 0, 6, 18, 21, 24}, xw[6] = {3, 3, 12, 3, 3, 12};
    for (int g = 0; g < 6; g++) rows(xw[g], [&](unsigned int* q, int s, int e) { putf(q, pb.tr[idx[s].first].Xbar[(size_t)idx[s].second * 36 + xo[g] + e]); });
    rows(12, [&](unsigned int* q, int s, int e) { putf(q, pb.tr[idx[s].first].Y[(size_t)idx[s].second * 12 + e]); });
    rows(432, [&](unsigned int* q, int s, int e) { putf(q, pb.tr[idx[s].first].K[(size_t)idx[s].second * 432 + e]); });
    rows(12, [&](unsigned int* q, int s, int e) { putf(q, pb.tr[idx[s].first].Qu[(size_t)idx[s].second * 12 + e]); });
    rows(144, [&](unsigned int* q, int s, int e) { putf(q, pb.tr[idx[s].first].Quu[(size_t)idx[s].second * 144 + e]); });
    rows(432, [&](unsigned int* q, int s, int e) { putf(q, pb.tr[idx[s].first].Qux[(size_t)idx[s].second * 432 + e]); });
    rows(4, [&](unsigned int* q, int s, int e) { *q = (unsigned int)h->s.ph[idx[s].first].d.contact[e]; });
    rows(4, [&](unsigned int* q, int s, int e) { float f = status_times ? status_times[idx[s].first * 4 + e] : 0.f; std::memcpy(q, &f, 4); });
    return 0;
}
int hsddp_get_kernel_times(hsddp_handle_t*, int, abi_f64*, long long*, char*, int) { return 0; }
int hsddp_get_kernel_units(hsddp_handle_t*, const char*, long long* units) { if (units) *units = 0; return HSDDP_ENOTSUP; }
int hsddp_reset_kernel_times(hsddp_handle_t*) { return 0; }
long long hsddp_debug_malloc_count(void) { return 0; }
// MultiPhaseDDP::get_solver_info(cost, dyn_feas, eqn_feas, ineq_feas) (MultiPhaseDDP.cpp:551-559)
int hsddp_get_history(hsddp_handle_t* h, int problem, int cap, float* cost, float* dyn_feas, float* eqn_feas, float* ineq_feas, int* n) {
    if (!h || problem < 0 || problem >= h->s.batch || cap < 0 || !n) return HSDDP_EINVAL;
    const Problem& q = h->s.pb[problem];
    *n = (int)q.cost_buffer.size();
    const int m = std::min(*n, cap);
    const std::vector<float>* src[4] = {&q.cost_buffer, &q.dyn_feas_buffer, &q.eqn_feas_buffer, &q.ineq_feas_buffer};
    float* dst[4] = {cost, dyn_feas, eqn_feas, ineq_feas};
    for (int k = 0; k < 4; k++) if (dst[k]) std::copy(src[k]->begin(), src[k]->begin() + m, dst[k]);
    return 0;
}

#ifndef ORC_LONG_DOUBLE
// ---------------------------------------------------------------- model-level probes (tests only)
// continuous-time WB contact dynamics: qdd(18), grf(12)   (WBM.cpp:38-57 / testKKTDynamics.cpp:97-121)
void oracle_wb_forward(const double* x, const double* u, const int* contact, double psi_dyn, double alpha, double* qdd, double* grf) {
    WbParams P; P.psi_dyn = psi_dyn; P.bg_alpha = alpha;
    double tau[18] = {0}; for (int i = 0; i < 12; i++) tau[6 + i] = u[i];
    WbKKT K; wb_forward(P, x, x + 18, tau, contact, K);
    std::memcpy(qdd, K.qdd, sizeof(K.qdd)); std::memcpy(grf, K.grf, sizeof(K.grf));
}
void oracle_wb_dynamics(const double* x, const double* u, const int* contact, double psi_dyn, double psi_kin, double alpha, double dt, double* xnext, double* y) {
    WbParams P; P.psi_dyn = psi_dyn; P.psi_kin = psi_kin; P.bg_alpha = alpha; wb_dynamics(P, x, u, contact, dt, xnext, y);
}
void oracle_wb_dynamics_partial(const double* x, const double* u, const int* contact, double psi_dyn, double psi_kin, double alpha, double dt,
                                double* A, double* B, double* C, double* D) {
    WbParams P; P.psi_dyn = psi_dyn; P.psi_kin = psi_kin; P.bg_alpha = alpha; wb_dynamics_partial(P, x, u, contact, dt, A, B, C, D);
}
void oracle_wb_impact(const double* x, const int* cur, const int* nxt, double psi_dyn, double psi_kin, int impulse_quirk, double* xnext, double* Px) {
    WbParams P; P.psi_dyn = psi_dyn; P.psi_kin = psi_kin; P.impulse_quirk = impulse_quirk != 0;
    if (xnext) wb_impact(P, x, cur, nxt, xnext);
    if (Px) wb_impact_partial(P, x, cur, nxt, Px);
}
// foot kinematics: pos(4x3), vel(4x3), J(4x3x18 row-major), Jv(4x3x18)
void oracle_wb_foot_kin(const double* x, double psi_dyn, double psi_kin, double* pos, double* vel, double* J, double* Jv) {
    WbParams P; P.psi_dyn = psi_dyn; P.psi_kin = psi_kin; WbFootKin F; wb_foot_kin(P, x, F, true);
    std::memcpy(pos, F.pos, sizeof(F.pos)); std::memcpy(vel, F.vel, sizeof(F.vel)); std::memcpy(J, F.J, sizeof(F.J)); std::memcpy(Jv, F.Jv, sizeof(F.Jv));
}
// the four CasADi-equivalent kinematic derivative families at (q, v, qdd, F): each 4 x (3x18) or 4 x (18x18), column-major per block
void oracle_wb_kin_derivs(const double* q, const double* v, const double* qdd, const double* F12, double psi_kin,
                          double* dvel_dq, double* dacc_dq, double* dacc_dv, double* dJTF_dq) {
    WbParams P; P.psi_dyn = psi_kin; P.psi_kin = psi_kin;
    static WbDeriv D; int mask[4] = {1, 1, 1, 1};
    for (int l = 0; l < 4; l++) {   // per-foot force derivative: evaluate with only foot l loaded
        int ml[4] = {0, 0, 0, 0}; ml[l] = 1;
        wb_derivs(P, q, v, qdd, false, F12, ml, D);
        for (int j = 0; j < 18; j++) for (int i = 0; i < 18; i++) dJTF_dq[l * 324 + i + 18 * j] = D.dJTF[i][j];
    }
    wb_derivs(P, q, v, qdd, false, F12, mask, D);
    for (int l = 0; l < 4; l++) for (int j = 0; j < 18; j++) for (int r = 0; r < 3; r++) {
        dvel_dq[l * 54 + r + 3 * j] = D.dvel[l][r][j];
        dacc_dq[l * 54 + r + 3 * j] = D.dacc[l][r][j];
        dacc_dv[l * 54 + r + 3 * j] = D.dacc[l][r][18 + j];
    }
}
void oracle_srb_xdot(const double* x, const double* u, const double* pf, const int* c, double* xd, double* Ac, double* Bc) {
    srb_xdot<double>(x, u, pf, c, xd);
    if (Ac && Bc) srb_partials_ct(x, u, pf, c, Ac, Bc);
}

// HKD model probes (tests/test_oracle_models.py): Euler step + partials; leg kinematics; reset map + partial
void oracle_hkd_step(const double* x, const double* u, double dt, const int* c, double* xn, double* A, double* B) {
    hkd_dynamics(x, u, c, dt, xn);
    if (A && B) hkd_dynamics_partial(x, u, c, dt, A, B);
}
void oracle_hkd_foot(const double* pos, const double* eul, const double* ql, int leg, double psi, double* pf, double* J27) {
    double J[3][9]; hkd_foot_jac(pos, eul, ql, leg, psi, pf, J);
    for (int a = 0; a < 3; a++) for (int j = 0; j < 9; j++) J27[a * 9 + j] = J[a][j];
}
void oracle_hkd_reset(const double* x, const int* c, const int* cn, double psi, double* xn, double* Px) {
    hkd_resetmap(x, c, cn, psi, xn);
    if (Px) hkd_resetmap_partial(x, c, cn, psi, Px);
}

#endif
}  // extern "C"
