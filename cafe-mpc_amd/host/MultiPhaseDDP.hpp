// C++ host-side mirror of the reference solver surface, over the C-ABI of include/hsddp.h.
//
// Same method names / argument meaning as MultiPhaseDDP<T> (HSDDPSolver/header/MultiPhaseDDP.h:22-93) so that
// reference-shaped host code (problem builders, MPC drivers) can switch solvers by changing one type:
//     hsddp::MultiPhaseDDP<double> solver;                     // reference: MultiPhaseDDP<double> solver;
//     solver.set_initial_condition(x0);                        //   same
//     solver.set_multiPhaseProblem(phases);                    //   deque<shared_ptr<SinglePhaseBase>> -> descriptors
//     solver.solve(ddp_setting, max_cputime_ms);               //   same
//     solver.get_solver_info(n_iters, n_ls, n_reg, ms);        //   same
// The reference passes closures (SinglePhase.h:65-96); a GPU cannot call them, so phases are the POD descriptors
// `hsddp_phase_desc_t` (what MHPCProblem::create_problem_one_phase binds, MHPCProblem.cpp:403-601).  Error
// behaviour follows the reference (void returns, failures reported through status/prints); `last_error()` exposes
// the C-ABI return code that the reference has no equivalent of.  Header-only, no Eigen needed (plain vectors).
#pragma once
#include <vector>
#include <stdexcept>
#include <cstring>
#include "hsddp.h"

namespace hsddp {

using HSDDP_OPTION = hsddp_option_t;

inline HSDDP_OPTION default_option() {   // HSDDP_CompoundTypes.h:15-36 defaults
    HSDDP_OPTION o{};
    o.alpha = 0.1; o.gamma = 0.1; o.update_penalty = 8; o.update_relax = 0.1; o.update_regularization = 2; o.update_ReB = 7;
    o.max_DDP_iter = 3; o.max_AL_iter = 2; o.max_DDP_iter_runtime = 1; o.max_AL_iter_runtime = 2;
    o.cost_thresh = 1e-3; o.tconstr_thresh = 1e-3; o.pconstr_thresh = 1e-3; o.dynamics_feas_thresh = 1e-3;
    o.merit_rho = 1e4; o.merit_scale = 0.2; o.merit_offset = 10; o.AL_active = 1; o.ReB_active = 1; o.smooth_active = 0; o.MS = 1; o.nsteps_per_node = 1;
    return o;
}

template <typename T = double>
class MultiPhaseDDP {
    static_assert(sizeof(T) == sizeof(double), "the reference instantiates double only (MultiPhaseDDP.cpp:562)");
public:
    explicit MultiPhaseDDP(int batch = 1, int device = 0) : batch_(batch), device_(device) {}
    ~MultiPhaseDDP() { if (h_) hsddp_destroy(h_); }
    MultiPhaseDDP(const MultiPhaseDDP&) = delete;
    MultiPhaseDDP& operator=(const MultiPhaseDDP&) = delete;

    // set_multiPhaseProblem (MultiPhaseDDP.h:33-41): also resets the cost / violation trackers (new handle).
    void set_multiPhaseProblem(const std::vector<hsddp_phase_desc_t>& phases_in, const hsddp_model_param_t* mp = nullptr) {
        if (h_) { hsddp_destroy(h_); h_ = nullptr; }
        n_phases = (int)phases_in.size();
        rc_ = hsddp_create(&h_, n_phases, phases_in.data(), mp, batch_, device_);
        if (rc_ == HSDDP_OK && !x0_.empty()) rc_ = hsddp_set_initial_condition(h_, x0_.data());
    }
    // set_initial_condition (MultiPhaseDDP.h:43): x0 is batch x n0 (one row per ensemble member)
    void set_initial_condition(const std::vector<T>& x0_in) { x0_ = x0_in; if (h_) rc_ = hsddp_set_initial_condition(h_, x0_.data()); }
    // what the builders write into Trajectory::Xbar / Ubar before solve (MHPCProblem.cpp:186-193)
    void set_nominal(int phase, const T* Xbar, const T* Ubar, bool per_problem = false) { rc_ = hsddp_set_nominal(h_, phase, Xbar, Ubar, per_problem ? 1 : 0); }

    void solve(HSDDP_OPTION& option, const float& max_cputime = 1e6) { rc_ = hsddp_solve(h_, &option, max_cputime); refresh(); }

    // public step methods (MultiPhaseDDP.h:51-75)
    bool hybrid_rollout(T eps, HSDDP_OPTION& option) { rc_ = hsddp_hybrid_rollout(h_, eps, &option); return rc_ == HSDDP_OK; }
    void linear_rollout(T eps, HSDDP_OPTION& option) { rc_ = hsddp_linear_rollout(h_, eps, &option); }
    void compute_cost(const HSDDP_OPTION& option) { rc_ = hsddp_compute_cost(h_, &option); }
    void LQ_approximation(HSDDP_OPTION& option) { rc_ = hsddp_LQ_approximation(h_, &option); }
    std::vector<int> backward_sweep(T regularization) { std::vector<int> ok(batch_); rc_ = hsddp_backward_sweep(h_, regularization, ok.data()); return ok; }
    void update_nominal_trajectory() { rc_ = hsddp_update_nominal_trajectory(h_); }

    // getters (MultiPhaseDDP.h:77-93), per problem b of the batch
    T get_actual_cost(int b = 0) const { return info_.at(b).actual_cost; }
    T get_dyn_infeasibility(int b = 0) const { return info_.at(b).dyn_feas; }
    T get_path_constraint_violation(int b = 0) const { return info_.at(b).max_pconstr; }
    T get_terminal_constraint_violation(int b = 0) const { return info_.at(b).max_tconstr; }
    void get_solver_info(int& n_iters, int& n_ls_iters, int& n_reg_iters, float& solve_time, int b = 0) const {
        n_iters = info_.at(b).n_iters; n_ls_iters = info_.at(b).n_ls_iters; n_reg_iters = info_.at(b).n_reg_iters; solve_time = hsddp_get_solve_time_ms(h_);
    }
    // the history overload (MultiPhaseDDP.h:85): cost / dynamics feasibility / terminal / path constraint violation after the initial
    // rollout and after every completed inner iteration
    void get_solver_info(std::vector<float>& cost, std::vector<float>& dyn_feas, std::vector<float>& eqn_feas, std::vector<float>& ineq_feas, int b = 0) {
        int n = 0; rc_ = hsddp_get_history(h_, b, 0, nullptr, nullptr, nullptr, nullptr, &n);
        cost.assign(n, 0.f); dyn_feas.assign(n, 0.f); eqn_feas.assign(n, 0.f); ineq_feas.assign(n, 0.f);
        if (rc_ == HSDDP_OK && n > 0) rc_ = hsddp_get_history(h_, b, n, cost.data(), dyn_feas.data(), eqn_feas.data(), ineq_feas.data(), &n);
    }
    int status(int b = 0) const { return info_.at(b).status; }

    // results live in the handle (the reference mutates caller-owned Trajectory objects in place); copy a field out
    std::vector<T> get_field(int phase, hsddp_field f, int b0 = 0, int nb = 1) const {
        int count = 0, elems = 0; hsddp_field_shape(h_, phase, f, &count, &elems);
        std::vector<T> out((size_t)nb * count * elems);
        if (!out.empty()) hsddp_get_field(h_, phase, f, b0, nb, out.data());
        return out;
    }
    int last_error() const { return rc_; }
    // publish_mpc_cmd (MHPC/MHPCLocomotion.cpp:190-287): the first n_steps control knots of one problem in MHPC_Command_lcmt field order,
    // packed on the device (fp32); the caller copies the rows into its lcm-gen struct
    std::vector<unsigned int> export_mpc_command(int problem, int n_steps, double mpc_time, double dt, const float* status_times = nullptr) {
        std::vector<unsigned int> words(1 + (size_t)n_steps * HSDDP_CMD_WORDS_PER_STEP);
        rc_ = hsddp_export_mpc_command(h_, problem, n_steps, mpc_time, dt, status_times, words.data());
        return words;
    }
    // receding-horizon step (MHPCProblem::update): phase `dphase` continues phase `sphase` of the previous window (sphase < 0: new phase)
    void warm_start_phase(int dphase, MultiPhaseDDP<T>* prev, int sphase, int popped_front) {
        rc_ = hsddp_warm_start_phase(h_, dphase, prev ? prev->h_ : nullptr, sphase, popped_front);
    }
    // receding-horizon step INSIDE the handle (what MHPCLocomotion::update gets from MHPCProblem::update + a new solver object,
    // MHPC/MHPCLocomotion.cpp:102-122): new phase table, trajectories and constraint parameters moved device to device, allocations reused
    void reconfigure(const std::vector<hsddp_phase_desc_t>& phases_in, const std::vector<int>& src_phase, const std::vector<int>& shift) {
        n_phases = (int)phases_in.size();
        rc_ = hsddp_reconfigure(h_, n_phases, phases_in.data(), src_phase.data(), shift.data());
    }
    // Trajectory::Ubar[k] written by the caller between solves (HKDProblem::update: Ubar[0].setZero(), HKDProblem.cpp:220); u = nullptr: zeros
    void set_control_knot(int phase, int k, const T* u = nullptr) { rc_ = hsddp_set_control_knot(h_, phase, k, u); }
    // solver_info_lcmt (lcmtypes/solver_info_lcmt.lcm) as MHPCLocomotion fills it after a solve (MHPC/MHPCLocomotion.cpp:74-79)
    struct SolverInfo { int n_iter, n_ls_iter, n_reg_iter; float solve_time, cost, dyn_feas, ineq_violation, eq_violation; };
    SolverInfo export_solver_info(int problem = 0) {
        static_assert(sizeof(SolverInfo) == 4 * HSDDP_SOLVER_INFO_WORDS, "solver_info_lcmt: eight 32-bit fields");
        SolverInfo s{}; rc_ = hsddp_export_solver_info(h_, problem, reinterpret_cast<unsigned int*>(&s)); return s;
    }
    hsddp_handle_t* handle() { return h_; }

private:
    void refresh() { info_.resize(batch_); if (h_) hsddp_get_info(h_, info_.data()); }
    hsddp_handle_t* h_ = nullptr;
    int batch_, device_, n_phases = 0, rc_ = 0;
    std::vector<T> x0_;
    std::vector<hsddp_info_t> info_;
};

}  // namespace hsddp
