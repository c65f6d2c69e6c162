/*
 * hsddp.h — C-ABI of the MI355X-native Hybrid-Systems DDP solver (libhsddp_hip.so).
 *
 * Drop-in boundary: this interface replaces, for the reference ruihuang1124/CAFE-MPC, the surface
 *     MultiPhaseDDP<T>::{set_multiPhaseProblem, set_initial_condition, solve, get_*}
 *         (HSDDPSolver/header/MultiPhaseDDP.h:31-93)
 * and everything beneath it (SinglePhase.cpp, TrajectoryManagement.cpp, ConstraintsBase.h,
 * SinglePhaseInterface.cpp, WBM.cpp, MHPCCost.cpp, MHPCConstraint.cpp, MHPCReset.cpp).
 *
 * The reference hands phases to the solver as std::function closures + virtual cost/constraint
 * objects (HSDDPSolver/header/SinglePhase.h:42-96), which cannot cross to a GPU.  This ABI carries
 * what those closures capture as plain-old-data "phase descriptors" (MHPCProblem.cpp:403-601):
 * contact pattern, dt, horizon, cost weights, constraint set + ReB/AL parameters, reset-map type and
 * per-knot reference arrays.  One handle solves a BATCH of independent problems (an ensemble of
 * initial states) that share the descriptors; batch is the outermost array dimension everywhere.
 *
 * Conventions: fp64, matrices column-major (Eigen default), arrays [batch][knot][elem].
 * All functions return 0 on success, a negative HSDDP_E* code otherwise.  Calls on one handle must be
 * externally serialised (the reference solver is not re-entrant either).
 * The identical ABI is exported by oracle/liboracle_hsddp.so (CPU restatement, test infrastructure).
 */
#ifndef HSDDP_H
#define HSDDP_H
#ifdef __cplusplus
extern "C" {
#endif

#define HSDDP_OK 0
#define HSDDP_EINVAL (-1)   /* bad argument / shape mismatch */
#define HSDDP_ENODEV (-2)   /* no HIP device / kernel launch failure */
#define HSDDP_ENOMEM (-3)
#define HSDDP_ENOTSUP (-4)  /* model or feature not supported by this build */

/* model ids: the three instantiations of SinglePhase<T,xs,us,ys> (SinglePhase.cpp:565-567) */
#define HSDDP_MODEL_WB 0  /* whole body 36/12/12  (MHPC/MHPC-Trajopt/WBM.h:14-19) */
#define HSDDP_MODEL_SRB 1 /* single rigid body 12/12/0 (SRBM.h) */
#define HSDDP_MODEL_HKD 2 /* hybrid kinodynamic 24/24/0 (HKDModel.h) */

/* HSDDP_OPTION  (HSDDPSolver/common/HSDDP_CompoundTypes.h:13-36), same field names. */
typedef struct hsddp_option {
    double alpha, gamma, update_penalty, update_relax, update_regularization, update_ReB;
    int max_DDP_iter, max_AL_iter, max_DDP_iter_runtime, max_AL_iter_runtime;
    double cost_thresh, tconstr_thresh, pconstr_thresh, dynamics_feas_thresh;
    double merit_rho, merit_scale, merit_offset;
    int AL_active, ReB_active, smooth_active, MS, nsteps_per_node;
} hsddp_option_t;

/* ReB parameter triple (ConstraintsBase.h:73-86) and AL triple (ConstraintsBase.h:58-70). */
typedef struct { double delta, delta_min, eps; } hsddp_reb_t;
typedef struct { double sigma, lambda, sigma_max; } hsddp_al_t;

/* One phase = what MHPCProblem::create_problem_one_phase / update_resetmap / add_tconstr_one_phase
 * bind into a SinglePhase (MHPCProblem.cpp:403-482, 524-601). */
typedef struct hsddp_phase_desc {
    int model;            /* HSDDP_MODEL_* */
    int horizon;          /* h: h controls, h+1 states */
    double dt;            /* traj->timeStep */
    double t_offset;      /* SinglePhase::set_time_offset */
    int contact[4];       /* contact pattern of the phase dynamics: FL,FR,HL,HR */
    int next_contact[4];  /* pattern after the phase (reset map + touchdown constraint) */
    int next_model;       /* model of the following phase (WB->SRB projection, MHPCReset.h:24-26); -1 = last */
    int shooting;         /* 1: every knot is a shooting node (update_SS_config(h+1)); 0: single shooting */
    double BG_alpha;      /* Baumgarte gain (mhpc_config.info:8) */
    /* quadratic tracking cost weights [q(n), r(m), qf(n)] (SinglePhaseInterface.cpp:6-18) */
    double q[36], r[24], qf[36];
    /* WB foot costs: 3-vectors (MHPCCost.cpp:119,188,245); <0 in [0] disables the cost */
    double w_foot_reg[3], w_swing_pos[3], w_swing_vel[3];
    double w_td_vel;      /* TDVelocityPenalty qFoot (MHPCCost.h:222); added iff a touchdown follows */
    /* path constraints (MHPCConstraint.cpp) — flags select which are added */
    int c_torque, c_joint, c_minheight, c_grf;
    double torque_limit;          /* 17.0 */
    double joint_lb[3], joint_ub[3];
    double h_min;                 /* 0.20 WB / 0.18 SRB */
    double mu;                    /* 0.6 WB / 0.7 SRB,HKD */
    hsddp_reb_t reb_torque, reb_joint, reb_minheight, reb_grf;
    /* joint-speed box on x[24..35] (BarrelRoll::JointSpeedLimit, BarrelRoll/BarrelRollConstraints.cpp:143-186); placed
     * after the torque limit in the constraint list like BarrelRollTO.cpp:178-199 does */
    int c_jointspeed;
    double jointspeed_lb, jointspeed_ub;   /* -20 / 20 (BarrelRollConstraints.h:69-70) */
    hsddp_reb_t reb_jointspeed;
    /* terminal touchdown constraint (WBTouchDown, MHPCConstraint.cpp:238-288), added iff a touchdown follows */
    int c_touchdown;
    double ground_height;
    hsddp_al_t al_td;
    /* per-knot reference arrays, h+1 entries each, shared by the whole batch (host pointers, copied) */
    const double *xr;        /* (h+1) x n  tracking state reference */
    const double *ur;        /* (h+1) x m  */
    const double *yr;        /* (h+1) x p  (may be NULL when p==0) */
    const double *foot_pos;  /* (h+1) x 12 reference foot placements (QuadAugmentedState::foot_placements) */
    const double *foot_vel;  /* (h+1) x 12 */
    const double *body_pos;  /* (h+1) x 3  reference body position (body_state.head<3>) */
    const int *ref_contact;  /* (h+1) x 4  reference contact flags at the knot time (MHPCCost.cpp:12) */
} hsddp_phase_desc_t;

/* model-level constants that the reference reads from the URDF / generated code */
typedef struct hsddp_model_param {
    double psi_dyn; /* thigh yaw offset used by the Pinocchio-equivalent terms (URDF literal 3.1415) */
    double psi_kin; /* the one baked into the CasADi kinematic-derivative functions (exact pi) */
} hsddp_model_param_t;

/* per-problem result (MultiPhaseDDP::get_* + get_solver_info, MultiPhaseDDP.h:77-93) */
typedef struct hsddp_info {
    double actual_cost, dyn_feas, max_tconstr, max_pconstr; /* get_actual_cost ... */
    int n_iters, n_ls_iters, n_reg_iters;                  /* iter_, ls_iter_total_, reg_iter_total_ */
    int status; /* 0 finished, 1 regularisation failure (bad_solve), 2 stopped by max_cputime */
} hsddp_info_t;

/* Trajectory fields (HSDDPSolver/header/TrajectoryManagement.h:54-84) readable through hsddp_get_field.
 * count = h+1 for state-like, h for control-like fields; *_T fields are per phase (count 1). */
enum hsddp_field {
    HSDDP_F_X = 0, HSDDP_F_XBAR, HSDDP_F_XSIM, HSDDP_F_DEFECT, HSDDP_F_DX, HSDDP_F_G,  /* (h+1) x n   */
    HSDDP_F_U, HSDDP_F_UBAR, HSDDP_F_DU, HSDDP_F_QU,                                 /* h x m       */
    HSDDP_F_Y,                                                                       /* h x p       */
    HSDDP_F_K, HSDDP_F_QUX,                                                          /* h x (m x n) */
    HSDDP_F_QUU,                                                                     /* h x (m x m) */
    HSDDP_F_A, HSDDP_F_B, HSDDP_F_C, HSDDP_F_D,                                      /* h x ...     */
    HSDDP_F_L, HSDDP_F_LX, HSDDP_F_LU, HSDDP_F_LY, HSDDP_F_LXX, HSDDP_F_LUX, HSDDP_F_LUU, HSDDP_F_LYY, /* rcostData */
    HSDDP_F_PHI, HSDDP_F_PHIX, HSDDP_F_PHIXX,                                        /* tcostData   */
    HSDDP_F_H0,                                                                      /* H at knot 0 of the phase, n x n */
    /* constraint parameters the solver updates between AL iterations and carries across MPC ticks (REB_Param_Struct / AL_Param_Struct,
     * ConstraintsBase.h:58-86): per knot per path constraint (h x ng; order torque, joint speed, joint, height, GRF), per terminal constraint */
    HSDDP_F_REB_EPS, HSDDP_F_REB_DELTA,                                              /* h x ng      */
    HSDDP_F_AL_SIGMA, HSDDP_F_AL_LAMBDA,                                             /* 1 x nt      */
    HSDDP_F_COUNT
};

typedef struct hsddp_handle hsddp_handle_t;

/* -- lifetime: replaces MultiPhaseDDP ctor + set_multiPhaseProblem (MultiPhaseDDP.h:31-41).
 * device: HIP device ordinal.  The handle owns all trajectory storage (device-resident across solves,
 * SURVEY 8b "Ownership"). */
int hsddp_create(hsddp_handle_t **out, int n_phases, const hsddp_phase_desc_t *phases,
                 const hsddp_model_param_t *mp, int batch, int device);
/* the same with a precision flag.  HSDDP_PREC_F32 (kinodynamic HKD 24/24/0 and single-rigid-body phases, HKDModel.h:33-61,
 * SinglePhase.cpp:566-567): the LQ records (A, B, lxx, luu, lx, lu) are kept in fp32 and the Riccati sweep and the linear rollout run in
 * fp32 on the fp32 matrix cores (half the HBM bytes of the sweep, a third of its LDS); rollouts, the LQ knot evaluation, the merit
 * function and every trajectory the ABI hands out stay fp64.  The reference instantiates double only (MultiPhaseDDP.cpp:562): fp32
 * results are held to a measured tolerance against the fp64 oracle (tests), not to north_star's 1e-6. */
#define HSDDP_PREC_F64 0
#define HSDDP_PREC_F32 1
int hsddp_create_ex(hsddp_handle_t **out, int n_phases, const hsddp_phase_desc_t *phases,
                    const hsddp_model_param_t *mp, int batch, int device, int precision);
int hsddp_precision(hsddp_handle_t *h);
void hsddp_destroy(hsddp_handle_t *h);

/* -- MultiPhaseDDP::set_initial_condition (MultiPhaseDDP.h:43); x0: batch x n(phase 0) */
int hsddp_set_initial_condition(hsddp_handle_t *h, const double *x0);
/* -- nominal trajectories, what builders write into Trajectory::Xbar/Ubar (MHPCProblem.cpp:186-193,
 * testMHPCProblem.cpp:70-84).  Xbar: [batch|1] x (h+1) x n, Ubar: [batch|1] x h x m;
 * per_problem=0 broadcasts one trajectory to the whole batch. Also zeroes K, dU, dX like a fresh Trajectory. */
int hsddp_set_nominal(hsddp_handle_t *h, int phase, const double *Xbar, const double *Ubar, int per_problem);

/* -- one control knot of the nominal trajectory, what a caller writes into Trajectory::Ubar[k] between solves: HKDProblem::update ends with
 * trajectory_ptrs.front()->Ubar[0].setZero() (HKDMPC/HKD-TrajOpt/HKDProblem.cpp:220).  u: batch x m, or NULL for zeros.  Ubar[k] and U[k]
 * are set; everything else of the trajectory stays. */
int hsddp_set_control_knot(hsddp_handle_t *h, int phase, int k, const double *u);

/* -- MultiPhaseDDP::solve (MultiPhaseDDP.cpp:216-447).  max_cputime_ms as in the reference.  opt->MS = 0: single shooting over the whole
 * horizon (MultiPhaseDDP.cpp:65-68: no shooting nodes, no defects, no linear rollout; dV from the backward sweep). */
int hsddp_solve(hsddp_handle_t *h, const hsddp_option_t *opt, float max_cputime_ms);

/* -- the public step methods of MultiPhaseDDP (MultiPhaseDDP.h:51-75), exposed for per-iterate parity tests */
int hsddp_hybrid_rollout(hsddp_handle_t *h, double eps, const hsddp_option_t *opt);   /* MultiPhaseDDP.cpp:49 */
int hsddp_compute_cost(hsddp_handle_t *h, const hsddp_option_t *opt);                 /* :450 */
int hsddp_LQ_approximation(hsddp_handle_t *h, const hsddp_option_t *opt);             /* :461 */
int hsddp_backward_sweep(hsddp_handle_t *h, double regularization, int *success /*batch*/); /* :174 */
int hsddp_linear_rollout(hsddp_handle_t *h, double eps, const hsddp_option_t *opt);   /* :12  */
int hsddp_update_nominal_trajectory(hsddp_handle_t *h);                               /* :524 */
int hsddp_get_exp_cost_change(hsddp_handle_t *h, double *dV_1, double *dV_2 /*batch each*/);
int hsddp_measure_dynamics_feasibility(hsddp_handle_t *h, double *feas /*batch*/);    /* :533 */

/* -- results */
int hsddp_get_info(hsddp_handle_t *h, hsddp_info_t *info /* batch */);
/* MultiPhaseDDP::get_solver_info(cost, dyn_feas, eqn_feas, ineq_feas) (MultiPhaseDDP.h:85, MultiPhaseDDP.cpp:551-559): the four history
 * buffers of one problem (std::vector<float> in the reference, MultiPhaseDDP.h:133-136): cleared by solve, one entry after the initial
 * rollout (MultiPhaseDDP.cpp:258-261) and one per inner iteration that ran to its end (:382-385).  Each destination holds `cap`
 * floats (NULL: skipped); *n receives the number of entries the solver buffered (entries beyond `cap` are not copied). */
int hsddp_get_history(hsddp_handle_t *h, int problem, int cap, float *cost, float *dyn_feas, float *eqn_feas, float *ineq_feas, int *n);
/* copies field `f` of `phase` for problems [b0, b0+nb) into dst (host), layout [nb][count][elems] */
int hsddp_get_field(hsddp_handle_t *h, int phase, int field, int b0, int nb, double *dst);
/* elems per knot and knot count of a field for a phase (so callers can size dst) */
int hsddp_field_shape(hsddp_handle_t *h, int phase, int field, int *count, int *elems);
/* time of the last hsddp_solve in ms (solve_time_, MultiPhaseDDP.cpp:444-446) */
float hsddp_get_solve_time_ms(hsddp_handle_t *h);

/* -- measurement hooks (bench.py): HIP-event time (ms) spent in each kernel family during the last solve
 * and number of launches; names returned as a NUL-separated list. Optional for the CPU backend. */
int hsddp_get_kernel_times(hsddp_handle_t *h, int max_n, double *ms, long long *launches, char *names, int names_cap);
/* knots processed by the launches of kernel family `name` ("k_rollout", "k_lq", "k_sweep") since the last reset: launches are masked per
 * problem, so this is what the algorithmic-bytes figure of the roofline is multiplied with */
int hsddp_get_kernel_units(hsddp_handle_t *h, const char *name, long long *units);
int hsddp_reset_kernel_times(hsddp_handle_t *h);
/* number of device allocations the library has made so far in this process (an MPC tick must not add to it once the handle is warm) */
long long hsddp_debug_malloc_count(void);

/* -- receding-horizon warm start (MHPCProblem::update, MHPCProblem.cpp:252-397): the phase `dphase` of handle `dst` takes its
 * nominal trajectory from phase `sphase` of handle `src` the way SinglePhase::pop_front / push_back_default shift the
 * Trajectory deques (SinglePhase.cpp:513-528, TrajectoryManagement.cpp:130-228), device to device:
 *   state   k <- Xbar_src[k + shift]              for k + shift <= h_src, else X_src[h_src]   (push_back_state(X.back()))
 *   control k <- Ubar_src[k + shift], K_src[..]   for k + shift <  h_src, else 0
 * sphase < 0: a phase created by the update (zero trajectory).  Same model and batch on both sides. */
int hsddp_warm_start_phase(hsddp_handle_t *dst, int dphase, hsddp_handle_t *src, int sphase, int shift);
/* -- the same receding-horizon update INSIDE one handle: the phase table is replaced by `phases`; phase i of the new window continues phase
 * src_phase[i] of the old one shifted by shift[i] knots (as hsddp_warm_start_phase) or is new (src_phase[i] < 0: zero trajectory, initial
 * constraint parameters).  Both calls also carry the constraint parameters the way the reference's phase objects do: the per-knot ReB
 * parameters travel with their knots and a pushed knot copies the last knot's (PathConstraintBase::pop_front / push_back,
 * ConstraintsBase.h:296-306; reset_params() is a no-op, :192), the AL parameters stay with the phase's terminal constraint (:375).
 * The handle keeps its solver state (reg_iter_total_ goes on counting like the reference's solver object, MultiPhaseDDP.cpp:218-221) and
 * reuses its device allocations: after the first ticks a call performs no hipMalloc / hipFree (the 18 ms budget of MHPCLocomotion.cpp:122).
 * The initial condition is kept; set the new one with hsddp_set_initial_condition. */
int hsddp_reconfigure(hsddp_handle_t *h, int n_phases, const hsddp_phase_desc_t *phases, const int *src_phase, const int *shift);

/* -- policy export in the field order of lcmtypes/MHPC_Command_lcmt.lcm, filled the way MHPCLocomotion::publish_mpc_cmd does
 * (MHPC/MHPCLocomotion.cpp:190-287): the first n_steps control knots of problem `problem`, walking the whole-body phases in
 * order, everything cast to fp32, matrices column-major (Eigen .data()).  `out` receives 32-bit words (host byte order; LCM's
 * own wire encoding is big-endian and adds a fingerprint — that stays with the caller's lcm-gen code):
 *   [0]                       int32  N_mpcsteps
 *   then, each as N_mpcsteps consecutive rows:  float mpc_times[1] | torque[12] (Ubar) | eul[3] (Xbar 3..5) | pos[3] (Xbar 0..2) |
 *   qJ[12] (Xbar 6..17) | vWorld[3] (Xbar 18..20) | eulrate[3] (Xbar 21..23) | qJd[12] (Xbar 24..35) | GRF[12] (Y) |
 *   feedback[432] (K 12x36) | Qu[12] | Quu[144] | Qux[432] | int32 contacts[4] (phase contact) | float statusTimes[4]
 * mpc_times[k] = mpc_time + k*dt; status_times: n_phases x 4 (wb_contact_durations) or NULL (zeros).
 * Fails with HSDDP_EINVAL if the first n_steps knots are not all whole-body knots. */
#define HSDDP_CMD_WORDS_PER_STEP 1089
int hsddp_export_mpc_command(hsddp_handle_t *h, int problem, int n_steps, double mpc_time, double dt, const float *status_times,
                             unsigned int *out /* 1 + n_steps*HSDDP_CMD_WORDS_PER_STEP words */);

/* -- lcmtypes/solver_info_lcmt.lcm as MHPCLocomotion fills it after every solve (MHPC/MHPCLocomotion.cpp:74-79, 127-131): eight 32-bit words
 * in field order  int32 n_iter | int32 n_ls_iter | int32 n_reg_iter | float solve_time (ms) | float cost | float dyn_feas |
 * float ineq_violation (get_path_constraint_violation) | float eq_violation (get_terminal_constraint_violation)  of one problem. */
#define HSDDP_SOLVER_INFO_WORDS 8
int hsddp_export_solver_info(hsddp_handle_t *h, int problem, unsigned int *out /* HSDDP_SOLVER_INFO_WORDS words */);

const char *hsddp_backend_name(void); /* "hip-gfx950" or "cpu-oracle" */

#ifdef __cplusplus
}
#endif
#endif
