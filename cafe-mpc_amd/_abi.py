"""ctypes mirror of include/hsddp.h (struct layouts + prototypes).

Used by the product wrapper (`MultiPhaseDDP` in __init__.py, bound to libhsddp_hip.so) and by the
tests to drive the CPU checker library through the *same* ABI (the tests pass its path in).  Nothing here computes anything.
"""
import ctypes as C
import numpy as np

MODEL_WB, MODEL_SRB, MODEL_HKD = 0, 1, 2
MODEL_DIMS = {MODEL_WB: (36, 12, 12), MODEL_SRB: (12, 12, 0), MODEL_HKD: (24, 24, 0)}

FIELDS = ["X", "XBAR", "XSIM", "DEFECT", "DX", "G", "U", "UBAR", "DU", "QU", "Y", "K", "QUX", "QUU",
          "A", "B", "C", "D", "L", "LX", "LU", "LY", "LXX", "LUX", "LUU", "LYY", "PHI", "PHIX", "PHIXX", "H0",
          "REB_EPS", "REB_DELTA", "AL_SIGMA", "AL_LAMBDA"]
FIELD_ID = {n: i for i, n in enumerate(FIELDS)}


class Option(C.Structure):
    """HSDDP_OPTION (HSDDPSolver/common/HSDDP_CompoundTypes.h:13-36); defaults = the struct defaults there."""
    _fields_ = [(n, C.c_double) for n in ("alpha", "gamma", "update_penalty", "update_relax", "update_regularization", "update_ReB")] + \
               [(n, C.c_int) for n in ("max_DDP_iter", "max_AL_iter", "max_DDP_iter_runtime", "max_AL_iter_runtime")] + \
               [(n, C.c_double) for n in ("cost_thresh", "tconstr_thresh", "pconstr_thresh", "dynamics_feas_thresh",
                                          "merit_rho", "merit_scale", "merit_offset")] + \
               [(n, C.c_int) for n in ("AL_active", "ReB_active", "smooth_active", "MS", "nsteps_per_node")]

    def __init__(self, **kw):
        super().__init__()
        d = dict(alpha=0.1, gamma=0.1, update_penalty=8, update_relax=0.1, update_regularization=2, update_ReB=7,
                 max_DDP_iter=3, max_AL_iter=2, max_DDP_iter_runtime=1, max_AL_iter_runtime=2, cost_thresh=1e-3,
                 tconstr_thresh=1e-3, pconstr_thresh=1e-3, dynamics_feas_thresh=1e-3, merit_rho=1e4, merit_scale=0.2,
                 merit_offset=10, AL_active=1, ReB_active=1, smooth_active=0, MS=1, nsteps_per_node=1)
        d.update(kw)
        for k, v in d.items():
            setattr(self, k, v)


def mhpc_ddp_setting(**kw):
    """MHPC/settings/ddp_setting.info as loadHSDDPSetting reads it (update_regularization is NOT read: quirk xiii)."""
    d = dict(alpha=0.5, gamma=0.1, update_penalty=5, update_relax=1, update_ReB=1, update_regularization=2,
             max_DDP_iter=10, max_AL_iter=20, max_DDP_iter_runtime=1, max_AL_iter_runtime=4, cost_thresh=1e-2,
             tconstr_thresh=1e-3, pconstr_thresh=1e-3, dynamics_feas_thresh=1e-3, merit_rho=1e3, merit_scale=0.2,
             merit_offset=1, AL_active=1, ReB_active=1, smooth_active=0, MS=1, nsteps_per_node=1)
    d.update(kw)
    return Option(**d)


class Reb(C.Structure):
    _fields_ = [("delta", C.c_double), ("delta_min", C.c_double), ("eps", C.c_double)]


class Al(C.Structure):
    _fields_ = [("sigma", C.c_double), ("lambda_", C.c_double), ("sigma_max", C.c_double)]


DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int)


class PhaseDesc(C.Structure):
    _fields_ = [("model", C.c_int), ("horizon", C.c_int), ("dt", C.c_double), ("t_offset", C.c_double),
                ("contact", C.c_int * 4), ("next_contact", C.c_int * 4), ("next_model", C.c_int), ("shooting", C.c_int),
                ("BG_alpha", C.c_double),
                ("q", C.c_double * 36), ("r", C.c_double * 24), ("qf", C.c_double * 36),
                ("w_foot_reg", C.c_double * 3), ("w_swing_pos", C.c_double * 3), ("w_swing_vel", C.c_double * 3),
                ("w_td_vel", C.c_double),
                ("c_torque", C.c_int), ("c_joint", C.c_int), ("c_minheight", C.c_int), ("c_grf", C.c_int),
                ("torque_limit", C.c_double), ("joint_lb", C.c_double * 3), ("joint_ub", C.c_double * 3),
                ("h_min", C.c_double), ("mu", C.c_double),
                ("reb_torque", Reb), ("reb_joint", Reb), ("reb_minheight", Reb), ("reb_grf", Reb),
                ("c_jointspeed", C.c_int), ("jointspeed_lb", C.c_double), ("jointspeed_ub", C.c_double), ("reb_jointspeed", Reb),
                ("c_touchdown", C.c_int), ("ground_height", C.c_double), ("al_td", Al),
                ("xr", DP), ("ur", DP), ("yr", DP), ("foot_pos", DP), ("foot_vel", DP), ("body_pos", DP),
                ("ref_contact", IP)]


class ModelParam(C.Structure):
    _fields_ = [("psi_dyn", C.c_double), ("psi_kin", C.c_double)]


class Info(C.Structure):
    _fields_ = [("actual_cost", C.c_double), ("dyn_feas", C.c_double), ("max_tconstr", C.c_double), ("max_pconstr", C.c_double),
                ("n_iters", C.c_int), ("n_ls_iters", C.c_int), ("n_reg_iters", C.c_int), ("status", C.c_int)]


PREC_F64, PREC_F32 = 0, 1

EXPORTS = ["hsddp_create", "hsddp_create_ex", "hsddp_precision", "hsddp_destroy", "hsddp_set_initial_condition", "hsddp_set_nominal", "hsddp_solve",
           "hsddp_hybrid_rollout", "hsddp_compute_cost", "hsddp_LQ_approximation", "hsddp_backward_sweep",
           "hsddp_linear_rollout", "hsddp_update_nominal_trajectory", "hsddp_get_exp_cost_change",
           "hsddp_measure_dynamics_feasibility", "hsddp_get_info", "hsddp_get_field", "hsddp_field_shape",
           "hsddp_get_solve_time_ms", "hsddp_get_kernel_times", "hsddp_get_kernel_units", "hsddp_reset_kernel_times", "hsddp_get_history",
           "hsddp_export_mpc_command", "hsddp_warm_start_phase", "hsddp_reconfigure", "hsddp_set_control_knot", "hsddp_export_solver_info", "hsddp_debug_malloc_count", "hsddp_backend_name"]


def bind(lib):
    """Attach argtypes/restypes for every entry point of include/hsddp.h to a loaded CDLL."""
    H = C.c_void_p
    OP = C.POINTER(Option)
    lib.hsddp_create.argtypes = [C.POINTER(H), C.c_int, C.POINTER(PhaseDesc), C.POINTER(ModelParam), C.c_int, C.c_int]
    lib.hsddp_create_ex.argtypes = [C.POINTER(H), C.c_int, C.POINTER(PhaseDesc), C.POINTER(ModelParam), C.c_int, C.c_int, C.c_int]
    lib.hsddp_precision.argtypes = [H]
    lib.hsddp_destroy.argtypes = [H]
    lib.hsddp_destroy.restype = None
    lib.hsddp_set_initial_condition.argtypes = [H, DP]
    lib.hsddp_set_nominal.argtypes = [H, C.c_int, DP, DP, C.c_int]
    lib.hsddp_solve.argtypes = [H, OP, C.c_float]
    lib.hsddp_hybrid_rollout.argtypes = [H, C.c_double, OP]
    lib.hsddp_compute_cost.argtypes = [H, OP]
    lib.hsddp_LQ_approximation.argtypes = [H, OP]
    lib.hsddp_backward_sweep.argtypes = [H, C.c_double, IP]
    lib.hsddp_linear_rollout.argtypes = [H, C.c_double, OP]
    lib.hsddp_update_nominal_trajectory.argtypes = [H]
    lib.hsddp_get_exp_cost_change.argtypes = [H, DP, DP]
    lib.hsddp_measure_dynamics_feasibility.argtypes = [H, DP]
    lib.hsddp_get_info.argtypes = [H, C.POINTER(Info)]
    lib.hsddp_get_field.argtypes = [H, C.c_int, C.c_int, C.c_int, C.c_int, DP]
    lib.hsddp_field_shape.argtypes = [H, C.c_int, C.c_int, IP, IP]
    lib.hsddp_get_solve_time_ms.argtypes = [H]
    lib.hsddp_get_solve_time_ms.restype = C.c_float
    lib.hsddp_get_kernel_times.argtypes = [H, C.c_int, DP, C.POINTER(C.c_longlong), C.c_char_p, C.c_int]
    lib.hsddp_get_kernel_units.argtypes = [H, C.c_char_p, C.POINTER(C.c_longlong)]
    lib.hsddp_reset_kernel_times.argtypes = [H]
    FP = C.POINTER(C.c_float)
    lib.hsddp_get_history.argtypes = [H, C.c_int, C.c_int, FP, FP, FP, FP, IP]
    lib.hsddp_export_mpc_command.argtypes = [H, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_uint)]
    lib.hsddp_warm_start_phase.argtypes = [H, C.c_int, H, C.c_int, C.c_int]
    lib.hsddp_reconfigure.argtypes = [H, C.c_int, C.POINTER(PhaseDesc), IP, IP]
    lib.hsddp_set_control_knot.argtypes = [H, C.c_int, C.c_int, C.c_void_p]
    lib.hsddp_export_solver_info.argtypes = [H, C.c_int, C.c_void_p]
    lib.hsddp_debug_malloc_count.argtypes = []
    lib.hsddp_debug_malloc_count.restype = C.c_longlong
    lib.hsddp_backend_name.argtypes = []
    lib.hsddp_backend_name.restype = C.c_char_p
    return lib


def _dp(a):
    return a.ctypes.data_as(DP)


class Solver:
    """Thin object wrapper over one hsddp handle of a bound library (either backend).

    Mirrors MultiPhaseDDP<T> (HSDDPSolver/header/MultiPhaseDDP.h:31-93): set_multiPhaseProblem happens in
    the constructor (descriptors instead of closures), then set_initial_condition / solve / get_*.
    """

    def __init__(self, lib, phases, batch=1, device=0, psi_dyn=3.1415, psi_kin=np.pi, precision=PREC_F64):
        self.lib = lib
        self.phases = phases            # list of dicts produced by problems.py (keeps numpy buffers alive)
        self.batch = batch
        n = len(phases)
        arr = (PhaseDesc * n)(*[p["desc"] for p in phases])
        mp = ModelParam(psi_dyn, psi_kin)
        self.h = C.c_void_p()
        rc = lib.hsddp_create_ex(C.byref(self.h), n, arr, C.byref(mp), batch, device, precision)
        if rc != 0:
            raise RuntimeError(f"hsddp_create_ex failed rc={rc}")
        self.precision = precision
        self.dims = [MODEL_DIMS[p["desc"].model] for p in phases]
        self.horizons = [p["desc"].horizon for p in phases]

    def close(self):
        if self.h:
            self.lib.hsddp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed rc={rc}")

    def set_initial_condition(self, x0):
        x0 = np.ascontiguousarray(np.broadcast_to(np.asarray(x0, dtype=np.float64), (self.batch, self.dims[0][0])))
        self._ck(self.lib.hsddp_set_initial_condition(self.h, _dp(x0)), "set_initial_condition")

    def set_nominal(self, phase, Xbar, Ubar):
        Xbar = np.ascontiguousarray(Xbar, dtype=np.float64)
        Ubar = np.ascontiguousarray(Ubar, dtype=np.float64)
        per = 1 if Xbar.ndim == 3 else 0
        self._ck(self.lib.hsddp_set_nominal(self.h, phase, _dp(Xbar), _dp(Ubar), per), "set_nominal")

    def solve(self, opt, max_cputime_ms=1e6):
        self._ck(self.lib.hsddp_solve(self.h, C.byref(opt), C.c_float(max_cputime_ms)), "solve")

    def hybrid_rollout(self, eps, opt):
        self._ck(self.lib.hsddp_hybrid_rollout(self.h, eps, C.byref(opt)), "hybrid_rollout")

    def compute_cost(self, opt):
        self._ck(self.lib.hsddp_compute_cost(self.h, C.byref(opt)), "compute_cost")

    def LQ_approximation(self, opt):
        self._ck(self.lib.hsddp_LQ_approximation(self.h, C.byref(opt)), "LQ_approximation")

    def backward_sweep(self, reg):
        ok = np.zeros(self.batch, dtype=np.int32)
        self._ck(self.lib.hsddp_backward_sweep(self.h, reg, ok.ctypes.data_as(IP)), "backward_sweep")
        return ok

    def linear_rollout(self, eps, opt):
        self._ck(self.lib.hsddp_linear_rollout(self.h, eps, C.byref(opt)), "linear_rollout")

    def update_nominal_trajectory(self):
        self._ck(self.lib.hsddp_update_nominal_trajectory(self.h), "update_nominal_trajectory")

    def get_exp_cost_change(self):
        a = np.zeros(self.batch)
        b = np.zeros(self.batch)
        self._ck(self.lib.hsddp_get_exp_cost_change(self.h, _dp(a), _dp(b)), "get_exp_cost_change")
        return a, b

    def measure_dynamics_feasibility(self):
        a = np.zeros(self.batch)
        self._ck(self.lib.hsddp_measure_dynamics_feasibility(self.h, _dp(a)), "measure_dynamics_feasibility")
        return a

    def get_info(self):
        info = (Info * self.batch)()
        self._ck(self.lib.hsddp_get_info(self.h, info), "get_info")
        return info

    def info_arrays(self):
        info = self.get_info()
        return {k: np.array([getattr(i, k) for i in info]) for k, _ in Info._fields_}

    def field(self, phase, name, b0=0, nb=None):
        """Trajectory field as numpy [nb, count, ...] with matrices in (rows, cols) numpy order."""
        nb = self.batch - b0 if nb is None else nb
        cnt, el = C.c_int(), C.c_int()
        fid = FIELD_ID[name]
        self._ck(self.lib.hsddp_field_shape(self.h, phase, fid, C.byref(cnt), C.byref(el)), "field_shape")
        out = np.zeros((nb, cnt.value, el.value))
        if out.size:
            self._ck(self.lib.hsddp_get_field(self.h, phase, fid, b0, nb, _dp(out)), "get_field")
        n, m, p = self.dims[phase]
        shp = {"K": (m, n), "QUX": (m, n), "QUU": (m, m), "A": (n, n), "B": (n, m), "C": (p, n), "D": (p, m),
               "LXX": (n, n), "LUX": (m, n), "LUU": (m, m), "LYY": (p, p), "PHIXX": (n, n), "H0": (n, n)}.get(name)
        if shp is not None and out.size:
            out = out.reshape(nb, cnt.value, shp[1], shp[0]).transpose(0, 1, 3, 2)  # column-major -> numpy
        return out

    def solve_time_ms(self):
        return float(self.lib.hsddp_get_solve_time_ms(self.h))

    def warm_start_phase(self, dphase, src, sphase, shift):
        """Receding-horizon shift of one phase's nominal trajectory from another solver of the same backend (device to device)."""
        self._ck(self.lib.hsddp_warm_start_phase(self.h, dphase, src.h if src is not None else None, sphase, shift), "warm_start_phase")

    def reconfigure(self, phases, src_phase, shift):
        """Receding-horizon update in place (include/hsddp.h hsddp_reconfigure): new phase table, warm start + constraint parameters moved
        inside the handle, allocations reused."""
        n = len(phases)
        arr = (PhaseDesc * n)(*[p["desc"] for p in phases])
        sp = np.ascontiguousarray(src_phase, dtype=np.int32); sh = np.ascontiguousarray(shift, dtype=np.int32)
        assert sp.size == n and sh.size == n
        self._ck(self.lib.hsddp_reconfigure(self.h, n, arr, sp.ctypes.data_as(IP), sh.ctypes.data_as(IP)), "reconfigure")
        self.phases = phases
        self.dims = [MODEL_DIMS[p["desc"].model] for p in phases]
        self.horizons = [p["desc"].horizon for p in phases]

    def export_solver_info(self, problem=0):
        """solver_info_lcmt content (MHPCLocomotion.cpp:74-79) of one problem: dict + the eight raw 32-bit words."""
        w = np.zeros(8, dtype=np.uint32)
        self._ck(self.lib.hsddp_export_solver_info(self.h, problem, w.ctypes.data_as(C.c_void_p)), "export_solver_info")
        f = w[3:].view(np.float32)
        return {"n_iter": int(w[0].view(np.int32)), "n_ls_iter": int(w[1].view(np.int32)), "n_reg_iter": int(w[2].view(np.int32)), "solve_time": float(f[0]), "cost": float(f[1]),
                "dyn_feas": float(f[2]), "ineq_violation": float(f[3]), "eq_violation": float(f[4]), "raw": w}

    def set_control_knot(self, phase, k, u=None):
        """Trajectory::Ubar[k] (and U[k]) of one phase for the whole batch; u: [batch, m] or None for zeros (HKDProblem.cpp:220)."""
        if u is not None:
            u = np.ascontiguousarray(u, dtype=np.float64); assert u.shape == (self.batch, self.dims[phase][1])
        self._ck(self.lib.hsddp_set_control_knot(self.h, phase, k, u.ctypes.data_as(C.c_void_p) if u is not None else None), "set_control_knot")

    CMD_FIELDS = (("mpc_times", 1, "f"), ("torque", 12, "f"), ("eul", 3, "f"), ("pos", 3, "f"), ("qJ", 12, "f"), ("vWorld", 3, "f"),
                  ("eulrate", 3, "f"), ("qJd", 12, "f"), ("GRF", 12, "f"), ("feedback", 432, "f"), ("Qu", 12, "f"), ("Quu", 144, "f"),
                  ("Qux", 432, "f"), ("contacts", 4, "i"), ("statusTimes", 4, "f"))

    def export_mpc_command(self, problem=0, n_steps=8, mpc_time=0.0, dt=0.01, status_times=None):
        """MHPC_Command_lcmt content (MHPCLocomotion.cpp:190-287) of one problem as a dict of fp32/int32 arrays + the raw words."""
        words = np.zeros(1 + n_steps * 1089, dtype=np.uint32)
        st = None
        if status_times is not None:
            st = np.ascontiguousarray(status_times, dtype=np.float32); assert st.shape == (len(self.phases), 4)
        rc = self.lib.hsddp_export_mpc_command(self.h, problem, n_steps, float(mpc_time), float(dt),
                                               st.ctypes.data_as(C.POINTER(C.c_float)) if st is not None else None,
                                               words.ctypes.data_as(C.POINTER(C.c_uint)))
        if rc != 0:
            raise RuntimeError(f"hsddp_export_mpc_command failed: {rc}")
        out = {"N_mpcsteps": int(words[0].view(np.int32)), "raw": words}
        pos = 1
        for name, w, kind in self.CMD_FIELDS:
            seg = words[pos:pos + n_steps * w]; pos += n_steps * w
            out[name] = seg.view(np.float32 if kind == "f" else np.int32).reshape(n_steps, w).copy()
        return out

    def get_history(self, problem=0, cap=4096):
        """MultiPhaseDDP::get_solver_info(cost, dyn_feas, eqn_feas, ineq_feas) (MultiPhaseDDP.h:85): the four float history buffers."""
        bufs = [np.zeros(cap, dtype=np.float32) for _ in range(4)]
        n = C.c_int()
        FP = C.POINTER(C.c_float)
        self._ck(self.lib.hsddp_get_history(self.h, problem, cap, *[b.ctypes.data_as(FP) for b in bufs], C.byref(n)), "get_history")
        m = min(n.value, cap)
        return {k: b[:m].copy() for k, b in zip(("cost", "dyn_feas", "eqn_feas", "ineq_feas"), bufs)}

    def kernel_units(self):
        out = {}
        for k in ("k_rollout", "k_lq", "k_sweep", "k_ls_probe"):
            v = C.c_longlong()
            if self.lib.hsddp_get_kernel_units(self.h, k.encode(), C.byref(v)) == 0:
                out[k] = int(v.value)
        return out

    def kernel_times(self, max_n=32):
        ms = np.zeros(max_n)
        cnt = np.zeros(max_n, dtype=np.int64)
        buf = C.create_string_buffer(2048)
        n = self.lib.hsddp_get_kernel_times(self.h, max_n, _dp(ms), cnt.ctypes.data_as(C.POINTER(C.c_longlong)), buf, 2048)
        names = [s.decode() for s in buf.raw.split(b"\0") if s][:n]
        return {names[i]: (float(ms[i]), int(cnt[i])) for i in range(min(n, len(names)))}
