"""Host-side problem builder: the counterpart of MHPCProblem<T>::initialization (MHPC/MHPC-Trajopt/MHPCProblem.cpp:14-249,
403-601) and of the gait-reference loader QuadReference (Reference/QuadReference.cpp:5-408), emitting the POD phase
descriptors of include/hsddp.h instead of SinglePhase objects with closures.

Everything the reference does in `float` is done in numpy float32 here (time accumulation, nearest-sample lookup,
std::stof parsing: SURVEY quirk vii), so that phase boundaries and reference indices come out identically.
"""
import json
import os
import re

import numpy as np

from ._abi import MODEL_WB, MODEL_SRB, Reb, Al, mhpc_ddp_setting
from . import problems

F32 = np.float32


# --------------------------------------------------------------------------------------------- settings files
def load_info(path):
    """boost::property_tree INFO file -> {section: {key: string}} (loadMHPCConfig MHPCProblem.h:66-84, load_reb_params ...)."""
    out, cur = {}, None
    for raw in open(path):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line == "{" or line == "}":
            if line == "}":
                cur = None
            continue
        toks = line.split()
        if len(toks) == 1:
            cur = toks[0]; out[cur] = {}
        elif cur is not None:
            out[cur][toks[0]] = toks[1]
    return out


def load_mhpc_config(path):
    c = load_info(path)["config"]
    return dict(plan_dur_wb=float(c["plan_dur_wb"]), plan_dur_srb=float(c["plan_dur_srb"]), dt_mpc=F32(float(c["dt_mpc"])),
                dt_wb=float(c["dt_wb"]), dt_srb=float(c["dt_srb"]), BG_alpha=float(F32(float(c["BG_alpha"]))),
                referenceFile=c["referenceFile"], costFile=c["costFile"], constraintParamFile=c["constraintParamFile"])


def load_constraint_params(path):
    p = load_info(path)
    reb = lambda n: Reb(float(p[n + "_ReB"]["delta"]), float(p[n + "_ReB"]["delta_min"]), float(p[n + "_ReB"]["eps"]))
    td = p["TD_AL"]
    return dict(grf=reb("GRF"), torque=reb("Torque"), jointspeed=reb("JointSpeed") if "JointSpeed_ReB" in p else reb("Joint"), joint=reb("Joint"),
                minheight=reb("MinHeight"), td=Al(float(td["sigma"]), float(td["lambda"]), float(td["sigma_max"])))


def load_cost_weights(path):
    """loadCostWeights (MHPCCostUtil.h:21-140): [q(36), r(12), qf(36)] for WB, [q(12), r(12), qf(12)] for SRB, 3-vectors per foot cost."""
    j = json.load(open(path))
    w = j["WB_Tracking_Cost"]; s = j["SRB_Tracking_Cost"]
    return dict(
        wb_q=list(w["qw_qB"]) + list(w["qw_qJ"]) * 4 + list(w["qw_vB"]) + list(w["qw_vJ"]) * 4, wb_r=[w["rw"]] * 12,
        wb_qf=list(w["qfw_qB"]) + list(w["qfw_qJ"]) * 4 + list(w["qfw_vB"]) + list(w["qfw_vJ"]) * 4,
        srb_q=list(s["qw_qB"]) + list(s["qw_vB"]), srb_r=[s["rw"]] * 12, srb_qf=list(s["qfw_qB"]) + list(s["qfw_vB"]),
        foot_reg=list(j["WB_FootPlace_Reg"]["qw_per_foot"]), swing_pos=list(j["Swing_Pos_Tracking"]["qw_per_foot"]),
        swing_vel=list(j["Swing_Vel_Tracking"]["qw_per_foot"]))


# --------------------------------------------------------------------------------------------- gait reference
class QuadReference:
    """QuadReference (Reference/QuadReference.h, .cpp): top-level gait data + nearest-sample lookups in float arithmetic."""
    FIELDS = {"body_state": 12, "jnt_angle": 12, "jnt_vel": 12, "foot_placements": 12, "foot_velocities": 12, "grf": 12, "torque": 12,
              "contact": 4, "status_dur": 4}

    def __init__(self, path, reorder=False):
        cols = {k: [] for k in self.FIELDS}
        cur = {k: np.zeros(w) for k, w in self.FIELDS.items()}
        lines = open(path).read().split("\n")
        i = 0
        self.dt = F32(0)
        while i < len(lines):
            line = lines[i]; i += 1
            if line == "dt":
                self.dt = F32(lines[i]); i += 1; continue
            key = next((k for k in ("body_state", "jnt_angle", "jnt_vel", "foot_placements", "foot_velocities", "foot_height", "grf", "torque",
                                    "contact", "status_dur") if k in line), None)     # same test order as load_top_level_data (:160-332)
            if key is None:
                continue
            words = lines[i].split(); i += 1
            if key == "foot_height":
                continue
            if key == "body_state":
                cur = {k: np.zeros(w) for k, w in self.FIELDS.items()}       # quad_state.SetZero()
            w = self.FIELDS[key]
            vals = [int(x) for x in words[:w]] if key == "contact" else [float(F32(x)) for x in words[:w]]      # std::stoi / std::stof
            cur[key][:len(vals)] = vals
            if key == "status_dur":
                for k in cols:
                    cols[k].append(cur[k].copy())
        self.tp = {k: np.array(v) for k, v in cols.items()}
        b = self.tp["body_state"]                                             # reorder_body_states: [eul,pos,omega,v] -> [pos,eul,v,omega]
        self.tp["body_state"] = np.concatenate([b[:, 3:6], b[:, 0:3], b[:, 9:12], b[:, 6:9]], axis=1)
        if reorder:                                                            # reorder_leg_dependent_states (:362-407)
            sw = lambda a: np.concatenate([a[:, 3:6], a[:, 0:3], a[:, 9:12], a[:, 6:9]], axis=1)
            for k in ("jnt_angle", "foot_placements", "foot_velocities", "grf", "torque"):
                self.tp[k] = sw(self.tp[k])
            self.tp["jnt_vel"] = np.zeros_like(self.tp["jnt_vel"])
            self.tp["contact"] = self.tp["contact"][:, [1, 0, 3, 2]]; self.tp["status_dur"] = self.tp["status_dur"][:, [1, 0, 3, 2]]
        self.k_cur = 0; self.sz = 0

    def __len__(self):
        return self.tp["contact"].shape[0]

    def initialize(self, plan_horizon):                                        # QuadReference::initialize (:5-21)
        self.k_cur = 0
        self.sz = int(round(float(F32(plan_horizon)) / float(self.dt))) + 1
        assert self.sz + 1 <= len(self), "reference file shorter than the planning horizon"

    def step(self, dt_sim):                                                    # QuadReference::step (:28-45)
        i = 1
        while float(F32(i) * self.dt) < float(F32(dt_sim)) or abs(float(F32(i) * self.dt) - float(F32(dt_sim))) <= 1e-6:
            self.k_cur += 1; i += 1

    def index(self, t):                                                        # get_a_reference_ptr_at_t (:63-76)
        t = F32(t)
        k = int(np.floor(t / self.dt))
        if float(F32(t - F32(k) * self.dt)) > 0.5 * float(self.dt):
            k += 1
        return min(k, self.sz - 1)

    def at(self, t):
        k = self.k_cur + self.index(t)
        return {n: self.tp[n][k] for n in self.tp}

    def contact_at(self, t):
        return self.tp["contact"][self.k_cur + self.index(t)].astype(np.int32)


def _approx_eq(a, b):      # HSDDP_Utils.h:46-56 (float tolerance, float error)
    return float(F32(abs(float(a) - float(b)))) <= float(F32(1e-6))


def _approx_leq(a, b):
    return float(a) < float(b) or _approx_eq(a, b)


# --------------------------------------------------------------------------------------------- MHPC problem
class MHPCProblemData:
    """Phase table of MHPCProblemData + the rules that evolve it: MHPCProblem::prepare_initialization (MHPCProblem.cpp:32-122)
    and the receding-horizon update (update_WB_plan / update_SRB_plan, :252-397).  `describe()` emits the phase descriptors of
    the current window; `update()` advances one MPC step and returns, per new phase, where every state / control slot comes
    from in the previous window (what pop_front / push_back_default do to the Trajectory deques, TrajectoryManagement.cpp:130-228)."""

    def __init__(self, ref, config, costs, cpar):
        self.ref, self.cfg, self.costs, self.cpar = ref, config, costs, cpar
        self.dt_wb, self.dt_srb = config["dt_wb"], config["dt_srb"]
        self.ref.initialize(F32(config["plan_dur_wb"] + config["plan_dur_srb"]))
        self.start, self.end, self.h, self.contact, self.dur, self.reach_end, self.has_td, self.shooting, self.uid = [], [], [], [], [], [], [], [], []
        self._next_uid = 0
        ref, cfg = self.ref, config
        if cfg["plan_dur_wb"] > 1e-5:                                          # MHPCProblem.cpp:69-108
            t = F32(0); start = F32(0)
            c_prev = ref.contact_at(t); d_prev = ref.at(t)["status_dur"].copy()
            while _approx_leq(t, cfg["plan_dur_wb"]):
                c_cur = ref.contact_at(t)
                if (c_cur != c_prev).any() or _approx_eq(t, cfg["plan_dur_wb"]):
                    end = t
                    self._push_phase(start, end, int(round(float(F32(end - start)) / self.dt_wb)), c_prev.copy(), d_prev.copy(), shooting=1)
                    self.reach_end[-1] = False      # (contact_prev != contact_prev).any(): always false (quirk viii)
                    c_prev = c_cur; d_prev = ref.at(t)["status_dur"].copy(); start = end
                t = F32(float(t) + self.dt_wb)
        self.srb_h = int(round(cfg["plan_dur_srb"] / self.dt_srb)) if cfg["plan_dur_srb"] > 1e-5 else 0
        self.srb_start = F32(cfg["plan_dur_wb"])
        for i in range(len(self.h)):                                           # add_tconstr_one_phase at initialisation (:198)
            self.has_td[i] = bool(self._touchdown(i).any())
        self.ref_start = F32(0)                                                # QuadReference::get_start_time

    def _push_phase(self, start, end, h, contact, dur, shooting):
        self.start.append(F32(start)); self.end.append(F32(end)); self.h.append(h); self.contact.append(contact); self.dur.append(dur)
        self.reach_end.append(False); self.has_td.append(False); self.shooting.append(shooting); self.uid.append(self._next_uid); self._next_uid += 1

    def _next_contact(self, i):                                                # update_resetmap / add_tconstr_one_phase (:524-540, 560-580)
        if i + 1 < len(self.h):
            return self.contact[i + 1]
        return self.ref.contact_at(F32(self.cfg["plan_dur_wb"] + float(self.cfg["dt_mpc"])))

    def _touchdown(self, i):
        nxt = self._next_contact(i)
        return np.array([1 if (self.contact[i][l] == 0 and nxt[l] == 1) else 0 for l in range(4)])

    # ---- receding-horizon update: one MPC step -----------------------------------------------------------
    def update(self):
        """MHPCProblem::update (:252-268).  Returns the slot maps {uid: (front_popped, pushed)} of surviving phases."""
        cfg, ref = self.cfg, self.ref
        old = {u: dict(h=h) for u, h in zip(self.uid, self.h)}
        nsteps = int(round(float(cfg["dt_mpc"]) / self.dt_wb))
        for _ in range(int(np.floor(float(cfg["dt_mpc"]) / float(ref.dt) + 1e-6)) if False else 0):
            pass
        k_before = ref.k_cur
        ref.step(cfg["dt_mpc"])
        self.ref_start = F32(float(self.ref_start) + (ref.k_cur - k_before) * float(ref.dt))
        popped = {u: 0 for u in self.uid}; pushed = {u: 0 for u in self.uid}
        if cfg["plan_dur_wb"] > 0:                                             # update_WB_plan (:271-352)
            new_start_time = self.ref_start
            for _ in range(nsteps):
                first = F32(float(self.start[0]) + self.dt_wb)
                if _approx_eq(self.end[0], first):
                    for lst in (self.start, self.end, self.h, self.contact, self.dur, self.reach_end, self.has_td, self.shooting, self.uid):
                        lst.pop(0)
                else:
                    popped[self.uid[0]] += 1; self.h[0] -= 1; self.start[0] = first
            for _ in range(nsteps):
                new_end = F32(float(self.end[-1]) + self.dt_wb)
                t_rel = F32(new_end - new_start_time)
                new_contact = ref.contact_at(t_rel)
                change = bool((new_contact != self.contact[-1]).any())
                if change and self.reach_end[-1]:
                    self._push_phase(self.end[-1], new_end, 1, new_contact.copy(), ref.at(t_rel)["status_dur"].copy(), shooting=0)   # SS_set empty
                    popped[self.uid[-1]] = 0; pushed[self.uid[-1]] = 0
                else:
                    self.end[-1] = new_end; self.h[-1] += 1
                    if change:
                        self.reach_end[-1] = True; self.has_td[-1] = bool(self._touchdown(len(self.h) - 1).any())
                    pushed[self.uid[-1]] += 1
            n = len(self.h)
            for i in range(n):                                                 # "other updates" (:340-351)
                if i < n - 1 or self.h[i] > nsteps:
                    self.shooting[i] = 1
        if cfg["plan_dur_srb"] > 0:                                            # update_SRB_plan (:355-377): nsteps = floor(dt_mpc / dt_srb) pop/push pairs
            self.srb_steps = int(np.floor(float(cfg["dt_mpc"]) / self.dt_srb + 1e-6))
            self.srb_start = F32(float(self.ref_start) + cfg["plan_dur_wb"])
        return {u: (popped[u], pushed[u], old[u]["h"] if u in old else None) for u in self.uid}

    # ---- descriptors of the current window --------------------------------------------------------------------
    def describe(self, ubar_mode="zero"):
        cfg, ref, costs, cpar = self.cfg, self.ref, self.costs, self.cpar
        dt_wb, dt_srb = self.dt_wb, self.dt_srb
        n_wb = len(self.h)
        phases = []
        for i in range(n_wb):
            h = self.h[i]
            nxt = self._next_contact(i)
            t_off = float(F32(self.start[i] - self.start[0]))
            xr = np.zeros((h + 1, 36)); ur = np.zeros((h + 1, 12)); yr = np.zeros((h + 1, 12)); fp = np.zeros((h + 1, 12)); fv = np.zeros((h + 1, 12))
            bp = np.zeros((h + 1, 3)); rc = np.zeros((h + 1, 4), dtype=np.int32); X0 = np.zeros((h + 1, 36))
            for k in range(h + 1):
                a = ref.at(F32(t_off + k * dt_wb))                             # cost / constraint lookups at t_offset + k dt (SinglePhase.cpp:243,298)
                b = a["body_state"]
                xr[k] = np.concatenate([b[:6], a["jnt_angle"], b[6:], a["jnt_vel"]])  # WBReference::get_reference_at_t (MHPCReference.cpp:24-39)
                ur[k] = a["torque"]; yr[k] = a["grf"]; fp[k] = a["foot_placements"]; fv[k] = a["foot_velocities"]; bp[k] = b[:3]; rc[k] = a["contact"]
                a0 = ref.at(F32(float(F32(self.start[i] - self.ref_start)) + k * dt_wb))   # initial guess lookup (MHPCProblem.cpp:186-193)
                b0 = a0["body_state"]
                X0[k] = np.concatenate([b0[:6], a0["jnt_angle"], b0[6:], a0["jnt_vel"]])
            refs = dict(xr=xr, ur=ur, yr=yr, foot_pos=fp, foot_vel=fv, body_pos=bp, ref_contact=rc)
            last = i == n_wb - 1
            ph = problems.wb_phase(h, dt_wb, t_off, self.contact[i], nxt, refs, next_model=MODEL_SRB if (last and self.srb_h > 0) else MODEL_WB,
                                   bg_alpha=cfg["BG_alpha"], shooting=self.shooting[i], ubar_mode="zero")
            d = ph["desc"]
            problems._set(d.q, costs["wb_q"]); problems._set(d.r, costs["wb_r"]); problems._set(d.qf, costs["wb_qf"])
            problems._set(d.w_foot_reg, costs["foot_reg"]); problems._set(d.w_swing_pos, costs["swing_pos"]); problems._set(d.w_swing_vel, costs["swing_vel"])
            d.reb_torque, d.reb_joint, d.reb_minheight, d.reb_grf, d.al_td = cpar["torque"], cpar["joint"], cpar["minheight"], cpar["grf"], cpar["td"]
            if not self.has_td[i]:                                             # no WBTouchDown / TDVelocityPenalty object on this phase (yet)
                d.c_touchdown = 0; d.w_td_vel = -1.0
            ph["Xbar"] = X0; ph["uid"] = self.uid[i]
            if ubar_mode == "gravity_comp":
                ph["Ubar"][:] = problems.wb_gravity_comp_torque(X0[0, :18], self.contact[i])
            phases.append(ph)
        if self.srb_h > 0:                                                     # MHPCProblem.cpp:216-247, 488-521
            h = self.srb_h; t_off = float(F32(self.srb_start - self.ref_start))
            xr = np.zeros((h + 1, 12)); ur = np.zeros((h + 1, 12)); fp = np.zeros((h + 1, 12)); rc = np.zeros((h + 1, 4), dtype=np.int32); bp = np.zeros((h + 1, 3))
            for k in range(h + 1):
                a = ref.at(F32(t_off + k * dt_srb))
                xr[k] = a["body_state"]; ur[k] = a["grf"]; fp[k] = a["foot_placements"]; rc[k] = a["contact"]; bp[k] = a["body_state"][:3]
            ph = problems.srb_phase(h, dt_srb, t_off, dict(xr=xr, ur=ur, foot_pos=fp, foot_vel=np.zeros((h + 1, 12)), body_pos=bp, ref_contact=rc))
            d = ph["desc"]
            problems._set(d.q, costs["srb_q"]); problems._set(d.r, costs["srb_r"]); problems._set(d.qf, costs["srb_qf"])
            d.reb_minheight = cpar["minheight"]
            ph["Xbar"] = xr.copy(); ph["uid"] = -1
            if ubar_mode == "gravity_comp":
                ph["Ubar"] = ur[:h].copy()
            phases.append(ph)
        info = dict(start_times=[float(s) for s in self.start], end_times=[float(e) for e in self.end], horizons=list(self.h),
                    contacts=[np.asarray(c).tolist() for c in self.contact], shooting=list(self.shooting), has_td=list(self.has_td),
                    status_durations=np.array(self.dur, dtype=np.float32) if self.dur else np.zeros((0, 4), np.float32), srb_horizon=self.srb_h,
                    x0=phases[0]["Xbar"][0].copy())
        return phases, info


def build_mhpc_problem(ref, config, costs, cpar, ubar_mode="zero"):
    """MHPCProblem::prepare_initialization + initialize_multiPhaseProblem at the start of the reference.
    Returns (phases, info) where info holds the phase table of MHPCProblemData (start/end times, horizons, contacts, durations)."""
    return MHPCProblemData(ref, config, costs, cpar).describe(ubar_mode=ubar_mode)


def build_from_tree(root, gait=None, ubar_mode="zero"):
    """Convenience: read MHPC/settings/{mhpc_config.info, cost/constraint files} and Reference/Data/<gait>/quad_reference.csv
    below a CAFE-MPC checkout `root`, like MHPCLocomotion::initialize (MHPC/MHPCLocomotion.cpp:20-40)."""
    cfg = load_mhpc_config(os.path.join(root, "MHPC", "settings", "mhpc_config.info"))
    costs = load_cost_weights(os.path.join(root, cfg["costFile"]))
    cpar = load_constraint_params(os.path.join(root, cfg["constraintParamFile"]))
    ref = QuadReference(os.path.join(root, "Reference", "Data", gait or cfg["referenceFile"], "quad_reference.csv"), reorder=False)
    phases, info = build_mhpc_problem(ref, cfg, costs, cpar, ubar_mode=ubar_mode)
    return phases, info, cfg


def load_ddp_setting(path):
    """loadHSDDPSetting (HSDDP_CompoundTypes.h:62-81): update_regularization and smooth_active are NOT read (quirk xiii)."""
    d = load_info(path)["ddp"]
    tf = lambda s: 1 if s.strip().lower() == "true" else 0
    return mhpc_ddp_setting(alpha=float(d["alpha"]), gamma=float(d["gamma"]), update_penalty=float(d["update_penalty"]), update_relax=float(d["update_relax"]),
                            update_ReB=float(d["update_ReB"]), max_DDP_iter=int(d["max_DDP_iter"]), max_AL_iter=int(d["max_AL_iter"]),
                            max_DDP_iter_runtime=int(d["max_DDP_iter_runtime"]), max_AL_iter_runtime=int(d["max_AL_iter_runtime"]),
                            cost_thresh=float(d["cost_thresh"]), tconstr_thresh=float(d["tconstr_thresh"]), pconstr_thresh=float(d["pconstr_thresh"]),
                            dynamics_feas_thresh=float(d["dynamics_feas_thresh"]), merit_rho=float(d["merit_rho"]), merit_scale=float(d["merit_scale"]),
                            merit_offset=float(d["merit_offset"]), AL_active=tf(d["AL_active"]), ReB_active=tf(d["ReB_active"]), MS=tf(d["MS"]),
                            nsteps_per_node=int(d["nsteps_per_node"]))


def shift_solver(solver_cls, lib, old_solver, old_phases, pd, slot_map, ubar_mode="zero", **solver_kw):
    """One receding-horizon step on the solver side (what MHPCLocomotion::update, MHPC/MHPCLocomotion.cpp:109-122, gets from
    MHPCProblem::update): build the solver of the shifted window and move the nominal trajectories over, device to device.
    `slot_map` is the return value of pd.update().  Returns (new_solver, new_phases, info)."""
    phases, info = pd.describe(ubar_mode=ubar_mode)
    new = solver_cls(lib, phases, batch=old_solver.batch, **solver_kw)
    old_index = {p.get("uid"): i for i, p in enumerate(old_phases)}
    for i, p in enumerate(phases):
        uid = p.get("uid")
        if uid == -1:                                  # SRB tail: update_SRB_plan pops/pushes floor(dt_mpc/dt_srb) knots (0 for the shipped config)
            new.warm_start_phase(i, old_solver, old_index[-1], getattr(pd, "srb_steps", 0))
        elif uid in old_index:
            new.warm_start_phase(i, old_solver, old_index[uid], slot_map[uid][0])
        else:
            new.warm_start_phase(i, None, -1, 0)       # phase created by the update: zero trajectory (Trajectory constructor)
    return new, phases, info


def shift_solver_in_place(solver, old_phases, pd, slot_map, ubar_mode="zero"):
    """The same receding-horizon step INSIDE the solver's handle (hsddp_reconfigure): new phase table, trajectories and constraint
    parameters moved device to device, allocations reused (no hipMalloc per tick), solver counters kept.  Returns (phases, info)."""
    phases, info = pd.describe(ubar_mode=ubar_mode)
    old_index = {p.get("uid"): i for i, p in enumerate(old_phases)}
    src, shift = [], []
    for p in phases:
        uid = p.get("uid")
        if uid == -1:
            src.append(old_index[-1]); shift.append(getattr(pd, "srb_steps", 0))
        elif uid in old_index:
            src.append(old_index[uid]); shift.append(slot_map[uid][0])
        else:
            src.append(-1); shift.append(0)
    solver.reconfigure(phases, src, shift)
    return phases, info


# --------------------------------------------------------------------------------------------- HKD-MPC problem
def load_hkd_constraint_params(path):
    p = load_info(path)
    reb = lambda n: Reb(float(p[n + "_ReB"]["delta"]), float(p[n + "_ReB"]["delta_min"]), float(p[n + "_ReB"]["eps"]))
    td = p["TD_AL"]
    return dict(grf=reb("GRF"), swing=reb("Swing"), td=Al(float(td["sigma"]), float(td["lambda"]), float(td["sigma_max"])))


class HKDProblemData:
    """Phase table of HKDProblemData + the rules that evolve it: HKDProblem::initialization (HKDMPC/HKD-TrajOpt/HKDProblem.cpp:14-111) with the
    constants of HKDMPCSolver::initialize (HKDMPC/HKDMPC.cpp:26-29), and the receding-horizon HKDProblem::update (:117-222).  `ref` must have
    been loaded with reorder=True (HKDMPC.h:32: legs FR FL HR HL, qJd zeroed).  `describe()` emits the descriptors of the current window,
    `update()` advances one MPC step (nsteps_between_mpc simulation steps) and returns the slot map {uid: (front_popped, pushed, old_h)} of
    the surviving phases - the same contract as MHPCProblemData, so shift_solver_in_place / hsddp_reconfigure serve both."""

    def __init__(self, ref, cpar, plan_duration=0.6, dt_sim=0.01, nsteps_between_mpc=2):
        self.ref, self.cpar = ref, cpar
        self.plan = F32(plan_duration); self.dt = F32(dt_sim); self.nsteps = int(nsteps_between_mpc); self.dt_mpc = F32(self.dt * F32(nsteps_between_mpc))
        ref.initialize(self.plan)
        self.start, self.end, self.h, self.contact, self.reach_end, self.has_td, self.shooting, self.uid = [], [], [], [], [], [], [], []
        self._next_uid = 0
        self.ref_start = F32(0)                                                # QuadReference::get_start_time
        self.dup_td = 0                                                        # phases that would carry TWO TouchDownConstraint objects in the reference (see update)
        t = F32(0); start = F32(0)
        c_prev = ref.contact_at(t)
        while _approx_leq(t, self.plan):                                       # HKDProblem.cpp:34-63
            c_cur = ref.contact_at(t)
            if (c_cur != c_prev).any() or (float(t) > float(self.plan) or _approx_eq(t, self.plan)):
                end = t
                self._push_phase(start, end, int(round(float(F32(end - start) / self.dt))), c_prev.copy(), shooting=1)      # reach_end: (c != c).any() = false (:51)
                c_prev = c_cur; start = end
            t = F32(t + self.dt)
        for i in range(len(self.h)):                                           # add_tconstr_one_phase for every phase at initialisation (:93)
            self.has_td[i] = bool(self._touchdown(i).any())

    def _push_phase(self, start, end, h, contact, shooting):
        self.start.append(F32(start)); self.end.append(F32(end)); self.h.append(h); self.contact.append(contact); self.reach_end.append(False)
        self.has_td.append(False); self.shooting.append(shooting); self.uid.append(self._next_uid); self._next_uid += 1

    def _next_contact(self, i):                                                # add_tconstr_one_phase (:283-291)
        if i + 1 < len(self.h):
            return self.contact[i + 1]
        return self.ref.contact_at(F32(self.plan + self.dt_mpc))

    def _touchdown(self, i):
        nxt = self._next_contact(i)
        return np.array([1 if (self.contact[i][l] == 0 and nxt[l] == 1) else 0 for l in range(4)])

    def update(self):
        """HKDProblem::update (:117-222): per simulation step the reference steps, the front phase loses a knot (or disappears), the back either
        grows by a knot or - once its end has been seen (is_phase_reach_end) - a new one-knot phase is appended; afterwards every phase but a
        last one of at most two knots gets its shooting nodes (:211-216) and the first control of the window is zeroed (:218, the caller does
        that through hsddp_set_control_knot).  Deviation (documented, counted in dup_td): add_tconstr_one_phase runs again on a phase that
        already got its TouchDownConstraint at initialisation (the initial LAST phase, when a touchdown follows it within dt_mpc of the
        horizon end), which leaves two identical constraint objects on it in the reference; here a phase has the constraint once."""
        ref = self.ref
        old = {u: h for u, h in zip(self.uid, self.h)}
        popped = {u: 0 for u in self.uid}; pushed = {u: 0 for u in self.uid}
        for _ in range(self.nsteps):
            k_before = ref.k_cur
            ref.step(self.dt)
            self.ref_start = F32(float(self.ref_start) + (ref.k_cur - k_before) * float(ref.dt))
            new_start = self.ref_start; new_end = F32(new_start + self.plan)
            self.start[0] = F32(self.start[0] + self.dt)                       # front end (:131-146)
            if _approx_leq(self.end[0], new_start):
                for lst in (self.start, self.end, self.h, self.contact, self.reach_end, self.has_td, self.shooting, self.uid):
                    lst.pop(0)
            else:
                popped[self.uid[0]] += 1; self.h[0] -= 1; self.start[0] = new_start
            new_contact = ref.contact_at(F32(new_end - new_start))             # back end (:149-199)
            change = bool((new_contact != self.contact[-1]).any())
            if change and self.reach_end[-1]:
                nstart = self.end[-1]
                self._push_phase(nstart, new_end, int(round(float(F32(new_end - nstart) / self.dt))), new_contact.copy(), shooting=0)      # no update_SS_config yet
                popped[self.uid[-1]] = 0; pushed[self.uid[-1]] = 0
            else:
                self.end[-1] = new_end; self.h[-1] += 1
                if change:
                    self.reach_end[-1] = True
                pushed[self.uid[-1]] += 1
            if self.reach_end[-1]:                                             # add_tconstr_one_phase on the last phase (:201-204)
                td = bool(self._touchdown(len(self.h) - 1).any())
                if td and self.has_td[-1]:
                    self.dup_td += 1
                self.has_td[-1] = self.has_td[-1] or td
        n = len(self.h)
        for i in range(n):                                                     # :207-216
            if i < n - 1 or self.h[i] > 2:
                self.shooting[i] = 1
        return {u: (popped[u], pushed[u], old.get(u)) for u in self.uid}

    @staticmethod
    def hkd_x(a):                                                              # HKDSinglePhaseReference::get_reference_at_t (HKDReference.cpp:23-61)
        b = a["body_state"]
        q = np.concatenate([a["foot_placements"][3 * l:3 * l + 3] if a["contact"][l] > 0 else a["jnt_angle"][3 * l:3 * l + 3] for l in range(4)])
        return np.concatenate([b[3:6], b[0:3], b[9:12], b[6:9], q])

    def describe(self, ubar_mode=None):      # (ubar_mode: signature of MHPCProblemData.describe; the kinodynamic nominal controls start at zero, HKDProblem.cpp:75)
        ref, dt, cpar = self.ref, self.dt, self.cpar
        phases = []
        n = len(self.h)
        for i in range(n):
            h = self.h[i]
            nxt = self._next_contact(i)
            t_off = float(F32(self.start[i] - self.start[0]))
            xr = np.zeros((h + 1, 24)); ur = np.zeros((h + 1, 24)); fp = np.zeros((h + 1, 12)); bp = np.zeros((h + 1, 3)); rc = np.zeros((h + 1, 4), dtype=np.int32)
            X0 = np.zeros((h + 1, 24))
            for k in range(h + 1):
                a = ref.at(F32(t_off + k * float(dt)))
                xr[k] = self.hkd_x(a); ur[k] = np.concatenate([a["grf"], a["jnt_vel"]]); fp[k] = a["foot_placements"]; bp[k] = a["body_state"][:3]; rc[k] = a["contact"]
                X0[k] = self.hkd_x(ref.at(F32(float(F32(self.start[i] - self.ref_start)) + k * float(dt))))      # initial guess (:80-85; only used by the first window)
            ph = problems.hkd_phase(h, float(dt), t_off, self.contact[i], nxt, dict(xr=xr, ur=ur, foot_pos=fp, foot_vel=np.zeros((h + 1, 12)), body_pos=bp, ref_contact=rc))
            d = ph["desc"]
            d.reb_grf = cpar["grf"]; d.al_td = cpar["td"]; d.shooting = self.shooting[i]
            if not self.has_td[i]:
                d.c_touchdown = 0
            ph["Xbar"] = X0; ph["Ubar"] = np.zeros((h, 24)); ph["uid"] = self.uid[i]
            phases.append(ph)
        info = dict(start_times=[float(s) for s in self.start], end_times=[float(e) for e in self.end], horizons=list(self.h), contacts=[np.asarray(c).tolist() for c in self.contact],
                    shooting=list(self.shooting), has_td=list(self.has_td), x0=phases[0]["Xbar"][0].copy())
        return phases, info


def build_hkd_problem(ref, cpar, plan_duration=0.6, dt_sim=0.01, nsteps_between_mpc=2):
    """HKDProblem::initialization (HKDMPC/HKD-TrajOpt/HKDProblem.cpp:14-97) at the start of the reference.  Returns (phases, info)."""
    return HKDProblemData(ref, cpar, plan_duration, dt_sim, nsteps_between_mpc).describe()


def hkd_next_footholds(solver, contacts, problem=0):
    """HKDMPCSolver::update_foot_placement (HKDMPC/HKDMPC.cpp:207-240): for every leg the qdummy entries at the start of the first
    phase it lands in (pattern 0 -> 1 along the contact sequence, search stops after the fifth phase pair).  Returns {leg: pf(3) float32}."""
    out = {}
    n = len(contacts)
    for i in range(n - 1):
        for l in range(4):
            if l not in out and contacts[i][l] == 0 and contacts[i + 1][l] == 1:
                out[l] = solver.field(i + 1, "XBAR", b0=problem, nb=1)[0, 0, 12 + 3 * l:15 + 3 * l].astype(np.float32)
        if i >= 4:
            break
    return out
