"""Synthetic Mini-Cheetah problem descriptors for the BASELINE.json configurations.

Host-side counterpart of what MHPCProblem (MHPC/MHPC-Trajopt/MHPCProblem.cpp:88-137, 174-249, 403-601)
binds into SinglePhase objects, emitted as POD phase descriptors (include/hsddp.h).  Inputs are synthetic
(no CSV gait file): weights from MHPC/settings/cost_weights_regular.JSON, ReB/AL parameters from
MHPC/settings/constraint_params_regular.info, limits from MHPCConstraint.{h,cpp}.
"""
import ctypes as C
import numpy as np

from ._abi import PhaseDesc, Reb, Al, MODEL_WB, MODEL_SRB, MODEL_HKD, MODEL_DIMS, DP, IP

QJ_NOM = np.array([0.0, -1.0, 2.0] * 4)            # Loco_TO.cpp:53
Z_NOM = 0.2183                                      # Loco_TO.cpp:54
MASS_WB = 8.252                                     # URDF link-mass sum


def wb_foot_positions(q, psi=np.pi):
    """World foot positions (4x3, FL FR HL HR) of the 18-dof model (PinocchioInteface.cpp:17-56 + URDF)."""
    def rx(a): c, s = np.cos(a), np.sin(a); return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    def ry(a): c, s = np.cos(a), np.sin(a); return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    def rz(a): c, s = np.cos(a), np.sin(a); return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    R = rz(q[3]) @ ry(q[4]) @ rx(q[5])
    p = np.asarray(q[:3], dtype=float)
    out = np.zeros((4, 3))
    for l, (sx, sy) in enumerate([(1, 1), (1, -1), (-1, 1), (-1, -1)]):
        qa, qh, qk = q[6 + 3 * l: 9 + 3 * l]
        Ra = R @ rx(qa); pa = p + R @ np.array([sx * 0.19, sy * 0.049, 0])
        Rh = Ra @ rz(psi) @ ry(qh); ph = pa + Ra @ np.array([0, sy * 0.062, 0])
        Rk = Rh @ ry(qk); pk = ph + Rh @ np.array([0, 0, -0.209])
        out[l] = pk + Rk @ np.array([0, 0, -0.195])
    return out


def wb_gravity_comp_torque(q, contact, psi=np.pi):
    """Quasi-static joint torques tau = -J_leg^T F for stance legs sharing the weight equally (builder-side
    initial control guess; the reference's tests start from Ubar = 0, testMHPCProblem.cpp:70-76)."""
    nc = max(1, int(np.sum(contact)))
    F = np.array([0.0, 0.0, MASS_WB * 9.81 / nc])
    u = np.zeros(12)
    e = 1e-6
    for l in range(4):
        if not contact[l]:
            continue
        for j in range(3):
            qp = np.array(q, dtype=float); qm = np.array(q, dtype=float)
            qp[6 + 3 * l + j] += e; qm[6 + 3 * l + j] -= e
            dp = (wb_foot_positions(qp, psi)[l] - wb_foot_positions(qm, psi)[l]) / (2 * e)
            u[3 * l + j] = -dp @ F
    return u


class SplitMix64:
    """splitmix64 -> double in [0,1): the portable PRNG SURVEY 8(d) prescribes for the x0 ensemble."""
    M = (1 << 64) - 1

    def __init__(self, seed):
        self.s = seed & self.M

    def skip(self, n):
        """Advance the stream by n draws (splitmix64's state is a counter: any shard can jump to its own slice)."""
        self.s = (self.s + n * 0x9E3779B97F4A7C15) & self.M

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & self.M
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.M
        z ^= z >> 31
        return (z >> 11) * (1.0 / 9007199254740992.0)


def wb_nominal_state():
    return np.concatenate([[0, 0, Z_NOM, 0, 0, 0], QJ_NOM, np.zeros(18)])


def wb_ensemble_x0(batch, seed, first=0):
    """x0 = x_nom + uniform delta: pos 0.02, eul 0.05, qJ 0.1, v 0.1, eul-rate 0.2, qJd 0.5.
    Problem b draws its 36 numbers from stream position b*36 (so any shard reproduces its own slice)."""
    amp = np.concatenate([[0.02] * 3, [0.05] * 3, [0.1] * 12, [0.1] * 3, [0.2] * 3, [0.5] * 12])
    xn = wb_nominal_state()
    out = np.zeros((batch, 36))
    rng = SplitMix64(seed)
    rng.skip(first * 36)
    for b in range(batch):
        u = np.array([rng.next() for _ in range(36)])
        out[b] = xn + amp * (2 * u - 1)
    return out


def _set(arr, vals):
    for i, v in enumerate(vals):
        arr[i] = v


def wb_phase(horizon, dt, t_offset, contact, next_contact, refs, next_model=MODEL_WB, bg_alpha=10.0, shooting=1,
             ubar_mode="gravity_comp"):
    """One WB phase descriptor with the shipped 'regular' weights/limits. refs: dict of per-knot arrays."""
    d = PhaseDesc()
    d.model, d.horizon, d.dt, d.t_offset = MODEL_WB, horizon, dt, t_offset
    _set(d.contact, contact); _set(d.next_contact, next_contact)
    d.next_model, d.shooting, d.BG_alpha = next_model, shooting, bg_alpha
    # cost_weights_regular.JSON -> [q, r, qf] in loadCostWeights order (MHPCCostUtil.h:21-78)
    q = [0, 0, 10, 1, 2, 2] + [1.0] * 12 + [2, 2, 2, 1, 2, 2] + [0.01] * 12
    qf = [0, 0, 1, 1, 1, 1] + [0.5] * 12 + [1.0] * 6 + [0.01] * 12
    _set(d.q, q); _set(d.r, [0.1] * 12); _set(d.qf, qf)
    _set(d.w_foot_reg, [20.0, 20.0, 1.0]); _set(d.w_swing_pos, [10.0, 10.0, 10.0]); _set(d.w_swing_vel, [2.0, 2.0, 2.0])
    d.w_td_vel = 1.0
    d.c_torque = d.c_joint = d.c_minheight = d.c_grf = 1
    d.torque_limit = 17.0
    _set(d.joint_lb, [-1.3, -5.0, -np.pi]); _set(d.joint_ub, [1.3, 5.0, np.pi])
    d.h_min, d.mu = 0.20, 0.6
    d.reb_grf = Reb(0.1, 0.1, 0.3); d.reb_torque = Reb(0.1, 0.1, 0.1); d.reb_joint = Reb(0.1, 0.1, 0.1)
    d.reb_minheight = Reb(0.01, 0.01, 0.1)
    d.c_touchdown, d.ground_height = 1, 0.0
    d.al_td = Al(10.0, 0.0, 1e4)
    bufs = {}
    n, m, p = MODEL_DIMS[MODEL_WB]
    for name, w in (("xr", n), ("ur", m), ("yr", p), ("foot_pos", 12), ("foot_vel", 12), ("body_pos", 3)):
        a = np.ascontiguousarray(refs[name], dtype=np.float64)
        assert a.shape == (horizon + 1, w), (name, a.shape)
        bufs[name] = a
        setattr(d, name, a.ctypes.data_as(DP))
    rc = np.ascontiguousarray(refs["ref_contact"], dtype=np.int32)
    assert rc.shape == (horizon + 1, 4)
    bufs["ref_contact"] = rc
    d.ref_contact = rc.ctypes.data_as(IP)
    ubar = np.zeros((horizon, m))
    if ubar_mode == "gravity_comp":  # "zero" reproduces testMHPCProblem.cpp:70-76
        ubar[:] = wb_gravity_comp_torque(bufs["xr"][0, :18], contact)
    return {"desc": d, "bufs": bufs, "Xbar": bufs["xr"].copy(), "Ubar": ubar}


def wb_trot_problem(schedule=((1, 1, 1, 1), (0, 1, 1, 0), (1, 0, 0, 1), (0, 1, 1, 0)), horizons=(50, 50, 50, 50),
                    dt=0.01, vx=0.5, last_next=(1, 0, 0, 1), swing_height=0.06, ubar_mode="gravity_comp"):
    """Configs 2/3 of BASELINE.json: WB, 4 contact phases x 50 knots, trot-like."""
    nph = len(schedule)
    starts = np.concatenate([[0], np.cumsum(horizons)])
    nominal_feet = wb_foot_positions(wb_nominal_state()[:18])   # relative to body at origin
    nominal_feet[:, 2] = 0.0

    def body_x(t):
        return vx * t

    # foothold of foot f while in stance during phase i: under the nominal foot at mid-phase
    def foothold(i, f):
        tm = (starts[i] + 0.5 * horizons[i]) * dt
        return nominal_feet[f] + np.array([body_x(tm), 0, 0])

    phases = []
    for i in range(nph):
        h = horizons[i]
        nxt = schedule[i + 1] if i + 1 < nph else last_next
        xr = np.zeros((h + 1, 36)); fp = np.zeros((h + 1, 12)); fv = np.zeros((h + 1, 12)); rc = np.zeros((h + 1, 4), dtype=np.int32)
        yr = np.zeros((h + 1, 12))
        for k in range(h + 1):
            t = (starts[i] + k) * dt
            xr[k, 0] = body_x(t); xr[k, 2] = Z_NOM; xr[k, 6:18] = QJ_NOM; xr[k, 18] = vx
            c = schedule[i] if k < h else nxt          # reference lookup at the knot's absolute time
            rc[k] = c
            nc = max(1, int(np.sum(c)))
            for f in range(4):
                if schedule[i][f]:
                    fp[k, 3 * f:3 * f + 3] = foothold(i, f)
                else:
                    # swing from the previous foothold to the next one over this phase
                    p0 = foothold(i - 1, f) if i > 0 else foothold(i, f) - np.array([vx * h * dt, 0, 0])
                    p1 = foothold(i + 1, f) if i + 1 < nph else foothold(i, f) + np.array([vx * h * dt, 0, 0])
                    s = k / h
                    fp[k, 3 * f:3 * f + 3] = p0 + (p1 - p0) * s + np.array([0, 0, swing_height * np.sin(np.pi * s)])
                    fv[k, 3 * f:3 * f + 3] = (p1 - p0) / (h * dt) + np.array([0, 0, swing_height * np.pi / (h * dt) * np.cos(np.pi * s)])
                if c[f]:
                    yr[k, 3 * f + 2] = MASS_WB * 9.81 / nc
        refs = dict(xr=xr, ur=np.zeros((h + 1, 12)), yr=yr, foot_pos=fp, foot_vel=fv, body_pos=xr[:, :3].copy(), ref_contact=rc)
        phases.append(wb_phase(h, dt, starts[i] * dt, schedule[i], nxt, refs, ubar_mode=ubar_mode))
    return phases


def wb_stance_problem(horizon=50, dt=0.01, contact=(1, 1, 1, 1), ubar_mode="gravity_comp"):
    """Config 1 of BASELINE.json: WB, one phase N=50, constant stance reference."""
    h = horizon
    xn = wb_nominal_state()
    feet = wb_foot_positions(xn[:18]); feet[:, 2] = 0.0
    xr = np.tile(xn, (h + 1, 1))
    rc = np.tile(np.array(contact, dtype=np.int32), (h + 1, 1))
    nc = max(1, int(np.sum(contact)))
    yr = np.zeros((h + 1, 12))
    for f in range(4):
        if contact[f]:
            yr[:, 3 * f + 2] = MASS_WB * 9.81 / nc
    refs = dict(xr=xr, ur=np.zeros((h + 1, 12)), yr=yr, foot_pos=np.tile(feet.reshape(-1), (h + 1, 1)),
                foot_vel=np.zeros((h + 1, 12)), body_pos=xr[:, :3].copy(), ref_contact=rc)
    return [wb_phase(h, dt, 0.0, contact, contact, refs, ubar_mode=ubar_mode)]


def srb_phase(horizon, dt, t_offset, refs):
    """SRB tail phase (MHPCProblem.cpp:488-521) with cost_weights_regular.JSON SRB weights."""
    d = PhaseDesc()
    d.model, d.horizon, d.dt, d.t_offset = MODEL_SRB, horizon, dt, t_offset
    d.next_model, d.shooting, d.BG_alpha = -1, 1, 0.0
    _set(d.q, [0, 0, 10, 1, 2, 2, 2, 2, 2, 1, 2, 2]); _set(d.r, [0.01] * 12); _set(d.qf, [0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1])
    _set(d.w_foot_reg, [-1, 0, 0]); _set(d.w_swing_pos, [-1, 0, 0]); _set(d.w_swing_vel, [-1, 0, 0]); d.w_td_vel = -1
    d.c_minheight, d.h_min = 1, 0.18
    d.reb_minheight = Reb(0.01, 0.01, 0.1)
    bufs = {}
    for name, w in (("xr", 12), ("ur", 12), ("foot_pos", 12), ("foot_vel", 12), ("body_pos", 3)):
        a = np.ascontiguousarray(refs[name], dtype=np.float64); assert a.shape == (horizon + 1, w), (name, a.shape)
        bufs[name] = a; setattr(d, name, a.ctypes.data_as(DP))
    rc = np.ascontiguousarray(refs["ref_contact"], dtype=np.int32); bufs["ref_contact"] = rc; d.ref_contact = rc.ctypes.data_as(IP)
    return {"desc": d, "bufs": bufs, "Xbar": bufs["xr"].copy(), "Ubar": np.zeros((horizon, 12))}


def mhpc_problem(wb_schedule=((1, 1, 1, 1), (0, 1, 1, 0)), wb_horizons=(25, 25), srb_schedule=((1, 0, 0, 1), (0, 1, 1, 0)),
                 srb_horizons=(5, 5), dt_wb=0.01, dt_srb=0.05, vx=0.5, swing_height=0.06, ubar_mode="gravity_comp"):
    """The MHPC horizon proper (MHPCProblem.cpp:373-521, mhpc_config.yaml): whole-body phases followed by a
    single-rigid-body tail at a coarser time step; the last WB phase resets into the SRB state through the
    impact map and the state projection (MHPCReset.cpp:4-52).  Synthetic trot references as in wb_trot_problem."""
    sched = list(wb_schedule) + list(srb_schedule)
    hs_ = list(wb_horizons) + list(srb_horizons)
    dts = [dt_wb] * len(wb_schedule) + [dt_srb] * len(srb_schedule)
    nph, nwb = len(sched), len(wb_schedule)
    t0 = np.concatenate([[0.0], np.cumsum([h * d for h, d in zip(hs_, dts)])])
    nominal_feet = wb_foot_positions(wb_nominal_state()[:18]); nominal_feet[:, 2] = 0.0

    def foothold(i, f):
        i = min(max(i, 0), nph - 1)
        return nominal_feet[f] + np.array([vx * 0.5 * (t0[i] + t0[i + 1]), 0, 0])

    phases = []
    for i in range(nph):
        h, dt = hs_[i], dts[i]
        nxt = sched[i + 1] if i + 1 < nph else sched[i]
        n = 36 if i < nwb else 12
        xr = np.zeros((h + 1, n)); fp = np.zeros((h + 1, 12)); fv = np.zeros((h + 1, 12)); rc = np.zeros((h + 1, 4), dtype=np.int32)
        grf = np.zeros((h + 1, 12)); bp = np.zeros((h + 1, 3))
        for k in range(h + 1):
            t = t0[i] + k * dt
            bp[k] = [vx * t, 0.0, Z_NOM]
            if i < nwb:
                xr[k, 0] = vx * t; xr[k, 2] = Z_NOM; xr[k, 6:18] = QJ_NOM; xr[k, 18] = vx
            else:
                xr[k, 0] = vx * t; xr[k, 2] = Z_NOM; xr[k, 6] = vx
            c = sched[i] if k < h else nxt
            rc[k] = c
            nc = max(1, int(np.sum(c)))
            for f in range(4):
                if sched[i][f]:
                    fp[k, 3 * f:3 * f + 3] = foothold(i, f)
                else:
                    p0, p1 = foothold(i - 1, f), foothold(i + 1, f)
                    if i == 0:
                        p0 = foothold(0, f) - np.array([vx * h * dt, 0, 0])
                    if i == nph - 1:
                        p1 = foothold(i, f) + np.array([vx * h * dt, 0, 0])
                    s = k / h
                    fp[k, 3 * f:3 * f + 3] = p0 + (p1 - p0) * s + np.array([0, 0, swing_height * np.sin(np.pi * s)])
                    fv[k, 3 * f:3 * f + 3] = (p1 - p0) / (h * dt) + np.array([0, 0, swing_height * np.pi / (h * dt) * np.cos(np.pi * s)])
                if c[f]:
                    grf[k, 3 * f + 2] = MASS_WB * 9.81 / nc
        if i < nwb:
            refs = dict(xr=xr, ur=np.zeros((h + 1, 12)), yr=grf, foot_pos=fp, foot_vel=fv, body_pos=bp, ref_contact=rc)
            phases.append(wb_phase(h, dt, t0[i], sched[i], nxt, refs, next_model=MODEL_WB if i + 1 < nwb else MODEL_SRB, ubar_mode=ubar_mode))
        else:
            # SRB knots take their contact set from the reference (SRBM.h:43-60): stance feet of THIS phase at every knot
            rc[:] = np.array(sched[i], dtype=np.int32)
            nc = max(1, int(np.sum(sched[i])))
            grf[:] = 0.0
            for f in range(4):
                if sched[i][f]:
                    grf[:, 3 * f + 2] = 8.912 * 9.81 / nc
            refs = dict(xr=xr, ur=grf, foot_pos=fp, foot_vel=fv, body_pos=bp, ref_contact=rc)
            ph = srb_phase(h, dt, t0[i], refs)
            if ubar_mode == "gravity_comp":
                ph["Ubar"] = grf[:h].copy()
            phases.append(ph)
    return phases


# ---- barrel roll (MHPC/MHPC-Trajopt/BarrelRoll/BarrelRollTO.cpp): values of setting/br_cost_weights.JSON,
#      br_constraint_params.info and load_desired_final_states (BarrelRollTO.cpp:262-341), restated as data
_BR_WEIGHTS = [
    dict(qB=[0, 5, 10, 2, 2, 2], vB=[1, .1, 1, 1, 1, 1], qJ=[.01] * 3, vJ=[.01] * 3, rw=.2, fqB=[0, 1, 10, 2, 2, 10], fvB=[1, .5, 5, 2, 2, 5], fqJ=[.05] * 3, fvJ=[.1] * 3),
    dict(qB=[0, 1, 10, 2, 2, 10], vB=[1] * 6, qJ=[.1] * 3, vJ=[.1] * 3, rw=.05, fqB=[0, 1, 10, 5, 5, 10], fvB=[1, 1, 5, 1, 1, 5], fqJ=[.1] * 3, fvJ=[.01] * 3),
    dict(qB=[0, 1, 5, 2, 2, 2], vB=[1] * 6, qJ=[1, .1, .1], vJ=[.1] * 3, rw=.5, fqB=[0, 1, 5, 5, 5, 5], fvB=[1, 1, 2, 1, 1, 1], fqJ=[.5, .1, .1], fvJ=[.01] * 3),
    dict(qB=[0, 1, 5, 2, 2, 2], vB=[1] * 6, qJ=[.1] * 3, vJ=[.1] * 3, rw=.1, fqB=[0, 1, 5, 5, 5, 10], fvB=[2, 2, .5, 1, 1, 1], fqJ=[.1] * 3, fvJ=[.01] * 3),
    dict(qB=[0, 1, 5, 2, 2, 2], vB=[1, 1, .5, 1, 1, 1], qJ=[1] * 3, vJ=[.1] * 3, rw=.1, fqB=[0, 0, 1, 5, 5, 10], fvB=[2, 2, .2, 1, 1, 1], fqJ=[1] * 3, fvJ=[.01] * 3),
    dict(qB=[0, 1, 5, 2, 2, 2], vB=[1] * 6, qJ=[.1] * 3, vJ=[.1] * 3, rw=.1, fqB=[0, 1, 5, 5, 5, 10], fvB=[2, 2, .5, 1, 1, 1], fqJ=[.1] * 3, fvJ=[.01] * 3),
]


def _br_state(pos, eul, qJ, v, euld):
    return np.concatenate([pos, eul, qJ, v, euld, np.zeros(12)])


def barrel_roll_states():
    """xinit and the six desired phase-final states of BarrelRollTO.cpp (:93-105, 262-341)."""
    pi = np.pi
    xinit = _br_state([0, 0, 0.2183], [0, 0, 0], np.tile([0, -1.0, 2.0], 4), [0, 0, 0], [0, 0, 0])
    q3 = np.array([0.3, -1.1, 2.2, -0.3, -1.1, 2.2, 0.3, -1.1, 2.2, -0.3, -1.1, 2.2])
    xf = [
        _br_state([0, -0.15, 0.26], [0, 0, pi / 6], np.tile([0, -1.2, 2.4], 4), [0, -1.0, 2.0], [0, 0, 3 * pi]),
        _br_state([0, -0.25, 0.33], [0, 0, 0.5 * pi], np.tile([pi / 6, -1.0, 2.0, -pi / 5, -0.5, 1.0], 2), [0, -1.2, 2.0], [0, 0, 3 * pi]),
        _br_state([0, -0.55, 0.22], [0, 0, 2 * pi], q3, [0, -1.5, -2.5], [0, 0, 3 * pi]),
        _br_state([0, -0.55, 0.25], [0, 0, 2 * pi], q3, [0, 0, 0], [0, 0, 0]),
        _br_state([0, -0.55, 0.25], [0, 0, 2 * pi], np.tile([0, -1.0, 2.0], 4), [0, 0, 0], [0, 0, 0]),
        _br_state([0, -0.55, 0.25], [0, 0, 2 * pi], np.tile([0, -1.0, 2.0], 4), [0, 0, 0], [0, 0, 0]),
    ]
    return xinit, xf


def barrel_roll_problem(switching_times=(0.0, 0.12, 0.33, 0.75, 0.90, 1.10, 1.25), dt=0.01,
                        contacts=((1, 1, 1, 1), (0, 1, 0, 1), (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0), (1, 1, 1, 1)), repeat=1):
    """The barrel-roll trajectory optimisation of BarrelRollTO.cpp: 6 hybrid phases (stance, right-side stance,
    flight, landing, flight, stance), tracking cost towards a constant per-phase target, torque / joint-speed /
    joint / height / GRF barriers and a four-foot touchdown constraint at the end of the flight phases.
    Returns (phases, xinit).  `repeat` > 1 chains the last four phases again ("running" variant, BASELINE config 3)."""
    xinit, xf = barrel_roll_states()
    sw = list(switching_times); cts = list(contacts); tgt = list(xf); wts = list(_BR_WEIGHTS)
    for _ in range(repeat - 1):   # flight -> landing pairs appended with the same durations
        for j in (4, 5):
            sw.append(sw[-1] + (switching_times[j + 1] - switching_times[j])); cts.append(contacts[j]); tgt.append(xf[j]); wts.append(_BR_WEIGHTS[j])
    return _br_build(sw, cts, tgt, wts, xinit, dt), xinit


def barrel_roll_running_problem(knots=(12, 21, 42, 15, 20, 15, 100, 125), dt=0.01):
    """BASELINE config 4 as SURVEY 8(d) specifies it: the barrel-roll schedule of BarrelRollTO.cpp:70-81 with a running lead-out,
    1111(12) 0101(21) 0000(42) 1111(15) 0000(20) 1111(15) 0101(100) 1010(125) = 8 hybrid phases / 350 knots; tracking targets and
    weights of the six shipped sets (BarrelRoll/setting/br_cost_weights.JSON, BarrelRollTO.cpp:277-338), the last set reused for the
    two running phases; constraints and initial guess as in barrel_roll_problem.  Returns (phases, xinit)."""
    xinit, xf = barrel_roll_states()
    cts = [(1, 1, 1, 1), (0, 1, 0, 1), (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0), (1, 1, 1, 1), (0, 1, 0, 1), (1, 0, 1, 0)]
    assert len(knots) == len(cts)
    sw = [0.0]
    for h in knots:
        sw.append(sw[-1] + h * dt)
    idx = [0, 1, 2, 3, 4, 5, 5, 5]
    return _br_build(sw, cts, [xf[i] for i in idx], [_BR_WEIGHTS[i] for i in idx], xinit, dt), xinit


def barrel_roll_ensemble_x0(batch, seed, xinit, first=0):
    """Ensemble of initial joint poses around the barrel roll's crouch (+-0.02 rad per joint); problem b draws from stream position b*12."""
    g = SplitMix64(seed); g.skip(first * 12)
    x = np.tile(xinit, (batch, 1))
    x[:, 6:18] += 0.04 * (np.array([g.next() for _ in range(batch * 12)]).reshape(batch, 12) - 0.5)
    return x


def _br_build(sw, cts, tgt, wts, xinit, dt):
    nph = len(cts)
    phases = []
    for i in range(nph):
        h = int(round((sw[i + 1] - sw[i]) / dt))
        nxt = cts[i + 1] if i + 1 < nph else cts[i]
        w = wts[i]
        xr = np.tile(tgt[i], (h + 1, 1))
        rc = np.tile(np.array(cts[i], dtype=np.int32), (h + 1, 1))
        refs = dict(xr=xr, ur=np.zeros((h + 1, 12)), yr=np.zeros((h + 1, 12)), foot_pos=np.zeros((h + 1, 12)), foot_vel=np.zeros((h + 1, 12)),
                    body_pos=xr[:, :3].copy(), ref_contact=rc)
        ph = wb_phase(h, dt, sw[i], cts[i], nxt, refs, ubar_mode="zero")
        d = ph["desc"]
        _set(d.q, list(w["qB"]) + list(w["qJ"]) * 4 + list(w["vB"]) + list(w["vJ"]) * 4)
        _set(d.r, [w["rw"]] * 12)
        _set(d.qf, list(w["fqB"]) + list(w["fqJ"]) * 4 + list(w["fvB"]) + list(w["fvJ"]) * 4)
        _set(d.w_foot_reg, [-1, 0, 0]); _set(d.w_swing_pos, [-1, 0, 0]); _set(d.w_swing_vel, [-1, 0, 0]); d.w_td_vel = -1.0   # tracking cost only
        d.c_jointspeed, d.jointspeed_lb, d.jointspeed_ub = 1, -20.0, 20.0
        d.h_min = 0.13
        d.reb_grf = Reb(0.02, 0.02, 0.1); d.reb_torque = Reb(0.01, 0.01, 0.1); d.reb_jointspeed = Reb(0.1, 0.1, 0.1)
        d.reb_joint = Reb(0.01, 0.01, 0.1); d.reb_minheight = Reb(0.01, 0.01, 0.1)
        d.al_td = Al(20.0, 0.0, 1e4)
        # initial guess: linear interpolation between consecutive targets with float time accumulation (BarrelRollTO.cpp:131-145)
        x_from = xinit if i == 0 else tgt[i - 1]
        t = np.float32(0.0); dur = np.float32(sw[i + 1] - sw[i])
        X = np.zeros((h + 1, 36))
        for k in range(h + 1):
            X[k] = x_from + (tgt[i] - x_from) * float(t / dur)
            t = np.float32(t + np.float32(dt))
        ph["Xbar"] = X
        phases.append(ph)
    return phases


def br_ddp_setting(**kw):
    """br_ddp_setting.info as loadHSDDPSetting reads it (update_regularization stays at the struct default 2: quirk xiii)."""
    from ._abi import mhpc_ddp_setting
    base = dict(alpha=0.5, gamma=0.1, update_penalty=5, update_relax=1, update_ReB=1, max_DDP_iter=10, max_AL_iter=30,
                cost_thresh=1e-2, tconstr_thresh=1e-3, pconstr_thresh=1e-3, dynamics_feas_thresh=1e-3, merit_scale=0.1, merit_offset=1,
                AL_active=1, ReB_active=1, MS=1)
    base.update(kw)
    return mhpc_ddp_setting(**base)


# ---- hybrid kinodynamic model (HKDMPC/HKD-TrajOpt): x = [eul, pos, omega, v, qdummy(12)], u = [GRF(12), qJdot(12)],
#      legs FR, FL, HR, HL
HKD_LEG_TO_WB = (1, 0, 3, 2)


def hkd_phase(horizon, dt, t_offset, contact, next_contact, refs):
    """One HKD phase as HKDProblem::create_problem_one_phase / add_tconstr_one_phase assemble it (HKDProblem.cpp:236-323):
    HKDTrackingCost weights (HKDCost.h:10-40), HKDFootPlaceReg (Qfoot = 100 on stance-foot x,y; HKDCost.h:55-76),
    GRF pyramid on u (mu 0.7) and the touchdown constraint, parameters of HKDMPC/settings/constraint_params.info."""
    d = PhaseDesc()
    d.model, d.horizon, d.dt, d.t_offset = MODEL_HKD, horizon, dt, t_offset
    _set(d.contact, contact); _set(d.next_contact, next_contact)
    d.next_model, d.shooting, d.BG_alpha = MODEL_HKD, 1, 0.0
    q = [1, 4, 4, 1, 1, 30, 1.0, 0.5, 0.2, 1, 1, 1] + [0.1 * (1 - contact[l]) for l in range(4) for _ in range(3)]
    scale = [1, 1, 2, 1, 1, 20, 1.0, 0.2, 0.1, 1, 1, 1] + [0.01] * 12
    _set(d.q, q); _set(d.qf, [20 * s * w for s, w in zip(scale, q)]); _set(d.r, [0.1] * 24)
    _set(d.w_foot_reg, [100.0, 100.0, 0.0]); _set(d.w_swing_pos, [-1, 0, 0]); _set(d.w_swing_vel, [-1, 0, 0]); d.w_td_vel = -1.0
    d.c_grf, d.mu = 1, 0.7
    d.reb_grf = Reb(0.1, 0.1, 0.5)
    d.c_touchdown, d.ground_height = 1, 0.0
    d.al_td = Al(20.0, 0.0, 1e4)
    bufs = {}
    for name, w in (("xr", 24), ("ur", 24), ("foot_pos", 12), ("foot_vel", 12), ("body_pos", 3)):
        a = np.ascontiguousarray(refs[name], dtype=np.float64); assert a.shape == (horizon + 1, w), (name, a.shape)
        bufs[name] = a; setattr(d, name, a.ctypes.data_as(DP))
    rc = np.ascontiguousarray(refs["ref_contact"], dtype=np.int32); bufs["ref_contact"] = rc; d.ref_contact = rc.ctypes.data_as(IP)
    return {"desc": d, "bufs": bufs, "Xbar": bufs["xr"].copy(), "Ubar": bufs["ur"][:horizon].copy()}


def hkd_trot_problem(schedule=((1, 1, 1, 1), (1, 0, 0, 1), (0, 1, 1, 0), (1, 0, 0, 1)), horizons=(10, 10, 10, 10), dt=0.01, vx=0.5,
                     last_next=(0, 1, 1, 0)):
    """BASELINE config 4 in shape: HKDMPC trot (diagonal pairs FR+HL / FL+HR), synthetic references built the way
    HKDSinglePhaseReference::get_reference_at_t lays them out (HKDReference.cpp:23-61): stance legs track the planned
    foothold, swing legs the nominal joint angles; GRF reference = weight split over the stance feet."""
    nph = len(schedule)
    t0 = np.concatenate([[0.0], np.cumsum([h * dt for h in horizons])])
    feet_wb = wb_foot_positions(wb_nominal_state()[:18]); feet_wb[:, 2] = 0.0
    feet = feet_wb[list(HKD_LEG_TO_WB)]          # nominal footholds under the hips, HKD leg order

    def foothold(i, f):
        i = min(max(i, 0), nph - 1)
        return feet[f] + np.array([vx * 0.5 * (t0[i] + t0[i + 1]), 0, 0])

    phases = []
    for i in range(nph):
        h = horizons[i]; c = schedule[i]
        nxt = schedule[i + 1] if i + 1 < nph else last_next
        xr = np.zeros((h + 1, 24)); ur = np.zeros((h + 1, 24)); fp = np.zeros((h + 1, 12)); bp = np.zeros((h + 1, 3))
        rc = np.tile(np.array(c, dtype=np.int32), (h + 1, 1))
        nc = max(1, int(np.sum(c)))
        for k in range(h + 1):
            t = t0[i] + k * dt
            bp[k] = [vx * t, 0.0, Z_NOM]
            xr[k, 3:6] = bp[k]; xr[k, 9] = vx
            for f in range(4):
                fp[k, 3 * f:3 * f + 3] = foothold(i if c[f] else i + 1, f)
                xr[k, 12 + 3 * f:15 + 3 * f] = fp[k, 3 * f:3 * f + 3] if c[f] else QJ_NOM[:3]
                if c[f]:
                    ur[k, 3 * f + 2] = 8.912 * 9.81 / nc
        refs = dict(xr=xr, ur=ur, foot_pos=fp, foot_vel=np.zeros((h + 1, 12)), body_pos=bp, ref_contact=rc)
        phases.append(hkd_phase(h, dt, t0[i], c, nxt, refs))
    return phases


def hkd_bound_problem(n_knots=200, dt=0.01, vx=0.5):
    """BASELINE config 5's workload: HKD 24/24/0 phases in the contact pattern of the shipped bound gait
    (Reference/Data/bound: 1111(6), then 1100(10) 0000(10) 0011(10) 0000(10) repeating; HKD leg order FR, FL, HR, HL), cut at n_knots."""
    sched, hor = [(1, 1, 1, 1)], [6]
    cyc = [((1, 1, 0, 0), 10), ((0, 0, 0, 0), 10), ((0, 0, 1, 1), 10), ((0, 0, 0, 0), 10)]
    i = 0
    while sum(hor) < n_knots:
        c, h = cyc[i % 4]; h = min(h, n_knots - sum(hor))
        sched.append(c); hor.append(h); i += 1
    return hkd_trot_problem(schedule=tuple(sched), horizons=tuple(hor), dt=dt, vx=vx, last_next=cyc[i % 4][0])


def hkd_ensemble_x0(batch, seed, phases, first=0):
    """Perturbed copies of the first reference state (body pose / twist noise; qdummy kept consistent with the contact set); problem b
    draws its 12 numbers from stream position b*12."""
    g = SplitMix64(seed); g.skip(first * 12)
    x0 = np.tile(phases[0]["bufs"]["xr"][0], (batch, 1))
    amp = np.array([0.06] * 6 + [0.2] * 6)
    x0[:, :12] += (np.array([g.next() for _ in range(batch * 12)]).reshape(batch, 12) - 0.5) * amp
    return x0


def hkd_ddp_setting(**kw):
    """HKDMPC/settings/ddp_setting.info as loadHSDDPSetting reads it (update_regularization stays 2: quirk xiii)."""
    from ._abi import mhpc_ddp_setting
    base = dict(alpha=0.1, gamma=0.01, update_penalty=5, update_relax=1, update_ReB=1, max_DDP_iter=10, max_AL_iter=5,
                cost_thresh=1e-3, tconstr_thresh=1e-3, pconstr_thresh=1e-3, dynamics_feas_thresh=1e-3, merit_scale=0.2, merit_offset=1e2,
                AL_active=1, ReB_active=1, MS=1)
    base.update(kw)
    return mhpc_ddp_setting(**base)
