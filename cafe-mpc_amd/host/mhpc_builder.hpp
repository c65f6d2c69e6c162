// Host-side problem builder in C++ (header-only): the counterpart of MHPCProblem<T>::initialization / update
// (MHPC/MHPC-Trajopt/MHPCProblem.cpp:14-397, 403-601) and of the gait loader QuadReference (Reference/QuadReference.cpp:5-408),
// emitting the POD phase descriptors of include/hsddp.h instead of SinglePhase objects with closures.  Same logic as the Python
// mirror cafe-mpc_amd/builder.py (tests/test_builder.py compares the two bit for bit); everything the reference does in `float`
// (time accumulation, nearest-sample lookup, std::stof parsing: SURVEY quirk vii) is done in float here as well.
#pragma once
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include "hsddp.h"

namespace hsddp {

// ------------------------------------------------------------------------------------------------ settings files
// boost::property_tree INFO file -> section -> key -> string
inline std::map<std::string, std::map<std::string, std::string>> load_info(const std::string& path) {
    std::map<std::string, std::map<std::string, std::string>> out; std::ifstream f(path); std::string line, cur;
    while (std::getline(f, line)) {
        const size_t sc = line.find(';'); if (sc != std::string::npos) line = line.substr(0, sc);
        std::istringstream ls(line); std::vector<std::string> tok; std::string w; while (ls >> w) tok.push_back(w);
        if (tok.empty()) continue;
        if (tok[0] == "{") continue;
        if (tok[0] == "}") { cur.clear(); continue; }
        if (tok.size() == 1) { cur = tok[0]; out[cur]; } else if (!cur.empty()) out[cur][tok[0]] = tok[1];
    }
    return out;
}
// numbers of   "key": [a, b, c]   or   "key": a   inside the object named `section` of a JSON text (MHPCCostUtil.h reads exactly these)
inline std::vector<double> json_numbers(const std::string& text, const std::string& section, const std::string& key) {
    size_t p = text.find("\"" + section + "\""); if (p == std::string::npos) return {};
    const size_t end = text.find('}', p);
    p = text.find("\"" + key + "\"", p); if (p == std::string::npos || p > end) return {};
    p = text.find(':', p) + 1;
    while (p < text.size() && (text[p] == ' ' || text[p] == '\t')) p++;
    std::vector<double> v;
    if (text[p] == '[') { const size_t q = text.find(']', p); std::string body = text.substr(p + 1, q - p - 1); for (auto& c : body) if (c == ',') c = ' '; std::istringstream ls(body); double x; while (ls >> x) v.push_back(x); }
    else v.push_back(std::strtod(text.c_str() + p, nullptr));
    return v;
}

struct MhpcConfig { double plan_dur_wb = 0, plan_dur_srb = 0, dt_wb = 0, dt_srb = 0; float dt_mpc = 0, BG_alpha = 0; std::string referenceFile, costFile, constraintParamFile; };
inline MhpcConfig load_mhpc_config(const std::string& path) {     // loadMHPCConfig (MHPCProblem.h:66-84)
    auto c = load_info(path)["config"]; MhpcConfig m;
    m.plan_dur_wb = std::stod(c["plan_dur_wb"]); m.plan_dur_srb = std::stod(c["plan_dur_srb"]); m.dt_mpc = (float)std::stod(c["dt_mpc"]);
    m.dt_wb = std::stod(c["dt_wb"]); m.dt_srb = std::stod(c["dt_srb"]); m.BG_alpha = (float)std::stod(c["BG_alpha"]);
    m.referenceFile = c["referenceFile"]; m.costFile = c["costFile"]; m.constraintParamFile = c["constraintParamFile"];
    return m;
}
struct ConstraintParams { hsddp_reb_t grf, torque, joint, minheight; hsddp_al_t td; };
inline ConstraintParams load_constraint_params(const std::string& path) {
    auto p = load_info(path);
    auto reb = [&](const std::string& n) { auto& s = p[n + "_ReB"]; return hsddp_reb_t{std::stod(s["delta"]), std::stod(s["delta_min"]), std::stod(s["eps"])}; };
    auto& t = p["TD_AL"];
    return ConstraintParams{reb("GRF"), reb("Torque"), reb("Joint"), reb("MinHeight"), hsddp_al_t{std::stod(t["sigma"]), std::stod(t["lambda"]), std::stod(t["sigma_max"])}};
}
struct CostWeights { std::vector<double> wb_q, wb_r, wb_qf, srb_q, srb_r, srb_qf, foot_reg, swing_pos, swing_vel; };
inline CostWeights load_cost_weights(const std::string& path) {    // loadCostWeights (MHPCCostUtil.h:21-140)
    std::ifstream f(path); std::stringstream ss; ss << f.rdbuf(); const std::string t = ss.str();
    auto cat = [](std::vector<double> a, const std::vector<double>& b) { a.insert(a.end(), b.begin(), b.end()); return a; };
    auto rep4 = [&](const std::vector<double>& a) { std::vector<double> o; for (int i = 0; i < 4; i++) o.insert(o.end(), a.begin(), a.end()); return o; };
    CostWeights w; const std::string W = "WB_Tracking_Cost", S = "SRB_Tracking_Cost";
    w.wb_q = cat(cat(cat(json_numbers(t, W, "qw_qB"), rep4(json_numbers(t, W, "qw_qJ"))), json_numbers(t, W, "qw_vB")), rep4(json_numbers(t, W, "qw_vJ")));
    w.wb_qf = cat(cat(cat(json_numbers(t, W, "qfw_qB"), rep4(json_numbers(t, W, "qfw_qJ"))), json_numbers(t, W, "qfw_vB")), rep4(json_numbers(t, W, "qfw_vJ")));
    w.wb_r.assign(12, json_numbers(t, W, "rw")[0]);
    w.srb_q = cat(json_numbers(t, S, "qw_qB"), json_numbers(t, S, "qw_vB")); w.srb_qf = cat(json_numbers(t, S, "qfw_qB"), json_numbers(t, S, "qfw_vB"));
    w.srb_r.assign(12, json_numbers(t, S, "rw")[0]);
    w.foot_reg = json_numbers(t, "WB_FootPlace_Reg", "qw_per_foot"); w.swing_pos = json_numbers(t, "Swing_Pos_Tracking", "qw_per_foot");
    w.swing_vel = json_numbers(t, "Swing_Vel_Tracking", "qw_per_foot");
    return w;
}

// ------------------------------------------------------------------------------------------------ gait reference
struct QuadSample { double body[12], qJ[12], qJd[12], foot_pos[12], foot_vel[12], grf[12], torque[12], status_dur[4]; int contact[4]; };
class QuadReference {       // Reference/QuadReference.{h,cpp}
public:
    std::vector<QuadSample> tp; float dt = 0; int k_cur = 0, sz = 0;
    bool load(const std::string& path, bool reorder) {
        std::ifstream f(path); if (!f.is_open()) return false;
        std::string line; QuadSample cur; std::memset(&cur, 0, sizeof(cur));
        auto fill = [](const std::string& l, double* dst, int n) { std::istringstream ls(l); std::string w; int i = 0; while (i < n && ls >> w) dst[i++] = (double)std::stof(w); };
        while (std::getline(f, line)) {
            if (line == "dt") { std::getline(f, line); dt = std::stof(line); continue; }
            static const char* keys[] = {"body_state", "jnt_angle", "jnt_vel", "foot_placements", "foot_velocities", "foot_height", "grf", "torque", "contact", "status_dur"};
            int key = -1; for (int i = 0; i < 10; i++) if (line.find(keys[i]) != std::string::npos) { key = i; break; }   // same test order as load_top_level_data
            if (key < 0) continue;
            std::string vals; std::getline(f, vals);
            switch (key) {
                case 0: std::memset(&cur, 0, sizeof(cur)); fill(vals, cur.body, 12); break;
                case 1: fill(vals, cur.qJ, 12); break;
                case 2: fill(vals, cur.qJd, 12); break;
                case 3: fill(vals, cur.foot_pos, 12); break;
                case 4: fill(vals, cur.foot_vel, 12); break;
                case 5: break;
                case 6: fill(vals, cur.grf, 12); break;
                case 7: fill(vals, cur.torque, 12); break;
                case 8: { std::istringstream ls(vals); std::string w; int i = 0; while (i < 4 && ls >> w) cur.contact[i++] = std::stoi(w); } break;
                case 9: fill(vals, cur.status_dur, 4); tp.push_back(cur); break;
            }
        }
        for (auto& s : tp) {      // reorder_body_states: [eul, pos, omega, vWorld] -> [pos, eul, vWorld, omega]
            double b[12]; std::memcpy(b, s.body, sizeof(b));
            for (int i = 0; i < 3; i++) { s.body[i] = b[3 + i]; s.body[3 + i] = b[i]; s.body[6 + i] = b[9 + i]; s.body[9 + i] = b[6 + i]; }
            if (reorder) {        // reorder_leg_dependent_states: swap left / right legs, zero the joint velocities
                auto sw = [](double* a) { double t[12]; std::memcpy(t, a, sizeof(t)); for (int i = 0; i < 3; i++) { a[i] = t[3 + i]; a[3 + i] = t[i]; a[6 + i] = t[9 + i]; a[9 + i] = t[6 + i]; } };
                sw(s.qJ); sw(s.foot_pos); sw(s.foot_vel); sw(s.grf); sw(s.torque); std::memset(s.qJd, 0, sizeof(s.qJd));
                std::swap(s.contact[0], s.contact[1]); std::swap(s.contact[2], s.contact[3]); std::swap(s.status_dur[0], s.status_dur[1]); std::swap(s.status_dur[2], s.status_dur[3]);
            }
        }
        return !tp.empty();
    }
    void initialize(float plan_horizon) { k_cur = 0; sz = (int)std::round(plan_horizon / dt) + 1; }
    int step(float dt_sim) {      // returns the number of samples advanced
        int adv = 0;
        for (int i = 1; (float)i * dt < dt_sim || std::fabs((float)i * dt - dt_sim) <= 1e-6f; i++) { k_cur++; adv++; }
        return adv;
    }
    int index(float t) const {    // get_a_reference_ptr_at_t
        int k = (int)std::floor(t / dt);
        if (t - k * dt > 0.5 * dt) k++;
        return k >= sz ? sz - 1 : k;
    }
    const QuadSample& at(float t) const { return tp[k_cur + index(t)]; }
};

inline bool approx_eq(double a, double b) { return (float)std::fabs(a - b) <= 1e-6f; }      // HSDDP_Utils.h:46-56
inline bool approx_leq(double a, double b) { return a < b || approx_eq(a, b); }

// ------------------------------------------------------------------------------------------------ MHPC problem
struct PhaseBuffers { std::vector<double> xr, ur, yr, foot_pos, foot_vel, body_pos, Xbar, Ubar; std::vector<int> ref_contact; };
struct SlotMove { int uid, popped, pushed, old_h; };

class MhpcProblemData {
public:
    struct Row { float start, end; int h; std::array<int, 4> contact; std::array<double, 4> dur; bool reach_end, has_td; int shooting, uid; };
    std::vector<Row> wb; int srb_h = 0, srb_steps = 0; float srb_start = 0, ref_start = 0;
    MhpcProblemData(QuadReference& r, const MhpcConfig& c, const CostWeights& w, const ConstraintParams& p) : ref(r), cfg(c), costs(w), cpar(p) {
        ref.initialize((float)(cfg.plan_dur_wb + cfg.plan_dur_srb));
        if (cfg.plan_dur_wb > 1e-5) {      // MHPCProblem.cpp:69-108
            float t = 0, start = 0; auto c_prev = contact_at(t); auto d_prev = dur_at(t);
            while (approx_leq(t, cfg.plan_dur_wb)) {
                auto c_cur = contact_at(t);
                if (c_cur != c_prev || approx_eq(t, cfg.plan_dur_wb)) {
                    push(start, t, (int)std::round((double)(t - start) / cfg.dt_wb), c_prev, d_prev, 1);
                    c_prev = c_cur; d_prev = dur_at(t); start = t;
                }
                t = (float)((double)t + cfg.dt_wb);
            }
        }
        srb_h = cfg.plan_dur_srb > 1e-5 ? (int)std::round(cfg.plan_dur_srb / cfg.dt_srb) : 0;
        srb_start = (float)cfg.plan_dur_wb;
        for (size_t i = 0; i < wb.size(); i++) wb[i].has_td = touchdown(i);
    }
    // MHPCProblem::update (MHPCProblem.cpp:252-377)
    std::vector<SlotMove> update() {
        std::map<int, int> old_h; for (auto& r : wb) old_h[r.uid] = r.h;
        const int nsteps = (int)std::round((double)cfg.dt_mpc / cfg.dt_wb);
        const int adv = ref.step(cfg.dt_mpc); ref_start = (float)((double)ref_start + adv * (double)ref.dt);
        std::map<int, int> popped, pushed;
        if (cfg.plan_dur_wb > 0) {
            for (int j = 0; j < nsteps; j++) {
                const float first = (float)((double)wb.front().start + cfg.dt_wb);
                if (approx_eq(wb.front().end, first)) wb.erase(wb.begin());
                else { popped[wb.front().uid]++; wb.front().h--; wb.front().start = first; }
            }
            for (int j = 0; j < nsteps; j++) {
                const float new_end = (float)((double)wb.back().end + cfg.dt_wb), t_rel = new_end - ref_start;
                auto nc = contact_at(t_rel); const bool change = nc != wb.back().contact;
                if (change && wb.back().reach_end) push(wb.back().end, new_end, 1, nc, dur_at(t_rel), 0);
                else {
                    wb.back().end = new_end; wb.back().h++;
                    if (change) { wb.back().reach_end = true; wb.back().has_td = touchdown(wb.size() - 1); }
                    pushed[wb.back().uid]++;
                }
            }
            for (size_t i = 0; i < wb.size(); i++) if (i + 1 < wb.size() || wb[i].h > nsteps) wb[i].shooting = 1;
        }
        if (cfg.plan_dur_srb > 0) { srb_steps = (int)std::floor((double)cfg.dt_mpc / cfg.dt_srb + 1e-6); srb_start = (float)((double)ref_start + cfg.plan_dur_wb); }
        std::vector<SlotMove> out;
        for (auto& r : wb) out.push_back({r.uid, popped[r.uid], pushed[r.uid], old_h.count(r.uid) ? old_h[r.uid] : -1});
        return out;
    }
    // descriptors of the current window; `bufs` owns the arrays the descriptors point to
    std::vector<hsddp_phase_desc_t> describe(std::vector<PhaseBuffers>& bufs) const {
        std::vector<hsddp_phase_desc_t> out; const size_t n_wb = wb.size(); bufs.assign(n_wb + (srb_h > 0 ? 1 : 0), PhaseBuffers());
        for (size_t i = 0; i < n_wb; i++) {
            const Row& r = wb[i]; const int h = r.h; PhaseBuffers& B = bufs[i];
            const auto nxt = next_contact(i); const double t_off = (double)(float)(r.start - wb[0].start);
            B.xr.resize((h + 1) * 36); B.ur.resize((h + 1) * 12); B.yr.resize((h + 1) * 12); B.foot_pos.resize((h + 1) * 12); B.foot_vel.resize((h + 1) * 12);
            B.body_pos.resize((h + 1) * 3); B.ref_contact.resize((h + 1) * 4); B.Xbar.resize((h + 1) * 36); B.Ubar.assign(h * 12, 0.0);
            for (int k = 0; k <= h; k++) {
                const QuadSample& a = ref.at((float)(t_off + k * cfg.dt_wb)); wb_state(a, &B.xr[k * 36]);
                std::memcpy(&B.ur[k * 12], a.torque, 96); std::memcpy(&B.yr[k * 12], a.grf, 96); std::memcpy(&B.foot_pos[k * 12], a.foot_pos, 96);
                std::memcpy(&B.foot_vel[k * 12], a.foot_vel, 96); std::memcpy(&B.body_pos[k * 3], a.body, 24); std::memcpy(&B.ref_contact[k * 4], a.contact, 16);
                wb_state(ref.at((float)((double)(float)(r.start - ref_start) + k * cfg.dt_wb)), &B.Xbar[k * 36]);
            }
            hsddp_phase_desc_t d; std::memset(&d, 0, sizeof(d));
            d.model = HSDDP_MODEL_WB; d.horizon = h; d.dt = cfg.dt_wb; d.t_offset = t_off;
            for (int l = 0; l < 4; l++) { d.contact[l] = r.contact[l]; d.next_contact[l] = nxt[l]; }
            d.next_model = (i + 1 == n_wb && srb_h > 0) ? HSDDP_MODEL_SRB : HSDDP_MODEL_WB; d.shooting = r.shooting; d.BG_alpha = cfg.BG_alpha;
            for (int j = 0; j < 36; j++) { d.q[j] = costs.wb_q[j]; d.qf[j] = costs.wb_qf[j]; } for (int j = 0; j < 12; j++) d.r[j] = costs.wb_r[j];
            for (int j = 0; j < 3; j++) { d.w_foot_reg[j] = costs.foot_reg[j]; d.w_swing_pos[j] = costs.swing_pos[j]; d.w_swing_vel[j] = costs.swing_vel[j]; }
            d.w_td_vel = r.has_td ? 1.0 : -1.0;                                   // TDVelocityPenalty::qFoot (MHPCCost.h:222)
            d.c_torque = d.c_joint = d.c_minheight = d.c_grf = 1; d.c_touchdown = r.has_td ? 1 : 0;
            d.torque_limit = 17.0; const double lb[3] = {-1.3, -5.0, -M_PI}, ub[3] = {1.3, 5.0, M_PI};       // MHPCConstraint.cpp:172-173
            for (int j = 0; j < 3; j++) { d.joint_lb[j] = lb[j]; d.joint_ub[j] = ub[j]; }
            d.h_min = 0.20; d.mu = 0.6; d.ground_height = 0.0;
            d.reb_torque = cpar.torque; d.reb_joint = cpar.joint; d.reb_minheight = cpar.minheight; d.reb_grf = cpar.grf; d.al_td = cpar.td;
            d.xr = B.xr.data(); d.ur = B.ur.data(); d.yr = B.yr.data(); d.foot_pos = B.foot_pos.data(); d.foot_vel = B.foot_vel.data(); d.body_pos = B.body_pos.data();
            d.ref_contact = B.ref_contact.data();
            out.push_back(d);
        }
        if (srb_h > 0) {          // MHPCProblem.cpp:216-247, 488-521
            const int h = srb_h; PhaseBuffers& B = bufs[n_wb]; const double t_off = (double)(float)(srb_start - ref_start);
            B.xr.resize((h + 1) * 12); B.ur.resize((h + 1) * 12); B.foot_pos.resize((h + 1) * 12); B.foot_vel.assign((h + 1) * 12, 0.0); B.body_pos.resize((h + 1) * 3);
            B.ref_contact.resize((h + 1) * 4); B.Ubar.assign(h * 12, 0.0);
            for (int k = 0; k <= h; k++) {
                const QuadSample& a = ref.at((float)(t_off + k * cfg.dt_srb));
                std::memcpy(&B.xr[k * 12], a.body, 96); std::memcpy(&B.ur[k * 12], a.grf, 96); std::memcpy(&B.foot_pos[k * 12], a.foot_pos, 96);
                std::memcpy(&B.body_pos[k * 3], a.body, 24); std::memcpy(&B.ref_contact[k * 4], a.contact, 16);
            }
            B.Xbar = B.xr;
            hsddp_phase_desc_t d; std::memset(&d, 0, sizeof(d));
            d.model = HSDDP_MODEL_SRB; d.horizon = h; d.dt = cfg.dt_srb; d.t_offset = t_off; d.next_model = -1; d.shooting = 1;
            for (int j = 0; j < 12; j++) { d.q[j] = costs.srb_q[j]; d.qf[j] = costs.srb_qf[j]; d.r[j] = costs.srb_r[j]; }
            d.w_foot_reg[0] = d.w_swing_pos[0] = d.w_swing_vel[0] = -1.0; d.w_td_vel = -1.0;
            d.c_minheight = 1; d.h_min = 0.18; d.reb_minheight = cpar.minheight;
            d.xr = B.xr.data(); d.ur = B.ur.data(); d.foot_pos = B.foot_pos.data(); d.foot_vel = B.foot_vel.data(); d.body_pos = B.body_pos.data(); d.ref_contact = B.ref_contact.data();
            out.push_back(d);
        }
        return out;
    }

private:
    QuadReference& ref; MhpcConfig cfg; CostWeights costs; ConstraintParams cpar; int next_uid = 0;
    std::array<int, 4> contact_at(float t) const { const QuadSample& s = ref.at(t); return {s.contact[0], s.contact[1], s.contact[2], s.contact[3]}; }
    std::array<double, 4> dur_at(float t) const { const QuadSample& s = ref.at(t); return {s.status_dur[0], s.status_dur[1], s.status_dur[2], s.status_dur[3]}; }
    void push(float start, float end, int h, const std::array<int, 4>& c, const std::array<double, 4>& d, int shooting) { wb.push_back(Row{start, end, h, c, d, false, false, shooting, next_uid++}); }
    std::array<int, 4> next_contact(size_t i) const { return i + 1 < wb.size() ? wb[i + 1].contact : contact_at((float)(cfg.plan_dur_wb + (double)cfg.dt_mpc)); }
    bool touchdown(size_t i) const { const auto n = next_contact(i); for (int l = 0; l < 4; l++) if (wb[i].contact[l] == 0 && n[l] == 1) return true; return false; }
    static void wb_state(const QuadSample& a, double* x) {       // WBReference::get_reference_at_t (MHPCReference.cpp:24-39)
        std::memcpy(x, a.body, 48); std::memcpy(x + 6, a.qJ, 96); std::memcpy(x + 18, a.body + 6, 48); std::memcpy(x + 24, a.qJd, 96);
    }
};

// ------------------------------------------------------------------------------------------------ HKD-MPC problem
struct HkdConstraintParams { hsddp_reb_t grf, swing; hsddp_al_t td; };
inline HkdConstraintParams load_hkd_constraint_params(const std::string& path) {      // loadConstrintParameters (HKDMPC/HKD-TrajOpt/HKDProblem.cpp:66)
    auto p = load_info(path); HkdConstraintParams c;
    auto reb = [&](const std::string& n) { auto& m = p[n]; return hsddp_reb_t{std::stod(m["delta"]), std::stod(m["delta_min"]), std::stod(m["eps"])}; };
    c.grf = reb("GRF_ReB"); c.swing = reb("Swing_ReB");
    c.td = hsddp_al_t{std::stod(p["TD_AL"]["sigma"]), std::stod(p["TD_AL"]["lambda"]), std::stod(p["TD_AL"]["sigma_max"])};
    return c;
}

// Phase table of HKDProblemData + the rules that evolve it: HKDProblem::initialization (HKDProblem.cpp:14-111) with the constants of
// HKDMPCSolver::initialize (HKDMPC/HKDMPC.cpp:26-29) and the receding-horizon HKDProblem::update (:117-222).  The reference must have been loaded
// with reorder = true (HKDMPC.h:32).  Same contract as MhpcProblemData: update() returns where every surviving phase's knots come from, the
// caller hands that to hsddp::MultiPhaseDDP::reconfigure and zeroes the window's first control (set_control_knot(0, 0), HKDProblem.cpp:220).
class HkdProblemData {
public:
    struct Row { float start, end; int h; std::array<int, 4> contact; bool reach_end, has_td; int shooting, uid; };
    std::vector<Row> ph; float ref_start = 0; int dup_td = 0;
    HkdProblemData(QuadReference& r, const HkdConstraintParams& p, float plan_duration = 0.6f, float dt_sim = 0.01f, int nsteps_between_mpc = 2)
        : ref(r), cpar(p), plan(plan_duration), dt(dt_sim), nsteps(nsteps_between_mpc), dt_mpc(dt_sim * (float)nsteps_between_mpc) {
        ref.initialize(plan);
        float t = 0, start = 0; auto c_prev = contact_at(t);
        while (approx_leq(t, plan)) {      // HKDProblem.cpp:34-63
            auto c_cur = contact_at(t);
            if (c_cur != c_prev || t > plan || approx_eq(t, plan)) { push(start, t, (int)std::round((float)(t - start) / dt), c_prev, 1); c_prev = c_cur; start = t; }
            t = t + dt;
        }
        for (size_t i = 0; i < ph.size(); i++) ph[i].has_td = touchdown(i);
    }
    std::vector<SlotMove> update() {       // HKDProblem::update (:117-222)
        std::map<int, int> old_h, popped, pushed; for (auto& r : ph) old_h[r.uid] = r.h;
        for (int j = 0; j < nsteps; j++) {
            const int adv = ref.step(dt); ref_start = (float)((double)ref_start + adv * (double)ref.dt);
            const float new_start = ref_start, new_end = new_start + plan;
            ph.front().start = ph.front().start + dt;
            if (approx_leq(ph.front().end, new_start)) ph.erase(ph.begin());
            else { popped[ph.front().uid]++; ph.front().h--; ph.front().start = new_start; }
            auto nc = contact_at(new_end - new_start); const bool change = nc != ph.back().contact;
            if (change && ph.back().reach_end) { const float ns = ph.back().end; push(ns, new_end, (int)std::round((float)(new_end - ns) / dt), nc, 0); }
            else { ph.back().end = new_end; ph.back().h++; if (change) ph.back().reach_end = true; pushed[ph.back().uid]++; }
            if (ph.back().reach_end) { const bool td = touchdown(ph.size() - 1); if (td && ph.back().has_td) dup_td++; ph.back().has_td = ph.back().has_td || td; }
        }
        for (size_t i = 0; i < ph.size(); i++) if (i + 1 < ph.size() || ph[i].h > 2) ph[i].shooting = 1;
        std::vector<SlotMove> out;
        for (auto& r : ph) out.push_back({r.uid, popped[r.uid], pushed[r.uid], old_h.count(r.uid) ? old_h[r.uid] : -1});
        return out;
    }
    std::vector<hsddp_phase_desc_t> describe(std::vector<PhaseBuffers>& bufs) const {
        std::vector<hsddp_phase_desc_t> out; bufs.assign(ph.size(), PhaseBuffers());
        for (size_t i = 0; i < ph.size(); i++) {
            const Row& r = ph[i]; const int h = r.h; PhaseBuffers& B = bufs[i];
            const auto nxt = next_contact(i); const double t_off = (double)(float)(r.start - ph[0].start);
            B.xr.resize((h + 1) * 24); B.ur.resize((h + 1) * 24); B.foot_pos.resize((h + 1) * 12); B.foot_vel.assign((h + 1) * 12, 0.0); B.body_pos.resize((h + 1) * 3);
            B.ref_contact.resize((h + 1) * 4); B.Xbar.resize((h + 1) * 24); B.Ubar.assign(h * 24, 0.0);
            for (int k = 0; k <= h; k++) {
                const QuadSample& a = ref.at((float)(t_off + k * (double)dt)); hkd_state(a, &B.xr[k * 24]);
                std::memcpy(&B.ur[k * 24], a.grf, 96); std::memcpy(&B.ur[k * 24 + 12], a.qJd, 96); std::memcpy(&B.foot_pos[k * 12], a.foot_pos, 96);
                std::memcpy(&B.body_pos[k * 3], a.body, 24); std::memcpy(&B.ref_contact[k * 4], a.contact, 16);
                hkd_state(ref.at((float)((double)(float)(r.start - ref_start) + k * (double)dt)), &B.Xbar[k * 24]);
            }
            hsddp_phase_desc_t d; std::memset(&d, 0, sizeof(d));
            d.model = HSDDP_MODEL_HKD; d.horizon = h; d.dt = (double)dt; d.t_offset = t_off; d.next_model = HSDDP_MODEL_HKD; d.shooting = r.shooting; d.BG_alpha = 0.0;
            for (int l = 0; l < 4; l++) { d.contact[l] = r.contact[l]; d.next_contact[l] = nxt[l]; }
            // HKDTrackingCost weights (HKDCost.h:10-40), terminal weights = 20 x scale x running ones
            const double q0[12] = {1, 4, 4, 1, 1, 30, 1.0, 0.5, 0.2, 1, 1, 1}, sc[12] = {1, 1, 2, 1, 1, 20, 1.0, 0.2, 0.1, 1, 1, 1};
            for (int j = 0; j < 12; j++) { d.q[j] = q0[j]; d.qf[j] = 20 * sc[j] * q0[j]; }
            for (int l = 0; l < 4; l++) for (int j = 0; j < 3; j++) { const double w = 0.1 * (1 - r.contact[l]); d.q[12 + 3 * l + j] = w; d.qf[12 + 3 * l + j] = 20 * 0.01 * w; }
            for (int j = 0; j < 24; j++) d.r[j] = 0.1;
            d.w_foot_reg[0] = 100.0; d.w_foot_reg[1] = 100.0; d.w_foot_reg[2] = 0.0; d.w_swing_pos[0] = -1; d.w_swing_vel[0] = -1; d.w_td_vel = -1.0;      // HKDFootPlaceReg (HKDCost.h:55-76)
            d.c_grf = 1; d.mu = 0.7; d.reb_grf = cpar.grf; d.c_touchdown = r.has_td ? 1 : 0; d.ground_height = 0.0; d.al_td = cpar.td;
            d.xr = B.xr.data(); d.ur = B.ur.data(); d.foot_pos = B.foot_pos.data(); d.foot_vel = B.foot_vel.data(); d.body_pos = B.body_pos.data(); d.ref_contact = B.ref_contact.data();
            out.push_back(d);
        }
        return out;
    }

private:
    QuadReference& ref; HkdConstraintParams cpar; float plan, dt; int nsteps; float dt_mpc; int next_uid = 0;
    std::array<int, 4> contact_at(float t) const { const QuadSample& s = ref.at(t); return {s.contact[0], s.contact[1], s.contact[2], s.contact[3]}; }
    void push(float start, float end, int h, const std::array<int, 4>& c, int shooting) { ph.push_back(Row{start, end, h, c, false, false, shooting, next_uid++}); }
    std::array<int, 4> next_contact(size_t i) const { return i + 1 < ph.size() ? ph[i + 1].contact : contact_at(plan + dt_mpc); }
    bool touchdown(size_t i) const { const auto n = next_contact(i); for (int l = 0; l < 4; l++) if (ph[i].contact[l] == 0 && n[l] == 1) return true; return false; }
    static void hkd_state(const QuadSample& a, double* x) {      // HKDSinglePhaseReference::get_reference_at_t (HKDReference.cpp:23-61): [eul, pos, omega, v, qdummy]
        for (int i = 0; i < 3; i++) { x[i] = a.body[3 + i]; x[3 + i] = a.body[i]; x[6 + i] = a.body[9 + i]; x[9 + i] = a.body[6 + i]; }
        for (int l = 0; l < 4; l++) for (int j = 0; j < 3; j++) x[12 + 3 * l + j] = a.contact[l] > 0 ? a.foot_pos[3 * l + j] : a.qJ[3 * l + j];
    }
};

}  // namespace hsddp
