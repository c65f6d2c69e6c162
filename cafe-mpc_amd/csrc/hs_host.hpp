// Host-side construction of the per-phase device descriptors from the C-ABI phase descriptors.
// Templated on a memory policy so that the SAME layout code is used by libhsddp_hip.so (HIP memory) and by the
// test-only lane emulator tests/_emu (plain host memory, used to debug kernel logic where no GPU exists).
#pragma once
#include <cstring>
#include <vector>
#include <algorithm>
#include "hsddp.h"
#include "hs_types.hpp"

#if !defined(__HIP_DEVICE_COMPILE__)   // host only: the device pass sees address-space qualified descriptor pointers (HS_GLOBAL)
namespace hs {

// reset maps that exist in the reference: WB->WB, WB->SRB (MHPCReset.cpp:4-52), SRB->SRB (identity), HKD->HKD (HKDReset.h)
inline bool phase_chain_ok(int model, int next_model) {
    if (model == HSDDP_MODEL_WB) return next_model == HSDDP_MODEL_WB || next_model == HSDDP_MODEL_SRB;
    return model == next_model;
}

// Mem policy: void* alloc(size_t bytes) (zero-filled, nullptr on failure); void upload(void* dst, const void* src, size_t bytes);
//             void replicate(void* base, size_t bytes_one, size_t count): copies record 0 into records 1..count-1
template <class Mem>
int setup_phase(Mem& mem, const hsddp_phase_desc_t& d, const hsddp_phase_desc_t* next, bool is_last, size_t B, PhaseDev& P, int slot0, bool f32 = false) {
    std::memset(&P, 0, sizeof(P));
    P.model = d.model; P.h = d.horizon; P.dt = d.dt; P.bg_alpha = d.BG_alpha;
    if (d.model == HSDDP_MODEL_WB) { P.n = 36; P.m = 12; P.p = 12; } else if (d.model == HSDDP_MODEL_SRB) { P.n = 12; P.m = 12; P.p = 0; } else { P.n = 24; P.m = 24; P.p = 0; }
    const size_t n = P.n, m = P.m, py = P.p;
    for (int l = 0; l < 4; l++) {
        P.contact[l] = d.contact[l]; P.next_contact[l] = d.next_contact[l];
        if (d.contact[l] > 0) P.feet[P.nc++] = l;
        P.td[l] = (d.contact[l] == 0 && d.next_contact[l] == 1) ? 1 : 0; P.n_td += P.td[l];
    }
    P.has_impact = P.n_td > 0; P.next_model = d.next_model; P.shooting = d.shooting; P.is_last = is_last ? 1 : 0;
    P.next_n = next ? (next->model == HSDDP_MODEL_WB ? 36 : next->model == HSDDP_MODEL_SRB ? 12 : 24) : P.n;
    std::memcpy(P.q, d.q, sizeof(P.q)); std::memcpy(P.r, d.r, sizeof(P.r)); std::memcpy(P.qf, d.qf, sizeof(P.qf));
    std::memcpy(P.w_foot_reg, d.w_foot_reg, 24); std::memcpy(P.w_swing_pos, d.w_swing_pos, 24); std::memcpy(P.w_swing_vel, d.w_swing_vel, 24);
    P.w_td_vel = d.w_td_vel;
    P.c_torque = d.c_torque; P.c_joint = d.c_joint; P.c_minheight = d.c_minheight; P.c_grf = d.c_grf; P.c_touchdown = d.c_touchdown;
    P.torque_limit = d.torque_limit; std::memcpy(P.joint_lb, d.joint_lb, 24); std::memcpy(P.joint_ub, d.joint_ub, 24);
    P.h_min = d.h_min; P.mu = d.mu; P.ground_height = d.ground_height;
    P.c_jspeed = d.c_jointspeed; P.jspeed_lb = d.jointspeed_lb; P.jspeed_ub = d.jointspeed_ub;
    const hsddp_reb_t* rb[5] = {&d.reb_torque, &d.reb_joint, &d.reb_minheight, &d.reb_grf, &d.reb_jointspeed};
    for (int g = 0; g < 5; g++) { P.reb_init[g][0] = rb[g]->delta; P.reb_init[g][1] = rb[g]->delta_min; P.reb_init[g][2] = rb[g]->eps; }
    P.al_init[0] = d.al_td.sigma; P.al_init[1] = d.al_td.lambda; P.al_init[2] = d.al_td.sigma_max;
    int ng = 0; P.go_torque = P.go_joint = P.go_height = P.go_grf = P.go_jspeed = -1;
    const bool wb = d.model == HSDDP_MODEL_WB;
    if (wb && d.c_torque) { P.go_torque = ng; ng += 24; }
    if (wb && d.c_jointspeed) { P.go_jspeed = ng; ng += 24; }
    if (wb && d.c_joint) { P.go_joint = ng; ng += 24; }
    if (d.c_minheight) { P.go_height = ng; ng += 1; }
    const bool hkd = d.model == HSDDP_MODEL_HKD, srb = d.model == HSDDP_MODEL_SRB;
    if (hkd) { ng = 0; P.go_height = -1; }                      // the HKD problem has the GRF pyramid only (HKDProblem.cpp:262-271)
    if ((wb || hkd) && d.c_grf && P.nc > 0) { P.go_grf = ng; ng += 5 * P.nc; }
    if (srb) { P.n_td = 0; P.has_impact = 0; for (int l = 0; l < 4; l++) P.td[l] = 0; }
    P.ng = ng; P.nt = (!srb && d.c_touchdown) ? P.n_td : 0; P.slot0 = slot0;
    { int offs[5], sz[5]; P.nobj = constraint_objects(P, offs, sz); P.obj_off = P.obj_sz = 0;
      for (int o = 0; o < P.nobj; o++) { P.obj_off |= (unsigned long long)offs[o] << (8 * o); P.obj_sz |= (unsigned long long)sz[o] << (8 * o); } }
    const size_t h1 = P.h + 1, hh = P.h;
    bool ok = true;
    auto up = [&](const double** dst, const double* src, size_t cnt) { void* p = mem.alloc(cnt * 8); if (!p) { ok = false; return; } if (src) mem.upload(p, src, cnt * 8); *dst = (const double*)p; };
    up(&P.xr, d.xr, h1 * n); up(&P.ur, d.ur, h1 * m); up(&P.yr, py ? d.yr : nullptr, h1 * std::max<size_t>(py, 1));
    up(&P.foot_pos, d.foot_pos, h1 * 12); up(&P.foot_vel, d.foot_vel, h1 * 12); up(&P.body_pos, d.body_pos, h1 * 3);
    { void* p = mem.alloc(h1 * 4 * sizeof(int)); if (!p) ok = false; else { if (d.ref_contact) mem.upload(p, d.ref_contact, h1 * 4 * sizeof(int)); P.ref_contact = (const int*)p; } }
    if (d.model == HSDDP_MODEL_WB) {      // packed per-knot reference record (PhaseDev::rref)
        std::vector<double> rr(h1 * 80, 0.0);
        for (size_t k = 0; k < h1; k++) {
            double* r = rr.data() + k * 80;
            if (d.xr) for (int i = 0; i < 36; i++) r[i] = d.xr[k * 36 + i];
            if (d.ur) for (int i = 0; i < 12; i++) r[36 + i] = d.ur[k * 12 + i];
            if (d.foot_vel) for (int i = 0; i < 12; i++) r[48 + i] = d.foot_vel[k * 12 + i];
            if (d.ref_contact) for (int i = 0; i < 4; i++) r[60 + i] = (double)d.ref_contact[k * 4 + i];
            for (int i = 0; i < 12; i++) r[64 + i] = (d.foot_pos ? d.foot_pos[k * 12 + i] : 0.0) - (d.body_pos ? d.body_pos[k * 3 + i % 3] : 0.0);
        }
        up(&P.rref, rr.data(), h1 * 80);
    }
    auto al = [&](double** dst, size_t cnt) { void* p = mem.alloc(std::max<size_t>(cnt, 1) * 8); if (!p) ok = false; *dst = (double*)p; };
    double** sx[] = {&P.X, &P.Xbar, &P.Xsim, &P.Defect, &P.Defect_bar, &P.dX, &P.G};
    for (auto p : sx) al(p, B * h1 * n);
    double** su[] = {&P.U, &P.Ubar, &P.dU, &P.Qu, &P.KdX};
    for (auto p : su) al(p, B * hh * m);
    al(&P.Y, B * hh * py);
    double** smn[] = {&P.K, &P.Qux};
    for (auto p : smn) al(p, B * hh * m * n);
    al(&P.Quu, B * hh * m * m);
    {   // record layout: the runtime twin of RecLayout<N,M,PY> (hs_types.hpp)
        auto rnd = [](size_t x) { return (x + 255) / 256 * 256; };
        const size_t ar = (size_t)rec_arows((int)n);      // stored rows of A and B (whole body: the lower 18)
        const size_t oA = 0, oLxx = oA + rnd(ar * n), oB = oLxx + rnd(n * n), oC = oB + rnd(ar * m), oD = oC + rnd(py * n), oLuu = oD + rnd(py * m),
                     oLyy = oLuu + rnd(m * m), oLx = oLyy + rnd(py * py), oLu = oLx + n, oLy = oLu + m, size = oLx + rnd(n + m + py);
        P.rs = (int)size;
        P.oA = (int)oA; P.oLxx = (int)oLxx; P.oB = (int)oB; P.oC = (int)oC; P.oD = (int)oD; P.oLuu = (int)oLuu; P.oLyy = (int)oLyy; P.oLx = (int)oLx; P.oLu = (int)oLu; P.oLy = (int)oLy;
        if (f32) { void* p = mem.alloc(std::max<size_t>(B * hh * size, 1) * sizeof(float)); if (!p) ok = false; P.rec32 = (float*)p; }
        else al(&P.rec, B * hh * size);
        if (P.rec) { P.A = P.rec + oA; P.lxx = P.rec + oLxx; P.B = P.rec + oB; P.C = P.rec + oC; P.D = P.rec + oD; P.luu = P.rec + oLuu;
                     P.lyy = P.rec + oLyy; P.lx = P.rec + oLx; P.lu = P.rec + oLu; P.ly = P.rec + oLy; }
    }
    al(&P.l, B * hh); al(&P.lbase, B * hh);
    al(&P.Phi, B); al(&P.Phibase, B); al(&P.Phix, B * n); al(&P.Phixx, B * n * n); al(&P.H0, B * n * n); al(&P.Px, B * (size_t)P.next_n * n);
    al(&P.g, B * hh * ng); al(&P.delta, B * hh * ng); al(&P.eps, B * hh * ng);
    if (wb) al(&P.kc, B * hh * KC_SIZE);
    al(&P.th, B * P.nt); al(&P.sigma, B * P.nt); al(&P.lambda, B * P.nt);
    if (!ok) return HSDDP_ENOMEM;
    // initial ReB / AL parameters (initialize_params, ConstraintsBase.h:173-180, 362-366)
    if (ng > 0) {
        std::vector<double> dl(hh * ng), ep(hh * ng);
        for (size_t k = 0; k < hh; k++) for (int c = 0; c < ng; c++) { int grp = constraint_group(P, c); dl[k * ng + c] = P.reb_init[grp][0]; ep[k * ng + c] = P.reb_init[grp][2]; }
        mem.upload(P.delta, dl.data(), dl.size() * 8); mem.replicate(P.delta, dl.size() * 8, B);
        mem.upload(P.eps, ep.data(), ep.size() * 8); mem.replicate(P.eps, ep.size() * 8, B);
    }
    if (P.nt > 0) { std::vector<double> sg(B * P.nt, P.al_init[0]), lm(B * P.nt, P.al_init[1]); mem.upload(P.sigma, sg.data(), sg.size() * 8); mem.upload(P.lambda, lm.data(), lm.size() * 8); }
    return HSDDP_OK;
}

inline const double* field_dev(const PhaseDev& P, int f, int& count, int& elems) {
    const int n = P.n, m = P.m, p = P.p, h = P.h;
    switch (f) {
        case HSDDP_F_X: count = h + 1; elems = n; return P.X;
        case HSDDP_F_XBAR: count = h + 1; elems = n; return P.Xbar;
        case HSDDP_F_XSIM: count = h + 1; elems = n; return P.Xsim;
        case HSDDP_F_DEFECT: count = h + 1; elems = n; return P.Defect;
        case HSDDP_F_DX: count = h + 1; elems = n; return P.dX;
        case HSDDP_F_G: count = h + 1; elems = n; return P.G;
        case HSDDP_F_U: count = h; elems = m; return P.U;
        case HSDDP_F_UBAR: count = h; elems = m; return P.Ubar;
        case HSDDP_F_DU: count = h; elems = m; return P.dU;
        case HSDDP_F_QU: count = h; elems = m; return P.Qu;
        case HSDDP_F_Y: count = h; elems = p; return P.Y;
        case HSDDP_F_K: count = h; elems = m * n; return P.K;
        case HSDDP_F_QUX: count = h; elems = m * n; return P.Qux;
        case HSDDP_F_QUU: count = h; elems = m * m; return P.Quu;
        case HSDDP_F_A: count = h; elems = n * n; return P.A;
        case HSDDP_F_B: count = h; elems = n * m; return P.B;
        case HSDDP_F_C: count = h; elems = p * n; return P.C;
        case HSDDP_F_D: count = h; elems = p * m; return P.D;
        case HSDDP_F_L: count = h; elems = 1; return P.l;
        case HSDDP_F_LX: count = h; elems = n; return P.lx;
        case HSDDP_F_LU: count = h; elems = m; return P.lu;
        case HSDDP_F_LY: count = h; elems = p; return P.ly;
        case HSDDP_F_LXX: count = h; elems = n * n; return P.lxx;
        case HSDDP_F_LUX: count = h; elems = m * n; return nullptr;   // identically zero for every shipped cost: not stored
        case HSDDP_F_LUU: count = h; elems = m * m; return P.luu;
        case HSDDP_F_LYY: count = h; elems = p * p; return P.lyy;
        case HSDDP_F_PHI: count = 1; elems = 1; return P.Phi;
        case HSDDP_F_PHIX: count = 1; elems = n; return P.Phix;
        case HSDDP_F_PHIXX: count = 1; elems = n * n; return P.Phixx;
        case HSDDP_F_H0: count = 1; elems = n * n; return P.H0;
        case HSDDP_F_REB_EPS: count = h; elems = P.ng; return P.eps;
        case HSDDP_F_REB_DELTA: count = h; elems = P.ng; return P.delta;
        case HSDDP_F_AL_SIGMA: count = 1; elems = P.nt; return P.sigma;
        case HSDDP_F_AL_LAMBDA: count = 1; elems = P.nt; return P.lambda;
        default: count = 0; elems = 0; return nullptr;
    }
}

// Whole-body A / B as the ABI hands them out (36 x 36 / 36 x 12, column-major) from the stored lower halves (RecLayout): the upper
// rows are the forward-Euler identities A = [I, dt I; ...], B = [0; ...] (WBM.cpp:68, 122-125).
inline bool wb_structured(const PhaseDev& P, int f) { return P.model == HSDDP_MODEL_WB && (f == HSDDP_F_A || f == HSDDP_F_B); }
inline void wb_expand_ab(int f, double dt, const double* stored, double* full) {
    const int cols = f == HSDDP_F_A ? 36 : 12;
    for (int c = 0; c < cols; c++) for (int r = 0; r < 36; r++) {
        double v;
        if (r >= 18) v = stored[(r - 18) + 18 * c];
        else v = f == HSDDP_F_A ? ((c == r ? 1.0 : 0.0) + (c == 18 + r ? dt : 0.0)) : 0.0;
        full[r + 36 * c] = v;
    }
}

// returns the base pointer of field f; knot record kk of problem b starts at base + (b*count + kk)*stride
inline const double* field_dev(const PhaseDev& P, int f, int& count, int& elems, int& stride) {
    const double* base = field_dev(P, f, count, elems);
    const bool in_rec = (f == HSDDP_F_A || f == HSDDP_F_B || f == HSDDP_F_C || f == HSDDP_F_D || f == HSDDP_F_LX || f == HSDDP_F_LU || f == HSDDP_F_LY ||
                         f == HSDDP_F_LXX || f == HSDDP_F_LUU || f == HSDDP_F_LYY);
    stride = in_rec ? P.rs : elems;
    return base;
}

}  // namespace hs
#endif
