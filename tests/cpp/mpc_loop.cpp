// The receding-horizon loop of the reference on the reference-shaped C++ host path, timed per tick INCLUDING the host work:
//   MHPCLocomotion::update (MHPC/MHPCLocomotion.cpp:91-142): opt_problem.update(), new solver, set_initial_condition, set_multiPhaseProblem,
//   solve(ddp_setting, dt_mpc * 1000 * 0.9), publish_mpc_cmd, solver_info  -- and its simulator-free form testTrajOptInLoop.cpp:85-117.
// Here: hsddp::MhpcProblemData::update + describe (cafe-mpc_amd/host/mhpc_builder.hpp), hsddp::MultiPhaseDDP<double>::reconfigure /
// set_initial_condition / solve(opt, 0.9 dt_mpc) / export_mpc_command / export_solver_info (cafe-mpc_amd/host/MultiPhaseDDP.hpp) over the C-ABI
// of whatever backend library the binary is linked against.  Prints one JSON object: per-tick wall time (mean / max over the warm ticks), the
// same split by call, iterations and cost per tick (tests compare them with the Python path), device allocations during the warm ticks.
//   mpc_loop <cafe_tree> <gait> <option.bin> <n_ticks> [budget: 1 = solve under max_cputime = 0.9 dt_mpc (default), 0 = unlimited]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include "mhpc_builder.hpp"
#include "MultiPhaseDDP.hpp"

using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const std::string root = argv[1], gait = argv[2], optfile = argv[3]; const int n_ticks = std::atoi(argv[4]);
    const bool budget = argc < 6 || std::atoi(argv[5]) != 0;
    hsddp::HSDDP_OPTION opt0 = hsddp::default_option();
    { std::ifstream f(optfile, std::ios::binary); if (!f.read(reinterpret_cast<char*>(&opt0), sizeof(opt0))) return 4; }
    hsddp::HSDDP_OPTION opt_rt = opt0; opt_rt.max_AL_iter = opt0.max_AL_iter_runtime; opt_rt.max_DDP_iter = opt0.max_DDP_iter_runtime;      // MHPCLocomotion.cpp:113-115
    auto cfg = hsddp::load_mhpc_config(root + "/MHPC/settings/mhpc_config.info");
    auto costs = hsddp::load_cost_weights(root + "/" + cfg.costFile);
    auto cpar = hsddp::load_constraint_params(root + "/" + cfg.constraintParamFile);
    hsddp::QuadReference ref; if (!ref.load(root + "/Reference/Data/" + gait + "/quad_reference.csv", false)) return 3;
    hsddp::MhpcProblemData pd(ref, cfg, costs, cpar);
    std::vector<hsddp::PhaseBuffers> bufs; auto descs = pd.describe(bufs);
    std::vector<int> uids; for (auto& r : pd.wb) uids.push_back(r.uid); if (pd.srb_h > 0) uids.push_back(-1);

    hsddp::MultiPhaseDDP<double> solver(1, 0);
    solver.set_initial_condition(std::vector<double>(bufs[0].Xbar.begin(), bufs[0].Xbar.begin() + 36));
    solver.set_multiPhaseProblem(descs);
    if (solver.last_error()) { std::fprintf(stderr, "create failed: %d\n", solver.last_error()); return 5; }
    for (size_t i = 0; i < descs.size(); i++) solver.set_nominal((int)i, bufs[i].Xbar.data(), bufs[i].Ubar.data());
    solver.solve(opt0);
    if (solver.last_error()) { std::fprintf(stderr, "initial solve failed: %d\n", solver.last_error()); return 6; }
    const int nst = (int)std::round((double)cfg.dt_mpc / cfg.dt_wb);
    const float max_cputime = budget ? cfg.dt_mpc * 1000.0f * 0.9f : 1e6f;      // MHPCLocomotion.cpp:122

    struct Tick { double total, state, build, reconf, setic, solve, exprt; int iters, status; double cost; };
    std::vector<Tick> ticks;
    long long m0 = 0, m1 = 0;
    for (int tick = 1; tick <= n_ticks; tick++) {
        Tick T{}; auto t0 = clk::now(), ta = t0;
        // state after one MPC step on the previous plan = next initial condition (testTrajOptInLoop.cpp:103-106 feeds the plan back)
        std::vector<double> x0(36);
        { auto xb = solver.get_field(0, HSDDP_F_XBAR); const int h0 = (int)descs[0].horizon;
          if (h0 >= nst) std::copy(xb.begin() + (size_t)nst * 36, xb.begin() + (size_t)(nst + 1) * 36, x0.begin());
          else { auto xb1 = solver.get_field(1, HSDDP_F_XBAR); std::copy(xb1.begin() + (size_t)(nst - h0) * 36, xb1.begin() + (size_t)(nst - h0 + 1) * 36, x0.begin()); } }
        T.state = ms_since(ta); ta = clk::now();
        auto moves = pd.update();
        std::vector<hsddp::PhaseBuffers> nb; auto nd = pd.describe(nb);
        std::map<int, int> old_index; for (size_t i = 0; i < uids.size(); i++) old_index[uids[i]] = (int)i;
        std::vector<int> nu; for (auto& r : pd.wb) nu.push_back(r.uid); if (pd.srb_h > 0) nu.push_back(-1);
        std::vector<int> src(nu.size(), -1), shift(nu.size(), 0);
        for (size_t i = 0; i < nu.size(); i++) {
            if (nu[i] == -1) { src[i] = old_index[-1]; shift[i] = pd.srb_steps; }
            else if (old_index.count(nu[i])) { src[i] = old_index[nu[i]]; for (auto& mv : moves) if (mv.uid == nu[i]) shift[i] = mv.popped; }
        }
        T.build = ms_since(ta); ta = clk::now();
        solver.reconfigure(nd, src, shift);
        if (solver.last_error()) { std::fprintf(stderr, "reconfigure failed at tick %d: %d\n", tick, solver.last_error()); return 7; }
        T.reconf = ms_since(ta); ta = clk::now();
        solver.set_initial_condition(x0);
        T.setic = ms_since(ta); ta = clk::now();
        solver.solve(opt_rt, max_cputime);
        T.solve = ms_since(ta); ta = clk::now();
        auto cmd = solver.export_mpc_command(0, 8, 0.01 * tick, cfg.dt_wb);      // publish_mpc_cmd: first 8 knots (MHPCLocomotion.cpp:190-287)
        auto si = solver.export_solver_info(0);
        T.exprt = ms_since(ta);
        T.total = ms_since(t0);
        if (solver.last_error() || cmd[0] != 8u) { std::fprintf(stderr, "tick %d failed: %d\n", tick, solver.last_error()); return 8; }
        T.iters = si.n_iter; T.status = solver.status(); T.cost = solver.get_actual_cost();
        ticks.push_back(T);
        descs = nd; bufs.swap(nb); uids = nu;
        if (tick == 4) m0 = hsddp_debug_malloc_count();
        m1 = hsddp_debug_malloc_count();
    }
    auto stat = [&](double Tick::*f, double& mean, double& mx) { mean = 0; mx = 0; int n = 0; for (size_t i = 4; i < ticks.size(); i++) { mean += ticks[i].*f; mx = std::max(mx, ticks[i].*f); n++; } mean /= std::max(n, 1); };
    std::printf("{\"ticks\":%d,\"budget_ms\":%.3f,\"max_cputime_ms\":%.3f", n_ticks, (double)cfg.dt_mpc * 1000.0, (double)max_cputime);
    const char* names[] = {"total", "state_readback", "descriptor_build", "reconfigure", "set_initial_condition", "solve", "export"};
    double Tick::*fields[] = {&Tick::total, &Tick::state, &Tick::build, &Tick::reconf, &Tick::setic, &Tick::solve, &Tick::exprt};
    for (int q = 0; q < 7; q++) { double mean, mx; stat(fields[q], mean, mx); std::printf(",\"%s_ms_mean\":%.4f,\"%s_ms_max\":%.4f", names[q], mean, names[q], mx); }
    std::printf(",\"device_allocations_in_warm_ticks\":%lld,\"iters\":[", m1 - m0);
    for (size_t i = 0; i < ticks.size(); i++) std::printf("%s%d", i ? "," : "", ticks[i].iters);
    std::printf("],\"status\":[");
    for (size_t i = 0; i < ticks.size(); i++) std::printf("%s%d", i ? "," : "", ticks[i].status);
    std::printf("],\"cost\":[");
    for (size_t i = 0; i < ticks.size(); i++) std::printf("%s%.17g", i ? "," : "", ticks[i].cost);
    std::printf("],\"total_ms\":[");
    for (size_t i = 0; i < ticks.size(); i++) std::printf("%s%.3f", i ? "," : "", ticks[i].total);
    std::printf("]}\n");
    return 0;
}
