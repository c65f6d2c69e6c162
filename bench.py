#!/usr/bin/env python3
"""bench.py — DDP iterations/sec of the batched HS-DDP solve (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--strong]

Workload (config.workload), weak scaling (default): BASELINE.json configs[2] — Mini-Cheetah whole-body, N=200 knots as 4 contact
phases (1111 -> 0110 -> 1001 -> 0110, dt=0.01), an ensemble of 4096 initial states PER GPU, synthetic inputs,
fixed-work mode: max_AL_iter=1, cost_thresh=0, so every problem runs exactly `--steps` DDP iterations (10).
`--strong`: BASELINE.json configs[3] — the running barrel roll, 8 hybrid phases / 350 knots, 8192 problems IN TOTAL split over the
ranks (8192/8 = 1024 per GPU at N=8).
A "step" = one DDP iteration (cost, LQ approximation, regularised Riccati sweep, linear rollout, line search, nominal update:
MultiPhaseDDP.cpp:277-387) of the whole batch.  Inputs are resident in HBM before the timed region.  One process per GPU; the only
collective is the RCCL all-gather of the per-problem results (cafe-mpc_amd/launch.py).  Started without a launcher, `--gpus N`
starts its own N ranks (before touching the GPU) and fails loudly if the node has fewer GPUs.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

# algorithmic HBM bytes per knot per DDP iteration (WB 36/12/12, fp64), by kernel family
ALG_BYTES = {
    "k_lq": (150 + 4381) * 8,               # K2: read x,u,y,refs; write A,B,C,D + RCostData
    "k_sweep": (4417 + 1068 + 4128 + 36) * 8,   # K3 Riccati + K4 linear rollout (one fused launch)
    "k_rollout": (528 + 133 + 69) * 8,      # K1 per line-search trial
    "k_ls_probe": (528 + 133 + 69) * 8,     # K1 per candidate step of a batched line-search launch
}
# the same for the kinodynamic model (HKD 24/24/0, BASELINE config 5): elements per knot x bytes per element of each stream.  LQ: reads x, u, refs
# (~100 fp64), writes A, B, lxx, luu, lx, lu (2352 record elements); sweep: Riccati reads the record + Defect and writes K, dU, Qu, Quu, Qux, G
# (1800 fp64), the linear rollout reads the record again + K, dU, Defect and writes dX; rollout: K, Xbar, dX, Ubar, dU + outputs (672 + 117 fp64)
def hkd_alg_bytes(rec_bytes):
    return {"k_lq": 100 * 8 + 2352 * rec_bytes, "k_sweep": 2 * 2352 * rec_bytes + (24 + 1800 + 576 + 24 + 24 + 24) * 8,
            "k_rollout": (672 + 117) * 8, "k_ls_probe": (672 + 117) * 8}


HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
METRIC = "DDP iterations/sec, Mini-Cheetah WB N=200, batch=4096; 1/2/4/8 GPU"


def cpu_baseline(pkg, phases_fn, x0_fn, opt_fn, steps, nprob=64):
    """Oracle (CPU restatement of the reference) on this box's host cores, SAME workload and SAME number of DDP iterations per
    problem as the GPU run, on a bounded sample of the ensemble — reported baseline, not the target."""
    path = os.path.join(ROOT, "oracle", "liboracle_hsddp.so")
    if not os.path.exists(path):
        return None, None
    lib = pkg._abi.bind(ctypes.CDLL(path))
    keep = None
    lib.oracle_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    # the GPU box exposes every host CPU in the affinity mask but a one-GPU job owns a 16-core share: size the pool to that
    cores = min(len(os.sched_getaffinity(0)), 16)
    phases = phases_fn()
    opt = opt_fn(steps)
    out = {}
    for kind, n, lq_thr, pb_thr in (("all_cores", nprob, 1, cores), ("reference_shaped_4thr", 2, 4, 1)):
        s = pkg.Solver(lib, phases, batch=n)
        for i, p in enumerate(phases):
            s.set_nominal(i, p["Xbar"], p["Ubar"])
        s.set_initial_condition(x0_fn(n))
        lib.oracle_set_threads(s.h, lq_thr, pb_thr)
        t0 = time.time(); s.solve(opt); dt = time.time() - t0
        info = s.info_arrays()
        out[kind] = (float(info["n_iters"].sum()) / dt, float(info["n_ls_iters"].sum()) / max(float(info["n_iters"].sum()), 1.0), dt)
        if kind == "all_cores":
            keep = s            # (kept open: parity_sample compares its results with the GPU's first problems)
        else:
            s.close()
    return keep, {"value": out["all_cores"][0], "unit": "DDP iterations/s", "cores": cores, "kind": "port",
            "mean_ls_trials_per_iter": out["all_cores"][1], "seconds": round(out["all_cores"][2], 2),
            "sample": f"the first {nprob} problems of the same ensemble x {steps} iterations each (same options as the GPU run), one problem "
                      f"per core on {cores} threads; reference-shaped run (1 problem at a time, 4 OpenMP threads over knots in "
                      f"LQ_approximation only, 2 problems): {out['reference_shaped_4thr'][0]:.3f} it/s"}


def parity_sample(sg, so, nph):
    """The problems cpu_baseline solved on the oracle against the SAME problems of the timed GPU run (the oracle as checker of the bench's own
    output): decisions must be identical, values agree to the parity tests' per-solve tolerances (tests/test_gpu_parity.py)."""
    n = so.batch
    ig, io = sg.info_arrays(), so.info_arrays()
    out = {"problems": n, "n_iters_equal": bool(np.array_equal(ig["n_iters"][:n], io["n_iters"])), "n_ls_iters_equal": bool(np.array_equal(ig["n_ls_iters"][:n], io["n_ls_iters"])),
           "n_reg_iters_equal": bool(np.array_equal(ig["n_reg_iters"][:n], io["n_reg_iters"])), "status_equal": bool(np.array_equal(ig["status"][:n], io["status"])),
           "max_rel_dcost": float(np.max(np.abs(ig["actual_cost"][:n] - io["actual_cost"]) / np.maximum(np.abs(io["actual_cost"]), 1e-300))),
           "max_abs_dfeas": float(np.max(np.abs(ig["dyn_feas"][:n] - io["dyn_feas"])))}
    dK0 = dKall = dX = 0.0; kmax = 0.0
    for i in range(nph):
        kg, ko = sg.field(i, "K", 0, n), so.field(i, "K")
        if i == 0:
            dK0 = float(np.abs(kg[:, 0] - ko[:, 0]).max())
        dKall = max(dKall, float(np.abs(kg - ko).max())); kmax = max(kmax, float(np.abs(ko).max()))
        dX = max(dX, float(np.abs(sg.field(i, "XBAR", 0, n) - so.field(i, "XBAR")).max()))
    out.update({"max_dK_inf_knot0": dK0, "max_dK_inf_all_knots": dKall, "max_abs_K": kmax, "max_dXbar_inf": dX,
                "pass": bool(out["n_iters_equal"] and out["n_ls_iters_equal"] and out["status_equal"] and dKall < 1e-6 and out["max_rel_dcost"] < 1e-6)})
    return out


def load_profile_json(pkg, name):
    """profiles/<name>: cited only while it was measured on the kernel sources this run executes (kernel_source_hash recorded by tools/pmc_summary.py)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    return d if d.get("kernel_source_hash") == pkg.kernel_source_hash() else None


def latency_probe(pkg, steps):
    """BASELINE config 2 (batch 1, N=200): ms per DDP iteration of ONE problem — the latency side of the same kernels."""
    phases = pkg.problems.wb_trot_problem()
    s = pkg.MultiPhaseDDP(phases, batch=1)
    s.set_initial_condition(pkg.problems.wb_ensemble_x0(1, 20241220 + 2))
    opt = pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=2, cost_thresh=0.0)
    s.solve(opt)                                           # warm-up (module load, first launches)
    s2 = pkg.MultiPhaseDDP(phases, batch=1)
    s2.set_initial_condition(pkg.problems.wb_ensemble_x0(1, 20241220 + 2))
    opt = pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=steps, cost_thresh=0.0)
    t0 = time.perf_counter(); s2.solve(opt); dt = time.perf_counter() - t0
    info = s2.info_arrays()
    out = {"config": "WB N=200 (4 x 50), batch 1", "ms_per_ddp_iteration": dt * 1e3 / max(int(info["n_iters"][0]), 1),
           "ls_trials_per_iter": float(info["n_ls_iters"][0]) / max(int(info["n_iters"][0]), 1)}
    s.close(); s2.close()
    return out


def mpc_tick_probe(pkg, ticks=24):
    """The receding-horizon loop the reference runs under an 18 ms budget (MHPCLocomotion.cpp:109-122, testTrajOptInLoop.cpp:85-117): shipped
    bound gait, 25 whole-body + 10 SRB knots, runtime limits (4 AL x 1 DDP), batch 1, one handle kept across ticks (hsddp_reconfigure).
    Timed per tick: reconfigure + set_initial_condition + solve + policy export (the ABI calls); the host-side descriptor building is
    reported separately (here it is the Python mirror of the builder, in the reference it is C++)."""
    import importlib
    builder = importlib.import_module(pkg.__name__ + ".builder")
    tree = os.path.join(ROOT, "tests", "golden", "cafe_tree")
    if not os.path.isdir(tree):
        return None
    cfg = builder.load_mhpc_config(tree + "/MHPC/settings/mhpc_config.info")
    pd = builder.MHPCProblemData(builder.QuadReference(tree + "/Reference/Data/bound/quad_reference.csv"), cfg,
                                 builder.load_cost_weights(tree + "/" + cfg["costFile"]), builder.load_constraint_params(tree + "/" + cfg["constraintParamFile"]))
    opt0 = builder.load_ddp_setting(tree + "/MHPC/settings/ddp_setting.info")
    opt_rt = builder.load_ddp_setting(tree + "/MHPC/settings/ddp_setting.info")
    opt_rt.max_AL_iter, opt_rt.max_DDP_iter = opt_rt.max_AL_iter_runtime, opt_rt.max_DDP_iter_runtime
    phases, info = pd.describe(ubar_mode="gravity_comp")
    s = pkg.MultiPhaseDDP(phases, batch=1)
    s.set_initial_condition(info["x0"][None]); s.solve(opt0)
    nst = int(round(float(cfg["dt_mpc"]) / cfg["dt_wb"]))
    abi_ms, host_ms, iters = [], [], []
    lib = pkg.load_hip_library()
    m0 = None
    for tick in range(ticks):
        xg = s.field(0, "XBAR")
        x0n = np.ascontiguousarray(xg[:, nst] if xg.shape[1] > nst else s.field(1, "XBAR")[:, nst - xg.shape[1] + 1])
        t0 = time.perf_counter()
        m = pd.update(); new_phases, inf = pd.describe(ubar_mode="gravity_comp")
        old_index = {p.get("uid"): i for i, p in enumerate(phases)}
        src = [old_index[-1] if p.get("uid") == -1 else old_index.get(p.get("uid"), -1) for p in new_phases]
        shift = [getattr(pd, "srb_steps", 0) if p.get("uid") == -1 else (m[p.get("uid")][0] if p.get("uid") in old_index else 0) for p in new_phases]
        t1 = time.perf_counter()
        s.reconfigure(new_phases, src, shift); s.set_initial_condition(x0n); s.solve(opt_rt)
        s.export_mpc_command(problem=0, n_steps=8, mpc_time=0.01 * tick, dt=cfg["dt_wb"])
        t2 = time.perf_counter()
        phases = new_phases
        if tick == 4:
            m0 = lib.hsddp_debug_malloc_count()
        host_ms.append((t1 - t0) * 1e3); abi_ms.append((t2 - t1) * 1e3); iters.append(int(s.info_arrays()["n_iters"][0]))
    warm = abi_ms[4:]
    out = {"config": "shipped bound gait, 25 WB + 10 SRB knots, 4 AL x 1 DDP per tick, batch 1, one handle (hsddp_reconfigure)",
           "budget_ms": 18.0, "ms_per_tick_mean": float(np.mean(warm)), "ms_per_tick_max": float(np.max(warm)), "ddp_iterations_per_tick": float(np.mean(iters[4:])),
           "device_allocations_in_warm_ticks": int(lib.hsddp_debug_malloc_count() - m0), "host_descriptor_build_ms_python": float(np.mean(host_ms[4:]))}
    s.close()
    return out


def mpc_tick_cpp_probe(ticks=24):
    """The same loop on the reference-shaped C++ host path (tests/cpp/mpc_loop.cpp: C++ problem builder update + describe, reconfigure,
    set_initial_condition, solve(opt, 0.9 dt_mpc), command + solver-info export), per-tick wall time INCLUDING the descriptor building - the
    figure to hold against the reference's 18 ms (MHPCLocomotion.cpp:122).  Compiled here with g++ against libhsddp_hip.so."""
    import shutil, subprocess, tempfile
    tree = os.path.join(ROOT, "tests", "golden", "cafe_tree")
    if shutil.which("g++") is None or not os.path.isdir(tree):
        return None
    pkg = ge.load_package()
    import importlib
    builder = importlib.import_module(pkg.__name__ + ".builder")
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "mpc_loop")
        try:
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cafe-mpc_amd", "host"),
                                   os.path.join(ROOT, "tests", "cpp", "mpc_loop.cpp"), "-L", os.path.join(ROOT, "cafe-mpc_amd"), "-lhsddp_hip",
                                   "-Wl,-rpath," + os.path.join(ROOT, "cafe-mpc_amd"), "-o", exe], timeout=120)
            open(os.path.join(td, "opt.bin"), "wb").write(bytes(builder.load_ddp_setting(tree + "/MHPC/settings/ddp_setting.info")))
            out = json.loads(subprocess.check_output([exe, tree, "bound", os.path.join(td, "opt.bin"), str(ticks)], timeout=120))
        except Exception as e:      # (a measurement leg: never takes the bench line down)
            return {"error": str(e)[:200]}
    keep = {k: v for k, v in out.items() if k.endswith(("_mean", "_max")) or k in ("ticks", "budget_ms", "max_cputime_ms", "device_allocations_in_warm_ticks")}
    keep["config"] = "tests/cpp/mpc_loop.cpp on libhsddp_hip.so: shipped bound gait, 25 WB + 10 SRB knots, runtime limits (4 AL x 1 DDP), batch 1, max_cputime = 0.9 dt_mpc"
    keep["all_ticks_status_0"] = all(v == 0 for v in out["status"])
    keep["total_ms_per_tick"] = [round(float(v), 3) for v in out["total_ms"]]      # (every tick, the first four are not in the statistics)
    keep["iterations_per_tick"] = out["iters"]
    return keep


def host_boundary_probe(s, x0, opt, dt_resident):
    """The C-ABI takes HOST buffers (include/hsddp.h: set_initial_condition, set_nominal) and hands host buffers back (get_info).  `value` is timed with the inputs
    resident; this leg times the same solve on the warm handle INCLUDING the transfers, twice: (a) what the ensemble workload needs - x0 per problem, one shared nominal
    trajectory per phase, the 64-byte result struct per problem back; (b) the worst case the boundary allows - a per-problem nominal trajectory for every phase as well."""
    import time as _t
    B = s.batch
    shared = [(p["Xbar"], p["Ubar"]) for p in s.phases]
    per_problem = [(np.ascontiguousarray(np.broadcast_to(np.asarray(xb)[None], (B,) + np.asarray(xb).shape)), np.ascontiguousarray(np.broadcast_to(np.asarray(ub)[None], (B,) + np.asarray(ub).shape)))
                   for xb, ub in shared]
    out = {}
    for name, noms in (("shared_nominal", shared), ("per_problem_nominal", per_problem)):
        t0 = _t.perf_counter()
        s.set_initial_condition(x0)
        for i, (xb, ub) in enumerate(noms):
            s.set_nominal(i, xb, ub)
        t1 = _t.perf_counter()
        s.solve(opt)
        t2 = _t.perf_counter()
        info = s.info_arrays()
        t3 = _t.perf_counter()
        up = x0.nbytes + sum(np.asarray(xb).nbytes + np.asarray(ub).nbytes for xb, ub in noms)
        out[name] = {"upload_MB": round(up / 1e6, 2), "upload_ms": round((t1 - t0) * 1e3, 2), "solve_ms": round((t2 - t1) * 1e3, 2), "download_ms": round((t3 - t2) * 1e3, 2),
                     "download_MB": round(B * 64 / 1e6, 3), "value_inclusive": float(info["n_iters"].sum()) / (t3 - t0)}
    out["note"] = ("same handle, same options as the timed solve (resident inputs: %.1f ms); the result download is Python-side struct unpacking more than PCIe" % (dt_resident * 1e3))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="problems per GPU (weak, default 4096) / in total (--strong, default 8192)")
    ap.add_argument("--strong", action="store_true", help="fixed total work: config 4 (barrel roll, 8 phases, N=350), 8192 problems split over the ranks")
    ap.add_argument("--hkd", choices=["f32", "f64"], default=None, help="BASELINE config 5 instead: HKD 24/24/0, N=200 bound-gait schedule, 16384 problems per GPU, "
                                                                             "on an fp32 handle (fp32 LQ records + fp32 MFMA sweep) or an fp64 one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    args = ap.parse_args()

    pkg = ge.load_package()
    launch = pkg.launch
    rc = launch.maybe_spawn(args.gpus, os.path.abspath(__file__), sys.argv[1:])      # before anything touches the GPU
    if rc is not None:
        sys.exit(rc)
    import torch
    rank, world, local, dist = launch.init_ranks("nccl")
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but {world} rank(s) are running", file=sys.stderr); sys.exit(2)
    dev = f"cuda:{local}"

    if args.strong:
        total = args.batch or 8192
        first, B = launch.shard(total, world, rank)
        seed = 20241220 + 4
        phases_fn = lambda: pkg.problems.barrel_roll_running_problem()[0]   # noqa: E731
        xinit = pkg.problems.barrel_roll_running_problem()[1]

        x0_fn = lambda n, first=0: pkg.problems.barrel_roll_ensemble_x0(n, seed, xinit, first=first)   # noqa: E731
        opt_fn = lambda k: pkg.problems.br_ddp_setting(max_AL_iter=1, max_DDP_iter=k, cost_thresh=0.0)   # noqa: E731
        workload = ("barrel roll with running lead-out, WB 36/12/12, 8 hybrid phases / N=350: 1111(12) 0101(21) 0000(42) 1111(15) 0000(20) "
                    "1111(15) 0101(100) 1010(125), dt=0.01, zero-torque start, ensemble of initial joint poses, fixed-work mode")
    elif args.hkd:
        B = args.batch or 16384
        total = B * world; first = rank * B
        seed = 20241220 + 5
        phases_fn = pkg.problems.hkd_bound_problem
        hkd_ref = pkg.problems.hkd_bound_problem()
        x0_fn = lambda n, first=0: pkg.problems.hkd_ensemble_x0(n, seed, hkd_ref, first=first)   # noqa: E731
        opt_fn = lambda k: pkg.problems.hkd_ddp_setting(max_AL_iter=1, max_DDP_iter=k, cost_thresh=0.0)   # noqa: E731
        workload = ("HKD 24/24/0 (hybrid kinodynamic), N=200 in the contact pattern of the shipped bound gait: 1111(6), then 1100(10) 0000(10) 0011(10) "
                    "0000(10) repeating (21 phases), dt=0.01, ensemble of initial body states, fixed-work mode; " +
                    ("fp32 handle: fp32 LQ records, Riccati sweep + linear rollout on v_mfma_f32_16x16x4_f32, rollouts / LQ knots / merit in fp64" if args.hkd == "f32" else "fp64 handle"))
    else:
        B = args.batch or 4096
        total = B * world; first = rank * B
        seed = 20241220 + 3
        phases_fn = pkg.problems.wb_trot_problem
        x0_fn = lambda n, first=0: pkg.problems.wb_ensemble_x0(n, seed, first=first)   # noqa: E731
        opt_fn = lambda k: pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=k, cost_thresh=0.0)   # noqa: E731
        workload = ("WB 36/12/12, N=200 = 4 contact phases x 50 knots (1111,0110,1001,0110), dt=0.01, ensemble of initial states, "
                    "fixed-work mode (max_AL_iter=1, cost_thresh=0)")
    x0 = x0_fn(B, first)

    prec = pkg.PREC_F32 if args.hkd == "f32" else pkg.PREC_F64
    alg_bytes = hkd_alg_bytes(4 if args.hkd == "f32" else 8) if args.hkd else ALG_BYTES
    alg_total = sum(alg_bytes[k] for k in ("k_lq", "k_sweep", "k_rollout"))

    def run(iters):
        s = pkg.MultiPhaseDDP(phases_fn(), batch=B, device=local, precision=prec)
        s.set_initial_condition(x0)
        return s, opt_fn(iters)

    if args.warmup > 0:
        s, opt = run(args.warmup); s.solve(opt); s.close()
    s, opt = run(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    s.solve(opt)                       # synchronises its stream before returning
    my_solve_ms = (time.perf_counter() - t0) * 1e3
    rows = launch.result_rows(s.info_arrays())
    res_all = launch.gather_results(dist, rows, dev)      # C1: all-gather of the per-problem result struct (64 B/problem) over RCCL/xGMI
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt = launch.max_over_ranks(dist, dt, dev)
    ranks_seen = int(round(launch.sum_over_ranks(dist, 1.0, dev)))
    per_rank = launch.gather_scalars(dist, [my_solve_ms, float(rows[:, 4].sum()), float(rows[:, 5].sum())], dev)      # solve ms, iterations, line-search trials of every rank
    assert res_all.shape[0] == total, (res_all.shape, total)
    iters_done = float(res_all[:, 4].sum())
    if rank == 0:
        kt = s.kernel_times()
        knots = sum(p["desc"].horizon for p in s.phases)
        dom = max((k for k in kt if k in alg_bytes), key=lambda k: kt[k][0], default=None)
        units_all = s.kernel_units()
        traffic_all = None if args.hkd else load_profile_json(pkg, "traffic.json")       # PMC bytes per knot of the whole-body workload (tools/profile_round.sh)
        counters = None if args.hkd else load_profile_json(pkg, "counters.json")          # SQ counters per knot (tools/pmc_lanes.sh)
        per_kernel = {}
        for kname in alg_bytes:
            if kname not in kt or kt[kname][1] == 0:
                continue
            ms, n = kt[kname]
            units = units_all.get(kname)
            per_launch_bytes = alg_bytes[kname] * (units / n if units else knots * B)
            ach = per_launch_bytes / (ms / n * 1e-3) / 1e9
            upl = (units / n if units else knots * B)
            tr = (traffic_all or {}).get("kernels", {}).get(kname)
            if tr is not None:      # PMC bytes per knot (measured at batch 512 on these kernel sources) x the knots of one launch of THIS run = HBM bytes per launch, like `achieved`
                tr = dict(tr, hbm_bytes_per_launch_raw=tr["hbm_bytes_per_knot_raw"] * upl, hbm_bytes_per_launch_fetch_x2=tr["hbm_bytes_per_knot_fetch_x2"] * upl,
                          over_algorithmic_raw=tr["hbm_bytes_per_knot_raw"] / alg_bytes[kname], over_algorithmic_fetch_x2=tr["hbm_bytes_per_knot_fetch_x2"] / alg_bytes[kname])
            per_kernel[kname] = {"avg_launch_ms": ms / n, "launches": n, "units_per_launch": upl, "alg_bytes_per_unit": alg_bytes[kname],
                                 "alg_bytes_per_launch": per_launch_bytes, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                                 "traffic": tr}
        roof = None
        if dom:
            d = per_kernel[dom]
            roof = {"bound": "hbm", "kernel": dom, "achieved": d["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac"], "traffic": d["traffic"],
                    "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"], "alg_bytes_per_launch": d["alg_bytes_per_launch"],
                    "note": "algorithmic bytes / measured launch time against the HBM roof (the yardstick of SURVEY 8d); `kernels` carries the same figure for every hot "
                            "kernel, `counters` what the SQ counters say binds them (only while measured on these kernel sources)",
                    "kernels": per_kernel, "counters": (counters or {}).get("kernels"),
                    "kernel_ms": {k: round(v[0], 3) for k, v in kt.items()}, "kernel_launches": {k: v[1] for k, v in kt.items()},
                    "kernel_units_knots": units_all,
                    "whole_iteration": {"alg_bytes_per_knot_iteration": alg_total, "achieved_GBs": alg_total * knots * iters_done / dt / 1e9 / world,
                                        "frac_of_peak": alg_total * knots * iters_done / dt / 1e9 / world / HBM_PEAK_GBS}}
        line = {"metric": METRIC, "value": iters_done / dt, "unit": "DDP iterations/s",
                "n_gpus": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
                "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32" if args.hkd == "f32" else "f64", "data": "synthetic",
                "config": {"workload": workload, "batch_per_gpu": B, "global_batch": total,
                           "parallelism": f"ensemble-sharded x{world}", "n_status_ok": int((res_all[:, 7] == 0).sum()),
                           "mean_ls_trials_per_iter": float(res_all[:, 5].sum() / max(iters_done, 1))},
                "roofline": roof}
        line["collectives"] = "rccl" if dist is not None else "none (single process)"
        line["per_rank"] = {"solve_ms": [round(float(v), 3) for v in per_rank[:, 0]], "iterations": [int(v) for v in per_rank[:, 1]], "ls_trials": [int(v) for v in per_rank[:, 2]],
                            "iter_imbalance": float(per_rank[:, 1].max() / max(per_rank[:, 1].mean(), 1.0)), "time_imbalance": float(per_rank[:, 0].max() / max(per_rank[:, 0].mean(), 1e-9))}
        if world == 1 and not args.no_cpu_baseline:
            so, line["cpu_baseline"] = cpu_baseline(pkg, phases_fn, x0_fn, opt_fn, args.steps)
            if so is not None:
                if args.hkd != "f32":          # (an fp32 handle is outside the fp64 tolerances: tests/test_gpu_parity.py holds it to its own)
                    line["parity_sample"] = parity_sample(s, so, len(s.phases))
                so.close()
        if world == 1 and not args.no_latency:
            line["host_boundary"] = host_boundary_probe(s, x0, opt, dt)
        if world == 1 and not args.no_latency and not args.strong and not args.hkd:
            line["latency"] = latency_probe(pkg, args.steps)
            line["latency"]["mpc_tick"] = mpc_tick_probe(pkg)
            line["latency"]["mpc_tick_cpp"] = mpc_tick_cpp_probe()
        print(json.dumps(line), flush=True)
    s.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
