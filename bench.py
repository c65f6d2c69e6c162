#!/usr/bin/env python3
"""bench.py — DDP iterations/sec of the batched HS-DDP solve (BASELINE.json metric) on N MI355X GPUs.

Workload (config.workload): BASELINE.json configs[2] — Mini-Cheetah whole-body, N=200 knots as 4 contact phases
(1111 -> 0110 -> 1001 -> 0110, dt=0.01), batch of 4096 initial states PER GPU (weak scaling), synthetic inputs
(SURVEY 8d), fixed-work mode: max_AL_iter=1, cost_thresh=0 so that every problem runs exactly `--steps` DDP
iterations.  A "step" = one DDP iteration (cost, LQ approximation, regularised Riccati sweep, linear rollout,
line search, nominal update: MultiPhaseDDP.cpp:277-387) of the whole batch.  Inputs are resident in HBM before
the timed region.  One process per GPU; the only collective is the RCCL all-gather of the per-problem results.
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
}
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(pkg, phases_fn, seed, seconds_hint=20.0):
    """Oracle (CPU restatement of the reference) on this box's host cores — reported baseline, not the target."""
    path = os.path.join(ROOT, "oracle", "liboracle_hsddp.so")
    if not os.path.exists(path):
        return None
    lib = pkg._abi.bind(ctypes.CDLL(path))
    lib.oracle_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    # the GPU box exposes every host CPU in the affinity mask but a one-GPU job owns a 16-core share: size the pool to that
    cores = min(len(os.sched_getaffinity(0)), 16)
    K = 2
    phases = phases_fn()
    opt = pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=K, cost_thresh=0.0)
    out = {}
    for kind, nprob, lq_thr, pb_thr in (("all_cores", max(cores, 1), 1, cores), ("reference_shaped_4thr", 2, 4, 1)):
        s = pkg.Solver(lib, phases, batch=nprob)
        for i, p in enumerate(phases):
            s.set_nominal(i, p["Xbar"], p["Ubar"])
        s.set_initial_condition(pkg.problems.wb_ensemble_x0(nprob, seed))
        lib.oracle_set_threads(s.h, lq_thr, pb_thr)
        t0 = time.time(); s.solve(opt); dt = time.time() - t0
        out[kind] = nprob * K / dt
        s.close()
    return {"value": out["all_cores"], "unit": "DDP iterations/s", "cores": cores, "kind": "port",
            "sample": f"{max(cores,1)} problems x {K} iterations of the same WB N=200 workload, one problem per core "
                      f"({cores} threads); reference-shaped run (1 problem at a time, 4 OpenMP threads over knots in "
                      f"LQ_approximation only): {out['reference_shaped_4thr']:.3f} it/s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="problems per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    pkg = ge.load_package()
    seed = 20241220 + 3
    B = args.batch
    x0 = pkg.problems.wb_ensemble_x0(B, seed, first=rank * B)

    def run(iters):
        phases = pkg.problems.wb_trot_problem()
        s = pkg.MultiPhaseDDP(phases, batch=B, device=local)
        s.set_initial_condition(x0)
        opt = pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=iters, cost_thresh=0.0)
        return s, opt

    if args.warmup > 0:
        s, opt = run(args.warmup); s.solve(opt); s.close()
    s, opt = run(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    s.solve(opt)                       # synchronises its stream before returning
    info = s.info_arrays()
    res = torch.tensor(np.stack([info["actual_cost"], info["dyn_feas"], info["max_tconstr"], info["max_pconstr"],
                                 info["n_iters"].astype(np.float64), info["n_ls_iters"].astype(np.float64),
                                 info["n_reg_iters"].astype(np.float64), info["status"].astype(np.float64)], axis=1),
                       device=f"cuda:{local}")
    if dist is not None:               # C1: all-gather of the per-problem result struct (64 B/problem) over RCCL/xGMI
        gathered = [torch.empty_like(res) for _ in range(world)]
        dist.all_gather(gathered, res)
        res_all = torch.cat(gathered)
    else:
        res_all = res
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=f"cuda:{local}", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
    res_all = res_all.cpu().numpy()
    iters_done = float(res_all[:, 4].sum())
    if rank == 0:
        kt = s.kernel_times()
        knots = sum(p["desc"].horizon for p in s.phases)
        dom = max((k for k in kt if k in ALG_BYTES), key=lambda k: kt[k][0], default=None)
        roof = None
        if dom:
            ms, n = kt[dom]
            per_launch_bytes = ALG_BYTES[dom] * knots * B        # every launch covers the whole batch
            achieved = per_launch_bytes / (ms / n * 1e-3) / 1e9
            traffic = None
            tf = os.path.join(ROOT, "profiles", "r01_traffic.json")
            if os.path.exists(tf):
                traffic = json.load(open(tf)).get(dom)
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "avg_launch_ms": ms / n, "launches": n,
                    "kernel_ms": {k: round(v[0], 3) for k, v in kt.items()}}
        line = {"metric": "DDP iterations/sec, Mini-Cheetah WB N=200, batch=4096; 1/2/4/8 GPU", "value": iters_done / dt, "unit": "DDP iterations/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "WB 36/12/12, N=200 = 4 contact phases x 50 knots (1111,0110,1001,0110), dt=0.01, ensemble of initial states, "
                                       "fixed-work mode (max_AL_iter=1, cost_thresh=0)", "batch_per_gpu": B, "global_batch": B * world,
                           "parallelism": f"ensemble-sharded x{world}", "n_status_ok": int((res_all[:, 7] == 0).sum()),
                           "mean_ls_trials_per_iter": float(res_all[:, 5].sum() / max(iters_done, 1))},
                "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pkg, pkg.problems.wb_trot_problem, seed)
        print(json.dumps(line), flush=True)
    s.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
