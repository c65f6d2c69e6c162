"""CPU test of the N>1 path (world_size 2, gloo): the ensemble is sharded into contiguous blocks of problems, each
rank solves its block with NO data-path collective, and one all-gather of the per-problem result struct puts the
whole ensemble's results on every rank (bench.py does the same with the HIP backend over RCCL).  The oracle library
stands in for the solver here only because this container has no GPU; what is under test is the sharding logic:
x0 stream offsets, result ordering, gathered == unsharded."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from conftest import pkg, ROOT

WORKER = textwrap.dedent("""
    import ctypes, os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    import __graft_entry__ as ge
    pkg = ge.load_package()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    lib = pkg._abi.bind(ctypes.CDLL(os.path.join(sys.argv[1], "oracle", "liboracle_hsddp.so")))
    B = 3
    phases = pkg.problems.wb_trot_problem(horizons=(4, 3, 3, 3))
    s = pkg.Solver(lib, phases, batch=B)
    for i, p in enumerate(phases):
        s.set_nominal(i, p["Xbar"], p["Ubar"])
    s.set_initial_condition(pkg.problems.wb_ensemble_x0(B, 99, first=rank * B))     # this rank's slice of the stream
    s.solve(pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=2, cost_thresh=0.0))
    info = s.info_arrays()
    res = torch.tensor(np.stack([info["actual_cost"], info["dyn_feas"], info["n_iters"].astype(float)], axis=1))
    out = [torch.empty_like(res) for _ in range(world)]
    dist.all_gather(out, res)
    if rank == 0:
        np.save(sys.argv[2], torch.cat(out).numpy())
    dist.barrier(); dist.destroy_process_group()
""")


def test_sharded_equals_unsharded(oracle_lib, tmp_path):
    w = tmp_path / "worker.py"; w.write_text(WORKER)
    out = tmp_path / "gathered.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29533", str(w), ROOT, str(out)], env=env, timeout=600)
    gathered = np.load(out)
    B = 6
    phases = pkg.problems.wb_trot_problem(horizons=(4, 3, 3, 3))
    s = pkg.Solver(oracle_lib, phases, batch=B)
    for i, p in enumerate(phases):
        s.set_nominal(i, p["Xbar"], p["Ubar"])
    s.set_initial_condition(pkg.problems.wb_ensemble_x0(B, 99))
    s.solve(pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=2, cost_thresh=0.0))
    info = s.info_arrays()
    ref = np.stack([info["actual_cost"], info["dyn_feas"], info["n_iters"].astype(float)], axis=1)
    assert gathered.shape == ref.shape
    assert np.array_equal(gathered, ref)          # same oracle, same inputs -> bit-identical, ordering included


def test_ensemble_stream_offsets():
    a = pkg.problems.wb_ensemble_x0(8, 5)
    b = np.vstack([pkg.problems.wb_ensemble_x0(4, 5, first=0), pkg.problems.wb_ensemble_x0(4, 5, first=4)])
    assert np.array_equal(a, b)
