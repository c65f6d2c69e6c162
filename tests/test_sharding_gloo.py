"""CPU rehearsal of the N>1 path (world_size 2, gloo) through the SAME launcher code bench.py uses (cafe-mpc_amd/launch.py): the parent
starts the ranks before touching any GPU (maybe_spawn), every rank solves its contiguous block of the ensemble with NO data-path
collective, and one all-gather of the per-problem result struct puts the whole ensemble's results on every rank.  The oracle library
stands in for the solver here only because this container has no GPU; what is under test is the launch / sharding / gather logic:
rank start-up, x0 stream offsets, uneven blocks, result ordering, gathered == unsharded."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from conftest import pkg, ROOT

WORKER = textwrap.dedent("""
    import ctypes, os, sys
    import numpy as np
    sys.path.insert(0, sys.argv[1])
    import __graft_entry__ as ge
    pkg = ge.load_package()
    launch = pkg.launch
    TOTAL = int(sys.argv[3])
    rc = launch.maybe_spawn(2, os.path.abspath(__file__), sys.argv[1:], require_gpus=False)      # the parent: start the two ranks
    if rc is not None:
        sys.exit(rc)
    rank, world, local, dist = launch.init_ranks("gloo")
    assert world == 2 and dist.get_world_size() == 2
    lib = pkg._abi.bind(ctypes.CDLL(os.path.join(sys.argv[1], "oracle", "liboracle_hsddp.so")))
    first, B = launch.shard(TOTAL, world, rank)
    if len(sys.argv) > 4 and sys.argv[4] == "strong":      # bench.py --strong in shape: the barrel-roll ensemble, a FIXED total split over the ranks
        phases, xinit = pkg.problems.barrel_roll_problem(switching_times=(0.0, 0.03, 0.06, 0.10, 0.13, 0.16, 0.19))
        x0 = pkg.problems.barrel_roll_ensemble_x0(B, 20241220 + 4, xinit, first=first)
        opt = pkg.problems.br_ddp_setting(max_AL_iter=1, max_DDP_iter=2, cost_thresh=0.0)
    else:
        phases = pkg.problems.wb_trot_problem(horizons=(4, 3, 3, 3))
        x0 = pkg.problems.wb_ensemble_x0(B, 99, first=first)                     # this rank's slice of the stream
        opt = pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=2, cost_thresh=0.0)
    s = pkg.Solver(lib, phases, batch=B)
    for i, p in enumerate(phases):
        s.set_nominal(i, p["Xbar"], p["Ubar"])
    s.set_initial_condition(x0)
    s.solve(opt)
    rows = launch.result_rows(s.info_arrays())
    gathered = launch.gather_results(dist, rows, "cpu")
    slowest = launch.max_over_ranks(dist, float(rank), "cpu")
    per_rank = launch.gather_scalars(dist, [float(rank) + 0.5, float(rows[:, 4].sum())], "cpu")      # what bench.py reports as per_rank
    if rank == 0:
        assert slowest == 1.0
        assert per_rank.shape == (2, 2) and per_rank[0, 0] == 0.5 and per_rank[1, 0] == 1.5 and per_rank[:, 1].sum() == gathered[:, 4].sum()
        np.save(sys.argv[2], gathered)
    dist.barrier(); dist.destroy_process_group()
""")


def test_sharded_equals_unsharded(oracle_lib, tmp_path):
    w = tmp_path / "worker.py"; w.write_text(WORKER)
    out = tmp_path / "gathered.npy"
    TOTAL = 7                                               # uneven split: 4 + 3
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    subprocess.check_call([sys.executable, str(w), ROOT, str(out), str(TOTAL)], env=env, timeout=600)     # no launcher: the worker spawns its ranks itself
    gathered = np.load(out)
    phases = pkg.problems.wb_trot_problem(horizons=(4, 3, 3, 3))
    s = pkg.Solver(oracle_lib, phases, batch=TOTAL)
    for i, p in enumerate(phases):
        s.set_nominal(i, p["Xbar"], p["Ubar"])
    s.set_initial_condition(pkg.problems.wb_ensemble_x0(TOTAL, 99))
    s.solve(pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=2, cost_thresh=0.0))
    ref = pkg.launch.result_rows(s.info_arrays())
    assert gathered.shape == ref.shape
    assert np.array_equal(gathered, ref)          # same oracle, same inputs -> bit-identical, ordering included


def test_strong_scaling_split_of_the_barrel_roll_ensemble(oracle_lib, tmp_path):
    """`bench.py --strong` in shape (config 4's ensemble, a fixed total split over the ranks, uneven blocks 3 + 2): every rank draws ITS slice of
    the joint-pose stream (barrel_roll_ensemble_x0(first=...)), gathered == unsharded bit for bit, the per-rank scalars arrive in rank order."""
    w = tmp_path / "worker.py"; w.write_text(WORKER)
    out = tmp_path / "gathered.npy"
    TOTAL = 5
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    subprocess.check_call([sys.executable, str(w), ROOT, str(out), str(TOTAL), "strong"], env=env, timeout=900)
    gathered = np.load(out)
    phases, xinit = pkg.problems.barrel_roll_problem(switching_times=(0.0, 0.03, 0.06, 0.10, 0.13, 0.16, 0.19))
    s = pkg.Solver(oracle_lib, phases, batch=TOTAL)
    for i, p in enumerate(phases):
        s.set_nominal(i, p["Xbar"], p["Ubar"])
    s.set_initial_condition(pkg.problems.barrel_roll_ensemble_x0(TOTAL, 20241220 + 4, xinit))
    s.solve(pkg.problems.br_ddp_setting(max_AL_iter=1, max_DDP_iter=2, cost_thresh=0.0))
    assert np.array_equal(gathered, pkg.launch.result_rows(s.info_arrays()))


def test_gpu_count_without_the_hip_runtime(tmp_path, monkeypatch):
    """The parent of the ranks counts GPUs from the KFD topology (no HIP call): here there is no amdgpu driver -> None; a fake topology is parsed
    (nodes with SIMDs are GPUs, the CPU node is not) and cut down by HIP_VISIBLE_DEVICES."""
    assert pkg.launch.count_gpus_without_hip() is None or isinstance(pkg.launch.count_gpus_without_hip(), int)
    import builtins
    root = tmp_path / "nodes"
    for i, simd in enumerate((0, 1024, 1024, 1024)):
        (root / str(i)).mkdir(parents=True); (root / str(i) / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\n")
    real_isdir, real_listdir, real_open = os.path.isdir, os.listdir, builtins.open
    kfd = "/sys/class/kfd/kfd/topology/nodes"
    monkeypatch.setattr(os.path, "isdir", lambda p: True if p == kfd else real_isdir(p))
    monkeypatch.setattr(os, "listdir", lambda p: real_listdir(str(root)) if p == kfd else real_listdir(p))
    monkeypatch.setattr(builtins, "open", lambda p, *a, **k: real_open(str(p).replace(kfd, str(root)), *a, **k))
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False); monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    assert pkg.launch.count_gpus_without_hip() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert pkg.launch.count_gpus_without_hip() == 2


def test_ensemble_stream_offsets():
    a = pkg.problems.wb_ensemble_x0(8, 5)
    b = np.vstack([pkg.problems.wb_ensemble_x0(4, 5, first=0), pkg.problems.wb_ensemble_x0(4, 5, first=4)])
    assert np.array_equal(a, b)
    xinit = pkg.problems.barrel_roll_states()[0]
    c = pkg.problems.barrel_roll_ensemble_x0(6, 9, xinit)
    d = np.vstack([pkg.problems.barrel_roll_ensemble_x0(2, 9, xinit), pkg.problems.barrel_roll_ensemble_x0(4, 9, xinit, first=2)])
    assert np.array_equal(c, d)
    assert [pkg.launch.shard(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` on a node without two GPUs (this container has none) must fail loudly with a non-zero exit code, never
    run fewer ranks than asked; a launcher that started a different number of ranks than --gpus is refused as well."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr
    env2 = dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
