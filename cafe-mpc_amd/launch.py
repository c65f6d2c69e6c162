"""One process per GPU: rank start-up, ensemble sharding and the single collective of the multi-GPU path.

The HS-DDP path shards by independent problems: rank r owns a contiguous block of the ensemble, solves it with its own handle and no
data-path collective; one all-gather of the 64-byte per-problem result struct (RCCL over xGMI; `nccl` IS RCCL on ROCm) puts every
problem's outcome on every rank (arg-min over contact-schedule candidates, MHPCLocomotion-style consumers).  bench.py and the CPU
rehearsal in tests/test_sharding_gloo.py (gloo) go through the same functions.
"""
import os
import socket
import subprocess
import sys

import numpy as np

RESULT_FIELDS = ("actual_cost", "dyn_feas", "max_tconstr", "max_pconstr", "n_iters", "n_ls_iters", "n_reg_iters", "status")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def count_gpus_without_hip():
    """GPUs this process may use, counted WITHOUT touching the HIP runtime (the parent of the ranks must not initialise the GPU): KFD topology
    nodes with SIMDs under /sys/class/kfd, cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None when the topology is not readable
    (no amdgpu driver on this host): the ranks then find out themselves and fail loudly."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(root):
        return None
    n = 0
    for node in sorted(os.listdir(root)):
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(root, node, "properties")).read().splitlines() if " " in line)
        except OSError:        # (a node of another container's cgroup: not ours)
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def maybe_spawn(n_ranks, script, argv, require_gpus=True):
    """Parent side of `bench.py --gpus N` started WITHOUT a launcher: start N ranks (torch.distributed.run, one per GPU) as child
    processes and relay their output.  Must run before this process makes any GPU call (a process that has initialised the GPU must
    not exec or fork GPU work).  Returns None when this process is itself a rank (or N == 1), else the children's exit code."""
    if "WORLD_SIZE" in os.environ:
        if int(os.environ["WORLD_SIZE"]) != n_ranks:
            print(f"[launch] --gpus {n_ranks} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks", file=sys.stderr)
            raise SystemExit(2)
        return None
    if n_ranks <= 1:
        return None
    if require_gpus:
        have = count_gpus_without_hip()
        if have is None:
            have = 0 if not os.path.exists("/dev/kfd") else None      # no amdgpu device node at all: certainly no GPU here
        if have is not None and have < n_ranks:
            print(f"[launch] --gpus {n_ranks} requested but this node exposes {have} GPU(s): refusing to run fewer ranks than asked", file=sys.stderr)
            return 3
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    # HSA_ENABLE_IPC_MODE_LEGACY: the pool's host driver only supports dmabuf IPC (RCCL otherwise fails with hipIpcGetMemHandle: invalid argument);
    # the image exports 0 already - an explicit setting of the caller is passed through untouched
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def init_ranks(backend):
    """(rank, world, local_rank, dist or None).  backend: "nccl" (RCCL, one GPU per rank) or "gloo" (CPU rehearsal)."""
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and os.environ.get("HSDDP_FORCE_PROCESS_GROUP", "0") != "1":
        return 0, 1, local, None      # (HSDDP_FORCE_PROCESS_GROUP=1: a one-rank group all the same - runs every RCCL call of the N > 1 path on a one-GPU box)
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend=backend)
    seen = dist.get_world_size()
    if seen != world:
        raise RuntimeError(f"process group reports {seen} ranks, launcher announced {world}")
    return rank, world, local, dist


def shard(total, world, rank):
    """Contiguous block [first, first + count) of `total` problems owned by `rank` (blocks differ by at most one problem)."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def result_rows(info):
    """hsddp_info_t arrays -> [problems, 8] fp64 rows (the 64-byte result struct of SURVEY 8e)."""
    return np.stack([np.asarray(info[k], dtype=np.float64) for k in RESULT_FIELDS], axis=1)


def gather_results(dist, rows, device):
    """All-gather of the per-problem result rows (blocks may differ in length by one: padded to the longest, trimmed after)."""
    import torch
    if dist is None:
        return rows
    world = dist.get_world_size()
    n = torch.tensor([rows.shape[0]], device=device, dtype=torch.int64)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    nmax = int(max(int(x.item()) for x in ns))
    pad = torch.zeros((nmax, rows.shape[1]), device=device, dtype=torch.float64)
    pad[:rows.shape[0]] = torch.as_tensor(rows, device=device)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return np.concatenate([o[:int(c.item())].cpu().numpy() for o, c in zip(out, ns)])


def max_over_ranks(dist, value, device):
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device):
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_scalars(dist, values, device):
    """[world, len(values)] fp64: a few per-rank scalars (solve time, iteration count) on every rank - explains a scaling result by rank."""
    row = np.asarray(values, dtype=np.float64)[None]
    if dist is None:
        return row
    import torch
    t = torch.as_tensor(row, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.concatenate([o.cpu().numpy() for o in out])
