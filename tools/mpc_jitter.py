#!/usr/bin/env python3
"""Tick-time jitter of the C++ MPC loop (tests/cpp/mpc_loop.cpp on libhsddp_hip.so): run the harness R times with T ticks each and list every tick above a threshold
with its index, iteration count and the run's per-call maxima.  Usage (GPU box): tools/mpc_jitter.py [runs=12] [ticks=48] [threshold_ms=6]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import importlib
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 48
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
budget = sys.argv[4] if len(sys.argv) > 4 else "1"      # 0: solve without max_cputime (no time checkpoints)
pkg = ge.load_package(); builder = importlib.import_module(pkg.__name__ + ".builder")
tree = os.path.join(ROOT, "tests", "golden", "cafe_tree")
with tempfile.TemporaryDirectory() as td:
    exe = os.path.join(td, "mpc_loop")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cafe-mpc_amd", "host"), os.path.join(ROOT, "tests", "cpp", "mpc_loop.cpp"),
                           "-L", os.path.join(ROOT, "cafe-mpc_amd"), "-lhsddp_hip", "-Wl,-rpath," + os.path.join(ROOT, "cafe-mpc_amd"), "-o", exe])
    open(os.path.join(td, "opt.bin"), "wb").write(bytes(builder.load_ddp_setting(tree + "/MHPC/settings/ddp_setting.info")))
    worst = []
    for r in range(runs):
        out = json.loads(subprocess.check_output([exe, tree, "bound", os.path.join(td, "opt.bin"), str(ticks), budget], timeout=300))
        tm = out["total_ms"]
        slow = [(i + 1, round(t, 2), out["iters"][i]) for i, t in enumerate(tm) if i >= 4 and t > thr]
        worst.append(max(tm[4:]))
        print(f"run {r:2d}: mean {out['total_ms_mean']:.3f} max {out['total_ms_max']:.3f} solve max {out['solve_ms_max']:.3f} reconf max {out['reconfigure_ms_max']:.3f} allocs {out['device_allocations_in_warm_ticks']}"
              + (f"  slow ticks (tick, ms, iters): {slow}" if slow else ""), flush=True)
    print("worst tick over all runs:", round(max(worst), 3), "ms; runs with a tick above", thr, "ms:", sum(1 for w in worst if w > thr), "of", runs)
