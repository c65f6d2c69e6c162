#!/usr/bin/env python3
"""VERDICT r02 item 2, measured instead of argued: do the latency-bound Riccati sweep and the VALU-bound knot kernels overlap when two
half-batches run on two streams?  Two handles of batch B/2 solved from two host threads (ctypes drops the GIL for the call, each handle owns
its stream and runs its own line-search control flow) against one handle of batch B, same problems, fixed-work mode.

    python tools/costream_probe.py [--batch 4096] [--steps 20] [--stagger-ms 0,5,10,20]
"""
import argparse, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--stagger-ms", default="0,8,16"); a = ap.parse_args()
pkg = ge.load_package()
phases = pkg.problems.wb_trot_problem()
x0 = pkg.problems.wb_ensemble_x0(a.batch, 20241220 + 3)
opt = pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=a.steps, cost_thresh=0.0)


def make(xs):
    s = pkg.MultiPhaseDDP(phases, batch=xs.shape[0]); s.set_initial_condition(xs); return s


w = make(x0[:64]); w.solve(pkg.mhpc_ddp_setting(max_AL_iter=1, max_DDP_iter=3, cost_thresh=0.0)); w.close()      # module load, first launches
s = make(x0); t0 = time.perf_counter(); s.solve(opt); t1 = time.perf_counter() - t0
ref = s.info_arrays(); kref = s.field(0, "K", 0, 8); s.close()
print(f"one stream, batch {a.batch}: {t1 * 1e3:8.1f} ms  {ref['n_iters'].sum() / t1:9.0f} it/s")
h = a.batch // 2
for stg in [float(v) for v in a.stagger_ms.split(",")]:
    sa, sb = make(x0[:h]), make(x0[h:])
    def run(sv, delay):
        if delay: time.sleep(delay * 1e-3)
        sv.solve(opt)
    th = [threading.Thread(target=run, args=(sa, 0.0)), threading.Thread(target=run, args=(sb, stg))]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t2 = time.perf_counter() - t0
    ia, ib = sa.info_arrays(), sb.info_arrays()
    same = np.array_equal(np.concatenate([ia["n_ls_iters"], ib["n_ls_iters"]]), ref["n_ls_iters"]) and np.array_equal(sa.field(0, "K", 0, 8), kref)
    print(f"two streams, 2 x {h}, second started {stg:4.1f} ms later: {t2 * 1e3:8.1f} ms  {(ia['n_iters'].sum() + ib['n_iters'].sum()) / t2:9.0f} it/s   ({t1 / t2:5.3f} x one stream; results identical: {same})")
    sa.close(); sb.close()
