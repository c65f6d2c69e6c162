#!/usr/bin/env python3
"""After `tools/profile_round.sh <tag>`, `tools/pmc_lanes.sh <tag>` and `tools/profile_hkd.sh <tag> f32` ran on the GPU box: build the tracked
summaries under profiles/ from gpurun_out/<tag> (the box's own profiles/ directory does not travel back) and print the headline figures."""
import json, subprocess, sys, glob, os
tag = sys.argv[1]
subprocess.check_call([sys.executable, "tools/pmc_summary.py", tag])
subprocess.check_call([sys.executable, "tools/pmc_summary.py", tag, "hkdf32"])
d = json.load(open(f"gpurun_out/{tag}/lanes/lanes.json"))
json.dump(d, open(f"profiles/{tag}_lane_occupancy.json", "w"), indent=1); json.dump(d, open("profiles/counters.json", "w"), indent=1)
sys.path.insert(0, ".")
import __graft_entry__ as ge
h = ge.load_package().kernel_source_hash()
print("hash", h, d["kernel_source_hash"], json.load(open("profiles/traffic.json"))["kernel_source_hash"])
for n in ("bench_steps20", "bench_steps10"):
    b = json.loads(open(f"gpurun_out/{tag}/{n}.json").read().strip().splitlines()[-1])
    print(n, round(b["value"]), round(b["ms_per_step"], 2), "cpu", round(b["cpu_baseline"]["value"], 1), "lat", round(b["latency"]["ms_per_ddp_iteration"], 2),
          "tick", b["latency"]["mpc_tick_cpp"]["total_ms_mean"], b["latency"]["mpc_tick_cpp"]["total_ms_max"], "parity", b["parity_sample"]["pass"], b["parity_sample"]["max_dK_inf_all_knots"])
    for k, v in b["roofline"]["kernels"].items():
        print("   ", k, round(v["avg_launch_ms"], 2), v["launches"], round(v["frac"], 3))
for n in ("hkdf32_bench_steps20", "hkdf32_bench_steps10"):
    b = json.loads(open(f"gpurun_out/{tag}/{n}.json").read().strip().splitlines()[-1])
    print(n, round(b["value"]), round(b["ms_per_step"], 2), {k: round(v["avg_launch_ms"], 2) for k, v in b["roofline"]["kernels"].items()})
for k, v in d["kernels"].items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if "per_knot" in a or "fraction" in a})
for k, v in json.load(open("profiles/traffic.json"))["kernels"].items():
    print(k, round(v["hbm_bytes_per_knot_raw"]), round(v["hbm_bytes_per_knot_fetch_x2"]))
# superseded tags of the same round (same files under another tag) are removed by hand: git rm profiles/<old>_*
