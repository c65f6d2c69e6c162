#!/usr/bin/env python3
"""Summarise the rocprofv3 outputs of tools/profile_round.sh into profiles/<tag>_* (tracked files the bench line cites)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1]
out = f"gpurun_out/{tag}"
knots_512 = 512 * 204                      # problems x slots per launch of the PMC runs


def find(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    return hits[0] if hits else None


st = find("stats/**/*kernel_stats.csv")
if st:
    shutil.copy(st, f"profiles/{tag}_kernel_stats_batch4096_steps5.csv")
if os.path.exists(f"{out}/bench.json"):
    shutil.copy(f"{out}/bench.json", f"profiles/{tag}_bench_batch4096_steps5.json")


def counter_by_kernel(d, counter):
    f = find(f"{d}/**/*counter_collection.csv")
    acc, n = defaultdict(float), defaultdict(int)
    if not f:
        return {}, {}
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") != counter:
            continue
        name = row["Kernel_Name"].split("(")[0]
        acc[name] += float(row["Counter_Value"]); n[name] += 1
    return acc, n


fetch, nf = counter_by_kernel("pmc_fetch", "FETCH_SIZE")
write, nw = counter_by_kernel("pmc_write", "WRITE_SIZE")
res = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes (kernel-trace only); command: bench.py --steps 2 --warmup 0 "
                "--batch 512 --no-cpu-baseline; 512 problems x 204 slots = 104448 knots per launch.  Counters are in KB.  MI355X_MICROARCH.md: on "
                "gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of a wide (16 B/lane) coalesced stream; these kernels read 8 B/lane "
                "(uncalibrated width), so both the raw value and the x2 upper bound are listed.", "kernels": {}}
traffic = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    fl = fetch.get(k, 0.0) * 1024 / max(nf.get(k, 1), 1); wl = write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)
    res["kernels"][k] = {"fetch_bytes_per_launch_raw": fl, "write_bytes_per_launch": wl, "fetch_bytes_per_knot_raw": fl / knots_512,
                         "fetch_bytes_per_knot_x2": 2 * fl / knots_512, "write_bytes_per_knot": wl / knots_512}
    traffic[k] = {"hbm_bytes_per_launch_batch4096_raw": 8 * (fl + wl), "hbm_bytes_per_launch_batch4096_fetch_x2": 8 * (2 * fl + wl),
                  "source": f"profiles/{tag}_pmc_batch512.json scaled x8 (traffic is linear in the batch)"}
if not res["kernels"]:
    sys.exit(f"no PMC data under {out}: nothing written")
json.dump(res, open(f"profiles/{tag}_pmc_batch512.json", "w"), indent=1)
json.dump(traffic, open("profiles/r01_traffic.json", "w"), indent=1)
print("wrote profiles/%s_*" % tag, list(res["kernels"]))
