#!/usr/bin/env python3
"""Summarise the rocprofv3 outputs of tools/profile_round.sh into profiles/<tag>_* (tracked files the bench line cites)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
SRC_HASH = ge.load_package().kernel_source_hash()
tag = sys.argv[1]
out = f"gpurun_out/{tag}"
HKD = sys.argv[2] if len(sys.argv) > 2 else None      # "hkdf32" / "hkdf64": the config-5 outputs of tools/profile_hkd.sh instead of the whole-body ones


def find(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    return hits[0] if hits else None


def last_json_line(path):
    if not os.path.exists(path):
        return None
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None


if HKD:
    st = find(f"{HKD}_stats/**/*kernel_stats.csv")
    if st:
        shutil.copy(st, f"profiles/{tag}_{HKD}_kernel_stats_batch16384_steps20.csv")
    for n in ("steps20", "steps10"):
        if os.path.exists(f"{out}/{HKD}_bench_{n}.json"):
            shutil.copy(f"{out}/{HKD}_bench_{n}.json", f"profiles/{tag}_{HKD}_bench_batch16384_{n}.json")
else:
    st = find("stats/**/*kernel_stats.csv")
    if st:
        shutil.copy(st, f"profiles/{tag}_kernel_stats_batch4096_steps20.csv")
    for n in ("steps20", "steps10"):
        if os.path.exists(f"{out}/bench_{n}.json"):
            shutil.copy(f"{out}/bench_{n}.json", f"profiles/{tag}_bench_batch4096_{n}.json")


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


pre = f"{HKD}_" if HKD else ""
fetch, nf = counter_by_kernel(pre + "pmc_fetch", "FETCH_SIZE")
write, nw = counter_by_kernel(pre + "pmc_write", "WRITE_SIZE")
bl = last_json_line(f"{out}/{pre}pmc_fetch_bench.json") or {}
units = (bl.get("roofline") or {}).get("kernel_units_knots", {})
# The rollout family = k_rollout_quad (whole-body running knots on lane quads) + k_rollout (terminal knots, single-rigid-body tail), ordinary
# rollouts and the probe launches of the batched line search alike: bytes of both kernels over the knots (x candidates) both families processed
for acc in (fetch, write):
    if "k_rollout_quad" in acc:
        acc["k_rollout"] = acc.get("k_rollout", 0.0) + acc.pop("k_rollout_quad")
for cnt in (nf, nw):
    if "k_rollout_quad" in cnt:
        cnt["k_rollout"] = cnt.get("k_rollout", 0) + cnt.pop("k_rollout_quad")
units_by_kernel = {"k_rollout": units.get("k_rollout", 0) + units.get("k_ls_probe", 0), "k_lq": units.get("k_lq", 0), "k_sweep": units.get("k_sweep", 0), "k_sweep32": units.get("k_sweep", 0),
                   "k_rollout_hkd": units.get("k_rollout", 0) + units.get("k_ls_probe", 0), "k_lq_hkd": units.get("k_lq", 0), "k_sweep_hkd": units.get("k_sweep", 0)}
res = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes (kernel-trace only); command: bench.py --steps 12 --warmup 0 "
                "--batch 512 --no-cpu-baseline --no-latency.  Counters are in KB, summed over every launch of a kernel in the run and divided by the knots "
                "(x line-search candidates) those launches processed (hsddp_get_kernel_units of the same run).  MI355X_MICROARCH.md: on gfx950 "
                "FETCH_SIZE reads exactly 1/2 of the bytes of a wide (16 B/lane) coalesced stream; these kernels mostly read 8 B/lane, so both the "
                "raw value and the x2 upper bound are listed.", "kernels": {}}
traffic = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    fb = fetch.get(k, 0.0) * 1024; wb = write.get(k, 0.0) * 1024
    u = units_by_kernel.get(k, 0)
    e = {"launches": nf.get(k, 0), "fetch_bytes_total_raw": fb, "write_bytes_total": wb, "knots_processed": u}
    if u:
        e.update({"fetch_bytes_per_knot_raw": fb / u, "fetch_bytes_per_knot_x2": 2 * fb / u, "write_bytes_per_knot": wb / u,
                  "hbm_bytes_per_knot_raw": (fb + wb) / u, "hbm_bytes_per_knot_fetch_x2": (2 * fb + wb) / u})
        traffic[k] = {"hbm_bytes_per_knot_raw": (fb + wb) / u, "hbm_bytes_per_knot_fetch_x2": (2 * fb + wb) / u, "source": f"profiles/{tag}_pmc_batch512.json"}
    res["kernels"][k] = e
if "k_rollout" in traffic:
    traffic["k_ls_probe"] = dict(traffic["k_rollout"], note="same kernel function as k_rollout (probe launches): run-wide average per knot x candidate")
if not res["kernels"]:
    sys.exit(f"no PMC data under {out}: nothing written")
res["kernel_source_hash"] = SRC_HASH
if HKD:
    res["_note"] = res["_note"].replace("--batch 512", f"--hkd {HKD[3:]} --batch 2048").replace("--steps 12", "--steps 8")
    json.dump(res, open(f"profiles/{tag}_{HKD}_pmc_batch2048.json", "w"), indent=1)
    print("wrote profiles/%s_%s_*" % (tag, HKD), list(res["kernels"])); sys.exit(0)
json.dump(res, open(f"profiles/{tag}_pmc_batch512.json", "w"), indent=1)
# the file bench.py cites (roofline.traffic), valid only for the kernel sources it was measured on
json.dump({"kernel_source_hash": SRC_HASH, "tag": tag, "kernels": traffic}, open("profiles/traffic.json", "w"), indent=1)
print("wrote profiles/%s_*" % tag, list(res["kernels"]))
