#!/bin/bash
# Lane occupancy of the per-knot kernels: SQ_THREAD_CYCLES_VALU (lane-cycles the VALU spent on ACTIVE lanes) against
# SQ_ACTIVE_INST_VALU (quad-cycles the VALU was busy) and SQ_INSTS_VALU.  Usage (GPU box, repo root): tools/pmc_lanes.sh <tag>
# Counters in their own run (kernel-trace only, no other trace domain); same command as tools/pmc_probe.sh so that the per-knot figures compare.
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r03x}
OUT=gpurun_out/$TAG/lanes
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
grep -o -i "SQ_THREAD_CYCLES_VALU\|SQ_ACTIVE_INST_VALU\|SQ_INSTS_VALU\b\|SQ_INST_CYCLES_VMEM\|SQ_VALU_THREAD\w*" $OUT/counters_list.txt | sort -u > $OUT/counters_found.txt
i=0
for C in "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o run --output-format csv -- python3 bench.py --steps 8 --warmup 0 --batch 1024 --no-cpu-baseline --no-latency > $OUT/p$i.log 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (k, r.get('Dispatch_Id'))
        if key not in seen and r['Counter_Name'] in ('SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES'): seen.add(key); calls[(k, r['Counter_Name'])] += 1
units = {}
try:
    line = [l for l in open(out + '/p1.log') if l.startswith('{')][-1]
    units = json.loads(line)['roofline']['kernel_units_knots']
except Exception as e:
    print('no bench line:', e)
sys.path.insert(0, '.')
import __graft_entry__ as ge
res = {'kernel_source_hash': ge.load_package().kernel_source_hash(), 'tag': sys.argv[2] if len(sys.argv) > 2 else '',
       'command': 'bench.py --steps 8 --warmup 0 --batch 1024 --no-cpu-baseline --no-latency', 'kernel_units_knots': units, 'kernels': {}}
# knots (x line-search candidates) behind each kernel function's counters: the rollout kernels serve ordinary rollouts and probe launches alike
kn = {'k_rollout_quad': units.get('k_rollout', 0) + units.get('k_ls_probe', 0), 'k_lq': units.get('k_lq', 0), 'k_sweep': units.get('k_sweep', 0)}
for k, c in acc.items():
    if not k.startswith(('k_rollout', 'k_lq', 'k_sweep', 'k_probe')): continue
    d = dict(c)
    if kn.get(k):
        d['knots'] = kn[k]
        for cn in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS'):
            if cn in d: d[cn.lower().replace('sq_insts_', '') + '_instructions_per_knot'] = d[cn] / kn[k]
        if 'SQ_WAVE_CYCLES' in d: d['wave_quad_cycles_per_knot'] = d['SQ_WAVE_CYCLES'] / kn[k]
    if d.get('SQ_ACTIVE_INST_VALU') and d.get('SQ_WAVE_CYCLES'): d['valu_active_fraction_of_wave_cycles'] = d['SQ_ACTIVE_INST_VALU'] / d['SQ_WAVE_CYCLES']
    if 'SQ_THREAD_CYCLES_VALU' in d and d.get('SQ_ACTIVE_INST_VALU'):
        # SQ_ACTIVE_INST_VALU counts quad-cycles (4 clocks); a wave64 instruction on 16 lanes per clock keeps 64 lane-slots per quad-cycle busy
        d['active_lane_fraction'] = d['SQ_THREAD_CYCLES_VALU'] / (64.0 * d['SQ_ACTIVE_INST_VALU'])
    if 'SQ_THREAD_CYCLES_VALU' in d and d.get('SQ_INSTS_VALU'):
        d['thread_cycles_per_valu_inst'] = d['SQ_THREAD_CYCLES_VALU'] / d['SQ_INSTS_VALU']
    res['kernels'][k] = d
json.dump(res, open(out + '/lanes.json', 'w'), indent=1)
json.dump(res, open('profiles/' + res['tag'] + '_lane_occupancy.json', 'w'), indent=1)
json.dump(res, open('profiles/counters.json', 'w'), indent=1)      # the file bench.py cites (roofline.counters), valid for these kernel sources only
print(json.dumps(res, indent=1))
PY
