#!/bin/bash
# Instruction-fetch behaviour of the hot kernels (they are 70-250 KB of mostly straight-line code): I-cache requests / hits / misses, fetch latency.
# Usage (GPU box, repo root): tools/pmc_icache.sh <tag> [extra bench args].  Counters in their own runs (kernel-trace only).
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r03x}; shift
OUT=gpurun_out/$TAG/icache
mkdir -p $OUT
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o run --output-format csv -- python3 bench.py --steps 8 --warmup 0 --batch 1024 --no-cpu-baseline --no-latency "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']] += float(r['Counter_Value'])
units = {}
try:
    units = json.loads([l for l in open(out + '/p1.log') if l.startswith('{')][-1])['roofline']['kernel_units_knots']
except Exception as e:
    print('no bench line:', e)
sys.path.insert(0, '.')
import __graft_entry__ as ge
res = {'kernel_source_hash': ge.load_package().kernel_source_hash(), 'tag': sys.argv[2], 'command': 'bench.py --steps 8 --warmup 0 --batch 1024 --no-cpu-baseline --no-latency', 'kernels': {}}
kn = {'k_rollout_quad': units.get('k_rollout', 0) + units.get('k_ls_probe', 0), 'k_lq': units.get('k_lq', 0), 'k_sweep': units.get('k_sweep', 0)}
for k, c in acc.items():
    if not k.startswith(('k_rollout', 'k_lq', 'k_sweep')): continue
    d = dict(c)
    if d.get('SQC_ICACHE_REQ'): d['icache_miss_fraction'] = d.get('SQC_ICACHE_MISSES', 0) / d['SQC_ICACHE_REQ']
    if d.get('SQ_IFETCH'): d['ifetch_latency_cycles'] = d.get('SQ_IFETCH_LEVEL', 0) / d['SQ_IFETCH']
    if kn.get(k):
        d['knots'] = kn[k]
        for cn in ('SQC_ICACHE_REQ', 'SQC_ICACHE_MISSES', 'SQ_IFETCH', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY'):
            if cn in d: d[cn.lower() + '_per_knot'] = d[cn] / kn[k]
    res['kernels'][k] = d
json.dump(res, open(out + '/icache.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
