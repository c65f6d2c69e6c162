#!/bin/bash
# LDS pipe and matrix-core occupancy of the hot kernels: bank conflicts against LDS busy cycles, MFMA busy cycles.  Usage (GPU box, repo root): tools/pmc_lds.sh <tag>
# Counters in their own runs (kernel-trace only); the same command as tools/pmc_lanes.sh.
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r03x}
OUT=gpurun_out/$TAG/lds
mkdir -p $OUT
i=0
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o run --output-format csv -- python3 bench.py --steps 8 --warmup 0 --batch 1024 --no-cpu-baseline --no-latency > $OUT/p$i.log 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
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
    if d.get('SQ_LDS_IDX_ACTIVE'): d['bank_conflict_fraction_of_lds_active'] = d.get('SQ_LDS_BANK_CONFLICT', 0) / d['SQ_LDS_IDX_ACTIVE']
    if kn.get(k):
        d['knots'] = kn[k]
        for cn in ('SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU_MFMA_F64', 'SQ_WAIT_INST_LDS', 'SQ_WAVE_CYCLES', 'SQ_INSTS_LDS'):
            if cn in d: d[cn.lower() + '_per_knot'] = d[cn] / kn[k]
    res['kernels'][k] = d
json.dump(res, open('profiles/' + res['tag'] + '_lds_mfma_counters.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
