set -e -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmcx
mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F64" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o run --output-format csv -- python3 bench.py --steps 8 --warmup 0 --batch 1024 --no-cpu-baseline --no-latency > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmcx/p*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][:12]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        for k in acc:
            if k.startswith(('k_sweep','k_lq','k_rollout')): print(k, dict(acc[k]))
PY
