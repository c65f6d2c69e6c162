# address-translation and memory-latency counters of the per-knot kernels (diagnostic): tools/pmc_tlb.sh, run on the GPU box
set -e -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmctlb
mkdir -p $OUT
i=0
for C in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o run --output-format csv -- python3 bench.py --steps 6 --warmup 0 --batch 1024 --no-cpu-baseline --no-latency > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmctlb/p*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][:12]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        for k in acc:
            if k.startswith(('k_sweep','k_lq','k_rollout')): print(k, dict(acc[k]))
PY
