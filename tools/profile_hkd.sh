#!/bin/bash
# BASELINE config 5 evidence, run on the GPU box from the repo root:  tools/profile_hkd.sh r03j [f32|f64]
#   1. bench lines of `bench.py --hkd <prec>` at --steps 10 and at the driver's shape --steps 20 --warmup 5 (HKD 24/24/0, N=200, 21 phases, batch 16 384)
#   2. rocprofv3 --kernel-trace --stats of the steps-20 command
#   3. two separate PMC passes (FETCH_SIZE / WRITE_SIZE, kernel-trace only) at batch 2048, 8 steps, each with the bench line of the same run
#   4. tools/pmc_summary.py <tag> hkd<prec> -> profiles/<tag>_hkd<prec>_*
set -e -o pipefail
TAG=${1:-r03x}; PREC=${2:-f32}
OUT=gpurun_out/$TAG; mkdir -p $OUT profiles
export TMPDIR=/tmp
python3 bench.py --hkd $PREC --steps 20 --warmup 5 --no-latency > $OUT/hkd${PREC}_bench_steps20.json 2> $OUT/hkd${PREC}_bench20.err
python3 bench.py --hkd $PREC --steps 10 --warmup 2 --no-latency > $OUT/hkd${PREC}_bench_steps10.json 2> $OUT/hkd${PREC}_bench10.err
rocprofv3 --kernel-trace --stats -d $OUT/hkd${PREC}_stats -o run --output-format csv -- python3 bench.py --hkd $PREC --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $OUT/hkd${PREC}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/hkd${PREC}_pmc_fetch -o run --output-format csv -- python3 bench.py --hkd $PREC --steps 8 --warmup 0 --batch 2048 --no-cpu-baseline --no-latency > $OUT/hkd${PREC}_pmc_fetch_bench.json 2> $OUT/hkd${PREC}_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/hkd${PREC}_pmc_write -o run --output-format csv -- python3 bench.py --hkd $PREC --steps 8 --warmup 0 --batch 2048 --no-cpu-baseline --no-latency > $OUT/hkd${PREC}_pmc_write_bench.json 2> $OUT/hkd${PREC}_pmc_write.err
python3 tools/pmc_summary.py $TAG hkd$PREC
