#!/bin/bash
# Collect the per-round evidence the bench line refers to (run on the GPU box from the repo root):
#   tools/profile_round.sh r01d
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/<tag>/stats
# 2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as MI355X_MICROARCH.md prescribes) at batch 512
# 3. tools/pmc_summary.py turns them into profiles/<tag>_*.{csv,json}
set -e -o pipefail
TAG=${1:-r01x}
OUT=gpurun_out/$TAG
mkdir -p $OUT profiles
export TMPDIR=/tmp
python3 bench.py --steps 5 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 bench.py --steps 5 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run --output-format csv -- python3 bench.py --steps 2 --warmup 0 --batch 512 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run --output-format csv -- python3 bench.py --steps 2 --warmup 0 --batch 512 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python3 tools/pmc_summary.py $TAG
