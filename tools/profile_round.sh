#!/bin/bash
# Collect the per-round evidence the bench line refers to (run on the GPU box from the repo root):  tools/profile_round.sh r02a
# 1. the bench line at the DRIVER's command (--steps 20 --warmup 5) and at the default (--steps 10)
# 2. rocprofv3 --kernel-trace --stats of the driver's command                                  -> gpurun_out/<tag>/stats
# 3. two separate PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as MI355X_MICROARCH.md prescribes) at batch 512, 12 steps (so that
#    the batched line-search launches are in the trace), each with the bench line of the same run (kernel unit counts)
# 4. tools/pmc_summary.py turns them into profiles/<tag>_*.{csv,json} and profiles/r02_traffic.json
set -e -o pipefail
TAG=${1:-r02x}
OUT=gpurun_out/$TAG
mkdir -p $OUT profiles
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench20.err
python3 bench.py > $OUT/bench_steps10.json 2> $OUT/bench10.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run --output-format csv -- python3 bench.py --steps 12 --warmup 0 --batch 512 --no-cpu-baseline --no-latency > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run --output-format csv -- python3 bench.py --steps 12 --warmup 0 --batch 512 --no-cpu-baseline --no-latency > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
python3 tools/pmc_summary.py $TAG
