#!/bin/bash
# Bench line of measurement builds (cafe-mpc_amd/variants/libhsddp_hip_<name>.so, made by `make -C cafe-mpc_amd/csrc variant NAME=.. EXTRA=..`).
# Usage (GPU box): tools/variant_bench.sh <tag> "<bench args>" <name|default> [...]
set -o pipefail
TAG=$1; ARGS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for V in "$@"; do
  if [ "$V" = default ]; then unset HSDDP_HIP_VARIANT; else export HSDDP_HIP_VARIANT=$V; fi
  timeout -k 10 420 python3 bench.py $ARGS --no-cpu-baseline --no-latency > $OUT/variant_$V.json 2> $OUT/variant_$V.err || { echo "$V failed"; tail -3 $OUT/variant_$V.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$OUT/variant_$V.json').read().strip().splitlines()[-1]); print('$V', round(d['value']), round(d['ms_per_step'],2), {k: round(v['avg_launch_ms'],2) for k,v in d['roofline']['kernels'].items()})" | tee -a $OUT/variants.txt
done
