#!/bin/bash
# One GPU-box session: tests, bench at the driver's command, in-kernel stamps (diagnostic builds).  Usage: tools/gpu_session.sh <tag> [steps...]
set -o pipefail
TAG=${1:-s}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for step in "$@"; do
  case $step in
    tests) timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt; tail -5 $OUT/tests.log ;;
    bench20) timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; echo "bench20 rc=$?" | tee -a $OUT/summary.txt; cut -c1-1500 $OUT/bench20.json ;;
    bench10) timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 > $OUT/bench10.json 2> $OUT/bench10.err; echo "bench10 rc=$?" | tee -a $OUT/summary.txt; cut -c1-1500 $OUT/bench10.json ;;
    bench5) timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench5.json 2> $OUT/bench5.err; echo "bench5 rc=$?" | tee -a $OUT/summary.txt; cut -c1-1500 $OUT/bench5.json ;;
    hkd32) timeout -k 10 420 python3 bench.py --hkd f32 --steps 5 --warmup 1 > $OUT/hkd32.json 2> $OUT/hkd32.err; echo "hkd32 rc=$?" | tee -a $OUT/summary.txt; cut -c1-1800 $OUT/hkd32.json; tail -3 $OUT/hkd32.err ;;
    hkd64) timeout -k 10 420 python3 bench.py --hkd f64 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/hkd64.json 2> $OUT/hkd64.err; echo "hkd64 rc=$?" | tee -a $OUT/summary.txt; cut -c1-1800 $OUT/hkd64.json; tail -3 $OUT/hkd64.err ;;
    wpe3) make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=3" > $OUT/build_wpe3.log 2>&1 && timeout -k 10 300 python3 tools/microbench.py > $OUT/wpe3.log 2>&1; echo "wpe3 rc=$?" | tee -a $OUT/summary.txt; cat $OUT/wpe3.log ;;
    sw2) make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=2 -DSW_WB_WAVES=2" > $OUT/build_sw2.log 2>&1 && timeout -k 10 300 python3 tools/microbench.py > $OUT/sw2.log 2>&1; echo "sw2 rc=$?" | tee -a $OUT/summary.txt; cat $OUT/sw2.log ;;
    micro) timeout -k 10 300 python3 tools/microbench.py > $OUT/micro.log 2>&1; echo "micro rc=$?" | tee -a $OUT/summary.txt; cat $OUT/micro.log ;;
    rollprof) make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=2 -DROLL_PROF" > $OUT/build_rollprof.log 2>&1 && ROLL_PROF=1 timeout -k 10 300 python3 tools/microbench.py > $OUT/rollprof.log 2>&1; echo "rollprof rc=$?" | tee -a $OUT/summary.txt; cat $OUT/rollprof.log ;;
    swprof) make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=2 -DSW_PROF" > $OUT/build_swprof.log 2>&1 && timeout -k 10 300 python3 tools/microbench.py > $OUT/swprof.log 2>&1; echo "swprof rc=$?" | tee -a $OUT/summary.txt; cat $OUT/swprof.log ;;
    lqprof) make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=2 -DLQ_PROF" > $OUT/build_lqprof.log 2>&1 && timeout -k 10 300 python3 tools/microbench.py > $OUT/lqprof.log 2>&1; echo "lqprof rc=$?" | tee -a $OUT/summary.txt; cat $OUT/lqprof.log ;;
    quad2) make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=2 -DQUAD_WPE=2" > $OUT/build_quad2.log 2>&1 && timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $OUT/quad2_bench20.json 2> $OUT/quad2.err; echo "quad2 rc=$?" | tee -a $OUT/summary.txt; python3 -c "import json,sys; d=json.loads(open('$OUT/quad2_bench20.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" ;;
    quadprof) make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=2 -DQUAD_PROF" > $OUT/build_quadprof.log 2>&1 && timeout -k 10 300 python3 tools/microbench.py > $OUT/quadprof.log 2>&1; echo "quadprof rc=$?" | tee -a $OUT/summary.txt; cat $OUT/quadprof.log ;;
    quadnext) make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=2 -DQUAD_NEXT_EARLY" > $OUT/build_quadnext.log 2>&1 && timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $OUT/quadnext_bench20.json 2> $OUT/quadnext.err; echo "quadnext rc=$?" | tee -a $OUT/summary.txt; python3 -c "import json,sys; d=json.loads(open('$OUT/quadnext_bench20.json').read().strip().splitlines()[-1]); print('QUAD_NEXT_EARLY', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" ;;
    benchq) timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $OUT/benchq.json 2> $OUT/benchq.err; echo "benchq rc=$?" | tee -a $OUT/summary.txt; python3 -c "import json,sys; d=json.loads(open('$OUT/benchq.json').read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" ;;
    maxilp) make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=2 -mllvm -amdgpu-sched-strategy=max-ilp" > $OUT/build_maxilp.log 2>&1 && timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $OUT/maxilp_bench20.json 2> $OUT/maxilp.err; echo "maxilp rc=$?" | tee -a $OUT/summary.txt; python3 -c "import json,sys; d=json.loads(open('$OUT/maxilp_bench20.json').read().strip().splitlines()[-1]); print('max-ilp', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" ;;
    maxocc) make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=2 -mllvm -amdgpu-sched-strategy=max-memory-clause" > $OUT/build_maxocc.log 2>&1 && timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $OUT/maxocc_bench20.json 2> $OUT/maxocc.err; echo "maxocc rc=$?" | tee -a $OUT/summary.txt; python3 -c "import json,sys; d=json.loads(open('$OUT/maxocc_bench20.json').read().strip().splitlines()[-1]); print('max-memory-clause', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" ;;
    rebuild) make -C cafe-mpc_amd/csrc clean all > $OUT/build.log 2>&1; echo "rebuild rc=$?" | tee -a $OUT/summary.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
