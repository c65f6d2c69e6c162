set -o pipefail
OUT=gpurun_out/r03j; mkdir -p $OUT
tools/profile_hkd.sh r03j f32 > $OUT/profile.log 2>&1; echo "profile rc=$?"; tail -2 $OUT/profile.log
run() { python3 bench.py --hkd f32 --steps 10 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],2), {k: round(v['avg_launch_ms'],2) for k,v in d['roofline']['kernels'].items()})"; }
run default | tee -a $OUT/variants.txt
make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=2 -DSW_F32_WAVES=3" > $OUT/build_a.log 2>&1 && run "SW_F32_WAVES=3 (no spill in k_sweep32?)" | tee -a $OUT/variants.txt
make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=2 -DROLL_HKD_WPE=3 -DLQ_HKD_WPE=4" > $OUT/build_b.log 2>&1 && run "ROLL_HKD_WPE=3 LQ_HKD_WPE=4" | tee -a $OUT/variants.txt
make -C cafe-mpc_amd/csrc clean all EXTRA="-DROLL_WPE=2 -DROLL_HKD_WPE=2 -DLQ_HKD_WPE=3" > $OUT/build_c.log 2>&1 && run "ROLL_HKD_WPE=2 LQ_HKD_WPE=3" | tee -a $OUT/variants.txt
