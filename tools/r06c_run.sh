set -x
O=gpurun_out/r06c; mkdir -p $O
export NRC_HPM_LIB=$PWD/nrc-hpm-renderer_amd/lib_stamps/libnrc_hpm.so
for w in -1 0 16; do timeout -k 10 120 python tools/xcd_balance.py --config c2 --window $w > $O/xcd_c2_w$w.txt 2>&1 || exit 1; done
timeout -k 10 120 python tools/xcd_balance.py --config c2 --train 1 > $O/xcd_c2_train.txt 2>&1 || exit 1
timeout -k 10 200 python tools/xcd_balance.py --config c5 > $O/xcd_c5.txt 2>&1 || exit 1
unset NRC_HPM_LIB
timeout -k 10 400 python tools/grad_drift.py --seeds 12 > $O/grad_drift.txt 2>&1 || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_quality.py -x -q > $O/pytest_quality.log 2>&1; tail -3 $O/pytest_quality.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --compat-fix 2 --no-cpu-baseline > $O/bench_q2.json 2> $O/bench_q2.err || exit 1
echo done
