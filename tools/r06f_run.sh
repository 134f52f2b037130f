set -x
O=gpurun_out/r06f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
timeout -k 10 300 python tools/tune_schedules.py --out $O/schedules.txt > $O/tune.log 2>&1; tail -12 $O/tune.log
for L in 64 32 16 8; do NRC_PREP_RAYS_PER_WAVE=$L timeout -k 10 200 python bench.py --steps 50 --warmup 10 --compat-fix 2 --no-quality --no-cpu-baseline > $O/bench_q2_L$L.json 2>/dev/null || exit 1; done
echo done
