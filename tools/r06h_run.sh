set -x
O=gpurun_out/r06h; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo done
