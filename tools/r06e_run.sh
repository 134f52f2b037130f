set -x
O=gpurun_out/r06e; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout -k 10 300 python tools/tune_schedules.py --out $O/schedules.txt > $O/tune.log 2>&1; tail -12 $O/tune.log
echo done
