set -x
O=gpurun_out/r06i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_integrator.py tests/test_gpu_frame_graph.py tests/test_gpu_baseline_configs.py -m gpu -q --maxfail=8 -k "fp16_exchange or pipelined or prep_train or ring or frame_graph or schedule or c4 or baseline" > $O/pytest.log 2>&1; tail -8 $O/pytest.log
for i in 1 2; do timeout -k 10 200 python bench.py --steps 50 --warmup 10 --compat-fix 2 --no-quality --no-cpu-baseline > $O/bench_q2_$i.json 2>/dev/null || exit 1; done
timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-quality --no-cpu-baseline > $O/bench_default.json 2>/dev/null
for f in bench_q2_1 bench_q2_2 bench_default; do python3 -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), round(d['ms_per_frame'],4), {k:round(v,3) for k,v in d['stage_ms'].items()}, d['schedule']['source'])"; done
echo done
