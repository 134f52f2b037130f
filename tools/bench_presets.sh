#!/bin/bash
# the three bench presets of the round (default 6x64, configs[4] 8x128 on the 512^3 smoke, HashGrid) with the library as built:
#   tools/bench_presets.sh <tag> [steps]
TAG=$1; STEPS=${2:-200}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for P in "c2:" "c5:--config c5" "hashgrid:--pos-id 0"; do
  N=${P%%:*}; A=${P#*:}
  timeout -k 10 400 python3 bench.py $A --steps $STEPS --warmup 10 --no-cpu-baseline > $OUT/$N.json 2> $OUT/$N.err || { tail -5 $OUT/$N.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-9s %8.1f Msamples/s  frame %.4f ms  host %.3f ms  schedule %s  stages %s' % (sys.argv[2], d['value'], d['ms_per_frame'], d['host_enqueue_ms_per_frame'], d.get('schedule'), {k: round(v, 3) for k, v in d['stage_ms'].items()}))" $OUT/$N.json $N
done
