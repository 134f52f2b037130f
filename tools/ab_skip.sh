#!/bin/bash
# What each launch of the training step costs the FRAME: tools/ab_skip.sh <tag> "<bench args>" <mask>...   (NRC_DIAG_SKIP, nrc_mlp.hip)
TAG=$1; ARGS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for M in "$@"; do
  NRC_DIAG_SKIP=$M timeout -k 10 300 python3 bench.py $ARGS --no-cpu-baseline > $OUT/bench_$M.json 2> $OUT/bench_$M.err || { tail -5 $OUT/bench_$M.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('skip %-3s %8.1f Msamples/s  frame %.4f ms  stages %s' % (sys.argv[2], d['value'], d['ms_per_frame'], {k: round(v, 3) for k, v in d['stage_ms'].items()}))" $OUT/bench_$M.json $M
done
