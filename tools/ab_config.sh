#!/bin/bash
# A/B of library builds on one bench preset: tools/ab_config.sh <tag> "<bench args>" <lib dir name>...   (lib dirs under nrc-hpm-renderer_amd/)
TAG=$1; ARGS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for L in "$@"; do
  export NRC_HPM_LIB=$PWD/nrc-hpm-renderer_amd/$L/libnrc_hpm.so
  timeout -k 10 300 python3 bench.py $ARGS --no-cpu-baseline > $OUT/bench_$L.json 2> $OUT/bench_$L.err || { tail -5 $OUT/bench_$L.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-12s %8.1f Msamples/s  frame %.4f ms  stages %s' % (sys.argv[2], d['value'], d['ms_per_frame'], {k: round(v, 3) for k, v in d['stage_ms'].items()}))" $OUT/bench_$L.json $L
done
