#!/bin/bash
# stand-alone k_gen_rays (single stream, no training) for several builds / volumes: tools/ab_alone.sh <tag> <volume> <lib dir>...
TAG=$1; VOL=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for L in "$@"; do
  NRC_HPM_LIB=$PWD/nrc-hpm-renderer_amd/$L/libnrc_hpm.so NRC_DEBUG=single_stream timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --train 0 --no-cpu-baseline --volume $VOL > $OUT/alone_${L}_$VOL.json 2> $OUT/alone_${L}_$VOL.err || { tail -3 $OUT/alone_${L}_$VOL.err; continue; }
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline_integrator']
print('%-10s vol %4s: gen_rays %.4f ms  fetch/px %.2f executed %.2f' % (sys.argv[2], sys.argv[3], d['stage_ms']['gen_rays'], r['fetches_per_pixel'], r.get('fetches_executed_per_pixel', -1)))" $OUT/alone_${L}_$VOL.json $L $VOL
done
