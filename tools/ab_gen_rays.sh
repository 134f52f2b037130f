#!/bin/bash
# A/B of k_gen_rays builds on the GPU box: tools/ab_gen_rays.sh <tag> <lib dir name>...   (lib dirs under nrc-hpm-renderer_amd/)
# per build: the bench frame (four streams, training on) and the stand-alone kernel (single stream, no training)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
summ() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-28s %8.1f Msamples/s  frame %.4f ms  gen_rays %.4f ms  infer %.4f  fetch/px %.2f' % (sys.argv[2], d['value'], d['ms_per_frame'], d['stage_ms']['gen_rays'], d['stage_ms']['infer'], d['roofline_integrator']['fetches_per_pixel']))" $1 "$2"; }
for L in "$@"; do
  export NRC_HPM_LIB=$PWD/nrc-hpm-renderer_amd/$L/libnrc_hpm.so
  timeout -k 10 200 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline > $OUT/bench_$L.json 2> $OUT/bench_$L.err || { tail -5 $OUT/bench_$L.err; exit 1; }
  summ $OUT/bench_$L.json "$L frame"
  NRC_DEBUG=single_stream timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --train 0 --no-cpu-baseline > $OUT/alone_$L.json 2> $OUT/alone_$L.err || { tail -5 $OUT/alone_$L.err; exit 1; }
  summ $OUT/alone_$L.json "$L alone"
done
