#!/bin/bash
# split-tile sweep on the GPU box: tools/ab_split.sh <tag> "<max:min_cycles> ..."   (NRC_SPLIT_TILES / NRC_SPLIT_MIN_CYCLES)
# per setting: the bench frame (four streams, training on) and the stand-alone kernel (single stream, no training)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
summ() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-22s %8.1f Msamples/s  frame %.4f ms  gen_rays %.4f ms  infer %.4f train %.4f' % (sys.argv[2], d['value'], d['ms_per_frame'], d['stage_ms']['gen_rays'], d['stage_ms']['infer'], d['stage_ms']['train']))" $1 "$2"; }
for S in $1; do
  export NRC_SPLIT_TILES=${S%%:*} NRC_SPLIT_MIN_CYCLES=${S##*:}
  timeout -k 10 200 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline $EXTRA_ARGS > $OUT/bench_$S.json 2> $OUT/bench_$S.err || { tail -5 $OUT/bench_$S.err; exit 1; }
  summ $OUT/bench_$S.json "$S frame"
  NRC_SINGLE_STREAM=1 timeout -k 10 200 python3 bench.py --steps 40 --warmup 10 --train 0 --no-cpu-baseline $EXTRA_ARGS > $OUT/alone_$S.json 2> $OUT/alone_$S.err || { tail -5 $OUT/alone_$S.err; exit 1; }
  summ $OUT/alone_$S.json "$S alone"
done
