#!/bin/bash
# bench preset under a list of environment settings: tools/ab_env_args.sh <tag> "<bench args>" "VAR=1 VAR2=x" "..."   ("-" = no extra environment)
TAG=$1; ARGS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for E in "$@"; do
  i=$((i+1))
  [ "$E" = "-" ] && E=""
  env $E timeout -k 10 200 python3 bench.py $ARGS --no-cpu-baseline > $OUT/env_$i.json 2> $OUT/env_$i.err || { tail -5 $OUT/env_$i.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-40s %7.1f Msamples/s frame %.4f host %.3f' % (sys.argv[2] or '(default)', d['value'], d['ms_per_frame'], d.get('host_enqueue_ms_per_frame', -1)), {k: round(v,3) for k,v in d['stage_ms'].items()})" $OUT/env_$i.json "$E"
done
