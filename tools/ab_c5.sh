#!/bin/bash
# configs[4] (512^3 smoke, 8x128) frame + dense MLP figure per library build: tools/ab_c5.sh <tag> <lib dir name>...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for L in "$@"; do
  NRC_HPM_LIB=$PWD/nrc-hpm-renderer_amd/$L/libnrc_hpm.so timeout -k 10 300 python3 bench.py --config c5 --no-cpu-baseline > $OUT/c5_$L.json 2> $OUT/c5_$L.err || { tail -5 $OUT/c5_$L.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); m=d['roofline_mlp']
print('%-10s c5 %7.1f Msamples/s frame %.4f ms' % (sys.argv[2], d['value'], d['ms_per_frame']), {k: round(v,3) for k,v in d['stage_ms'].items()}, 'dense MLP %.4f ms = %.1f%%, on-frame %.4f ms' % (m['ms_per_launch'], 100*m['frac'], m['on_frame_queries']['ms_per_launch']))" $OUT/c5_$L.json $L
done
