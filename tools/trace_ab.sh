#!/bin/bash
# kernel-trace durations of the stand-alone camera stage for several builds: tools/trace_ab.sh <tag> <lib dir>...
set -o pipefail
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp NRC_DEBUG=single_stream
PY=$(readlink -f "$(command -v python3)")
for L in "$@"; do
  export NRC_HPM_LIB=$REPO/nrc-hpm-renderer_amd/$L/libnrc_hpm.so
  (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --kernel-trace --stats -d "$OUT/$L" -o t -- $PY $REPO/bench.py --steps 40 --warmup 5 --train 0 --no-cpu-baseline) > "$OUT/$L.log" 2>&1 || exit 1
  find "$OUT" -name "*.db" -delete 2>/dev/null
  echo "=== $L"; grep -h "k_gen_rays\|k_infer\|k_composite\|k_tile\|k_flight" $(find "$OUT/$L" -name "*kernel_stats.csv") | cut -c1-200
done
