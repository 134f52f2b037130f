#!/bin/bash
# Kernel trace of one bench preset: tools/trace_bench.sh <tag> "<bench args>" [lib dir name]
TAG=$1; ARGS=$2; L=${3:-lib}
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp NRC_HPM_LIB=$REPO/nrc-hpm-renderer_amd/$L/libnrc_hpm.so
PY=$(readlink -f $(which python3))
(cd /tmp && timeout -k 10 400 rocprofv3 -f csv --kernel-trace --stats -d "$OUT/prof" -o t -- "$PY" "$REPO/bench.py" $ARGS --steps 25 --warmup 2 --no-cpu-baseline) > "$OUT/prof.log" 2>&1 || { tail -5 $OUT/prof.log; exit 1; }
python3 - "$OUT" <<PYEOF
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) >= 40:
            print("%-60s calls %5s  avg %7.1f us  min %7.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PYEOF
