#!/bin/bash
# Kernel trace of the stand-alone training chain of one model: tools/trace_train.sh <tag> <width> <depth> [pos_id] [lib dir name]
TAG=$1; W=$2; D=$3; P=${4:-3}; L=${5:-lib}
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp NRC_HPM_LIB=$REPO/nrc-hpm-renderer_amd/$L/libnrc_hpm.so
PY=$(readlink -f $(which python3))
timeout -k 10 200 python3 tools/train_step_rate.py 16384 300 $W $D $P || exit 1
(cd /tmp && timeout -k 10 300 rocprofv3 -f csv --kernel-trace --stats -d "$OUT/prof" -o t -- "$PY" "$REPO/tools/train_step_rate.py" 16384 200 $W $D $P) > "$OUT/prof.log" 2>&1 || { tail -5 $OUT/prof.log; exit 1; }
python3 - "$OUT" <<PYEOF
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) >= 100:
            print("%-60s calls %5s  avg %7.1f us  min %7.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PYEOF
