#!/bin/bash
REPO=$PWD; OUT=$REPO/gpurun_out/r05_hiptrace; mkdir -p $OUT
export TMPDIR=/tmp
PY=$(readlink -f $(which python3))
(cd /tmp && timeout -k 10 400 rocprofv3 -f csv --hip-runtime-trace --stats -d "$OUT/prof" -o t -- "$PY" "$REPO/bench.py" --steps 40 --warmup 5 --no-cpu-baseline) > "$OUT/prof.log" 2>&1 || { tail -5 $OUT/prof.log; exit 1; }
ls $OUT/prof
python3 - "$OUT" <<PYEOF
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/prof/**/*hip_api_stats.csv", recursive=True) + glob.glob(sys.argv[1] + "/prof/**/*hip*stats.csv", recursive=True):
    print(f)
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-40s calls %7s  avg %8.2f us  total %8.2f ms" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
    break
PYEOF
