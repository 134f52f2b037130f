#!/bin/bash
# kernel-trace A/B of library builds on the bench frame: tools/ab_trace.sh <tag> <lib dir name>...  (per build: rocprofv3 --kernel-trace --stats
# of bench.py, then the average / minimum duration of the frame's kernels)
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
PY=$(readlink -f "$(command -v python3)")
REPO=$PWD
export TMPDIR=/tmp
for L in "$@"; do
  export NRC_HPM_LIB=$REPO/nrc-hpm-renderer_amd/$L/libnrc_hpm.so
  (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --kernel-trace --stats -d $OUT/prof_$L -o b -- $PY $REPO/bench.py --steps 25 --warmup 2 --no-cpu-baseline ${BENCH_ARGS}) > $OUT/prof_$L.log 2>&1 || { tail -5 $OUT/prof_$L.log; exit 1; }
  find $OUT/prof_$L -name "*.db" -delete
  python3 - $OUT/prof_$L $L $OUT/prof_$L.log <<'PYEOF'
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
line = [l for l in open(sys.argv[3]) if l.startswith("{")]
d = json.loads(line[-1]) if line else None
print(sys.argv[2], ("frame %.4f ms  %.1f Msamples/s" % (d["ms_per_frame"], d["value"])) if d else "")
for row in csv.DictReader(open(f)):
    n = row["Name"]
    for key in ("k_infer", "k_gen_rays", "k_train_gen", "k_train_fwd_bwd", "k_encode", "k_wgrad", "k_prep_train", "k_composite", "k_reduce", "k_train_scan"):
        if key in n:
            print("   %-16s calls %4s avg %8.1f us  min %8.1f  %5s%%" % (key, row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, row["Percentage"]))
            break
PYEOF
done
