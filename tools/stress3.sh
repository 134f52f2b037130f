#!/bin/bash
# verification stage: the comparison and the build that reproduced (tiles-vs-whole, -DNRC_DIAG_LOWPRIO=8: the camera kernels at wave priority 0 beside raised neighbours), many fresh processes
set -u
cd "$(dirname "$0")/.."
N=${1:-150}; OUT=${2:-gpurun_out/stress3}; mkdir -p "$OUT"; : > "$OUT/summary.txt"
for spec in "prio_quiet stress_main_prio 0" "prio_spin stress_main_prio 1"; do
    set -- $spec; name=$1; bin=$2; pert=$3; bad=0; : > "$OUT/$name.log"
    for i in $(seq 1 "$N"); do
        GPU_MAX_HW_QUEUES=8 timeout -k 5 150 tests/cpp/_build/$bin tiles 1 "$pert" >> "$OUT/$name.log" 2>&1; rc=$?
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name: time limit -- stopping" | tee -a "$OUT/summary.txt"; exit 1; fi
        [ $rc -ne 0 ] && bad=$((bad + 1))
        [ $((i % 50)) -eq 0 ] && echo "$name: $bad of $i so far"
    done
    echo "$name: $bad of $N" | tee -a "$OUT/summary.txt"
done
