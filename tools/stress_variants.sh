#!/bin/bash
# which co-resident kernel's raised wave priority makes k_gen_rays misbehave?  builds (no GPU): tools/stress_variants.sh build
# run (GPU box): tools/stress_variants.sh <processes per variant> [out dir]
set -u
cd "$(dirname "$0")/.."
BIN=tests/cpp/_build
VARIANTS="p1:-DNRC_DIAG_SETPRIO=1 p2:-DNRC_DIAG_SETPRIO=2 p4:-DNRC_DIAG_SETPRIO=4 p7nopair:-DNRC_DIAG_SETPRIO=7,-DNRC_PAIR_TAIL=0 p7:-DNRC_DIAG_SETPRIO=7"
if [ "${1:-}" = "build" ]; then
    for v in $VARIANTS; do
        name=${v%%:*}; flags=$(echo "${v#*:}" | tr ',' ' ')
        make -C nrc-hpm-renderer_amd/csrc ARCH=gfx950 OUT=../lib_$name "EXTRA=$flags" > /dev/null || exit 1
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O2 -Iinclude tests/cpp/stress_main.cpp -o $BIN/stress_main_$name \
            -Lnrc-hpm-renderer_amd/lib_$name -lnrc_hpm -pthread "-Wl,-rpath,\$ORIGIN/../../../nrc-hpm-renderer_amd/lib_$name" || exit 1
        echo built $name
    done
    exit 0
fi
N=${1:-120}; OUT=${2:-gpurun_out/stress7}; mkdir -p "$OUT"; : > "$OUT/summary.txt"
for v in $VARIANTS; do
    name=${v%%:*}; bad=0; : > "$OUT/$name.log"
    for i in $(seq 1 "$N"); do
        GPU_MAX_HW_QUEUES=8 timeout -k 5 150 $BIN/stress_main_$name tiles 1 1 >> "$OUT/$name.log" 2>&1; rc=$?
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name: time limit -- stopping" | tee -a "$OUT/summary.txt"; exit 1; fi
        [ $rc -ne 0 ] && bad=$((bad + 1))
    done
    echo "$name: $bad of $N" | tee -a "$OUT/summary.txt"
done
