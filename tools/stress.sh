#!/bin/bash
# Stress job for the non-determinism hunt (tests/cpp/stress_main.cpp): fresh processes x environment variants, every process
# renders tiles-vs-whole (configs[3] shape, training off) and pipelined-vs-single-stream (training on) with a perturbing kernel
# co-resident.  Usage (on the GPU box, from the repo root):  tools/stress.sh <processes per variant> [out dir]
# Build first (no GPU needed):  tools/stress.sh build
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
BIN=tests/cpp/_build
if [ "${1:-}" = "build" ]; then
    mkdir -p $BIN
    make -C nrc-hpm-renderer_amd/csrc ARCH=gfx950 || exit 1
    make -C nrc-hpm-renderer_amd/csrc ARCH=gfx950 OUT=../lib_prio EXTRA=-DNRC_DIAG_LOWPRIO=8 || exit 1
    for v in "" _prio; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O2 -Iinclude tests/cpp/stress_main.cpp -o $BIN/stress_main$v \
            -Lnrc-hpm-renderer_amd/lib$v -lnrc_hpm -pthread "-Wl,-rpath,\$ORIGIN/../../../nrc-hpm-renderer_amd/lib$v" || exit 1
    done
    exit 0
fi
N=${1:-20}
OUT=${2:-gpurun_out/stress}
mkdir -p "$OUT"
run_variant() {      # name, binary, perturb, env...
    local name=$1 bin=$2 perturb=$3; shift 3
    local log="$OUT/$name.log" bad=0
    : > "$log"
    for i in $(seq 1 "$N"); do
        env "$@" timeout -k 5 150 $BIN/$bin both 1 "$perturb" >> "$log" 2>&1
        rc=$?
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name: process $i hit its time limit -- stopping the job" | tee -a "$OUT/summary.txt"; exit 1; fi
        if [ $rc -ne 0 ]; then bad=$((bad + 1)); fi
    done
    echo "$name: $bad of $N processes reported a mismatch or a guard violation" | tee -a "$OUT/summary.txt"
}
: > "$OUT/summary.txt"
run_variant first_processes    stress_main      0 GPU_MAX_HW_QUEUES=8
run_variant spin_wave_per_cu   stress_main      1 GPU_MAX_HW_QUEUES=8
run_variant spin_lds_workgroup stress_main      2 GPU_MAX_HW_QUEUES=8
run_variant poison_alloc       stress_main      1 GPU_MAX_HW_QUEUES=8 NRC_DEBUG=poison_alloc
run_variant guard_alloc        stress_main      1 GPU_MAX_HW_QUEUES=8 NRC_DEBUG=guard_alloc,poison_alloc
run_variant hw_queues_2        stress_main      1 GPU_MAX_HW_QUEUES=2
run_variant hw_queues_4        stress_main      1 GPU_MAX_HW_QUEUES=4
run_variant setprio_build      stress_main_prio 1 GPU_MAX_HW_QUEUES=8
run_variant setprio_build_quiet stress_main_prio 0 GPU_MAX_HW_QUEUES=8
cat "$OUT/summary.txt"
