#!/bin/bash
# NOTE (round 6): -DNRC_DIAG_LASTDIR / -DNRC_DIAG_BISECT left the product source; this tool builds them from the tree of commit aa01da1 (round 5): git worktree add /tmp/r05 aa01da1
# the failing arrangement (camera kernels at wave priority 0 beside raised neighbours) with the last-direction probes in
# (-DNRC_DIAG_LASTDIR -DNRC_DIAG_LOWPRIO=8): for every affected pixel the log says whether the direction the kernel stored at its
# end is still the one new_ray_dir had produced (written to memory right behind the call) -- i.e. whether the value changed in
# the register file or was computed differently.   tools/stress_lastdir.sh build | tools/stress_lastdir.sh <processes> [out dir]
# (the product is compiled with -fno-slp-vectorize since the cause was found: this diagnostic build switches the vectoriser back ON)
set -u
cd "$(dirname "$0")/.."
BIN=tests/cpp/_build
if [ "${1:-}" = "build" ]; then
    make -C nrc-hpm-renderer_amd/csrc ARCH=gfx950 OUT=../lib_ld "EXTRA=-fslp-vectorize -DNRC_DIAG_LASTDIR -DNRC_DIAG_LOWPRIO=8" > /dev/null || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O2 -Iinclude tests/cpp/stress_main.cpp -o $BIN/stress_main_ld \
        -Lnrc-hpm-renderer_amd/lib_ld -lnrc_hpm -pthread "-Wl,-rpath,\$ORIGIN/../../../nrc-hpm-renderer_amd/lib_ld" || exit 1
    exit 0
fi
N=${1:-300}; OUT=${2:-gpurun_out/stress_lastdir}; mkdir -p "$OUT"; : > "$OUT/lastdir.log"; bad=0
for i in $(seq 1 "$N"); do
    GPU_MAX_HW_QUEUES=8 timeout -k 5 150 $BIN/stress_main_ld tiles 1 1 > "$OUT/run.log" 2>&1; rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "time limit -- stopping" | tee -a "$OUT/summary.txt"; exit 1; fi
    if [ $rc -ne 0 ]; then bad=$((bad + 1)); echo "=== process $i" >> "$OUT/lastdir.log"; grep -E "MISMATCH|LASTDIR" "$OUT/run.log" >> "$OUT/lastdir.log"; fi
    [ $((i % 50)) -eq 0 ] && echo "$bad of $i so far"
done
echo "lastdir probe build: $bad of $N" | tee "$OUT/summary.txt"
grep -c "moved_in_registers tiles 1\|whole 1" "$OUT/lastdir.log" | sed 's/^/lines with a value that moved in the register file: /' | tee -a "$OUT/summary.txt"
grep -c "LASTDIR" "$OUT/lastdir.log" | sed 's/^/LASTDIR lines: /' | tee -a "$OUT/summary.txt"
