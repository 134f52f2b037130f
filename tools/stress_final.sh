#!/bin/bash
# The determinism harness on the shipped build: tools/stress_final.sh [N]   (N fresh processes per arrangement; run on the GPU box)
#   raised   the product's default: every kernel of the library at s_setprio 3, the perturbing kernel raised too
#   lowered  NRC_DEBUG=wave_priority_raise=0 (what nrc_cache_comm_init selects for world > 1): the library at the default priority UNDER a raised
#            perturbing kernel -- the arrangement that failed 1-3 % of the time before the integrator was compiled without the SLP vectoriser
cd "$(dirname "$0")/.."
N=${1:-40}; OUT=gpurun_out/stress_final; mkdir -p $OUT; : > $OUT/summary.txt
EXE=$(python3 -c "import __graft_entry__ as e; print(e.build_cpp_stress())") || exit 1
#   q2       round 6: the long-train-path graph (quirk Q2 fixed: traces of four frames on streams of their own) against the single-stream order
for spec in "raised 1 both" "lowered 0 both" "q2 1 pipeq2"; do
  set -- $spec; name=$1; val=$2; what=$3; bad=0; : > $OUT/$name.log
  for i in $(seq 1 $N); do
    NRC_DEBUG=wave_priority_raise=$val GPU_MAX_HW_QUEUES=8 timeout -k 5 150 $EXE $what 1 1 >> $OUT/$name.log 2>&1; rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name: time limit -- stopping" | tee -a $OUT/summary.txt; exit 1; fi
    [ $rc -ne 0 ] && bad=$((bad + 1))
    [ $((i % 10)) -eq 0 ] && echo "$name: $bad of $i so far"
  done
  echo "$name: $bad of $N fresh processes with mismatches" | tee -a $OUT/summary.txt
done
