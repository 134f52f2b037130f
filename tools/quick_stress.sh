#!/bin/bash
# quick_stress.sh <binary suffix> <perturb 0|1> <n>
cd "$(dirname "$0")/.."
bad=0
for i in $(seq 1 $3); do GPU_MAX_HW_QUEUES=8 timeout -k 5 150 tests/cpp/_build/stress_main_$1 tiles 1 $2 > /tmp/qs.log 2>&1 || bad=$((bad+1)); done
echo "stress_main_$1 perturb $2: $bad of $3"
