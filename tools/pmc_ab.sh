#!/bin/bash
# SQ counters of the stand-alone k_gen_rays launch for several builds: tools/pmc_ab.sh <tag> <lib dir>...
# (counter passes of their own, the interpreter binary itself after `--`)
set -o pipefail
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp NRC_DEBUG=single_stream
PY=$(readlink -f "$(command -v python3)")
for L in "$@"; do
  export NRC_HPM_LIB=$REPO/nrc-hpm-renderer_amd/$L/libnrc_hpm.so
  B="$PY $REPO/bench.py --steps 6 --warmup 2 --train 0 --no-cpu-baseline"
  (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d "$OUT/${L}_a" -o b -- $B) > "$OUT/${L}_a.log" 2>&1 || exit 1
  (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA -d "$OUT/${L}_b" -o b -- $B) > "$OUT/${L}_b.log" 2>&1 || exit 1
  find "$OUT" -name "*.db" -delete 2>/dev/null
  echo "=== $L"; python3 tools/pmc_summary.py "$OUT/${L}_a" "$OUT/${L}_b" | awk '/^k_gen_rays/{p=1;print;next} /^k_/{p=0} p{print}'
done
