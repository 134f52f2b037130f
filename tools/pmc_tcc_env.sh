#!/bin/bash
# L2 counters of k_gen_rays in the bench frame under a list of environment settings: tools/pmc_tcc_env.sh <tag> "<bench args>" "VAR=1" ...  ("-" = none)
set -o pipefail
TAG=$1; ARGS=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
PY=$(readlink -f "$(command -v python3)")
i=0
for E in "$@"; do
  i=$((i+1)); [ "$E" = "-" ] && E=""
  for kv in $E; do export "$kv"; done
  B="$PY $REPO/bench.py $ARGS --steps 12 --warmup 6 --no-cpu-baseline"
  (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d "$OUT/e${i}_tcc" -o b -- $B) > "$OUT/e${i}_tcc.log" 2>&1 || exit 1
  (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --pmc FETCH_SIZE -d "$OUT/e${i}_fetch" -o b -- $B) > "$OUT/e${i}_fetch.log" 2>&1 || exit 1
  for kv in $E; do unset "${kv%%=*}"; done
  find "$OUT" -name "*.db" -delete 2>/dev/null
  echo "=== ${E:-(default)}"; python3 tools/pmc_summary.py "$OUT/e${i}_tcc" "$OUT/e${i}_fetch" | awk '/^k_gen_rays<false>/{p=1;print;next} /^k_/{p=0} p{print}'
done
