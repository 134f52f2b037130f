#!/bin/bash
# Round 6's ONE profile collection (one per round, at the end): tools/r06_collect.sh, run on the MI355X box from the repo root.
# Everything lands under gpurun_out/r06z/; python tools/profile_collect.py gpurun_out/r06z r06 + the copies at the end of this file's header
# bring the summaries into profiles/.
set -o pipefail
O=gpurun_out/r06z; mkdir -p $O
PY=$(readlink -f "$(command -v python3)")
REPO=$(pwd)
export TMPDIR=/tmp
echo "== driver command" && timeout -k 10 300 "$PY" bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err || exit 1
bash tools/profile_round.sh r06z micro mlp bench pmc c5 hash issue c4 train timeline || exit 1
echo "== Q2 fixed (the algorithm the CLI asks for)" && timeout -k 10 300 "$PY" bench.py --compat-fix 2 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_q2.json 2> $O/bench_q2.err || exit 1
(cd /tmp && timeout -k 10 400 rocprofv3 -f csv --kernel-trace --stats -d "$REPO/$O/prof_q2" -o q2 -- "$PY" "$REPO/bench.py" --compat-fix 2 --steps 25 --warmup 2 --no-cpu-baseline --no-quality) > $O/prof_q2.log 2>&1 || exit 1
echo "== fp16 exchange has no N > 1 here; convergence" && timeout -k 10 600 "$PY" tools/convergence.py --scenes 0,4 --frames 2048 --hash --out $O/convergence > $O/convergence.log 2>&1 || exit 1
echo "== quality calibration" && timeout -k 10 300 "$PY" tests/quality.py --calibrate --out $O/quality_calibration.txt > $O/quality.log 2>&1 || exit 1
echo "== gradient drift" && timeout -k 10 400 "$PY" tools/grad_drift.py --seeds 12 2>&1 | grep -v amdgpu.ids > $O/grad_drift.txt || exit 1
echo "== XCD balance (stamps-only build)"
export NRC_HPM_LIB=$REPO/nrc-hpm-renderer_amd/lib_stamps/libnrc_hpm.so
for a in "--config c2" "--config c2 --train 1" "--config c5" "--config c5 --train 1"; do timeout -k 10 200 "$PY" tools/xcd_balance.py $a 2>&1 | grep -v amdgpu.ids >> $O/xcd_balance.txt || exit 1; echo >> $O/xcd_balance.txt; done
unset NRC_HPM_LIB
find $O -name "*.db" -delete 2>/dev/null
du -sh $O
echo "== all done"
