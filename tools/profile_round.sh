#!/bin/bash
# One GPU session's worth of measurements for profiles/rNN_* (run on the MI355X box from the repo root):
#   tools/profile_round.sh r02 [steps...]      steps: micro mlp bench pmc pin c5 hash issue c4 train timeline   (default: micro mlp bench pmc pin)
# Output goes to gpurun_out/<tag>/ ; then: python tools/profile_collect.py gpurun_out/<tag> rNN   (copies the summaries the documents quote into profiles/).
# rocprofv3 is always given the interpreter binary itself after `--` (no env / bash -c / shebang hop) and counters are
# collected in passes of their own (no trace domains beside --pmc).
set -o pipefail
TAG=${1:-r02}; shift
STEPS=${*:-micro mlp bench pmc pin}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
PY=$(readlink -f "$(command -v python3)")
has() { [[ " $STEPS " == *" $1 "* ]]; }

if has micro; then
  echo "== micro" && timeout -k 10 120 tools/_build/valu_rate > "$OUT/valu_rate.txt" 2>&1 || exit 1
fi
if has mlp; then
  echo "== mlp kernel trace (dense k_infer alone)"
  (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --kernel-trace --stats -d "$OUT/prof_mlp" -o mlp -- "$PY" "$REPO/tools/bench_mlp.py" 2073600 20) > "$OUT/prof_mlp.log" 2>&1 || exit 1
  python3 tools/trace_tail.py "$(find "$OUT/prof_mlp" -name "*kernel_trace.csv" | head -1)" k_infer 100 > "$OUT/mlp_trace_tail.txt" 2>&1
  export NRC_BENCH_MLP_WARM=5      # (the counter passes serialise every dispatch: no clock ramp to wait out, and 400 more dispatches to count)
  echo "== mlp MFMA counters"
  (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
      -d "$OUT/pmc_mlp_mfma" -o mlp -- "$PY" "$REPO/tools/bench_mlp.py" 2073600 4) > "$OUT/pmc_mlp_mfma.log" 2>&1 || exit 1
  (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_VALU_MFMA_COEXEC_CYCLES \
      -d "$OUT/pmc_mlp_sq" -o mlp -- "$PY" "$REPO/tools/bench_mlp.py" 2073600 4) > "$OUT/pmc_mlp_sq.log" 2>&1 || exit 1
  unset NRC_BENCH_MLP_WARM
fi
if has bench; then
  echo "== bench (plain)" && timeout -k 10 400 "$PY" bench.py --steps 100 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
  echo "== bench kernel trace"
  (cd /tmp && timeout -k 10 400 rocprofv3 -f csv --kernel-trace --stats -d "$OUT/prof_bench" -o bench -- "$PY" "$REPO/bench.py" --steps 25 --warmup 2 --no-cpu-baseline) > "$OUT/prof_bench.log" 2>&1 || exit 1
fi
if has pmc; then
  B="$PY $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
  echo "== pmc FETCH_SIZE";  (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o b -- $B) > "$OUT/pmc_fetch.log" 2>&1 || exit 1
  echo "== pmc WRITE_SIZE";  (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o b -- $B) > "$OUT/pmc_write.log" 2>&1 || exit 1
  echo "== pmc TCC";         (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_tcc" -o b -- $B) > "$OUT/pmc_tcc.log" 2>&1 || exit 1
  echo "== pmc SQ";          (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -o b -- $B) > "$OUT/pmc_sq.log" 2>&1 || exit 1
fi
if has c5; then
  echo "== configs[4] bench (plain)" && timeout -k 10 400 "$PY" bench.py --config c5 > "$OUT/c5_bench.json" 2> "$OUT/c5_bench.err" || exit 1
  echo "== configs[4] kernel trace"
  (cd /tmp && timeout -k 10 400 rocprofv3 -f csv --kernel-trace --stats -d "$OUT/prof_c5" -o c5 -- "$PY" "$REPO/bench.py" --config c5 --steps 12 --warmup 2 --no-cpu-baseline) > "$OUT/prof_c5.log" 2>&1 || exit 1
  echo "== dense 8x128 inference kernel trace"
  (cd /tmp && timeout -k 10 300 rocprofv3 -f csv --kernel-trace --stats -d "$OUT/prof_mlp128" -o mlp128 -- "$PY" "$REPO/tools/bench_mlp.py" 2073600 20 128 8) > "$OUT/prof_mlp128.log" 2>&1 || exit 1
  python3 tools/trace_tail.py "$(find "$OUT/prof_mlp128" -name "*kernel_trace.csv" | head -1)" k_infer_gen 100 > "$OUT/mlp128_trace_tail.txt" 2>&1
fi
if has hash; then
  echo "== HashGrid model (the reference's default posID 0) bench (plain)" && timeout -k 10 400 "$PY" bench.py --pos-id 0 --steps 60 --warmup 10 --no-cpu-baseline > "$OUT/hashgrid_bench.json" 2> "$OUT/hashgrid_bench.err" || exit 1
  echo "== HashGrid kernel trace"
  (cd /tmp && timeout -k 10 400 rocprofv3 -f csv --kernel-trace --stats -d "$OUT/prof_hash" -o hash -- "$PY" "$REPO/bench.py" --pos-id 0 --steps 12 --warmup 2 --no-cpu-baseline) > "$OUT/prof_hash.log" 2>&1 || exit 1
fi
if has issue; then
  echo "== issue-mix micro" && timeout -k 10 60 tools/_build/issue_mix > "$OUT/issue_mix.txt" 2>&1 || exit 1
fi
if has c4; then
  echo "== configs[3]: one rank's share of the 4K frame" && timeout -k 10 300 "$PY" tools/c4_rank_emulation.py > "$OUT/c4_rank_emulation.txt" 2>&1 || exit 1
  echo "== configs[3] on one GPU" && timeout -k 10 400 "$PY" bench.py --config c4 --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/c4_bench.json" 2> "$OUT/c4_bench.err" || exit 1
fi
if has train; then
  echo "== training step alone, kernel traces (6x64, 8x128)"
  bash tools/trace_train.sh "$TAG/train64" 64 6 > "$OUT/train_6x64.txt" 2>&1 || exit 1
  bash tools/trace_train.sh "$TAG/train128" 128 8 > "$OUT/train_8x128.txt" 2>&1 || exit 1
fi
if has timeline; then
  echo "== frame timelines (default preset, configs[4], HashGrid)"
  { echo "# default preset"; timeout -k 10 200 "$PY" tools/frame_timeline.py --frames 8; echo "# configs[4]"; timeout -k 10 200 "$PY" tools/frame_timeline.py --config c5 --frames 8;
    echo "# HashGrid model"; timeout -k 10 200 "$PY" tools/frame_timeline.py --pos-id 0 --frames 8; } > "$OUT/frame_timeline.txt" 2>&1 || exit 1
fi
if has pin; then
  echo "== exr pin calibration" && timeout -k 10 600 "$PY" tests/exr_pin_calibrate.py --backend gpu --frames 8192 --variant-frames 2048 --out "$OUT/exr_pin_gpu.json" > "$OUT/exr_pin_gpu.log" 2>&1 || exit 1
fi
# keep the merge-back small: the raw rocprofv3 databases are large, the CSVs are what gets read
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT"
echo "== done"
