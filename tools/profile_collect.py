#!/usr/bin/env python3
"""Copy what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/<round>_*: the summaries the documents quote.

    python tools/profile_collect.py gpurun_out/r02z r02

Only files that exist are copied (a round may have run a subset of the steps).  The raw per-dispatch counter CSVs of the
FETCH_SIZE / WRITE_SIZE passes are kept as they are (bench.py and tools/pmc_traffic.py read the derived JSON, the reader of
profiles/ can re-derive it from them); everything else is a rocprofv3 --stats table or a text summary."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def first(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    dst = os.path.join(ROOT, "profiles")
    copied = []

    def put(path, name):
        if path and os.path.exists(path):
            shutil.copyfile(path, os.path.join(dst, "%s_%s" % (rnd, name)))
            copied.append(name)

    def last_json_line(path, name):
        if not os.path.exists(path):
            return
        lines = [l for l in open(path) if l.startswith("{")]
        if lines:
            with open(os.path.join(dst, "%s_%s" % (rnd, name)), "w") as f:
                f.write(lines[-1])
            copied.append(name)

    def summary(dirs, name):
        dirs = [d for d in dirs if os.path.isdir(d)]
        if not dirs:
            return
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] + dirs, capture_output=True, text=True, check=True).stdout
        with open(os.path.join(dst, "%s_%s" % (rnd, name)), "w") as f:
            f.write(out)
        copied.append(name)

    last_json_line(os.path.join(src, "bench.json"), "bench.json")
    last_json_line(os.path.join(src, "c5_bench.json"), "c5_bench.json")
    put(first(src + "/prof_bench/**/*kernel_stats.csv"), "bench_kernel_stats.csv")
    put(first(src + "/prof_c5/**/*kernel_stats.csv"), "c5_kernel_stats.csv")
    put(first(src + "/prof_mlp/**/*kernel_stats.csv"), "mlp_kernel_stats.csv")
    put(first(src + "/prof_mlp128/**/*kernel_stats.csv"), "mlp128_kernel_stats.csv")
    put(os.path.join(src, "train_6x64.txt"), "train_6x64_kernel_trace.txt")
    put(os.path.join(src, "train_8x128.txt"), "train_8x128_kernel_trace.txt")
    put(os.path.join(src, "frame_timeline.txt"), "frame_timeline.txt")
    put(os.path.join(src, "mlp_trace_tail.txt"), "mlp_kernel_trace_tail.txt")
    put(os.path.join(src, "mlp128_trace_tail.txt"), "mlp128_kernel_trace_tail.txt")
    put(first(src + "/pmc_fetch/**/*counter_collection.csv"), "pmc_fetch_size.csv")
    put(first(src + "/pmc_write/**/*counter_collection.csv"), "pmc_write_size.csv")
    if all(os.path.isdir(os.path.join(src, d)) for d in ("pmc_fetch", "pmc_write", "pmc_tcc")):
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(dst, rnd + "_pmc_traffic.json"),
                        os.path.join(src, "pmc_fetch"), os.path.join(src, "pmc_write"), os.path.join(src, "pmc_tcc")], check=True)
        copied.append("pmc_traffic.json")
    summary([os.path.join(src, "pmc_sq")], "pmc_sq_counters.txt")
    summary([os.path.join(src, "pmc_mlp_mfma"), os.path.join(src, "pmc_mlp_sq")], "mlp_pmc_mfma.txt")
    put(os.path.join(src, "valu_rate.txt"), "micro_valu_rate.txt")
    put(os.path.join(src, "issue_mix.txt"), "micro_issue_mix.txt")
    last_json_line(os.path.join(src, "hashgrid_bench.json"), "hashgrid_bench.json")
    put(first(src + "/prof_hash/**/*kernel_stats.csv"), "hashgrid_kernel_stats.csv")
    last_json_line(os.path.join(src, "c4_bench.json"), "c4_bench.json")
    put(os.path.join(src, "c4_rank_emulation.txt"), "c4_rank_emulation.txt")
    # round 6 (tools/r06_collect.sh): the driver's command, the Q2-fixed preset, the convergence curves, the quality calibration, the gradient
    # attribution, the XCD balance
    last_json_line(os.path.join(src, "bench_driver_command.json"), "bench_driver_command.json")
    last_json_line(os.path.join(src, "bench_q2.json"), "bench_q2.json")
    put(first(src + "/prof_q2/**/*kernel_stats.csv"), "q2_kernel_stats.csv")
    for sid in (0, 4):
        put(os.path.join(src, "convergence_%d.txt" % sid), "convergence_%d.txt" % sid)
    put(os.path.join(src, "quality_calibration.txt"), "quality_calibration.txt")
    put(os.path.join(src, "grad_drift.txt"), "grad_drift.txt")
    put(os.path.join(src, "xcd_balance.txt"), "xcd_balance.txt")
    print("profiles/%s_*: %s" % (rnd, ", ".join(copied) if copied else "nothing found under " + src))


if __name__ == "__main__":
    main()
