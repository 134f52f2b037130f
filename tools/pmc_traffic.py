#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes into profiles/rNN_pmc_traffic.json (per-kernel HBM-side bytes per launch).

    python tools/pmc_traffic.py OUT.json DIR_FETCH DIR_WRITE DIR_TCC

The three directories hold separate passes (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_HIT TCC_MISS) of the
same command.  FETCH_SIZE / WRITE_SIZE are reported in KB.  On gfx950 FETCH_SIZE tallies a wide coalesced streaming read at half
its bytes (MI355X_MICROARCH.md, HBM section): doubled for the streaming kernels (k_infer, k_composite); left as reported for the
integrator's 1-byte gathers, whose access width that correction was not calibrated for.
"""
import collections
import csv
import glob
import json
import sys

STREAMING = ("k_infer", "k_composite")


def load(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].replace("void ", "").replace("nrc::(anonymous namespace)::", "").split("(")[0].split("<")[0]
                if name.startswith("_ZN3nrc"):
                    name = "k_" + name.split("k_", 1)[1].split("E", 1)[0] if "k_" in name else name
                if name.startswith("k_"):
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = load(dirs)
    kernels = {}
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        rd = m.get("FETCH_SIZE", 0.0) * 1024.0
        if k in STREAMING:
            rd *= 2.0
        wr = m.get("WRITE_SIZE", 0.0) * 1024.0
        hit, miss = m.get("TCC_HIT_sum", m.get("TCC_HIT", 0.0)), m.get("TCC_MISS_sum", m.get("TCC_MISS", 0.0))
        kernels[k] = {"FETCH_SIZE_KB": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
                      "TCC_EA0_RDREQ_sum": m.get("TCC_EA0_RDREQ_sum", m.get("TCC_EA0_RDREQ")),
                      "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "bytes_read": rd, "bytes_written": wr,
                      "traffic_bytes": rd + wr, "l2_hit_rate": hit / (hit + miss) if hit + miss else None}
    note = ("rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | TCC_*), per-launch means over `python3 bench.py --steps 3 "
            "--warmup 1 --no-cpu-baseline` (1920x1080, 256^3 cloud, train on), MI355X. FETCH_SIZE/WRITE_SIZE in KB as reported; on "
            "gfx950 FETCH_SIZE reports 1/2 of a wide coalesced streaming read (MI355X_MICROARCH.md HBM section) -> bytes_read doubles "
            "it for the streaming kernels (k_infer, k_composite); for the integrator's 1-byte gathers it is left uncorrected.")
    json.dump({"note": note, "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print("%-18s read %10.0f KB  write %10.0f KB  L2 hit %s" % (k, v["bytes_read"] / 1024, v["bytes_written"] / 1024, v["l2_hit_rate"]))


if __name__ == "__main__":
    main()
