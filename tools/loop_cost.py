#!/usr/bin/env python3
"""Static issue-cost model of the camera kernels' loops (round 4): compiles nrc_integrator.hip for the device, finds the loops of a
kernel in the disassembly and prices each one with the per-instruction SIMD issue costs tools/issue_mix.hip measured at five waves per
SIMD (s_memtime clocks: plain VALU 1.64, packed FP32 2.52, v_cndmask on an SGPR mask 2.66, v_cmp 1.7, any scalar instruction 2.0 --
scalar instructions are NOT free --, s_nop 0.43).  The model reproduces the measured trip time of the tracking loops within 10 %.

    python3 tools/loop_cost.py [-DNAME=VALUE ...] [--kernel k_gen_raysILb0] [--min 100]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COST = {"VALU": 1.64, "packed": 2.52, "cndmask": 2.66, "cmp": 1.7, "SALU": 2.0, "branch": 2.0, "nop": 0.43, "LDS": 1.64, "VMEM": 1.64, "wait": 0.43, "trans": 6.5}


def classify(op):
    if op.startswith("v_pk_"): return "packed"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_cmp"): return "cmp"
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_log", "v_exp", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_"): return "VALU"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("buffer_", "global_", "flat_")): return "VMEM"
    return "VALU"


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    kernel = "k_gen_raysILb0"
    min_len = 100
    args = sys.argv[1:]
    for i, a in enumerate(args):
        if a == "--kernel": kernel = args[i + 1]
        if a == "--min": min_len = int(args[i + 1])
    src = os.path.join(ROOT, "nrc-hpm-renderer_amd", "csrc", "nrc_integrator.hip")
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "integ.co")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
                               "--cuda-device-only", "--no-gpu-bundle-output", "-c", "-o", co, src] + defs, stderr=subprocess.DEVNULL)
        dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", notes):
        if kernel in m.group(1):
            print("%s: scratch %s B, %s SGPRs (%s spilled), %s VGPRs" % (m.group(1)[:40], m.group(2), m.group(3), m.group(4), m.group(5)))
    for block in re.split(r"\n(?=[0-9a-f]{16} <[^>]+>:)", dis):
        m = re.match(r"[0-9a-f]{16} <([^>]+)>:", block)
        if not m or kernel not in m.group(1):
            continue
        lines = block.splitlines()[1:]
        addr = []
        for l in lines:
            mm = re.search(r"//\s*([0-9A-Fa-f]{12}):", l)
            addr.append(int(mm.group(1), 16) if mm else None)
        print("%s: %d instructions, %d bytes" % (m.group(1)[:40], len(lines), (addr[-1] or 0) - (addr[0] or 0)))
        seen = set()
        for i, l in enumerate(lines):
            mm = re.match(r"\s*(s_cbranch_\w+|s_branch)\s+(\d+)", l)
            if not (mm and addr[i] is not None):
                continue
            off = int(mm.group(2))
            if off < 32768:
                continue
            tgt = addr[i] + 4 + (off - 65536) * 4
            if tgt not in addr:
                continue
            j = addr.index(tgt)
            body = [re.sub(r"\s*//.*", "", x).strip().split()[0] for x in lines[j:i + 1]]
            if len(body) < min_len or len(body) > 400 or (j, i) in seen:
                continue
            seen.add((j, i))
            ops = collections.Counter(classify(o) for o in body)
            cost = sum(COST[k] * v for k, v in ops.items())
            loads = sum(1 for x in lines[j:i + 1] if "buffer_load_ubyte" in x)
            print("  loop %5d..%5d  %3d instr  %d byte loads  cost %6.1f   %s" % (j, i, len(body), loads, cost, " ".join("%s %d" % kv for kv in sorted(ops.items()))))


if __name__ == "__main__":
    main()
