"""Instruction mix of the loops of one kernel in a device assembly listing (hipcc --cuda-device-only -S):
    python3 tools/loop_isa.py <file.s> <kernel name substring> [min instructions]
For every natural loop (label .. last backward branch to it) prints its size and how many instructions are VALU, packed VALU, SALU,
branches, waits, LDS, vector memory; nested loops are listed with their depth."""
import re, subprocess, sys

def main():
    path, flt = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = open(path).read().splitlines()
    # kernel extents
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout
            if flt in name and start is None:
                start = i
        if start is not None and l.startswith(".Lfunc_end") and i > start:
            end = i
            break
    body = lines[start:end]
    labels = {}
    insts = []       # (index in body, text)
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        insts.append(t.split(";")[0].strip())
    loops = []
    for k, t in enumerate(insts):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
        if m:
            lab = m.group(1) or m.group(2)
            if lab in labels and labels[lab] <= k:
                loops.append((labels[lab], k, lab))
    # merge loops with the same header (keep the farthest back edge)
    best = {}
    for a, b, lab in loops:
        if lab not in best or b > best[lab][1]:
            best[lab] = (a, b)
    print("kernel: %d instructions" % len(insts))
    for lab, (a, b) in sorted(best.items(), key=lambda x: x[1][0]):
        seg = insts[a:b + 1]
        if len(seg) < min_n:
            continue
        depth = sum(1 for l2, (a2, b2) in best.items() if a2 <= a and b2 >= b) 
        c = dict(valu=0, pk=0, trans=0, salu=0, branch=0, wait=0, lds=0, vmem=0, lane=0, other=0)
        for t in seg:
            op = t.split()[0]
            if op.startswith("v_pk_"): c["pk"] += 1
            elif op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32") or op.startswith("v_permlane"): c["lane"] += 1
            elif re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op): c["trans"] += 1
            elif op.startswith("v_"): c["valu"] += 1
            elif op.startswith("s_cbranch") or op == "s_branch": c["branch"] += 1
            elif op.startswith("s_waitcnt") or op.startswith("s_nop"): c["wait"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"): c["vmem"] += 1
            else: c["other"] += 1
        print("%-12s depth %d  insts %5d [%5d..%5d]  " % (lab, depth, len(seg), a, b) + "  ".join("%s %d" % kv for kv in c.items() if kv[1]))

if __name__ == "__main__":
    main()
