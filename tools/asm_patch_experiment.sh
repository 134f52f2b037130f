#!/bin/bash
# NOTE (round 6): -DNRC_DIAG_LASTDIR / -DNRC_DIAG_BISECT left the product source; this tool builds them from the tree of commit aa01da1 (round 5): git worktree add /tmp/r05 aa01da1
# Which instruction of the SLP-made packed-FP32 cluster in new_ray_dir's second rotation is it?  
# Builds the every-process-failing probe library (tools/bisect_build.sh 0) from its DEVICE ASSEMBLY with one edit applied to the cluster
# of k_gen_rays<false>, through the steps hipcc itself takes (device -S, assemble, lld, clang-offload-bundler, host compile with
# -fcuda-include-gpubinary), plus the harness:   tools/asm_patch_experiment.sh <name> <python expression editing the list `L` of cluster lines>
# e.g.  tools/asm_patch_experiment.sh nopafter 'sum(([l, "\ts_nop 4"] if "v_pk_" in l else [l] for l in L), [])'
set -e
cd "$(dirname "$0")/.."
NAME=$1; EDIT=$2
W=/tmp/asmexp_$NAME; mkdir -p $W
LL=/opt/rocm/lib/llvm/bin
SRC=$PWD/nrc-hpm-renderer_amd/csrc/nrc_integrator.hip
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -fslp-vectorize -DNRC_DIAG_LASTDIR -DNRC_DIAG_LOWPRIO=8 -DNRC_GEN_WAVES_PER_SIMD=4 -DNRC_DIAG_BISECT=0 -ffp-contract=off"
[ -f nrc-hpm-renderer_amd/lib_bs0/nrc_api.o ] || tools/bisect_build.sh 0 > /dev/null
/opt/rocm/bin/hipcc $FL --cuda-device-only -S $SRC -o $W/int.s 2>/dev/null
python3 - "$W/int.s" "$EDIT" <<'PY'
import sys, re
path, edit = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3nrc12_GLOBAL__N_110k_gen_raysILb0E"))
# the cluster: from the first v_pk_mul_f32 behind the kernel's last-but-one v_div_fixup ... 1.0 up to the last packed op in front of the final normalisation
idx = [i for i in range(start, len(lines)) if "v_pk_mov_b32" in lines[i]]
mov = idx[0]
a = mov
while "v_div_fixup_f32" not in lines[a]: a -= 1
a += 1
b = mov
while not ("v_mul_f32_e32" in lines[b] and lines[b + 1].strip().startswith("v_fmac_f32") and lines[b + 2].strip().startswith("v_fmac_f32") and "v_pk_" not in lines[b + 3]): b += 1
L = lines[a:b]
print("cluster: %d lines, %d packed" % (len(L), sum("v_pk_" in l for l in L)))
def scal(l, t0="v120", t1="v121"):
    """a packed FP32 instruction as two scalar ones (results in temporaries first: destination and sources may overlap)"""
    m = re.match(r"\s*v_pk_(fma|mul)_f32 v\[(\d+):\d+\], (.*)$", l)
    if not m: return [l]
    op, d, rest = m.group(1), int(m.group(2)), m.group(3)
    mods = {k: [int(x) for x in v.split(",")] for k, v in re.findall(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]", rest)}
    srcs = [int(x) for x in re.findall(r"v\[(\d+):\d+\]", rest)]
    n = len(srcs)
    sel, selhi = mods.get("op_sel", [0] * n), mods.get("op_sel_hi", [1] * n)
    nlo, nhi = mods.get("neg_lo", [0] * n), mods.get("neg_hi", [0] * n)
    def operand(i, hi):
        r = srcs[i] + ((selhi[i] if hi else sel[i]))
        return ("-" if (nhi[i] if hi else nlo[i]) else "") + "v%d" % r
    out = []
    for hi, t in ((0, t0), (1, t1)):
        ops = ", ".join(operand(i, hi) for i in range(n))
        out.append("\tv_%s_f32%s %s, %s" % (op, "_e64" if op == "mul" else "", t, ops))
    out += ["\tv_mov_b32_e32 v%d, %s" % (d, t0), "\tv_mov_b32_e32 v%d, %s" % (d + 1, t1)]
    return out
L2 = eval(edit, {"L": L, "re": re, "scal": scal})
# (temporaries v120 / v121: the kernel is built for four waves per SIMD, 128 VGPRs)
lines[a:b] = L2
out = "\n".join(lines)
out = re.sub(r"(\.amdhsa_kernel _ZN3nrc12_GLOBAL__N_110k_gen_raysILb0E.*?\.amdhsa_next_free_vgpr )\d+", r"\g<1>128", out, flags=re.S)
out = re.sub(r"(\.amdhsa_kernel _ZN3nrc12_GLOBAL__N_110k_gen_raysILb0E.*?\.amdhsa_accum_offset )\d+", r"\g<1>128", out, flags=re.S)
open(path, "w").write(out)
print("\n".join(L2))
PY
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/int.s -o $W/int.o
$LL/ld.lld -shared $W/int.o -o $W/int.hsaco
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/int.hsaco -output=$W/int.hipfb
/opt/rocm/bin/hipcc $FL --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/int.hipfb -c $SRC -o $W/nrc_integrator.o 2>/dev/null
OUT=nrc-hpm-renderer_amd/lib_px$NAME; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libnrc_hpm.so nrc-hpm-renderer_amd/lib_bs0/nrc_mlp.o $W/nrc_integrator.o nrc-hpm-renderer_amd/lib_bs0/nrc_api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O2 -Iinclude tests/cpp/stress_main.cpp -o tests/cpp/_build/stress_main_px$NAME \
    -L$OUT -lnrc_hpm -pthread "-Wl,-rpath,\$ORIGIN/../../../$OUT"
echo "built px$NAME"
