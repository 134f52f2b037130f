"""Register / LDS / scratch use of the library's kernels as hipcc reports them:  python3 tools/kernel_regs.py <file.hip> [name filter] [extra flags...]"""
import re, subprocess, sys, os
src = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
strict = ["-ffp-contract=off"] if "integrator" in src or "api" in src else []
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-fno-slp-vectorize"] + strict + extra + \
      ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(src))).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0].replace("void ", "")
    if flt in d:
        print("%-46s VGPR %3d AGPR %3d  scratch %3d  LDS %6d  waves/SIMD %d  SGPR spill %d" % (d[:46], r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("ScratchSize", 0),
              r.get("LDS Size", 0), r.get("Occupancy", 0), r.get("SGPRs Spill", 0)))
