"""The C-ABI library loads and exports every symbol include/nrc_hpm.h declares; without a GPU the product fails
loudly (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "nrc_hpm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(nrc_[a-z0-9_]+)\s*\(", txt))
    return sorted(n for n in names if n != "nrc_grad_hook")


def test_library_exports_every_declared_symbol(api):
    L = api.load_library()
    decl = declared_symbols()
    assert len(decl) >= 50
    missing = [s for s in decl if not hasattr(L, s)]
    assert missing == []
    assert sorted(api.ABI_SYMBOLS) == decl           # the Python mirror binds exactly the declared surface


def test_version_and_default_config(api):
    L = api.load_library()
    assert b"gfx950" in L.nrc_version()
    c = api.NrcConfig()
    L.nrc_config_default(C.byref(c))
    assert c.nn_width == 64 and c.nn_depth == 6 and c.pos_id == 3 and c.dir_id == 0 and c.seed == 1337


def test_null_arguments_return_error_codes(api):
    L = api.load_library()
    assert L.nrc_cache_create(None, None) == -1
    assert b"SkyRenderer ERROR" in L.nrc_last_error()
    assert L.nrc_renderer_render(None, 0) == -1
    assert L.nrc_cache_destroy(None) == 0


def test_no_cpu_fallback_without_gpu(api):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="SkyRenderer ERROR"):
        api.NeuralRadianceCache(api.AppConfig())


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "nrc-hpm-renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "nrc_oracle" not in txt, f


def test_headers_compile_standalone():
    """include/nrc_hpm.h is plain C99 and include/nrc_hpm.hpp is self-contained C++17 (no HIP, no torch in the boundary)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = "-I" + os.path.join(root, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", inc,
                           os.path.join(root, "tests", "cpp", "header_check.c")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", inc,
                           os.path.join(root, "tests", "cpp", "header_check.cpp")])



LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "nrc-hpm-renderer_amd", "lib", "libnrc_hpm.so")
CSRC = os.path.join(ROOT, "nrc-hpm-renderer_amd", "csrc")


def shipped_code_objects(tmp):
    """the gfx950 code objects inside the SHIPPED lib/libnrc_hpm.so (one per translation unit), as (disassembly, notes) pairs.
    llvm-objdump --offloading writes the bundles' members next to its input, so it runs on a copy."""
    import glob
    import shutil
    import subprocess
    so = os.path.join(tmp, "libnrc_hpm.so")
    shutil.copy(LIB, so)
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = []
    for co in sorted(glob.glob(so + ".*gfx950")):
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
        out.append((dis, notes))
    return out


def kernel_bodies(dis):
    """{mangled name: disassembly text} of one code object"""
    out = {}
    for block in re.split(r"\n(?=[0-9a-f]{16} <[^>]+>:)", dis):
        m = re.match(r"[0-9a-f]{16} <([^>]+)>:", block)
        if m:
            out[m.group(1)] = block
    return out


def test_shipped_library_is_the_build_of_these_sources_and_flags(api):
    """round 3 shipped a lib/ older than its Makefile's flags and nothing noticed.  The Makefile hashes every source
    of the library, the boundary header, the compiler version and the flags into NRC_BUILD_ID (what nrc_version() returns); a lib/ that
    was not rebuilt after an edit of any of them fails here."""
    import subprocess
    want = subprocess.run(["make", "-s", "-C", CSRC, "print-build-id"], check=True, capture_output=True, text=True).stdout.strip()
    assert re.fullmatch(r"[0-9a-f]{16}", want), want
    got = api.load_library().nrc_version().decode()
    assert got.endswith("build " + want), (got, want, "lib/libnrc_hpm.so is stale: run `python __graft_entry__.py` (make -C nrc-hpm-renderer_amd/csrc)")
    assert api.build_id() == want


def test_shipped_kernels_have_no_scratch_in_the_camera_kernels_and_no_swizzled_packed_fp32():
    """The checks of DESIGN.md section 7.1 on the code objects INSIDE the shipped library (round 3 made them on a fresh compile of the
    sources and passed while lib/ held 21 such instructions): no v_pk_mov_b32, no packed FP32 instruction of any kind with op_sel: /
    neg_lo: / neg_hi: operand modifiers (op_sel_hi:[...] is how the hand-written f2 arithmetic broadcasts a scalar operand: allowed),
    no scratch access in k_gen_rays / k_mc_render, and every kernel that can be co-resident with the camera kernels raises its wave
    priority (s_setprio) so that none outranks them."""
    import subprocess
    import tempfile
    swizzled = re.compile(r"v_pk_[a-z0-9_]*_f32[^\n]*\b(op_sel:|neg_hi:|neg_lo:)")
    # kernels that never run beside a frame (set-up, metrics, tests) or run at the default priority, below the camera kernels
    no_prio_ok = ("k_flight_table", "k_query_layout", "k_compare_", "k_test_", "k_publish_loss")
    n_kernels = n_camera = n_spill_checked = 0
    with tempfile.TemporaryDirectory() as d:
        objs = shipped_code_objects(d)
    assert len(objs) == 3                                    # nrc_mlp, nrc_integrator, nrc_api
    for dis, notes in objs:
        assert "v_pk_mov_b32" not in dis
        bad = swizzled.search(dis)
        assert bad is None, bad.group(0)
        for name, body in kernel_bodies(dis).items():
            n_kernels += 1
            if "s_setprio" not in body:
                assert any(k in name for k in no_prio_ok), name
        # (a camera kernel may CARRY a few bytes of private segment -- the register allocator of this compiler leaves a 16-byte spill slot
        # behind whose spills it removed again, plus the 4-byte scavenging slot that comes with any slot -- but it must not TOUCH scratch:
        # the hazard of DESIGN.md 7.1 was a reload; the bodies are checked instruction by instruction)
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", notes):
            if any(k in m.group(1) for k in ("k_gen_rays", "k_mc_render")):
                n_camera += 1
                assert int(m.group(2)) <= 20, (m.group(1), m.group(2))
        # (the instruction regex below only knows two addressing forms.  The code object's own notes are form-independent: a
        # camera kernel has spilled NO vector register -- the only kind of spill that goes to scratch memory -- and the scalar registers it
        # spilled (the fetch-counting build of k_gen_rays keeps 55 of them in the lanes of a VGPR: v_writelane / v_readlane, no memory) cannot
        # be in its private segment, which is smaller than they are)
        for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", notes):
            if any(k in m.group(1) for k in ("k_gen_rays", "k_mc_render")):
                n_spill_checked += 1
                private, sgpr_spills, vgpr_spills = int(m.group(2)), int(m.group(3)), int(m.group(4))
                assert vgpr_spills == 0, (m.group(1), vgpr_spills)
                assert sgpr_spills == 0 or 4 * sgpr_spills > private, (m.group(1), sgpr_spills, private)
        for name, body in kernel_bodies(dis).items():
            if any(k in name for k in ("k_gen_rays", "k_mc_render")):
                touching = re.search(r"\b(scratch_(load|store)\w*|buffer_(load|store)_dword\w*\s+[^\n]*\boffen\b[^\n]*\bs\[0:3\])", body)
                assert touching is None, (name, touching.group(0))
    assert n_spill_checked >= 4
    assert n_kernels >= 70 and n_camera >= 4       # k_gen_rays<0/1>, k_mc_render<0/1>  (k_prep_train -- 16 384 rays, latency-bound -- keeps a few bytes)
    # the workaround was validated with this compiler; another one has to be stressed again (tests/test_gpu_stress.py, tools/stress*.sh)
    ver = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout
    assert "HIP version: 7.2" in ver, "re-validate -fno-slp-vectorize + s_setprio (DESIGN.md 7.1) on this compiler: " + ver.splitlines()[0]
