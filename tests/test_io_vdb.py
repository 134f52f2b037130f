"""The from-scratch VDB reader (nrc-hpm-renderer_amd/io_vdb.py; Texture3D::FromVDB semantics, src/Texture3D.cpp:12-82) on the
reference's own file: data/volume/wdas_cloud_sixteenth.vdb -- 415 642 active voxels (= the file's `file_voxel_count` metadata),
bounding box (-66,-21,-90)..(59,64,63), and, quantised like src/Texture3D.cpp:106, a dense R8 volume equal to the committed fixture
byte for byte.  Skipped where the reference checkout is absent (the GPU box); tests/golden/make_golden.py wrote the fixture."""
import os

import numpy as np
import pytest

VDB = "/root/reference/data/volume/wdas_cloud_sixteenth.vdb"


@pytest.mark.skipif(not os.path.exists(VDB), reason="reference checkout not present")
def test_vdb_reader_reproduces_the_fixture_and_the_files_voxel_count(sc, cloud16):
    from nrc_hpm_renderer_amd import io_vdb
    vol, info = io_vdb.from_vdb(VDB)
    assert info["active_voxels"] == 415642 == info["file_voxel_count"]
    assert tuple(info["bbox_min"]) == (-66, -21, -90) and tuple(info["bbox_max"]) == (59, 64, 63) and info["extent"] == (126, 86, 154)
    assert vol.dtype == np.float32 and vol.shape == (126, 86, 154) and float(vol.max()) == 1.0      # src/Texture3D.cpp:74
    u8 = sc.quantize_density(vol)
    assert u8.shape == cloud16.shape == (154, 86, 126) and np.array_equal(u8, cloud16)


def _build(name):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "tests", "cpp", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, name)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", name + ".cpp"), "-o", exe])
    return exe


@pytest.mark.skipif(not os.path.exists(VDB), reason="reference checkout not present")
def test_cpp_vdb_reader_equals_the_fixture(cloud16, tmp_path):
    """include/nrc_vdb.hpp -- the reader behind en::HpmScene(appConfig, "file.vdb") (src/HpmScene.cpp:44 -> Texture3D::FromVDB) -- on
    the reference's file: the file's voxel count, its bounding box, and the dense R8 volume of the committed fixture byte for byte"""
    import json
    import subprocess
    exe = _build("vdb_main")
    out = tmp_path / "cloud.u8"
    r = subprocess.run([exe, VDB, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout)
    assert info["active_voxels"] == 415642 == info["file_voxel_count"] and info["max"] == 1.0 and info["grid"] == "density"
    assert (info["nx"], info["ny"], info["nz"]) == (126, 86, 154) and info["bbox_min"] == [-66, -21, -90] and info["bbox_max"] == [59, 64, 63]
    assert np.array_equal(np.fromfile(out, np.uint8), cloud16.reshape(-1))


@pytest.mark.skipif(not os.path.exists("/root/reference/reference/0/0.exr"), reason="reference checkout not present")
def test_cpp_exr_reader_equals_the_python_reader(tmp_path):
    """include/nrc_exr.hpp -- what en::Reference loads its ground truth with (tinyexr's LoadEXR in src/Reference.cpp:617-631) -- on the
    reference's own ZIP-compressed 1920x1080 images: bit-identical to io_exr.read_exr, whose statistics tests/golden/exr_stats.json pins"""
    import subprocess
    from nrc_hpm_renderer_amd import io_exr
    exe = _build("exr_main")
    for sid in (0, 4):
        src = "/root/reference/reference/%d/0.exr" % sid
        out = tmp_path / ("ref%d.f32" % sid)
        r = subprocess.run([exe, src, str(out)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        a = np.fromfile(out, np.float32).reshape(1080, 1920, 4)
        assert np.array_equal(a.view(np.uint32), io_exr.read_exr(src).view(np.uint32))


def _build_asan(name):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "tests", "cpp", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, name + "_asan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall",
                           "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", name + ".cpp"), "-o", exe])
    return exe


def _mutations(data, rng, n):
    """truncations, bytes flipped in the header region, and 32/64-bit fields overwritten with huge / negative values"""
    out = [data[:k] for k in (0, 3, 8, 40, 100, len(data) // 2, len(data) - 1)]
    for _ in range(n):
        b = bytearray(data)
        pos = int(rng.integers(0, min(len(b), 600)))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            b[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            b[pos:pos + 4] = (0xFFFFFFF0).to_bytes(4, "little")
        else:
            b[pos:pos + 8] = (0x7FFFFFFFFFFFFFF0).to_bytes(8, "little")
        out.append(bytes(b[:len(data)]))
    return out


def test_cpp_readers_reject_malformed_files_without_reading_out_of_bounds(tmp_path):
    """offsets and sizes the VDB / EXR readers take from the file are checked overflow-safely -- a truncated or corrupted
    file ends in `SkyRenderer ERROR` (exit 1) or is read (exit 0), never in an out-of-bounds access (the readers run under
    AddressSanitizer + UBSan here: a report aborts with another status)"""
    import subprocess
    from nrc_hpm_renderer_amd import io_exr
    rng = np.random.default_rng(3)
    img = rng.random((24, 20, 4), dtype=np.float32)
    good = tmp_path / "good.exr"
    io_exr.write_exr(str(good), img, compression="zip")
    exe = _build_asan("exr_main")
    r = subprocess.run([exe, str(good), str(tmp_path / "o.f32")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(tmp_path / "o.f32", np.float32).reshape(24, 20, 4), img)
    data = good.read_bytes()
    seen = {0: 0, 1: 0}
    for k, m in enumerate(_mutations(data, rng, 150)):
        bad = tmp_path / "bad.exr"
        bad.write_bytes(m)
        r = subprocess.run([exe, str(bad), str(tmp_path / "o.f32")], capture_output=True, text=True, timeout=60)
        assert r.returncode in (0, 1), (k, r.returncode, r.stderr[-1500:])
        assert r.returncode == 0 or "SkyRenderer ERROR" in r.stderr, r.stderr[-500:]
        seen[r.returncode] += 1
    assert seen[1] > 20
    if not os.path.exists(VDB):
        return
    exe = _build_asan("vdb_main")
    data = open(VDB, "rb").read()
    seen = {0: 0, 1: 0}
    for k, m in enumerate(_mutations(data, rng, 40)):
        bad = tmp_path / "bad.vdb"
        bad.write_bytes(m)
        r = subprocess.run([exe, str(bad), str(tmp_path / "o.u8")], capture_output=True, text=True, timeout=120)
        assert r.returncode in (0, 1), (k, r.returncode, r.stderr[-1500:])
        assert r.returncode == 0 or "SkyRenderer ERROR" in r.stderr, r.stderr[-500:]
        seen[r.returncode] += 1
    assert seen[1] > 10
