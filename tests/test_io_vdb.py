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
