#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ from the reference checkout's DATA files.

Run in the authoring container only (needs /root/reference):  python tests/golden/make_golden.py
Inputs  (data, not source):  /root/reference/data/volume/wdas_cloud_sixteenth.vdb
                             /root/reference/reference/{0,1,2,4,5}/0.exr
Outputs: cloud_sixteenth_u8.npz   density volume quantised like src/Texture3D.cpp:106 (texture memory order [k][j][i])
         exr_stats.json           per-image statistics
         exr_{0,4}_240x135.npz    8x8 box-filtered RGBA of the two EXRs whose estimator matches the checked-in shaders
         exr_{0,4}_1920x1080.npz  the same two EXRs at full size, lossless: the images are grey (R = G = B, asserted), so
                                  {L = R, A} in fp32 is every number they hold (3.5 MB each) -- what Reference::CompareNrc
                                  (src/Reference.cpp:72-107) compares a frame with; tools/convergence.py and the quality tests use them
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nrc_hpm_renderer_amd.io_exr as io_exr  # noqa: E402
import nrc_hpm_renderer_amd.io_vdb as io_vdb  # noqa: E402
from nrc_hpm_renderer_amd import scene  # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    vol, info = io_vdb.from_vdb(os.path.join(REF, "data/volume/wdas_cloud_sixteenth.vdb"))
    assert info["active_voxels"] == info["file_voxel_count"] == 415642
    u8 = scene.quantize_density(vol)
    np.savez_compressed(os.path.join(OUT, "cloud_sixteenth_u8.npz"), density=u8,
                        bbox_min=np.array(info["bbox_min"]), bbox_max=np.array(info["bbox_max"]))
    stats = {}
    for sid in (0, 1, 2, 4, 5):
        img = io_exr.read_exr(os.path.join(REF, "reference/%d/0.exr" % sid))
        a = img[..., 3]
        rgb = img[..., :3]
        stats[str(sid)] = dict(
            width=int(img.shape[1]), height=int(img.shape[0]),
            coverage=float((a > 0).mean()), alpha_ge_999=float((a >= 0.999).mean()), mean_alpha=float(a.mean()),
            mean_rgb_valid=float(rgb[a > 0].mean()), mean_rgb_all=float(rgb.mean()), max_rgb=float(rgb.max()),
            background=float(rgb[0, 0, 0]), centre=[float(x) for x in img[540, 960]],
            radiance_pin=bool(sid in (0, 4)),
        )
        if sid in (0, 4):
            ds = img.reshape(135, 8, 240, 8, 4).mean(axis=(1, 3)).astype(np.float32)
            np.savez_compressed(os.path.join(OUT, "exr_%d_240x135.npz" % sid), rgba=ds)
            assert (img[..., 0] == img[..., 1]).all() and (img[..., 0] == img[..., 2]).all()
            np.savez_compressed(os.path.join(OUT, "exr_%d_1920x1080.npz" % sid), L=img[..., 0].copy(), A=img[..., 3].copy())
    with open(os.path.join(OUT, "exr_stats.json"), "w") as f:
        json.dump(stats, f, indent=1, sort_keys=True)
    print(json.dumps(stats, indent=1))


if __name__ == "__main__":
    main()
