"""Lane-utilisation profile of gen_rays on the bench workload.

    make -C nrc-hpm-renderer_amd/csrc OUT=../lib_prof EXTRA=-DNRC_LOOP_PROFILE
    NRC_HPM_LIB=nrc-hpm-renderer_amd/lib_prof/libnrc_hpm.so python tools/loop_profile.py

Per loop kind: iterations summed over lanes ("useful"), 64 x iterations the wave issued ("issued"), their ratio (lane
utilisation) and the issued iterations per pixel.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nrc_hpm_renderer_amd import api, scene as sc  # noqa: E402
KINDS = ["find_entry_exit primary", "find_entry_exit in-volume", "delta_track step", "ratio_track step", "new_ray_dir call",
         "trace_scene call", "pixel", "-"]


def main():
    W, H, N = 1920, 1080, 256
    cache_file = "/tmp/nrc_cloud_%d_1337.npy" % N
    vol = np.load(cache_file) if os.path.exists(cache_file) else sc.quantize_density(sc.fbm_cloud_volume(N, seed=1337))
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky())
    cam = sc.make_camera(aspect=W / H)
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=14, log2_infer_batch_size=21, scene_id=4,
                        primary_ray_length=1, primary_ray_prob=0.0, train_spp=1, train_ring_buf_size=1.0, seed=1337)
    torch.cuda.set_device(0)
    nrc = api.NeuralRadianceCache(cfg)
    r = api.NrcHpmRenderer(W, H, True, cam, cfg, scene, nrc)
    L = api.load_library()
    out = (C.c_ulonglong * 16)()
    r.Render(None, False)
    assert L.nrc_debug_loop_profile(out, 1) == 0
    frames = 4
    for _ in range(frames):
        r.Render(None, False)
    assert L.nrc_debug_loop_profile(out, 0) == 0
    px = W * H * frames
    print("%-28s %14s %14s %8s %12s" % ("kind", "useful", "issued", "util", "issued/px"))
    for k, name in enumerate(KINDS[:7]):
        u, i = out[k], out[8 + k]
        print("%-28s %14d %14d %8.3f %12.2f" % (name, u, i, u / max(i, 1), i / px))


if __name__ == "__main__":
    main()
