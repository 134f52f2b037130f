"""What a flat, refilling tracking loop could gain: per-pixel density look-up counts of one bench frame (a -DNRC_LOOP_PROFILE
build writes them into the origin image) and the lane utilisation of a few wave organisations.

    make -C nrc-hpm-renderer_amd/csrc OUT=../lib_prof EXTRA="-DNRC_LOOP_PROFILE -DNRC_NO_LOOP_COUNTERS"
    NRC_HPM_LIB=nrc-hpm-renderer_amd/lib_prof/libnrc_hpm.so python tools/lane_model.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nrc_hpm_renderer_amd import api, scene as sc  # noqa: E402


def main():
    W, H, N = 1920, 1080, 256
    vol = sc.cached_volume("cloud", N, seed=1337)
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky())
    cam = sc.make_camera(aspect=W / H)
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=14, log2_infer_batch_size=21, scene_id=4, seed=1337)
    torch.cuda.set_device(0)
    nrc = api.NeuralRadianceCache(cfg)
    r = api.NrcHpmRenderer(W, H, True, cam, cfg, scene, nrc)
    r.SetFullVertexImages(True)
    r.Render(None, False)
    f = r.Buffer("origin").cpu().numpy().reshape(H, W, 4)[..., 3].astype(np.float64)
    print("look-ups per pixel: mean %.2f max %d, pixels with none: %.1f %%" % (f.mean(), f.max(), 100.0 * (f == 0).mean()))
    t = f.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(H // 8, W // 8, 64)      # [tile row][tile col][lane]
    live = t.max(-1) > 0
    print("tiles: %d, with work: %d" % (live.size, live.sum()))

    def util(groups):      # groups: [..., k tiles, 64 lanes] -> lane l walks its pixel of each of the k tiles one after another
        per_lane = groups.sum(-2)
        return per_lane.sum() / (64.0 * per_lane.max(-1).sum())

    print("one tile per wave, flat loop (wave time = its longest lane):      utilisation %.3f" % util(t[:, :, None, :]))
    for k in (2, 4, 8):
        g = t[:, :(W // 8) // k * k].reshape(H // 8, -1, k, 64)
        print("%d horizontally adjacent tiles per wave, lanes refill:              utilisation %.3f" % (k, util(g)))
    g = t[:H // 8 // 2 * 2].reshape(-1, 2, W // 8, 64).transpose(0, 2, 1, 3)
    print("2 vertically adjacent tiles per wave:                               utilisation %.3f" % util(g))
    # the bound of any scheme that keeps a pixel on one lane: the costliest pixel of a wave
    print("look-ups of the costliest pixel of a tile / mean over its pixels: median %.2f" % np.median(t.max(-1)[live] / t.mean(-1)[live]))


if __name__ == "__main__":
    main()
