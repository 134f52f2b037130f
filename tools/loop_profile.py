"""Lane-utilisation profile of gen_rays on the bench workload.

    make -C nrc-hpm-renderer_amd/csrc OUT=../lib_prof EXTRA=-DNRC_LOOP_PROFILE
    NRC_HPM_LIB=nrc-hpm-renderer_amd/lib_prof/libnrc_hpm.so python tools/loop_profile.py

The loop counters slow the kernel ~10x (and their same-address atomics make some XCDs look slower than others); for an
undistorted per-wave timeline build with EXTRA="-DNRC_LOOP_PROFILE -DNRC_NO_LOOP_COUNTERS" (time stamps only).

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
    for k, name in enumerate(KINDS[:6]):
        u, i = out[k], out[8 + k]
        print("%-28s %14d %14d %8.3f %12.2f" % (name, u, i, u / max(i, 1), i / px))
    trk = out[8 + 2] + out[8 + 3]
    if trk:
        print("tracking-loop trips issued with <= 32 lanes active: %.1f %%, with <= 16: %.1f %% (issued kinds 6/7 of the "
              "counter build)" % (100.0 * out[8 + 6] / trk, 100.0 * out[8 + 7] / trk))


    # occupancy over time of the last gen_rays launch (one 8x8 tile per wave)
    nw = ((W + 7) // 8) * ((H + 7) // 8)

    def wave_times():
        tbuf = (C.c_ulonglong * (4 * nw))()
        assert L.nrc_debug_wave_times(tbuf, nw) == 0
        raw = np.frombuffer(tbuf, dtype=np.uint64).reshape(nw, 4)
        t = raw[:, :2].astype(np.float64) * 0.01       # 100 MHz -> us
        t0 = t[:, 0].min()
        wave_times.xcc = (raw[:, 2] & 0xf).astype(np.int64)
        wave_times.hw = raw[:, 3].astype(np.int64)
        return t[:, 0] - t0, t[:, 1] - t0

    start, end = wave_times()
    xcc, hw = wave_times.xcc, wave_times.hw
    dur = end - start
    r.Render(None, False)
    start2, end2 = wave_times()
    dur2 = end2 - start2
    print("per-slot duration correlation between two consecutive frames: %.3f" % np.corrcoef(dur, dur2)[0, 1])
    span = end.max()
    print("waves %d, launch span %.1f us, wave duration mean %.1f / p50 %.1f / p90 %.1f / max %.1f us" %
          (nw, span, dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max()))
    print("mean resident waves %.0f (sum of durations / span)" % (dur.sum() / span))
    edges = np.linspace(0, span, 21)
    for a, b in zip(edges[:-1], edges[1:]):
        resident = (np.minimum(end, b) - np.maximum(start, a)).clip(min=0).sum() / (b - a)
        sel = (start >= a) & (start < b)
        per_xcc = [int(((np.minimum(end, b) - np.maximum(start, a)).clip(min=0) * (xcc == k)).sum() / (b - a)) for k in range(8)]
        print("  %6.1f-%6.1f us: %5.0f waves resident; %5d waves start here, mean duration %.1f us; per XCC %s" %
              (a, b, resident, sel.sum(), dur[sel].mean() if sel.any() else 0.0, per_xcc))
    print("XCC of the first 64 workgroups:", xcc[0:256:4].tolist())
    tiles_x = (W + 7) // 8
    blk_cost = dur.reshape(-1, 4).sum(axis=1)
    print("cost by (workgroup index mod 8):", [round(float(blk_cost[k::8].sum()) / 1e3, 1) for k in range(8)])
    print("cost by (workgroup index mod 16):", [round(float(blk_cost[k::16].sum()) / 1e3, 1) for k in range(16)])
    print("waves per XCC:", [int((xcc == k).sum()) for k in range(8)])
    print("work (sum of durations, ms) per XCC:", [round(float(dur[xcc == k].sum()) / 1e3, 1) for k in range(8)])
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    simd = (hw >> 4) & 0x3
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    skey = key * 4 + simd
    # peak concurrent waves per SIMD / per CU at the middle of the launch
    tmid = 0.35 * span
    live = (start <= tmid) & (end > tmid)
    per_simd = np.bincount(skey[live], minlength=int(skey.max()) + 1)
    per_cu = np.bincount(key[live], minlength=int(key.max()) + 1)
    print("waves live at t=%.0f us: %d; per-SIMD histogram %s; per-CU histogram %s" %
          (tmid, live.sum(), np.bincount(per_simd[per_simd > 0]).tolist(), np.bincount(per_cu[per_cu > 0]).tolist()))
    print("distinct (xcc,se,sh,cu) seen:", len(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))))


if __name__ == "__main__":
    main()
