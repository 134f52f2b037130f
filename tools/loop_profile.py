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
    # more frames before the time line is read: the launch order in use then comes from a warm frame's tile costs (sampled every
    # 16 frames, applied two frames later)
    for _ in range(int(os.environ.get("LOOP_PROFILE_WARM_FRAMES", "0"))):
        r.Render(None, False)
    px = W * H * frames
    print("%-28s %14s %14s %8s %12s" % ("kind", "useful", "issued", "util", "issued/px"))
    for k, name in enumerate(KINDS[:6]):
        u, i = out[k], out[8 + k]
        print("%-28s %14d %14d %8.3f %12.2f" % (name, u, i, u / max(i, 1), i / px))
    if hasattr(L, "nrc_debug_live_hist") and any(out[:8]):
        hist = (C.c_ulonglong * 24)()
        assert L.nrc_debug_live_hist(hist) == 0
        names = ["ratio tracking, 64-lane trips", "ratio tracking, pair trips (walks alive)", "delta tracking trips"]
        total = float(sum(hist)) or 1.0
        print("wave-level trips by walks alive (1, 2, 3-4, 5-8, 9-16, 17-32, 33-64), %% of all %d trips:" % sum(hist))
        for w in range(3):
            print("  %-44s %s" % (names[w], "  ".join("%5.1f" % (100.0 * hist[8 * w + k] / total) for k in range(7))))
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
    cheap = dur < 6.0
    if cheap.any() and (~cheap).any():      # (the counter build slows every wave: no cheap class there)
        # the cheap waves (tiles the empty-space mask rejects, tiles off the cloud) against the walking ones
        print("waves shorter than 6 us: %d (%.1f %% of the waves, %.2f %% of the wave time); they start between %.1f and %.1f us; the last "
              "longer wave ends at %.1f us of %.1f; wave time still to run after the first cheap wave starts: %.1f %% in longer waves" %
              (cheap.sum(), 100.0 * cheap.mean(), 100.0 * dur[cheap].sum() / dur.sum(), np.percentile(start[cheap], 1), start[cheap].max(),
               end[~cheap].max(), span,
               100.0 * (np.minimum(end[~cheap], span) - np.maximum(start[~cheap], np.percentile(start[cheap], 1))).clip(min=0).sum() / dur.sum()))
        late = (~cheap) & (start > np.percentile(start[cheap], 1))
        print("longer waves that start after the first cheap ones: %d, durations p50 %.1f / p90 %.1f / max %.1f us; the 12 waves that end last "
              "(start, duration): %s" % (late.sum(), np.median(dur[late]) if late.any() else 0, np.percentile(dur[late], 90) if late.any() else 0,
                                         dur[late].max() if late.any() else 0,
                                         [(round(float(start[i]), 1), round(float(dur[i]), 1)) for i in np.argsort(end)[-12:]]))
        last = np.argsort(end)[-400:]
        print("the 400 waves that end last: start p10 %.1f / p50 %.1f / p90 %.1f us, duration p10 %.1f / p50 %.1f / p90 %.1f us, the same launch "
              "slots one frame later: duration p10 %.1f / p50 %.1f / p90 %.1f us" %
              (np.percentile(start[last], 10), np.median(start[last]), np.percentile(start[last], 90), np.percentile(dur[last], 10),
               np.median(dur[last]), np.percentile(dur[last], 90), np.percentile(dur2[last], 10), np.median(dur2[last]), np.percentile(dur2[last], 90)))
        first = np.argsort(start)[:4096]
        print("the 4096 waves that start first: duration p10 %.1f / p50 %.1f / p90 %.1f us" %
              (np.percentile(dur[first], 10), np.median(dur[first]), np.percentile(dur[first], 90)))
    # what a better launch order could give: greedy list scheduling of the measured wave durations on the chip's wave slots (no
    # contention model: durations as measured), in the launch order used, in the order of this frame's own durations (the ideal
    # predictor) and in the order of the next frame's durations (a one-sample predictor)
    import heapq

    def makespan(order, slots=5120):
        free = [0.0] * slots
        heapq.heapify(free)
        t_end = 0.0
        for i in order:
            t = heapq.heappop(free) + dur[i]
            t_end = max(t_end, t)
            heapq.heappush(free, t)
        return t_end
    print("list-scheduling model on 5120 slots: launch order used %.1f us, sorted by this frame's durations %.1f us, by the next frame's "
          "durations %.1f us, by the mean of both %.1f us; sum of durations / slots %.1f us" %
          (makespan(np.argsort(start)), makespan(np.argsort(-dur)), makespan(np.argsort(-dur2)), makespan(np.argsort(-(dur + dur2))),
           dur.sum() / 5120))
    # how much a better predictor could give: order by the mean duration of K earlier frames (same launch order, so the same
    # launch-position bias), evaluated on a frame that is not in the mean
    if os.environ.get("LOOP_PROFILE_PREDICTOR"):
        hist = [dur, dur2]
        for _ in range(9):
            r.Render(None, False)
            s_k, e_k = wave_times()
            hist.append(e_k - s_k)
        held = hist[-1]

        def makespan_on(order, d, slots=5120):
            free = [0.0] * slots
            heapq.heapify(free)
            t_end = 0.0
            for i in order:
                t = heapq.heappop(free) + d[i]
                t_end = max(t_end, t)
                heapq.heappush(free, t)
            return t_end
        print("held-out frame: sum / slots %.1f us, its own order %.1f us" % (held.sum() / 5120, makespan_on(np.argsort(-held), held)))
        for K in (1, 2, 4, 8):
            mean_k = np.mean(hist[-1 - K:-1], axis=0)
            max_k = np.max(hist[-1 - K:-1], axis=0)
            print("  predictor = mean of %d earlier frames: %.1f us; max of them: %.1f us; rank correlation of the mean with the held-out frame among the 11 000 longest: %.3f" %
                  (K, makespan_on(np.argsort(-mean_k), held), makespan_on(np.argsort(-max_k), held),
                   np.corrcoef(mean_k[np.argsort(-held)[:11000]], held[np.argsort(-held)[:11000]])[0, 1]))
    for thr in (0.90, 0.95, 0.99, 0.999):
        order_e = np.sort(end[~cheap])
        print("  %.1f %% of the longer waves have ended at %.1f us" % (100 * thr, order_e[int(thr * (order_e.size - 1))]))
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
