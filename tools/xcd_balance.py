#!/usr/bin/env python3
"""Is gen_rays' work balanced over the eight XCDs?  (VERDICT r05 "weak" 3: the counter build's profile showed 1.9 x between the XCCs' summed
wave durations; tools/loop_profile.py itself warns that the counters' same-address atomics distort exactly that.)

A STAMPS-ONLY production build -- the shipped kernel plus one wall-clock stamp at each wave's start and end and its HW_REG_XCC_ID:

    make -C nrc-hpm-renderer_amd/csrc OUT=../lib_stamps EXTRA="-DNRC_LOOP_PROFILE -DNRC_NO_LOOP_COUNTERS"
    NRC_HPM_LIB=nrc-hpm-renderer_amd/lib_stamps/libnrc_hpm.so python tools/xcd_balance.py [--config c2|c5] [--window W]

Per XCC of the last gen_rays launch of a warm renderer: waves, summed wave time, the moment its last walking wave ends, and how full its
640 wave slots (32 CUs x 4 SIMDs x 5 waves) were until then.  The hardware deals workgroups to the XCCs round-robin, so an XCC whose share of
the launch is heavier than the others' ends later while the others idle: `makespan spread` = (latest XCC end - mean XCC end) / launch span.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
from nrc_hpm_renderer_amd import api, scene as sc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=["c2", "c5"])
    ap.add_argument("--window", type=int, default=-1, help="pin nrc_schedule.xcd_window (default: the library's)")
    ap.add_argument("--train", type=int, default=0, help="1: the whole frame graph runs beside the launch (training on)")
    ap.add_argument("--frames", type=int, default=48)
    args = ap.parse_args()
    W, H = 1920, 1080
    if args.config == "c5":
        vol = sc.cached_volume("smoke", 512, seed=1337)
        model = dict(nn_width=128, nn_depth=8)
    else:
        vol = sc.cached_volume("cloud", 256, seed=1337)
        model = dict()
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky())
    cam = sc.make_camera(aspect=W / H)
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=14, log2_infer_batch_size=21, scene_id=4, primary_ray_length=1,
                        primary_ray_prob=0.0, train_spp=1, train_ring_buf_size=1.0, seed=1337, **model)
    torch.cuda.set_device(0)
    nrc = api.NeuralRadianceCache(cfg)
    r = api.NrcHpmRenderer(W, H, True, cam, cfg, scene, nrc)
    if args.window >= 0:
        r.SetSchedule(xcd_window=args.window)
    L = api.load_library()
    if not hasattr(L, "nrc_debug_wave_times"):
        raise SystemExit("this library has no stamps: build with EXTRA=\"-DNRC_LOOP_PROFILE -DNRC_NO_LOOP_COUNTERS\" and set NRC_HPM_LIB")
    frs = sc.frame_randoms(args.frames, seed=1337)
    r.RenderFrames(frs, bool(args.train))
    torch.cuda.synchronize()
    nw = ((W + 7) // 8) * ((H + 7) // 8)
    n_slots = min(65536, (nw + 3) // 4 * 4 + 64)
    tbuf = (C.c_ulonglong * (4 * n_slots))()
    assert L.nrc_debug_wave_times(tbuf, n_slots) == 0
    raw = np.frombuffer(tbuf, dtype=np.uint64).reshape(n_slots, 4)
    t = raw[:, :2].astype(np.float64) * 0.01       # 100 MHz wall clock -> us
    ok = (raw[:, 0] != 0) & (raw[:, 1] >= raw[:, 0])
    # stamps of EARLIER launches survive in slots this launch's padding waves did not finish: keep the newest launch only
    ok &= t[:, 0] > np.median(t[ok, 0]) - 600.0
    start, end, xcc = t[ok, 0], t[ok, 1], (raw[ok, 2] & 0xf).astype(np.int64)
    t0 = start.min()
    start, end = start - t0, end - t0
    dur = end - start
    span = end.max()
    walking = dur > 6.0
    print("# tools/xcd_balance.py --config %s --train %d: schedule %s, build %s" % (args.config, args.train, r.GetSchedule(), api.build_id()))
    print("waves stamped %d, launch span %.1f us, wave time %.1f ms, walking waves (> 6 us) %d holding %.1f %% of the wave time"
          % (ok.sum(), span, dur.sum() / 1e3, walking.sum(), 100.0 * dur[walking].sum() / dur.sum()))
    print("%4s %7s %12s %14s %12s %18s" % ("XCC", "waves", "wave ms", "last end us", "walking", "slots busy to end"))
    ends, sums = [], []
    for k in range(8):
        m = xcc == k
        if not m.any():
            continue
        e = end[m & walking].max() if (m & walking).any() else 0.0
        ends.append(e)
        sums.append(dur[m].sum())
        busy = (np.minimum(end[m], e) - np.minimum(start[m], e)).sum() / max(e * 640.0, 1e-9)
        print("%4d %7d %12.2f %14.1f %12d %17.1f %%" % (k, m.sum(), dur[m].sum() / 1e3, e, (m & walking).sum(), 100.0 * busy))
    ends, sums = np.array(ends), np.array(sums)
    print("wave time per XCC: max / mean = %.3f, min / mean = %.3f" % (sums.max() / sums.mean(), sums.min() / sums.mean()))
    print("XCC end of walking: latest %.1f us, mean %.1f us, earliest %.1f us; makespan spread (latest - mean) / span = %.1f %%"
          % (ends.max(), ends.mean(), ends.min(), 100.0 * (ends.max() - ends.mean()) / span))
    # resident waves per XCC over the launch, ten bins
    print("resident waves per XCC (of 640 slots) in ten equal time bins:")
    edges = np.linspace(0.0, span, 11)
    for k in range(8):
        m = xcc == k
        row = [((np.minimum(end[m], b) - np.maximum(start[m], a)).clip(min=0).sum() / (b - a)) for a, b in zip(edges[:-1], edges[1:])]
        print("  XCC %d: %s" % (k, " ".join("%4.0f" % v for v in row)))
    r.Destroy()
    nrc.Destroy()


if __name__ == "__main__":
    main()
