#!/usr/bin/env python3
"""Measures the statistics of tests/exr_pin.py for the reference's scenes and for perturbed variants -- the numbers the bounds in
tests/test_oracle_golden.py and tests/test_gpu_integrator.py are set from.

  python tests/exr_pin_calibrate.py --backend gpu --frames 8192 --variant-frames 2048 --out gpurun_out/exr_pin_gpu.json
  python tests/exr_pin_calibrate.py --backend oracle --frames 256 --out /tmp/exr_pin_oracle.json
gpu: McHpmRenderer at 1920x1080 (PATH_LENGTH 32, progressive blend), down-sampled 8x8 like the fixtures.
oracle: the CPU restatement at 240x135 (every pixel = the top-left pixel of the fixture's 8x8 block)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VARIANTS = ["none", "dir_x1.01", "dir_x1.02", "dir_x1.05", "dir_x0.98", "env_x1.1", "env_x1.25", "env_x0.8", "g=0.75", "g=0.85",
            "g=-0.8", "density_x1.1", "density_x0.9", "light_from_opposite_side", "light_from_above"]


def render_pair(backend, sc, cloud, name, frames, seed):
    import exr_pin
    s0, s4 = exr_pin.perturbed(sc, cloud, name)
    cam = sc.make_camera(aspect=1920 / 1080)
    out = []
    if backend == "gpu":
        from nrc_hpm_renderer_amd import api
        for scene in (s0, s4):
            mc = api.McHpmRenderer(1920, 1080, 32, True, cam, scene)
            frs = sc.frame_randoms(frames, seed=seed)
            for f in range(frames):
                mc.SetFrameRandom(frs[f])
                mc.Render()
            out.append(exr_pin.downsample8(mc.GetImage().cpu().numpy()))
            mc.Destroy()
    else:
        from oracle import Oracle
        orc = Oracle()
        for scene in (s0, s4):
            img = np.zeros((exr_pin.DS_H, exr_pin.DS_W, 4), np.float32)
            frs = sc.frame_randoms(frames, seed=seed)
            for f in range(frames):
                img, _, _ = orc.mc_render(scene, cam, exr_pin.DS_W, exr_pin.DS_H, 32, frs[f], blend=1.0 / (f + 1), out=img,
                                          threads=os.cpu_count() or 1)
            out.append(img)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gpu")
    ap.add_argument("--frames", type=int, default=8192)
    ap.add_argument("--variant-frames", type=int, default=2048)
    ap.add_argument("--seeds", type=int, default=3, help="independent repetitions of the unperturbed pair (statistical spread)")
    ap.add_argument("--variants", default=",".join(VARIANTS))
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import exr_pin
    from nrc_hpm_renderer_amd import scene as sc
    cloud = np.load(os.path.join(exr_pin.GOLDEN, "cloud_sixteenth_u8.npz"))["density"]
    res = {}
    for name in args.variants.split(","):
        reps = args.seeds if name == "none" else 1
        for rep in range(reps):
            n = args.frames if name == "none" and rep == 0 else args.variant_frames
            t0 = time.time()
            o0, o4 = render_pair(args.backend, sc, cloud, name, n, seed=7 + 100 * rep)
            st = exr_pin.pin_statistics(o0, o4)
            st["frames"], st["seconds"] = n, time.time() - t0
            key = name if rep == 0 else "%s#%d" % (name, rep)
            res[key] = st
            print(key, json.dumps({k: (round(v, 5) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(dict(backend=args.backend, results=res), f, indent=1)


if __name__ == "__main__":
    main()
