#!/usr/bin/env python3
"""Where the fused training kernels' gradient differs from the oracle's, and why the smoke test's figure moved from 7.5e-6 (rounds 2-4) to
3.5e-4 (round 5, tiny-cuda-nn's initialisation) -- VERDICT r05 "weak" 2.

Both sides round at the same points (weights, activations and deltas to fp16; products summed wider: fp32 MFMA accumulators here,
double in the oracle), so they differ only where a sum lands close enough to a rounding boundary for the accumulation order to decide:
  * an fp16 rounding flip of an activation / delta (one ulp = 2^-11 relative of ONE number of ONE ray): invisible in the gradient's norm;
  * a ReLU decision: a pre-activation within accumulation error of zero is kept by one side and dropped by the other -- the whole
    delta row of that neuron for that ray appears or vanishes.
This tool measures, on the smoke test's batch (1 024 rays) and on the parity test's (2 048 rays):
  1. the gradient's rel-L2 per layer, for the shipped initialisation and for seeds 1..S (the same model, other draws);
  2. the rays that carry the difference: the batch is run in tiles of 32 rays against the whole batch's normaliser, each tile's gradient
     compared with the oracle's for the same tile; for the worst tiles every ray alone (32 copies of it fill a tile);
  3. the inference outputs of those rays (a forward-side flip shows there too).
  python tools/grad_drift.py [--seeds 12] > profiles/r06_grad_drift.txt
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LAYERS = [(64, 80)] + [(64, 64)] * 5 + [(3, 64)]


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / max(np.linalg.norm(b), 1e-30))


def per_layer(g, g_ref):
    out, off = [], 0
    for o, i in LAYERS:
        out.append(rel(g[off:off + o * i], g_ref[off:off + o * i]))
        off += o * i
    return out


def gpu_grad(torch, c, x, t, n_norm):
    c.Backward(torch.from_numpy(np.ascontiguousarray(x)).cuda(), torch.from_numpy(np.ascontiguousarray(t)).cuda(), nNorm=n_norm)
    return c.GetParams(4).astype(np.float64) / 128.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    args = ap.parse_args()
    import torch
    from nrc_hpm_renderer_amd import api
    from oracle import Oracle
    orc = Oracle()
    print("# tools/grad_drift.py, build %s" % api.build_id())

    def batch(kind):
        if kind == "smoke":          # __graft_entry__.smoke()
            rng = np.random.default_rng(0)
            x = rng.random((4096, 5), dtype=np.float32)
            x[:, :3] = x[:, :3] + 31.0
            t = rng.random((1024, 3), dtype=np.float32)
            return x[:1024].copy(), t
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        rng = np.random.default_rng(21)      # tests/test_gpu_mlp.py::test_backward_matches_oracle (queries(2048, seed=21), targets seed 22)
        x = rng.random((2048, 5), dtype=np.float32)
        x[:, :3] += 31.0
        x[:, 3] = x[:, 3] * 2.0 - 0.5
        x[rng.random(2048) < 0.1, 4] = np.nan
        t = (np.random.default_rng(22).random((2048, 3), dtype=np.float32) * 2).astype(np.float32)
        return x, t

    for kind in ("smoke", "parity_test"):
        x, t = batch(kind)
        n = x.shape[0]
        print("\n## batch '%s': %d rays" % (kind, n))
        print("# seed | gradient rel-L2 | per layer (first, hidden 1..5, output) | loss rel diff")
        worst = None
        for seed in [1337] + list(range(1, args.seeds + 1)):
            c = api.NeuralRadianceCache(api.AppConfig(train_batch_count=1, log2_train_batch_size=10, seed=seed))
            onn = orc.nn_create(seed=seed)
            assert np.array_equal(c.GetParams(0), onn.buffer(0))
            g = gpu_grad(torch, c, x, t, n)
            loss = c.GetLoss()
            loss_ref = onn.backward(x, t)
            g_ref = np.array(onn.buffer(4), np.float64)
            r = rel(g, g_ref)
            print("%d %.3e  %s  %.2e" % (seed, r, " ".join("%.2e" % v for v in per_layer(g, g_ref)), abs(loss - loss_ref) / abs(loss_ref)))
            if seed == 1337:
                worst = (seed, c, onn, g, g_ref)
            else:
                c.Destroy()
        seed, c, onn, g, g_ref = worst
        # 2. which rays carry it
        err_total = np.linalg.norm(g - g_ref)
        tiles = []
        for a in range(0, n, 32):
            gt = gpu_grad(torch, c, x[a:a + 32], t[a:a + 32], n)
            onn.backward(x[a:a + 32], t[a:a + 32], n_norm=n)
            gr = np.array(onn.buffer(4), np.float64)
            tiles.append((float(np.linalg.norm(gt - gr)), a))
        tiles.sort(reverse=True)
        e2 = np.array([e * e for e, _ in tiles])
        print("# seed %d: |g - g_ref| = %.3e (|g_ref| = %.3e).  Tiles of 32 rays, error norm: top 5 = %s; they hold %.1f %% of the summed squared "
              "tile errors; median tile %.2e" % (seed, err_total, np.linalg.norm(g_ref), ", ".join("%.2e@%d" % tt for tt in tiles[:5]),
                                                 100.0 * e2[:5].sum() / e2.sum(), float(np.median([e for e, _ in tiles]))))
        out_gpu = torch.empty((n, 3), device="cuda")
        c.Infer(torch.from_numpy(x).cuda(), out_gpu, False)
        out_gpu = out_gpu.cpu().numpy()
        out_ref = onn.forward(x, use_ema=False, mode=1)
        for e_tile, a in tiles[:3]:
            rays = []
            for s in range(a, a + 32):
                xs, ts = np.repeat(x[s:s + 1], 32, axis=0), np.repeat(t[s:s + 1], 32, axis=0)
                gs = gpu_grad(torch, c, xs, ts, n)
                onn.backward(xs, ts, n_norm=n)
                gr = np.array(onn.buffer(4), np.float64)
                rays.append((float(np.linalg.norm(gs - gr)) / 32.0, s, per_layer(gs, gr)))
            rays.sort(reverse=True)
            e, s, pl = rays[0]
            print("#   tile @%d (error %.2e): worst ray %d carries %.2e (next %.2e); its per-layer rel-L2 %s; forward output GPU %s oracle %s"
                  % (a, e_tile, s, e, rays[1][0], " ".join("%.1e" % v for v in pl), np.array2string(out_gpu[s], precision=5),
                     np.array2string(out_ref[s], precision=5)))
        # what is left without the three worst rays' tiles
        keep = np.ones(n, bool)
        for _, a in tiles[:3]:
            keep[a:a + 32] = False
        gk = gpu_grad(torch, c, x[keep], t[keep], n)
        onn.backward(x[keep], t[keep], n_norm=n)
        print("# without those three tiles (%d rays): rel-L2 %.3e" % (int(keep.sum()), rel(gk, np.array(onn.buffer(4), np.float64))))
        c.Destroy()
    return 0


if __name__ == "__main__":
    sys.exit(main())
