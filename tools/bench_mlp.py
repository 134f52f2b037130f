#!/usr/bin/env python3
"""Micro-benchmark of the fused encode+MLP inference kernel (event-timed on the launch stream)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nrc_hpm_renderer_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1920 * 1080
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.cuda.set_device(0)
width = int(sys.argv[3]) if len(sys.argv) > 3 else 64
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 6
c = api.NeuralRadianceCache(api.AppConfig(nn_width=width, nn_depth=depth))
flop = 2.0 * (80 * width + (depth - 1) * width * width + width * 3)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand((n, 5), device="cuda", generator=g)
x[:, :3] += 31.0
y = torch.empty((n, 3), device="cuda")
for _ in range(5):
    c.Infer(x, y, True)
times = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        c.Infer(x, y, True)
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) / reps)
ms = float(np.median(times))
print("%dx%d threads=%s n=%d  %.4f ms  %.1f TFLOP/s (%.1f%% of 2500)  %.2f Gsamples/s" %
      (depth, width, os.environ.get("NRC_INFER_THREADS", "512"), n, ms, flop * n / ms / 1e9, flop * n / ms / 1e9 / 25.0, n / ms / 1e6))
