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
# The GPU reaches its load clocks only after ~25 ms of work behind an idle period (launch by launch: 0.106, 0.125, 0.116, 0.108 ...
# 0.094 ms for blocks of 20 launches of the 6x64 model): 40 ms of untimed launches first, then the timed blocks without a pause.
# Under `rocprofv3 --kernel-trace` the ramp is in the trace: tools/trace_tail.py reports the steady part.
warm = int(os.environ.get("NRC_BENCH_MLP_WARM", "400"))
for _ in range(warm):
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
