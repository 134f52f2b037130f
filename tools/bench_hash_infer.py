#!/usr/bin/env python3
"""Dense inference of the HashGrid model (k_encode_hash_lm + k_infer_gen), event-timed:  python3 tools/bench_hash_infer.py [n] [reps]
(the level-major mapping is what a device without eight XCDs gets: NRC_DEBUG=assume_xcds=4)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nrc_hpm_renderer_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1920 * 1080
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.cuda.set_device(0)
c = api.NeuralRadianceCache(api.AppConfig(pos_id=0))
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand((n, 5), device="cuda", generator=g)
y = torch.empty((n, 3), device="cuda")
for _ in range(60):
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
print("HashGrid 6x64 dense inference, n=%d: %.4f ms  %.2f Gsamples/s  (%.0f G gathers/s)" % (n, ms, n / ms / 1e6, n * 128 / ms / 1e6))
