"""Stand-alone rate of the training chain (backward -> optimizer step) on an idle GPU:
python3 tools/train_step_rate.py [batch] [steps] [nn_width] [nn_depth] [pos_id]      (NRC_DEBUG=no_fused_opt for the three-launch optimizer)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nrc_hpm_renderer_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
width = int(sys.argv[3]) if len(sys.argv) > 3 else 64
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 6
pos_id = int(sys.argv[5]) if len(sys.argv) > 5 else 3
c = api.NeuralRadianceCache(api.AppConfig(nn_width=width, nn_depth=depth, pos_id=pos_id))
rng = np.random.default_rng(5)
x = torch.from_numpy(rng.random((n, 5), dtype=np.float32)).cuda()
t = torch.from_numpy(rng.random((n, 3), dtype=np.float32)).cuda()
for _ in range(20):
    c.Backward(x, t); c.OptimizerStep()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    c.Backward(x, t); c.OptimizerStep()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("batch %d, %dx%d pos_id %d: %.1f us per training step (fused optimizer: %s), loss %.5f"
      % (n, depth, width, pos_id, dt * 1e6, "no_fused_opt" not in os.environ.get("NRC_DEBUG", ""), c.GetLoss()))
