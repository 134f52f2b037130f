import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
"""how the frame time settles after start-up: 5 steps (20 frames) per line, host-synchronised between lines"""
import bench
args = bench.parse_args(["--gpus", "1", "--no-cpu-baseline"])
bench.apply_preset(args)
job = bench.Job(args, False, 0, 1, False, False)
from nrc_hpm_renderer_amd import api, scene as sc
mode = sys.argv[1] if len(sys.argv) > 1 else "mc"
if mode == "mc":
    bench.gpu_mc_baseline(api, sc, job.scene, args.width, args.height)      # (returns (figures, None))
job.randoms = sc.frame_randoms(400 * 4 + 8, seed=1337)
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for pas in range(passes):
  if pas: time.sleep(float(sys.argv[3]) if len(sys.argv) > 3 else 0.05)
  out = []
  for blk in range(12):
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(5): job.step()
      torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 20 * 1e3)
      st = job.ren.StageStats(reset=True)
      out[-1] = (out[-1], st)
  for k, (t, st) in enumerate(out[:12]):
      print("pass %d %s frames %3d-%3d: %.4f ms/frame  %s" % (pas, mode, 20 * k, 20 * k + 19, t, {a: round(b, 3) for a, b in st.items() if b and a not in ("frames", "total")}))
