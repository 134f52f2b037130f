"""Per-frame event timeline of the four-stream frame graph on one bench preset (nrc_renderer_frame_timeline): which stream does a
pipelined frame wait for?   python3 tools/frame_timeline.py [bench args] [--frames K] [--settle STEPS]
prints, per frame: the start of gen_rays and, relative to it, the end of gen_rays / train rays / training / inference / compositing,
plus the idle time of the render stream in front of the launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

argv = sys.argv[1:]
frames = 12
if "--frames" in argv:
    i = argv.index("--frames"); frames = int(argv[i + 1]); del argv[i:i + 2]
settle = 40      # steps in front of the recorded ones (--settle 300: behind the renderer's schedule trials)
if "--settle" in argv:
    i = argv.index("--settle"); settle = int(argv[i + 1]); del argv[i:i + 2]
args = bench.parse_args(argv)
strong = bench.apply_preset(args)
import torch
job = bench.Job(args, strong, 0, 1, False, False)
job.prepare(200, 10)
for _ in range(settle):
    job.step()
torch.cuda.synchronize()
job.ren.ResetStageStats()
for _ in range(10):
    job.step()
tl = job.ren.FrameTimeline()
torch.cuda.synchronize()
n = len(tl)
print("%d frames; frame interval %.4f ms" % (n, (tl[-1, 0] - tl[0, 0]) / (n - 1)))
print("frame  start   gap_before | gen_rays  train_rays  training  inference  composite   (ms after the frame's start)")
for f in range(max(1, n - frames), n):
    s = tl[f, 0]
    print("%4d %8.3f %8.3f   | %7.3f  %9.3f  %8.3f  %9.3f  %9.3f" % (f, s, s - tl[f - 1, 1], tl[f, 1] - s, tl[f, 2] - s, tl[f, 5] - s, tl[f, 3] - s, tl[f, 4] - s))
m = tl[5:]
d = m - m[:, :1]
print("mean after start: gen_rays %.3f  train rays %.3f  training %.3f  inference %.3f  composite %.3f; render-stream gap %.3f"
      % (d[:, 1].mean(), d[:, 2].mean(), d[:, 5].mean(), d[:, 3].mean(), d[:, 4].mean(), (tl[6:, 0] - tl[5:-1, 1]).mean()))
# per-stream busy time per frame: from the moment a stream's work for the frame COULD start (its predecessor on the stream and the events
# it waits for are done) to its end; columns of tl: 0 gen start, 1 gen done, 2 train rays done, 3 inference done, 4 composite done, 5 training done
deferred = False      # (nrc_schedule.composite_defer: off unless the caller sets it)
busy = {"A gen_rays": [], "D train rays": [], "B training": [], "C inference": [], "composite": []}
for f in range(6, n):
    busy["A gen_rays"].append(tl[f, 1] - tl[f, 0])
    busy["D train rays"].append(tl[f, 2] - max(tl[f, 1], tl[f - 2, 5], tl[f - 1, 2]))
    busy["B training"].append(tl[f, 5] - max(tl[f, 2], tl[f - 1, 5]))
    busy["C inference"].append(tl[f, 3] - max(tl[f, 1], tl[f - 1, 5], tl[f - 1, 3] if deferred else tl[f - 1, 4]))
    busy["composite"].append(tl[f, 4] - max(tl[f, 3], tl[f - 1, 4]))
iv = (tl[-1, 0] - tl[5, 0]) / (n - 6)
print("busy per frame (ms; interval %.4f): " % iv + "  ".join("%s %.3f" % (k, float(np.mean(v))) for k, v in busy.items()))
print("schedule:", job.ren.GetSchedule())
job.close()
