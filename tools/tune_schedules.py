#!/usr/bin/env python3
"""Lets the renderer's tuner settle on the BASELINE presets' frames and saves what it chose (nrc_schedule_cache_save) -- the table the Python
mirror loads with the library (nrc-hpm-renderer_amd/schedules.txt), so that a run shorter than the tuner's ~400 frames (the driver's 25-frame
bench) starts on the schedule a long run would have found.  No knob changes a pixel; the keys name device, model, volume and frame.

    NRC_SCHEDULE_CACHE= python tools/tune_schedules.py --out gpurun_out/schedules.txt [--rounds 3]

Every preset is tuned `rounds` times in fresh renderers (the table is cleared of the key in between by running with an empty cache); a knob
is written only if all rounds agree on it, otherwise the default stays -- a choice that flips between runs is noise, not a preference."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NRC_SCHEDULE_CACHE"] = ""          # tune from the defaults, not from an earlier table
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

PRESETS = [
    ("c2", []),
    ("c2 q2-fixed", ["--compat-fix", "2"]),
    ("c5", ["--config", "c5"]),
    ("hashgrid (the reference's default model)", ["--pos-id", "0"]),
    ("c4 on one GPU", ["--config", "c4"]),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "schedules.txt"))
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--max-steps", type=int, default=600)
    a = ap.parse_args()
    import bench
    import torch
    from nrc_hpm_renderer_amd import api
    torch.cuda.set_device(0)
    lines = ["# tools/tune_schedules.py on %s, build %s: <key> camera_priority_low cost_order_lag xcd_window" % (torch.cuda.get_device_name(0), api.build_id())]
    for name, extra in PRESETS:
        votes, key = [], None
        for _ in range(a.rounds):
            api.clear_schedule_cache()      # (the previous round's result is in the process-wide table: tune from the defaults again)
            args = bench.parse_args(extra + ["--no-quality", "--no-cpu-baseline"])
            strong = bench.apply_preset(args)
            job = bench.Job(args, strong, 0, 1, False, False)
            job.prepare(a.max_steps, 0)
            s = job.ren.GetSchedule()
            key = s["key"]
            steps = 0
            while not s["tuning_done"] and steps < a.max_steps:
                for _ in range(10):
                    job.step()
                steps += 10
                torch.cuda.synchronize()
                s = job.ren.GetSchedule()
            votes.append((s["camera_priority_low"], s["cost_order_lag"], s["xcd_window"], s["tuning_done"], steps * args.spp))
            job.close()
        print(name, key, votes, flush=True)
        done = [v for v in votes if v[3]]
        if len(done) < a.rounds:
            lines.append("# %s: the tuner did not finish in %d frames in %d of %d rounds -- default kept" % (name, a.max_steps * 4, a.rounds - len(done), a.rounds))
            continue
        knobs = []
        for k, default in ((0, 0), (1, 2), (2, 2)):
            c = collections.Counter(v[k] for v in done)
            val, cnt = c.most_common(1)[0]
            knobs.append(val if cnt == len(done) else default)
        lines.append("# %s: rounds chose %s" % (name, [v[:3] for v in done]))
        lines.append("%s %d %d %d" % (key, knobs[0], knobs[1], knobs[2]))
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
