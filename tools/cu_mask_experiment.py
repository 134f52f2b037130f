"""Does the frame gain when gen_rays leaves a few CUs to the side streams?  The renderer's stream A (gen_rays) is the caller's stream:
this script makes a stream created with hipExtStreamCreateWithCUMask that excludes `reserve` CUs the current torch stream, builds the
bench job on it and prints the frame time and the stage times.   python tools/cu_mask_experiment.py [reserve CUs ...] [c5]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from nrc_hpm_renderer_amd import scene as sc  # noqa: E402

reserves = [int(a) for a in sys.argv[1:] if a.isdigit()] or [0, 16, 32]
config = "c5" if "c5" in sys.argv else "c2"
torch.cuda.set_device(0)
hip = C.CDLL("libamdhip64.so")
args = bench.parse_args(["--gpus", "1", "--no-cpu-baseline", "--config", config])
bench.apply_preset(args)
for reserve in reserves:
    ptr = C.c_void_p(0)
    if reserve:
        bits = [1] * 256                      # bit i = CU i; leave out `reserve` CUs spread evenly over the mask
        for k in range(reserve):
            bits[(k * 256) // reserve + (256 // reserve) - 1] = 0
        words = (C.c_uint32 * 8)(*[sum(bits[32 * w + b] << b for b in range(32)) for w in range(8)])
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(ptr), C.c_uint32(8), words)
        assert rc == 0, rc
        stream = torch.cuda.ExternalStream(ptr.value)
    else:
        stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        job = bench.Job(args, False, 0, 1, False, False)
        job.prepare(60, 30)
        res = job.timed(60, 30)
        st = job.stats
        print("%s reserve %3d CUs: %.1f Msamples/s, %.4f ms/frame, stages %s" % (config, reserve, res["value"], res["ms_per_step"] / args.spp,
                                                                                {k: round(v, 3) for k, v in st.items() if k not in ("frames", "clear", "prep_infer")}))
        job.close()
