#!/bin/bash
export NRC_HPM_LIB=$PWD/nrc-hpm-renderer_amd/lib_diag/libnrc_hpm.so
mkdir -p gpurun_out/r02g
for cfg in "512 2 2" "256 2 4" "256 2 2" "256 1 4" "256 3 4" "512 1 2" "256 2 3"; do
  set -- $cfg
  NRC_INFER_THREADS=$1 NRC_INFER_NT=$2 NRC_INFER_BPC=$3 timeout -k 10 200 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/r02g/b_$1_$2_$3.json 2>gpurun_out/r02g/b_$1_$2_$3.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('threads %s nt %s bpc %s: %8.1f Msamples/s frame %.4f ms gen %.4f infer %.4f train %.4f prep %.4f | k_infer dense %.4f ms (%.1f%%)' % (sys.argv[2], sys.argv[3], sys.argv[4], d['value'], d['ms_per_frame'], d['stage_ms']['gen_rays'], d['stage_ms']['infer'], d['stage_ms']['train'], d['stage_ms']['prep_train'], d['roofline_mlp']['ms_per_launch'], 100*d['roofline_mlp']['frac']))" gpurun_out/r02g/b_$1_$2_$3.json $1 $2 $3
done
