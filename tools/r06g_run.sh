set -x
O=gpurun_out/r06g; mkdir -p $O
summ() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-34s %8.1f Msamples/s  frame %.4f ms  stages %s  sched %s' % (sys.argv[2], d['value'], d['ms_per_frame'], {k: round(v,3) for k,v in d['stage_ms'].items()}, [d['schedule'][k] for k in ('camera_priority_low','cost_order_lag','xcd_window','source')]))" $1 "$2" >> $O/summary.txt; }
for W in 5 4; do
 for cfg in "c2:" "c5:--config c5" "hash:--pos-id 0"; do
  name=${cfg%%:*}; args=${cfg#*:}
  NRC_CAMERA_WAVES=$W timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-quality $args > $O/bench_${name}_w$W.json 2> $O/bench_${name}_w$W.err || { tail -5 $O/bench_${name}_w$W.err; exit 1; }
  summ $O/bench_${name}_w$W.json "$name waves/SIMD $W"
 done
done
# upper bounds on any scheme for the thin tail trips (diagnostic builds, WRONG frames): stand-alone gen_rays
for L in lib lib_cutd2_4 lib_cutd2_64 lib_cutall_4; do
  NRC_HPM_LIB=$PWD/nrc-hpm-renderer_amd/$L/libnrc_hpm.so NRC_DEBUG=single_stream timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --train 0 --no-cpu-baseline --no-quality > $O/alone_$L.json 2> $O/alone_$L.err || { tail -5 $O/alone_$L.err; exit 1; }
  summ $O/alone_$L.json "alone $L"
done
cat $O/summary.txt
echo done
