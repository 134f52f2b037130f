#!/bin/bash
# HashGrid model under a one-rank torch.distributed.run job (the native RCCL exchange path): list exchange of the table gradient
# against the dense all-reduce (NRC_DEBUG=dense_grid_exchange), and no communicator at all.   tools/ab_grid_exchange.sh <tag>
TAG=${1:-grid}; OUT=gpurun_out/$TAG; mkdir -p $OUT
show() { python3 -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('%-28s %7.1f Msamples/s frame %.4f' % (sys.argv[2], d['value'], d['ms_per_frame']), {k: round(v,3) for k,v in d['stage_ms'].items()}, d.get('exchange'))" $1 "$2"; }
ARGS="bench.py --gpus 1 --pos-id 0 --steps 30 --warmup 5 --no-cpu-baseline"
timeout -k 10 250 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 $ARGS > $OUT/sparse.json 2> $OUT/sparse.err || { tail -5 $OUT/sparse.err; exit 1; }
show $OUT/sparse.json "lists (default)"
NRC_DEBUG=dense_grid_exchange timeout -k 10 250 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29562 $ARGS > $OUT/dense.json 2> $OUT/dense.err || { tail -5 $OUT/dense.err; exit 1; }
show $OUT/dense.json "dense all-reduce"
timeout -k 10 250 python3 $ARGS > $OUT/none.json 2> $OUT/none.err || { tail -5 $OUT/none.err; exit 1; }
show $OUT/none.json "no communicator"
