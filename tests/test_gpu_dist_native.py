"""The product's multi-GPU exchange path on the GPU box: a fresh `python -m torch.distributed.run` job (one rank per GPU; this
box has one) brings up the library's own RCCL communicator (nrc_cache_comm_init), the renderer all-reduces gradient vector + loss
cell with ncclAllReduce on its training stream every frame, and the results equal a run without any communicator bit for bit (a
one-rank all-reduce is the identity); the torch.distributed hook path gives the same bits.  The N > 1 arithmetic of the exchange
(sharded batches against the global normaliser sum to the full-batch gradient, replicas stay identical) is covered on CPU ranks by
tests/test_dist_gloo.py and on one device by test_gpu_mlp.py::test_sharded_backward_sums_to_full_batch."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_native_rccl_exchange_under_torch_distributed_run(torch_gpu, tmp_path):
    out = str(tmp_path / "dist.json")
    port = 29600 + (os.getpid() % 300)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "workers", "dist_native_worker.py"), out],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out + ".0"))
    assert res["world"] == 1 and res["comm"] == [0, 1]          # what ncclCommUserRank / ncclCommCount report
    assert res["comm_hook"] == [0, 0]                            # the hook path has no native communicator
    assert res["finite"] and res["step"] == 5 and len(res["losses"]) == 5
    assert res["native_equals_no_communicator"]
    assert res["native_equals_hook"]
    # the fp16 exchange (nrc_cache_set_exchange_dtype): RCCL's ncclHalf all-reduce gives the bits the hook path's statement of the protocol
    # gives, trains to finite losses within fp16's resolution of the fp32 run, and is a different run (the rounding happened)
    assert res["fp16_native_equals_hook"] and res["fp16_finite"] and res["fp16_differs_from_fp32"]
    assert res["fp16_weight_rel_diff"] < 2e-2 and all(abs(a - b) <= 2e-2 * abs(b) for a, b in zip(res["fp16_losses"], res["losses"]))
    assert res["hashgrid_fp16_finite"] and res["hashgrid_fp16_weight_rel_diff"] < 2e-2
    # HashGrid model: the sparse list exchange is the one in use (and not for the dense run or the model without a table), it
    # trains to the same losses and weights as the dense all-reduce up to the run-to-run noise of the packed-fp16 atomics
    assert res["hashgrid_sparse_flags"] == [True, False, False] and res["hashgrid_finite"]
    ls, ld = res["hashgrid_losses_sparse"], res["hashgrid_losses_dense"]
    assert len(ls) == 5 and all(abs(a - b) <= 2e-2 * abs(b) for a, b in zip(ls, ld)), (ls, ld)
    assert res["hashgrid_weight_rel_diff"] < 1e-2
