"""bench.py's N > 1 code path on one GPU: SE2_BENCH_FORCE_DIST=1 makes the single rank create an RCCL communicator
(se2gpu_comm_*), shard the landmarks (rank 0 of 1) and all-reduce the packed lower triangle of [S | b | scalars] through
RCCL inside every LM trial - exactly what every rank of `torch.distributed.run --nproc-per-node N bench.py --gpus N` does.
The 8-GPU run itself is the driver's; this keeps the path from rotting between rounds."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_sharded_path_with_a_one_rank_rccl_communicator(synth):
    from se2lam_amd.optimizer import SlamOptimizer
    env = dict(os.environ)
    env.update({"SE2_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29611"})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "10", "--kf", "50",
           "--landmarks", "5000", "--no-orb", "--no-cpu-baseline", "--ba-windows", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # stdout carries the JSON line and nothing else (RCCL's banner goes to stderr)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["unit"] == "iters/s" and d["value"] > 0 and d["steps"] >= 20
    assert "RCCL" in d["config"]["parallelism"] or "single" in d["config"]["parallelism"]
    # the exchange is broken out: the packed lower triangle, timed in the per-kernel pass
    ex = d["exchange"]
    from se2lam_amd import capi
    # (the packed triangle of the system the handle factorises: the natural order's size, or a little more when the poses
    # were re-ordered into padded partitions - always less than the rectangle [S; b^T])
    assert capi.lib().se2gpu_ba_exchange_doubles(50) <= ex["doubles_per_trial"] < (3 * 50 + 1) * 160
    assert ex["comm_ranks"] == 1                   # ncclCommCount of the communicator bench.py created
    assert d["timed_s"] >= 0.1                      # the timed-region floor is enforced, whatever --steps says
    assert ex["allreduce_system_us"] > 0 and ex["pack_unpack_us"] > 0 and ex["allreduce_us_per_iteration"] > 0
    # the same window without the communicator: identical LM run (one rank: the all-reduce is the identity)
    o = SlamOptimizer()
    o.load(synth.ba_graph(50, 5000))
    o.initializeOptimization(0)
    o.optimize(10)
    assert np.isclose(d["config"]["chi2_final"], o.stats["chi2_final"], rtol=1e-9)
