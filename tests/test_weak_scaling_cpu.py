"""The WEAK form of the multi-GPU BA metric on CPU, world_size 2: every rank optimises ITS OWN windows, there
is no data-path collective, the job's figure is all ranks' LM iterations over the slowest rank's time - the form that can scale with
the number of GPUs (north_star: "independent keyframe windows shard across the 8 GPUs"), as `bench.py --gpus N` reports it in
`value_weak`.  No GPU here and the library has no CPU fallback, so a rank's windows are optimised by the oracle; what is exercised is
what the N > 1 run relies on besides the device: se2lam_amd/rendezvous.py (the stdlib TCP rendezvous bench.py uses instead of
torch.distributed - see its header - with broadcast, barrier, max and sum over ranks), the dealing of distinct windows to ranks, and
the aggregation rule.  The strong form (one window, landmark shards, all-reduce of the reduced system) is tests/test_distributed_cpu.py."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import time
    from oracle import oracle
    from se2lam_amd import synth
    from se2lam_amd.rendezvous import Rendezvous
    rv = Rendezvous(rank, world, "127.0.0.1", port)
    token = rv.broadcast(bytes(range(128)) if rank == 0 else None, 128)          # what carries the ncclUniqueId in the strong form
    windows = synth.mixed_windows(6, p_range=(8, 14), l_range=(60, 160), kidnapped_every=5, seed=31)
    mine = windows[rank::world]                                                    # distinct windows, dealt round-robin
    rv.barrier()
    t0 = time.perf_counter()
    its, chi = 0, []
    for g in mine:
        _, _, st = oracle.ba_optimize(g, 5, 0)
        its += st["iterations"]
        chi.append(st["chi2_final"])
    dt = time.perf_counter() - t0
    dt_max = rv.allreduce_max(dt)
    total = rv.allreduce_sum(float(its))
    rv.barrier()
    rv.close()
    q.put((rank, token == bytes(range(128)), its, dt, dt_max, total, chi))


def test_weak_form_two_ranks_own_windows_no_collective():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [True, True]
    assert got[0][4] == got[1][4] == max(got[0][3], got[1][3])                     # every rank holds the slowest rank's time
    assert got[0][5] == got[1][5] == got[0][2] + got[1][2]                         # ... and the job's iteration count
    # the windows a rank optimised are its own: together the two ranks covered every window exactly once, with the results of a
    # single process doing all of them
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from se2lam_amd import synth
    windows = synth.mixed_windows(6, p_range=(8, 14), l_range=(60, 160), kidnapped_every=5, seed=31)
    want = [oracle.ba_optimize(g, 5, 0)[2]["chi2_final"] for g in windows]
    assert np.allclose(got[0][6], want[0::2], rtol=1e-12) and np.allclose(got[1][6], want[1::2], rtol=1e-12)
