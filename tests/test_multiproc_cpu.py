"""world_size-2 `gloo` test of the N>1 bench contract on CPU: the path shards by replicas (no data-path
collective); the only distributed pieces are the barrier around the timed region and the MAX-over-ranks of the
elapsed time.  bench.timed_region is GPU-only, so its protocol is restated here with CPU work in the middle."""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    # every rank owns an independent replica of the workload: same FLOP / byte counts, different data
    torch.manual_seed(rank)
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))          # rank 1 is the slow one
    dist.barrier()
    wall = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    flops = bench.prefill_flops(bench.PREFILL) * world
    q.put((rank, wall.item(), flops))
    dist.destroy_process_group()


def test_two_rank_timing_contract():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, w0, f0), (r1, w1, f1) = res
    assert abs(w0 - w1) < 1e-9          # both ranks agree on the MAX
    assert w0 >= 0.1                    # and it is the slow rank's time
    assert f0 == f1 == 2 * 4.0 * 48 * 24 * 1024 * 1024 * 128 * 0.5


def test_bench_work_formulas():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.prefill_flops(bench.PREFILL) == 309237645312.0       # 309.24 GFLOP
    assert bench.prefill_bytes(bench.PREFILL) == 1207959552.0         # 1 207.96 MB
    assert bench.decode_bytes(bench.DECODE) == 805601280.0            # 805 601 280 B
