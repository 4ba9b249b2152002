"""world_size-2 `gloo` test of the N > 1 path of bench.py on CPU: frames are sharded by contiguous ranges with
no data-path collective; the only communication is the barrier and the max-over-ranks of the elapsed time."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, total_frames, q):
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.shard(total_frames, world, rank)
    # every rank "processes" its own frames: here it just records which global frame indices it owns
    owned = torch.zeros(total_frames, dtype=torch.int64)
    owned[lo:hi] = 1
    dt = torch.tensor([0.5 + 0.25 * rank], dtype=torch.float64)   # rank 1 is the slow one
    dist.barrier()
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dist.all_reduce(owned, op=dist.ReduceOp.SUM)                   # test-only: proves the shards partition the range
    q.put((rank, lo, hi, float(dt.item()), owned.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [2048, 17])
def test_frame_sharding_and_max_time_over_two_ranks(total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (0 if total == 2048 else 1)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, dt0, owned0), (r1, lo1, hi1, dt1, owned1) = got
    assert lo0 == 0 and hi0 == lo1 and hi1 == total              # contiguous, disjoint, complete
    assert abs((hi0 - lo0) - (hi1 - lo1)) <= 1
    assert dt0 == dt1 == 0.75                                    # max over ranks
    assert owned0 == [1] * total and owned1 == [1] * total


def test_shard_covers_everything_for_any_world():
    import bench
    for world in (1, 2, 3, 4, 8):
        for total in (8, 1024, 16384, 1000):
            ranges = [bench.shard(total, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in ranges) - min(h - l for l, h in ranges) <= 1
