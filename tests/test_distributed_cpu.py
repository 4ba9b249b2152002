"""The N > 1 path of bench.py on CPU (world_size-2 `gloo`): frames are sharded by contiguous ranges with no data-path
collective; the only communication is bench.Ranks — a barrier, the max-over-ranks of the elapsed time and the gather
of the per-rank shard reports — and it is exercised here exactly as bench.main() uses it.  The launcher logic (who
spawns the ranks, what a world-size mismatch does) is tested without a GPU as well."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, total_frames, q):
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    w, r, lr = bench.world_from_env(world, {"WORLD_SIZE": str(world), "RANK": str(rank), "LOCAL_RANK": str(rank)})
    assert (w, r, lr) == (world, rank, rank)
    ranks = bench.Ranks(w, r)                       # gloo, CPU tensors: what bench.main() builds
    lo, hi = bench.shard(total_frames, world, rank)
    ranks.barrier()
    dt_max = ranks.max(0.5 + 0.25 * rank)           # rank 1 is the slow one
    shards = ranks.gather({"rank": rank, "frames": [lo, hi]})
    q.put((rank, lo, hi, dt_max, shards))
    ranks.close()


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
    (r0, lo0, hi0, dt0, sh0), (r1, lo1, hi1, dt1, sh1) = got
    assert lo0 == 0 and hi0 == lo1 and hi1 == total              # contiguous, disjoint, complete
    assert abs((hi0 - lo0) - (hi1 - lo1)) <= 1
    assert dt0 == dt1 == 0.75                                    # max over ranks
    assert sh1 is None                                           # only rank 0 receives the reports
    assert [s["rank"] for s in sh0] == [0, 1] and sh0[0]["frames"] == [lo0, hi0] and sh0[1]["frames"] == [lo1, hi1]


def test_shard_covers_everything_for_any_world():
    import bench
    for world in (1, 2, 3, 4, 8):
        for total in (8, 1024, 16384, 1000):
            ranges = [bench.shard(total, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in ranges) - min(h - l for l, h in ranges) <= 1
    # BASELINE configs[3]: 16384 frames over 8 GPUs = 2048 per GPU
    assert [bench.shard(16384, 8, r) for r in (0, 7)] == [(0, 2048), (14336, 16384)]


def test_launch_command_is_the_drivers_command():
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--frames", "2048"], port=29777)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "8", "--frames", "2048"]


def test_world_size_mismatch_and_missing_devices_are_errors():
    import bench
    with pytest.raises(SystemExit) as e:
        bench.world_from_env(8, {"WORLD_SIZE": "1", "RANK": "0"})          # round 1 silently ran 1 GPU here
    assert "WORLD_SIZE=1" in str(e.value)
    with pytest.raises(SystemExit):
        bench.world_from_env(1, {"WORLD_SIZE": "2", "RANK": "0"})
    assert bench.world_from_env(2, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}) == (2, 1, 1)
    assert bench.device_for_rank(3, 8, {}) == 3
    with pytest.raises(SystemExit) as e:
        bench.device_for_rank(1, 1, {})                                       # 2 ranks, 1 GPU: no oversubscription
    assert "1 GPU(s) are visible" in str(e.value)
    assert bench.device_for_rank(1, 1, {"SSD_BENCH_DEVICE": "0"}) == 0       # unless asked for (the 2-rank GPU test)


def test_bench_refuses_a_foreign_world_size_before_touching_the_gpu():
    """`bench.py --gpus 2` inside a 1-rank environment must not print a 1-GPU line"""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout) and "n_gpus" not in p.stdout
