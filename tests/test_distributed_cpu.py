"""The N > 1 path of bench.py on CPU (world_size-2 and world_size-8 `gloo`): frames are sharded by contiguous ranges with no data-path
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


def test_eight_ranks_shard_configs3_and_gather_their_reports():
    """BASELINE configs[3] as the driver launches it — 8 ranks — on CPU: 16384 frames in 8 shards of 2048, the barrier, the
    max over 8 ranks and the gather of 8 shard reports (each with its device's identity) on rank 0, whose count of distinct
    devices is what lets a SCALE line prove 8 physical GPUs."""
    import bench
    import torch.multiprocessing as mp
    world, total = 8, 16384
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7
    procs = [ctx.Process(target=_worker8, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [g[1:3] for g in got] == [(2048 * r, 2048 * (r + 1)) for r in range(world)]
    assert all(g[3] == 0.5 + 0.25 * 7 for g in got)                      # every rank holds the slowest rank's time
    assert all(g[4] is None for g in got[1:])
    reports = got[0][4]
    assert [s["rank"] for s in reports] == list(range(world)) and [s["frames"] for s in reports] == [[2048 * r, 2048 * (r + 1)] for r in range(world)]
    assert bench.distinct_devices([s["where"] for s in reports]) == 8
    assert bench.distinct_devices([reports[0]["where"]] * 8) == 1        # eight ranks on one GPU would show


def _worker8(rank, world, port, total_frames, q):
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    w, r, lr = bench.world_from_env(world, {"WORLD_SIZE": str(world), "RANK": str(rank), "LOCAL_RANK": str(rank)})
    ranks = bench.Ranks(w, r)
    lo, hi = bench.shard(total_frames, world, rank)
    ranks.barrier()
    dt_max = ranks.max(0.5 + 0.25 * rank)
    where = {"device": lr, "pci_bus_id": "0000:%02x:00.0" % (0x10 + 0x10 * lr), "uuid": "%032x" % lr, "numa_node": lr // 4}
    shards = ranks.gather({"rank": rank, "frames": [lo, hi], "where": where})
    q.put((rank, lo, hi, dt_max, shards))
    ranks.close()


def test_a_rank_binds_itself_next_to_its_gpu():
    """bench.bind_rank_to_device: the library binds the calling thread (ssd_bind_thread_to_device: threads started afterwards
    inherit its mask), bench reports what happened and remembers the mask it started with (the all-cores CPU baseline restores
    it).  The library is played by a stand-in (no GPU here)."""
    import bench

    class FakeSsd:
        def __init__(self, bound):
            self.bound = bound

        def device_info(self, device):
            return {"pci_bus_id": "0000:c5:00.0", "uuid": "ab" * 16, "numa_node": 1, "n_local_cpus": self.bound, "cpu_list": "64-127"}

        def bind_thread_to_device(self, device):
            return self.bound

    before = os.sched_getaffinity(0)
    info = bench.bind_rank_to_device(FakeSsd(64), 3)
    assert info["device"] == 3 and info["bound"] and info["cpus_bound"] == 64 and info["pci_bus_id"] == "0000:c5:00.0"
    assert info["cpus_before"] == len(before) and bench._AFFINITY_AT_START == before
    info = bench.bind_rank_to_device(FakeSsd(0), 0)
    assert not info["bound"] and info["cpus_bound"] == 0
    assert os.sched_getaffinity(0) == before              # the stand-in binds nothing; bench itself never touches the mask


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
