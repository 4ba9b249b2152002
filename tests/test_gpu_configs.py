"""BASELINE.json configurations at their full sizes, and the N > 1 launch of bench.py, on the one GPU of the box."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import parity
import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _state_counts(ssd, det, frame):
    raw, lay = det.frame_state(frame)
    hist = np.frombuffer(raw, dtype=np.uint32, count=ssd.MAX_BINS, offset=lay["hist"])
    n_nonzero, n_inrange = np.frombuffer(raw, dtype=np.uint32, count=2, offset=lay["hist"] + 4 * ssd.MAX_BINS)
    return hist, int(n_nonzero), int(n_inrange)


def test_config3_full_batch(ssd, oracle, gpu_device):
    """BASELINE configs[2] (SURVEY.md section 8(d) "config 3") at its full size: 1024 XGA frames resident in HBM (9.66 GB),
    one batch (grid.x = 1024, ~25 k blocks per streaming kernel).  EVERY frame against the oracle (round 4: the oracle on a pool
    of threads, parity.check_batch_against_oracle; round 3 sampled every 64th); the second run
    bitwise equal (workspace left clean by 1024 frames' worth of consumers); histogram mass = in-range count <= non-zero
    count <= W H for all 1024 frames."""
    n, W, H = 1024, 1024, 768
    sc_list = scenes.batch_scenes(ssd, W, H, n, base_seed=100000, rng_seed=1000)      # the frames bench.py times on rank 0
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=n, batches_in_flight=ssd.BATCHES_IN_FLIGHT_THROUGHPUT)
    buf = ssd.DeviceBuffer(W * H * 12 * n, gpu_device)
    ssd.synth_device(sc_list, buf.ptr, device=gpu_device)
    det = ssd.Detector(cfg, trans, gpu_device)
    assert det.batches_in_flight == 3                  # asked for (what bench.py runs): three workspaces used in turn
    det.enqueue(buf.ptr, n)
    r1 = det.fetch_list(n)
    for i in range(n):
        hist, n_nonzero, n_inrange = _state_counts(ssd, det, i)
        assert int(hist.sum()) == n_inrange <= n_nonzero <= W * H, "frame %d" % i
        assert n_inrange > W * H // 4, "frame %d: implausibly few points in range" % i
    for _ in range(3):                                 # the other two workspaces, then the first one again
        det.enqueue(buf.ptr, n)
        r2 = det.fetch_list(n)
        assert [bytes(x) for x in r1] == [bytes(x) for x in r2]
    rep = {}
    assert parity.check_batch_against_oracle(ssd, oracle, cfg, trans.constants, buf, W * H * 12, r1, W, H, report=rep) == 1024
    assert rep["frames_checked"] == 1024
    assert rep.get("max_corner_err", 0.0) == 0.0 and rep.get("max_height_err", 0.0) <= parity.TOL_HEIGHT
    assert sum(1 for r in r1 if r.n_steps >= 3) >= n * 9 // 10
    det.close()
    buf.free()


def test_config4_rank_share(ssd, oracle, gpu_device):
    """BASELINE configs[3] (SURVEY.md section 8(d) "config 4": 16,384 frames over 8 GPUs) at the size ONE rank gets: 2048 XGA
    frames resident in HBM (19.3 GB), max_frames_per_batch = 2048, one batch per call (grid.x = 2048, ~49 k blocks per streaming
    kernel, 3.6 GB per workspace).  The frames are rank 0's of that run (bench.py --gpus 8 --frames 2048).  EVERY frame
    against the oracle (round 4; round 3: every 128th); four calls (every workspace, the first one twice) bitwise equal; histogram mass = in-range count
    <= non-zero count <= W H for all 2048 frames."""
    n, W, H = 2048, 1024, 768
    sc_list = scenes.batch_scenes(ssd, W, H, n, base_seed=100000, rng_seed=1000)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=n, batches_in_flight=ssd.BATCHES_IN_FLIGHT_THROUGHPUT)
    buf = ssd.DeviceBuffer(W * H * 12 * n, gpu_device)
    ssd.synth_device(sc_list, buf.ptr, device=gpu_device)
    det = ssd.Detector(cfg, trans, gpu_device)
    assert det.batches_in_flight == 3
    det.enqueue(buf.ptr, n)
    r1 = det.fetch_list(n)
    for i in range(n):
        hist, n_nonzero, n_inrange = _state_counts(ssd, det, i)
        assert int(hist.sum()) == n_inrange <= n_nonzero <= W * H, "frame %d" % i
        assert n_inrange > W * H // 4, "frame %d: implausibly few points in range" % i
    for _ in range(3):
        det.enqueue(buf.ptr, n)
        r2 = det.fetch_list(n)
        assert [bytes(x) for x in r1] == [bytes(x) for x in r2]
    rep = {}
    assert parity.check_batch_against_oracle(ssd, oracle, cfg, trans.constants, buf, W * H * 12, r1, W, H, report=rep) == 2048
    assert rep["frames_checked"] == 2048
    assert rep.get("max_corner_err", 0.0) == 0.0 and rep.get("max_height_err", 0.0) <= parity.TOL_HEIGHT
    assert sum(1 for r in r1 if r.n_steps >= 3) >= n * 9 // 10
    det.close()
    buf.free()


def test_config5_batch(ssd, oracle, gpu_device):
    """BASELINE configs[4] (SURVEY.md "config 5"): a batch of 64 FHD stress frames (8 noisy steps, 5 % outliers) in HBM; EVERY
    frame against the oracle (round 4; round 3: every 16th), second run bitwise equal, histogram mass = in-range count."""
    n, W, H = 64, 1920, 1080
    sc_list = scenes.fhd_stress_scenes(ssd, n, base_seed=9000)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=n)
    buf = ssd.DeviceBuffer(W * H * 12 * n, gpu_device)
    ssd.synth_device(sc_list, buf.ptr, device=gpu_device)
    det = ssd.Detector(cfg, trans, gpu_device)
    det.enqueue(buf.ptr, n)
    r1 = det.fetch_list(n)
    for i in range(n):
        hist, n_nonzero, n_inrange = _state_counts(ssd, det, i)
        assert int(hist.sum()) == n_inrange <= n_nonzero <= W * H
    det.enqueue(buf.ptr, n)
    r2 = det.fetch_list(n)
    assert [bytes(x) for x in r1] == [bytes(x) for x in r2]
    rep = {}
    assert parity.check_batch_against_oracle(ssd, oracle, cfg, trans.constants, buf, W * H * 12, r1, W, H, chunk=16, report=rep) == 64
    assert rep["frames_checked"] == 64
    assert rep.get("max_corner_err", 0.0) == 0.0 and rep.get("max_height_err", 0.0) <= parity.TOL_HEIGHT
    assert all(5 <= r.n_steps <= 9 for r in r1)      # the reference reports fewer than the 8 built steps (SURVEY.md 8(a) probe note)
    det.close()
    buf.free()


def test_bench_runs_two_ranks_on_this_gpu():
    """`python bench.py --gpus 2` launches its own two ranks (torch.distributed.run, gloo); SSD_BENCH_DEVICE=0 lets both
    use the one GPU of this box.  The line must say n_gpus 2, the shards must be disjoint contiguous frame ranges, and
    each rank must have checked frames of ITS shard against the oracle."""
    env = dict(os.environ, SSD_BENCH_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "64", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["frames_per_gpu_per_step"] == 64
    assert "configs[3]" in d["config"]["workload"]
    ranks = sorted(d["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1] and [r["device"] for r in ranks] == [0, 0]
    assert ranks[0]["frames"] == [0, 64] and ranks[1]["frames"] == [64, 128]
    for r in ranks:
        assert r["parity"]["frames_checked_against_oracle"] == 5
        assert r["parity"]["max_abs_corner_err_m"] == 0.0 and r["parity"]["max_abs_height_err_m"] <= parity.TOL_HEIGHT
        assert r["steps_found"] > 0
    assert d["value"] == pytest.approx(2 * 64 * 2 / (d["ms_per_step"] * 2 * 1e-3), rel=1e-6)
    assert "cpu_baseline" not in d                      # reported at N = 1 only


def test_bench_launcher_with_four_ranks_on_this_gpu():
    """The launcher end to end with more ranks than this box has GPUs: `bench.py --gpus 4 --frames 64`, every rank on device 0
    (SSD_BENCH_DEVICE=0).  Four bindings to the device's socket, four disjoint shards of the global frame range, four shard
    reports with their own oracle checks, `distinct_devices` 1 and the line's own `warning` that this is no 4-GPU measurement.
    (Eight ranks on one card are more processes than this pool lets a job put on a GPU - six, the test runner included;
    world size 8 itself runs on the CPU in tests/test_distributed_cpu.py.)"""
    env = dict(os.environ, SSD_BENCH_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--frames", "64", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["scaling"] == "weak" and d["config"]["frames_per_gpu_per_step"] == 64
    assert d["config"]["workload"].startswith("BASELINE configs[3]: 256-frame batch frame-sharded over 4 GPUs")
    ranks = sorted(d["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1, 2, 3] and [r["device"] for r in ranks] == [0, 0, 0, 0]
    assert [r["frames"] for r in ranks] == [[0, 64], [64, 128], [128, 192], [192, 256]]
    assert len(d["devices"]) == 4 and d["distinct_devices"] == 1
    assert "not an 4-GPU measurement" in d["warning"] and "SSD_BENCH_DEVICE" in d["warning"]
    for r in ranks:
        assert r["where"]["pci_bus_id"] == ranks[0]["where"]["pci_bus_id"]
        assert r["parity"]["frames_checked_against_oracle"] == 5
        assert r["parity"]["max_abs_corner_err_m"] == 0.0 and r["parity"]["max_abs_height_err_m"] <= parity.TOL_HEIGHT
        assert r["steps_found"] > 0
    assert d["value"] == pytest.approx(4 * 64 * 2 / (d["ms_per_step"] * 2 * 1e-3), rel=1e-6)


def test_bench_config4_two_ranks_at_their_full_share_on_this_gpu():
    """BASELINE configs[3] as far as one GPU can carry it: 2 of the 8 ranks, each with its full share of 2048 XGA frames
    (19.3 GB of frames + 3 x 3.6 GB of workspaces per rank, both on this one device via SSD_BENCH_DEVICE=0).  Frame ranges
    [0, 2048) and [2048, 4096) of the global index space, each rank's own frames checked against the oracle, per-rank stage
    times and K1's fraction of the HBM peak in the line, the workload named as configs[3]."""
    env = dict(os.environ, SSD_BENCH_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "2048", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["frames_per_gpu_per_step"] == 2048
    assert d["config"]["workload"].startswith("BASELINE configs[3]: 4096-frame batch frame-sharded over 2 GPUs")
    assert d["config"]["batches_in_flight"] == 3
    ranks = sorted(d["ranks"], key=lambda r: r["rank"])
    assert [r["frames"] for r in ranks] == [[0, 2048], [2048, 4096]] and [r["device"] for r in ranks] == [0, 0]
    for r in ranks:
        assert r["parity"]["frames_checked_against_oracle"] == 5
        assert r["parity"]["max_abs_corner_err_m"] == 0.0 and r["parity"]["max_abs_height_err_m"] <= parity.TOL_HEIGHT
        assert r["steps_found"] >= 2048 * 3                       # ground + >= 3 steps in nearly every frame
        assert set(r["stage_ms"]) == {"hist", "peaks", "raster", "outline", "quads", "inquad", "final"}
        assert all(v > 0.0 for v in r["stage_ms"].values()) and 0.0 < r["k1_frac_of_hbm_peak"] < 1.0
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["algorithmic_bytes_per_launch"] == 12.0 * 1024 * 768 * 2048
    assert d["value"] == pytest.approx(2 * 2048 * 3 / (d["ms_per_step"] * 3 * 1e-3), rel=1e-6)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):                                         # kept for profiles/ (the builder copies it)
        with open(os.path.join(out, "config4_two_ranks_one_gpu.json"), "w") as f:
            f.write(lines[0] + "\n")


def test_bench_refuses_two_ranks_without_two_gpus():
    """without the override, 2 ranks on a 1-GPU box must fail loudly instead of printing a line"""
    import importlib
    ssd = importlib.import_module("stair-step-detector_amd")
    if ssd.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SSD_BENCH_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "8", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode != 0 and "n_gpus" not in p.stdout


def test_native_driver_shards_frames_over_device_threads(ssd):
    """detect-stairs-amd --device-list: one host thread + one handle per listed device, contiguous frame ranges, frames kept in
    HBM; the lines must come out in frame order and equal the single-handle run's.  Three
    shards on the one GPU of this box (a device may be listed more than once)."""
    exe = os.path.join(os.path.dirname(ssd.LIB_PATH), "detect-stairs-amd")
    args = [exe, "--width", "640", "--height", "480", "--frames", "10", "--steps", "3", "--seed", "99"]
    single = subprocess.run(args, check=True, capture_output=True, text=True, timeout=600).stdout.splitlines()
    sharded = subprocess.run(args + ["--device-list", "0,0,0"], check=True, capture_output=True, text=True, timeout=600)
    assert sharded.stdout.splitlines() == single and len(single) == 10
    assert "3 device shard(s)" in sharded.stderr
    assert all(l.startswith('["stairs",["stairSteps",4]') for l in single)
    assert sharded.stderr.count("resident frames x 1 pass(es)") == 3 and "generation outside the timed region" in sharded.stderr
    # more shards than frames (empty shards are skipped), and several passes over the resident frames
    few = subprocess.run(args[:6] + ["2"] + args[7:] + ["--device-list", "0,0,0", "--passes", "3"], check=True, capture_output=True, text=True, timeout=600)
    assert few.stdout.splitlines() == single[:2] and few.stderr.count("resident frames x 3 pass(es)") == 2
    bad = subprocess.run(args + ["--device-list", "0,7"], capture_output=True, text=True, timeout=600)
    if ssd.device_count() < 8:
        assert bad.returncode != 0 and "not present" in bad.stderr


def test_host_fed_ingest_slices_pinned_and_pageable(ssd, oracle, gpu_device):
    """ssd_process_host / ssd_process_depth_host on 70 VGA frames = three ingest slices (32 + 32 + 6) over two staging
    buffers, from pinned (ssd_host_alloc) and from pageable memory, as vertices and as 16-bit depth: every result equals the
    single-frame call's bytewise, and every 10th frame the oracle's."""
    n, W, H = 70, 640, 480
    sc_list = scenes.batch_scenes(ssd, W, H, n, base_seed=31000, rng_seed=5)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=64)
    det = ssd.Detector(cfg, trans, gpu_device)
    intr = ssd.intrinsics_for_scene(sc_list[0])
    det.set_intrinsics(intr)
    xyz = ssd.synth_host(sc_list)
    depth = ssd.synth_depth_host(sc_list)
    pin_x = ssd.PinnedArray(xyz.shape, np.float32)
    pin_x.array[...] = xyz
    pin_d = ssd.PinnedArray(depth.shape, np.uint16)
    pin_d.array[...] = depth
    ref_x = [det.process_host(xyz[i])[0] for i in range(n)]
    ref_d = [det.process_depth_host(depth[i])[0] for i in range(n)]
    for arr, run, ref in ((xyz, det.process_host, ref_x), (pin_x.array, det.process_host, ref_x),
                          (depth, det.process_depth_host, ref_d), (pin_d.array, det.process_depth_host, ref_d)):
        for _ in range(2):                                     # twice: the staging buffers and the workspace are reused
            got = run(arr)
            assert len(got) == n
            assert [bytes(g) for g in got] == [bytes(r) for r in ref]
    for i in range(0, n, 10):
        parity.check_results_only(ssd, oracle, cfg, trans.constants, xyz[i], ref_x[i])
        parity.check_results_only(ssd, oracle, cfg, trans.constants, oracle.deproject(intr, depth[i]), ref_d[i])
    # the same through a handle with three workspaces: the slices run on the handle's own streams and overlap, a staging buffer
    # is released by the end of the slice that read it (an event on that slice's stream), not by the ingest stream's position
    det3 = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=128, batches_in_flight=3), trans, gpu_device)
    assert det3.batches_in_flight == 3
    det3.set_intrinsics(intr)
    for _ in range(2):
        assert [bytes(g) for g in det3.process_host(pin_x.array)] == [bytes(r) for r in ref_x]
        assert [bytes(g) for g in det3.process_depth_host(depth)] == [bytes(r) for r in ref_d]
    det3.close()
    pin_x.free()
    pin_d.free()
    det.close()


def test_calls_on_different_streams_are_ordered_by_the_library(ssd, oracle, gpu_device):
    """A handle's workspace is single-buffered: an enqueue on one stream followed at once by work on another
    (a second user stream, then ssd_process_host's own ingest streams, then the default stream) must see its predecessor
    finished — the library inserts the event wait.  Every result must equal the oracle's."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")          # the runtime libssd_hip.so already runs on (torch would bring its own copy into this process)
    streams = []
    for _ in range(2):
        st = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0          # hipStreamNonBlocking
        streams.append(st)
    n, W, H = 24, 640, 480
    sc_list = scenes.batch_scenes(ssd, W, H, 2 * n, base_seed=52000, rng_seed=9)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=n, batches_in_flight=1)        # one workspace: strict stream order is the contract
    host = ssd.synth_host(sc_list)
    a = ssd.DeviceBuffer(W * H * 12 * n, gpu_device)
    b = ssd.DeviceBuffer(W * H * 12 * n, gpu_device)
    a.upload(host[:n])
    b.upload(host[n:])
    det = ssd.Detector(cfg, trans, gpu_device)
    for rep in range(3):
        det.enqueue(a.ptr, n, stream=streams[0].value)
        det.enqueue(b.ptr, n, stream=streams[1].value)         # other stream, no host synchronisation in between
        rb = det.fetch_list(n)
        ra = [ssd.FrameResult.from_buffer_copy(r) for r in det.fetch(n, back=1)]
        rh = det.process_host(host[:n])                         # the ingest streams
        det.enqueue(b.ptr, n)                                   # the default stream
        rb2 = det.fetch_list(n)
        for i in range(0, n, 5):
            parity.check_results_only(ssd, oracle, cfg, trans.constants, host[i], ra[i])
            parity.check_results_only(ssd, oracle, cfg, trans.constants, host[n + i], rb[i])
        assert [bytes(x) for x in rh] == [bytes(x) for x in ra]
        assert [bytes(x) for x in rb2] == [bytes(x) for x in rb]
    det.close()
    a.free()
    b.free()
    for st in streams:
        hip.hipStreamDestroy(st)


def test_default_handle_orders_a_refill_of_its_frames_behind_the_batch(ssd, oracle, gpu_device):
    """a handle created with the DEFAULT configuration (ssd_default_config: batches_in_flight = 0)
    has one workspace and runs in strict stream order, whatever its batch size — so `enqueue, then overwrite the same frame
    buffer on the same stream` (here the NULL stream: ssd_device_upload is a plain hipMemcpy) needs no fetch in between.
    Round 3 resolved 0 to three workspaces on streams of the handle's own for batches of >= 16 frames, and exactly this
    raced.  Every result must be the FIRST contents' (oracle), five refills in a row."""
    n, W, H = 32, 640, 480
    sc_list = scenes.batch_scenes(ssd, W, H, 6 * n, base_seed=77000, rng_seed=3)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(W, H)                      # as ssd_default_config leaves it
    assert cfg.batches_in_flight == 0 and cfg.max_frames_per_batch >= n
    det = ssd.Detector(cfg, trans, gpu_device)
    assert det.batches_in_flight == 1
    host = ssd.synth_host(sc_list)
    buf = ssd.DeviceBuffer(W * H * 12 * n, gpu_device)
    buf.upload(host[:n])
    for r in range(5):
        det.enqueue(buf.ptr, n)                         # NULL stream
        buf.upload(host[(r + 1) * n:(r + 2) * n])       # refill at once: ordered behind the batch by the stream
        got = det.fetch_list(n)
        for i in range(n):
            parity.check_results_only(ssd, oracle, cfg, trans.constants, host[r * n + i], got[i])
    det.close()
    buf.free()


def test_device_identity_and_local_cpus(ssd, gpu_device):
    """ssd_device_info_get / ssd_bind_thread_to_device (8-GPU readiness): the PCI bus id has sysfs's
    spelling, the UUID is there, and binding leaves the thread on a non-empty subset of the CPUs it had."""
    import re
    info = ssd.device_info(gpu_device)
    assert re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.[0-9a-f]", info["pci_bus_id"]), info
    assert len(info["uuid"]) in (0, 32)
    before = os.sched_getaffinity(0)
    try:
        bound = ssd.bind_thread_to_device(gpu_device)
        after = os.sched_getaffinity(0)
        assert after and after <= before
        assert bound == 0 and after == before or bound == len(after)
        if info["n_local_cpus"] > 0 and info["numa_node"] >= 0:
            assert os.path.isdir("/sys/devices/system/node/node%d" % info["numa_node"])
    finally:
        os.sched_setaffinity(0, before)


def test_handle_keeps_three_batches_in_flight(ssd, oracle, gpu_device):
    """ssd_config::batches_in_flight: a handle with three workspaces takes them in turn on streams of its own.  Nine batches of
    different frames and sizes are enqueued two ahead of the fetches (ssd_fetch_back(back = 2)); in between, calls that the
    library holds in the first workspace — debug capture, the riser pass, a partial run (ssd_enqueue_stages) — are mixed in
    without any host synchronisation.  Every batch must equal, bytewise, what a one-workspace handle (strict stream order)
    returns for it, risers and debug records included; every 9th frame the oracle's.  ssd_stream_wait orders a producer
    that overwrites a batch's frames behind that batch."""
    import ctypes as C
    W, H, cap = 640, 480, 40
    sizes = [40, 17, 40, 33, 40, 8, 40, 25, 40]
    sc_list = scenes.batch_scenes(ssd, W, H, sum(sizes), base_seed=71000, rng_seed=13)
    trans = ssd.transformation_for_scene(sc_list[0])
    fb = W * H * 12
    buf = ssd.DeviceBuffer(fb * sum(sizes), gpu_device)
    ssd.synth_device(sc_list, buf.ptr, device=gpu_device)
    host = ssd.synth_host(sc_list)
    starts = [sum(sizes[:i]) for i in range(len(sizes))]
    one = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=cap, batches_in_flight=1), trans, gpu_device)
    assert one.batches_in_flight == 1
    want = []
    for at, n in zip(starts, sizes):
        one.enqueue(buf.ptr + at * fb, n)
        want.append([bytes(r) for r in one.fetch(n)])
    one_ws = one.workspace_bytes
    one.set_risers(True, 0.03, 200)
    one.enqueue(buf.ptr + starts[3] * fb, sizes[3])
    one.fetch(sizes[3])
    want_risers = [bytes(r) for r in one.fetch_risers(sizes[3])]
    one.set_risers(False)
    one.set_debug(True)
    one.enqueue(buf.ptr + starts[5] * fb, sizes[5])
    one.fetch(sizes[5])
    want_dbg = bytes(one.debug(sizes[5] - 1))
    one.close()

    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=cap, batches_in_flight=3), trans, gpu_device)
    assert det.batches_in_flight == 3 and det.workspace_bytes > 2.5 * one_ws
    for rep in range(2):
        got = [None] * len(sizes)
        for i, (at, n) in enumerate(zip(starts, sizes)):
            det.enqueue(buf.ptr + at * fb, n)
            if i >= 2:
                got[i - 2] = [bytes(r) for r in det.fetch(sizes[i - 2], back=2)]
        got[-2] = [bytes(r) for r in det.fetch(sizes[-2], back=1)]
        got[-1] = [bytes(r) for r in det.fetch(sizes[-1], back=0)]
        assert got == want, "round %d" % rep
        # calls held in the first workspace, straight behind free-running ones and followed by free-running ones
        det.enqueue(buf.ptr + starts[0] * fb, sizes[0])
        det.enqueue(buf.ptr + starts[1] * fb, sizes[1])
        det.set_risers(True, 0.03, 200)
        det.enqueue(buf.ptr + starts[3] * fb, sizes[3])
        det.set_risers(False)
        det.enqueue(buf.ptr + starts[2] * fb, sizes[2])
        assert [bytes(r) for r in det.fetch(sizes[3], back=1)] == want[3]
        assert [bytes(r) for r in det.fetch(sizes[2], back=0)] == want[2]
        assert [bytes(r) for r in det.fetch(sizes[1], back=2)] == want[1]
        det.set_risers(True, 0.03, 200)
        det.enqueue(buf.ptr + starts[3] * fb, sizes[3])
        assert [bytes(r) for r in det.fetch_risers(sizes[3])] == want_risers
        det.set_risers(False)
        det.enqueue(buf.ptr + starts[4] * fb, sizes[4])
        det.set_debug(True)
        det.enqueue(buf.ptr + starts[5] * fb, sizes[5])
        assert [bytes(r) for r in det.fetch(sizes[5])] == want[5]
        assert bytes(det.debug(sizes[5] - 1)) == want_dbg
        det.set_debug(False)
        det.enqueue(buf.ptr + starts[6] * fb, sizes[6], stages=ssd.STAGE_HIST)          # K1 alone: leaves accumulators behind
        det.enqueue(buf.ptr + starts[7] * fb, sizes[7])
        det.enqueue(buf.ptr + starts[8] * fb, sizes[8])
        assert [bytes(r) for r in det.fetch(sizes[7], back=1)] == want[7]
        assert [bytes(r) for r in det.fetch(sizes[8], back=0)] == want[8]
    flat = [r for w in want for r in w]
    for i in range(0, len(flat), 9):
        parity.check_results_only(ssd, oracle, det.cfg, trans.constants, host[i], ssd.FrameResult.from_buffer_copy(flat[i]))
    # other workspace counts, the largest included: as many batches ahead of the fetches as there are workspaces
    for depth in (2, 5, 8):
        d = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=cap, batches_in_flight=depth), trans, gpu_device)
        assert d.batches_in_flight == depth
        ahead = depth - 1
        order = list(range(len(sizes))) * 2
        got = {}
        for k, i in enumerate(order):
            d.enqueue(buf.ptr + starts[i] * fb, sizes[i])
            if k >= ahead:
                j = order[k - ahead]
                got[k - ahead] = (j, [bytes(r) for r in d.fetch(sizes[j], back=ahead)])
        for back in range(ahead - 1, -1, -1):
            j = order[len(order) - 1 - back]
            got[len(order) - 1 - back] = (j, [bytes(r) for r in d.fetch(sizes[j], back=back)])
        assert len(got) == len(order) and all(res == want[j] for j, res in got.values()), "depth %d" % depth
        with pytest.raises(ssd.SsdError, match="no enqueue at that position"):
            d.fetch(1, back=max(2, depth))
        d.close()
    # a producer that recycles a batch's frame buffer: ordered behind the batch by ssd_stream_wait, no host synchronisation
    hip = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
    scratch = ssd.DeviceBuffer(fb * cap, gpu_device)
    for rep in range(3):
        ssd.synth_device(sc_list[starts[0]:starts[0] + cap], scratch.ptr, device=gpu_device, stream=st.value)
        det.enqueue(scratch.ptr, cap, stream=st.value)
        det.stream_wait(0, st.value)
        assert hip.hipMemsetAsync(C.c_void_p(scratch.ptr), 0, C.c_size_t(fb * cap), st) == 0       # the "next producer"
        assert [bytes(r) for r in det.fetch(cap)] == want[0]
    hip.hipStreamDestroy(st)
    scratch.free()
    det.close()
    buf.free()


def test_pipeline_overlaps_batches_and_returns_them_in_order(ssd, oracle, gpu_device):
    """ssd_pipeline_*: two handles on two streams, batches dealt out round-robin.  Seven batches of different sizes through a
    depth-2 pipeline come back in submission order, each bytewise what a single handle returns; a third unfetched submit is
    refused (SSD_E_CAP) and every 7th frame equals the oracle's."""
    W, H = 640, 480
    sizes = [12, 5, 12, 1, 9, 12, 3]
    sc_list = scenes.batch_scenes(ssd, W, H, sum(sizes), base_seed=61000, rng_seed=11)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=12)
    fb = W * H * 12
    buf = ssd.DeviceBuffer(fb * sum(sizes), gpu_device)
    ssd.synth_device(sc_list, buf.ptr, device=gpu_device)
    host = ssd.synth_host(sc_list)
    det = ssd.Detector(cfg, trans, gpu_device)
    want, at = [], 0
    for n in sizes:
        det.enqueue(buf.ptr + at * fb, n)
        want.append(det.fetch_list(n))
        at += n
    det.close()
    pipe = ssd.Pipeline(cfg, trans, gpu_device, depth=2)
    got, at = [], 0
    for i, n in enumerate(sizes):
        if pipe.pending() == 2:
            got.append(pipe.next())
        pipe.submit(buf.ptr + at * fb, n, after_stream=None if i % 2 else False)     # alternately behind the default stream
        at += n
    with pytest.raises(ssd.SsdError, match="unfetched"):
        pipe.submit(buf.ptr, 1)                                    # both handles hold a batch
    while pipe.pending():
        got.append(pipe.next())
    with pytest.raises(ssd.SsdError, match="nothing submitted"):
        pipe.next()
    pipe.close()
    assert [len(g) for g in got] == sizes
    for g, w in zip(got, want):
        assert [bytes(x) for x in g] == [bytes(x) for x in w]
    flat = [r for g in got for r in g]
    for i in range(0, len(flat), 7):
        parity.check_results_only(ssd, oracle, cfg, trans.constants, host[i], flat[i])
    buf.free()


def test_pipeline_stage_times_belong_to_the_batch_fetched_last(ssd, gpu_device):
    """ssd_pipeline_set_timing / ssd_pipeline_stage_times: seven positive stage times for the batch ssd_pipeline_next returned
    last; refused before anything was fetched and once that batch's handle has been given the next one; results unchanged by
    the events."""
    W, H = 640, 480
    sc_list = scenes.batch_scenes(ssd, W, H, 6, base_seed=62000, rng_seed=12)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=6)
    buf = ssd.DeviceBuffer(W * H * 12 * 6, gpu_device)
    ssd.synth_device(sc_list, buf.ptr, device=gpu_device)
    plain = ssd.Pipeline(cfg, trans, gpu_device, depth=2)
    plain.submit(buf.ptr, 6)
    want = plain.next()
    plain.close()
    pipe = ssd.Pipeline(cfg, trans, gpu_device, depth=2)
    pipe.set_timing(True)
    with pytest.raises(ssd.SsdError, match="no fetched batch"):
        pipe.stage_times_ms()
    pipe.submit(buf.ptr, 6)
    pipe.submit(buf.ptr, 3)
    got = pipe.next()
    st = pipe.stage_times_ms()
    assert list(st) == list(ssd.STAGE_NAMES) and all(v > 0.0 for v in st.values()), st
    assert [bytes(r) for r in got] == [bytes(r) for r in want]
    pipe.submit(buf.ptr, 2)                                          # goes to the handle whose batch was fetched last
    with pytest.raises(ssd.SsdError, match="no fetched batch"):
        pipe.stage_times_ms()
    while pipe.pending():
        pipe.next()
    assert all(v > 0.0 for v in pipe.stage_times_ms().values())
    pipe.close()
    buf.free()
