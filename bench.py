#!/usr/bin/env python3
"""bench.py — frames/s of the per-frame point-cloud path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (N = 1: BASELINE.json configs[2]): a batch of F = 1024 synthetic 1024x768 frames (3-step staircases
with randomised rise / tread / yaw / noise), resident in HBM when the timed region starts.  One "step" = one
pass of the whole path (K1 hist .. K5 final + results to the host) over the batch.  For N > 1 every rank owns
its own F frames on its own GPU (frames are independent: no collective on the data path; weak scaling) and
value = N * F * K / max-over-ranks time.

The timed steps go through the plain handle API (ssd_enqueue / ssd_fetch_back) of a handle asked for
ssd_config::batches_in_flight = 3 (SSD_BATCHES_IN_FLIGHT_THROUGHPUT; opt-in since round 4 — the library's default is one
workspace in strict stream order): the bench enqueues ahead of its fetches and never touches its frames, so the
steps overlap, so an event-bracketed stage inside them is no kernel duration.  `stage_ms` and `roofline` therefore come
from a few EXTRA steps after the timed region, run one at a time with HIP events between the launches
("stage_ms_source": "separate timed steps"), next to a plain read stream over the same buffer (`k1_over_plain_stream`).

PyTorch is plumbing here: device memory, the stream, the barrier and the max-reduce (torch.distributed over
`gloo` on CPU tensors: the data path has no exchange step, so RCCL is never loaded).  The hot path is
libssd_hip.so through its C ABI.  The CPU oracle is used only for the cpu_baseline leg and the parity
spot check, both outside the timed region.

`python bench.py --gpus N` with N > 1 and no RANK in the environment launches the N ranks itself (a child
`python -m torch.distributed.run ... bench.py --gpus N ...`, started before this process touches the GPU) and
exits with the child's code.  A world size that differs from --gpus, or fewer visible GPUs than ranks, is an
error, never a silent 1-GPU run.  BASELINE configs[3] (16384 frames over 8 GPUs): `--gpus 8 --frames 2048`.
SSD_BENCH_DEVICE=<index> puts every rank on that one device (tests: 2 ranks on a 1-GPU box).

Every rank binds itself to the CPUs of its GPU's NUMA node (ssd_bind_thread_to_device: PCI bus id of the HIP device ->
/sys/bus/pci/devices/<id>/local_cpulist) as its first GPU-runtime call, before torch initialises the device and before any
stream, pinned buffer or helper thread of the run exists, and reports the device's PCI bus id / UUID / NUMA node: the line of an
N-GPU run lists N distinct physical GPUs (`devices`, `distinct_devices`; fewer than N is flagged in the line as `warning`).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec; ~6.3 TB/s achievable by a float4 copy)
VALU_CYCLES = 4.5         # SIMD cycles per wave64 vector instruction of this path, measured (tools/instr_rate.hip, profiles/r05_instr_rate.txt:
                          # fp64 add / mul / fma 4.6-4.7, conversions and compares 4.4-4.5, DPP 4.5, v_pk_fma_f32 4.6; only plain f32 add / mul,
                          # 32-bit add / and and v_mov run at 2.4-2.5)
METRIC = "point-cloud frames/sec (1024×768 pts) at 1/2/4/8 GPUs; step height/corner max-abs err"
METRIC_FHD = "point-cloud frames/sec (1920×1080 pts, stress); step height/corner max-abs err"


def shard(total, world, rank):
    """Contiguous frame range of `rank`: frame i -> rank i*world//total."""
    lo = (total * rank + world - 1) // world
    hi = (total * (rank + 1) + world - 1) // world
    return lo, hi


def launch_command(n_gpus, argv, port=None):
    """The command that runs this script as n_gpus ranks on this node (what the driver uses for N > 1)."""
    if port is None:
        port = 29500 + os.getpid() % 2000
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def world_from_env(n_gpus, environ=None):
    """(world, rank, local_rank) from the launcher's environment; --gpus must agree with it."""
    env = os.environ if environ is None else environ
    world = int(env.get("WORLD_SIZE", "1"))
    rank = int(env.get("RANK", "0"))
    local_rank = int(env.get("LOCAL_RANK", str(rank)))
    if world != max(n_gpus, 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a line for another configuration" % (n_gpus, world))
    if not 0 <= rank < world:
        raise SystemExit("bench.py: RANK=%d outside WORLD_SIZE=%d" % (rank, world))
    return world, rank, local_rank


def device_for_rank(local_rank, n_devices, environ=None):
    """One GPU per rank; SSD_BENCH_DEVICE forces all ranks onto one device (2-rank test on a 1-GPU box)."""
    env = os.environ if environ is None else environ
    forced = env.get("SSD_BENCH_DEVICE")
    dev = int(forced) if forced not in (None, "") else local_rank
    if not 0 <= dev < n_devices:
        raise SystemExit("bench.py: rank needs device %d but %d GPU(s) are visible (one GPU per rank; no oversubscription "
                         "unless SSD_BENCH_DEVICE is set)" % (dev, n_devices))
    return dev


_AFFINITY_AT_START = None      # the main thread's CPUs before bind_rank_to_device narrowed them (the all-cores CPU baseline uses them all)


def bind_rank_to_device(ssd, device):
    """This rank onto the CPUs next to its GPU (the NUMA node the device hangs off), and the device's identity for the report.
    The library maps HIP device -> PCI bus id -> /sys/bus/pci/devices/<id>/{numa_node,local_cpulist} and binds the CALLING
    thread — the rank's main thread, which every thread it starts afterwards inherits from (torch's, the oracle pool's, the HIP
    runtime's later ones).  Where the platform names no local CPUs (numa_node -1, single-socket boxes, containers that hide
    sysfs) the affinity is left alone and the report says so."""
    global _AFFINITY_AT_START
    info = ssd.device_info(device)
    _AFFINITY_AT_START = os.sched_getaffinity(0)
    before = len(_AFFINITY_AT_START)
    bound = ssd.bind_thread_to_device(device)
    info.update({"device": device, "cpus_before": before, "cpus_bound": bound, "bound": bound > 0})
    return info


def distinct_devices(reports):
    """how many different physical GPUs the ranks' reports name (PCI bus id, else UUID, else index)"""
    return len(set((r.get("pci_bus_id") or r.get("uuid") or str(r.get("device"))) for r in reports))


class Ranks:
    """The only communication of the bench: a barrier and a max over ranks, on CPU tensors over gloo."""

    def __init__(self, world, rank):
        self.world, self.rank = world, rank
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def max(self, value):
        if self.world == 1:
            return float(value)
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(value)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj):
        """-> list of every rank's obj on rank 0 (small, outside the timed region), None elsewhere"""
        if self.world == 1:
            return [obj]
        import torch.distributed as dist
        out = [None] * self.world if self.rank == 0 else None
        dist.gather_object(obj, out, dst=0)
        return out

    def close(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


def cpu_quota():
    """how many CPUs' worth of time this process's control group may use (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us), or
    None without a limit.  A box can show 256 CPUs in the affinity mask and grant 16 of them: 256 busy threads are then throttled to a
    sixteenth of the time each, and a per-core figure from them says nothing."""
    try:
        q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(p_)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p_ > 0:
            return float(q) / float(p_)
    except Exception:
        pass
    return None


def numa_node_of(address):
    """the NUMA node that holds the page at `address` (get_mempolicy(MPOL_F_NODE | MPOL_F_ADDR)), or None where the call is not allowed"""
    import ctypes
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        node = ctypes.c_int(-1)
        SYS_get_mempolicy = {"x86_64": 239, "aarch64": 236}.get(os.uname().machine)
        if SYS_get_mempolicy is None:
            return None
        rc = libc.syscall(ctypes.c_long(SYS_get_mempolicy), ctypes.byref(node), None, ctypes.c_ulong(0), ctypes.c_void_p(address), ctypes.c_ulong(1 | 2))
        return int(node.value) if rc == 0 else None
    except Exception:
        return None


def host_fed_leg(ssd, scenes, n_frames=256, reps=3, device=0):
    """PCIe-inclusive rates of ssd_process_host / ssd_process_depth_host (frames in HOST memory; double-buffered ingest:
    the copy of a slice overlaps the kernels of the one before) — the deployment a camera implies.  Reported beside the
    HBM-resident `value`, never as it.  XGA frames; float3 vertices and 16-bit depth; pinned and pageable sources."""
    import numpy as np
    W, H = 1024, 768
    sc = scenes.batch_scenes(ssd, W, H, n_frames, base_seed=4242)
    trans = ssd.transformation_for_scene(sc[0])
    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=64), trans, device)
    det.set_intrinsics(ssd.intrinsics_for_scene(sc[0]))
    out = {"frames": n_frames, "width": W, "height": H}
    dev = ssd.DeviceBuffer(W * H * 12 * n_frames, device)
    for kind, item, shape, dtype in (("float3", 12, (n_frames, H, W, 3), np.float32), ("depth16", 2, (n_frames, H, W), np.uint16)):
        if kind == "float3":
            ssd.synth_device(sc, dev.ptr, device=device)
        else:
            ssd.synth_depth_device(sc, dev.ptr, device=device)
        pinned = ssd.PinnedArray(shape, dtype)
        pinned.array[...] = dev.download(W * H * item * n_frames, dtype=dtype).reshape(shape)
        pageable = np.array(pinned.array)
        run = det.process_host if kind == "float3" else det.process_depth_host
        # both sources in both orders (pinned, pageable, pageable, pinned), two warm-up calls in front of every timed leg: round 5's
        # line had the pinned source 23 % BEHIND the pageable one on the driver's box - measured first, behind a single warm-up call
        # (VERDICT round 5, item 7); with the order taken out, the two legs of a source say whether a difference is the source's
        legs = {"pinned": [], "pageable": []}
        found = 0
        for name, arr in (("pinned", pinned.array), ("pageable", pageable), ("pageable", pageable), ("pinned", pinned.array)):
            run(arr)
            run(arr)
            t0 = time.perf_counter()
            for _ in range(reps):
                res = run(arr)
            legs[name].append((time.perf_counter() - t0) / reps)
            found = int(sum(1 for r in res if r.n_steps >= 3))
        for name, arr in (("pinned", pinned.array), ("pageable", pageable)):
            dt = min(legs[name])
            out["%s_%s" % (kind, name)] = {"frames_per_s": n_frames / dt, "host_to_device_GBps": n_frames * W * H * item / dt / 1e9,
                                          "legs_GBps": [n_frames * W * H * item / d / 1e9 for d in legs[name]],
                                          "numa_node_of_source": numa_node_of(arr.ctypes.data), "stairs_found": found}
        pinned.free()
    dev.free()
    det.close()
    return out


def file_sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def counters_stamp(name, lib_sha):
    """The stamp of a committed counter file (profiles/<name>: written by tools/r06_final.sh with the git revision and the
    sha256 of the libssd_hip.so its passes ran) against the library this process loaded: -> (source string, stale)."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        st = json.load(open(path)).get("stamp") or {}
    except Exception:
        return None, None
    src = "profiles/%s (rocprofv3 --pmc passes of this command, committed; not re-measured in this run; git %s, libssd_hip.so sha256 %s)" % (
        name, st.get("git_head", "unknown"), (st.get("lib_sha256") or "unknown")[:16])
    return src, (st.get("lib_sha256") != lib_sha)


def secondary_leg(ssd, scenes, torch, device, workload, steps, warmup, check_frames, with_cpu):
    """One of the other BASELINE configurations timed in the same run: `fhd_stress` = configs[4], 256
    frames of 1920x1080 (8 noisy steps, 5 % outliers); `depth16` = the metric's 1024 XGA frames as 16-bit depth (SURVEY 8(f)
    rank 1).  Same procedure as the primary region - frames resident in HBM, three batches in flight, warm-up, K timed steps
    between synchronisations, then a few passes one at a time with events for K1's own duration - and a few frames of the last
    pass against the CPU oracle."""
    import numpy as np
    fhd = workload == "fhd_stress"
    depth_in = workload == "depth16"
    W, H = (1920, 1080) if fhd else (1024, 768)
    F = 256 if fhd else 1024
    sc = scenes.fhd_stress_scenes(ssd, F, base_seed=9000) if fhd else scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
    trans = ssd.transformation_for_scene(sc[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=3)
    frame_bytes = W * H * (2 if depth_in else 12)
    frames = torch.empty(F * frame_bytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    det = ssd.Detector(cfg, trans, device)
    intr = ssd.intrinsics_for_scene(sc[0])
    if depth_in:
        ssd.synth_depth_device(sc, frames.data_ptr(), device=device, stream=stream)
        det.set_intrinsics(intr)
    else:
        ssd.synth_device(sc, frames.data_ptr(), device=device, stream=stream)
    enqueue = (lambda: det.enqueue_depth(frames.data_ptr(), F, stream=stream)) if depth_in else (lambda: det.enqueue(frames.data_ptr(), F, stream=stream))
    ahead = max(det.batches_in_flight, 2) - 1

    def run(n):
        res = None
        for i in range(n):
            enqueue()
            if i >= ahead:
                res = det.fetch(F, back=ahead)
        for back in range(min(ahead, n) - 1, -1, -1):
            res = det.fetch(F, back=back)
        return res

    run(max(warmup, 4))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    det.set_timing(True)
    stage = {k: 0.0 for k in ssd.STAGE_NAMES}
    n_extra = 4
    for b in range(n_extra + 1):
        enqueue()
        det.fetch(F)
        if b > 0:
            for k, v in det.stage_times_ms().items():
                stage[k] += v / n_extra
    det.set_timing(False)
    alg = float(frame_bytes) * F
    k1 = stage["hist"]
    out = {"workload": ("BASELINE configs[4]: %d synthetic %dx%d frames (8 noisy steps, 5 %% outliers) resident in HBM" % (F, W, H)) if fhd else
                       ("the metric's %d XGA frames as 16-bit depth images resident in HBM, deprojected on the fly (SURVEY.md section 8(f) rank 1)" % F),
           "value": F * steps / dt, "unit": "frames/s", "steps": steps, "ms_per_step": dt / steps * 1e3, "frames_per_step": F,
           "width": W, "height": H, "input": "depth16" if depth_in else "float3", "batches_in_flight": det.batches_in_flight,
           "stage_ms": stage, "k1_ms": k1, "algorithmic_bytes_per_launch": alg,
           "k1_frac_of_hbm_peak": (alg / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS) if k1 > 0 else None,
           "whole_path_frac_of_hbm_peak": alg * steps / dt / 1e9 / HBM_PEAK_GBS,
           "steps_found": int(sum(r.n_steps for r in res))}
    if with_cpu and check_frames > 0:
        import oracle_binding as ob
        import parity
        oracle = ob.load_oracle()
        rep = {}
        idx = sorted(set(int(i) for i in np.linspace(0, F - 1, check_frames)))
        for i in idx:
            x = frames[i * frame_bytes:(i + 1) * frame_bytes].cpu().numpy()
            x = oracle.deproject(intr, x.view(np.uint16).reshape(H, W)) if depth_in else x.view(np.float32)
            parity.check_results_only(ssd, oracle, cfg, trans.constants, x, res[i], rep)
        out["parity"] = {"frames_checked_against_oracle": len(idx), "max_abs_height_err_m": rep.get("max_height_err", 0.0),
                         "max_abs_corner_err_m": rep.get("max_corner_err", 0.0), "bar_m": 1e-4}
    det.close()
    del frames
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (default: 1024 XGA / 256 FHD)")
    ap.add_argument("--workload", choices=["xga_batch", "fhd_stress"], default="xga_batch",
                    help="xga_batch = BASELINE configs[2] (the metric's configuration); fhd_stress = configs[4]")
    ap.add_argument("--cpu-frames", type=int, default=0, help="bounded CPU-baseline sample in frames (default: ~15 s of CPU work)")
    ap.add_argument("--input", choices=["float3", "depth16"], default="float3",
                    help="float3 = xyz vertices (the metric's input); depth16 = 16-bit depth frames deprojected on the fly (SURVEY 8f rank 1)")
    ap.add_argument("--batches-in-flight", type=int, default=3,
                    help="workspaces of the handle (ssd_config::batches_in_flight), asked for explicitly: 3 = SSD_BATCHES_IN_FLIGHT_THROUGHPUT "
                         "(the bench enqueues ahead of its fetches); 1 (or 0, the library's default) = strictly one batch at a time in stream "
                         "order — what the profiling passes use, so that a kernel's traced duration is its own")
    ap.add_argument("--two-pass", action="store_true", help="A/B: ssd_set_single_pass(h, 0): K1 then k_raster over every frame, as before round 4's single pass")
    ap.add_argument("--pad-mib", type=int, default=0, help="tools: hold this many MiB of device memory allocated BEFORE the frames (K1 has two states by where the "
                    "input buffer lands, profiles/r06_box_spread.txt; a pad moves it); 0 = none")
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed load in front of the W warm-up steps (an idle device needs more than a few steps to reach its clocks)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-hostfed", action="store_true", help="skip the host-fed (PCIe-inclusive) leg reported beside `value`")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-frame latency leg (profiling passes: its 56 one-frame "
                                                               "launches would be averaged into the per-kernel statistics)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary legs (BASELINE configs[4] = FHD stress, and 16-bit depth input) the default run times after the primary region")
    ap.add_argument("--risers", action="store_true",
                    help="also gather the evidence of the vertical faces (extension beyond the reference, SURVEY 8f rank 4); off for the metric")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    if args.gpus > 1 and "RANK" not in os.environ:
        # not under a launcher: start the ranks as a child BEFORE anything here touches the GPU, exit with its code
        import subprocess
        raise SystemExit(subprocess.call(launch_command(args.gpus, sys.argv[1:])))

    import numpy as np
    import torch

    world, rank, local_rank = world_from_env(args.gpus)
    ssd = importlib.import_module("stair-step-detector_amd")
    n_devices = ssd.device_count()
    if n_devices < 1:
        raise SystemExit("bench.py: no GPU visible; the HIP path is mandatory (there is no CPU fallback)")
    device = device_for_rank(local_rank, n_devices)
    where = bind_rank_to_device(ssd, device)          # first: the rank onto its GPU's socket, then everything else
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible to torch; the HIP path is mandatory (there is no CPU fallback)")
    torch.cuda.set_device(device)
    ranks = Ranks(world, rank)

    import scenes

    fhd = args.workload == "fhd_stress"
    W, H = (1920, 1080) if fhd else (1024, 768)
    F = args.frames or (256 if fhd else 1024)
    frame_bytes = W * H * 12
    # frames of this rank: a contiguous range of the global frame index space (no overlap between ranks)
    lo, hi = shard(F * world, world, rank)
    assert hi - lo == F
    if fhd:
        sc_list = scenes.fhd_stress_scenes(ssd, F, base_seed=9000 + lo)
    else:
        sc_list = scenes.batch_scenes(ssd, W, H, F, base_seed=100000 + lo, rng_seed=1000 + rank)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=args.batches_in_flight)
    depth_in = args.input == "depth16"
    if depth_in:
        frame_bytes = W * H * 2
    pad_hold = torch.empty(args.pad_mib << 20, dtype=torch.uint8, device="cuda") if args.pad_mib > 0 else None       # tools: see --pad-mib
    frames = torch.empty(F * frame_bytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    det = ssd.Detector(cfg, trans, device)
    depth = det.batches_in_flight
    if args.two_pass:
        det.set_single_pass(False)           # ssd_set_single_pass: the planes' memory back, two passes
    if args.risers:
        det.set_risers(True, tolerance=0.03, min_support=200)
    intr = ssd.intrinsics_for_scene(sc_list[0])
    if depth_in:
        ssd.synth_depth_device(sc_list, frames.data_ptr(), device=device, stream=stream)
        det.set_intrinsics(intr)
    else:
        ssd.synth_device(sc_list, frames.data_ptr(), device=device, stream=stream)

    def enqueue():
        if depth_in:
            det.enqueue_depth(frames.data_ptr(), F, stream=stream)
        else:
            det.enqueue(frames.data_ptr(), F, stream=stream)

    ahead = max(depth, 2) - 1            # enqueues kept ahead of the fetches (the handle's result slots: max(2, depth))

    def run(n_steps):
        """n_steps passes over the batch, every pass's results fetched to the host: pass i + `ahead` is enqueued before the
        results of pass i are read (they travel with their own enqueue), so the GPU never waits for the host and the handle
        has its `depth` batches in flight."""
        res = None
        for i in range(n_steps):
            enqueue()
            if i >= ahead:
                res = det.fetch(F, back=ahead)
        for back in range(min(ahead, n_steps) - 1, -1, -1):
            res = det.fetch(F, back=back)
        return res

    # Before the W warm-up steps of the contract: load until the device has been busy for a while.  The first process on an idle
    # box measured 227 k frames/s over its 10 timed steps (34 ms) after 2 warm-up steps, the same command seconds later 291-300 k
    # (profiles/r04_cold_start.txt): clocks and memory ramp up over more than a few milliseconds.  Untimed; reported as `prewarm`.
    c0 = time.perf_counter()
    prewarm_steps = 0
    while time.perf_counter() - c0 < args.prewarm_seconds:
        run(4)
        prewarm_steps += 4
    prewarm = {"seconds": time.perf_counter() - c0, "steps": prewarm_steps}

    res = run(args.warmup)

    def fence():
        torch.cuda.synchronize()
        ranks.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    res = run(args.steps)
    fence()
    dt = time.perf_counter() - t0

    # Per-stage device times: EXTRA steps after the timed region, one at a time (enqueue, fetch, then the next), HIP events
    # recorded between the launches on the stream the kernels run on.  The timed steps above overlap (depth batches in
    # flight), where an event-bracketed stage is a kernel's share of a machine it splits with the other batches' kernels.
    # A plain read stream over the same buffer is timed before and after (what the memory system delivers right now).
    n_extra = 6
    det.set_timing(True)
    stream_ms = [ssd.stream_read_ms(frames.data_ptr(), F * frame_bytes, reps=5, device=device)] if not depth_in else []
    stage = {k: 0.0 for k in ssd.STAGE_NAMES}
    per_pass = []
    predict_ms = 0.0
    for b in range(n_extra + 1):
        c0 = time.perf_counter()
        enqueue()
        det.fetch(F)
        if b > 0:                                        # the first one warms the events up
            per_pass.append((time.perf_counter() - c0) * 1e3)
            for k, v in det.stage_times_ms().items():
                stage[k] += v / n_extra
            predict_ms += det.predict_time_ms() / n_extra
    single_pass = det.single_pass_stats(F, scan_planes=False)     # of the last pass: did K1 raster the step plateaus itself, for how many frames
    one_at_a_time_ms = sorted(per_pass)[len(per_pass) // 2]      # median: a host hiccup in one pass is not the handle's rate
    if not depth_in:
        stream_ms.append(ssd.stream_read_ms(frames.data_ptr(), F * frame_bytes, reps=5, device=device))
    det.set_timing(False)

    dt_max = ranks.max(dt)
    # which frames of the global index space each rank processed, and on which device (rank 0 reports it); with more
    # than one rank every rank also checks a few frames of ITS shard against the CPU oracle (outside the timed region)
    k1_ms = stage["hist"]
    alg_bytes = (2.0 if depth_in else 12.0) * W * H * F    # 12 B per raw point (2 B per depth pixel), read once
    mine = {"rank": rank, "device": device, "where": where, "frames": [lo, hi], "seconds": dt, "steps_found": int(sum(r.n_steps for r in res)),
            "stage_ms": stage, "predict_ms": predict_ms, "k1_frac_of_hbm_peak": (alg_bytes / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k1_ms > 0 else 0.0,
            "one_batch_at_a_time_ms": one_at_a_time_ms}
    if world > 1 and not args.no_cpu:
        import oracle_binding as ob
        import parity
        oracle = ob.load_oracle()
        rep = {}
        for i in sorted(set(int(i) for i in np.linspace(0, F - 1, min(5, F)))):
            x = frames[i * frame_bytes:(i + 1) * frame_bytes].cpu().numpy()
            x = oracle.deproject(intr, x.view(np.uint16).reshape(H, W)) if depth_in else x.view(np.float32)
            parity.check_results_only(ssd, oracle, cfg, trans.constants, x, res[i], rep)
        mine["parity"] = {"frames_checked_against_oracle": min(5, F), "max_abs_height_err_m": rep.get("max_height_err", 0.0),
                          "max_abs_corner_err_m": rep.get("max_corner_err", 0.0)}
    shards = ranks.gather(mine)

    if rank == 0:
        lib_sha = file_sha256(ssd.LIB_PATH)
        value = world * F * args.steps / dt_max
        achieved = alg_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        plain_ms = sum(stream_ms) / len(stream_ms) if stream_ms else None
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_k_hist.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch_at_%dx%dx%d" % (W, H, F))
                if depth_in:
                    traffic = None
                elif traffic is None and not fhd:     # measured at 1024 frames: scale per frame
                    ref = json.load(open(pmc)).get("hbm_bytes_per_launch_at_1024x768x1024")
                    traffic = None if ref is None else ref * F / 1024.0
            except Exception:
                traffic = None
        pipeline_moved = None
        pm = os.path.join(ROOT, "profiles", "pmc_pipeline.json")
        if os.path.exists(pm) and not fhd and not depth_in and F == 1024:
            try:
                j = json.load(open(pm))
                pipeline_moved = {"hbm_read_bytes": j["hbm_read_bytes"], "hbm_write_bytes": j["hbm_write_bytes"],
                                  "over_algorithmic": j["bytes_moved_over_algorithmic"]}
                pipeline_moved["source"], pipeline_moved["stale"] = counters_stamp("pmc_pipeline.json", lib_sha)
            except Exception:
                pipeline_moved = None
        # The two-sided floor of the whole pass, from the committed counters of this command (profiles/pmc_issue.json: vector
        # instructions per launch and kernel; profiles/pmc_pipeline.json: HBM bytes): every vector instruction occupies one of the
        # chip's 1024 SIMDs for 4 cycles (tools/instr_rate.hip: fp64, conversions, compares, integer ops alike), every byte
        # crosses HBM at the plain stream's rate at best.  The pass cannot be shorter than either; how far above both it is says
        # what the overlap of the kernels leaves on the table.
        floors = None
        pi = os.path.join(ROOT, "profiles", "pmc_issue.json")
        if os.path.exists(pi) and pipeline_moved is not None:
            try:
                kernels = json.load(open(pi))["kernels"]
                valu = sum(v.get("SQ_INSTS_VALU", 0.0) for k, v in kernels.items() if "k_stream_read" not in k and "synth" not in k)
                simds, clock_hz = 1024, 2.4e9
                floors = {"valu_instructions_per_launch": valu, "valu_floor_ms": valu * VALU_CYCLES / (simds * clock_hz) * 1e3,
                          "hbm_floor_ms": None if plain_ms is None else (pipeline_moved["hbm_read_bytes"] + pipeline_moved["hbm_write_bytes"]) / (alg_bytes / plain_ms),
                          "hbm_floor_ms_at_peak": (pipeline_moved["hbm_read_bytes"] + pipeline_moved["hbm_write_bytes"]) / (HBM_PEAK_GBS * 1e9) * 1e3,
                          "measured_ms": dt_max / args.steps * 1e3,
                          "note": "valu floor = vector instructions x %.1f cycles / (1024 SIMDs x 2.4 GHz) (what a wave64 instruction of this path costs a SIMD: "
                                  "fp64, conversions, compares, DPP, packed fp32 alike - profiles/r05_instr_rate.txt); hbm floor = HBM bytes of the pass / "
                                  "the plain read stream's rate measured in this run; counters: profiles/pmc_issue.json, profiles/pmc_pipeline.json "
                                  "(committed rocprofv3 --pmc passes of this command, not re-measured here)" % VALU_CYCLES}
                floors["source"], floors["stale"] = counters_stamp("pmc_issue.json", lib_sha)
            except Exception:
                floors = None
        out = {
            "metric": METRIC_FHD if fhd else METRIC, "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[4]: %d synthetic %dx%d frames (8 noisy steps, 5 %% outliers) resident in HBM; per GPU"
                                    if fhd else
                                    ("BASELINE configs[3]: %d-frame batch frame-sharded over %d GPUs; " % (F * world, world) if world > 1 else
                                     "BASELINE configs[2]: ") +
                                    "batch of %d synthetic %dx%d frames (3-step staircases, randomised rise/"
                                    "tread/yaw/noise) resident in HBM, streamed through the whole per-frame path; per GPU") % (F, W, H),
                       "frames_per_gpu_per_step": F, "width": W, "height": H, "input": args.input, "risers": bool(args.risers),
                       "batches_in_flight": depth, "parallelism": "frame-sharded x%d, no collective" % world},
            "roofline": {"bound": "hbm", "kernel": ("k_hist_planes (K1 of the single pass: transform+crop+bin+histogram+cell records, and the raster of "
                                                    "the step plateaus into planes)" if single_pass["ran"] else "k_hist (K1: transform+crop+bin+histogram)"),
                         "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": None if traffic is None else counters_stamp("pmc_k_hist.json", lib_sha)[0],
                         "stale": None if traffic is None else counters_stamp("pmc_k_hist.json", lib_sha)[1],
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": k1_ms,
                         "plain_stream_GBps": None if plain_ms is None else alg_bytes / (plain_ms * 1e-3) / 1e9,
                         "k1_over_plain_stream": None if plain_ms is None or k1_ms <= 0 else plain_ms / k1_ms,
                         "plain_stream_note": "ssd_test_stream_read (tools/loadbench.hip variant C: 16-byte loads, wave-contiguous) over the "
                                              "same buffer, 5 launches before and 5 after the separate timed steps"},
            "floors": floors,
            "prewarm": prewarm,
            "stage_ms": stage,
            "predict_ms": predict_ms,
            "single_pass": {"ran": single_pass["ran"], "frames_with_steps": single_pass["with_steps"], "frames_covered_by_the_predictor": single_pass["covered"],
                            "planes_per_frame": single_pass["planes"] / float(F),
                            "note": "K1 rasters the step plateaus itself into planes of the height bins k_predict (predict_ms, in front of the seven stages) "
                                    "expects them in; k_peaks checks the planes against the complete histogram frame by frame and k_raster (stage "
                                    "'raster') does only the frames not covered; of the last of the separate timed steps"},
            "stage_ms_source": "separate timed steps: %d extra passes after the timed region, one batch at a time (enqueue, fetch), HIP events "
                               "between the launches on the kernels' stream; the timed region itself keeps %d batches in flight" % (n_extra, depth),
            "one_batch_at_a_time": {"ms_per_step": one_at_a_time_ms, "frames_per_s": F / one_at_a_time_ms * 1e3,
                                    "note": "the same handle fed one batch at a time (no overlap; with the stage events): median of the %d extra passes, host clock around enqueue + fetch" % n_extra},
            "pipeline_bytes_moved": pipeline_moved,
            "pipeline_bytes_algorithmic_frac_of_peak": (alg_bytes * args.steps / dt_max / 1e9) / HBM_PEAK_GBS,
        }
        # the dominant kernel's own issue floor: in the single pass K1 also rasters, and sits nearer to the vector-issue roof than to HBM's
        if os.path.exists(pi):
            try:
                kernels = json.load(open(pi))["kernels"]
                want = "k_hist_planes" if single_pass["ran"] else "k_hist<"
                hit = [v for k, v in kernels.items() if want in k and "SQ_INSTS_VALU" in v]
                if hit and k1_ms > 0 and not depth_in and not fhd and F == 1024:
                    vf = hit[0]["SQ_INSTS_VALU"] * VALU_CYCLES / (1024 * 2.4e9) * 1e3
                    out["roofline"]["issue"] = {"valu_instructions_per_launch": hit[0]["SQ_INSTS_VALU"], "valu_floor_ms": vf, "frac_of_issue_peak": vf / k1_ms,
                                                "valu_busy_percent_one_batch_at_a_time": hit[0].get("VALUBusy"),
                                                "note": "the kernel's vector instructions x %.1f cycles / (1024 SIMDs x 2.4 GHz) over its measured launch time; counters: "
                                                        "profiles/pmc_issue.json (committed pass of this command)" % VALU_CYCLES,
                                                "stale": counters_stamp("pmc_issue.json", lib_sha)[1]}
            except Exception:
                pass
        if floors is not None:           # the whole pass's floors beside the dominant kernel's roofline
            out["roofline"]["valu_floor_ms"] = floors["valu_floor_ms"]
            out["roofline"]["hbm_floor_ms"] = floors["hbm_floor_ms"]
            out["roofline"]["pass_ms"] = floors["measured_ms"]
        # Which K1 number is which: `frac` is THIS run's - HIP events on the stream, this GPU -; frac_profiles is
        # the committed rocprofv3 --kernel-trace --stats average of the same command on the GPU named beside it.  GPUs of this pool
        # differ by several per cent on this kernel (profiles/r06_box_spread.txt): when the two differ by more than 3 %, the line says so.
        out["roofline"]["frac_source"] = "this run: HIP events around the kernel on its stream, %d timed launches, GPU %s" % (
            args.steps, where.get("uuid") or where.get("pci_bus_id") or str(device))
        kp = os.path.join(ROOT, "profiles", "k1_rocprof.json")
        if os.path.exists(kp) and not fhd and not depth_in and F == 1024:
            try:
                j = json.load(open(kp))
                out["roofline"]["frac_profiles"] = j["frac"]
                if j.get("other_rocprof_passes_same_library"):
                    out["roofline"]["frac_profiles_other_passes"] = j["other_rocprof_passes_same_library"]
                out["roofline"]["frac_profiles_source"] = {"file": "profiles/k1_rocprof.json", "avg_launch_ms": j["avg_launch_ms"], "gpu": j.get("gpu"),
                                                           "summary": j.get("summary"), "stale": counters_stamp("k1_rocprof.json", lib_sha)[1]}
                if j["frac"] > 0 and abs(out["roofline"]["frac"] - j["frac"]) / j["frac"] > 0.03:
                    out["roofline"]["frac_warning"] = ("this run's K1 fraction %.3f and the committed rocprofv3 average %.3f differ by more than 3 %%: another GPU of the "
                                                       "pool, or another clock state of the same one (profiles/r06_box_spread.txt); the claim against the north star's 0.60 "
                                                       "rests on `frac`, measured here" % (out["roofline"]["frac"], j["frac"]))
            except Exception:
                pass
        out["library"] = {"path": os.path.relpath(ssd.LIB_PATH, ROOT), "sha256": lib_sha}
        out["devices"] = [sh["where"] for sh in shards]
        out["distinct_devices"] = distinct_devices(out["devices"])
        if world > 1:
            out["ranks"] = shards
            if out["distinct_devices"] != world:
                # said in the line, not raised: the other ranks are already waiting in the closing barrier
                out["warning"] = "%d ranks on %d distinct GPU(s)%s: this is not an %d-GPU measurement" % (
                    world, out["distinct_devices"], " (SSD_BENCH_DEVICE puts every rank on one device: a test of the launcher)" if os.environ.get("SSD_BENCH_DEVICE") else "", world)
        steps_hist = [r.n_steps for r in res]
        out["steps_histogram"] = {str(k): int(sum(1 for n in steps_hist if n == k)) for k in sorted(set(steps_hist))}

        if not args.no_cpu:
            import oracle_binding as ob
            import parity
            oracle = ob.load_oracle()
            ocfg, ocal = ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants)
            # cpu_baseline is reported at N = 1 only; with more ranks rank 0 still spot-checks its shard against the oracle
            n_cpu = max(1, min(args.cpu_frames or (17 if world > 1 else 256 if fhd else 1280), F))
            idx = [int(i) for i in np.linspace(0, F - 1, n_cpu)]
            rep = {}
            cdt = 0.0
            checked = 0
            keep = []                                       # host copies for the all-cores leg below
            for i in idx:                                   # one frame at a time: download (untimed), oracle (timed)
                x = frames[i * frame_bytes:(i + 1) * frame_bytes].cpu().numpy()
                x = oracle.deproject(intr, x.view(np.uint16).reshape(H, W)) if depth_in else x.view(np.float32)
                c0 = time.perf_counter()
                oracle.process_lean(ocfg, ocal, x)
                cdt += time.perf_counter() - c0
                if len(keep) < (64 if fhd else 256):
                    keep.append(x)
                if i % 64 == 0 or i == idx[-1]:
                    parity.check_results_only(ssd, oracle, cfg, trans.constants, x, res[i], rep)
                    checked += 1
            if world == 1:
                out["cpu_baseline"] = {"value": n_cpu / cdt, "unit": "frames/s", "cores": 1, "kind": "port",
                                       "sample": "%d of the %d frames of rank 0's batch (evenly spaced), the whole per-frame path in "
                                                 "oracle/ssd_oracle.cpp (CPU restatement of the reference, one thread as the reference runs), "
                                                 "%.1f s of CPU time" % (n_cpu, F, cdt)}
            # the same port on all host cores, one frame per thread (ctypes drops the GIL)
            if world == 1 and _AFFINITY_AT_START:
                os.sched_setaffinity(0, _AFFINITY_AT_START)      # "all host cores": not only the GPU's socket (pool threads inherit this)
            cpus_all = sorted(os.sched_getaffinity(0))
            visible = len(cpus_all)
            quota = cpu_quota()
            if world == 1 and visible > 1 and len(keep) > 1:
                # The native runner (oracle/ssd_oracle_mt.cpp): pinned std::threads, each on a private copy of a
                # frame it touched first (its own NUMA node), all from one start line, frames pre-loaded - what the host could do if frames
                # were sharded across its cores as they are across GPUs.  (Round 5 drove the same oracle from a pool of Python threads
                # through ctypes: 4 frames/s per core against 90 on one.)  How many threads: a box of this pool shows every CPU of the
                # host in the affinity mask and grants the job a share of them; where the control group says how many (cpu_quota) that
                # many threads run, spread over the mask; where it does not, a few counts are tried and the best one is the figure.
                single = n_cpu / cdt
                if quota is not None and quota >= 1.0:
                    counts = [min(visible, int(quota))]
                else:
                    counts = sorted(set(min(visible, c) for c in (8, 16, 32, 64, visible)))
                reps = 3 if fhd else 6
                tried, best = [], None
                for n_use in counts:
                    cpus = [cpus_all[(i * visible) // n_use] for i in range(n_use)]
                    many = oracle.process_many(ocfg, ocal, keep[:min(len(keep), 32)], cpus, reps=reps)
                    tried.append({"threads": n_use, "frames_per_s": many["frames_per_s"], "wall_s": many["wall_s"], "host_read_GBps": many["read_gb_per_s"]})
                    if best is None or many["frames_per_s"] > best[1]["frames_per_s"]:
                        best = (n_use, many)
                cores, many = best
                per_core = many["frames_per_s"] / cores
                out["cpu_baseline_all_cores"] = {
                    "value": many["frames_per_s"], "unit": "frames/s", "cores": cores, "kind": "port",
                    "sample": "%d frames: %d pinned threads x %d frames each, every thread on a private copy of one of %d distinct frames of the batch, "
                              "%.1f s wall from the first thread's start to the last one's end (oracle/ssd_oracle_mt.cpp)"
                              % (many["frames"], cores, reps, min(len(keep), 32), many["wall_s"]),
                    "cpus_visible": visible, "cpu_quota_of_the_control_group": quota, "thread_counts_tried": tried,
                    "frames_per_s_per_core": per_core, "single_thread_frames_per_s": single, "scaling_vs_cores_x_single_thread": per_core / single,
                    "host_read_GBps_same_threads": many["read_gb_per_s"],
                    "limited_by": ("the cores: within 2 x of threads x the single thread's rate" if per_core * 2.0 >= single else
                                   "not the cores alone: %.1f frames/s per thread against %.1f on one thread alone; the same %d threads read %.0f GB/s from "
                                   "their private frames in a plain loop, and a frame costs the oracle its 12 B x %d points and several times that in "
                                   "per-frame lists and images (write-allocate traffic) - the host's memory, or a CPU share smaller than the thread count"
                                   % (per_core, single, cores, many["read_gb_per_s"], W * H)),
                    "input_GBps": many["frames_per_s"] * 12.0 * W * H / 1e9}
            out["parity"] = {"frames_checked_against_oracle": checked, "max_abs_height_err_m": rep.get("max_height_err", 0.0),
                             "max_abs_corner_err_m": rep.get("max_corner_err", 0.0), "bar_m": 1e-4}
        if world == 1 and not depth_in and not args.no_latency:
            # BASELINE configs[1]: ONE frame resident in HBM through the whole path (enqueue + fetch); GPU-latency-bound
            one = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=1), trans, device)
            lat, st1 = [], {k: 0.0 for k in ssd.STAGE_NAMES}
            for i in range(8 + 48):                       # the latency: no timing events between the launches
                c0 = time.perf_counter()
                one.enqueue(frames.data_ptr(), 1, stream=stream)
                one.fetch(1)
                if i >= 8:
                    lat.append(time.perf_counter() - c0)
            one.set_timing(True)
            for i in range(4 + 16):                       # the stages: events between the launches (they add ~20 us per call)
                one.enqueue(frames.data_ptr(), 1, stream=stream)
                one.fetch(1)
                if i >= 4:
                    for k, v in one.stage_times_ms().items():
                        st1[k] += v / 16
            one.close()
            out["single_frame"] = {"latency_ms_device_resident": sorted(lat)[len(lat) // 2] * 1e3, "stage_ms": st1,
                                   "note": "one %dx%d frame already in HBM, ssd_enqueue + ssd_fetch (7 dependent launches, the last "
                                           "one storing the result into pinned host memory); median of 48 calls without timing events; stage_ms "
                                           "from 16 further calls with events between the launches; bounded by the GPU-side latency of the "
                                           "kernels' short phases (DESIGN.md section 3), not by HBM" % (W, H)}
        if world == 1 and not depth_in and not args.no_latency:
            # The same overlap across handles (ssd_pipeline_*: depth handles with one workspace each on depth streams, fed
            # round-robin), as a cross-check of what `value` — one handle with its own workspaces — delivers.
            out["pipelined"] = {"note": "same batch, same K steps through ssd_pipeline_* (depth single-workspace handles on depth HIP "
                                         "streams fed round-robin, per-stage timing events off; INTEGRATION.md section 4)", "unit": "frames/s"}
            for depth in (2, 3, 4):
                pipe = ssd.Pipeline(cfg, trans, device, depth=depth)
                torch.cuda.synchronize()

                def overlapped(n_steps):
                    for i in range(n_steps):
                        if pipe.pending() == depth:
                            pipe.next(copy=False)
                        pipe.submit(frames.data_ptr(), F)
                    while pipe.pending():
                        pipe.next(copy=False)

                overlapped(depth)
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                overlapped(args.steps)
                torch.cuda.synchronize()
                odt = time.perf_counter() - c0
                pipe.close()
                out["pipelined"]["depth%d" % depth] = {"value": F * args.steps / odt, "ms_per_step": odt / args.steps * 1e3}
        if world == 1 and not fhd and not depth_in and not args.no_secondary:
            # the other BASELINE configurations, timed by the same run; the primary region above is untouched
            out["secondary"] = {}
            for wl in ("fhd_stress", "depth16"):
                out["secondary"][wl] = secondary_leg(ssd, scenes, torch, device, wl, steps=10, warmup=args.warmup, check_frames=4 if wl == "fhd_stress" else 9,
                                                     with_cpu=not args.no_cpu)
        if world == 1 and not args.no_hostfed and not fhd:
            det.close()
            del frames
            torch.cuda.empty_cache()
            out["host_fed"] = host_fed_leg(ssd, scenes, device=device)
            out["host_fed"]["gpu_numa_node"] = where.get("numa_node")
            out["host_fed"]["note"] = ("frames in host memory through ssd_process_host / ssd_process_depth_host (PCIe-inclusive, "
                                       "double-buffered ingest); NOT the metric `value`, which is measured on frames resident in HBM.  Every source "
                                       "is timed twice, in the order pinned, pageable, pageable, pinned, behind two warm-up calls each; the figure is "
                                       "the better leg, legs_GBps holds both, numa_node_of_source the node of the source's first page (ssd_host_alloc "
                                       "= hipHostMalloc: wherever the calling thread's policy puts it) beside gpu_numa_node")
        print(json.dumps(out), flush=True)

    det.close()
    ranks.close()


if __name__ == "__main__":
    main()
