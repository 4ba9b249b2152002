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

PyTorch is plumbing here: device memory, the stream, the barrier and the max-reduce.  The hot path is
libssd_hip.so through its C ABI.  The CPU oracle is used only for the cpu_baseline leg and the parity
spot check, both outside the timed region.
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
METRIC = "point-cloud frames/sec (1024×768 pts) at 1/2/4/8 GPUs; step height/corner max-abs err"
METRIC_FHD = "point-cloud frames/sec (1920×1080 pts, stress); step height/corner max-abs err"


def shard(total, world, rank):
    """Contiguous frame range of `rank`: frame i -> rank i*world//total."""
    lo = (total * rank + world - 1) // world
    hi = (total * (rank + 1) + world - 1) // world
    return lo, hi


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
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--risers", action="store_true",
                    help="also gather the evidence of the vertical faces (extension beyond the reference, SURVEY 8f rank 4); off for the metric")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and world > 1:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the HIP path is mandatory (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    ssd = importlib.import_module("stair-step-detector_amd")
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
    cfg = ssd.default_config(W, H, max_frames_per_batch=F)
    depth_in = args.input == "depth16"
    if depth_in:
        frame_bytes = W * H * 2
    frames = torch.empty(F * frame_bytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    det = ssd.Detector(cfg, trans, local_rank)
    det.set_timing(True)
    if args.risers:
        det.set_risers(True, tolerance=0.03, min_support=200)
    intr = ssd.intrinsics_for_scene(sc_list[0])
    if depth_in:
        ssd.synth_depth_device(sc_list, frames.data_ptr(), device=local_rank, stream=stream)
        det.set_intrinsics(intr)
    else:
        ssd.synth_device(sc_list, frames.data_ptr(), device=local_rank, stream=stream)

    def enqueue():
        if depth_in:
            det.enqueue_depth(frames.data_ptr(), F, stream=stream)
        else:
            det.enqueue(frames.data_ptr(), F, stream=stream)

    def run(n_steps):
        """n_steps passes over the batch, every pass's results fetched to the host: pass i+1 is enqueued before the
        results of pass i are read (they travel with their own enqueue), so the GPU never waits for the host."""
        res = None
        for i in range(n_steps):
            enqueue()
            if i > 0:
                res = det.fetch(F, back=1)
        return det.fetch(F) if n_steps > 0 else res

    res = run(args.warmup)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    res = run(args.steps)
    fence()
    dt = time.perf_counter() - t0

    # per-stage device times of the timed steps (HIP events recorded on the kernels' stream)
    n_timed = min(args.steps, 63)
    stage = {k: 0.0 for k in ssd.STAGE_NAMES}
    for b in range(n_timed):
        for k, v in det.stage_times_ms(b).items():
            stage[k] += v / n_timed

    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    if rank == 0:
        value = world * F * args.steps / dt_max
        k1_ms = stage["hist"]
        alg_bytes = (2.0 if depth_in else 12.0) * W * H * F    # 12 B per raw point (2 B per depth pixel), read once
        achieved = alg_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
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
        out = {
            "metric": METRIC_FHD if fhd else METRIC, "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[4]: %d synthetic %dx%d frames (8 noisy steps, 5 %% outliers) resident in HBM; per GPU"
                                    if fhd else
                                    "BASELINE configs[2]: batch of %d synthetic %dx%d frames (3-step staircases, randomised rise/"
                                    "tread/yaw/noise) resident in HBM, streamed through the whole per-frame path; per GPU") % (F, W, H),
                       "frames_per_gpu_per_step": F, "width": W, "height": H, "input": args.input, "risers": bool(args.risers), "parallelism": "frame-sharded x%d, no collective" % world},
            "roofline": {"bound": "hbm", "kernel": "k_hist (K1: transform+crop+bin+histogram)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": k1_ms},
            "stage_ms": stage,
            "pipeline_bytes_algorithmic_frac_of_peak": (alg_bytes * args.steps / dt_max / 1e9) / HBM_PEAK_GBS,
        }
        steps_hist = [r.n_steps for r in res]
        out["steps_histogram"] = {str(k): int(sum(1 for n in steps_hist if n == k)) for k in sorted(set(steps_hist))}

        if not args.no_cpu:
            import oracle_binding as ob
            import parity
            oracle = ob.load_oracle()
            ocfg, ocal = ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants)
            n_cpu = max(1, min(args.cpu_frames or (256 if fhd else 1280), F))
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
            out["cpu_baseline"] = {"value": n_cpu / cdt, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "%d of the %d frames of rank 0's batch (evenly spaced), the whole per-frame path in "
                                             "oracle/ssd_oracle.cpp (CPU restatement of the reference, one thread as the reference runs), "
                                             "%.1f s of CPU time" % (n_cpu, F, cdt)}
            # the same port on all host cores, one frame per thread (ctypes drops the GIL)
            cores = len(os.sched_getaffinity(0))
            if cores > 1 and len(keep) > 1:
                from concurrent.futures import ThreadPoolExecutor
                reps = max(1, (4 * cores + len(keep) - 1) // len(keep))
                work = keep * reps
                with ThreadPoolExecutor(cores) as pool:
                    c0 = time.perf_counter()
                    list(pool.map(lambda a: oracle.process_lean(ocfg, ocal, a), work))
                    adt = time.perf_counter() - c0
                out["cpu_baseline_all_cores"] = {"value": len(work) / adt, "unit": "frames/s", "cores": cores, "kind": "port",
                                                 "sample": "%d frames (%d distinct), one frame per thread, %.1f s wall" % (len(work), len(keep), adt)}
            out["parity"] = {"frames_checked_against_oracle": checked, "max_abs_height_err_m": rep.get("max_height_err", 0.0),
                             "max_abs_corner_err_m": rep.get("max_corner_err", 0.0), "bar_m": 1e-4}
        print(json.dumps(out), flush=True)

    det.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
