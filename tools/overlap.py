#!/usr/bin/env python3
"""tools/overlap.py — does splitting a batch over two streams (two handles, half the frames each) overlap the HBM-bound
K1 of one half with the VALU-bound K2/K4 of the other?  Experiment; prints frames/s for 1 and 2 and 4 streams."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
F, W, H = 1024, 1024, 768
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
trans = ssd.transformation_for_scene(sc[0])
frames = torch.empty(F * W * H * 12, dtype=torch.uint8, device="cuda")
ssd.synth_device(sc, frames.data_ptr(), device=0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for parts in (1, 2, 4, 8):
    n = F // parts
    dets = [ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=n), trans, 0) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def step():
        for k in range(parts):
            dets[k].enqueue(frames.data_ptr() + k * n * W * H * 12, n, stream=streams[k].cuda_stream)
        out = []
        for k in range(parts):
            out.append(dets[k].fetch(n, stream=streams[k].cuda_stream))
        return out
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%d stream(s) x %d frames: %.3f ms per %d frames = %.0f frames/s" % (parts, n, dt * 1e3, F, F / dt), flush=True)
    for d in dets: d.close()
