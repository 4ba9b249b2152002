#!/usr/bin/env python3
"""tools/two_handles.py [F ...] — mid-size batches: one handle on one stream against TWO handles (each with its own workspace)
on two streams fed alternately.  With a few dozen frames per batch the small kernels (outline, quads, final: one block per
frame or image, latency-bound) leave the GPU nearly empty; a second handle's batch fills it.  GPU box, via gpurun."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes  # noqa: E402
import torch  # noqa: E402

W, H = 1024, 768
out = {}
for F in [int(a) for a in sys.argv[1:]] or [16, 64, 256]:
    sc = scenes.batch_scenes(ssd, W, H, 2 * F, base_seed=100000, rng_seed=1000)
    trans = ssd.transformation_for_scene(sc[0])
    buf = ssd.DeviceBuffer(W * H * 12 * 2 * F, 0)
    ssd.synth_device(sc, buf.ptr, device=0)
    half = W * H * 12 * F
    streams = [torch.cuda.Stream(device=0), torch.cuda.Stream(device=0)]
    dets = [ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F), trans, 0) for _ in range(2)]
    reps = max(8, 2048 // F)

    def run(n_handles):
        for k in range(n_handles):                      # warm up
            dets[k].enqueue(buf.ptr + k * half, F, stream=streams[k].cuda_stream)
            dets[k].fetch(F)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            k = i % n_handles
            if i >= n_handles:
                dets[k].fetch(F)                          # the batch this handle ran last
            dets[k].enqueue(buf.ptr + (i % 2) * half, F, stream=streams[k].cuda_stream)
        for k in range(n_handles):
            dets[k].fetch(F)
        torch.cuda.synchronize()
        return reps * F / (time.perf_counter() - t0)

    for d in dets:
        d.set_timing(True)
    one = run(1)
    st1 = {k: round(sum(dets[0].stage_times_ms(b)[k] for b in range(4)) / 4, 4) for k in ssd.STAGE_NAMES}
    two = run(2)
    st2 = {k: round(sum(dets[j].stage_times_ms(b)[k] for j in range(2) for b in range(2)) / 4, 4) for k in ssd.STAGE_NAMES}
    out[F] = {"one_handle_frames_per_s": round(one), "two_handles_frames_per_s": round(two), "gain": round(two / one, 3),
              "stage_ms_one": st1, "stage_ms_two": st2}
    for d in dets:
        d.close()
    buf.free()
print(json.dumps(out))
