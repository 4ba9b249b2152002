#!/usr/bin/env python3
"""tools/inflight.py — batches in flight inside ONE handle (ssd_config::batches_in_flight) against the same overlap across
handles (ssd_pipeline_*), same process, alternating, XGA.  For each frames-per-batch F and depth D: frames/s of
  handle_null   the plain handle API, caller's stream = the null stream
  handle_own    the plain handle API, caller's stream = a non-blocking stream of the caller's
  pipeline      ssd_pipeline_* at the same depth
Prints one JSON object; copy it to profiles/ when it is to be cited.
"""
import ctypes as C
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

W, H = 1024, 768
hip = C.CDLL("libamdhip64.so")
own = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(own), 1) == 0
out = {}
for F in [int(a) for a in sys.argv[1:]] or (64, 256, 1024):
    sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
    trans = ssd.transformation_for_scene(sc[0])
    buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
    ssd.synth_device(sc, buf.ptr, device=0)
    reps = max(12, 8192 // F)
    for rnd in range(2):
        for depth in (1, 2, 3, 4):
            cfg = ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=depth)
            for name, stream in (("handle_null", None), ("handle_own", own.value)):
                det = ssd.Detector(cfg, trans, 0)
                ahead = max(depth, 2) - 1

                def run(n):
                    for i in range(n):
                        det.enqueue(buf.ptr, F, stream=stream)
                        if i >= ahead:
                            det.fetch(F, back=ahead)
                    for back in range(min(ahead, n) - 1, -1, -1):
                        det.fetch(F, back=back)

                run(depth + 1)
                ssd.lib().ssd_device_sync(0)
                t0 = time.perf_counter()
                run(reps)
                ssd.lib().ssd_device_sync(0)
                out.setdefault("F%d_depth%d_%s" % (F, depth, name), []).append(round(reps * F / (time.perf_counter() - t0)))
                det.close()
            pipe = ssd.Pipeline(ssd.default_config(W, H, max_frames_per_batch=F), trans, 0, depth=depth)

            def prun(n):
                for i in range(n):
                    if pipe.pending() == depth:
                        pipe.next(copy=False)
                    pipe.submit(buf.ptr, F)
                while pipe.pending():
                    pipe.next(copy=False)

            prun(depth + 1)
            ssd.lib().ssd_device_sync(0)
            t0 = time.perf_counter()
            prun(reps)
            ssd.lib().ssd_device_sync(0)
            out.setdefault("F%d_depth%d_pipeline" % (F, depth), []).append(round(reps * F / (time.perf_counter() - t0)))
            pipe.close()
    buf.free()
print(json.dumps(out))
