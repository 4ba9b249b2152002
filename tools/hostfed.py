#!/usr/bin/env python3
"""tools/hostfed.py — PCIe-inclusive throughput of ssd_process_host (frames in host memory), reported
beside bench.py's HBM-resident number.  Never the bench `value`."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sc = scenes.batch_scenes(ssd, 1024, 768, n, base_seed=4242)
xyz = ssd.synth_host(sc)
det = ssd.Detector(ssd.default_config(1024, 768, max_frames_per_batch=n), ssd.transformation_for_scene(sc[0]), 0)
det.process_host(xyz)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    det.process_host(xyz)
dt = (time.perf_counter() - t0) / reps
one = ssd.Detector(ssd.default_config(1024, 768, max_frames_per_batch=1), ssd.transformation_for_scene(sc[0]), 0)
one.process_host(xyz[0])
t0 = time.perf_counter()
for i in range(32):
    one.process_host(xyz[i % n])
lat = (time.perf_counter() - t0) / 32
# one frame already resident in HBM: enqueue + fetch (7 launches + one 1.2 KB result copy)
buf = ssd.DeviceBuffer(xyz[0].nbytes, 0)
buf.upload(xyz[0])
one.enqueue(buf.ptr, 1)
one.fetch(1)
t0 = time.perf_counter()
for i in range(64):
    one.enqueue(buf.ptr, 1)
    one.fetch(1)
lat_dev = (time.perf_counter() - t0) / 64
print({"single_frame_latency_ms_device_resident": lat_dev * 1e3,
       "host_fed_frames_per_s": n / dt, "h2d_GBps_equiv": n * xyz[0].nbytes / dt / 1e9,
       "single_frame_latency_ms_incl_h2d": lat * 1e3})
