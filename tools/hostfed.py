#!/usr/bin/env python3
"""tools/hostfed.py [frames] — PCIe-inclusive throughput of the host-fed entry points (bench.host_fed_leg) and the
single-frame latencies, reported beside bench.py's HBM-resident number.  Never the bench `value`."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import bench  # noqa: E402
import scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = bench.host_fed_leg(ssd, scenes, n_frames=n)
sc = scenes.batch_scenes(ssd, 1024, 768, 8, base_seed=4242)
xyz = ssd.synth_host(sc)
one = ssd.Detector(ssd.default_config(1024, 768, max_frames_per_batch=1), ssd.transformation_for_scene(sc[0]), 0)
one.process_host(xyz[0])
t0 = time.perf_counter()
for i in range(32):
    one.process_host(xyz[i % 8])
out["single_frame_latency_ms_incl_h2d_pageable"] = (time.perf_counter() - t0) / 32 * 1e3
# one frame already resident in HBM: enqueue + fetch
buf = ssd.DeviceBuffer(xyz[0].nbytes, 0)
buf.upload(xyz[0])
one.enqueue(buf.ptr, 1)
one.fetch(1)
t0 = time.perf_counter()
for i in range(64):
    one.enqueue(buf.ptr, 1)
    one.fetch(1)
out["single_frame_latency_ms_device_resident"] = (time.perf_counter() - t0) / 64 * 1e3
print(json.dumps(out))
