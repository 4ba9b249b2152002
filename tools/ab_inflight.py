"""tools/ab_inflight.py DIR_A DIR_B ... — frames/s of the plain handle with three batches in flight (1024 XGA frames per call, 24 calls)
and K1's one-at-a-time stage time, alternately for the library builds stair-step-detector_amd/DIR_*; each measurement in a
process of its own (a library is loaded once per process)."""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import importlib, os, sys, time
R = %r
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=3), ssd.transformation_for_scene(sc[0]), 0)
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
def run(n, ahead=2):
    for i in range(n):
        det.enqueue(buf.ptr, F)
        if i >= ahead: det.fetch(F, back=ahead)
    for back in range(min(ahead, n) - 1, -1, -1): det.fetch(F, back=back)
run(5); ssd.lib().ssd_device_sync(0)
t0 = time.perf_counter(); run(24); ssd.lib().ssd_device_sync(0); dt = time.perf_counter() - t0
det.set_timing(True)
k1 = 0.0
for i in range(7):
    det.enqueue(buf.ptr, F); det.fetch(F)
    if i: k1 += det.stage_times_ms()["hist"] / 6
print("%%.0f %%.3f" %% (24 * F / dt, k1))
""" % R
dirs = sys.argv[1:]
out = {d: [] for d in dirs}
for rnd in range(4):
    for d in dirs:
        env = dict(os.environ, SSD_HIP_LIB=os.path.join(R, "stair-step-detector_amd", d, "libssd_hip.so"))
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        out[d].append(p.stdout.strip() or p.stderr[-200:])
for d in dirs:
    print(d, out[d])
