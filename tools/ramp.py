"""tools/ramp.py — how much of a 20-step timed region is the GPU's clock ramp?  After 3 s of idle: optionally a plain read
stream for `pre` seconds, then 5 warm-up + 20 timed steps of the 1024-frame batch through the plain handle (3 batches in
flight), as bench.py does.  Prints frames/s per variant, three rounds."""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F), ssd.transformation_for_scene(sc[0]), 0)
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
ahead = 2
def run(n):
    for i in range(n):
        det.enqueue(buf.ptr, F)
        if i >= ahead: det.fetch(F, back=ahead)
    for back in range(min(ahead, n) - 1, -1, -1): det.fetch(F, back=back)
for rnd in range(3):
    for pre in (0.0, 0.1, 0.3, 1.0, 3.0):
        time.sleep(3.0)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < pre:
            ssd.stream_read_ms(buf.ptr, W * H * 12 * F, reps=5)
        run(5)
        ssd.lib().ssd_device_sync(0)
        t0 = time.perf_counter()
        run(20)
        ssd.lib().ssd_device_sync(0)
        dt = time.perf_counter() - t0
        print("round %d  pre-stream %.1f s: %.0f frames/s (%.3f ms per step)" % (rnd, pre, 20 * F / dt, dt / 20 * 1e3), flush=True)
