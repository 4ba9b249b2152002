"""tools/place_overlap.py [N] — throughput with three batches in flight (1024 XGA frames per call) against WHERE the three workspaces'
cell records lie: one process, one input buffer, one handle; the record arrays of all workspaces re-allocated N times (test hook).
How much of the run-to-run spread of bench.py's `value` is placement."""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
trans = ssd.transformation_for_scene(sc[0])
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=3), trans, 0)
def run(n, ahead=2):
    for i in range(n):
        det.enqueue(buf.ptr, F)
        if i >= ahead: det.fetch(F, back=ahead)
    for back in range(min(ahead, n) - 1, -1, -1): det.fetch(F, back=back)
for k in range(N):
    if k:
        det.record_realloc()
    run(6); ssd.lib().ssd_device_sync(0)
    rates = []
    for rep in range(3):
        t0 = time.perf_counter(); run(24); ssd.lib().ssd_device_sync(0)
        rates.append(24 * F / (time.perf_counter() - t0))
    print("placement %d: %s frames/s" % (k, " ".join("%.0f" % r for r in rates)), flush=True)
