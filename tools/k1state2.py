"""tools/k1state2.py — does K1's slow state follow the INPUT buffer, the HANDLE's buffers, or the process?  One process: three input
buffers (1024 XGA frames each, same content) x five handles, all alive at once; K1's time for every pair."""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
trans = ssd.transformation_for_scene(sc[0])
bufs = []
for b in range(3):
    buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
    ssd.synth_device(sc, buf.ptr, device=0)
    bufs.append(buf)
dets = []
for d in range(5):
    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=1), trans, 0)
    det.set_timing(True)
    dets.append(det)
print("input buffers at", [hex(b.ptr) for b in bufs])
for rnd in range(2):
    for bi, buf in enumerate(bufs):
        row = []
        for det in dets:
            t = []
            for i in range(5):
                det.enqueue(buf.ptr, F); det.fetch(F)
                if i >= 2: t.append(det.stage_times_ms()["hist"])
            row.append(min(t))
        print("round %d input %d: K1 ms per handle %s   stream %.3f" % (rnd, bi, " ".join("%.3f" % x for x in row), ssd.stream_read_ms(buf.ptr, W * H * 12 * F, reps=3, device=0)), flush=True)
