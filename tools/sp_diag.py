"""tools/sp_diag.py [FRAMES] — the single pass frame by frame on an XGA batch: the predictor's planes against the plateaus found"""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F), ssd.transformation_for_scene(sc[0]), 0)
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
det.enqueue(buf.ptr, F); res = det.fetch_list(F)
print(det.single_pass_stats(F))
shown = 0
for i in range(F):
    table, planes, covered, steps = det.single_pass_frame(i)
    if covered or shown >= 12:
        continue
    shown += 1
    raw, lay = det.frame_state(i)
    hist = np.frombuffer(raw, dtype=np.uint32, count=ssd.MAX_BINS, offset=lay["hist"])
    lut = np.frombuffer(raw, dtype=np.uint8, count=ssd.MAX_BINS, offset=lay["lut"])
    print("frame", i, "planes", planes, "steps", steps, "plane of bin", [(int(b), int(table[b])) for b in np.nonzero(table != 255)[0]])
    print("   plateau of bin", [(int(b), int(lut[b])) for b in np.nonzero(lut != 255)[0]])
    print("   hist", [(b, int(hist[b])) for b in range(80) if hist[b] > 5000 or (b > 0 and hist[b-1] > 5000) or hist[b+1] > 5000])
