"""tools/sp_diag_fhd.py — the single pass frame by frame on the FHD stress batch: the predictor's planes against the plateaus found"""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1920, 1080, 64
sc = scenes.fhd_stress_scenes(ssd, F, base_seed=9000)
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F), ssd.transformation_for_scene(sc[0]), 0)
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
det.set_debug(True, images=False)
det.enqueue(buf.ptr, F); res = det.fetch_list(F)
det.set_debug(False)
det.enqueue(buf.ptr, F); res = det.fetch_list(F)
print(det.single_pass_stats(F))
for i in range(3):
    table, planes, covered, steps = det.single_pass_frame(i)
    raw, lay = det.frame_state(i)
    hist = np.frombuffer(raw, dtype=np.uint32, count=ssd.MAX_BINS, offset=lay["hist"])
    lut = np.frombuffer(raw, dtype=np.uint8, count=ssd.MAX_BINS, offset=lay["lut"])
    print("frame", i, "planes", planes, "covered", covered, "steps", steps)
    print("   plane of bin", [(int(b), int(table[b])) for b in np.nonzero(table != 255)[0]])
    print("   plateau of bin", [(int(b), int(lut[b])) for b in np.nonzero(lut != 255)[0]])
    print("   hist", hist[:100].tolist())
