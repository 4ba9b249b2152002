"""tools/sp_time.py — per-stage times of 1024 XGA frames, one batch at a time: two-pass pipeline, single pass with a predictor that
gives no planes (k_predict + K1's idle raster code), single pass"""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F), ssd.transformation_for_scene(sc[0]), 0)
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
det.set_timing(True)
for rnd in range(2):
    for name, mode, sab in (("two-pass", 0, 0), ("no planes", -1, 2), ("single", -1, 0)):
        det.single_pass(mode, sab)
        acc = {}
        for i in range(7):
            det.enqueue(buf.ptr, F); det.fetch(F)
            if i:
                for k, v in det.stage_times_ms().items():
                    acc[k] = acc.get(k, 0.0) + v / 6
        print("%-10s" % name, " ".join("%s %.3f" % (k[:5], v) for k, v in acc.items()), det.single_pass_stats(F))
