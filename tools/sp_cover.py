"""tools/sp_cover.py — single pass on 1024 XGA frames of three scene seeds: coverage of the predictor and the stage times"""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
for seed in (100000, 200000, 300000):
    sc = scenes.batch_scenes(ssd, W, H, F, base_seed=seed, rng_seed=seed // 100)
    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F), ssd.transformation_for_scene(sc[0]), 0)
    ssd.synth_device(sc, buf.ptr, device=0)
    det.set_timing(True)
    acc = {}
    for i in range(5):
        det.enqueue(buf.ptr, F); det.fetch(F)
        if i:
            for k, v in det.stage_times_ms().items():
                acc[k] = acc.get(k, 0.0) + v / 4
    st = det.single_pass_stats(F)
    print(os.path.basename(os.path.dirname(os.environ.get("SSD_HIP_LIB", "lib/x"))), seed, "covered %d / %d planes %d" % (st["covered"], st["with_steps"], st["planes"]),
          " ".join("%s %.3f" % (k[:5], v) for k, v in acc.items()))
    det.close()
