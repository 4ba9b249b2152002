"""tools/sp_prof.py [MODE SABOTAGE] — 8 batches of 1024 XGA frames one at a time, for rocprofv3 --kernel-trace --stats"""
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
if len(sys.argv) > 2:
    det.single_pass(int(sys.argv[1]), int(sys.argv[2]))
for i in range(8):
    det.enqueue(buf.ptr, F); det.fetch(F)
