"""tools/stages.py [frames] [reps] — per-stage device times (ms, HIP events between the launches) of one handle fed one batch at
a time, XGA; honours SSD_HIP_LIB (build variants) and, in -DSSD_TUNING builds, the SSD_* geometry variables.  One line."""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
W, H = (1920, 1080) if os.environ.get("STAGES_FHD") else (1024, 768)
sc = scenes.fhd_stress_scenes(ssd, F, base_seed=9000) if os.environ.get("STAGES_FHD") else scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=1), ssd.transformation_for_scene(sc[0]), 0)
det.set_timing(True)
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
acc = {k: 0.0 for k in ssd.STAGE_NAMES}
for it in range(reps + 3):
    det.enqueue(buf.ptr, F); res = det.fetch(F)
    if it >= 3:
        for k, v in det.stage_times_ms().items():
            acc[k] += v / reps
tag = os.environ.get("STAGES_TAG", "")
print("%-44s total %.3f  %s  steps %d" % (tag, sum(acc.values()), " ".join("%s %.3f" % (k[:4], v) for k, v in acc.items()), sum(r.n_steps for r in res)))
