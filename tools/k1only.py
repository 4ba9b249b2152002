"""tools/k1only.py — K1 alone (ssd_enqueue_stages(HIST | PEAKS)) and the plain read stream, alternately; honours SSD_HIP_LIB."""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
F = 1024
sc = scenes.batch_scenes(ssd, 1024, 768, F, base_seed=100000, rng_seed=1000)
det = ssd.Detector(ssd.default_config(1024, 768, max_frames_per_batch=F, batches_in_flight=1), ssd.transformation_for_scene(sc[0]), 0)
det.set_timing(True)
buf = ssd.DeviceBuffer(1024 * 768 * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
k1 = st = 0.0
N = 10
for it in range(N + 2):
    s = ssd.stream_read_ms(buf.ptr, 1024 * 768 * 12 * F, reps=3)
    det.enqueue(buf.ptr, F, stages=ssd.STAGE_HIST | ssd.STAGE_PEAKS); ssd.lib().ssd_device_sync(0)
    if it >= 2:
        k1 += det.stage_times_ms()["hist"] / N; st += s / N
print("%-24s K1 %.3f ms  plain stream %.3f ms  ratio %.3f" % (os.environ.get("STAGES_TAG", ""), k1, st, st / k1))
