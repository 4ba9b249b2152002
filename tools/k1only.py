"""tools/k1only.py [FRAMES] — K1 alone (ssd_enqueue_stages(HIST | PEAKS)) and the plain read stream, alternately; honours SSD_HIP_LIB.
With 16 - 24 frames the input (9.4 MB a frame) stays in the 256 MiB Infinity Cache between launches: K1 fed from there."""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = scenes.batch_scenes(ssd, 1024, 768, F, base_seed=100000, rng_seed=1000)
det = ssd.Detector(ssd.default_config(1024, 768, max_frames_per_batch=F, batches_in_flight=1), ssd.transformation_for_scene(sc[0]), 0)
det.set_timing(True)
buf = ssd.DeviceBuffer(1024 * 768 * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
k1 = st = 0.0
N = 10 if F >= 256 else 40
for it in range(N + 2):
    s = ssd.stream_read_ms(buf.ptr, 1024 * 768 * 12 * F, reps=3)
    det.enqueue(buf.ptr, F, stages=ssd.STAGE_HIST | ssd.STAGE_PEAKS); ssd.lib().ssd_device_sync(0)
    if it >= 2:
        k1 += det.stage_times_ms()["hist"] / N; st += s / N
print("%-24s %4d frames  K1 %.4f ms (%.3f us / frame)  plain stream %.4f ms (%.3f us / frame)  ratio %.3f" % (os.environ.get("STAGES_TAG", ""), F, k1, 1e3 * k1 / F, st, 1e3 * st / F, st / k1))
