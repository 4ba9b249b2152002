"""tools/stride_states.py STRIDE_PAD_BYTES [HOLD_MIB] — K1's time with the frames STRIDE_PAD_BYTES apart beyond their own size (a multiple of 16), optionally
with HOLD_MIB of device memory held in front of them: does the frames' stride (9 MiB exactly for XGA vertices: every frame's chunk k on the same low address
bits) have a say in K1's two states?  One line."""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
pad = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hold = int(sys.argv[2]) << 20 if len(sys.argv) > 2 else 0
W, H, F = 1024, 768, 1024
stride = W * H * 12 + pad
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
trans = ssd.transformation_for_scene(sc[0])
held = ssd.DeviceBuffer(hold, 0) if hold else None
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=1), trans, 0)
buf = ssd.DeviceBuffer(stride * F, 0)
ssd.synth_device(sc, buf.ptr, stride_bytes=stride, device=0)
det.set_timing(True)
t, n = [], 0
for i in range(8):
    det.enqueue(buf.ptr, F, stride_bytes=stride); res = det.fetch(F)
    if i >= 3:
        t.append(det.stage_times_ms()["hist"])
n = sum(r.n_steps for r in res)
print("stride pad %7d B  hold %5d MiB  K1 %.3f ms  (steps %d)" % (pad, hold >> 20, sum(t) / len(t), n), flush=True)
