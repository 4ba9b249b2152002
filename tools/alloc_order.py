"""tools/alloc_order.py A|B [pad_mib] — K1's state by the ORDER of the process's allocations: A = the input buffer first, then the handle's workspaces;
B = the handle first, then the input (tools/stages.py's order).  One line: K1 ms (one batch at a time, 1024 XGA frames)."""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
order = sys.argv[1] if len(sys.argv) > 1 else "A"
pad = int(sys.argv[2]) << 20 if len(sys.argv) > 2 else 0
W, H, F = 1024, 768, 1024
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
trans = ssd.transformation_for_scene(sc[0])
cfg = ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=int(os.environ.get("LANES", "1")))
padbuf = ssd.DeviceBuffer(pad, 0) if pad else None
if order == "A":
    buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
    det = ssd.Detector(cfg, trans, 0)
else:
    det = ssd.Detector(cfg, trans, 0)
    buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
det.set_timing(True)
t = []
for i in range(8):
    det.enqueue(buf.ptr, F); det.fetch(F)
    if i >= 3:
        t.append(det.stage_times_ms()["hist"])
print("order %s pad %4d MiB  input at 0x%x  K1 %.3f ms" % (order, pad >> 20, buf.ptr, sum(t) / len(t)), flush=True)
