"""tools/k1place3.py [N] — all stage times (one batch at a time) for N freshly allocated arrays of cell records against one input
buffer: which stages follow the records' placement, and how far the whole pass moves."""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
trans = ssd.transformation_for_scene(sc[0])
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=1), trans, 0)
det.set_timing(True)
print("input at 0x%x" % buf.ptr)
for k in range(N):
    addr = det.record_realloc() if k else 0
    acc = {n: 1e9 for n in ssd.STAGE_NAMES}
    tot = 1e9
    for i in range(5):
        det.enqueue(buf.ptr, F); det.fetch(F)
        if i >= 1:
            st = det.stage_times_ms()
            tot = min(tot, sum(st.values()))
            for n in acc: acc[n] = min(acc[n], st[n])
    print("records 0x%012x  total %.3f  %s" % (addr, tot, " ".join("%s %.3f" % (n[:4], v) for n, v in acc.items())), flush=True)
