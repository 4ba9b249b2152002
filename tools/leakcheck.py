"""tools/leakcheck.py — device memory before and after many handles / batches of the kinds the fuzz sweep makes"""
import ctypes as C, importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
hip = C.CDLL("libamdhip64.so")
def free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value
ssd.lib().ssd_device_sync(0)
base = None
for rnd in range(12):
    for (W, H, F, depth) in ((640, 480, 384, True), (1024, 768, 384, False), (600, 450, 384, False), (1920, 1080, 48, True)):
        sc = scenes.batch_scenes(ssd, W, H, 4, base_seed=1000 + rnd, rng_seed=rnd)
        cfg = ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=3)
        det = ssd.Detector(cfg, ssd.transformation_for_scene(sc[0]), 0)
        if rnd % 2:
            det.set_risers(True, 0.03, 200)
        scl = (sc * (F // 4 + 1))[:F]
        if depth:
            det.set_intrinsics(ssd.intrinsics_for_scene(sc[0]))
            buf = ssd.DeviceBuffer(F * W * H * 2, 0)
            ssd.synth_depth_device(scl, buf.ptr, device=0)
            det.enqueue_depth(buf.ptr, F)
        else:
            buf = ssd.DeviceBuffer(F * W * H * 12, 0)
            ssd.synth_device(scl, buf.ptr, device=0)
            det.enqueue(buf.ptr + W * H * 12, F - 1); det.fetch_list(F - 1)
            det.enqueue(buf.ptr, F)
        det.fetch_list(F)
        det.close(); buf.free()
    fb = free_bytes()
    if base is None:
        base = fb
    print("round", rnd, "free GiB %.3f" % (fb / 2**30), "delta MiB %.1f" % ((fb - base) / 2**20), flush=True)
