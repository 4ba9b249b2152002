import importlib, sys, time, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H = 1024, 768
out = {}
for F in (64, 256, 1024):
    sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
    trans = ssd.transformation_for_scene(sc[0])
    buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
    ssd.synth_device(sc, buf.ptr, device=0)
    cfg = ssd.default_config(W, H, max_frames_per_batch=F)
    reps = max(10, 4096 // F)
    for depth in (1, 2, 3, 4):
        pipe = ssd.Pipeline(cfg, trans, 0, depth=depth)
        def run(n):
            for i in range(n):
                if pipe.pending() == depth:
                    pipe.next(copy=False)
                pipe.submit(buf.ptr, F)
            while pipe.pending():
                pipe.next(copy=False)
        run(depth + 1)
        ssd.lib().ssd_device_sync(0)
        t0 = time.perf_counter()
        run(reps)
        ssd.lib().ssd_device_sync(0)
        out["F%d_depth%d" % (F, depth)] = round(reps * F / (time.perf_counter() - t0))
        pipe.close()
    buf.free()
print(json.dumps(out))
