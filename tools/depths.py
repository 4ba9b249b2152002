"""tools/depths.py — frames/s of ssd_pipeline_* (= a handle with as many workspaces, tools/inflight.py) by depth, alternating, XGA."""
import importlib, json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H = 1024, 768
out = {}
for F in [int(a) for a in sys.argv[1:]] or (256, 1024):
    sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
    trans = ssd.transformation_for_scene(sc[0])
    buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
    ssd.synth_device(sc, buf.ptr, device=0)
    cfg = ssd.default_config(W, H, max_frames_per_batch=F)
    reps = max(16, 16384 // F)
    for rnd in range(3):
        for depth in (1, 2, 3, 4, 5, 6, 8):
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
            out.setdefault("F%d_depth%d" % (F, depth), []).append(round(reps * F / (time.perf_counter() - t0)))
            pipe.close()
    buf.free()
print(json.dumps(out))
