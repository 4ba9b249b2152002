"""tools/pipeline_timing.py — the pipeline (depth 1-4, 1024 XGA frames per batch) with per-stage timing events on every handle:
frames/s and the stages' mean device times while the batches overlap (GPU box)."""
import importlib, sys, time, json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
trans = ssd.transformation_for_scene(sc[0])
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
cfg = ssd.default_config(W, H, max_frames_per_batch=F)
out = {}
for timing in (False, True):
    for depth in (1, 2, 3, 4):
        pipe = ssd.Pipeline(cfg, trans, 0, depth=depth)
        pipe.set_timing(timing)
        st = {}
        got = [0]
        def run(n, collect):
            for i in range(n):
                if pipe.pending() == depth:
                    pipe.next(copy=False)
                    if collect and timing:
                        for k, v in pipe.stage_times_ms().items():
                            st[k] = st.get(k, 0.0) + v
                        got[0] += 1
                pipe.submit(buf.ptr, F)
            while pipe.pending():
                pipe.next(copy=False)
        run(depth + 2, False)
        ssd.lib().ssd_device_sync(0)
        reps = 24
        t0 = time.perf_counter()
        run(reps, True)
        ssd.lib().ssd_device_sync(0)
        dt = time.perf_counter() - t0
        out["depth%d_%s" % (depth, "events" if timing else "plain")] = {"frames_per_s": round(reps * F / dt),
            **({"stage_ms": {k: round(v / max(got[0], 1), 3) for k, v in st.items()}} if timing else {})}
        pipe.close()
print(json.dumps(out, indent=1))
