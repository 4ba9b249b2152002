import importlib, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
sc = scenes.batch_scenes(ssd, 1024, 768, 4, base_seed=4242)
xyz = ssd.synth_host(sc)
one = ssd.Detector(ssd.default_config(1024, 768, max_frames_per_batch=1), ssd.transformation_for_scene(sc[0]), 0)
buf = ssd.DeviceBuffer(xyz[0].nbytes, 0); buf.upload(xyz[0])
one.set_timing(True)
for i in range(8):
    one.enqueue(buf.ptr, 1); one.fetch(1)
acc = {}
for b in range(4):
    for k, v in one.stage_times_ms(b).items(): acc[k] = acc.get(k, 0) + v / 4
print({k: round(v * 1e3, 1) for k, v in acc.items()}, "us; sum", round(sum(acc.values()) * 1e3, 1))
one.set_timing(False)
t0 = time.perf_counter()
for i in range(200):
    one.enqueue(buf.ptr, 1); one.fetch(1)
print("latency us", (time.perf_counter() - t0) / 200 * 1e6)
