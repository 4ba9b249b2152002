"""tools/ab_env.py "ENV=VAL .." "ENV=VAL .." ... — frames/s of the plain handle with three batches in flight (1024 XGA frames per call, 24
calls), alternately for environment settings of the tools build (lib_tuning: SSD_* variables, ssd_handle.h); each measurement in a
process of its own; AB_DEPTH = batches in flight (default 3), AB_ROUNDS (default 4)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import importlib, os, sys, time
R = %r
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
depth = int(os.environ.get("AB_DEPTH", "3"))
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=depth), ssd.transformation_for_scene(sc[0]), 0)
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
ahead = max(depth, 2) - 1
def run(n):
    for i in range(n):
        det.enqueue(buf.ptr, F)
        if i >= ahead: res = det.fetch(F, back=ahead)
    for back in range(min(ahead, n) - 1, -1, -1): res = det.fetch(F, back=back)
    return res
run(6); ssd.lib().ssd_device_sync(0)
t0 = time.perf_counter(); res = run(30); ssd.lib().ssd_device_sync(0); dt = time.perf_counter() - t0
print("%%.0f (steps %%d)" %% (30 * F / dt, sum(r.n_steps for r in res)))
""" % R
settings = sys.argv[1:] or [""]
out = {s: [] for s in settings}
for rnd in range(int(os.environ.get("AB_ROUNDS", "4"))):
    for st in settings:
        env = dict(os.environ, SSD_HIP_LIB=os.path.join(R, "stair-step-detector_amd", "lib_tuning", "libssd_hip.so"))
        env.pop("AB_LIB", None)
        for kv in st.split():
            k, v = kv.split("=", 1); env[k] = v
            if k == "AB_LIB":            # another build variant: stair-step-detector_amd/<v>/libssd_hip.so
                env["SSD_HIP_LIB"] = os.path.join(R, "stair-step-detector_amd", v, "libssd_hip.so")
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        out[st].append(p.stdout.strip() or p.stderr[-300:])
for st in settings:
    print("%-40s %s" % (st or "(default)", out[st]))
