"""tools/k1state.py PAD [PAD ..] — K1's duration (one batch of 1024 XGA frames at a time, events around the launch) in fresh processes
(fresh allocations), several per setting of SSD_RECORD_PAD (cell records added to the stride between the frames' record arrays;
tools build: lib_tuning).  Behind the question whether K1's "slow state" (its record store costing 0.2 ms instead of 0.08) belongs to
the box or to where an allocation's pages lie.  Prints per setting the sorted K1 times and the plain stream beside them."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import importlib, os, sys
R = %r
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=1), ssd.transformation_for_scene(sc[0]), 0)
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
det.set_timing(True)
k1 = []
for i in range(9):
    det.enqueue(buf.ptr, F); det.fetch(F)
    if i >= 3: k1.append(det.stage_times_ms()["hist"])
s = ssd.stream_read_ms(buf.ptr, W * H * 12 * F, reps=5, device=0)
print("%%.3f %%.3f %%.3f" %% (min(k1), sum(k1) / len(k1), s))
""" % R
reps = int(os.environ.get("K1STATE_REPS", "6"))
for pad in sys.argv[1:] or ["0"]:
    env = dict(os.environ, SSD_HIP_LIB=os.path.join(R, "stair-step-detector_amd", "lib_tuning", "libssd_hip.so"), SSD_RECORD_PAD=pad)
    got = []
    for r in range(reps):
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        got.append(p.stdout.strip() or p.stderr[-200:])
    print("pad %6s:\n   %s" % (pad, "\n   ".join(got)), flush=True)
