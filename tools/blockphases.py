"""tools/blockphases.py — where the blocks of k_raster / k_inquad spend their lives, summed over all waves of one launch on
1024 XGA frames (GPU box).  Needs the tools-only build:
    make -C stair-step-detector_amd/csrc OUT=../lib_phase EXTRA=-DSSD_PHASE_TIMING ../lib_phase/libssd_hip.so
Prints per kernel: waves, mean wave life (us) and its split over the marked phases."""
import ctypes as C, importlib, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
os.environ.setdefault("SSD_HIP_LIB", os.path.join(R, "stair-step-detector_amd", "lib_phase", "libssd_hip.so"))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NAMES = {0: ("k_raster", {0: "tables+clear", 1: "cell list", 2: "walk", 5: "flush", 3: "barrier wait", 4: "sums out"}),
         1: ("k_inquad", {0: "tables", 1: "cell list", 2: "walk", 3: "barrier wait", 4: "sums out"})}
sc = scenes.batch_scenes(ssd, 1024, 768, F, base_seed=100000, rng_seed=1000)
det = ssd.Detector(ssd.default_config(1024, 768, max_frames_per_batch=F, batches_in_flight=1), ssd.transformation_for_scene(sc[0]), 0)
det.set_timing(True)
buf = ssd.DeviceBuffer(1024 * 768 * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
L = ssd.lib()
out = (C.c_ulonglong * (4 * 64 * 8))()
for it in range(3):
    det.enqueue(buf.ptr, F); det.fetch(F)
    assert L.ssd_blockphase_read(out) == 0
print("stage ms:", {k: round(v, 3) for k, v in det.stage_times_ms().items()})
for k, (name, labels) in NAMES.items():
    col = lambda i: sum(out[(k * 64 + c) * 8 + i] for c in range(64))
    waves = col(7)
    if not waves:
        continue
    tot = sum(col(i) for i in labels)
    print("%s: %d waves, mean wave life %.2f us" % (name, waves, tot / waves / 100.0))
    for i, lab in labels.items():
        print("   %-14s %7.2f us  %5.1f %%" % (lab, col(i) / waves / 100.0, 100.0 * col(i) / tot))
