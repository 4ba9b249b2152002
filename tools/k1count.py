"""tools/k1count.py [fhd] — what K1's per-wave image windows do on the bench's frames (a -DSSD_COUNT build: make OUT=../lib_count EXTRA=-DSSD_COUNT):
wave-tiles, wave-tiles with candidate points, window moves (flush + re-anchor), words flushed, candidate pixels, pixels that missed the window."""
import ctypes as C, importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.environ.setdefault("SSD_HIP_LIB", os.path.join(R, "stair-step-detector_amd", "lib_count", "libssd_hip.so"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
fhd = len(sys.argv) > 1 and sys.argv[1] == "fhd"
W, H, F = (1920, 1080, 256) if fhd else (1024, 768, 1024)
sc = scenes.fhd_stress_scenes(ssd, F, base_seed=9000) if fhd else scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=1), ssd.transformation_for_scene(sc[0]), 0)
buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
ssd.synth_device(sc, buf.ptr, device=0)
out = (C.c_ulonglong * 8)()
L = ssd.lib()
det.enqueue(buf.ptr, F); det.fetch(F)
L.ssd_tools_k1_counters(out)
det.enqueue(buf.ptr, F); det.fetch(F)
L.ssd_tools_k1_counters(out)
t, tc, mv, fw, px, miss, cp = [int(v) & 0xffffffff for v in out[:7]]
print("%s: wave-tiles %d, with candidates %d (%.1f %%), window moves %d (one per %.2f candidate tiles), words flushed %d (%.1f per move), "
      "candidate points %d (%.1f per candidate tile), pixels %d, missed the window %d (%.2f %%)"
      % ("FHD stress" if fhd else "XGA", t, tc, 100.0 * tc / max(t, 1), mv, tc / max(mv, 1), fw, fw / max(mv, 1), cp, cp / max(tc, 1), px, miss, 100.0 * miss / max(px, 1)))
