"""tools/k1place2.py [N] — K1's time for N freshly allocated arrays of cell records against one input buffer (addresses in hex):
looking for the rule behind the two states."""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
EXTRA = int(sys.argv[2]) << 20 if len(sys.argv) > 2 else 0          # MiB added to every allocation
OFF = int(sys.argv[3]) << 20 if len(sys.argv) > 3 else 0            # the records this many MiB into it
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
trans = ssd.transformation_for_scene(sc[0])
for b in range(2):
    buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
    ssd.synth_device(sc, buf.ptr, device=0)
    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=1), trans, 0)
    det.set_timing(True)
    print("input %d at 0x%x" % (b, buf.ptr))
    for k in range(N):
        addr = det.record_realloc(EXTRA, OFF)
        t = []
        for i in range(4):
            det.enqueue(buf.ptr, F); det.fetch(F)
            if i >= 1: t.append(det.stage_times_ms()["hist"])
        # the region by itself: a plain read stream over the records' 100 MB, and a memset of them (20 in a row)
        rd = ssd.stream_read_ms(addr, 96 << 20, reps=5, device=0)
        import ctypes, time
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipDeviceSynchronize()
        c0 = time.perf_counter()
        for _ in range(20):
            hip.hipMemsetAsync(ctypes.c_void_p(addr), 0, ctypes.c_size_t(96 << 20), None)
        hip.hipDeviceSynchronize()
        wr = (time.perf_counter() - c0) / 20 * 1e3
        print("  records 0x%x  delta %+d MiB  K1 %.3f %-4s  region alone: read %.1f GB/s  memset %.1f GB/s" % (addr, (addr - buf.ptr) >> 20, min(t), "SLOW" if min(t) > float(os.environ.get("K1_SLOW_MS", "1.87")) else "", (96 << 20) / rd / 1e6, (96 << 20) / wr / 1e6), flush=True)
    det.close(); buf.free()
    ssd.hooks_lib().ssd_test_record_release()
