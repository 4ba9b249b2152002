"""tools/sp_frames.py [W H F,F,..] — from how many frames per batch does the single pass pay?  XGA, three batches in flight, two passes against
the single pass forced on, alternating"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 768)
for F in ([int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (8, 16, 32, 48, 64, 128)):
    sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=3), ssd.transformation_for_scene(sc[0]), 0)
    buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
    ssd.synth_device(sc, buf.ptr, device=0)
    def run(n, ahead=2):
        for i in range(n):
            det.enqueue(buf.ptr, F)
            if i >= ahead: det.fetch(F, back=ahead)
        for back in range(min(ahead, n) - 1, -1, -1): det.fetch(F, back=back)
    out = []
    for rnd in range(3):
        for mode in (0, 1):
            det.single_pass(mode)
            run(20); ssd.lib().ssd_device_sync(0)
            n = max(40, 4096 // F)
            t0 = time.perf_counter(); run(n); ssd.lib().ssd_device_sync(0); dt = time.perf_counter() - t0
            out.append("%s %.0f" % ("single" if mode else "two", n * F / dt))
    print(F, out)
    det.close(); buf.free()
