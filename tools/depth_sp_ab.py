"""tools/depth_sp_ab.py [rounds] — 16-bit depth input, 1024 XGA frames resident in HBM, three batches in flight: the two-pass pipeline (what a depth
batch runs by default) against the single pass forced on it (ssd_set_single_pass's forced mode), alternating on one handle; frames/s over 20 enqueues
each and the stage times one batch at a time.  VERDICT round 5, item 4: decide the single pass on depth input on evidence from this build."""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=3), ssd.transformation_for_scene(sc[0]), 0)
det.set_intrinsics(ssd.intrinsics_for_scene(sc[0]))
buf = ssd.DeviceBuffer(W * H * 2 * F, 0)
ssd.synth_depth_device(sc, buf.ptr, device=0)
one = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=1), ssd.transformation_for_scene(sc[0]), 0)
one.set_intrinsics(ssd.intrinsics_for_scene(sc[0]))
one.set_timing(True)
ref = None
for rnd in range(rounds):
    for name, mode in (("two passes", 0), ("single pass", 1)):
        det.single_pass(mode, 0)
        for i in range(4):                                   # warm-up
            det.enqueue_depth(buf.ptr, F)
            if i >= 2:
                det.fetch(F, back=2)
        det.fetch(F, back=1); det.fetch(F, back=0)
        t0 = time.perf_counter()
        n = 20
        for i in range(n):
            det.enqueue_depth(buf.ptr, F)
            if i >= 2:
                det.fetch(F, back=2)                         # two batches ahead of the one that is read
        det.fetch(F, back=1)
        res = det.fetch(F, back=0)
        dt = time.perf_counter() - t0
        lines = [bytes(res[i]) for i in range(0, F, 16)]
        if ref is None:
            ref = lines
        assert lines == ref, "results differ between the modes"
        one.single_pass(mode, 0)
        acc = {}
        for i in range(5):
            one.enqueue_depth(buf.ptr, F); one.fetch(F)
            if i:
                for k, v in one.stage_times_ms().items():
                    acc[k] = acc.get(k, 0.0) + v / 4
        print("%-12s %8.0f frames/s  (%.3f ms per batch)   one at a time: %s  ran single pass: %s" % (
            name, n * F / dt, dt / n * 1e3, " ".join("%s %.3f" % (k[:5], v) for k, v in acc.items()), one.single_pass_stats(F)["ran"]), flush=True)
