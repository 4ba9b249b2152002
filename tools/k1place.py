"""tools/k1place.py — K1's time against WHERE its cell records lie: one input buffer, one handle, the records shifted inside their
allocation (test hook ssd_test_record_offset) by the listed byte offsets; two rounds.  Behind DESIGN.md's "placement" paragraph."""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
W, H, F = 1024, 768, 1024
sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
trans = ssd.transformation_for_scene(sc[0])
offsets = [int(a, 0) for a in sys.argv[1:]] or [0, 4096, 65536, 1 << 20, 8 << 20]
for b in range(2):
    buf = ssd.DeviceBuffer(W * H * 12 * F, 0)
    ssd.synth_device(sc, buf.ptr, device=0)
    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=F, batches_in_flight=1), trans, 0)
    det.set_timing(True)
    det.record_realloc(max(offsets), 0)                   # an allocation with room for the largest offset
    for rnd in range(2):
        row = []
        for off in offsets:
            det.record_offset(off)
            t = []
            for i in range(4):
                det.enqueue(buf.ptr, F); det.fetch(F)
                if i >= 1: t.append(det.stage_times_ms()["hist"])
            row.append("%d:%.3f" % (off >> 10, min(t)))
        print("input %d round %d (offset KiB : K1 ms)  %s" % (b, rnd, "  ".join(row)), flush=True)
    # the same handle, the same input: only the array of the cell records is allocated anew, eight times
    det.record_offset(0)
    row = []
    for k in range(8):
        addr = det.record_realloc() if k else 0
        t = []
        for i in range(4):
            det.enqueue(buf.ptr, F); det.fetch(F)
            if i >= 1: t.append(det.stage_times_ms()["hist"])
        row.append("%x:%.3f" % (addr >> 20, min(t)))
    print("input %d at %x MiB, records re-allocated (address MiB : K1 ms)  %s" % (b, buf.ptr >> 20, "  ".join(row)), flush=True)
    det.close(); buf.free()
