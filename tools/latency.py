import importlib, sys, time, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
sc = scenes.batch_scenes(ssd, 1024, 768, 8, base_seed=4242)
xyz = ssd.synth_host(sc)
for nf in (1, 8):
    one = ssd.Detector(ssd.default_config(1024, 768, max_frames_per_batch=nf), ssd.transformation_for_scene(sc[0]), 0)
    buf = ssd.DeviceBuffer(xyz[0].nbytes * nf, 0)
    buf.upload(xyz[:nf])
    one.set_timing(True)
    for _ in range(5):
        one.enqueue(buf.ptr, nf); one.fetch(nf)
    te = tf = 0.0; st = {}
    N = 100
    for i in range(N):
        t0 = time.perf_counter(); one.enqueue(buf.ptr, nf); t1 = time.perf_counter(); one.fetch(nf); t2 = time.perf_counter()
        te += t1 - t0; tf += t2 - t1
        for k, v in one.stage_times_ms().items(): st[k] = st.get(k, 0) + v / N
    print(nf, "frames: enqueue cpu %.1f us, fetch wait %.1f us, total %.1f us; gpu stages (ms):" % (te / N * 1e6, tf / N * 1e6, (te + tf) / N * 1e6), {k: round(v, 4) for k, v in st.items()}, "sum %.4f" % sum(st.values()))
    one.set_timing(False)
    t0 = time.perf_counter()
    for i in range(N):
        one.enqueue(buf.ptr, nf); one.fetch(nf)
    print("   without event timing: %.1f us per call" % ((time.perf_counter() - t0) / N * 1e6))
