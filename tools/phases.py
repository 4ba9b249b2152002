"""tools/phases.py — phase clocks of k_outline / k_quads / k_final for one resident XGA frame (GPU box).
Needs the tools-only build: make -C stair-step-detector_amd/csrc OUT=../lib_phase EXTRA=-DSSD_PHASE_TIMING ../lib_phase/libssd_hip.so
and SSD_HIP_LIB=stair-step-detector_amd/lib_phase/libssd_hip.so.  Prints the mean over 20 calls of the time between
consecutive markers (block (0, 0) of each kernel: frame 0, image slot 0), in microseconds."""
import ctypes as C, importlib, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
os.environ.setdefault("SSD_HIP_LIB", os.path.join(R, "stair-step-detector_amd", "lib_phase", "libssd_hip.so"))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
NAMES = {0: ("k_outline", ["fs loads", "init", "closing+extents", "scans+lists", "BestLine", "bounds+base", "probes", "compaction",
                           "rank", "ties", "corners+store", "clear"]),
         1: ("k_quads", ["entry", "loads+ballots", "ground quad", "-", "build_quad_test", "table+segs", "lutLive", "wanted+tail"]),
         2: ("k_final", ["entry", "closing+extents", "scan", "BestLine", "surfaces+results", "risers+sync", "clear"])}
FHD = len(sys.argv) > 1 and sys.argv[1] == "fhd"          # python tools/phases.py fhd: one frame of the FHD stress workload (config 5)
sc = scenes.fhd_stress_scenes(ssd, 1, base_seed=9000) if FHD else scenes.batch_scenes(ssd, 1024, 768, 1, base_seed=4242)
xyz = ssd.synth_host(sc)
one = ssd.Detector(ssd.default_config(sc[0].width, sc[0].height, max_frames_per_batch=1), ssd.transformation_for_scene(sc[0]), 0)
buf = ssd.DeviceBuffer(xyz[0].nbytes, 0)
buf.upload(xyz[:1])
L = ssd.lib()
acc = {}
N = 20
for it in range(N + 5):
    one.enqueue(buf.ptr, 1); one.fetch(1)
    out = (C.c_ulonglong * (4 * 32))()
    rc = L.ssd_phase_read(out)
    assert rc == 0, rc
    if it < 5:
        continue
    for k, (name, labels) in NAMES.items():
        t = [out[k * 32 + i] for i in range(len(labels))]
        for i in range(1, len(labels)):
            acc[(k, i)] = acc.get((k, i), 0.0) + (t[i] - t[i - 1]) / 100.0 / N
for k, (name, labels) in NAMES.items():
    tot = sum(acc[(k, i)] for i in range(1, len(labels)))
    print("%s: %.2f us between first and last marker" % (name, tot))
    for i in range(1, len(labels)):
        print("   %-22s %6.2f" % (labels[i], acc[(k, i)]))
