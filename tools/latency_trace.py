"""tools/latency_trace.py [nframes] — run under `rocprofv3 --kernel-trace` on the GPU box: 30 single-handle calls of a small
batch without the library's timing events, so that the trace's begin / end timestamps give the kernels' own durations
and the gaps between them.  tools/latency_gaps.py condenses the trace."""
import importlib, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import scenes
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sc = scenes.batch_scenes(ssd, 1024, 768, max(nf, 1), base_seed=4242)
xyz = ssd.synth_host(sc)
one = ssd.Detector(ssd.default_config(1024, 768, max_frames_per_batch=nf), ssd.transformation_for_scene(sc[0]), 0)
buf = ssd.DeviceBuffer(xyz[0].nbytes * nf, 0)
buf.upload(xyz[:nf])
for _ in range(30):
    one.enqueue(buf.ptr, nf); one.fetch(nf)
