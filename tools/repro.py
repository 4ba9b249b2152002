#!/usr/bin/env python3
"""tools/repro.py — re-runs the failures of a tools/fuzz.py summary (JSON on stdin or file) with full intermediates."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import oracle_binding as ob, parity
oracle = ob.load_oracle()
d = json.load(open(sys.argv[1]))
for f in d["failures"]:
    W, H = f["res"]
    sc = ssd.make_scene(W, H, **f["scene"])
    trans = ssd.transformation_for_scene(sc)
    cfg = ssd.default_config(W, H, max_frames_per_batch=1)
    det = ssd.Detector(cfg, trans, 0)
    xyz = ssd.synth_host([sc])[0]
    try:
        rep = parity.check_frame(ssd, oracle, det, cfg, trans.constants, xyz, images=True)
        print("frame", f["frame"], "OK in single-frame debug mode", rep)
    except parity.Mismatch as e:
        print("frame", f["frame"], "MISMATCH:", str(e)[:600])
    det.set_debug(True)
    det.process_host(xyz)
    dbg = det.debug(0)
    res, *_ = oracle.process(ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants), xyz)
    for k in range(res.n_plateaus):
        o, g = res.plateaus[k], dbg.plateaus[k]
        print(" plateau", k, "peak", o.peak_bin, "valid", o.valid, g.valid, "n_in_quad", o.n_in_quad, g.n_in_quad, "mean_z %.17g %.17g" % (o.mean_z, g.mean_z),
              "quad_world", [float("%.17g" % v) for v in o.quad_world])
    det.close()
