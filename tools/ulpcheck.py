#!/usr/bin/env python3
"""tools/ulpcheck.py — where do device and oracle doubles differ in the last bit? (diagnostic, GPU box)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
ssd = importlib.import_module("stair-step-detector_amd")
import oracle_binding as ob, scenes
oracle = ob.load_oracle()
F = 1024
sc_list = scenes.batch_scenes(ssd, 1024, 768, F, base_seed=100000, rng_seed=1000)
trans = ssd.transformation_for_scene(sc_list[0]); cfg = ssd.default_config(1024, 768, max_frames_per_batch=1)
intr = ssd.intrinsics_for_scene(sc_list[0])
det = ssd.Detector(cfg, trans, 0); det.set_intrinsics(intr); det.set_debug(True)
ocfg, ocal = ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants)
for i in list(range(0, F, 64)) + [F - 1]:
    depth = ssd.synth_depth_host([sc_list[i]])[0]
    fr = det.process_depth_host(depth)[0]; dbg = det.debug(0)
    res, *_ = oracle.process(ocfg, ocal, oracle.deproject(intr, depth))
    for k in range(res.n_plateaus):
        d, o = dbg.plateaus[k], res.plateaus[k]
        if not o.is_step or not o.outline_found: continue
        for name in ("bounds", "base_line", "vline", "quad_img", "quad_world"):
            a = np.array(getattr(d, name), dtype=np.float64).ravel(); b = np.array(getattr(o, name), dtype=np.float64).ravel()
            if not np.array_equal(a, b):
                j = np.nonzero(a != b)[0]
                print("frame", i, "plateau", k, name, "idx", j.tolist(), "dev", a[j].tolist(), "ora", b[j].tolist(), "best", list(d.best_pt[0]), list(o.best_pt[0]), list(d.best_pt[1]), list(o.best_pt[1]))
    for s in range(res.n_steps):
        a = np.array(fr.steps[s].quad); b = np.array(res.steps_ext[s][1:9])
        if not np.array_equal(a, b):
            print("frame", i, "step", s, "quad differs", (a - b).tolist())
print("done")
