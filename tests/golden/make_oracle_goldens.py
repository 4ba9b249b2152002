#!/usr/bin/env python3
"""tests/golden/make_oracle_goldens.py — writes tests/golden/oracle_goldens.json: the CPU oracle's own output, at full
precision, for the 31 named scenes of tests/scenes.py and the 40 frames of the survey's probe (tests/survey_anchor.py).

What this pins: the ORACLE against itself.  Every GPU parity test compares the HIP path with the oracle as built at
test time; an accidental edit of oracle/ssd_oracle.cpp would move that reference silently.  With these vectors
committed, such an edit turns tests/test_oracle_goldens.py red on the CPU, and the HIP path is checked against the
same frozen numbers on the GPU.  What it does NOT do: pin the oracle to the reference — only output of the
reference's own pointcloud.cpp / segmentation.cpp / transformation.cpp could, and those cannot be built in this image
(DESIGN.md section 5).

Per frame: SHA-256 of the input cloud (so a drift of the frame generators is told apart from a drift of the oracle), the
calibration constants, counts, hist[n_bins], peaks, the plateau table, ground / first valid index, status, the steps as
9 IEEE doubles each in hex (height, 4 corners in external world coordinates) and the 3-decimal line.

Run from the repository root after building (python -c 'import __graft_entry__ as g; g.build()'):
    python tests/golden/make_oracle_goldens.py
"""
import hashlib
import importlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

OUT = os.path.join(HERE, "oracle_goldens.json")


def hexd(v):
    return float(v).hex()


def record(oracle, ob, ocfg, ocal, xyz):
    """one frame through the oracle -> the golden record (plain JSON types)"""
    a = np.ascontiguousarray(xyz, dtype=np.float32)
    res, *_ = oracle.process(ocfg, ocal, a)
    nb = res.n_bins
    rec = {
        "sha256": hashlib.sha256(a.tobytes()).hexdigest(),
        "calibration": [hexd(v) for v in list(ocal.a) + list(ocal.b) + list(ocal.r2) + list(ocal.t2) + [ocal.world_z]],
        "status": res.status, "n_nonzero": res.n_nonzero, "n_inrange": res.n_inrange, "n_oob": res.n_oob, "n_bins": nb,
        "hist": [int(v) for v in res.hist[:nb]],
        "peaks": [int(v) for v in res.peaks[:res.n_peaks]],
        "plateaus": [[p.peak_bin, p.bin_lo, p.bin_hi, p.n_points, p.is_step, p.outline_found, p.valid, p.n_in_quad]
                     for p in (res.plateaus[i] for i in range(res.n_plateaus))],
        "ground_ind": res.ground_ind, "first_valid_ind": res.first_valid_ind, "ground_n_in_quad": res.ground_n_in_quad,
        "n_steps": res.n_steps,
        "steps": [[hexd(v) for v in res.steps_ext[i]] for i in range(res.n_steps)],
        "line": res.line.decode(),
    }
    return rec


def frames(ssd, tmp):
    """yields (name, width, height, xyz float32, (world points, camera points) or None) for all 71 golden frames"""
    import scenes
    import survey_anchor
    for name in sorted(scenes.scene_params()):
        sc = scenes.make(ssd, name)
        yield "scene:" + name, sc.width, sc.height, ssd.synth_host([sc])[0], ssd.calibration_points(sc)
    xyz, cam = survey_anchor.frame(tmp)
    yield "probe:appendix_a", 1024, 768, xyz, (survey_anchor.WORLD_POINTS, cam)
    for k, case in enumerate(survey_anchor.probe_cases()):
        xyz, cam = survey_anchor.frame(tmp, case=case)
        yield "probe:%02d" % k, case["width"], case["height"], xyz, (survey_anchor.WORLD_POINTS, cam)


def build(ssd, oracle, ob, tmp):
    out = {}
    for name, w, h, xyz, (world, cam) in frames(ssd, tmp):
        ocfg = oracle.config(w, h)
        rc, ocal = oracle.calibration(world, cam)
        assert rc == 0, name
        rec = record(oracle, ob, ocfg, ocal, xyz)
        rec["width"], rec["height"] = w, h
        out[name] = rec
    return out


def main():
    ssd = importlib.import_module("stair-step-detector_amd")
    import oracle_binding as ob
    oracle = ob.load_oracle()
    with tempfile.TemporaryDirectory() as tmp:
        frames_out = build(ssd, oracle, ob, tmp)
    doc = {"what": "output of oracle/ssd_oracle.cpp (the CPU restatement) frozen at full precision; regression pin of the oracle "
                   "itself, NOT a pin against the reference (see the generating script's header)",
           "generator": "tests/golden/make_oracle_goldens.py", "frames": frames_out}
    json.dump(doc, open(OUT, "w"), indent=0, sort_keys=True)
    print("%d frames -> %s (%d bytes)" % (len(frames_out), OUT, os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
