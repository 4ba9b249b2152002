"""The oracle against its own committed full-precision output (tests/golden/oracle_goldens.json, generator
tests/golden/make_oracle_goldens.py): a regression pin of the checker, so that an edit of oracle/ssd_oracle.cpp cannot
move the reference of the GPU parity tests unnoticed.  `-m gpu`: the HIP path against the same frozen vectors.
NOT a pin against the reference (only reference-built output could be)."""
import importlib.util
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
import parity

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "oracle_goldens.json")


def _generator():
    spec = importlib.util.spec_from_file_location("make_oracle_goldens", os.path.join(HERE, "golden", "make_oracle_goldens.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _goldens():
    return json.load(open(GOLDEN))["frames"]


def test_golden_set_is_the_71_frames():
    g = _goldens()
    assert len(g) == 71
    assert sum(1 for k in g if k.startswith("scene:")) == 31 and sum(1 for k in g if k.startswith("probe:")) == 40
    assert {r["n_steps"] for r in g.values()} >= {0, 4}                    # empty lines and full staircases
    assert any(r["status"] & ob.ST_THROW for r in g.values())               # would-have-thrown frames are in the set


def test_oracle_reproduces_its_committed_goldens(ssd, oracle, tmp_path):
    """every field, every frame: integers equal, doubles identical (hex), the line byte for byte"""
    gen = _generator()
    golden = _goldens()
    live = gen.build(ssd, oracle, ob, tmp_path)
    assert sorted(live) == sorted(golden)
    for name in sorted(golden):
        g, l = golden[name], live[name]
        assert l["sha256"] == g["sha256"], "%s: the frame generator no longer reproduces this input cloud" % name
        for key in sorted(g):
            assert l[key] == g[key], "%s: oracle field %r moved" % (name, key)


@pytest.mark.gpu
def test_hip_path_reproduces_the_oracle_goldens(ssd, gpu_device, tmp_path):
    """the HIP path (debug capture on, through the C ABI) against the frozen vectors — no live oracle involved:
    counts, histogram, peaks, plateau table, in-quad counts exact; corners identical doubles; heights within the
    declared 1e-9 m; the serialized line byte for byte"""
    gen = _generator()
    golden = _goldens()
    worst_h = 0.0
    for name, w, h, xyz, (world, cam) in gen.frames(ssd, tmp_path):
        g = golden[name]
        trans = ssd.GeometricTransformation(world, cam)
        cal = trans.constants
        got = [float(v).hex() for v in list(cal.a) + list(cal.b) + list(cal.r2) + list(cal.t2) + [cal.world_z]]
        assert got == g["calibration"], name
        cfg = ssd.default_config(w, h, max_frames_per_batch=1)
        det = ssd.Detector(cfg, trans, gpu_device)
        det.set_debug(True)
        fr = det.process_host(np.ascontiguousarray(xyz, dtype=np.float32).reshape(h, w, 3))[0]
        d = det.debug(0)
        assert (fr.status, d.n_nonzero, d.n_inrange, d.n_oob, d.n_bins) == (g["status"], g["n_nonzero"], g["n_inrange"], g["n_oob"], g["n_bins"]), name
        assert list(d.hist[:g["n_bins"]]) == g["hist"], name
        assert list(d.peaks[:d.n_peaks]) == g["peaks"], name
        assert d.n_plateaus == len(g["plateaus"]) and d.ground_ind == g["ground_ind"] and d.first_valid_ind == g["first_valid_ind"], name
        for k, gp in enumerate(g["plateaus"]):
            p = d.plateaus[k]
            assert [p.peak_bin, p.bin_lo, p.bin_hi, p.n_points, p.is_step] == gp[:5], "%s plateau %d" % (name, k)
            if not gp[4]:
                continue                            # outline and validity exist for step plateaus only (the oracle marks the ground itself)
            assert [p.outline_found, p.valid] == gp[5:7], "%s plateau %d" % (name, k)
            if gp[6] and not (g["status"] & ob.ST_THROW) and g["first_valid_ind"] >= 0 and k >= g["first_valid_ind"]:
                assert p.n_in_quad == gp[7], "%s plateau %d: in-quadrilateral count" % (name, k)
        if g["first_valid_ind"] >= 0 and g["ground_ind"] >= 0 and not (g["status"] & ob.ST_THROW):
            assert d.ground_n_in_quad == g["ground_n_in_quad"], name
        assert fr.n_steps == g["n_steps"], name
        for i, gs in enumerate(g["steps"]):
            want = [float.fromhex(v) for v in gs]
            assert [float(v).hex() for v in fr.steps[i].quad] == gs[1:], "%s step %d: corners" % (name, i)
            err = abs(fr.steps[i].height - want[0])
            assert err <= parity.TOL_HEIGHT, "%s step %d: height" % (name, i)
            worst_h = max(worst_h, err)
        if not (g["status"] & ob.ST_THROW):
            assert ssd.Stairs(fr).serialize() == g["line"], name
        det.close()
    assert worst_h <= parity.TOL_HEIGHT
