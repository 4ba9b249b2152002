"""The frame of SURVEY.md Appendix A and the values the survey observed for it.

TEST INFRASTRUCTURE.  SANITY ANCHOR, NOT A PIN: the survey obtained these numbers from the reference's own
pointcloud.cpp / segmentation.cpp / transformation.cpp compiled against stand-in headers for Boost.QVM, OpenCV
and librealsense — a build this repository does not (and must not) make.
SURVEY.md Appendix A quotes the first-tread height 0.17005879162516252, the corner x -0.40078125000000003 and
the 3-decimal line; the remaining digits below are the same run's full-precision output.
"""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

WORLD_POINTS = np.array([-0.5, 1.2, 0.0, 0.5, 1.2, 0.0, 0.4, 0.3, 0.0])

# height, then front-left, front-right, back-left, back-right (x, y) in the external world
OBSERVED = np.array([
    [4.0658096429012029e-05, -0.40078125000000003, 0.22343749999999996, 0.39960937499999993, 0.22343749999999996,
     -0.40078125000000003, 0.44999999999999996, 0.39960937499999988, 0.44999999999999996],
    [0.17005879162516252, -0.40078125000000003, 0.44999999999999996, 0.39960937499999988, 0.44999999999999996,
     -0.40078125000000003, 0.73125000000000007, 0.39960937499999988, 0.73125000000000007],
    [0.34004724137684084, -0.40078125000000003, 0.72968750000000004, 0.39843749999999994, 0.72968750000000004,
     -0.40078125000000003, 1.0125, 0.39843749999999994, 1.0125],
    [0.50953343572393495, -0.39960937500000004, 1.0093749999999999, 0.39843749999999994, 1.0093749999999999,
     -0.39960937500000004, 1.1828125, 0.39843749999999994, 1.1828125],
])

OBSERVED_LINE = ('["stairs",["stairSteps",4],[[["height",0.000],["quadrilateral",[-0.401,0.223],[0.400,0.223],[-0.401,0.450],[0.400,0.450]]],'
                 '[["height",0.170],["quadrilateral",[-0.401,0.450],[0.400,0.450],[-0.401,0.731],[0.400,0.731]]],'
                 '[["height",0.340],["quadrilateral",[-0.401,0.730],[0.398,0.730],[-0.401,1.012],[0.398,1.012]]],'
                 '[["height",0.510],["quadrilateral",[-0.400,1.009],[0.398,1.009],[-0.400,1.183],[0.398,1.183]]]]]')


def _generator(tmp_path):
    exe = os.path.join(str(tmp_path), "survey_probe_frame")
    if not os.path.exists(exe):
        subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(HERE, "golden", "survey_probe_frame.cpp"), "-o", exe], check=True)
    return exe


def frame(tmp_path, width=1024, height=768, case=None):
    """Compiles tests/golden/survey_probe_frame.cpp (once per tmp_path), runs it; returns (xyz float32[W*H*3], camera
    points[9]).  `case`: one entry of tests/golden/survey_probe_lines.json (scene knobs)."""
    exe = _generator(tmp_path)
    out = os.path.join(str(tmp_path), "frame.bin")
    args = [exe, str(width), str(height), out]
    if case is not None:
        args = [exe, str(case["width"]), str(case["height"]), out, str(case["steps"]), repr(case["sigma"]), repr(case["cam_height"]),
                repr(case["pitch_deg"]), repr(case["first_riser_y"]), repr(case["tread"]), repr(case["rise"]), repr(case["half_width"]),
                repr(case["outliers"])]
    p = subprocess.run(args, check=True, capture_output=True, text=True)
    cam = np.array(p.stdout.split(), dtype=np.float64)
    assert cam.shape == (9,)
    return np.fromfile(out, dtype=np.float32), cam


def probe_cases():
    """The 39 further runs of the survey's probe binaries (see the file's own header): parameters + printed line."""
    import json
    return json.load(open(os.path.join(HERE, "golden", "survey_probe_lines.json")))["cases"]
