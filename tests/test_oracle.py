"""CPU tests of the oracle: pinned against the real reference where the reference compiles here
(stairs.cpp, quadrilateralTest.cpp -> oracle/_ref), against the committed golden vectors those produced,
and cross-checked against independent restatements (scipy morphology, brute-force best line)."""
import itertools
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _unhex(lst, shape):
    return np.array([float.fromhex(x) for x in lst], dtype=np.float64).reshape(shape)


# ------------------------------------------------------------------ sanity anchor: the survey's probe run
def test_oracle_reproduces_the_surveys_observed_run(oracle, tmp_path):
    """SURVEY.md Appendix A's frame through the oracle: every printed digit of the values the survey observed from the
    reference (stand-in build; an anchor, not a pin — see tests/survey_anchor.py)."""
    import survey_anchor as sa
    xyz, cam = sa.frame(tmp_path)
    rc, cal = oracle.calibration(sa.WORLD_POINTS, cam)
    assert rc == 0
    n, steps, status = oracle.process_lean(oracle.config(1024, 768), cal, xyz)
    assert (n, status) == (4, 0)
    assert ["%.17g" % v for v in steps.reshape(-1)] == ["%.17g" % v for v in sa.OBSERVED.reshape(-1)]
    assert oracle.serialize(steps) == sa.OBSERVED_LINE


def test_oracle_reproduces_39_further_lines_of_the_surveys_probe(oracle, tmp_path):
    """VGA / XGA / FHD x 0-8 steps, noise 0-5 mm, up to 20 % outliers, other camera poses and stair geometries: the
    oracle's serialized line against what the survey's probe binaries printed (anchors, not pins: tests/golden/
    survey_probe_lines.json says where they come from)."""
    import survey_anchor as sa
    cases = sa.probe_cases()
    assert len(cases) == 39
    n_with_steps = 0
    for c in cases:
        xyz, cam = sa.frame(tmp_path, case=c)
        rc, cal = oracle.calibration(sa.WORLD_POINTS, cam)
        assert rc == 0
        n, steps, status = oracle.process_lean(oracle.config(c["width"], c["height"]), cal, xyz)
        assert status == 0
        line = oracle.serialize(steps) if n > 0 else '["stairs",["stairSteps",0]]'
        assert line == c["line"], (c, line)
        n_with_steps += 1 if n > 0 else 0
    assert n_with_steps >= 30


# ------------------------------------------------------------------ golden vectors from the real reference
def test_serialize_matches_reference_goldens(oracle):
    cases = json.load(open(os.path.join(HERE, "golden", "ref_serialize.json")))
    assert len(cases) >= 8
    for c in cases:
        steps = _unhex(c["steps"], (c["n"], 9))
        assert oracle.serialize(steps) == c["line"]


def test_quadtest_matches_reference_goldens(oracle):
    cases = json.load(open(os.path.join(HERE, "golden", "ref_quadtest.json")))
    assert len(cases) >= 70
    n_inside = 0
    for c in cases:
        quad = _unhex(c["quad"], (4, 2))
        pts = _unhex(c["pts"], (-1, 2))
        rc, inside = oracle.quad_test(quad, pts)
        assert rc == c["rc"]
        if rc == 0:
            assert "".join(str(int(v)) for v in inside) == c["inside"]
            n_inside += int(inside.sum())
    assert n_inside > 500


# ------------------------------------------------------------------ live against oracle/_ref (when built)
def test_serialize_live_against_reference(oracle, ref):
    rng = np.random.default_rng(11)
    for n in (0, 1, 3, 7, 17, 40):
        steps = rng.normal(0, 3, (n, 9))
        steps[rng.random(steps.shape) < 0.1] = -0.0
        assert oracle.serialize(steps) == ref.serialize(steps)
    third = np.arange(-20, 20)[:, None] * 0.0005 + np.zeros((1, 9))
    assert oracle.serialize(third) == ref.serialize(third)


def test_quadtest_live_against_reference(oracle, ref):
    rng = np.random.default_rng(12)
    codes = set()
    for i in range(400):
        if i % 2:
            q = rng.uniform(-1, 1, (4, 2))
        else:
            a = rng.uniform(0, np.pi)
            w, d = rng.uniform(0.02, 1.0), rng.uniform(0.02, 0.5)
            base = np.array([[-w, -d], [w, -d], [-w, d], [w, d]]) * (1 + rng.normal(0, 0.1, (4, 2)))
            q = base @ np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
        lo, hi = q.min(0), q.max(0)
        pts = np.concatenate([rng.uniform(lo - 0.1, hi + 0.1, (300, 2)), q, np.nextafter(q, np.inf), np.nextafter(q, -np.inf)])
        rc_o, in_o = oracle.quad_test(q, pts)
        rc_r, in_r = ref.quad_test(q, pts)
        assert rc_o == rc_r
        codes.add(rc_o)
        if rc_o == 0:
            assert np.array_equal(in_o, in_r)
    assert 0 in codes and -1 in codes


def test_reference_cell_map_is_not_the_geometric_test(oracle, ref):
    """Quirk Q9 (found while building the GPU path): for some convex quadrilaterals the reference accepts
    (even at 5-15 degrees of yaw) QuadrilateralTest::isPointWithin (quadrilateralTest.cpp:275-451) returns
    true for points far outside the quadrilateral: a cell of its 3x3 map tests only the segments whose
    bounding boxes overlap the cell.  The build therefore reproduces the cell map (oracle and HIP kernels
    alike) instead of a half-plane test; this test pins that against the real reference code."""
    rng = np.random.default_rng(5)
    n_quads = n_disagree = 0
    for i in range(1500):
        a = rng.uniform(0.05, 0.5)
        w, d = rng.uniform(0.05, 0.6), rng.uniform(0.03, 0.4)
        base = np.array([[-w, -d], [w, -d], [-w, d], [w, d]]) * (1 + rng.normal(0, 0.1, (4, 2)))
        q = base @ np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
        lo, hi = q.min(0), q.max(0)
        pts = rng.uniform(lo - 0.02, hi + 0.02, (600, 2))
        rc, inside = oracle.quad_test(q, pts)
        rc_r, inside_r = ref.quad_test(q, pts)
        assert rc == rc_r
        if rc != 0:
            continue
        assert np.array_equal(inside, inside_r)
        n_quads += 1
        c = q.mean(0)
        dmin = np.full(len(pts), np.inf)
        for s, e in ((0, 1), (1, 3), (3, 2), (2, 0)):
            ax, ay = q[e][1] - q[s][1], q[s][0] - q[e][0]
            cc = q[e][0] * q[s][1] - q[s][0] * q[e][1]
            sg = 1.0 if ax * c[0] + ay * c[1] + cc >= 0 else -1.0
            dmin = np.minimum(dmin, sg * (ax * pts[:, 0] + ay * pts[:, 1] + cc) / np.hypot(ax, ay))
        clear = np.abs(dmin) > 1e-6
        if np.any((inside.astype(bool) != (dmin > 0)) & clear):
            n_disagree += 1
    assert n_quads > 1000 and n_disagree > 10


# ------------------------------------------------------------------ morphology (OpenCV absent: documented semantics)
def test_close3x3_against_scipy(oracle):
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(13)
    st = np.ones((3, 3), dtype=bool)
    for shape, density in (((37, 53), 0.3), ((64, 64), 0.05), ((5, 200), 0.5), ((48, 70), 0.9), ((3, 3), 0.5), ((1, 9), 0.5)):
        img = (rng.random(shape) < density)
        img[0, :] |= rng.random(shape[1]) < 0.5          # exercise the borders
        img[:, -1] |= rng.random(shape[0]) < 0.5
        want = ndi.binary_erosion(ndi.binary_dilation(img, st, border_value=0), st, border_value=1)
        got = oracle.close3x3(img.astype(np.uint8) * 255)
        assert np.array_equal(got, want.astype(np.uint8) * 255)


def test_close3x3_basic_properties(oracle):
    rng = np.random.default_rng(14)
    img = (rng.random((60, 80)) < 0.2).astype(np.uint8) * 255
    c = oracle.close3x3(img)
    assert np.all(c >= img)                      # closing is extensive
    assert np.array_equal(oracle.close3x3(c), c)  # and idempotent
    assert not oracle.close3x3(np.zeros((9, 9), np.uint8)).any()
    assert oracle.close3x3(np.full((9, 9), 255, np.uint8)).all()
    one = np.zeros((9, 9), np.uint8)
    one[4, 4] = 255
    assert np.array_equal(oracle.close3x3(one), one)
    gap = np.zeros((9, 9), np.uint8)
    gap[4, 2] = gap[4, 4] = 255                   # a one-pixel gap closes
    assert oracle.close3x3(gap)[4, 3] == 255


# ------------------------------------------------------------------ BestLine (segmentation.cpp:409-487)
def _best_line_bruteforce(pts, hyp):
    best = None
    for p, q in itertools.combinations(range(len(pts)), 2):
        (x1, y1), (x2, y2) = pts[p], pts[q]
        a, b, c = y2 - y1, x1 - x2, x2 * y1 - x1 * y2
        others = [abs(x * a + y * b + c) for i, (x, y) in enumerate(pts) if i not in (p, q)]
        if len(pts) <= 2:
            res = 0.0
        else:
            n = (len(others) - 1) // 2 if len(others) > 4 else 1
            res = sum(sorted(others)[:n]) / (n * hyp(float(a), float(b)))
        if best is None or res < best[0]:
            best = (res, (a, b, c))
    return best[1]


def test_best_line_against_bruteforce(oracle):
    rng = np.random.default_rng(15)
    for m in (2, 3, 4, 5, 6, 7, 9, 14, 21, 40):
        for trial in range(4):
            xs = 500 + 25 * np.arange(m) * (1 if trial % 2 else -1)
            ys = (300 + 0.1 * (xs - 500) + rng.integers(-3, 4, m)).astype(int)
            if trial == 2:
                ys[rng.integers(0, m)] += 40      # an outlier scan
            if trial == 3:
                ys[:] = 300                       # all collinear: every pair ties at residual 0 -> first pair wins
            pts = [(int(x), int(y)) for x, y in zip(xs, ys)]
            rc, line = oracle.best_line(pts)
            assert rc == 0
            assert tuple(line) == _best_line_bruteforce(pts, oracle.hypot)


# ------------------------------------------------------------------ hypot: oracle = libm, product = restated glibc algorithm
def test_sort_restatement_leaves_the_permutation_std_sort_leaves(ssd, oracle, tmp_path):
    """csrc/ssd_sort.h (libstdc++'s introsort restated, for the tie order of segmentation.cpp:724) against std::sort
    itself: random, tie-heavy, sorted, reversed and organ-pipe keys, and McIlroy's adversary (tests/golden/antiqsort.cpp)
    that drives std::sort into its heapsort fallback."""
    import subprocess
    rng = np.random.default_rng(0)
    for trial in range(1200):
        n = int(rng.integers(1, 400))
        kind = trial % 6
        if kind == 0:
            d = rng.uniform(0, 10, n)
        elif kind == 1:
            d = rng.integers(0, 4, n).astype(float)
        elif kind == 2:
            d = np.sort(rng.integers(0, 30, n)).astype(float)
        elif kind == 3:
            d = np.sort(rng.uniform(0, 1, n))[::-1].copy()
        elif kind == 4:
            d = np.round(rng.normal(5, 2, n), 1)
        else:
            d = np.abs(np.arange(n) - n // 2).astype(float)
        assert np.array_equal(oracle.sort_perm(d), ssd.sort_perm(d)), (trial, n)
    exe = str(tmp_path / "antiqsort")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(HERE, "golden", "antiqsort.cpp"), "-o", exe], check=True)
    for n in (40, 120, 300, 1000):
        killer = np.array(subprocess.run([exe, str(n)], check=True, capture_output=True, text=True).stdout.split(), dtype=np.float64)
        for q in (1, 2, 7, 50):
            d = np.floor(killer / q)
            assert np.array_equal(oracle.sort_perm(d), ssd.sort_perm(d)), (n, q)


def test_product_hypot_equals_libm_on_this_image(ssd, oracle):
    """The oracle calls the host's std::hypot as the reference does; the kernels restate the glibc 2.35
    algorithm (the same function, ssd_test_hypot_host, compiled for the host).  They must agree here."""
    rng = np.random.default_rng(16)
    L = ssd.hooks_lib()
    vals = []
    for a in range(-60, 61, 7):
        for b in range(-1100, 1101, 13):
            vals.append((float(a), float(b)))
    vals += [(0.0, 0.0), (0.0, 5.0), (3.0, 0.0), (3.0, 4.0), (1e-200, 1e-200), (1e200, 1e200), (1.0, 1e-17), (1e-300, 1.0)]
    vals += [tuple(v) for v in rng.standard_normal((20000, 2))]
    vals += [tuple(v) for v in rng.standard_normal((5000, 2)) * [1.0, 1e-3]]
    vals += [tuple(v) for v in rng.integers(-3000, 3000, (20000, 2)).astype(float)]
    bad = [(a, b) for a, b in vals if L.ssd_test_hypot_host(a, b) != oracle.hypot(a, b)]
    assert not bad, bad[:5]


def test_the_all_cores_runner_is_the_single_threaded_oracle_on_every_thread(ssd, oracle):
    """oracle/ssd_oracle_mt.cpp (bench.py's cpu_baseline_all_cores): pinned threads behind one start line, each on a private copy
    of a frame - the steps they find are those ssdo_process_lean finds on the same frames, thread for thread and repetition for
    repetition; more threads than frames wrap around."""
    import oracle_binding as ob
    import scenes
    W, H = 640, 480
    sc = scenes.batch_scenes(ssd, W, H, 3, base_seed=100000, rng_seed=7)
    trans = ssd.transformation_for_scene(sc[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=3)
    xs = list(ssd.synth_host(sc))
    ocfg, ocal = ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants)
    alone = [oracle.process_lean(ocfg, ocal, x)[0] for x in xs]
    assert all(n >= 0 for n in alone) and sum(alone) > 0
    cpus = sorted(os.sched_getaffinity(0))[:4] or [0]
    cpus = (cpus * 5)[:5]                                         # five threads on up to four CPUs, three frames: wraps
    many = oracle.process_many(ocfg, ocal, xs, cpus, reps=2)
    assert many["frames"] == 10 and many["steps"] == 2 * sum(alone[t % 3] for t in range(5))
    assert many["wall_s"] > 0 and many["frames_per_s"] > 0 and many["read_gb_per_s"] > 0
