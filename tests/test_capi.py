"""CPU tests of the C-ABI library: it loads, exports every symbol include/ssd_hip.h declares, its host-only
entry points (configuration, calibration, serialisation, synthetic frame source) behave like the
reference / oracle, and the compute entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_binding as ob
import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header_name):
    header = open(os.path.join(ROOT, "include", header_name)).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    return set(re.findall(r"\b(ssd_[a-z0-9_]+)\s*\(", header))


def test_library_exports_every_declared_symbol(ssd):
    """each of the three libraries exports exactly what its header declares — and the product library nothing of the
    frame source or the test hooks (they are not part of the product ABI)"""
    import subprocess
    for header, L, names, path in (("ssd_hip.h", ssd.lib(), ssd.EXPORTS, ssd.LIB_PATH),
                                   ("ssd_source.h", ssd.source_lib(), ssd.SOURCE_EXPORTS, ssd.SOURCE_LIB_PATH),
                                   ("ssd_testhooks.h", ssd.hooks_lib(), ssd.HOOK_EXPORTS, ssd.HOOKS_LIB_PATH)):
        declared = _declared(header)
        missing = [s for s in sorted(declared) if not hasattr(L, s)]
        assert not missing, (header, missing)
        assert set(names) == declared, (header, sorted(set(names) ^ declared))
        exported = set(re.findall(r" T (ssd_[a-z0-9_]+)", subprocess.run(["nm", "-D", "--defined-only", path], check=True,
                                                                          capture_output=True, text=True).stdout))
        assert exported == declared, (header, sorted(exported ^ declared))
    assert len(_declared("ssd_hip.h")) >= 25
    assert not [n for n in _declared("ssd_hip.h") if n.startswith("ssd_test_") or n.startswith("ssd_synth_")]


def test_header_is_plain_c_and_links(ssd, tmp_path):
    """include/ssd_hip.h must be consumable from C (cgo / JNI / ctypes-style bindings): a C99 translation unit that
    includes it, takes the address of every declared entry point and links against libssd_hip.so."""
    import subprocess
    names = sorted(_declared("ssd_hip.h") | _declared("ssd_source.h") | _declared("ssd_testhooks.h"))
    src = tmp_path / "use_header.c"
    src.write_text('#include "ssd_hip.h"\n#include "ssd_source.h"\n#include "ssd_testhooks.h"\n#include <stdio.h>\ntypedef void (*fn)(void);\nint main(void)\n{\n  const fn f[] = { %s };\n'
                   '  ssd_config cfg; ssd_frame_result r; ssd_frame_risers rr; (void)r; (void)rr;\n'
                   '  if(ssd_default_config(&cfg, 640, 480) != SSD_OK) return 1;\n'
                   '  printf("%%d %%d\\n", (int)(sizeof f / sizeof f[0]), cfg.width);\n  return 0;\n}\n'
                   % ", ".join("(fn)%s" % n for n in names))
    exe = tmp_path / "use_header"
    libdir = os.path.dirname(ssd.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-lssd_hip", "-lssd_source", "-lssd_testhooks", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == len(names) and out[1] == "640"


def test_struct_layouts_match_the_header(ssd):
    """ctypes mirrors must have the C sizes (computed from the header's constants)."""
    assert C.sizeof(ssd.Step) == 72
    assert C.sizeof(ssd.FrameResult) == 8 + 72 * ssd.MAX_STEPS
    assert C.sizeof(ssd.Calibration) == 19 * 8
    assert C.sizeof(ssd.Config) == 8 + 9 * 8 + 8 + 8          # 3 trailing int32 + padding to 8
    assert C.sizeof(ssd.Scene) % 8 == 0
    assert C.sizeof(ssd.Riser) == 8 + 7 * 8
    assert C.sizeof(ssd.FrameRisers) == 8 + C.sizeof(ssd.Riser) * (ssd.MAX_STEPS - 1)


def test_riser_entry_points_validate_their_arguments(ssd):
    L = ssd.lib()
    assert L.ssd_set_risers(None, 1, 0.03, 200) == -1            # SSD_E_ARG: null handle
    assert L.ssd_fetch_risers(None, None, 1, None) == -1


def test_default_config_is_the_reference_configuration(ssd):
    cfg = ssd.default_config(640, 480)   # configuration.h:27-52
    assert (cfg.x_min, cfg.x_max, cfg.y_min, cfg.y_max, cfg.z_min, cfg.z_max) == (-0.6, 0.6, 0.1, 1.3, -0.1, 1.1)
    assert (cfg.height_interval, cfg.min_height_above_ground, cfg.min_step_depth) == (0.01, 0.05, 0.1)


def test_compute_entry_points_fail_loudly_without_gpu(ssd):
    if ssd.device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = ssd.default_config(640, 480)
    with pytest.raises(ssd.SsdError, match="no HIP device"):
        ssd.Detector(cfg, ssd.GeometricTransformation())
    with pytest.raises(ssd.SsdError):
        ssd.Pointcloud(ssd.Window("w"), ssd.GeometricTransformation()).process(np.zeros((480, 640, 3), np.float32))
    with pytest.raises(ssd.SsdError, match="no HIP device"):
        ssd.PinnedArray((4, 4), np.float32)                       # ssd_host_alloc
    with pytest.raises(ssd.SsdError, match="no HIP device"):
        ssd.DeviceBuffer(1024)                                    # ssd_device_alloc
    with pytest.raises(ssd.SsdError, match="no HIP device"):
        ssd.device_info(0)                                        # ssd_device_info_get
    before = os.sched_getaffinity(0)
    with pytest.raises(ssd.SsdError, match="no HIP device"):
        ssd.bind_thread_to_device(0)                              # ... and nothing was bound
    assert os.sched_getaffinity(0) == before
    sc = ssd.make_scene(64, 48)
    with pytest.raises(ssd.SsdError, match="no HIP device"):
        ssd.synth_device([sc], 4096)                              # the frame source's device generator (libssd_source.so)
    # the reference's main over this build (lib/detect-stairs-ref) ends with an error, not with a made-up line
    import subprocess
    exe = os.path.join(os.path.dirname(ssd.LIB_PATH), "detect-stairs-ref")
    if os.path.exists(exe):
        p = subprocess.run([exe], capture_output=True, text=True, timeout=120, cwd=os.path.dirname(exe))
        assert p.returncode != 0 and "stairs" not in p.stdout


def test_create_rejects_bad_arguments(ssd):
    L = ssd.lib()
    cfg = ssd.default_config(640, 480)
    cal = ssd.GeometricTransformation().constants
    h = C.c_void_p()
    assert L.ssd_create(None, C.byref(cal), 0, C.byref(h)) == -1
    cfg.height_interval = 0.001           # 1201 bins > SSD_MAX_BINS
    assert L.ssd_create(C.byref(cfg), C.byref(cal), 0, C.byref(h)) == -1
    assert b"bins" in L.ssd_last_error()
    cfg = ssd.default_config(640, 480)
    cfg.max_step_plateaus = 99
    assert L.ssd_create(C.byref(cfg), C.byref(cal), 0, C.byref(h)) == -1
    # workspaces of the handle: 0 (automatic) .. 8
    for bad in (-1, 9):
        cfg = ssd.default_config(640, 480, batches_in_flight=bad)
        assert L.ssd_create(C.byref(cfg), C.byref(cal), 0, C.byref(h)) == -1 and b"batches_in_flight" in L.ssd_last_error()
    # a pixel key's row field and the one-subtraction window test of the rasterising kernels: height <= 8064
    assert L.ssd_create(C.byref(ssd.default_config(640, 8065)), C.byref(cal), 0, C.byref(h)) == -1 and b"8064" in L.ssd_last_error()
    assert L.ssd_create(C.byref(ssd.default_config(8193, 480)), C.byref(cal), 0, C.byref(h)) == -1
    assert ssd.default_config(640, 480).batches_in_flight == 0
    assert L.ssd_stream_wait(None, 0, None) == -1 and L.ssd_batches_in_flight(None) == 0


def test_calibration_matches_oracle_bitwise(ssd, oracle):
    rng = np.random.default_rng(21)
    for trial in range(20):
        sc = ssd.make_scene(640, 480, cam_height=float(rng.uniform(0.6, 1.6)), pitch_deg=float(rng.uniform(30, 70)),
                            roll_deg=float(rng.uniform(-5, 5)))
        marks = [(float(rng.uniform(-0.5, -0.1)), float(rng.uniform(0.7, 1.2)), 0.0),
                 (float(rng.uniform(0.1, 0.5)), float(rng.uniform(0.7, 1.2)), 0.0),
                 (float(rng.uniform(-0.3, 0.3)), float(rng.uniform(0.2, 0.5)), 0.0)]
        world, cam = ssd.calibration_points(sc, marks)
        t = ssd.GeometricTransformation(world, cam)
        rc, cal = oracle.calibration(world, cam)
        assert rc == 0
        assert bytes(t.constants) == bytes(cal)
        a = np.array(t.constants.a).reshape(3, 3)
        assert np.allclose(a @ a.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(a) - 1) < 1e-14
        # the three marks lie on the ground: their world z is 0
        z = (a @ cam.T).T[:, 2] + t.constants.b[2]
        assert np.all(np.abs(z) < 1e-12)
    with pytest.raises(ssd.SsdError):
        ssd.GeometricTransformation(np.zeros((3, 3)), np.zeros((3, 3)))


def test_reference_calibration_triangle_world_points(ssd, oracle):
    """The shipped calibration-triangle world points (reference calibration-triangle:2-4)."""
    world = np.array([[-1.121, 1.79826, 0.004], [1.121, 1.79826, 0.004], [0.769229, 0.3, 0.004]])
    sc = ssd.make_scene(640, 480, cam_height=1.2, pitch_deg=45.0)
    _, cam = ssd.calibration_points(sc, [tuple(w[:2]) + (0.0,) for w in world])
    t = ssd.GeometricTransformation(world, cam)
    rc, cal = oracle.calibration(world, cam)
    assert rc == 0 and bytes(t.constants) == bytes(cal)
    assert t.constants.world_z == 0.004
    # external world = camera-dependent world here (marks given in the same frame): R2 = I, t2 = 0
    assert np.allclose(np.array(t.constants.r2), [1, 0, 0, 1], atol=1e-12)
    assert np.allclose(np.array(t.constants.t2), [0, 0], atol=1e-12)


def test_serialize_cabi_matches_oracle_and_reference_goldens(ssd, oracle):
    import json
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_serialize.json")))
    for c in cases:
        n = c["n"]
        if n > ssd.MAX_STEPS:
            continue
        steps = np.array([float.fromhex(x) for x in c["steps"]]).reshape(n, 9)
        fr = ssd.FrameResult()
        fr.n_steps = n
        for i in range(n):
            fr.steps[i].height = steps[i][0]
            for k in range(8):
                fr.steps[i].quad[k] = steps[i][1 + k]
        assert ssd.Stairs(fr).serialize() == c["line"]
    fr = ssd.FrameResult()
    fr.status = ssd.ST_THROW
    assert ssd.Stairs(fr).serialize() == ""


def test_serialize_prints_every_digit_and_ignores_the_c_locale(ssd, oracle):
    """ostream << fixed << setprecision(3) prints all 309 integral digits of DBL_MAX and never a decimal comma; so must
    ssd_serialize (it used snprintf into 64 bytes).  Against the oracle's ostream; against the real stairs.cpp where
    oracle/_ref is present."""
    import locale
    vals = [0.17, -0.0001, 1e60, -1e300, 1.7976931348623157e308, float("inf"), float("-inf"), float("nan"), 0.0005, 0.0015, -0.0]
    ref = ob.load_ref()
    old = locale.setlocale(locale.LC_NUMERIC)
    for loc in (None, "de_DE.UTF-8", "de_DE", "fr_FR.UTF-8"):
        if loc is not None:
            try:
                locale.setlocale(locale.LC_NUMERIC, loc)
            except locale.Error:
                continue
        for v in vals:
            r = ssd.FrameResult()
            r.n_steps = 1
            r.steps[0].height = v
            r.steps[0].quad[0] = -v
            buf = C.create_string_buffer(1 << 13)
            n = ssd.lib().ssd_serialize(C.byref(r), buf, len(buf))
            steps = np.zeros((1, 9))
            steps[0, 0], steps[0, 1] = v, -v
            assert n == len(buf.value) and buf.value.decode() == oracle.serialize(steps), (loc, v)
            if ref is not None:
                assert buf.value.decode() == ref.serialize(steps), (loc, v)
            assert b"," + b"0" * 3 not in buf.value.split(b"[")[-1][:3]
    locale.setlocale(locale.LC_NUMERIC, old)
    small = C.create_string_buffer(64)
    r = ssd.FrameResult()
    r.n_steps = 1
    r.steps[0].height = 1e300
    assert ssd.lib().ssd_serialize(C.byref(r), small, len(small)) == -5          # SSD_E_CAP, never a truncated line


def test_create_rejects_ranges_the_fixed_point_mean_cannot_hold(ssd):
    """the mean height is a sum of round(z * 2^40) in int64 via a magic-constant add: |z| < 2048 m and
    max|z| * W * H < 2^23; such a configuration must be refused, not answered wrongly"""
    cal = ssd.GeometricTransformation().constants
    for zmin, zmax, w, h in ((-0.1, 2500.0, 640, 480), (-3000.0, 1.0, 640, 480), (-0.1, 10.0, 1920, 1080)):
        cfg = ssd.default_config(w, h)
        cfg.z_min, cfg.z_max, cfg.height_interval = zmin, zmax, (zmax - zmin) / 100.0
        hnd = C.c_void_p()
        assert ssd.lib().ssd_create(C.byref(cfg), C.byref(cal), 0, C.byref(hnd)) == -1, (zmin, zmax, w, h)
        assert b"2^23" in ssd.lib().ssd_last_error()


def _consumer_reads(line):
    """What the reference's consumers take out of one stdout line, restated: ros/stair_step_detector_pkg/.../stair_step_detector.py:33-56
    (ident, count, then per step height = step[0][1] and the corners step[1][1..4] as x, y) — print-stairs.py:55-71 indexes the
    same way.  -> (ident, count, [(height, [(x, y) x 4])])"""
    import json
    jdata = json.loads(line)
    ident, count = jdata[0], jdata[1][1]
    steps = []
    if count > 0:
        for step in jdata[2]:
            steps.append((step[0][1], [(step[1][k][0], step[1][k][1]) for k in (1, 2, 3, 4)]))
    return ident, count, steps


def test_wire_format_parses_like_the_ros_node(ssd, oracle):
    """ros/stair_step_detector_pkg/.../stair_step_detector.py:34-61 and print-stairs.py:55-71 index the line like this."""
    import json
    fr = ssd.FrameResult()
    fr.n_steps = 2
    for i in range(2):
        fr.steps[i].height = 0.17 * i
        for k in range(8):
            fr.steps[i].quad[k] = 0.1 * k - 0.35
    jdata = json.loads(ssd.Stairs(fr).serialize())
    assert jdata[0] == "stairs" and jdata[1][1] == 2
    assert jdata[2][1][0][1] == pytest.approx(0.17)
    assert [len(jdata[2][0][1][k]) for k in range(1, 5)] == [2, 2, 2, 2]
    ident, count, steps = _consumer_reads(ssd.Stairs(fr).serialize())
    assert (ident, count) == ("stairs", 2) and steps[1][0] == pytest.approx(0.17)
    assert steps[0][1] == [(pytest.approx(0.1 * (2 * k) - 0.35, abs=5e-4), pytest.approx(0.1 * (2 * k + 1) - 0.35, abs=5e-4)) for k in range(4)]
    fr.n_steps = 0
    assert json.loads(ssd.Stairs(fr).serialize()) == ["stairs", ["stairSteps", 0]]


def test_the_references_own_consumer_reads_the_same_numbers(ssd):
    """Reference pin of the wire format's CONSUMER side: tests/golden/ref_print_stairs.json holds
    what /root/reference/print-stairs.py — run unmodified as a subprocess in the build container (tests/golden/make_ref_goldens.py) —
    printed for the 69 golden lines.  print-stairs.py:53-77 draws every step (last one first) as its height and the corners
    quadri[3], quadri[4], quadri[1], quadri[2] at fixed terminal positions, each number as `6.3f`.  The hand-restated indexing
    above (_consumer_reads, the ROS node's) must pull exactly those numbers out of the same lines, and ssd_serialize must print
    those lines from the numbers (round trip)."""
    import json
    import re
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_print_stairs.json")))
    screens = fx["stdout"].split("\x1b[1;1H")[1:]
    assert len(screens) == len(fx["stdin_lines"]) == 69
    number = re.compile(r"\x1b\[(\d+);(\d+)H([ -]?\d+\.\d{3})(?![\d])")
    seen_steps = 0
    for line, screen in zip(fx["stdin_lines"], screens):
        ident, count, steps = _consumer_reads(line)
        head = re.match(r"(\w+), number of stair steps: (\d+)\n", screen)
        assert head and head.group(1) == ident == "stairs" and int(head.group(2)) == count
        drawn = number.findall(screen)
        assert len(drawn) == 9 * count
        top = 4
        for i in range(count - 1, -1, -1):                      # the terminal shows the highest step first
            height, quad = steps[i]
            got = {(int(r), int(c)): txt for r, c, txt in drawn[9 * (count - 1 - i):9 * (count - i)]}
            want = {(top + 2, 13): height, (top, 3): quad[2][0], (top + 1, 3): quad[2][1], (top, 23): quad[3][0], (top + 1, 23): quad[3][1],
                    (top + 3, 3): quad[0][0], (top + 4, 3): quad[0][1], (top + 3, 23): quad[1][0], (top + 4, 23): quad[1][1]}
            assert set(got) == set(want), (line, i)
            for pos, v in want.items():
                assert got[pos] == "%6.3f" % v, (line, i, pos)
            top += 7
            seen_steps += 1
        # and back: the numbers the consumer read, through ssd_serialize, give the line again
        fr = ssd.FrameResult()
        fr.n_steps = count
        for i, (height, quad) in enumerate(steps):
            fr.steps[i].height = height
            for k in range(4):
                fr.steps[i].quad[2 * k], fr.steps[i].quad[2 * k + 1] = quad[k]
        assert ssd.Stairs(fr).serialize() == line
    assert seen_steps > 200


def _same(a, b):
    """bitwise equality of doubles, NaNs of either sign / payload counted equal (a NaN's sign is not specified through arithmetic)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


def test_lines_are_the_references_line_coordinates(ssd, oracle):
    """Reference pin of types.h:117-163 (LineCoordinates<T>, the base of every line of the path: Line<int> / Line<double>,
    segmentation.cpp:321-407; StairsDetector::Line, pointcloud.cpp:514-526): the line through two points in int and in double and
    the determinants det / detx / dety, as the reference's own header computes them (compiled into oracle/_ref; golden
    tests/golden/ref_lines.json) against the oracle's lineThrough / LineT and the kernels' line_through_i / line_through_d /
    intersect60 (csrc/ssd_math.h, host build)."""
    import json
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_lines.json")))
    r = ob.load_ref()
    unhex = lambda xs: np.array([float.fromhex(x) for x in xs])
    for c in fx["double"]:
        pq, want = unhex(c["pq"]), unhex(c["abc"])
        assert _same(oracle.line(pq), want) and _same(ssd.line_host(pq)[0], want), c
        if r is not None:
            assert _same(r.line(pq), want)
    for c in fx["int"]:
        pq, want = np.array(c["pq"], dtype=np.int32), np.array(c["abc"], dtype=np.int32)
        assert np.array_equal(oracle.line(pq, integer=True), want) and np.array_equal(ssd.line_host(pq.astype(np.float64))[1], want), c
        if r is not None:
            assert np.array_equal(r.line(pq, integer=True), want)
    kTan60 = 1.7320508075688772
    met = 0
    for c in fx["dets"]:
        l, o, want = unhex(c["l"]), unhex(c["o"]), unhex(c["dets"])
        assert _same(oracle.line_dets(l, o), want), c
        if r is not None:
            assert _same(r.line_dets(l, o), want)
        # the kernels' intersection = detx / det, dety / det whenever |det| > |dot| tan 60 degrees (segmentation.cpp:350-362)
        found, x, y = ssd.intersect_host(l, o)
        with np.errstate(all="ignore"):
            dot = l[0] * o[0] + l[1] * o[1]
            expect = bool(abs(want[0]) > abs(dot) * kTan60)
            assert found == expect, c
            if found:
                assert _same([x, y], [want[1] / want[0], want[2] / want[0]]), c
                met += 1
    assert met > 100


def test_default_config_is_the_references_configuration(ssd, oracle):
    """Reference pin of configuration.h:27-52: ssd_default_config (and the oracle's) against stairs::Configuration{} as the
    reference's own header defines it — compiled into oracle/_ref (live, where that exists) and as the committed golden
    tests/golden/ref_configuration.json (made by tests/golden/make_ref_goldens.py from the same call)."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_configuration.json")))
    names = ["x_min", "x_max", "y_min", "y_max", "z_min", "z_max", "height_interval", "min_height_above_ground", "min_step_depth"]
    want = [float.fromhex(gold["values"][n]) for n in names]
    W, H = gold["depth_stream"]["width"], gold["depth_stream"]["height"]
    assert (W, H) == (640, 480)
    r = ob.load_ref()
    if r is not None:
        vals, wh = r.configuration()
        assert [float(v).hex() for v in vals] == [gold["values"][n] for n in names] and wh == (W, H)
    cfg = ssd.default_config(W, H)
    assert [getattr(cfg, n) for n in names] == want and (cfg.width, cfg.height) == (W, H)
    ocfg = oracle.config(W, H)
    assert [getattr(ocfg, n) for n in names] == want and (ocfg.width, ocfg.height) == (W, H)


def test_host_generator_is_deterministic_and_plausible(ssd):
    sc = scenes.make(ssd, "vga_3steps_noise2mm")
    a, b = ssd.synth_host([sc])[0], ssd.synth_host([sc])[0]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    sc.seed += 1
    assert not np.array_equal(a, ssd.synth_host([sc])[0])
    assert (a[..., 2] > 0).mean() > 0.95 and a[..., 2].max() < 9.0


def test_oracle_runs_the_config1_frame(ssd, oracle):
    """BASELINE.json config 1: one XGA frame, 3 steps, on the CPU path: ground + 3 treads at 0.17 m pitch."""
    sc = scenes.make(ssd, "xga_config1")
    t = ssd.transformation_for_scene(sc)
    cfg = ssd.default_config(1024, 768)
    xyz = ssd.synth_host([sc])[0]
    res, *_ = oracle.process(ob.to_oracle_config(cfg), ob.to_oracle_calibration(t.constants), xyz)
    assert res.n_steps == 4 and res.status == 0
    heights = [res.steps_ext[i][0] for i in range(4)]
    assert np.allclose(heights, [0.004, 0.174, 0.344, 0.514], atol=2e-3)
    # pixel-pitch quantisation of corners: 1.2 m / 1024 px (SURVEY.md appendix A observed -0.40078125 on this scene type)
    x0 = res.steps_ext[1][1]
    assert abs(x0 - (-0.6 + 170 * 1.2 / 1024)) < 2 * 1.2 / 1024
    line = res.line.decode()
    assert line.startswith('["stairs",["stairSteps",4],[[["height",0.004],')
    n, steps, status = oracle.process_lean(ob.to_oracle_config(cfg), ob.to_oracle_calibration(t.constants), xyz)
    assert n == 4 and np.array_equal(steps, np.array([list(res.steps_ext[i]) for i in range(4)]))


# ------------------------------------------------------------------ calibration files
REFERENCE_TRIANGLE = """calibration triangle
x1 = -1.121, y1 = 1.79826, z1 = 0.004
x2 = 1.121, y2 = 1.79826, z2 = 0.004
x3 = 0.769229, y3 = 0.3, z3 = 0.004
lowerQuadrant = right
"""          # the data file the reference ships (calibration-triangle:1-5)


def _write_calibration(tmp_path, ssd, triangle=REFERENCE_TRIANGLE, rows=10, header="calibration points", jitter=1e-3, seed=3):
    """Writes the two files as the reference's calibrate tool does (geometricCalibration.cpp:43-71:
    `fixed << setprecision(6) << setw(9) x << ", " << setw(9) y << ", " << setw(8) z`, sets joined by "; ")."""
    rng = np.random.default_rng(seed)
    world = np.array([[-1.121, 1.79826, 0.004], [1.121, 1.79826, 0.004], [0.769229, 0.3, 0.004]])
    sc = ssd.make_scene(640, 480, cam_height=1.3, pitch_deg=42.0)
    _, cam = ssd.calibration_points(sc, [tuple(w[:2]) + (0.0,) for w in world])
    (tmp_path / "calibration-triangle").write_text(triangle)
    lines = [header]
    sets = []
    for _ in range(rows):
        s = cam + rng.normal(0, jitter, cam.shape)
        sets.append(s)
        lines.append("; ".join("%9.6f, %9.6f, %8.6f" % tuple(p) for p in s))
    (tmp_path / "calibration-points").write_text("\n".join(lines) + "\n")
    return world, np.array(sets)


def test_calibration_loader_matches_oracle_and_reference_triangle(ssd, oracle, tmp_path):
    world, sets = _write_calibration(tmp_path, ssd)
    t, loaded = ssd.GeometricCalibration.load(str(tmp_path))
    assert loaded
    rc, w_o, c_o = oracle.calibration_load(str(tmp_path))
    assert rc == 0
    assert np.array_equal(t.world_points, w_o) and np.array_equal(t.camera_points, c_o)
    assert np.array_equal(t.world_points, world)
    # mean of the ten point sets as float32 values summed in double (geometricCalibration.cpp:127-141)
    want = np.float32(np.round(sets, 6)).astype(np.float64).sum(0) / 10.0
    assert np.allclose(t.camera_points, want, atol=1e-12)
    rc2, cal = oracle.calibration(w_o, c_o)
    assert rc2 == 0 and bytes(cal) == bytes(t.constants)
    import oracle_binding
    ref = oracle_binding.load_ref()
    if ref is not None:                       # the real CalibrationTriangle::load / isValid
        rc3, w_r, side = ref.load_triangle(str(tmp_path))
        assert rc3 == 0 and side == 2 and np.array_equal(w_r, t.world_points)
        # ... and the real CalibrationTriangle::save (calibrationTriangle.cpp:127-146): the file the reference itself writes
        # must load here to what the reference loads from it (its default stream precision drops digits: compare after the trip)
        import shutil
        out = tmp_path / "resaved"
        out.mkdir()
        assert ref.resave_triangle(str(tmp_path), str(out)) == 0
        shutil.copy(tmp_path / "calibration-points", out / "calibration-points")
        t2, loaded2 = ssd.GeometricCalibration.load(str(out))
        rc4, w_r2, side2 = ref.load_triangle(str(out))
        assert loaded2 and rc4 == 0 and side2 == 2 and np.array_equal(t2.world_points, w_r2)
        assert np.allclose(t2.world_points, t.world_points, atol=1e-5)
        assert oracle.calibration_load(str(out))[0] == 0 and np.array_equal(oracle.calibration_load(str(out))[1], w_r2)


def test_calibration_loader_falls_back_to_identity_like_the_reference(ssd, oracle, tmp_path):
    """geometricCalibration.cpp:199-202: missing / short / invalid files -> error message, identity transformation."""
    ident = bytes(ssd.GeometricTransformation().constants)
    t, loaded = ssd.GeometricCalibration.load(str(tmp_path))                 # no files at all
    assert not loaded and bytes(t.constants) == ident
    _write_calibration(tmp_path, ssd, rows=9)                                # fewer than 10 rows
    t, loaded = ssd.GeometricCalibration.load(str(tmp_path))
    assert not loaded and bytes(t.constants) == ident and oracle.calibration_load(str(tmp_path))[0] == -2
    _write_calibration(tmp_path, ssd, header="calibration pts")              # wrong header
    assert not ssd.GeometricCalibration.load(str(tmp_path))[1]
    _write_calibration(tmp_path, ssd, triangle=REFERENCE_TRIANGLE.replace("lowerQuadrant = right", "lowerQuadrant = up"))
    assert not ssd.GeometricCalibration.load(str(tmp_path))[1] and oracle.calibration_load(str(tmp_path))[0] == -1
    _write_calibration(tmp_path, ssd, triangle=REFERENCE_TRIANGLE.replace("x2 = 1.121", "x2 = -1.121"))   # coincident corners
    assert not ssd.GeometricCalibration.load(str(tmp_path))[1]
    import oracle_binding
    ref = oracle_binding.load_ref()
    if ref is not None:
        assert ref.load_triangle(str(tmp_path))[0] == 2                      # loaded but not valid
    _write_calibration(tmp_path, ssd)                                        # and a good pair loads again
    assert ssd.GeometricCalibration.load(str(tmp_path))[1]


# ------------------------------------------------------------------ 16-bit depth input
def test_deprojection_matches_oracle_bitwise(ssd, oracle):
    sc = scenes.make(ssd, "vga_3steps_noise2mm")
    intr = ssd.intrinsics_for_scene(sc)
    depth = ssd.synth_depth_host([sc])[0]
    assert depth.dtype == np.uint16 and (depth > 0).mean() > 0.95
    assert np.array_equal(depth, ssd.synth_depth_host([sc])[0])
    a, b = ssd.deproject_host(intr, depth), oracle.deproject(intr, depth)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # the documented formula on a few pixels, and zero depth -> the invalid point
    z = np.float32(depth[100, 200]) * np.float32(intr.depth_units)
    assert a[100, 200, 2] == z and a[100, 200, 0] == z * ((np.float32(200) - np.float32(intr.ppx)) / np.float32(intr.fx))
    d0 = depth.copy()
    d0[5, 7] = 0
    assert tuple(ssd.deproject_host(intr, d0)[5, 7]) == (0.0, 0.0, 0.0)
    # the depth frame is the float frame quantised to 0.25 mm: same staircase through the oracle
    t = ssd.transformation_for_scene(sc)
    cfg = ssd.default_config(640, 480)
    res, *_ = oracle.process(ob.to_oracle_config(cfg), ob.to_oracle_calibration(t.constants), a)
    assert res.n_steps == 4


def test_host_build_of_the_kernels_quadrilateral_test_against_the_reference_goldens(ssd):
    """PINNED, and without a GPU: csrc/ssd_quadtest.h (the register-resident builder k_quads runs, the constant cell, the
    evaluator k_inquad runs) is host + device code; compiled for the host (ssd_test_quad_host) it must reproduce the
    vectors the reference's own quadrilateralTest.cpp produced (tests/golden/ref_quadtest.json, generator
    make_ref_goldens.py) — every throw code and every point.  The -m gpu twin runs the device build of the same code."""
    import json
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_quadtest.json")))
    seen = set()
    for c in cases:
        quad = np.array([float.fromhex(x) for x in c["quad"]]).reshape(4, 2)
        pts = np.array([float.fromhex(x) for x in c["pts"]]).reshape(-1, 2)
        err, inside = ssd.quad_test_host(quad, pts)
        assert err == c["rc"], (err, c["rc"])
        seen.add(err)
        if err == 0:
            assert "".join(str(int(v)) for v in inside) == c["inside"]
    assert 0 in seen and -1 in seen


def test_host_build_of_the_quadrilateral_test_against_the_oracle_on_random_quadrilaterals(ssd, oracle):
    """The same host build against the oracle's restatement (itself checked against the real reference in test_oracle.py) on
    random treads: turned, sheared, with equal coordinates (merged cells) — throw code and every point."""
    rng = np.random.default_rng(977)
    codes = set()
    for it in range(400):
        yaw = np.radians(rng.uniform(-50, 50)) if it % 3 else 0.0
        w, d = rng.uniform(0.2, 1.0), rng.uniform(0.08, 0.4)
        base = np.array([[-w / 2, -d / 2], [w / 2, -d / 2], [-w / 2, d / 2], [w / 2, d / 2]])
        if it % 4 != 1:
            base += rng.normal(0, 0.03 if it % 4 else 0.15, base.shape)
        rot = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
        quad = base @ rot.T + [rng.uniform(-0.2, 0.2), rng.uniform(0.4, 1.0)]
        if it % 7 == 0:
            quad = np.round(quad * 8) / 8
        lo, hi = quad.min(axis=0) - 0.05, quad.max(axis=0) + 0.05
        pts = rng.uniform(lo, hi, (300, 2))
        pts[:8] = np.vstack([quad, quad + 1e-12])            # the corners themselves and just beside them
        err, inside = ssd.quad_test_host(quad, pts)
        want_err, want = oracle.quad_test(quad, pts)
        codes.add(err)
        assert err == want_err, (it, err, want_err)
        if err == 0:
            assert np.array_equal(inside.astype(bool), np.asarray(want).astype(bool)), it
    assert 0 in codes and len(codes) > 1


def test_host_build_of_the_kernels_closing_against_the_oracle(ssd, oracle):
    """csrc/ssd_closing.h — the 3x3 closing on bit images as k_outline / k_final compute it, word by word (closed_word: probe
    rows, debug capture) and by pixel column from the raw 5 x 5 neighbourhood (closed_scan_column: the scans) — is host + device
    code.  Its host build against the oracle's closing (checked against scipy.ndimage and the border behaviour of
    cv::morphologyEx in test_oracle.py): blobs, noise, lit borders, widths that are no multiple of 64 or 32, widths on both sides of 1024, every band
    height the kernels use; the first and last closed row of every 25th column must be those of the closed image."""
    rng = np.random.default_rng(4711)
    for case in range(64):
        # (above 1024 pixels the column scan fetches a row's five pixels with one 8-byte load and a 64-bit shift: ColumnStrip)
        w = int(rng.choice([64, 65, 97, 128, 200, 427, 600, 1025, 1057, 1100, 1920, 2049]))
        h = int(rng.integers(12, 90))
        img = np.zeros((h, w), np.uint8)
        for _ in range(int(rng.integers(1, 6))):                      # rectangles, some touching the borders
            x0, y0 = int(rng.integers(-5, w - 3)), int(rng.integers(-5, h - 3))
            x1, y1 = x0 + int(rng.integers(2, w // 2 + 3)), y0 + int(rng.integers(2, h // 2 + 3))
            img[max(y0, 0):max(y1, 0), max(x0, 0):max(x1, 0)] = 255
        img[rng.random((h, w)) < rng.choice([0.0, 0.02, 0.2, 0.5])] = 255    # salt
        img[rng.random((h, w)) < rng.choice([0.0, 0.05, 0.3])] = 0           # pepper: holes for the closing to fill (or not)
        if case % 5 == 0:
            img[0, :] = 255; img[:, 0] = 255
        if case % 7 == 0:
            img[-1, :] = 255; img[:, -1] = 255
        want = oracle.close3x3(img.copy())
        xs0 = int(rng.integers(0, 25))
        y_from = int(rng.integers(0, h // 2))
        band = int(rng.choice([1, 3, 7, 16, 64]))
        closed, first, last = ssd.closing_host(img, xs0, 25, y_from, band)
        assert np.array_equal(closed, want), "case %d: word-wise closing differs (%dx%d)" % (case, w, h)
        for j, x in enumerate(range(xs0, w, 25)):
            rows = np.nonzero(want[y_from:, x])[0] + y_from
            assert first[j] == (rows[0] if len(rows) else -1) and last[j] == (rows[-1] if len(rows) else -1), \
                "case %d: column x = %d (%dx%d, rows from %d, bands of %d)" % (case, x, w, h, y_from, band)


def test_host_build_of_the_kernels_best_line_against_the_oracle(ssd, oracle):
    """csrc/ssd_bestline.h — the residual of a pair's line in the three forms k_outline / k_final use (any list; distinct keys
    taken four per pass; ONE pass that keeps the n + 2 smallest distances sorted, equal distances included) — is host + device
    code.  BestLine over its host build against the oracle's (itself checked against a brute-force statement in
    test_oracle.py): scan-like lists of 2..64 points, outliers, collinear lists (every pair ties: the first must win), many
    equal distances (rows of equal y), duplicated points' distances."""
    rng = np.random.default_rng(20261)
    checked = 0
    for m in (2, 3, 4, 5, 6, 7, 8, 9, 11, 14, 21, 30, 42, 64):
        for trial in range(6):
            xs = 12 + 25 * np.arange(m)
            if trial % 2:
                xs = xs[::-1].copy()
            ys = (300 + 0.07 * (xs - 500) + rng.integers(-3, 4, m)).astype(int)
            if trial == 2:
                ys[rng.integers(0, m)] += 40                      # an outlier scan
            if trial == 3:
                ys[:] = 300                                       # collinear
            if trial == 4:
                ys = 300 + (np.arange(m) % 2) * 2                 # two rows: distances repeat massively
            if trial == 5:
                ys = rng.integers(0, 768, m)                      # no line at all
            pts = [(int(x), int(y)) for x, y in zip(xs, ys)]
            rc, want = oracle.best_line(pts)
            assert rc == 0
            for form in (0, 1, 2):
                got = ssd.best_line_host(pts, form)
                assert got == tuple(int(v) for v in want), "m = %d, trial %d, form %d: %s vs %s" % (m, trial, form, got, tuple(want))
                checked += 1
    assert checked == 14 * 6 * 3
