"""The C++ class surface (include/stairs/*.h -> libssd_hip.so, libssd_source.so) under the reference's OWN main.

north_star: "keeping the existing Pointcloud/Stairs C++ class surface so detect-stairs.cpp ... link against it
unchanged".  In the build container (where /root/reference exists) the reference's detect-stairs.cpp is compiled IN
PLACE, UNMODIFIED, against this build's headers of the same names and linked against its libraries; on the GPU box the
binary built here (lib/detect-stairs-ref, see __graft_entry__.build) is run and its stdout must be the oracle's lines.
The functors of GeometricTransformation are checked on the CPU against the same operation sequence in numpy."""
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MAIN = "/root/reference/detect-stairs.cpp"
INC = os.path.join(ROOT, "include", "stairs")
MARKS = np.array([[-0.35, 0.9, 0.004], [0.35, 0.9, 0.004], [0.2, 0.35, 0.004]])     # external-world marks (ssd.CALIBRATION_MARKS + z offset)


def _libdir(ssd):
    return os.path.dirname(ssd.LIB_PATH)


def _compile_reference_main(ssd, out):
    libdir = _libdir(ssd)
    # the file's bytes go in through stdin, unmodified: a quoted #include looks in the including file's own directory
    # first, which would find the reference's window.h (GLFW, librealsense) before this build's header of that name
    with open(REF_MAIN, "rb") as src:
        subprocess.run(["g++", "-x", "c++", "-std=c++20", "-O2", "-Wall", "-I", INC, "-", "-o", out, "-L", libdir, "-lssd_hip", "-lssd_source",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,$ORIGIN"], check=True, stdin=src)


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference tree exists only in the build container")
def test_reference_main_compiles_unmodified_against_this_build(ssd, tmp_path):
    """every name detect-stairs.cpp:21-45 uses resolves: Window, GeometricCalibration::load, GeometricTransformation,
    Pointcloud, Camera::start / waitForFrames, Frameset::depthFrame, Window::show / operator bool"""
    exe = str(tmp_path / "detect-stairs-ref")
    _compile_reference_main(ssd, exe)
    undefined = subprocess.run(["nm", "-D", "--undefined-only", exe], check=True, capture_output=True, text=True).stdout
    for sym in ("Pointcloud", "GeometricCalibration", "Camera", "Window"):
        assert sym in undefined, sym                 # resolved from this build's libraries, not inlined stubs
    # the binary kept next to the libraries (it travels to the GPU box) is this same translation unit
    shipped = os.path.join(_libdir(ssd), "detect-stairs-ref")
    assert os.path.exists(shipped), "__graft_entry__.build() did not build lib/detect-stairs-ref"


def test_transformation_functors_follow_the_reference_operation_order(ssd, tmp_path):
    """cameraToWorld / worldToCamera / toExternalWorld (transformation.h:79-100, .cpp:185-194) through a small C++
    program against the same IEEE operation sequence in numpy (no FMA on either side)"""
    sc = ssd.make_scene(640, 480, seed=3)
    world, cam = ssd.calibration_points(sc)
    src = tmp_path / "functors.cpp"
    src.write_text('#include "transformation.h"\n#include <cstdio>\nusing namespace stairs;\nint main()\n{\n'
                   '  const GeometricTransformation::RefPoints w{ Point3{%s}, Point3{%s}, Point3{%s} }, c{ Point3{%s}, Point3{%s}, Point3{%s} };\n'
                   '  const GeometricTransformation t(w, c);\n'
                   '  const Point3f v{ 0.125f, -0.37f, 1.21f };\n'
                   '  const Point3 a = t.cameraToWorld()(v), b = t.worldToCamera()(a), e = t.toExternalWorld()(a);\n'
                   '  std::printf("%%a %%a %%a %%a %%a %%a %%a %%a %%a\\n", a.x, a.y, a.z, b.x, b.y, b.z, e.x, e.y, e.z);\n  return 0;\n}\n'
                   % tuple(", ".join(repr(float(x)) for x in p) for p in list(world) + list(cam)))
    exe = tmp_path / "functors"
    libdir = _libdir(ssd)
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", INC, str(src), "-o", str(exe), "-L", libdir, "-lssd_hip", "-Wl,-rpath," + libdir], check=True)
    got = [float.fromhex(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    k = ssd.GeometricTransformation(world, cam).constants
    A, b, R, t2 = np.array(k.a).reshape(3, 3), np.array(k.b), np.array(k.r2).reshape(2, 2), np.array(k.t2)
    v = np.array([np.float32(0.125), np.float32(-0.37), np.float32(1.21)], dtype=np.float64)
    w = np.array([(A[r, 0] * v[0] + A[r, 1] * v[1]) + A[r, 2] * v[2] for r in range(3)]) + b
    d = w - b
    c = np.array([(A[0, r] * d[0] + A[1, r] * d[1]) + A[2, r] * d[2] for r in range(3)])
    e = np.array([R[0, 0] * w[0] + R[0, 1] * w[1], R[1, 0] * w[0] + R[1, 1] * w[1]]) + t2
    want = list(w) + list(c) + [e[0], e[1], k.world_z + w[2]]
    assert [x.hex() for x in got] == [float(x).hex() for x in want]
    assert np.allclose(c, v, atol=1e-12)                                   # and the round trip returns the camera point


@pytest.mark.gpu
def test_reference_main_prints_the_oracles_lines(ssd, oracle, gpu_device, tmp_path):
    """lib/detect-stairs-ref = /root/reference/detect-stairs.cpp, unmodified, over this build: run it on three synthetic
    frames with the calibration files GeometricCalibration::load() reads; stdout must be the oracle's lines"""
    exe = os.path.join(_libdir(ssd), "detect-stairs-ref")
    if os.path.exists(REF_MAIN):
        exe = str(tmp_path / "detect-stairs-ref")
        _compile_reference_main(ssd, exe)
    assert os.path.exists(exe), "lib/detect-stairs-ref missing: it is built by __graft_entry__.build() where /root/reference exists"
    W, H, n, seed = 640, 480, 3, 4242
    import ctypes as C
    sc0 = ssd.Scene()
    S = ssd.source_lib()
    assert S.ssd_source_default_scene(C.byref(sc0), W, H, 3, seed) == 0
    marks = (C.c_double * 9)(*MARKS.reshape(9))
    assert S.ssd_source_write_calibration(C.byref(sc0), marks, str(tmp_path).encode()) == 0
    env = dict(os.environ, SSD_SOURCE_WIDTH=str(W), SSD_SOURCE_HEIGHT=str(H), SSD_SOURCE_FRAMES=str(n), SSD_SOURCE_STEPS="3",
               SSD_SOURCE_SEED=str(seed))
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env).stdout.strip().splitlines()
    assert len(out) == n
    rc, wpts, cpts = oracle.calibration_load(str(tmp_path))                 # the oracle reads the same two files
    assert rc == 0
    rc, ocal = oracle.calibration(wpts, cpts)
    assert rc == 0
    ocfg = oracle.config(W, H)
    for f in range(n):
        sc = ssd.Scene()
        S.ssd_source_default_scene(C.byref(sc), W, H, 3, seed + f)
        res, *_ = oracle.process(ocfg, ocal, ssd.synth_host([sc])[0])
        assert res.n_steps == 4
        assert out[f] == res.line.decode(), "frame %d" % f
